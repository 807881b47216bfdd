"""GPU tests of the callers either side of the path: DataStore layout, the
reference.* API, convert_file and the CLI, all running their statistics through the HIP kernels.
They restate the assertions of the reference's tests/unit/test_reference.py, test_convert.py and
test_cli.py against this package."""
from __future__ import annotations

import json
import math
import zipfile

import numpy as np
import pytest

from conftest import load_model

pytestmark = pytest.mark.gpu


def _write_model(root, name, draws, params, meta=None):
    import pyarrow as pa
    import pyarrow.parquet as pq
    P, C, N = draws.shape
    (root / "draws").mkdir(parents=True, exist_ok=True)
    (root / "meta").mkdir(parents=True, exist_ok=True)
    cols = {"chain": np.repeat(np.arange(C), N), "draw": np.tile(np.arange(N), C)}
    for i, p in enumerate(params):
        cols[p] = draws[i].reshape(-1)
    pq.write_table(pa.table(cols), root / "draws" / f"{name}.draws.parquet")
    if meta is not None:
        (root / "meta" / f"{name}.meta.json").write_text(json.dumps(meta))


@pytest.fixture()
def store(tmp_path):
    from mcmc_ref_hip.store import DataStore
    draws, params, rec = load_model("eight_schools-eight_schools_noncentered")
    meta = {"model": "eight", "parameters": params, "n_chains": 10, "n_draws_per_chain": 1000,
            "diagnostics": rec["meta_diagnostics"], "checks": rec["meta_checks"]}
    _write_model(tmp_path / "pkg", "eight", draws, params, meta)
    _write_model(tmp_path / "pkg", "eight_nometa", draws, params, None)
    d2, p2, _ = load_model("radon_pooled")
    _write_model(tmp_path / "local", "radon", d2, p2, {"model": "radon", "diagnostics": {}})
    return DataStore(local_root=tmp_path / "local", packaged_root=tmp_path / "pkg"), rec, params, draws


def test_reference_stats_and_diagnostics(store):
    from mcmc_ref_hip import reference
    st, rec, params, draws = store
    assert reference.list_models(st) == ["eight", "eight_nometa", "radon"]
    stats = reference.stats("eight", params=["mu", "tau"], store=st)
    assert set(stats) == {"mu", "tau"} and set(stats["mu"]) == {"mean", "std", "q5", "q50", "q95"}
    for p in ("mu", "tau"):
        for k, v in rec["stats"]["numpy"][p].items():
            assert stats[p][k] == pytest.approx(v, rel=1e-12)
    # cached diagnostics are returned verbatim (reference.py:82-90)
    assert reference.diagnostics_for_model("eight", store=st) == rec["meta_diagnostics"]
    assert list(reference.diagnostics_for_model("eight", params=["tau"], store=st)) == ["tau"]
    # no cache -> computed on the GPU, equals the reference's packaged goldens
    diag = reference.diagnostics_for_model("eight_nometa", store=st)
    assert list(diag) == params
    for p in params:
        for k in ("rhat", "ess_bulk", "ess_tail"):
            assert diag[p][k] == pytest.approx(rec["meta_diagnostics"][p][k], rel=1e-6)
    # empty diagnostics dict in meta -> computed as well (radon: 4 x 2500)
    d2 = reference.diagnostics_for_model("radon", store=st)
    _, p2, r2 = load_model("radon_pooled")
    for p in p2:
        assert d2[p]["ess_bulk"] == pytest.approx(r2["meta_diagnostics"][p]["ess_bulk"], rel=1e-6)
    summ = reference.summary_for_model("eight_nometa", store=st)
    assert summ["mu"]["rhat"] == pytest.approx(rec["meta_diagnostics"]["mu"]["rhat"], rel=1e-6)
    assert summ["mu"]["q50"] == rec["stats"]["numpy"]["mu"]["q50"]
    with pytest.raises(FileNotFoundError):
        reference.stats("nope", store=st)
    with pytest.raises(ValueError, match="Unknown backend: bogus"):
        reference.stats("eight", backend="bogus", store=st)
    # the reference's own backends stay selectable by name (reference.py:33,112 default to "arrow") and give exactly
    # what the reference returned for this model
    for b in ("arrow", "numpy"):
        sb = reference.stats("eight", params=["mu", "tau"], backend=b, store=st)
        assert sb == {p: rec["stats"][b][p] for p in ("mu", "tau")}, b
    assert reference.compare("eight", {"mu": list(draws[params.index("mu")].reshape(-1))}, backend="arrow", store=st).passed


def test_reference_draws_and_compare(store):
    from mcmc_ref_hip import reference
    st, rec, params, draws = store
    arr = reference.draws("eight", params=["mu", "tau"], chains=[0, 1, 2, 3], return_="numpy", store=st)
    assert arr.shape == (4000, 2)
    assert np.array_equal(arr[:, 0], draws[params.index("mu"), :4].reshape(-1))
    d = reference.draws("eight", return_="draws", store=st)
    assert d.params == params
    with pytest.raises(ValueError, match="Unknown return type"):
        reference.draws("eight", return_="bogus", store=st)
    actual = {"mu": list(draws[params.index("mu")].reshape(-1)[::3]), "tau": list(draws[params.index("tau")].reshape(-1)[::3])}
    res = reference.compare("eight", actual=actual, store=st)
    assert res.passed is True and res.failures == []
    bad = {"mu": [v + 100.0 for v in actual["mu"]]}
    res = reference.compare("eight", actual=bad, store=st)
    assert res.passed is False and res.failures[0].startswith("mu.mean rel_error=")


def test_convert_file_csv_and_jsonzip(tmp_path):
    from mcmc_ref_hip.convert import convert_file
    rng = np.random.default_rng(0)
    # single chain CSV: rejected by default, allowed with force and rhat is NaN (reference test_convert.py:28-62)
    csv = tmp_path / "one.csv"
    csv.write_text("chain,draw,mu\n" + "\n".join(f"0,{i},{rng.normal():.17g}" for i in range(50)))
    out = tmp_path / "out"
    (out / "d").mkdir(parents=True); (out / "m").mkdir()
    with pytest.raises(ValueError, match="at least 4 chains"):
        convert_file(csv, "one", out / "d", out / "m")
    res = convert_file(csv, "one", out / "d", out / "m", force=True)
    assert res.draws_path.exists() and res.meta_path.exists()
    assert math.isnan(res.meta["diagnostics"]["mu"]["rhat"])
    assert res.meta["n_chains"] == 1 and res.meta["n_draws_per_chain"] == 50
    assert res.meta["checks"]["nchains_is_gte_4"] is False
    on_disk = json.loads(res.meta_path.read_text())
    assert set(on_disk) == {"model", "parameters", "n_chains", "n_draws_per_chain", "diagnostics", "generated_date",
                            "checks", "source"}
    # CSV without chain/draw columns: chain 0, draw = row number
    csv2 = tmp_path / "bare.csv"
    csv2.write_text("a,b\n" + "\n".join(f"{rng.normal():.17g},{rng.normal():.17g}" for _ in range(20)))
    res2 = convert_file(csv2, "bare", out / "d", out / "m", force=True)
    assert res2.meta["parameters"] == ["a", "b"]
    # chain-list json-zip, 4 chains x 2500 = 10000 draws: passes every quality check
    chains = [{"x": list(rng.normal(size=2500)), "y": list(rng.normal(size=2500))} for _ in range(4)]
    jz = tmp_path / "m.json.zip"
    with zipfile.ZipFile(jz, "w") as zf:
        zf.writestr("m.json", json.dumps(chains))
    res3 = convert_file(jz, "m", out / "d", out / "m", source="unit")
    assert res3.meta["checks"] == {"ndraws_is_10k": True, "nchains_is_gte_4": True, "ess_above_400": True,
                                   "rhat_below_1_01": True}
    assert res3.meta["source"] == "unit" and res3.meta["parameters"] == ["x", "y"]
    # too few draws: the quality gate fails with the reference's message
    small = [{"x": list(rng.normal(size=100))} for _ in range(4)]
    jz2 = tmp_path / "s.json.zip"
    with zipfile.ZipFile(jz2, "w") as zf:
        zf.writestr("s.json", json.dumps(small))
    with pytest.raises(ValueError, match="quality checks failed: ndraws_is_10k"):
        convert_file(jz2, "s", out / "d", out / "m")
    with pytest.raises(ValueError, match="Unsupported input format"):
        convert_file(tmp_path / "x.txt", "x", out / "d", out / "m")


def test_cli(tmp_path, monkeypatch):
    from click.testing import CliRunner
    from mcmc_ref_hip.cli import main
    draws, params, rec = load_model("gp_pois_regr-gp_regr")
    root = tmp_path / "root"
    _write_model(root, "gp", draws, params, None)
    monkeypatch.setenv("MCMC_REF_LOCAL_ROOT", str(root))
    import mcmc_ref_hip.store as store_mod
    monkeypatch.setattr(store_mod, "default_packaged_root", lambda: None)
    r = CliRunner()
    out = r.invoke(main, ["list"])
    assert out.exit_code == 0 and out.output.split() == ["gp"]
    out = r.invoke(main, ["stats", "gp", "--format", "json", "--include-diagnostics"])
    assert out.exit_code == 0, out.output
    js = json.loads(out.output)
    assert "rhat" in js[params[0]] and js[params[0]]["mean"] == pytest.approx(rec["stats"]["numpy"][params[0]]["mean"])
    out = r.invoke(main, ["diagnostics", "gp", "--format", "csv"])
    assert out.output.splitlines()[0] == "param,rhat,ess_bulk,ess_tail" and len(out.output.splitlines()) == 1 + len(params)
    actual = tmp_path / "actual.csv"
    cols = {p: draws[i].reshape(-1)[::7] for i, p in enumerate(params)}
    actual.write_text(",".join(params) + "\n" + "\n".join(",".join(f"{cols[p][j]:.17g}" for p in params)
                                                           for j in range(len(cols[params[0]]))))
    out = r.invoke(main, ["compare", "gp", "--actual", str(actual)])
    assert out.exit_code == 0 and out.output.startswith("passed")
    actual.write_text(params[0] + "\n" + "\n".join("1e6" for _ in range(10)))
    out = r.invoke(main, ["compare", "gp", "--actual", str(actual), "--format", "json"])
    assert out.exit_code == 2 and json.loads(out.output)["passed"] is False
    csv = tmp_path / "c.csv"
    csv.write_text("chain,draw,mu\n" + "\n".join(f"{c},{i},{(c * 31 + i * 17) % 97 / 97.0}" for c in range(4) for i in range(60)))
    out = r.invoke(main, ["convert", str(csv), "--name", "conv", "--force"])
    assert out.exit_code == 0 and "converted conv" in out.output
    assert (root / "draws" / "conv.draws.parquet").exists() and (root / "meta" / "conv.meta.json").exists()
