"""provenance-generate -> provenance-publish and the CmdStan CSV route with the statistics on the GPU (SURVEY 8(f)
N2 / N3), held to what the imported reference produced for the same inputs (tests/golden/pipeline_cases.json):
the per-recipe errors{} / counters / exit codes of src/mcmc_ref/generate.py:46-103 + cli.py:248-291 and the
meta.json diagnostics convert_file wrote for the reference's own fake runner."""
from __future__ import annotations

import json

import numpy as np
import pytest

from conftest import GOLDEN, load_json
from test_pipeline_cpu import make_scaffold

pytestmark = pytest.mark.gpu
CASES = load_json("pipeline_cases.json")["generate"]


def _fail_blr(**kw):
    from mcmc_ref_hip import generate
    if kw["model_name"] == "blr":
        raise RuntimeError("sampler exploded")
    generate.fake_jsonzip_runner(**kw)


def test_generate_reference_corpus_matches_the_reference(tmp_path):
    from mcmc_ref_hip import generate
    scaffold = make_scaffold(tmp_path / "scaffold", files_for=[m for m in CASES["scaffold_models"] if m != "earn_height"])
    cfg = generate.GenerationConfig(chains=4, iter_sampling=300)
    # forced: meta.json carries the hot path's output for the reference's fake payload
    res = generate.generate_reference_corpus(scaffold_root=scaffold, output_root=tmp_path / "g1", models=["dugongs"],
                                             config=cfg, force=True, runner=generate.fake_jsonzip_runner)
    exp = CASES["forced"]
    assert (res.generated, res.failed, res.errors) == (exp["generated"], exp["failed"], exp["errors"])
    assert (tmp_path / "g1" / "archives" / "dugongs.json.zip").exists()
    assert (tmp_path / "g1" / "draws" / "dugongs.draws.parquet").exists()
    text = (tmp_path / "g1" / "meta" / "dugongs.meta.json").read_text()
    meta = json.loads(text)
    assert text == json.dumps(meta, indent=2, sort_keys=True)
    meta.pop("generated_date")
    diag, gold = meta.pop("diagnostics"), dict(exp["meta"])
    gdiag = gold.pop("diagnostics")
    assert meta == gold                                             # model, parameters, counts, checks, source
    for p in gdiag:
        for k in ("rhat", "ess_bulk", "ess_tail"):
            assert diag[p][k] == pytest.approx(gdiag[p][k], rel=1e-9), (p, k)
    # the quality gate of an unforced run lands in errors{}, nothing is written
    res = generate.generate_reference_corpus(scaffold_root=scaffold, output_root=tmp_path / "g2", models=["dugongs"],
                                             config=cfg, force=False, runner=generate.fake_jsonzip_runner)
    assert (res.generated, res.failed, res.errors) == (0, 1, CASES["gated"]["errors"])
    assert (tmp_path / "g2" / "draws" / "dugongs.draws.parquet").exists() is CASES["gated"]["wrote_draws"]
    # failing runner + missing scaffold file + a good recipe in one run
    res = generate.generate_reference_corpus(scaffold_root=scaffold, output_root=tmp_path / "g3",
                                             models=["blr", "earn_height", "dugongs"], config=cfg, force=True, runner=_fail_blr)
    assert (res.generated, res.failed, res.errors) == (CASES["mixed"]["generated"], CASES["mixed"]["failed"], CASES["mixed"]["errors"])
    # too few chains
    res = generate.generate_reference_corpus(scaffold_root=scaffold, output_root=tmp_path / "g5", models=["dugongs"],
                                             config=generate.GenerationConfig(chains=1, iter_sampling=50), force=False,
                                             runner=generate.fake_jsonzip_runner)
    assert (res.generated, res.failed, res.errors) == (0, 1, CASES["single_chain"]["errors"])


def test_many_recipes_one_batch_and_a_poisoned_one(tmp_path, oracle):
    """All recipes of a scaffold in one call: every model goes through the rolling window, a recipe whose archive holds
    a NaN draw fails alone (the reference's per-recipe isolation, generate.py:77-96), the others equal the oracle."""
    import zipfile
    from mcmc_ref_hip import generate
    names = [f"m{i:02d}" for i in range(13)]
    scaffold = make_scaffold(tmp_path / "scaffold", models=names)
    rng = np.random.default_rng(7)
    payloads = {}

    def runner(*, model_name, archive_path, config, **kw):
        k = names.index(model_name)
        C, N, P = 4 + k % 3, 150 + 37 * k, 1 + k % 4
        x = rng.normal(size=(C, P, N)).cumsum(axis=2) * 0.1 + rng.normal(size=(C, P, N))
        if model_name == "m05":
            x[1, 0, 17] = np.nan
        payloads[model_name] = x
        chains = [{f"p{j}": [float(v) for v in x[c, j]] for j in range(P)} for c in range(C)]
        with zipfile.ZipFile(archive_path, "w") as zf:
            zf.writestr(f"{model_name}.json", json.dumps(chains).replace("NaN", "NaN"))
    res = generate.generate_reference_corpus(scaffold_root=scaffold, output_root=tmp_path / "out", runner=runner, force=True)
    assert res.generated == 12 and res.failed == 1 and list(res.errors) == ["m05"] and "non-finite" in res.errors["m05"]
    for name in names:
        if name == "m05":
            assert not (tmp_path / "out" / "meta" / f"{name}.meta.json").exists()
            continue
        meta = json.loads((tmp_path / "out" / "meta" / f"{name}.meta.json").read_text())
        x = payloads[name]
        exp = oracle.summarize(np.ascontiguousarray(x.transpose(1, 0, 2)), "pcn", min_chains=1)
        for j in range(x.shape[1]):
            for k in ("rhat", "ess_bulk", "ess_tail"):
                assert meta["diagnostics"][f"p{j}"][k] == pytest.approx(float(exp[k][j]), rel=1e-9, nan_ok=True), (name, j, k)
        assert meta["n_chains"] == x.shape[0] and meta["n_draws_per_chain"] == x.shape[2]


def test_cli_provenance_generate_and_publish(tmp_path):
    """The reference's tests/unit/test_cli.py:127-169 flow (scaffold given on disk): exit codes and echo strings."""
    from click.testing import CliRunner
    from mcmc_ref_hip.cli import main
    scaffold = make_scaffold(tmp_path / "scaffold", files_for=["dugongs", "blr"])
    r = CliRunner()
    out = r.invoke(main, ["provenance-generate", "--scaffold-root", str(scaffold), "--output-root", str(tmp_path / "gen"),
                          "--models", "dugongs", "--fake-runner", "--force"])
    assert out.exit_code == 0, out.output
    assert out.output.strip() == f"generated=1 failed=0 output={tmp_path / 'gen'}"
    assert (tmp_path / "gen" / "draws" / "dugongs.draws.parquet").exists()
    meta = json.loads((tmp_path / "gen" / "meta" / "dugongs.meta.json").read_text())
    assert meta["n_chains"] == 10 and meta["n_draws_per_chain"] == 10000 and meta["checks"]["nchains_is_gte_4"] is True
    # without --force the fake draws fail the gate: exit 1 and one line per failing recipe, sorted
    out = r.invoke(main, ["provenance-generate", "--scaffold-root", str(scaffold), "--output-root", str(tmp_path / "gen2"),
                          "--models", "earn_height,dugongs", "--fake-runner"])
    assert out.exit_code == 1
    lines = out.output.strip().splitlines()
    assert lines[0] == f"generated=0 failed=2 output={tmp_path / 'gen2'}"
    assert lines[1].startswith("- dugongs: quality checks failed: ") and lines[2] == "- earn_height: missing scaffold files"
    out = r.invoke(main, ["provenance-publish", "--source-root", str(tmp_path / "gen"), "--scaffold-root", str(scaffold),
                          "--package-root", str(tmp_path / "pkg")])
    assert out.exit_code == 0, out.output
    assert out.output.strip() == f"published draws=1 meta=1 pairs=2 to={tmp_path / 'pkg'}"
    assert (tmp_path / "pkg" / "draws" / "dugongs.draws.parquet").exists()
    assert (tmp_path / "pkg" / "pairs" / "neals_funnel" / "pair.json").exists()
    # the published package is a store the reference API reads (store.py:102-119 layout)
    from mcmc_ref_hip import reference
    from mcmc_ref_hip.store import DataStore
    st = DataStore(local_root=tmp_path / "none", packaged_root=tmp_path / "pkg")
    assert reference.list_models(st) == ["dugongs"]
    assert reference.diagnostics_for_model("dugongs", store=st) == meta["diagnostics"]


def test_cmdstan_csv_chains_through_the_kernels(oracle):
    from mcmc_ref_hip import _ffi
    from mcmc_ref_hip.cmdstan_generate import chains_tensor
    files = [GOLDEN / "cmdstan" / "chain_1.csv", GOLDEN / "cmdstan" / "chain_2.csv"]
    names, x = chains_tensor(files + files[::-1])            # 4 chains of 40 draws
    with _ffi.Context(0) as ctx:
        got = ctx.summarize(x, "pcn")
    exp = oracle.summarize(x, "pcn")
    assert np.array_equal(got["lag_bulk"], exp["lag_bulk"]) and np.array_equal(got["q"], exp["q"])
    for k in ("mean", "std", "rhat", "ess_bulk", "ess_tail"):
        assert np.allclose(got[k], exp[k], rtol=1e-9, atol=0, equal_nan=True), k
