"""Host logic of the callers around the hot path against what the imported reference returned
(tests/golden/pipeline_cases.json, made by tests/golden/make_golden_pipeline.py): the CmdStan CSV reader and its
name normalisation (src/mcmc_ref/cmdstan_generate.py:13-41), the chain-list JSON-zip reader (convert.py:78-102),
recipe selection / error bookkeeping of provenance-generate before any kernel runs (generate.py:46-103), and
provenance-publish (generate.py:106-174).  No GPU."""
from __future__ import annotations

import json
import zipfile

import numpy as np
import pytest

from conftest import GOLDEN, load_json

CASES = load_json("pipeline_cases.json")


def make_scaffold(root, models=None, files_for=None, pairs=("eight_schools", "neals_funnel")):
    """A scaffold directory in the layout provenance.materialize_scaffold writes (provenance.py:101-140)."""
    models = list(models if models is not None else CASES["generate"]["scaffold_models"])
    (root / "stan_models").mkdir(parents=True)
    (root / "stan_data").mkdir()
    for m in (files_for if files_for is not None else models):
        (root / "stan_models" / f"{m}.stan").write_text("parameters { real mu; } model { mu ~ normal(0, 1); }\n")
        (root / "stan_data" / f"{m}.json").write_text("{}\n")
    for p in pairs:
        (root / "pairs" / p / "centered").mkdir(parents=True)
        (root / "pairs" / p / "pair.json").write_text(json.dumps({"name": p}) + "\n")
        (root / "pairs" / p / "centered" / "model.stan").write_text("// stan\n")
    (root / "provenance_manifest.json").write_text(json.dumps(
        {"schema_version": 1, "generator": {"name": "mcmc-ref"}, "cmdstan": {"chains": 10}, "models": models,
         "pairs": list(pairs), "files": {}}, indent=2, sort_keys=True) + "\n")
    return root


def test_cmdstan_csv_reader_matches_the_reference():
    from mcmc_ref_hip import cmdstan_generate as cs
    for fname, exp in CASES["cmdstan"]["files"].items():
        got = cs.parse_cmdstan_csv(GOLDEN / "cmdstan" / fname)
        assert set(got) == set(exp)
        assert not any(k.endswith("__") for k in got)
        for k in exp:
            assert got[k] == exp[k], (fname, k)             # float(text) on both sides: bit-equal
        names, x = cs.read_cmdstan_csv(GOLDEN / "cmdstan" / fname)
        assert x.shape == (len(exp), 40) and x.dtype == np.float64
    for raw, norm in CASES["cmdstan"]["normalize"].items():
        assert cs._normalize_cmdstan_param_name(raw) == norm, raw
    names, x = cs.chains_tensor([GOLDEN / "cmdstan" / "chain_1.csv", GOLDEN / "cmdstan" / "chain_2.csv"])
    assert x.shape == (11, 2, 40)
    for c, fname in enumerate(("chain_1.csv", "chain_2.csv")):
        for i, n in enumerate(names):
            assert x[i, c].tolist() == CASES["cmdstan"]["files"][fname][n]


def test_cmdstan_payload_validation_messages(tmp_path):
    from mcmc_ref_hip import cmdstan_generate as cs
    payloads = {"empty": [], "no_params": [{}], "key_mismatch": [{"a": [1.0]}, {"b": [1.0]}],
                "ragged": [{"a": [1.0], "b": [1.0, 2.0]}]}
    for label, msg in CASES["cmdstan"]["payload_errors"].items():
        with pytest.raises(ValueError) as e:
            cs.build_posteriordb_payload(payloads[label])
        assert str(e.value) == msg
    ok = [{"a": [1.0, 2.0]}, {"a": [3.0, 4.0]}]
    assert cs.build_posteriordb_payload(ok) is ok
    z = cs.write_posteriordb_json_zip(ok, tmp_path / "sub" / "m.json.zip", model_name="m")
    with zipfile.ZipFile(z) as zf:
        assert zf.namelist() == ["m.json"] and json.loads(zf.read("m.json")) == ok
    p = cs.write_provenance(tmp_path / "prov" / "p.json", {"b": 1, "a": 2})
    assert p.read_text() == json.dumps({"b": 1, "a": 2}, indent=2, sort_keys=True)
    # empty file / header only
    (tmp_path / "e.csv").write_text("# nothing\n")
    assert cs.parse_cmdstan_csv(tmp_path / "e.csv") == {}
    (tmp_path / "h.csv").write_text("lp__,mu\n")
    assert cs.parse_cmdstan_csv(tmp_path / "h.csv") == {}


def test_json_zip_reader_keeps_types_and_errors(tmp_path):
    from mcmc_ref_hip.convert import _read_json_zip

    def table_of(payload):
        p = tmp_path / "x.json.zip"
        with zipfile.ZipFile(p, "w") as zf:
            zf.writestr("x.json", json.dumps(payload))
        return _read_json_zip(p)
    exp = CASES["jsonzip"]
    t = table_of([{"k": [1, 2, 3], "m": [1, 2.5, 3], "f": [0.5, 1.5, 2.5]}, {"k": [4, 5, 6], "m": [4, 5, 6], "f": [1.0, 2.0, 3.0]}])
    assert t.column_names == exp["columns"]
    assert {n: str(t.schema.field(n).type) for n in t.column_names} == exp["schema"]
    assert t.column("k").to_pylist() == exp["k"]
    t = table_of([{"a": [1.0, 2.0], "b": [1.0, 2.0]}, {"a": [3.0, 4.0, 5.0], "b": [3.0, 4.0, 5.0]}])
    assert t.num_rows == exp["long_chain_rows"]
    kind, msg = exp["short_chain"]
    with pytest.raises(IndexError, match=msg):
        table_of([{"a": [1.0, 2.0]}, {"a": [3.0]}])
    assert kind == "IndexError"
    with pytest.raises(ValueError, match="non-empty list of chains"):
        table_of([])


def test_generate_bookkeeping_without_kernels(tmp_path):
    """Unknown names, missing scaffold files and failing runners are settled before any archive reaches the GPU."""
    from mcmc_ref_hip import generate
    g = CASES["generate"]
    scaffold = make_scaffold(tmp_path / "scaffold", files_for=[m for m in g["scaffold_models"] if m != "earn_height"])
    assert [r.name for r in generate.scaffold_recipes(scaffold)] == sorted(g["scaffold_models"])
    with pytest.raises(ValueError) as e:
        generate.generate_reference_corpus(scaffold_root=scaffold, output_root=tmp_path / "o", models=["zzz", "aaa"],
                                           runner=generate.fake_jsonzip_runner)
    assert str(e.value) == g["unknown"]

    def exploding(**kw):
        raise RuntimeError("sampler exploded")
    res = generate.generate_reference_corpus(scaffold_root=scaffold, output_root=tmp_path / "o2",
                                             models=["blr", "earn_height"], runner=exploding, force=True)
    assert (res.generated, res.failed) == (0, 2)
    assert res.errors == {k: v for k, v in g["mixed"]["errors"].items()}
    assert all((tmp_path / "o2" / d).is_dir() for d in ("archives", "draws", "meta"))
    # a scaffold without a manifest: recipes are the file stems
    bare = tmp_path / "bare"
    make_scaffold(bare, models=["m1", "m2"])
    (bare / "provenance_manifest.json").unlink()
    (bare / "stan_data" / "m2.json").unlink()
    assert [r.name for r in generate.scaffold_recipes(bare)] == ["m1", "m2"]
    cfg = generate.GenerationConfig()
    assert (cfg.chains, cfg.iter_sampling, cfg.iter_warmup, cfg.thin, cfg.seed) == (10, 10000, 10000, 10, 4711)


def test_fake_runner_payload(tmp_path):
    from mcmc_ref_hip import generate
    a = tmp_path / "arch" / "m.json.zip"
    generate.fake_jsonzip_runner(model_name="m", recipe=None, stan_file=None, data_file=None, archive_path=a,
                                 config=generate.GenerationConfig(chains=3, iter_sampling=5))
    with zipfile.ZipFile(a) as zf:
        payload = json.loads(zf.read("m.json"))
    assert len(payload) == 3 and payload[2]["mu"] == [2.0 + 0.001 * i for i in range(5)]
    assert payload[1]["sigma"] == [1.0 + v for v in payload[1]["mu"]]
    d = np.arange(24.0).reshape(4, 2, 3)                  # draws x chains x columns
    pl = generate._draws_to_chain_payload(d, ["lp__", "a", "b"])
    assert len(pl) == 2 and list(pl[0]) == ["a", "b"] and pl[1]["a"] == [4.0, 10.0, 16.0, 22.0]
    assert generate._draws_to_chain_payload(d.transpose(1, 0, 2), ["lp__", "a", "b"]) == pl
    with pytest.raises(ValueError, match="Unexpected CmdStan draws shape"):
        generate._draws_to_chain_payload(np.zeros((4, 2)), ["a"])


def test_publish(tmp_path):
    from mcmc_ref_hip import generate
    g = CASES["generate"]
    scaffold = make_scaffold(tmp_path / "scaffold", pairs=("p1", "p2", "p3"))
    src = tmp_path / "src"
    (src / "draws").mkdir(parents=True); (src / "meta").mkdir()
    (src / "draws" / "fresh.draws.parquet").write_bytes(b"parquet")
    (src / "draws" / "ignored.txt").write_text("x")
    (src / "meta" / "fresh.meta.json").write_text("{}")
    pkg = tmp_path / "pkg"
    (pkg / "draws").mkdir(parents=True); (pkg / "pairs" / "stale").mkdir(parents=True)
    (pkg / "draws" / "stale.draws.parquet").write_bytes(b"stale")
    (pkg / "provenance_manifest.json").write_text('{"old": true}')
    res = generate.publish_reference_data(source_root=src, scaffold_root=scaffold, package_root=pkg)
    assert (res.draws_copied, res.meta_copied, res.pairs_copied, res.package_root) == (1, 1, 3, pkg)
    assert not (pkg / "draws" / "stale.draws.parquet").exists() and not (pkg / "pairs" / "stale").exists()
    text = (pkg / "provenance_manifest.json").read_text()
    man = json.loads(text)
    assert text.endswith("\n") and text == json.dumps(man, indent=2, sort_keys=True) + "\n"
    assert sorted(man) == g["publish"]["manifest_keys"]
    import hashlib
    assert man["files"]["draws/fresh.draws.parquet"] == hashlib.sha256(b"parquet").hexdigest()
    assert sorted(man["files"]) == sorted(p.relative_to(pkg).as_posix() for p in pkg.rglob("*")
                                          if p.is_file() and p.name != "provenance_manifest.json")
    assert sum(1 for k in man["files"] if k.startswith("pairs/")) == 6
    with pytest.raises(FileNotFoundError) as e:
        generate.publish_reference_data(source_root=tmp_path / "nope", scaffold_root=scaffold, package_root=tmp_path / "p2")
    assert str(e.value).startswith(g["publish_missing_prefix"])
    with pytest.raises(FileNotFoundError, match="scaffold pairs directory not found"):
        generate.publish_reference_data(source_root=src, scaffold_root=tmp_path / "empty", package_root=tmp_path / "p3")


def test_compare_gate_on_the_host():
    """compare_stats evaluates a handful of pairs on the host (compare.py:41-43): no device, no library call."""
    from mcmc_ref_hip import compare
    for c in load_json("compare_cases.json")["compare"]:
        r = compare.compare_stats(c["ref"], c["actual"], c["tolerance"], c["metrics"])
        assert r.passed == c["passed"] and r.failures == c["failures"]
