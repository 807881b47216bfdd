#!/usr/bin/env python3
"""Golden vectors for the callers around the hot path, made by IMPORTING the reference.

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden_pipeline.py

Runs only in the build container (needs /root/reference).  Emits data only:
  cmdstan/chain_{1,2}.csv   two small CmdStan-format chain files WRITTEN BY THIS SCRIPT (comment lines, sampler
                            internals `lp__` ..., scalar / vector / matrix parameter columns)
  pipeline_cases.json       what the reference returns for them and for its provenance-generate / -publish flow:
     cmdstan:    parse_cmdstan_csv() of both files, the name-normalisation table, payload validation messages
     generate:   generate_reference_corpus() with the reference's own fake_jsonzip_runner (4 chains x 300 draws):
                 the GenerationResult and the meta.json it wrote (diagnostics = the hot path's output), with and
                 without --force, a failing runner, missing scaffold files, an unknown model name
     jsonzip:    Arrow schema types convert._read_json_zip infers for integer / mixed / float draws, short chain error
     publish:    PublishResult counters and the set of files hashed into the manifest
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
import zipfile
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
sys.dont_write_bytecode = True
sys.path[:0] = [str(REF / "src"), str(REF / "packages/mcmc-ref-data/src")]
os.environ.setdefault("MCMC_REF_LOCAL_ROOT", "/tmp/nonexistent-mcmc-ref-root")
HERE = Path(__file__).resolve().parent

from mcmc_ref import cmdstan_generate as ref_cs  # noqa: E402
from mcmc_ref import convert as ref_convert  # noqa: E402
from mcmc_ref import generate as ref_gen  # noqa: E402
from mcmc_ref import provenance as ref_prov  # noqa: E402


def write_cmdstan_files():
    rng = np.random.default_rng(4711)
    header = ["lp__", "accept_stat__", "stepsize__", "treedepth__", "n_leapfrog__", "divergent__", "energy__",
              "mu", "tau", "theta.1", "theta.2", "theta.3", "Sigma.1.1", "Sigma.1.2", "Sigma.2.1", "Sigma.2.2",
              "log_lik.10", "x_raw"]
    (HERE / "cmdstan").mkdir(exist_ok=True)
    for chain in (1, 2):
        lines = ["# model = demo_model", f"# id = {chain}", "#     num_samples = 40", "#     seed = 4711",
                 ",".join(header), "# Adaptation terminated", "# Step size = 0.35", "# Diagonal elements of inverse mass matrix:",
                 "# 1.1, 0.9"]
        for _ in range(40):
            row = [rng.normal(-7, 1), rng.uniform(0.6, 1), 0.35, float(rng.integers(2, 5)), float(rng.integers(3, 31)), 0.0,
                   rng.normal(10, 2)] + list(rng.normal(size=len(header) - 7) * 10.0 ** rng.integers(-3, 3))
            lines.append(",".join(repr(float(v)) if i not in (3, 4, 5) else str(int(v)) for i, v in enumerate(row)))
        lines += ["# ", "#  Elapsed Time: 0.01 seconds (Warm-up)", "#                0.02 seconds (Sampling)"]
        (HERE / "cmdstan" / f"chain_{chain}.csv").write_text("\n".join(lines) + "\n")


def cmdstan_cases():
    out = {"files": {}}
    for chain in (1, 2):
        out["files"][f"chain_{chain}.csv"] = ref_cs.parse_cmdstan_csv(HERE / "cmdstan" / f"chain_{chain}.csv")
    names = ["mu", "theta.1", "theta.12.3", "Sigma.1.2.3", "a.b", "x.1a", "lp__", "theta.", ".1", "z_9.0", "beta[1]", "_t.2"]
    out["normalize"] = {n: ref_cs._normalize_cmdstan_param_name(n) for n in names}
    msgs = {}
    for label, payload in (("empty", []), ("no_params", [{}]), ("key_mismatch", [{"a": [1.0]}, {"b": [1.0]}]),
                           ("ragged", [{"a": [1.0], "b": [1.0, 2.0]}])):
        try:
            ref_cs.build_posteriordb_payload(payload)
            msgs[label] = None
        except ValueError as e:
            msgs[label] = str(e)
    out["payload_errors"] = msgs
    return out


def _failing_runner(**kw):
    if kw["model_name"] == "blr":
        raise RuntimeError("sampler exploded")
    ref_gen.fake_jsonzip_runner(**kw)


def generate_cases():
    out = {}
    cfg = ref_gen.GenerationConfig(chains=4, iter_sampling=300)
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        scaffold = td / "scaffold"
        ref_prov.materialize_scaffold(scaffold)
        out["scaffold_models"] = json.loads((scaffold / "provenance_manifest.json").read_text())["models"]
        # forced: diagnostics of the fake payload are the hot path's output
        res = ref_gen.generate_reference_corpus(scaffold_root=scaffold, output_root=td / "g1", models=["dugongs"],
                                                config=cfg, force=True, runner=ref_gen.fake_jsonzip_runner)
        meta = json.loads((td / "g1" / "meta" / "dugongs.meta.json").read_text())
        meta.pop("generated_date")
        out["forced"] = {"generated": res.generated, "failed": res.failed, "errors": res.errors, "meta": meta}
        # not forced: the quality gate message lands in errors{}
        res = ref_gen.generate_reference_corpus(scaffold_root=scaffold, output_root=td / "g2", models=["dugongs"],
                                                config=cfg, force=False, runner=ref_gen.fake_jsonzip_runner)
        out["gated"] = {"generated": res.generated, "failed": res.failed, "errors": res.errors,
                        "wrote_draws": (td / "g2" / "draws" / "dugongs.draws.parquet").exists()}
        # one failing runner, one missing scaffold file, one good recipe
        (scaffold / "stan_data" / "earn_height.json").unlink()
        res = ref_gen.generate_reference_corpus(scaffold_root=scaffold, output_root=td / "g3",
                                                models=["blr", "earn_height", "dugongs"], config=cfg, force=True,
                                                runner=_failing_runner)
        out["mixed"] = {"generated": res.generated, "failed": res.failed, "errors": res.errors}
        try:
            ref_gen.generate_reference_corpus(scaffold_root=scaffold, output_root=td / "g4", models=["zzz", "aaa"],
                                              config=cfg, runner=ref_gen.fake_jsonzip_runner)
        except ValueError as e:
            out["unknown"] = str(e)
        # single chain, not forced
        res = ref_gen.generate_reference_corpus(scaffold_root=scaffold, output_root=td / "g5", models=["dugongs"],
                                                config=ref_gen.GenerationConfig(chains=1, iter_sampling=50), force=False,
                                                runner=ref_gen.fake_jsonzip_runner)
        out["single_chain"] = {"generated": res.generated, "failed": res.failed, "errors": res.errors}
        # publish
        pub = ref_gen.publish_reference_data(source_root=td / "g1", scaffold_root=scaffold, package_root=td / "pkg")
        man = json.loads((td / "pkg" / "provenance_manifest.json").read_text())
        out["publish"] = {"draws_copied": pub.draws_copied, "meta_copied": pub.meta_copied, "pairs_copied": pub.pairs_copied,
                          "manifest_keys": sorted(man), "hashed_draws_meta": sorted(k for k in man["files"] if k.split("/")[0] in ("draws", "meta")),
                          "n_hashed": len(man["files"]), "n_pair_files": sum(1 for k in man["files"] if k.startswith("pairs/"))}
        try:
            ref_gen.publish_reference_data(source_root=td / "nope", scaffold_root=scaffold, package_root=td / "pkg2")
        except FileNotFoundError as e:
            out["publish_missing_prefix"] = str(e).split(":")[0]
    return out


def jsonzip_cases():
    out = {}
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)

        def table_of(payload):
            p = td / "x.json.zip"
            with zipfile.ZipFile(p, "w") as zf:
                zf.writestr("x.json", json.dumps(payload))
            return ref_convert._read_json_zip(p)
        t = table_of([{"k": [1, 2, 3], "m": [1, 2.5, 3], "f": [0.5, 1.5, 2.5]}, {"k": [4, 5, 6], "m": [4, 5, 6], "f": [1.0, 2.0, 3.0]}])
        out["schema"] = {n: str(t.schema.field(n).type) for n in t.column_names}
        out["columns"] = t.column_names
        out["k"] = t.column("k").to_pylist()
        t = table_of([{"a": [1.0, 2.0], "b": [1.0, 2.0]}, {"a": [3.0, 4.0, 5.0], "b": [3.0, 4.0, 5.0]}])   # longer chain: cut
        out["long_chain_rows"] = t.num_rows
        try:
            table_of([{"a": [1.0, 2.0]}, {"a": [3.0]}])
            out["short_chain"] = None
        except Exception as e:  # noqa: BLE001
            out["short_chain"] = [type(e).__name__, str(e)]
    return out


def main():
    write_cmdstan_files()
    out = {"cmdstan": cmdstan_cases(), "generate": generate_cases(), "jsonzip": jsonzip_cases()}
    (HERE / "pipeline_cases.json").write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")
    print("wrote", HERE / "pipeline_cases.json")


if __name__ == "__main__":
    main()
