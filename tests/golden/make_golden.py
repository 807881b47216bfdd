#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference); its outputs -- plain
data: inputs + the reference's outputs -- are committed so that the GPU box and
CI can check the oracle and the HIP path without the reference being present.

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden.py

Emits
  unit_vectors.json      small known-answer cases (reference's own test inputs + edge cases)
  synth_c1_subset.json   reference outputs for 12 params of the C1 synthetic model (4x10000)
  models/<name>.npz      draws of a few packaged models as [P][C][N] f64 + param names
  models/<name>.json     their packaged meta.json diagnostics (the reference's own goldens),
                         reference.stats() for backend arrow and numpy, recomputed diagnostics
  compare_cases.json     compare_stats / compute_basic_stats known answers
"""
from __future__ import annotations

import json
import os
import sys
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
sys.dont_write_bytecode = True
sys.path[:0] = [str(REF / "src"), str(REF / "packages/mcmc-ref-data/src")]
os.environ.setdefault("MCMC_REF_LOCAL_ROOT", "/tmp/nonexistent-mcmc-ref-root")

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parents[1] / "mcmc-db_amd"))

from mcmc_ref import compare as ref_compare  # noqa: E402
from mcmc_ref import diagnostics as ref_diag  # noqa: E402
from mcmc_ref import reference as ref_api  # noqa: E402
from mcmc_ref.backends import get_backend  # noqa: E402
from mcmc_ref.convert import _chains_from_table, _checks  # noqa: E402
from mcmc_ref.store import DataStore  # noqa: E402

from mcmc_ref_hip import synth  # noqa: E402  (this repo's workload generator)


def _count_lags(z):
    """Number of rho terms the reference accumulates in _ess (diagnostics.py:171-177)."""
    calls = {"n": 0, "neg": False}
    orig = ref_diag._autocorr

    def wrapped(chains, lag, var_hat):
        r = orig(chains, lag, var_hat)
        calls["n"] += 1
        if r < 0:
            calls["neg"] = True
        return r

    ref_diag._autocorr = wrapped
    try:
        ess = ref_diag._ess(z)
    finally:
        ref_diag._autocorr = orig
    terms = calls["n"] - 1 if calls["neg"] else calls["n"]
    return ess, terms


def ref_full(chains, min_chains=4):
    """Everything the reference computes for one parameter, incl. integer intermediates."""
    out = {}
    for name, fn in (("rhat", ref_diag.split_rhat), ("ess_bulk", ref_diag.ess_bulk),
                     ("ess_tail", ref_diag.ess_tail)):
        try:
            out[name] = fn(chains, min_chains=min_chains)
        except ValueError as e:
            out[name] = {"error": str(e)}
    if isinstance(out["rhat"], dict) or len(chains) < 2:
        return out
    z = ref_diag._rank_normalize(chains)
    folded = ref_diag._fold_chains(chains)
    zf = ref_diag._rank_normalize(folded)
    out["rhat_bulk"] = ref_diag._rhat(ref_diag._split_chains(z))
    out["rhat_tail"] = ref_diag._rhat(ref_diag._split_chains(zf))
    e1, l1 = _count_lags(z)
    e2, l2 = _count_lags(zf)
    assert (e1 == out["ess_bulk"] or (e1 != e1 and out["ess_bulk"] != out["ess_bulk"]))
    assert (e2 == out["ess_tail"] or (e2 != e2 and out["ess_tail"] != out["ess_tail"]))
    out["lag_bulk"], out["lag_tail"] = l1, l2
    return out


def unit_vectors():
    rng = np.random.default_rng(20260104)
    cases = {}
    # the reference's own test inputs (tests/unit/test_diagnostics.py:10-58, tests/unit/test_cli.py:13-21)
    cases["ref_ess_positive"] = [[1.0, 2.0, 3.0, 4.0], [1.1, 2.1, 3.1, 4.1], [0.9, 1.9, 2.9, 3.9],
                                 [1.05, 2.05, 3.05, 4.05]]
    cases["ref_scale_diff"] = [[0.0] * 4, [10.0] * 4, [0.0] * 4, [10.0] * 4]
    cases["ref_identical"] = [[1.0] * 4] * 4
    cases["ref_cli_4x2"] = [[1.0, 2.0], [1.5, 2.5], [1.0, 2.0], [1.5, 2.5]]
    cases["ref_two_chains"] = [[1.0, 2.0, 3.0, 4.0], [1.1, 2.1, 3.1, 4.1]]
    cases["ref_single_chain"] = [[1.0, 2.0, 3.0, 4.0]]
    # edge cases
    cases["n1"] = [[1.0], [2.0], [3.0], [4.0]]
    cases["n3_odd"] = [[0.3, -1.2, 2.2], [0.1, 0.4, -0.7], [1.5, 1.4, -2.0], [0.0, 0.9, 0.8]]
    cases["n0_empty"] = [[], [], [], []]
    cases["neg_zero_ties"] = [[0.0, -0.0, 1.0, -1.0, 0.0], [-0.0, 2.0, -2.0, 0.0, 1.0],
                              [1.0, 1.0, -1.0, 0.0, 3.0], [0.5, -0.5, 0.0, -0.0, 0.25]]
    cases["ragged"] = [list(rng.normal(size=9)), list(rng.normal(size=7)), list(rng.normal(size=8)),
                       list(rng.normal(size=11))]
    cases["iid_4x64"] = [list(rng.normal(size=64)) for _ in range(4)]
    cases["iid_4x101_odd"] = [list(rng.normal(loc=3.0, scale=0.01, size=101)) for _ in range(4)]
    cases["iid_10x50"] = [list(rng.standard_t(3, size=50)) for _ in range(10)]
    cases["iid_2x33"] = [list(rng.normal(size=33)) for _ in range(2)]
    cases["iid_7x40"] = [list(rng.exponential(size=40)) for _ in range(7)]
    x = rng.normal(size=(4, 200))
    for c in range(4):                      # AR(1) phi=0.9 -> long truncation lags
        for t in range(1, 200):
            x[c, t] = 0.9 * x[c, t - 1] + np.sqrt(1 - 0.81) * x[c, t]
    cases["ar1_4x200"] = [list(r) for r in x]
    cases["rounded_ties_4x100"] = [list(np.round(rng.normal(size=100), 1)) for _ in range(4)]
    cases["few_levels_4x60"] = [list(rng.integers(0, 3, size=60).astype(float)) for _ in range(4)]
    cases["shifted_chain_4x80"] = [list(rng.normal(size=80) + (2.0 if c == 3 else 0.0)) for c in range(4)]
    cases["one_const_chain"] = [[2.0] * 6, list(rng.normal(size=6)), list(rng.normal(size=6)),
                                list(rng.normal(size=6))]
    cases["huge_scale"] = [list(rng.normal(loc=1e9, scale=1e-3, size=32)) for _ in range(4)]
    cases["tiny_scale"] = [list(rng.normal(loc=0, scale=1e-200, size=32)) for _ in range(4)]
    cases["monotone_trend"] = [list(np.arange(40.0) + c * 0.25) for c in range(4)]

    out = {}
    for name, chains in cases.items():
        chains = [[float(v) for v in c] for c in chains]
        rec = {"chains": chains}
        rec["min4"] = ref_full(chains, 4)
        rec["min1"] = ref_full(chains, 1)
        if sum(len(c) for c in chains) > 0 and len(chains) >= 1:
            z = ref_diag._rank_normalize(chains)
            rec["z"] = z
            f = ref_diag._fold_chains(chains)
            rec["folded"] = f
            rec["z_folded"] = ref_diag._rank_normalize(f)
        out[name] = rec
    try:
        ref_diag.split_rhat(cases["ref_identical"], min_chains=0)
    except ValueError as e:
        out["_min_chains_0_error"] = str(e)
    return out


def stats_both(table, params):
    return {b: get_backend(b).stats(table, params) for b in ("arrow", "numpy")}


def synth_c1_subset():
    C, N, P = 4, 10000, 100
    pick = [0, 1, 19, 24, 33, 49, 50, 59, 74, 79, 98, 99]
    x = synth.c1_model(C, N, P, seed=4711, params=pick)        # [len(pick)][C][N]
    recs = {}
    import pyarrow as pa
    for k, p in enumerate(pick):
        chains = [list(map(float, x[k, c])) for c in range(C)]
        r = ref_full(chains, 4)
        tbl = pa.table({"v": pa.array(x[k].reshape(-1))})
        r["stats"] = stats_both(tbl, ["v"])
        r["basic"] = ref_compare.compute_basic_stats(list(map(float, x[k].reshape(-1))))
        recs[str(p)] = r
        print("synth param", p, r["rhat"], r["ess_bulk"], r["lag_bulk"], r["lag_tail"], flush=True)
    return {"C": C, "N": N, "P": P, "seed": 4711, "params": pick, "results": recs}


MODELS = [
    "eight_schools-eight_schools_noncentered",   # BASELINE config 0
    "radon_pooled",                               # 4 x 2500
    "wells_data-wells_dist",                      # smallest P
    "arK-arK",
    "gp_pois_regr-gp_regr",
    "garch-garch11",
]


def model_fixture(store, name, outdir):
    table = store.open_draws(name).read_all()
    meta = store.read_meta(name)
    params = [c for c in table.column_names if c not in {"chain", "draw"}]
    chains0 = _chains_from_table(table, params[0])
    C, N = len(chains0), len(chains0[0])
    arr = np.empty((len(params), C, N), dtype=np.float64)
    recomputed = {}
    for i, p in enumerate(params):
        ch = _chains_from_table(table, p)
        arr[i] = np.asarray(ch, dtype=np.float64)
        recomputed[p] = ref_full(ch, 4)
    np.savez(outdir / f"{name}.npz", draws=arr, params=np.array(params))
    rec = {
        "model": name, "params": params, "n_chains": meta["n_chains"],
        "n_draws_per_chain": meta["n_draws_per_chain"],
        "meta_diagnostics": meta["diagnostics"],        # the reference's own packaged goldens
        "meta_checks": meta["checks"],
        "recomputed": recomputed,                       # imported reference, python 3.10 here
        "recomputed_checks": _checks(C, N, {p: recomputed[p] for p in params}),
        "stats": stats_both(table, params),
        "chain_order_is_chain_major": bool(
            np.array_equal(np.asarray(table.column("chain")), np.repeat(np.arange(C), N))),
    }
    (outdir / f"{name}.json").write_text(json.dumps(rec, indent=1))
    print("model", name, arr.shape, flush=True)


def compare_cases():
    out = {}
    out["basic"] = []
    rng = np.random.default_rng(7)
    for vals in ([1, 2, 1.5, 2.5, 1, 2, 1.5, 2.5], [], [3.0], list(rng.normal(5, 2, size=257)),
                 [float(i) * 0.1 for i in range(100)]):
        vals = [float(v) for v in vals]
        out["basic"].append({"values": vals, "out": ref_compare.compute_basic_stats(vals)})
    cc = []
    for ref, act, tol, metrics in (
        ({"mu": {"mean": 1.0, "std": 1.0}}, {"mu": {"mean": 1.05, "std": 0.95}}, 0.1, ["mean", "std"]),
        ({"mu": {"mean": 1.0, "std": 1.0}}, {"mu": {"mean": 2.0, "std": 1.0}}, 0.1, ["mean", "std"]),
        ({"mu": {"mean": 0.0, "std": 1.0}}, {"mu": {"mean": 1e-13, "std": 1.0}}, 0.15, ["mean", "std"]),
        ({"mu": {"mean": 1.0}, "tau": {"mean": 2.0}}, {"mu": {"mean": 1.0}}, 0.15, ["mean"]),
        ({"mu": {"mean": 1.0, "std": 1.0}}, {"mu": {"mean": 1.0, "std": 1.0}}, 0.15, ["mean", "q5"]),
        ({"mu": {"mean": float("nan"), "std": 1.0}}, {"mu": {"mean": 1.0, "std": float("inf")}}, 0.15,
         ["mean", "std"]),
    ):
        r = ref_compare.compare_stats(ref, act, tol, metrics)
        cc.append({"ref": ref, "actual": act, "tolerance": tol, "metrics": metrics,
                   "passed": r.passed, "failures": r.failures,
                   "details": {p: {m: {"ref": d.ref, "actual": d.actual, "rel_error": d.rel_error,
                                       "passed": d.passed} for m, d in ms.items()}
                               for p, ms in r.details.items()}})
    out["compare"] = cc
    import pyarrow as pa
    tb = pa.table({"mu": pa.array([float(i) * 0.1 for i in range(100)], type=pa.float64())})
    out["backends_consistency"] = stats_both(tb, ["mu"])       # tests/unit/test_backends_consistency.py
    qs = [0.0, 0.01, 0.25, 0.5, 0.75, 0.999, 1.0]
    v = rng.normal(size=1001)
    tb = pa.table({"v": pa.array(v)})
    out["quantile_grid"] = {"values": [float(t) for t in v], "quantiles": qs,
                            "arrow": get_backend("arrow").stats(tb, ["v"], quantiles=qs),
                            "numpy": get_backend("numpy").stats(tb, ["v"], quantiles=qs)}
    return out


def main():
    (HERE / "models").mkdir(exist_ok=True)
    (HERE / "unit_vectors.json").write_text(json.dumps(unit_vectors(), indent=1))
    (HERE / "compare_cases.json").write_text(json.dumps(compare_cases(), indent=1))
    store = DataStore()
    for m in MODELS:
        model_fixture(store, m, HERE / "models")
    (HERE / "synth_c1_subset.json").write_text(json.dumps(synth_c1_subset(), indent=1))
    assert ref_api  # imported to prove the API module loads


if __name__ == "__main__":
    main()
