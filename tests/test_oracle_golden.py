"""Pin the CPU oracle (oracle/mcr_oracle.c) against the reference's golden vectors.

Golden vectors come from tests/golden/make_golden.py (imported reference) and from
the reference's own packaged meta.json diagnostics.  CPU only.
"""
from __future__ import annotations

import math
from statistics import NormalDist

import numpy as np
import pytest

from conftest import MODEL_NAMES, load_json, load_model, rel_close, same_float

UNIT = load_json("unit_vectors.json")
CASES = [k for k in UNIT if not k.startswith("_")]


def test_inv_cdf_matches_stdlib_bitwise(oracle):
    nd = NormalDist()
    rng = np.random.default_rng(1)
    ps = list(rng.uniform(1e-12, 1 - 1e-12, size=20000)) + [0.5, 0.075, 0.925, 0.0750001, 1e-300,
                                                           5e-5, 1 - 5e-5, 1.25e-6, 1e-12]
    ps += [(r - 0.5) / 40000 for r in (1, 1.5, 2, 3000, 20000, 20000.5, 39999.5, 40000)]
    for p in ps:
        assert oracle.inv_cdf(p) == nd.inv_cdf(p), p


@pytest.mark.parametrize("name", CASES)
def test_unit_vectors_bitwise(oracle, name):
    rec = UNIT[name]
    chains = rec["chains"]
    for key, mc in (("min4", 4), ("min1", 1)):
        exp = rec[key]
        if isinstance(exp["rhat"], dict):
            with pytest.raises(ValueError) as ei:
                oracle.split_rhat(chains, min_chains=mc)
            assert str(ei.value) == exp["rhat"]["error"]
            with pytest.raises(ValueError) as ei:
                oracle.ess_bulk(chains, min_chains=mc)
            assert str(ei.value) == exp["ess_bulk"]["error"]
            with pytest.raises(ValueError) as ei:
                oracle.ess_tail(chains, min_chains=mc)
            assert str(ei.value) == exp["ess_tail"]["error"]
            continue
        got = oracle.diag(chains, mc)
        for k in ("rhat", "ess_bulk", "ess_tail"):
            assert same_float(got[k], exp[k]), (name, key, k, got[k], exp[k])
        if "rhat_bulk" in exp:
            for k in ("rhat_bulk", "rhat_tail"):
                assert same_float(got[k], exp[k]), (name, key, k)
            assert got["lag_bulk"] == exp["lag_bulk"] and got["lag_tail"] == exp["lag_tail"]
    if "z" in rec:
        z, ranks = oracle.rank_normalize(chains)
        for a, b in zip(z, rec["z"]):
            assert list(a) == b
        f, _ = oracle.fold(chains)
        for a, b in zip(f, rec["folded"]):
            assert list(a) == b
        zf, _ = oracle.rank_normalize([list(c) for c in f])
        for a, b in zip(zf, rec["z_folded"]):
            assert list(a) == b
        # ranks are exact multiples of 0.5 and agree with scipy's average ranks
        from scipy.stats import rankdata
        flat = np.concatenate([np.asarray(c, dtype=float) for c in chains])
        assert np.array_equal(np.concatenate(ranks), rankdata(flat, method="average"))


def test_min_chains_zero_message(oracle):
    with pytest.raises(ValueError) as ei:
        oracle.split_rhat([[1.0] * 4] * 4, min_chains=0)
    assert str(ei.value) == UNIT["_min_chains_0_error"]


@pytest.mark.parametrize("name", MODEL_NAMES)
def test_packaged_model_goldens(oracle, name):
    draws, params, rec = load_model(name)
    P, C, N = draws.shape
    assert (C, N) == (rec["n_chains"], rec["n_draws_per_chain"])
    s = oracle.summarize(draws, "pcn")
    n_bit = 0
    for i, p in enumerate(params):
        # (1) imported reference, same interpreter family: bitwise
        r = rec["recomputed"][p]
        for k in ("rhat", "ess_bulk", "ess_tail", "rhat_bulk", "rhat_tail"):
            assert same_float(float(s[k][i]), r[k]), (p, k, s[k][i], r[k])
        assert int(s["lag_bulk"][i]) == r["lag_bulk"] and int(s["lag_tail"][i]) == r["lag_tail"]
        # (2) the reference's packaged meta.json goldens (written by another Python build): 1e-12
        m = rec["meta_diagnostics"][p]
        for k in ("rhat", "ess_bulk", "ess_tail"):
            assert rel_close(float(s[k][i]), m[k], 1e-12), (p, k, s[k][i], m[k])
            n_bit += float(s[k][i]) == m[k]
        # (3) Backend.stats of both reference backends
        for b in ("arrow", "numpy"):
            st = rec["stats"][b][p]
            assert abs(float(s["mean"][i]) - st["mean"]) <= 1e-13 * (abs(st["mean"]) + st["std"])
            assert rel_close(float(s["std"][i]), st["std"], 1e-12)
            for j, qk in enumerate(("q5", "q50", "q95")):
                assert rel_close(float(s["q"][i, j]), st[qk], 1e-14), (b, p, qk, s["q"][i, j], st[qk])
        # numpy quantile arithmetic is restated exactly
        for j, qk in enumerate(("q5", "q50", "q95")):
            assert float(s["q"][i, j]) == rec["stats"]["numpy"][p][qk]
    assert n_bit > 0
    ess_ok = all(v > 400 for v in s["ess_bulk"])
    rhat_ok = all(v < 1.01 for v in s["rhat"])
    chk = {"ndraws_is_10k": C * N == 10000, "nchains_is_gte_4": C >= 4, "ess_above_400": ess_ok,
           "rhat_below_1_01": rhat_ok}
    assert chk == rec["meta_checks"] == rec["recomputed_checks"]


def test_synth_c1_subset(oracle):
    from mcmc_ref_hip import synth
    g = load_json("synth_c1_subset.json")
    x = synth.c1_model(g["C"], g["N"], g["P"], seed=g["seed"], params=g["params"])
    s = oracle.summarize(x, "pcn")
    for i, p in enumerate(g["params"]):
        r = g["results"][str(p)]
        for k in ("rhat", "ess_bulk", "ess_tail", "rhat_bulk", "rhat_tail"):
            assert same_float(float(s[k][i]), r[k]), (p, k)
        assert (int(s["lag_bulk"][i]), int(s["lag_tail"][i])) == (r["lag_bulk"], r["lag_tail"])
        assert float(s["q"][i, 1]) == r["stats"]["numpy"]["v"]["q50"]
        # mean of a location-0 parameter is a cancelling sum: tolerance is relative to the scale
        assert abs(float(s["mean"][i]) - r["stats"]["arrow"]["v"]["mean"]) <= 1e-13 * (
            abs(float(s["mean"][i])) + float(s["std"][i]))
        assert rel_close(float(s["std"][i]), r["stats"]["numpy"]["v"]["std"], 1e-11)
        b = oracle.basic_stats(x[i].reshape(-1))
        assert b["mean"] == r["basic"]["mean"] and b["std"] == r["basic"]["std"]
    # strided layouts give identical answers
    xc = np.ascontiguousarray(np.transpose(x[:3], (1, 2, 0)))       # [C][N][P]
    s2 = oracle.summarize(xc, "cnp")
    for k in ("rhat", "ess_bulk", "ess_tail", "mean", "std"):
        assert np.array_equal(s2[k], s[k][:3])


def test_compare_and_basic_cases(oracle):
    g = load_json("compare_cases.json")
    for rec in g["basic"]:
        out = oracle.basic_stats(rec["values"])
        assert same_float(out["mean"], rec["out"]["mean"]) and same_float(out["std"], rec["out"]["std"])
    for rec in g["compare"]:
        for p, ms in rec["details"].items():
            names = list(ms)
            rel, ok = oracle.compare([ms[m]["ref"] for m in names], [ms[m]["actual"] for m in names],
                                     rec["tolerance"])
            for j, m in enumerate(names):
                assert same_float(float(rel[j]), ms[m]["rel_error"]) and bool(ok[j]) == ms[m]["passed"]
    qg = g["quantile_grid"]
    st = oracle.stats(qg["values"], qg["quantiles"])
    for q in qg["quantiles"]:
        k = f"q{int(q * 100)}"
        assert st[k] == qg["numpy"]["v"][k]
        assert rel_close(st[k], qg["arrow"]["v"][k], 1e-14)
    n = len(qg["values"])
    assert st["_q_lo"] == [min(int(math.floor((n - 1) * q)), n - 1) for q in qg["quantiles"]]
    bc = g["backends_consistency"]
    st = oracle.stats([float(i) * 0.1 for i in range(100)])
    for b in ("arrow", "numpy"):
        assert rel_close(st["mean"], bc[b]["mu"]["mean"], 1e-10) and rel_close(st["std"], bc[b]["mu"]["std"], 1e-10)


def test_numpy_path_agrees_with_the_c_oracle(oracle):
    """oracle/numpy_path.py (the "NumPy CPU path" timed by bench.py) against the C restatement: integers exact,
    floats to 1e-9 (ndtri / FFT differ from AS241 / direct sums in the last bits)."""
    from oracle import numpy_path
    rng = np.random.default_rng(3)
    cases = [rng.normal(size=(3, 4, 500)), np.round(rng.normal(size=(2, 4, 301)), 1),
             np.cumsum(rng.normal(size=(2, 3, 400)), axis=2) * 0.05 + rng.normal(size=(2, 3, 400)),
             np.stack([np.full((4, 50), 2.5), rng.normal(size=(4, 50))])]
    for x in cases:
        a, b = numpy_path.summarize(x), oracle.summarize(x, "pcn", min_chains=1)
        assert np.array_equal(a["lag_bulk"], b["lag_bulk"]) and np.array_equal(a["lag_tail"], b["lag_tail"])
        assert np.array_equal(a["q"], b["q"]) and np.array_equal(a["median"], b["median"])
        for k in ("mean", "std", "rhat", "rhat_bulk", "rhat_tail", "ess_bulk", "ess_tail"):
            assert np.allclose(a[k], b[k], rtol=1e-9, atol=1e-12, equal_nan=True), k
