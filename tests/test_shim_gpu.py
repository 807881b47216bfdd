"""GPU tests of the host-side mirror of the reference interface (mcmc_ref_hip.*).

The first block restates the reference's own unit tests (tests/unit/test_diagnostics.py:10-58,
test_compare.py:6-24, test_backends_consistency.py:17-28) against this package; the rest checks
the same calls against the golden vectors produced by the imported reference.
"""
from __future__ import annotations

import math

import numpy as np
import pytest

from conftest import load_json, load_model

pytestmark = pytest.mark.gpu

UNIT = load_json("unit_vectors.json")
CMP = load_json("compare_cases.json")


# ---- the reference's tests/unit/test_diagnostics.py, verbatim assertions ----------------------
def test_split_rhat_identical_chains():
    from mcmc_ref_hip.diagnostics import split_rhat
    rhat = split_rhat([[1.0, 1.0, 1.0, 1.0]] * 4)
    assert 0.99 <= rhat <= 1.01


def test_ess_positive():
    from mcmc_ref_hip.diagnostics import ess_bulk
    chains = [[1.0, 2.0, 3.0, 4.0], [1.1, 2.1, 3.1, 4.1], [0.9, 1.9, 2.9, 3.9], [1.05, 2.05, 3.05, 4.05]]
    assert ess_bulk(chains) > 0


def test_split_rhat_detects_scale_diff():
    from mcmc_ref_hip.diagnostics import split_rhat
    chains = [[0.0] * 4, [10.0] * 4, [0.0] * 4, [10.0] * 4]
    assert split_rhat(chains) > 1.1


def test_split_rhat_requires_four_chains_by_default():
    from mcmc_ref_hip.diagnostics import split_rhat
    with pytest.raises(ValueError, match="at least 4 chains"):
        split_rhat([[1.0, 2.0, 3.0, 4.0], [1.1, 2.1, 3.1, 4.1]])


def test_split_rhat_allows_single_chain_when_explicitly_overridden():
    from mcmc_ref_hip.diagnostics import split_rhat
    assert math.isnan(split_rhat([[1.0, 2.0, 3.0, 4.0]], min_chains=1))


def test_ess_bulk_requires_four_chains_by_default():
    from mcmc_ref_hip.diagnostics import ess_bulk
    with pytest.raises(ValueError, match="at least 4 chains"):
        ess_bulk([[1.0, 2.0, 3.0, 4.0], [1.1, 2.1, 3.1, 4.1]])


# ---- exact values of those inputs (captured from the imported reference) ---------
@pytest.mark.parametrize("name", ["ref_ess_positive", "ref_scale_diff", "ref_identical", "ref_cli_4x2", "ragged",
                                  "ar1_4x200", "rounded_ties_4x100", "n3_odd", "neg_zero_ties"])
def test_diagnostics_known_answers(name):
    from mcmc_ref_hip import diagnostics
    rec = UNIT[name]
    exp = rec["min4"]
    for fn, key in ((diagnostics.split_rhat, "rhat"), (diagnostics.ess_bulk, "ess_bulk"),
                    (diagnostics.ess_tail, "ess_tail")):
        got = fn(rec["chains"])
        e = exp[key]
        assert (math.isnan(got) and math.isnan(e)) or got == pytest.approx(e, rel=1e-9), (name, key)


# ---- tests/unit/test_compare.py ---------------------------------------------------------------
def test_compare_stats_passes():
    from mcmc_ref_hip.compare import compare_stats
    result = compare_stats({"mu": {"mean": 1.0, "std": 1.0}}, {"mu": {"mean": 1.05, "std": 0.95}},
                           tolerance=0.1, metrics=["mean", "std"])
    assert result.passed is True
    assert result.failures == []
    assert result.details["mu"]["mean"].passed is True


def test_compare_stats_fails():
    from mcmc_ref_hip.compare import compare_stats
    result = compare_stats({"mu": {"mean": 1.0, "std": 1.0}}, {"mu": {"mean": 2.0, "std": 1.0}},
                           tolerance=0.1, metrics=["mean", "std"])
    assert result.passed is False
    assert result.failures


def test_compare_stats_matches_reference_strings_and_details():
    from mcmc_ref_hip.compare import compare_stats
    for rec in CMP["compare"]:
        r = compare_stats(rec["ref"], rec["actual"], rec["tolerance"], rec["metrics"])
        assert r.passed == rec["passed"]
        assert r.failures == rec["failures"]
        assert set(r.details) == set(rec["details"])
        for p, ms in rec["details"].items():
            for m, d in ms.items():
                g = r.details[p][m]
                assert g.passed == d["passed"]
                for a, b in ((g.ref, d["ref"]), (g.actual, d["actual"]), (g.rel_error, d["rel_error"])):
                    assert (a != a and b != b) or a == b


def test_compute_basic_stats_known_answers():
    from mcmc_ref_hip.compare import compute_basic_stats, compute_stats_from_draws
    for rec in CMP["basic"]:
        out = compute_basic_stats(rec["values"])
        for k in ("mean", "std"):
            e = rec["out"][k]
            assert (math.isnan(out[k]) and math.isnan(e)) or out[k] == pytest.approx(e, rel=1e-12, abs=1e-300)
    draws = {"a": [1.0, 2.0, 3.0, 4.0], "b": [2.0, 2.0, 2.0, 2.0]}
    st = compute_stats_from_draws(draws)
    assert st["a"]["mean"] == 2.5 and st["a"]["std"] == pytest.approx(math.sqrt(1.25), rel=1e-15)
    assert st["b"] == {"mean": 2.0, "std": 0.0}
    st = compute_stats_from_draws({"a": [1.0, 3.0], "b": [1.0, 2.0, 3.0]})
    assert st["a"]["mean"] == 2.0 and st["b"]["mean"] == 2.0


# ---- tests/unit/test_backends_consistency.py ---------------------------------------------------
def test_hip_backend_agrees_with_arrow_and_numpy():
    import pyarrow as pa
    from mcmc_ref_hip.backends import get_backend
    table = pa.table({"chain": pa.array([0] * 100, type=pa.int32()),
                      "draw": pa.array(list(range(100)), type=pa.int32()),
                      "mu": pa.array([float(i) * 0.1 for i in range(100)], type=pa.float64())})
    be = get_backend("hip")
    assert be.name == "hip"
    st = be.stats(table, ["mu"])
    assert set(st["mu"]) == {"mean", "std", "q5", "q50", "q95"}
    for b in ("arrow", "numpy"):
        ref = CMP["backends_consistency"][b]["mu"]
        assert st["mu"]["mean"] == pytest.approx(ref["mean"], rel=1e-10)
        assert st["mu"]["std"] == pytest.approx(ref["std"], rel=1e-10)
        for k in ("q5", "q50", "q95"):
            assert st["mu"][k] == pytest.approx(ref[k], rel=1e-14)
    qg = CMP["quantile_grid"]
    tb = pa.table({"v": pa.array(qg["values"])})
    st = be.stats(tb, ["v"], quantiles=qg["quantiles"])
    assert st["v"] == pytest.approx(qg["numpy"]["v"], rel=1e-12)
    for q in qg["quantiles"]:
        assert st["v"][f"q{int(q * 100)}"] == qg["numpy"]["v"][f"q{int(q * 100)}"]      # bit-exact
    # nulls are skipped, as ArrowBackend.stats does (backends_arrow.py:38-42: pc.mean / pc.stddev / quantile(skip_nulls=True));
    # expected values from pyarrow itself on the same table, columns of different surviving lengths in one call
    import pyarrow.compute as pc
    tn = pa.table({"v": pa.array([1.0, None, 2.0, 4.5, None, -3.0]), "w": pa.array([0.5, 1.5, None, 2.5, 3.5, 9.0]),
                   "u": pa.array([1.0, 2.0, 3.0, 4.0, 5.0, 6.0])})
    got = be.stats(tn, ["v", "w", "u"])
    for name in ("v", "w", "u"):
        col = tn.column(name)
        assert got[name]["mean"] == pytest.approx(pc.mean(col).as_py(), rel=1e-14)
        assert got[name]["std"] == pytest.approx(pc.stddev(col).as_py(), rel=1e-14)
        qv = pc.quantile(col, q=[0.05, 0.5, 0.95], interpolation="linear", skip_nulls=True).to_pylist()
        assert [got[name]["q5"], got[name]["q50"], got[name]["q95"]] == pytest.approx(qv, rel=1e-14)
    with pytest.raises(ValueError):
        be.stats(pa.table({"v": pa.array([None, None], type=pa.float64())}), ["v"])     # nothing left
    with pytest.raises(ValueError):
        be.stats(pa.table({"v": pa.array([1.0, float("nan"), 2.0])}), ["v"])           # NaN is a value, and rejected


# ---- convert._compute_diagnostics on Arrow tables (the provenance-generate call site) ----------
def _table(draws, params, shuffle=None):
    import pyarrow as pa
    P, C, N = draws.shape
    cols = {"chain": np.repeat(np.arange(C), N), "draw": np.tile(np.arange(N), C)}
    for i, p in enumerate(params):
        cols[p] = draws[i].reshape(-1)
    if shuffle is not None:
        perm = np.random.default_rng(shuffle).permutation(C * N)
        cols = {k: v[perm] for k, v in cols.items()}
    return pa.table(cols)


@pytest.mark.parametrize("name", ["eight_schools-eight_schools_noncentered", "radon_pooled"])
def test_compute_diagnostics_matches_packaged_meta(name):
    from mcmc_ref_hip import convert
    draws, params, rec = load_model(name)
    for shuffle in (None, 5):
        tbl = _table(draws, params, shuffle)
        diag = convert._compute_diagnostics(tbl, params)
        assert list(diag) == params
        for p in params:
            for k in ("rhat", "ess_bulk", "ess_tail"):
                assert diag[p][k] == pytest.approx(rec["meta_diagnostics"][p][k], rel=1e-6)
                assert diag[p][k] == pytest.approx(rec["recomputed"][p][k], rel=1e-9)
        n_chains, n_draws = convert._count_chains_draws(tbl)
        assert convert._checks(n_chains, n_draws, diag) == rec["meta_checks"]
    with pytest.raises(ValueError, match="at least 11 chains"):
        convert._compute_diagnostics(tbl, params, min_chains=11)


def test_compute_diagnostics_ragged_table(oracle):
    import pyarrow as pa
    from mcmc_ref_hip import convert
    rng = np.random.default_rng(2)
    lens = [40, 37, 45, 40]
    chain = np.concatenate([np.full(n, c) for c, n in enumerate(lens)])
    draw = np.concatenate([np.arange(n) for n in lens])
    a = rng.normal(size=chain.size)
    tbl = pa.table({"chain": chain, "draw": draw, "a": a})
    diag = convert._compute_diagnostics(tbl, ["a"])
    off = np.concatenate([[0], np.cumsum(lens)])
    exp = oracle.diag([a[off[c]:off[c + 1]] for c in range(4)], 4)
    for k in ("rhat", "ess_bulk", "ess_tail"):
        assert diag["a"][k] == pytest.approx(exp[k], rel=1e-9)


def test_summarize_models_single_process(oracle):
    from mcmc_ref_hip import _ffi, shard
    rng = np.random.default_rng(8)
    models = [(rng.normal(size=(3, 4, 500)), "pcn"), (rng.normal(size=(4, 300, 2)), "cnp"),
              (rng.normal(size=(1, 10, 100)), "pcn"), (rng.normal(size=(5, 4, 64)), "pcn"),
              (rng.normal(size=(2, 6, 1000)), "pcn"), (rng.normal(size=(2, 4, 4096)), "pcn")]
    ctx = _ffi.default_context()
    rec = shard.summarize_models(ctx, models)
    assert rec.shape == (3 + 2 + 1 + 5 + 2 + 2, shard.RECORD_DOUBLES)
    F = shard.RECORD_FIELDS
    row = 0
    for mi, (arr, layout) in enumerate(models):
        exp = oracle.summarize(arr, layout)
        for p in range(len(exp["mean"])):
            r = rec[row]
            assert (r[F.index("model_idx")], r[F.index("param_idx")]) == (mi, p)
            assert r[F.index("q50")] == exp["q"][p, 1]
            assert r[F.index("lag_bulk")] == exp["lag_bulk"][p] and r[F.index("lag_tail")] == exp["lag_tail"][p]
            for k in ("std", "rhat", "ess_bulk", "ess_tail"):
                assert r[F.index(k)] == pytest.approx(exp[k][p], rel=1e-9)
            row += 1


def test_corpus_shaped_batch_matches_per_model(oracle):
    """BASELINE config 2 geometry: 57 model shapes; batched pipelines == per-model pipelines, and the
    six real fixture models inside the batch still reproduce their packaged meta.json goldens."""
    from conftest import MODEL_NAMES
    from mcmc_ref_hip import _ffi, corpus, shard
    real = {}
    recs = {}
    for name in MODEL_NAMES:
        draws, params, rec = load_model(name)
        real[name], recs[name] = draws, (params, rec)
    models = corpus.synthetic_corpus(seed=99, real_models=real)
    assert len(models) == 57 and sum(a.shape[0] for _, a in models) == 460
    ctx = _ffi.default_context()
    pairs = [(a, "pcn") for _, a in models]
    batched = shard.summarize_models(ctx, pairs, batch=True)
    single = shard.summarize_models(ctx, pairs, batch=False)
    assert batched.shape == (460, shard.RECORD_DOUBLES)
    assert np.array_equal(batched, single, equal_nan=True)
    F = shard.RECORD_FIELDS
    names = [n for n, _ in models]
    for name, (params, rec) in recs.items():
        mi = names.index(name)
        rows = batched[batched[:, F.index("model_idx")] == mi]
        assert rows.shape[0] == len(params)
        for j, pn in enumerate(params):
            for k in ("rhat", "ess_bulk", "ess_tail"):
                assert rows[j, F.index(k)] == pytest.approx(rec["meta_diagnostics"][pn][k], rel=1e-6)
    # spot-check three synthetic models against the oracle
    for mi in (0, 20, 41):
        if names[mi] in recs:
            continue
        exp = oracle.summarize(models[mi][1], "pcn")
        rows = batched[batched[:, F.index("model_idx")] == mi]
        assert np.array_equal(rows[:, F.index("lag_bulk")], exp["lag_bulk"])
        assert np.allclose(rows[:, F.index("ess_tail")], exp["ess_tail"], rtol=1e-9)


def test_summarize_models_c_abi(oracle):
    """mcr_summarize_models: many independent models in one C call (rolling window over the lanes)."""
    from mcmc_ref_hip import _ffi, synth
    ctx = _ffi.default_context()
    shapes = [(4, 700, 3), (10, 1000, 2), (4, 2500, 1), (6, 64, 4), (4, 4097, 2)] * 3      # 15 models > window
    xs = [synth.c1_model(C, N, P, seed=100 + i) for i, (C, N, P) in enumerate(shapes)]
    ts = [ctx.upload(x, "pcn") for x in xs]
    try:
        res = ctx.summarize_models(ts)
        assert len(res) == len(xs)
        for x, r in zip(xs, res):
            exp = oracle.summarize(x, "pcn")
            assert np.array_equal(r["lag_bulk"], exp["lag_bulk"]) and np.array_equal(r["q"], exp["q"])
            assert np.allclose(r["ess_tail"], exp["ess_tail"], rtol=1e-9) and np.allclose(r["rhat"], exp["rhat"], rtol=1e-9)
        # a failing model in the middle: its code comes back, the context stays usable
        bad = ctx.upload(np.zeros((1, 2, 10)), "pcn")
        with pytest.raises(_ffi.McrError) as ei:
            ctx.summarize_models([ts[0], bad, ts[1]])
        assert ei.value.code == _ffi.MCR_EMINCHAINS
        bad.free()
        assert len(ctx.summarize_models(ts[:2])) == 2
    finally:
        for t in ts:
            t.free()
