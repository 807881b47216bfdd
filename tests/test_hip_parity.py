"""GPU parity tests: HIP path (through the C ABI) vs the CPU oracle and the committed goldens.

Gates (BASELINE.json north_star): integer / rank / index outputs bit-exact; floating-point
outputs within 1e-6 relative (we assert much tighter where the arithmetic allows).
"""
from __future__ import annotations

import math

import numpy as np
import pytest

from conftest import MODEL_NAMES, load_json, load_model

pytestmark = pytest.mark.gpu

REL = 1e-6          # north_star tolerance for floating-point outputs
TIGHT = 1e-9        # what fp64 on both sides actually delivers for these statistics


@pytest.fixture(scope="module")
def ctx():
    from mcmc_ref_hip import _ffi
    c = _ffi.Context(0)
    yield c
    c.close()


def close(a, b, rel, scale=0.0):
    a, b = float(a), float(b)
    if math.isnan(a) or math.isnan(b):
        return math.isnan(a) and math.isnan(b)
    if math.isinf(a) or math.isinf(b):
        return a == b
    return abs(a - b) <= rel * max(abs(a), abs(b)) + scale


def check_summary(got, exp, rel=TIGHT, what=""):
    P = len(exp["mean"])
    for p in range(P):
        sd = float(exp["std"][p])
        # a mean is a cancelling sum: its error scales with the spread, not with |mean|
        assert close(got["mean"][p], exp["mean"][p], rel, scale=rel * sd), (what, p, "mean")
        assert close(got["std"][p], exp["std"][p], rel), (what, p, "std", got["std"][p], exp["std"][p])
        assert close(got["median"][p], exp["median"][p], 0.0), (what, p, "median")
        for k in ("rhat", "rhat_bulk", "rhat_tail", "ess_bulk", "ess_tail"):
            assert close(got[k][p], exp[k][p], rel), (what, p, k, got[k][p], exp[k][p])
        assert int(got["lag_bulk"][p]) == int(exp["lag_bulk"][p]), (what, p, "lag_bulk")
        assert int(got["lag_tail"][p]) == int(exp["lag_tail"][p]), (what, p, "lag_tail")
    # quantiles are order statistics + one lerp: bit-exact
    assert np.array_equal(got["q"], exp["q"]), what


def test_version_and_device(ctx):
    assert ctx.lib.mcr_version() == 100
    assert ctx.lib.mcr_device_count() >= 1


@pytest.mark.parametrize("name", MODEL_NAMES)
def test_packaged_models_vs_oracle_and_goldens(ctx, oracle, name):
    draws, params, rec = load_model(name)
    got = ctx.summarize(draws, "pcn")
    exp = oracle.summarize(draws, "pcn")
    check_summary(got, exp, what=name)
    # the reference's own packaged goldens and the imported-reference vectors
    for i, p in enumerate(params):
        for k in ("rhat", "ess_bulk", "ess_tail"):
            assert close(got[k][i], rec["meta_diagnostics"][p][k], REL)
            assert close(got[k][i], rec["recomputed"][p][k], TIGHT)
        assert int(got["lag_bulk"][i]) == rec["recomputed"][p]["lag_bulk"]
        assert int(got["lag_tail"][i]) == rec["recomputed"][p]["lag_tail"]
        for b in ("arrow", "numpy"):
            st = rec["stats"][b][p]
            assert close(got["mean"][i], st["mean"], TIGHT, scale=TIGHT * st["std"])
            assert close(got["std"][i], st["std"], TIGHT)
            for j, qk in enumerate(("q5", "q50", "q95")):
                assert close(got["q"][i, j], st[qk], 1e-14)
        assert [float(v) for v in got["q"][i]] == [rec["stats"]["numpy"][p][qk] for qk in ("q5", "q50", "q95")]


def test_eight_schools_c0_four_chains(ctx, oracle):
    """BASELINE config 0: eight_schools chains 0-3 -> 4 x 1000 x 10."""
    draws, params, _ = load_model("eight_schools-eight_schools_noncentered")
    sub = np.ascontiguousarray(draws[:, :4, :])
    check_summary(ctx.summarize(sub, "pcn"), oracle.summarize(sub, "pcn"), what="C0")
    # the same tensor in Draws.to_numpy layout [C][N][P] and in f32
    cnp = np.ascontiguousarray(np.transpose(sub, (1, 2, 0)))
    check_summary(ctx.summarize(cnp, "cnp"), oracle.summarize(cnp, "cnp"), what="C0-cnp")
    f32 = sub.astype(np.float32)
    check_summary(ctx.summarize(f32, "pcn"), oracle.summarize(f32, "pcn"), what="C0-f32")


def test_synth_c1_subset_vs_reference_goldens(ctx, oracle):
    from mcmc_ref_hip import synth
    g = load_json("synth_c1_subset.json")
    x = synth.c1_model(g["C"], g["N"], g["P"], seed=g["seed"], params=g["params"])
    got = ctx.summarize(x, "pcn")
    check_summary(got, oracle.summarize(x, "pcn"), what="c1-subset")
    for i, p in enumerate(g["params"]):
        r = g["results"][str(p)]
        for k in ("rhat", "ess_bulk", "ess_tail", "rhat_bulk", "rhat_tail"):
            assert close(got[k][i], r[k], TIGHT), (p, k)
        assert (int(got["lag_bulk"][i]), int(got["lag_tail"][i])) == (r["lag_bulk"], r["lag_tail"])


UNIT = load_json("unit_vectors.json")


@pytest.mark.parametrize("name", [k for k in UNIT if not k.startswith("_")])
def test_unit_vectors_ragged_api(ctx, name):
    rec = UNIT[name]
    chains = rec["chains"]
    from mcmc_ref_hip._ffi import McrError, MCR_EMINCHAINS
    for key, mc in (("min4", 4), ("min1", 1)):
        exp = rec[key]
        if isinstance(exp["rhat"], dict):
            with pytest.raises(McrError) as ei:
                ctx.diagnose_chains(chains, mc)
            assert ei.value.code == MCR_EMINCHAINS
            continue
        got = ctx.diagnose_chains(chains, mc, debug=True)
        for k in ("rhat", "ess_bulk", "ess_tail"):
            assert close(got[k], exp[k], TIGHT), (name, key, k, got[k], exp[k])
        if "rhat_bulk" in exp:
            assert close(got["rhat_bulk"], exp["rhat_bulk"], TIGHT)
            assert close(got["rhat_tail"], exp["rhat_tail"], TIGHT)
            assert (got["lag_bulk"], got["lag_tail"]) == (exp["lag_bulk"], exp["lag_tail"]), (name, key)
        if "z" in rec and len(chains) >= 1:
            from scipy.stats import rankdata
            flat = np.concatenate([np.asarray(c, dtype=float) for c in chains])
            assert np.array_equal(np.concatenate(got["rank_bulk"]), rankdata(flat, method="average"))
            fold = np.concatenate([np.asarray(c, dtype=float) for c in rec["folded"]])
            assert np.array_equal(np.concatenate(got["rank_tail"]), rankdata(fold, method="average"))
            for a, b in zip(got["z_bulk"], rec["z"]):
                assert np.allclose(a, np.asarray(b), rtol=1e-13, atol=0)
            for a, b in zip(got["z_tail"], rec["z_folded"]):
                assert np.allclose(a, np.asarray(b), rtol=1e-13, atol=0)


def test_error_codes(ctx):
    from mcmc_ref_hip._ffi import McrError, MCR_EMINCHAINS, MCR_EMINCHAINS_ARG, MCR_ENONFINITE
    x = np.random.default_rng(0).normal(size=(2, 3, 50))
    with pytest.raises(McrError) as ei:
        ctx.summarize(x, "pcn", min_chains=4)
    assert ei.value.code == MCR_EMINCHAINS
    with pytest.raises(McrError) as ei:
        ctx.summarize(x, "pcn", min_chains=0)
    assert ei.value.code == MCR_EMINCHAINS_ARG
    x[1, 2, 7] = np.nan
    with pytest.raises(McrError) as ei:
        ctx.summarize(x, "pcn", min_chains=1)
    assert ei.value.code == MCR_ENONFINITE
    x[1, 2, 7] = np.inf
    with pytest.raises(McrError) as ei:
        ctx.summarize(x, "pcn", min_chains=1)
    assert ei.value.code == MCR_ENONFINITE
    # the context stays usable after an error
    x[1, 2, 7] = 0.5
    ctx.summarize(x, "pcn", min_chains=1)


def test_shapes_edge_cases(ctx, oracle):
    rng = np.random.default_rng(11)
    for (P, C, N) in [(1, 4, 2), (3, 4, 3), (2, 4, 17), (5, 2, 64), (1, 1, 10), (2, 10, 1), (1, 4, 4097),
                      (3, 5, 4096), (2, 4, 8193), (1, 3, 30001)]:
        x = rng.normal(loc=2.0, scale=3.0, size=(P, C, N))
        got = ctx.summarize(x, "pcn", min_chains=1)
        exp = oracle.summarize(x, "pcn", min_chains=1)
        check_summary(got, exp, what=f"{(P, C, N)}")
    # the order-statistics kernel: one wave per parameter (blocks of four), lane j = quantile j (32 at most), 64-ary
    # search for the fold split -- parameter counts off the block size, every lane busy, ties across the median
    qs = np.linspace(0.0, 1.0, 32)
    for (P, C, N) in [(7, 4, 16), (5, 2, 33), (9, 4, 1031), (6, 3, 4099)]:
        x = rng.normal(size=(P, C, N))
        x[1] = np.round(x[1])                                   # heavy ties, the median inside a long run
        x[2, :, : N // 2] = 0.5                                  # half of the draws equal
        got = ctx.summarize(x, "pcn", min_chains=1, quantiles=qs)
        exp = oracle.summarize(x, "pcn", min_chains=1, quantiles=qs)
        check_summary(got, exp, what=f"32 quantiles {(P, C, N)}")
    # empty tensors
    got = ctx.summarize(np.zeros((2, 4, 0)), "pcn")
    assert np.isnan(got["mean"]).all() and np.isnan(got["rhat"]).all() and np.isnan(got["ess_bulk"]).all()
    got = ctx.summarize(np.zeros((0, 4, 10)), "pcn")
    assert got["mean"].shape == (0,)


def test_heavy_ties_and_constant(ctx, oracle):
    rng = np.random.default_rng(5)
    x = np.empty((4, 4, 3000))
    x[0] = np.round(rng.normal(size=(4, 3000)), 1)            # ~60 distinct values, long tie runs
    x[1] = rng.integers(0, 2, size=(4, 3000)).astype(float)   # two values
    x[2] = 7.25                                               # constant: one run of length M
    x[3] = np.round(rng.normal(size=(4, 3000)), 3)
    x[3, 2] = 1.0                                             # one constant chain
    check_summary(ctx.summarize(x, "pcn"), oracle.summarize(x, "pcn"), what="ties")


def test_constant_halves_of_a_chain_that_is_not_constant(ctx, oracle):
    """Found by tests/manual/fuzz_long.py (seed 13, case 390): every draw equal but the LAST one of an odd-length chain,
    which `_split_chains` drops (diagnostics.py:76-85) -- all half-chains are constant although one chain is not.  The
    reference's within variance is then exactly 0 (`_variance` on equal values) and `_rhat` returns inf or 1.0 by whether
    the mean of the equal half means rounds (diagnostics.py:148-150); Q - S m is zero only up to rounding, so the
    kernels detect constant halves on the rank codes (k_acov_seg: SG_MIN0 .. SG_MAX1)."""
    x = np.full((3, 6, 5), -171.0)
    x[0, 2, 4] = -170.0                     # the dropped draw
    x[1, 0, 4] = -170.0; x[1, 5, 4] = -172.0
    x[2, 3, 1] = -170.0                     # a draw the split keeps: an ordinary parameter
    y = np.full((2, 4, 9), 2.5)
    y[0, 1, 8] = 7.0
    y[1, 1, 3] = 7.0
    for arr in (x, y):
        exp = oracle.summarize(arr, "pcn", min_chains=2)
        got = ctx.summarize(arr, "pcn", min_chains=2)
        for k in ("rhat", "rhat_bulk", "rhat_tail", "ess_bulk", "ess_tail"):
            for a, b in zip(got[k], exp[k]):
                assert close(a, b, TIGHT), (k, got[k], exp[k])
        assert np.array_equal(got["lag_bulk"], exp["lag_bulk"]) and np.array_equal(got["lag_tail"], exp["lag_tail"])
    assert np.isinf(oracle.summarize(x, "pcn", min_chains=2)["rhat"][0])


def test_basic_stats_and_compare(ctx, oracle):
    g = load_json("compare_cases.json")
    for rec in g["basic"]:
        out = ctx.basic_stats(rec["values"])
        exp = rec["out"]
        assert close(out["mean"], exp["mean"], 1e-12, scale=1e-12 * (exp["std"] if exp["std"] == exp["std"] else 0))
        assert close(out["std"], exp["std"], 1e-12)
    for rec in g["compare"]:
        for p, ms in rec["details"].items():
            names = list(ms)
            rel, ok = ctx.compare([ms[m]["ref"] for m in names], [ms[m]["actual"] for m in names], rec["tolerance"])
            for j, m in enumerate(names):
                assert close(rel[j], ms[m]["rel_error"], 0.0) and bool(ok[j]) == ms[m]["passed"]
    v = np.random.default_rng(3).normal(100.0, 1e-3, size=1_000_003)
    out, exp = ctx.basic_stats(v), oracle.basic_stats(v)
    assert close(out["mean"], exp["mean"], 1e-12) and close(out["std"], exp["std"], 1e-9)
    out32 = ctx.basic_stats(v.astype(np.float32))
    exp32 = oracle.basic_stats(v.astype(np.float32).astype(np.float64))
    assert close(out32["mean"], exp32["mean"], 1e-12) and close(out32["std"], exp32["std"], 1e-9)


def test_device_resident_async_and_profile(ctx, oracle):
    from mcmc_ref_hip import synth
    x = synth.c1_model(4, 2000, 6, seed=9)
    t = ctx.upload(x, "pcn")
    try:
        ctx.profile(True)
        ctx.profile_reset()
        bufs = [ctx.enqueue(t) for _ in range(3)]
        ctx.wait()
        exp = oracle.summarize(x, "pcn")
        for b in bufs:
            check_summary(b.result(), exp, what="async")
        prof = ctx.profile_get()
        assert prof["k_tile_sort"]["launches"] == 3 and prof["k_diag"]["total_ms"] > 0
        m, s = ctx.moments(t)
        assert np.allclose(m, exp["mean"], rtol=1e-12) and np.allclose(s, exp["std"], rtol=1e-10)
    finally:
        ctx.profile(False)
        t.free()


def test_workspace_chunking_gives_identical_results(ctx):
    from mcmc_ref_hip import synth
    x = synth.c1_model(4, 3000, 9, seed=3)
    full = ctx.summarize(x, "pcn")
    ctx._check(ctx.lib.mcr_set_workspace_limit(ctx.handle, 2 << 20))   # forces ~3 params per chunk
    try:
        chunked = ctx.summarize(x, "pcn")
    finally:
        ctx._check(ctx.lib.mcr_set_workspace_limit(ctx.handle, 8 << 30))
    for k in full:
        assert np.array_equal(full[k], chunked[k], equal_nan=True), k


def test_long_chains_merge_pass_path(ctx, oracle):
    """M = 4 x 70000 = 280k pooled draws: beyond the bucket path (69 tiles), so the pairwise merge
    passes, the standalone rank kernel and 35 autocovariance segments per chain are exercised
    (the geometry of BASELINE config 4, N = 100000)."""
    from mcmc_ref_hip import synth
    x = synth.c1_model(4, 70000, 3, seed=21)
    x[2] = np.round(x[2], 1)                       # heavy ties across tile and block edges
    got = ctx.summarize(x, "pcn")
    exp = oracle.summarize(x, "pcn")
    check_summary(got, exp, what="long")
    f32 = x[:2].astype(np.float32)
    cnp = np.ascontiguousarray(np.transpose(f32, (1, 2, 0)))          # [C][N][P] f32 through k_ingest_transpose
    check_summary(ctx.summarize(cnp, "cnp"), oracle.summarize(cnp, "cnp"), what="long-f32-cnp")


@pytest.mark.parametrize("C,N", [(4, 25000), (4, 50000), (4, 100000), (5, 100000), (2, 9000)])
def test_sample_ranking_by_merge_equals_pairwise_and_oracle(C, N, oracle, monkeypatch):
    """k_splitters ranks the regular samples of the sorted runs (every 64th order statistic).  Up to 1024 samples (pooled
    arrays of at most 16 tiles) pair by pair; more -- runs pre-merged to 8192 .. 32768 draws, 1 664 / 3 328 / 6 656 / 8 192
    samples here: every MVT of the kernel -- by merging the runs' sample lists in the LDS (round 4).  Both must cut the
    same buckets: bit-identical results, and equal to the oracle; f64 keys and f32 records; heavy ties; chains far apart /
    a random walk / sorted input (runs that hardly overlap: every cross rank is 0 or a whole run)."""
    from mcmc_ref_hip import _ffi
    rng = np.random.default_rng(C * N)
    x = rng.normal(size=(4, C, N))
    x[1] = np.round(x[1], 1)                                            # 80 distinct values: tie runs across every cut
    x[2] += 20.0 * np.arange(C)[:, None]                               # chains far apart: the runs' value ranges hardly overlap
    if C * N <= 100000:                                                 # (the oracle walks ~N/3 lags of these two: small shapes only)
        x[2] = np.cumsum(rng.normal(size=(C, N)), axis=1) * 0.01
        x[3] = np.sort(rng.normal(size=C * N)).reshape(C, N)            # sorted input: the runs are disjoint value ranges
    else:
        x[3] = np.floor(x[3] * 3.0)                                     # a dozen distinct values
    exp = oracle.summarize_mt(x, "pcn", min_chains=2)
    monkeypatch.setenv("MCR_SPLITTERS_PAIRWISE", "1")
    pair = _ffi.Context(0)
    monkeypatch.delenv("MCR_SPLITTERS_PAIRWISE")
    merge = _ffi.Context(0)
    try:
        for xx, tag in ((x, "f64"), (x.astype(np.float32), "f32")):
            e = exp if tag == "f64" else oracle.summarize_mt(xx, "pcn", min_chains=2)
            a = pair.summarize(xx, "pcn", min_chains=2)
            b = merge.summarize(xx, "pcn", min_chains=2)
            for k in a:
                assert np.array_equal(a[k], b[k], equal_nan=True), (tag, k)
            check_summary(b, e, what=f"splitters-{tag}-{C}x{N}")
    finally:
        pair.close(); merge.close()


def test_sample_ranking_modes_agree_on_many_parameters(monkeypatch):
    """The same comparison without the oracle on 600 parameters of the stress generator (4 x 100 000 f32, mu = p, seven
    scales, the f32 grid's ties): a merge whose threads' outputs straddled two pairs of runs went wrong on ~4 % of such
    parameters only (caught by the 16 GB test while this file was green), so the modes are held to each other in bulk."""
    from mcmc_ref_hip import _ffi
    monkeypatch.setenv("MCR_LANES", "1")
    monkeypatch.setenv("MCR_SPLITTERS_PAIRWISE", "1")
    pair = _ffi.Context(0)
    monkeypatch.delenv("MCR_SPLITTERS_PAIRWISE")
    merge = _ffi.Context(0)
    try:
        for (C, N, P, p0) in ((4, 100000, 600, 7900), (4, 50000, 300, 0), (3, 40000, 300, 4000)):
            t = merge.alloc_tensor(C, N, P, np.float32)
            merge.fill_synthetic(t, 4711, p0=p0)
            a = pair.summarize(_ffi.DeviceTensor(pair, t.buf, t.targs), min_chains=2)
            b = merge.summarize(t, min_chains=2)
            t.free()
            for k in a:
                assert np.array_equal(a[k], b[k], equal_nan=True), (C, N, k, np.flatnonzero(a[k] != b[k])[:10])
    finally:
        pair.close(); merge.close()


def test_lone_calls_fork_and_pipelined_calls_do_not_and_both_agree(monkeypatch):
    """A lone synchronous call runs the bulk half of the diagnostics' tiers 1 and 2 on the lane's second stream under the
    fold kernel (round 4); pipelined calls and MCR_FORK=0 keep one stream.  Same bits either way, in any interleaving."""
    from mcmc_ref_hip import _ffi, synth
    monkeypatch.setenv("MCR_FORK", "0")
    plain = _ffi.Context(0)
    monkeypatch.delenv("MCR_FORK")
    forked = _ffi.Context(0)
    rng = np.random.default_rng(4)
    try:
        shapes = [(4, 10000, 60), (10, 1000, 250), (4, 1000, 3), (2, 50, 5), (4, 30000, 20)]      # forks from 2 M param-draws up
        xs = [synth.c1_model(C, N, P, seed=5 + i) for i, (C, N, P) in enumerate(shapes)]
        xs.append(np.cumsum(rng.normal(size=(6, 4, 5000)), axis=2) * 0.01)          # tier 3 behind the join
        for x in xs:
            a = plain.summarize(x, "pcn", min_chains=2)
            t = forked.upload(x, "pcn")
            b = forked.summarize(t, min_chains=2)                                     # lone: forks
            bufs = [forked.enqueue(t, min_chains=2) for _ in range(5)]                # pipelined: the first forks, the rest do not
            forked.wait()
            c = forked.summarize(t, min_chains=2)                                     # lone again
            t.free()
            for r in [b, c] + [q.result() for q in bufs]:
                for k in a:
                    assert np.array_equal(a[k], r[k], equal_nan=True), (x.shape, k)
    finally:
        plain.close(); forked.close()


def test_sticky_chains_continuation_paths(ctx, oracle):
    """AR(1) with phi = 0.995: the first negative rho lies hundreds of lags out, so tier 2 (lags 64..255) and the
    first tier-3 round beyond it decide the truncation lag."""
    rng = np.random.default_rng(17)
    C, N = 4, 6000
    x = np.empty((2, C, N))
    for p, phi in enumerate((0.97, 0.995)):
        e = rng.normal(size=(C, N))
        for c in range(C):
            y = np.empty(N); y[0] = e[c, 0]
            for t in range(1, N):
                y[t] = phi * y[t - 1] + np.sqrt(1 - phi * phi) * e[c, t]
            x[p, c] = y
    got = ctx.summarize(x, "pcn")
    exp = oracle.summarize(x, "pcn")
    assert int(exp["lag_bulk"].max()) > 255, exp["lag_bulk"]
    check_summary(got, exp, what="sticky")


def test_long_lag_tier3_random_walks(ctx, oracle):
    """Truncation lags in the THOUSANDS: random walks and the provenance fake runner's ramps leave rho
    positive for a large fraction of the chain, so tiers 1 and 2 (lags < 256) decide nothing and the chip-wide
    rounds of k_acov_long / k_diag_long_scan do.  Lags exact, ESS to 1e-9; more listed
    pairs (2 x 14) than the 16 slots of a launch; and the wall time of the
    L ~ 6000 case, which one workgroup used to walk lag by lag (49 ms in round 1)."""
    import time
    rng = np.random.default_rng(5)
    C, N = 4, 20000
    x = np.cumsum(rng.normal(size=(14, C, N)), axis=2) * 0.01
    x[12] = rng.normal(size=(C, N))                                   # an easy parameter among the sticky ones
    x[13] = np.arange(C)[:, None] + 0.001 * np.arange(N)[None, :]     # generate.fake_jsonzip_runner's ramp
    exp = oracle.summarize(x, "pcn")
    got = ctx.summarize(x, "pcn")
    check_summary(got, exp, what="tier3")
    lags = np.concatenate([exp["lag_bulk"], exp["lag_tail"]])
    assert (lags > 4096).sum() >= 4 and ((lags > 256) & (lags < 4096)).sum() >= 2, lags
    one = np.ascontiguousarray(x[:1])
    t = ctx.upload(one, "pcn")
    try:
        ctx.summarize(t)
        t0 = time.perf_counter()
        r = ctx.summarize(t)
        ms = (time.perf_counter() - t0) * 1e3
    finally:
        t.free()
    assert int(r["lag_bulk"][0]) == int(exp["lag_bulk"][0])
    print(f"\nrandom walk 4 x {N}, truncation lag {int(r['lag_bulk'][0])} / {int(r['lag_tail'][0])}: {ms:.2f} ms per call")
    assert ms < 10.0, ms
    # ragged chains through mcr_diagnose_chains: n = the shortest chain
    chains = [list(np.cumsum(rng.normal(size=n)) * 0.01) for n in (5000, 4100, 6000, 4500)]
    g, e = ctx.diagnose_chains(chains, 2), oracle.diag(chains, 2)
    assert (g["lag_bulk"], g["lag_tail"]) == (e["lag_bulk"], e["lag_tail"]) and e["lag_bulk"] > 224
    for k in ("rhat", "ess_bulk", "ess_tail"):
        assert close(g[k], e[k], TIGHT), (k, g[k], e[k])


def test_long_chains_fft_tier_equals_direct_tier_and_oracle(ctx, oracle, monkeypatch):
    """Chains of more than 16 384 draws take their long lags from the FFT tier (mcr_fft.hpp: zero-padded four-step
    FFT, |.|^2 summed over chains, second FFT); MCR_FFT=0 keeps them on the direct products.  Both must give the
    oracle's integer truncation lags exactly and its ESS to 1e-9: random walks (lags in the thousands and tens of
    thousands), more listed pairs than the FFT has slots (the rest fall through to the direct rounds), N not a power
    of two, two chains, ragged chains."""
    import time
    from mcmc_ref_hip import _ffi
    monkeypatch.setenv("MCR_FFT", "0")
    direct = _ffi.Context(0)
    monkeypatch.delenv("MCR_FFT")
    rng = np.random.default_rng(8)
    try:
        x = np.cumsum(rng.normal(size=(18, 4, 17001)), axis=2) * 0.01           # 36 listed pairs > 32 FFT slots (N = 2^16)
        x[5] = rng.normal(size=(4, 17001))
        exp = oracle.summarize(x, "pcn")
        assert int(min(exp["lag_bulk"][0], exp["lag_tail"][0])) > 256
        for c, name in ((ctx, "fft"), (direct, "direct")):
            check_summary(c.summarize(x, "pcn"), exp, what=f"long-{name}")
        y = np.cumsum(rng.normal(size=(1, 2, 60000)), axis=2) * 0.01             # N = 2^17, two chains
        expy = oracle.summarize(y, "pcn", min_chains=2)
        t = ctx.upload(y, "pcn"); td = direct.upload(y, "pcn")
        try:
            for c, tt, name in ((ctx, t, "fft"), (direct, td, "direct")):
                c.summarize(tt, min_chains=2)
                t0 = time.perf_counter()
                got = c.summarize(tt, min_chains=2)
                ms = (time.perf_counter() - t0) * 1e3
                check_summary(got, expy, what=f"long2-{name}")
                print(f"\nrandom walk 2 x 60000, lags {int(got['lag_bulk'][0])} / {int(got['lag_tail'][0])}, {name}: {ms:.2f} ms per call")
        finally:
            t.free(); td.free()
        chains = [list(np.cumsum(rng.normal(size=n)) * 0.01) for n in (21000, 17500, 30000)]     # ragged: n = 17500
        e = oracle.diag(chains, 2)
        for c in (ctx, direct):
            g = c.diagnose_chains(chains, 2)
            assert (g["lag_bulk"], g["lag_tail"]) == (e["lag_bulk"], e["lag_tail"])
            for k in ("rhat", "ess_bulk", "ess_tail"):
                assert close(g[k], e[k], TIGHT), (k, g[k], e[k])
    finally:
        direct.close()


def test_moments_with_an_outlying_first_draw(ctx, oracle):
    """the moments used to be shifted by the parameter's FIRST draw in a single pass, so an unconverged
    start (first draw d sigma away) cost d^2 eps of accuracy in std.  Now every tile / slice is two-pass around its
    own mean and the pieces are merged with Chan's update: the reference's two-pass result (compare.py:58-64,
    np.std) to 1e-12 whatever the first draw is."""
    rng = np.random.default_rng(12)
    x = rng.normal(size=(4, 4, 6000))
    x[0, 0, 0] = 1e9                      # a single wild first draw
    x[1, :, :200] += 1e7                  # an unconverged start of every chain
    x[2] = x[2] * 1e-6 + 3.0              # small spread around an offset
    x[3, 0, 0] = -1e12
    exp = oracle.summarize(x, "pcn")
    got = ctx.summarize(x, "pcn")
    for p in range(4):
        assert close(got["std"][p], exp["std"][p], 1e-12), (p, got["std"][p], exp["std"][p])
        assert close(got["mean"][p], exp["mean"][p], 1e-12, scale=1e-12 * exp["std"][p]), p
    t = ctx.upload(x, "pcn")
    try:
        mean, std = ctx.moments(t)                              # the streaming one-pass kernel (slice pivots + Chan)
    finally:
        t.free()
    for p in range(4):
        assert close(std[p], exp["std"][p], 1e-12), (p, std[p], exp["std"][p])
    for p in range(4):
        b = ctx.basic_stats(x[p].reshape(-1))
        e = oracle.basic_stats(x[p].reshape(-1)) if hasattr(oracle, "basic_stats") else {"std": exp["std"][p]}
        assert close(b["std"], e["std"], 1e-12), (p, b, e)
    cnp = np.ascontiguousarray(np.transpose(x, (1, 2, 0)))      # strided variant (k_moments_cols)
    t = ctx.upload(cnp, "cnp")
    try:
        _, std2 = ctx.moments(t)
    finally:
        t.free()
    for p in range(4):
        assert close(std2[p], exp["std"][p], 1e-12), (p, std2[p])


def test_f32_records_equal_the_widened_path(ctx, oracle, monkeypatch):
    """f32 tensors in the Arrow layout are sorted as packed (key, position) records (mcr_sort32.hpp, VERDICT r1 item
    4).  Widening f32 -> f64 keeps order and equality, so EVERY output must be bit-identical to the f64 kernels run on
    the same draws (MCR_F32_RECORDS=0), and equal to the oracle: one tile, many tiles, the pre-merged long path
    (N = 100000 geometry), heavy ties, +-0, denormals, huge magnitudes."""
    from mcmc_ref_hip import _ffi
    monkeypatch.setenv("MCR_F32_RECORDS", "0")
    wide = _ffi.Context(0)
    monkeypatch.delenv("MCR_F32_RECORDS")
    rng = np.random.default_rng(32)
    cases = []
    for C, N, P in ((4, 500, 3), (4, 1000, 5), (4, 10000, 4), (7, 4097, 2), (4, 16384, 2), (4, 100000, 2), (2, 131071, 1)):
        x = (rng.normal(size=(P, C, N)) * 10.0 ** rng.integers(-3, 3, size=(P, 1, 1)) + rng.normal(size=(P, 1, 1)) * 5).astype(np.float32)
        cases.append(x)
    t = np.round(rng.normal(size=(3, 4, 9000)), 1).astype(np.float32)             # heavy ties
    t[1, :, ::3] = 0.0; t[1, :, 1::3] = -0.0                                       # +-0 tie with each other
    t[2] = (rng.integers(-3, 4, size=(4, 9000)) * 1e-42).astype(np.float32)        # denormals, few distinct values
    cases.append(t)
    big = (rng.normal(size=(2, 4, 5000)) * 1e30).astype(np.float32)
    cases.append(big)
    try:
        for x in cases:
            got = ctx.summarize(x, "pcn", min_chains=2)
            ref = wide.summarize(x, "pcn", min_chains=2)
            for k in got:
                assert np.array_equal(got[k], ref[k], equal_nan=True), (x.shape, k)
            if x.shape[1] * x.shape[2] <= 70000:
                check_summary(got, oracle.summarize(x, "pcn", min_chains=2), what=f"f32rec {x.shape}")
        bad = cases[2].copy(); bad[1, 2, 777] = np.nan
        for c in (ctx, wide):
            with pytest.raises(_ffi.McrError) as e:
                c.summarize(bad, "pcn")
            assert e.value.code == _ffi.MCR_ENONFINITE
    finally:
        wide.close()


def test_random_shapes_fuzz(ctx, oracle):
    """Seeded fuzz over shapes that cross every internal boundary: tile (4096) and bucket edges,
    1..16 tiles, segment (2048) edges, odd N, tie-heavy and constant columns, both layouts, f32."""
    rng = np.random.default_rng(2026)
    n_choices = [2, 3, 5, 63, 64, 65, 127, 511, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097, 6000, 9001]
    for it in range(48):
        C = int(rng.integers(2, 9))
        N = int(rng.choice(n_choices))
        if C * N > 65000 and it % 4:
            N = 65000 // C
        P = int(rng.integers(1, 5))
        kind = it % 6
        x = rng.normal(size=(P, C, N)) * 10.0 ** rng.integers(-3, 4) + rng.normal() * 100.0
        if kind == 1:
            x = np.round(x, 0)                                   # few distinct values
        elif kind == 2:
            x[0] = 3.25                                          # constant parameter
        elif kind == 3:
            x = np.cumsum(rng.normal(size=(P, C, N)), axis=2) * 0.05     # random walks: long lags
        elif kind == 4 and C > 2:
            x[:, 0, :] += 5.0                                    # one shifted chain
        layout = "pcn"
        arr = x
        if it % 5 == 0:
            arr = np.ascontiguousarray(np.transpose(x, (1, 2, 0))); layout = "cnp"
        if it % 7 == 0:
            arr = arr.astype(np.float32)
        got = ctx.summarize(arr, layout, min_chains=2)
        exp = oracle.summarize(arr, layout, min_chains=2)
        check_summary(got, exp, what=f"fuzz{it} C={C} N={N} P={P} kind={kind} {layout} {arr.dtype}")


def test_ragged_fuzz(ctx, oracle):
    rng = np.random.default_rng(77)
    for it in range(24):
        C = int(rng.integers(2, 7))
        lens = [int(rng.integers(1, 700)) for _ in range(C)]
        if it % 3 == 0:
            lens[int(rng.integers(0, C))] = int(rng.integers(2050, 5000))     # second-half range beyond min length
        chains = [list(np.round(rng.normal(size=n), 2 if it % 2 else 8)) for n in lens]
        got = ctx.diagnose_chains(chains, 2)
        exp = oracle.diag(chains, 2)
        for k in ("rhat", "rhat_bulk", "rhat_tail", "ess_bulk", "ess_tail"):
            assert close(got[k], exp[k], TIGHT), (it, lens, k, got[k], exp[k])
        assert (got["lag_bulk"], got["lag_tail"]) == (exp["lag_bulk"], exp["lag_tail"]), (it, lens)


def test_rolling_window_wait_one(ctx, oracle):
    """mcr_summarize_wait_one: a rolling window of outstanding calls over two lanes delivers every
    result, in order, without draining the device."""
    from mcmc_ref_hip import _ffi, synth
    xs = [synth.c1_model(4, 1500 + 100 * i, 3, seed=40 + i) for i in range(3)]
    ts = [ctx.upload(x, "pcn") for x in xs]
    exps = [oracle.summarize(x, "pcn") for x in xs]
    done = []
    try:
        for k in range(20):
            if ctx.inflight >= _ffi.MCR_MAX_INFLIGHT:
                done.append(ctx.wait_one())
            ctx.enqueue(ts[k % 3])
        while ctx.inflight:
            done.append(ctx.wait_one())
        assert len(done) == 20 and ctx.wait_one() is None
        for k, b in enumerate(done):
            check_summary(b.result(), exps[k % 3], what=f"rolling{k}")
    finally:
        for t in ts:
            t.free()


def test_inflight_limit_and_context_reuse(ctx, oracle):
    """More than MCR_MAX_INFLIGHT outstanding calls is an argument error, not a hang; the context
    keeps working afterwards and shapes may change freely between calls (graph cache, lanes)."""
    from mcmc_ref_hip import _ffi, synth
    x = synth.c1_model(4, 600, 2, seed=5)
    t = ctx.upload(x, "pcn")
    try:
        for _ in range(_ffi.MCR_MAX_INFLIGHT):
            ctx.enqueue(t)
        with pytest.raises(_ffi.McrError) as ei:
            ctx.enqueue(t)
        assert ei.value.code == _ffi.MCR_EINVAL
        ctx.wait()
        exp = oracle.summarize(x, "pcn")
        for shape in [(4, 600, 2), (4, 300, 2), (4, 600, 1), (4, 600, 2)]:
            C, N, P = shape
            y = np.ascontiguousarray(x[:P, :, :N])
            check_summary(ctx.summarize(y, "pcn"), oracle.summarize(y, "pcn"), what=f"reuse{shape}")
        check_summary(ctx.summarize(t), exp, what="reuse-dev")
    finally:
        t.free()


def _bits_equal(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.dtype == np.float64:
        return np.array_equal(a.view(np.int64), b.view(np.int64))
    return np.array_equal(a, b)


@pytest.mark.parametrize("shape,dtype", [((4, 10000, 100), np.float64),      # BASELINE config 1, full size
                                         ((4, 100000, 24), np.float32)])     # config 4 geometry (N = 100000), P slice
def test_full_size_properties(ctx, shape, dtype):
    """Size-independent properties at the sizes the oracle is too slow for:
      * scaling the draws by 2 is exact in floating point and keeps every order, so the rank-based diagnostics
        (rhat*, ess*, truncation lags) must come back BIT-identical and mean / std / quantiles exactly doubled;
      * parameters are independent: permuting them permutes the results bit for bit (what sharding relies on);
      * a chunked workspace gives the same bits as one pass; ESS <= C*N; quantiles are ordered;
      * the moments agree with the generator (iid N(p, sigma_p), sigma_p = 10^((p mod 7) - 3))."""
    C_, N, P = shape
    t = ctx.alloc_tensor(C_, N, P, dtype)
    ctx.fill_synthetic(t, 4711)
    host = t.buf.download(dtype, C_ * N * P).reshape(P, C_, N)
    base = ctx.summarize(t)
    M = C_ * N
    sig = 10.0 ** ((np.arange(P) % 7) - 3)
    assert np.all(np.abs(base["mean"] - np.arange(P)) < 6 * sig / np.sqrt(M))
    assert np.all(np.abs(base["std"] / sig - 1) < 0.02)
    assert np.all(base["ess_bulk"] <= M) and np.all(base["ess_tail"] <= M) and np.all(base["ess_bulk"] > 0.5 * M)
    assert np.all(np.abs(base["rhat"] - 1) < 0.01)
    assert np.all(np.diff(base["q"], axis=1) > 0) and _bits_equal(base["q"][:, 1], base["median"])
    # x -> 2x
    dbl = ctx.summarize(host * dtype(2), "pcn")
    for k in ("rhat", "rhat_bulk", "rhat_tail", "ess_bulk", "ess_tail", "lag_bulk", "lag_tail"):
        assert _bits_equal(dbl[k], base[k]), k
    for k in ("mean", "std", "q", "median"):
        assert _bits_equal(dbl[k], 2 * base[k]), k
    # parameter permutation (and the host-upload path against the device-resident one)
    perm = np.random.default_rng(0).permutation(P)
    pr = ctx.summarize(np.ascontiguousarray(host[perm]), "pcn")
    for k in ("mean", "std", "q", "median", "rhat", "ess_bulk", "ess_tail", "lag_bulk", "lag_tail"):
        assert _bits_equal(pr[k], base[k][perm]), k
    # chunked workspace
    ctx._check(ctx.lib.mcr_set_workspace_limit(ctx.handle, 96 << 20))
    try:
        ch = ctx.summarize(t)
    finally:
        ctx._check(ctx.lib.mcr_set_workspace_limit(ctx.handle, 8 << 30))
    for k in ("mean", "std", "q", "rhat", "ess_bulk", "ess_tail", "lag_bulk", "lag_tail"):
        assert _bits_equal(ch[k], base[k]), k
    t.free()


def test_extreme_shapes_and_scales(ctx, oracle):
    """Grid and value-range extremes: more parameters than one grid dimension holds, the chain-count limit, one
    long pooled array with heavy ties, slow-mixing random walks (truncation lag in the thousands: the direct
    continuation loop), denormal- and overflow-scale draws (std overflows to inf exactly like np.std)."""
    rng = np.random.default_rng(0)
    cases = {
        "P=70000": rng.normal(size=(70000, 2, 8)),
        "C=256": rng.normal(size=(3, 256, 40)),
        "C=1": rng.normal(size=(2, 1, 5000)),
        "ties-1.2M": np.round(rng.normal(size=(1, 4, 300000)), 2),
        "random-walk": np.cumsum(rng.normal(size=(2, 4, 20000)), axis=2),
        "denormal": rng.normal(size=(4, 4, 4096)) * 1e-300,
        "overflow": rng.normal(size=(4, 4, 4096)) * 1e300,
    }
    for what, x in cases.items():
        got = ctx.summarize(x, "pcn", min_chains=1)
        exp = oracle.summarize(x, "pcn", min_chains=1)
        if what == "overflow":
            assert np.all(np.isinf(got["std"])) and np.all(np.isinf(exp["std"]))
            got["std"][:] = exp["std"][:] = 1e300          # check_summary scales the mean tolerance by std
        check_summary(got, exp, what=what)
    assert cases["random-walk"].shape and int(ctx.summarize(cases["random-walk"], "pcn")["lag_bulk"].min()) > 1000


def test_host_upload_in_pieces(ctx, oracle):
    """Host tensors of >= 8 MB in parameter-major layout are uploaded in pieces that overlap the kernels of the
    previous piece: same bits as the device-resident path, errors still reported, strided layouts untouched."""
    from mcmc_ref_hip._ffi import McrError, MCR_ENONFINITE
    from mcmc_ref_hip import synth
    x = synth.c1_model(4, 10000, 37, seed=5)                         # 11.8 MB: 4 pieces of 9-10 parameters
    got = ctx.summarize(x, "pcn")
    t = ctx.upload(x, "pcn")
    dev = ctx.summarize(t)
    t.free()
    for k in ("mean", "std", "q", "median", "rhat", "rhat_bulk", "rhat_tail", "ess_bulk", "ess_tail", "lag_bulk", "lag_tail", "q_lo"):
        assert _bits_equal(got[k], dev[k]), k
    check_summary({k: v[:3] for k, v in got.items() if k != "q_lo"},
                  oracle.summarize(x[:3], "pcn"), what="pieces")
    st = ctx.summarize(x, "pcn", diagnostics=False, quantiles=(0.1, 0.9))
    assert _bits_equal(st["mean"], dev["mean"]) and np.isnan(st["rhat"]).all() and st["q"].shape == (37, 2)
    for poison in (np.nan, np.inf, -np.inf):                         # non-finite draws in a multi-tile parameter:
        bad = x.copy()                                               # rejected, and no kernel walks a broken partition
        bad[30, 2, 777] = poison                                     # lands in the last piece
        bad[31, :, ::7] = poison
        with pytest.raises(McrError) as ei:
            ctx.summarize(bad, "pcn")
        assert ei.value.code == MCR_ENONFINITE
        t = ctx.upload(bad[28:34], "pcn")
        with pytest.raises(McrError) as ei:
            ctx.summarize(t)
        assert ei.value.code == MCR_ENONFINITE
        t.free()
    got3 = ctx.summarize(x, "pcn")                                   # the context is healthy afterwards
    assert _bits_equal(got3["ess_bulk"], dev["ess_bulk"])
    cnp = np.ascontiguousarray(np.transpose(x, (1, 2, 0)))           # [C][N][P]: not separable, single upload
    got2 = ctx.summarize(cnp, "cnp")
    for k in ("mean", "rhat", "ess_bulk", "lag_bulk"):
        assert _bits_equal(got2[k], dev[k]), k
    assert ctx.inflight == 0


def test_nonfinite_in_long_ragged_chains(ctx):
    """mcr_diagnose_chains with NaN / Inf in chains that span several sort tiles (and unequal lengths)."""
    from mcmc_ref_hip._ffi import McrError, MCR_ENONFINITE
    rng = np.random.default_rng(8)
    for poison in (np.nan, np.inf):
        chains = [rng.normal(size=n) for n in (30000, 25000, 30000, 28000)]
        chains[2][12345] = poison
        with pytest.raises(McrError) as ei:
            ctx.diagnose_chains(chains, min_chains=4)
        assert ei.value.code == MCR_ENONFINITE
    ok = ctx.diagnose_chains([rng.normal(size=n) for n in (30000, 25000, 30000, 28000)], min_chains=4)
    assert abs(ok["rhat"] - 1.0) < 0.01


def test_hipgraph_replay_path(oracle, monkeypatch):
    """MCR_GRAPH=1 (capture once per shape / buffers / slot, then replay) gives the same bits as direct launches."""
    from mcmc_ref_hip._ffi import Context
    from mcmc_ref_hip import synth
    x = synth.c1_model(4, 3000, 6, seed=17)
    y = synth.c1_model(4, 700, 3, seed=18)
    with Context(0) as plain:
        a = [plain.summarize(x, "pcn"), plain.summarize(y, "pcn")]
    monkeypatch.setenv("MCR_GRAPH", "1")
    with Context(0) as g:
        tx, ty = g.upload(x, "pcn"), g.upload(y, "pcn")
        for _ in range(3):                                   # first round captures, later rounds replay
            b = [g.summarize(tx), g.summarize(ty)]
            for got, exp in zip(b, a):
                for k in ("mean", "std", "q", "rhat", "ess_bulk", "ess_tail", "lag_bulk", "lag_tail"):
                    assert _bits_equal(got[k], exp[k]), k
        bufs = [g.enqueue(tx) for _ in range(6)]
        g.wait()
        assert all(_bits_equal(bb.result()["ess_bulk"], a[0]["ess_bulk"]) for bb in bufs)
        tx.free(); ty.free()


def test_two_contexts_in_two_threads(oracle):
    """One Context per thread (the documented threading model): concurrent calls do not disturb each other."""
    import threading
    from mcmc_ref_hip._ffi import Context
    from mcmc_ref_hip import synth
    xs = [synth.c1_model(4, 2000 + 500 * k, 5 + k, seed=30 + k) for k in range(2)]
    exp = [oracle.summarize(x, "pcn") for x in xs]
    out, errs = [None, None], []

    def work(k):
        try:
            with Context(0) as c:
                for _ in range(20):
                    out[k] = c.summarize(xs[k], "pcn")
        except Exception as exc:          # pragma: no cover
            errs.append(exc)

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for k in range(2):
        check_summary(out[k], exp[k], what=f"thread {k}")
