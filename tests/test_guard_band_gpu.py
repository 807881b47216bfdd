"""Guard band of the ESS walk in all three tiers + the order of the tier-3 list.

Spec: src/mcmc_ref/diagnostics.py:154-193 -- `_ess` walks the lags and stops at the first `rho < 0`.  For chains of more
than 16 384 draws the long lags come from FFTs (csrc/mcr_fft.hpp), whose rho carries ~1e-14 of round-off; the scan
(k_diag_long_scan, csrc/mcr_diag.hpp) therefore decides no lag whose rho lies within 1e-10 of zero on that value: it
re-derives the lag with _autocorr's own left-to-right sums first.  tests/golden/band_cases.json holds square-wave
chains whose rho is exactly zero in exact arithmetic at the deciding lag, with what the IMPORTED REFERENCE returned
(both outcomes occur: its rounding makes that rho +tiny in some cases, -tiny in others).  `low_cases` (round 4) are the
same construction with the zero lag at 4 ... 250: lags the segment records of k_acov_seg and the mean correction of
k_diag_combine / k_diag_combine2 produce (tiers 1 and 2), which take the same re-derivation (ref_cov_sum)."""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"


@pytest.fixture(scope="module")
def ctx():
    from mcmc_ref_hip import _ffi
    c = _ffi.Context(0)
    yield c
    c.close()


def square_wave(n: int, T: int, shift: int) -> np.ndarray:      # as in tests/golden/make_band_fixture.py
    i = (np.arange(n) + shift) % T
    return np.where(i < T // 2, 1.0, -1.0)


def _cases(key="cases"):
    return json.loads((GOLD / "band_cases.json").read_text())[key]


def test_tiers_one_and_two_decide_a_zero_rho_as_the_reference_does(ctx, oracle):
    """74 square-wave chain sets (C in {2, 4}, n in {1000 ... 10 000}) whose rho is exactly zero in exact arithmetic at a
    lag between 4 and 250, run through the imported reference (53 keep the zero lag, 21 stop one short of it): the GPU's
    integer term count equals the reference's on every one, ESS to 1e-12, and each case went through the re-derivation."""
    cases = _cases("low_cases")
    assert len(cases) >= 60
    lags = sorted(c["zero_lag"] for c in cases)
    assert lags[0] < 8 and any(64 <= l < 256 for l in lags) and lags[-1] < 256          # both tiers
    outcomes = set()
    for c in cases:
        x = np.stack([square_wave(c["n"], c["T"], s) for s in c["shifts"]])[None]
        exp = oracle.summarize(x, "pcn", min_chains=2)
        assert int(exp["lag_bulk"][0]) == c["ref_terms"] and float(exp["ess_bulk"][0]) == c["ref_ess_bulk"]
        outcomes.add(c["ref_terms"] - c["zero_lag"])
        before = ctx.rho_guard_count()
        got = ctx.summarize(x, "pcn", min_chains=2)
        assert int(got["lag_bulk"][0]) == c["ref_terms"], (c, int(got["lag_bulk"][0]))
        assert abs(got["ess_bulk"][0] - c["ref_ess_bulk"]) <= 1e-12 * c["ref_ess_bulk"], c
        assert got["ess_tail"][0] == c["C"] * c["n"] and int(got["lag_tail"][0]) == 0        # |x - med| is constant
        assert ctx.rho_guard_count() > before, c
    assert outcomes == {0, -1}


def test_tiers_one_and_two_without_the_band(ctx, monkeypatch):
    """MCR_RHO_BAND=0: the segment-record rho decides on its own rounding.  Reported, not asserted (beyond +-1)."""
    from mcmc_ref_hip import _ffi
    monkeypatch.setenv("MCR_RHO_BAND", "0")
    raw = _ffi.Context(0)
    monkeypatch.delenv("MCR_RHO_BAND")
    try:
        cases = _cases("low_cases")
        off = 0
        for c in cases:
            x = np.stack([square_wave(c["n"], c["T"], s) for s in c["shifts"]])[None]
            got = raw.summarize(x, "pcn", min_chains=2)
            assert abs(int(got["lag_bulk"][0]) - c["ref_terms"]) <= 1
            off += int(got["lag_bulk"][0]) != c["ref_terms"]
        assert raw.rho_guard_count() == 0
        print(f"\nwithout the guard band {off} of {len(cases)} tier-1/2 truncation lags differ from the reference's")
    finally:
        raw.close()


def test_a_batch_of_band_cases_in_one_call(ctx):
    """Many parameters of one call inside the band at once (every workgroup of k_diag_combine / combine2 re-deriving):
    the cases that share (C, n) stacked into one tensor give the same answers as one by one."""
    cases = [c for c in _cases("low_cases") if c["C"] == 4 and c["n"] == 4000]
    assert len(cases) >= 8
    x = np.stack([np.stack([square_wave(c["n"], c["T"], s) for s in c["shifts"]]) for c in cases])
    got = ctx.summarize(x, "pcn")
    assert [int(v) for v in got["lag_bulk"]] == [c["ref_terms"] for c in cases]
    assert np.allclose(got["ess_bulk"], [c["ref_ess_bulk"] for c in cases], rtol=1e-12, atol=0)


def test_tier3_lists_do_not_depend_on_when_a_workgroup_starts(ctx, oracle):
    """k_tier3's workgroups each compact the list of undecided pairs, and a scan that finished early used to
    take its pair off the list a late workgroup saw.  Several calls in flight on the lanes (staggered workgroup starts),
    each with many listed pairs of very different truncation lags against two slots per launch; every call must equal
    the oracle."""
    rng = np.random.default_rng(77)
    xs = []
    for k in range(6):
        x = np.cumsum(rng.normal(size=(10, 4, 3000 + 500 * k)), axis=2) * 0.01
        x[2 * (k % 3)] = rng.normal(size=x.shape[1:])               # some pairs decided at lag 1
        x[5, :, : 1500] = rng.normal(size=(4, 1500)) * 3.0            # short memory: a scan that ends early
        xs.append(x)
    exps = [oracle.summarize(x, "pcn") for x in xs]
    assert sum(int((e["lag_bulk"] > 256).sum() + (e["lag_tail"] > 256).sum()) for e in exps) >= 40
    ts = [ctx.upload(x, "pcn") for x in xs]
    try:
        for _ in range(3):
            bufs = [ctx.enqueue(t) for t in ts]
            ctx.wait()
            for b, e in zip(bufs, exps):
                g = b.result()
                for k in ("lag_bulk", "lag_tail"):
                    assert np.array_equal(g[k], e[k]), k
                for k in ("ess_bulk", "ess_tail", "rhat"):
                    assert np.allclose(g[k], e[k], rtol=1e-9, atol=0), k
    finally:
        for t in ts:
            t.free()


def test_fft_direct_and_reference_agree_where_rho_is_zero(ctx, oracle, monkeypatch):
    """FFT tier == direct tier == the reference on the integer truncation lag (and ESS to 1e-12) for every case, and the
    guard did run: each case has at least one lag inside the band."""
    from mcmc_ref_hip import _ffi
    monkeypatch.setenv("MCR_FFT", "0")
    direct = _ffi.Context(0)
    monkeypatch.delenv("MCR_FFT")
    try:
        outcomes = set()
        for c in _cases():
            x = np.stack([square_wave(c["n"], c["T"], s) for s in c["shifts"]])[None]      # [1][C][n]
            assert c["n"] > 16384 and c["zero_lag"] >= 300
            exp = oracle.summarize(x, "pcn", min_chains=2)
            assert int(exp["lag_bulk"][0]) == c["ref_terms"] and float(exp["ess_bulk"][0]) == c["ref_ess_bulk"]
            outcomes.add(c["ref_terms"] - c["zero_lag"])
            for cx, name in ((ctx, "fft"), (direct, "direct")):
                before = cx.rho_guard_count()
                got = cx.summarize(x, "pcn", min_chains=2)
                assert int(got["lag_bulk"][0]) == c["ref_terms"], (name, c, int(got["lag_bulk"][0]))
                assert abs(got["ess_bulk"][0] - c["ref_ess_bulk"]) <= 1e-12 * c["ref_ess_bulk"], (name, c)
                assert got["ess_tail"][0] == c["C"] * c["n"] and int(got["lag_tail"][0]) == 0    # |x - med| is constant
                assert cx.rho_guard_count() > before, (name, c)
        assert outcomes == {0, -1}            # the reference's rounding went both ways across the cases
    finally:
        direct.close()


def test_without_the_band_the_engines_are_on_their_own(ctx, monkeypatch):
    """MCR_RHO_BAND=0 switches the re-derivation off: the lag is then decided on the FFT's / the tree sums' own value of
    a rho that is zero up to round-off.  Nothing is asserted about which way each engine falls -- only that the guard is
    what the agreement above rests on (the counter stays put) and that the lag stays within one of the reference's."""
    from mcmc_ref_hip import _ffi
    monkeypatch.setenv("MCR_RHO_BAND", "0")
    raw = _ffi.Context(0)
    monkeypatch.delenv("MCR_RHO_BAND")
    try:
        off = 0
        for c in _cases():
            x = np.stack([square_wave(c["n"], c["T"], s) for s in c["shifts"]])[None]
            got = raw.summarize(x, "pcn", min_chains=2)
            assert abs(int(got["lag_bulk"][0]) - c["ref_terms"]) <= 1
            off += int(got["lag_bulk"][0]) != c["ref_terms"]
        assert raw.rho_guard_count() == 0
        print(f"\nwithout the guard band {off} of {len(_cases())} truncation lags differ from the reference's")
    finally:
        raw.close()


def test_tier3_list_order_is_a_function_of_the_data(ctx):
    """36 listed pairs against 32 FFT slots (4 x 17 001 random walks): which pairs the FFT serves must not depend on
    the order in which workgroups finished -- the list is built in ascending pair order -- so repeated calls return
    the same bits, and so does a context that works through the parameters in two chunks."""
    rng = np.random.default_rng(8)
    x = np.cumsum(rng.normal(size=(18, 4, 17001)), axis=2) * 0.01
    t = ctx.upload(x, "pcn")
    try:
        first = ctx.summarize(t)
        assert int(first["lag_bulk"].min()) > 256
        for _ in range(5):
            again = ctx.summarize(t)
            for k in ("ess_bulk", "ess_tail", "rhat", "lag_bulk", "lag_tail"):
                assert np.array_equal(first[k], again[k]), k
    finally:
        t.free()


def test_tier3_single_launch_and_listed_routes_agree_with_the_oracle(ctx, oracle):
    """Chains of at most 16 384 draws take tier 3 in ONE launch (k_tier3: the list compacted into every workgroup's LDS,
    256-lag groups, the pair's last group runs the scan) when a chunk holds at most 2 048 pairs, and the listed route
    (k_long_list + k_dev_fill + k_acov_long + k_diag_long_scan) beyond that.  Random walks put many pairs on the list --
    more than the two slots of a launch -- with truncation lags on both sides of 256."""
    rng = np.random.default_rng(21)
    x = np.cumsum(rng.normal(size=(12, 4, 6000)), axis=2) * 0.01            # 24 pairs, k_tier3
    x[3] = rng.normal(size=(4, 6000))
    exp = oracle.summarize(x, "pcn")
    got = ctx.summarize(x, "pcn")
    lags = np.concatenate([exp["lag_bulk"], exp["lag_tail"]])
    assert (lags > 256).sum() >= 12 and (lags < 64).sum() >= 2
    for k in ("lag_bulk", "lag_tail"):
        assert np.array_equal(got[k], exp[k]), k
    for k in ("ess_bulk", "ess_tail", "rhat"):
        assert np.allclose(got[k], exp[k], rtol=1e-9, atol=0), k
    y = np.cumsum(rng.normal(size=(1030, 4, 1200)), axis=2) * 0.01          # 2 060 pairs in one chunk: the listed route
    expy = oracle.summarize(y, "pcn")
    goty = ctx.summarize(y, "pcn")
    ly = np.concatenate([expy["lag_bulk"], expy["lag_tail"]])
    assert (ly > 256).sum() >= 100
    for k in ("lag_bulk", "lag_tail"):
        assert np.array_equal(goty[k], expy[k]), k
    for k in ("ess_bulk", "ess_tail", "rhat"):
        assert np.allclose(goty[k], expy[k], rtol=1e-9, atol=0), k


def test_two_valued_columns_hit_the_band_by_chance_and_still_match(ctx, oracle):
    """Not a construction: 3 000 indicator-type columns (4 x 500 draws, half of every chain's draws one, shuffled).  z is
    then +-w with chain means of zero up to rounding, every lag product an integer times w^2, and rho is zero up to round-off
    at the deciding lag of a few columns in a hundred -- its sign in the reference is whatever its left-to-right sums
    leave.  Every integer term count must equal the oracle's (the reference's algorithm bit for bit), and the guard must
    have run."""
    rng = np.random.default_rng(2024)
    P, C, N = 3000, 4, 500
    base = np.concatenate([np.zeros(N // 2), np.ones(N // 2)])
    x = np.stack([rng.permutation(base) for _ in range(P * C)]).reshape(P, C, N)
    x[::3] *= 2.5                                        # other value pairs: the ranks are what matters
    x[1::3] -= 7.0
    exp = oracle.summarize_mt(x, "pcn")
    before = ctx.rho_guard_count()
    got = ctx.summarize(x, "pcn")
    assert ctx.rho_guard_count() - before >= 5           # a few dozen columns in the band (seeded: deterministic)
    for k in ("lag_bulk", "lag_tail"):
        assert np.array_equal(got[k], exp[k]), (k, np.flatnonzero(got[k] != exp[k])[:10])
    for k in ("ess_bulk", "ess_tail", "rhat"):
        assert np.allclose(got[k], exp[k], rtol=1e-9, atol=0, equal_nan=True), k


def test_tier3_stages_pairs_that_end_in_different_stages(ctx, oracle):
    """k_tier3 walks a listed pair's lags in stages of 2 048 (round 4): a stage is scanned by whichever workgroup completes
    the later of "its products are there" / "the stage in front is through", and later items skip decided pairs.  AR(1)
    chains of 4 x 10 000 draws with phi from 0.9 to 0.9995 plus random walks end their walks before tier 3 and in its first
    two stages; trends, a level shift and a walk on 4 x 16 000 draws (62 lag groups = 8 stages) in the second and third.
    Many listed pairs against the launch's 256 workgroups, several calls in flight, repeated: lags exact every time."""
    from scipy.signal import lfilter
    rng = np.random.default_rng(123)
    phis = np.array([0.9, 0.97, 0.985, 0.99, 0.993, 0.996, 0.998, 0.999, 0.9993, 0.9995] * 2)
    e = rng.normal(size=(len(phis), 4, 10000)) * np.sqrt(1 - phis * phis)[:, None, None]
    x = np.stack([lfilter([1.0], [1.0, -p], e[i], axis=1) for i, p in enumerate(phis)])
    w = np.cumsum(rng.normal(size=(8, 4, 10000)), axis=2) * 0.01
    w[4:] += rng.normal(size=(4, 4, 10000)) * 0.3
    x = np.concatenate([x, w])
    n = 16000
    tt = np.arange(n) / n
    y = np.empty((4, 4, n))
    y[0] = 3 * tt + rng.normal(size=(4, n)) * 0.2
    y[1] = np.where(tt < 0.5, -1.0, 1.0) + rng.normal(size=(4, n)) * 0.3
    y[2] = np.cumsum(rng.normal(size=(4, n)), axis=1) * 0.01
    y[3] = np.cos(np.pi * tt) + rng.normal(size=(4, n)) * 0.1
    seen = set()
    for arr in (x, y):
        exp = oracle.summarize_mt(arr, "pcn")
        lags = np.concatenate([exp["lag_bulk"], exp["lag_tail"]])
        seen |= set(((lags[lags > 255] - 256) // 2048).tolist()) | ({-1} if (lags <= 255).any() else set())
        t = ctx.upload(arr, "pcn")
        try:
            for _ in range(3):
                bufs = [ctx.enqueue(t) for _ in range(4)]
                ctx.wait()
                for b in bufs:
                    g = b.result()
                    for k in ("lag_bulk", "lag_tail"):
                        assert np.array_equal(g[k], exp[k]), (k, g[k], exp[k])
                    for k in ("ess_bulk", "ess_tail", "rhat"):
                        assert np.allclose(g[k], exp[k], rtol=1e-9, atol=0), k
        finally:
            t.free()
    assert {-1, 0, 1, 2}.issubset(seen), sorted(seen)


@pytest.mark.parametrize("n", [4097, 6001, 9999, 12001])
def test_tier3_stages_with_chain_lengths_that_are_not_multiples_of_16(ctx, oracle, n):
    """The products of a listed pair live in acov[pair][n]: with n not a multiple of 16 a 128-byte cache line can hold the
    last lags of one stage and the first of the next, which another workgroup writes later -- the scan reads them at agent
    scope (and behind an acquire fence), never from a line its CU cached while scanning the stage in front.  Random walks
    whose ESS walks cross several stages, many pairs, calls in flight on every lane, three times over."""
    rng = np.random.default_rng(n)
    x = np.cumsum(rng.normal(size=(24, 4, n)), axis=2) * 0.01
    x[::3] += rng.normal(size=(8, 4, n)) * 0.2
    exp = oracle.summarize_mt(x, "pcn")
    lags = np.concatenate([exp["lag_bulk"], exp["lag_tail"]])
    assert (lags > 768).sum() >= 10 and ((lags > 2304).sum() >= 1 or n < 6000), lags      # (4 097 draws: stages 1 and 2 only)
    t = ctx.upload(x, "pcn")
    try:
        for _ in range(3):
            bufs = [ctx.enqueue(t) for _ in range(8)]
            ctx.wait()
            for b in bufs:
                g = b.result()
                for k in ("lag_bulk", "lag_tail"):
                    assert np.array_equal(g[k], exp[k]), (k, np.flatnonzero(g[k] != exp[k]))
                for k in ("ess_bulk", "ess_tail", "rhat"):
                    assert np.allclose(g[k], exp[k], rtol=1e-9, atol=0), k
    finally:
        t.free()
