"""Extensions named in the north star but absent from the reference:
PARITY UNPINNED by the reference; pinned here to scipy / numpy, whose definitions the kernels restate."""
from __future__ import annotations

import numpy as np
import pytest

from conftest import load_model

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from mcmc_ref_hip import _ffi
    return _ffi.default_context()


def test_two_sample_matches_scipy(ctx):
    from scipy.stats import ks_2samp, wasserstein_distance
    rng = np.random.default_rng(4)
    cases = [(3, 10000, 4000), (2, 5000, 5000), (4, 37, 41), (1, 1, 1), (2, 70000, 9000), (1, 4096, 8192)]
    for P, Mr, Ma in cases:
        ref = rng.normal(size=(P, Mr))
        act = rng.normal(loc=0.1, scale=1.2, size=(P, Ma))
        if P > 1:
            ref[1] = np.round(ref[1], 1); act[1] = np.round(act[1], 1)        # ties within and across samples
        ks, w1 = ctx.two_sample(ref, act)
        for p in range(P):
            exact = ks_2samp(ref[p], act[p], method="exact" if max(Mr, Ma) <= 10000 else "asymp").statistic
            if max(Mr, Ma) <= 10000:
                assert ks[p] == exact, (P, Mr, Ma, p)            # scipy's exact mode returns the same rational
            else:
                assert ks[p] == pytest.approx(exact, rel=1e-13)   # asymp mode: float CDF differences
            assert w1[p] == pytest.approx(wasserstein_distance(ref[p], act[p]), rel=1e-12, abs=1e-15)
    same = rng.normal(size=(2, 1000))
    ks, w1 = ctx.two_sample(same, same)
    assert np.all(ks == 0.0) and np.all(w1 == 0.0)
    with pytest.raises(ValueError):
        ctx.two_sample(np.array([[1.0, np.nan]]), np.array([[1.0, 2.0]]))
    # non-finite draws in long (multi-tile) samples, either side, are reported and never walked as a partition
    big_r, big_a = rng.normal(size=(3, 40000)), rng.normal(size=(3, 25000))
    for poison in (np.nan, np.inf, -np.inf):
        for side in (0, 1):
            r, a = big_r.copy(), big_a.copy()
            (r if side == 0 else a)[1, ::11] = poison
            with pytest.raises(ValueError):
                ctx.two_sample(r, a)
    ks, w1 = ctx.two_sample(big_r, big_a)
    assert np.all(np.isfinite(ks)) and np.all(np.isfinite(w1))


def test_covariance_mfma_matches_numpy(ctx):
    rng = np.random.default_rng(5)
    for P, M in [(1, 10), (5, 1000), (16, 4096), (33, 10001), (65, 77), (130, 3001), (100, 40000)]:
        L = rng.normal(size=(P, P))
        x = (L @ rng.normal(size=(P, M))) * 1e-2 + rng.normal(size=(P, 1)) * 50.0   # correlated, offset means
        cov = ctx.covariance(x)
        exp = np.cov(x, ddof=0).reshape(P, P)
        assert np.allclose(cov, exp, rtol=1e-10, atol=1e-12 * np.abs(exp).max()), (P, M)
        assert np.allclose(cov, cov.T, rtol=1e-12, atol=0)
    draws, params, rec = load_model("eight_schools-eight_schools_noncentered")
    x = draws.reshape(len(params), -1)
    cov = ctx.covariance(x)
    assert np.allclose(np.sqrt(np.diag(cov)), [rec["stats"]["numpy"][p]["std"] for p in params], rtol=1e-10)


def test_validate_api(tmp_path):
    import json
    import pyarrow as pa
    import pyarrow.parquet as pq
    from scipy.stats import ks_2samp
    from mcmc_ref_hip.store import DataStore
    from mcmc_ref_hip.validate import validate
    draws, params, rec = load_model("radon_pooled")
    P, C, N = draws.shape
    root = tmp_path / "r"
    (root / "draws").mkdir(parents=True); (root / "meta").mkdir()
    cols = {"chain": np.repeat(np.arange(C), N), "draw": np.tile(np.arange(N), C)}
    cols.update({p: draws[i].reshape(-1) for i, p in enumerate(params)})
    pq.write_table(pa.table(cols), root / "draws" / "radon.draws.parquet")
    (root / "meta" / "radon.meta.json").write_text(json.dumps({"diagnostics": rec["meta_diagnostics"]}))
    st = DataStore(local_root=root, packaged_root=tmp_path / "none")
    rng = np.random.default_rng(1)
    good = {p: list(rng.choice(draws[i].reshape(-1), size=3000)) for i, p in enumerate(params)}
    res = validate("radon", good, ks_max=0.1, w1_scaled_max=0.1, store=st)
    assert res.passed and res.compare.passed and res.failures == []
    for i, p in enumerate(params):
        assert res.ks[p] == pytest.approx(ks_2samp(draws[i].reshape(-1), np.asarray(good[p])).statistic, rel=1e-13)
    bad = dict(good)
    bad[params[0]] = [v * 3.0 for v in good[params[0]]]
    res = validate("radon", bad, ks_max=0.1, store=st)
    assert not res.passed and any(f.startswith(f"{params[0]}.ks=") for f in res.failures)
    assert any(f.startswith(f"{params[0]}.std rel_error=") for f in res.failures)   # the reference's gate fires too
