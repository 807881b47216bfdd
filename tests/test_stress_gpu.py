"""BASELINE config 4 at FULL size in the `-m gpu` suite: 4 x 100000 x 10000 f32 =
16 GB of draws generated on the device, ALL 10 000 parameters through the full pipeline (chunked through the
workspace; f32 tensors are sorted as packed records, csrc/mcr_sort32.hpp), checked by

  * size-independent properties over every parameter (the generator's location / scale, ordered quantiles, median ==
    q50, 0 < ESS <= M, rhat = max(bulk, tail) < 1.01 for iid draws, non-negative integer lags), and
  * the oracle (the reference's algorithm, oracle/mcr_oracle.c) on 128 parameters -- the first and last parameter of every
    workspace chunk, every scale of the generator, the all-ties parameters at the f32 grid -- integer outputs exact, floats
    to 1e-9.

The reference itself is fp64-only and has no golden for this shape; parity is against the oracle on the widened draws."""
from __future__ import annotations

import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _row(ctx, t, i: int, M: int) -> np.ndarray:
    out = np.empty(M, dtype=np.float32)
    ctx._check(ctx.lib.mcr_memcpy_d2h(ctx.handle, out.ctypes.data_as(C.c_void_p), C.c_void_p(t.buf.ptr.value + i * M * 4),
                                      out.nbytes))
    return out


def test_config4_full_size_all_parameters(oracle, monkeypatch):
    from mcmc_ref_hip import _ffi
    monkeypatch.setenv("MCR_LANES", "1")                    # one 0.25 s call: a single lane, a single 8 GiB workspace
    Cn, N, P = 4, 100000, 10000
    M = Cn * N
    with _ffi.Context(0) as ctx:
        t = ctx.alloc_tensor(Cn, N, P, np.float32)
        try:
            ctx.fill_synthetic(t, 4711)
            got = ctx.summarize(t)
            p = np.arange(P)
            sig = 10.0 ** ((p % 7) - 3)
            fine = sig >= 64.0 * np.spacing(np.maximum(p, 1).astype(np.float32)).astype(np.float64)   # sigma above the f32 grid at p
            assert fine.sum() > 4000
            assert np.all(np.abs(got["mean"] - p)[fine] < 0.02 * sig[fine])
            assert np.all(np.abs(got["std"] / sig - 1)[fine] < 0.02)
            assert np.all(got["std"] > 0) and np.all(np.isfinite(got["std"]))
            assert np.all(got["q"][:, 0] <= got["q"][:, 1]) and np.all(got["q"][:, 1] <= got["q"][:, 2])
            assert np.array_equal(got["q"][:, 1], got["median"])
            for k in ("ess_bulk", "ess_tail"):
                assert np.all((got[k] > 0) & (got[k] <= M)), k
            assert np.all(got["rhat"] >= got["rhat_bulk"]) and np.all(got["rhat"] >= got["rhat_tail"])
            assert np.all(got["rhat"] < 1.01)                                   # iid chains
            assert np.all(got["lag_bulk"] >= 0) and np.all(got["lag_tail"] >= 0)
            # the oracle on 128 parameters: both sides of every workspace-chunk edge, every scale of the generator,
            # >= 8 parameters whose sigma is at the f32 grid (all ties: still exact ranks) -- synth.stress_check_sample
            from mcmc_ref_hip import synth
            per_chunk = ctx.params_per_chunk(t)
            assert 0 < per_chunk < P                        # the 16 GB tensor does stream through the workspace in chunks
            sel = synth.stress_check_sample(P, per_chunk, 128)
            edges = set(range(0, P, per_chunk)) | {min(p0 + per_chunk, P) - 1 for p0 in range(0, P, per_chunk)}
            assert len(sel) >= 128 and edges <= set(sel.tolist()) and {int(p) % 7 for p in sel} == set(range(7))
            sub = np.stack([_row(ctx, t, int(i), M) for i in sel]).reshape(len(sel), Cn, N)
            grid = np.spacing(sel.astype(np.float32)).astype(np.float64)
            n_tied = sum(len(np.unique(sub[k])) < 64 for k in range(len(sel)) if 10.0 ** ((sel[k] % 7) - 3) <= 2.1 * grid[k])
            assert n_tied >= 8, n_tied
        finally:
            t.free()
    exp = oracle.summarize_mt(sub, "pcn")
    for k in ("lag_bulk", "lag_tail"):
        assert np.array_equal(got[k][sel], exp[k]), k
    assert np.array_equal(got["q"][sel], exp["q"]) and np.array_equal(got["median"][sel], exp["median"])
    for k in ("mean", "std", "rhat", "rhat_bulk", "rhat_tail", "ess_bulk", "ess_tail"):
        assert np.allclose(got[k][sel], exp[k], rtol=1e-9, atol=0), (k, got[k][sel], exp[k])
