#!/usr/bin/env python3
"""The two long-chain cases of tests/test_hip_parity.py (random walks whose ESS truncation lags are in the thousands:
4 x 20 000 and 2 x 60 000 draws) called repeatedly, for `rocprofv3 --kernel-trace --stats` (per-kernel time of the
FFT tier, csrc/mcr_fft.hpp).  usage: fft_prof.py [reps]"""
import sys
import time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "mcmc-db_amd")]
from mcmc_ref_hip import _ffi

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = _ffi.Context(0)
rng = np.random.default_rng(8)
for shape, mc in (((1, 4, 20000), 4), ((1, 2, 60000), 2), ((8, 4, 100000), 4)):
    x = np.cumsum(rng.normal(size=shape), axis=2) * 0.01
    t = ctx.upload(x, "pcn")
    ctx.summarize(t, min_chains=mc)
    t0 = time.perf_counter()
    for _ in range(reps):
        r = ctx.summarize(t, min_chains=mc)
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"{shape}: {ms:.3f} ms per call, lags {r['lag_bulk'].tolist()} / {r['lag_tail'].tolist()}", flush=True)
    t.free()
ctx.close()
