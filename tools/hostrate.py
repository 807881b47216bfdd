#!/usr/bin/env python3
"""Host-side cost of one pipelined call: a tiny tensor (the kernels take no time), rolling window of 8 calls, with and
without reuse of the result buffers.  What is measured is Python + ctypes + the library's launch sequence."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "mcmc-db_amd")]
import numpy as np
from mcmc_ref_hip import _ffi
ctx = _ffi.Context(0)
x = np.random.default_rng(0).normal(size=(1, 4, 100))
t = ctx.upload(x, "pcn")
for reuse in (False, True):
    ring = []
    def run(n):
        for k in range(n):
            if ctx.inflight >= 8:
                b = ctx.wait_one()
                if reuse: ring.append(b)
            ctx.enqueue(t, bufs=ring.pop() if ring else None)
        ctx.wait()
    run(100)
    t0 = time.perf_counter()
    run(2000)
    print(f"tiny call, buffer reuse {reuse}, MCR_GRAPH={os.environ.get('MCR_GRAPH', '0')}: {(time.perf_counter() - t0) / 2000 * 1e6:.1f} us per call")
