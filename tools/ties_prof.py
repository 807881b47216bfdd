#!/usr/bin/env python3
"""Heavily tied columns at the C1 shape (two values / a dozen / 80 / constant): per-kernel times of a lone call and the
pipelined step, next to continuous draws -- do ties cost anything?"""
import os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "mcmc-db_amd")]
from mcmc_ref_hip import _ffi
rng = np.random.default_rng(2)
base = rng.normal(size=(100, 4, 10000))
cases = {"continuous": base, "two values": np.sign(base), "a dozen values": np.floor(base * 2.0), "80 values": np.round(base, 1),
         "constant": np.full_like(base, 3.25), "half constant": np.where(np.arange(100)[:, None, None] % 2 == 0, 1.5, base)}
for name, x in cases.items():
    os.environ["MCR_LANES"] = "1"
    c1 = _ffi.Context(0); os.environ.pop("MCR_LANES")
    t = c1.upload(np.ascontiguousarray(x), "pcn")
    c1.summarize(t)
    g0 = c1.rho_guard_count()
    c1.profile(True); c1.profile_reset()
    for _ in range(5): c1.enqueue(t); c1.wait()
    pr = c1.profile_get(); c1.profile(False)
    guards = (c1.rho_guard_count() - g0) // 5
    ctx = _ffi.Context(0); t2 = _ffi.DeviceTensor(ctx, t.buf, t.targs)
    def run(k):
        for _ in range(k):
            if ctx.inflight >= 8: ctx.wait_one()
            ctx.enqueue(t2)
        ctx.wait()
    run(20); t0 = time.perf_counter(); run(200); dt = (time.perf_counter() - t0) / 200
    print(f"{name:15s}: pipelined {dt*1e6:6.0f} us/call, band lags re-derived per call {guards:4d}; alone: " +
          "  ".join(f"{k[2:]}={v['total_ms']/v['launches']*1e3:.0f}" for k, v in sorted(pr.items(), key=lambda kv: -kv[1]['total_ms'])[:7]), flush=True)
    c1.close(); ctx.close()
