#!/usr/bin/env python3
"""Sticky chains at the C1 shape (every parameter AR(1) with phi = 0.99 / 0.999: dozens to hundreds of pairs in tier 3):
per-kernel times of a lone call and the pipelined step; MCR_T3_WG sets how many workgroups a tier-3 launch aims at."""
import os, sys, time
from pathlib import Path
import numpy as np
from scipy.signal import lfilter
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "mcmc-db_amd")]
from mcmc_ref_hip import _ffi
rng = np.random.default_rng(1)
for phi, (C, N, P) in ((0.99, (4, 10000, 100)), (0.999, (4, 10000, 100)), (0.99, (10, 1000, 460))):
    e = rng.normal(size=(P, C, N)) * np.sqrt(1 - phi * phi)
    x = lfilter([1.0], [1.0, -phi], e, axis=2)
    os.environ["MCR_LANES"] = "1"
    c1 = _ffi.Context(0); os.environ.pop("MCR_LANES")
    t = c1.upload(x, "pcn")
    r = c1.summarize(t)
    c1.profile(True); c1.profile_reset()
    for _ in range(5): c1.enqueue(t); c1.wait()
    pr = c1.profile_get(); c1.profile(False)
    lags = np.concatenate([r["lag_bulk"], r["lag_tail"]])
    ctx = _ffi.Context(0); t2 = _ffi.DeviceTensor(ctx, t.buf, t.targs)
    def run(k):
        for _ in range(k):
            if ctx.inflight >= 8: ctx.wait_one()
            ctx.enqueue(t2)
        ctx.wait()
    run(20); t0 = time.perf_counter(); run(100); dt = (time.perf_counter() - t0) / 100
    print(f"T3_WG={os.environ.get('MCR_T3_WG', '256')} phi={phi} {C}x{N}x{P}: pairs beyond lag 255: {(lags > 255).sum()} of {lags.size}, max lag {lags.max()}; "
          f"pipelined {dt*1e6:.0f} us/call; alone: " + "  ".join(f"{k[2:]}={v['total_ms']/v['launches']*1e3:.0f}" for k, v in sorted(pr.items(), key=lambda kv: -kv[1]['total_ms'])[:6]), flush=True)
    c1.close(); ctx.close()
