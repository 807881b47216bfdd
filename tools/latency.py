#!/usr/bin/env python3
"""Single-call latencies: host-buffer boundary (PCIe-inclusive) vs device-resident, C1 and C0 shapes; MCR_FORK=0/1."""
import os, sys, time, statistics
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "mcmc-db_amd")]
from mcmc_ref_hip import _ffi, synth
ctx = _ffi.Context(0)
for name, (C, N, P) in {"C1 4x10000x100": (4, 10000, 100), "C0 4x1000x10": (4, 1000, 10), "corpus 10x1000x45": (10, 1000, 45)}.items():
    x = synth.c1_model(C, N, P)
    t = ctx.upload(x, "pcn")
    for _ in range(5):
        ctx.summarize(x); ctx.summarize(t)
    def med(fn, n):
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        return statistics.median(ts), min(ts)
    host, hmin = med(lambda: ctx.summarize(x), 30)
    dev, dmin = med(lambda: ctx.summarize(t), 60)
    print(f"fork={os.environ.get('MCR_FORK','1')} {name}: host-buffer call {host*1e6:.0f} us (min {hmin*1e6:.0f}), "
          f"device-resident synchronous call {dev*1e6:.0f} us (min {dmin*1e6:.0f})", flush=True)
    t.free()
