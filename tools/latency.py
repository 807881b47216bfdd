#!/usr/bin/env python3
"""Single-call latencies: host-buffer boundary (PCIe-inclusive) vs device-resident, C1 and C0 shapes."""
import os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "mcmc-db_amd")]
from mcmc_ref_hip import _ffi, synth
ctx = _ffi.Context(0)
for name, (C, N, P) in {"C1 4x10000x100": (4, 10000, 100), "C0 4x1000x10": (4, 1000, 10)}.items():
    x = synth.c1_model(C, N, P)
    t = ctx.upload(x, "pcn")
    for _ in range(3):
        ctx.summarize(x); ctx.summarize(t)
    n = 30
    t0 = time.perf_counter()
    for _ in range(n): ctx.summarize(x)
    host = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n): ctx.summarize(t)
    dev = (time.perf_counter() - t0) / n
    pd = C * N * P
    print(f"{name}: host-buffer call {host*1e3:.3f} ms ({pd/host/1e9:.2f} G pd/s incl. H2D of {x.nbytes/1e6:.1f} MB), "
          f"device-resident synchronous call {dev*1e3:.3f} ms ({pd/dev/1e9:.2f} G pd/s)", flush=True)
    t.free()
