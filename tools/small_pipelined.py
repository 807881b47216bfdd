#!/usr/bin/env python3
"""Pipelined throughput of SMALL calls (4 x 10000 x 10, 4 x 1000 x 10, 10 x 1000 x 8): host-bound?  MCR_GRAPH=0/1."""
import os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "mcmc-db_amd")]
from mcmc_ref_hip import _ffi, synth
ctx = _ffi.Context(0)
for (C, N, P) in ((4, 10000, 10), (4, 1000, 10), (10, 1000, 8), (4, 10000, 100)):
    x = synth.c1_model(C, N, P, seed=3)
    t = ctx.upload(x, "pcn")
    ring = []
    def run(k):
        for _ in range(k):
            if ctx.inflight >= 8: ring.append(ctx.wait_one())
            ctx.enqueue(t, bufs=ring.pop() if ring else None)
        ctx.wait()
    run(50)
    ws = []
    for _ in range(5):
        t0 = time.perf_counter(); run(400); ws.append((time.perf_counter() - t0) / 400 * 1e6)
    print(f"MCR_GRAPH={os.environ.get('MCR_GRAPH','0')} {C}x{N}x{P}: {sorted(ws)[2]:.1f} us per call pipelined", flush=True)
    t.free()
