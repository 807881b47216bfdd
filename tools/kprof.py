#!/usr/bin/env python3
"""Per-kernel HIP-event timings for a few synthetic shapes (development aid)."""
import os, sys, time
os.environ.setdefault("MCR_LANES", "1")   # per-kernel timings of kernels running alone
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "mcmc-db_amd")]
from mcmc_ref_hip import _ffi, synth

def run(ctx, x, name, steps=30):
    t = ctx.upload(x, "pcn")
    for _ in range(3):
        ctx.enqueue(t); ctx.wait()
    ctx.profile(True); ctx.profile_reset()
    t0 = time.perf_counter()
    for k in range(steps):
        ctx.enqueue(t)
        if (k + 1) % 4 == 0: ctx.wait()
    ctx.wait()
    el = (time.perf_counter() - t0) / steps
    pr = ctx.profile_get(); ctx.profile(False)
    print(f"{name}: {el*1e6:.0f} us/step  " + "  ".join(f"{k[2:]}={v['total_ms']/v['launches']*1e3:.1f}x{v['launches']//steps}" for k, v in pr.items()), flush=True)
    t.free()

if __name__ == "__main__":
    ctx = _ffi.Context(0)
    rng = np.random.default_rng(0)
    which = sys.argv[1:] or ["iid", "c1", "corpus"]
    if "iid" in which:
        run(ctx, rng.normal(size=(100, 4, 10000)), "iid 4x10000x100")
    if "c1" in which:
        run(ctx, synth.c1_model(4, 10000, 100), "c1  4x10000x100")
    if "corpus" in which:
        run(ctx, rng.normal(size=(45, 10, 1000)), "iid 10x1000x45")
        run(ctx, rng.normal(size=(460, 10, 1000)), "iid 10x1000x460")
    if "big" in which:
        run(ctx, rng.normal(size=(1000, 4, 10000)), "iid 4x10000x1000", steps=8)
