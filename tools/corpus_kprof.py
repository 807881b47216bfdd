#!/usr/bin/env python3
"""Per-kernel HIP-event times of the statistics pipeline on the REAL packaged corpus (device-resident, one lane), next to
the same shapes filled with synthetic draws: what real MCMC draws cost in tiers 2 and 3."""
import os, sys
os.environ.setdefault("MCR_LANES", "1")
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "mcmc-db_amd")]
from mcmc_ref_hip import _ffi, parquet, synth

ctx = _ffi.Context(0)
paths = sorted((ROOT / "tests/golden/corpus/draws").glob("*.draws.parquet"))
draws = parquet.read_draws_many(ctx, paths)
def run(tensors, label):
    for t in tensors: ctx.enqueue(t); ctx.wait()
    ctx.profile(True); ctx.profile_reset()
    res = []
    for t in tensors:
        b = ctx.enqueue(t); ctx.wait(); res.append(b.result())
    pr = ctx.profile_get(); ctx.profile(False)
    tot = sum(v["total_ms"] for v in pr.values())
    print(f"{label}: kernels {tot:.3f} ms  " + "  ".join(f"{k[2:]}={v['total_ms']*1e3:.0f}us/{v['launches']}" for k, v in sorted(pr.items(), key=lambda kv: -kv[1]["total_ms"])))
    return res
real = run([d.tensor for d in draws], "real corpus (57 calls)")
lags = np.concatenate([np.concatenate([r["lag_bulk"], r["lag_tail"]]) for r in real])
print("truncation lags: median %d, p90 %d, p99 %d, max %d; pairs beyond lag 63: %d, beyond 255: %d of %d" % (
    np.median(lags), np.percentile(lags, 90), np.percentile(lags, 99), lags.max(), (lags > 63).sum(), (lags > 255).sum(), lags.size))
syn = [ctx.upload(synth.c1_model(*d.tensor.shape_cnp[:2], d.tensor.shape_cnp[2], seed=7 + i), "pcn") for i, d in enumerate(draws)]
run(syn, "same shapes, synthetic draws")
