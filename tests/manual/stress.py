#!/usr/bin/env python3
"""BASELINE config 4 (stress): 4 x 100000 x P fp32 synthetic draws generated on the device.

  python tests/manual/stress.py --params 10000        # 16 GB tensor: streaming-moments roofline + full pipeline
Reports the moments kernel's achieved HBM GB/s (algorithmic bytes = 4 B per param-draw) and the
full-pipeline param-draws/s, and checks a 16-parameter slice against the CPU oracle.
"""
import argparse, json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "mcmc-db_amd")]
from mcmc_ref_hip import _ffi

ap = argparse.ArgumentParser()
ap.add_argument("--chains", type=int, default=4)
ap.add_argument("--draws", type=int, default=100000)
ap.add_argument("--params", type=int, default=10000)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--pipeline-params", type=int, default=1000)
ap.add_argument("--check-params", type=int, default=16)
a = ap.parse_args()
C, N, P = a.chains, a.draws, a.params
dt = np.float32 if a.dtype == "f32" else np.float64
es = np.dtype(dt).itemsize
ctx = _ffi.Context(0)
t = ctx.alloc_tensor(C, N, P, dt)
ctx.fill_synthetic(t, 4711)
out = {"shape": [C, N, P], "dtype": a.dtype, "bytes": C * N * P * es}
# --- streaming moments (the HBM-roofline kernel) ---
ctx.moments(t)
ctx.profile(True); ctx.profile_reset()
t0 = time.perf_counter()
for _ in range(a.reps):
    mean, std = ctx.moments(t)
wall = (time.perf_counter() - t0) / a.reps
pr = ctx.profile_get(); ctx.profile(False)
kms = pr["k_moments"]["total_ms"] / pr["k_moments"]["launches"]
out["moments"] = {"kernel_ms": kms, "wall_ms": wall * 1e3, "GBps": C * N * P * es / kms / 1e6,
                  "frac_of_8TBps": C * N * P * es / kms / 1e6 / 8000.0, "param_draws_per_s": C * N * P / (kms * 1e-3)}
exp_mean = np.arange(P, dtype=np.float64)
sig = 10.0 ** ((np.arange(P) % 7) - 3)
out["moments"]["max_mean_err_sigma"] = float(np.max(np.abs(mean - exp_mean) / sig))
out["moments"]["max_std_relerr"] = float(np.max(np.abs(std / sig - 1)))
print(json.dumps(out), flush=True)
# --- parity of a slice against the oracle ---
if a.check_params:
    from oracle import oracle as orc
    k = a.check_params
    sl = t.buf.download(dt, C * N * k).reshape(k, C, N)
    got = ctx.summarize(sl, "pcn")
    exp = orc.summarize(sl, "pcn")
    ok = np.array_equal(got["lag_bulk"], exp["lag_bulk"]) and np.array_equal(got["lag_tail"], exp["lag_tail"]) \
        and np.array_equal(got["q"], exp["q"])
    worst = max(float(np.max(np.abs(got[f] - exp[f]) / np.abs(exp[f]))) for f in ("std", "rhat", "ess_bulk", "ess_tail"))
    m, s = ctx.moments(ctx.upload(sl, "pcn"))
    print(json.dumps({"slice_parity": bool(ok and worst < 1e-6), "worst_rel": worst,
                      "moments_vs_oracle": float(np.max(np.abs(s - exp["std"]) / exp["std"]))}), flush=True)
# --- full pipeline on the first pipeline-params parameters (chunked through the workspace) ---
if a.pipeline_params:
    k = min(a.pipeline_params, P)
    sub = _ffi.DeviceTensor(ctx, t.buf, (t.targs[0], C, N, k, N, 1, C * N))
    walls = []
    for _ in range(6):                       # consecutive calls rotate over the lanes: each lane allocates its workspace once
        t0 = time.perf_counter(); ctx.summarize(sub); walls.append(round(time.perf_counter() - t0, 4))
    print(json.dumps({"pipeline_params": k, "wall_seconds_of_consecutive_calls": walls}), flush=True)
    ctx.profile(True); ctx.profile_reset()
    t0 = time.perf_counter()
    r = ctx.summarize(sub)
    el = time.perf_counter() - t0
    pr = ctx.profile_get(); ctx.profile(False)
    print(json.dumps({"pipeline_params": k, "seconds": el, "param_draws_per_s": C * N * k / el,
                      "kernels_ms": {n: round(v["total_ms"], 2) for n, v in pr.items()}}), flush=True)
t.free(); ctx.close()
