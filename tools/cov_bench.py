#!/usr/bin/env python3
"""fp64 MFMA covariance (extension X3): achieved TFLOP/s of k_cov_mfma against the dense-fp64 matrix peak.

    python tools/cov_bench.py [--out gpurun_out/cov.json]

`executed_*` counts the flops the kernel really executes: 2 * M * 128 * 128 per tile on or above the block diagonal
(`frac_of_peak` is on those).  `priced_*` counts 2 * P^2 * M, the full Gram matrix a dense SYRK is priced at, of which the
upper-triangle tiling executes about half -- kept for comparison with round 2, never as the utilisation.  Peak: 78.6
TFLOP/s (256 CUs x 4 SIMDs x 32 fp64 FMA per clock x 2.4 GHz; MI355X spec fp64 matrix = fp64 vector rate).  What the
instruction sustains on this chip depends on how densely it is issued (the chip gives clock back under load): a bare loop
of back-to-back independent `v_mfma_f64_16x16x4f64` on every CU holds 46.1 TFLOP/s, the same instruction at 5 MFMAs per 4
LDS loads 72.5 TFLOP/s (tools/ubench/mfma64_rate.hip, profiles/r03_mfma64_rate.txt); `frac_of_dense_loop` is against
the first figure.
Parity unpinned by the reference (no covariance there): the result is checked against numpy.cov(ddof=0)."""
import argparse, json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "mcmc-db_amd")]
from mcmc_ref_hip import _ffi

PEAK_TF = 78.6
MFMA_LOOP_TF = 46.1      # measured: tools/ubench/mfma64_rate, 5 independent accumulators, every CU busy
TILE = 128
ap = argparse.ArgumentParser()
ap.add_argument("--out", default=None)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
ctx = _ffi.Context(0)
rows = []
for P, M in ((100, 40000), (1000, 40000), (2048, 40000), (4096, 20000)):
    rng = np.random.default_rng(P)
    x = rng.normal(size=(P, M)) * (1.0 + np.arange(P)[:, None] % 7) + np.arange(P)[:, None]
    dx = _ffi.DeviceBuffer(ctx, x.nbytes).upload(x)
    dc = _ffi.DeviceBuffer(ctx, P * P * 8)
    call = lambda: ctx._check(ctx.lib.mcr_covariance_dev(ctx.handle, dx.ptr, M, P, dc.ptr))
    call()
    cov = dc.download(np.float64, P * P).reshape(P, P)
    exp = np.cov(x, ddof=0) if P <= 1000 else None
    err = float(np.max(np.abs(cov - exp)) / np.max(np.abs(exp))) if exp is not None else None
    ctx.profile(True); ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        call()
    wall = (time.perf_counter() - t0) / a.reps
    pr = ctx.profile_get(); ctx.profile(False)
    kms = pr["k_cov_mfma"]["total_ms"] / pr["k_cov_mfma"]["launches"]
    nb = (P + TILE - 1) // TILE
    priced = 2.0 * P * P * M
    executed = 2.0 * M * (nb * (nb + 1) / 2) * TILE * TILE
    ex_tf = executed / (kms * 1e-3) / 1e12
    rows.append({"P": P, "M": M, "kernel_ms": kms, "call_ms": wall * 1e3,
                 "executed_TFLOPs": ex_tf, "frac_of_peak": ex_tf / PEAK_TF, "frac_of_dense_loop": ex_tf / MFMA_LOOP_TF,
                 "priced_TFLOPs": priced / (kms * 1e-3) / 1e12, "priced_frac_of_peak": priced / (kms * 1e-3) / 1e12 / PEAK_TF,
                 "executed_over_priced": executed / priced, "max_rel_err_vs_numpy": err,
                 "other_kernels_ms": {k: v["total_ms"] / v["launches"] for k, v in pr.items() if k != "k_cov_mfma"}})
    print(json.dumps(rows[-1]), flush=True)
    dx.free(); dc.free()
out = {"kernel": "k_cov_mfma (v_mfma_f64_16x16x4f64, 128x128 tiles = 4x4 MFMA tiles per wave, LDS-staged, double-buffered)",
       "peak_TFLOPs": PEAK_TF, "mfma_f64_16x16x4_loop_TFLOPs_measured": MFMA_LOOP_TF, "rows": rows}
if a.out:
    Path(a.out).write_text(json.dumps(out, indent=1) + "\n")
ctx.close()
