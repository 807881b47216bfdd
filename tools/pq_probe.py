#!/usr/bin/env python3
"""Decode the corpus-shaped Parquet set a few times (for rocprofv3 --kernel-trace / HIP-event timing of
k_pq_snappy and k_pq_decode).  usage: pq_probe.py [reps] [compression]"""
import io
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "mcmc-db_amd")]
import pyarrow as pa
import pyarrow.parquet as pq
from mcmc_ref_hip import _ffi, corpus, parquet

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
comp = sys.argv[2] if len(sys.argv) > 2 else "snappy"
ctx = _ffi.Context(0)
imgs = []
for name, x in corpus.synthetic_corpus(seed=4711):
    P, C, N = x.shape
    cols = {"chain": np.repeat(np.arange(C), N), "draw": np.tile(np.arange(N), C)}
    for i in range(P):
        cols[f"p[{i + 1}]"] = x[i].reshape(-1)
    b = io.BytesIO()
    pq.write_table(pa.table(cols), b, compression=comp)
    imgs.append(b.getvalue())
files = [parquet.ParquetFile(i, ctx) for i in imgs]
for k in range(reps + 1):
    if k == 1:
        ctx.profile(True); ctx.profile_reset()
    t0 = time.perf_counter()
    ds = parquet.read_draws_many(ctx, files)
    dt = time.perf_counter() - t0
    for d in ds:
        d.free()
    print(f"pass {k}: {dt * 1e3:.2f} ms")
for k, v in ctx.profile_get().items():
    print(k, v["launches"], f"{v['total_ms'] / v['launches'] * 1e3:.1f} us/launch")
