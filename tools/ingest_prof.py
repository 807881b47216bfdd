#!/usr/bin/env python3
"""cProfile of the native Parquet route (host-side overheads of tools/ingest_bench.py's native pass)."""
import cProfile, io, pstats, sys, tempfile, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "mcmc-db_amd")]
import pyarrow as pa, pyarrow.parquet as pq
from mcmc_ref_hip import _ffi, corpus, parquet
ctx = _ffi.Context(0)
with tempfile.TemporaryDirectory() as td:
    paths = []
    for name, x in corpus.synthetic_corpus(seed=4711):
        P, C, N = x.shape
        cols = {"chain": np.repeat(np.arange(C), N), "draw": np.tile(np.arange(N), C)}
        for i in range(P):
            cols[f"p[{i + 1}]"] = x[i].reshape(-1)
        path = Path(td) / f"{name}.draws.parquet"; pq.write_table(pa.table(cols), path); paths.append(path)
    for _ in range(2):
        parquet.summarize_files(ctx, paths)
    t0 = time.perf_counter(); parquet.summarize_files(ctx, paths); print("wall", time.perf_counter() - t0)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5):
        parquet.summarize_files(ctx, paths)
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats(22); print(s.getvalue()[:6000])
