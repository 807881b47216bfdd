#!/usr/bin/env python3
"""Phase table of mcr_summarize_files on the committed corpus files: python tools/ingest_phases.py [reps]"""
import json, statistics, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "mcmc-db_amd")]
from mcmc_ref_hip import _ffi, parquet

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
paths = sorted((ROOT / "tests/golden/corpus/draws").glob("*.draws.parquet"))
ctx = _ffi.Context(0)
sp = [str(p) for p in paths]
for _ in range(3):
    parquet.summarize_files(ctx, paths)
runs, wall_c, wall_py = [], [], []
for _ in range(reps):
    ph = {}
    t0 = time.perf_counter(); parquet._summarize_paths(ctx, sp, 4, [0.05, 0.5, 0.95], True, ph); wall_c.append((time.perf_counter() - t0) * 1e3)
    runs.append(ph)
    t0 = time.perf_counter(); parquet.summarize_files(ctx, paths); wall_py.append((time.perf_counter() - t0) * 1e3)
out = {"files": len(paths), "file_bytes": sum(p.stat().st_size for p in paths),
       "phases_ms_median": {k: round(statistics.median(r[k] for r in runs), 3) for k in runs[0]},
       "io_threads": int(__import__("os").environ.get("MCR_IO_THREADS", "8")), "fork": __import__("os").environ.get("MCR_FORK", "1"),
       "c_call_plus_dicts_ms_median": round(statistics.median(wall_c), 3), "summarize_files_ms_median": round(statistics.median(wall_py), 3),
       "summarize_files_ms_min": round(min(wall_py), 3)}
print(json.dumps(out))
ctx.close()
