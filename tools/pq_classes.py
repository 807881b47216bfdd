#!/usr/bin/env python3
"""k_pq_snappy / k_pq_decode time by column class on the committed corpus files: the parameter columns, `chain`, `draw`."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "mcmc-db_amd")]
from mcmc_ref_hip import _ffi, parquet
from mcmc_ref_hip._ffi import MCR_PQ_F64, MCR_PQ_I64, DeviceBuffer

paths = sorted((ROOT / "tests/golden/corpus/draws").glob("*.draws.parquet"))
ctx = _ffi.Context(0)
files = [parquet.ParquetFile(p, ctx) for p in paths]
buf = DeviceBuffer(ctx, 64 << 20)
for label, pick in (("parameters", lambda n: n not in ("chain", "draw")), ("chain", lambda n: n == "chain"), ("draw", lambda n: n == "draw"),
                    ("all", lambda n: True)):
    reqs, off = [], 0
    for f in files:
        for i, n in enumerate(f.column_names):
            if pick(n):
                kind = MCR_PQ_I64 if n in ("chain", "draw") else MCR_PQ_F64
                reqs.append((f, i, kind, buf.ptr.value + off))
                off += f.num_rows * 8
    for _ in range(2):
        parquet.decode(ctx, reqs)
    ctx.profile(True); ctx.profile_reset()
    for _ in range(5):
        parquet.decode(ctx, reqs)
    pr = ctx.profile_get(); ctx.profile(False)
    pages = sum(1 for f, i, _, _ in reqs for pg in f.pages() if pg["column"] == i)
    print(f"{label:10s} {len(reqs):4d} columns {pages:5d} pages  " + "  ".join(f"{k}={v['total_ms'] / v['launches'] * 1e3:.0f}us" for k, v in pr.items() if k.startswith("k_pq")))
f = files[0]
for pg in f.pages():
    if f.column_names[pg["column"]] in ("chain", "draw") or pg["column"] == 2:
        print(f.column_names[pg["column"]], {k: pg[k] for k in ("kind", "encoding", "compressed_size", "uncompressed_size", "num_values")})

# which files' parameter columns are slow, and what their pages look like
rows = []
for f in files:
    reqs, off = [], 0
    for i, n in enumerate(f.column_names):
        if n not in ("chain", "draw"):
            reqs.append((f, i, MCR_PQ_F64, buf.ptr.value + off)); off += f.num_rows * 8
    parquet.decode(ctx, reqs)
    ctx.profile(True); ctx.profile_reset()
    for _ in range(3):
        parquet.decode(ctx, reqs)
    pr = ctx.profile_get(); ctx.profile(False)
    us = pr["k_pq_snappy"]["total_ms"] / pr["k_pq_snappy"]["launches"] * 1e3
    rows.append((us, f.path.name, len(reqs)))
rows.sort(reverse=True)
print("slowest files (parameter columns only):", [(round(u), n[:40], k) for u, n, k in rows[:6]], "fastest:", [(round(u), n[:30]) for u, n, k in rows[-3:]])
slow = [f for f in files if f.path.name == rows[0][1]][0]
pg = [p for p in slow.pages() if slow.column_names[p["column"]] not in ("chain", "draw")]
pg.sort(key=lambda p: p["compressed_size"] / max(p["uncompressed_size"], 1))
print("its most compressible pages:", [(slow.column_names[p["column"]], p["kind"], p["compressed_size"], p["uncompressed_size"], p["num_values"]) for p in pg[:6]])
