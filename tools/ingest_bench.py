#!/usr/bin/env python3
"""End-to-end corpus pass from Parquet files: native GPU ingest vs the pyarrow route.

Writes the 57 corpus-shaped models (synthetic draws, pyarrow's default writer = what the reference's convert.py
produces: SNAPPY + RLE_DICTIONARY, one row group) into a temp directory, then times, page cache warm:

  arrow : pq.read_table -> numpy [P][M] -> H2D -> pipelined summarise      (the reference's reader feeding the kernels)
  native: mmap -> mcr_parquet_decode (one batched call) -> pipelined summarise, draws never decoded on the host

and checks that both give identical statistics.  Prints one JSON object.
"""
from __future__ import annotations

import json
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "mcmc-db_amd")]


def main():
    import pyarrow as pa
    import pyarrow.parquet as pq
    from mcmc_ref_hip import _ffi, corpus, parquet
    from mcmc_ref_hip.convert import table_to_tensor

    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    real_dir = Path(sys.argv[2]) if len(sys.argv) > 2 else None      # directory of real *.draws.parquet files
    ctx = _ffi.Context(0)
    with tempfile.TemporaryDirectory() as td:
        paths = []
        if real_dir is not None:
            paths = sorted(real_dir.glob("*.draws.parquet"))
            total_pd = 0
            for p in paths:
                md = pq.ParquetFile(p).metadata
                total_pd += md.num_rows * (md.num_columns - 2)
        else:
            models = corpus.synthetic_corpus(seed=4711)
            total_pd = sum(int(np.prod(m.shape)) for _, m in models)
            for name, x in models:
                P, C, N = x.shape
                cols = {"chain": np.repeat(np.arange(C), N), "draw": np.tile(np.arange(N), C)}
                for i in range(P):
                    cols[f"p[{i + 1}]"] = x[i].reshape(-1)
                path = Path(td) / f"{name}.draws.parquet"
                pq.write_table(pa.table(cols), path)
                paths.append(path)
        file_bytes = sum(p.stat().st_size for p in paths)

        def arrow_pass():
            t_dec = t_rest = 0.0
            out, pend, tens = [], [], []
            for path in paths:
                t0 = time.perf_counter()
                table = pq.read_table(path)
                params = [c for c in table.column_names if c not in ("chain", "draw")]
                x, counts = table_to_tensor(table, params)
                t1 = time.perf_counter()
                t = ctx.upload(x.reshape(len(params), len(counts), int(counts[0])), "pcn")
                if ctx.inflight >= _ffi.MCR_MAX_INFLIGHT:
                    ctx.wait_one()
                pend.append((params, ctx.enqueue(t)))
                tens.append(t)
                t_dec += t1 - t0
                t_rest += time.perf_counter() - t1
            t1 = time.perf_counter()
            ctx.wait()
            t_rest += time.perf_counter() - t1
            for t in tens:
                t.free()
            qs = (0.05, 0.5, 0.95)
            return [{p: parquet._entry(b.result(), i, qs, True) for i, p in enumerate(params)} for params, b in pend], t_dec, t_rest

        def native_pass():
            return parquet.summarize_files(ctx, paths)

        def native_phases():
            """Per-phase host clock of ONE mcr_summarize_files call (mcr_fileset_phases)."""
            ph = {}
            parquet._summarize_paths(ctx, [str(p) for p in paths], 4, [0.05, 0.5, 0.95], True, ph)
            return ph

        ra, _, _ = arrow_pass()
        rn = native_pass()
        same = ra == rn
        ta = []
        for _ in range(reps):
            t0 = time.perf_counter(); _, td_, tr_ = arrow_pass(); ta.append((time.perf_counter() - t0, td_, tr_))
        tn = []
        for _ in range(reps):
            t0 = time.perf_counter(); native_pass(); tn.append(time.perf_counter() - t0)
        phase_runs = [native_phases() for _ in range(reps)]
        phases = {k: sorted(r[k] for r in phase_runs)[len(phase_runs) // 2] for k in phase_runs[0]}      # medians
        # decode alone (metadata parse + upload + kernels), and the kernels by HIP events
        files = [parquet.ParquetFile(p, ctx) for p in paths]
        t0 = time.perf_counter()
        for _ in range(reps):
            ds = parquet.read_draws_many(ctx, files)
            for d in ds:
                d.free()
        t_decode = (time.perf_counter() - t0) / reps
        ctx.profile(True); ctx.profile_reset()
        ds = parquet.read_draws_many(ctx, files)
        prof = ctx.profile_get()
        ctx.profile(False)
        for d in ds:
            d.free()
        for f in files:
            f.close()
    best_a = min(ta)
    out = {
        "workload": (f"{len(paths)} REAL packaged corpus files" if real_dir is not None else
                     f"{len(paths)} corpus-shaped Parquet files (synthetic draws)") +
                    f", {total_pd} param-draws, {file_bytes} file bytes (SNAPPY + RLE_DICTIONARY), page cache warm",
        "identical_statistics": bool(same),
        "arrow_route_s": best_a[0], "arrow_decode_s": best_a[1], "arrow_upload_summarise_s": best_a[2],
        "native_route_s": min(tn),
        "native_phases_ms_median": {k: round(v, 3) for k, v in phases.items()},
        "native_decode_only_s": t_decode,
        "speedup_end_to_end": best_a[0] / min(tn),
        "param_draws_per_s_native": total_pd / min(tn), "param_draws_per_s_arrow": total_pd / best_a[0],
        "decode_kernels_ms": {k: v["total_ms"] for k, v in prof.items() if k.startswith("k_pq") or k == "k_gather_rows"},
        "decoded_GB_per_s_kernels": total_pd * 8 / 1e9 / (sum(v["total_ms"] for k, v in prof.items() if k.startswith("k_pq")) * 1e-3),
    }
    print(json.dumps(out))
    ctx.close()
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
