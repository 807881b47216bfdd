import io, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "mcmc-db_amd")]
import pyarrow as pa, pyarrow.parquet as pq
from mcmc_ref_hip import _ffi, corpus, parquet
ctx = _ffi.Context(0)
imgs = []
for name, x in corpus.synthetic_corpus(seed=4711)[:20]:
    P, C, N = x.shape
    cols = {"chain": np.repeat(np.arange(C), N), "draw": np.tile(np.arange(N), C)}
    for i in range(P):
        cols[f"p[{i + 1}]"] = x[i].reshape(-1)
    b = io.BytesIO(); pq.write_table(pa.table(cols), b); imgs.append(b.getvalue())
files = [parquet.ParquetFile(i, ctx) for i in imgs]
buf = _ffi.DeviceBuffer(ctx, 64 << 20)
def run(sel, kind):
    reqs = []; off = 0
    for f in files:
        for j, n in enumerate(f.column_names):
            if sel(n):
                reqs.append((f, j, kind, buf.ptr.value + off)); off += f.num_rows * 8
    parquet.decode(ctx, reqs)
    ctx.profile(True); ctx.profile_reset()
    parquet.decode(ctx, reqs)
    pr = ctx.profile_get(); ctx.profile(False)
    return len(reqs), {k: round(v["total_ms"] * 1e3, 1) for k, v in pr.items()}
print("chain  ", run(lambda n: n == "chain", 1))
print("draw   ", run(lambda n: n == "draw", 1))
print("doubles", run(lambda n: n not in ("chain", "draw"), 0))
print("one dbl", run(lambda n: n == "p[1]", 0))
