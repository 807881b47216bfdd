"""Native Parquet ingest: draws file -> device tensor, no pyarrow on the path.

Counterpart of the reference's `pq.read_table` / `pq.ParquetFile(...).iter_batches` + `to_numpy`
(src/mcmc_ref/store.py:79-95, src/mcmc_ref/convert.py:61-65, src/mcmc_ref/backends_numpy.py:35) for the
on-disk layout `draws/<model>.draws.parquet` (long table: `chain`, `draw`, one DOUBLE column per parameter).
Footer and page headers are parsed on the host by the C library, the page payloads are decompressed and
decoded by HIP kernels straight into HBM; the statistics then run on that tensor without a host round trip.
"""
from __future__ import annotations

import ctypes as C
import mmap
import os
from pathlib import Path
from typing import Iterable, Sequence

import numpy as np

from . import _ffi
from ._ffi import MCR_F64, MCR_PQ_F64, MCR_PQ_I64, DeviceBuffer, DeviceTensor, McrError, ParquetRequest

PHYSICAL_TYPES = {0: "BOOLEAN", 1: "INT32", 2: "INT64", 3: "INT96", 4: "FLOAT", 5: "DOUBLE", 6: "BYTE_ARRAY",
                  7: "FIXED_LEN_BYTE_ARRAY"}
NUMERIC = (1, 2, 4, 5)


class ParquetFile:
    """Parsed metadata of one Parquet file image (bytes, mmap or path).  Parsing needs no GPU."""

    def __init__(self, source, context: "_ffi.Context | None" = None):
        self.lib = _ffi.load_library()
        self.ctx = context
        self._mm = None
        if isinstance(source, (str, os.PathLike)):
            self.path = Path(source)
            with open(self.path, "rb") as fh:
                size = os.fstat(fh.fileno()).st_size
                if size == 0:
                    raise McrError(_ffi.MCR_EINVAL, f"parquet: {self.path} is empty")
                self._mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
            self._image = np.frombuffer(self._mm, dtype=np.uint8)
        else:
            self.path = None
            self._image = np.frombuffer(source, dtype=np.uint8)
        self.handle = C.c_void_p()
        rc = self.lib.mcr_parquet_open(context.handle if context else None, self._image.ctypes.data_as(C.c_void_p),
                                       self._image.size, C.byref(self.handle))
        if rc != _ffi.MCR_OK:
            msg = (self.lib.mcr_last_error(context.handle if context else None) or b"").decode()
            self.handle = None
            raise McrError(rc, msg)
        self.num_rows = int(self.lib.mcr_parquet_num_rows(self.handle))
        n = self.lib.mcr_parquet_num_columns(self.handle)
        self.column_names = [self.lib.mcr_parquet_column_name(self.handle, i).decode() for i in range(n)]
        self.column_types = [self.lib.mcr_parquet_column_type(self.handle, i) for i in range(n)]

    def pages(self) -> list[dict]:
        keys = ("column", "kind", "encoding", "codec", "payload_offset", "compressed_size", "uncompressed_size",
                "num_values", "first_row", "dictionary_page")
        out, info = [], np.zeros(10, dtype=np.int64)
        for i in range(self.lib.mcr_parquet_num_pages(self.handle)):
            self.lib.mcr_parquet_page_info(self.handle, i, info.ctypes.data_as(C.POINTER(C.c_int64)))
            out.append(dict(zip(keys, (int(v) for v in info))))
        return out

    def index(self, name: str) -> int:
        try:
            return self.column_names.index(name)
        except ValueError:
            raise KeyError(f"column {name!r} not in {self.path or 'parquet image'}") from None

    def close(self):
        if getattr(self, "handle", None):
            self.lib.mcr_parquet_close(self.handle)
            self.handle = None
        self._image = None
        if self._mm is not None:
            try:
                self._mm.close()
            except BufferError:      # a numpy view is still alive somewhere; the mapping goes with it
                pass
            self._mm = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def decode(ctx: "_ffi.Context", requests: Sequence[tuple[ParquetFile, int, int, C.c_void_p]]):
    """One upload + two launches for all (file, column index, out_kind, device pointer) requests."""
    arr = (ParquetRequest * max(len(requests), 1))()
    for i, (f, col, kind, ptr) in enumerate(requests):
        arr[i].file, arr[i].column, arr[i].out_kind = f.handle, col, kind
        arr[i].out_dev = ptr if isinstance(ptr, int) else ptr.value
    ctx._check(ctx.lib.mcr_parquet_decode(ctx.handle, arr, len(requests)))


def _layout(chain: np.ndarray, draw: np.ndarray):
    """(chain ids, order or None, draws per chain) -- the integer bookkeeping of `_chains_from_table`
    (src/mcmc_ref/convert.py:150-161); same rule as convert.chain_layout."""
    if chain.size:
        dc = np.diff(chain)
        if np.all(dc >= 0):                           # chain-major file (every packaged one): no sort needed
            cut = np.flatnonzero(dc) + 1
            starts = np.concatenate([[0], cut])
            counts = np.diff(np.concatenate([starts, [chain.size]]))
            dd = np.diff(draw)
            dd[cut - 1] = 0                           # steps across a chain boundary do not count
            if np.all(dd >= 0):
                return chain[starts].copy(), None, counts
    ids, counts = np.unique(chain, return_counts=True)
    order = None if chain.size == 0 else np.lexsort((draw, chain))
    return ids, order, counts


class _Arena:
    """One device allocation shared by the models of a batch (hipMalloc / hipFree cost ~0.1 ms each, which
    would dominate a 57-file corpus pass); freed when the last view is."""

    def __init__(self, ctx, nbytes: int):
        self.buf = DeviceBuffer(ctx, max(nbytes, 8))
        self.refs = 0

    def view(self, offset: int, nbytes: int) -> "_View":
        self.refs += 1
        return _View(self, offset, nbytes)

    def release(self):
        self.refs -= 1
        if self.refs <= 0:
            self.buf.free()


class _View:
    """Slice of an arena with DeviceBuffer's interface (ptr / download / free)."""

    def __init__(self, arena: _Arena, offset: int, nbytes: int):
        self.arena, self.ctx, self.nbytes = arena, arena.buf.ctx, nbytes
        self.ptr = C.c_void_p(arena.buf.ptr.value + offset)

    def download(self, dtype, count: int) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        if out.nbytes:
            self.ctx._check(self.ctx.lib.mcr_memcpy_d2h(self.ctx.handle, out.ctypes.data_as(C.c_void_p), self.ptr,
                                                        out.nbytes))
        return out

    def free(self):
        if self.arena is not None:
            self.arena.release()
            self.arena = None
            self.ptr = C.c_void_p()


class DeviceDraws:
    """Draws of one model in HBM: `tensor` is [P][M] f64 in (chain, draw) order; counts = draws per chain."""

    def __init__(self, tensor: DeviceTensor | None, buf, params: list[str], chain_ids: np.ndarray,
                 counts: np.ndarray):
        self.tensor, self.buf, self.params, self.chain_ids, self.counts = tensor, buf, params, chain_ids, counts

    @property
    def rectangular(self) -> bool:
        return len(self.counts) > 0 and bool(np.all(self.counts == self.counts[0]))

    def to_host(self) -> np.ndarray:
        M = int(self.counts.sum())
        return self.buf.download(np.float64, len(self.params) * M).reshape(len(self.params), M)

    def free(self):
        self.buf.free()


def read_draws_many(ctx: "_ffi.Context", sources: Sequence, params: Sequence[Iterable[str] | None] | None = None
                    ) -> list[DeviceDraws]:
    """Decodes many draws files with one batched decode call (all pages of all files in one grid)."""
    files = [s if isinstance(s, ParquetFile) else ParquetFile(s, ctx) for s in sources]
    owned = [not isinstance(s, ParquetFile) for s in sources]
    try:
        reqs, plan, bufs = [], [], []
        wants, sizes, colidx, ididx = [], [], [], []
        for k, f in enumerate(files):
            want = list(params[k]) if params is not None and params[k] is not None else \
                [n for n, t in zip(f.column_names, f.column_types) if n not in ("chain", "draw") and t in NUMERIC]
            wants.append(want)
            colidx.append([f.index(n) for n in want])      # KeyError for an unknown name BEFORE any device allocation
            ididx.append((f.index("chain"), f.index("draw")))
            sizes.append(len(want) * f.num_rows * 8)       # packed: same-shape neighbours form ONE [P][C][N] tensor
        arena = _Arena(ctx, sum(sizes))
        id_rows = sum(f.num_rows for f in files)
        ids_all = DeviceBuffer(ctx, max(2 * id_rows * 8, 8))
        off = ioff = 0
        for f, want, size, cols, (i_chain, i_draw) in zip(files, wants, sizes, colidx, ididx):
            M = f.num_rows
            buf = arena.view(off, len(cols) * M * 8)
            base, ibase = buf.ptr.value, ids_all.ptr.value + ioff * 8
            for j, c in enumerate(cols):
                reqs.append((f, c, MCR_PQ_F64, base + j * M * 8))
            reqs.append((f, i_chain, MCR_PQ_I64, ibase))
            reqs.append((f, i_draw, MCR_PQ_I64, ibase + M * 8))
            plan.append((want, M, buf, ioff))
            bufs.append(buf)
            off += size
            ioff += 2 * M
        try:
            decode(ctx, reqs)
            ids_host = ids_all.download(np.int64, 2 * id_rows)
        except Exception:
            for b in bufs:
                b.free()
            raise
        finally:
            ids_all.free()
        out = []
        for want, M, buf, ioff in plan:
            cd = ids_host[ioff:ioff + 2 * M]
            chain_ids, order, counts = _layout(cd[:M], cd[M:])
            if order is not None and want and M:
                dst = DeviceBuffer(ctx, len(want) * M * 8)
                order = np.ascontiguousarray(order, dtype=np.int64)
                ctx._check(ctx.lib.mcr_gather_rows_dev(ctx.handle, buf.ptr, len(want), M,
                                                       order.ctypes.data_as(C.POINTER(C.c_int64)), dst.ptr))
                buf.free()
                buf = dst
            tensor = None
            if len(counts) and np.all(counts == counts[0]):
                Cn, N = len(counts), int(counts[0])
                tensor = DeviceTensor(ctx, buf, (MCR_F64, Cn, N, len(want), N, 1, Cn * N))
            out.append(DeviceDraws(tensor, buf, want, chain_ids, counts))
        return out
    finally:
        for f, o in zip(files, owned):
            if o:
                f.close()


def read_draws(ctx: "_ffi.Context", source, params: Iterable[str] | None = None) -> DeviceDraws:
    return read_draws_many(ctx, [source], [params])[0]


def read_columns(ctx: "_ffi.Context", source, columns: Iterable[str] | None = None) -> dict[str, np.ndarray]:
    """Host arrays of numeric columns decoded on the device (ints as int64, floats as float64)."""
    f = source if isinstance(source, ParquetFile) else ParquetFile(source, ctx)
    try:
        names = list(columns) if columns is not None else [n for n, t in zip(f.column_names, f.column_types) if t in NUMERIC]
        M = f.num_rows
        buf = DeviceBuffer(ctx, max(len(names) * M * 8, 8))
        try:
            kinds = [MCR_PQ_I64 if f.column_types[f.index(n)] in (1, 2) else MCR_PQ_F64 for n in names]
            decode(ctx, [(f, f.index(n), k, buf.ptr.value + j * M * 8) for j, (n, k) in enumerate(zip(names, kinds))])
            raw = buf.download(np.int64, len(names) * M).reshape(len(names), M)
        finally:
            buf.free()
        return {n: (raw[j].copy() if k == MCR_PQ_I64 else raw[j].view(np.float64).copy())
                for j, (n, k) in enumerate(zip(names, kinds))}
    finally:
        if not isinstance(source, ParquetFile):
            f.close()


def _entry(r: dict, i: int, qs, diagnostics: bool) -> dict[str, float]:
    e = {"mean": float(r["mean"][i]), "std": float(r["std"][i])}
    for q, v in zip(qs, r["q"][i]):
        e[f"q{int(q * 100)}"] = float(v)
    if diagnostics:
        e.update(rhat=float(r["rhat"][i]), ess_bulk=float(r["ess_bulk"][i]), ess_tail=float(r["ess_tail"][i]))
    return e


_FS_FIELDS = (("mean", 0), ("std", 1), ("rhat", 4), ("ess_bulk", 5), ("ess_tail", 6))


FS_PHASES = ("open_ms", "read_pinned_parse_ms", "plan_ms", "upload_decode_layout_ms", "statistics_ms",
             "collect_ms", "close_ms", "total_ms")          # MCR_FS_PH_* of include/mcmcref_hip.h


def _summarize_paths(ctx: "_ffi.Context", paths: list[str], min_chains: int, qs: list[float], diagnostics: bool,
                     phases: dict | None = None):
    """All of summarize_files in ONE C call (mcr_summarize_files: mmap, parse, batched decode, layout check,
    pipelined statistics).  Returns None when a file needs the general route (rows out of (chain, draw) order,
    chains of unequal length)."""
    L = ctx.lib
    arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    q = np.ascontiguousarray(qs, dtype=np.float64)
    fs = C.c_void_p()
    rc = L.mcr_summarize_files(ctx.handle, arr, len(paths), int(min_chains), q.ctypes.data_as(C.POINTER(C.c_double)),
                               q.size, 1 if diagnostics else 0, C.byref(fs))
    if rc == _ffi.MCR_ELAYOUT:
        return None
    if rc != _ffi.MCR_OK:
        msg = (L.mcr_last_error(ctx.handle) or b"").decode()
        if rc in (_ffi.MCR_EMINCHAINS, _ffi.MCR_EMINCHAINS_ARG, _ffi.MCR_ENONFINITE):
            raise ValueError(msg.split(": ", 1)[-1] if rc == _ffi.MCR_EMINCHAINS else msg)
        if "cannot compute stats of empty columns" in msg:
            raise ValueError("cannot compute stats of empty columns")
        raise McrError(rc, msg)
    try:
        if phases is not None:
            ms = (C.c_double * len(FS_PHASES))()
            L.mcr_fileset_phases(fs, ms, len(FS_PHASES))
            phases.update(zip(FS_PHASES, (float(v) for v in ms)))
            phases["jobs"] = int(L.mcr_fileset_jobs(fs))
        nq = len(qs)
        qkeys = [f"q{int(v * 100)}" for v in qs]
        nfiles = L.mcr_fileset_size(fs)
        counts = [int(L.mcr_fileset_params(fs, i)) for i in range(nfiles)]
        total = sum(counts)
        rows = np.empty((max(total, 1), 10 + nq))                  # one export of the whole set (mcr_fileset_export)
        L.mcr_fileset_export(fs, rows.ctypes.data_as(C.POINTER(C.c_double)), total)
        need = int(L.mcr_fileset_names(fs, None, 0))
        buf = C.create_string_buffer(max(need, 1))
        L.mcr_fileset_names(fs, buf, need)
        names = buf.raw[:need].decode().split("\0")[:total]
        table = rows[:total].tolist()
        out, r = [], 0
        for P in counts:
            res = {}
            for name, row in zip(names[r:r + P], table[r:r + P]):
                e = {"mean": row[0], "std": row[1]}
                e.update(zip(qkeys, row[10:]))
                if diagnostics:
                    e["rhat"], e["ess_bulk"], e["ess_tail"] = row[3], row[4], row[5]
                res[name] = e
            out.append(res)
            r += P
        return out
    finally:
        L.mcr_fileset_free(fs)


def summarize_files(ctx: "_ffi.Context", sources: Sequence, params: Sequence[Iterable[str] | None] | None = None, *,
                    min_chains: int = 4, quantiles=(0.05, 0.5, 0.95), diagnostics: bool = True
                    ) -> list[dict[str, dict[str, float]]]:
    """File images -> per-parameter statistics without the draws ever visiting host memory decoded:
    one batched decode, then the models are pipelined through the summarise lanes (rolling window).

    diagnostics=False is `Backend.stats` (pooled mean / std / quantiles, any chain structure);
    diagnostics=True adds rhat / ess_bulk / ess_tail and needs equal-length chains (what
    `convert._compute_diagnostics` + `Backend.stats` give for the same file).
    """
    qs = list(quantiles)
    if sources and all(isinstance(s_, (str, os.PathLike)) for s_ in sources) and \
            (params is None or all(p_ is None for p_ in params)):
        fast = _summarize_paths(ctx, [os.fspath(s_) for s_ in sources], min_chains, qs, diagnostics)
        if fast is not None:
            return fast
    models = read_draws_many(ctx, sources, params)
    try:
        # Jobs: maximal runs of neighbouring models that sit back to back in the arena and share (C, N) are ONE
        # tensor with their parameters concatenated -- one kernel pipeline instead of one per model.
        jobs = []                                    # (first model, n models, tensor args, min_chains, diagnostics)
        for k, d in enumerate(models):
            P, M = len(d.params), int(d.counts.sum())
            if P == 0:
                continue
            if diagnostics and len(d.counts) < min_chains:
                raise ValueError(f"R-hat diagnostics require at least {min_chains} chains; got {len(d.counts)} chain(s)")
            if not diagnostics and M == 0:
                raise ValueError("cannot compute stats of empty columns")
            full = diagnostics and d.tensor is not None      # ragged chains: pooled stats here, diagnostics below
            shape = (len(d.counts), int(d.counts[0])) if full else (1, M)
            if jobs and M > 0:
                j = jobs[-1]
                last = models[j["first"] + j["count"] - 1]
                if (k == j["first"] + j["count"] and j["shape"] == shape and j["full"] == full
                        and isinstance(d.buf, _View) and isinstance(last.buf, _View)
                        and last.buf.ptr.value + len(last.params) * M * 8 == d.buf.ptr.value):
                    j["count"] += 1
                    j["P"] += P
                    continue
            jobs.append({"first": k, "count": 1, "shape": shape, "full": full, "P": P, "buf": d.buf})
        pending = []
        for j in jobs:
            Cn, N = j["shape"]
            t = DeviceTensor(ctx, j["buf"], (MCR_F64, Cn, N, j["P"], N, 1, Cn * N))
            if ctx.inflight >= _ffi.MCR_MAX_INFLIGHT:
                ctx.wait_one()
            try:
                pending.append(ctx.enqueue(t, min_chains=min_chains if j["full"] else 1, quantiles=qs,
                                           diagnostics=j["full"]))
            except McrError as exc:
                raise ValueError(exc.message) from exc
        try:
            ctx.wait()
        except McrError as exc:
            raise ValueError(exc.message) from exc
        results = [None] * len(models)
        for j, b in zip(jobs, pending):
            r = b.result()
            p0 = 0
            for k in range(j["first"], j["first"] + j["count"]):
                P = len(models[k].params)
                if P == 0:
                    continue
                results[k] = {key: (v[p0:p0 + P] if key != "q_lo" else v) for key, v in r.items()}
                p0 += P
        out = []
        for d, r in zip(models, results):
            if r is not None and diagnostics and d.tensor is None:
                x = d.to_host()
                off = np.concatenate([[0], np.cumsum(d.counts)])
                r = dict(r)
                for k in ("rhat", "ess_bulk", "ess_tail"):
                    r[k] = np.full(len(d.params), np.nan)
                for i in range(len(d.params)):
                    try:
                        g = ctx.diagnose_chains([x[i, off[c]:off[c + 1]] for c in range(len(d.counts))],
                                                min_chains=min_chains)
                    except McrError as exc:
                        raise ValueError(exc.message) from exc
                    for k in ("rhat", "ess_bulk", "ess_tail"):
                        r[k][i] = g[k]
            out.append({p: _entry(r, i, qs, diagnostics) for i, p in enumerate(d.params)})
        return out
    finally:
        try:
            if ctx.inflight:
                ctx.wait()
        except McrError:
            pass
        for d in models:
            d.free()
