"""Multi-GPU sharding of independent models and the one collective of the path.

Every statistic is per (model, parameter) (src/mcmc_ref/convert.py:140-147 is an independent
loop), so models shard embarrassingly: one process per GPU, no data-path collective, and a single
all_gather of fixed-size per-parameter summary records at the end (RCCL over xGMI when the
process group backend is "nccl"; gloo on CPU for tests).  The gather is latency-bound: the whole
packaged corpus is 460 records x 128 B = 59 KB.
"""
from __future__ import annotations

from collections.abc import Sequence

import numpy as np

RECORD_FIELDS = ("mean", "std", "q5", "q50", "q95", "rhat_bulk", "rhat_tail", "rhat", "ess_bulk", "ess_tail",
                 "lag_bulk", "lag_tail", "n_chains", "n_draws", "param_idx", "model_idx")
RECORD_DOUBLES = len(RECORD_FIELDS)          # 16 doubles = 128 bytes


def plan_shards(costs: Sequence[float], world: int) -> list[list[int]]:
    """Greedy longest-processing-time assignment of models (by cost, e.g. C*N*P) to `world` ranks.

    Deterministic: ties broken by model index, so every rank computes the same plan.
    """
    if world < 1:
        raise ValueError("world must be >= 1")
    load = [0.0] * world
    shards: list[list[int]] = [[] for _ in range(world)]
    for i in sorted(range(len(costs)), key=lambda k: (-float(costs[k]), k)):
        r = min(range(world), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += float(costs[i])
    for s in shards:
        s.sort()
    return shards


def pack_records(summary: dict, model_idx: int, n_chains: int, n_draws: int) -> np.ndarray:
    """[P][16] float64 records from a summarize() result computed with quantiles (0.05, 0.5, 0.95)."""
    P = len(summary["mean"])
    rec = np.empty((P, RECORD_DOUBLES), dtype=np.float64)
    q = summary["q"]
    cols = [summary["mean"], summary["std"], q[:, 0], q[:, 1], q[:, 2], summary["rhat_bulk"],
            summary["rhat_tail"], summary["rhat"], summary["ess_bulk"], summary["ess_tail"],
            summary["lag_bulk"].astype(np.float64), summary["lag_tail"].astype(np.float64),
            np.full(P, float(n_chains)), np.full(P, float(n_draws)), np.arange(P, dtype=np.float64),
            np.full(P, float(model_idx))]
    for j, c in enumerate(cols):
        rec[:, j] = c
    return rec


def gather_records(local: np.ndarray, dist=None, device=None, force: bool = False) -> np.ndarray:
    """all_gather of every rank's [n_r][16] records -> [sum n_r][16] on every rank, ordered by
    (model_idx, param_idx).  `dist` is torch.distributed (initialised) or None for a single process.
    force=True runs the collective even for a world of one (exercises the RCCL path on a 1-GPU box)."""
    local = np.ascontiguousarray(local, dtype=np.float64).reshape(-1, RECORD_DOUBLES)
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        out = local
    else:
        import torch
        world = dist.get_world_size()
        dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
        n = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
        nmax = n.clone()
        dist.all_reduce(nmax, op=dist.ReduceOp.MAX)
        cap = int(nmax.item())
        padded = torch.full((cap + 1, RECORD_DOUBLES), float("nan"), dtype=torch.float64, device=dev)
        padded[0, 0] = float(local.shape[0])                      # row 0 carries the valid count
        if local.shape[0]:
            padded[1:1 + local.shape[0]] = torch.from_numpy(local).to(dev)
        allp = torch.empty((world * (cap + 1), RECORD_DOUBLES), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allp, padded)      # concatenation along dim 0 (gloo and nccl)
        allp = allp.cpu().numpy().reshape(world, cap + 1, RECORD_DOUBLES)
        out = np.concatenate([allp[r, 1:1 + int(allp[r, 0, 0])] for r in range(world)], axis=0)
    if out.shape[0]:
        order = np.lexsort((out[:, RECORD_FIELDS.index("param_idx")], out[:, RECORD_FIELDS.index("model_idx")]))
        out = out[order]
    return out


def summarize_models(ctx, models: Sequence[tuple[np.ndarray, str]], rank: int = 0, world: int = 1, dist=None,
                     min_chains: int = 4, batch: bool = True) -> np.ndarray:
    """Shard `models` ((array, layout) pairs) over ranks, summarise this rank's share on `ctx` and
    gather all records on every rank.

    batch=True: models of this rank that share (chains, draws, dtype) and are C-contiguous [P][C][N]
    are concatenated along the parameter axis and go through ONE kernel pipeline (the corpus has 55
    models of 10 x 1000 draws: 55 pipelines of ~12 launches become one).  Parameters are
    independent, so the results are identical to per-model calls.
    """
    from . import _ffi
    costs = [float(np.prod(a.shape)) for a, _ in models]
    mine = plan_shards(costs, world)[rank]
    groups: dict = {}
    for i in mine:
        arr, layout = models[i]
        if batch and layout == "pcn" and arr.ndim == 3 and arr.flags.c_contiguous and arr.shape[0] > 0:
            groups.setdefault(("b", arr.shape[1], arr.shape[2], arr.dtype.str), []).append(i)
        else:
            groups[("s", i)] = [i]
    recs = []
    pending = []

    def drain():
        ctx.wait()
        for members, bufs, t, dims in pending:
            r = bufs.result()
            p0 = 0
            for i in members:
                P = models[i][0].shape[models[i][1].index("p")]
                part = {k: (v[p0:p0 + P] if k != "q_lo" else v) for k, v in r.items()}
                recs.append(pack_records(part, i, dims[0], dims[1]))
                p0 += P
            t.free()
        pending.clear()

    for key, members in groups.items():
        if key[0] == "b" and len(members) > 1:
            big = np.concatenate([models[i][0] for i in members], axis=0)
            t = ctx.upload(big, "pcn")
        else:
            arr, layout = models[members[0]]
            t = ctx.upload(np.ascontiguousarray(arr), layout)
        pending.append((members, ctx.enqueue(t, min_chains=min_chains), t, t.shape_cnp))
        if len(pending) == _ffi.MCR_MAX_INFLIGHT:
            drain()
    drain()
    local = np.concatenate(recs, axis=0) if recs else np.empty((0, RECORD_DOUBLES))
    return gather_records(local, dist)


def summarize_paths(ctx, paths: Sequence, rank: int = 0, world: int = 1, dist=None, min_chains: int = 4,
                    device=None) -> np.ndarray:
    """The corpus straight from disk on N GPUs: `paths` (draws/<model>.draws.parquet files, the same list on every
    rank) are assigned to ranks by greedy LPT over file size, each rank turns its share into statistics with ONE
    `mcr_summarize_files` call (native Parquet ingest + kernels), and one all_gather of the 128-byte records puts
    every model's summary on every rank.  `model_idx` in the records indexes `paths`."""
    import ctypes as C
    import os
    from . import _ffi, parquet
    paths = [os.fspath(p) for p in paths]
    mine = plan_shards([float(os.path.getsize(p)) for p in paths], world)[rank]
    qs = np.array([0.05, 0.5, 0.95])
    recs = []
    if mine:
        L = ctx.lib
        arr = (C.c_char_p * len(mine))(*[paths[i].encode() for i in mine])
        fs = C.c_void_p()
        rc = L.mcr_summarize_files(ctx.handle, arr, len(mine), int(min_chains), qs.ctypes.data_as(C.POINTER(C.c_double)),
                                   3, 1, C.byref(fs))
        if rc == _ffi.MCR_OK:
            try:
                for k, i in enumerate(mine):
                    P = int(L.mcr_fileset_params(fs, k))
                    if P == 0:
                        continue
                    fld = lambda f, w=1: np.ctypeslib.as_array(L.mcr_fileset_field(fs, k, f), shape=(P * w,)).copy()  # noqa: E731
                    q = fld(2, 3).reshape(P, 3)
                    rec = np.empty((P, RECORD_DOUBLES))
                    cols = [fld(0), fld(1), q[:, 0], q[:, 1], q[:, 2], fld(7), fld(8), fld(4), fld(5), fld(6), fld(9), fld(10),
                            np.full(P, float(L.mcr_fileset_chains(fs, k))), np.full(P, float(L.mcr_fileset_draws(fs, k))),
                            np.arange(P, dtype=np.float64), np.full(P, float(i))]
                    for j, c in enumerate(cols):
                        rec[:, j] = c
                    recs.append(rec)
            finally:
                L.mcr_fileset_free(fs)
        elif rc == _ffi.MCR_ELAYOUT:          # shuffled rows somewhere in the share: the per-file route (device gather)
            for d, i in zip(parquet.read_draws_many(ctx, [paths[i] for i in mine]), mine):
                try:
                    if d.tensor is None:
                        raise ValueError(f"{paths[i]}: chains of unequal length")
                    r = ctx.summarize(d.tensor, min_chains=min_chains)
                    recs.append(pack_records(r, i, len(d.counts), int(d.counts[0])))
                finally:
                    d.free()
        else:
            ctx._check(rc)
    local = np.concatenate(recs, axis=0) if recs else np.empty((0, RECORD_DOUBLES))
    return gather_records(local, dist, device=device)
