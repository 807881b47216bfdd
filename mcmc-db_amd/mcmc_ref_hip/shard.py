"""Multi-GPU sharding of independent models / parameter blocks and the one collective of the path.

Every statistic is per (model, parameter) (src/mcmc_ref/convert.py:140-147 is an independent loop), so the work shards
with no data-path exchange: one process per GPU, whole models assigned by greedy LPT (`plan_shards`) or ONE large model
cut into contiguous parameter blocks (`param_block`), and a single all-gather of fixed-size per-parameter
summary records (16 doubles = 128 bytes) at the end.  The collective is `ncclAllGather` inside libmcmcref_hip
(`mcr_comm_all_gather`, RCCL over xGMI): no torch, no MPI.  Ranks find each other through the environment a
`torch.distributed.run` / torchrun launch provides (RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT): rank 0
hands the 128-byte ncclUniqueId to the others through files keyed on those (`exchange_unique_id`).

The gather is latency-bound (packaged corpus: 460 records = 59 KB; a 10 000-parameter model: 1.3 MB), so xGMI link
bandwidth is irrelevant to it.  What to expect from N GPUs: independent models of C1 size scale weakly (each rank
runs the single-GPU pipeline; the gather adds tens of microseconds per run); the packaged corpus is 0.3 ms of work on
ONE GPU, so sharding it is launch- and gather-latency bound by construction and does not speed up.

`comm` arguments take anything with `world`, `rank`, `all_gather(np.ndarray) -> np.ndarray[world, ...]`: the RCCL
`Communicator` below, or a stand-in (the CPU tests rehearse the control flow with a gloo-backed one).
"""
from __future__ import annotations

import ctypes as C
import os
import tempfile
import time
from collections.abc import Sequence
from pathlib import Path

import numpy as np

RECORD_FIELDS = ("mean", "std", "q5", "q50", "q95", "rhat_bulk", "rhat_tail", "rhat", "ess_bulk", "ess_tail",
                 "lag_bulk", "lag_tail", "n_chains", "n_draws", "param_idx", "model_idx")
RECORD_DOUBLES = len(RECORD_FIELDS)          # 16 doubles = 128 bytes (MCR_RECORD_DOUBLES)
COMM_ID_BYTES = 128


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


def param_block(P: int, world: int, rank: int) -> tuple[int, int]:
    """[p0, p1): the contiguous column block of rank `rank` when ONE model's P parameters are split over `world`
    ranks (no halo, no exchange; block sizes differ by at most one)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad world / rank")
    return P * rank // world, P * (rank + 1) // world


# ---- the communicator ------------------------------------------------------------------------------------------
# Rendezvous of the ranks of ONE launch, torch-free: rank 0 creates the 128-byte ncclUniqueId and hands it to the other
# ranks through small files in a directory every rank of the launch can see (the local temp dir for a single node,
# MCR_COMM_DIR on a shared file system otherwise).
#
# Which launch a file belongs to (`comm_key`):
#   * MCR_COMM_KEY, when exported, is the key -- any launcher (srun, mpirun, a shell loop) can provide one;
#   * otherwise MASTER_ADDR / MASTER_PORT / world size / TORCHELASTIC_RESTART_COUNT, plus TORCHELASTIC_RUN_ID when the
#     launcher was given a real run id (`torchrun --rdzv-id X`), plus -- only when the run id is absent or torchrun's
#     default "none" -- the pid of the launcher (os.getppid()): `python -m torch.distributed.run` / torchrun start the
#     ranks as direct children of one agent process, so they agree on it.  A launcher that puts a shell between itself
#     and each rank must export MCR_COMM_KEY (or a run id); the TimeoutError of a rank that never finds rank 0 prints
#     the resolved path and the key parts, so a disagreement is visible at once.
#
# The hand-over itself (`exchange_unique_id`) cannot pick up a file a crashed earlier launch left under the same key:
# every rank r > 0 publishes a fresh random nonce (`<base>.hello.<r>`), rank 0 publishes `id || nonce_1 .. nonce_{w-1}`
# (`<base>.id`, atomic rename, republished when a hello changes under it), rank r accepts an id file only if it carries
# ITS nonce and then acknowledges with `<base>.ack.<r>`; rank 0 returns once every ack matches and removes the files.
COMM_NONCE_BYTES = 16


def comm_key(world: int) -> tuple[str, list[str]]:
    """(key, parts) of this launch; see the block comment above for the resolution order."""
    explicit = os.environ.get("MCR_COMM_KEY")
    if explicit:
        parts = ["MCR_COMM_KEY=" + explicit]
        key = explicit
    else:
        run_id = os.environ.get("TORCHELASTIC_RUN_ID", "")
        parts = ["MASTER_ADDR=" + os.environ.get("MASTER_ADDR", "127.0.0.1"),
                 "MASTER_PORT=" + os.environ.get("MASTER_PORT", "0"), f"world={world}",
                 "TORCHELASTIC_RESTART_COUNT=" + os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")]
        if run_id and run_id != "none":
            parts.append("TORCHELASTIC_RUN_ID=" + run_id)
        else:
            parts.append(f"launcher_pid={os.getppid()}")
        key = "_".join(x.split("=", 1)[1] for x in parts)
    return "".join(ch if ch.isalnum() or ch in "._-" else "-" for ch in key), parts


def _id_file(world: int) -> Path:
    """Base path of the rendezvous files of this launch (`<base>.id`, `<base>.hello.<r>`, `<base>.ack.<r>`)."""
    d = Path(os.environ.get("MCR_COMM_DIR", tempfile.gettempdir()))
    return d / f"mcr_rccl_id_{comm_key(world)[0]}"


def _publish(path: Path, payload: bytes) -> None:
    tmp = path.with_name(path.name + f".{os.getpid()}.tmp")
    tmp.write_bytes(payload)
    os.replace(tmp, path)                               # atomic: a reader sees the whole payload or no file


def _read(path: Path) -> bytes:
    try:
        return path.read_bytes()
    except (FileNotFoundError, PermissionError):
        return b""


def exchange_unique_id(rank: int, world: int, make_id, *, timeout: float = 300.0, base: Path | None = None,
                       poll: float = 0.005) -> bytes:
    """Rank 0 calls `make_id()` (-> COMM_ID_BYTES bytes) and every rank returns those bytes.  File protocol described
    above; raises TimeoutError (naming the path and the key parts) when the other side never shows up."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad world / rank")
    if world == 1:
        return bytes(make_id())
    lws = os.environ.get("LOCAL_WORLD_SIZE")
    if base is None and lws and int(lws) != world and not os.environ.get("MCR_COMM_DIR"):
        raise RuntimeError(f"multi-node launch (LOCAL_WORLD_SIZE={lws}, WORLD_SIZE={world}): export MCR_COMM_DIR = a "
                           "directory every node sees; the default temp dir is node-local")
    base = _id_file(world) if base is None else Path(base)
    idp = base.with_name(base.name + ".id")
    hello = lambda r: base.with_name(base.name + f".hello.{r}")       # noqa: E731
    ack = lambda r: base.with_name(base.name + f".ack.{r}")           # noqa: E731
    t0 = time.monotonic()

    def expired(what: str):
        if time.monotonic() - t0 > timeout:
            raise TimeoutError(f"rank {rank}/{world}: {what} after {timeout:.0f} s; rendezvous files {base}.* "
                               f"(key parts: {', '.join(comm_key(world)[1])}).  Ranks behind different parent processes "
                               "need MCR_COMM_KEY (or torchrun --rdzv-id) to agree on the key")
        time.sleep(poll)

    if rank == 0:
        uid = bytes(make_id())
        if len(uid) != COMM_ID_BYTES:
            raise ValueError(f"make_id() returned {len(uid)} bytes, want {COMM_ID_BYTES}")
        published: list[bytes] | None = None
        try:
            while True:
                nonces = [_read(hello(r)) for r in range(1, world)]
                if all(len(x) == COMM_NONCE_BYTES for x in nonces):
                    if nonces != published:             # first time, or a rank replaced a stale hello by its own
                        _publish(idp, uid + b"".join(nonces))
                        published = nonces
                    if all(_read(ack(r)) == nonces[r - 1] for r in range(1, world)):
                        return uid
                expired("not every rank said hello / acknowledged the id")
        finally:
            for f in [idp] + [hello(r) for r in range(1, world)] + [ack(r) for r in range(1, world)]:
                try:
                    f.unlink()
                except FileNotFoundError:
                    pass
    nonce = os.urandom(COMM_NONCE_BYTES)
    _publish(hello(rank), nonce)
    want = COMM_ID_BYTES + COMM_NONCE_BYTES * (world - 1)
    o = COMM_ID_BYTES + COMM_NONCE_BYTES * (rank - 1)
    while True:
        raw = _read(idp)
        if len(raw) == want and raw[o:o + COMM_NONCE_BYTES] == nonce:
            _publish(ack(rank), nonce)
            return raw[:COMM_ID_BYTES]
        if not hello(rank).exists():                   # a crashed launch's rank 0 clean-up may have taken it: say hello again
            _publish(hello(rank), nonce)
        expired("no RCCL id carrying this rank's nonce from rank 0")


class Communicator:
    """RCCL communicator of libmcmcref_hip over the ranks of a torchrun-style launch (one process per GPU)."""

    def __init__(self, ctx, world: int | None = None, rank: int | None = None, timeout: float = 300.0):
        from . import _ffi
        self.ctx = ctx
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        L = ctx.lib

        def make_id() -> bytes:
            uid = (C.c_ubyte * COMM_ID_BYTES)()
            rc = L.mcr_comm_unique_id(uid, COMM_ID_BYTES)
            if rc != _ffi.MCR_OK:
                raise _ffi.McrError(rc, (L.mcr_last_error(None) or b"").decode())
            return bytes(uid)

        raw = exchange_unique_id(self.rank, self.world, make_id, timeout=timeout)
        uid = (C.c_ubyte * COMM_ID_BYTES).from_buffer_copy(raw)
        h = C.c_void_p()
        ctx._check(L.mcr_comm_init(ctx.handle, uid, self.world, self.rank, C.byref(h)))   # collective
        self.handle = h

    @property
    def has_deadline(self) -> bool:
        """True: every collective on this communicator gives up after MCR_COMM_TIMEOUT_S (default 300 s) with an
        McrError(MCR_ECOMM) instead of waiting for a dead peer for ever (mcr_comm_has_deadline)."""
        return bool(self.ctx.lib.mcr_comm_has_deadline(self.handle))

    def all_gather(self, arr: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(arr, dtype=np.float64)
        out = np.empty((self.world,) + a.shape, dtype=np.float64)
        dp = C.POINTER(C.c_double)
        self.ctx._check(self.ctx.lib.mcr_comm_all_gather(self.handle, a.ctypes.data_as(dp), a.size, out.ctypes.data_as(dp)))
        return out

    def all_reduce(self, vals, op: str = "max") -> np.ndarray:
        a = np.array(vals, dtype=np.float64, ndmin=1)
        self.ctx._check(self.ctx.lib.mcr_comm_all_reduce(self.handle, a.ctypes.data_as(C.POINTER(C.c_double)), a.size,
                                                         {"sum": 0, "max": 1, "min": 2}[op]))
        return a

    def barrier(self) -> None:
        self.ctx._check(self.ctx.lib.mcr_comm_barrier(self.handle))

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.ctx.lib.mcr_comm_free(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


# ---- records ---------------------------------------------------------------------------------------------------
def pack_records(summary: dict, model_idx: int, n_chains: int, n_draws: int, param0: int = 0) -> np.ndarray:
    """[P][16] float64 records from a summarize() result computed with quantiles (0.05, 0.5, 0.95); `param0` = index of
    the first parameter when the result covers one block of a P-split model."""
    P = len(summary["mean"])
    rec = np.empty((P, RECORD_DOUBLES), dtype=np.float64)
    q = summary["q"]
    cols = [summary["mean"], summary["std"], q[:, 0], q[:, 1], q[:, 2], summary["rhat_bulk"],
            summary["rhat_tail"], summary["rhat"], summary["ess_bulk"], summary["ess_tail"],
            summary["lag_bulk"].astype(np.float64), summary["lag_tail"].astype(np.float64),
            np.full(P, float(n_chains)), np.full(P, float(n_draws)), np.arange(P, dtype=np.float64) + float(param0),
            np.full(P, float(model_idx))]
    for j, c in enumerate(cols):
        rec[:, j] = c
    return rec


def gather_records(local: np.ndarray, comm=None) -> np.ndarray:
    """all-gather of every rank's [n_r][16] records -> [sum n_r][16] on every rank, ordered by (model_idx, param_idx).
    comm = None: single process.  Two `all_gather` calls: the record counts (so that every rank pads to the same
    length), then the padded records."""
    local = np.ascontiguousarray(local, dtype=np.float64).reshape(-1, RECORD_DOUBLES)
    if comm is None:
        out = local
    else:
        counts = comm.all_gather(np.array([float(local.shape[0])])).reshape(-1).astype(np.int64)
        cap = int(counts.max()) if counts.size else 0
        padded = np.full((max(cap, 1), RECORD_DOUBLES), np.nan)
        padded[:local.shape[0]] = local
        allp = comm.all_gather(padded).reshape(comm.world, max(cap, 1), RECORD_DOUBLES)
        out = np.concatenate([allp[r, :int(counts[r])] for r in range(comm.world)], axis=0)
    if out.shape[0]:
        order = np.lexsort((out[:, RECORD_FIELDS.index("param_idx")], out[:, RECORD_FIELDS.index("model_idx")]))
        out = out[order]
    return out


def _agree(comm, error: BaseException | None, what: str) -> None:
    """Every rank learns whether any rank failed BEFORE the data collective, so that one bad file or model ends the
    job everywhere with a message instead of leaving the healthy ranks blocked inside the gather."""
    if comm is None:
        if error is not None:
            raise error
        return
    flags = comm.all_gather(np.array([0.0 if error is None else 1.0])).reshape(-1)
    if error is not None:
        raise error
    bad = [int(r) for r in np.nonzero(flags)[0]]
    if bad:
        raise RuntimeError(f"{what} failed on rank(s) {bad}; this rank ({comm.rank}) stops with them")


def summarize_models(ctx, models: Sequence[tuple[np.ndarray, str]], comm=None, min_chains: int = 4,
                     batch: bool = True) -> np.ndarray:
    """Shard `models` ((array, layout) pairs) over the ranks of `comm`, summarise this rank's share on `ctx` and
    gather all records on every rank.

    batch=True: models of this rank that share (chains, draws, dtype) and are C-contiguous [P][C][N]
    are concatenated along the parameter axis and go through ONE kernel pipeline (the corpus has 55
    models of 10 x 1000 draws: 55 pipelines of ~12 launches become one).  Parameters are
    independent, so the results are identical to per-model calls.
    """
    from . import _ffi
    world, rank = (comm.world, comm.rank) if comm is not None else (1, 0)
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
        try:
            ctx.wait()
            for members, bufs, t, dims in pending:
                r = bufs.result()
                p0 = 0
                for i in members:
                    P = models[i][0].shape[models[i][1].index("p")]
                    part = {k: (v[p0:p0 + P] if k != "q_lo" else v) for k, v in r.items()}
                    recs.append(pack_records(part, i, dims[0], dims[1]))
                    p0 += P
        finally:
            for _, _, t, _ in pending:
                t.free()
            pending.clear()

    error = None
    try:
        for key, members in groups.items():
            if key[0] == "b" and len(members) > 1:
                big = np.concatenate([models[i][0] for i in members], axis=0)
                t = ctx.upload(big, "pcn")
            else:
                arr, layout = models[members[0]]
                t = ctx.upload(np.ascontiguousarray(arr), layout)
            pending.append((members, None, t, t.shape_cnp))
            pending[-1] = (members, ctx.enqueue(t, min_chains=min_chains), t, t.shape_cnp)
            if len(pending) == _ffi.MCR_MAX_INFLIGHT:
                drain()
        drain()
    except Exception as exc:  # noqa: BLE001 - settled with the other ranks below
        error = exc
        for _, _, t, _ in pending:
            t.free()
        pending.clear()
    _agree(comm, error, "summarize_models")
    local = np.concatenate(recs, axis=0) if recs else np.empty((0, RECORD_DOUBLES))
    return gather_records(local, comm)


def summarize_param_split(ctx, draws: np.ndarray, comm=None, min_chains: int = 4, model_idx: int = 0) -> np.ndarray:
    """ONE model [P][C][N] (any array-like that slices along its first axis, e.g. a memmap) split over the ranks by
    contiguous parameter blocks: each rank uploads and summarises only draws[p0:p1], the records carry
    global parameter indices, one gather."""
    world, rank = (comm.world, comm.rank) if comm is not None else (1, 0)
    P, Cn, N = draws.shape
    p0, p1 = param_block(P, world, rank)
    error, recs = None, np.empty((0, RECORD_DOUBLES))
    try:
        if p1 > p0:
            r = ctx.summarize(np.ascontiguousarray(draws[p0:p1]), "pcn", min_chains=min_chains)
            recs = pack_records(r, model_idx, Cn, N, param0=p0)
    except Exception as exc:  # noqa: BLE001
        error = exc
    _agree(comm, error, "summarize_param_split")
    return gather_records(recs, comm)


def summarize_paths(ctx, paths: Sequence, comm=None, min_chains: int = 4) -> np.ndarray:
    """The corpus straight from disk on N GPUs: `paths` (draws/<model>.draws.parquet files, the same list on every
    rank) are assigned to ranks by greedy LPT over file size, each rank turns its share into statistics with ONE
    `mcr_summarize_files` call (native Parquet ingest + kernels), and one all-gather of the 128-byte records puts
    every model's summary on every rank.  `model_idx` in the records indexes `paths`."""
    from . import _ffi, parquet
    world, rank = (comm.world, comm.rank) if comm is not None else (1, 0)
    paths = [os.fspath(p) for p in paths]
    mine = plan_shards([float(os.path.getsize(p)) for p in paths], world)[rank]
    qs = np.array([0.05, 0.5, 0.95])
    recs = []
    error = None
    try:
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
                decoded = parquet.read_draws_many(ctx, [paths[i] for i in mine])
                try:
                    for d, i in zip(decoded, mine):
                        if d.tensor is None:
                            raise ValueError(f"{paths[i]}: chains of unequal length")
                        r = ctx.summarize(d.tensor, min_chains=min_chains)
                        recs.append(pack_records(r, i, len(d.counts), int(d.counts[0])))
                finally:
                    for d in decoded:
                        d.free()
            else:
                ctx._check(rc)
    except Exception as exc:  # noqa: BLE001
        error = exc
    _agree(comm, error, "summarize_paths")
    local = np.concatenate(recs, axis=0) if recs else np.empty((0, RECORD_DOUBLES))
    return gather_records(local, comm)
