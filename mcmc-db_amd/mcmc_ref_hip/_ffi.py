"""ctypes binding of libmcmcref_hip.so (include/mcmcref_hip.h).  No torch, no fallback.

If the shared library or a HIP device is missing every entry point raises
`HipUnavailableError`: the product path never silently computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import threading
from pathlib import Path

import numpy as np

MCR_OK = 0
MCR_EINVAL, MCR_EMINCHAINS, MCR_EMINCHAINS_ARG, MCR_ENONFINITE = -1, -2, -3, -4
MCR_EHIP, MCR_ENOMEM, MCR_ENODEVICE, MCR_ECOMM, MCR_ELAYOUT = -5, -6, -7, -8, -9
MCR_F64, MCR_F32 = 0, 1
MCR_MAX_QUANTILES = 32
MCR_MAX_INFLIGHT = 8

_PKG = Path(__file__).resolve().parent
DEFAULT_LIB = _PKG.parent / "lib" / "libmcmcref_hip.so"


class HipUnavailableError(RuntimeError):
    """libmcmcref_hip.so could not be loaded, or no MI355X/HIP device is usable."""


class McrError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"[mcr {code}] {message}")
        self.code = code
        self.message = message


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)


class Summary(C.Structure):
    _fields_ = [(n, _dp) for n in ("mean", "std", "q", "median", "rhat", "rhat_bulk", "rhat_tail",
                                   "ess_bulk", "ess_tail")] + \
               [("lag_bulk", _ip), ("lag_tail", _ip), ("q_lo", _ip)]


class ModelDesc(C.Structure):
    _fields_ = [("draws_dev", C.c_void_p), ("dtype", C.c_int), ("min_chains", C.c_int),
                ("C", C.c_int64), ("N", C.c_int64), ("P", C.c_int64),
                ("stride_c", C.c_int64), ("stride_n", C.c_int64), ("stride_p", C.c_int64)]


class ParquetRequest(C.Structure):
    _fields_ = [("file", C.c_void_p), ("column", C.c_int), ("out_kind", C.c_int), ("out_dev", C.c_void_p)]


MCR_PQ_F64, MCR_PQ_I64 = 0, 1


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("total_ms", C.c_double)]


# every symbol include/mcmcref_hip.h declares: (restype, argtypes)
_I64 = C.c_int64
_TENSOR = [C.c_void_p, C.c_int, _I64, _I64, _I64, _I64, _I64, _I64]
SYMBOLS = {
    "mcr_version": (C.c_int, []),
    "mcr_device_count": (C.c_int, []),
    "mcr_init": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "mcr_free": (None, [C.c_void_p]),
    "mcr_last_error": (C.c_char_p, [C.c_void_p]),
    "mcr_set_workspace_limit": (C.c_int, [C.c_void_p, C.c_size_t]),
    "mcr_rho_guard_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "mcr_plan_chunks": (C.c_int, [C.c_void_p] + [C.c_int64] * 6 + [C.c_int, C.POINTER(C.c_int64)]),
    "mcr_dev_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "mcr_dev_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mcr_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "mcr_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "mcr_sync": (C.c_int, [C.c_void_p]),
    "mcr_summarize": (C.c_int, [C.c_void_p] + _TENSOR + [C.c_int, _dp, C.c_int, C.POINTER(Summary)]),
    "mcr_summarize_dev": (C.c_int, [C.c_void_p] + _TENSOR + [C.c_int, _dp, C.c_int, C.POINTER(Summary)]),
    "mcr_summarize_enqueue": (C.c_int, [C.c_void_p] + _TENSOR + [C.c_int, _dp, C.c_int, C.POINTER(Summary)]),
    "mcr_summarize_models": (C.c_int, [C.c_void_p, C.POINTER(ModelDesc), C.c_int, _dp, C.c_int, C.POINTER(Summary)]),
    "mcr_summarize_wait": (C.c_int, [C.c_void_p]),
    "mcr_summarize_wait_one": (C.c_int, [C.c_void_p]),
    "mcr_diagnose_chains": (C.c_int, [C.c_void_p, _dp, _ip, C.c_int, C.c_int, C.POINTER(Summary),
                                      _dp, _dp, _dp, _dp]),
    "mcr_basic_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _I64, _dp, _dp]),
    "mcr_moments_dev": (C.c_int, [C.c_void_p] + _TENSOR + [_dp, _dp]),
    "mcr_compare": (C.c_int, [C.c_void_p, _dp, _dp, _I64, C.c_double, _dp, C.POINTER(C.c_uint8)]),
    "mcr_two_sample": (C.c_int, [C.c_void_p, _dp, _I64, _dp, _I64, _I64, _dp, _dp]),
    "mcr_covariance": (C.c_int, [C.c_void_p, _dp, _I64, _I64, _dp]),
    "mcr_covariance_dev": (C.c_int, [C.c_void_p, C.c_void_p, _I64, _I64, C.c_void_p]),
    "mcr_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "mcr_profile_reset": (C.c_int, [C.c_void_p]),
    "mcr_profile_get": (C.c_int, [C.c_void_p, C.POINTER(KernelTime), C.c_int, C.POINTER(C.c_int)]),
    "mcr_fill_synthetic": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _I64, _I64, _I64, C.c_uint64]),
    "mcr_fill_synthetic_at": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _I64, _I64, _I64, _I64, C.c_uint64]),
    "mcr_hbm_probe": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, _dp, _dp]),
    "mcr_comm_unique_id": (C.c_int, [C.c_void_p, C.c_size_t]),
    "mcr_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "mcr_comm_free": (None, [C.c_void_p]),
    "mcr_comm_world": (C.c_int, [C.c_void_p]),
    "mcr_comm_rank": (C.c_int, [C.c_void_p]),
    "mcr_comm_has_deadline": (C.c_int, [C.c_void_p]),
    "mcr_comm_all_gather": (C.c_int, [C.c_void_p, _dp, _I64, _dp]),
    "mcr_comm_all_reduce": (C.c_int, [C.c_void_p, _dp, _I64, C.c_int]),
    "mcr_comm_barrier": (C.c_int, [C.c_void_p]),
    "mcr_parquet_open": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "mcr_parquet_close": (None, [C.c_void_p]),
    "mcr_parquet_num_rows": (C.c_int64, [C.c_void_p]),
    "mcr_parquet_num_columns": (C.c_int, [C.c_void_p]),
    "mcr_parquet_column_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "mcr_parquet_column_type": (C.c_int, [C.c_void_p, C.c_int]),
    "mcr_parquet_num_pages": (C.c_int, [C.c_void_p]),
    "mcr_parquet_page_info": (C.c_int, [C.c_void_p, C.c_int, _ip]),
    "mcr_parquet_decode": (C.c_int, [C.c_void_p, C.POINTER(ParquetRequest), C.c_int]),
    "mcr_gather_rows_dev": (C.c_int, [C.c_void_p, C.c_void_p, _I64, _I64, _ip, C.c_void_p]),
    "mcr_summarize_files": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.c_int, C.c_int, _dp, C.c_int, C.c_int,
                                      C.POINTER(C.c_void_p)]),
    "mcr_fileset_size": (C.c_int, [C.c_void_p]),
    "mcr_fileset_params": (C.c_int64, [C.c_void_p, C.c_int]),
    "mcr_fileset_chains": (C.c_int64, [C.c_void_p, C.c_int]),
    "mcr_fileset_draws": (C.c_int64, [C.c_void_p, C.c_int]),
    "mcr_fileset_param_name": (C.c_char_p, [C.c_void_p, C.c_int, C.c_int64]),
    "mcr_fileset_field": (_dp, [C.c_void_p, C.c_int, C.c_int]),
    "mcr_fileset_phases": (C.c_int, [C.c_void_p, _dp, C.c_int]),
    "mcr_fileset_jobs": (C.c_int, [C.c_void_p]),
    "mcr_fileset_export": (C.c_int64, [C.c_void_p, _dp, C.c_int64]),
    "mcr_fileset_names": (C.c_int64, [C.c_void_p, C.c_char_p, C.c_int64]),
    "mcr_fileset_free": (None, [C.c_void_p]),
}

_lib = None
_lib_lock = threading.Lock()


def lib_path() -> Path:
    return Path(os.environ.get("MCMC_REF_HIP_LIB", str(DEFAULT_LIB)))


def load_library():
    """dlopen the C ABI and bind every declared symbol (works without a GPU)."""
    global _lib
    with _lib_lock:
        if _lib is None:
            path = lib_path()
            if not path.exists():
                raise HipUnavailableError(
                    f"{path} not found: build it with `python mcmc-db_amd/build.py` "
                    "(there is no CPU fallback)")
            # Multi-process RCCL (N > 1 ranks, one per GPU) exchanges device buffers between processes over IPC handles.  The
            # host driver of this platform supports only the dmabuf form; with the HSA runtime's legacy IPC mode (its
            # default) `hipIpcGetMemHandle` fails with "invalid argument" inside ncclCommInitRank.  The runtime reads the
            # variable when it initialises, i.e. at the first HIP call, so it is set here, before the library is mapped --
            # it must precede the first HIP call of the process, whoever makes it; a value the launcher exported wins.
            # Only for multi-rank launches: a single process has no IPC to do, and the variable is process-global (it would
            # change the runtime's behaviour for any other HIP library the host application uses: ADVICE r3).
            if int(os.environ.get("WORLD_SIZE", "1") or "1") > 1:
                os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            try:
                L = C.CDLL(str(path))
            except OSError as exc:
                raise HipUnavailableError(f"cannot load {path}: {exc}") from exc
            for name, (res, args) in SYMBOLS.items():
                fn = getattr(L, name)   # AttributeError if the ABI is incomplete
                fn.restype = res
                fn.argtypes = args
            _lib = L
    return _lib


def _as_dp(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _as_ip(a):
    return a.ctypes.data_as(_ip) if a is not None else None


SUMMARY_F64 = ("mean", "std", "median", "rhat", "rhat_bulk", "rhat_tail", "ess_bulk", "ess_tail")


class SummaryBuffers:
    """Host arrays an mcr_summary points into (kept alive by this object)."""

    def __init__(self, P: int, nq: int, diagnostics: bool = True):
        self.P, self.nq, self.diagnostics = P, nq, diagnostics
        n = max(P, 1)
        self.arrays = {k: np.full(n, np.nan) for k in SUMMARY_F64}
        self.arrays["q"] = np.full(max(P * nq, 1), np.nan)
        self.arrays["lag_bulk"] = np.zeros(n, dtype=np.int64)
        self.arrays["lag_tail"] = np.zeros(n, dtype=np.int64)
        self.arrays["q_lo"] = np.zeros(max(nq, 1), dtype=np.int64)
        kw = {k: _as_dp(self.arrays[k]) for k in SUMMARY_F64 + ("q",)}
        kw.update({k: _as_ip(self.arrays[k]) for k in ("lag_bulk", "lag_tail", "q_lo")})
        if not diagnostics:     # NULL members: the library skips the diagnostics kernels
            for k in ("rhat", "rhat_bulk", "rhat_tail", "ess_bulk", "ess_tail", "lag_bulk", "lag_tail"):
                kw[k] = None
        self.struct = Summary(**kw)

    def result(self) -> dict:
        P, nq = self.P, self.nq
        out = {k: self.arrays[k][:P].copy() for k in SUMMARY_F64 + ("lag_bulk", "lag_tail")}
        out["q"] = self.arrays["q"][:P * nq].reshape(P, nq).copy()
        out["q_lo"] = self.arrays["q_lo"][:nq].copy()
        return out


def tensor_args(draws: np.ndarray, layout: str):
    """(dtype code, C, N, P, stride_c, stride_n, stride_p) for a 3-D array whose axes are `layout`."""
    if draws.ndim != 3 or sorted(layout) != ["c", "n", "p"]:
        raise ValueError("draws must be 3-D and layout a permutation of 'cnp'")
    if draws.dtype == np.float64:
        code = MCR_F64
    elif draws.dtype == np.float32:
        code = MCR_F32
    else:
        raise TypeError(f"unsupported dtype {draws.dtype}; use float64 or float32")
    ax = {a: i for i, a in enumerate(layout)}
    dims = [draws.shape[ax[a]] for a in "cnp"]
    strides = [draws.strides[ax[a]] // draws.itemsize for a in "cnp"]
    if any(s < 0 for s in strides):
        raise ValueError("negative strides are not supported")
    return (code, *dims, *strides)


class DeviceBuffer:
    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx, self.nbytes = ctx, nbytes
        self.ptr = C.c_void_p()
        ctx._check(ctx.lib.mcr_dev_alloc(ctx.handle, nbytes, C.byref(self.ptr)))

    def upload(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.ctx._check(self.ctx.lib.mcr_memcpy_h2d(self.ctx.handle, self.ptr, arr.ctypes.data_as(C.c_void_p),
                                                    arr.nbytes))
        return self

    def download(self, dtype, count: int) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        self.ctx._check(self.ctx.lib.mcr_memcpy_d2h(self.ctx.handle, out.ctypes.data_as(C.c_void_p), self.ptr,
                                                    out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.ctx.lib.mcr_dev_free(self.ctx.handle, self.ptr)
            self.ptr = C.c_void_p()


class DeviceTensor:
    """A draw tensor resident in HBM: device buffer + the (dtype, dims, strides) it was uploaded with."""

    def __init__(self, ctx: "Context", buf: DeviceBuffer, targs):
        self.ctx, self.buf, self.targs = ctx, buf, targs

    @property
    def shape_cnp(self):
        return self.targs[1:4]

    def free(self):
        self.buf.free()


class Context:
    """One GPU + one HIP stream (mcr_ctx).  Not thread-safe: one Context per thread."""

    def __init__(self, device: int | None = None):
        self.lib = load_library()
        if device is None:
            device = int(os.environ.get("MCMC_REF_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        h = C.c_void_p()
        rc = self.lib.mcr_init(int(device), C.byref(h))
        if rc != MCR_OK:
            msg = (self.lib.mcr_last_error(None) or b"").decode()
            if rc == MCR_ENODEVICE:
                raise HipUnavailableError(msg)
            raise McrError(rc, msg)
        self.handle = h
        self.device = int(device)
        self._pending: list[SummaryBuffers] = []

    def close(self):
        if getattr(self, "handle", None):
            self.lib.mcr_free(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass

    # -- errors ------------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != MCR_OK:
            raise McrError(rc, (self.lib.mcr_last_error(self.handle) or b"").decode())

    # -- device memory --------------------------------------------------------------------------
    def upload(self, draws: np.ndarray, layout: str = "pcn") -> DeviceTensor:
        targs = tensor_args(draws, layout)
        if not draws.flags.c_contiguous:
            raise ValueError("upload() needs a C-contiguous array (permute the layout string instead)")
        buf = DeviceBuffer(self, max(draws.nbytes, 8)).upload(draws)
        return DeviceTensor(self, buf, targs)

    def alloc_tensor(self, C_: int, N: int, P: int, dtype=np.float64) -> DeviceTensor:
        """Uninitialised [P][C][N] device tensor (fill it with fill_synthetic)."""
        es = np.dtype(dtype).itemsize
        buf = DeviceBuffer(self, max(C_ * N * P * es, 8))
        code = MCR_F64 if np.dtype(dtype) == np.float64 else MCR_F32
        return DeviceTensor(self, buf, (code, C_, N, P, N, 1, C_ * N))

    def fill_synthetic(self, t: DeviceTensor, seed: int = 4711, p0: int = 0):
        """Synthetic stress draws; p0 > 0: `t` is the parameter block [p0, p0 + P) of a larger tensor."""
        code, C_, N, P = t.targs[:4]
        self._check(self.lib.mcr_fill_synthetic_at(self.handle, t.buf.ptr, code, C_, N, P, int(p0), seed))

    def hbm_probe(self, nbytes: int = 4 << 30, iters: int = 5) -> dict:
        """Measured HBM rates of this device (GB/s): read-only streaming kernel and device-to-device copy."""
        rd, cp = C.c_double(0.0), C.c_double(0.0)
        self._check(self.lib.mcr_hbm_probe(self.handle, int(nbytes), int(iters), C.byref(rd), C.byref(cp)))
        return {"read_GBps": rd.value, "copy_GBps": cp.value}

    def sync(self):
        self._check(self.lib.mcr_sync(self.handle))

    # -- hot path -----------------------------------------------------------------------------
    @staticmethod
    def _quantiles(quantiles):
        qs = np.ascontiguousarray(list(quantiles), dtype=np.float64)
        if qs.size > MCR_MAX_QUANTILES:
            raise ValueError(f"at most {MCR_MAX_QUANTILES} quantiles")
        return qs

    def summarize(self, draws, layout: str = "pcn", min_chains: int = 4,
                  quantiles=(0.05, 0.5, 0.95), diagnostics: bool = True) -> dict:
        """All per-parameter statistics of a host array or a DeviceTensor (synchronous).

        diagnostics=False: Backend.stats only (mean/std/quantiles/median); the rank, fold and
        autocovariance kernels are not launched and the diagnostics come back as NaN.
        """
        qs = self._quantiles(quantiles)
        if isinstance(draws, DeviceTensor):
            targs, ptr, fn = draws.targs, draws.buf.ptr, self.lib.mcr_summarize_dev
        else:
            targs, fn = tensor_args(draws, layout), self.lib.mcr_summarize
            ptr = draws.ctypes.data_as(C.c_void_p)
        bufs = SummaryBuffers(targs[3], qs.size, diagnostics)
        self._check(fn(self.handle, ptr, *targs, int(min_chains), _as_dp(qs), qs.size, C.byref(bufs.struct)))
        return bufs.result()

    def enqueue(self, t: DeviceTensor, min_chains: int = 4, quantiles=(0.05, 0.5, 0.95),
                diagnostics: bool = True, bufs: "SummaryBuffers | None" = None) -> SummaryBuffers:
        """Asynchronous summarize.  `bufs`: result buffers of an EARLIER, already delivered call of the same shape to
        write into again (a caller that streams many same-shape models keeps a ring of them instead of allocating
        twelve arrays per call)."""
        qs = self._quantiles(quantiles)
        if bufs is None or bufs.P != t.targs[3] or bufs.nq != qs.size or bufs.diagnostics != diagnostics:
            bufs = SummaryBuffers(t.targs[3], qs.size, diagnostics)
        self._check(self.lib.mcr_summarize_enqueue(self.handle, t.buf.ptr, *t.targs, int(min_chains), _as_dp(qs),
                                                   qs.size, C.byref(bufs.struct)))
        self._pending.append(bufs)
        return bufs

    def summarize_models(self, tensors, min_chains: int = 4, quantiles=(0.05, 0.5, 0.95)) -> list[dict]:
        """One C call for a list of DeviceTensors (independent models), pipelined through the lanes."""
        qs = self._quantiles(quantiles)
        n = len(tensors)
        descs = (ModelDesc * max(n, 1))()
        outs = (Summary * max(n, 1))()
        bufs = []
        for i, t in enumerate(tensors):
            code, C_, N, P, sc, sn, sp = t.targs
            descs[i] = ModelDesc(t.buf.ptr, code, int(min_chains), C_, N, P, sc, sn, sp)
            b = SummaryBuffers(P, qs.size)
            outs[i] = b.struct
            bufs.append(b)
        self._check(self.lib.mcr_summarize_models(self.handle, descs, n, _as_dp(qs), qs.size, outs))
        return [b.result() for b in bufs]

    def wait(self):
        try:
            self._check(self.lib.mcr_summarize_wait(self.handle))
        finally:
            self._pending.clear()

    def wait_one(self) -> SummaryBuffers | None:
        """Wait for the oldest outstanding enqueue only; returns its buffers (None if nothing is pending)."""
        if not self._pending:
            return None
        bufs = self._pending.pop(0)
        self._check(self.lib.mcr_summarize_wait_one(self.handle))
        return bufs

    @property
    def inflight(self) -> int:
        return len(self._pending)

    def diagnose_chains(self, chains, min_chains: int = 4, debug: bool = False) -> dict:
        """split_rhat / ess_bulk / ess_tail of ONE parameter given as (possibly ragged) chains."""
        off = np.zeros(len(chains) + 1, dtype=np.int64)
        for i, c in enumerate(chains):
            off[i + 1] = off[i] + len(c)
        M = int(off[-1])
        pooled = np.empty(max(M, 1), dtype=np.float64)
        for i, c in enumerate(chains):
            pooled[off[i]:off[i + 1]] = np.asarray(c, dtype=np.float64)
        bufs = SummaryBuffers(1, 0)
        dbg = [np.full(max(M, 1), np.nan) for _ in range(4)] if debug else [None] * 4
        self._check(self.lib.mcr_diagnose_chains(self.handle, _as_dp(pooled), _as_ip(off), len(chains),
                                                 int(min_chains), C.byref(bufs.struct), *[_as_dp(d) for d in dbg]))
        r = bufs.result()
        out = {k: (float(r[k][0]) if k in SUMMARY_F64 else int(r[k][0]))
               for k in ("rhat", "rhat_bulk", "rhat_tail", "ess_bulk", "ess_tail", "median", "lag_bulk",
                         "lag_tail")}
        if debug:
            for name, d in zip(("z_bulk", "z_tail", "rank_bulk", "rank_tail"), dbg):
                out[name] = [d[off[i]:off[i + 1]].copy() for i in range(len(chains))]
        return out

    def basic_stats(self, values) -> dict:
        v = np.ascontiguousarray(values)
        if v.dtype not in (np.float64, np.float32):
            v = v.astype(np.float64)
        mean, std = C.c_double(math.nan), C.c_double(math.nan)
        code = MCR_F64 if v.dtype == np.float64 else MCR_F32
        self._check(self.lib.mcr_basic_stats(self.handle, v.ctypes.data_as(C.c_void_p), code, v.size,
                                             C.byref(mean), C.byref(std)))
        return {"mean": mean.value, "std": std.value}

    def moments(self, t: DeviceTensor) -> tuple[np.ndarray, np.ndarray]:
        P = t.targs[3]
        mean, std = np.full(max(P, 1), np.nan), np.full(max(P, 1), np.nan)
        self._check(self.lib.mcr_moments_dev(self.handle, t.buf.ptr, *t.targs, _as_dp(mean), _as_dp(std)))
        return mean[:P], std[:P]

    def compare(self, ref, actual, tol: float):
        r = np.ascontiguousarray(ref, dtype=np.float64)
        a = np.ascontiguousarray(actual, dtype=np.float64)
        if r.shape != a.shape:
            raise ValueError("ref and actual must have the same shape")
        rel = np.empty(max(r.size, 1))
        ok = np.zeros(max(r.size, 1), dtype=np.uint8)
        self._check(self.lib.mcr_compare(self.handle, _as_dp(r), _as_dp(a), r.size, float(tol), _as_dp(rel),
                                         ok.ctypes.data_as(C.POINTER(C.c_uint8))))
        return rel[:r.size], ok[:r.size].astype(bool)

    # -- extensions (not in the reference) -------------------------------------------------------------
    def two_sample(self, ref, actual) -> tuple[np.ndarray, np.ndarray]:
        """(KS statistic, Wasserstein-1) per parameter; ref [P][Mr], actual [P][Ma] (2-D, finite)."""
        r = np.ascontiguousarray(ref, dtype=np.float64)
        a = np.ascontiguousarray(actual, dtype=np.float64)
        if r.ndim != 2 or a.ndim != 2 or r.shape[0] != a.shape[0]:
            raise ValueError("ref and actual must be 2-D with the same number of parameters")
        if not (np.isfinite(r).all() and np.isfinite(a).all()):
            raise ValueError("draws contain non-finite values")
        P = r.shape[0]
        ks, w1 = np.full(max(P, 1), np.nan), np.full(max(P, 1), np.nan)
        self._check(self.lib.mcr_two_sample(self.handle, _as_dp(r), r.shape[1], _as_dp(a), a.shape[1], P,
                                            _as_dp(ks), _as_dp(w1)))
        return ks[:P], w1[:P]

    def covariance(self, draws) -> np.ndarray:
        """Population covariance (ddof=0) of draws [P][M] -> [P][P] (fp64 MFMA)."""
        x = np.ascontiguousarray(draws, dtype=np.float64)
        if x.ndim != 2:
            raise ValueError("draws must be 2-D [P][M]")
        P = x.shape[0]
        cov = np.full((max(P, 1), max(P, 1)), np.nan)
        self._check(self.lib.mcr_covariance(self.handle, _as_dp(x), x.shape[1], P, _as_dp(cov)))
        return cov[:P, :P] if P else np.empty((0, 0))

    # -- measurement -----------------------------------------------------------------------------
    def params_per_chunk(self, t: "DeviceTensor", diagnostics: bool = True) -> int:
        """Parameters per workspace chunk of a call on this tensor (mcr_plan_chunks): chunk k is [k * n, (k + 1) * n)."""
        _, C_, N, P, sc, sn, sp = t.targs
        v = C.c_int64(0)
        self._check(self.lib.mcr_plan_chunks(self.handle, C_, N, P, sc, sn, sp, int(diagnostics), C.byref(v)))
        return int(v.value)

    def rho_guard_count(self) -> int:
        """Band lags of the tier-3 ESS scan re-derived with the reference's own sums so far (mcr_rho_guard_count)."""
        v = C.c_int64(0)
        self._check(self.lib.mcr_rho_guard_count(self.handle, C.byref(v)))
        return int(v.value)

    def profile(self, on: bool):
        self._check(self.lib.mcr_profile_enable(self.handle, 1 if on else 0))

    def profile_reset(self):
        self._check(self.lib.mcr_profile_reset(self.handle))

    def profile_get(self) -> dict:
        arr = (KernelTime * 32)()
        n = C.c_int(0)
        self._check(self.lib.mcr_profile_get(self.handle, arr, 32, C.byref(n)))
        return {arr[i].name.decode(): {"launches": int(arr[i].launches), "total_ms": float(arr[i].total_ms)}
                for i in range(min(n.value, 32))}


_default = threading.local()


def default_context() -> Context:
    """The calling THREAD's Context on MCMC_REF_HIP_DEVICE / LOCAL_RANK / device 0 (created on first use).

    A Context is one GPU + its streams and is not thread-safe, so the module-level convenience functions
    (`reference.stats`, `compare.compute_basic_stats`, ...) never share one between threads: each thread gets its own."""
    ctx = getattr(_default, "ctx", None)
    if ctx is None or ctx.handle is None:
        ctx = _default.ctx = Context()
    return ctx
