"""ctypes loader for oracle/libmcr_oracle.so (TEST INFRASTRUCTURE ONLY).

The C file restates the reference's algorithms (see mcr_oracle.c for the
file:line citations); this module only marshals numpy arrays in and out.
"""
from __future__ import annotations

import ctypes as C
import math
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "libmcr_oracle.so"


def build(force: bool = False) -> Path:
    src = _HERE / "mcr_oracle.c"
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "-B", "libmcr_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Diag(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("rhat", "rhat_bulk", "rhat_tail", "ess_bulk", "ess_tail", "median")] + \
               [("lag_bulk", C.c_int64), ("lag_tail", C.c_int64)]


class _Summary(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_double)) for n in
                ("mean", "std", "q", "rhat", "rhat_bulk", "rhat_tail", "ess_bulk", "ess_tail",
                 "median")] + [("lag_bulk", C.POINTER(C.c_int64)), ("lag_tail", C.POINTER(C.c_int64))]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(_LIB_PATH))
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int64)
        L.orc_inv_cdf.restype = C.c_double
        L.orc_inv_cdf.argtypes = [C.c_double]
        L.orc_rank_normalize.argtypes = [dp, C.c_int64, dp, dp]
        L.orc_fold.argtypes = [dp, C.c_int64, dp, dp]
        L.orc_split_rhat_of_z.restype = C.c_double
        L.orc_split_rhat_of_z.argtypes = [dp, ip, C.c_int]
        L.orc_ess_of_z.restype = C.c_double
        L.orc_ess_of_z.argtypes = [dp, ip, C.c_int, ip]
        L.orc_diag.argtypes = [dp, ip, C.c_int, C.c_int, C.POINTER(_Diag)]
        L.orc_basic_stats.argtypes = [dp, C.c_int64, dp, dp]
        L.orc_basic_stats.restype = None
        L.orc_stats.argtypes = [dp, C.c_int64, dp, C.c_int, dp, dp, dp, ip]
        L.orc_compare.argtypes = [dp, dp, C.c_int64, C.c_double, dp, C.POINTER(C.c_uint8)]
        L.orc_compare.restype = None
        L.orc_summarize.argtypes = [C.c_void_p, C.c_int] + [C.c_int64] * 6 + \
                                   [C.c_int, dp, C.c_int, C.POINTER(_Summary)]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _pool(chains):
    """list-of-lists (possibly ragged) -> pooled f64 array + int64 offsets."""
    off = np.zeros(len(chains) + 1, dtype=np.int64)
    for i, c in enumerate(chains):
        off[i + 1] = off[i] + len(c)
    x = np.empty(max(int(off[-1]), 1), dtype=np.float64)
    for i, c in enumerate(chains):
        x[off[i]:off[i + 1]] = np.asarray(c, dtype=np.float64)
    return x, off


def inv_cdf(p: float) -> float:
    return lib().orc_inv_cdf(float(p))


def rank_normalize(chains):
    x, off = _pool(chains)
    m = int(off[-1])
    z = np.empty(max(m, 1)); r = np.empty(max(m, 1))
    lib().orc_rank_normalize(_dp(x), m, _dp(z), _dp(r))
    return ([z[off[i]:off[i + 1]].copy() for i in range(len(chains))],
            [r[off[i]:off[i + 1]].copy() for i in range(len(chains))])


def fold(chains):
    x, off = _pool(chains)
    m = int(off[-1])
    f = np.empty(max(m, 1)); med = C.c_double(math.nan)
    lib().orc_fold(_dp(x), m, _dp(f), C.byref(med))
    return [f[off[i]:off[i + 1]].copy() for i in range(len(chains))], med.value


_ERR = {
    -3: lambda k, m, what: ValueError(f"min_chains must be >= 1; got {k}"),
    -2: lambda k, m, what: ValueError(f"{what} diagnostics require at least {k} chains; got {m} chain(s)"),
}


def diag(chains, min_chains: int = 4, what: str = "R-hat") -> dict:
    """All three diagnostics + integer intermediates for one parameter."""
    x, off = _pool(chains)
    d = _Diag()
    rc = lib().orc_diag(_dp(x), _ip(off), len(chains), int(min_chains), C.byref(d))
    if rc in _ERR:
        raise _ERR[rc](min_chains, len(chains), what)
    if rc:
        raise MemoryError("oracle allocation failed")
    return {n: getattr(d, n) for n, _ in _Diag._fields_}


def split_rhat(chains, *, min_chains: int = 4) -> float:
    return diag(chains, min_chains, "R-hat")["rhat"]


def ess_bulk(chains, *, min_chains: int = 4) -> float:
    return diag(chains, min_chains, "ESS")["ess_bulk"]


def ess_tail(chains, *, min_chains: int = 4) -> float:
    return diag(chains, min_chains, "ESS")["ess_tail"]


def basic_stats(values) -> dict:
    v = np.ascontiguousarray(values, dtype=np.float64)
    mean, std = C.c_double(), C.c_double()
    lib().orc_basic_stats(_dp(v if v.size else np.zeros(1)), v.size, C.byref(mean), C.byref(std))
    return {"mean": mean.value, "std": std.value}


def stats(values, quantiles=(0.05, 0.5, 0.95)) -> dict:
    v = np.ascontiguousarray(values, dtype=np.float64)
    qs = np.ascontiguousarray(quantiles, dtype=np.float64)
    mean, std = C.c_double(), C.c_double()
    qo = np.empty(max(qs.size, 1)); lo = np.zeros(max(qs.size, 1), dtype=np.int64)
    rc = lib().orc_stats(_dp(v), v.size, _dp(qs), qs.size, C.byref(mean), C.byref(std), _dp(qo), _ip(lo))
    if rc:
        raise ValueError("orc_stats failed")
    out = {"mean": mean.value, "std": std.value}
    for q, val in zip(qs, qo):
        out[f"q{int(q * 100)}"] = float(val)
    out["_q_lo"] = [int(i) for i in lo[:qs.size]]
    return out


def compare(ref, act, tol):
    r = np.ascontiguousarray(ref, dtype=np.float64); a = np.ascontiguousarray(act, dtype=np.float64)
    rel = np.empty(max(r.size, 1)); ok = np.zeros(max(r.size, 1), dtype=np.uint8)
    lib().orc_compare(_dp(r), _dp(a), r.size, float(tol), _dp(rel), ok.ctypes.data_as(C.POINTER(C.c_uint8)))
    return rel[:r.size], ok[:r.size].astype(bool)


def summarize(draws: np.ndarray, layout: str = "pcn", min_chains: int = 4,
              quantiles=(0.05, 0.5, 0.95)) -> dict:
    """draws: 3-D array; layout names the axis order ('pcn' = [P][C][N], 'cnp' = [C][N][P])."""
    assert draws.ndim == 3 and draws.dtype in (np.float64, np.float32)
    ax = {a: i for i, a in enumerate(layout)}
    Cn, N, P = (draws.shape[ax[a]] for a in "cnp")
    es = draws.itemsize
    sc, sn, sp = (draws.strides[ax[a]] // es for a in "cnp")
    qs = np.ascontiguousarray(quantiles, dtype=np.float64)
    nq = qs.size
    arrs = {n: np.full(max(P, 1), np.nan) for n in
            ("mean", "std", "rhat", "rhat_bulk", "rhat_tail", "ess_bulk", "ess_tail", "median")}
    arrs["q"] = np.full(max(P * nq, 1), np.nan)
    lb = np.zeros(max(P, 1), dtype=np.int64); lt = np.zeros(max(P, 1), dtype=np.int64)
    s = _Summary(**{n: _dp(a) for n, a in arrs.items()}, lag_bulk=_ip(lb), lag_tail=_ip(lt))
    rc = lib().orc_summarize(draws.ctypes.data_as(C.c_void_p), 0 if draws.dtype == np.float64 else 1,
                             Cn, N, P, sc, sn, sp, int(min_chains), _dp(qs), nq, C.byref(s))
    if rc in _ERR:
        raise _ERR[rc](min_chains, Cn, "R-hat")
    if rc:
        raise MemoryError("oracle allocation failed")
    out = {n: a[:P].copy() for n, a in arrs.items() if n != "q"}
    out["q"] = arrs["q"][:P * nq].reshape(P, nq).copy()
    out["lag_bulk"] = lb[:P].copy(); out["lag_tail"] = lt[:P].copy()
    return out


def summarize_mt(draws: np.ndarray, layout: str = "pcn", threads: int | None = None, **kw) -> dict:
    """`summarize` with the parameters split over host threads (ctypes releases the GIL; parameters are independent)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    ax = layout.index("p")
    P = draws.shape[ax]
    T = max(1, min(threads or (os.cpu_count() or 1), P))
    if T == 1:
        return summarize(draws, layout, **kw)
    cuts = [P * i // T for i in range(T + 1)]
    parts = [np.ascontiguousarray(np.take(draws, range(cuts[i], cuts[i + 1]), axis=ax)) for i in range(T)]
    with ThreadPoolExecutor(T) as pool:
        outs = list(pool.map(lambda x: summarize(x, layout, **kw), parts))
    return {k: np.concatenate([o[k] for o in outs], axis=0) for k in outs[0]}
