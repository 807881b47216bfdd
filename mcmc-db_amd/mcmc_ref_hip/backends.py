"""Backend registry with the MI355X "hip" backend.

Mirrors src/mcmc_ref/backends.py:14-55 of the reference: the `Backend` protocol, the frozen
`BackendSpec`, the `BACKENDS` dict and `get_backend(name)` with the same error text.  Entries: "hip"
(this library) plus the reference's own "arrow" and "numpy" (lazy pass-throughs to pyarrow.compute / numpy,
backends.py:27-48), so every `backend=` value the reference accepts keeps working.
`register_into(registry)` adds "hip" to the reference's own `BACKENDS` so that
`reference.stats(model, backend="hip")` works unchanged.
"""
from __future__ import annotations

from collections.abc import Callable, Iterable
from dataclasses import dataclass
from typing import Any, Protocol

import numpy as np


class Backend(Protocol):
    name: str

    def stats(
        self,
        table: Any,
        params: Iterable[str],
        quantiles: Iterable[float] = (0.05, 0.5, 0.95),
        quantile_mode: str = "exact",
    ) -> dict[str, dict[str, float]]:
        """Compute per-parameter stats from an Arrow Table or batches."""


@dataclass(frozen=True)
class BackendSpec:
    name: str
    loader: Callable[[], Backend]


def columns_as_arrays(table: Any, params: list[str]) -> list[np.ndarray]:
    """The requested columns as float64 arrays (Arrow Table, reader, or mapping of arrays).  Arrow nulls are DROPPED, as
    `pc.mean` / `pc.stddev` / `pc.quantile(skip_nulls=True)` do in ArrowBackend.stats (src/mcmc_ref/backends_arrow.py:38-42),
    so the columns may come back with different lengths."""
    if hasattr(table, "read_all"):
        table = table.read_all()
    cols = []
    for p in params:
        if hasattr(table, "column"):
            col = table.column(p)              # KeyError from pyarrow for unknown columns, as in the reference
            if getattr(col, "null_count", 0):
                col = col.drop_null()
            arr = col.to_numpy(zero_copy_only=False) if hasattr(col, "to_numpy") else np.asarray(col)
        else:
            arr = np.asarray(table[p])
        cols.append(np.asarray(arr, dtype=np.float64).reshape(-1))
    return cols


def columns_to_matrix(table: Any, params: list[str]) -> np.ndarray:
    """[P][M] float64 matrix of the requested columns; all of them must have the same length once nulls are dropped."""
    cols = columns_as_arrays(table, params)
    if not cols:
        return np.empty((0, 0), dtype=np.float64)
    m = len(cols[0])
    if any(len(c) != m for c in cols):
        raise ValueError("all parameter columns must have the same length")
    out = np.empty((len(cols), m), dtype=np.float64)
    for i, c in enumerate(cols):
        out[i] = c
    return out


class HipBackend:
    """`Backend.stats` on the GPU: sort + order statistics + streaming moments kernels.

    Same contract as ArrowBackend.stats / NumpyBackend.stats (src/mcmc_ref/backends_arrow.py:22-52,
    backends_numpy.py:17-49): pooled mean, population std, linear-interpolated quantiles under the
    keys f"q{int(q*100)}".  `quantile_mode` is accepted and ignored, as in the reference.
    Arrow nulls are skipped like ArrowBackend does (backends_arrow.py:38-42: columns that keep different numbers of
    draws go through the kernels in groups of equal length); a column with no draw left raises ValueError, and so do
    NaN / infinite draws (the reference's two backends disagree on them; the kernels reject them).
    """

    name = "hip"

    def __init__(self, context=None) -> None:
        from . import _ffi
        try:
            self._ctx = context or _ffi.default_context()
        except _ffi.HipUnavailableError as exc:   # same shape as the reference's import guards
            raise ImportError(f"libmcmcref_hip + an MI355X are required for the hip backend: {exc}") from exc

    def stats(
        self,
        table,
        params: Iterable[str],
        quantiles: Iterable[float] = (0.05, 0.5, 0.95),
        quantile_mode: str = "exact",
    ) -> dict[str, dict[str, float]]:
        from . import _ffi
        params = list(params)
        qs = list(quantiles)
        cols = columns_as_arrays(table, params)
        if not params:
            return {}
        if any(len(c) == 0 for c in cols):
            raise ValueError("cannot compute stats of empty columns")
        by_len: dict[int, list[int]] = {}
        for i, c in enumerate(cols):
            by_len.setdefault(len(c), []).append(i)
        entries: dict[int, dict[str, float]] = {}
        for M, members in by_len.items():          # one group unless nulls were dropped unevenly
            x = np.empty((len(members), 1, M), dtype=np.float64)
            for k, i in enumerate(members):
                x[k, 0] = cols[i]
            try:
                r = self._ctx.summarize(x, "pcn", min_chains=1, quantiles=qs, diagnostics=False)
            except _ffi.McrError as exc:
                raise ValueError(exc.message) from exc
            for k, i in enumerate(members):
                entry = {"mean": float(r["mean"][k]), "std": float(r["std"][k])}
                for q, v in zip(qs, r["q"][k], strict=False):
                    entry[f"q{int(q * 100)}"] = float(v)
                entries[i] = entry
        return {param: entries[i] for i, param in enumerate(params)}


def _load_hip() -> Backend:
    return HipBackend()


def _load_arrow() -> Backend:
    from .backends_arrow import ArrowBackend      # lazy: pyarrow is imported only when asked for
    return ArrowBackend()


def _load_numpy() -> Backend:
    from .backends_numpy import NumpyBackend
    return NumpyBackend()


BACKENDS: dict[str, BackendSpec] = {
    "hip": BackendSpec(name="hip", loader=_load_hip),
    "arrow": BackendSpec(name="arrow", loader=_load_arrow),
    "numpy": BackendSpec(name="numpy", loader=_load_numpy),
}


def get_backend(name: str) -> Backend:
    spec = BACKENDS.get(name)
    if spec is None:
        raise ValueError(f"Unknown backend: {name}")
    return spec.loader()


def register_into(registry: dict, spec_cls=BackendSpec) -> None:
    """Add the "hip" entry to another registry (e.g. the reference's mcmc_ref.backends.BACKENDS)."""
    registry["hip"] = spec_cls(name="hip", loader=_load_hip)
