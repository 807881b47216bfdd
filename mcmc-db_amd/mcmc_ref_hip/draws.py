"""`Draws`: what `reference.draws()` hands back (interface of the reference's src/mcmc_ref/draws.py:9-53),
plus `to_device()`, which puts the draws into HBM in the layout the kernels consume."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Callable


def _materialise(obj: Any) -> Any:
    """A RecordBatchReader is drained into a Table; a Table is returned as it is."""
    reader = getattr(obj, "read_all", None)
    return reader() if callable(reader) else obj


@dataclass
class Draws:
    data: Any                                   # pyarrow Table or RecordBatchReader
    params: list[str]
    chains: list[int] | None = None
    meta: dict[str, Any] | None = field(default=None)

    # -- conversions named as in the reference ------------------------------------------------
    def to_arrow(self) -> Any:
        return self.data

    def to_numpy(self) -> Any:
        """Rows x parameters (`(C*N, P)`, parameter index fastest): `np.stack(columns, axis=-1)` as the reference does
        (draws.py:28-29), columns keeping their own dtypes (numpy promotes mixed ones); reshaped to (C, N, P) this is
        the "cnp" input layout of mcr_summarize."""
        import numpy as np
        table = _materialise(self.data)
        return np.stack([table.column(p).to_numpy(zero_copy_only=False) for p in self.params], axis=-1)

    def to_list(self) -> list[dict[str, Any]]:
        table = _materialise(self.data)
        rows = getattr(table, "to_pylist", None)
        return rows() if callable(rows) else [r for r in table]

    # -- addition ----------------------------------------------------------------------------
    def to_device(self, context=None):
        """[P][C][N] device tensor in (chain, draw) order (equal-length chains), ready for Context.enqueue."""
        from . import _ffi
        from .convert import table_to_tensor
        x, counts = table_to_tensor(_materialise(self.data), self.params)
        if len(counts) == 0 or (counts != counts[0]).any():
            raise ValueError("to_device needs chains of equal length")
        ctx = context or _ffi.default_context()
        return ctx.upload(x.reshape(len(self.params), len(counts), int(counts[0])), "pcn")


_RETURN_FORMS: dict[str, Callable[[Draws], Any]] = {
    "draws": lambda d: d,
    "arrow": Draws.to_arrow,
    "numpy": Draws.to_numpy,
    "list": Draws.to_list,
}


def coerce_return(draws: Draws, return_: str) -> Any:
    try:
        form = _RETURN_FORMS[return_]
    except KeyError:
        raise ValueError(f"Unknown return type: {return_}") from None
    return form(draws)
