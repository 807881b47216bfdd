"""Draws wrapper (reference src/mcmc_ref/draws.py): Arrow object + conversion helpers."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any


@dataclass
class Draws:
    data: Any
    params: list[str]
    chains: list[int] | None = None
    meta: dict[str, Any] | None = None

    def _table(self) -> Any:
        return self.data.read_all() if hasattr(self.data, "read_all") else self.data

    def to_arrow(self) -> Any:
        return self.data

    def to_numpy(self) -> Any:
        """(C*N, P) float array, parameters fastest -- the layout mcr_summarize takes as
        stride_p = 1 (layout string "cnp" after a reshape to (C, N, P))."""
        import numpy as np
        table = self._table()
        return np.stack([table.column(p).to_numpy(zero_copy_only=False) for p in self.params], axis=-1)

    def to_list(self) -> list[dict[str, Any]]:
        table = self._table()
        return table.to_pylist() if hasattr(table, "to_pylist") else list(table)


def coerce_return(draws: Draws, return_: str) -> Any:
    if return_ == "draws":
        return draws
    if return_ == "arrow":
        return draws.to_arrow()
    if return_ == "numpy":
        return draws.to_numpy()
    if return_ == "list":
        return draws.to_list()
    raise ValueError(f"Unknown return type: {return_}")
