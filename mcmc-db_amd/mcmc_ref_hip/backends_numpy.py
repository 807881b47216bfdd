"""The reference's optional "numpy" stats backend, kept selectable next to "hip" (src/mcmc_ref/backends_numpy.py:17-49).

A pass-through to numpy's own mean / std(ddof=0) / quantile on the columns, selected explicitly by name; it is not a
fallback for "hip".
"""
from __future__ import annotations

from collections.abc import Iterable


class NumpyBackend:
    name = "numpy"

    def __init__(self) -> None:
        try:
            import numpy  # noqa: F401
        except Exception as exc:  # pragma: no cover - import guard
            raise ImportError("numpy is required for the numpy backend") from exc

    def stats(self, table, params: Iterable[str], quantiles: Iterable[float] = (0.05, 0.5, 0.95),
              quantile_mode: str = "exact") -> dict[str, dict[str, float]]:
        import numpy as np
        table = table.read_all() if hasattr(table, "read_all") else table
        qs = list(quantiles)
        keys = [f"q{int(q * 100)}" for q in qs]

        def column(name):
            if not hasattr(table, "column"):
                return np.asarray(table[name])
            col = table.column(name)
            return col.to_numpy(zero_copy_only=False) if hasattr(col, "to_numpy") else np.asarray(col)

        out: dict[str, dict[str, float]] = {}
        for param in params:
            data = column(param)
            entry = {"mean": float(np.mean(data)), "std": float(np.std(data, ddof=0))}
            entry.update({k: float(v) for k, v in zip(keys, np.quantile(data, qs), strict=False)})
            out[param] = entry
        return out
