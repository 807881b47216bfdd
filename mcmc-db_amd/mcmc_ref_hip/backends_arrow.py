"""The reference's "arrow" stats backend, kept selectable next to "hip".

`reference.stats(model, backend="arrow")` is the reference's default call (src/mcmc_ref/reference.py:33); this entry
keeps it working through `mcmc_ref_hip.reference`.  It is a pass-through to pyarrow.compute -- the third-party
arithmetic the reference itself calls (src/mcmc_ref/backends_arrow.py:36-51) -- selected explicitly by name, never
as a fallback for "hip".
"""
from __future__ import annotations

from collections.abc import Iterable


class ArrowBackend:
    name = "arrow"

    def __init__(self) -> None:
        try:
            import pyarrow.compute  # noqa: F401
        except Exception as exc:  # pragma: no cover - import guard
            raise ImportError("pyarrow is required for the arrow backend") from exc

    def stats(self, table, params: Iterable[str], quantiles: Iterable[float] = (0.05, 0.5, 0.95),
              quantile_mode: str = "exact") -> dict[str, dict[str, float]]:
        import pyarrow.compute as pc
        table = table.read_all() if hasattr(table, "read_all") else table
        qs = list(quantiles)
        keys = [f"q{int(q * 100)}" for q in qs]
        out: dict[str, dict[str, float]] = {}
        for param in params:
            column = table.column(param)
            entry = {"mean": float(pc.mean(column).as_py()), "std": float(pc.stddev(column).as_py())}   # ddof = 0
            qv = pc.quantile(column, q=qs, interpolation="linear", skip_nulls=True)
            values = qv.to_pylist() if hasattr(qv, "to_pylist") else [qv.as_py()]
            entry.update({k: float(v) for k, v in zip(keys, values, strict=False)})
            out[param] = entry
        return out
