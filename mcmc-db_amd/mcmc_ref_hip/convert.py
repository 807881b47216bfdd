"""The diagnostics call site of the conversion pipeline, batched for the GPU.

Mirrors the hot-path part of src/mcmc_ref/convert.py: `_chains_from_table` (:150-161, here a
whole-table integer gather `table_to_tensor`), `_count_chains_draws` (:123-131),
`_compute_diagnostics` (:134-147, one kernel pipeline per model instead of a Python loop per
parameter), `_checks` (:164-172) and `_enforce_checks` (:175-178).
"""
from __future__ import annotations

from collections.abc import Iterable
from typing import Any

import numpy as np

from . import _ffi
from .backends import columns_to_matrix


def _col(table: Any, name: str) -> np.ndarray:
    if hasattr(table, "column"):
        col = table.column(name)
        return np.asarray(col.to_numpy(zero_copy_only=False) if hasattr(col, "to_numpy") else col)
    return np.asarray(table[name])


def chain_layout(table: Any):
    """Integer bookkeeping of the long table: (chain ids sorted, row order, per-chain lengths).

    order[k] = row index of the k-th draw in (chain id asc, draw idx asc) order, which is how
    `_chains_from_table` orders values (robust to unordered rows).
    """
    if hasattr(table, "read_all"):
        table = table.read_all()
    chain = _col(table, "chain").astype(np.int64)
    draw = _col(table, "draw").astype(np.int64)
    ids, counts = np.unique(chain, return_counts=True)
    already = chain.size == 0 or (np.all(np.diff(chain) >= 0) and
                                  all(np.all(np.diff(draw[s:s + n]) >= 0)
                                      for s, n in zip(np.concatenate([[0], np.cumsum(counts)[:-1]]), counts)))
    order = None if already else np.lexsort((draw, chain))     # stable: ties keep row order, like sorted()
    return ids, order, counts


def table_to_tensor(table: Any, params: Iterable[str]):
    """Returns (x, counts): x is [P][M] float64 in (chain, draw) order; counts = draws per chain."""
    if hasattr(table, "read_all"):
        table = table.read_all()
    params = list(params)
    _, order, counts = chain_layout(table)
    x = columns_to_matrix(table, params)
    if order is not None and x.size:
        x = np.ascontiguousarray(x[:, order])
    return x, counts


def _count_chains_draws(table: Any) -> tuple[int, int]:
    _, _, counts = chain_layout(table)
    return int(len(counts)), int(counts.min()) if len(counts) else 0


def _compute_diagnostics(table: Any, params: Iterable[str], *, min_chains: int = 4,
                         context=None) -> dict[str, dict[str, float]]:
    params = list(params)
    if min_chains < 1:
        raise ValueError(f"min_chains must be >= 1; got {min_chains}")
    if not params:
        return {}
    x, counts = table_to_tensor(table, params)
    C = len(counts)
    if C < min_chains:
        raise ValueError(f"R-hat diagnostics require at least {min_chains} chains; got {C} chain(s)")
    ctx = context or _ffi.default_context()
    diag: dict[str, dict[str, float]] = {}
    try:
        if C >= 1 and np.all(counts == counts[0]):
            r = ctx.summarize(x.reshape(len(params), C, int(counts[0])), "pcn", min_chains=min_chains,
                              quantiles=())
            for i, p in enumerate(params):
                diag[p] = {"rhat": float(r["rhat"][i]), "ess_bulk": float(r["ess_bulk"][i]),
                           "ess_tail": float(r["ess_tail"][i])}
        else:                                       # ragged chains: one pipeline per parameter
            off = np.concatenate([[0], np.cumsum(counts)])
            for i, p in enumerate(params):
                chains = [x[i, off[c]:off[c + 1]] for c in range(C)]
                d = ctx.diagnose_chains(chains, min_chains=min_chains)
                diag[p] = {"rhat": d["rhat"], "ess_bulk": d["ess_bulk"], "ess_tail": d["ess_tail"]}
    except _ffi.McrError as exc:
        raise ValueError(exc.message) from exc
    return diag


def _checks(n_chains: int, n_draws: int, diag: dict[str, dict[str, float]]) -> dict[str, bool]:
    ess_ok = all(values.get("ess_bulk", 0.0) > 400 for values in diag.values())
    rhat_ok = all(values.get("rhat", 1.0) < 1.01 for values in diag.values())
    return {
        "ndraws_is_10k": n_chains * n_draws == 10_000,
        "nchains_is_gte_4": n_chains >= 4,
        "ess_above_400": ess_ok,
        "rhat_below_1_01": rhat_ok,
    }


def _enforce_checks(checks: dict[str, bool]) -> None:
    failures = [name for name, ok in checks.items() if not ok]
    if failures:
        raise ValueError(f"quality checks failed: {', '.join(failures)}")
