"""Rank-normalised split R-hat and ESS on the GPU, behind the reference's function signatures.

Mirrors src/mcmc_ref/diagnostics.py:13-73 (`split_rhat`, `ess_bulk`, `ess_tail`): same arguments,
same guards, same ValueError texts, NaN for fewer than two chains.  The arithmetic (pooled sort,
tie-averaged ranks, AS241 inverse normal, fold, split-chain variances, first-negative-rho
autocovariance sum) runs in libmcmcref_hip; chains may be ragged.
"""
from __future__ import annotations

from collections.abc import Sequence

from . import _ffi


def _validate_min_chains(min_chains: int) -> None:
    if min_chains < 1:
        raise ValueError(f"min_chains must be >= 1; got {min_chains}")


def _guard(chains, min_chains: int, what: str) -> bool:
    """Reference guards (diagnostics.py:24-30); returns True when the answer is NaN."""
    _validate_min_chains(min_chains)
    if len(chains) < min_chains:
        raise ValueError(f"{what} diagnostics require at least {min_chains} chains; got {len(chains)} chain(s)")
    return len(chains) < 2


def diagnose(chains: Sequence[Sequence[float]], *, min_chains: int = 4, context=None) -> dict:
    """All three diagnostics (+ integer truncation lags) from ONE pass of the kernels."""
    if _guard(chains, min_chains, "R-hat"):
        nan = float("nan")
        return {"rhat": nan, "ess_bulk": nan, "ess_tail": nan}
    ctx = context or _ffi.default_context()
    try:
        return ctx.diagnose_chains(chains, min_chains=min_chains)
    except _ffi.McrError as exc:
        raise ValueError(exc.message) from exc


def split_rhat(chains: Sequence[Sequence[float]], *, min_chains: int = 4) -> float:
    """Rank-normalized split R-hat with folded variant (returns max of both)."""
    if _guard(chains, min_chains, "R-hat"):
        return float("nan")
    return diagnose(chains, min_chains=min_chains)["rhat"]


def ess_bulk(chains: Sequence[Sequence[float]], *, min_chains: int = 4) -> float:
    if _guard(chains, min_chains, "ESS"):
        return float("nan")
    return diagnose(chains, min_chains=min_chains)["ess_bulk"]


def ess_tail(chains: Sequence[Sequence[float]], *, min_chains: int = 4) -> float:
    if _guard(chains, min_chains, "ESS"):
        return float("nan")
    return diagnose(chains, min_chains=min_chains)["ess_tail"]
