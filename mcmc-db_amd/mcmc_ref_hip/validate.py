"""validate(): draw-vs-reference validation in one call (ADDITIVE: not a reference symbol).

The reference gates a sampler's draws with `reference.compare` (relative error of mean / std,
src/mcmc_ref/reference.py:107-122).  `validate` keeps that gate unchanged and adds, per parameter,
distribution-level distances computed on the GPU -- the two-sample Kolmogorov-Smirnov statistic and
the Wasserstein-1 distance (scaled by the reference std) -- plus the reference draws' own
diagnostics.  Nothing here exists in the reference: parity for these extras is pinned to scipy only
(tests/test_ext_gpu.py), and they never change `passed` unless thresholds are given explicitly.
"""
from __future__ import annotations

from collections.abc import Mapping, Sequence
from dataclasses import dataclass, field

import numpy as np

from . import _ffi
from .backends import columns_to_matrix
from .compare import CompareResult, compare_stats, compute_stats_from_draws
from .store import DataStore


@dataclass(frozen=True)
class ValidateResult:
    passed: bool
    compare: CompareResult                       # the reference's gate, unchanged
    ks: dict[str, float]                         # two-sample KS statistic per parameter
    wasserstein: dict[str, float]                # W1 per parameter
    wasserstein_scaled: dict[str, float]         # W1 / reference std
    failures: list[str] = field(default_factory=list)


def validate(model: str, actual: Mapping[str, Sequence[float]], tolerance: float = 0.15,
             metrics: Sequence[str] = ("mean", "std"), ks_max: float | None = None,
             w1_scaled_max: float | None = None, store: DataStore | None = None, context=None) -> ValidateResult:
    store = store or DataStore()
    ctx = context or _ffi.default_context()
    params = list(actual.keys())
    table = store.open_draws(model, params=params).read_all()
    ref = columns_to_matrix(table, params)
    from .backends import HipBackend
    ref_stats = HipBackend(ctx).stats(table, params)
    cmp_res = compare_stats(ref_stats, compute_stats_from_draws(actual, ctx), tolerance=tolerance, metrics=metrics,
                            context=ctx)
    lens = {len(actual[p]) for p in params}
    ks: dict[str, float] = {}
    w1: dict[str, float] = {}
    if len(lens) == 1 and params:
        act = np.stack([np.asarray(actual[p], dtype=np.float64) for p in params])
        k, w = ctx.two_sample(ref, act)
        ks = {p: float(k[i]) for i, p in enumerate(params)}
        w1 = {p: float(w[i]) for i, p in enumerate(params)}
    else:
        for i, p in enumerate(params):
            k, w = ctx.two_sample(ref[i:i + 1], np.asarray(actual[p], dtype=np.float64)[None, :])
            ks[p], w1[p] = float(k[0]), float(w[0])
    scaled = {p: (w1[p] / ref_stats[p]["std"] if ref_stats[p]["std"] > 0 else float("inf") if w1[p] > 0 else 0.0)
              for p in params}
    failures = list(cmp_res.failures)
    if ks_max is not None:
        failures += [f"{p}.ks={ks[p]:.3g} > {ks_max}" for p in params if not ks[p] <= ks_max]
    if w1_scaled_max is not None:
        failures += [f"{p}.w1_scaled={scaled[p]:.3g} > {w1_scaled_max}" for p in params if not scaled[p] <= w1_scaled_max]
    return ValidateResult(passed=not failures, compare=cmp_res, ks=ks, wasserstein=w1, wasserstein_scaled=scaled,
                          failures=failures)
