"""Comparison utilities for reference vs actual draws (GPU arithmetic, reference's result types).

Mirrors src/mcmc_ref/compare.py:9-68: frozen dataclasses `ParamResult` / `CompareResult` with the
same fields, `compare_stats` with the same failure strings, `compute_basic_stats` /
`compute_stats_from_draws`.  The moments of the `actual` draws run in libmcmcref_hip; the O(P) relative-error
arithmetic stays on the host unless there are enough pairs to be worth a launch (`mcr_compare`).
"""
from __future__ import annotations

from collections.abc import Mapping, Sequence
from dataclasses import dataclass

import numpy as np

from . import _ffi


GPU_COMPARE_MIN_PAIRS = 4096      # below this many (ref, actual) pairs the gate is evaluated on the host


@dataclass(frozen=True)
class ParamResult:
    ref: float
    actual: float
    rel_error: float
    passed: bool


@dataclass(frozen=True)
class CompareResult:
    passed: bool
    details: dict[str, dict[str, ParamResult]]
    failures: list[str]


def compare_stats(
    ref_stats: Mapping[str, Mapping[str, float]],
    actual_stats: Mapping[str, Mapping[str, float]],
    tolerance: float,
    metrics: Sequence[str],
    context=None,
) -> CompareResult:
    details: dict[str, dict[str, ParamResult]] = {}
    failures: list[str] = []
    order: list[tuple[int, str | None, str | None]] = []      # keeps the reference's failure order
    refs: list[float] = []
    acts: list[float] = []
    for param, stats in ref_stats.items():
        if param not in actual_stats:
            order.append((-1, param, None))
            continue
        for metric in metrics:
            order.append((len(refs), param, metric))
            refs.append(float(stats.get(metric, float("nan"))))
            acts.append(float(actual_stats[param].get(metric, float("nan"))))
    rel = ok = None
    if len(refs) >= GPU_COMPARE_MIN_PAIRS:
        ctx = context or _ffi.default_context()
        rel, ok = ctx.compare(refs, acts, float(tolerance))
    elif refs:      # a handful of scalars: compare.py:41-43 as it stands (a launch + two copies would be pure latency)
        rel = [abs(a - r) / max(abs(r), 1e-12) for r, a in zip(refs, acts)]
        ok = [e <= tolerance for e in rel]
    for k, param, metric in order:
        if k < 0:
            failures.append(f"missing param: {param}")
            continue
        rel_error, passed = float(rel[k]), bool(ok[k])
        if not passed:
            failures.append(f"{param}.{metric} rel_error={rel_error:.3g} > {tolerance}")
        details.setdefault(param, {})[metric] = ParamResult(ref=refs[k], actual=acts[k], rel_error=rel_error,
                                                            passed=passed)
    for param in ref_stats:
        if param in actual_stats:
            details.setdefault(param, {})
    return CompareResult(passed=len(failures) == 0, details=details, failures=failures)


def compute_basic_stats(values: Sequence[float], context=None) -> dict[str, float]:
    n = len(values)
    if n == 0:
        return {"mean": float("nan"), "std": float("nan")}
    ctx = context or _ffi.default_context()
    try:
        out = ctx.basic_stats(np.asarray(values, dtype=np.float64))
    except _ffi.McrError as exc:
        raise ValueError(exc.message) from exc
    return {"mean": float(out["mean"]), "std": float(out["std"])}


def compute_stats_from_draws(draws: Mapping[str, Sequence[float]], context=None) -> dict[str, dict[str, float]]:
    """All parameters of `actual` in one launch when they have equal lengths, else one by one."""
    names = list(draws)
    lens = {len(draws[p]) for p in names}
    if len(names) > 1 and len(lens) == 1 and next(iter(lens)) > 0:
        ctx = context or _ffi.default_context()
        x = np.stack([np.asarray(draws[p], dtype=np.float64).reshape(-1) for p in names])
        t = ctx.upload(x.reshape(len(names), 1, -1), "pcn")
        try:
            mean, std = ctx.moments(t)
        except _ffi.McrError as exc:
            raise ValueError(exc.message) from exc
        finally:
            t.free()
        return {p: {"mean": float(mean[i]), "std": float(std[i])} for i, p in enumerate(names)}
    return {param: compute_basic_stats(values, context) for param, values in draws.items()}
