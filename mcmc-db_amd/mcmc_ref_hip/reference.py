"""Python API for reference draws, with the statistics on the GPU.

Same surface as the reference's src/mcmc_ref/reference.py:15-122 (`list_models`, `stan_data`,
`model_code`, `stats`, `draws`, `diagnostics_for_model`, `compare`); the default backend is "hip".
`summary_for_model` is an addition: stats and diagnostics of a model from ONE kernel pipeline.
"""
from __future__ import annotations

import os
from collections.abc import Mapping, Sequence

from . import convert
from .backends import get_backend
from .compare import compare_stats, compute_stats_from_draws
from .draws import Draws, coerce_return
from .store import DataStore


def list_models(store: DataStore | None = None) -> list[str]:
    return (store or DataStore()).list_models()


def stan_data(model: str, store: DataStore | None = None) -> dict:
    return (store or DataStore()).read_stan_data(model)


def model_code(model: str, store: DataStore | None = None) -> str:
    return (store or DataStore()).read_stan_code(model)


def _table_and_params(store: DataStore, model: str, params):
    table = store.open_draws(model, params=params).read_all()
    if params is None:
        params = [c for c in table.column_names if c not in {"chain", "draw"}]
    return table, list(params)


def _native_reader() -> bool:
    """The "hip" backend reads `draws/<model>.draws.parquet` with the library's own Parquet ingest (decode on the
    GPU, no pyarrow) unless MCMC_REF_HIP_READER=arrow asks for the reference's pyarrow route."""
    return os.environ.get("MCMC_REF_HIP_READER", "native").lower() != "arrow"


def _native_summaries(store: DataStore, models: Sequence[str], params, *, diagnostics: bool, min_chains: int = 4):
    from . import _ffi, parquet
    ctx = _ffi.default_context()
    paths = [store.resolve_draws_path(m) for m in models]
    try:
        return parquet.summarize_files(ctx, paths, [params] * len(paths), min_chains=min_chains,
                                       diagnostics=diagnostics)
    except KeyError as exc:                      # unknown column: the same exception type pyarrow's path ends in
        raise KeyError(str(exc)) from exc


def stats(model: str, params: Sequence[str] | None = None, backend: str = "hip", quantile_mode: str = "exact",
          store: DataStore | None = None) -> dict[str, dict[str, float]]:
    """Summary statistics (mean, std, q5, q50, q95) of a model's reference draws."""
    store = store or DataStore()
    if backend == "hip" and _native_reader():
        get_backend(backend)                     # ImportError without the library / a device, as for any backend
        return _native_summaries(store, [model], params, diagnostics=False)[0]
    table, params = _table_and_params(store, model, params)
    return get_backend(backend).stats(table, params, quantile_mode=quantile_mode)


def summaries_for_models(models: Sequence[str], store: DataStore | None = None, min_chains: int = 4
                         ) -> dict[str, dict[str, dict[str, float]]]:
    """All statistics of many models from one batched Parquet decode + pipelined kernel passes
    (an addition; the reference loops `stats` / `diagnostics_for_model` per model)."""
    store = store or DataStore()
    out = _native_summaries(store, list(models), None, diagnostics=True, min_chains=min_chains)
    return dict(zip(models, out))


def draws(model: str, params: Sequence[str] | None = None, chains: Sequence[int] | None = None,
          return_: str = "arrow", store: DataStore | None = None):
    """return_: "arrow" | "draws" | "numpy" | "list" (as in the reference)."""
    store = store or DataStore()
    reader = store.open_draws(model, params=params, chains=chains)
    if params is None:
        reader = reader.read_all()
        params = [c for c in reader.column_names if c not in {"chain", "draw"}]
    return coerce_return(Draws(data=reader, params=list(params), chains=list(chains) if chains else None), return_)


def diagnostics_for_model(model: str, params: Sequence[str] | None = None,
                          store: DataStore | None = None) -> dict[str, dict[str, float]]:
    """meta.json's cached diagnostics when present (reference.py:82-90), else computed on the GPU for
    all parameters in one pipeline."""
    store = store or DataStore()
    try:
        meta = store.read_meta(model)
    except FileNotFoundError:
        meta = {}
    diag = meta.get("diagnostics")
    if isinstance(diag, dict) and diag:
        return diag if params is None else {p: diag[p] for p in params if p in diag}
    if _native_reader():
        full = _native_summaries(store, [model], params, diagnostics=True)[0]
        return {p: {k: v[k] for k in ("rhat", "ess_bulk", "ess_tail")} for p, v in full.items()}
    table, params = _table_and_params(store, model, params)
    return convert._compute_diagnostics(table, params)


def summary_for_model(model: str, params: Sequence[str] | None = None, store: DataStore | None = None,
                      min_chains: int = 4) -> dict[str, dict[str, float]]:
    """mean, std, q5, q50, q95, rhat, ess_bulk, ess_tail per parameter from one pass of the kernels."""
    store = store or DataStore()
    if _native_reader():
        return _native_summaries(store, [model], params, diagnostics=True, min_chains=min_chains)[0]
    table, params = _table_and_params(store, model, params)
    return convert.summarize_table(table, params, min_chains=min_chains)


def compare(model: str, actual: Mapping[str, Sequence[float]], tolerance: float = 0.15,
            metrics: Sequence[str] = ("mean", "std"), backend: str = "hip", store: DataStore | None = None):
    """Compare actual draws against the reference statistics (reference.py:107-122)."""
    ref_stats = stats(model, params=list(actual.keys()), backend=backend, store=store)
    actual_stats = compute_stats_from_draws(actual)
    return compare_stats(ref_stats, actual_stats, tolerance=tolerance, metrics=metrics)
