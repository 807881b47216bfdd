"""Python API for reference draws, with the statistics on the GPU.

Same surface as the reference's src/mcmc_ref/reference.py:15-122 (`list_models`, `stan_data`,
`model_code`, `stats`, `draws`, `diagnostics_for_model`, `compare`); the default backend is "hip".
`summary_for_model` is an addition: stats and diagnostics of a model from ONE kernel pipeline.
"""
from __future__ import annotations

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


def stats(model: str, params: Sequence[str] | None = None, backend: str = "hip", quantile_mode: str = "exact",
          store: DataStore | None = None) -> dict[str, dict[str, float]]:
    """Summary statistics (mean, std, q5, q50, q95) of a model's reference draws."""
    table, params = _table_and_params(store or DataStore(), model, params)
    return get_backend(backend).stats(table, params, quantile_mode=quantile_mode)


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
    table, params = _table_and_params(store, model, params)
    return convert._compute_diagnostics(table, params)


def summary_for_model(model: str, params: Sequence[str] | None = None, store: DataStore | None = None,
                      min_chains: int = 4) -> dict[str, dict[str, float]]:
    """mean, std, q5, q50, q95, rhat, ess_bulk, ess_tail per parameter from one pass of the kernels."""
    table, params = _table_and_params(store or DataStore(), model, params)
    return convert.summarize_table(table, params, min_chains=min_chains)


def compare(model: str, actual: Mapping[str, Sequence[float]], tolerance: float = 0.15,
            metrics: Sequence[str] = ("mean", "std"), backend: str = "hip", store: DataStore | None = None):
    """Compare actual draws against the reference statistics (reference.py:107-122)."""
    ref_stats = stats(model, params=list(actual.keys()), backend=backend, store=store)
    actual_stats = compute_stats_from_draws(actual)
    return compare_stats(ref_stats, actual_stats, tolerance=tolerance, metrics=metrics)
