"""Locate and open reference draws: the on-disk layout contract of mcmc-ref, unchanged.

Layout (reference src/mcmc_ref/store.py:102-167, generate.py:130-155):

    <root>/draws/<model>.draws.parquet      long table: chain, draw, <param>...  (row = one draw)
    <root>/meta/<model>.meta.json           parameters, n_chains, n_draws_per_chain, diagnostics, checks
    <root>/stan_data/<model>.data.json
    <root>/stan_code|stan_models/<model>.stan

Two roots are searched, the packaged corpus first (`mcmc_ref_data` or `mcmc_ref` package data), then the
local one ($MCMC_REF_LOCAL_ROOT or ~/.mcmc-ref), exactly like the reference, so existing stores and the
provenance-generate -> provenance-publish flow keep working.  Host I/O only; no arithmetic on draws.
"""
from __future__ import annotations

import json
import os
from collections.abc import Sequence
from pathlib import Path

_SUBDIRS = ("draws", "meta", "pairs", "stan_data", "stan_code", "stan_models")


def _usable(root: Path | None) -> Path | None:
    if root is None:
        return None
    root = Path(root)
    return root if any((root / d).exists() for d in _SUBDIRS) else None


def default_local_root() -> Path:
    env = os.environ.get("MCMC_REF_LOCAL_ROOT")
    return Path(env) if env else Path.home() / ".mcmc-ref"


def default_packaged_root() -> Path | None:
    from importlib import resources
    for pkg in ("mcmc_ref_data", "mcmc_ref"):
        try:
            data = Path(str(resources.files(pkg).joinpath("data")))
        except Exception:
            continue
        if (data / "draws").exists() or (data / "meta").exists():
            return data
    return None


class DataStore:
    def __init__(self, local_root: Path | None = None, packaged_root: Path | None = None) -> None:
        self._local = _usable(local_root or default_local_root())
        self._packaged = _usable(packaged_root or default_packaged_root())

    # packaged first, then local (reference store.py:121-122)
    def _roots(self) -> list[Path]:
        return [r for r in (self._packaged, self._local) if r is not None]

    def _find(self, model: str, candidates: Sequence[tuple[str, str]], what: str) -> Path:
        for root in self._roots():
            for subdir, suffix in candidates:
                path = root / subdir / f"{model}{suffix}"
                if path.exists():
                    return path
        raise FileNotFoundError(f"{what} not found for model: {model}")

    def list_models(self) -> list[str]:
        names = set()
        for root in self._roots():
            for path in (root / "draws").glob("*.draws.parquet"):
                names.add(path.name[: -len(".draws.parquet")])
        return sorted(names)

    def resolve_draws_path(self, model: str) -> Path:
        return self._find(model, [("draws", ".draws.parquet")], "draws")

    def resolve_meta_path(self, model: str) -> Path:
        return self._find(model, [("meta", ".meta.json")], "metadata")

    def resolve_stan_data_path(self, model: str) -> Path:
        return self._find(model, [("stan_data", ".data.json")], "stan data")

    def resolve_stan_code_path(self, model: str) -> Path:
        return self._find(model, [("stan_code", ".stan"), ("stan_models", ".stan")], "stan code")

    def read_meta(self, model: str) -> dict:
        return json.loads(self.resolve_meta_path(model).read_text())

    def read_stan_data(self, model: str) -> dict:
        return json.loads(self.resolve_stan_data_path(model).read_text())

    def read_stan_code(self, model: str) -> str:
        return self.resolve_stan_code_path(model).read_text()

    def open_draws(self, model: str, params: Sequence[str] | None = None,
                   chains: Sequence[int] | None = None, batch_size: int = 1024):
        """RecordBatchReader over chain, draw and the selected parameter columns (column and chain
        push-down through pyarrow.dataset, as the reference does)."""
        import pyarrow.dataset as ds
        dataset = ds.dataset(self.resolve_draws_path(model), format="parquet")
        if params is None:
            params = [c for c in dataset.schema.names if c not in {"chain", "draw"}]
        filt = ds.field("chain").isin(list(chains)) if chains is not None else None
        return dataset.scanner(columns=["chain", "draw", *params], filter=filt, batch_size=batch_size).to_reader()
