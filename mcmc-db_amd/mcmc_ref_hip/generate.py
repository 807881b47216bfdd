"""provenance-generate / provenance-publish with the diagnostics batched on the GPU.

Same flow, result types, error texts and exit behaviour as the reference's src/mcmc_ref/generate.py:46-162
(`generate_reference_corpus`, `publish_reference_data`, `fake_jsonzip_runner`): a sampler runner writes one chain-list
JSON-zip per recipe, `convert` turns it into `draws/<m>.draws.parquet` + `meta/<m>.meta.json`, failures are collected
per recipe in `errors{name: message}` and never stop the run.  What differs is the schedule: the reference runs
runner -> convert_file -> (per-parameter Python diagnostics) one recipe at a time; here every archive is produced
first and `convert.convert_files` then pushes ALL models through the kernel pipeline with a rolling window of
MCR_MAX_INFLIGHT calls over the context's lanes.

Recipes are read from the scaffold on disk (`provenance_manifest.json["models"]`, else the file stems of
`stan_models/` and `stan_data/`): the Stan programs and data literals of the reference's provenance.py are data
definitions outside the statistics path and are not restated here.
"""
from __future__ import annotations

import hashlib
import importlib
import json
import shutil
import zipfile
from collections.abc import Callable
from dataclasses import dataclass
from pathlib import Path

from . import convert


@dataclass(frozen=True)
class GenerationConfig:
    """CmdStan run of the published corpus (src/mcmc_ref/provenance.py:16-22)."""
    chains: int = 10
    iter_sampling: int = 10_000
    iter_warmup: int = 10_000
    thin: int = 10
    seed: int = 4711


@dataclass(frozen=True)
class ModelRecipe:
    """What a runner needs to know about a recipe here: its name (files come as arguments)."""
    name: str


@dataclass(frozen=True)
class GenerationResult:
    generated: int
    failed: int
    output_root: Path
    errors: dict[str, str]


@dataclass(frozen=True)
class PublishResult:
    draws_copied: int
    meta_copied: int
    pairs_copied: int
    package_root: Path


RecipeRunner = Callable[..., None]


def scaffold_recipes(scaffold_root: Path) -> list[ModelRecipe]:
    """Recipe names of a scaffold, sorted: the manifest's model list, else whatever Stan programs / data it holds."""
    scaffold_root = Path(scaffold_root)
    manifest = scaffold_root / "provenance_manifest.json"
    names: set[str] = set()
    if manifest.is_file():
        try:
            names.update(str(n) for n in json.loads(manifest.read_text()).get("models", []))
        except (ValueError, AttributeError):
            pass
    if not names:
        names.update(p.name[:-len(".stan")] for p in (scaffold_root / "stan_models").glob("*.stan"))
        names.update(p.name[:-len(".json")] for p in (scaffold_root / "stan_data").glob("*.json"))
    return [ModelRecipe(n) for n in sorted(names)]


def _selected_recipes(scaffold_root: Path, models: list[str] | None) -> list[ModelRecipe]:
    known = {r.name: r for r in scaffold_recipes(scaffold_root)}
    if models is None:
        return list(known.values())
    missing = [m for m in models if m not in known]
    if missing:
        raise ValueError(f"unknown model recipe(s): {', '.join(sorted(missing))}")
    return [known[m] for m in models]


def generate_reference_corpus(*, scaffold_root: Path, output_root: Path, models: list[str] | None = None,
                              config: GenerationConfig | None = None, force: bool = False,
                              runner: RecipeRunner | None = None, context=None) -> GenerationResult:
    scaffold_root, output_root = Path(scaffold_root), Path(output_root)
    archives_dir, draws_dir, meta_dir = (output_root / d for d in ("archives", "draws", "meta"))
    for d in (archives_dir, draws_dir, meta_dir):
        d.mkdir(parents=True, exist_ok=True)
    selected = _selected_recipes(scaffold_root, models)
    config = config or GenerationConfig()
    runner = runner or _cmdstan_jsonzip_runner
    errors: dict[str, str] = {}
    jobs: list[tuple[Path, str]] = []
    # 1. samplers (CPU, external): one archive per recipe; a failing recipe is recorded and skipped
    for recipe in selected:
        stan_file = scaffold_root / "stan_models" / f"{recipe.name}.stan"
        data_file = scaffold_root / "stan_data" / f"{recipe.name}.json"
        archive_path = archives_dir / f"{recipe.name}.json.zip"
        if not stan_file.exists() or not data_file.exists():
            errors[recipe.name] = "missing scaffold files"
            continue
        try:
            runner(model_name=recipe.name, recipe=recipe, stan_file=stan_file, data_file=data_file,
                   archive_path=archive_path, config=config)
        except Exception as exc:  # noqa: BLE001 - the reference records any failure per recipe (generate.py:95-96)
            errors[recipe.name] = str(exc)
            continue
        jobs.append((archive_path, recipe.name))
    # 2. every archive through the kernels in one pipelined batch, then the quality gate + files per model
    results = convert.convert_files(jobs, out_draws_dir=draws_dir, out_meta_dir=meta_dir, force=force,
                                    source=_cmdstan_source(), context=context)
    generated = 0
    for (_, name), res in zip(jobs, results):
        if isinstance(res, Exception):
            errors[name] = str(res)
        else:
            generated += 1
    return GenerationResult(generated=generated, failed=len(errors), output_root=output_root, errors=errors)


def publish_reference_data(*, source_root: Path, scaffold_root: Path, package_root: Path) -> PublishResult:
    """Copies generated draws / meta and the scaffold's pairs into the data package and writes the sha256 manifest
    (reference generate.py:106-174; the layout `store.DataStore` resolves)."""
    source_root, scaffold_root, package_root = Path(source_root), Path(scaffold_root), Path(package_root)
    sources = [source_root / "draws", source_root / "meta"]
    absent = [str(p) for p in sources if not p.is_dir()]
    if absent:
        raise FileNotFoundError(f"source draws/meta directories must exist: {', '.join(absent)}")
    pairs_src = scaffold_root / "pairs"
    if not pairs_src.is_dir():
        raise FileNotFoundError(f"scaffold pairs directory not found: {pairs_src}")
    manifest_src = scaffold_root / "provenance_manifest.json"
    if not manifest_src.is_file():
        raise FileNotFoundError(f"scaffold provenance manifest not found: {manifest_src}")

    package_root.mkdir(parents=True, exist_ok=True)
    targets = {k: package_root / k for k in ("draws", "meta", "pairs")}
    for t in targets.values():                      # stale artefacts never survive a publish
        if t.exists():
            shutil.rmtree(t)
        t.mkdir(parents=True)

    def copy_all(src: Path, dst: Path, pattern: str) -> int:
        files = [p for p in sorted(src.glob(pattern)) if p.is_file()]
        for p in files:
            shutil.copy2(p, dst / p.name)
        return len(files)

    draws_copied = copy_all(sources[0], targets["draws"], "*.draws.parquet")
    meta_copied = copy_all(sources[1], targets["meta"], "*.meta.json")
    pair_dirs = [d for d in sorted(pairs_src.iterdir()) if d.is_dir()]
    for d in pair_dirs:
        shutil.copytree(d, targets["pairs"] / d.name, dirs_exist_ok=True)

    hashes = {}
    for p in sorted(package_root.rglob("*")):
        rel = p.relative_to(package_root).as_posix()
        if p.is_file() and rel != "provenance_manifest.json":
            hashes[rel] = hashlib.sha256(p.read_bytes()).hexdigest()
    manifest = {**json.loads(manifest_src.read_text()), "files": hashes}
    (package_root / "provenance_manifest.json").write_text(json.dumps(manifest, indent=2, sort_keys=True) + "\n")
    return PublishResult(draws_copied=draws_copied, meta_copied=meta_copied, pairs_copied=len(pair_dirs),
                         package_root=package_root)


def _write_jsonzip(path: Path, model_name: str, payload: list[dict[str, list[float]]]) -> None:
    path.parent.mkdir(parents=True, exist_ok=True)
    with zipfile.ZipFile(path, "w", compression=zipfile.ZIP_DEFLATED) as zf:
        zf.writestr(f"{model_name}.json", json.dumps(payload))


def fake_jsonzip_runner(*, model_name: str, recipe, stan_file: Path, data_file: Path, archive_path: Path,
                        config: GenerationConfig) -> None:
    """The reference's deterministic test double (`--fake-runner`, generate.py:177-193): chain c draws
    mu_i = c + 0.001 i, sigma_i = 1 + mu_i."""
    payload = []
    for c in range(config.chains):
        mu = [float(c) + 0.001 * float(i) for i in range(config.iter_sampling)]
        payload.append({"mu": mu, "sigma": [1.0 + v for v in mu]})
    _write_jsonzip(archive_path, model_name, payload)


def _draws_to_chain_payload(draws, names: list[str]) -> list[dict[str, list[float]]]:
    """cmdstanpy `fit.draws()` (draws x chains x columns, or a permutation of it) -> one dict per chain, sampler
    internals (`*__`) dropped (reference generate.py:230-254)."""
    if draws.ndim != 3:
        raise ValueError(f"Unexpected CmdStan draws shape: {draws.shape}")
    if draws.shape[2] != len(names):
        draws = draws.transpose(1, 0, 2)
        if draws.shape[2] != len(names):
            raise ValueError(f"Unexpected CmdStan draws shape: {draws.shape}")
    if draws.shape[1] > draws.shape[0]:
        draws = draws.transpose(1, 0, 2)
    keep = [k for k, n in enumerate(names) if not n.endswith("__")]
    return [{names[k]: [float(v) for v in draws[:, c, k]] for k in keep} for c in range(draws.shape[1])]


def _cmdstan_jsonzip_runner(*, model_name: str, recipe, stan_file: Path, data_file: Path, archive_path: Path,
                            config: GenerationConfig) -> None:
    try:
        model_cls = importlib.import_module("cmdstanpy").CmdStanModel
    except Exception as exc:  # pragma: no cover - cmdstanpy is absent in this image
        raise RuntimeError(
            "cmdstanpy is required for provenance generation. Install with: uv add --dev cmdstanpy") from exc
    fit = model_cls(stan_file=str(stan_file)).sample(
        data=str(data_file), chains=config.chains, iter_sampling=config.iter_sampling, iter_warmup=config.iter_warmup,
        thin=config.thin, seed=config.seed, show_progress=False)
    _write_jsonzip(archive_path, model_name, _draws_to_chain_payload(fit.draws(), list(fit.column_names)))


def _cmdstan_source() -> str:
    try:
        import cmdstanpy
        ver = cmdstanpy.cmdstan_version()
        return f"cmdstan-{ver[0]}.{ver[1]}" + (f".{ver[2]}" if len(ver) > 2 else ".0")
    except Exception:  # noqa: BLE001
        return "cmdstan-unknown"
