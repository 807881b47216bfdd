"""`mcmc-ref-hip` CLI: the hot-path commands of the reference's `mcmc-ref` CLI on the GPU.

Mirrors src/mcmc_ref/cli.py for `list`, `stats`, `diagnostics`, `info`, `compare`, `convert`,
`provenance-generate` and `provenance-publish` (same options, echo strings and exit codes: compare exits 2 when the
gate fails, provenance-generate exits 1 when any recipe failed); `--backend` accepts "hip" (default) and the
reference's "arrow" / "numpy".  `provenance-scaffold` (Stan programs + data literals), pairs and draws-export
commands are outside the statistics path and not included.
"""
from __future__ import annotations

import json
from pathlib import Path

import click

from . import convert as convert_mod
from . import generate as generate_mod
from . import reference
from .store import DataStore


@click.group()
def main() -> None:
    """mcmc-ref-hip CLI."""


def _headers(stats: dict) -> list[str]:
    keys: set[str] = set()
    for metrics in stats.values():
        keys.update(metrics.keys())
    return sorted(keys)


def _print_table(stats: dict) -> None:
    headers = ["param"] + _headers(stats)
    widths = [max(len(h), 6) for h in headers]
    click.echo(" ".join(h.ljust(w) for h, w in zip(headers, widths, strict=False)))
    for param, metrics in stats.items():
        row = [param] + [f"{metrics.get(h, float('nan')):.6g}" for h in headers[1:]]
        click.echo(" ".join(v.ljust(w) for v, w in zip(row, widths, strict=False)))


@main.command("list")
@click.option("--format", "format_", type=click.Choice(["table", "json"], case_sensitive=False), default="table")
def list_cmd(format_: str) -> None:
    models = reference.list_models()
    if format_ == "json":
        click.echo(json.dumps(models, indent=2))
        return
    for m in models:
        click.echo(m)


@main.command("stats")
@click.argument("model")
@click.option("--params", default=None, help="Comma-separated parameter list")
@click.option("--format", "format_", type=click.Choice(["table", "csv", "json"], case_sensitive=False), default="table")
@click.option("--backend", type=click.Choice(["hip", "arrow", "numpy"], case_sensitive=False), default="hip")
@click.option("--quantile-mode", type=click.Choice(["exact"], case_sensitive=False), default="exact")
@click.option("--include-diagnostics", is_flag=True, help="Include rhat/ess metrics")
def stats_cmd(model, params, format_, backend, quantile_mode, include_diagnostics) -> None:
    param_list = params.split(",") if params else None
    stats = reference.stats(model, params=param_list, backend=backend, quantile_mode=quantile_mode)
    if include_diagnostics:
        for param, metrics in reference.diagnostics_for_model(model, params=param_list).items():
            stats.setdefault(param, {}).update(metrics)
    if format_ == "json":
        click.echo(json.dumps(stats, indent=2, sort_keys=True))
    elif format_ == "csv":
        headers = ["param"] + _headers(stats)
        click.echo(",".join(headers))
        for param, metrics in stats.items():
            click.echo(",".join([param] + [str(metrics.get(h, "")) for h in headers[1:]]))
    else:
        _print_table(stats)


@main.command("diagnostics")
@click.argument("model")
@click.option("--format", "format_", type=click.Choice(["table", "csv", "json"], case_sensitive=False), default="table")
def diagnostics_cmd(model: str, format_: str) -> None:
    diag = reference.diagnostics_for_model(model)
    if format_ == "json":
        click.echo(json.dumps(diag, indent=2, sort_keys=True))
    elif format_ == "csv":
        click.echo("param,rhat,ess_bulk,ess_tail")
        for param, m in diag.items():
            click.echo(",".join([param, str(m.get("rhat")), str(m.get("ess_bulk")), str(m.get("ess_tail"))]))
    else:
        _print_table(diag)


@main.command("info")
@click.argument("model")
def info_cmd(model: str) -> None:
    click.echo(json.dumps(DataStore().read_meta(model), indent=2, sort_keys=True))


def _read_actual_csv(path: Path) -> dict[str, list[float]]:
    import pyarrow.csv as pacsv
    table = pacsv.read_csv(path)
    return {p: [float(v) for v in table.column(p).to_pylist()]
            for p in table.column_names if p not in {"chain", "draw"}}


@main.command("compare")
@click.argument("model")
@click.option("--actual", "actual_path", type=click.Path(path_type=Path), required=True)
@click.option("--tolerance", default=0.15, type=float)
@click.option("--format", "format_", type=click.Choice(["table", "json"], case_sensitive=False), default="table")
def compare_cmd(model: str, actual_path: Path, tolerance: float, format_: str) -> None:
    result = reference.compare(model, actual=_read_actual_csv(actual_path), tolerance=tolerance)
    if format_ == "json":
        details = {p: {k: vars(v) for k, v in ms.items()} for p, ms in result.details.items()}
        click.echo(json.dumps({"passed": result.passed, "failures": result.failures, "details": details},
                              indent=2, sort_keys=True))
    else:
        click.echo("passed" if result.passed else "failed")
        for failure in result.failures:
            click.echo(f"- {failure}")
    raise SystemExit(0 if result.passed else 2)


@main.command("convert")
@click.argument("input_path", type=click.Path(path_type=Path))
@click.option("--name", required=True)
@click.option("--force", is_flag=True)
def convert_cmd(input_path: Path, name: str, force: bool) -> None:
    from .store import default_local_root
    local_root = default_local_root()
    draws_dir, meta_dir = local_root / "draws", local_root / "meta"
    draws_dir.mkdir(parents=True, exist_ok=True)
    meta_dir.mkdir(parents=True, exist_ok=True)
    convert_mod.convert_file(input_path, name=name, out_draws_dir=draws_dir, out_meta_dir=meta_dir, force=force)
    click.echo(f"converted {name} -> {draws_dir}")


@main.command("provenance-generate")
@click.option("--scaffold-root", type=click.Path(path_type=Path), required=True)
@click.option("--output-root", type=click.Path(path_type=Path), required=True)
@click.option("--models", default=None, help="Optional comma-separated recipe names.")
@click.option("--force", is_flag=True, help="Forward --force to convert quality checks.")
@click.option("--fake-runner", is_flag=True, help="Use deterministic fake runner (testing only).")
def provenance_generate_cmd(scaffold_root: Path, output_root: Path, models: str | None, force: bool,
                            fake_runner: bool) -> None:
    """Sampler archives -> draws/meta for every recipe of a scaffold (reference cli.py:248-274); the diagnostics of all
    models run as one pipelined batch on the GPU."""
    result = generate_mod.generate_reference_corpus(
        scaffold_root=scaffold_root, output_root=output_root, models=models.split(",") if models else None,
        force=force, runner=generate_mod.fake_jsonzip_runner if fake_runner else None)
    click.echo(f"generated={result.generated} failed={result.failed} output={result.output_root}")
    if result.errors:
        for name, message in sorted(result.errors.items()):
            click.echo(f"- {name}: {message}")
        raise SystemExit(1)


@main.command("provenance-publish")
@click.option("--source-root", type=click.Path(path_type=Path), required=True)
@click.option("--scaffold-root", type=click.Path(path_type=Path), required=True)
@click.option("--package-root", type=click.Path(path_type=Path), required=True)
def provenance_publish_cmd(source_root: Path, scaffold_root: Path, package_root: Path) -> None:
    result = generate_mod.publish_reference_data(source_root=source_root, scaffold_root=scaffold_root,
                                                 package_root=package_root)
    click.echo(f"published draws={result.draws_copied} meta={result.meta_copied} pairs={result.pairs_copied} "
               f"to={result.package_root}")


if __name__ == "__main__":
    main()
