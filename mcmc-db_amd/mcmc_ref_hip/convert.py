"""The diagnostics call site of the conversion pipeline, batched for the GPU.

Mirrors the hot-path part of src/mcmc_ref/convert.py: `_chains_from_table` (:150-161, here a
whole-table integer gather `table_to_tensor`), `_count_chains_draws` (:123-131),
`_compute_diagnostics` (:134-147, one kernel pipeline per model instead of a Python loop per
parameter), `_checks` (:164-172) and `_enforce_checks` (:175-178).
"""
from __future__ import annotations

import json
import zipfile
from collections.abc import Iterable
from dataclasses import dataclass
from datetime import date
from pathlib import Path
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


def summarize_table(table: Any, params: Iterable[str], *, min_chains: int = 4, context=None,
                    quantiles=(0.05, 0.5, 0.95)) -> dict[str, dict[str, float]]:
    """Backend.stats + diagnostics of every parameter of a long table from ONE kernel pipeline."""
    params = list(params)
    if not params:
        return {}
    x, counts = table_to_tensor(table, params)
    C = len(counts)
    if C == 0 or not np.all(counts == counts[0]):
        raise ValueError("summarize_table needs chains of equal length")
    ctx = context or _ffi.default_context()
    qs = list(quantiles)
    try:
        r = ctx.summarize(x.reshape(len(params), C, int(counts[0])), "pcn", min_chains=min_chains, quantiles=qs)
    except _ffi.McrError as exc:
        raise ValueError(exc.message) from exc
    out: dict[str, dict[str, float]] = {}
    for i, p in enumerate(params):
        e = {"mean": float(r["mean"][i]), "std": float(r["std"][i])}
        for q, v in zip(qs, r["q"][i], strict=False):
            e[f"q{int(q * 100)}"] = float(v)
        e.update(rhat=float(r["rhat"][i]), ess_bulk=float(r["ess_bulk"][i]), ess_tail=float(r["ess_tail"][i]))
        out[p] = e
    return out


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


# ---- file conversion (reference src/mcmc_ref/convert.py:26-120): JSON-zip / CSV -> Parquet + meta ----
@dataclass(frozen=True)
class ConvertResult:
    draws_path: Path
    meta_path: Path
    meta: dict


def _read_json_zip(path: Path):
    """Chain-list JSON-zip: [ {param: [draws...]}, ... ] (one dict per chain) -> long Arrow table.

    As the reference builds it (convert.py:78-102): parameters in sorted order, `n_draws` taken from the first chain's
    first parameter, longer chains cut at `n_draws`, a SHORTER chain is an IndexError, and a column keeps the type its
    JSON numbers have (all-integer draws stay int64 in the written Parquet file, any float makes it double)."""
    import pyarrow as pa
    with zipfile.ZipFile(path) as zf:
        payload = json.loads(zf.read(zf.namelist()[0]))
    if not isinstance(payload, list) or not payload:
        raise ValueError("json-zip payload must be a non-empty list of chains")
    params = sorted(payload[0].keys())
    n_draws = len(next(iter(payload[0].values())))
    n_chains = len(payload)
    cols = {"chain": np.repeat(np.arange(n_chains, dtype=np.int64), n_draws),
            "draw": np.tile(np.arange(n_draws, dtype=np.int64), n_chains)}
    for p in params:
        parts = []
        for ch in payload:
            if n_draws and len(ch[p]) < n_draws:
                raise IndexError("list index out of range")
            parts.append(np.asarray(ch[p][:n_draws]))
        col = np.concatenate(parts) if n_draws else np.empty(0)
        if col.dtype.kind not in "iuf":           # bools / None / strings: let pyarrow infer (and refuse) as it would
            col = [v for part in parts for v in part.tolist()]
        cols[p] = col
    return pa.table(cols)


def _read_input(path: Path):
    import pyarrow.csv as pacsv
    if path.suffix == ".csv":
        return pacsv.read_csv(path)
    if path.suffixes[-2:] == [".json", ".zip"]:
        return _read_json_zip(path)
    raise ValueError(f"Unsupported input format: {path}")


def _ensure_chain_draw(table):
    """Add the missing bookkeeping columns like the reference does (convert.py:105-120): a missing
    `draw` is the row number, a missing `chain` is chain 0 (int32, appended after the parameters)."""
    import pyarrow as pa
    cols = set(table.column_names)
    n = table.num_rows
    chain = pa.array(np.zeros(n, dtype=np.int32))
    draw = pa.array(np.arange(n, dtype=np.int32))
    if "chain" in cols and "draw" in cols:
        return table
    if "chain" in cols:
        return table.append_column("draw", draw)
    if "draw" in cols:
        return table.append_column("chain", chain)
    return table.append_column("chain", chain).append_column("draw", draw)


def convert_files(jobs, out_draws_dir: Path, out_meta_dir: Path, force: bool = False, source: str = "converted",
                  context=None) -> list:
    """`convert_file` for many inputs with the kernels pipelined: jobs = [(input_path, name), ...]; returns, per job,
    a ConvertResult or the exception that `convert_file` would have raised for it (the per-recipe try/except of
    generate.generate_reference_corpus, src/mcmc_ref/generate.py:77-96, becomes per-entry results).

    All inputs are read and laid out first, the rectangular models are uploaded and enqueued with a rolling window of
    MCR_MAX_INFLIGHT calls (consecutive models overlap on the context's lanes; a NaN draw or any other kernel-side
    failure stays confined to its model), ragged models take the per-parameter route, then the quality gate and the
    two files of every model are written."""
    import pyarrow.parquet as pq
    out_draws_dir, out_meta_dir = Path(out_draws_dir), Path(out_meta_dir)
    min_chains = 1 if force else 4
    n = len(jobs)
    results: list = [None] * n
    prepared: dict[int, tuple] = {}
    for i, (input_path, _name) in enumerate(jobs):
        try:
            table = _ensure_chain_draw(_read_input(Path(input_path)))
            params = [c for c in table.column_names if c not in {"chain", "draw"}]
            n_chains, n_draws = _count_chains_draws(table)
            x, counts = table_to_tensor(table, params)
            if params and len(counts) < min_chains:
                raise ValueError(f"R-hat diagnostics require at least {min_chains} chains; got {len(counts)} chain(s)")
            prepared[i] = (table, params, n_chains, n_draws, x, counts)
        except Exception as exc:  # noqa: BLE001 - reported per job
            results[i] = exc
    diags: dict[int, dict] = {}
    ctx = None
    if any(v[1] for v in prepared.values()):
        ctx = context or _ffi.default_context()
    window: list[tuple[int, object, object]] = []          # (job, device tensor, the result buffers enqueue() handed back)
    if ctx is not None and ctx.inflight:
        ctx.wait()       # somebody else's calls on this context: wait_one() below must deliver OUR oldest call, nobody else's

    def retire():
        i, t, mine = window.pop(0)
        try:
            got = ctx.wait_one()
            if got is not mine:        # cannot happen on a context drained above; never hand a model another call's numbers
                raise RuntimeError("convert_files: the context delivered a result that is not this model's")
            r = got.result()
            diags[i] = {p: {"rhat": float(r["rhat"][k]), "ess_bulk": float(r["ess_bulk"][k]),
                            "ess_tail": float(r["ess_tail"][k])} for k, p in enumerate(prepared[i][1])}
        except _ffi.McrError as exc:
            results[i] = ValueError(exc.message)
        finally:
            t.free()

    ragged = []
    try:
        for i, (table, params, n_chains, n_draws, x, counts) in prepared.items():
            if not params:
                diags[i] = {}
                continue
            if not np.all(counts == counts[0]):
                ragged.append(i)
                continue
            if len(window) == _ffi.MCR_MAX_INFLIGHT:
                retire()
            try:
                t = ctx.upload(x.reshape(len(params), len(counts), int(counts[0])), "pcn")
                try:
                    bufs = ctx.enqueue(t, min_chains=min_chains, quantiles=())
                except Exception:
                    t.free()
                    raise
                window.append((i, t, bufs))
            except _ffi.McrError as exc:
                results[i] = ValueError(exc.message)
        while window:
            retire()
    finally:
        # anything but a kernel-side failure of one model (handled above) ends the batch: nothing stays in flight and no
        # device tensor stays allocated behind the exception
        if window:
            try:
                ctx.wait()
            except Exception:  # noqa: BLE001 - the original exception is the one to report
                pass
            for _i, t, _b in window:
                t.free()
            window.clear()
    for i in ragged:                                  # chains of unequal length: one pipeline per parameter
        table, params = prepared[i][:2]
        try:
            diags[i] = _compute_diagnostics(table, params, min_chains=min_chains, context=ctx)
        except Exception as exc:  # noqa: BLE001
            results[i] = exc
    for i, (table, params, n_chains, n_draws, _x, _counts) in prepared.items():
        if results[i] is not None:
            continue
        name = jobs[i][1]
        try:
            checks = _checks(n_chains, n_draws, diags[i])
            if not force:
                _enforce_checks(checks)
            meta = {"model": name, "parameters": params, "n_chains": n_chains, "n_draws_per_chain": n_draws,
                    "diagnostics": diags[i], "generated_date": date.today().isoformat(), "checks": checks,
                    "source": source}
            draws_path = out_draws_dir / f"{name}.draws.parquet"
            meta_path = out_meta_dir / f"{name}.meta.json"
            pq.write_table(table, draws_path)
            meta_path.write_text(json.dumps(meta, indent=2, sort_keys=True))
            results[i] = ConvertResult(draws_path=draws_path, meta_path=meta_path, meta=meta)
        except Exception as exc:  # noqa: BLE001
            results[i] = exc
    return results


def convert_file(input_path: Path, name: str, out_draws_dir: Path, out_meta_dir: Path, force: bool = False,
                 source: str = "converted") -> ConvertResult:
    """Same contract as the reference's convert_file (convert.py:26-67): diagnostics of every parameter (one GPU
    pipeline per file), quality checks (raise ValueError("quality checks failed: ...") unless `force`), then
    `<name>.draws.parquet` and `<name>.meta.json` (sorted keys, indent 2)."""
    res = convert_files([(input_path, name)], out_draws_dir, out_meta_dir, force=force, source=source)[0]
    if isinstance(res, Exception):
        raise res
    return res
