"""CmdStan chain CSVs -> parameter draws -> chain-list JSON-zip (the step in front of `convert_file`).

Behaviour of the reference's src/mcmc_ref/cmdstan_generate.py:13-88: `#` comment lines are dropped, sampler-internal
columns (names ending in `__`) are skipped, `theta.1.2` becomes `theta[1,2]`, a payload is one dict per chain with
equal parameter sets and equal draw counts.  The CSV body is parsed column-wise with numpy (one pass over the text)
instead of a DictReader row loop; `chains_tensor` additionally hands the chains over as the `[P][C][N]` array the
kernels consume, so CmdStan output can go to `Context.summarize` without the JSON round trip.
"""
from __future__ import annotations

import io
import json
import re
import zipfile
from pathlib import Path
from typing import Any

import numpy as np

_INDEXED = re.compile(r"^([A-Za-z_][A-Za-z0-9_]*)((?:\.\d+)+)$")


def _normalize_cmdstan_param_name(name: str) -> str:
    """`theta.1.2` -> `theta[1,2]`; names without a numeric dotted suffix are returned unchanged."""
    m = _INDEXED.match(name)
    if m is None:
        return name
    return f"{m.group(1)}[{','.join(m.group(2)[1:].split('.'))}]"


def read_cmdstan_csv(path: Path) -> tuple[list[str], np.ndarray]:
    """(normalised parameter names, draws[P][N] float64) of one CmdStan chain file."""
    with Path(path).open() as f:
        body = [line for line in f if not line.startswith("#")]
    if not body:
        return [], np.empty((0, 0))
    header = [h.strip() for h in body[0].rstrip("\r\n").split(",")]
    keep = [i for i, h in enumerate(header) if h and not h.endswith("__")]
    rows = [ln for ln in body[1:] if ln.strip()]
    if rows:
        data = np.loadtxt(io.StringIO("".join(rows)), delimiter=",", dtype=np.float64, ndmin=2)
        if data.shape[1] != len(header):
            raise ValueError(f"{path}: rows have {data.shape[1]} fields, header has {len(header)}")
    else:
        data = np.empty((0, len(header)))
    names: list[str] = []
    cols: list[np.ndarray] = []
    for i in keep:                      # two raw names normalising to one parameter are concatenated, as setdefault/append does
        n = _normalize_cmdstan_param_name(header[i])
        if n in names:
            k = names.index(n)
            cols[k] = np.concatenate([cols[k], data[:, i]])
        else:
            names.append(n)
            cols.append(data[:, i])
    if not cols:
        return [], np.empty((0, data.shape[0]))
    if len({c.size for c in cols}) != 1:
        raise ValueError(f"{path}: columns of unequal length after name normalisation")
    return names, np.ascontiguousarray(np.stack(cols))


def parse_cmdstan_csv(path: Path) -> dict[str, list[float]]:
    """{param: draws} of one chain file (the reference's return shape; no rows -> empty dict)."""
    names, x = read_cmdstan_csv(path)
    if x.shape[1] == 0:
        return {}
    return {n: x[i].tolist() for i, n in enumerate(names)}


def build_posteriordb_payload(chain_draws: list[dict[str, list[float]]]) -> list[dict[str, list[float]]]:
    """Validates list[chain][param] -> draws and returns it (same messages as the reference)."""
    if not chain_draws:
        raise ValueError("no chain draws provided")
    params = set(chain_draws[0])
    if not params:
        raise ValueError("chain draws contain no parameters")
    for idx, chain in enumerate(chain_draws):
        if set(chain) != params:
            raise ValueError(f"chain {idx} parameter keys mismatch")
        if len({len(v) for v in chain.values()}) != 1:
            raise ValueError(f"chain {idx} has inconsistent draw counts")
    return chain_draws


def chains_tensor(paths: list[Path]) -> tuple[list[str], np.ndarray]:
    """Chain files of one model -> (parameter names, x[P][C][N]) for `Context.summarize(x, "pcn")`."""
    per_chain = [read_cmdstan_csv(p) for p in paths]
    if not per_chain:
        raise ValueError("no chain draws provided")
    names = per_chain[0][0]
    if not names:
        raise ValueError("chain draws contain no parameters")
    for idx, (n, x) in enumerate(per_chain):
        if set(n) != set(names):
            raise ValueError(f"chain {idx} parameter keys mismatch")
    n_draws = min(x.shape[1] for _, x in per_chain)
    out = np.empty((len(names), len(per_chain), n_draws))
    for c, (n, x) in enumerate(per_chain):
        for i, name in enumerate(names):
            out[i, c] = x[n.index(name), :n_draws]
    return names, out


def write_posteriordb_json_zip(payload: list[dict[str, list[float]]], out_path: Path, *, model_name: str) -> Path:
    out_path = Path(out_path)
    out_path.parent.mkdir(parents=True, exist_ok=True)
    with zipfile.ZipFile(out_path, "w", compression=zipfile.ZIP_DEFLATED) as zf:
        zf.writestr(f"{model_name}.json", json.dumps(payload))
    return out_path


def write_provenance(path: Path, data: dict[str, Any]) -> Path:
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    path.write_text(json.dumps(data, indent=2, sort_keys=True))
    return path
