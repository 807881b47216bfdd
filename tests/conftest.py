"""Shared pytest setup.

`gpu` marks tests that need a real MI355X (run by the driver with `-m gpu`);
everything else must pass on a CPU-only box with `-m "not gpu"`.
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
GOLDEN = ROOT / "tests" / "golden"
for p in (ROOT, ROOT / "mcmc-db_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (HIP device)")


def load_json(name: str):
    return json.loads((GOLDEN / name).read_text())


MODEL_NAMES = sorted(p.stem for p in (GOLDEN / "models").glob("*.npz"))


def load_model(name: str):
    z = np.load(GOLDEN / "models" / f"{name}.npz", allow_pickle=False)
    rec = json.loads((GOLDEN / "models" / f"{name}.json").read_text())
    return z["draws"], [str(s) for s in z["params"]], rec


def same_float(a: float, b: float) -> bool:
    """Bit-level equality that treats NaN == NaN."""
    return (a != a and b != b) or a == b


def rel_close(a: float, b: float, rel: float) -> bool:
    if a != a or b != b:
        return a != a and b != b
    if a in (float("inf"), float("-inf")) or b in (float("inf"), float("-inf")):
        return a == b
    return abs(a - b) <= rel * max(abs(a), abs(b), 1e-300)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc
