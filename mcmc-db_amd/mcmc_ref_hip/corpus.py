"""Corpus-shaped synthetic workload (BASELINE configs 2 and 3).

The packaged reference corpus itself (41 MB of Parquet) does not travel to the GPU box; what does
is its shape table -- (chains, draws, params) of the 57 packaged models, data/corpus_shapes.json --
and the six real models kept as fixtures under tests/golden/models.  `synthetic_corpus` fills every
shape with the deterministic AR(1) generator of synth.py, so the workload has the corpus's launch
geometry (55 models of 10 x 1000 draws, two of 4 x 2500, 2..45 parameters each, 4.6 M param-draws).
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np

from . import synth

_SHAPES = Path(__file__).resolve().parent / "data" / "corpus_shapes.json"


def corpus_shapes() -> list[dict]:
    return json.loads(_SHAPES.read_text())["models"]


def synthetic_corpus(seed: int = 4711, real_models: dict | None = None) -> list[tuple[str, np.ndarray]]:
    """[(model name, draws [P][C][N] float64)] for every packaged model shape.

    real_models: optional {name: array} of real draws to substitute (e.g. the test fixtures)."""
    out = []
    for k, m in enumerate(corpus_shapes()):
        name = m["model"]
        if real_models and name in real_models:
            out.append((name, np.ascontiguousarray(real_models[name], dtype=np.float64)))
            continue
        out.append((name, synth.c1_model(m["n_chains"], m["n_draws"], m["n_params"], seed=seed + 17 * k)))
    return out
