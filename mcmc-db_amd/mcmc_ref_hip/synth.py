"""Deterministic synthetic draw tensors for tests and bench.

Counter-based: value (p, c, t) depends only on (seed, P, C, N, p, c, t), so any
parameter subset reproduces the same numbers as the full model.

  e      = BoxMuller(splitmix64(seed*G + 2i), splitmix64(seed*G + 2i + 1)),  i = (p*C + c)*N + t
  y_0    = e_0 ;  y_t = phi_p * y_{t-1} + sqrt(1 - phi_p^2) * e_t            (AR(1), unit variance)
  phi_p  = 0.95 * p / (P - 1)
  x      = p + 10**((p mod 7) - 3) * y
  p mod 25 == 24 : chain 3 shifted by +0.5 * sigma_p      (R-hat > 1.01 path)
  p mod 20 == 19 : rounded to 2 decimals                  (ties -> average-rank path)
"""
from __future__ import annotations

import numpy as np

_G = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x.astype(np.uint64) + _G
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def std_normal(seed: int, index: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        base = np.uint64(seed) * _G + np.uint64(2) * index.astype(np.uint64)
        h1 = splitmix64(base)
        h2 = splitmix64(base + np.uint64(1))
    u1 = ((h1 >> np.uint64(11)).astype(np.float64) + 0.5) * 2.0 ** -53
    u2 = (h2 >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def c1_model(C: int = 4, N: int = 10000, P: int = 100, seed: int = 4711, params=None,
             dtype=np.float64) -> np.ndarray:
    """Returns draws as [len(params)][C][N] (param-major, chain-major rows: the Arrow column layout)."""
    from scipy.signal import lfilter

    plist = list(range(P)) if params is None else list(params)
    out = np.empty((len(plist), C, N), dtype=np.float64)
    t = np.arange(N, dtype=np.uint64)
    for k, p in enumerate(plist):
        phi = 0.95 * p / (P - 1) if P > 1 else 0.0
        sigma = 10.0 ** ((p % 7) - 3)
        for c in range(C):
            e = std_normal(seed, np.uint64((p * C + c) * N) + t)
            if N > 0:
                e[1:] *= np.sqrt(1.0 - phi * phi)
                y = lfilter([1.0], [1.0, -phi], e)
            else:
                y = e
            x = p + sigma * y
            if p % 25 == 24 and c == 3:
                x = x + 0.5 * sigma
            if p % 20 == 19:
                x = np.round(x, 2)
            out[k, c] = x
    return out.astype(dtype, copy=False)


def stress_check_sample(P: int, per_chunk: int, count: int = 128) -> np.ndarray:
    """Which parameters of a large tensor (the stress shape: mu_p = p, sigma_p = 10**((p mod 7) - 3), f32) a test or bench.py re-computes on the host:

    * the first and the last parameter of EVERY workspace chunk (per_chunk = Context.params_per_chunk: chunk k is
      [k * per_chunk, (k + 1) * per_chunk)), i.e. both sides of every chunk edge;
    * every residue of p mod 7 (the generator's seven scales) at the low and at the high end of the tensor;
    * at least eight parameters whose sigma is at or below the f32 grid at mu_p (p >= 4096 with p mod 7 == 0: sigma = 1e-3
      against a grid step of 4.9e-4 .. 9.8e-4, so the 400 000 draws take a handful of distinct values: all ties);
    * an even spread over the tensor for the rest, `count` parameters in all (fewer if P is smaller)."""
    per_chunk = max(int(per_chunk), 1)
    sel = set()
    for p0 in range(0, P, per_chunk):
        sel.add(p0)
        sel.add(min(p0 + per_chunk, P) - 1)
    sel.update(range(min(7, P)))
    sel.update(range(max(P - 7, 0), P))
    tied = [p for p in range(P - 1, 4095, -1) if p % 7 == 0]
    sel.update(tied[:: max(len(tied) // 8, 1)][:10])
    want = min(count, P)
    if len(sel) < want:                         # an even spread over the parameters not chosen yet
        rest = np.array([p for p in range(P) if p not in sel])
        need = want - len(sel)
        sel.update(int(p) for p in rest[((np.arange(need) + 0.5) * len(rest) / need).astype(int)])
    return np.array(sorted(sel), dtype=np.int64)
