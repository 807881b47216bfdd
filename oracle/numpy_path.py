"""NumPy restatement of the hot path (test infrastructure + the "NumPy CPU path" timing of SURVEY 8(d)).

Same algorithm as oracle/mcr_oracle.c (which follows src/mcmc_ref/diagnostics.py line by line) but vectorised the
way a NumPy user would write it: one argsort per parameter, tie groups by run boundaries, scipy's ndtri for
Phi^-1, autocovariances by FFT (all lags at once) followed by the reference's first-negative-rho truncation.
It is checked against the C oracle (tests/test_oracle_golden.py) to 1e-9 -- ndtri and the FFT differ from
AS241 / direct sums in the last bits -- and is NEVER imported by the product (mcmc-db_amd/).
"""
from __future__ import annotations

import numpy as np
from scipy.special import ndtri


def _rank_z(flat: np.ndarray) -> np.ndarray:
    """diagnostics.py:101-133: average ranks of the pooled draws -> Phi^-1((r - 0.5) / M)."""
    M = flat.size
    order = np.argsort(flat, kind="stable")
    s = flat[order]
    new = np.empty(M, dtype=bool)
    new[0] = True
    np.not_equal(s[1:], s[:-1], out=new[1:])
    start = np.flatnonzero(new)                       # first sorted position of every tie run
    end = np.append(start[1:], M)
    avg = (start + 1 + end) / 2.0                     # (i + 1 + j) / 2, 1-based average rank
    ranks = np.empty(M)
    ranks[order] = np.repeat(avg, end - start)
    return ndtri((ranks - 0.5) / M)


def _rhat(y: np.ndarray) -> float:
    """diagnostics.py:136-151 on y[m][n]."""
    m, n = y.shape
    if m < 2 or n < 2:
        return float("nan")
    mu = y.mean(axis=1)
    B = n * np.sum((mu - mu.mean()) ** 2) / (m - 1)
    W = np.mean(y.var(axis=1, ddof=1))
    if W == 0.0:
        return 1.0 if B == 0.0 else float("inf")
    return float(np.sqrt(((n - 1) / n * W + B / n) / W))


def _ess(z: np.ndarray) -> tuple[float, int]:
    """diagnostics.py:154-193 on z[m][n] (unsplit chains): FFT autocovariances, first-negative truncation."""
    m, n = z.shape
    if n < 2:
        return float("nan"), 0
    mu = z.mean(axis=1, keepdims=True)
    W = np.mean(z.var(axis=1, ddof=1))
    B = n * np.sum((mu[:, 0] - mu.mean()) ** 2) / (m - 1) if m > 1 else 0.0
    vhat = (n - 1) / n * W + B / n
    if vhat == 0.0:
        return float(m * n), 0
    d = z - mu
    f = np.fft.rfft(d, 2 * n, axis=1)
    ac = np.fft.irfft(f * np.conj(f), 2 * n, axis=1)[:, :n]          # sum_i d_i d_{i+l}
    rho = (ac[:, 1:] / (n - np.arange(1, n))).sum(axis=0) / (m * vhat)
    neg = np.flatnonzero(rho < 0)
    L = int(neg[0]) if neg.size else n - 1                            # lags 1..L enter the sum
    return float(m * n / (1.0 + 2.0 * rho[:L].sum())), L


def summarize(draws: np.ndarray, quantiles=(0.05, 0.5, 0.95)) -> dict:
    """draws [P][C][N] float64, equal-length chains, C >= 2.  Same keys as oracle.summarize."""
    P, C, N = draws.shape
    half = N // 2
    out = {k: np.full(P, np.nan) for k in ("mean", "std", "median", "rhat", "rhat_bulk", "rhat_tail", "ess_bulk", "ess_tail")}
    out["q"] = np.full((P, len(quantiles)), np.nan)
    out["lag_bulk"] = np.zeros(P, dtype=np.int64)
    out["lag_tail"] = np.zeros(P, dtype=np.int64)
    for p in range(P):
        x = draws[p]
        flat = x.reshape(-1)
        out["mean"][p] = flat.mean()
        out["std"][p] = flat.std()
        out["q"][p] = np.quantile(flat, quantiles)
        med = np.median(flat)
        out["median"][p] = med
        zb = _rank_z(flat).reshape(C, N)
        zt = _rank_z(np.abs(flat - med)).reshape(C, N)
        split = lambda z: np.concatenate([z[:, :half], z[:, half:2 * half]], axis=0)   # noqa: E731
        rb, rt = _rhat(split(zb)), _rhat(split(zt))
        out["rhat_bulk"][p], out["rhat_tail"][p] = rb, rt
        out["rhat"][p] = rt if rt > rb else rb
        out["ess_bulk"][p], out["lag_bulk"][p] = _ess(zb)
        out["ess_tail"][p], out["lag_tail"][p] = _ess(zt)
    return out
