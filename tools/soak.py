#!/usr/bin/env python3
"""Soak: random shapes through the pipelined path (4 lanes, rolling window of 8 calls, tensors freed and reallocated)
against a single-lane synchronous context -- results must be bit-identical.  usage: soak.py [seconds]"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "mcmc-db_amd")]
import os
from mcmc_ref_hip import _ffi

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(time.time()) % 100000)
fast = _ffi.Context(0)
os.environ["MCR_LANES"] = "1"
slow = _ffi.Context(0)
os.environ.pop("MCR_LANES")
KEYS = ("mean", "std", "q", "median", "rhat", "rhat_bulk", "rhat_tail", "ess_bulk", "ess_tail", "lag_bulk", "lag_tail")
t_end = time.time() + budget
calls = pd = 0
rounds = 0
while time.time() < t_end:
    batch = []
    for _ in range(int(rng.integers(3, 14))):
        C = int(rng.choice([1, 2, 4, 4, 4, 10]))
        N = int(rng.choice([1, 2, 7, 100, 1000, 1000, 2500, 4097, 10000, 30000, 60000]))
        P = int(rng.integers(1, 40 if N <= 10000 else 6))
        kind = rng.integers(0, 6)
        x = rng.normal(size=(P, C, N))
        if kind == 1: x = np.round(x, 1)
        if kind == 2: x = np.cumsum(x, axis=2) * 0.02 + rng.normal(size=(P, C, N))
        if kind == 3: x[0] = 3.25
        if kind == 4: x = x.astype(np.float32)
        if kind == 5: x = np.sign(x) * (1.0 + (np.arange(P) % 3)[:, None, None])      # two-valued columns: rho exactly zero now and then (guard band)
        layout = "pcn" if rng.integers(0, 3) else "cnp"
        if layout == "cnp": x = np.ascontiguousarray(np.transpose(x, (1, 2, 0)))
        batch.append((x, layout))
    tens = [fast.upload(x, l) for x, l in batch]
    bufs = []
    for t in tens:
        if fast.inflight >= _ffi.MCR_MAX_INFLIGHT:
            fast.wait_one()
        bufs.append(fast.enqueue(t, min_chains=1))
    fast.wait()
    for (x, l), t, b in zip(batch, tens, bufs):
        ref = slow.summarize(x, l, min_chains=1)
        got = b.result()
        for k in KEYS:
            a, r = np.ascontiguousarray(got[k]), np.ascontiguousarray(ref[k])
            same = np.array_equal(a.view(np.int64), r.view(np.int64)) if a.dtype == np.float64 else np.array_equal(a, r)
            if not same:
                print("MISMATCH", k, x.shape, l, x.dtype); sys.exit(1)
        t.free()
        calls += 1; pd += x.size
    rounds += 1
print(f"soak ok: {rounds} rounds, {calls} calls, {pd / 1e6:.0f} M param-draws, pipelined == single-lane bit for bit")
