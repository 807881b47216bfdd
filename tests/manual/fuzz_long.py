#!/usr/bin/env python3
"""Long seeded differential fuzz of the C ABI against the CPU oracle (manual: minutes on the GPU box).

  python tests/manual/fuzz_long.py [iterations] [seed]
Shapes cross every internal boundary (tile 4096, bucket edges, segment 2048, 64-lane waves, u16 / u32 positions), data
kinds: continuous, heavily tied, constant parameters, random walks (long truncation lags), shifted chains, tiny spread
around a large offset; both layouts, f32 and f64, 1..32 quantiles.  Integer outputs and quantiles must be identical,
floating-point outputs within 1e-9."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "mcmc-db_amd")]
from mcmc_ref_hip import _ffi
from oracle import oracle

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1      # replay the stream, check this iteration alone (verbose)
rng = np.random.default_rng(seed)
ctx = _ffi.Context(0)
n_choices = [2, 3, 5, 31, 63, 64, 65, 127, 128, 129, 511, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097, 6000,
             8191, 8192, 8193, 12001, 16383, 16385, 20011]


def close(a, b, rel):
    a, b = float(a), float(b)
    if np.isnan(a) or np.isnan(b):
        return np.isnan(a) and np.isnan(b)
    if np.isinf(a) or np.isinf(b):
        return a == b
    return abs(a - b) <= rel * max(abs(a), abs(b), 1e-300)


bad = 0
for it in range(iters):
    C = int(rng.integers(1, 11))
    N = int(rng.choice(n_choices))
    if C * N > 200000:
        N = 200000 // C
    P = int(rng.integers(1, 9))
    kind = int(rng.integers(0, 7))
    x = rng.normal(size=(P, C, N)) * 10.0 ** rng.integers(-3, 4) + rng.normal() * 100.0
    if kind == 1:
        x = np.round(x, int(rng.integers(0, 2)))
    elif kind == 2:
        x[0] = 3.25
    elif kind == 3:
        x = np.cumsum(rng.normal(size=(P, C, N)), axis=2) * 0.05
    elif kind == 4 and C > 2:
        x[:, 0, :] += 5.0
    elif kind == 5:
        x = 3.0 + 1e-6 * rng.normal(size=(P, C, N))
    elif kind == 6:
        x[:, :, : N // 2] = np.round(x[:, :, : N // 2])
    layout, arr = "pcn", x
    if it % 5 == 0:
        arr = np.ascontiguousarray(np.transpose(x, (1, 2, 0))); layout = "cnp"
    if it % 7 == 0:
        arr = arr.astype(np.float32)
    nq = int(rng.integers(1, 33))
    qs = np.sort(rng.uniform(size=nq)); qs[0] = 0.0 if it % 3 == 0 else qs[0]
    mc = 1 if C < 2 else 2
    if only >= 0 and it != only:
        continue
    try:
        got = ctx.summarize(arr, layout, min_chains=min(mc, C), quantiles=qs)
        exp = oracle.summarize(arr, layout, min_chains=min(mc, C), quantiles=qs)
    except Exception as exc:  # noqa: BLE001
        print(f"it {it}: C={C} N={N} P={P} kind={kind}: {type(exc).__name__}: {exc}")
        bad += 1
        continue
    what = f"it {it}: C={C} N={N} P={P} kind={kind} {layout} {arr.dtype} nq={nq}"
    ok = np.array_equal(got["q"], exp["q"])
    for p in range(P):
        ok &= close(got["median"][p], exp["median"][p], 0.0)
        ok &= int(got["lag_bulk"][p]) == int(exp["lag_bulk"][p]) and int(got["lag_tail"][p]) == int(exp["lag_tail"][p])
        # (a constant column whose value is no binary fraction: sequential sums leave the reference a std of O(eps |mean|)
        #  and a mean one ulp off; the tile-wise two-pass form gives exactly 0 and the value itself)
        ok &= abs(float(got["std"][p]) - float(exp["std"][p])) <= 1e-9 * float(exp["std"][p]) + 64 * 2.2e-16 * abs(float(exp["mean"][p]))
        ok &= abs(float(got["mean"][p]) - float(exp["mean"][p])) <= 1e-9 * max(float(exp["std"][p]), abs(float(exp["mean"][p])) * 1e-6, 1e-300)
        for k in ("rhat", "rhat_bulk", "rhat_tail", "ess_bulk", "ess_tail"):
            ok &= close(got[k][p], exp[k][p], 1e-9)
    if not ok:
        bad += 1
        print("MISMATCH", what)
        if only >= 0:
            print(" q equal:", np.array_equal(got["q"], exp["q"]))
            for p in range(P):
                for k in ("mean", "std", "median", "rhat", "rhat_bulk", "rhat_tail", "ess_bulk", "ess_tail", "lag_bulk", "lag_tail"):
                    print(f"  p{p} {k}: got {got[k][p]!r} exp {exp[k][p]!r}")
    if it % 50 == 49:
        print(f"{it + 1} cases, {bad} bad", flush=True)
print(f"fuzz done: {iters} cases, {bad} bad")
sys.exit(1 if bad else 0)
