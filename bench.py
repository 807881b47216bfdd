#!/usr/bin/env python3
"""bench.py -- validated param-draws/s of the MI355X statistics hot path.

One "step" = one full pass of the hot path (pooled mean/std, q5/q50/q95, rank-normalised split
R-hat, ESS bulk, ESS tail, truncation lags) over one synthetic model resident in HBM.

  python bench.py                                   # N=1, BASELINE config 1 (4 x 10000 x 100 f64)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W      # one rank per GPU, weak scaling

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` (dominant
kernel, HIP-event timed inside this process) and `cpu_baseline` (the C oracle on this host's cores).
No torch anywhere: for N > 1 the ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment
the launcher sets and talk through the library's RCCL communicator (mcmc_ref_hip.shard.Communicator:
barrier, MAX-over-ranks clock, and the one all-gather of the per-parameter summary records).

At N = 1 the default run also carries `configs` (every other single-GPU BASELINE configuration, driver-timed in the same
process: the 57 committed corpus files end to end, the corpus shapes device-resident, the full pipeline on the 16 GB
stress tensor), `d_sweep` (P = 10 / 100 / 1000 at 4 x 10000), `sync_call_us` / `host_call_us` (ONE synchronous call, which
is what reference.compare pays) -- each with its own `validated` flag; --no-extras skips them.

Workloads (--workload):
  c1       BASELINE config 1, the headline: every rank its own 4 x 10000 x 100 f64 model (weak scaling)
  c1split  ONE such model, its P axis cut into contiguous blocks over the ranks (strong scaling)
  corpus   BASELINE configs 2/3: the 57 packaged model shapes, LPT-sharded over ranks (strong scaling)
  stress   BASELINE config 4: 4 x 100000 x 10000 f32 (16 GB) generated on the device, P-split over ranks
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path[:0] = [str(ROOT), str(ROOT / "mcmc-db_amd")]

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
TRAFFIC_FILE = ROOT / "profiles" / "pmc_traffic.json"      # rocprofv3 --pmc passes of tools/collect_profiles.sh

# Algorithmic (compulsory) HBM bytes per param-draw and launch of each kernel, f64 input
# (DESIGN.md "Kernels and rooflines").  Intermediate traffic a kernel causes
# beyond these is implementation overhead and shows up as a lower fraction.
ALG_BYTES_PER_PD = {
    "k_tile_sort": 8.0,      # the one compulsory read of the draw tensor (its sorted tiles are intermediates)
    "k_bucket_merge": 4.0,   # the 4-byte bulk rank code it must produce per draw (sorted order is an intermediate)
    "k_fold_merge": 4.0,     # the 4-byte folded rank code it must produce per draw
    "k_acov_seg": 8.0,       # reads the bulk and the folded rank code once each
    "k_merge": 8.0,          # (long-array path, per pass) re-reads the keys it merges
    "k_rank_z": 4.0,         # (long-array path) the rank code it must produce
    "k_ingest": 16.0,        # read + write of the layout change
    "k_moments": 8.0,
}


# profile_get() name -> name in profiles/pmc_traffic.json where they differ
TRAFFIC_NAMES: dict[str, str] = {}


def cpu_model() -> str:
    try:
        for ln in Path("/proc/cpuinfo").read_text().splitlines():
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def alg_bytes(name: str, es: int) -> float:
    """Algorithmic bytes per param-draw of one launch of kernel `name` for element size `es`."""
    if name in ("k_tile_sort", "k_moments"):
        return float(es)                  # the compulsory read scales with the input dtype
    if name == "k_ingest":
        return float(es) + 8.0
    return ALG_BYTES_PER_PD.get(name, 0.0)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--windows", type=int, default=0,
                    help="number of K-step timed windows (ms_per_step is the MEDIAN window); 0 = repeat until they cover "
                         "0.3 s of timed work, at least 5 and at most 64 (a 5 ms region is fragile: clock ramp-up, fill "
                         "and drain of the rolling window)")
    ap.add_argument("--workload", choices=["c1", "c1split", "corpus", "stress"], default="c1")
    ap.add_argument("--chains", type=int, default=4)
    ap.add_argument("--draws", type=int, default=10000)
    ap.add_argument("--params", type=int, default=100)
    ap.add_argument("--layout", choices=["pcn", "cnp"], default="pcn",
                    help="pcn = Arrow column layout [P][C][N]; cnp = Draws.to_numpy layout [C][N][P]")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--inflight", type=int, default=8, help="steps enqueued before a host wait (1..8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-validate", action="store_true")
    ap.add_argument("--no-moments", action="store_true",
                    help="skip the streaming-moments HBM roofline leg (16 GB f32 tensor generated on the device)")
    ap.add_argument("--no-probe", action="store_true", help="skip the measured-HBM-peak probe")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the other single-GPU configurations of the default line (configs / d_sweep / call latencies)")
    return ap.parse_args()


def validate(got: dict, exp: dict) -> tuple[bool, float]:
    """Parity gate of the run that was timed: integers exact, floats <= 1e-6 relative."""
    worst = 0.0
    ok = np.array_equal(got["lag_bulk"], exp["lag_bulk"]) and np.array_equal(got["lag_tail"], exp["lag_tail"])
    ok = ok and np.array_equal(got["q"], exp["q"]) and np.array_equal(got["median"], exp["median"])
    for k in ("std", "rhat", "rhat_bulk", "rhat_tail", "ess_bulk", "ess_tail"):
        a, b = got[k], exp[k]
        fin = np.isfinite(b)
        ok = ok and np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~fin & ~np.isnan(b)], b[~fin & ~np.isnan(b)])
        if fin.any():
            worst = max(worst, float(np.max(np.abs(a[fin] - b[fin]) / np.maximum(np.abs(b[fin]), 1e-300))))
    err_mean = np.abs(got["mean"] - exp["mean"]) / (np.abs(exp["mean"]) + exp["std"])
    worst = max(worst, float(np.max(err_mean)))
    return bool(ok and worst <= 1e-6), worst


class Ranks:
    """The ranks of this launch: RCCL communicator of the library for N > 1 (or MCR_BENCH_FORCE_DIST=1, which runs
    every collective with a world of one so that the RCCL calls are exercised on a 1-GPU box)."""

    def __init__(self, ctx, world: int, rank: int):
        self.ctx, self.world, self.rank, self.comm = ctx, world, rank, None
        if world > 1 or os.environ.get("MCR_BENCH_FORCE_DIST") == "1":
            from mcmc_ref_hip import shard
            self.comm = shard.Communicator(ctx, world, rank)

    def barrier(self):
        """Drains this rank's device work, then (N > 1) meets the other ranks."""
        self.ctx.sync()
        if self.comm is not None:
            self.comm.barrier()

    def max(self, x: float) -> float:
        return float(self.comm.all_reduce([x], "max")[0]) if self.comm is not None else x

    def gather(self, x: float) -> list[float]:
        """x of every rank, in rank order (the per-rank clocks of the rank-0 line: a slow rank shows in SCALE_rNN.json)."""
        if self.comm is None:
            return [float(x)]
        return [float(v) for v in np.asarray(self.comm.all_gather(np.array([x], dtype=np.float64))).reshape(-1)]

    def all_true(self, flag: bool) -> bool:
        return bool(self.comm.all_reduce([1.0 if flag else 0.0], "min")[0] > 0.5) if self.comm is not None else flag

    def close(self):
        if self.comm is not None:
            self.comm.close()


MIN_WINDOWS, MAX_WINDOWS, MIN_TIMED_SECONDS = 5, 64, 0.3


def timed_windows(a, ranks: Ranks, run) -> tuple[float, list[float], object]:
    """W warm-up steps, then timed regions of EXACTLY K steps each, every one bracketed by a barrier + device sync on
    both sides and clocked as the MAX over ranks.  `--windows N` fixes their number; the default (0) repeats them until
    they cover MIN_TIMED_SECONDS of timed work (at least MIN_WINDOWS, at most MAX_WINDOWS): a 20-step window is 3.6 ms,
    and the box needs tens of milliseconds of load before its clocks settle (the first windows of a short run are
    10 - 20 % slower; all of them are printed).  Returns (median seconds per window, all windows, the last result
    handle of the last window)."""
    run(a.warmup)
    secs, last = [], None
    ranks.own_secs = []                                           # this rank's own clock per window (before the MAX)
    while True:
        ranks.barrier()
        t0 = time.perf_counter()
        last = run(a.steps)
        ranks.ctx.sync()
        ranks.own_secs.append(time.perf_counter() - t0)
        ranks.barrier()
        secs.append(ranks.max(time.perf_counter() - t0))          # the same value on every rank
        if a.windows > 0:
            if len(secs) >= a.windows:
                break
        elif len(secs) >= MAX_WINDOWS or (len(secs) >= MIN_WINDOWS and sum(secs) >= MIN_TIMED_SECONDS):
            break
    return statistics.median(secs), secs, last


def timing_note(a, windows: list[float]) -> str:
    how = (f"{len(windows)} windows" if a.windows > 0 else
           f"{len(windows)} windows (repeated until {MIN_TIMED_SECONDS} s of timed work, {MIN_WINDOWS}..{MAX_WINDOWS})")
    return f"median of {how} of {a.steps} steps, each bracketed by barrier + device sync, MAX over ranks"


def hbm_probe(ctx, a) -> dict | None:
    if a.no_probe:
        return None
    try:
        return ctx.hbm_probe(4 << 30, 5)
    except Exception as exc:  # noqa: BLE001 - the probe is context for the roofline, never a reason to fail the bench
        return {"error": str(exc)}


def corpus_bench(a, ctx, ranks: Ranks):
    """BASELINE configs 2/3: one step = one pass over the whole 57-model corpus (shapes of the packaged
    reference set, synthetic draws), models LPT-sharded over ranks, same-shape models batched into one
    kernel pipeline, one RCCL all-gather of 128-byte records at the end."""
    from mcmc_ref_hip import _ffi, corpus, shard
    world, rank = ranks.world, ranks.rank
    models = corpus.synthetic_corpus(seed=4711)
    costs = [float(np.prod(m.shape)) for _, m in models]
    mine = shard.plan_shards(costs, world)[rank]
    groups = {}
    for i in mine:
        arr = models[i][1]
        groups.setdefault((arr.shape[1], arr.shape[2]), []).append(i)
    tensors = []
    for (C, N), members in groups.items():
        big = np.concatenate([models[i][1] for i in members], axis=0)
        tensors.append((members, big, ctx.upload(big, "pcn")))
    total_pd = int(sum(costs))

    def run(steps):
        """Rolling window over (step, tensor group): the device never drains between corpus passes."""
        last = None
        for _ in range(steps):
            cur = []
            for _, _, t in tensors:
                if ctx.inflight >= _ffi.MCR_MAX_INFLIGHT:
                    ctx.wait_one()
                cur.append(ctx.enqueue(t))
            last = cur
        ctx.wait()
        return last

    elapsed, windows, last = timed_windows(a, ranks, run)
    recs = []
    valid = True
    from oracle import oracle as orc
    for (members, big, t), bufs in zip(tensors, last or []):
        r = bufs.result()
        p0 = 0
        for i in members:
            P = models[i][1].shape[0]
            part = {k: (v[p0:p0 + P] if k != "q_lo" else v) for k, v in r.items()}
            recs.append(shard.pack_records(part, i, big.shape[1], big.shape[2]))
            p0 += P
        if not a.no_validate:
            ok, _ = validate(r, orc.summarize(big, "pcn"))
            valid = valid and ok
        t.free()
    local = np.concatenate(recs) if recs else np.empty((0, shard.RECORD_DOUBLES))
    allrec = shard.gather_records(local, ranks.comm)
    valid = ranks.all_true(valid and allrec.shape[0] == 460)
    per_rank_ms = [x / a.steps * 1e3 for x in ranks.gather(statistics.median(ranks.own_secs))]
    if rank == 0:
        value = a.steps * total_pd / elapsed
        print(json.dumps({
            "metric": "validated param-draws/sec", "value": value if valid else 0.0, "unit": "param-draws/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "packaged mcmc-ref-data corpus shapes: 57 models, 460 params, 4.6 M param-draws "
                                   "(BASELINE configs 2/3), synthetic draws", "layout": "pcn",
                       "sharding": f"whole models, greedy LPT over {world} rank(s), same-shape models batched, "
                                   "one RCCL all-gather of 128-byte records"},
            "validated": valid, "pipeline_alg_GBps": value * 8 / 1e9,
            "ms_per_step_per_rank": [round(x, 5) for x in per_rank_ms],
            "ms_per_step_windows": [s / a.steps * 1e3 for s in windows]}), flush=True)
    return 0 if valid else 1


def split_bench(a, ctx, ranks: Ranks):
    """ONE model, its parameter axis cut into contiguous blocks over the ranks (strong scaling).
    c1split: the C1 model (host-generated, validated against the oracle on every rank's block).
    stress : BASELINE config 4, 4 x 100000 x 10000 f32 = 16 GB, generated on the device block by block; validated by
             size-independent properties over ALL parameters of the block + the oracle on 16 parameters."""
    from mcmc_ref_hip import shard, synth
    world, rank = ranks.world, ranks.rank
    stress = a.workload == "stress"
    C, N, P = (4, 100000, 10000) if stress else (a.chains, a.draws, a.params)
    dt = np.float32 if stress else (np.float64 if a.dtype == "f64" else np.float32)
    p0, p1 = shard.param_block(P, world, rank)
    pb = p1 - p0
    host = None
    if stress:
        t = ctx.alloc_tensor(C, N, max(pb, 1), dt)
        if pb:
            ctx.fill_synthetic(t, 4711, p0=p0)
    else:
        host = synth.c1_model(C, N, P, seed=4711, params=range(p0, p1), dtype=dt)
        t = ctx.upload(host if pb else np.zeros((1, C, N), dtype=dt), "pcn")
    def run(steps):
        last = None
        for _ in range(steps):
            if ctx.inflight >= (1 if stress else 8):
                ctx.wait_one()
            last = ctx.enqueue(t) if pb else None
        ctx.wait()
        return last

    elapsed, windows, last = timed_windows(a, ranks, run)
    valid, worst, checked = True, 0.0, "nothing"
    recs = np.empty((0, shard.RECORD_DOUBLES))
    if pb:
        got = last.result()
        recs = shard.pack_records(got, 0, C, N, param0=p0)
        if not a.no_validate:
            from oracle import oracle as orc
            if stress:
                valid = stress_properties(got, np.arange(p0, p1), C * N)
                ok, worst, how = stress_oracle_check(ctx, t, got, p0, pb, C, N, synth, orc)
                valid = valid and ok
                checked = f"properties over all {pb} parameters of the block + {how}"
            else:
                valid, worst = validate(got, orc.summarize(host, "pcn"))
                checked = f"oracle on all {pb} parameters of the block"
    allrec = shard.gather_records(recs, ranks.comm)
    valid = ranks.all_true(valid and allrec.shape[0] == P and
                           np.array_equal(allrec[:, shard.RECORD_FIELDS.index("param_idx")], np.arange(P)))
    per_rank_ms = [x / a.steps * 1e3 for x in ranks.gather(statistics.median(ranks.own_secs))]
    if rank == 0:
        value = a.steps * C * N * P / elapsed
        es = np.dtype(dt).itemsize
        print(json.dumps({
            "metric": "validated param-draws/sec", "value": value if valid else 0.0, "unit": "param-draws/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32" if es == 4 else "f64",
            "data": "synthetic",
            "config": {"workload": (f"stress: {C}x{N}x{P} f32 (16 GB) generated on the device (BASELINE config 4)" if stress else
                                    f"one {C}x{N}x{P} model (C1)") + f", parameter axis split over {world} rank(s)",
                       "layout": "pcn", "sharding": "contiguous parameter blocks, no halo, one RCCL all-gather of records"},
            "validated": valid, "max_rel_err": worst, "validation": checked,
            "pipeline_alg_GBps": value * es / 1e9, "pipeline_frac_of_hbm": value * es / 1e9 / HBM_PEAK_GBS / world,
            "ms_per_step_per_rank": [round(x, 5) for x in per_rank_ms],
            "ms_per_step_windows": [s / a.steps * 1e3 for s in windows]}), flush=True)
    t.free()
    return 0 if valid else 1


# =====================================================================================================================
# The other single-GPU BASELINE configurations, inside the default run.  Rank 0, N = 1 only.  Every
# leg times its own work between device syncs with the host clock, validates the results of the calls it timed, and
# reports `validated`; none of them touches `value`.
# =====================================================================================================================
def _median_ms(fn, reps: int, sync) -> tuple[float, list[float]]:
    ts = []
    for _ in range(reps):
        sync()
        t0 = time.perf_counter()
        fn()
        sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts), ts


def leg_call_latency(ctx, t, host, got_pipelined: dict) -> dict:
    """ONE synchronous call -- what reference.compare(model, actual) pays (src/mcmc_ref/reference.py:107-122) -- on the
    device-resident tensor (`sync_call_us`) and from host memory, upload included (`host_call_us`, PCIe-inclusive)."""
    keys = ("mean", "std", "q", "rhat", "ess_bulk", "ess_tail", "lag_bulk", "lag_tail")
    res = {}
    for _ in range(3):
        ctx.summarize(t)
    dev_ms, dev_all = _median_ms(lambda: res.__setitem__("d", ctx.summarize(t)), 40, ctx.sync)
    for _ in range(2):
        ctx.summarize(host, "pcn")
    host_ms, host_all = _median_ms(lambda: res.__setitem__("h", ctx.summarize(host, "pcn")), 15, ctx.sync)
    same = all(np.array_equal(res["d"][k], got_pipelined[k], equal_nan=True) and
               np.array_equal(res["h"][k], got_pipelined[k], equal_nan=True) for k in keys)
    pd = int(np.prod(host.shape))
    return {"sync_call_us": round(dev_ms * 1e3, 1), "sync_call_min_us": round(min(dev_all) * 1e3, 1),
            "host_call_us": round(host_ms * 1e3, 1), "host_call_min_us": round(min(host_all) * 1e3, 1),
            "sync_call_param_draws_per_s": pd / (dev_ms * 1e-3), "host_call_param_draws_per_s": pd / (host_ms * 1e-3),
            "host_bytes_uploaded": int(host.nbytes),
            "what": "median of 40 / 15 synchronous mcr_summarize_dev / mcr_summarize calls on the C1 tensor, one at a time "
                    "(no pipelining); results bit-equal to the pipelined calls that `value` timed",
            "validated": bool(same)}


def leg_d_sweep(ctx, synth, orc, C: int, N: int) -> list[dict]:
    """north_star: throughput on synthetic (4 chains x 10 000 draws x D params).  Pipelined (8 calls in flight) and one
    synchronous call, D = 10 / 100 / 1000, each validated against the oracle on the calls that were timed."""
    out = []
    for P in (10, 100, 1000):
        host = synth.c1_model(C, N, P, seed=4711)
        t = ctx.upload(host, "pcn")
        steps = {10: 400, 100: 200, 1000: 40}[P]

        ring = []                                   # result buffers of delivered calls, reused (as the headline's loop does)

        def run(k):
            last = None
            for _ in range(k):
                if ctx.inflight >= 8:
                    ring.append(ctx.wait_one())
                last = ctx.enqueue(t, bufs=ring.pop() if ring else None)
            ctx.wait()
            return last
        run(max(steps // 10, 4))
        ws = []
        for _ in range(5):
            ctx.sync(); t0 = time.perf_counter(); last = run(steps); ctx.sync()
            ws.append((time.perf_counter() - t0) / steps * 1e3)
        ms = statistics.median(ws)
        sync_ms, _ = _median_ms(lambda: ctx.summarize(t), 15, ctx.sync)
        ok, worst = validate(last.result(), orc.summarize_mt(host, "pcn"))
        t.free()
        out.append({"params": P, "shape": f"{C}x{N}x{P} f64", "ms_per_call_pipelined": round(ms, 5),
                    "param_draws_per_s": C * N * P / (ms * 1e-3), "sync_call_us": round(sync_ms * 1e3, 1),
                    "us_per_100_params": round(ms * 1e3 * 100 / P, 2), "validated": ok, "max_rel_err": worst})
    return out


def leg_sticky(ctx, orc, C: int, N: int, P: int) -> dict:
    """The C1 shape with STICKY chains: every parameter an AR(1) walk with phi = 0.99 (integrated autocorrelation time ~ 200
    draws, as hierarchical posteriors have), so that most (parameter, kind) pairs are still undecided at lag 256 and go
    through tier 3 -- the synthetic C1 model of the headline stops at phi = 0.95 and never does.  Pipelined and lone call,
    oracle on every parameter."""
    from scipy.signal import lfilter
    phi = 0.99
    rng = np.random.default_rng(99)
    x = lfilter([1.0], [1.0, -phi], rng.normal(size=(P, C, N)) * np.sqrt(1 - phi * phi), axis=2)
    x += np.arange(P)[:, None, None]
    t = ctx.upload(x, "pcn")

    def run(k):
        last = None
        for _ in range(k):
            if ctx.inflight >= 8:
                ctx.wait_one()
            last = ctx.enqueue(t)
        ctx.wait()
        return last
    run(10)
    ws = []
    for _ in range(5):
        ctx.sync(); t0 = time.perf_counter(); last = run(40); ctx.sync()
        ws.append((time.perf_counter() - t0) / 40 * 1e3)
    ms = statistics.median(ws)
    sync_ms, _ = _median_ms(lambda: ctx.summarize(t), 10, ctx.sync)
    got = last.result()
    ok, worst = validate(got, orc.summarize_mt(x, "pcn"))
    t.free()
    lags = np.concatenate([got["lag_bulk"], got["lag_tail"]])
    return {"workload": f"{C}x{N}x{P} f64, every parameter AR(1) with phi = {phi} (sticky chains: tiers 2 and 3 of the ESS lags)",
            "pairs_beyond_lag_255": int((lags > 255).sum()), "pairs": int(lags.size), "max_truncation_lag": int(lags.max()),
            "ms_per_call_pipelined": round(ms, 4), "param_draws_per_s": C * N * P / (ms * 1e-3), "sync_call_us": round(sync_ms * 1e3, 1),
            "validated": ok, "max_rel_err": worst, "validation": "oracle on all parameters (integer truncation lags exact)"}


def leg_corpus_device(ctx, _ffi, orc, steps: int = 20) -> dict:
    """BASELINE config 2, device-resident: the 57 packaged model shapes (synthetic draws), same-shape models batched."""
    from mcmc_ref_hip import corpus
    models = corpus.synthetic_corpus(seed=4711)
    groups = {}
    for i, (_, arr) in enumerate(models):
        groups.setdefault((arr.shape[1], arr.shape[2]), []).append(i)
    tensors = []
    for members in groups.values():
        big = np.concatenate([models[i][1] for i in members], axis=0)
        tensors.append((big, ctx.upload(big, "pcn")))
    total_pd = int(sum(np.prod(m.shape) for _, m in models))

    def run(k):
        last = None
        for _ in range(k):
            cur = []
            for _, t in tensors:
                if ctx.inflight >= _ffi.MCR_MAX_INFLIGHT:
                    ctx.wait_one()
                cur.append(ctx.enqueue(t))
            last = cur
        ctx.wait()
        return last
    run(5)
    ws = []
    for _ in range(7):
        ctx.sync(); t0 = time.perf_counter(); last = run(steps); ctx.sync()
        ws.append((time.perf_counter() - t0) / steps * 1e3)
    ms = statistics.median(ws)
    ok = True
    for (big, t), b in zip(tensors, last):
        v, _ = validate(b.result(), orc.summarize_mt(big, "pcn"))
        ok = ok and v
        t.free()
    return {"workload": "57 packaged model shapes, 460 params, 4.6 M param-draws, synthetic draws resident in HBM "
                        "(BASELINE config 2; = --workload corpus)", "ms_per_corpus_pass": round(ms, 5),
            "param_draws_per_s": total_pd / (ms * 1e-3), "windows_ms": [round(w, 5) for w in ws], "steps_per_window": steps,
            "validated": bool(ok), "validation": "oracle on all 460 parameters of the last timed pass"}


def leg_corpus_files(ctx) -> dict | None:
    """BASELINE config 2 END TO END: the 57 committed draws files of the reference's packaged corpus (tests/golden/corpus,
    data files, 42.7 MB) -> mcr_summarize_files (mmap, footer parse, upload, Snappy + page decode on the GPU, statistics)
    -> the 1 380 goldens the reference packaged in its meta.json files.  File-bytes-to-statistics, page cache warm."""
    from mcmc_ref_hip import parquet
    root = ROOT / "tests" / "golden" / "corpus"
    paths = sorted((root / "draws").glob("*.draws.parquet"))
    if len(paths) != 57:
        return None
    res = parquet.summarize_files(ctx, paths, min_chains=4)
    ms, all_ms = _median_ms(lambda: parquet.summarize_files(ctx, paths, min_chains=4), 9, ctx.sync)
    runs = []

    def c_call():
        ph = {}
        parquet._summarize_paths(ctx, [str(p) for p in paths], 4, [0.05, 0.5, 0.95], True, ph)
        runs.append(ph)
    c_ms, c_all = _median_ms(c_call, 9, ctx.sync)
    phases = {k: round(statistics.median(r[k] for r in runs), 3) for k in runs[0]}
    res = parquet.summarize_files(ctx, paths, min_chains=4)
    n_vals, worst, total_pd = 0, 0.0, 0
    for path, got in zip(paths, res):
        meta = json.loads((root / "meta" / (path.name[: -len(".draws.parquet")] + ".meta.json")).read_text())
        total_pd += meta["n_chains"] * meta["n_draws_per_chain"] * len(meta["parameters"])
        for prm, gold in meta["diagnostics"].items():
            for k in ("rhat", "ess_bulk", "ess_tail"):
                worst = max(worst, abs(got[prm][k] - gold[k]) / max(abs(gold[k]), 1e-300))
                n_vals += 1
    return {"workload": f"the {len(paths)} REAL packaged draws files ({sum(p.stat().st_size for p in paths)} file bytes, SNAPPY) "
                        "-> native Parquet ingest -> statistics, one mcr_summarize_files call, page cache warm "
                        "(BASELINE config 2 end to end)",
            "ms_end_to_end": round(ms, 3), "ms_min": round(min(all_ms), 3), "param_draws": total_pd,
            "param_draws_per_s": total_pd / (ms * 1e-3),
            "ms_c_call_only": round(c_ms, 3), "phases_ms": phases,
            "goldens_checked": n_vals, "max_rel_err_vs_packaged_goldens": worst,
            "validated": bool(n_vals >= 1300 and worst <= 1e-6),
            "validation": "rhat / ess_bulk / ess_tail of every parameter against the reference's own meta.json goldens (<= 1e-6)"}


def stress_properties(got: dict, p: np.ndarray, M: int) -> bool:
    """Size-independent properties of the stress tensor's statistics (generator: mu_p = p, sigma_p = 10**((p mod 7) - 3), iid)."""
    sig = 10.0 ** ((p % 7) - 3)
    fine = sig >= 64.0 * np.spacing(np.maximum(p, 1).astype(np.float32)).astype(np.float64)   # sigma well above the f32 grid at p
    return bool(np.all(np.abs(got["mean"] - p)[fine] < 0.02 * sig[fine]) and np.all(np.abs(got["std"] / sig - 1)[fine] < 0.02)
                and np.all(got["std"] > 0) and np.all(np.abs(got["mean"] - p) < 0.02 * sig + 1e-3)
                and np.all(got["q"][:, 0] <= got["q"][:, 1]) and np.all(got["q"][:, 1] <= got["q"][:, 2])
                and np.array_equal(got["q"][:, 1], got["median"])
                and np.all((got["ess_bulk"] > 0) & (got["ess_bulk"] <= M)) and np.all((got["ess_tail"] > 0) & (got["ess_tail"] <= M))
                and np.all(got["rhat"] >= got["rhat_bulk"]) and np.all(got["rhat"] < 1.01)
                and np.all(got["lag_bulk"] >= 0) and np.all(got["lag_tail"] >= 0))


def stress_oracle_check(ctx, t, got: dict, p0: int, pb: int, C: int, N: int, synth, orc) -> tuple[bool, float, str]:
    """The oracle on 128 parameters of the block: both sides of every workspace-chunk edge, every scale of the generator,
    the all-ties parameters at the f32 grid (synth.stress_check_sample), on all host threads."""
    sel = synth.stress_check_sample(pb, ctx.params_per_chunk(t), 128)
    sub = np.stack([ctx_row(ctx, t, int(i), C * N) for i in sel])      # rows of the block, not the block
    ok, worst = validate({k: (v[sel] if k != "q_lo" else v) for k, v in got.items()},
                         orc.summarize_mt(sub.reshape(len(sel), C, N), "pcn"))
    return ok, worst, f"oracle on {len(sel)} (chunk edges of {ctx.params_per_chunk(t)}-parameter chunks, all 7 scales, all-ties)"


def leg_stress(_ffi, synth, orc, device: int, peak_measured, pipeline: bool = True) -> tuple[dict | None, dict | None]:
    """BASELINE config 4 on its own single-lane context: 4 x 100000 x 10000 f32 = 16 GB generated on the device; the
    streaming-moments roofline (one HBM pass) and the FULL pipeline over all 10 000 parameters (chunked through the
    workspace), validated by properties over every parameter + the oracle on 128."""
    os.environ["MCR_LANES"] = "1"
    ctx = _ffi.Context(device)
    os.environ.pop("MCR_LANES", None)
    mc, mn, mp = 4, 100000, 10000
    try:
        try:
            big = ctx.alloc_tensor(mc, mn, mp, np.float32)
        except _ffi.McrError:                      # a smaller device: quarter-size tensor
            mp = 2500
            big = ctx.alloc_tensor(mc, mn, mp, np.float32)
        ctx.fill_synthetic(big, 4711)
        ctx.moments(big)
        ctx.profile(True)
        ctx.profile_reset()
        for _ in range(5):
            mm, ms = ctx.moments(big)
        pm = ctx.profile_get()["k_moments"]
        ctx.profile(False)
        kms = pm["total_ms"] / pm["launches"]
        sig = 10.0 ** ((np.arange(mp) % 7) - 3)
        ok_m = bool(np.max(np.abs(mm - np.arange(mp)) / sig) < 0.05 and np.max(np.abs(ms / sig - 1)) < 0.05)
        gbs = mc * mn * mp * 4 / (kms * 1e-3) / 1e9
        moments = {"kernel": "k_moments_rows<float>", "workload": f"{mc}x{mn}x{mp} f32 ({mc * mn * mp * 4 / 1e9:.0f} GB) synthetic, on-device",
                   "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "peak_measured": peak_measured, "unit": "GB/s",
                   "frac": gbs / HBM_PEAK_GBS, "frac_of_measured": (gbs / peak_measured) if peak_measured else None,
                   "avg_launch_us": kms * 1e3, "param_draws_per_s": mc * mn * mp / (kms * 1e-3), "sane": ok_m,
                   "traffic": None, "traffic_source": None}
        tr = traffic_table().get("moments-4x100000x2500-f32")
        if tr is not None and mp == 2500:
            moments["traffic"], moments["traffic_source"] = tr.get("k_moments"), str(TRAFFIC_FILE.relative_to(ROOT))
        if not pipeline:
            big.free()
            return moments, None
        # ---- the full pipeline on the same tensor ----
        ctx.summarize(big)                                     # warm-up: workspace allocation, z table
        ts = []
        for _ in range(3):
            ctx.sync(); t0 = time.perf_counter(); got = ctx.summarize(big); ctx.sync()
            ts.append(time.perf_counter() - t0)
        sec = statistics.median(ts)
        ok = stress_properties(got, np.arange(mp), mc * mn)
        ok2, worst, how = stress_oracle_check(ctx, big, got, 0, mp, mc, mn, synth, orc)
        big.free()
        pd = mc * mn * mp
        stress = {"workload": f"stress: {mc}x{mn}x{mp} f32 ({pd * 4 / 1e9:.0f} GB) generated on the device, ALL {mp} parameters "
                              "through the full pipeline, chunked through the workspace (BASELINE config 4)",
                  "ms_per_step": round(sec * 1e3, 2), "steps_ms": [round(x * 1e3, 2) for x in ts], "param_draws_per_s": pd / sec,
                  "alg_GBps": pd * 4 / sec / 1e9, "frac": pd * 4 / sec / 1e9 / HBM_PEAK_GBS,
                  "frac_of_measured": (pd * 4 / sec / 1e9 / peak_measured) if peak_measured else None,
                  "validated": bool(ok and ok2), "max_rel_err": worst,
                  "validation": f"properties over all {mp} parameters + {how}"}
        return moments, stress
    finally:
        ctx.close()


def ctx_row(ctx, t, i: int, M: int) -> np.ndarray:
    """Row i (M f32 draws) of a device tensor, without pulling the whole block to the host."""
    import ctypes as C
    out = np.empty(M, dtype=np.float32)
    ctx._check(ctx.lib.mcr_memcpy_d2h(ctx.handle, out.ctypes.data_as(C.c_void_p), C.c_void_p(t.buf.ptr.value + i * M * 4),
                                      out.nbytes))
    return out


def main():
    a = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    if a.workload == "stress":
        os.environ.setdefault("MCR_LANES", "1")     # one 0.3 s call at a time: a single lane, a single workspace
    from mcmc_ref_hip import _ffi, synth
    ndev = max(_ffi.load_library().mcr_device_count(), 1)
    ctx = _ffi.Context(local_rank % ndev)
    ranks = Ranks(ctx, world, rank)
    try:
        if a.workload == "corpus":
            return corpus_bench(a, ctx, ranks)
        if a.workload in ("c1split", "stress"):
            return split_bench(a, ctx, ranks)
        return c1_bench(a, ctx, ranks, _ffi, synth)
    finally:
        ranks.close()
        ctx.close()


def c1_bench(a, ctx, ranks: Ranks, _ffi, synth):
    world, rank = ranks.world, ranks.rank
    C, N, P = a.chains, a.draws, a.params
    dt = np.float64 if a.dtype == "f64" else np.float32
    # independent models shard across ranks: rank r validates its own model (weak scaling)
    x = synth.c1_model(C, N, P, seed=4711 + rank, dtype=dt)                 # [P][C][N]
    host = x if a.layout == "pcn" else np.ascontiguousarray(np.transpose(x, (1, 2, 0)))
    t = ctx.upload(host, a.layout)
    inflight = max(1, min(a.inflight, _ffi.MCR_MAX_INFLIGHT))

    ring = []                                   # result buffers of delivered calls, reused (no allocation per step)

    def run(steps):
        """Rolling window: at most `inflight` steps outstanding, the device never drains in between."""
        last = None
        for k in range(steps):
            if ctx.inflight >= inflight:
                ring.append(ctx.wait_one())
            last = ctx.enqueue(t, bufs=ring.pop() if ring else None)
        ctx.wait()
        return last

    # ---- timed region: windows of exactly K steps (rolling window of up to `inflight` calls over the lanes) ----
    elapsed, windows, last = timed_windows(a, ranks, run)
    # ---- same K steps again with a HIP event pair around every kernel launch (per-kernel roofline).
    #      With several lanes the kernels of consecutive steps overlap, which would inflate every
    #      per-kernel duration: this pass runs on a single-lane
    #      context (same device buffer), so the durations are those of each kernel running alone
    #      and agree with rocprofv3 --kernel-trace of `MCR_LANES=1 python bench.py`. ----
    os.environ["MCR_LANES"] = "1"
    ctx1 = _ffi.Context(ctx.device)
    t_one = _ffi.DeviceTensor(ctx1, t.buf, t.targs)
    for _ in range(2):
        ctx1.enqueue(t_one); ctx1.wait()
    ctx1.profile(True)
    ctx1.profile_reset()
    t1 = time.perf_counter()
    for k in range(a.steps):
        ctx1.enqueue(t_one)
        if (k + 1) % inflight == 0:
            ctx1.wait()
    ctx1.wait()
    elapsed_prof = time.perf_counter() - t1
    prof = ctx1.profile_get()
    ctx1.profile(False)
    ctx1.close()
    os.environ.pop("MCR_LANES", None)

    got = last.result()

    # final summary gather over RCCL: fixed-size (128 B) per-parameter records to every rank
    gathered_ok = True
    if ranks.comm is not None:
        from mcmc_ref_hip import shard
        mine = shard.pack_records(got, rank, C, N)
        allrec = shard.gather_records(mine, ranks.comm)
        sel = allrec[allrec[:, shard.RECORD_FIELDS.index("model_idx")] == rank]
        gathered_ok = allrec.shape[0] == world * P and np.array_equal(sel, mine, equal_nan=True)

    valid, worst = True, 0.0
    cpu = None
    if not a.no_validate or not a.no_cpu_baseline:
        from oracle import oracle as orc          # checker + cpu_baseline leg only
        t1 = time.perf_counter()
        exp = orc.summarize(host, a.layout)
        cpu_s = time.perf_counter() - t1
        if not a.no_validate:
            valid, worst = validate(got, exp)
        if rank == 0 and not a.no_cpu_baseline:
            # Bounded sample (~12 s of wall time): the same model, parameters split over T host threads
            # (ctypes releases the GIL; parameters are independent, so this is how a CPU deployment of the
            # restatement would run).  The single-thread rate of the validation pass is reported beside it.
            from concurrent.futures import ThreadPoolExecutor

            def split(T):
                cuts = [P * i // T for i in range(T + 1)]
                sl = [(slice(cuts[i], cuts[i + 1]),) if a.layout == "pcn" else (Ellipsis, slice(cuts[i], cuts[i + 1]))
                      for i in range(T)]
                return [np.ascontiguousarray(host[ix]) for ix in sl]

            def timed_passes(T, parts, budget_s, fn):
                passes, total_s = 0, 0.0
                with ThreadPoolExecutor(T) as pool:
                    while total_s < budget_s and passes < 64:
                        t1 = time.perf_counter()
                        list(pool.map(fn, parts))
                        total_s += time.perf_counter() - t1
                        passes += 1
                return passes, total_s

            ncpu = os.cpu_count() or 1
            # the restatement on 1 core and on all cores.  `value` = every visible core (one thread per
            # parameter at most: the parameters are the unit of work), `value_16` = 16 threads, the CPU share of a
            # one-GPU box on this pool, `value_1core` = the single-threaded validation pass above.
            T_all = max(1, min(ncpu, P))
            T = max(1, min(16, ncpu, P))
            parts = split(T)
            passes, total_s = timed_passes(T, parts, 8.0, lambda x: orc.summarize(x, a.layout))
            v16 = passes * C * N * P / total_s
            if T_all > T:
                pa, sa = timed_passes(T_all, split(T_all), 8.0, lambda x: orc.summarize(x, a.layout))
                v_all = pa * C * N * P / sa
            else:
                pa, sa, v_all = passes, total_s, v16
            cpu = {"value": v_all, "unit": "param-draws/s", "cores": T_all, "kind": "port",
                   "cpu_model": cpu_model(), "cores_visible": ncpu,
                   "sample": f"the same {C}x{N}x{P} {a.dtype} model through oracle/mcr_oracle.c, its parameters split over "
                             f"{T_all} threads ({pa} passes, {sa:.1f} s); value_16: over {T} threads ({passes} passes, "
                             f"{total_s:.1f} s); value_1core: one pass on one thread ({cpu_s:.1f} s)",
                   "value_16": v16, "cores_16": T, "value_1core": C * N * P / cpu_s}
            if a.layout == "pcn" and a.dtype == "f64" and C >= 2:
                # the "NumPy CPU path" of SURVEY 8(d): the same statistics vectorised with numpy / scipy
                # (argsort ranks, ndtri, FFT autocovariances), same thread split, ~6 s sample
                from oracle import numpy_path
                npass, nsec = timed_passes(T, parts, 6.0, numpy_path.summarize)
                cpu["numpy_path"] = {"value": npass * C * N * P / nsec, "unit": "param-draws/s", "cores": T,
                                     "sample": f"{npass} passes of oracle/numpy_path.py over {T} threads ({nsec:.1f} s)"}
    valid = ranks.all_true(valid and gathered_ok)

    probe = hbm_probe(ctx, a) if rank == 0 else None
    peak_measured = probe.get("read_GBps") if probe and "error" not in probe else None

    moments, extras = None, {}
    if rank == 0 and world == 1 and not a.no_extras and a.layout == "pcn" and a.dtype == "f64":
        from oracle import oracle as orc          # checker of every leg below
        extras["configs"] = {}
        lat = leg_call_latency(ctx, t, host, got)
        extras["sync_call_us"], extras["host_call_us"] = lat["sync_call_us"], lat["host_call_us"]
        extras["call_latency"] = lat
        extras["d_sweep"] = leg_d_sweep(ctx, synth, orc, C, N)
        extras["configs"]["sticky_c1"] = leg_sticky(ctx, orc, C, N, P)
        extras["configs"]["corpus_device"] = leg_corpus_device(ctx, _ffi, orc)
        cf = leg_corpus_files(ctx)
        if cf is not None:
            extras["configs"]["corpus_files"] = cf
    if rank == 0 and not a.no_moments:
        from oracle import oracle as orc
        # BASELINE config 4 itself: 4 x 100000 x 10000 f32 = 16 GB, generated on the device: the streaming-moments
        # roofline (one HBM pass per launch, 4 B per param-draw) and, with the extras, the full pipeline on it.
        moments, stress = leg_stress(_ffi, synth, orc, ctx.device, peak_measured, pipeline=bool(extras))
        if stress is not None:
            extras["configs"]["stress_pipeline"] = stress
    if extras:
        extras["extra_keys"] = sorted(["sync_call_us", "host_call_us", "call_latency", "d_sweep"] +
                                      [f"configs.{k}" for k in extras["configs"]])
        extras["extras_validated"] = bool(extras["call_latency"]["validated"] and all(d["validated"] for d in extras["d_sweep"])
                                          and all(c["validated"] for c in extras["configs"].values()))
    per_rank_ms = [x / a.steps * 1e3 for x in ranks.gather(statistics.median(ranks.own_secs))]

    if rank == 0:
        pd_step = C * N * P
        es = 8 if a.dtype == "f64" else 4
        kern = {}
        dom, dom_ms = None, -1.0
        ktraffic = traffic_table().get(f"{C}x{N}x{P}-{a.dtype}-{a.layout}", {})
        if N <= 16384 and "k_acov_long" in prof:       # chains up to 16 384 draws: the tier-3 launch behind this HIP-event label is k_tier3
            prof["k_tier3"] = prof.pop("k_acov_long")
        for name, r in prof.items():
            avg_ms = r["total_ms"] / max(r["launches"], 1)
            alg = alg_bytes(name, es)
            tb = ktraffic.get(TRAFFIC_NAMES.get(name, name), ktraffic.get(name))
            kern[name] = {"launches_per_step": r["launches"] / a.steps, "avg_us": round(avg_ms * 1e3, 2),
                          "alg_GBps": round(alg * pd_step / (avg_ms * 1e-3) / 1e9, 1) if avg_ms > 0 else None,
                          # FETCH_SIZE + WRITE_SIZE of the kernel / its time.  The bytes are per launch from the
                          # committed rocprofv3 --pmc passes of this configuration (traffic_source below), the time is this
                          # run's HIP-event average.
                          "traffic_bytes": tb,
                          "traffic_GBps": round(tb / (avg_ms * 1e-3) / 1e9, 1) if tb is not None and avg_ms > 0 else None}
            if r["total_ms"] > dom_ms and name in ALG_BYTES_PER_PD:
                dom, dom_ms = name, r["total_ms"]
        dom_avg_s = prof[dom]["total_ms"] / prof[dom]["launches"] * 1e-3
        dom_alg = alg_bytes(dom, es) * pd_step
        traffic = traffic_table().get(f"{C}x{N}x{P}-{a.dtype}-{a.layout}", {}).get(dom)
        achieved = dom_alg / dom_avg_s / 1e9
        value = world * a.steps * pd_step / elapsed
        out = {
            "metric": "validated param-draws/sec", "value": value if valid else 0.0, "unit": "param-draws/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype,
            "data": "synthetic",
            "config": {"workload": f"single posteriordb-shaped model, {C}x{N}x{P} synthetic draws per GPU "
                                   "(BASELINE config 1; AR(1) chains, ties and a shifted chain; SURVEY 8(d) C1)",
                       "layout": a.layout, "statistics": "mean,std,q5,q50,q95,split_rhat,ess_bulk,ess_tail",
                       "sharding": f"independent models, {world} rank(s), RCCL all-gather of summaries (library, no torch)"},
            "validated": valid, "max_rel_err": worst,
            "timing": timing_note(a, windows), "ms_per_step_first_window": round(windows[0] / a.steps * 1e3, 5),
            "ms_per_step_windows": [round(s / a.steps * 1e3, 5) for s in windows],
            "ms_per_step_event_pass": elapsed_prof / a.steps * 1e3,
            "pipeline_alg_GBps": value / world * es / 1e9,
            "pipeline_frac_of_hbm": value / world * es / 1e9 / HBM_PEAK_GBS,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "peak_measured": peak_measured, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "frac_of_measured": (achieved / peak_measured) if peak_measured else None,
                         "traffic": traffic,
                         "traffic_source": (str(TRAFFIC_FILE.relative_to(ROOT)) + " (rocprofv3 --pmc passes of an earlier run of this "
                                            "configuration, not counters of this run)") if traffic is not None else None,
                         "avg_launch_us": dom_avg_s * 1e6, "alg_bytes_per_launch": dom_alg},
            "hbm_probe": probe,
            "kernels": kern,
            "kernels_traffic_source": (str(TRAFFIC_FILE.relative_to(ROOT)) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                       "an earlier run of this configuration, MCR_LANES=1; not counters of this run)") if ktraffic else None,
            "traffic_bytes_per_step": int(sum(v for k, v in ktraffic.items() if isinstance(v, (int, float)))) if ktraffic else None,
            "moments_roofline": moments,
            "cpu_baseline": cpu,
            "ms_per_step_per_rank": [round(x, 5) for x in per_rank_ms],
        }
        out.update(extras)
        print(json.dumps(out), flush=True)
    t.free()
    return 0 if valid else 1


def traffic_table() -> dict:
    if not TRAFFIC_FILE.exists():
        return {}
    try:
        return json.loads(TRAFFIC_FILE.read_text())
    except Exception:  # noqa: BLE001
        return {}


if __name__ == "__main__":
    try:
        sys.exit(main())
    except SystemExit:
        raise
    except BaseException:  # noqa: BLE001
        # A rank that fails -- a collective past its deadline (MCR_COMM_TIMEOUT_S: a peer died or never came), a bad file --
        # reports and LEAVES: with several ranks a normal interpreter exit can wait for ever inside librccl's teardown
        # when a peer is gone, and the launcher only ends the job when every rank has ended.
        import traceback
        traceback.print_exc()
        sys.stderr.flush(); sys.stdout.flush()
        if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("MCR_BENCH_FORCE_DIST") == "1":
            os._exit(3)
        sys.exit(1)
