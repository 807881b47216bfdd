"""The N > 1 code path of bench.py over RCCL, exercised on one GPU: one rank launched exactly as the driver launches
N ranks (`python -m torch.distributed.run ...`; torch lives in the launcher only, bench.py imports none), with
MCR_BENCH_FORCE_DIST=1 so that the unique-id hand-over, ncclCommInitRank, the barrier, the MAX all-reduce of the
clock and the all-gather of the 128-byte records really go through librccl inside libmcmcref_hip
(mcr_comm_*).  World sizes > 1 are rehearsed with a gloo stand-in in tests/test_host_cpu.py and
tests/test_shard_files_gpu.py (RCCL refuses two ranks on one GPU)."""
from __future__ import annotations

import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("workload", ["c1", "corpus", "c1split"])
def test_bench_one_rank_over_rccl(workload):
    env = dict(os.environ, MCR_BENCH_FORCE_DIST="1")     # (HSA_ENABLE_IPC_MODE_LEGACY=0 is set by _ffi.load_library itself)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "1", "--steps", "20",
           "--warmup", "2", "--windows", "2", "--no-cpu-baseline", "--no-moments", "--no-probe", "--workload", workload]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["validated"] is True and out["value"] > 0
    assert out["metric"] == "validated param-draws/sec"


@pytest.mark.parametrize("nonblocking", ["0", "1"])
def test_communicator_collectives_one_rank(tmp_path, monkeypatch, nonblocking):
    """mcr_comm_* directly: a world of one over RCCL (ncclAllGather / ncclAllReduce on this GPU), gather_records on top.
    Both communicator modes: the default (blocking communicator, init on a watched helper thread, polled stream waits) and
    MCR_COMM_NONBLOCKING=1 (ncclCommInitRankConfig with blocking = 0)."""
    import numpy as np
    from mcmc_ref_hip import _ffi, shard
    monkeypatch.setenv("MCR_COMM_NONBLOCKING", nonblocking)
    monkeypatch.setenv("MCR_COMM_DIR", str(tmp_path))
    monkeypatch.setenv("MASTER_PORT", str(_free_port()))
    with _ffi.Context(0) as ctx, shard.Communicator(ctx, world=1, rank=0) as comm:
        assert (comm.world, comm.rank) == (1, 0)
        assert comm.has_deadline                                    # every wait below is bounded by MCR_COMM_TIMEOUT_S
        assert list(tmp_path.iterdir()) == []                       # rank 0 removed the id file after the init
        x = np.arange(48.0).reshape(3, 16)
        assert np.array_equal(comm.all_gather(x), x[None])
        assert comm.all_reduce([1.5, -2.0], "max").tolist() == [1.5, -2.0]
        assert comm.all_reduce([3.0], "sum").tolist() == [3.0]
        comm.barrier()
        rec = np.random.default_rng(0).normal(size=(5, shard.RECORD_DOUBLES))
        rec[:, shard.RECORD_FIELDS.index("model_idx")] = [2, 0, 1, 0, 2]
        rec[:, shard.RECORD_FIELDS.index("param_idx")] = [1, 0, 0, 1, 0]
        out = shard.gather_records(rec, comm)
        assert np.array_equal(out, rec[[1, 3, 2, 4, 0]])
        assert shard.gather_records(np.empty((0, shard.RECORD_DOUBLES)), comm).shape == (0, shard.RECORD_DOUBLES)
        x3 = np.random.default_rng(1).normal(size=(6, 4, 500))
        split = shard.summarize_param_split(ctx, x3, comm)
        whole = ctx.summarize(x3, "pcn")
        assert np.array_equal(split[:, shard.RECORD_FIELDS.index("ess_bulk")], whole["ess_bulk"])
        assert split[:, shard.RECORD_FIELDS.index("param_idx")].tolist() == list(range(6))


_ABSENT_PEER = r"""
import ctypes as C, sys, time
sys.path[:0] = [{root!r}, {pkg!r}]
from mcmc_ref_hip import _ffi
L = _ffi.load_library()
ctx = _ffi.Context(0)
uid = C.create_string_buffer(_ffi.MCR_COMM_ID_BYTES if hasattr(_ffi, "MCR_COMM_ID_BYTES") else 128)
assert L.mcr_comm_unique_id(uid, len(uid)) == 0
h = C.c_void_p()
t0 = time.time()
rc = L.mcr_comm_init(ctx.handle, uid, 2, 0, C.byref(h))            # a world of two, and rank 1 never comes
dt = time.time() - t0
print("RC", rc, "SECONDS", round(dt, 1), "MSG", (L.mcr_last_error(ctx.handle) or b"").decode(), flush=True)
ok = ctx.summarize(__import__("numpy").random.default_rng(0).normal(size=(2, 4, 200)), "pcn")    # the context still works
print("STILL_WORKS", bool(ok["ess_bulk"][0] > 0), flush=True)
import os
os._exit(0)     # RCCL 2.27.7 does not return from aborting an init that waits for a peer: the helper thread that tried is
                # still inside it, and a normal interpreter exit would wait for librccl's teardown (bench.py leaves the same way)
"""


@pytest.mark.parametrize("nonblocking", ["0", "1"])
def test_a_peer_that_never_arrives_ends_in_an_error_not_a_hang(tmp_path, nonblocking):
    """ncclCommInitRank with a world of two and only this rank present.  The call gives up after
    MCR_COMM_TIMEOUT_S and returns MCR_ECOMM naming the call and the rank; the process goes on (its context still
    computes).  Both communicator modes (the default blocking one under the watchdog, MCR_COMM_NONBLOCKING=1).  Runs in a
    child with a hard limit so that a hang would fail, not stall, the suite."""
    from mcmc_ref_hip import _ffi
    env = dict(os.environ, MCR_COMM_TIMEOUT_S="4", MCR_COMM_NONBLOCKING=nonblocking)
    code = _ABSENT_PEER.format(root=str(ROOT), pkg=str(ROOT / "mcmc-db_amd"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=75, cwd=str(ROOT))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RC ")][0].split()
    assert int(line[1]) == _ffi.MCR_ECOMM, r.stdout
    assert 3.0 <= float(line[3]) <= 60.0, r.stdout
    assert "rank 0 of 2 gave up" in r.stdout and "ncclCommInitRank" in r.stdout and "did not arrive" in r.stdout
    assert "STILL_WORKS True" in r.stdout
