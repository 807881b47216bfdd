"""The N > 1 code path of bench.py over RCCL, exercised on one GPU: a one-rank process group launched exactly as
the driver launches N ranks (`python -m torch.distributed.run ...`), with MCR_BENCH_FORCE_DIST=1 so that the
barrier, the MAX all_reduce of the timing and the all_gather of the 128-byte records really go through
backend "nccl" (= RCCL) on device tensors.  World sizes > 1 are covered with gloo in tests/test_host_cpu.py."""
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


@pytest.mark.parametrize("workload", ["c1", "corpus"])
def test_bench_one_rank_over_rccl(workload):
    env = dict(os.environ, MCR_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "1", "--steps", "20",
           "--warmup", "2", "--no-cpu-baseline", "--no-moments", "--workload", workload]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["validated"] is True and out["value"] > 0
    assert out["metric"] == "validated param-draws/sec"
