"""The N > 1 rendezvous of mcmc_ref_hip.shard executed for real: two worker processes under
`python -m torch.distributed.run --nproc-per-node 2` (the driver's launch line), no GPU call in them.

What is parallelised over those ranks is the model loop of src/mcmc_ref/generate.py:77-96 / convert.py:140-147; this
file only covers how the ranks of one launch find each other."""
from __future__ import annotations

import json
import os
import socket
import subprocess
import sys
import threading
import time
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "mcmc-db_amd")]
from mcmc_ref_hip import shard  # noqa: E402


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(tmp_path: Path, world: int = 2, **env) -> dict[int, dict]:
    out = tmp_path / "out"
    out.mkdir(exist_ok=True)
    e = dict(os.environ, RDZV_OUT=str(out), MCR_COMM_DIR=str(tmp_path / "comm"), **{k: str(v) for k, v in env.items()})
    e.pop("MCR_COMM_KEY", None)
    (tmp_path / "comm").mkdir(exist_ok=True)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "tests" / "rdzv_worker.py")]
    r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    return {k: json.loads((out / f"rank{k}.json").read_text()) for k in range(world)}


def test_two_ranks_under_torchrun_agree_on_the_id(tmp_path):
    recs = _launch(tmp_path)
    assert recs[0]["base"] == recs[1]["base"]                          # both resolve the same rendezvous path ...
    assert recs[0]["ppid"] == recs[1]["ppid"] and f"launcher_pid={recs[0]['ppid']}" in recs[0]["key_parts"]   # ... through the agent's pid
    assert recs[0]["id_sha"] == recs[1]["id_sha"] == recs[0]["made"]   # rank 1 received rank 0's 128 bytes
    assert recs[1]["made"] is None
    assert list((tmp_path / "comm").iterdir()) == []                   # nothing left behind


def test_three_ranks_and_an_explicit_run_id(tmp_path):
    recs = _launch(tmp_path, world=3, TORCHELASTIC_RUN_ID="job-5")     # (torchrun overwrites it with its own --rdzv-id default)
    assert len({r["base"] for r in recs.values()}) == 1
    assert recs[0]["id_sha"] == recs[1]["id_sha"] == recs[2]["id_sha"]
    assert list((tmp_path / "comm").iterdir()) == []


def test_missing_rank0_times_out_with_the_path_in_the_message(tmp_path):
    t0 = time.monotonic()
    recs = _launch(tmp_path, RDZV_MODE="absent0", RDZV_TIMEOUT="2")
    assert time.monotonic() - t0 < 120
    assert recs[0].get("skipped") and "timeout" in recs[1]
    msg = recs[1]["timeout"]
    assert recs[1]["base"] in msg and "launcher_pid=" in msg and "MCR_COMM_KEY" in msg


def _threads(base: Path, world: int, timeout: float = 20.0, delay0: float = 0.0):
    res: dict[int, object] = {}

    def run(r):
        try:
            if r == 0 and delay0:
                time.sleep(delay0)
            res[r] = shard.exchange_unique_id(r, world, lambda: bytes(range(128)), timeout=timeout, base=base)
        except Exception as exc:  # noqa: BLE001
            res[r] = exc
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return res


def test_stale_files_of_a_crashed_launch_are_not_picked_up(tmp_path):
    base = tmp_path / "mcr_rccl_id_k"
    stale_id = bytes([7]) * shard.COMM_ID_BYTES
    # a previous launch under the same key died after rank 0 had published and rank 1 had said hello / acknowledged
    base.with_name(base.name + ".id").write_bytes(stale_id + bytes([9]) * shard.COMM_NONCE_BYTES)
    base.with_name(base.name + ".hello.1").write_bytes(bytes([9]) * shard.COMM_NONCE_BYTES)
    base.with_name(base.name + ".ack.1").write_bytes(bytes([9]) * shard.COMM_NONCE_BYTES)
    res = _threads(base, 2, delay0=0.3)          # rank 1 sees the stale id file for a while before rank 0 shows up
    assert res[0] == res[1] == bytes(range(128)) != stale_id
    assert list(tmp_path.iterdir()) == []


def test_exchange_edge_cases(tmp_path, monkeypatch):
    assert shard.exchange_unique_id(0, 1, lambda: b"x" * 128) == b"x" * 128        # a world of one touches no file
    with pytest.raises(ValueError):
        shard.exchange_unique_id(2, 2, lambda: b"")
    res = _threads(tmp_path / "b", 4)
    assert all(res[r] == bytes(range(128)) for r in range(4)) and list(tmp_path.iterdir()) == []
    with pytest.raises(TimeoutError) as ei:      # nobody else: rank 0 gives up too and cleans up
        shard.exchange_unique_id(0, 2, lambda: bytes(128), timeout=0.3, base=tmp_path / "c")
    assert str(tmp_path / "c") in str(ei.value) and list(tmp_path.iterdir()) == []
    # a multi-node launch with a node-local directory fails at once, before any polling
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")
    monkeypatch.delenv("MCR_COMM_DIR", raising=False)
    t0 = time.monotonic()
    with pytest.raises(RuntimeError, match="MCR_COMM_DIR"):
        shard.exchange_unique_id(1, 4, lambda: bytes(128), timeout=30)
    assert time.monotonic() - t0 < 1.0


def test_a_rank_that_dies_after_the_rendezvous(tmp_path):
    """VERDICT r3 item 5, the control flow on CPU: two ranks over the gloo stand-in, rank 1 dies (os._exit) right after
    the barrier that ends the rendezvous; rank 0's failure agreement / record gather (mcmc_ref_hip.shard) must raise
    within the collective's deadline and the process must leave with a non-zero code -- nobody waits for ever.  (The
    product's own deadline lives in the library: MCR_COMM_TIMEOUT_S on the non-blocking RCCL communicator, exercised on
    the GPU box by tests/test_dist_rccl_gpu.py::test_a_peer_that_never_arrives_ends_in_an_error_not_a_hang.)"""
    import socket
    import time
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    worker = str(ROOT / "tests" / "dead_peer_worker.py")
    deadline = 20.0
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(port), str(deadline)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240) for p in procs]
    took = time.time() - t0
    assert procs[1].returncode == 17                                   # the rank that died
    assert procs[0].returncode == 3, (outs[0][0][-800:], outs[0][1][-800:])
    assert "RAISED" in outs[0][0] and "NO ERROR" not in outs[0][0]
    assert took < 60 + deadline
