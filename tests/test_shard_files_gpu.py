"""Sharded corpus-from-disk (shard.summarize_paths): one rank, and two ranks sharing the GPU over gloo (the real
N > 1 runs use RCCL; the collective code path is the same `gather_records`)."""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _write(tmp_path, shapes, seed=3, shuffle=()):
    rng = np.random.default_rng(seed)
    paths = []
    for k, (C, N, P) in enumerate(shapes):
        cols = {"chain": np.repeat(np.arange(C), N), "draw": np.tile(np.arange(N), C)}
        for p in range(P):
            cols[f"b[{p + 1}]"] = rng.normal(size=C * N) * (p + 1)
        t = pa.table(cols)
        if k in shuffle:
            t = t.take(pa.array(rng.permutation(C * N)))
        path = tmp_path / f"model{k:02d}.draws.parquet"
        pq.write_table(t, path)
        paths.append(path)
    return paths


SHAPES = [(4, 500, 3), (4, 500, 5), (10, 100, 2), (4, 2500, 4), (4, 500, 1), (10, 100, 6), (4, 500, 2)]


def test_single_rank_records_match_summarize_files(tmp_path):
    from mcmc_ref_hip import _ffi, parquet, shard
    paths = _write(tmp_path, SHAPES)
    with _ffi.Context(0) as ctx:
        rec = shard.summarize_paths(ctx, paths)
        ref = parquet.summarize_files(ctx, paths)
        assert rec.shape == (sum(s[2] for s in SHAPES), shard.RECORD_DOUBLES)
        F = shard.RECORD_FIELDS.index
        row = 0
        for i, res in enumerate(ref):
            for j, (name, e) in enumerate(res.items()):
                r = rec[row]
                assert r[F("model_idx")] == i and r[F("param_idx")] == j
                assert (r[F("n_chains")], r[F("n_draws")]) == (SHAPES[i][0], SHAPES[i][1])
                for k in ("mean", "std", "q5", "q50", "q95", "rhat", "ess_bulk", "ess_tail"):
                    assert r[F(k)] == e[k], (i, name, k)
                assert r[F("rhat")] == max(r[F("rhat_bulk")], r[F("rhat_tail")]) and r[F("lag_bulk")] >= 0
                row += 1
        # a shuffled file in the share: same records through the per-file route
        paths2 = _write(tmp_path / "s", SHAPES[:3], shuffle=(1,)) if (tmp_path / "s").mkdir() is None else None
        rec2 = shard.summarize_paths(ctx, paths2)
        ref2 = parquet.summarize_files(ctx, paths2)
        assert rec2.shape[0] == 10 and rec2[3, F("ess_bulk")] == list(ref2[1].values())[0]["ess_bulk"]


def test_two_ranks_over_gloo_share_the_gpu(tmp_path):
    paths = _write(tmp_path, SHAPES)
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path[:0] = [{str(ROOT)!r}, {str(ROOT / 'mcmc-db_amd')!r}]
        sys.path.insert(0, {str(ROOT / 'tests')!r})
        import numpy as np, torch.distributed as dist
        from gloo_comm import GlooComm
        from mcmc_ref_hip import _ffi, shard
        dist.init_process_group("gloo")
        comm = GlooComm(dist)
        rank, world = comm.rank, comm.world
        paths = sorted(p for p in os.listdir({str(tmp_path)!r}) if p.endswith(".parquet"))
        paths = [os.path.join({str(tmp_path)!r}, p) for p in paths]
        with _ffi.Context(0) as ctx:
            rec = shard.summarize_paths(ctx, paths, comm)
            if rank == 0:
                one = shard.summarize_paths(ctx, paths)
                assert rec.shape == one.shape and np.array_equal(rec, one, equal_nan=True), "sharded != single"
                print("OK", rec.shape[0])
        dist.destroy_process_group()
    """))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=dict(os.environ))
    assert r.returncode == 0, r.stderr[-2000:]
    assert f"OK {sum(s[2] for s in SHAPES)}" in r.stdout
