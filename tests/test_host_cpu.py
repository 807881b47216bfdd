"""CPU-only tests: the C ABI loads and exports every declared symbol, the host-side mirror of the
reference interface behaves (no compute calls without a GPU), and the multi-rank gather works on
gloo with world_size 2."""
from __future__ import annotations

import ctypes
import json
import os
import re
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def built_lib():
    import importlib.util
    spec = importlib.util.spec_from_file_location("mcr_build", ROOT / "mcmc-db_amd" / "build.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build()


def test_abi_exports_every_declared_symbol(built_lib):
    header = (ROOT / "include" / "mcmcref_hip.h").read_text()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(mcr_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    L = ctypes.CDLL(str(built_lib))
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in mcmcref_hip.h but not exported"
    from mcmc_ref_hip import _ffi
    assert set(_ffi.SYMBOLS) == declared
    lib = _ffi.load_library()
    assert lib.mcr_version() == 100


def test_no_gpu_fails_loudly(built_lib):
    from mcmc_ref_hip import _ffi, backends, diagnostics
    if _ffi.load_library().mcr_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_ffi.HipUnavailableError):
        _ffi.Context(0)
    with pytest.raises(ImportError):
        backends.get_backend("hip")
    with pytest.raises(_ffi.HipUnavailableError):      # never a silent CPU answer
        diagnostics.split_rhat([[1.0, 2.0, 3.0, 4.0]] * 4)


def test_product_package_never_imports_oracle():
    for py in (ROOT / "mcmc-db_amd").rglob("*.py"):
        text = py.read_text()
        assert "oracle" not in text, f"{py} mentions the oracle"
    for src in (ROOT / "mcmc-db_amd" / "csrc").iterdir():
        assert "oracle" not in src.read_text()


def test_reference_guards_and_messages_without_gpu():
    from mcmc_ref_hip import backends, convert, diagnostics
    with pytest.raises(ValueError, match="at least 4 chains"):
        diagnostics.split_rhat([[1.0, 2.0, 3.0, 4.0], [1.1, 2.1, 3.1, 4.1]])
    with pytest.raises(ValueError, match=r"ESS diagnostics require at least 4 chains; got 2 chain\(s\)"):
        diagnostics.ess_bulk([[1.0, 2.0, 3.0, 4.0], [1.1, 2.1, 3.1, 4.1]])
    with pytest.raises(ValueError, match="min_chains must be >= 1; got 0"):
        diagnostics.ess_tail([[1.0]] * 4, min_chains=0)
    import math
    assert math.isnan(diagnostics.split_rhat([[1.0, 2.0, 3.0, 4.0]], min_chains=1))     # no GPU needed
    with pytest.raises(ValueError, match="Unknown backend: nope"):
        backends.get_backend("nope")
    assert convert._checks(10, 1000, {"a": {"rhat": 1.001, "ess_bulk": 900.0}}) == {
        "ndraws_is_10k": True, "nchains_is_gte_4": True, "ess_above_400": True, "rhat_below_1_01": True}
    chk = convert._checks(2, 10, {"a": {"rhat": float("nan"), "ess_bulk": 10.0}})
    assert chk == {"ndraws_is_10k": False, "nchains_is_gte_4": False, "ess_above_400": False,
                   "rhat_below_1_01": False}
    with pytest.raises(ValueError, match="quality checks failed: ndraws_is_10k, nchains_is_gte_4"):
        convert._enforce_checks({"ndraws_is_10k": False, "nchains_is_gte_4": False, "ess_above_400": True})


def test_table_to_tensor_orders_like_chains_from_table():
    import pyarrow as pa
    from mcmc_ref_hip import convert
    rng = np.random.default_rng(0)
    C, N = 4, 25
    chain = np.repeat(np.arange(C), N)
    draw = np.tile(np.arange(N), C)
    mu = rng.normal(size=C * N)
    perm = rng.permutation(C * N)
    tbl = pa.table({"chain": chain[perm], "draw": draw[perm], "mu": mu[perm], "tau": (mu * 2)[perm]})
    x, counts = convert.table_to_tensor(tbl, ["mu", "tau"])
    assert list(counts) == [N] * C
    assert np.array_equal(x[0], mu) and np.array_equal(x[1], mu * 2)
    assert convert._count_chains_draws(tbl) == (C, N)
    # already ordered: no gather
    tbl2 = pa.table({"chain": pa.array(chain, type=pa.int32()), "draw": pa.array(draw, type=pa.int32()), "mu": mu})
    ids, order, counts = convert.chain_layout(tbl2)
    assert order is None and list(ids) == [0, 1, 2, 3]
    # ragged
    keep = np.ones(C * N, dtype=bool)
    keep[-3:] = False
    tbl3 = pa.table({"chain": chain[keep], "draw": draw[keep], "mu": mu[keep]})
    assert convert._count_chains_draws(tbl3) == (C, N - 3)


def test_tensor_args_layouts():
    from mcmc_ref_hip import _ffi
    x = np.zeros((5, 4, 7))
    assert _ffi.tensor_args(x, "pcn") == (_ffi.MCR_F64, 4, 7, 5, 7, 1, 28)
    y = np.zeros((4, 7, 5), dtype=np.float32)
    assert _ffi.tensor_args(y, "cnp") == (_ffi.MCR_F32, 4, 7, 5, 35, 5, 1)
    with pytest.raises(TypeError):
        _ffi.tensor_args(np.zeros((1, 1, 1), dtype=np.int32), "pcn")
    with pytest.raises(ValueError):
        _ffi.tensor_args(np.zeros((2, 2)), "pcn")


def test_plan_shards_lpt():
    from mcmc_ref_hip import shard
    costs = [450000, 20000, 330000, 30000, 100000, 100000, 70000, 1750000]
    plan = shard.plan_shards(costs, 3)
    assert sorted(i for s in plan for i in s) == list(range(len(costs)))
    loads = [sum(costs[i] for i in s) for s in plan]
    assert max(loads) == 1750000                      # the big model sits alone
    assert shard.plan_shards(costs, 3) == plan        # deterministic
    assert shard.plan_shards([], 2) == [[], []]
    assert shard.plan_shards([1, 1, 1], 1) == [[0, 1, 2]]


def _gloo_worker(rank: int, world: int, port: int, tmp: str):
    import torch.distributed as dist
    sys.path[:0] = [str(ROOT), str(ROOT / "mcmc-db_amd"), str(ROOT / "tests")]
    from gloo_comm import GlooComm
    from mcmc_ref_hip import shard
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = GlooComm(dist)
        # 5 models with different parameter counts; every rank derives the same plan
        sizes = [3, 1, 4, 2, 5]
        plan = shard.plan_shards([s * 1000 for s in sizes], world)
        recs = []
        for m in plan[rank]:
            P = sizes[m]
            summ = {k: np.full(P, 10.0 * m + j) for j, k in enumerate(
                ("mean", "std", "rhat_bulk", "rhat_tail", "rhat", "ess_bulk", "ess_tail"))}
            summ["q"] = np.tile(np.array([[m + 0.05, m + 0.5, m + 0.95]]), (P, 1))
            summ["lag_bulk"] = np.arange(P, dtype=np.int64) + m
            summ["lag_tail"] = np.arange(P, dtype=np.int64) * 2
            recs.append(shard.pack_records(summ, m, 4, 1000))
        local = np.concatenate(recs) if recs else np.empty((0, shard.RECORD_DOUBLES))
        allrec = shard.gather_records(local, comm)
        np.save(os.path.join(tmp, f"rank{rank}.npy"), allrec)
        # one model split by parameter blocks: the records carry global parameter indices
        P = 7
        p0, p1 = shard.param_block(P, world, rank)
        part = {k: np.arange(p0, p1, dtype=np.float64) for k in ("mean", "std", "rhat_bulk", "rhat_tail", "rhat", "ess_bulk", "ess_tail")}
        part["q"] = np.tile(np.arange(p0, p1, dtype=np.float64)[:, None], (1, 3))
        part["lag_bulk"] = part["lag_tail"] = np.arange(p0, p1, dtype=np.int64)
        split = shard.gather_records(shard.pack_records(part, 0, 4, 10, param0=p0), comm)
        np.save(os.path.join(tmp, f"split{rank}.npy"), split)
        # a failure on ONE rank is agreed on before the data collective: every rank raises, nobody hangs
        try:
            shard._agree(comm, ValueError("bad file") if rank == 1 else None, "unit")
            outcome = "no error"
        except ValueError as e:
            outcome = f"own: {e}"
        except RuntimeError as e:
            outcome = f"peer: {e}"
        with open(os.path.join(tmp, f"agree{rank}.txt"), "w") as f:
            f.write(outcome)
        shard._agree(comm, None, "unit")                      # and a healthy round passes
    finally:
        dist.destroy_process_group()


def test_gather_records_gloo_world2(tmp_path):
    import torch.multiprocessing as mp
    from mcmc_ref_hip import shard
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_gloo_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "rank0.npy")
    b = np.load(tmp_path / "rank1.npy")
    assert np.array_equal(a, b, equal_nan=True)
    assert a.shape == (3 + 1 + 4 + 2 + 5, shard.RECORD_DOUBLES)
    mi, pi = shard.RECORD_FIELDS.index("model_idx"), shard.RECORD_FIELDS.index("param_idx")
    assert list(a[:, mi]) == [0] * 3 + [1] + [2] * 4 + [3] * 2 + [4] * 5
    assert list(a[:3, pi]) == [0, 1, 2]
    assert a[4, shard.RECORD_FIELDS.index("q50")] == 2.5 and a[4, shard.RECORD_FIELDS.index("n_draws")] == 1000
    # single-process path is the identity (sorted)
    assert np.array_equal(shard.gather_records(a[::-1]), a)
    s0, s1 = np.load(tmp_path / "split0.npy"), np.load(tmp_path / "split1.npy")
    assert np.array_equal(s0, s1) and list(s0[:, pi]) == list(range(7)) and list(s0[:, 0]) == list(range(7))
    assert (tmp_path / "agree0.txt").read_text() == "peer: unit failed on rank(s) [1]; this rank (0) stops with them"
    assert (tmp_path / "agree1.txt").read_text() == "own: bad file"


def test_param_block_and_id_file(monkeypatch, tmp_path):
    from mcmc_ref_hip import shard
    for P, world in ((10000, 8), (7, 3), (2, 4), (0, 2)):
        blocks = [shard.param_block(P, world, r) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == P
        assert all(blocks[r][1] == blocks[r + 1][0] for r in range(world - 1))
        sizes = [b - a for a, b in blocks]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.param_block(4, 2, 2)
    monkeypatch.setenv("MCR_COMM_DIR", str(tmp_path))
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1"); monkeypatch.setenv("MASTER_PORT", "29511")
    for k in ("MCR_COMM_KEY", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT"):
        monkeypatch.delenv(k, raising=False)
    a = shard._id_file(2)
    monkeypatch.setenv("MASTER_PORT", "29512")
    b = shard._id_file(2)
    assert a.parent == tmp_path and a != b and "29511" in a.name and a != shard._id_file(4)
    assert str(os.getppid()) in a.name                       # no run id: the launcher's pid tells launches apart
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "none")        # torchrun's default run id is no better than none at all
    assert shard._id_file(2) == b
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "job-17")      # a real run id replaces the pid
    c = shard._id_file(2)
    assert "job-17" in c.name and str(os.getppid()) not in c.name.replace("29512", "")
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "1")
    assert shard._id_file(2) != c
    monkeypatch.setenv("MCR_COMM_KEY", "my/key:1")           # an explicit key is the whole key
    d = shard._id_file(2)
    monkeypatch.setenv("MASTER_PORT", "29513")
    assert shard._id_file(2) == d and d.name == "mcr_rccl_id_my-key-1"
    assert shard.comm_key(2)[1] == ["MCR_COMM_KEY=my/key:1"]


def test_sorting_network_is_a_sorting_network(tmp_path):
    """0-1 principle: the 16- and 8-input networks of csrc/mcr_sortnet.h sort all binary inputs."""
    import subprocess
    src = tmp_path / "check.c"
    src.write_text(r'''
#include <stdio.h>
#include "mcr_sortnet.h"
#define X(a, b) {a, b},
static const int net16[][2] = { MCR_NET16(X) };
static const int net8[][2] = { MCR_NET8(X) };
static long check(const int (*net)[2], int n, int w) {
    long bad = 0;
    for (unsigned m = 0; m < (1u << w); ++m) {
        int v[16];
        for (int i = 0; i < w; ++i) v[i] = (m >> i) & 1;
        for (int c = 0; c < n; ++c) { int a = net[c][0], b = net[c][1]; if (a >= b || b >= w) return -1;
            if (v[b] < v[a]) { int t = v[a]; v[a] = v[b]; v[b] = t; } }
        for (int i = 0; i + 1 < w; ++i) if (v[i] > v[i + 1]) { ++bad; break; }
    }
    return bad;
}
int main(void) {
    const int n16 = (int)(sizeof(net16) / sizeof(net16[0])), n8 = (int)(sizeof(net8) / sizeof(net8[0]));
    const long b16 = check(net16, n16, 16), b8 = check(net8, n8, 8);
    printf("%d %ld %d %ld\n", n16, b16, n8, b8);
    return b16 != 0 || b8 != 0;
}
''')
    exe = tmp_path / "check"
    subprocess.run(["gcc", "-O2", "-I", str(ROOT / "mcmc-db_amd" / "csrc"), "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert out == ["60", "0", "19", "0"]


def test_register_merge_levels_merge_every_pair_of_sorted_01_runs(tmp_path):
    """The tile sort's merge levels in registers (csrc/mcr_kernels.hpp lane_merge_levels_16_to_64 and its third level):
    a group of 2 / 4 / 8 lanes with 16 registers each holds two sorted runs; the mirror exchange with the partner lane
    (lane ^ 1, mirror lane of the quad, mirror lane of the group of 8; mirror register), the half-cleaners across lanes
    (lane ^ 2, lane ^ 1; same register) and the 16-input bitonic merge of mcr_sortnet.h inside each lane must leave the
    group sorted.  0-1 principle for merging networks: every pair of sorted 0-1 runs."""
    import subprocess
    src = tmp_path / "levels.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "mcr_sortnet.h"
#define X(a, b) {a, b},
static const int bit16[][2] = { MCR_BITONIC16(X) };
static int v[8][16];
/* one exchange stage as the kernel does it: every lane reads the partner's OLD registers */
static void stage(int lanes, int xor_or_mirror, int mirror_regs, int min_mask) {
    int o[8][16];
    memcpy(o, v, sizeof(o));
    for (int l = 0; l < lanes; ++l) {
        const int partner = xor_or_mirror < 0 ? (l & ~(-xor_or_mirror - 1)) + ((-xor_or_mirror - 1) - (l & (-xor_or_mirror - 1))) : l ^ xor_or_mirror;
        const int keep_min = (l & min_mask) == 0;
        for (int r = 0; r < 16; ++r) {
            const int other = o[partner][mirror_regs ? 15 - r : r], own = o[l][r];
            v[l][r] = keep_min ? (other < own ? other : own) : (own < other ? other : own);
        }
    }
}
static void local(int lanes) {
    const int n = (int)(sizeof(bit16) / sizeof(bit16[0]));
    for (int l = 0; l < lanes; ++l)
        for (int c = 0; c < n; ++c) { const int a = bit16[c][0], b = bit16[c][1];
            if (v[l][b] < v[l][a]) { const int t = v[l][a]; v[l][a] = v[l][b]; v[l][b] = t; } }
}
static long level(int lanes) {   /* lanes = 2, 4, 8: two sorted runs of 8 * lanes inputs each */
    const int half = 8 * lanes;
    long bad = 0;
    for (int za = 0; za <= half; ++za)
        for (int zb = 0; zb <= half; ++zb) {       /* run A: za zeros then ones; run B: zb zeros then ones */
            for (int e = 0; e < half; ++e) { v[e / 16][e % 16] = e >= za; v[lanes / 2 + e / 16][e % 16] = e >= zb; }
            stage(lanes, -lanes, 1, lanes / 2);            /* mirror lane of the group, mirror register */
            for (int d = lanes / 4; d >= 1; d /= 2) stage(lanes, d, 0, d);      /* half-cleaners across lanes */
            local(lanes);
            int prev = 0, ones = 0;
            for (int e = 0; e < 2 * half; ++e) { const int x = v[e / 16][e % 16]; if (x < prev) { ++bad; break; } prev = x; ones += x; }
            if (ones != 2 * half - za - zb) ++bad;
        }
    return bad;
}
int main(void) {
    const long b2 = level(2), b4 = level(4), b8 = level(8);
    printf("%ld %ld %ld\n", b2, b4, b8);
    return b2 != 0 || b4 != 0 || b8 != 0;
}
''')
    exe = tmp_path / "levels"
    subprocess.run(["gcc", "-O2", "-I", str(ROOT / "mcmc-db_amd" / "csrc"), "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert out == ["0", "0", "0"]


def test_store_layout_and_precedence(tmp_path):
    import pyarrow as pa
    import pyarrow.parquet as pq
    from mcmc_ref_hip.store import DataStore
    pkg, loc = tmp_path / "pkg", tmp_path / "loc"
    for root, val in ((pkg, 1.0), (loc, 2.0)):
        (root / "draws").mkdir(parents=True); (root / "meta").mkdir(); (root / "stan_data").mkdir()
        pq.write_table(pa.table({"chain": [0, 0, 1, 1], "draw": [0, 1, 0, 1], "mu": [val] * 4, "tau": [3.0] * 4}),
                       root / "draws" / "m.draws.parquet")
        (root / "meta" / "m.meta.json").write_text(json.dumps({"who": root.name}))
    (loc / "stan_models").mkdir()
    (loc / "stan_models" / "m.stan").write_text("parameters { real mu; }")
    pq.write_table(pa.table({"chain": [0], "draw": [0], "x": [1.0]}), loc / "draws" / "only_local.draws.parquet")
    st = DataStore(local_root=loc, packaged_root=pkg)
    assert st.list_models() == ["m", "only_local"]
    assert st.read_meta("m") == {"who": "pkg"}                       # packaged root wins
    assert st.resolve_draws_path("only_local").parent.parent == loc
    assert st.read_stan_code("m").startswith("parameters")
    with pytest.raises(FileNotFoundError, match="stan data not found for model: m"):
        st.read_stan_data("m")
    with pytest.raises(FileNotFoundError, match="draws not found for model: zzz"):
        st.resolve_draws_path("zzz")
    t = st.open_draws("m", params=["tau"], chains=[1]).read_all()
    assert t.column_names == ["chain", "draw", "tau"] and t.num_rows == 2
    assert st.open_draws("m").read_all().column("mu").to_pylist() == [1.0] * 4
    assert DataStore(local_root=tmp_path / "none", packaged_root=tmp_path / "none2").list_models() == []


def test_convert_readers_cpu(tmp_path):
    import zipfile
    import pyarrow as pa
    from mcmc_ref_hip import convert
    jz = tmp_path / "a.json.zip"
    with zipfile.ZipFile(jz, "w") as zf:
        zf.writestr("a.json", json.dumps([{"b": [1.0, 2.0], "a": [3.0, 4.0]}, {"b": [5.0, 6.0], "a": [7.0, 8.0]}]))
    t = convert._read_json_zip(jz)
    assert t.column_names == ["chain", "draw", "a", "b"]
    assert t.column("chain").to_pylist() == [0, 0, 1, 1] and t.column("a").to_pylist() == [3.0, 4.0, 7.0, 8.0]
    with zipfile.ZipFile(jz, "w") as zf:
        zf.writestr("a.json", json.dumps({"not": "a list"}))
    with pytest.raises(ValueError, match="non-empty list of chains"):
        convert._read_json_zip(jz)
    bare = convert._ensure_chain_draw(pa.table({"x": [1.0, 2.0, 3.0]}))
    assert bare.column_names == ["x", "chain", "draw"] and bare.column("draw").to_pylist() == [0, 1, 2]
    only_chain = convert._ensure_chain_draw(pa.table({"chain": [0, 0], "x": [1.0, 2.0]}))
    assert only_chain.column_names == ["chain", "x", "draw"]
