"""Native Parquet ingest on the GPU: `mcr_parquet_decode` through the C ABI against
pyarrow's decode of the same file image -- pyarrow IS the reference's reader for this step
(`pq.read_table`, src/mcmc_ref/store.py:79-95), so equality here is parity with the reference's ingest.
Bit-exact for every value (doubles compared as bit patterns)."""
from __future__ import annotations

import io

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest

from conftest import GOLDEN, load_model

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from mcmc_ref_hip._ffi import Context
    c = Context(0)
    yield c
    c.close()


def image(table, **kw) -> bytes:
    buf = io.BytesIO()
    pq.write_table(table, buf, **kw)
    return buf.getvalue()


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.int64) if a.dtype == np.float64 else a


def check_roundtrip(ctx, img: bytes, columns=None):
    from mcmc_ref_hip.parquet import read_columns
    t = pq.read_table(io.BytesIO(img))
    got = read_columns(ctx, img, columns)
    names = columns if columns is not None else [f.name for f in t.schema if pa.types.is_integer(f.type) or
                                                 pa.types.is_floating(f.type)]
    assert list(got) == list(names)
    for n in names:
        col = t[n].to_numpy()
        exp = col.astype(np.int64) if np.issubdtype(col.dtype, np.integer) else col.astype(np.float64)
        assert got[n].dtype == exp.dtype, n
        assert np.array_equal(bits(got[n]), bits(exp)), n
    return got


def mixed_table(n=6000, seed=3):
    rng = np.random.default_rng(seed)
    x = rng.normal(size=n)
    x[::97] = -0.0
    x[5] = np.inf; x[6] = -np.inf; x[7] = np.nan; x[8] = 5e-324
    return pa.table({
        "chain": np.repeat(np.arange(4), n // 4), "draw": np.tile(np.arange(n // 4), 4),
        "x": x,
        "ties": np.round(rng.normal(size=n), 1),                 # ~60 distinct values: small dictionary, 6-bit indices
        "const": np.full(n, 2.5),                                # one dictionary entry: bit width 0
        "runs": np.repeat(rng.normal(size=n // 50), 50),         # RLE runs inside the index stream
        "f32": rng.normal(size=n).astype(np.float32),
        "i32": rng.integers(-1000, 1000, n).astype(np.int32),
        "i64": rng.integers(-2**40, 2**40, n),
        "label": pa.array([str(i % 5) for i in range(n)]),       # not numeric: present, never requested
    })


@pytest.mark.parametrize("kw", [
    dict(),
    dict(compression="none"),
    dict(use_dictionary=False),
    dict(data_page_version="2.0"),
    dict(data_page_version="2.0", compression="none"),
    dict(row_group_size=700),
    dict(data_page_size=2048, write_batch_size=64),
    dict(data_page_size=1500, write_batch_size=32, row_group_size=1700, data_page_version="2.0"),
    dict(use_dictionary=["ties", "runs"], compression={"x": "none", "ties": "snappy"}),
])
def test_roundtrip_writer_options(ctx, kw):
    check_roundtrip(ctx, image(mixed_table(), **kw))


def test_required_columns_and_subsets(ctx):
    n = 3000
    rng = np.random.default_rng(1)
    schema = pa.schema([pa.field("chain", pa.int32(), nullable=False), pa.field("draw", pa.int32(), nullable=False),
                        pa.field("a", pa.float64(), nullable=False), pa.field("b", pa.float64())])
    t = pa.table({"chain": np.repeat(np.arange(3, dtype=np.int32), n // 3),
                  "draw": np.tile(np.arange(n // 3, dtype=np.int32), 3),
                  "a": rng.normal(size=n), "b": rng.normal(size=n)}, schema=schema)
    img = image(t)
    check_roundtrip(ctx, img)
    check_roundtrip(ctx, img, ["b"])
    check_roundtrip(ctx, img, ["b", "chain", "a"])


def test_dictionary_fallback_and_large_pages(ctx):
    # > 1 MB of distinct doubles: the writer abandons the dictionary mid-chunk (dict pages followed by PLAIN pages)
    rng = np.random.default_rng(2)
    n = 300_000
    t = pa.table({"x": rng.normal(size=n), "small": rng.integers(0, 3000, n).astype(np.float64)})
    img = image(t)
    from mcmc_ref_hip.parquet import ParquetFile
    with ParquetFile(img) as f:
        encs = {p["encoding"] for p in f.pages() if p["column"] == 0 and p["kind"] == 0}
    assert encs == {0, 8}, encs
    check_roundtrip(ctx, img)
    check_roundtrip(ctx, image(t, data_page_size=1 << 22, row_group_size=n))


def test_snappy_back_references(ctx):
    """Element kinds of the Snappy stream: long literals, overlapping copies (runs), 1/2-byte-offset copies and
    copies that reach back beyond the 32 KB LDS window of the kernel."""
    rng = np.random.default_rng(4)
    period = rng.normal(size=5200)                               # 41.6 KB period: offsets > 32 KB
    far = np.tile(period, 12)
    n = far.size
    zeros = np.zeros(n)
    near = np.tile(rng.normal(size=37), n // 37 + 1)[:n]         # 296-byte period: short offsets
    ramp = np.arange(n, dtype=np.float64)                        # shared high bytes: many short copies
    t = pa.table({"far": far, "zeros": zeros, "near": near, "ramp": ramp, "idx": np.arange(n) % 1000})
    for kw in (dict(use_dictionary=False), dict(use_dictionary=False, data_page_size=1 << 20), dict()):
        check_roundtrip(ctx, image(t, **kw))


def test_rejections(ctx):
    from mcmc_ref_hip._ffi import McrError
    from mcmc_ref_hip.parquet import read_columns
    n = 1000
    x = np.arange(n, dtype=np.float64)
    with pytest.raises(McrError, match="null values"):
        read_columns(ctx, image(pa.table({"x": pa.array([1.0, None, 3.0] * 100)})))
    with pytest.raises(McrError, match="null values"):
        read_columns(ctx, image(pa.table({"x": pa.array([None if i == 777 else float(i) for i in range(n)])}),
                                data_page_version="2.0"))
    with pytest.raises(McrError, match="codec"):
        read_columns(ctx, image(pa.table({"x": x}), compression="zstd"))
    with pytest.raises(McrError, match="encoding"):
        read_columns(ctx, image(pa.table({"x": x}), use_dictionary=False, column_encoding={"x": "BYTE_STREAM_SPLIT"}))
    with pytest.raises(McrError, match="physical type"):
        read_columns(ctx, image(pa.table({"s": pa.array(["a", "b"])})), ["s"])
    # a corrupted Snappy payload is reported, not decoded into garbage silently
    img = bytearray(image(pa.table({"x": np.tile(np.arange(50.0), 200)}), use_dictionary=False))
    from mcmc_ref_hip.parquet import ParquetFile
    with ParquetFile(bytes(img)) as f:
        pg = f.pages()[0]
    img[pg["payload_offset"]] ^= 0x55                       # breaks the length preamble
    with pytest.raises(McrError, match="Snappy"):
        read_columns(ctx, bytes(img))
    # the context keeps working after errors
    check_roundtrip(ctx, image(pa.table({"x": x})))
    # empty table
    got = read_columns(ctx, image(pa.table({"x": pa.array([], pa.float64())})))
    assert got["x"].shape == (0,)


def test_packaged_files_match_the_reference_reader(ctx):
    """Real corpus files (parquet-cpp-arrow 23.0.0) against the draws the reference's own reader produced
    (tests/golden/models/*.npz, written by tests/golden/make_golden.py) and against pyarrow here."""
    from mcmc_ref_hip.parquet import read_draws
    for name in ("wells_data-wells_dist", "radon_pooled"):
        path = GOLDEN / "parquet" / f"{name}.draws.parquet"
        check_roundtrip(ctx, path.read_bytes())
        draws, params, rec = load_model(name)                 # [P][C][N]
        d = read_draws(ctx, path)
        assert d.params == params and d.rectangular
        assert d.tensor.targs[1:4] == (draws.shape[1], draws.shape[2], draws.shape[0])
        assert np.array_equal(bits(d.to_host().reshape(draws.shape)), bits(draws))
        # straight into the statistics, no host round trip of the draws
        got = ctx.enqueue(d.tensor, min_chains=1)
        ctx.wait()
        exp = ctx.summarize(draws, "pcn", min_chains=1)
        for k in ("mean", "std", "q", "rhat", "ess_bulk", "ess_tail"):
            assert np.array_equal(bits(got.result()[k]), bits(exp[k])), (name, k)
        d.free()


def test_unordered_rows_and_batched_files(ctx):
    from mcmc_ref_hip.convert import table_to_tensor
    from mcmc_ref_hip.parquet import read_draws_many
    rng = np.random.default_rng(9)
    imgs, tables = [], []
    for k, (C, N, P) in enumerate([(4, 500, 3), (10, 100, 7), (2, 1000, 1), (4, 250, 5)]):
        cols = {"chain": np.repeat(np.arange(C), N), "draw": np.tile(np.arange(N), C)}
        for p in range(P):
            cols[f"theta[{p + 1}]"] = rng.normal(size=C * N)
        t = pa.table(cols)
        if k % 2 == 1:                                         # rows shuffled: needs the (chain, draw) gather
            t = t.take(pa.array(rng.permutation(C * N)))
        tables.append(t)
        imgs.append(image(t, row_group_size=600 if k == 0 else None))
    out = read_draws_many(ctx, imgs)
    for t, d in zip(tables, out):
        params = [n for n in t.column_names if n not in ("chain", "draw")]
        x, counts = table_to_tensor(t, params)
        assert d.params == params and np.array_equal(d.counts, counts)
        assert np.array_equal(bits(d.to_host()), bits(x))
        d.free()
    # ragged chains: decoded, but no rectangular tensor
    t = pa.table({"chain": [0] * 5 + [1] * 3, "draw": list(range(5)) + list(range(3)), "x": np.arange(8.0)})
    d = read_draws_many(ctx, [image(t)])[0]
    assert d.tensor is None and list(d.counts) == [5, 3] and np.array_equal(d.to_host()[0], np.arange(8.0))
    d.free()


def test_reference_api_native_reader_equals_arrow_reader(ctx, tmp_path, monkeypatch):
    """`reference.stats / diagnostics_for_model / summary_for_model` over a store: the native ingest (default)
    and the pyarrow route (MCMC_REF_HIP_READER=arrow) feed the same kernels and must agree bit for bit."""
    from mcmc_ref_hip import reference
    from mcmc_ref_hip.store import DataStore
    root = tmp_path / "pkg"
    (root / "draws").mkdir(parents=True)
    (root / "meta").mkdir()
    names = []
    for name in ("wells_data-wells_dist", "radon_pooled"):
        (root / "draws" / f"{name}.draws.parquet").write_bytes((GOLDEN / "parquet" / f"{name}.draws.parquet").read_bytes())
        names.append(name)
    rng = np.random.default_rng(12)
    t = pa.table({"chain": [0] * 40 + [1] * 30 + [2] * 35 + [3] * 40, "draw": list(range(40)) + list(range(30)) +
                  list(range(35)) + list(range(40)), "a": rng.normal(size=145), "b": rng.normal(size=145)})
    pq.write_table(t, root / "draws" / "ragged.draws.parquet")
    st = DataStore(local_root=tmp_path / "none", packaged_root=root)
    res = {}
    for reader in ("native", "arrow"):
        monkeypatch.setenv("MCMC_REF_HIP_READER", reader)
        res[reader] = {
            "stats": {m: reference.stats(m, store=st) for m in names + ["ragged"]},
            "stats_sub": reference.stats("radon_pooled", params=["sigma"], store=st),
            "diag": {m: reference.diagnostics_for_model(m, store=st) for m in names + ["ragged"]},
            "summ": {m: reference.summary_for_model(m, store=st) for m in names},
        }
    assert res["native"] == res["arrow"]
    assert list(res["native"]["stats_sub"]) == ["sigma"]
    monkeypatch.setenv("MCMC_REF_HIP_READER", "native")
    both = reference.summaries_for_models(names, store=st)
    assert both == res["native"]["summ"]
    # and the packaged goldens of the reference (meta.json diagnostics restated in tests/golden/models/*.json)
    for name in names:
        _, params, rec = load_model(name)
        for p in params:
            for k in ("rhat", "ess_bulk", "ess_tail"):
                assert both[name][p][k] == pytest.approx(rec["meta_diagnostics"][p][k], rel=1e-6)
    with pytest.raises(KeyError):
        reference.stats("radon_pooled", params=["nope"], store=st)


def test_corrupted_files_fail_cleanly_or_decode_in_bounds(ctx):
    """Random byte damage anywhere in a file (headers, Snappy streams, run headers, dictionary indices): the call
    either reports an error or returns arrays of the right shape -- never a crash, a hang or an out-of-bounds
    access (every device-side read and write is bounded by sizes the host validated)."""
    from mcmc_ref_hip._ffi import McrError
    from mcmc_ref_hip.parquet import read_columns
    rng = np.random.default_rng(99)
    n = 4000
    t = pa.table({"chain": np.repeat(np.arange(4), n // 4), "draw": np.tile(np.arange(n // 4), 4),
                  "x": rng.normal(size=n), "ties": np.round(rng.normal(size=n), 1),
                  "ramp": np.arange(n, dtype=np.float64)})
    outcomes = {"ok": 0, "error": 0}
    for kw in (dict(), dict(use_dictionary=False), dict(compression="none"), dict(data_page_version="2.0")):
        img = image(t, **kw)
        flen = int.from_bytes(img[-8:-4], "little")
        body = len(img) - 8 - flen
        for _ in range(60):
            b = bytearray(img)
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(4, body))] = int(rng.integers(0, 256))      # page headers and payloads
            try:
                got = read_columns(ctx, bytes(b))
                assert all(v.shape == (n,) for v in got.values())
                outcomes["ok"] += 1
            except McrError:
                outcomes["error"] += 1
    assert outcomes["error"] > 0 and outcomes["ok"] > 0, outcomes
    check_roundtrip(ctx, image(t))                             # the context is still healthy


def test_column_subset_of_a_file_with_an_inflated_page_is_rejected(ctx):
    """with a column subset only that column's chunk is staged; a page claiming more bytes than its chunk
    holds must be refused before any kernel reads past the staging buffer."""
    from mcmc_ref_hip._ffi import McrError
    from mcmc_ref_hip.parquet import read_columns
    from test_parquet_cpu import inflated_last_page_image
    with pytest.raises(McrError, match="extends beyond its column chunk"):
        read_columns(ctx, inflated_last_page_image(), columns=["a"])
    check_roundtrip(ctx, image(pa.table({"a": np.arange(1000.0), "b": np.arange(1000.0) + 0.5})))


def test_random_tables_and_writer_options(ctx):
    """Randomised schemas (column count, physical types, nullability flags, value distributions) and writer options
    against pyarrow's decode."""
    rng = np.random.default_rng(2024)
    for case in range(40):
        n = int(rng.integers(1, 20000))
        cols, fields = {}, []
        for j in range(int(rng.integers(1, 9))):
            kind = rng.integers(0, 7)
            if kind == 0:
                v = rng.normal(size=n) * 10.0 ** rng.integers(-6, 6)
            elif kind == 1:
                v = np.round(rng.normal(size=n), int(rng.integers(0, 3)))          # ties
            elif kind == 2:
                v = np.repeat(rng.normal(size=n // 97 + 1), 97)[:n]                # long runs
            elif kind == 3:
                v = rng.normal(size=n).astype(np.float32)
            elif kind == 4:
                v = rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)
            elif kind == 5:
                v = rng.integers(-2**62, 2**62, n)
            else:
                v = (np.arange(n) % int(rng.integers(1, 5000))).astype(np.int64)   # periodic: compressible
            name = f"c{j}"
            cols[name] = v
            fields.append(pa.field(name, pa.from_numpy_dtype(v.dtype), nullable=bool(rng.integers(0, 2))))
        t = pa.table(cols, schema=pa.schema(fields))
        kw = dict(compression=str(rng.choice(["snappy", "none"])),
                  use_dictionary=bool(rng.integers(0, 2)),
                  data_page_version=str(rng.choice(["1.0", "2.0"])))
        if rng.integers(0, 2):
            kw["row_group_size"] = int(rng.integers(1, n + 1))
        if rng.integers(0, 2):
            kw["data_page_size"] = int(rng.integers(64, 1 << 16))
            kw["write_batch_size"] = int(rng.integers(1, 1024))
        try:
            check_roundtrip(ctx, image(t, **kw))
        except AssertionError as exc:
            raise AssertionError(f"case {case}: n={n} kw={kw} schema={t.schema}") from exc


def test_one_call_file_batch_equals_the_general_route(ctx, tmp_path):
    """mcr_summarize_files (paths in, statistics out, one C call) against the Python-orchestrated route over the same
    files; files it cannot take (shuffled rows, ragged chains) fall back to that route transparently."""
    from mcmc_ref_hip import parquet
    rng = np.random.default_rng(77)
    paths = []
    for k, (C, N, P) in enumerate([(4, 500, 3), (4, 500, 2), (10, 100, 7), (4, 500, 1), (2, 1000, 4)]):
        cols = {"chain": np.repeat(np.arange(C), N), "draw": np.tile(np.arange(N), C)}
        for p in range(P):
            cols[f"theta[{p + 1}]"] = rng.normal(size=C * N)
        path = tmp_path / f"m{k}.draws.parquet"
        pq.write_table(pa.table(cols), path)
        paths.append(path)
    for diag, mc in ((True, 2), (False, 4)):
        fast = parquet.summarize_files(ctx, paths, min_chains=mc, diagnostics=diag)            # C entry point
        assert parquet._summarize_paths(ctx, [str(p) for p in paths], mc, [0.05, 0.5, 0.95], diag) == fast
        slow = parquet.summarize_files(ctx, [p.read_bytes() for p in paths], min_chains=mc, diagnostics=diag)
        assert fast == slow
        assert [list(f) for f in fast] == [[f"theta[{p + 1}]" for p in range(P)] for P in (3, 2, 7, 1, 4)]
    with pytest.raises(ValueError, match="require at least 4 chains; got 2"):
        parquet.summarize_files(ctx, paths, min_chains=4)
    # a shuffled file and a ragged one: the C call declines (MCR_ELAYOUT), the general route answers
    t = pq.read_table(paths[0]).take(pa.array(rng.permutation(2000)))
    pq.write_table(t, tmp_path / "shuffled.draws.parquet")
    assert parquet._summarize_paths(ctx, [str(tmp_path / "shuffled.draws.parquet")], 4, [0.5], True) is None
    a = parquet.summarize_files(ctx, [tmp_path / "shuffled.draws.parquet"])[0]
    b = parquet.summarize_files(ctx, [paths[0]])[0]
    assert a == b
    rag = pa.table({"chain": [0] * 40 + [1] * 30 + [2] * 35 + [3] * 40, "draw": list(range(40)) + list(range(30)) +
                    list(range(35)) + list(range(40)), "a": rng.normal(size=145)})
    pq.write_table(rag, tmp_path / "ragged.draws.parquet")
    assert parquet._summarize_paths(ctx, [str(tmp_path / "ragged.draws.parquet")], 4, [0.5], True) is None
    assert parquet._summarize_paths(ctx, [str(tmp_path / "ragged.draws.parquet")], 4, [0.5], False) is not None
    r = parquet.summarize_files(ctx, [tmp_path / "ragged.draws.parquet"])[0]
    assert set(r["a"]) == {"mean", "std", "q5", "q50", "q95", "rhat", "ess_bulk", "ess_tail"}
    with pytest.raises(Exception):
        parquet.summarize_files(ctx, [tmp_path / "missing.draws.parquet"])


def test_file_batch_reports_the_broken_file(ctx, tmp_path):
    from mcmc_ref_hip._ffi import McrError
    from mcmc_ref_hip import parquet
    rng = np.random.default_rng(5)
    good = tmp_path / "good.draws.parquet"
    pq.write_table(pa.table({"chain": np.repeat(np.arange(4), 100), "draw": np.tile(np.arange(100), 4),
                             "x": rng.normal(size=400)}), good)
    bad = tmp_path / "bad.draws.parquet"
    img = bytearray(good.read_bytes())
    img[-6] ^= 0xFF                                   # footer length
    bad.write_bytes(bytes(img))
    with pytest.raises(McrError, match="bad.draws.parquet"):
        parquet.summarize_files(ctx, [good, bad])
    (tmp_path / "empty.draws.parquet").write_bytes(b"")
    with pytest.raises(McrError, match="empty"):
        parquet.summarize_files(ctx, [good, tmp_path / "empty.draws.parquet"])
    nochain = tmp_path / "nochain.draws.parquet"
    pq.write_table(pa.table({"x": rng.normal(size=10)}), nochain)
    with pytest.raises(McrError, match="chain / draw"):
        parquet.summarize_files(ctx, [nochain])
    ok = parquet.summarize_files(ctx, [good])[0]
    assert set(ok) == {"x"} and abs(ok["x"]["rhat"] - 1) < 0.1
