"""Host side of the native Parquet ingest (no GPU): footer / schema / page-header parsing of
`mcr_parquet_open` against pyarrow's view of the same files.  Reference call sites this replaces:
src/mcmc_ref/store.py:79-95, src/mcmc_ref/convert.py:61-65."""
from __future__ import annotations

import io

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest

from conftest import GOLDEN
from mcmc_ref_hip._ffi import McrError
from mcmc_ref_hip.parquet import ParquetFile

PHYS = {"INT32": 1, "INT64": 2, "FLOAT": 4, "DOUBLE": 5, "BYTE_ARRAY": 6, "BOOLEAN": 0}


def image(table, **kw) -> bytes:
    buf = io.BytesIO()
    pq.write_table(table, buf, **kw)
    return buf.getvalue()


def check_against_pyarrow(img: bytes):
    f = ParquetFile(img)
    md = pq.ParquetFile(io.BytesIO(img)).metadata
    assert f.num_rows == md.num_rows
    assert f.column_names == [md.schema.column(i).name for i in range(md.num_columns)]
    assert f.column_types == [PHYS[md.schema.column(i).physical_type] for i in range(md.num_columns)]
    pages = f.pages()
    for c in range(md.num_columns):
        mine = [p for p in pages if p["column"] == c]
        data = [p for p in mine if p["kind"] in (0, 3)]
        assert sum(p["num_values"] for p in data) == md.num_rows
        assert [p["first_row"] for p in data] == list(np.cumsum([0] + [p["num_values"] for p in data])[:-1])
        comp = sum(md.row_group(r).column(c).total_compressed_size for r in range(md.num_row_groups))
        # payloads + headers = the chunk: payload bytes alone are a bit less
        assert 0 < sum(p["compressed_size"] for p in mine) <= comp
        n_dict = sum(1 for r in range(md.num_row_groups) if md.row_group(r).column(c).has_dictionary_page)
        assert sum(1 for p in mine if p["kind"] == 2) == n_dict
        for r in range(md.num_row_groups):
            cm = md.row_group(r).column(c)
            first = cm.dictionary_page_offset if cm.has_dictionary_page else cm.data_page_offset
            assert any(first < p["payload_offset"] <= first + 512 for p in mine)   # just behind the first page header
    f.close()
    return pages


def test_packaged_files_parse():
    for path in sorted((GOLDEN / "parquet").glob("*.parquet")):
        img = path.read_bytes()
        pages = check_against_pyarrow(img)
        assert all(p["codec"] == 1 for p in pages)                  # SNAPPY
        assert {p["encoding"] for p in pages if p["kind"] == 0} == {8}   # RLE_DICTIONARY
        with ParquetFile(path) as f:                                # path / mmap form
            assert f.num_rows == 10000 and f.column_names[:2] == ["chain", "draw"]


@pytest.mark.parametrize("kw", [
    dict(),
    dict(compression="none"),
    dict(use_dictionary=False),
    dict(data_page_version="2.0"),
    dict(row_group_size=700),
    dict(data_page_size=2048, write_batch_size=64),
    dict(compression="none", use_dictionary=False, row_group_size=1000, data_page_version="2.0"),
])
def test_writer_options_parse(kw):
    rng = np.random.default_rng(5)
    n = 5000
    t = pa.table({"chain": np.repeat(np.arange(4), n // 4), "draw": np.tile(np.arange(n // 4), 4),
                  "x": rng.normal(size=n), "y": np.round(rng.normal(size=n), 1),
                  "f": rng.normal(size=n).astype(np.float32), "i": rng.integers(-5, 5, n).astype(np.int32),
                  "s": pa.array([str(i % 7) for i in range(n)])})
    check_against_pyarrow(image(t, **kw))


def test_required_columns_and_empty_table():
    schema = pa.schema([pa.field("chain", pa.int64(), nullable=False), pa.field("x", pa.float64(), nullable=False)])
    check_against_pyarrow(image(pa.table({"chain": np.arange(10), "x": np.arange(10.0)}, schema=schema)))
    f = ParquetFile(image(pa.table({"chain": pa.array([], pa.int64()), "x": pa.array([], pa.float64())})))
    assert f.num_rows == 0 and f.column_names == ["chain", "x"]


def test_rejects_what_it_cannot_read():
    img = image(pa.table({"x": np.arange(100.0)}))
    with pytest.raises(McrError, match="magic"):
        ParquetFile(b"not a parquet file at all")
    with pytest.raises(McrError, match="magic"):
        ParquetFile(img[:-1])
    bad = bytearray(img)
    bad[-8:-4] = (len(img) * 2).to_bytes(4, "little")
    with pytest.raises(McrError, match="footer length"):
        ParquetFile(bytes(bad))
    with pytest.raises(McrError, match="nested"):
        ParquetFile(image(pa.table({"l": pa.array([[1.0, 2.0], [3.0]])})))
    # a footer cut in the middle must fail cleanly, never crash
    for cut in (20, 40, 80):
        flen = int.from_bytes(img[-8:-4], "little")
        trunc = img[:len(img) - 8 - flen] + img[len(img) - 8 - flen:len(img) - 8 - cut] + \
            (flen - cut).to_bytes(4, "little") + b"PAR1"
        with pytest.raises(McrError):
            ParquetFile(trunc)


def test_fuzzed_footers_never_crash():
    rng = np.random.default_rng(0)
    img = bytearray(image(pa.table({"chain": np.arange(50), "x": np.arange(50.0)})))
    flen = int.from_bytes(img[-8:-4], "little")
    lo = len(img) - 8 - flen
    for _ in range(300):
        b = bytearray(img)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(lo, len(img) - 8))] = int(rng.integers(0, 256))
        try:
            ParquetFile(bytes(b)).close()
        except McrError:
            pass


def inflated_last_page_image() -> bytes:
    """Two uncompressed PLAIN columns of 1000 doubles; the compressed_page_size in the page header of column `a`
    is raised to 8191 (same varint length), so its page runs past its column chunk, into `b`."""
    img = bytearray(image(pa.table({"a": np.arange(1000.0), "b": np.arange(1000.0) + 0.5}), compression="none",
                          use_dictionary=False, write_statistics=False))
    at = pq.ParquetFile(io.BytesIO(bytes(img))).metadata.row_group(0).column(0).data_page_offset
    # page header: 15 00 (DATA_PAGE)  15 vv (uncompressed_page_size)  15 vv (compressed_page_size), 2-byte varints
    assert img[at:at + 3] == b"\x15\x00\x15" and img[at + 5] == 0x15 and img[at + 3:at + 5] == img[at + 6:at + 8]
    assert img[at + 3] & 0x80 and not img[at + 4] & 0x80
    img[at + 6:at + 8] = bytes([0xFE, 0x7F])      # varint(zigzag(8191)) > the 8 0xx bytes the chunk holds
    return bytes(img)


def test_page_running_past_its_column_chunk_is_rejected():
    """only the chunks of the requested columns are staged on the device, so a page payload that extends
    beyond its chunk (a damaged compressed_page_size, an understated total_compressed_size) must fail in open()."""
    with pytest.raises(McrError, match="extends beyond its column chunk"):
        ParquetFile(inflated_last_page_image())


def test_thrift_containers_with_huge_counts_fail_fast():
    """A footer whose unknown field is a map<bool,bool> / list with an absurd count must not spin."""
    import time
    img = image(pa.table({"x": np.arange(10.0)}))
    flen = int.from_bytes(img[-8:-4], "little")
    foot = img[len(img) - 8 - flen:len(img) - 8]
    for evil in (bytes([0xFB]) + b"\xff" * 9 + b"\x01" + bytes([0x11]),      # field +15: map, count 2^63-1, <bool,bool>
                 bytes([0xF9, 0xF1]) + b"\xff" * 9 + b"\x01"):                # field +15: list<bool>, count 2^63-1
        new_foot = evil + foot
        b = img[:len(img) - 8 - flen] + new_foot + len(new_foot).to_bytes(4, "little") + b"PAR1"
        t0 = time.perf_counter()
        with pytest.raises(McrError):
            ParquetFile(b)
        assert time.perf_counter() - t0 < 1.0
