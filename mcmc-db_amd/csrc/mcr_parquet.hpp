// mcr_parquet.hpp -- Parquet draws file -> device tensor.
//
// The reference reads its draws with pyarrow (`pq.read_table` / `pq.ParquetFile`, src/mcmc_ref/store.py:79-95,
// src/mcmc_ref/convert.py:61-65) and spends >90 % of an end-to-end `stats` call there once the statistics run on
// the GPU.  This file is the replacement for that step, written against the published Apache Parquet format
// (parquet.thrift; Thrift compact protocol; Snappy format description; RLE / bit-packing hybrid), not against
// pyarrow's sources:
//
//   host  : footer + page headers (Thrift compact protocol) -> a flat page table.  Bytes are never decoded on the host.
//   device: k_pq_snappy  one wavefront per compressed page: raw Snappy -> scratch
//           k_pq_decode  one workgroup per data page: definition levels checked (nulls are rejected), PLAIN or
//                        RLE_DICTIONARY / PLAIN_DICTIONARY values -> out[row] as f64 or i64
//
// Supported (everything the packaged corpus and pyarrow's default writer produce for flat numeric tables):
// flat schemas, REQUIRED / OPTIONAL columns without nulls, DOUBLE / FLOAT / INT32 / INT64, UNCOMPRESSED / SNAPPY,
// data pages v1 and v2, dictionary fallback to PLAIN inside a column chunk, any number of row groups and pages.
// Everything else fails the call with a message naming the feature -- never a partial answer.
#pragma once
#include "mcr_device.hpp"

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace mcr {
namespace pq {

// ---- format constants (parquet.thrift) ---------------------------------------------------------
enum : int { T_BOOLEAN = 0, T_INT32 = 1, T_INT64 = 2, T_INT96 = 3, T_FLOAT = 4, T_DOUBLE = 5, T_BYTE_ARRAY = 6, T_FLBA = 7 };
enum : int { ENC_PLAIN = 0, ENC_PLAIN_DICT = 2, ENC_RLE = 3, ENC_BIT_PACKED = 4, ENC_RLE_DICT = 8 };
enum : int { CODEC_NONE = 0, CODEC_SNAPPY = 1 };
enum : int { PAGE_DATA = 0, PAGE_INDEX = 1, PAGE_DICT = 2, PAGE_DATA_V2 = 3 };

// device-side error codes (first error wins; reported with the page index)
enum : int { PQE_OK = 0, PQE_SNAPPY = 1, PQE_NULLS = 2, PQE_LEVELS = 3, PQE_RUNS = 4, PQE_DICT_INDEX = 5, PQE_SHORT = 6 };

// One page as the kernels see it.  Offsets are bytes into the device staging buffer (`stage`, the uploaded column
// chunks) or the decompression scratch.
struct PageDev {
    u64 src_off;        // payload in stage (for v2: start of the level bytes)
    u64 dst_off;        // decompressed bytes in scratch (only when compressed)
    u32 comp_size;      // bytes to decompress (v2: without the levels)
    u32 uncomp_size;    // size after decompression (v2: without the levels)
    u32 num_values;
    u32 lvl_bytes;      // v2: rep + def level bytes in front of the values (never compressed)
    u32 def_bytes;      // v2: def level bytes (the last `def_bytes` of lvl_bytes)
    int dict_page;      // data page: index of its dictionary page in the table, -1 = none
    u32 dict_count;     // data page: entries of that dictionary
    unsigned char kind, encoding, compressed, phys_type, max_def, out_kind, pad0, pad1;
    u64 out_off;        // data page: first output element
    void* out;          // data page: column output (8-byte elements)
};

// ---- Thrift compact protocol reader (host) -----------------------------------------------------
struct Thrift {
    const unsigned char* p; const unsigned char* end; bool ok = true;
    Thrift(const void* b, size_t n) : p((const unsigned char*)b), end((const unsigned char*)b + n) {}
    unsigned char byte() { if (p >= end) { ok = false; return 0; } return *p++; }
    u64 varint() { u64 v = 0; for (int s = 0; s < 64; s += 7) { const unsigned char b = byte(); v |= (u64)(b & 0x7F) << s; if (!(b & 0x80)) return v; } ok = false; return v; }
    i64 zigzag() { const u64 v = varint(); return (i64)(v >> 1) ^ -(i64)(v & 1); }
    bool binary(const unsigned char** s, size_t* n) { const u64 l = varint(); if (!ok || l > (u64)(end - p)) { ok = false; return false; } *s = p; *n = (size_t)l; p += l; return true; }
    // field header: returns type (0 = stop), updates *id
    int field(int* id) {
        const unsigned char b = byte();
        if (!ok || b == 0) return 0;
        const int delta = b >> 4, type = b & 15;
        if (delta) *id += delta; else *id = (int)zigzag();
        return type;
    }
    void list_header(int* etype, u64* n) { const unsigned char b = byte(); *etype = b & 15; *n = b >> 4; if (*n == 15) *n = varint(); }
    void skip(int type, int depth = 0) {
        if (!ok || depth > 32) { ok = false; return; }
        switch (type) {
            case 1: case 2: break;                       // bool carried in the field header
            case 3: byte(); break;
            case 4: case 5: case 6: varint(); break;
            case 7: if (end - p < 8) ok = false; else p += 8; break;
            case 8: { const unsigned char* s; size_t n; binary(&s, &n); break; }
            // every element occupies at least one byte (a bool inside a container is one byte, unlike a bool field,
            // which lives in its header): a count larger than what is left is a damaged footer, not a long loop
            case 9: case 10: { int et; u64 n; list_header(&et, &n); if (n > (u64)(end - p)) { ok = false; break; }
                               for (u64 i = 0; i < n && ok; ++i) { if (et == 1 || et == 2) byte(); else skip(et, depth + 1); } break; }
            case 11: { const u64 n = varint(); if (n > (u64)(end - p)) { ok = false; break; }
                       if (n) { const unsigned char kv = byte(); const int kt = kv >> 4, vt = kv & 15;
                                for (u64 i = 0; i < n && ok; ++i) {
                                    if (kt == 1 || kt == 2) byte(); else skip(kt, depth + 1);
                                    if (vt == 1 || vt == 2) byte(); else skip(vt, depth + 1); } } break; }
            case 12: { int id = 0; for (;;) { const int t = field(&id); if (!t || !ok) break; skip(t, depth + 1); } break; }
            default: ok = false;
        }
    }
};

struct Column { std::string name; int type = -1; int max_def = 0; bool leaf = true; };

struct Page {              // host view of a page
    int col; int kind; int encoding; int codec;
    u64 payload_off;       // file offset of the bytes after the page header
    u32 comp_size, uncomp_size, num_values;
    u32 rep_bytes = 0, def_bytes = 0; bool v2_compressed = true;
    u64 row_off = 0;       // data pages: first row of the page within the column
    int dict = -1;         // data pages: index (into `pages`) of the chunk's dictionary page
    u32 dict_count = 0;
};

struct Chunk { int col; u64 start, end; int first_page, n_pages; };

struct File {
    const unsigned char* bytes = nullptr; size_t len = 0;
    i64 num_rows = 0;
    std::vector<Column> cols;
    std::vector<Page> pages;
    std::vector<Chunk> chunks;          // per (row group, column), file order
    std::string created_by;
    std::string error;
};

inline bool fail(File& f, const std::string& m) { f.error = m; return false; }

inline bool parse_schema(Thrift& t, File& f)
{
    int et; u64 n; t.list_header(&et, &n);
    if (!t.ok || et != 12 || n < 1) return fail(f, "bad schema list");
    for (u64 i = 0; i < n; ++i) {
        int id = 0, type = -1, rep = 0, nchild = -1; std::string name;
        for (;;) {
            const int ft = t.field(&id);
            if (!ft || !t.ok) break;
            if (id == 1 && ft == 5) type = (int)t.zigzag();
            else if (id == 3 && ft == 5) rep = (int)t.zigzag();
            else if (id == 4 && ft == 8) { const unsigned char* s; size_t l; if (t.binary(&s, &l)) name.assign((const char*)s, l); }
            else if (id == 5 && ft == 5) nchild = (int)t.zigzag();
            else t.skip(ft);
        }
        if (!t.ok) return fail(f, "truncated schema element");
        if (i == 0) { if ((u64)nchild != n - 1) return fail(f, "nested schemas are not supported (only flat tables)"); continue; }
        if (nchild > 0) return fail(f, "nested schemas are not supported (only flat tables)");
        if (rep == 2) return fail(f, "REPEATED column '" + name + "' is not supported");
        Column c; c.name = name; c.type = type; c.max_def = rep == 1 ? 1 : 0;
        f.cols.push_back(c);
    }
    return true;
}

struct ChunkMeta { int type = -1, codec = 0; i64 num_values = 0, data_off = -1, dict_off = -1, comp_total = 0; };

inline bool parse_column_meta(Thrift& t, ChunkMeta& m)
{
    int id = 0;
    for (;;) {
        const int ft = t.field(&id);
        if (!ft || !t.ok) break;
        if (id == 1 && ft == 5) m.type = (int)t.zigzag();
        else if (id == 4 && ft == 5) m.codec = (int)t.zigzag();
        else if (id == 5 && ft == 6) m.num_values = t.zigzag();
        else if (id == 7 && ft == 6) m.comp_total = t.zigzag();
        else if (id == 9 && ft == 6) m.data_off = t.zigzag();
        else if (id == 11 && ft == 6) m.dict_off = t.zigzag();
        else t.skip(ft);
    }
    return t.ok;
}

// Page header at `off`; fills pg (payload_off = first byte after the header).
inline bool parse_page_header(File& f, u64 off, Page& pg)
{
    if (off >= f.len) return fail(f, "page header beyond the end of the file");
    Thrift t(f.bytes + off, f.len - off);
    int id = 0; i64 type = -1, usz = -1, csz = -1;
    pg.num_values = 0; pg.encoding = -1;
    for (;;) {
        const int ft = t.field(&id);
        if (!ft || !t.ok) break;
        if (id == 1 && ft == 5) type = t.zigzag();
        else if (id == 2 && ft == 5) usz = t.zigzag();
        else if (id == 3 && ft == 5) csz = t.zigzag();
        else if ((id == 5 || id == 7 || id == 8) && ft == 12) {
            int sid = 0;
            for (;;) {
                const int st = t.field(&sid);
                if (!st || !t.ok) break;
                if (id == 5 || id == 7) {                 // DataPageHeader / DictionaryPageHeader
                    if (sid == 1 && st == 5) pg.num_values = (u32)t.zigzag();
                    else if (sid == 2 && st == 5) pg.encoding = (int)t.zigzag();
                    else t.skip(st);
                } else {                                  // DataPageHeaderV2
                    if (sid == 1 && st == 5) pg.num_values = (u32)t.zigzag();
                    else if (sid == 4 && st == 5) pg.encoding = (int)t.zigzag();
                    else if (sid == 5 && st == 5) pg.def_bytes = (u32)t.zigzag();
                    else if (sid == 6 && st == 5) pg.rep_bytes = (u32)t.zigzag();
                    else if (sid == 7 && (st == 1 || st == 2)) pg.v2_compressed = (st == 1);
                    else t.skip(st);
                }
            }
        } else t.skip(ft);
    }
    if (!t.ok || type < 0 || usz < 0 || csz < 0) return fail(f, "truncated or malformed page header");
    if (usz > (i64)(1u << 30) || csz > (i64)(1u << 30)) return fail(f, "page larger than 1 GiB");
    pg.kind = (int)type; pg.uncomp_size = (u32)usz; pg.comp_size = (u32)csz;
    pg.payload_off = off + (u64)(t.p - (f.bytes + off));
    if (pg.payload_off + pg.comp_size > f.len) return fail(f, "page payload beyond the end of the file");
    return true;
}

// Parses footer, schema, row groups and walks the page headers of every column chunk.
inline bool open(File& f, const void* bytes, size_t len)
{
    f.bytes = (const unsigned char*)bytes; f.len = len;
    if (len < 12 || memcmp(f.bytes, "PAR1", 4) != 0 || memcmp(f.bytes + len - 4, "PAR1", 4) != 0)
        return fail(f, len >= 4 && memcmp(f.bytes + len - 4, "PARE", 4) == 0 ? "encrypted Parquet files are not supported"
                                                                            : "not a Parquet file (magic bytes)");
    u32 flen; memcpy(&flen, f.bytes + len - 8, 4);
    if ((u64)flen + 12 > len) return fail(f, "footer length out of range");
    Thrift t(f.bytes + len - 8 - flen, flen);
    int id = 0; bool have_schema = false;
    struct RG { std::vector<ChunkMeta> cm; i64 rows = 0; };
    std::vector<RG> rgs;
    for (;;) {
        const int ft = t.field(&id);
        if (!ft || !t.ok) break;
        if (id == 2 && ft == 9) { if (!parse_schema(t, f)) return false; have_schema = true; }
        else if (id == 3 && ft == 6) f.num_rows = t.zigzag();
        else if (id == 4 && ft == 9) {
            int et; u64 n; t.list_header(&et, &n);
            for (u64 r = 0; r < n && t.ok; ++r) {
                RG rg; int rid = 0;
                for (;;) {
                    const int rt = t.field(&rid);
                    if (!rt || !t.ok) break;
                    if (rid == 1 && rt == 9) {
                        int cet; u64 cn; t.list_header(&cet, &cn);
                        for (u64 c = 0; c < cn && t.ok; ++c) {
                            ChunkMeta m; int cid = 0; bool external = false;
                            for (;;) {
                                const int ct = t.field(&cid);
                                if (!ct || !t.ok) break;
                                if (cid == 1 && ct == 8) { const unsigned char* s; size_t l; t.binary(&s, &l); external = l > 0; }
                                else if (cid == 3 && ct == 12) { if (!parse_column_meta(t, m)) return fail(f, "truncated column metadata"); }
                                else t.skip(ct);
                            }
                            if (external) return fail(f, "column chunks in external files are not supported");
                            rg.cm.push_back(m);
                        }
                    } else if (rid == 3 && rt == 6) rg.rows = t.zigzag();
                    else t.skip(rt);
                }
                rgs.push_back(rg);
            }
        } else if (id == 6 && ft == 8) { const unsigned char* s; size_t l; if (t.binary(&s, &l)) f.created_by.assign((const char*)s, l); }
        else t.skip(ft);
    }
    if (!t.ok || !have_schema) return fail(f, "truncated or malformed footer");
    if (f.num_rows < 0 || f.num_rows >= (i64)0x7FFFFFFF) return fail(f, "row count out of range");
    std::vector<u64> rows_done(f.cols.size(), 0);
    for (const RG& rg : rgs) {
        if (rg.cm.size() != f.cols.size()) return fail(f, "row group column count differs from the schema");
        for (size_t c = 0; c < rg.cm.size(); ++c) {
            const ChunkMeta& m = rg.cm[c];
            if (m.type != f.cols[c].type) return fail(f, "column chunk type differs from the schema");
            if (m.data_off < 0 || m.comp_total < 0) return fail(f, "column chunk without offsets");
            u64 off = (m.dict_off > 0 && m.dict_off < m.data_off) ? (u64)m.dict_off : (u64)m.data_off;
            const u64 end = off + (u64)m.comp_total;
            if (end > len) return fail(f, "column chunk beyond the end of the file");
            Chunk ch; ch.col = (int)c; ch.start = off; ch.end = end; ch.first_page = (int)f.pages.size(); ch.n_pages = 0;
            int dict = -1; u32 dict_count = 0; i64 seen = 0;
            while (off < end && seen < m.num_values) {
                Page pg; pg.col = (int)c; pg.codec = m.codec;
                if (!parse_page_header(f, off, pg)) return false;
                off = pg.payload_off + pg.comp_size;
                // only the chunks of the requested columns are uploaded: a page that runs past its chunk would make the
                // decode kernels read outside the staged bytes
                if (off > end) return fail(f, "page of column '" + f.cols[c].name + "' extends beyond its column chunk");
                if (pg.kind == PAGE_INDEX) continue;
                if (pg.kind == PAGE_DICT) { dict = (int)f.pages.size(); dict_count = pg.num_values; }
                else if (pg.kind == PAGE_DATA || pg.kind == PAGE_DATA_V2) {
                    pg.row_off = rows_done[c] + (u64)seen; pg.dict = dict; pg.dict_count = dict_count;
                    seen += pg.num_values;
                    if (pg.kind == PAGE_DATA_V2 && (u64)pg.rep_bytes + pg.def_bytes > pg.comp_size) return fail(f, "v2 level bytes exceed the page");
                } else return fail(f, "unknown page type " + std::to_string(pg.kind));
                f.pages.push_back(pg); ++ch.n_pages;
            }
            if (seen != m.num_values) return fail(f, "pages of column '" + f.cols[c].name + "' do not add up to its value count");
            rows_done[c] += (u64)seen;
            f.chunks.push_back(ch);
        }
    }
    for (size_t c = 0; c < f.cols.size(); ++c)
        if ((i64)rows_done[c] != f.num_rows) return fail(f, "column '" + f.cols[c].name + "' does not have num_rows values");
    return true;
}

// ---- device helpers ----------------------------------------------------------------------------
// Loads from arbitrarily aligned addresses as aligned dwords + v_alignbit (no reliance on the memory system's
// unaligned mode).  May touch up to 3 bytes in front of and 7 bytes behind the value: stage and scratch buffers are
// padded accordingly.
__device__ __forceinline__ u64 ld64u(const unsigned char* p)
{
    const uintptr_t a = (uintptr_t)p;
    const u32* w = (const u32*)(a & ~(uintptr_t)3);
    const u32 sh = (u32)(a & 3) * 8;
    const u32 w0 = w[0], w1 = w[1], w2 = w[2];
    return (u64)__funnelshift_r(w0, w1, sh) | ((u64)__funnelshift_r(w1, w2, sh) << 32);
}
__device__ __forceinline__ u32 ld32u(const unsigned char* p)
{
    const uintptr_t a = (uintptr_t)p;
    const u32* w = (const u32*)(a & ~(uintptr_t)3);
    const u32 sh = (u32)(a & 3) * 8;
    return __funnelshift_r(w[0], w[1], sh);
}

__device__ __forceinline__ void pq_error(int* err, int code, int page)
{
    if (atomicCAS(err, 0, code) == 0) err[1] = page;
}

// ---- Snappy (raw format) -----------------------------------------------------------------------
// One wavefront per page.  A Snappy stream is a chain of elements whose boundaries are only known by walking it, so
// the kernel works in batches of 64 input bytes:
//   1. every lane decodes "the element that would start at my byte" (type, header size, lengths, offset);
//   2. a scalar walk over those candidates (v_readlane + s_add per element, no memory access) marks the real
//      element starts; a wave scan of their output lengths gives every element its output position;
//   3. all literal bytes of the batch go to the output in ONE step (each lane owns one input byte);
//   4. the copies run in stream order, all lanes sharing one copy.
// The output is assembled in an LDS ring (the last kRing bytes) and flushed to global memory in 8 KB pieces with
// 16-byte stores; back references (always within 64 KB in practice, almost always within a few KB) read the ring,
// farther ones flush and read the output back behind a workgroup fence.  Literals that do not fit the window go
// alone: up to 64 bytes through LDS, longer ones from the input straight to global memory at 16 bytes per lane.
constexpr int kRing = 32768, kInWin = 4096, kFlush = 8192;
constexpr int kBatchOut = 64 * 64 + 64;      // most output bytes one batch can produce

// Orders LDS traffic between the lanes of ONE wavefront: LDS executes a wave's instructions in issue order, so only
// the compiler has to be kept from moving accesses across this point.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(64) void k_pq_snappy(const unsigned char* __restrict__ stage, unsigned char* scratch,
                                                  const PageDev* __restrict__ pages, const int* __restrict__ list,
                                                  int* err)
{
    __shared__ __attribute__((aligned(16))) unsigned char ring[kRing];
    __shared__ __attribute__((aligned(16))) unsigned char inb[kInWin];
    const int pi = list[blockIdx.x];
    const PageDev pg = pages[pi];
    const int lane = threadIdx.x;
    const unsigned char* src = stage + pg.src_off + pg.lvl_bytes;
    unsigned char* dst = scratch + pg.dst_off;          // 16-byte aligned
    const u32 n_in = pg.comp_size, n_out = pg.uncomp_size;
    constexpr u32 RM = kRing - 1;
    auto uni = [](u32 x) { return (u32)__builtin_amdgcn_readfirstlane((int)x); };   // wave-uniform value -> SGPR
    auto rl = [](u32 v, u32 l) { return (u32)__builtin_amdgcn_readlane((int)v, (int)l); };
    // preamble: uncompressed length
    u32 ip = 0, ulen = 0;
    for (int s = 0; s < 35 && ip < n_in; s += 7) { const u32 b = src[ip++]; ulen |= (b & 0x7F) << s; if (!(b & 0x80)) break; }
    ip = uni(ip);
    if (uni(ulen) != n_out) { if (lane == 0) pq_error(err, PQE_SNAPPY, pi); return; }
    i64 win = -(i64)kInWin * 2;          // input offset of inb[0]; the LDS window is [win, win + kInWin)
    const uintptr_t src_a = (uintptr_t)src;
    u32 op = 0, flushed = 0;             // output bytes produced / already in global memory
    bool bad = false;

    auto flush = [&](u32 upto) {         // ring -> dst for [flushed, upto); upto - flushed <= kRing
        wave_sync();
        u32 a = flushed;
        const u32 h = min(upto - a, (16u - (a & 15)) & 15);
        if (lane < (int)h) dst[a + lane] = ring[(a + lane) & RM];
        a += h;
        const u32 nv = (upto - a) >> 4;
        for (u32 v = lane; v < nv; v += 64) *(uint4*)(dst + a + 16 * v) = *(const uint4*)(ring + ((a + 16 * v) & RM));
        a += 16 * nv;
        if (lane < (int)(upto - a)) dst[a + lane] = ring[(a + lane) & RM];
        flushed = upto;
    };

    while (ip < n_in) {
        // LDS input window covers [ip, ip + 136): 64 candidate headers of 5 bytes, or a header + a 64-byte literal
        if ((i64)ip < win || (i64)ip + 136 > win + kInWin) {
            win = (i64)ip - (i64)((src_a + ip) & 15);                    // 16-byte aligned global loads
            wave_sync();
            for (int o = lane * 16; o < kInWin; o += 64 * 16)
                *(uint4*)(inb + o) = *(const uint4*)(src + win + o);     // reads past n_in stay inside the padded stage
            wave_sync();
        }
        // 1. the element that would start at byte ip + lane
        const unsigned char* q = inb + ((i64)ip - win) + lane;
        const u32 c0 = q[0], c1 = q[1], c2 = q[2], c3 = q[3], c4 = q[4];
        const u32 type = c0 & 3;
        u32 hdr, olen, off = 0;
        if (type == 0) {
            const u32 L = c0 >> 2;
            if (L < 60) { hdr = 1; olen = L + 1; }
            else {
                const u32 nb = L - 59;
                u32 v = c1 | (c2 << 8) | (c3 << 16) | (c4 << 24);
                if (nb < 4) v &= (1u << (8 * nb)) - 1;
                hdr = 1 + nb; olen = v >= 0x7FFFFFF0u ? 0x7FFFFFF0u : v + 1;
            }
        } else if (type == 1) { hdr = 2; olen = 4 + ((c0 >> 2) & 7); off = ((c0 >> 5) << 8) | c1; }
        else if (type == 2) { hdr = 3; olen = (c0 >> 2) + 1; off = c1 | (c2 << 8); }
        else { hdr = 5; olen = (c0 >> 2) + 1; off = c1 | (c2 << 8) | (c3 << 16) | (c4 << 24); }
        const u32 ilen = type == 0 ? hdr + olen : hdr;                   // input bytes of the element
        // 2. walk the chain of real element starts inside the window
        u64 starts = 0; u32 pos = 0;
        const u32 avail = min(64u, n_in - ip);
        while (pos < avail) {
            const u32 il = rl(ilen, pos);
            if (pos + il > 64) break;                                    // a literal that leaves the window goes alone
            starts |= 1ull << pos; pos += il;
        }
        if (starts == 0) {                                               // lone literal at ip
            const u32 len = rl(olen, 0);
            ip += rl(hdr, 0);
            if (ip > n_in || len > n_in - ip || len > n_out - op) { bad = true; break; }
            if (len <= 64) {                                             // LDS window -> ring
                if (lane < (int)len) ring[(op + lane) & RM] = inb[(i64)ip - win + lane];
            } else {                                                     // input -> global + ring
                flush(op);
                const unsigned char* s = src + ip;
                const u32 h = min(len, (16u - (op & 15)) & 15);
                if (lane < (int)h) { const unsigned char b = s[lane]; ring[(op + lane) & RM] = b; dst[op + lane] = b; }
                const unsigned char* s2 = s + h;
                const u32 nv = (len - h) >> 4, o2 = op + h;
                const uintptr_t a2 = (uintptr_t)s2;
                const u32* w = (const u32*)(a2 & ~(uintptr_t)3);
                const u32 sh = (u32)(a2 & 3) * 8;
                for (u32 v = lane; v < nv; v += 64) {
                    const u32 w0 = w[4 * v], w1 = w[4 * v + 1], w2 = w[4 * v + 2], w3 = w[4 * v + 3], w4 = w[4 * v + 4];
                    uint4 x;
                    x.x = __funnelshift_r(w0, w1, sh); x.y = __funnelshift_r(w1, w2, sh);
                    x.z = __funnelshift_r(w2, w3, sh); x.w = __funnelshift_r(w3, w4, sh);
                    *(uint4*)(dst + o2 + 16 * v) = x;
                    *(uint4*)(ring + ((o2 + 16 * v) & RM)) = x;
                }
                const u32 tl = (len - h) & 15, o3 = o2 + 16 * nv;
                if (lane < (int)tl) { const unsigned char b = s2[16 * nv + lane]; ring[(o3 + lane) & RM] = b; dst[o3 + lane] = b; }
                flushed = op + len;
            }
            ip += len; op += len;
        } else {
            const bool isstart = (starts >> lane) & 1;
            // output position of every element: exclusive scan of the output lengths over the start lanes
            u32 incl = isstart ? olen : 0;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const u32 t = (u32)__shfl_up((int)incl, o, kWave); if (lane >= o) incl += t; }
            const u32 total = rl(incl, 63);
            const u32 opos = op + incl - (isstart ? olen : 0);
            // validation of the whole batch (in-window elements are small: no overflow in these sums)
            const bool lb = isstart && (opos + olen > n_out || ip + lane + ilen > n_in ||
                                        (type != 0 && (off == 0 || off > opos)));
            if (__ballot(lb)) { bad = true; break; }
            // 3. literal bodies: lane l belongs to the last start s <= l; it is payload if l >= s + hdr_s
            const u64 below = starts & (lane == 63 ? ~0ull : ((2ull << lane) - 1));
            const int s_l = 63 - __clzll((long long)below);              // starts has bit 0 set
            const u32 s_type = (u32)__shfl((int)type, s_l, kWave), s_hdr = (u32)__shfl((int)hdr, s_l, kWave),
                      s_opos = (u32)__shfl((int)opos, s_l, kWave);
            if ((u32)lane < pos && s_type == 0 && (u32)lane >= (u32)s_l + s_hdr)
                ring[(s_opos + (u32)lane - (u32)s_l - s_hdr) & RM] = (unsigned char)c0;
            wave_sync();
            u64 cm = __ballot(isstart && type != 0);
            // 4a. copies whose source lies wholly IN FRONT of this batch's output (and inside the ring) read bytes that are
            //     final and write ranges of their own: they do not depend on one another and go all at once, one byte per
            //     copy and step (round 4).  Stray 4-byte matches in otherwise incompressible pages are of this kind -- the
            //     `Intercept` dictionary page of the packaged `diamonds` model is 1 868 of them between literals of a few
            //     bytes, at offsets anywhere in Snappy's 64 KB window: one after the other, two LDS round trips each, that
            //     page took 0.81 ms against 45 us for its literal-only neighbours, and the launch waits for its slowest page
            //     (tools/pq_classes.py).
            {
                const bool indep = isstart && type != 0 && opos - off + olen <= op && off <= (u32)(kRing - kBatchOut);
                const u64 im = __ballot(indep);
                if (im & (im - 1)) {                                         // two or more
                    const u32 mylen = indep ? olen : 0u;
                    for (u32 i2 = 0;; ++i2) {
                        const bool act = i2 < mylen;
                        if (!__ballot(act)) break;
                        if (act) ring[(opos + i2) & RM] = ring[(opos - off + i2) & RM];
                    }
                    cm &= ~im;
                    wave_sync();
                }
            }
            // 4b. the others, in stream order
            while (cm) {
                const u32 s_c = (u32)__builtin_ctzll(cm);
                cm &= cm - 1;
                const u32 len = rl(olen, s_c), of = rl(off, s_c), o = rl(opos, s_c);
                u32 k = (u32)lane;
                if (of < len) k = (u32)lane % of;                        // overlapping copies repeat the last `of` bytes (scalar branch)
                unsigned char b = 0;
                if (of <= (u32)(kRing - kBatchOut)) {                     // the ring still holds the source
                    if (lane < (int)len) b = ring[(o - of + k) & RM];
                } else {                                                 // beyond the ring: read the output back
                    flush(o);
                    __threadfence_block();
                    __syncthreads();
                    if (lane < (int)len) b = ((volatile unsigned char*)dst)[o - of + k];
                }
                wave_sync();
                if (lane < (int)len) ring[(o + lane) & RM] = b;
                wave_sync();
            }
            op += total; ip += pos;
        }
        if ((op ^ flushed) >= (u32)kFlush) flush(op & ~(u32)(kFlush - 1));   // crossed an 8 KB boundary
        wave_sync();
    }
    if (!bad) flush(op);
    if (bad || op != n_out) { if (lane == 0) pq_error(err, PQE_SNAPPY, pi); }
}

// ---- page decode -------------------------------------------------------------------------------
__device__ __forceinline__ u32 pq_varint(const unsigned char*& p, const unsigned char* end)
{
    u32 v = 0;
    for (int s = 0; s < 35 && p < end; s += 7) { const u32 b = *p++; v |= (b & 0x7F) << s; if (!(b & 0x80)) break; }
    return v;
}

template <int OUT_I64>
__device__ __forceinline__ void pq_store(void* out, u64 i, u64 bits, int phys)
{
    // bits: the raw little-endian value (8 bytes for INT64 / DOUBLE, low 4 for INT32 / FLOAT)
    if (OUT_I64) {
        ((i64*)out)[i] = phys == T_INT64 ? (i64)bits : (i64)(int)(u32)bits;
    } else {
        double v;
        if (phys == T_DOUBLE) v = __longlong_as_double((i64)bits);
        else if (phys == T_FLOAT) v = (double)__uint_as_float((u32)bits);
        else if (phys == T_INT64) v = (double)(i64)bits;
        else v = (double)(int)(u32)bits;
        ((double*)out)[i] = v;
    }
}

// One workgroup per data page.  All threads walk the run headers in lock step (uniform loads); the values of a run are
// spread over the threads.
__global__ __launch_bounds__(256) void k_pq_decode(const unsigned char* __restrict__ stage,
                                                   const unsigned char* __restrict__ scratch,
                                                   const PageDev* __restrict__ pages, const int* __restrict__ list,
                                                   int* err)
{
    const int pi = list[blockIdx.x];
    const PageDev pg = pages[pi];
    const int tid = threadIdx.x;
    const unsigned char* lvl = nullptr; u32 lvl_len = 0;
    const unsigned char* val; const unsigned char* vend;
    if (pg.kind == PAGE_DATA_V2) {
        lvl = stage + pg.src_off + (pg.lvl_bytes - pg.def_bytes); lvl_len = pg.def_bytes;
        val = pg.compressed ? scratch + pg.dst_off : stage + pg.src_off + pg.lvl_bytes;
        vend = val + pg.uncomp_size;
    } else {
        val = pg.compressed ? scratch + pg.dst_off : stage + pg.src_off;
        vend = val + pg.uncomp_size;
        if (pg.max_def) {
            if (vend - val < 4) { if (tid == 0) pq_error(err, PQE_SHORT, pi); return; }
            lvl_len = ld32u(val); lvl = val + 4;
            if (lvl_len > (u32)(vend - lvl)) { if (tid == 0) pq_error(err, PQE_LEVELS, pi); return; }
            val = lvl + lvl_len;
        }
    }
    const u32 nv = pg.num_values;
    // definition levels (bit width 1): every value must be defined
    if (pg.max_def && nv) {
        const unsigned char* p = lvl; const unsigned char* pe = lvl + lvl_len;
        u32 seen = 0; int flag = 0;
        while (seen < nv && p < pe) {
            const u32 h = pq_varint(p, pe);
            if (h & 1) {
                const u32 groups = h >> 1, cnt = min(groups * 8, nv - seen);
                if (groups > (u32)(pe - p)) { flag = PQE_LEVELS; break; }
                for (u32 g = tid; g * 8 < cnt; g += 256) {
                    const u32 m = (cnt - g * 8 >= 8) ? 0xFFu : ((1u << (cnt - g * 8)) - 1);
                    if ((p[g] & m) != m) flag = PQE_NULLS;
                }
                p += groups; seen += cnt;
            } else {
                const u32 cnt = h >> 1;
                if (p >= pe) { flag = PQE_LEVELS; break; }
                if (cnt && (*p & 1) == 0) flag = PQE_NULLS;
                p += 1; seen += min(cnt, nv - seen);
                if (cnt == 0) { flag = PQE_LEVELS; break; }
            }
        }
        if (!flag && seen < nv) flag = PQE_LEVELS;
        if (__syncthreads_or(flag)) { if (flag) pq_error(err, flag, pi); return; }
    }
    const int phys = pg.phys_type;
    const u32 es = (phys == T_INT64 || phys == T_DOUBLE) ? 8 : 4;
    void* out = (char*)pg.out + pg.out_off * 8;
    if (pg.encoding == ENC_PLAIN) {
        if ((u64)nv * es > (u64)(vend - val)) { if (tid == 0) pq_error(err, PQE_SHORT, pi); return; }
        for (u32 i = tid; i < nv; i += 256) {
            const u64 bits = es == 8 ? ld64u(val + (u64)i * 8) : (u64)ld32u(val + (u64)i * 4);
            if (pg.out_kind) pq_store<1>(out, i, bits, phys); else pq_store<0>(out, i, bits, phys);
        }
        return;
    }
    // dictionary indices: 1 byte bit width, then RLE / bit-packed hybrid runs
    const PageDev dp = pages[pg.dict_page];
    const unsigned char* dict = dp.compressed ? scratch + dp.dst_off : stage + dp.src_off;
    const u32 dn = pg.dict_count;
    if (nv == 0) return;
    if (val >= vend) { if (tid == 0) pq_error(err, PQE_SHORT, pi); return; }
    const u32 bw = *val++;
    if (bw > 32) { if (tid == 0) pq_error(err, PQE_RUNS, pi); return; }
    const u64 mask = bw == 32 ? 0xFFFFFFFFull : ((1ull << bw) - 1);
    u32 done = 0; int flag = 0;
    while (done < nv) {
        if (val >= vend) { flag = PQE_RUNS; break; }
        const u32 h = pq_varint(val, vend);
        if (h & 1) {
            const u32 groups = h >> 1;
            const u64 bytes = (u64)groups * bw;
            const u32 cnt = (u32)min((u64)groups * 8, (u64)(nv - done));
            if (groups == 0 || ((u64)cnt * bw + 7) / 8 > (u64)(vend - val)) { flag = PQE_RUNS; break; }   // the last run may be cut short
            for (u32 i = tid; i < cnt; i += 256) {
                const u64 bit = (u64)i * bw;
                const u32 idx = (u32)((ld64u(val + (bit >> 3)) >> (bit & 7)) & mask);
                if (idx >= dn) { flag = PQE_DICT_INDEX; continue; }
                const u64 bits = es == 8 ? ld64u(dict + (u64)idx * 8) : (u64)ld32u(dict + (u64)idx * 4);
                if (pg.out_kind) pq_store<1>(out, done + i, bits, phys); else pq_store<0>(out, done + i, bits, phys);
            }
            val += min(bytes, (u64)(vend - val)); done += cnt;
        } else {
            const u32 cnt0 = h >> 1, nb = (bw + 7) / 8;
            if (cnt0 == 0 || nb > (u32)(vend - val)) { flag = PQE_RUNS; break; }
            u32 idx = 0;
            for (u32 b = 0; b < nb; ++b) idx |= (u32)val[b] << (8 * b);
            val += nb;
            const u32 cnt = min(cnt0, nv - done);
            if (idx >= dn) { flag = PQE_DICT_INDEX; break; }
            const u64 bits = es == 8 ? ld64u(dict + (u64)idx * 8) : (u64)ld32u(dict + (u64)idx * 4);
            for (u32 i = tid; i < cnt; i += 256) {
                if (pg.out_kind) pq_store<1>(out, done + i, bits, phys); else pq_store<0>(out, done + i, bits, phys);
            }
            done += cnt;
        }
    }
    if (__syncthreads_or(flag)) { if (flag) pq_error(err, flag, pi); }
}

// dst[p][k] = src[p][order[k]]  (rows of a long table that are not in (chain, draw) order; `_chains_from_table`,
// src/mcmc_ref/convert.py:150-161)
__global__ __launch_bounds__(256) void k_gather_rows(const double* __restrict__ src, const i64* __restrict__ order,
                                                     i64 M, double* __restrict__ dst)
{
    const i64 k = (i64)blockIdx.x * 256 + threadIdx.x;
    const i64 p = blockIdx.y;
    if (k < M) dst[p * M + k] = src[p * M + order[k]];
}

// The integer bookkeeping of `_chains_from_table` (src/mcmc_ref/convert.py:150-161) for files whose rows are already in
// (chain, draw) order, on the device: one workgroup per file walks its decoded chain / draw id columns and leaves
//   out[4 f + 0] = number of chains (maximal runs of equal chain id), [1] = length of the first run,
//   [2] = 1 if every run has that length, [3] = 1 if the rows are in (chain id ascending, draw ascending) order
// so that only 32 bytes per file come back to the host instead of the id columns (9 MB for the packaged corpus).
// Runs are all `L` long iff every run boundary sits at a multiple of L and there are M / L runs.
struct FileIds { const i64* chain; const i64* draw; i64 M; };
__global__ __launch_bounds__(256) void k_chain_layout(const FileIds* __restrict__ files, i64* __restrict__ out)
{
    __shared__ unsigned long long s_first;
    __shared__ unsigned s_bounds, s_bad, s_uneven;
    const FileIds f = files[blockIdx.x];
    const int tid = threadIdx.x;
    if (tid == 0) { s_first = (unsigned long long)f.M; s_bounds = 0u; s_bad = 0u; s_uneven = 0u; }
    __syncthreads();
    unsigned nb = 0, bad = 0;
    unsigned long long first = (unsigned long long)f.M;
    for (i64 r = 1 + tid; r < f.M; r += 256) {
        const i64 c0 = f.chain[r - 1], c1 = f.chain[r];
        if (c1 != c0) { ++nb; if (c1 < c0) bad = 1u; if ((unsigned long long)r < first) first = (unsigned long long)r; }
        else if (f.draw[r] < f.draw[r - 1]) bad = 1u;
    }
    if (nb) { atomicAdd(&s_bounds, nb); atomicMin(&s_first, first); }
    if (bad) atomicOr(&s_bad, 1u);
    __syncthreads();
    const i64 L = (i64)s_first;          // length of the first run (M when there is one chain)
    unsigned uneven = 0;
    if (L > 0)
        for (i64 r = 1 + tid; r < f.M; r += 256)
            if (f.chain[r] != f.chain[r - 1] && r % L != 0) uneven = 1u;
    if (uneven) atomicOr(&s_uneven, 1u);
    __syncthreads();
    if (tid == 0) {
        const i64 Cn = f.M > 0 ? (i64)s_bounds + 1 : 0;
        const bool equal = f.M == 0 || (!s_uneven && L > 0 && f.M % L == 0 && Cn == f.M / L);
        out[4 * blockIdx.x + 0] = Cn;
        out[4 * blockIdx.x + 1] = f.M > 0 ? L : 0;
        out[4 * blockIdx.x + 2] = equal ? 1 : 0;
        out[4 * blockIdx.x + 3] = s_bad ? 0 : 1;
    }
}

}  // namespace pq
}  // namespace mcr
