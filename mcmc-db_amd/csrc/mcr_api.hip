// mcr_api.hip -- the C ABI of libmcmcref_hip.so (see include/mcmcref_hip.h) and the host-side
// launch logic: workspace carving, parameter chunking, async result slots, HIP-event profiling.
// Plain HIP runtime only: no torch, no hipify, no compatibility layers.
#include "../../include/mcmcref_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <exception>
#include <memory>
#include <thread>
#include <string>
#include <functional>
#include <new>
#include <vector>

#include "mcr_kernels.hpp"
#include "mcr_sort32.hpp"
#include "mcr_diag.hpp"
#include "mcr_fft.hpp"
#include "mcr_ext.hpp"
#include "mcr_parquet.hpp"
#include "mcr_comm.hpp"

using namespace mcr;

namespace {

constexpr int kTile = 4096;          // pooled draws per sorted tile / bucket capacity (256 lanes x 16 or 512 x 8)
constexpr i64 kIdx16Max = 65535;      // pooled arrays up to this length carry 16-bit positions through the sort
constexpr int kMaxChains = 256;
constexpr int kMaxGridY = 65535;

enum KernelId {
    K_INGEST = 0, K_MOMENTS, K_MOMENTS_FINAL, K_TILE_SORT, K_MERGE, K_ORDER_STATS, K_RANK_Z, K_FOLD_MERGE,
    K_DIAG, K_FINALIZE, K_COMPARE, K_FILL, K_SPLITTERS, K_BUCKET_MERGE, K_ACOV_MORE,
    K_DIAG2, K_ACOV_SEG, K_TWO_SAMPLE, K_COV, K_ZTABLE, K_PQ_SNAPPY, K_PQ_DECODE, K_GATHER, K_ACOV_LONG, K_DIAG_LONG, K_COV_FINAL, K_FFT, K_COUNT
};
const char* const kKernelNames[K_COUNT] = {
    "k_ingest", "k_moments", "k_moments_final", "k_tile_sort", "k_merge", "k_order_stats", "k_rank_z",
    "k_fold_merge", "k_diag", "k_finalize", "k_compare", "k_fill_synth", "k_splitters",
    "k_bucket_merge", "k_acov_more", "k_diag_combine2", "k_acov_seg", "k_two_sample", "k_cov_mfma", "k_ztable",
    "k_pq_snappy", "k_pq_decode", "k_gather_rows", "k_acov_long", "k_diag_long_scan", "k_cov_final", "k_fft"};

struct EvPair { hipEvent_t a, b; int kid; };

struct Chunk { i64 p0, pc; size_t res_off; };

struct Slot {
    bool busy = false;
    double* d_res = nullptr; double* h_res = nullptr; size_t res_cap = 0;   // doubles
    i64* d_off = nullptr; i64* h_off = nullptr; size_t off_cap = 0;          // entries
    i64 off_C = -1, off_N = -1;     // regular chain offsets c * N already resident in d_off (no upload per call)
    mcr_summary out{};
    i64 P = 0, M = 0; int nq = 0; int C = 0;
    bool trivial_nan = false;  // M == 0: no kernels ran
    hipEvent_t done = nullptr;  // recorded on the slot's lane after its last copy (mcr_summarize_wait_one)
    int lane = 0;
    i64 qlo[MCR_MAX_QUANTILES];
    std::vector<Chunk> chunks;
};

struct GraphEntry {
    std::vector<uint64_t> key;
    hipGraphExec_t exec = nullptr;
    std::vector<Chunk> chunks;
};

char g_init_err[512] = "";

}  // namespace

struct mcr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    char err[512] = "";
    size_t ws_limit = 0;
    void* ws = nullptr; size_t ws_bytes = 0;          // workspace of the CURRENT lane (see use_lane)
    // Lanes = streams with a workspace each (MCR_LANES, default 4): consecutive enqueues rotate lanes, so the
    // tail of one call's kernels (partial last waves of workgroups) overlaps the next call's head.
    hipStream_t lane_stream[MCR_MAX_INFLIGHT] = {};
    hipStream_t lane_aux[MCR_MAX_INFLIGHT] = {};      // second stream of a lane + its fork / join events (lone calls: run_pipeline)
    hipEvent_t lane_fork[MCR_MAX_INFLIGHT] = {}, lane_join[MCR_MAX_INFLIGHT] = {};
    bool fork_lone = true;                            // MCR_FORK=0: never fork
    int t3_workgroups = 256;                          // MCR_T3_WG: workgroups of a k_tier3 launch (they share the (listed pair, lag group) items)
    void* lane_ws[MCR_MAX_INFLIGHT] = {};
    size_t lane_ws_bytes[MCR_MAX_INFLIGHT] = {};
    int lane = 0, n_lanes = 4;
    void* stage = nullptr; size_t stage_bytes = 0;  // device copy of host tensors (mcr_summarize)
    hipStream_t copy_stream = nullptr;              // uploads of mcr_summarize, overlapped with the lanes' kernels
    // Parquet ingest (mcr_parquet_decode): uploaded column chunks, decompression scratch, page table + error word
    void* pq_stage = nullptr; size_t pq_stage_bytes = 0;
    void* pq_scratch = nullptr; size_t pq_scratch_bytes = 0;
    void* pq_tab = nullptr; size_t pq_tab_bytes = 0;
    void* fs_arena = nullptr; size_t fs_arena_bytes = 0;   // mcr_summarize_files: decoded draws + chain / draw ids
    void* pq_pin = nullptr; size_t pq_pin_bytes = 0;       // mcr_summarize_files: pinned host staging of the files' column chunks
    int io_threads = 8;                                    // MCR_IO_THREADS: host threads that read the file images and parse their footers
    size_t io_piece = (size_t)2 << 20;                     // MCR_IO_PIECE_KB: the finished prefix is uploaded in pieces of at least this size
    // z tables (k_ztable): a function of M alone, so they are computed once per pooled length and kept for the life of
    // the context instead of being relaunched by every call (most recently used first; a handful of shapes is typical)
    struct ZTab { i64 M; double* tab; };
    std::vector<ZTab> ztabs;
    struct Twiddle { int L; double2* tab; };          // exp(-2 pi i t / L), t < L / 2, per sub-transform length (mcr_fft.hpp)
    std::vector<Twiddle> twiddles;
    Slot slots[MCR_MAX_INFLIGHT];
    int n_inflight = 0, next_slot = 0;
    std::vector<int> order;  // busy slots in enqueue order
    bool prof = false;
    std::vector<EvPair> pending;
    std::vector<hipEvent_t> free_ev;
    int64_t k_launches[K_COUNT] = {0};
    double k_ms[K_COUNT] = {0};
    // hipGraph cache: the launch sequence of one summarize call is static for a given shape,
    // buffer set and slot, so it is captured once and replayed (removes ~5 us of host launch gap
    // between each of the ~12 kernels).  Disabled while profiling (events sit between kernels).
    bool fft_on = true;      // MCR_FFT=0: long chains take the direct tier-3 rounds only (A/B measurements, parity tests)
    bool f32_records = true; // MCR_F32_RECORDS=0: f32 tensors take the f64 kernels (widened by the tile sort) instead of mcr_sort32.hpp
    bool splitters_pairwise = false;   // MCR_SPLITTERS_PAIRWISE=1: rank the regular samples pair by pair whatever their number (A/B, parity tests)
    int sort_cfg = 10;       // MCR_SORT_CFG = tile + 10 * merge geometry (see sort_stage_i); default: tile 256 x 16, merges 512 x 8
    size_t dbg_lds_pad[3] = {0, 0, 0};   // MCR_DBG_LDS_PAD="t,b,f": extra dynamic LDS bytes for tile sort / bucket merge / fold (occupancy experiments)
    double rho_band = kRhoBand;   // MCR_RHO_BAND: half-width of the guard band of the tier-3 scan (0 = decide on the raw values)
    unsigned* guard_count = nullptr;   // device counter: band lags re-derived the reference's way (mcr_rho_guard_count)
    bool graph_on = false;   // MCR_GRAPH=1: capture / replay (measured: no throughput gain, +0.17 ms per synchronous call)
    std::vector<GraphEntry> graphs;
};

namespace {

int fail(mcr_ctx* ctx, int code, const char* fmt, ...)
{
    char* dst = ctx ? ctx->err : g_init_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(ctx, call)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(ctx, e_ == hipErrorOutOfMemory ? MCR_ENOMEM : MCR_EHIP, "%s failed: %s", \
                        #call, hipGetErrorString(e_));                                            \
    } while (0)

void prof_begin(mcr_ctx* ctx, int kid)
{
    if (!ctx->prof) return;
    EvPair pr{nullptr, nullptr, kid};
    for (hipEvent_t* e : {&pr.a, &pr.b}) {
        if (!ctx->free_ev.empty()) { *e = ctx->free_ev.back(); ctx->free_ev.pop_back(); }
        else if (hipEventCreate(e) != hipSuccess) *e = nullptr;
    }
    if (pr.a && pr.b) { hipEventRecord(pr.a, ctx->stream); ctx->pending.push_back(pr); }
}
void prof_end(mcr_ctx* ctx)
{
    if (!ctx->prof || ctx->pending.empty()) return;
    hipEventRecord(ctx->pending.back().b, ctx->stream);
}
// Stream must be idle (synchronised) when this is called.
void prof_resolve(mcr_ctx* ctx)
{
    for (EvPair& pr : ctx->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pr.a, pr.b) == hipSuccess) {
            ctx->k_ms[pr.kid] += ms;
            ctx->k_launches[pr.kid] += 1;
        }
        ctx->free_ev.push_back(pr.a);
        ctx->free_ev.push_back(pr.b);
    }
    ctx->pending.clear();
}

#define LAUNCH(ctx, kid, kern, grid, block, smem, ...)                                         \
    do {                                                                                       \
        prof_begin(ctx, kid);                                                                  \
        hipLaunchKernelGGL(kern, grid, block, smem, (ctx)->stream, __VA_ARGS__);               \
        prof_end(ctx);                                                                         \
        hipError_t le_ = hipGetLastError();                                                    \
        if (le_ != hipSuccess)                                                                 \
            return fail(ctx, MCR_EHIP, "launch %s failed: %s", kKernelNames[kid],             \
                        hipGetErrorString(le_));                                               \
    } while (0)

void use_lane(mcr_ctx* ctx, int lane)
{
    ctx->lane_ws[ctx->lane] = ctx->ws; ctx->lane_ws_bytes[ctx->lane] = ctx->ws_bytes;   // save current
    ctx->lane = lane;
    ctx->stream = ctx->lane_stream[lane];
    ctx->ws = ctx->lane_ws[lane]; ctx->ws_bytes = ctx->lane_ws_bytes[lane];
}
void sync_all(mcr_ctx* ctx)
{
    for (hipStream_t s : ctx->lane_aux) if (s) hipStreamSynchronize(s);     // (joined into the lane's stream in normal operation)
    for (hipStream_t s : ctx->lane_stream) if (s) hipStreamSynchronize(s);
}

void drop_graphs(mcr_ctx* ctx)   // waits for every lane first (a cached graph may be in flight on any)
{
    sync_all(ctx);
    for (GraphEntry& g : ctx->graphs)
        if (g.exec) hipGraphExecDestroy(g.exec);
    ctx->graphs.clear();
}

int ensure_ws(mcr_ctx* ctx, size_t bytes)
{
    if (bytes <= ctx->ws_bytes) return MCR_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    drop_graphs(ctx);
    if (ctx->ws) { hipFree(ctx->ws); ctx->ws = nullptr; ctx->ws_bytes = 0; }
    const size_t want = bytes + (bytes >> 3);  // a little slack so near-equal shapes do not realloc
    hipError_t e = hipMalloc(&ctx->ws, want);
    if (e != hipSuccess) {
        e = hipMalloc(&ctx->ws, bytes);
        if (e != hipSuccess) return fail(ctx, MCR_ENOMEM, "workspace hipMalloc(%zu) failed: %s", bytes,
                                         hipGetErrorString(e));
        ctx->ws_bytes = bytes;
    } else {
        ctx->ws_bytes = want;
    }
    return MCR_OK;
}

// Device z table of pooled length M (2M doubles), from the context's cache; filled on first use.
constexpr size_t kMaxZTabs = 24;
int get_ztab(mcr_ctx* ctx, i64 M, double** out)
{
    for (size_t k = 0; k < ctx->ztabs.size(); ++k)
        if (ctx->ztabs[k].M == M) {
            const mcr_ctx::ZTab hit = ctx->ztabs[k];
            ctx->ztabs.erase(ctx->ztabs.begin() + (long)k);
            ctx->ztabs.insert(ctx->ztabs.begin(), hit);
            *out = hit.tab;
            return MCR_OK;
        }
    if (ctx->ztabs.size() >= kMaxZTabs) {           // the evicted table may still be read by a call in flight
        sync_all(ctx);
        drop_graphs(ctx);
        hipFree(ctx->ztabs.back().tab);
        ctx->ztabs.pop_back();
    }
    double* tab = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&tab, sizeof(double) * (size_t)(2 * M > 2 ? 2 * M : 2)));
    hipLaunchKernelGGL(k_ztable, dim3((unsigned)((2 * M + 255) / 256)), dim3(256), 0, ctx->stream, tab, M);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);     // other lanes will read it: complete before anyone can
    if (e != hipSuccess) { hipFree(tab); return fail(ctx, MCR_EHIP, "k_ztable failed: %s", hipGetErrorString(e)); }
    ctx->ztabs.insert(ctx->ztabs.begin(), mcr_ctx::ZTab{M, tab});
    *out = tab;
    return MCR_OK;
}

// Twiddle table of a sub-transform length (at most 11 distinct lengths exist), from the context's cache.
int get_twiddles(mcr_ctx* ctx, int L, double2** out)
{
    for (const mcr_ctx::Twiddle& t : ctx->twiddles)
        if (t.L == L) { *out = t.tab; return MCR_OK; }
    double2* tab = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&tab, sizeof(double2) * (size_t)(L / 2 > 1 ? L / 2 : 1)));
    hipLaunchKernelGGL(fft::k_fft_twiddles, dim3((unsigned)((L / 2 + 255) / 256)), dim3(256), 0, ctx->stream, tab, L);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { hipFree(tab); return fail(ctx, MCR_EHIP, "k_fft_twiddles failed: %s", hipGetErrorString(e)); }
    ctx->twiddles.push_back(mcr_ctx::Twiddle{L, tab});
    *out = tab;
    return MCR_OK;
}

// FFT tier of the ESS lags: chains of more than kFftMinN draws (and at most 2^21, so that N = 2^ceil(log2 2n) splits
// into two sub-transforms of at most 2048).  slots = listed pairs served per call (the rest take the direct rounds).
constexpr i64 kFftMinN = 16384;
constexpr int kFftChainBatch = 4;
struct FftPlan { bool on = false; int logN = 0, log1 = 0, log2 = 0, slots = 0, cb = 0; size_t bytes = 0; };
FftPlan plan_fft(i64 n, int C, bool enabled)
{
    FftPlan f;
    if (!enabled || n <= kFftMinN || C < 2) return f;
    int lg = 1;
    while (((i64)1 << lg) < 2 * n) ++lg;
    if (lg > 22) return f;
    f.on = true; f.logN = lg; f.log1 = lg / 2; f.log2 = lg - f.log1;
    const i64 N = (i64)1 << lg;
    i64 slots = ((i64)1 << 23) / N;
    f.slots = (int)(slots < 2 ? 2 : (slots > 32 ? 32 : slots));
    f.cb = (C + 1) / 2 < kFftChainBatch ? (C + 1) / 2 : kFftChainBatch;     // transforms per batch: two chains share one
    f.bytes = (size_t)f.slots * ((size_t)f.cb * N * 16 + (size_t)N * 8 + (size_t)N * 16) + 3 * 256;
    return f;
}

int ensure_slot(mcr_ctx* ctx, Slot& s, size_t res_doubles, size_t off_entries)
{
    if (res_doubles > s.res_cap || off_entries > s.off_cap) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        drop_graphs(ctx);
    }
    if (res_doubles > s.res_cap) {
        if (s.d_res) hipFree(s.d_res);
        if (s.h_res) hipHostFree(s.h_res);
        s.d_res = nullptr; s.h_res = nullptr; s.res_cap = 0;
        HIP_TRY(ctx, hipMalloc((void**)&s.d_res, res_doubles * sizeof(double)));
        HIP_TRY(ctx, hipHostMalloc((void**)&s.h_res, res_doubles * sizeof(double), hipHostMallocDefault));
        s.res_cap = res_doubles;
    }
    if (off_entries > s.off_cap) {
        if (s.d_off) hipFree(s.d_off);
        if (s.h_off) hipHostFree(s.h_off);
        s.d_off = nullptr; s.h_off = nullptr; s.off_cap = 0; s.off_C = s.off_N = -1;
        HIP_TRY(ctx, hipMalloc((void**)&s.d_off, off_entries * sizeof(i64)));
        HIP_TRY(ctx, hipHostMalloc((void**)&s.h_off, off_entries * sizeof(i64), hipHostMallocDefault));
        s.off_cap = off_entries;
    }
    return MCR_OK;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Carve {
    char* base; size_t off = 0;
    template <typename T> T* take(size_t n)
    {
        off = align_up(off, 256);
        T* p = reinterpret_cast<T*>(base + off);
        off += n * sizeof(T);
        return p;
    }
};

struct WsPlan {
    size_t per_param;  // bytes per parameter (upper bound incl. alignment slack handled separately)
    i64 ntiles;
    // bucket path: the pooled array is cut into bk_k <= 16 sorted runs of length bk_R (a tile, or tiles
    // pre-merged by a few pairwise passes); #buckets, samples per bucket
    int bk_B = 0, bk_D = 0, bk_k = 0;
    i64 bk_R = 0;
};

WsPlan plan_ws(i64 M, int C, bool ingest, bool ranks, i64 nstage)
{
    WsPlan w;
    w.ntiles = (M + kTile - 1) / kTile;
    {
        i64 R = kTile;
        while ((M + R - 1) / R > kMaxBucketTiles) R *= 2;
        const i64 k = (M + R - 1) / R;
        if (k * (R / 64) <= 8192) {                           // the samples of one parameter must fit LDS
            const i64 last = M - (k - 1) * R;
            const i64 sf = (k - 1) * (R / 64) + last / 64;    // finite regular samples
            w.bk_R = R; w.bk_k = (int)k;
            w.bk_D = (int)((4096 - 64 - 79 * k) / 64);        // 64*D + 64*k + 15*k <= 4032: the top 64 slots are the kernel's scratch
            w.bk_B = (int)((sf + w.bk_D - 1) / w.bk_D);
            if (w.bk_B < 1) w.bk_B = 1;
        }
    }
    w.per_param = (size_t)(w.ntiles + 16) * 64 * 8 + (size_t)(w.bk_B + 1) * ((size_t)w.bk_k + 1) * 4 + 16 +
                  (size_t)M * (8 + 4) * 2 + (size_t)M * 4 * 2 + (ingest ? (size_t)M * 8 : 0) +
                  (ranks ? (size_t)M * 32 + 1024 : 0) + (size_t)w.ntiles * kMomRec * 8 + 8 +
                  (size_t)2 * (size_t)(C > 0 ? C : 1) *
                      ((size_t)((nstage + kSeg - 1) / kSeg + 1) * (kSegRec + 64 * kMoreBlocks) + kChState) * 8 +
                  (size_t)2 * (size_t)(nstage > 0 ? nstage : 1) * 8 + 2 * 4 + 2 * (kT3Stages + 1) * 4 + 2 * kPairState * 8 +
                  8 + 64;
    return w;
}

struct PipeIn {
    const void* X;       // [pc][M] contiguous: f64 (user tensor or ingest buffer) or, x_f32, the user's f32 tensor itself
    bool x_f32 = false;
    bool no_records = false;   // MCR_F32_RECORDS=0: keep f32 tensors on the f64 kernels (A/B measurements, parity tests)
    i64 M, pc;
    int C;
    const i64* d_off;
    i64 n, nh;
    QArgs q;
    double* d_res;       // chunk's result table, stride pc
    // carved buffers
    double *kA, *kB, *part;
    u32 *zb, *zt;        // rank codes n2 of every draw in time order (z = ztab[n2], rank = (n2 + 1) / 2)
    void *iA, *iB;       // pooled-position payload of the sort: u16 when M <= kIdx16Max, else u32
    i64* split;
    double* rec;         // [pc][2][C][nseg][kSegRec] first-pass segment records
    unsigned* more;      // [pc][2] continuation flags
    double* state;       // [pc][2][4] rho scan state
    double* chstate;     // [pc][2][C][kChState]
    double* rec2;        // [pc][2][C][nseg][64*kMoreBlocks] lag products of tier 2 (lags 64..255)
    double* acov;        // [pc][2][n] deviation products of tier 3 (lags >= 256), listed pairs only
    unsigned* long_count; // [1] number of pairs in the tier-3 list of this call
    unsigned* long_list;  // [2 pc]
    unsigned* t3c;        // [kT3Stages + 1][2 pc] k_tier3: per pair and stage the finished lag groups (+ kT3Prev), then "decided"
    FftPlan fft;          // FFT tier for long chains (mcr_fft.hpp): buffers shared by the chunks of a call
    double2 *fft_A = nullptr, *fft_B = nullptr; double* fft_S = nullptr;
    double2 *tw1 = nullptr, *tw2 = nullptr;
    i64 nstage;          // longest chain prefix the segment grid has to cover
    double* ztab;        // [2M] z of every possible tie run (shared by all parameters of the call)
    double* samp;        // [pc][ntiles][64] regular samples of the sorted tiles
    u32* cut;            // [pc][B+1][ntiles]
    u32* boff;           // [pc][B+1]
    int bk_B = 0, bk_D = 0, bk_k = 0;
    i64 bk_R = 0;
    i64 ntiles;
    bool do_diag = true;  // false: Backend.stats only (sort + order statistics + moments)
    bool fork = false;    // lone call: the bulk half of the diagnostics on the lane's second stream (run_pipeline)
};

// Tiers 1 and 2 of one kind (0 bulk, 1 folded) or of both (kind_sel < 0): segment products, combine, the flagged pairs'
// lags 64..255 -- everything of the diagnostics that does not need the other kind.  On ctx->stream.
int launch_diag_front(mcr_ctx* ctx, const PipeIn& a, int kind_sel)
{
    // short chains (<= 1024 draws, e.g. the packaged corpus): half-size segments on 128-thread
    // workgroups, so twice as many of the latency-bound workgroups are resident per CU
    const bool small = a.nstage <= 1024;
    const int seg = small ? 1024 : kSeg;
    const int nseg = (int)((a.nstage + seg - 1) / seg);
    const unsigned pk = (unsigned)((kind_sel < 0 ? 2 : 1) * a.pc), ky = kind_sel < 0 ? 2u : 1u;
    if (small) {
        LAUNCH(ctx, K_ACOV_SEG, (k_acov_seg<128, 1024, true>), dim3((unsigned)nseg, (unsigned)a.C, pk), dim3(128), 0,
               (const u32*)a.zb, (const u32*)a.zt, (const double*)a.ztab, a.M, a.d_off, a.C, a.n, a.nh, nseg,
               (const unsigned*)nullptr, a.rec, kind_sel);
    } else {
        LAUNCH(ctx, K_ACOV_SEG, (k_acov_seg<256, kSeg, true>), dim3((unsigned)nseg, (unsigned)a.C, pk), dim3(256), 0,
               (const u32*)a.zb, (const u32*)a.zt, (const double*)a.ztab, a.M, a.d_off, a.C, a.n, a.nh, nseg,
               (const unsigned*)nullptr, a.rec, kind_sel);
    }
    LAUNCH(ctx, K_DIAG, k_diag_combine, dim3((unsigned)a.pc, ky), dim3(64u * (unsigned)(a.C < kCombineWaves ? a.C : kCombineWaves)), (size_t)6 * a.C * 8, (const u32*)a.zb,
           (const u32*)a.zt, (const double*)a.ztab, a.M, a.d_off, a.C, a.n, a.nh, nseg, (const double*)a.rec, a.d_res, a.pc, a.more, a.state, a.chstate,
           a.long_count, a.t3c,
           ctx->rho_band, ctx->guard_count, kind_sel);
    // tier 2 for pairs whose first negative rho lies beyond lag 63 (others exit at once)
    if (small) {
        LAUNCH(ctx, K_ACOV_MORE, (k_acov_seg<128, 1024, false>), dim3((unsigned)nseg, (unsigned)a.C, pk), dim3(128), 0,
               (const u32*)a.zb, (const u32*)a.zt, (const double*)a.ztab, a.M, a.d_off, a.C, a.n, a.nh, nseg,
               (const unsigned*)a.more, a.rec2, kind_sel);
    } else {
        LAUNCH(ctx, K_ACOV_MORE, (k_acov_seg<256, kSeg, false>), dim3((unsigned)nseg, (unsigned)a.C, pk), dim3(256), 0,
               (const u32*)a.zb, (const u32*)a.zt, (const double*)a.ztab, a.M, a.d_off, a.C, a.n, a.nh, nseg,
               (const unsigned*)a.more, a.rec2, kind_sel);
    }
    return MCR_OK;
}

// Split R-hat + ESS (mcr_diag.hpp): the joint part -- tier 2's combine (+ finalize), tier 3.
int launch_diag(mcr_ctx* ctx, const PipeIn& a)
{
    const bool small = a.nstage <= 1024;
    const int seg = small ? 1024 : kSeg;
    const int nseg = (int)((a.nstage + seg - 1) / seg);
    const bool fused_tier3 = !a.fft.on && a.n <= 16384 && 2 * a.pc <= kTier3MaxPairs;     // k_tier3: one launch for the whole tier
    LAUNCH(ctx, K_DIAG2, k_diag_combine2, dim3((unsigned)a.pc, 2), dim3(1024), (size_t)2 * a.C * 8, (const u32*)a.zb,
           (const u32*)a.zt, (const double*)a.ztab, a.M, a.d_off, a.C,
           a.n, nseg, (const double*)a.rec2, (const unsigned*)a.more, a.state,
           a.chstate, a.d_res, a.pc, a.kA, a.kB,   // kA / kB: the sort's key buffers, free by now
           (const double*)a.part, (int)a.ntiles, fused_tier3 ? 0 : 1, ctx->rho_band, ctx->guard_count);
    if (a.n <= kLag2) return MCR_OK;       // chains short enough to be decided by lag 255 never reach tier 3
    if (fused_tier3) {
        // the common case in ONE launch: list, products of the lags [256, n) stage by stage, scans (k_tier3); a fixed number
        // of workgroups (MCR_T3_WG) shares the (listed pair, lag group) items
        const i64 groups = (a.n - kLag2 + kLongGroup - 1) / kLongGroup;
        const i64 most = 2 * a.pc * groups;                                  // work items if every pair were listed
        const unsigned wgs = (unsigned)(most < (i64)ctx->t3_workgroups ? most : (i64)ctx->t3_workgroups);
        LAUNCH(ctx, K_ACOV_LONG, k_tier3, dim3(wgs), dim3(256), 0, (const double*)a.kA, (const double*)a.kB, a.M, a.d_off,
               a.C, a.n, (const unsigned*)a.more, a.state, a.acov, a.d_res, a.pc, ctx->rho_band, ctx->guard_count, a.t3c,
               (const u32*)a.zb, (const u32*)a.zt, (const double*)a.ztab, a.chstate);
        return MCR_OK;
    }
    LAUNCH(ctx, K_DIAG2, k_long_list, dim3(1), dim3(1024), 0, (const unsigned*)a.more, (const double*)a.state, a.pc,
           a.long_count, a.long_list);
    {
        const unsigned slots = (unsigned)((2 * a.pc < 64) ? 2 * a.pc : 64);
        LAUNCH(ctx, K_DIAG2, k_dev_fill, dim3((unsigned)((a.M + 4095) / 4096), slots), dim3(256), 0, (const u32*)a.zb, (const u32*)a.zt,
               (const double*)a.ztab, a.M, a.d_off, a.C, a.n, (const double*)a.chstate, (const unsigned*)a.long_count,
               (const unsigned*)a.long_list, a.kA, a.kB);
    }
    // tier 3 for the pairs still undecided at lag 256.
    //  * chains of more than 16 384 draws: ALL lags of the first fft.slots listed pairs by FFT (mcr_fft.hpp);
    //  * everything else (and list entries beyond those slots): direct products over the whole chip, in rounds
    //    [256, 16384) -- one round, i.e. two near-empty launches, for chains up to 16 384 draws -- then growing 4x.
    unsigned slot_from = 0;
    if (a.fft.on) {
        const fft::Plan pl{a.fft.log1, a.fft.log2};
        const int N1 = 1 << pl.log1, N2 = 1 << pl.log2, cols = fft::kColElems >> pl.log1;
        const unsigned F = (unsigned)a.fft.slots;
        const size_t lds_cols = (size_t)fft::kColElems * 16, lds_rows = (size_t)N2 * 16;
        for (int c0 = 0; c0 < a.C; c0 += 2 * a.fft.cb) {          // a batch = fft.cb transforms = 2 fft.cb chains (two per transform)
            const int left = (a.C - c0 + 1) / 2;
            const int nb = (left < a.fft.cb) ? left : a.fft.cb;
            LAUNCH(ctx, K_FFT, (fft::k_fft_cols<256>), dim3((unsigned)(N2 / cols), (unsigned)nb, F), dim3(256), lds_cols,
                   (const double*)a.kA, (const double*)a.kB, a.M, a.d_off, c0, a.C, a.n, pl, (const double2*)a.tw1,
                   (const unsigned*)a.long_count, (const unsigned*)a.long_list, (const double*)a.state, a.fft_A, a.fft.cb);
            LAUNCH(ctx, K_FFT, (fft::k_fft_rows_power<256>), dim3((unsigned)N1, F), dim3(256), lds_rows, (const double2*)a.fft_A, pl,
                   (const double2*)a.tw2, (const unsigned*)a.long_count, (const unsigned*)a.long_list, (const double*)a.state,
                   a.fft_S, a.fft.cb, nb, c0 == 0 ? 1 : 0);
        }
        LAUNCH(ctx, K_FFT, (fft::k_fft_rows_spec<256>), dim3((unsigned)N1, F), dim3(256), lds_rows, (const double*)a.fft_S, pl,
               (const double2*)a.tw2, (const unsigned*)a.long_count, (const unsigned*)a.long_list, (const double*)a.state, a.fft_B);
        LAUNCH(ctx, K_FFT, (fft::k_fft_cols_out<256>), dim3((unsigned)(N2 / cols), F), dim3(256), lds_cols, (const double2*)a.fft_B, pl,
               (const double2*)a.tw1, a.n, (const unsigned*)a.long_count, (const unsigned*)a.long_list, (const double*)a.state,
               a.acov);
        slot_from = F;
    }
    const unsigned scan_slots = (unsigned)((2 * a.pc < 16) ? 2 * a.pc : 16);      // one light workgroup per listed pair at a time
    for (i64 L0 = kLag2; L0 < a.n;) {
        const i64 L1 = (L0 < 16384) ? 16384 : L0 * 4;
        const i64 lend = (L1 < a.n) ? L1 : a.n;
        const unsigned groups = (unsigned)((lend - L0 + kLongGroup - 1) / kLongGroup);
        unsigned slots = 256u / (groups ? groups : 1u);                             // ~256 workgroups when the round has few lag groups
        slots = slots < (unsigned)kLongSlots ? (unsigned)kLongSlots : (slots > 64u ? 64u : slots);
        if ((i64)slots > 2 * a.pc) slots = (unsigned)(2 * a.pc);
        LAUNCH(ctx, K_ACOV_LONG, (k_acov_long<256>), dim3(groups, slots), dim3(256), 0, (const double*)a.kA, (const double*)a.kB,
               a.M, a.d_off, a.C, a.n, L0, L1, (const unsigned*)a.long_count, (const unsigned*)a.long_list,
               (const double*)a.state, a.acov, slot_from);
        LAUNCH(ctx, K_DIAG_LONG, k_diag_long_scan, dim3(scan_slots), dim3(256), 0, a.C, a.n, L0, L1, (const unsigned*)a.long_count,
               (const unsigned*)a.long_list, a.state, a.acov, a.d_res, a.pc, (const u32*)a.zb, (const u32*)a.zt,
               (const double*)a.ztab, a.chstate, a.M, a.d_off, ctx->rho_band, ctx->guard_count);
        L0 = L1;
    }
    return MCR_OK;
}

// k_splitters for the runs in `keys`: few samples (S <= 1024) are ranked pair by pair, more by merging the runs' sample
// lists in the LDS (MVT outputs per thread and round: the smallest of 2, 4, 8 that holds S samples in 1024 MVT slots).
template <typename KT, int MVT>
int launch_splitters_v(mcr_ctx* ctx, const PipeIn& a, const KT* keys)
{
    const int S = a.bk_k * (int)(a.bk_R / 64);
    const size_t lds_spl = splitters_lds_bytes(S, a.bk_B, a.bk_k, MVT);
    if (lds_spl > 160 * 1024) return fail(ctx, MCR_ENOMEM, "k_splitters needs %zu bytes of LDS", lds_spl);
    if (lds_spl > 60 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_splitters<KT, MVT>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_spl));
    LAUNCH(ctx, K_SPLITTERS, (k_splitters<KT, MVT>), dim3((unsigned)a.pc), dim3(1024), lds_spl, keys, (const double*)a.samp,
           a.M, a.bk_k, a.bk_B, a.bk_D, a.bk_R, a.cut, a.boff);
    return MCR_OK;
}
template <typename KT>
int launch_splitters(mcr_ctx* ctx, const PipeIn& a, const KT* keys)
{
    const int S = a.bk_k * (int)(a.bk_R / 64);
    if (S <= 1024 || ctx->splitters_pairwise) return launch_splitters_v<KT, 0>(ctx, a, keys);
    if (S <= 2048) return launch_splitters_v<KT, 2>(ctx, a, keys);
    if (S <= 4096) return launch_splitters_v<KT, 4>(ctx, a, keys);
    return launch_splitters_v<KT, 8>(ctx, a, keys);       // (MVT must divide the runs' 64 j samples: a thread's outputs never straddle two pairs of runs)
}

// Tile sort + (bucket partition | merge passes): leaves the pooled ascending (key, idx) order of every
// parameter in *kin / *iin (one of the two ping-pong sets) and, on the bucket path with do_diag, z_bulk.
// (TNT, TVT): threads x draws per lane of the tile sort; (MNT, MVT): of the merge kernels (bucket merge, fold, passes).
template <typename IdxT, int TNT, int TVT, int MNT, int MVT>
int sort_stage_t(mcr_ctx* ctx, PipeIn& a, double** kin_o, void** iin_o, double** kout_o, void** iout_o, bool* ranked_o)
{
    static_assert(TNT * TVT == kTile && MNT * MVT == kTile, "tile geometry");
    const i64 M = a.M, pc = a.pc;
    const unsigned py = (unsigned)pc;
    constexpr size_t lds_tile = sort_lds_bytes<IdxT>(kTile);
    // (the z lookup table of this M comes from the context's cache: get_ztab)
    // 1. tile sort (+ moment partials, + regular samples when a tile is already a run)
    const bool bucket = a.bk_B > 0;
    double* const samp1 = (bucket && a.bk_R == kTile) ? a.samp : (double*)nullptr;
    if (a.x_f32) {
        LAUNCH(ctx, K_TILE_SORT, (k_tile_sort<TNT, TVT, IdxT, float>), dim3((unsigned)a.ntiles, py), dim3(TNT),
               lds_tile, (const float*)a.X, M, a.kA, (IdxT*)a.iA, a.part, (int)a.ntiles, samp1);
    } else {
        LAUNCH(ctx, K_TILE_SORT, (k_tile_sort<TNT, TVT, IdxT, double>), dim3((unsigned)a.ntiles, py), dim3(TNT),
               lds_tile + ctx->dbg_lds_pad[0], (const double*)a.X, M, a.kA, (IdxT*)a.iA, a.part, (int)a.ntiles, samp1);
    }
    double *kin = a.kA, *kout = a.kB;
    IdxT *iin = (IdxT*)a.iA, *iout = (IdxT*)a.iB;
    const unsigned nblk = (unsigned)((M + kTile - 1) / kTile);
    bool ranked = false;
    // 2. pairwise merge-path passes: up to the run length of the bucket partition, or all the way
    const i64 Rstop = bucket ? a.bk_R : M;
    for (i64 R = kTile; R < Rstop; R *= 2) {
        LAUNCH(ctx, K_MERGE, (k_merge<MNT, MVT, false, IdxT>), dim3(nblk, py), dim3(MNT), lds_tile + 256,
               (const double*)kin, (const IdxT*)iin, kout, iout, M, R, (double*)nullptr, pc,
               QArgs{}, (u32*)nullptr, (const i64*)nullptr);
        std::swap(kin, kout);
        std::swap(iin, iout);
    }
    if (bucket) {
        // 3. exact k-way partition + in-LDS bucket merge, fused with ranks -> z
        if (a.bk_R != kTile)
            LAUNCH(ctx, K_SPLITTERS, k_sample_runs<double>, dim3((unsigned)a.bk_k, py), dim3(256), 0, (const double*)kin, M,
                   a.bk_R, a.samp);
        {
            const int rc = launch_splitters<double>(ctx, a, (const double*)kin);
            if (rc) return rc;
        }
        const unsigned pgrp = (unsigned)((pc + 7) / 8 * 8);   // XCD-aware 1-D grid (xcd_map)
        LAUNCH(ctx, K_BUCKET_MERGE, (k_bucket_merge<MNT, MVT, IdxT>), dim3(pgrp * (unsigned)a.bk_B), dim3(MNT), lds_tile + ctx->dbg_lds_pad[1],
               (const double*)kin, (const IdxT*)iin, kout, iout, M, a.bk_k, a.bk_B, (const u32*)a.cut,
               (const u32*)a.boff, a.do_diag ? a.zb : (u32*)nullptr, pc, a.bk_R);
        std::swap(kin, kout);
        std::swap(iin, iout);
        ranked = true;
    }
    *kin_o = kin; *iin_o = iin; *kout_o = kout; *iout_o = iout; *ranked_o = ranked;
    return MCR_OK;
}

// Order statistics by the fold kernel's own workgroups (a launch less) while a parameter has few of them; a pooled
// array beyond 16-bit positions has a hundred fold workgroups per parameter, and one k_order_stats launch serves them.
inline bool order_stats_in_fold(const PipeIn& a) { return a.do_diag && a.M <= kIdx16Max; }
inline const i64* fold_split(const PipeIn& a) { return order_stats_in_fold(a) ? (const i64*)nullptr : (const i64*)a.split; }

template <typename IdxT, int NT, int VT>
int launch_fold_rec(mcr_ctx* ctx, PipeIn& a, double* kin, unsigned fgrid)
{
    LAUNCH(ctx, K_FOLD_MERGE, (k_merge<NT, VT, true, IdxT, u64>), dim3(fgrid), dim3(NT), sort_lds_bytes<IdxT>(kTile),
           (const u64*)kin, (const IdxT*)nullptr, (double*)nullptr, (IdxT*)nullptr, a.M, (i64)0, a.d_res, a.pc,
           a.q, a.zt, fold_split(a));
    return MCR_OK;
}

template <typename IdxT, int NT, int VT>
int launch_fold(mcr_ctx* ctx, PipeIn& a, double* kin, void* iin, double* kout, void* iout, unsigned fgrid)
{
    LAUNCH(ctx, K_FOLD_MERGE, (k_merge<NT, VT, true, IdxT>), dim3(fgrid), dim3(NT), sort_lds_bytes<IdxT>(kTile) + ctx->dbg_lds_pad[2],
           (const double*)kin, (const IdxT*)iin, kout, (IdxT*)iout, a.M, (i64)0, a.d_res, a.pc,
           a.q, a.zt, fold_split(a));
    return MCR_OK;
}

// f32 tensors (Arrow layout, bucket path): the same stage on packed (key, position) records (mcr_sort32.hpp).  The
// record arrays live in the f64 key buffers kA / kB (8 bytes per draw); the position buffers stay unused.
template <int TNT, int TVT, int MNT, int MVT>
int sort_stage_rec(mcr_ctx* ctx, PipeIn& a, double** kin_o, void** iin_o, double** kout_o, void** iout_o, bool* ranked_o)
{
    static_assert(TNT * TVT == kTile && MNT * MVT == kTile, "tile geometry");
    const i64 M = a.M, pc = a.pc;
    const unsigned py = (unsigned)pc;
    u64 *rin = reinterpret_cast<u64*>(a.kA), *rout = reinterpret_cast<u64*>(a.kB);
    LAUNCH(ctx, K_TILE_SORT, (k_tile_sort32<TNT, TVT>), dim3((unsigned)a.ntiles, py), dim3(TNT), (size_t)kTile * 8,
           (const float*)a.X, M, rin, a.part, (int)a.ntiles, (a.bk_R == kTile) ? a.samp : (double*)nullptr);
    const unsigned nblk = (unsigned)((M + kTile - 1) / kTile);
    for (i64 R = kTile; R < a.bk_R; R *= 2) {
        LAUNCH(ctx, K_MERGE, (k_merge32<MNT, MVT>), dim3(nblk, py), dim3(MNT), rec_lds_bytes(kTile), (const u64*)rin, rout, M, R);
        std::swap(rin, rout);
    }
    if (a.bk_R != kTile)
        LAUNCH(ctx, K_SPLITTERS, k_sample_runs<u64>, dim3((unsigned)a.bk_k, py), dim3(256), 0, (const u64*)rin, M, a.bk_R, a.samp);
    {
        const int rc = launch_splitters<u64>(ctx, a, (const u64*)rin);
        if (rc) return rc;
    }
    const unsigned pgrp = (unsigned)((pc + 7) / 8 * 8);
    LAUNCH(ctx, K_BUCKET_MERGE, (k_bucket_merge32<MNT, MVT>), dim3(pgrp * (unsigned)a.bk_B), dim3(MNT), rec_lds_bytes(kTile) + 512,
           (const u64*)rin, rout, M, a.bk_k, a.bk_B, (const u32*)a.cut, (const u32*)a.boff,
           a.do_diag ? a.zb : (u32*)nullptr, pc, a.bk_R);
    std::swap(rin, rout);
    *kin_o = reinterpret_cast<double*>(rin); *kout_o = reinterpret_cast<double*>(rout);
    *iin_o = a.iA; *iout_o = a.iB; *ranked_o = true;
    return MCR_OK;
}

int sort_stage_rec_i(mcr_ctx* ctx, PipeIn& a, double** kin_o, void** iin_o, double** kout_o, void** iout_o, bool* ranked_o)
{
    const int t = ctx->sort_cfg % 10, m = ctx->sort_cfg / 10;
    if (t == 0) {
        if (m == 0) return sort_stage_rec<256, 16, 256, 16>(ctx, a, kin_o, iin_o, kout_o, iout_o, ranked_o);
        if (m == 1) return sort_stage_rec<256, 16, 512, 8>(ctx, a, kin_o, iin_o, kout_o, iout_o, ranked_o);
        return sort_stage_rec<256, 16, 1024, 4>(ctx, a, kin_o, iin_o, kout_o, iout_o, ranked_o);
    }
    if (m == 0) return sort_stage_rec<512, 8, 256, 16>(ctx, a, kin_o, iin_o, kout_o, iout_o, ranked_o);
    if (m == 1) return sort_stage_rec<512, 8, 512, 8>(ctx, a, kin_o, iin_o, kout_o, iout_o, ranked_o);
    return sort_stage_rec<512, 8, 1024, 4>(ctx, a, kin_o, iin_o, kout_o, iout_o, ranked_o);
}

// MCR_SORT_CFG = tile + 10 * merge.  tile: 0 = 256 lanes x 16 draws, 1 = 512 x 8.  merge: 0 = 256 x 16, 1 = 512 x 8,
// 2 = 1024 x 4.  Measured on C1 (profiles/r02_*): the tile sort is fastest with long per-lane runs (the register
// network does four levels for free), the merge kernels with short ones (twice the waves per CU hide the dependent
// LDS chains of the serial merge, which has no register phase to amortise).
template <typename IdxT>
int sort_stage_i(mcr_ctx* ctx, PipeIn& a, double** kin_o, void** iin_o, double** kout_o, void** iout_o, bool* ranked_o)
{
    const int t = ctx->sort_cfg % 10, m = ctx->sort_cfg / 10;
    if (t == 0) {
        if (m == 0) return sort_stage_t<IdxT, 256, 16, 256, 16>(ctx, a, kin_o, iin_o, kout_o, iout_o, ranked_o);
        if (m == 1) return sort_stage_t<IdxT, 256, 16, 512, 8>(ctx, a, kin_o, iin_o, kout_o, iout_o, ranked_o);
        return sort_stage_t<IdxT, 256, 16, 1024, 4>(ctx, a, kin_o, iin_o, kout_o, iout_o, ranked_o);
    }
    if (m == 0) return sort_stage_t<IdxT, 512, 8, 256, 16>(ctx, a, kin_o, iin_o, kout_o, iout_o, ranked_o);
    if (m == 1) return sort_stage_t<IdxT, 512, 8, 512, 8>(ctx, a, kin_o, iin_o, kout_o, iout_o, ranked_o);
    return sort_stage_t<IdxT, 512, 8, 1024, 4>(ctx, a, kin_o, iin_o, kout_o, iout_o, ranked_o);
}

// Records are used for f32 tensors whenever the bucket partition applies (pooled arrays up to 512 K draws); beyond
// that the tile sort widens the f32 draws itself and the f64 kernels run.
inline bool use_records(const PipeIn& a) { return a.x_f32 && a.bk_B > 0 && !a.no_records; }

int sort_stage(mcr_ctx* ctx, PipeIn& a, double** kin_o, void** iin_o, double** kout_o, void** iout_o, bool* ranked_o)
{
    if (use_records(a)) return sort_stage_rec_i(ctx, a, kin_o, iin_o, kout_o, iout_o, ranked_o);
    return a.M <= kIdx16Max ? sort_stage_i<unsigned short>(ctx, a, kin_o, iin_o, kout_o, iout_o, ranked_o)
                            : sort_stage_i<u32>(ctx, a, kin_o, iin_o, kout_o, iout_o, ranked_o);
}

template <typename IdxT>
int fold_stage_i(mcr_ctx* ctx, PipeIn& a, double* kin, void* iin, double* kout, void* iout, unsigned fgrid)
{
    const int m = ctx->sort_cfg / 10;
    if (use_records(a)) {
        if (m == 0) return launch_fold_rec<IdxT, 256, 16>(ctx, a, kin, fgrid);
        if (m == 1) return launch_fold_rec<IdxT, 512, 8>(ctx, a, kin, fgrid);
        return launch_fold_rec<IdxT, 1024, 4>(ctx, a, kin, fgrid);
    }
    if (m == 0) return launch_fold<IdxT, 256, 16>(ctx, a, kin, iin, kout, iout, fgrid);
    if (m == 1) return launch_fold<IdxT, 512, 8>(ctx, a, kin, iin, kout, iout, fgrid);
    return launch_fold<IdxT, 1024, 4>(ctx, a, kin, iin, kout, iout, fgrid);
}

// The whole per-chunk pipeline on ctx->stream.  M >= 1, pc >= 1.
int run_pipeline(mcr_ctx* ctx, PipeIn& a)
{
    const i64 M = a.M, pc = a.pc;
    const unsigned py = (unsigned)pc;
    double *kin, *kout;
    void *iin, *iout;
    bool ranked;
    {
        const int rc = sort_stage(ctx, a, &kin, &iin, &kout, &iout, &ranked);
        if (rc) return rc;
    }
    // 3. order statistics: with diagnostics, by the fold kernel itself; a launch of their own for Backend.stats calls
    if (!order_stats_in_fold(a)) {
        if (use_records(a)) {
            LAUNCH(ctx, K_ORDER_STATS, k_order_stats<u64>, dim3((unsigned)((pc + 3) / 4)), dim3(256), 0,
                   (const u64*)kin, M, pc, a.q, a.d_res, a.split);
        } else {
            LAUNCH(ctx, K_ORDER_STATS, k_order_stats<double>, dim3((unsigned)((pc + 3) / 4)), dim3(256), 0,
                   (const double*)kin, M, pc, a.q, a.d_res, a.split);
        }
    }
    if (a.do_diag) {
        // 4. bulk ranks -> z (already done by k_bucket_merge on the bucket path)
        if (!ranked)
            LAUNCH(ctx, K_RANK_Z, k_rank_z, dim3((unsigned)((M + 255) / 256), py), dim3(256), 0,
                   (const double*)kin, (const u32*)iin, M, a.zb);       // no bucket path only beyond 512 K draws: u32 positions
        // A LONE call (nothing else in flight on the context: what reference.compare makes) forks here: the bulk half of
        // tiers 1 and 2 needs only the bulk rank codes, which exist now, so it runs on the lane's second stream UNDER the fold
        // kernel, and the two halves join in front of combine2 -- ~35 us off the critical path of a 0.3 ms call.  A pipelined
        // caller keeps the single stream: its lanes are full anyway, and three more launches per call would only cost.
        const bool fork = a.fork && a.C >= 2 && ctx->lane_aux[ctx->lane] != nullptr;
        hipStream_t main_stream = ctx->stream;
        if (fork) {
            HIP_TRY(ctx, hipEventRecord(ctx->lane_fork[ctx->lane], main_stream));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->lane_aux[ctx->lane], ctx->lane_fork[ctx->lane], 0));
            ctx->stream = ctx->lane_aux[ctx->lane];
            const int rc = launch_diag_front(ctx, a, 0);
            ctx->stream = main_stream;
            if (rc) return rc;
            HIP_TRY(ctx, hipEventRecord(ctx->lane_join[ctx->lane], ctx->lane_aux[ctx->lane]));
        }
        // 5+6. fold: one merge of the two monotone halves around the median, fused with ranks -> z
        const unsigned fgrid = (unsigned)((pc + 7) / 8 * 8) * (unsigned)((M + (kTile - 64) - 1) / (kTile - 64));   // 4032 outputs per fold workgroup
        {
            const int rc = M <= kIdx16Max ? fold_stage_i<unsigned short>(ctx, a, kin, iin, kout, iout, fgrid)
                                          : fold_stage_i<u32>(ctx, a, kin, iin, kout, iout, fgrid);
            if (rc) return rc;
        }
        // 7. R-hat + ESS (+ the finalize step, inside k_diag_combine2)
        if (a.C >= 2) {
            const int rc = launch_diag_front(ctx, a, fork ? 1 : -1);
            if (rc) return rc;
            if (fork) HIP_TRY(ctx, hipStreamWaitEvent(main_stream, ctx->lane_join[ctx->lane], 0));
            return launch_diag(ctx, a);
        }
    }
    // 8. finalize (calls without diagnostics, single chains)
    LAUNCH(ctx, K_FINALIZE, k_finalize, dim3((unsigned)((pc + 63) / 64)), dim3(64), 0, (const double*)a.part,
           (int)a.ntiles, M, pc, (a.do_diag ? a.C : 0), a.d_res);
    return MCR_OK;
}

int prep_quantiles(mcr_ctx* ctx, const double* qs, int nq, i64 M, QArgs& q, i64* qlo_out)
{
    if (nq < 0 || nq > MCR_MAX_QUANTILES) return fail(ctx, MCR_EINVAL, "n_q must be in [0, %d]; got %d", MCR_MAX_QUANTILES, nq);
    if (nq > 0 && !qs) return fail(ctx, MCR_EINVAL, "quantiles is NULL");
    q.nq = nq;
    for (int k = 0; k < nq; ++k) {
        if (!(qs[k] >= 0.0 && qs[k] <= 1.0)) return fail(ctx, MCR_EINVAL, "Quantiles must be in the range [0, 1]");
        const double h = (double)(M - 1) * qs[k];   // numpy: virtual index (n-1)*q ; arrow: same
        const double fl = std::floor(h);
        i64 lo = (i64)fl;
        if (lo < 0) lo = 0;
        if (M > 0 && lo > M - 1) lo = M - 1;
        q.lo[k] = lo;
        q.g[k] = h - fl;
        qlo_out[k] = lo;
    }
    return MCR_OK;
}

constexpr int res_fields(int nq) { return R_Q0 + nq; }

template <typename T>
int launch_ingest(mcr_ctx* ctx, const void* src, double* X, i64 C, i64 N, i64 pc, i64 sc, i64 sn, i64 sp, i64 p0)
{
    if (sp == 1 && sn != 1 && pc > 1) {
        const i64 nb = (N + 63) / 64;
        LAUNCH(ctx, K_INGEST, (k_ingest_transpose<T>), dim3((unsigned)(nb * C), (unsigned)((pc + 63) / 64)),
               dim3(256), 0, (const T*)src, X, C, N, pc, sc, sn, sp, p0);
    } else {
        const i64 nb = (N + 255) / 256;
        LAUNCH(ctx, K_INGEST, (k_ingest_rows<T>), dim3((unsigned)(nb * C), (unsigned)pc), dim3(256), 0,
               (const T*)src, X, C, N, sc, sn, sp, p0);
    }
    return MCR_OK;
}

void fill_nan(const mcr_summary& o, i64 P, int nq)
{
    double* arrs[] = {o.mean, o.std, o.median, o.rhat, o.rhat_bulk, o.rhat_tail, o.ess_bulk, o.ess_tail};
    for (double* a : arrs)
        if (a) for (i64 p = 0; p < P; ++p) a[p] = NAN;
    if (o.q) for (i64 i = 0; i < P * nq; ++i) o.q[i] = NAN;
    if (o.lag_bulk) for (i64 p = 0; p < P; ++p) o.lag_bulk[p] = 0;
    if (o.lag_tail) for (i64 p = 0; p < P; ++p) o.lag_tail[p] = 0;
}

// Copies slot results (already on the host) into the caller's arrays.  Returns MCR_ENONFINITE
// if any parameter saw NaN/Inf draws.
int unpack_slot(mcr_ctx* ctx, Slot& s)
{
    const mcr_summary& o = s.out;
    if (o.q_lo) for (int k = 0; k < s.nq; ++k) o.q_lo[k] = s.qlo[k];
    if (s.trivial_nan) { fill_nan(o, s.P, s.nq); return MCR_OK; }
    double nbad = 0.0;
    for (const Chunk& ch : s.chunks) {
        const double* r = s.h_res + ch.res_off;
        const i64 pc = ch.pc;
        auto cp = [&](double* dst, int field) {
            if (dst) memcpy(dst + ch.p0, r + (size_t)field * pc, sizeof(double) * (size_t)pc);
        };
        cp(o.mean, R_MEAN); cp(o.std, R_STD); cp(o.median, R_MEDIAN); cp(o.rhat, R_RHAT);
        cp(o.rhat_bulk, R_RHAT_BULK); cp(o.rhat_tail, R_RHAT_TAIL); cp(o.ess_bulk, R_ESS_BULK);
        cp(o.ess_tail, R_ESS_TAIL);
        for (i64 p = 0; p < pc; ++p) {
            if (o.lag_bulk) o.lag_bulk[ch.p0 + p] = (int64_t)r[(size_t)R_LAG_BULK * pc + p];
            if (o.lag_tail) o.lag_tail[ch.p0 + p] = (int64_t)r[(size_t)R_LAG_TAIL * pc + p];
            nbad += r[(size_t)R_BAD * pc + p];
            if (o.q) for (int k = 0; k < s.nq; ++k) o.q[(ch.p0 + p) * s.nq + k] = r[(size_t)(R_Q0 + k) * pc + p];
        }
    }
    if (nbad > 0.0) return fail(ctx, MCR_ENONFINITE, "draws contain %.0f non-finite value(s)", nbad);
    return MCR_OK;
}

int check_common(mcr_ctx* ctx, const void* draws, int dtype, i64 C, i64 N, i64 P, int min_chains,
                 const mcr_summary* out)
{
    if (!ctx) return fail(nullptr, MCR_EINVAL, "ctx is NULL");
    if (!out) return fail(ctx, MCR_EINVAL, "out is NULL");
    if (dtype != MCR_F64 && dtype != MCR_F32) return fail(ctx, MCR_EINVAL, "unsupported dtype %d", dtype);
    if (C < 0 || N < 0 || P < 0) return fail(ctx, MCR_EINVAL, "negative dimension (C=%lld N=%lld P=%lld)", C, N, P);
    if (min_chains < 1) return fail(ctx, MCR_EMINCHAINS_ARG, "min_chains must be >= 1; got %d", min_chains);
    if (C < min_chains)
        return fail(ctx, MCR_EMINCHAINS, "diagnostics require at least %d chains; got %lld chain(s)", min_chains, C);
    if (C > kMaxChains) return fail(ctx, MCR_EINVAL, "at most %d chains are supported; got %lld", kMaxChains, C);
    if (C * N >= (i64)0x7FFFFFFFll) return fail(ctx, MCR_EINVAL, "C*N must be < 2^31");   // 2M-1 rank codes in u32
    if (!draws && C * N * P > 0) return fail(ctx, MCR_EINVAL, "draws is NULL");
    return MCR_OK;
}

// How a call of this shape is cut into workspace chunks of parameters (a pure function of the shape, the workspace limit
// and MCR_FFT): shared by enqueue_impl and mcr_plan_chunks.
struct ChunkPlan { bool ingest = false; WsPlan wp{}; FftPlan fp{}; size_t slack = 0; i64 pcmax = 0; };
int plan_chunks(mcr_ctx* ctx, i64 C, i64 N, i64 P, i64 sc, i64 sn, i64 sp, bool do_diag, ChunkPlan& cp)
{
    const i64 M = C * N;
    // tensors in the Arrow column layout [P][C][N] are consumed in place, f64 and f32 alike (the tile sort widens f32
    // as it loads); anything else goes through one ingest pass into X[P][M] f64
    cp.ingest = !((N <= 1 || sn == 1) && (C <= 1 || sc == N) && (P <= 1 || sp == M));
    cp.wp = plan_ws(M, (int)C, cp.ingest, false, N);
    cp.fp = plan_fft(N, (int)C, do_diag && ctx->fft_on);
    // the FFT tier is an accelerator, not a requirement: under a tight workspace limit it gets fewer slots, or none
    // (the direct rounds then serve every listed pair), but at least a third of the limit stays with the parameters
    while (cp.fp.on && cp.fp.bytes > ctx->ws_limit / 3) {
        if (cp.fp.slots <= 1) { cp.fp = FftPlan{}; break; }
        cp.fp.slots /= 2;
        const size_t Nf = (size_t)1 << cp.fp.logN;
        cp.fp.bytes = (size_t)cp.fp.slots * ((size_t)cp.fp.cb * Nf * 16 + Nf * 8 + Nf * 16) + 3 * 256;
    }
    cp.slack = 40 * 256 + cp.fp.bytes;
    if (cp.wp.per_param + cp.slack > ctx->ws_limit)
        return fail(ctx, MCR_ENOMEM, "one parameter needs %zu bytes of workspace (%zu for the parameter + %zu shared, of which %zu "
                    "for the FFT tier); limit is %zu", cp.wp.per_param + cp.slack, cp.wp.per_param, cp.slack, cp.fp.bytes, ctx->ws_limit);
    i64 pcmax = (i64)((ctx->ws_limit - cp.slack) / cp.wp.per_param);
    if (pcmax > P) pcmax = P;
    if (pcmax > kMaxGridY / 2) pcmax = kMaxGridY / 2;   // k_acov_seg uses grid.z = 2 * pc
    cp.pcmax = pcmax;
    return MCR_OK;
}

int enqueue_impl(mcr_ctx* ctx, const void* draws_dev, int dtype, i64 C, i64 N, i64 P, i64 sc, i64 sn, i64 sp,
                 int min_chains, const double* quantiles, int nq, const mcr_summary* out)
{
    int rc = check_common(ctx, draws_dev, dtype, C, N, P, min_chains, out);
    if (rc) return rc;
    if (ctx->n_inflight >= MCR_MAX_INFLIGHT) return fail(ctx, MCR_EINVAL, "more than %d summaries in flight", MCR_MAX_INFLIGHT);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const i64 M = C * N;
    int si = -1;
    for (int k = 0; k < MCR_MAX_INFLIGHT; ++k) {
        const int c = (ctx->next_slot + k) % MCR_MAX_INFLIGHT;
        if (!ctx->slots[c].busy) { si = c; break; }
    }
    Slot& s = ctx->slots[si];
    QArgs q;
    rc = prep_quantiles(ctx, quantiles, nq, M, q, s.qlo);
    if (rc) return rc;
    use_lane(ctx, si % ctx->n_lanes);
    s.out = *out; s.P = P; s.M = M; s.nq = nq; s.C = (int)C;
    s.chunks.clear();
    s.trivial_nan = (M == 0 || P == 0);
    if (!s.trivial_nan) {
        const bool do_diag_early = out->rhat || out->rhat_bulk || out->rhat_tail || out->ess_bulk || out->ess_tail ||
                                   out->lag_bulk || out->lag_tail;
        ChunkPlan cp;
        rc = plan_chunks(ctx, C, N, P, sc, sn, sp, do_diag_early, cp);
        if (rc) return rc;
        const bool ingest = cp.ingest;
        const WsPlan& wp = cp.wp;
        const FftPlan& fp = cp.fp;
        const size_t slack = cp.slack;
        const i64 pcmax = cp.pcmax;
        double2 *tw1 = nullptr, *tw2 = nullptr;
        if (fp.on) {
            rc = get_twiddles(ctx, 1 << fp.log1, &tw1);
            if (!rc) rc = get_twiddles(ctx, 1 << fp.log2, &tw2);
            if (rc) return rc;
        }
        rc = ensure_ws(ctx, (size_t)pcmax * wp.per_param + slack);
        if (rc) return rc;
        const int R = res_fields(nq);
        rc = ensure_slot(ctx, s, (size_t)R * (size_t)P, (size_t)C + 1);
        if (rc) return rc;
        const bool off_resident = s.off_C == C && s.off_N == N;
        if (!off_resident) for (i64 c = 0; c <= C; ++c) s.h_off[c] = c * N;
        const bool do_diag = out->rhat || out->rhat_bulk || out->rhat_tail || out->ess_bulk || out->ess_tail ||
                             out->lag_bulk || out->lag_tail;
        double* ztab = nullptr;
        if (do_diag) { rc = get_ztab(ctx, M, &ztab); if (rc) return rc; }
        // Everything below only enqueues stream work with arguments that are a pure function of `key`.
        auto issue = [&]() -> int {
            if (!off_resident)
                HIP_TRY(ctx, hipMemcpyAsync(s.d_off, s.h_off, sizeof(i64) * (size_t)(C + 1), hipMemcpyHostToDevice, ctx->stream));
            for (i64 p0 = 0; p0 < P; p0 += pcmax) {
                const i64 pc = (P - p0 < pcmax) ? P - p0 : pcmax;
                Carve cv{reinterpret_cast<char*>(ctx->ws)};
                PipeIn a{};
                a.M = M; a.pc = pc; a.C = (int)C; a.d_off = s.d_off; a.n = N; a.nh = (N >= 2) ? N / 2 : 0; a.q = q;
                a.ntiles = wp.ntiles;
                a.kA = cv.take<double>((size_t)pc * M); a.kB = cv.take<double>((size_t)pc * M);
                a.iA = cv.take<u32>((size_t)pc * M);    a.iB = cv.take<u32>((size_t)pc * M);
                a.zb = cv.take<u32>((size_t)pc * M); a.zt = cv.take<u32>((size_t)pc * M);
                a.part = cv.take<double>((size_t)pc * wp.ntiles * kMomRec);
                a.split = cv.take<i64>((size_t)pc);
                {
                    const size_t cc = (size_t)(C > 0 ? C : 1), ns = (size_t)((N + kSeg - 1) / kSeg + 1);
                    a.rec = cv.take<double>((size_t)pc * 2 * cc * ns * kSegRec);
                    a.rec2 = cv.take<double>((size_t)pc * 2 * cc * ns * 64 * kMoreBlocks);
                    a.chstate = cv.take<double>((size_t)pc * 2 * cc * kChState);
                    a.more = cv.take<unsigned>((size_t)pc * 2);
                    a.state = cv.take<double>((size_t)pc * 2 * kPairState);
                    a.acov = cv.take<double>((size_t)pc * 2 * (size_t)(N > 0 ? N : 1));
                    a.long_count = cv.take<unsigned>(1);
                    a.long_list = cv.take<unsigned>((size_t)pc * 2);
                    a.t3c = cv.take<unsigned>((size_t)pc * 2 * (kT3Stages + 1));
                }
                a.nstage = N > 0 ? N : 1;
                a.ztab = ztab;
                a.samp = cv.take<double>((size_t)pc * (wp.ntiles + 16) * 64);
                a.cut = cv.take<u32>((size_t)pc * (wp.bk_B + 1) * (size_t)(wp.bk_k + 1));
                a.boff = cv.take<u32>((size_t)pc * (wp.bk_B + 1));
                a.bk_B = wp.bk_B; a.bk_D = wp.bk_D; a.bk_k = wp.bk_k; a.bk_R = wp.bk_R;
                a.do_diag = do_diag;
                // (a graph capture or the per-kernel event pairs of the profiling mode keep the single stream)
                a.fork = ctx->fork_lone && ctx->n_inflight == 0 && !ctx->prof && !ctx->graph_on && P <= pcmax &&
                         (double)M * (double)pc >= 2e6 && (double)M * (double)pc <= 8e6;    // kernels long enough to be worth
                                                   // three more launches and two event waits (measured: C1 4 M param-draws 306 -> 288 us,
                                                   // 10 x 1000 x 45 149 -> 166, 4 x 1000 x 10 124 -> 133), short enough not to fill the chip
                a.fft = fp; a.tw1 = tw1; a.tw2 = tw2;
                if (fp.on) {
                    const size_t Nf = (size_t)1 << fp.logN;
                    a.fft_A = cv.take<double2>((size_t)fp.slots * fp.cb * Nf);
                    a.fft_S = cv.take<double>((size_t)fp.slots * Nf);
                    a.fft_B = cv.take<double2>((size_t)fp.slots * Nf);
                }
                if (ingest) {
                    double* X = cv.take<double>((size_t)pc * M);
                    const int r2 = (dtype == MCR_F64) ? launch_ingest<double>(ctx, draws_dev, X, C, N, pc, sc, sn, sp, p0)
                                                      : launch_ingest<float>(ctx, draws_dev, X, C, N, pc, sc, sn, sp, p0);
                    if (r2) return r2;
                    a.X = X;
                } else {
                    a.x_f32 = dtype == MCR_F32;
                    a.no_records = !ctx->f32_records;
                    a.X = a.x_f32 ? (const void*)(reinterpret_cast<const float*>(draws_dev) + p0 * M)
                                  : (const void*)(reinterpret_cast<const double*>(draws_dev) + p0 * M);
                }
                const size_t res_off = (size_t)R * (size_t)p0;
                a.d_res = s.d_res + res_off;
                const int r3 = run_pipeline(ctx, a);
                if (r3) return r3;
                s.chunks.push_back(Chunk{p0, pc, res_off});
            }
            HIP_TRY(ctx, hipMemcpyAsync(s.h_res, s.d_res, sizeof(double) * (size_t)R * (size_t)P, hipMemcpyDeviceToHost,
                                        ctx->stream));
            return MCR_OK;
        };
        if (ctx->graph_on && !ctx->prof) {
            std::vector<uint64_t> key = {(uint64_t)(uintptr_t)draws_dev, (uint64_t)dtype, (uint64_t)C, (uint64_t)N,
                                         (uint64_t)P, (uint64_t)sc, (uint64_t)sn, (uint64_t)sp, (uint64_t)nq,
                                         (uint64_t)si, (uint64_t)do_diag, (uint64_t)(uintptr_t)ctx->ws,
                                         (uint64_t)(uintptr_t)s.d_res, (uint64_t)(uintptr_t)s.d_off,
                                         (uint64_t)pcmax, (uint64_t)(uintptr_t)ztab, (uint64_t)off_resident};
            for (int k = 0; k < nq; ++k) {
                uint64_t bits; memcpy(&bits, &q.g[k], 8);
                key.push_back((uint64_t)q.lo[k]); key.push_back(bits);
            }
            GraphEntry* hit = nullptr;
            for (GraphEntry& g : ctx->graphs) if (g.key == key) { hit = &g; break; }
            if (!hit) {
                if (ctx->graphs.size() >= 64) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); drop_graphs(ctx); }
                HIP_TRY(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
                rc = issue();
                hipGraph_t graph = nullptr;
                const hipError_t ce = hipStreamEndCapture(ctx->stream, &graph);
                if (rc) { if (graph) hipGraphDestroy(graph); return rc; }
                if (ce != hipSuccess) return fail(ctx, MCR_EHIP, "hipStreamEndCapture failed: %s", hipGetErrorString(ce));
                GraphEntry ge;
                ge.key = key;
                const hipError_t ie = hipGraphInstantiate(&ge.exec, graph, nullptr, nullptr, 0);
                hipGraphDestroy(graph);
                if (ie != hipSuccess) return fail(ctx, MCR_EHIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ie));
                ge.chunks = s.chunks;
                ctx->graphs.push_back(std::move(ge));
                hit = &ctx->graphs.back();
            }
            s.chunks = hit->chunks;
            HIP_TRY(ctx, hipGraphLaunch(hit->exec, ctx->stream));
        } else {
            rc = issue();
            if (rc) return rc;
        }
    }
    if (!s.trivial_nan) { s.off_C = C; s.off_N = N; }
    if (!s.done) HIP_TRY(ctx, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventRecord(s.done, ctx->stream));
    s.lane = ctx->lane;
    s.busy = true;
    ctx->order.push_back(si);
    ctx->n_inflight++;
    ctx->next_slot = (si + 1) % MCR_MAX_INFLIGHT;
    return MCR_OK;
}

// Waits for a slot's completion event.  The host polls the event for a while before it blocks in the runtime: a blocking
// wait is woken through an interrupt and the scheduler, which costs a lone synchronous call (what reference.compare makes,
// src/mcmc_ref/reference.py:107-122) some 20 - 40 us on top of its 0.3 ms; a pipelined caller seldom waits at all.
int wait_event(mcr_ctx* ctx, hipEvent_t ev)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (int spin = 0;; ++spin) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) return MCR_OK;
        if (q != hipErrorNotReady) return fail(ctx, MCR_EHIP, "hipEventQuery failed: %s", hipGetErrorString(q));
        if ((spin & 63) == 63 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2e-3) break;
    }
    HIP_TRY(ctx, hipEventSynchronize(ev));
    return MCR_OK;
}

// Waits for the OLDEST outstanding enqueue only and delivers its results; the others keep running.
int wait_one_impl(mcr_ctx* ctx)
{
    if (ctx->order.empty()) return MCR_OK;
    const int si = ctx->order.front();
    Slot& s = ctx->slots[si];
    { const int rc = wait_event(ctx, s.done); if (rc) return rc; }
    const int rc = unpack_slot(ctx, s);
    s.busy = false;
    ctx->order.erase(ctx->order.begin());
    ctx->n_inflight--;
    return rc;
}

int wait_impl(mcr_ctx* ctx)
{
    // every slot in flight records its event behind its last copy on its lane: waiting for the events (polled, see
    // wait_event) leaves every lane idle; the stream synchronisations below then return at once
    // (slots filled by mcr_diagnose_chains carry no event of their own: the stream synchronisation waits for those)
    for (int si : ctx->order)
        if (ctx->slots[si].done) { const int rc = wait_event(ctx, ctx->slots[si].done); if (rc) return rc; }
    for (hipStream_t st : ctx->lane_stream) if (st) HIP_TRY(ctx, hipStreamSynchronize(st));
    use_lane(ctx, 0);
    prof_resolve(ctx);
    int rc = MCR_OK;
    for (int si : ctx->order) {
        Slot& s = ctx->slots[si];
        const int r = unpack_slot(ctx, s);
        if (r && !rc) rc = r;
        s.busy = false;
    }
    ctx->order.clear();
    ctx->n_inflight = 0;
    return rc;
}

// Drops any enqueued-but-unwaited work after an error so the ctx stays usable.
void abort_inflight(mcr_ctx* ctx)
{
    sync_all(ctx);
    use_lane(ctx, 0);
    prof_resolve(ctx);
    for (int si : ctx->order) ctx->slots[si].busy = false;
    ctx->order.clear();
    ctx->n_inflight = 0;
}

int ensure_stage(mcr_ctx* ctx, size_t bytes)
{
    if (bytes <= ctx->stage_bytes) return MCR_OK;
    sync_all(ctx);
    if (ctx->stage) { hipFree(ctx->stage); ctx->stage = nullptr; ctx->stage_bytes = 0; }
    HIP_TRY(ctx, hipMalloc(&ctx->stage, bytes));
    ctx->stage_bytes = bytes;
    return MCR_OK;
}

// Extent (in elements) touched by a strided tensor, or -1 if strides are negative.
i64 tensor_extent(i64 C, i64 N, i64 P, i64 sc, i64 sn, i64 sp)
{
    if (C == 0 || N == 0 || P == 0) return 0;
    if (sc < 0 || sn < 0 || sp < 0) return -1;
    return (C - 1) * sc + (N - 1) * sn + (P - 1) * sp + 1;
}

template <typename T>
int moments_impl(mcr_ctx* ctx, const T* src, i64 C, i64 N, i64 P, i64 sc, i64 sn, i64 sp, double* d_mean,
                 double* d_std, double* part, int S, bool rows)
{
    const i64 M = C * N;
    if (rows) {
        LAUNCH(ctx, K_MOMENTS, (k_moments_rows<T>), dim3((unsigned)S, (unsigned)P), dim3(256), 0, src, M, sp, part, S);
    } else {
        LAUNCH(ctx, K_MOMENTS, (k_moments_cols<T>), dim3((unsigned)((P + 63) / 64), (unsigned)S), dim3(256), 0, src,
               C, N, P, sc, sn, sp, part, S);
    }
    LAUNCH(ctx, K_MOMENTS_FINAL, k_moments_final, dim3((unsigned)((P + 255) / 256)), dim3(256), 0,
           (const double*)part, S, P, d_mean, d_std, (double*)nullptr);
    return MCR_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

int mcr_version(void) { return MCR_VERSION; }

int mcr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* mcr_last_error(const mcr_ctx* ctx) { return ctx ? ctx->err : g_init_err; }

int mcr_init(int device, mcr_ctx** out)
{
    if (!out) return fail(nullptr, MCR_EINVAL, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, MCR_ENODEVICE, "no HIP device available (%s); libmcmcref_hip has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0 || device >= n) return fail(nullptr, MCR_ENODEVICE, "device %d out of range [0, %d)", device, n);
    mcr_ctx* ctx = new (std::nothrow) mcr_ctx();
    if (!ctx) return fail(nullptr, MCR_ENOMEM, "out of host memory");
    ctx->device = device;
    if (const char* env = getenv("MCR_LANES")) {
        const int v = atoi(env);
        ctx->n_lanes = v < 1 ? 1 : (v > MCR_MAX_INFLIGHT ? MCR_MAX_INFLIGHT : v);
    }
    if (const char* env = getenv("MCR_IO_PIECE_KB")) {
        const long v = atol(env);
        ctx->io_piece = (size_t)(v < 64 ? 64 : (v > (1 << 20) ? (1 << 20) : v)) << 10;
    }
    if (const char* env = getenv("MCR_IO_THREADS")) {
        const int v = atoi(env);
        ctx->io_threads = v < 1 ? 1 : (v > 64 ? 64 : v);
    }
    e = hipSetDevice(device);
    for (int l = 0; l < ctx->n_lanes && e == hipSuccess; ++l)
        e = hipStreamCreateWithFlags(&ctx->lane_stream[l], hipStreamNonBlocking);
    for (int l = 0; l < ctx->n_lanes && e == hipSuccess; ++l) {
        e = hipStreamCreateWithFlags(&ctx->lane_aux[l], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->lane_fork[l], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->lane_join[l], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
        for (hipStream_t st : ctx->lane_stream) if (st) hipStreamDestroy(st);
        for (hipStream_t st : ctx->lane_aux) if (st) hipStreamDestroy(st);
        for (hipEvent_t ev : ctx->lane_fork) if (ev) hipEventDestroy(ev);
        for (hipEvent_t ev : ctx->lane_join) if (ev) hipEventDestroy(ev);
        delete ctx;
        return fail(nullptr, MCR_EHIP, "device %d init failed: %s", device, hipGetErrorString(e));
    }
    ctx->stream = ctx->lane_stream[0];
    size_t mb = 8192;
    if (const char* env = getenv("MCR_WORKSPACE_MB")) { const long v = atol(env); if (v > 0) mb = (size_t)v; }
    ctx->ws_limit = mb << 20;
    if (const char* env = getenv("MCR_GRAPH")) ctx->graph_on = atoi(env) != 0;
    if (const char* env = getenv("MCR_F32_RECORDS")) ctx->f32_records = atoi(env) != 0;
    if (const char* env = getenv("MCR_FFT")) ctx->fft_on = atoi(env) != 0;
    if (const char* env = getenv("MCR_SPLITTERS_PAIRWISE")) ctx->splitters_pairwise = atoi(env) != 0;
    if (const char* env = getenv("MCR_FORK")) ctx->fork_lone = atoi(env) != 0;
    if (const char* env = getenv("MCR_T3_WG")) { const int v = atoi(env); if (v > 0) ctx->t3_workgroups = v; }
    if (const char* env = getenv("MCR_DBG_LDS_PAD")) {
        unsigned long t = 0, b = 0, f = 0;
        if (sscanf(env, "%lu,%lu,%lu", &t, &b, &f) >= 1) { ctx->dbg_lds_pad[0] = t; ctx->dbg_lds_pad[1] = b; ctx->dbg_lds_pad[2] = f; }
    }
    if (const char* env = getenv("MCR_RHO_BAND")) { const double v = atof(env); if (v >= 0.0 && v < 1.0) ctx->rho_band = v; }
    if (hipMalloc((void**)&ctx->guard_count, sizeof(unsigned)) != hipSuccess ||
        hipMemsetAsync(ctx->guard_count, 0, sizeof(unsigned), ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
        mcr_free(ctx);
        return fail(nullptr, MCR_ENOMEM, "device %d: cannot allocate the guard counter", device);
    }
    if (const char* env = getenv("MCR_SORT_CFG")) {
        const int v = atoi(env);
        if (v >= 0 && v % 10 <= 1 && v / 10 <= 2) ctx->sort_cfg = v;
    }
    *out = ctx;
    return MCR_OK;
}

void mcr_free(mcr_ctx* ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    sync_all(ctx);
    use_lane(ctx, 0);
    prof_resolve(ctx);
    drop_graphs(ctx);
    for (hipEvent_t e : ctx->free_ev) hipEventDestroy(e);
    for (Slot& s : ctx->slots) {
        if (s.done) hipEventDestroy(s.done);
        if (s.d_res) hipFree(s.d_res);
        if (s.h_res) hipHostFree(s.h_res);
        if (s.d_off) hipFree(s.d_off);
        if (s.h_off) hipHostFree(s.h_off);
    }
    ctx->lane_ws[0] = ctx->ws;
    for (void* w : ctx->lane_ws) if (w) hipFree(w);
    for (const mcr_ctx::ZTab& z : ctx->ztabs) hipFree(z.tab);
    for (const mcr_ctx::Twiddle& t : ctx->twiddles) hipFree(t.tab);
    if (ctx->guard_count) hipFree(ctx->guard_count);
    if (ctx->stage) hipFree(ctx->stage);
    if (ctx->pq_stage) hipFree(ctx->pq_stage);
    if (ctx->pq_scratch) hipFree(ctx->pq_scratch);
    if (ctx->pq_tab) hipFree(ctx->pq_tab);
    if (ctx->fs_arena) hipFree(ctx->fs_arena);
    if (ctx->pq_pin) hipHostFree(ctx->pq_pin);
    for (hipStream_t st : ctx->lane_stream) if (st) hipStreamDestroy(st);
    for (hipStream_t st : ctx->lane_aux) if (st) hipStreamDestroy(st);
    for (hipEvent_t ev : ctx->lane_fork) if (ev) hipEventDestroy(ev);
    for (hipEvent_t ev : ctx->lane_join) if (ev) hipEventDestroy(ev);
    if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
    delete ctx;
}

int mcr_rho_guard_count(mcr_ctx* ctx, int64_t* rederived)
{
    if (!ctx || !rederived) return fail(ctx, MCR_EINVAL, "NULL argument");
    hipSetDevice(ctx->device);
    sync_all(ctx);
    unsigned v = 0;
    HIP_TRY(ctx, hipMemcpy(&v, ctx->guard_count, sizeof(unsigned), hipMemcpyDeviceToHost));
    *rederived = (int64_t)v;
    return MCR_OK;
}

int mcr_set_workspace_limit(mcr_ctx* ctx, size_t bytes)
{
    if (!ctx) return fail(nullptr, MCR_EINVAL, "ctx is NULL");
    if (bytes < (1u << 20)) return fail(ctx, MCR_EINVAL, "workspace limit must be at least 1 MiB");
    ctx->ws_limit = bytes;
    return MCR_OK;
}

int mcr_plan_chunks(mcr_ctx* ctx, int64_t C, int64_t N, int64_t P, int64_t sc, int64_t sn, int64_t sp, int diagnostics,
                    int64_t* params_per_chunk)
{
    if (!ctx || !params_per_chunk) return fail(ctx, MCR_EINVAL, "mcr_plan_chunks: NULL argument");
    if (C <= 0 || N <= 0 || P <= 0) { *params_per_chunk = 0; return MCR_OK; }
    ChunkPlan cp;
    const int rc = plan_chunks(ctx, C, N, P, sc, sn, sp, diagnostics != 0, cp);
    if (rc) return rc;
    *params_per_chunk = cp.pcmax;
    return MCR_OK;
}

int mcr_dev_alloc(mcr_ctx* ctx, size_t bytes, void** dptr)
{
    if (!ctx || !dptr) return fail(ctx, MCR_EINVAL, "NULL argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMalloc(dptr, bytes ? bytes : 1));
    return MCR_OK;
}
int mcr_dev_free(mcr_ctx* ctx, void* dptr)
{
    if (!ctx) return fail(nullptr, MCR_EINVAL, "ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (dptr) HIP_TRY(ctx, hipFree(dptr));
    return MCR_OK;
}
int mcr_memcpy_h2d(mcr_ctx* ctx, void* dptr, const void* hptr, size_t bytes)
{
    if (!ctx) return fail(nullptr, MCR_EINVAL, "ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(dptr, hptr, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MCR_OK;
}
int mcr_memcpy_d2h(mcr_ctx* ctx, void* hptr, const void* dptr, size_t bytes)
{
    if (!ctx) return fail(nullptr, MCR_EINVAL, "ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(hptr, dptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MCR_OK;
}
int mcr_sync(mcr_ctx* ctx)
{
    if (!ctx) return fail(nullptr, MCR_EINVAL, "ctx is NULL");
    for (hipStream_t st : ctx->lane_stream) if (st) HIP_TRY(ctx, hipStreamSynchronize(st));
    return MCR_OK;
}

int mcr_summarize_enqueue(mcr_ctx* ctx, const void* draws_dev, int dtype, int64_t C, int64_t N, int64_t P,
                          int64_t sc, int64_t sn, int64_t sp, int min_chains, const double* quantiles, int n_q,
                          mcr_summary* out)
{
    const int rc = enqueue_impl(ctx, draws_dev, dtype, C, N, P, sc, sn, sp, min_chains, quantiles, n_q, out);
    if (rc && ctx && rc != MCR_EMINCHAINS && rc != MCR_EMINCHAINS_ARG && rc != MCR_EINVAL) {
        char keep[512];
        memcpy(keep, ctx->err, sizeof keep);
        abort_inflight(ctx);
        memcpy(ctx->err, keep, sizeof keep);
    }
    return rc;
}

int mcr_summarize_wait(mcr_ctx* ctx)
{
    if (!ctx) return fail(nullptr, MCR_EINVAL, "ctx is NULL");
    return wait_impl(ctx);
}

int mcr_summarize_wait_one(mcr_ctx* ctx)
{
    if (!ctx) return fail(nullptr, MCR_EINVAL, "ctx is NULL");
    return wait_one_impl(ctx);
}

int mcr_summarize_models(mcr_ctx* ctx, const mcr_model_desc* models, int n_models, const double* quantiles,
                         int n_q, mcr_summary* outs)
{
    if (!ctx) return fail(nullptr, MCR_EINVAL, "ctx is NULL");
    if (n_models < 0 || (n_models > 0 && (!models || !outs))) return fail(ctx, MCR_EINVAL, "bad argument");
    int rc = MCR_OK;
    for (int i = 0; i < n_models && !rc; ++i) {
        if (ctx->n_inflight >= MCR_MAX_INFLIGHT) rc = wait_one_impl(ctx);
        if (rc) break;
        const mcr_model_desc& m = models[i];
        rc = mcr_summarize_enqueue(ctx, m.draws_dev, m.dtype, m.C, m.N, m.P, m.stride_c, m.stride_n, m.stride_p,
                                   m.min_chains, quantiles, n_q, &outs[i]);
    }
    char keep[512];
    memcpy(keep, ctx->err, sizeof keep);
    const int rw = wait_impl(ctx);          // deliver everything that was enqueued
    if (rc) { memcpy(ctx->err, keep, sizeof keep); return rc; }
    return rw;
}

int mcr_summarize_dev(mcr_ctx* ctx, const void* draws_dev, int dtype, int64_t C, int64_t N, int64_t P, int64_t sc,
                      int64_t sn, int64_t sp, int min_chains, const double* quantiles, int n_q, mcr_summary* out)
{
    int rc = mcr_summarize_enqueue(ctx, draws_dev, dtype, C, N, P, sc, sn, sp, min_chains, quantiles, n_q, out);
    if (rc) return rc;
    return wait_impl(ctx);
}

int mcr_summarize(mcr_ctx* ctx, const void* draws, int dtype, int64_t C, int64_t N, int64_t P, int64_t sc,
                  int64_t sn, int64_t sp, int min_chains, const double* quantiles, int n_q, mcr_summary* out)
{
    int rc = check_common(ctx, draws, dtype, C, N, P, min_chains, out);
    if (rc) return rc;
    const i64 ext = tensor_extent(C, N, P, sc, sn, sp);
    if (ext < 0) return fail(ctx, MCR_EINVAL, "negative strides are not supported");
    const size_t es = dtype == MCR_F64 ? 8 : 4;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ext > 0) {
        rc = ensure_stage(ctx, (size_t)ext * es);
        if (rc) return rc;
    }
    // Parameter-major tensors (the Arrow column layout) of some size go up in a few pieces: while piece k + 1
    // crosses PCIe on the copy stream, the kernels of piece k already run on a lane (pageable host memory makes
    // the copy itself synchronous for the host, which has nothing better to do).
    const i64 per_param = tensor_extent(C, N, 1, sc, sn, 0);
    const bool separable = P >= 2 && C * N > 0 && sp >= per_param && ctx->n_inflight == 0 && ctx->n_lanes > 1;
    if (separable && (size_t)ext * es >= ((size_t)8 << 20)) {
        int pieces = ctx->n_lanes < 4 ? ctx->n_lanes : 4;
        if ((i64)pieces > P) pieces = (int)P;
        const int nqq = n_q > 0 ? n_q : 0;
        for (int k = 0; k < pieces && !rc; ++k) {
            const i64 p0 = P * k / pieces, p1 = P * (k + 1) / pieces, pc = p1 - p0;
            const size_t b0 = (size_t)p0 * (size_t)sp * es;
            const size_t bytes = (size_t)tensor_extent(C, N, pc, sc, sn, sp) * es;
            hipError_t e = hipMemcpyAsync((char*)ctx->stage + b0, (const char*)draws + b0, bytes, hipMemcpyHostToDevice, ctx->copy_stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->copy_stream);
            if (e != hipSuccess) { rc = fail(ctx, MCR_EHIP, "upload failed: %s", hipGetErrorString(e)); break; }
            mcr_summary o = *out;
            auto adv = [&](double*& q, i64 n) { if (q) q += n; };
            adv(o.mean, p0); adv(o.std, p0); adv(o.q, p0 * nqq); adv(o.median, p0); adv(o.rhat, p0); adv(o.rhat_bulk, p0);
            adv(o.rhat_tail, p0); adv(o.ess_bulk, p0); adv(o.ess_tail, p0);
            if (o.lag_bulk) o.lag_bulk += p0;
            if (o.lag_tail) o.lag_tail += p0;
            rc = mcr_summarize_enqueue(ctx, (const char*)ctx->stage + b0, dtype, C, N, pc, sc, sn, sp, min_chains, quantiles, n_q, &o);
        }
        char keep[512];
        memcpy(keep, ctx->err, sizeof keep);
        const int rw = wait_impl(ctx);
        if (rc) { memcpy(ctx->err, keep, sizeof keep); return rc; }
        return rw;
    }
    if (ext > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->stage, draws, (size_t)ext * es, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // the pipeline may run on the other lane
    }
    return mcr_summarize_dev(ctx, ctx->stage, dtype, C, N, P, sc, sn, sp, min_chains, quantiles, n_q, out);
}

int mcr_diagnose_chains(mcr_ctx* ctx, const double* pooled, const int64_t* chain_off, int C, int min_chains,
                        mcr_summary* out, double* z_bulk, double* z_tail, double* rank_bulk, double* rank_tail)
{
    if (!ctx) return fail(nullptr, MCR_EINVAL, "ctx is NULL");
    if (!out) return fail(ctx, MCR_EINVAL, "out is NULL");
    if (C < 0) return fail(ctx, MCR_EINVAL, "negative chain count");
    if (min_chains < 1) return fail(ctx, MCR_EMINCHAINS_ARG, "min_chains must be >= 1; got %d", min_chains);
    if (C < min_chains) return fail(ctx, MCR_EMINCHAINS, "diagnostics require at least %d chains; got %d chain(s)", min_chains, C);
    if (C > kMaxChains) return fail(ctx, MCR_EINVAL, "at most %d chains are supported; got %d", kMaxChains, C);
    if (C > 0 && !chain_off) return fail(ctx, MCR_EINVAL, "chain_off is NULL");
    if (ctx->n_inflight) return fail(ctx, MCR_EINVAL, "mcr_diagnose_chains with summaries in flight");
    const i64 M = C > 0 ? chain_off[C] : 0;
    if (M < 0 || M >= (i64)0x7FFFFFFFll) return fail(ctx, MCR_EINVAL, "pooled length out of range");
    i64 n = 0, nh = 0;
    bool have_h = false;
    for (int c = 0; c < C; ++c) {
        const i64 len = chain_off[c + 1] - chain_off[c];
        if (len < 0) return fail(ctx, MCR_EINVAL, "chain_off must be non-decreasing");
        if (c == 0 || len < n) n = len;
        if (len >= 2) { if (!have_h || len / 2 < nh) nh = len / 2; have_h = true; }
    }
    if (M > 0 && !pooled) return fail(ctx, MCR_EINVAL, "pooled is NULL");
    Slot& s = ctx->slots[0];
    s.out = *out; s.P = 1; s.M = M; s.nq = 0; s.C = C; s.chunks.clear();
    s.trivial_nan = (M == 0);
    if (s.trivial_nan) { s.busy = true; ctx->order.push_back(0); ctx->n_inflight = 1; return wait_impl(ctx); }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const bool want_dbg = z_bulk || z_tail || rank_bulk || rank_tail;
    i64 nstage = n > 0 ? n : 1;
    for (int c = 0; c < C; ++c) {
        const i64 len = chain_off[c + 1] - chain_off[c];
        if (len >= 2 && len / 2 + nh > nstage) nstage = len / 2 + nh;
    }
    const WsPlan wp = plan_ws(M, C, true, want_dbg, nstage);
    FftPlan fp = plan_fft(n, C, ctx->fft_on);
    fp.slots = fp.on ? 2 : 0;                       // one parameter: two (parameter, kind) pairs at most
    if (fp.on) { const size_t Nf = (size_t)1 << fp.logN; fp.bytes = (size_t)fp.slots * ((size_t)fp.cb * Nf * 16 + Nf * 8 + Nf * 16) + 3 * 256; }
    if (fp.on && wp.per_param + 40 * 256 + fp.bytes > ctx->ws_limit) fp = FftPlan{};
    const size_t slack = 40 * 256 + fp.bytes;
    if (wp.per_param + slack > ctx->ws_limit) return fail(ctx, MCR_ENOMEM, "workspace limit too small for %lld draws", M);
    int rc = ensure_ws(ctx, wp.per_param + slack);
    if (rc) return rc;
    const int R = res_fields(0);
    rc = ensure_slot(ctx, s, (size_t)R, (size_t)C + 1);
    if (rc) return rc;
    s.off_C = s.off_N = -1;
    memcpy(s.h_off, chain_off, sizeof(i64) * (size_t)(C + 1));
    HIP_TRY(ctx, hipMemcpyAsync(s.d_off, s.h_off, sizeof(i64) * (size_t)(C + 1), hipMemcpyHostToDevice, ctx->stream));
    Carve cv{reinterpret_cast<char*>(ctx->ws)};
    PipeIn a{};
    a.M = M; a.pc = 1; a.C = C; a.d_off = s.d_off; a.n = n; a.nh = nh; a.q.nq = 0; a.ntiles = wp.ntiles;
    a.kA = cv.take<double>((size_t)M); a.kB = cv.take<double>((size_t)M);
    a.iA = cv.take<u32>((size_t)M);    a.iB = cv.take<u32>((size_t)M);
    a.zb = cv.take<u32>((size_t)M); a.zt = cv.take<u32>((size_t)M);
    a.part = cv.take<double>((size_t)wp.ntiles * kMomRec);
    a.split = cv.take<i64>(1);
    {
        const size_t cc = (size_t)(C > 0 ? C : 1), ns = (size_t)((nstage + kSeg - 1) / kSeg + 1);
        a.rec = cv.take<double>((size_t)2 * cc * ns * kSegRec);
        a.rec2 = cv.take<double>((size_t)2 * cc * ns * 64 * kMoreBlocks);
        a.chstate = cv.take<double>((size_t)2 * cc * kChState);
        a.more = cv.take<unsigned>(2);
        a.state = cv.take<double>(2 * kPairState);
        a.acov = cv.take<double>((size_t)2 * (size_t)(n > 0 ? n : 1));
        a.long_count = cv.take<unsigned>(1);
        a.long_list = cv.take<unsigned>(2);
        a.t3c = cv.take<unsigned>(2 * (kT3Stages + 1));
    }
    a.nstage = nstage;
    rc = get_ztab(ctx, M, &a.ztab);
    if (rc) return rc;
    a.samp = cv.take<double>((size_t)(wp.ntiles + 16) * 64);
    a.cut = cv.take<u32>((size_t)(wp.bk_B + 1) * (size_t)(wp.bk_k + 1));
    a.boff = cv.take<u32>((size_t)(wp.bk_B + 1));
    a.bk_B = wp.bk_B; a.bk_D = wp.bk_D; a.bk_k = wp.bk_k; a.bk_R = wp.bk_R;
    a.fft = fp;
    if (fp.on) {
        rc = get_twiddles(ctx, 1 << fp.log1, &a.tw1);
        if (!rc) rc = get_twiddles(ctx, 1 << fp.log2, &a.tw2);
        if (rc) return rc;
        const size_t Nf = (size_t)1 << fp.logN;
        a.fft_A = cv.take<double2>((size_t)fp.slots * fp.cb * Nf);
        a.fft_S = cv.take<double>((size_t)fp.slots * Nf);
        a.fft_B = cv.take<double2>((size_t)fp.slots * Nf);
    }
    double* X = cv.take<double>((size_t)M);
    double* dbg[4] = {nullptr, nullptr, nullptr, nullptr};     // z_bulk, z_tail, rank_bulk, rank_tail (decoded codes)
    if (want_dbg) for (int i = 0; i < 4; ++i) dbg[i] = cv.take<double>((size_t)M);
    HIP_TRY(ctx, hipMemcpyAsync(X, pooled, sizeof(double) * (size_t)M, hipMemcpyHostToDevice, ctx->stream));
    a.X = X;
    a.d_res = s.d_res;
    rc = run_pipeline(ctx, a);
    if (rc) { abort_inflight(ctx); return rc; }
    s.chunks.push_back(Chunk{0, 1, 0});
    HIP_TRY(ctx, hipMemcpyAsync(s.h_res, s.d_res, sizeof(double) * (size_t)R, hipMemcpyDeviceToHost, ctx->stream));
    auto back = [&](double* dst, const double* srcp) -> hipError_t {
        return dst ? hipMemcpyAsync(dst, srcp, sizeof(double) * (size_t)M, hipMemcpyDeviceToHost, ctx->stream) : hipSuccess;
    };
    if (want_dbg) {
        const unsigned nb = (unsigned)((M + 255) / 256);
        hipLaunchKernelGGL(k_decode_codes, dim3(nb), dim3(256), 0, ctx->stream, (const u32*)a.zb, (const double*)a.ztab, M,
                           z_bulk ? dbg[0] : (double*)nullptr, rank_bulk ? dbg[2] : (double*)nullptr);
        hipLaunchKernelGGL(k_decode_codes, dim3(nb), dim3(256), 0, ctx->stream, (const u32*)a.zt, (const double*)a.ztab, M,
                           z_tail ? dbg[1] : (double*)nullptr, rank_tail ? dbg[3] : (double*)nullptr);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, back(z_bulk, dbg[0]));
    HIP_TRY(ctx, back(z_tail, dbg[1]));
    HIP_TRY(ctx, back(rank_bulk, dbg[2]));
    HIP_TRY(ctx, back(rank_tail, dbg[3]));
    s.busy = true; ctx->order.push_back(0); ctx->n_inflight = 1;
    return wait_impl(ctx);
}

int mcr_moments_dev(mcr_ctx* ctx, const void* draws_dev, int dtype, int64_t C, int64_t N, int64_t P, int64_t sc,
                    int64_t sn, int64_t sp, double* mean, double* std)
{
    if (!ctx) return fail(nullptr, MCR_EINVAL, "ctx is NULL");
    if (dtype != MCR_F64 && dtype != MCR_F32) return fail(ctx, MCR_EINVAL, "unsupported dtype %d", dtype);
    if (C < 0 || N < 0 || P < 0 || !mean || !std) return fail(ctx, MCR_EINVAL, "bad argument");
    const i64 M = C * N;
    if (P == 0) return MCR_OK;
    if (M == 0) { for (i64 p = 0; p < P; ++p) mean[p] = std[p] = NAN; return MCR_OK; }
    if (!draws_dev) return fail(ctx, MCR_EINVAL, "draws is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const bool rows = (N <= 1 || sn == 1) && (C <= 1 || sc == N);
    if (rows && P > kMaxGridY) return fail(ctx, MCR_EINVAL, "P > %d not supported by mcr_moments_dev rows path", kMaxGridY);
    int S;
    if (rows) {
        const i64 vec = dtype == MCR_F64 ? 2 : 4;
        i64 smax = (M + 256 * vec * 4 - 1) / (256 * vec * 4);   // >= one unrolled sweep per block
        i64 want = (4096 + P - 1) / P;
        S = (int)(want < smax ? want : smax);
        if (S < 1) S = 1;
    } else {
        i64 want = (2048 + (P + 63) / 64 - 1) / ((P + 63) / 64);
        i64 smax = (M + 63) / 64;
        S = (int)(want < smax ? want : smax);
        if (S < 1) S = 1;
        if (S > kMaxGridY) S = kMaxGridY;
    }
    const size_t need = align_up((size_t)P * S * kMomRec * 8, 256) + align_up((size_t)P * 8, 256) * 2 + 1024;
    int rc = ensure_ws(ctx, need);
    if (rc) return rc;
    Carve cv{reinterpret_cast<char*>(ctx->ws)};
    double* part = cv.take<double>((size_t)P * S * kMomRec);
    double* d_mean = cv.take<double>((size_t)P);
    double* d_std = cv.take<double>((size_t)P);
    rc = dtype == MCR_F64 ? moments_impl<double>(ctx, (const double*)draws_dev, C, N, P, sc, sn, sp, d_mean, d_std, part, S, rows)
                          : moments_impl<float>(ctx, (const float*)draws_dev, C, N, P, sc, sn, sp, d_mean, d_std, part, S, rows);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(mean, d_mean, sizeof(double) * (size_t)P, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(std, d_std, sizeof(double) * (size_t)P, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    prof_resolve(ctx);
    return MCR_OK;
}

int mcr_basic_stats(mcr_ctx* ctx, const void* values, int dtype, int64_t n, double* mean, double* std)
{
    if (!ctx) return fail(nullptr, MCR_EINVAL, "ctx is NULL");
    if (!mean || !std || n < 0) return fail(ctx, MCR_EINVAL, "bad argument");
    if (dtype != MCR_F64 && dtype != MCR_F32) return fail(ctx, MCR_EINVAL, "unsupported dtype %d", dtype);
    if (n == 0) { *mean = NAN; *std = NAN; return MCR_OK; }   /* compare.py:60-61 */
    if (!values) return fail(ctx, MCR_EINVAL, "values is NULL");
    const size_t es = dtype == MCR_F64 ? 8 : 4;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure_stage(ctx, (size_t)n * es);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->stage, values, (size_t)n * es, hipMemcpyHostToDevice, ctx->stream));
    return mcr_moments_dev(ctx, ctx->stage, dtype, 1, n, 1, n, 1, n, mean, std);
}

int mcr_compare(mcr_ctx* ctx, const double* ref, const double* actual, int64_t n, double tol, double* rel_error,
                uint8_t* passed)
{
    if (!ctx) return fail(nullptr, MCR_EINVAL, "ctx is NULL");
    if (n < 0 || (n > 0 && (!ref || !actual || !rel_error || !passed))) return fail(ctx, MCR_EINVAL, "bad argument");
    if (n == 0) return MCR_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t need = align_up((size_t)n * 8, 256) * 3 + align_up((size_t)n, 256) + 1024;
    int rc = ensure_ws(ctx, need);
    if (rc) return rc;
    Carve cv{reinterpret_cast<char*>(ctx->ws)};
    double* d_ref = cv.take<double>((size_t)n);
    double* d_act = cv.take<double>((size_t)n);
    double* d_rel = cv.take<double>((size_t)n);
    unsigned char* d_ok = cv.take<unsigned char>((size_t)n);
    HIP_TRY(ctx, hipMemcpyAsync(d_ref, ref, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_act, actual, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    LAUNCH(ctx, K_COMPARE, k_compare, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (const double*)d_ref,
           (const double*)d_act, (i64)n, tol, d_rel, d_ok);
    HIP_TRY(ctx, hipMemcpyAsync(rel_error, d_rel, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(passed, d_ok, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    prof_resolve(ctx);
    return MCR_OK;
}

// Sort-only carve of the workspace for P parameters of M draws each (no diagnostics buffers).
static int carve_sort(mcr_ctx* ctx, Carve& cv, i64 M, i64 P, PipeIn& a)
{
    const WsPlan wp = plan_ws(M, 1, false, false, 1);
    a = PipeIn{};
    a.M = M; a.pc = P; a.C = 1; a.ntiles = wp.ntiles; a.do_diag = false;
    a.kA = cv.take<double>((size_t)P * M); a.kB = cv.take<double>((size_t)P * M);
    a.iA = cv.take<u32>((size_t)P * M);    a.iB = cv.take<u32>((size_t)P * M);
    a.part = cv.take<double>((size_t)P * wp.ntiles * kMomRec);
    a.samp = cv.take<double>((size_t)P * (wp.ntiles + 16) * 64);
    a.cut = cv.take<u32>((size_t)P * (wp.bk_B + 1) * (size_t)(wp.bk_k + 1));
    a.boff = cv.take<u32>((size_t)P * (wp.bk_B + 1));
    a.bk_B = wp.bk_B; a.bk_D = wp.bk_D; a.bk_k = wp.bk_k; a.bk_R = wp.bk_R;
    return MCR_OK;
}

int mcr_two_sample(mcr_ctx* ctx, const double* ref, int64_t Mr, const double* act, int64_t Ma, int64_t P,
                   double* ks, double* w1)
{
    if (!ctx) return fail(nullptr, MCR_EINVAL, "ctx is NULL");
    if (P < 0 || Mr < 1 || Ma < 1 || (P > 0 && (!ref || !act || !ks || !w1)))
        return fail(ctx, MCR_EINVAL, "bad argument (both samples need at least one draw)");
    if (P == 0) return MCR_OK;
    if (P > kMaxGridY) return fail(ctx, MCR_EINVAL, "P > %d", kMaxGridY);
    if (Mr >= (i64)0xFFFFFFFFll || Ma >= (i64)0xFFFFFFFFll || (double)Mr * (double)Ma >= 9007199254740992.0)
        return fail(ctx, MCR_EINVAL, "samples too long (Mr * Ma must stay below 2^53)");
    if (ctx->n_inflight) return fail(ctx, MCR_EINVAL, "mcr_two_sample with summaries in flight");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const i64 Mx = Mr > Ma ? Mr : Ma;
    const int nblk = (int)((Mr + Ma + kTile - 1) / kTile);
    const size_t need = (size_t)P * (size_t)Mx * (8 + 4) * 2 + (size_t)P * (size_t)(Mr + Ma) * 8 * 2 +
                        (size_t)P * ((size_t)((Mx + kTile - 1) / kTile) * (32 + 512 + 4 * 80) + (size_t)nblk * 16 + 64) +
                        64 * 256;
    int rc = ensure_ws(ctx, need);
    if (rc) return rc;
    Carve cv{reinterpret_cast<char*>(ctx->ws)};
    double* Xr = cv.take<double>((size_t)P * Mr);
    double* Xa = cv.take<double>((size_t)P * Ma);
    double* Sr = cv.take<double>((size_t)P * Mr);
    double* part = cv.take<double>((size_t)P * nblk * 2);
    double* d_ks = cv.take<double>((size_t)P);
    double* d_w = cv.take<double>((size_t)P);
    double* bad = cv.take<double>((size_t)P * 2);
    const size_t base = cv.off;
    HIP_TRY(ctx, hipMemcpyAsync(Xr, ref, sizeof(double) * (size_t)P * Mr, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(Xa, act, sizeof(double) * (size_t)P * Ma, hipMemcpyHostToDevice, ctx->stream));
    double *kin, *kout; void *iin, *iout; bool ranked;
    PipeIn a;
    {   // ascending order of the reference sample, parked in Sr
        Carve c2{reinterpret_cast<char*>(ctx->ws), base};
        carve_sort(ctx, c2, Mr, P, a);
        a.X = Xr;
        rc = sort_stage(ctx, a, &kin, &iin, &kout, &iout, &ranked);
        if (rc) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(Sr, kin, sizeof(double) * (size_t)P * Mr, hipMemcpyDeviceToDevice, ctx->stream));
        hipLaunchKernelGGL(k_bad_count, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)a.part,
                           (int)a.ntiles, (i64)P, bad);
    }
    {   // ascending order of the actual sample, then one merge-path pass over both
        Carve c2{reinterpret_cast<char*>(ctx->ws), base};
        carve_sort(ctx, c2, Ma, P, a);
        a.X = Xa;
        rc = sort_stage(ctx, a, &kin, &iin, &kout, &iout, &ranked);
        if (rc) return rc;
        hipLaunchKernelGGL(k_bad_count, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)a.part,
                           (int)a.ntiles, (i64)P, bad + P);
    }
    LAUNCH(ctx, K_TWO_SAMPLE, (k_two_sample<256, 16>), dim3((unsigned)nblk, (unsigned)P), dim3(256), 0,
           (const double*)Sr, (i64)Mr, (const double*)kin, (i64)Ma, part, nblk);
    LAUNCH(ctx, K_TWO_SAMPLE, k_two_sample_final, dim3((unsigned)((P + 255) / 256)), dim3(256), 0,
           (const double*)part, nblk, (i64)P, (double)Mr * (double)Ma, d_ks, d_w);
    HIP_TRY(ctx, hipMemcpyAsync(ks, d_ks, sizeof(double) * (size_t)P, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(w1, d_w, sizeof(double) * (size_t)P, hipMemcpyDeviceToHost, ctx->stream));
    std::vector<double> h_bad((size_t)2 * (size_t)P);
    HIP_TRY(ctx, hipMemcpyAsync(h_bad.data(), bad, sizeof(double) * h_bad.size(), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    prof_resolve(ctx);
    for (i64 p = 0; p < P; ++p)
        if (h_bad[(size_t)p] != 0.0 || h_bad[(size_t)(P + p)] != 0.0 || !(ks[p] == ks[p]) || !(w1[p] == w1[p]) || std::isinf(w1[p]))
            return fail(ctx, MCR_ENONFINITE, "draws contain non-finite values");
    return MCR_OK;
}

// Device-resident form (draws_dev [P][M] f64, cov_dev [P][P] f64, both in this context's device memory): what
// tools/cov_bench.py times.  Synchronous.
int mcr_covariance_dev(mcr_ctx* ctx, const double* draws_dev, int64_t M, int64_t P, double* cov_dev)
{
    if (!ctx) return fail(nullptr, MCR_EINVAL, "ctx is NULL");
    if (P < 0 || M < 1 || (P > 0 && (!draws_dev || !cov_dev))) return fail(ctx, MCR_EINVAL, "bad argument");
    if (P == 0) return MCR_OK;
    if (P > 8192) return fail(ctx, MCR_EINVAL, "P > 8192 not supported by mcr_covariance");
    if (ctx->n_inflight) return fail(ctx, MCR_EINVAL, "mcr_covariance with summaries in flight");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int nb = (int)((P + kCovBM - 1) / kCovBM);
    const i64 P64 = (i64)nb * kCovBM;
    const i64 tiles = (i64)nb * (nb + 1) / 2;                        // workgroups per draw slice (none below the diagonal)
    int ksplit = (int)((512 + tiles - 1) / tiles);                   // >= ~512 workgroups: two per CU
    const i64 maxsplit = (M + 16 * kCovBK - 1) / (16 * kCovBK);      // a slice is at least 16 panels long
    if (ksplit > maxsplit) ksplit = (int)maxsplit;
    {   // the slices' partial tiles are summed by k_cov_final: at most 128 MB of them
        const i64 cap = ((i64)128 << 20) / (P64 * P64 * 8);
        if (ksplit > cap) ksplit = (int)(cap < 1 ? 1 : cap);
    }
    if (ksplit < 1) ksplit = 1;
    i64 kchunk = (M + ksplit - 1) / ksplit;
    kchunk = (kchunk + kCovBK - 1) / kCovBK * kCovBK;
    ksplit = (int)((M + kchunk - 1) / kchunk);
    const int S = 8;
    const size_t need = (size_t)ksplit * P64 * P64 * 8 + (size_t)P * (S * kMomRec * 8 + 16) + 16 * 256;
    int rc = ensure_ws(ctx, need);
    if (rc) return rc;
    Carve cv{reinterpret_cast<char*>(ctx->ws)};
    double* partial = cv.take<double>((size_t)ksplit * P64 * P64);
    double* mpart = cv.take<double>((size_t)P * S * kMomRec);
    double* d_mean = cv.take<double>((size_t)P);
    double* d_std = cv.take<double>((size_t)P);
    rc = moments_impl<double>(ctx, draws_dev, 1, M, P, M, 1, M, d_mean, d_std, mpart, (M >= 8 * 2048) ? S : 1, true);
    if (rc) return rc;
    if ((M & 1) == 0 && (reinterpret_cast<uintptr_t>(draws_dev) & 15) == 0) {
        LAUNCH(ctx, K_COV, (k_cov_mfma<true>), dim3((unsigned)tiles, (unsigned)ksplit), dim3(256), 0, draws_dev,
               (const double*)d_mean, (i64)M, (i64)P, nb, kchunk, partial);
    } else {
        LAUNCH(ctx, K_COV, (k_cov_mfma<false>), dim3((unsigned)tiles, (unsigned)ksplit), dim3(256), 0, draws_dev,
               (const double*)d_mean, (i64)M, (i64)P, nb, kchunk, partial);
    }
    LAUNCH(ctx, K_COV_FINAL, k_cov_final, dim3((unsigned)((P * P + 255) / 256)), dim3(256), 0, (const double*)partial,
           ksplit, nb, (i64)M, (i64)P, cov_dev);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    prof_resolve(ctx);
    return MCR_OK;
}

int mcr_covariance(mcr_ctx* ctx, const double* draws, int64_t M, int64_t P, double* cov)
{
    if (!ctx) return fail(nullptr, MCR_EINVAL, "ctx is NULL");
    if (P < 0 || M < 1 || (P > 0 && (!draws || !cov))) return fail(ctx, MCR_EINVAL, "bad argument");
    if (P == 0) return MCR_OK;
    if (ctx->n_inflight) return fail(ctx, MCR_EINVAL, "mcr_covariance with summaries in flight");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure_stage(ctx, (size_t)P * M * 8 + (size_t)P * P * 8 + 256);
    if (rc) return rc;
    double* X = (double*)ctx->stage;
    double* d_cov = (double*)((char*)ctx->stage + align_up((size_t)P * M * 8, 256));
    HIP_TRY(ctx, hipMemcpyAsync(X, draws, sizeof(double) * (size_t)P * M, hipMemcpyHostToDevice, ctx->stream));
    rc = mcr_covariance_dev(ctx, X, M, P, d_cov);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(cov, d_cov, sizeof(double) * (size_t)P * P, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MCR_OK;
}

int mcr_profile_enable(mcr_ctx* ctx, int on)
{
    if (!ctx) return fail(nullptr, MCR_EINVAL, "ctx is NULL");
    ctx->prof = on != 0;
    return MCR_OK;
}
int mcr_profile_reset(mcr_ctx* ctx)
{
    if (!ctx) return fail(nullptr, MCR_EINVAL, "ctx is NULL");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    prof_resolve(ctx);
    for (int k = 0; k < K_COUNT; ++k) { ctx->k_launches[k] = 0; ctx->k_ms[k] = 0.0; }
    return MCR_OK;
}
int mcr_profile_get(mcr_ctx* ctx, mcr_kernel_time* out, int max, int* n)
{
    if (!ctx || !n) return fail(ctx, MCR_EINVAL, "NULL argument");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    prof_resolve(ctx);
    int cnt = 0;
    for (int k = 0; k < K_COUNT; ++k) {
        if (!ctx->k_launches[k]) continue;
        if (out && cnt < max) {
            memset(&out[cnt], 0, sizeof(mcr_kernel_time));
            strncpy(out[cnt].name, kKernelNames[k], sizeof(out[cnt].name) - 1);
            out[cnt].launches = ctx->k_launches[k];
            out[cnt].total_ms = ctx->k_ms[k];
        }
        ++cnt;
    }
    *n = cnt;
    return MCR_OK;
}

int mcr_hbm_probe(mcr_ctx* ctx, size_t bytes, int iters, double* read_gbps, double* copy_gbps)
{
    if (!ctx) return fail(nullptr, MCR_EINVAL, "ctx is NULL");
    if (bytes < ((size_t)1 << 20) || iters < 1 || (!read_gbps && !copy_gbps)) return fail(ctx, MCR_EINVAL, "bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    bytes &= ~(size_t)4095;
    void *a = nullptr, *b = nullptr;
    HIP_TRY(ctx, hipMalloc(&a, bytes));
    if (hipMalloc(&b, copy_gbps ? bytes : 4096 * 4) != hipSuccess) { hipFree(a); return fail(ctx, MCR_ENOMEM, "probe buffer"); }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipStream_t st = ctx->stream;
    int rc = MCR_OK;
    auto best_of = [&](auto&& launch) -> double {
        float best = 1e30f;
        launch();                                                  // warm-up (page faults, clocks)
        for (int k = 0; k < iters; ++k) {
            hipEventRecord(e0, st); launch(); hipEventRecord(e1, st);
            hipEventSynchronize(e1);
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.f && ms < best) best = ms;
        }
        return (double)best;
    };
    if (hipMemsetAsync(a, 0x5A, bytes, st) != hipSuccess) rc = fail(ctx, MCR_EHIP, "memset failed");
    if (!rc && read_gbps) {
        const i64 nvec = (i64)(bytes / 16);
        double ms = 1e30;
        for (unsigned grid : {2048u, 4096u, 8192u, 16384u, 32768u}) {
            const double t = best_of([&] { hipLaunchKernelGGL(k_stream_read, dim3(grid), dim3(256), 0, st, (const uint4*)a, nvec, (u32*)b); });
            if (t < ms) ms = t;
        }
        *read_gbps = (double)bytes / (ms * 1e-3) / 1e9;
    }
    if (!rc && copy_gbps) {
        const double ms = best_of([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, st); });
        *copy_gbps = 2.0 * (double)bytes / (ms * 1e-3) / 1e9;     // bytes read + bytes written
    }
    hipStreamSynchronize(st);
    if (hipGetLastError() != hipSuccess && !rc) rc = fail(ctx, MCR_EHIP, "probe launch failed");
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(a); hipFree(b);
    return rc;
}

int mcr_fill_synthetic_at(mcr_ctx* ctx, void* draws_dev, int dtype, int64_t C, int64_t N, int64_t P, int64_t p0, uint64_t seed)
{
    if (!ctx) return fail(nullptr, MCR_EINVAL, "ctx is NULL");
    if (dtype != MCR_F64 && dtype != MCR_F32) return fail(ctx, MCR_EINVAL, "unsupported dtype %d", dtype);
    if (C < 0 || N < 0 || P < 0 || p0 < 0) return fail(ctx, MCR_EINVAL, "negative dimension");
    const i64 total = C * N * P;
    if (total <= 0) return MCR_OK;
    if (!draws_dev) return fail(ctx, MCR_EINVAL, "draws is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    i64 blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    const i64 first = p0 * C * N;       // element index of the block's first draw in the whole tensor
    if (dtype == MCR_F64) {
        LAUNCH(ctx, K_FILL, (k_fill_synth<double>), dim3((unsigned)blocks), dim3(256), 0, (double*)draws_dev, total,
               (i64)(C * N), (u64)seed, first);
    } else {
        LAUNCH(ctx, K_FILL, (k_fill_synth<float>), dim3((unsigned)blocks), dim3(256), 0, (float*)draws_dev, total,
               (i64)(C * N), (u64)seed, first);
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    prof_resolve(ctx);
    return MCR_OK;
}

int mcr_fill_synthetic(mcr_ctx* ctx, void* draws_dev, int dtype, int64_t C, int64_t N, int64_t P, uint64_t seed)
{
    return mcr_fill_synthetic_at(ctx, draws_dev, dtype, C, N, P, 0, seed);
}


// ---- multi-GPU: the RCCL communicator -------------------------------------------------------------

struct mcr_comm {
    mcr_ctx* ctx = nullptr;
    ncclComm_t nccl = nullptr;
    hipStream_t stream = nullptr;
    int world = 1, rank = 0;
    void* dbuf = nullptr; size_t dbytes = 0;      // device staging (send block + receive blocks)
    bool deadline = true;                         // every wait is bounded by timeout_s (MCR_COMM_BLOCKING=1 switches that off)
    bool nonblocking = false;                     // MCR_COMM_NONBLOCKING=1: communicator created with config.blocking = 0
    bool dead = false;                            // aborted after a deadline or an asynchronous error
    double timeout_s = 300.0;                     // MCR_COMM_TIMEOUT_S
};

namespace {
int comm_fail(mcr_ctx* ctx, const char* what, ncclResult_t r)
{
    mcr::comm::Api* a = mcr::comm::api();
    return fail(ctx, MCR_ECOMM, "%s failed: %s", what, (a && a->GetErrorString) ? a->GetErrorString(r) : "RCCL error");
}
int comm_buf(mcr_comm* c, size_t bytes)
{
    if (bytes <= c->dbytes) return MCR_OK;
    if (c->dbuf) { hipFree(c->dbuf); c->dbuf = nullptr; c->dbytes = 0; }
    HIP_TRY(c->ctx, hipMalloc(&c->dbuf, bytes));
    c->dbytes = bytes;
    return MCR_OK;
}

// A DEADLINE ON EVERY COLLECTIVE.  The ranks are separate processes; one that dies after the
// rendezvous (the reference's generate loop carries on past a failed recipe, src/mcmc_ref/generate.py:77-96 -- a sharded
// one must at least not hang on it) would leave its peers inside ncclCommInitRank / ncclAllGather for ever.  So:
//   * ncclCommInitRank runs on a helper thread and the caller only watches the clock (a rank whose peers never come does
//     not return from it, and on RCCL 2.27.7 neither does an abort of it);
//   * the collectives are enqueued on the communicator's stream as usual (the RCCL call returns once its kernel is in the
//     stream) and the wait for that stream is a poll of hipStreamQuery + ncclCommGetAsyncError against
//     MCR_COMM_TIMEOUT_S (default 300 s; the first init of a node can take tens of seconds) instead of a
//     hipStreamSynchronize;
//   * on expiry, or on an asynchronous RCCL error, the communicator is aborted (ncclCommAbort, itself bounded: it tears
//     down the stuck kernel and the proxy), marked dead, and the call returns MCR_ECOMM naming the call, the rank and the
//     time; every later call on it fails at once.
// The communicator itself is the ordinary BLOCKING one (what every RCCL application runs; its first world > 1 run here
// is the driver's); MCR_COMM_NONBLOCKING=1 creates it with ncclCommInitRankConfig(blocking = 0) instead, where the RCCL
// calls themselves return ncclInProgress and are polled too.  MCR_COMM_BLOCKING=1: no deadlines at all (round 3).
using comm_clock = std::chrono::steady_clock;

// ncclCommAbort with a bound of its own: measured on RCCL 2.27.7, aborting a communicator whose ncclCommInitRank is still
// waiting for a peer does not return (it joins the bootstrap it is meant to cancel).  The abort therefore runs on a helper
// thread that this call waits for at most kAbortGraceS; a helper that has not come back by then is left behind (detached:
// it owns nothing but the communicator handle, which nobody uses again) -- the caller gets its error either way.
constexpr double kAbortGraceS = 5.0;
void bounded_abort(ncclComm_t nccl)
{
    mcr::comm::Api* a = mcr::comm::api();
    if (!a || !a->CommAbort || !nccl) return;
    auto done = std::make_shared<std::atomic<int>>(0);
    std::thread([a, nccl, done]() { a->CommAbort(nccl); done->store(1, std::memory_order_release); }).detach();
    const comm_clock::time_point t0 = comm_clock::now();
    while (!done->load(std::memory_order_acquire) && std::chrono::duration<double>(comm_clock::now() - t0).count() < kAbortGraceS)
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
}

int comm_abort(mcr_comm* c, const char* what, const char* why)
{
    bounded_abort(c->nccl);
    c->nccl = nullptr;
    c->dead = true;
    return fail(c->ctx, MCR_ECOMM, "%s: rank %d of %d gave up: %s (MCR_COMM_TIMEOUT_S = %g s); the communicator was aborted",
                what, c->rank, c->world, why, c->timeout_s);
}

// Waits until the communicator's pending RCCL call has been issued (non-blocking mode: state leaves ncclInProgress) and,
// with `stream`, until the work on its stream has finished.
int comm_wait(mcr_comm* c, const char* what, bool stream)
{
    mcr::comm::Api* a = mcr::comm::api();
    const comm_clock::time_point t0 = comm_clock::now();
    auto expired = [&]() { return std::chrono::duration<double>(comm_clock::now() - t0).count() > c->timeout_s; };
    auto nap = [&](int& spins) { if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50)); };
    int spins = 0;
    if (c->nonblocking) {
        for (;;) {
            ncclResult_t st = ncclSuccess;
            const ncclResult_t r = a->CommGetAsyncError(c->nccl, &st);
            if (r != ncclSuccess) { comm_abort(c, what, "ncclCommGetAsyncError failed"); return MCR_ECOMM; }
            if (st == ncclSuccess) break;
            if (st != ncclInProgress) {
                char why[160];
                snprintf(why, sizeof why, "asynchronous RCCL error: %s", a->GetErrorString ? a->GetErrorString(st) : "?");
                return comm_abort(c, what, why);
            }
            if (expired()) return comm_abort(c, what, "a peer did not arrive before the deadline");
            nap(spins);
        }
    }
    if (!stream) return MCR_OK;
    if (!c->deadline) { HIP_TRY(c->ctx, hipStreamSynchronize(c->stream)); return MCR_OK; }
    for (;;) {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) return MCR_OK;
        if (q != hipErrorNotReady) return fail(c->ctx, MCR_EHIP, "%s: hipStreamQuery failed: %s", what, hipGetErrorString(q));
        ncclResult_t st = ncclSuccess;
        if (a->CommGetAsyncError && a->CommGetAsyncError(c->nccl, &st) == ncclSuccess && st != ncclSuccess && st != ncclInProgress) {
            char why[160];
            snprintf(why, sizeof why, "asynchronous RCCL error: %s", a->GetErrorString ? a->GetErrorString(st) : "?");
            return comm_abort(c, what, why);
        }
        if (expired()) return comm_abort(c, what, "the collective did not complete before the deadline (a peer is gone or stuck)");
        nap(spins);
    }
}
}  // namespace

int mcr_comm_unique_id(void* id, size_t len)
{
    if (!id || len < MCR_COMM_ID_BYTES) return fail(nullptr, MCR_EINVAL, "id buffer must hold %d bytes", MCR_COMM_ID_BYTES);
    mcr::comm::Api* a = mcr::comm::api();
    if (!a) return fail(nullptr, MCR_ECOMM, "librccl could not be loaded: %s", mcr::comm::api_why());
    static_assert(sizeof(ncclUniqueId) == MCR_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId uid;
    const ncclResult_t r = a->GetUniqueId(&uid);
    if (r != ncclSuccess) return comm_fail(nullptr, "ncclGetUniqueId", r);
    memcpy(id, &uid, sizeof uid);
    return MCR_OK;
}

int mcr_comm_init(mcr_ctx* ctx, const void* id, int world, int rank, mcr_comm** out)
{
    if (!ctx) return fail(nullptr, MCR_EINVAL, "ctx is NULL");
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return fail(ctx, MCR_EINVAL, "bad communicator arguments (world %d, rank %d)", world, rank);
    *out = nullptr;
    mcr::comm::Api* a = mcr::comm::api();
    if (!a) return fail(ctx, MCR_ECOMM, "librccl could not be loaded: %s", mcr::comm::api_why());
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    mcr_comm* c = new (std::nothrow) mcr_comm();
    if (!c) return fail(ctx, MCR_ENOMEM, "out of host memory");
    c->ctx = ctx; c->world = world; c->rank = rank;
    if (const char* env = getenv("MCR_COMM_TIMEOUT_S")) { const double v = atof(env); if (v > 0.0) c->timeout_s = v; }
    const char* blk = getenv("MCR_COMM_BLOCKING");
    const char* nbl = getenv("MCR_COMM_NONBLOCKING");
    c->deadline = !(blk && atoi(blk) != 0);
    c->nonblocking = c->deadline && a->nonblocking() && nbl && atoi(nbl) != 0;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(ctx, MCR_EHIP, "hipStreamCreate failed: %s", hipGetErrorString(e)); }
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    // collective: every rank of the world calls it
    if (c->deadline) {
        // The whole init runs on a helper thread and THIS thread only watches the clock: on RCCL 2.27.7 neither the init of a
        // rank whose peers never come nor an abort of it is guaranteed to return, and the caller must get its error anyway.
        struct InitJob {
            std::atomic<int> state{0}, cancel{0};      // state 1: finished (r / async say how)
            std::atomic<ncclComm_t> nccl{nullptr};
            ncclResult_t r = ncclSuccess, async = ncclSuccess;
        };
        auto job = std::make_shared<InitJob>();
        const int dev = ctx->device;
        const bool nonblocking = c->nonblocking;
        std::thread([a, job, dev, world, uid, rank, nonblocking]() {
            hipSetDevice(dev);
            ncclComm_t h = nullptr;
            if (nonblocking) {
                ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
                cfg.blocking = 0;
                job->r = a->CommInitRankConfig(&h, world, uid, rank, &cfg);
            } else {
                job->r = a->CommInitRank(&h, world, uid, rank);          // returns when every rank has arrived -- or never
            }
            job->nccl.store(h, std::memory_order_release);
            if (nonblocking && (job->r == ncclSuccess || job->r == ncclInProgress) && h) {
                ncclResult_t st = ncclInProgress;
                int spins = 0;
                while (!job->cancel.load(std::memory_order_acquire)) {
                    if (a->CommGetAsyncError(h, &st) != ncclSuccess) { st = ncclInternalError; break; }
                    if (st != ncclInProgress) break;
                    if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
                }
                job->async = st;
            }
            job->state.store(1, std::memory_order_release);
        }).detach();
        const comm_clock::time_point t0 = comm_clock::now();
        int spins = 0;
        while (!job->state.load(std::memory_order_acquire) &&
               std::chrono::duration<double>(comm_clock::now() - t0).count() <= c->timeout_s)
            if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(100));
        int rc = MCR_OK;
        if (!job->state.load(std::memory_order_acquire)) {
            job->cancel.store(1, std::memory_order_release);
            bounded_abort(job->nccl.load(std::memory_order_acquire));
            rc = fail(ctx, MCR_ECOMM, "ncclCommInitRank: rank %d of %d gave up: a peer did not arrive before the deadline "
                      "(MCR_COMM_TIMEOUT_S = %g s); the communicator was aborted", rank, world, c->timeout_s);
        } else if (job->r != ncclSuccess && job->r != ncclInProgress) rc = comm_fail(ctx, "ncclCommInitRank", job->r);
        else if (!job->nccl.load()) rc = fail(ctx, MCR_ECOMM, "ncclCommInitRank returned no communicator");
        else if (job->async != ncclSuccess) { bounded_abort(job->nccl.load()); rc = comm_fail(ctx, "ncclCommInitRank (asynchronous)", job->async); }
        if (rc) { hipStreamDestroy(c->stream); delete c; return rc; }
        c->nccl = job->nccl.load();
    } else {
        const ncclResult_t r = a->CommInitRank(&c->nccl, world, uid, rank);
        if (r != ncclSuccess) { hipStreamDestroy(c->stream); delete c; return comm_fail(ctx, "ncclCommInitRank", r); }
    }
    *out = c;
    return MCR_OK;
}

void mcr_comm_free(mcr_comm* c)
{
    if (!c) return;
    hipSetDevice(c->ctx->device);
    mcr::comm::Api* a = mcr::comm::api();
    if (c->nccl && !c->dead) {
        if (c->deadline) {
            if (comm_wait(c, "mcr_comm_free", true) == MCR_OK && c->nccl) {
                a->CommDestroy(c->nccl);            // (returns ncclInProgress on a non-blocking communicator: finalisation goes on inside RCCL)
            }
        } else {
            if (c->stream) hipStreamSynchronize(c->stream);
            if (a) a->CommDestroy(c->nccl);
        }
    }
    if (c->dbuf) hipFree(c->dbuf);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int mcr_comm_world(const mcr_comm* c) { return c ? c->world : -1; }
int mcr_comm_rank(const mcr_comm* c) { return c ? c->rank : -1; }
int mcr_comm_has_deadline(const mcr_comm* c) { return (c && c->deadline) ? 1 : 0; }

// THE collective of the path: every rank contributes `count` doubles, every rank receives world * count doubles in
// rank order (ncclAllGather over xGMI).  Host pointers; staged through a small device buffer on the communicator's stream.
int mcr_comm_all_gather(mcr_comm* c, const double* send, int64_t count, double* recv)
{
    if (!c) return fail(nullptr, MCR_EINVAL, "comm is NULL");
    mcr_ctx* ctx = c->ctx;
    if (c->dead) return fail(ctx, MCR_ECOMM, "ncclAllGather: the communicator of rank %d was aborted by an earlier failure", c->rank);
    if (count < 0 || (count > 0 && (!send || !recv))) return fail(ctx, MCR_EINVAL, "bad argument");
    if (count == 0) return MCR_OK;
    mcr::comm::Api* a = mcr::comm::api();
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t sb = (size_t)count * 8, rb = sb * (size_t)c->world;
    int rc = comm_buf(c, align_up(sb, 256) + rb);
    if (rc) return rc;
    char* d_send = (char*)c->dbuf;
    char* d_recv = d_send + align_up(sb, 256);
    HIP_TRY(ctx, hipMemcpyAsync(d_send, send, sb, hipMemcpyHostToDevice, c->stream));
    const ncclResult_t r = a->AllGather(d_send, d_recv, (size_t)count, ncclDouble, c->nccl, c->stream);
    if (r != ncclSuccess && r != ncclInProgress) return comm_fail(ctx, "ncclAllGather", r);
    rc = comm_wait(c, "ncclAllGather", false);                 // issued (non-blocking mode)
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(recv, d_recv, rb, hipMemcpyDeviceToHost, c->stream));
    return comm_wait(c, "ncclAllGather", true);
}

// Element-wise reduction of n host doubles over the ranks, in place (op: 0 sum, 1 max, 2 min): the bench's
// max-over-ranks clock and "every rank validated" flag, and (n = 1) a barrier.
int mcr_comm_all_reduce(mcr_comm* c, double* vals, int64_t n, int op)
{
    if (!c) return fail(nullptr, MCR_EINVAL, "comm is NULL");
    mcr_ctx* ctx = c->ctx;
    if (c->dead) return fail(ctx, MCR_ECOMM, "ncclAllReduce: the communicator of rank %d was aborted by an earlier failure", c->rank);
    if (n < 0 || (n > 0 && !vals) || op < 0 || op > 2) return fail(ctx, MCR_EINVAL, "bad argument");
    if (n == 0) return MCR_OK;
    mcr::comm::Api* a = mcr::comm::api();
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t b = (size_t)n * 8;
    int rc = comm_buf(c, b);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(c->dbuf, vals, b, hipMemcpyHostToDevice, c->stream));
    const ncclRedOp_t ops[3] = {ncclSum, ncclMax, ncclMin};
    const ncclResult_t r = a->AllReduce(c->dbuf, c->dbuf, (size_t)n, ncclDouble, ops[op], c->nccl, c->stream);
    if (r != ncclSuccess && r != ncclInProgress) return comm_fail(ctx, "ncclAllReduce", r);
    rc = comm_wait(c, "ncclAllReduce", false);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(vals, c->dbuf, b, hipMemcpyDeviceToHost, c->stream));
    return comm_wait(c, "ncclAllReduce", true);
}

int mcr_comm_barrier(mcr_comm* c)
{
    if (!c) return fail(nullptr, MCR_EINVAL, "comm is NULL");
    for (hipStream_t st : c->ctx->lane_stream) if (st) HIP_TRY(c->ctx, hipStreamSynchronize(st));
    double one = 1.0;
    return mcr_comm_all_reduce(c, &one, 1, 0);
}


// ---- Parquet ingest (replaces pq.read_table at store.py:79-95 / convert.py:61-65) -----------

struct mcr_parquet { mcr::pq::File f; };

namespace {
int ensure_buf(mcr_ctx* ctx, void** p, size_t* cap, size_t bytes)
{
    if (bytes <= *cap) return MCR_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (*p) { hipFree(*p); *p = nullptr; *cap = 0; }
    const size_t want = bytes + (bytes >> 2);
    const hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess) return fail(ctx, MCR_ENOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    *cap = want;
    return MCR_OK;
}
}  // namespace

int mcr_parquet_open(mcr_ctx* ctx, const void* bytes, size_t len, mcr_parquet** out)
{
    if (!bytes || !out) return fail(ctx, MCR_EINVAL, "NULL argument");     // ctx may be NULL: parsing needs no device
    mcr_parquet* f = new (std::nothrow) mcr_parquet();
    if (!f) return fail(ctx, MCR_ENOMEM, "out of host memory");
    bool ok = false;
    try { ok = mcr::pq::open(f->f, bytes, len); }
    catch (const std::exception& e) { f->f.error = std::string("host allocation failed: ") + e.what(); }
    if (!ok) {
        const int rc = fail(ctx, MCR_EINVAL, "parquet: %s", f->f.error.c_str());
        delete f;
        return rc;
    }
    *out = f;
    return MCR_OK;
}

void mcr_parquet_close(mcr_parquet* f) { delete f; }
int64_t mcr_parquet_num_rows(const mcr_parquet* f) { return f ? f->f.num_rows : -1; }
int mcr_parquet_num_columns(const mcr_parquet* f) { return f ? (int)f->f.cols.size() : -1; }
const char* mcr_parquet_column_name(const mcr_parquet* f, int col)
{
    return (f && col >= 0 && col < (int)f->f.cols.size()) ? f->f.cols[col].name.c_str() : nullptr;
}
int mcr_parquet_column_type(const mcr_parquet* f, int col)
{
    return (f && col >= 0 && col < (int)f->f.cols.size()) ? f->f.cols[col].type : -1;
}

int mcr_parquet_num_pages(const mcr_parquet* f) { return f ? (int)f->f.pages.size() : -1; }
int mcr_parquet_page_info(const mcr_parquet* f, int page, int64_t* info)
{
    if (!f || !info || page < 0 || page >= (int)f->f.pages.size()) return MCR_EINVAL;
    const mcr::pq::Page& p = f->f.pages[page];
    info[0] = p.col; info[1] = p.kind; info[2] = p.encoding; info[3] = p.codec; info[4] = (int64_t)p.payload_off;
    info[5] = p.comp_size; info[6] = p.uncomp_size; info[7] = p.num_values; info[8] = (int64_t)p.row_off; info[9] = p.dict;
    return MCR_OK;
}

namespace {
// A batched decode in three steps: pq_plan (host only: byte spans to stage, page table), pq_buffers (device buffers +
// table uploads), pq_launch (the two kernels).  Between buffers and launch the caller stages the spans' file bytes at
// pq_stage + span.stage_off: mcr_parquet_decode copies them from the caller's file image (pageable memory),
// mcr_summarize_files reads them into pinned memory on host threads and uploads behind them.
struct PqSpan { const mcr::pq::File* f; u64 start, end; size_t stage_off; };
struct PqPlan {
    std::vector<PqSpan> merged;
    size_t stage_total = 0, scratch_total = 0;
    std::vector<mcr::pq::PageDev> tab;
    std::vector<int> l_snappy, l_decode;
    mcr::pq::PageDev* d_tab = nullptr; int* d_ls = nullptr; int* d_ld = nullptr; int* d_err = nullptr;
};

// whole_files: (file, offset of its whole image in the staging buffer) sorted by file pointer, and the total staged
// bytes -- the caller stages whole file images instead of the packed column chunks (mcr_summarize_files, which wants
// nearly every byte of every file anyway and starts uploading before the footers are parsed).
using FileBase = std::pair<const mcr::pq::File*, size_t>;
int pq_plan(mcr_ctx* ctx, const mcr_parquet_request* reqs, int n_reqs, PqPlan& P,
            const std::vector<FileBase>* whole_files = nullptr, size_t whole_total = 0)
{
    namespace pq = mcr::pq;
    std::vector<PqSpan> spans;
    std::vector<pq::PageDev>& tab = P.tab;
    // 1. byte spans to upload: the column chunks of the requested columns, merged when (nearly) adjacent
    for (int r = 0; r < n_reqs; ++r) {
        const mcr_parquet_request& q = reqs[r];
        if (!q.file) return fail(ctx, MCR_EINVAL, "request %d: file is NULL", r);
        const pq::File& f = q.file->f;
        if (q.column < 0 || q.column >= (int)f.cols.size()) return fail(ctx, MCR_EINVAL, "request %d: column %d out of range", r, q.column);
        const pq::Column& col = f.cols[q.column];
        if (col.type != pq::T_INT32 && col.type != pq::T_INT64 && col.type != pq::T_FLOAT && col.type != pq::T_DOUBLE)
            return fail(ctx, MCR_EINVAL, "parquet: column '%s' has physical type %d; only INT32, INT64, FLOAT and DOUBLE columns are decoded", col.name.c_str(), col.type);
        if (q.out_kind != MCR_PQ_F64 && q.out_kind != MCR_PQ_I64) return fail(ctx, MCR_EINVAL, "request %d: bad out_kind %d", r, q.out_kind);
        if (q.out_kind == MCR_PQ_I64 && col.type != pq::T_INT32 && col.type != pq::T_INT64)
            return fail(ctx, MCR_EINVAL, "parquet: column '%s' is not an integer column", col.name.c_str());
        if (f.num_rows > 0 && !q.out_dev) return fail(ctx, MCR_EINVAL, "request %d: out_dev is NULL", r);
        for (const pq::Chunk& ch : f.chunks)
            if (ch.col == q.column && ch.end > ch.start) spans.push_back(PqSpan{&f, ch.start, ch.end, 0});
    }
    std::sort(spans.begin(), spans.end(), [](const PqSpan& a, const PqSpan& b) {
        return a.f != b.f ? std::less<const pq::File*>()(a.f, b.f) : a.start < b.start; });
    std::vector<PqSpan>& merged = P.merged;
    for (const PqSpan& s : spans) {
        if (!merged.empty() && merged.back().f == s.f && s.start <= merged.back().end + 4096) {
            if (s.end > merged.back().end) merged.back().end = s.end;
        } else merged.push_back(s);
    }
    size_t stage_total = 0;
    if (whole_files) {
        for (PqSpan& s : merged) {
            auto it = std::lower_bound(whole_files->begin(), whole_files->end(), s.f, [](const FileBase& a, const mcr::pq::File* key) {
                return std::less<const mcr::pq::File*>()(a.first, key); });
            if (it == whole_files->end() || it->first != s.f) return fail(ctx, MCR_EINVAL, "parquet: request on a file that is not staged");
            s.stage_off = it->second + (size_t)s.start;
        }
        stage_total = whole_total;
    } else {
        for (PqSpan& s : merged) { s.stage_off = align_up(stage_total, 256); stage_total = s.stage_off + (size_t)(s.end - s.start); }
    }
    P.stage_total = stage_total;
    // staged position of the file bytes [off, off + n): the WHOLE payload must lie inside one uploaded span
    // (spans are sorted by (file, start): binary search for the file's first span, then a short walk)
    auto stage_of = [&](const pq::File* f, u64 off, u64 n) -> size_t {
        auto it = std::lower_bound(merged.begin(), merged.end(), f, [](const PqSpan& s, const pq::File* key) {
            return std::less<const pq::File*>()(s.f, key); });
        for (; it != merged.end() && it->f == f; ++it)
            if (off >= it->start && off + n <= it->end) return it->stage_off + (size_t)(off - it->start);
        return (size_t)-1;
    };
    // 2. page table
    size_t scratch_total = 0;
    for (int r = 0; r < n_reqs; ++r) {
        const mcr_parquet_request& q = reqs[r];
        const pq::File& f = q.file->f;
        const pq::Column& col = f.cols[q.column];
        const u32 es = (col.type == pq::T_INT64 || col.type == pq::T_DOUBLE) ? 8 : 4;
        for (const pq::Chunk& ch : f.chunks) {
            if (ch.col != q.column) continue;
            int dict_idx = -1;
            for (int k = 0; k < ch.n_pages; ++k) {
                const pq::Page& pg = f.pages[(size_t)ch.first_page + k];
                if (pg.codec != pq::CODEC_NONE && pg.codec != pq::CODEC_SNAPPY)
                    return fail(ctx, MCR_EINVAL, "parquet: column '%s' uses compression codec %d; only UNCOMPRESSED and SNAPPY are decoded", col.name.c_str(), pg.codec);
                pq::PageDev d; memset(&d, 0, sizeof(d));
                const bool v2 = pg.kind == pq::PAGE_DATA_V2;
                const u32 lvl = v2 ? pg.rep_bytes + pg.def_bytes : 0;
                const bool comp = pg.codec == pq::CODEC_SNAPPY && (!v2 || pg.v2_compressed);
                if (pg.uncomp_size < lvl) return fail(ctx, MCR_EINVAL, "parquet: v2 page smaller than its levels");
                d.src_off = stage_of(&f, pg.payload_off, pg.comp_size);
                if (pg.comp_size > 0 && d.src_off == (u64)(size_t)-1) return fail(ctx, MCR_EINVAL, "parquet: page outside its column chunk");
                d.comp_size = pg.comp_size - lvl; d.uncomp_size = pg.uncomp_size - lvl;
                d.num_values = pg.num_values; d.lvl_bytes = lvl; d.def_bytes = v2 ? pg.def_bytes : 0;
                d.kind = (unsigned char)pg.kind; d.compressed = comp ? 1 : 0; d.phys_type = (unsigned char)col.type;
                d.max_def = (unsigned char)((v2 && pg.def_bytes == 0) ? 0 : col.max_def);   // v2 without level bytes: all defined
                d.out_kind = (unsigned char)q.out_kind;
                d.dict_page = -1;
                if (v2 && pg.rep_bytes) return fail(ctx, MCR_EINVAL, "parquet: repetition levels in a flat column");
                if (!comp && d.comp_size != d.uncomp_size) return fail(ctx, MCR_EINVAL, "parquet: uncompressed page with differing sizes");
                if (comp) { d.dst_off = align_up(scratch_total, 16); scratch_total = (size_t)d.dst_off + d.uncomp_size + 16; }
                if (pg.kind == pq::PAGE_DICT) {
                    if (pg.encoding != pq::ENC_PLAIN && pg.encoding != pq::ENC_PLAIN_DICT)
                        return fail(ctx, MCR_EINVAL, "parquet: dictionary page encoding %d is not supported", pg.encoding);
                    if ((u64)pg.num_values * es > d.uncomp_size) return fail(ctx, MCR_EINVAL, "parquet: dictionary page shorter than its entries");
                    d.encoding = pq::ENC_PLAIN;
                    dict_idx = (int)tab.size();
                } else {
                    if (pg.encoding == pq::ENC_PLAIN) d.encoding = pq::ENC_PLAIN;
                    else if (pg.encoding == pq::ENC_RLE_DICT || pg.encoding == pq::ENC_PLAIN_DICT) {
                        if (pg.dict < 0 || dict_idx < 0) return fail(ctx, MCR_EINVAL, "parquet: dictionary-encoded page without a dictionary page");
                        d.encoding = pq::ENC_RLE_DICT; d.dict_page = dict_idx; d.dict_count = pg.dict_count;
                    } else
                        return fail(ctx, MCR_EINVAL, "parquet: column '%s' uses value encoding %d; only PLAIN and RLE_DICTIONARY are decoded", col.name.c_str(), pg.encoding);
                    d.out = q.out_dev; d.out_off = pg.row_off;
                    P.l_decode.push_back((int)tab.size());
                }
                if (comp) P.l_snappy.push_back((int)tab.size());
                tab.push_back(d);
            }
        }
    }
    P.scratch_total = scratch_total;
    return MCR_OK;
}

// 3. device buffers + the page table and the two page lists, on ctx->stream
int pq_buffers(mcr_ctx* ctx, PqPlan& P)
{
    namespace pq = mcr::pq;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure_buf(ctx, &ctx->pq_stage, &ctx->pq_stage_bytes, P.stage_total + pq::kInWin + 256);
    if (rc) return rc;
    rc = ensure_buf(ctx, &ctx->pq_scratch, &ctx->pq_scratch_bytes, P.scratch_total + 256);
    if (rc) return rc;
    const size_t tab_bytes = align_up(P.tab.size() * sizeof(pq::PageDev), 256);
    const size_t ls_bytes = align_up(P.l_snappy.size() * 4 + 4, 256), ld_bytes = align_up(P.l_decode.size() * 4 + 4, 256);
    rc = ensure_buf(ctx, &ctx->pq_tab, &ctx->pq_tab_bytes, tab_bytes + ls_bytes + ld_bytes + 256);
    if (rc) return rc;
    char* tb = (char*)ctx->pq_tab;
    P.d_tab = (pq::PageDev*)tb;
    P.d_ls = (int*)(tb + tab_bytes); P.d_ld = (int*)(tb + tab_bytes + ls_bytes);
    P.d_err = (int*)(tb + tab_bytes + ls_bytes + ld_bytes);
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipMemsetAsync((char*)ctx->pq_stage + P.stage_total, 0, pq::kInWin + 256, st));
    if (!P.tab.empty()) HIP_TRY(ctx, hipMemcpyAsync(P.d_tab, P.tab.data(), P.tab.size() * sizeof(pq::PageDev), hipMemcpyHostToDevice, st));
    if (!P.l_snappy.empty()) HIP_TRY(ctx, hipMemcpyAsync(P.d_ls, P.l_snappy.data(), P.l_snappy.size() * 4, hipMemcpyHostToDevice, st));
    if (!P.l_decode.empty()) HIP_TRY(ctx, hipMemcpyAsync(P.d_ld, P.l_decode.data(), P.l_decode.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemsetAsync(P.d_err, 0, 8, st));
    return MCR_OK;
}

int pq_launch(mcr_ctx* ctx, const PqPlan& P)
{
    namespace pq = mcr::pq;
    if (!P.l_snappy.empty())
        LAUNCH(ctx, K_PQ_SNAPPY, pq::k_pq_snappy, dim3((unsigned)P.l_snappy.size()), dim3(64), 0, (const unsigned char*)ctx->pq_stage,
               (unsigned char*)ctx->pq_scratch, (const pq::PageDev*)P.d_tab, (const int*)P.d_ls, P.d_err);
    if (!P.l_decode.empty())
        LAUNCH(ctx, K_PQ_DECODE, pq::k_pq_decode, dim3((unsigned)P.l_decode.size()), dim3(256), 0, (const unsigned char*)ctx->pq_stage,
               (const unsigned char*)ctx->pq_scratch, (const pq::PageDev*)P.d_tab, (const int*)P.d_ld, P.d_err);
    return MCR_OK;
}

int pq_error(mcr_ctx* ctx, const int* h_err)
{
    if (!h_err[0]) return MCR_OK;
    static const char* const what[] = {"", "corrupt Snappy stream", "null values are not supported", "corrupt definition levels",
                                       "corrupt RLE / bit-packed runs", "dictionary index out of range", "page shorter than its values"};
    const int c = h_err[0];
    return fail(ctx, MCR_EINVAL, "parquet: %s (page %d of the request)", (c > 0 && c < 7) ? what[c] : "decode error", h_err[1]);
}
}  // namespace

int mcr_parquet_decode(mcr_ctx* ctx, const mcr_parquet_request* reqs, int n_reqs)
{
    if (!ctx) return fail(nullptr, MCR_EINVAL, "ctx is NULL");
    if (n_reqs < 0 || (n_reqs > 0 && !reqs)) return fail(ctx, MCR_EINVAL, "bad request list");
    if (n_reqs == 0) return MCR_OK;
    try {
        PqPlan P;
        int rc = pq_plan(ctx, reqs, n_reqs, P);
        if (rc) return rc;
        rc = pq_buffers(ctx, P);
        if (rc) return rc;
        hipStream_t st = ctx->stream;
        for (const PqSpan& s : P.merged)
            HIP_TRY(ctx, hipMemcpyAsync((char*)ctx->pq_stage + s.stage_off, s.f->bytes + s.start, (size_t)(s.end - s.start), hipMemcpyHostToDevice, st));
        rc = pq_launch(ctx, P);
        if (rc) return rc;
        int h_err[2] = {0, 0};
        HIP_TRY(ctx, hipMemcpyAsync(h_err, P.d_err, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        prof_resolve(ctx);
        rc = pq_error(ctx, h_err);
        if (rc) return rc;
    } catch (const std::exception& e) {
        return fail(ctx, MCR_ENOMEM, "parquet: host allocation failed: %s", e.what());
    }
    return MCR_OK;
}

int mcr_gather_rows_dev(mcr_ctx* ctx, const double* src_dev, int64_t P, int64_t M, const int64_t* order, double* dst_dev)
{
    if (!ctx) return fail(nullptr, MCR_EINVAL, "ctx is NULL");
    if (P < 0 || M < 0 || P > kMaxGridY) return fail(ctx, MCR_EINVAL, "bad shape");
    if (P == 0 || M == 0) return MCR_OK;
    if (!src_dev || !dst_dev || !order) return fail(ctx, MCR_EINVAL, "NULL argument");
    if (src_dev == dst_dev) return fail(ctx, MCR_EINVAL, "gather cannot run in place");
    for (i64 k = 0; k < M; ++k) if (order[k] < 0 || order[k] >= M) return fail(ctx, MCR_EINVAL, "order[%lld] out of range", (long long)k);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure_buf(ctx, &ctx->pq_tab, &ctx->pq_tab_bytes, (size_t)M * 8);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->pq_tab, order, (size_t)M * 8, hipMemcpyHostToDevice, ctx->stream));
    LAUNCH(ctx, K_GATHER, mcr::pq::k_gather_rows, dim3((unsigned)((M + 255) / 256), (unsigned)P), dim3(256), 0, src_dev,
           (const i64*)ctx->pq_tab, (i64)M, dst_dev);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    prof_resolve(ctx);
    return MCR_OK;
}


// ---- many files in one call -------------------------------------------------------------------------------

struct mcr_fileset {
    double phase_ms[MCR_FS_PHASES] = {0};
    int n_q = 0, n_jobs = 0;
    struct Entry {
        std::vector<std::string> names;
        i64 C = 0, N = 0;
        std::vector<double> f[MCR_FS_FIELDS];   // MCR_FS_* fields
    };
    std::vector<Entry> files;
};

namespace {
struct OpenFile {          // a draws file of mcr_summarize_files: descriptor, size, parsed metadata (its image lives in the pinned buffer)
    int fd = -1; size_t len = 0; mcr_parquet* pq = nullptr;
    ~OpenFile() {
        if (pq) mcr_parquet_close(pq);
        if (fd >= 0) close(fd);
    }
};
}  // namespace

int mcr_summarize_files(mcr_ctx* ctx, const char* const* paths, int n_paths, int min_chains, const double* quantiles,
                        int n_q, int diagnostics, mcr_fileset** out)
{
    namespace pq = mcr::pq;
    if (!ctx) return fail(nullptr, MCR_EINVAL, "ctx is NULL");
    if (!out || n_paths < 0 || (n_paths > 0 && !paths)) return fail(ctx, MCR_EINVAL, "bad argument");
    if (min_chains < 1) return fail(ctx, MCR_EMINCHAINS_ARG, "min_chains must be >= 1; got %d", min_chains);
    if (n_q < 0 || n_q > MCR_MAX_QUANTILES || (n_q > 0 && !quantiles)) return fail(ctx, MCR_EINVAL, "bad quantile list");
    if (ctx->n_inflight) return fail(ctx, MCR_EINVAL, "mcr_summarize_files with summaries in flight");
    using clk = std::chrono::steady_clock;
    const clk::time_point t_start = clk::now();
    double phase[MCR_FS_PHASES] = {0};
    clk::time_point t_prev = t_start;
    auto lap = [&](int k) { const clk::time_point now = clk::now(); phase[k] += std::chrono::duration<double, std::milli>(now - t_prev).count(); t_prev = now; };
    try {
        std::vector<OpenFile> mf((size_t)n_paths);
        struct Plan { std::vector<int> cols; int chain = -1, draw = -1; i64 M = 0; size_t off = 0, ioff = 0; i64 C = 0, N = 0; };
        std::vector<Plan> plan((size_t)n_paths);
        size_t arena = 0, ids = 0;
        // 1. The files' images go WHOLE into one pinned host buffer (nearly every byte of a draws file is a column chunk this
        //    call decodes): MCR_IO_THREADS host threads pread() them -- the page cache's copy lands in memory the DMA engine
        //    reads directly; no mapping is set up or torn down (57 munmaps cost 2 ms of TLB shoot-downs in a process with
        //    this many threads), no page faults, no pageable staging inside the runtime -- and parse each footer and its
        //    page headers from that image, while THIS thread uploads the finished prefix behind them in pieces of >= 2 MB.
        //    HIP calls stay on the calling thread.
        std::vector<size_t> img_off((size_t)n_paths + 1, 0);
        for (int i = 0; i < n_paths; ++i) {
            OpenFile& m = mf[(size_t)i];
            m.fd = open(paths[i], O_RDONLY);
            struct stat st;
            if (m.fd < 0 || fstat(m.fd, &st) != 0) return fail(ctx, MCR_EINVAL, "cannot open %s", paths[i]);
            m.len = (size_t)st.st_size;
            if (m.len == 0) return fail(ctx, MCR_EINVAL, "parquet: %s is empty", paths[i]);
            img_off[(size_t)i + 1] = align_up(img_off[(size_t)i] + m.len, 256);
        }
        const size_t img_total = img_off[(size_t)n_paths];
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        int rc = ensure_buf(ctx, &ctx->pq_stage, &ctx->pq_stage_bytes, img_total + pq::kInWin + 256);
        if (rc) return rc;
        if (img_total > ctx->pq_pin_bytes) {                    // pinned staging, kept for the life of the context
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->pq_pin) { hipHostFree(ctx->pq_pin); ctx->pq_pin = nullptr; ctx->pq_pin_bytes = 0; }
            const size_t want = img_total + (img_total >> 2);
            HIP_TRY(ctx, hipHostMalloc(&ctx->pq_pin, want, hipHostMallocDefault));
            ctx->pq_pin_bytes = want;
        }
        lap(MCR_FS_PH_OPEN);
        std::vector<int> frc((size_t)n_paths, MCR_OK);
        std::vector<std::string> ferr((size_t)n_paths);
        if (n_paths > 0) {
            char* pin = (char*)ctx->pq_pin;
            std::vector<std::atomic<int>> done((size_t)n_paths);
            for (auto& d : done) d.store(0, std::memory_order_relaxed);
            std::atomic<int> next{0};
            auto reader = [&]() {
                for (int i; (i = next.fetch_add(1)) < n_paths;) {
                    OpenFile& m = mf[(size_t)i];
                    auto bad = [&](int code, const std::string& msg) { frc[(size_t)i] = code; ferr[(size_t)i] = msg; };
                    try {
                        size_t got = 0;
                        while (got < m.len) {
                            const ssize_t r = pread(m.fd, pin + img_off[(size_t)i] + got, m.len - got, (off_t)got);
                            if (r <= 0) break;
                            got += (size_t)r;
                        }
                        if (got < m.len) bad(MCR_EINVAL, std::string("short read of ") + paths[i]);
                        else {
                            m.pq = new mcr_parquet();
                            if (!pq::open(m.pq->f, pin + img_off[(size_t)i], m.len)) bad(MCR_EINVAL, std::string(paths[i]) + ": parquet: " + m.pq->f.error);
                        }
                    } catch (const std::exception& e) { bad(MCR_ENOMEM, std::string("host allocation failed: ") + e.what()); }
                    done[(size_t)i].store(1, std::memory_order_release);
                }
            };
            std::vector<std::thread> th;
            const int TR = std::max(1, std::min(ctx->io_threads, n_paths));
            for (int t = 0; t < TR; ++t) th.emplace_back(reader);
            int sent = 0;                                        // files [0, sent) are uploaded
            hipError_t he = hipSuccess;
            for (int i = 0; i < n_paths; ++i) {
                while (done[(size_t)i].load(std::memory_order_acquire) == 0) std::this_thread::yield();
                const size_t from = img_off[(size_t)sent], upto = img_off[(size_t)i + 1];
                if (he == hipSuccess && (upto - from >= ctx->io_piece || i + 1 == n_paths)) {
                    he = hipMemcpyAsync((char*)ctx->pq_stage + from, pin + from, upto - from, hipMemcpyHostToDevice, ctx->stream);
                    sent = i + 1;
                }
            }
            for (std::thread& x : th) x.join();
            for (int i = 0; i < n_paths; ++i)
                if (frc[(size_t)i]) { hipStreamSynchronize(ctx->stream); return fail(ctx, frc[(size_t)i], "%s", ferr[(size_t)i].c_str()); }
            if (he != hipSuccess) return fail(ctx, MCR_EHIP, "hipMemcpyAsync of the file images failed: %s", hipGetErrorString(he));
        }
        lap(MCR_FS_PH_READ);
        for (int i = 0; i < n_paths; ++i) {
            const pq::File& f = mf[(size_t)i].pq->f;
            Plan& pl = plan[(size_t)i];
            for (int c = 0; c < (int)f.cols.size(); ++c) {
                const int ty = f.cols[(size_t)c].type;
                const bool numeric = ty == pq::T_INT32 || ty == pq::T_INT64 || ty == pq::T_FLOAT || ty == pq::T_DOUBLE;
                if (f.cols[(size_t)c].name == "chain") pl.chain = c;
                else if (f.cols[(size_t)c].name == "draw") pl.draw = c;
                else if (numeric) pl.cols.push_back(c);
            }
            if (pl.chain < 0 || pl.draw < 0) { hipStreamSynchronize(ctx->stream); return fail(ctx, MCR_EINVAL, "%s: no chain / draw columns", paths[i]); }
            pl.M = f.num_rows;
            pl.ioff = ids; ids += (size_t)2 * (size_t)pl.M * 8;
        }
        // arena order = path order.  Measured in round 4 on the packaged corpus -- BEFORE k_tier3 got its extra slots for short
        // chains, i.e. with 17 listed pairs per slot on the critical path of the statistics phase (1.18 - 1.20 ms then, 0.68 ms
        // now; five tensors as the files come): files of one hinted shape laid next to each other (two tensors): 1.42 - 1.48 ms;
        // tensors capped at 160 / 100 / 60 / 30 parameters (6 / 8 / 11 / 20 tensors): 1.29 / 1.33 / 1.40 / 1.90 ms; every tensor
        // forked over two streams: 1.31 - 1.34 ms.
        std::vector<int> order((size_t)n_paths);
        for (int i = 0; i < n_paths; ++i) {
            order[(size_t)i] = i;
            Plan& pl = plan[(size_t)i];
            pl.off = arena; arena += pl.cols.size() * (size_t)pl.M * 8;
        }
        // 2. one batched decode into the arena ([P][M] per file, packed) + chain / draw ids behind it
        const size_t ids_base = align_up(arena, 256);
        const size_t lay_base = align_up(ids_base + ids, 256), lay_out = align_up(lay_base + (size_t)n_paths * sizeof(pq::FileIds), 256);
        rc = ensure_buf(ctx, &ctx->fs_arena, &ctx->fs_arena_bytes, lay_out + (size_t)n_paths * 32 + 256);
        if (rc) return rc;
        char* base = (char*)ctx->fs_arena;
        std::vector<mcr_parquet_request> reqs;
        for (int i = 0; i < n_paths; ++i) {
            const Plan& pl = plan[(size_t)i];
            for (size_t j = 0; j < pl.cols.size(); ++j)
                reqs.push_back(mcr_parquet_request{mf[(size_t)i].pq, pl.cols[j], MCR_PQ_F64, base + pl.off + j * (size_t)pl.M * 8});
            reqs.push_back(mcr_parquet_request{mf[(size_t)i].pq, pl.chain, MCR_PQ_I64, base + ids_base + pl.ioff});
            reqs.push_back(mcr_parquet_request{mf[(size_t)i].pq, pl.draw, MCR_PQ_I64, base + ids_base + pl.ioff + (size_t)pl.M * 8});
        }
        PqPlan PP;
        if (!reqs.empty()) {
            std::vector<FileBase> bases;
            for (int i = 0; i < n_paths; ++i) bases.emplace_back(&mf[(size_t)i].pq->f, img_off[(size_t)i]);
            std::sort(bases.begin(), bases.end(), [](const FileBase& a, const FileBase& b) { return std::less<const pq::File*>()(a.first, b.first); });
            rc = pq_plan(ctx, reqs.data(), (int)reqs.size(), PP, &bases, img_total);
            if (rc) { hipStreamSynchronize(ctx->stream); return rc; }
            rc = pq_buffers(ctx, PP);            // (pq_stage is large enough already: no reallocation under the uploads in flight)
            if (rc) return rc;
        }
        lap(MCR_FS_PH_PLAN);
        // 2c. decode kernels + the chain / draw bookkeeping on the device (k_chain_layout: 32 bytes per file come back
        //     instead of the id columns)
        std::vector<i64> h_lay((size_t)n_paths * 4, 0);
        int h_err[2] = {0, 0};
        if (!reqs.empty()) {
            rc = pq_launch(ctx, PP);
            if (rc) return rc;
            std::vector<pq::FileIds> fids((size_t)n_paths);
            for (int i = 0; i < n_paths; ++i) {
                const Plan& pl = plan[(size_t)i];
                fids[(size_t)i] = pq::FileIds{(const i64*)(base + ids_base + pl.ioff), (const i64*)(base + ids_base + pl.ioff + (size_t)pl.M * 8), pl.M};
            }
            HIP_TRY(ctx, hipMemcpyAsync(base + lay_base, fids.data(), fids.size() * sizeof(pq::FileIds), hipMemcpyHostToDevice, ctx->stream));
            LAUNCH(ctx, K_GATHER, pq::k_chain_layout, dim3((unsigned)n_paths), dim3(256), 0, (const pq::FileIds*)(base + lay_base), (i64*)(base + lay_out));
            HIP_TRY(ctx, hipMemcpyAsync(h_lay.data(), base + lay_out, (size_t)n_paths * 32, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(h_err, PP.d_err, 8, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            prof_resolve(ctx);
            rc = pq_error(ctx, h_err);
            if (rc) return rc;
        }
        // 3. chain / draw bookkeeping (convert._chains_from_table): rows must already be in (chain, draw) order
        for (int i = 0; i < n_paths; ++i) {
            Plan& pl = plan[(size_t)i];
            const i64* L = h_lay.data() + (size_t)i * 4;
            if (pl.M > 0 && !L[3]) return fail(ctx, MCR_ELAYOUT, "%s: rows are not in (chain, draw) order", paths[i]);
            const i64 C = pl.M > 0 ? L[0] : 0;
            const bool equal = pl.M == 0 || L[2] != 0;
            pl.C = C; pl.N = pl.M > 0 ? L[1] : 0;
            if (diagnostics && !pl.cols.empty()) {
                if (C < min_chains) return fail(ctx, MCR_EMINCHAINS, "%s: R-hat diagnostics require at least %d chains; got %lld chain(s)", paths[i], min_chains, (long long)C);
                if (!equal) return fail(ctx, MCR_ELAYOUT, "%s: chains of unequal length", paths[i]);
            }
            if (!diagnostics && !pl.cols.empty() && pl.M == 0) return fail(ctx, MCR_EINVAL, "%s: cannot compute stats of empty columns", paths[i]);
        }
        lap(MCR_FS_PH_DECODE);
        // 4. result set + jobs (runs of neighbouring files of one shape are one tensor)
        std::unique_ptr<mcr_fileset> fs(new mcr_fileset());
        fs->n_q = n_q;
        fs->files.resize((size_t)n_paths);
        for (int i = 0; i < n_paths; ++i) {
            mcr_fileset::Entry& e = fs->files[(size_t)i];
            const Plan& pl = plan[(size_t)i];
            const size_t P = pl.cols.size();
            for (int c : pl.cols) e.names.push_back(mf[(size_t)i].pq->f.cols[(size_t)c].name);
            e.C = pl.C; e.N = pl.N;
            for (int k = 0; k < MCR_FS_FIELDS; ++k) e.f[k].assign(k == MCR_FS_Q ? P * (size_t)n_q : P, NAN);
        }
        struct Job { int first, count; i64 C, N, P; };
        std::vector<Job> jobs;
        for (int oi = 0; oi < n_paths; ++oi) {             // (first / count index `order`, the arena's sequence of files)
            const Plan& pl = plan[(size_t)order[(size_t)oi]];
            if (pl.cols.empty()) continue;
            const i64 Cj = diagnostics ? pl.C : 1, Nj = diagnostics ? pl.N : pl.M;
            if (!jobs.empty()) {
                Job& j = jobs.back();
                const Plan& last = plan[(size_t)order[(size_t)(j.first + j.count - 1)]];
                if (j.first + j.count == oi && j.C == Cj && j.N == Nj && last.off + last.cols.size() * (size_t)last.M * 8 == pl.off) {
                    ++j.count; j.P += (i64)pl.cols.size();
                    continue;
                }
            }
            jobs.push_back(Job{oi, 1, Cj, Nj, (i64)pl.cols.size()});
        }
        fs->n_jobs = (int)jobs.size();
        // per-job staging of the results (a job spans files; scattered back below)
        constexpr int NF = MCR_FS_FIELDS;
        std::vector<std::vector<double>> jf(jobs.size() * NF);
        std::vector<std::vector<int64_t>> jl(jobs.size() * 2);
        std::vector<int64_t> qlo((size_t)(n_q > 0 ? n_q : 1));
        int err = MCR_OK;
        char keep[512] = "";
        for (size_t k = 0; k < jobs.size() && !err; ++k) {
            const Job& j = jobs[k];
            for (int q = 0; q < NF; ++q) jf[k * NF + q].assign(q == MCR_FS_Q ? (size_t)j.P * (size_t)n_q : (size_t)j.P, NAN);
            jl[k * 2].assign((size_t)j.P, 0); jl[k * 2 + 1].assign((size_t)j.P, 0);
            mcr_summary o{};
            o.mean = jf[k * NF + MCR_FS_MEAN].data(); o.std = jf[k * NF + MCR_FS_STD].data();
            o.q = n_q > 0 ? jf[k * NF + MCR_FS_Q].data() : nullptr; o.median = jf[k * NF + MCR_FS_MEDIAN].data();
            o.q_lo = qlo.data();
            if (diagnostics) {
                o.rhat = jf[k * NF + MCR_FS_RHAT].data(); o.ess_bulk = jf[k * NF + MCR_FS_ESS_BULK].data();
                o.ess_tail = jf[k * NF + MCR_FS_ESS_TAIL].data();
                o.rhat_bulk = jf[k * NF + MCR_FS_RHAT_BULK].data(); o.rhat_tail = jf[k * NF + MCR_FS_RHAT_TAIL].data();
                o.lag_bulk = jl[k * 2].data(); o.lag_tail = jl[k * 2 + 1].data();
            }
            if (ctx->n_inflight >= MCR_MAX_INFLIGHT) err = wait_one_impl(ctx);
            if (!err)
                err = enqueue_impl(ctx, base + plan[(size_t)order[(size_t)j.first]].off, MCR_F64, j.C, j.N, j.P, j.N, 1, j.C * j.N,
                                   diagnostics ? min_chains : 1, quantiles, n_q, &o);
            if (err) memcpy(keep, ctx->err, sizeof keep);
        }
        const int rw = wait_impl(ctx);
        if (err) { memcpy(ctx->err, keep, sizeof keep); return err; }
        if (rw) return rw;
        lap(MCR_FS_PH_STATS);
        if (diagnostics)
            for (size_t k = 0; k < jobs.size(); ++k)
                for (size_t p = 0; p < (size_t)jobs[k].P; ++p) {
                    jf[k * NF + MCR_FS_LAG_BULK][p] = (double)jl[k * 2][p];
                    jf[k * NF + MCR_FS_LAG_TAIL][p] = (double)jl[k * 2 + 1][p];
                }
        for (size_t k = 0; k < jobs.size(); ++k) {
            size_t p0 = 0;
            for (int oi = jobs[k].first; oi < jobs[k].first + jobs[k].count; ++oi) {
                mcr_fileset::Entry& e = fs->files[(size_t)order[(size_t)oi]];
                const size_t P = e.names.size();
                for (int q = 0; q < NF; ++q) {
                    const size_t w = q == MCR_FS_Q ? (size_t)n_q : 1;
                    if (w) memcpy(e.f[q].data(), jf[k * NF + q].data() + p0 * w, P * w * sizeof(double));
                }
                p0 += P;
            }
        }
        lap(MCR_FS_PH_COLLECT);
        mf.clear();                                   // unmap, close, free the parsed metadata
        lap(MCR_FS_PH_CLOSE);
        phase[MCR_FS_PH_TOTAL] = std::chrono::duration<double, std::milli>(clk::now() - t_start).count();
        memcpy(fs->phase_ms, phase, sizeof phase);
        *out = fs.release();
        return MCR_OK;
    } catch (const std::exception& e) {
        return fail(ctx, MCR_ENOMEM, "mcr_summarize_files: host allocation failed: %s", e.what());
    }
}

int mcr_fileset_size(const mcr_fileset* fs) { return fs ? (int)fs->files.size() : -1; }
static const mcr_fileset::Entry* fs_entry(const mcr_fileset* fs, int file)
{
    return (fs && file >= 0 && file < (int)fs->files.size()) ? &fs->files[(size_t)file] : nullptr;
}
int64_t mcr_fileset_params(const mcr_fileset* fs, int file) { const auto* e = fs_entry(fs, file); return e ? (int64_t)e->names.size() : -1; }
int64_t mcr_fileset_chains(const mcr_fileset* fs, int file) { const auto* e = fs_entry(fs, file); return e ? e->C : -1; }
int64_t mcr_fileset_draws(const mcr_fileset* fs, int file) { const auto* e = fs_entry(fs, file); return e ? e->N : -1; }
const char* mcr_fileset_param_name(const mcr_fileset* fs, int file, int64_t param)
{
    const auto* e = fs_entry(fs, file);
    return (e && param >= 0 && param < (int64_t)e->names.size()) ? e->names[(size_t)param].c_str() : nullptr;
}
const double* mcr_fileset_field(const mcr_fileset* fs, int file, int field)
{
    const auto* e = fs_entry(fs, file);
    return (e && field >= 0 && field < MCR_FS_FIELDS) ? e->f[field].data() : nullptr;
}
int64_t mcr_fileset_export(const mcr_fileset* fs, double* rows, int64_t cap_rows)
{
    if (!fs) return -1;
    static const int order[10] = {MCR_FS_MEAN, MCR_FS_STD, MCR_FS_MEDIAN, MCR_FS_RHAT, MCR_FS_ESS_BULK, MCR_FS_ESS_TAIL,
                                  MCR_FS_RHAT_BULK, MCR_FS_RHAT_TAIL, MCR_FS_LAG_BULK, MCR_FS_LAG_TAIL};
    const size_t w = 10 + (size_t)fs->n_q;
    int64_t row = 0;
    for (const mcr_fileset::Entry& e : fs->files)
        for (size_t p = 0; p < e.names.size(); ++p, ++row) {
            if (!rows || row >= cap_rows) continue;
            double* r = rows + (size_t)row * w;
            for (int k = 0; k < 10; ++k) r[k] = e.f[order[k]][p];
            for (int q = 0; q < fs->n_q; ++q) r[10 + q] = e.f[MCR_FS_Q][p * (size_t)fs->n_q + (size_t)q];
        }
    return row;
}

int64_t mcr_fileset_names(const mcr_fileset* fs, char* buf, int64_t cap)
{
    if (!fs) return -1;
    int64_t need = 0;
    for (const mcr_fileset::Entry& e : fs->files)
        for (const std::string& n : e.names) {
            const int64_t len = (int64_t)n.size() + 1;
            if (buf && need + len <= cap) memcpy(buf + need, n.c_str(), (size_t)len);
            need += len;
        }
    return need;
}

int mcr_fileset_jobs(const mcr_fileset* fs) { return fs ? fs->n_jobs : -1; }
int mcr_fileset_phases(const mcr_fileset* fs, double* ms, int cap)
{
    if (!fs || !ms || cap < 0) return -1;
    const int n = cap < MCR_FS_PHASES ? cap : MCR_FS_PHASES;
    for (int k = 0; k < n; ++k) ms[k] = fs->phase_ms[k];
    return MCR_FS_PHASES;
}
void mcr_fileset_free(mcr_fileset* fs) { delete fs; }

}  // extern "C"
