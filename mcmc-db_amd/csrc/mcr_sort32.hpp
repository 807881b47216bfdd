// mcr_sort32.hpp -- the sort stage for f32 draw tensors (BASELINE config 4: 4 x 100000 x 10000 f32).
//
// Same algorithm as the f64 kernels of mcr_kernels.hpp (tile sort -> pairwise passes -> exact bucket partition ->
// in-LDS bucket merge fused with tie-averaged ranks), same results bit for bit (widening f32 -> f64 preserves order
// and equality, so ranks, order statistics and the widened values the moments / fold step see are identical), but the
// unit that moves is ONE 64-bit record per draw
//     record = order-preserving image of the f32 bits (f32_key) << 32 | pooled position
// instead of an f64 key plus a separate position: 8 bytes per LDS slot and per HBM element instead of 12, one integer
// compare per merge step that orders (value, position) at once, no payload gather after a merge level, and no widening
// ingest pass over the tensor.  +inf pads are kRecPad.  Reference spec of the whole stage: the `sorted(flat, key=...)`
// and average ranks of src/mcmc_ref/diagnostics.py:101-122.
#pragma once
#include "mcr_kernels.hpp"

namespace mcr {

__device__ __forceinline__ u32 rec_key(u64 r) { return (u32)(r >> 32); }
__device__ __forceinline__ u32 rec_pos(u64 r) { return (u32)r; }

// Register sort of VT records per lane (the networks of mcr_sortnet.h on 64-bit integers).
template <int VT>
__device__ __forceinline__ void thread_sort_rec(u64 (&r)[VT])
{
    static_assert(VT == 16 || VT == 8, "sorting networks exist for 16 and 8 items per lane");
#define MCR_CE(a, b)                                         \
    {                                                        \
        const bool sw = r[b] < r[a];                         \
        const u64 lo = sw ? r[b] : r[a], hi = sw ? r[a] : r[b]; \
        r[a] = lo; r[b] = hi;                                \
    }
    if constexpr (VT == 16) { MCR_NET16(MCR_CE) } else { MCR_NET8(MCR_CE) }
#undef MCR_CE
}

// lane_merge_levels of mcr_kernels.hpp on records: the first three merge levels of the tile sort (16 -> 128 records) as
// bitonic merges across 2, 4 and 8 lanes over DPP; records are distinct (the position is in the low word), so an
// exchange is one unsigned compare per side.
template <int CTRL>
__device__ __forceinline__ u64 dpp_u64(u64 v)
{
    return ((u64)dpp_u32<CTRL>((u32)(v >> 32)) << 32) | (u64)dpp_u32<CTRL>((u32)v);
}
__device__ __forceinline__ u64 select_by_mask(u64 a, u64 b, unsigned long long m)
{
    return ((u64)select_by_mask((u32)(a >> 32), (u32)(b >> 32), m) << 32) | (u64)select_by_mask((u32)a, (u32)b, m);
}

template <int CTRL, bool MIRROR, unsigned long long MIN_LANES>
__device__ __forceinline__ void lane_pair_stage_rec(u64 (&r)[16])
{
    auto exchange = [](u64& own, u64 o) {
        const unsigned long long lt = __builtin_amdgcn_ballot_w64(o < own), gt = __builtin_amdgcn_ballot_w64(own < o);
        own = select_by_mask(own, o, (lt & MIN_LANES) | (gt & ~MIN_LANES));
    };
#pragma unroll
    for (int i = 0; i < (MIRROR ? 8 : 16); ++i) {
        const int s = MIRROR ? 15 - i : i;
        const u64 o_i = dpp_u64<CTRL>(r[s]);
        if (MIRROR) {
            const u64 o_s = dpp_u64<CTRL>(r[i]);
            exchange(r[s], o_s);
        }
        exchange(r[i], o_i);
    }
}

__device__ __forceinline__ void lane_bitonic_merge16_rec(u64 (&r)[16])
{
#define MCR_CE(a, b)                                         \
    {                                                        \
        const bool sw = r[b] < r[a];                         \
        const u64 lo = sw ? r[b] : r[a], hi = sw ? r[a] : r[b]; \
        r[a] = lo; r[b] = hi;                                \
    }
    MCR_BITONIC16(MCR_CE)
#undef MCR_CE
}

__device__ __forceinline__ void lane_merge_levels_rec(u64 (&r)[16])
{
    constexpr unsigned long long kEvenLanes = 0x5555555555555555ull, kLowPairs = 0x3333333333333333ull,
                                 kLowQuads = 0x0F0F0F0F0F0F0F0Full;
    lane_pair_stage_rec<0xB1, true, kEvenLanes>(r);
    lane_bitonic_merge16_rec(r);
    lane_pair_stage_rec<0x1B, true, kLowPairs>(r);
    lane_pair_stage_rec<0xB1, false, kEvenLanes>(r);
    lane_bitonic_merge16_rec(r);
    lane_pair_stage_rec<0x141, true, kLowQuads>(r);
    lane_pair_stage_rec<0x4E, false, kLowPairs>(r);
    lane_pair_stage_rec<0xB1, false, kEvenLanes>(r);
    lane_bitonic_merge16_rec(r);
}

// serial_merge of mcr_kernels.hpp on records: the record IS the payload, so there is nothing to gather afterwards.
// LIM > 0: slot indices are clamped to LIM (k_tile_sort32 declares exactly T slots so that five tiles fit a CU, and
// the unconditional read one past the last run must not leave them).
template <int VT, int LIM = 0>
__device__ __forceinline__ void serial_merge_rec(const u64* srec, int a0, int na, int b0, int nb, int ai, int bi,
                                                 int nout, u64 (&out)[VT])
{
    int pa = a0 + ai, pb = b0 + bi;
    const int ea = a0 + na, eb = b0 + nb;
    auto slot = [](int e) { return pos16(LIM > 0 ? min(e, LIM) : e); };
    u64 ar = srec[slot(pa)], br = srec[slot(pb)];
#pragma unroll
    for (int i = 0; i < VT; ++i) {
        if (i < nout) {
            const bool takeA = (pb >= eb) | ((pa < ea) & !(br < ar));
            out[i] = takeA ? ar : br;
            pa += takeA ? 1 : 0;
            pb += takeA ? 0 : 1;
            const int pn = takeA ? pa : pb;
            const u64 nv = srec[slot(pn)];
            ar = takeA ? nv : ar;
            br = takeA ? br : nv;
        } else {
            out[i] = kRecPad;
        }
    }
}

// One more slot than records: the unconditional read "one past the last run" of serial_merge_rec stays in bounds.
constexpr size_t rec_lds_bytes(int T) { return (size_t)(T + 2) * 8; }

// ------------------------------------------------------------------------------------------------
// Tile sort: 4096 pooled f32 draws of one parameter -> sorted records, the tile's moments (two-pass on the widened
// values, as k_tile_sort) and every 64th order statistic.  The tensor is read from HBM exactly once, as f32.
// ------------------------------------------------------------------------------------------------
template <int NT, int VT>
__global__ __launch_bounds__(NT) void k_tile_sort32(const float* __restrict__ X, i64 M, u64* __restrict__ recs,
                                                    double* __restrict__ part, int ntiles, double* __restrict__ samp)
{
    constexpr int T = NT * VT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* srec = reinterpret_cast<u64*>(smem);
    double* red = reinterpret_cast<double*>(smem);      // only after the sorted tile has left LDS

    const int tid = threadIdx.x, tile = blockIdx.x;
    const i64 p = blockIdx.y;
    const i64 base = (i64)tile * T;
    const int count = (int)((M - base < (i64)T) ? M - base : (i64)T);
    const float* src = X + p * M + base;

    double bad = 0.0;
#pragma unroll
    for (int i = 0; i < VT; ++i) {
        const int e = i * NT + tid;
        u64 r = kRecPad;
        if (e < count) {
            const float v = src[e];
            bad += isfinite(v) ? 0.0 : 1.0;
            r = ((u64)f32_key(v) << 32) | (u64)(u32)(base + e);
        }
        srec[pos16(e)] = r;
    }
    __syncthreads();
    u64 r[VT];
#pragma unroll
    for (int i = 0; i < VT; ++i) r[i] = srec[pos16(tid * VT + i)];
    thread_sort_rec<VT>(r);
    constexpr bool kLaneLevels = MCR_TILE_DPP_LEVELS >= 3 && VT == 16;
    if constexpr (kLaneLevels) lane_merge_levels_rec(r);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VT; ++i) srec[pos16(tid * VT + i)] = r[i];
    __syncthreads();

    for (int coop = kLaneLevels ? 16 : 2; coop <= NT; coop <<= 1) {
        const int first = tid & ~(coop - 1);
        const int run = VT * (coop >> 1);
        const int a0 = first * VT, b0 = a0 + run;
        const int diag = VT * (tid - first);
        auto A = [&](int i) { return srec[pos16(a0 + i)]; };
        auto B = [&](int j) { return srec[pos16(b0 + j)]; };
        const int ai = merge_path32(A, run, B, run, diag);
        serial_merge_rec<VT, T - 1>(srec, a0, run, b0, run, ai, diag - ai, VT, r);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < VT; ++i) srec[pos16(tid * VT + i)] = r[i];
        __syncthreads();
    }
    for (int e = tid; e < count; e += NT) recs[p * M + base + e] = srec[pos16(e)];
    if (samp != nullptr && tid < T / 64) {
        const int e = 64 * tid + 63;
        samp[(p * ntiles + tile) * (T / 64) + tid] = (e < count) ? key_value(rec_key(srec[pos16(e)])) : INFINITY;
    }
    __syncthreads();                 // `red` aliases srec from here on
    double s1 = 0.0;
#pragma unroll
    for (int i = 0; i < VT; ++i) s1 += (tid * VT + i < count) ? key_value(rec_key(r[i])) : 0.0;
    s1 = block_sum<NT>(s1, red);
    const double mt = s1 / (double)count;
    double s2 = 0.0, e1 = 0.0;
#pragma unroll
    for (int i = 0; i < VT; ++i) { const double d = (tid * VT + i < count) ? key_value(rec_key(r[i])) - mt : 0.0; s2 = fma(d, d, s2); e1 += d; }
    block_sum3<NT>(s2, e1, bad, red);
    if (tid == 0) store_slice_moments(part + (p * ntiles + tile) * kMomRec, mt, e1, s2, bad, (double)count);
}

// ------------------------------------------------------------------------------------------------
// One pairwise merge-path pass over sorted runs of R records (pooled arrays longer than 16 tiles).
// ------------------------------------------------------------------------------------------------
template <int NT, int VT>
__global__ __launch_bounds__(NT) void k_merge32(const u64* __restrict__ rin, u64* __restrict__ rout, i64 M, i64 R)
{
    constexpr int OB = NT * VT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* srec = reinterpret_cast<u64*>(smem);
    __shared__ i64 sh[2];
    const int tid = threadIdx.x;
    const i64 p = blockIdx.y;
    const i64 o0 = (i64)blockIdx.x * OB;
    if (o0 >= M) return;
    const u64* rp = rin + p * M;
    const i64 pb = (o0 / (2 * R)) * (2 * R);
    const i64 abase = pb, na = (M - pb < R) ? M - pb : R;
    const i64 bbase = pb + na, nb = (M - bbase < R) ? M - bbase : R;
    const i64 d0 = o0 - pb, tot = na + nb;
    const i64 d1 = (d0 + OB < tot) ? d0 + OB : tot;
    auto GA = [&](i64 i) { return rp[abase + i]; };
    auto GB = [&](i64 j) { return rp[bbase + j]; };
    if (tid < 64) { const i64 r0 = merge_path_wave(GA, na, GB, nb, d0); if (tid == 0) sh[0] = r0; }
    else if (tid < 128) { const i64 r1 = merge_path_wave(GA, na, GB, nb, d1); if (tid == 64) sh[1] = r1; }
    __syncthreads();
    const i64 ai0 = sh[0], ai1 = sh[1], bi0 = d0 - ai0;
    const int ca = (int)(ai1 - ai0), cb = (int)((d1 - ai1) - bi0);
    const int total = ca + cb;
    {   // all VT loads of a lane in flight before the first LDS store (slots beyond `total` read record 0, not stored)
        u64 gr[VT];
#pragma unroll
        for (int j = 0; j < VT; ++j) {
            const int e = j * NT + tid;
            const i64 g = (e < ca) ? abase + ai0 + e : bbase + bi0 + (e - ca);
            gr[j] = rp[(e < total) ? g : 0];
        }
#pragma unroll
        for (int j = 0; j < VT; ++j) {
            const int e = j * NT + tid;
            if (e < total) srec[pos16(e)] = gr[j];
        }
    }
    __syncthreads();
    const int diag = (tid * VT < total) ? tid * VT : total;
    const int nout = (total - diag < VT) ? total - diag : VT;
    auto A = [&](int i) { return srec[pos16(i)]; };
    auto B = [&](int j) { return srec[pos16(ca + j)]; };
    const int ai = merge_path32(A, ca, B, cb, diag);
    u64 r[VT];
    serial_merge_rec<VT>(srec, 0, ca, ca, cb, ai, diag - ai, nout, r);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VT; ++i)
        if (i < nout) srec[pos16(diag + i)] = r[i];
    __syncthreads();
    for (int e = tid; e < total; e += NT) rout[p * M + o0 + e] = srec[pos16(e)];
}

// ------------------------------------------------------------------------------------------------
// Bucket merge on records (k_bucket_merge of mcr_kernels.hpp): gather <= k sorted pieces, merge in LDS, write the
// pooled ascending records and -- fused -- the rank code of every draw to time order.
// ------------------------------------------------------------------------------------------------
template <int NT, int VT>
__global__ __launch_bounds__(NT) void k_bucket_merge32(const u64* __restrict__ rin, u64* __restrict__ rout, i64 M,
                                                       int k, int B, const u32* __restrict__ cut,
                                                       const u32* __restrict__ boff, u32* __restrict__ z, i64 P, i64 R)
{
    constexpr int T = NT * VT;
    static_assert(T == 4096, "the partition bound of k_splitters assumes 4096-slot buckets");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* srec = reinterpret_cast<u64*>(smem);
    int* sst = reinterpret_cast<int*>(srec + T + 2);   // padded piece starts [k+1], then scratch
    int* spl = sst + 40;                                // piece lengths [k]
    int* sps = spl + 40;                                // piece source offsets in tile [k]
    i64* sedge = reinterpret_cast<i64*>(sps + 40);      // [4] global run bounds of the edge values

    const int tid = threadIdx.x;
    i64 p;
    int b;
    if (!xcd_map(P, B, p, b)) return;
    const u64* rp = rin + p * M;
    const u32* c0 = cut + (p * (B + 1) + b) * k;
    const u32* c1 = c0 + k;
    if (tid == 0) {
        int acc = 0;
        for (int t = 0; t < k; ++t) {
            const int len = (int)(c1[t] - c0[t]);
            sst[t] = acc; spl[t] = len; sps[t] = (int)c0[t];
            acc += (len + 15) & ~15;
        }
        sst[k] = acc;
    }
    __syncthreads();
    const int padded = sst[k];
    const int total = (int)(boff[p * (B + 1) + b + 1] - boff[p * (B + 1) + b]);
    const i64 obase = boff[p * (B + 1) + b];
    if (padded > T - 64 || total < 0 || total > padded || obase + total > M) return;   // never with a valid partition
    {   // gather: one piece search per 16-slot chunk, all VT loads in flight together (see k_bucket_merge)
        const int lane = tid & 63, wv = tid >> 6;
        u32 my_g = 0; int my_n = 0;
        if ((lane & 15) < VT) {
            const int e0 = (lane & 15) * NT + 64 * wv + 16 * (lane >> 4);
            if (e0 < padded) {
                int t = 0;
#pragma unroll
                for (int step = 8; step > 0; step >>= 1)
                    if (t + step < k && e0 >= sst[t + step]) t += step;
                const int o = e0 - sst[t];
                my_n = spl[t] - o;
                my_g = (u32)((i64)t * R) + (u32)sps[t] + (u32)o;
            }
        }
        u64 gr[VT];
#pragma unroll
        for (int j = 0; j < VT; ++j) {
            const int within = lane & 15;
            const u32 g0 = (u32)row_lane((int)my_g, j);      // lane j of my row of 16 looked the chunk up
            const int n = row_lane(my_n, j);
            const bool live = within < n;
            const u64 r = rp[live ? (i64)(g0 + (u32)within) : 0];
            gr[j] = live ? r : kRecPad;
        }
#pragma unroll
        for (int j = 0; j < VT; ++j) {
            const int e = j * NT + tid;
            if (e < padded) srec[pos16(e)] = gr[j];
        }
    }
    __syncthreads();
    const int chunk0 = tid * VT;
    int mypiece = 0;      // the piece that holds slot chunk0 (pieces are padded to 16 >= VT slots)
#pragma unroll
    for (int step = 8; step > 0; step >>= 1)
        if (mypiece + step < k && chunk0 >= sst[mypiece + step]) mypiece += step;
    for (int w = 1; w < k; w <<= 1) {
        u64 r[VT];
        bool moved = false;
        if (chunk0 < padded) {
            const int ra = mypiece & ~(2 * w - 1);
            const int a0 = sst[ra];
            const int a1 = sst[(ra + w < k) ? ra + w : k];
            const int b1 = sst[(ra + 2 * w < k) ? ra + 2 * w : k];
            const int na = a1 - a0, nb = b1 - a1, diag = chunk0 - a0;
            moved = nb > 0;
            if (moved) {
                auto A = [&](int i) { return srec[pos16(a0 + i)]; };
                auto Bf = [&](int j) { return srec[pos16(a1 + j)]; };
                const int ai = merge_path32(A, na, Bf, nb, diag);
                serial_merge_rec<VT>(srec, a0, na, a1, nb, ai, diag - ai, VT, r);
            }
        }
        __syncthreads();
        if (moved) {
#pragma unroll
            for (int i = 0; i < VT; ++i) srec[pos16(chunk0 + i)] = r[i];
        }
        __syncthreads();
    }
    for (int e = tid; e < total; e += NT) rout[p * M + obase + e] = srec[pos16(e)];
    if (z == nullptr || total == 0) return;
    // tie runs that continue in a neighbouring bucket: look at the record just before / after this bucket's piece in
    // every run; only then pay for the bound searches
    if (tid < 4) sedge[tid] = 0;
    const u32 kfirst = rec_key(srec[pos16(0)]), klast = rec_key(srec[pos16(total - 1)]);
    if (tid < 64) {
        bool e0 = false, e1 = false;
        if (tid < k) {
            const i64 tbase = (i64)tid * R;
            const int cnt = (int)((M - tbase < R) ? M - tbase : R);
            const int lo = sps[tid], hi = sps[tid] + spl[tid];
            if (lo > 0) e0 = rec_key(rp[tbase + lo - 1]) == kfirst;
            if (hi < cnt) e1 = rec_key(rp[tbase + hi]) == klast;
        }
        const bool a0 = __ballot(e0) != 0, a1 = __ballot(e1) != 0;
        if (tid == 0) { sst[36] = a0; sst[37] = a1; }
    }
    __syncthreads();
    const bool ext0 = sst[36] != 0, ext1 = sst[37] != 0;
    if ((ext0 || ext1) && tid < 2 * k) {
        const int t = tid % k, which = tid / k;           // 0: first value, 1: last value
        const u32 v = which ? klast : kfirst;
        const i64 tbase = (i64)t * R;
        const int cnt = (int)((M - tbase < R) ? M - tbase : R);
        const u64* a = rp + tbase;
        int lo = 0, hi = cnt;
        while (lo < hi) { const int m = (lo + hi) >> 1; if (rec_key(a[m]) < v) lo = m + 1; else hi = m; }
        const int lb = lo;
        hi = cnt;
        while (lo < hi) { const int m = (lo + hi) >> 1; if (!(v < rec_key(a[m]))) lo = m + 1; else hi = m; }
        atomicAdd(reinterpret_cast<unsigned long long*>(&sedge[2 * which]), (unsigned long long)lb);
        atomicAdd(reinterpret_cast<unsigned long long*>(&sedge[2 * which + 1]), (unsigned long long)lo);
    }
    __syncthreads();
    int rs[VT], re[VT];
    block_tie_runs<NT, VT>([&](int g) { return rec_key(srec[pos16(g)]); }, total, sst + 24, rs, re);
#pragma unroll
    for (int i = 0; i < VT; ++i) {
        const int e = tid * VT + i;
        if (e < total) {
            const u64 r = srec[pos16(e)];
            i64 gs = obase + rs[i], ge = obase + re[i];
            if (ext0 && rec_key(r) == kfirst) gs = sedge[0];
            if (ext1 && rec_key(r) == klast) ge = sedge[3];
            z[p * M + min(rec_pos(r), (u32)(M - 1))] = (u32)(gs + ge);   // code of the tie run: rank = (code + 1) / 2
        }
    }
}

}  // namespace mcr
