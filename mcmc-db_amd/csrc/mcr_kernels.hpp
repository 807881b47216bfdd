// mcr_kernels.hpp -- the HIP kernels of the statistics hot path (gfx950, wave64, fp64 VALU).
//
// Per model (C chains x N draws, M = C*N pooled draws per parameter, P parameters) the pipeline is
//
//   k_ingest_*      (only if the tensor is not in the Arrow layout [P][C][N])  strided -> X[P][M] f64
//   k_tile_sort     X (f64, or f32 widened on load) -> sorted 4096-draw tiles of (key f64, pooled position u16 | u32)
//                   + the tile's (count, mean, M2) + every 64th order statistic of each tile
//                   (f32 tensors: the packed-record kernels of mcr_sort32.hpp take the first four places of this list)
//   k_merge<false>  only for M > 64K: pairwise merge-path passes until at most 16 sorted runs remain
//                   (k_sample_runs then takes the regular samples of those runs)
//   k_splitters     exact k-way partition by regular sampling: deterministic bucket bound
//   k_bucket_merge  per bucket: gather <= 16 sorted pieces, merge in LDS, write the pooled ascending
//                   order, and (fused) tie-averaged ranks -> rank code -> time order                 (a7)
//   (order statistics: quantiles, median, fold split point -- by the first wave of every fold workgroup;
//    a k_order_stats launch of their own for Backend.stats calls and for pooled arrays beyond 64K)      (a2/a3/a8)
//   k_merge<true>   |x - med| order by ONE merge of the two monotone halves (no second sort), fused
//                   with the rank codes of the folded values, their scatter write-combined in LDS   (a8, a7)
//   k_acov_seg / k_diag_combine(2) / k_tier3 (or k_long_list + k_dev_fill + FFT | k_acov_long + k_diag_long_scan)
//                   (mcr_diag.hpp, mcr_fft.hpp) split R-hat + ESS for bulk and folded z, lags in three tiers;
//                   combine2 also packs mean / std / rhat = pymax(bulk, tail) into the result table (k_finalize
//                   does that for calls without diagnostics)                                       (a9-a13)
//   (k_rank_z: the standalone rank kernel, used only when M > 512K and no bucket partition applies)
//
// (aN) = row of SURVEY.md section 8(a); reference file:line citations are next to each kernel.
#pragma once
#include "mcr_device.hpp"
#include "mcr_sortnet.h"

namespace mcr {

// result table: SoA, res[field * P + p]
enum ResField {
    R_MEAN = 0, R_STD, R_MEDIAN, R_RHAT, R_RHAT_BULK, R_RHAT_TAIL, R_ESS_BULK, R_ESS_TAIL,
    R_LAG_BULK, R_LAG_TAIL, R_BAD, R_Q0  // R_Q0 .. R_Q0 + nq - 1
};

#ifndef MCR_WCOMB
#define MCR_WCOMB 1          // 0: the fold kernel scatters its codes as they come (A/B builds)
#endif
constexpr bool kWriteCombine = MCR_WCOMB != 0;

struct QArgs {
    int nq;
    i64 lo[32];
    double g[32];
};

// XCD-aware block -> (parameter, block-within-parameter) map for kernels that scatter into one
// parameter's z array.  Workgroups are dealt round-robin over the 8 XCDs (blocks L and L+8 share
// an XCD, each XCD has its own L2), so parameter p runs all its nb blocks on XCD p % 8, back to
// back: its 8-byte scattered z stores then merge into whole lines in ONE L2 instead of leaving
// partial sectors in eight.  Launch with a 1-D grid of ceil(P/8)*8*nb blocks.  Speed only.
__device__ __forceinline__ bool xcd_map(i64 P, int nb, i64& p, int& blk)
{
    const i64 L = blockIdx.x;
    const i64 xcd = L & 7, j = L >> 3;
    p = (j / nb) * 8 + xcd;
    blk = (int)(j % nb);
    return p < P;
}

// ------------------------------------------------------------------------------------------------
// z lookup table.  A draw's z depends only on its tie run [s, e) and M:
//     rank = (s + 1 + e) / 2,   z = Phi^-1((rank - 0.5) / M)          (diagnostics.py:117,130-131)
// so z is a function of the integer n2 = s + e in [1, 2M-1] alone -- the same for every parameter
// of the model.  One tiny launch evaluates it for all 2M values (same operations, hence the same
// bits as evaluating it per draw).  The rank kernels therefore scatter only the 4-byte CODE n2 of
// each draw to time order (half the bytes and half the L2 footprint of scattering z itself: the
// scattered stores are the expensive part), and the consumers (k_acov_seg,
// k_diag_combine*) turn codes into z with one read of this L2-resident table.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ztable(double* __restrict__ ztab, i64 M)
{
    const i64 n2 = (i64)blockIdx.x * 256 + threadIdx.x;
    if (n2 >= 2 * M) return;
    const double r = (double)(n2 + 1) / 2.0;
    ztab[n2] = (n2 >= 1) ? inv_cdf((r - 0.5) / (double)M) : 0.0;
}

// Codes -> z and average ranks in time order (debug outputs of mcr_diagnose_chains only).
__global__ __launch_bounds__(256) void k_decode_codes(const u32* __restrict__ code, const double* __restrict__ ztab,
                                                      i64 n, double* __restrict__ z, double* __restrict__ rank)
{
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u32 c = code[i];
    if ((i64)c > 2 * n - 1) c = (u32)(2 * n - 1);
    if (z) z[i] = ztab[c];
    if (rank) rank[i] = (double)((i64)c + 1) / 2.0;
}

// ------------------------------------------------------------------------------------------------
// Ingest: arbitrary element strides / f32 -> X[p][c*N + t] f64.
// rows variant: lanes run along t (coalesced when stride_n == 1).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_ingest_rows(const T* __restrict__ src, double* __restrict__ X,
                                                     i64 C, i64 N, i64 sc, i64 sn, i64 sp, i64 p0)
{
    const i64 nb = (N + 255) / 256;
    const i64 c = blockIdx.x / nb, t = (blockIdx.x % nb) * 256 + threadIdx.x;
    const i64 p = blockIdx.y;
    if (t < N) X[p * (C * N) + c * N + t] = (double)src[c * sc + t * sn + (p0 + p) * sp];
}

// transpose variant: lanes run along p on the read side (coalesced when stride_p == 1, i.e. the
// [C][N][P] layout of Draws.to_numpy, src/mcmc_ref/draws.py:28-29), along t on the write side.
template <typename T>
__global__ __launch_bounds__(256) void k_ingest_transpose(const T* __restrict__ src,
                                                          double* __restrict__ X, i64 C, i64 N, i64 P,
                                                          i64 sc, i64 sn, i64 sp, i64 p0)
{
    __shared__ double tile[64][65];
    const i64 nb = (N + 63) / 64;
    const i64 c = blockIdx.x / nb, t0 = (blockIdx.x % nb) * 64;
    const i64 pl0 = (i64)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const i64 t = t0 + r, p = pl0 + tx;
        tile[r][tx] = (t < N && p < P) ? (double)src[c * sc + t * sn + (p0 + p) * sp] : 0.0;
    }
    __syncthreads();
    for (int q = ty; q < 64; q += 4) {
        const i64 t = t0 + tx, p = pl0 + q;
        if (t < N && p < P) X[p * (C * N) + c * N + t] = tile[tx][q];
    }
}

// ------------------------------------------------------------------------------------------------
// Streaming moments (a2/a3/a4): one HBM pass.  Every workgroup accumulates shifted sums S1 = sum(x-K),
// S2 = sum((x-K)^2) of ITS slice around a pivot K taken from that slice (the mean of its first 64 draws, so a single
// outlying draw cannot become the pivot), converts them to the slice's (count, mean, M2 = sum (x - mean)^2), and the
// finisher merges the slices with Chan's pairwise update.  Same result as the reference's two-pass forms -- np.mean /
// np.std(ddof=0) (src/mcmc_ref/backends_numpy.py:41-42), pc.mean / pc.stddev (backends_arrow.py:38-39),
// compute_basic_stats (compare.py:58-64) -- to a few ulp even when a chain starts far from its bulk (an unconverged
// `actual` handed to compare()): the cancellation is bounded by the spread INSIDE a slice, not by |first draw - mean|.
// part[(p*S + s)*kMomRec + {0..4}] = slice pivot K, slice mean - K, slice M2 (about its mean), non-finite count, slice
// length.  The mean travels as the pair (K, mean - K): rounded to one double it would carry an error of ulp(mean) / 2,
// which Chan's update multiplies by the distance between slice means -- a relative error of eps |mean| / sigma /
// sqrt(slice) in the variance, 1e-12 for draws of spread 1e-6 around 3.
// rows variant: grid (S, P); the block streams a contiguous slice of X[p][.] with 16-byte loads.
// ------------------------------------------------------------------------------------------------
constexpr int kMomRec = 5;

__device__ __forceinline__ void store_slice_moments(double* o, double K, double s1, double s2, double bad, double n)
{
    const double dm = (n > 0.0) ? s1 / n : 0.0;         // s1 = sum (x - K), s2 = sum (x - K)^2
    double m2 = s2 - s1 * dm;
    if (!(m2 >= 0.0) && !isnan(m2)) m2 = 0.0;          // tiny negative from rounding; NaN / +inf pass through
    o[0] = K; o[1] = dm; o[2] = m2; o[3] = bad; o[4] = n;
}

// (mean, M2, bad) of a parameter from its S slice records (fixed order): Chan et al.'s combination
//     M2 = sum_s M2_s + sum_s n_s (mean_s - mean)^2,  mean = m_0 + sum_s n_s (mean_s - m_0) / N,
// with every mean difference formed as (K_s - K_0) + (dm_s - dm_0): the pivots of slices of one parameter lie close
// together, so the first term is exact, and the second is a difference of small corrections.
__device__ __forceinline__ void merge_slice_moments(const double* __restrict__ rec, int S, double& mean, double& m2,
                                                    double& bad, double& n)
{
    const double K0 = rec[0], d0 = rec[1];
    double acc = 0.0, N = 0.0, b = 0.0;
    for (int s = 0; s < S; ++s) {
        const double* o = rec + (i64)s * kMomRec;
        acc += o[4] * ((o[0] - K0) + (o[1] - d0)); N += o[4]; b += o[3];
    }
    const double mr = (N > 0.0) ? acc / N : NAN;        // the mean, relative to K0 + d0
    double q = 0.0;
    for (int s = 0; s < S; ++s) {
        const double* o = rec + (i64)s * kMomRec;
        const double d = ((o[0] - K0) + (o[1] - d0)) - mr;
        q += o[2]; q = fma(o[4] * d, d, q);
    }
    mean = K0 + (d0 + mr); m2 = q; bad = b; n = N;
}

template <typename T>
__global__ __launch_bounds__(256) void k_moments_rows(const T* __restrict__ X, i64 M, i64 pstride,
                                                      double* __restrict__ part, int S)
{
    __shared__ double red[12];
    const i64 p = blockIdx.y;
    const int s = blockIdx.x;
    const T* x = X + p * pstride;
    constexpr int V = 16 / sizeof(T);  // elements per 16-byte load
    i64 per = (M + S - 1) / S;
    per = (per + V - 1) / V * V;
    const i64 b = s * per, e = (b + per < M) ? b + per : M;
    // pivot: mean of the slice's first (up to) 64 draws, the same value in every lane
    double K = 0.0;
    {
        const int lane = threadIdx.x & 63;
        const i64 cnt = (e - b < 64) ? e - b : 64;
        K = (lane < cnt) ? (double)x[b + lane] : 0.0;
        K = wave_sum(K) / (double)(cnt > 0 ? cnt : 1);
        if (!isfinite(K)) K = 0.0;                      // non-finite draws: the call is rejected anyway; keep the sums defined
    }
    double s1 = 0.0, s2 = 0.0, bad = 0.0;
    auto acc1 = [&](double v) {
        const double d = v - K;
        s1 += d; s2 = fma(d, d, s2);
        bad += isfinite(v) ? 0.0 : 1.0;
    };
    i64 scalar_from = b;
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0 && e > b) {
        const i64 nvec = (e - b) / V;  // b is a multiple of V, so x + b stays 16-byte aligned
        const float4* xv = reinterpret_cast<const float4*>(x + b);
        i64 g = threadIdx.x;
        for (; g + 3 * 256 < nvec; g += 4 * 256) {  // 4 independent 16-byte loads in flight per lane
            float4 r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) r[u] = xv[g + u * 256];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const T* v = reinterpret_cast<const T*>(&r[u]);
#pragma unroll
                for (int j = 0; j < V; ++j) acc1((double)v[j]);
            }
        }
        for (; g < nvec; g += 256) {
            const float4 r = xv[g];
            const T* v = reinterpret_cast<const T*>(&r);
#pragma unroll
            for (int j = 0; j < V; ++j) acc1((double)v[j]);
        }
        scalar_from = b + nvec * V;
    }
    for (i64 j = scalar_from + threadIdx.x; j < e; j += 256) acc1((double)x[j]);
    block_sum3<256>(s1, s2, bad, red);
    if (threadIdx.x == 0) store_slice_moments(part + (p * S + s) * kMomRec, K, s1, s2, bad, (double)(e > b ? e - b : 0));
}

// strided variant ([C][N][P]-like tensors, stride_p == 1): lanes run along p, each thread owns one
// parameter and walks a slice of the C*N rows.  grid (ceil(P/64), S), block (64, 4).
template <typename T>
__global__ __launch_bounds__(256) void k_moments_cols(const T* __restrict__ src, i64 C, i64 N, i64 P,
                                                      i64 sc, i64 sn, i64 sp, double* __restrict__ part,
                                                      int S)
{
    __shared__ double sh[3][4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const i64 p = (i64)blockIdx.x * 64 + tx;
    const int s = blockIdx.y;
    const i64 M = C * N;
    const i64 per = (M + S - 1) / S, b = s * per, e = (b + per < M) ? b + per : M;
    double s1 = 0.0, s2 = 0.0, bad = 0.0, K = 0.0;
    if (p < P && b < e) {
        { const i64 c = b / N, t = b - c * N; K = (double)src[c * sc + t * sn + p * sp]; }     // pivot: first draw of the slice
        if (!isfinite(K)) K = 0.0;
        for (i64 r = b + ty; r < e; r += 4) {
            const i64 c = r / N, t = r - c * N;
            const double v = (double)src[c * sc + t * sn + p * sp], d = v - K;
            s1 += d; s2 = fma(d, d, s2);
            bad += isfinite(v) ? 0.0 : 1.0;
        }
    }
    sh[0][ty][tx] = s1; sh[1][ty][tx] = s2; sh[2][ty][tx] = bad;
    __syncthreads();
    if (ty == 0 && p < P)
        store_slice_moments(part + (p * S + s) * kMomRec, K, sh[0][0][tx] + sh[0][1][tx] + sh[0][2][tx] + sh[0][3][tx],
                            sh[1][0][tx] + sh[1][1][tx] + sh[1][2][tx] + sh[1][3][tx],
                            sh[2][0][tx] + sh[2][1][tx] + sh[2][2][tx] + sh[2][3][tx], (double)(e > b ? e - b : 0));
}

// mean/std from the S slice records of every parameter (fixed summation order).
// Writes mean[p], std[p] directly (mcr_moments_dev / mcr_basic_stats).
__global__ void k_moments_final(const double* __restrict__ part, int S, i64 P, double* __restrict__ mean,
                                double* __restrict__ stdv, double* __restrict__ bad)
{
    const i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    double mu, m2, b, n;
    merge_slice_moments(part + p * S * kMomRec, S, mu, m2, b, n);
    double var = m2 / n;
    if (isinf(m2)) var = INFINITY;          // squares overflowed: inf like np.std
    else if (!(var >= 0.0)) var = 0.0;
    mean[p] = mu;
    stdv[p] = sqrt(var);
    if (bad) bad[p] = b;
}

// ------------------------------------------------------------------------------------------------
// Tile sort: one workgroup sorts T = NT*VT pooled draws of one parameter entirely in LDS
// (thread-local odd-even network over VT registers, then log2(NT) merge-path levels), carrying the
// pooled position as a u32 payload.  Spec: the `sorted(flat, key=...)` of
// src/mcmc_ref/diagnostics.py:110 (stability is irrelevant: ties share one average rank).
// Also emits the moments (count, mean, M2) of its tile (so the draws are read from HBM once).
// Partial tiles are padded with +inf (draws are finite or the call fails with MCR_ENONFINITE).
// ------------------------------------------------------------------------------------------------
// 16 registers per lane, fully static 60-comparator network (mcr_sortnet.h): every index is a
// compile-time constant, so the keys stay in VGPRs (a loop-carried index here makes hipcc emulate
// dynamic register indexing with 16-way select chains: 10x the instructions).
template <int VT>
__device__ __forceinline__ void thread_sort(double (&k)[VT], u32 (&ix)[VT])
{
    static_assert(VT == 16 || VT == 8, "sorting networks exist for 16 and 8 items per lane");
#define MCR_CE(a, b)                                                        \
    {                                                                       \
        const bool sw = k[b] < k[a];                                        \
        const double lo = sw ? k[b] : k[a], hi = sw ? k[a] : k[b];          \
        const u32 ilo = sw ? ix[b] : ix[a], ihi = sw ? ix[a] : ix[b];       \
        k[a] = lo; k[b] = hi; ix[a] = ilo; ix[b] = ihi;                     \
    }
    if constexpr (VT == 16) { MCR_NET16(MCR_CE) } else { MCR_NET8(MCR_CE) }
#undef MCR_CE
}

#ifndef MCR_TILE_DPP_LEVELS
#define MCR_TILE_DPP_LEVELS 3      // 0: every merge level in LDS; 2 / 3: the first two / three in registers (below)
#endif

// The first two merge levels of the tile sort (16 -> 32 -> 64 sorted draws) without the LDS: bitonic merges across 2 and 4
// lanes.  Lane L of a group holds the sorted draws 16 L .. 16 L + 15 of the group's run in its registers.  To merge the
// ascending runs A and B of a group, draw e of A meets draw n-1-e of B (the partner lane is the mirror lane of the
// group, the partner register the mirror register): the A side keeps the minima, the B side the maxima, which leaves
// two bitonic sequences with every draw of the first at or below every draw of the second; half-cleaners at distance
// 16 (partner lane L ^ 1, same register; 4-lane level only) and then 8, 4, 2, 1 inside the lane sort each of them.
// The partner's registers arrive by DPP quad permutes (a VALU operand path, no LDS traffic, no waitcnt); a lane reads
// its partner's OLD registers because the wave executes the permutes of a register pair before it writes either.
// 52 VALU per draw for the two levels, no LDS, against 72 VALU + 9 LDS instructions per draw for the same two levels on
// the merge-path route.  A third level (8 lanes: row_half_mirror, then distances 32 and 16) costs the VALU of its LDS
// form and still wins for the LDS instructions and the barrier pair it saves; a fourth (16 lanes: lane ^ 4 takes two
// DPP movs per word) measured slower than its LDS form (tile sort 396 -> 408 us on 1000 parameters).  Ties keep their own side (stability is irrelevant: equal draws share one average rank).
// b where the lane's bit of m is set, else a: one v_cndmask on a mask built on the scalar unit (the compiler's own lowering
// of a per-lane choice between two compare results goes through 0 / 1 in VGPRs: six more VALU per exchange).
__device__ __forceinline__ u32 select_by_mask(u32 a, u32 b, unsigned long long m)
{
    u32 r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
    return r;
}
__device__ __forceinline__ double select_by_mask(double a, double b, unsigned long long m)
{
    return __hiloint2double((int)select_by_mask((u32)__double2hiint(a), (u32)__double2hiint(b), m),
                            (int)select_by_mask((u32)__double2loint(a), (u32)__double2loint(b), m));
}

// MIN_LANES: the lanes that keep the minimum of an exchange (the others keep the maximum)
template <int CTRL, bool MIRROR, unsigned long long MIN_LANES>
__device__ __forceinline__ void lane_pair_stage(double (&k)[16], u32 (&ix)[16])
{
    auto exchange = [](double& own, u32& own_i, double o, u32 oi) {
        const unsigned long long lt = __builtin_amdgcn_ballot_w64(o < own), gt = __builtin_amdgcn_ballot_w64(own < o);
        const unsigned long long take = (lt & MIN_LANES) | (gt & ~MIN_LANES);
        own = select_by_mask(own, o, take);
        own_i = select_by_mask(own_i, oi, take);
    };
#pragma unroll
    for (int r = 0; r < (MIRROR ? 8 : 16); ++r) {
        const int s = MIRROR ? 15 - r : r;                  // my register r meets the partner's register s
        const double o_r = dpp_f64<CTRL>(k[s]);
        const u32 oi_r = dpp_u32<CTRL>(ix[s]);
        if (MIRROR) {
            const double o_s = dpp_f64<CTRL>(k[r]);         // ... and my register s the partner's register r
            const u32 oi_s = dpp_u32<CTRL>(ix[r]);
            exchange(k[s], ix[s], o_s, oi_s);
        }
        exchange(k[r], ix[r], o_r, oi_r);
    }
}

__device__ __forceinline__ void lane_bitonic_merge16(double (&k)[16], u32 (&ix)[16])
{
#define MCR_CE(a, b)                                                        \
    {                                                                       \
        const bool sw = k[b] < k[a];                                        \
        const double lo = sw ? k[b] : k[a], hi = sw ? k[a] : k[b];          \
        const u32 ilo = sw ? ix[b] : ix[a], ihi = sw ? ix[a] : ix[b];       \
        k[a] = lo; k[b] = hi; ix[a] = ilo; ix[b] = ihi;                     \
    }
    // half-cleaners at distance 8, 4, 2, 1 (mcr_sortnet.h); every index a literal (a loop-carried index would make hipcc
    // emulate dynamic register indexing with 16-way select chains)
    MCR_BITONIC16(MCR_CE)
#undef MCR_CE
}

// registers sorted per lane -> runs of 64 draws sorted across every aligned group of 4 lanes
__device__ __forceinline__ void lane_merge_levels_16_to_64(double (&k)[16], u32 (&ix)[16])
{
    constexpr unsigned long long kEvenLanes = 0x5555555555555555ull, kLowPairs = 0x3333333333333333ull;
    lane_pair_stage<0xB1, true, kEvenLanes>(k, ix);       // quad_perm [1,0,3,2]: 2 x 16 -> 32
    lane_bitonic_merge16(k, ix);
    lane_pair_stage<0x1B, true, kLowPairs>(k, ix);        // quad_perm [3,2,1,0]: 2 x 32 -> 64, mirror lane of the quad
    lane_pair_stage<0xB1, false, kEvenLanes>(k, ix);      // half-cleaner at distance 16
    lane_bitonic_merge16(k, ix);
#if MCR_TILE_DPP_LEVELS >= 3
    constexpr unsigned long long kLowQuads = 0x0F0F0F0F0F0F0F0Full;
    lane_pair_stage<0x141, true, kLowQuads>(k, ix);       // row_half_mirror: 2 x 64 -> 128, mirror lane of the group of 8
    lane_pair_stage<0x4E, false, kLowPairs>(k, ix);       // quad_perm [2,3,0,1]: half-cleaner at distance 32
    lane_pair_stage<0xB1, false, kEvenLanes>(k, ix);      // half-cleaner at distance 16
    lane_bitonic_merge16(k, ix);
#endif
}

// Serial merge of up to VT outputs from LDS runs A = skey[pos16(a0 + .)] (na items) and
// B = skey[pos16(b0 + .)] (nb items), starting at (ai, bi).  src[i] = LDS slot the output came from.
// Branch-free.  Which run is exhausted is decided on the POINTERS (two integer compares whose lane masks are combined
// with the key compare on the scalar unit), so the head values need no sanitising: the LDS read of the next head is
// unconditional (one slot past a run is still inside the workgroup's LDS) and whatever it returns past the end of a
// run is never selected.  Ties take from A.  ~19 VALU per output (25 with the +inf substitution this replaced).
template <int VT>
__device__ __forceinline__ void serial_merge(const double* skey, int a0, int na, int b0, int nb, int ai,
                                             int bi, int nout, double (&k)[VT], int (&src)[VT])
{
    int pa = a0 + ai, pb = b0 + bi;
    const int ea = a0 + na, eb = b0 + nb;
    double ak = skey[pos16(pa)], bk = skey[pos16(pb)];
#pragma unroll
    for (int i = 0; i < VT; ++i) {
        if (i < nout) {
            const bool takeA = (pb >= eb) | ((pa < ea) & !(bk < ak));
            k[i] = takeA ? ak : bk;
            src[i] = takeA ? pa : pb;
            pa += takeA ? 1 : 0;
            pb += takeA ? 0 : 1;
            const int pn = takeA ? pa : pb;
            const double nv = skey[pos16(pn)];
            ak = takeA ? nv : ak;
            bk = takeA ? bk : nv;
        } else {
            src[i] = 0;
        }
    }
}

// IdxT = the pooled-position payload: u16 when M < 65536 (C1 and every packaged model: 10 bytes per LDS slot, so FOUR
// 4096-draw tiles are resident per CU instead of three, and 10 instead of 12 bytes per draw go to HBM and back),
// u32 otherwise.  LDS: skey[T] then sidx[T], nothing else -- the single read "one slot past the last run" of
// serial_merge lands on sidx[0] (in bounds, value never selected); the reduction scratch reuses skey at the end.
template <typename IdxT> constexpr size_t sort_lds_bytes(int T) { return (size_t)T * (8 + sizeof(IdxT)); }

template <int NT, int VT, typename IdxT, typename XT>
__global__ __launch_bounds__(NT) void k_tile_sort(const XT* __restrict__ X, i64 M,
                                                  double* __restrict__ keys, IdxT* __restrict__ idx,
                                                  double* __restrict__ part, int ntiles,
                                                  double* __restrict__ samp)
{
    constexpr int T = NT * VT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* skey = reinterpret_cast<double*>(smem);
    IdxT* sidx = reinterpret_cast<IdxT*>(skey + T);
    double* red = skey;      // only after the sorted tile has left LDS (barrier below)

    const int tid = threadIdx.x, tile = blockIdx.x;
    const i64 p = blockIdx.y;
    const i64 base = (i64)tile * T;
    const int count = (int)((M - base < (i64)T) ? M - base : (i64)T);
    const XT* src = X + p * M + base;    // XT = float: f32 tensors in the Arrow layout are widened here, not by an ingest pass

    double bad = 0.0;
#pragma unroll
    for (int i = 0; i < VT; ++i) {
        const int e = i * NT + tid;
        double v = INFINITY;
        if (e < count) {
            v = (double)src[e];
            bad += isfinite(v) ? 0.0 : 1.0;
        }
        skey[pos16(e)] = v;
    }
    __syncthreads();

    double k[VT];
    u32 ix[VT];
#pragma unroll
    for (int i = 0; i < VT; ++i) {
        const int e = tid * VT + i;
        k[i] = skey[pos16(e)];
        ix[i] = (e < count) ? (u32)(base + e) : 0xFFFFFFFFu;
    }
    thread_sort<VT>(k, ix);
    constexpr bool kLaneLevels = MCR_TILE_DPP_LEVELS >= 2 && VT == 16;
    if constexpr (kLaneLevels) lane_merge_levels_16_to_64(k, ix);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VT; ++i) {
        const int e = tid * VT + i;
        skey[pos16(e)] = k[i];
        sidx[posi(e)] = (IdxT)ix[i];
    }
    __syncthreads();

    for (int coop = kLaneLevels ? (MCR_TILE_DPP_LEVELS >= 3 ? 16 : 8) : 2; coop <= NT; coop <<= 1) {
        const int first = tid & ~(coop - 1);
        const int run = VT * (coop >> 1);
        const int a0 = first * VT, b0 = a0 + run;
        const int diag = VT * (tid - first);
        auto A = [&](int i) { return skey[pos16(a0 + i)]; };
        auto B = [&](int j) { return skey[pos16(b0 + j)]; };
        const int ai = merge_path32(A, run, B, run, diag);
        int srcs[VT];
        serial_merge<VT>(skey, a0, run, b0, run, ai, diag - ai, VT, k, srcs);
#pragma unroll
        for (int i = 0; i < VT; ++i) ix[i] = sidx[posi(srcs[i])];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < VT; ++i) {
            const int e = tid * VT + i;
            skey[pos16(e)] = k[i];
            sidx[posi(e)] = (IdxT)ix[i];
        }
        __syncthreads();
    }

    for (int e = tid; e < count; e += NT) {
        keys[p * M + base + e] = skey[pos16(e)];
        idx[p * M + base + e] = sidx[posi(e)];
    }
    // regular samples (every 64th order statistic of the tile) for the exact bucket partition
    if (samp != nullptr && tid < T / 64) {
        const int e = 64 * tid + 63;
        samp[(p * ntiles + tile) * (T / 64) + tid] = (e < count) ? skey[pos16(e)] : INFINITY;
    }
    __syncthreads();                 // `red` aliases skey from here on
    // Moments of the tile, two-pass like the reference (mean first, then squared deviations; compare.py:62-63), from the
    // lane's VT sorted draws still in registers (the draws are read from HBM exactly once): k_finalize merges the
    // tiles with Chan's update.  Slots at or beyond `count` hold the +inf pads.
    double s1 = 0.0;
#pragma unroll
    for (int i = 0; i < VT; ++i) s1 += (tid * VT + i < count) ? k[i] : 0.0;
    s1 = block_sum<NT>(s1, red);
    const double mt = s1 / (double)count;
    double s2 = 0.0, e1 = 0.0;       // second pass about mt; e1 = sum (x - mt) is what rounding left of the mean
#pragma unroll
    for (int i = 0; i < VT; ++i) { const double d = (tid * VT + i < count) ? k[i] - mt : 0.0; s2 = fma(d, d, s2); e1 += d; }
    block_sum3<NT>(s2, e1, bad, red);
    if (tid == 0) store_slice_moments(part + (p * ntiles + tile) * kMomRec, mt, e1, s2, bad, (double)count);
}

// ------------------------------------------------------------------------------------------------
// Write-combined scatter of the rank codes to time order (round 3).  A wave's store of 64 codes for 64 unrelated draws
// touches 64 different 128-byte lines of the parameter's code array, and the CU's L1 takes them one line per cycle: 4 M
// scattered stores per call were a third of the fold kernel.  But a workgroup's <= 4032 codes go to an array of M / 32
// lines (1 250 for C1): about three of them per line.  So the workgroup counting-sorts its (position, code) pairs by
// LINE inside its LDS first -- one LDS atomic per pair for the rank in its line, one block scan over the line counters
// -- and stores them in that order: consecutive lanes then write ascending addresses, ~20 lines per wave-store instead
// of 64.  Fold: 294 -> 250 us per 1000 parameters.  Which code lands where is unchanged.  For M < 65536 (16-bit
// positions; at most 2048 lines).
// hcode / hpos: the lane's VT (code, pooled position) pairs, live for e = tid * VT + i < total.  cnt_code: >= 2112 + 4096
// dead u32 of LDS (the key array), spos: 4096 dead u16 (the position array), swt: NT / 64 u32 of scratch.
// ------------------------------------------------------------------------------------------------
template <int NT, int VT>
__device__ __forceinline__ void scatter_codes_by_line(u32* __restrict__ zrow, i64 M, int total, const u32 (&hcode)[VT],
                                                      const u32 (&hpos)[VT], u32* cnt_code, unsigned short* spos, u32* swt)
{
    constexpr int kMaxLines = 2048, PER = kMaxLines / NT > 0 ? kMaxLines / NT : 1;
    static_assert(NT * PER >= kMaxLines, "every line counter has an owner in the scan");
    const int tid = threadIdx.x;
    const int nlines = (int)((M + 31) >> 5);
    u32* scnt = cnt_code;                   // [nlines + 1]: counts, then first slot of every line
    u32* scode = cnt_code + 2112;
    __syncthreads();                        // the caller's last reads of the arrays that are reused here
    for (int i = tid; i <= nlines; i += NT) scnt[i] = 0u;
    __syncthreads();
    u32 rk[VT];
#pragma unroll
    for (int i = 0; i < VT; ++i) rk[i] = (tid * VT + i < total) ? atomicAdd(&scnt[hpos[i] >> 5], 1u) : 0u;
    __syncthreads();
    {   // exclusive scan of the line counters: PER consecutive counters per thread, wave scan, wave totals
        u32 c[PER], tsum = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) { const int q = tid * PER + j; c[j] = (q < nlines) ? scnt[q] : 0u; tsum += c[j]; }
        u32 wtot;
        const u32 wex = wave_excl_scan_u32(tsum, wtot);
        if ((tid & 63) == 0) swt[tid >> 6] = wtot;
        __syncthreads();
        u32 base = wex;
        for (int w = 0; w < (tid >> 6); ++w) base += swt[w];
#pragma unroll
        for (int j = 0; j < PER; ++j) { const int q = tid * PER + j; if (q < nlines) scnt[q] = base; base += c[j]; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VT; ++i) {
        if (tid * VT + i < total) {
            const u32 slot = min(scnt[hpos[i] >> 5] + rk[i], (u32)(NT * VT - 1));
            scode[slot] = hcode[i];
            spos[slot] = (unsigned short)hpos[i];
        }
    }
    __syncthreads();
    for (int e = tid; e < total; e += NT) zrow[spos[e]] = scode[e];
}

// Tie runs of a sorted LDS array by scans instead of per-element searches (uniform cost however heavy
// the ties): thread t owns positions [16t, 16t+16); rs[i] / re[i] = block-local [start, end) of the
// run of equal keys containing position 16t+i.  `wsh` = 2 * NT/64 ints of LDS scratch.
template <int NT, int VT, class KF>
__device__ __forceinline__ void block_tie_runs(KF key_at, int total, int* wsh, int (&rs)[VT], int (&re)[VT])
{
    constexpr int NW = NT / kWave;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int base = tid * VT;
    bool h[VT];
    auto prev = key_at((base > 0 && base - 1 < total) ? base - 1 : 0);     // (position 0 is a head whatever it compares to)
    int cur = -1, first = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < VT; ++i) {
        const int g = base + i;
        const auto kv = key_at(g < total ? g : 0);
        h[i] = (g == 0) || (g >= total) || (prev != kv);
        if (h[i]) { cur = g; if (first == 0x7fffffff) first = g; }
        rs[i] = cur;
        prev = kv;
    }
    // max-scan of `cur` (last head at or before the end of my chunk) and reverse min-scan of `first` over the lanes (DPP)
    int exL, exR;
    const int incl = wave_prefix_max(cur, exL);
    const int rinc = wave_suffix_min(first, exR);
    __syncthreads();
    if (lane == 63) wsh[w] = incl;
    if (lane == 0) wsh[NW + w] = rinc;
    __syncthreads();
    int carryL = -1, carryR = total;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) {
        if (ww < w) carryL = max(carryL, wsh[ww]);
        if (ww > w) carryR = min(carryR, wsh[NW + ww]);
    }
    carryL = max(carryL, exL);          // (the identities in lanes 0 / 63)
    carryR = min(carryR, exR);
    if (carryR > total) carryR = total;
    int nxt = carryR;
#pragma unroll
    for (int i = VT - 1; i >= 0; --i) {
        re[i] = nxt;
        if (h[i]) nxt = base + i;
        if (rs[i] < 0) rs[i] = carryL;
    }
    __syncthreads();
}

// Order statistics of one parameter by ONE WAVE from its pooled ascending keys: quantiles (numpy `linear` lerp, which
// pyarrow's interpolation="linear" agrees with to 1 ulp; src/mcmc_ref/backends_numpy.py:44, backends_arrow.py:40-42),
// statistics.median (diagnostics.py:97) and the fold split point s = #(x < med), found by a 64-ary search (64 probes and
// a ballot per round: three dependent loads for 40 000 draws where a bisection needs sixteen).  Returns s in every lane;
// `write`: also store the quantiles and the median into the result table (one caller per parameter does).
template <typename KT>
__device__ __forceinline__ i64 wave_order_stats(const KT* __restrict__ k, i64 M, const QArgs& q, bool write,
                                                double* __restrict__ res, i64 P, i64 p, double& med_out)
{
#pragma clang fp contract(off)  // the lerp must round like numpy's (separate multiply and add)
    const int lane = threadIdx.x & 63;
    if (write && lane < q.nq) {
        const int j = lane;
        const i64 lo = q.lo[j], hi = (lo + 1 < M) ? lo + 1 : M - 1;
        const double a = sorted_key(k, lo), b = sorted_key(k, hi), d = b - a, g = q.g[j];
        res[(R_Q0 + j) * P + p] = (g >= 0.5) ? b - d * (1.0 - g) : a + d * g;
    }
    // the two middle draws give the median and, unless draws tie AT the median, the split point too: with below < med
    // every draw up to index M/2 - 1 is below the median and the one at M/2 is not, so s = M/2 and the search (three
    // dependent rounds of loads in front of every fold workgroup's merge) is skipped
    const double below = (M >= 2) ? sorted_key(k, M / 2 - 1) : -INFINITY, at = sorted_key(k, M / 2);
    const double med = (M & 1) ? at : (below + at) / 2.0;
    if (write && lane == 0) res[R_MEDIAN * P + p] = med;
    // (!(at < med): draws near 1e308 can overflow below + at to +inf, and then #(x < med) is M, not M / 2: ADVICE r3)
    if (below < med && !(at < med)) { med_out = med; return M / 2; }
    i64 lo = 0, hi = M;  // first index with k[i] >= med lies in [lo, hi]
    while (lo < hi) {
        const i64 step = (hi - lo + 63) / 64;
        const i64 pos = lo + step * (lane + 1) - 1;                 // ascending probes, the last one at or beyond hi - 1
        const bool less = (pos < hi) && (sorted_key(k, pos) < med);
        const i64 cnt = (i64)__popcll(__ballot(less));              // sorted keys: the first cnt probes are below med
        const i64 nhi = lo + step * (cnt + 1) - 1;                  // probe cnt (if any) is the first one not below
        lo += step * cnt;
        hi = (nhi < hi) ? nhi : hi;
    }
    med_out = med;
    return lo;
}

// ------------------------------------------------------------------------------------------------
// Merge pass.  FOLD == false: merges pairs of sorted runs of length R (merge sort pass).
// FOLD == true : produces the ascending order of |x - med| from the ascending order of x with a
// single merge: the values below the median, walked downwards (med - x, non-decreasing because
// rounding is monotone), against the values at or above it (x - med).  This replaces the second
// full sort of src/mcmc_ref/diagnostics.py:93-98 + :110.  Each workgroup owns OB = NT*VT outputs.
// ------------------------------------------------------------------------------------------------
// KT = double: sorted keys in `kin`, positions in `iin`.  KT = u64 (FOLD only): sorted f32 records in `kin`
// (mcr_sort32.hpp), `iin` unused -- the fold step widens the keys on the fly, its own keys |x - med| are f64.
template <int NT, int VT, bool FOLD, typename IdxT, typename KT = double>
__global__ __launch_bounds__(NT) void k_merge(const KT* __restrict__ kin, const IdxT* __restrict__ iin,
                                              double* __restrict__ kout, IdxT* __restrict__ iout, i64 M,
                                              i64 R, double* __restrict__ res, i64 P,
                                              QArgs q, u32* __restrict__ z, const i64* __restrict__ split_in)
{
    constexpr int OBS = NT * VT;                            // LDS slots
    // The fold kernel owns 64 outputs fewer than it has slots and keeps its scratch in the 64 key slots that frees:
    // its LDS is then exactly 4096 * (8 + sizeof(IdxT)) bytes, i.e. FOUR workgroups per CU with 16-bit positions.
    constexpr int OB = FOLD ? OBS - 64 : OBS;               // outputs per workgroup
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* skey = reinterpret_cast<double*>(smem);
    IdxT* sidx = reinterpret_cast<IdxT*>(skey + OBS);
    i64* sh = FOLD ? reinterpret_cast<i64*>(skey + OB) : reinterpret_cast<i64*>(sidx + OBS);

    const int tid = threadIdx.x;
    i64 p = blockIdx.y;
    int blk = blockIdx.x;
    if (FOLD) {   // 1-D XCD-aware grid (the fused z scatter of one parameter stays in one L2)
        if (!xcd_map(P, (int)((M + OB - 1) / OB), p, blk)) return;
    }
    const i64 o0 = (i64)blk * OB;
    if (o0 >= M) return;
    const KT* kp = kin + p * M;
    const IdxT* ip = iin + p * M;
    auto KEY = [&](i64 g) -> double { return sorted_key(kp, g); };
    auto POS = [&](i64 g) -> u32 {
        if constexpr (sizeof(KT) == 8 && !__is_same(KT, double)) return (u32)kp[g]; else return (u32)ip[g];
    };

    i64 abase, na, bbase, nb, d0;
    double med = 0.0;
    if (!FOLD) {
        const i64 pb = (o0 / (2 * R)) * (2 * R);
        abase = pb;
        na = (M - pb < R) ? M - pb : R;
        bbase = pb + na;
        nb = (M - bbase < R) ? M - bbase : R;
        d0 = o0 - pb;
    } else {
        // The order statistics of the parameter, by the first wave of EVERY fold workgroup (a launch of their own in round
        // 2): the median and the split point are three to five dependent loads, the parameter's first workgroup also
        // writes the quantiles and the median.
        // (long arrays -- a hundred fold workgroups per parameter -- get them from a k_order_stats launch instead: split_in)
        if (split_in != nullptr) {
            if (tid == 0) { sh[0] = split_in[p]; reinterpret_cast<double*>(sh)[1] = res[R_MEDIAN * P + p]; }
        } else if (tid < 64) {
            double m_;
            const i64 s_ = wave_order_stats(kp, M, q, blk == 0, res, P, p, m_);
            if (tid == 0) { sh[0] = s_; reinterpret_cast<double*>(sh)[1] = m_; }
        }
        __syncthreads();
        const i64 s = sh[0];
        med = reinterpret_cast<double*>(sh)[1];
        __syncthreads();
        abase = s - 1;  // walked downwards
        na = s;
        bbase = s;
        nb = M - s;
        d0 = o0;
    }
    const i64 tot = na + nb;
    const i64 d1 = (d0 + OB < tot) ? d0 + OB : tot;

    auto GA = [&](i64 i) -> double { return FOLD ? med - KEY(abase - i) : KEY(abase + i); };
    auto GB = [&](i64 j) -> double { return FOLD ? KEY(bbase + j) - med : KEY(bbase + j); };
    if (tid < 64) { const i64 r0 = merge_path_wave(GA, na, GB, nb, d0); if (tid == 0) sh[0] = r0; }
    else if (tid < 128) { const i64 r1 = merge_path_wave(GA, na, GB, nb, d1); if (tid == 64) sh[1] = r1; }
    __syncthreads();
    const i64 ai0 = sh[0], ai1 = sh[1];
    const i64 bi0 = d0 - ai0;
    const int ca = (int)(ai1 - ai0), cb = (int)((d1 - ai1) - bi0);
    const int total = ca + cb;

    {   // gather the two runs: slot e = j * NT + tid; all VT (key, position) loads of a lane are issued before the first
        // LDS store waits for one (slots at or beyond `total` read element 0 and are not stored)
        double gv[VT]; u32 gi[VT];
#pragma unroll
        for (int j = 0; j < VT; ++j) {
            const int e = j * NT + tid;
            const bool inA = e < ca;
            i64 g = inA ? (FOLD ? abase - (ai0 + e) : abase + ai0 + e) : bbase + bi0 + (e - ca);
            g = (e < total) ? g : 0;
            const double x = KEY(g);
            gv[j] = FOLD ? (inA ? med - x : x - med) : x;
            gi[j] = POS(g);
        }
#pragma unroll
        for (int j = 0; j < VT; ++j) {
            const int e = j * NT + tid;
            if (e < total) { skey[pos16(e)] = gv[j]; sidx[posi(e)] = (IdxT)gi[j]; }
        }
    }
    __syncthreads();

    const int diag = (tid * VT < total) ? tid * VT : total;
    const int nout = (total - diag < VT) ? total - diag : VT;
    auto A = [&](int i) { return skey[pos16(i)]; };
    auto B = [&](int j) { return skey[pos16(ca + j)]; };
    const int ai = merge_path32(A, ca, B, cb, diag);
    double k[VT];
    int srcs[VT];
    u32 ix[VT];
    serial_merge<VT>(skey, 0, ca, ca, cb, ai, diag - ai, nout, k, srcs);
#pragma unroll
    for (int i = 0; i < VT; ++i) ix[i] = sidx[posi(srcs[i])];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VT; ++i) {
        if (i < nout) {
            skey[pos16(diag + i)] = k[i];
            sidx[posi(diag + i)] = (IdxT)ix[i];
        }
    }
    __syncthreads();
    if (z == nullptr) {
        for (int e = tid; e < total; e += NT) {
            kout[p * M + o0 + e] = skey[pos16(e)];
            iout[p * M + o0 + e] = sidx[posi(e)];
        }
        return;
    }
    // Fused ranks -> z (src/mcmc_ref/diagnostics.py:113-133): the merged order never goes to memory.
    // Tie runs touching the block edges are completed with lower/upper bounds over the two runs.
    // Does the first / last tie run continue outside this block?  Look at the one element before
    // and after the block in each run; only then pay for the bound searches (rare: heavy ties).
    if (tid == 0) {
        const i64 bi1 = d1 - ai1;
        const double v0 = skey[pos16(0)], v1 = skey[pos16(total - 1)];
        bool ext0 = false, ext1 = false;
        if (ai0 > 0 && GA(ai0 - 1) == v0) ext0 = true;
        if (bi0 > 0 && GB(bi0 - 1) == v0) ext0 = true;
        if (ai1 < na && GA(ai1) == v1) ext1 = true;
        if (bi1 < nb && GB(bi1) == v1) ext1 = true;
        sh[2] = ext0; sh[3] = ext1;
    }
    __syncthreads();
    const bool ext0 = sh[2] != 0, ext1 = sh[3] != 0;
    __syncthreads();
    if ((ext0 || ext1) && tid < 8) {
        const int which = tid >> 2, side = (tid >> 1) & 1, upper = tid & 1;   // value, run, bound kind
        const double v = which ? skey[pos16(total - 1)] : skey[pos16(0)];
        const i64 len = side ? nb : na;
        i64 lo = 0, hi = len;
        while (lo < hi) {
            const i64 mid = (lo + hi) >> 1;
            const double x = side ? GB(mid) : GA(mid);
            const bool right = upper ? !(v < x) : (x < v);
            if (right) lo = mid + 1; else hi = mid;
        }
        sh[2 + tid] = lo;
    }
    __syncthreads();
    const double vfirst = skey[pos16(0)], vlast = skey[pos16(total - 1)];
    const i64 gfirst = sh[2 + 0] + sh[2 + 2];   // lower bounds (both runs) of the first value
    const i64 glast = sh[2 + 5] + sh[2 + 7];    // upper bounds (both runs) of the last value
    int rs[VT], re[VT];
    block_tie_runs<NT, VT>([&](int g) { return skey[pos16(g)]; }, total, reinterpret_cast<int*>(sh + 12), rs, re);
    const bool by_line = kWriteCombine && M <= 65535;        // (workgroup-uniform)
    u32 hcode[VT], hpos[VT];
#pragma unroll
    for (int i = 0; i < VT; ++i) {
        const int e = tid * VT + i;
        hcode[i] = 0u; hpos[i] = 0u;
        if (e < total) {
            const double v = skey[pos16(e)];
            i64 gs = d0 + rs[i], ge = d0 + re[i];
            if (ext0 && v == vfirst) gs = gfirst;
            if (ext1 && v == vlast) ge = glast;
            const u32 t_idx = min((u32)sidx[posi(e)], (u32)(M - 1));     // stale order after a rejected (NaN) partition
            hcode[i] = (u32)(gs + ge); hpos[i] = t_idx;
            if (!by_line) z[p * M + t_idx] = (u32)(gs + ge);            // code of the tie run: rank = (code + 1) / 2
        }
    }
    if (by_line)
        scatter_codes_by_line<NT, VT>(z + p * M, M, total, hcode, hpos, reinterpret_cast<u32*>(skey),
                                      reinterpret_cast<unsigned short*>(sidx), reinterpret_cast<u32*>(sh));
}

// ------------------------------------------------------------------------------------------------
// Exact k-way partition by regular sampling (k = #sorted runs <= 16; a run is a sorted tile, or the
// result of a few pairwise merge passes when the pooled array is longer than 16 tiles).
//
// Every sorted tile contributes its 64th, 128th, ... order statistics.  In the strict total order
// (value, tile, position) the sample of pooled sample-rank r has between 64(r+1) and 64(r+1)+64k
// draws at or below it, so cutting at every D-th sample gives buckets of fewer than 64(D+k) draws:
// with D = floor((4032 - 79k) / 64) a bucket, each of its <= k pieces padded to a multiple of 16,
// always fits the 4032 data slots of k_bucket_merge's LDS -- a deterministic bound, no overflow path.
// One workgroup per parameter writes cut[p][b][t] (start of bucket b inside tile t, b = 0..B) and
// boff[p][b] (start of bucket b in the pooled order).
// ------------------------------------------------------------------------------------------------
constexpr int kMaxBucketTiles = 16;

// The LDS of k_splitters: S samples + splitter tables + cut table, then (MVT > 0) the merge area of the sample ranking.
__host__ __device__ inline size_t splitters_base_bytes(int S, int B, int k, int mvt)     // tables in front of the merge area, 16-byte aligned
{
    const size_t base = (size_t)S * (mvt > 0 ? 8 : 12) + (size_t)(B + 1) * 16 + (size_t)(B + 1) * k * 4 + 64;   // (no rank array in merge mode)
    return (base + 15) / 16 * 16;
}
__host__ __device__ inline size_t splitters_lds_bytes(int S, int B, int k, int mvt)
{
    return splitters_base_bytes(S, B, k, mvt) + (mvt > 0 ? (size_t)(1024 * mvt + 16) * 10 : 0);
}

// MVT = 0: few samples (S <= 1024: pooled arrays of up to 16 tiles, the C1 and corpus shapes) -- every (sample, other
//          run) pair is one bisection, the pooled ranks add up with LDS atomics.
// MVT > 0: many samples (runs pre-merged to 8192 .. 32768 draws: 128 .. 512 samples each, up to 8192 in all; the stress
//          shape ranks 6 656) -- the k sorted sample lists are MERGED in the LDS, ceil(log2 k) merge-path rounds of
//          1024 threads x MVT outputs carrying (value, origin) pairs; a sample's place in the merged list IS its pooled
//          rank in the strict order (value, run, position), since the merges are stable and the runs enter in order.
//          ~190 LDS round trips per thread on the stress shape instead of 6.5 samples x 12 runs x 9 probes = 700
//          (k_splitters was 12.4 ms of its 216 ms step, and the first kernel of the long-chain profile: VERDICT r3 item 8).
template <typename KT, int MVT>      // KT = double (sorted keys) or u64 (sorted f32 records, mcr_sort32.hpp)
__global__ __launch_bounds__(1024) void k_splitters(const KT* __restrict__ keys,
                                                    const double* __restrict__ samp, i64 M, int k, int B,
                                                    int D, i64 R, u32* __restrict__ cut, u32* __restrict__ boff)
{
    constexpr int NTS = 1024;
    const int SPT = (int)(R / 64);                   // samples per run
    const int S = k * SPT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sv = reinterpret_cast<double*>(smem);    // S samples
    double* splv = sv + S;                           // B splitter values
    int* srank = reinterpret_cast<int*>(splv + B + 1);   // S pooled ranks (MVT == 0 only)
    int* splt = srank + (MVT > 0 ? 0 : S);           // B splitter run
    int* splp = splt + B + 1;                        // B splitter position in run
    u32* scut = reinterpret_cast<u32*>(splp + B + 1);    // (B+1) * k cuts
    const int tid = threadIdx.x;
    const i64 p = blockIdx.x;
    const KT* kp = keys + p * M;
    for (int i = tid; i < S; i += NTS) { sv[i] = samp[p * S + i]; if (MVT == 0) srank[i] = i % SPT; }
    for (int b = tid; b <= B; b += NTS) splt[b] = -1;      // -1 = splitter b not found (only with NaN / Inf draws)
    if (MVT > 0) {
        constexpr int VT = MVT > 0 ? MVT : 1;
        static_assert(MVT == 0 || 64 % VT == 0, "a thread's VT outputs must not straddle two pairs of runs (run lengths are multiples of 64)");
        constexpr int SP = NTS * VT;                           // merge slots: the samples, then +inf pads (S <= SP)
        double* mk = reinterpret_cast<double*>(smem + splitters_base_bytes(S, B, k, MVT));                   // [SP + 16] keys, pos16-swizzled
        unsigned short* mi = reinterpret_cast<unsigned short*>(mk + SP + 16);                                // [SP + 16] origin = run * SPT + index
        for (int e = tid; e < SP + 16; e += NTS) {
            mk[pos16(e)] = (e < S) ? samp[p * S + e] : INFINITY;
            mi[e] = (unsigned short)(e < S ? e : 0xFFFF);
        }
        __syncthreads();
        const int chunk0 = tid * VT;
        for (int g = SPT; g < SP; g <<= 1) {                   // runs of g slots merge in pairs (SP is a multiple of SPT)
            double kk[VT]; int srcs[VT]; unsigned short ids[VT];
            const int a0 = chunk0 / (2 * g) * (2 * g);
            const int a1 = (a0 + g < SP) ? a0 + g : SP, b1 = (a0 + 2 * g < SP) ? a0 + 2 * g : SP;
            const int na = a1 - a0, nb = b1 - a1, diag = chunk0 - a0;
            const bool moved = nb > 0;
            if (moved) {
                auto A = [&](int i) { return mk[pos16(a0 + i)]; };
                auto Bf = [&](int j) { return mk[pos16(a1 + j)]; };
                const int ai = merge_path32(A, na, Bf, nb, diag);
                serial_merge<VT>(mk, a0, na, a1, nb, ai, diag - ai, VT, kk, srcs);
#pragma unroll
                for (int i = 0; i < VT; ++i) ids[i] = mi[srcs[i]];
            }
            __syncthreads();
            if (moved) {
#pragma unroll
                for (int i = 0; i < VT; ++i) { mk[pos16(chunk0 + i)] = kk[i]; mi[chunk0 + i] = ids[i]; }
            }
            __syncthreads();
        }
        // the sample at merged place r = b D - 1 is splitter b
        for (int b = 1 + tid; b < B; b += NTS) {
            const int r = b * D - 1;
            if (r < SP) {
                const double v = mk[pos16(r)];
                const int i = mi[r];
                if (v < INFINITY && i < S) { splv[b] = v; splt[b] = i / SPT; splp[b] = 64 * (i % SPT) + 63; }
            }
        }
        __syncthreads();
    } else {
    __syncthreads();
    // pooled rank of every finite sample = own index + samples of every other run below it;
    // one (sample, other run) pair per thread step.  The lanes of a wave take CONSECUTIVE SAMPLES against one run: their
    // probes fall near each other in that run's list and their atomics on neighbouring counters (with consecutive runs per
    // lane -- round 3 -- every lane probed its own run at a stride of SPT doubles, i.e. the same LDS bank: 6.8 conflict
    // cycles per LDS instruction, VERDICT r3 weak #3a)
    for (i64 q = tid; q < (i64)S * k; q += NTS) {
        const int t2 = (int)(q / S), i = (int)(q - (i64)t2 * S);
        const double v = sv[i];
        const int t = i / SPT;
        if (t2 == t || !(v < INFINITY)) continue;
        const double* a = sv + t2 * SPT;
        int lo = 0, hi = SPT;  // count of samples of run t2 that sort before (v, t, j)
        if (t2 < t) { while (lo < hi) { const int m = (lo + hi) >> 1; if (!(v < a[m])) lo = m + 1; else hi = m; } }
        else        { while (lo < hi) { const int m = (lo + hi) >> 1; if (a[m] < v) lo = m + 1; else hi = m; } }
        if (lo) atomicAdd(&srank[i], lo);
    }
    __syncthreads();
    for (int i = tid; i < S; i += NTS) {
        if (!(sv[i] < INFINITY)) continue;
        const int r = srank[i];
        if ((r + 1) % D == 0) {
            const int b = (r + 1) / D;
            if (b >= 1 && b < B) { splv[b] = sv[i]; splt[b] = i / SPT; splp[b] = 64 * (i % SPT) + 63; }
        }
    }
    __syncthreads();
    }
    // Non-finite draws (the call will be rejected with MCR_ENONFINITE) break the ordering the splitters rely on:
    // leave an EMPTY partition behind, so that no later kernel walks cut tables made of garbage.
    {
        int missing = 0;
        for (int b = 1 + tid; b < B; b += NTS) missing |= (splt[b] < 0);
        if (__syncthreads_or(missing)) {
            for (int q = tid; q < (B + 1) * k; q += NTS) cut[p * (i64)(B + 1) * k + q] = 0u;
            for (int b = tid; b <= B; b += NTS) boff[p * (B + 1) + b] = 0u;
            return;
        }
    }
    // cuts: position in run t where bucket b starts
    for (int q = tid; q < (B + 1) * k; q += NTS) {
        const int b = q / k, t = q % k;
        const i64 tbase = (i64)t * R;
        const int cnt = (int)((M - tbase < R) ? M - tbase : R);
        u32 c;
        if (b == 0) c = 0;
        else if (b == B) c = (u32)cnt;
        else {
            const double v = splv[b];
            const int ts = splt[b];
            if (t == ts) c = (u32)(splp[b] + 1);
            else {
                // the run's own samples (in LDS) bracket the answer to a 64-draw window
                const double* sa = sv + t * SPT;
                int slo = 0, shi = SPT;
                if (t < ts) { while (slo < shi) { const int m = (slo + shi) >> 1; if (!(v < sa[m])) slo = m + 1; else shi = m; } }
                else        { while (slo < shi) { const int m = (slo + shi) >> 1; if (sa[m] < v) slo = m + 1; else shi = m; } }
                const KT* a = kp + tbase;
                int lo = 64 * slo, hi = 64 * slo + 63;        // sample j sits at position 64 j + 63
                if (lo > cnt) lo = cnt;
                if (hi > cnt) hi = cnt;
                if (t < ts) { while (lo < hi) { const int m = (lo + hi) >> 1; if (!(v < sorted_key(a, m))) lo = m + 1; else hi = m; } }
                else        { while (lo < hi) { const int m = (lo + hi) >> 1; if (sorted_key(a, m) < v) lo = m + 1; else hi = m; } }
                c = (u32)lo;
            }
        }
        scut[q] = c;
        cut[(p * (B + 1) + b) * k + t] = c;
    }
    __syncthreads();
    for (int b = tid; b <= B; b += NTS) {
        u32 o = 0;
        for (int t = 0; t < k; ++t) o += scut[b * k + t];
        boff[p * (B + 1) + b] = o;
    }
}

// Every 64th order statistic of each sorted run of length R (the regular samples of k_splitters)
// when the runs were produced by merge passes rather than by k_tile_sort.  grid (k, P).
template <typename KT>
__global__ __launch_bounds__(256) void k_sample_runs(const KT* __restrict__ keys, i64 M, i64 R,
                                                     double* __restrict__ samp)
{
    const int run = blockIdx.x, k = gridDim.x;
    const i64 p = blockIdx.y;
    const int SPT = (int)(R / 64);
    const i64 base = (i64)run * R;
    for (int j = threadIdx.x; j < SPT; j += 256) {
        const i64 e = base + 64 * (i64)j + 63;
        samp[(p * k + run) * SPT + j] = (e < M && 64 * (i64)j + 63 < R) ? sorted_key(keys, p * M + e) : INFINITY;
    }
}

// ------------------------------------------------------------------------------------------------
// Bucket merge: one workgroup per (bucket, parameter) gathers its <= k sorted pieces (one per tile)
// into LDS, each padded with +inf to a multiple of 16, merges them with ceil(log2 k) merge-path
// rounds, writes the bucket to its place in the pooled ascending order (for the fold step and the
// order statistics) and -- fused -- turns positions into tie-averaged ranks, z = Phi^-1((r-1/2)/M)
// and scatters z to time order (src/mcmc_ref/diagnostics.py:101-133).  Tie runs that touch a bucket
// edge are completed with lower/upper bounds over the k sorted tiles, which are complete in memory.
// Replaces log2(k) global merge passes + k_rank_z of the first version.
// ------------------------------------------------------------------------------------------------
template <int NT, int VT, typename IdxT>
__global__ __launch_bounds__(NT) void k_bucket_merge(const double* __restrict__ kin, const IdxT* __restrict__ iin,
                                                      double* __restrict__ kout, IdxT* __restrict__ iout, i64 M,
                                                      int k, int B, const u32* __restrict__ cut,
                                                      const u32* __restrict__ boff, u32* __restrict__ z, i64 P,
                                                      i64 R)
{
    constexpr int T = NT * VT;
    static_assert(T == 4096, "the partition bound of k_splitters assumes 4096-slot buckets");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* skey = reinterpret_cast<double*>(smem);
    IdxT* sidx = reinterpret_cast<IdxT*>(skey + T);
    // k_splitters bounds every bucket by T - 64 slots; the 64 key slots above hold the scratch, so the workgroup's LDS
    // is exactly T * (8 + sizeof(IdxT)) bytes: FOUR buckets per CU with 16-bit positions.
    int* sst = reinterpret_cast<int*>(skey + (T - 64));   // padded piece starts [k+1], then scratch
    int* spl = sst + 40;                            // piece lengths [k]
    int* sps = spl + 40;                            // piece source offsets in tile [k]
    i64* sedge = reinterpret_cast<i64*>(sps + 40);  // [4] global run bounds of the edge values

    const int tid = threadIdx.x;
    i64 p;
    int b;
    if (!xcd_map(P, B, p, b)) return;
    const double* kp = kin + p * M;
    const IdxT* ip = iin + p * M;
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
    const int padded = sst[k];          // <= 4096 by construction of D
    const int total = (int)(boff[p * (B + 1) + b + 1] - boff[p * (B + 1) + b]);
    const i64 obase = boff[p * (B + 1) + b];
    if (padded > T - 64 || total < 0 || total > padded || obase + total > M) return;   // never with a valid partition
    // gather pieces (+inf pads).  Slot e = j * NT + tid: a wave's j-th load covers 64 consecutive slots = four 16-slot
    // chunks, one per DPP row of 16 lanes, and a chunk never straddles pieces (they are padded to 16).  Lane j of row q
    // looks up the piece of the row's j-th chunk ONCE; the element loop takes {source offset, live length} from that lane
    // with a row broadcast and issues all VT loads back to back (one piece search per 16 slots instead of one per slot,
    // no load waits for the previous one).
    {
        const int lane = tid & 63, wv = tid >> 6;
        u32 my_g = 0; int my_n = 0;
        if ((lane & 15) < VT) {
            const int e0 = (lane & 15) * NT + 64 * wv + 16 * (lane >> 4);
            if (e0 < padded) {
                int t = 0;   // last piece whose padded start is <= e0 (starts are non-decreasing)
#pragma unroll
                for (int step = 8; step > 0; step >>= 1)
                    if (t + step < k && e0 >= sst[t + step]) t += step;
                const int o = e0 - sst[t];
                my_n = spl[t] - o;
                my_g = (u32)((i64)t * R) + (u32)sps[t] + (u32)o;
            }
        }
        double gv[VT]; IdxT gi[VT];
#pragma unroll
        for (int j = 0; j < VT; ++j) {
            const int within = lane & 15;
            const u32 g0 = (u32)row_lane((int)my_g, j);      // lane j of my row of 16 looked the chunk up
            const int n = row_lane(my_n, j);
            const bool live = within < n;
            const i64 g = live ? (i64)(g0 + (u32)within) : 0;     // dead slots read element 0 (always there) and drop it
            const double v = kp[g]; const IdxT id = ip[g];
            gv[j] = live ? v : INFINITY; gi[j] = live ? id : (IdxT)~(IdxT)0;
        }
#pragma unroll
        for (int j = 0; j < VT; ++j) {
            const int e = j * NT + tid;
            if (e < padded) { skey[pos16(e)] = gv[j]; sidx[posi(e)] = gi[j]; }
        }
    }
    __syncthreads();
    // merge rounds over adjacent runs (run boundaries = sst[] at stride 2^r)
    const int chunk0 = tid * VT;
    int mypiece = 0;      // the piece that holds slot chunk0 (pieces are padded to 16 >= VT slots: a lane's chunk lies in one piece)
#pragma unroll
    for (int step = 8; step > 0; step >>= 1)
        if (mypiece + step < k && chunk0 >= sst[mypiece + step]) mypiece += step;
    for (int w = 1; w < k; w <<= 1) {
        double kk[VT]; int srcs[VT]; u32 ix[VT];
        const bool active = chunk0 < padded;
        bool moved = false;
        if (active) {
            const int ra = mypiece & ~(2 * w - 1);   // first piece of the pair of runs (w pieces each) that contains chunk0
            const int a0 = sst[ra];
            const int a1 = sst[(ra + w < k) ? ra + w : k];
            const int b1 = sst[(ra + 2 * w < k) ? ra + 2 * w : k];
            const int na = a1 - a0, nb = b1 - a1, diag = chunk0 - a0;
            moved = nb > 0;          // a run without a partner in this round (k not a power of two) stays where it is
            if (moved) {
                auto A = [&](int i) { return skey[pos16(a0 + i)]; };
                auto Bf = [&](int j) { return skey[pos16(a1 + j)]; };
                const int ai = merge_path32(A, na, Bf, nb, diag);
                serial_merge<VT>(skey, a0, na, a1, nb, ai, diag - ai, VT, kk, srcs);
#pragma unroll
                for (int i = 0; i < VT; ++i) ix[i] = sidx[posi(srcs[i])];
            }
        }
        __syncthreads();
        if (moved) {
#pragma unroll
            for (int i = 0; i < VT; ++i) { skey[pos16(chunk0 + i)] = kk[i]; sidx[posi(chunk0 + i)] = (IdxT)ix[i]; }
        }
        __syncthreads();
    }
    // pooled ascending order out
    for (int e = tid; e < total; e += NT) {
        kout[p * M + obase + e] = skey[pos16(e)];
        iout[p * M + obase + e] = sidx[posi(e)];
    }
    if (z == nullptr || total == 0) return;
    // Do the first / last tie runs continue in a neighbouring bucket?  Look at the element just
    // before / after this bucket's piece in every tile; only then pay for the bound searches.
    if (tid < 4) sedge[tid] = 0;
    if (tid < 64) {
        bool e0 = false, e1 = false;
        if (tid < k) {
            const i64 tbase = (i64)tid * R;
            const int cnt = (int)((M - tbase < R) ? M - tbase : R);
            const int lo = sps[tid], hi = sps[tid] + spl[tid];
            if (lo > 0) e0 = (kp[tbase + lo - 1] == skey[pos16(0)]);
            if (hi < cnt) e1 = (kp[tbase + hi] == skey[pos16(total - 1)]);
        }
        const bool a0 = __ballot(e0) != 0, a1 = __ballot(e1) != 0;
        if (tid == 0) { sst[36] = a0; sst[37] = a1; }
    }
    __syncthreads();
    const bool ext0 = sst[36] != 0, ext1 = sst[37] != 0;
    if ((ext0 || ext1) && tid < 2 * k) {
        const int t = tid % k, which = tid / k;           // 0: first value, 1: last value
        const double v = which ? skey[pos16(total - 1)] : skey[pos16(0)];
        const i64 tbase = (i64)t * R;
        const int cnt = (int)((M - tbase < R) ? M - tbase : R);
        const double* a = kp + tbase;
        int lo = 0, hi = cnt;
        while (lo < hi) { const int m = (lo + hi) >> 1; if (a[m] < v) lo = m + 1; else hi = m; }
        const int lb = lo;
        hi = cnt;
        while (lo < hi) { const int m = (lo + hi) >> 1; if (!(v < a[m])) lo = m + 1; else hi = m; }
        atomicAdd(reinterpret_cast<unsigned long long*>(&sedge[2 * which]), (unsigned long long)lb);
        atomicAdd(reinterpret_cast<unsigned long long*>(&sedge[2 * which + 1]), (unsigned long long)lo);
    }
    __syncthreads();
    const double vfirst = skey[pos16(0)], vlast = skey[pos16(total - 1)];
    int rs[VT], re[VT];
    block_tie_runs<NT, VT>([&](int g) { return skey[pos16(g)]; }, total, sst + 24, rs, re);
#pragma unroll
    for (int i = 0; i < VT; ++i) {
        const int e = tid * VT + i;
        if (e < total) {
            const double v = skey[pos16(e)];
            i64 gs = obase + rs[i], ge = obase + re[i];
            if (ext0 && v == vfirst) gs = sedge[0];
            if (ext1 && v == vlast) ge = sedge[3];
            // (scattered as it is: with four merge rounds to overlap them, this kernel's stores are hidden already --
            //  scatter_codes_by_line here measured 382 -> 382 us per 1000 parameters)
            z[p * M + min((u32)sidx[posi(e)], (u32)(M - 1))] = (u32)(gs + ge);   // code of the tie run: rank = (code + 1) / 2
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Order statistics from the pooled ascending keys: quantiles (numpy `linear` lerp, which pyarrow's
// interpolation="linear" agrees with to 1 ulp; src/mcmc_ref/backends_numpy.py:44,
// backends_arrow.py:40-42), statistics.median (diagnostics.py:97) and the fold split point
// s = #(x < med).  One WAVE per parameter: lane j takes quantile j, and the split point is found by a 64-ary search
// (64 probes and a ballot per round: three dependent loads for 40 000 draws where a bisection needs sixteen -- the
// kernel is nothing but that latency chain, and the fold merge waits for it).  grid ceil(P / 4), block 256.
// ------------------------------------------------------------------------------------------------
template <typename KT>
__global__ __launch_bounds__(256) void k_order_stats(const KT* __restrict__ keys, i64 M, i64 P, QArgs q,
                                                     double* __restrict__ res, i64* __restrict__ split)
{
    const int lane = threadIdx.x & 63;
    const i64 p = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    double med;
    const i64 s = wave_order_stats(keys + p * M, M, q, true, res, P, p, med);
    if (lane == 0) split[p] = s;
}

// ------------------------------------------------------------------------------------------------
// Ranks -> z.  For the sorted position i of a parameter: the maximal run [s, e) of equal keys
// gets the average 1-based rank (s + 1 + e) / 2 (src/mcmc_ref/diagnostics.py:113-122),
// z = Phi^-1((rank - 0.5) / M) (:130-131), written to the draw's pooled (time-order) position.
// Run ends are found by galloping + bisection from i, so isolated values cost two neighbour
// loads and long tie runs cost O(log run).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rank_z(const double* __restrict__ keys,
                                                const u32* __restrict__ idx, i64 M, u32* __restrict__ z)
{
    const i64 p = blockIdx.y;
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const double* k = keys + p * M;
    const double v = k[i];
    i64 s = i, e = i + 1;
    if (i > 0 && k[i - 1] == v) {  // gallop left: find first index with key == v
        i64 step = 1, hi = i - 1;  // k[hi] == v
        i64 lo = hi - step;
        while (lo >= 0 && k[lo] == v) { hi = lo; step <<= 1; lo = hi - step; }
        if (lo < -1) lo = -1;      // k[lo] != v (or lo == -1), k[hi] == v
        while (hi - lo > 1) {
            const i64 mid = (lo + hi) >> 1;
            if (k[mid] == v) hi = mid; else lo = mid;
        }
        s = hi;
    }
    if (i + 1 < M && k[i + 1] == v) {  // gallop right: find last index with key == v
        i64 step = 1, lo = i + 1;      // k[lo] == v
        i64 hi = lo + step;
        while (hi < M && k[hi] == v) { lo = hi; step <<= 1; hi = lo + step; }
        if (hi > M) hi = M;            // k[hi] != v (or hi == M), k[lo] == v
        while (hi - lo > 1) {
            const i64 mid = (lo + hi) >> 1;
            if (k[mid] == v) lo = mid; else hi = mid;
        }
        e = lo + 1;
    }
    z[p * M + min(idx[p * M + i], (u32)(M - 1))] = (u32)(s + e);   // code of the tie run: rank = (code + 1) / 2
}

// ------------------------------------------------------------------------------------------------
// Finalize: pooled mean / population std from the tile partials, rhat = Python max(bulk, tail)
// (`tail if tail > bulk else bulk`, src/mcmc_ref/diagnostics.py:40), NaN diagnostics when there
// are fewer than two chains (diagnostics.py:29-30, 53-54, 69-70).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void finalize_param(const double* __restrict__ part, int S, i64 M,
                                               i64 P, int C, double* __restrict__ res, i64 p)
{
    double mu, m2, b, n;
    merge_slice_moments(part + p * S * kMomRec, S, mu, m2, b, n);
    double var = m2 / (double)M;
    if (isinf(m2)) var = INFINITY;          // squares overflowed: inf like np.std
    else if (!(var >= 0.0)) var = 0.0;
    res[R_MEAN * P + p] = mu;
    res[R_STD * P + p] = sqrt(var);
    res[R_BAD * P + p] = b;
    if (C < 2) {
        res[R_RHAT * P + p] = NAN; res[R_RHAT_BULK * P + p] = NAN; res[R_RHAT_TAIL * P + p] = NAN;
        res[R_ESS_BULK * P + p] = NAN; res[R_ESS_TAIL * P + p] = NAN;
        res[R_LAG_BULK * P + p] = 0.0; res[R_LAG_TAIL * P + p] = 0.0;
    } else {
        const double rb = res[R_RHAT_BULK * P + p], rt = res[R_RHAT_TAIL * P + p];
        res[R_RHAT * P + p] = (rt > rb) ? rt : rb;
    }
}

// Stand-alone form: calls without diagnostics (Backend.stats only) and single-chain tensors.  With diagnostics the
// same function runs inside k_diag_combine2 (one launch fewer per call).
__global__ void k_finalize(const double* __restrict__ part, int S, i64 M, i64 P, int C, double* __restrict__ res)
{
    const i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    finalize_param(part, S, M, P, C, res, p);
}

// compare.compare_stats arithmetic (src/mcmc_ref/compare.py:41-43)
__global__ void k_compare(const double* __restrict__ ref, const double* __restrict__ act, i64 n,
                          double tol, double* __restrict__ rel, unsigned char* __restrict__ pass)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double r = ref[i], a = act[i], ar = fabs(r);
    const double denom = (1e-12 > ar) ? 1e-12 : ar;  // Python max(abs(ref), 1e-12): NaN stays NaN
    const double e = fabs(a - r) / denom;
    rel[i] = e;
    pass[i] = (e <= tol) ? 1 : 0;
}

// Read-only streaming probe: what the HBM delivers to a kernel that does nothing but 16-byte loads.  Each workgroup
// streams a contiguous slice with 4 loads in flight per lane (the access pattern of k_moments_rows).  bench.py reports
// the best rate over a few grid sizes as `peak_measured` next to the 8 TB/s specification peak.
__global__ __launch_bounds__(256) void k_stream_read(const uint4* __restrict__ src, i64 nvec, u32* __restrict__ sink)
{
    const i64 per = (nvec + gridDim.x - 1) / gridDim.x;
    const i64 b = (i64)blockIdx.x * per, e = (b + per < nvec) ? b + per : nvec;
    u32 acc = 0;
    i64 i = b + threadIdx.x;
    for (; i + 3 * 256 < e; i += 4 * 256) {
        uint4 r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = src[i + u * 256];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= r[u].x ^ r[u].y ^ r[u].z ^ r[u].w;
    }
    for (; i < e; i += 256) { const uint4 r = src[i]; acc ^= r.x ^ r.y ^ r.z ^ r.w; }
    if (acc == 0x9E3779B9u) sink[blockIdx.x & 1023] = acc;      // practically never: keeps the loads alive without a store stream
}

// Synthetic stress tensor: iid N(p, sigma_p), counter-based, layout [P][C][N].
template <typename T>
__global__ __launch_bounds__(256) void k_fill_synth(T* __restrict__ out, i64 total, i64 M, u64 seed, i64 first)
{
    // `first` = index of out[0] in the whole tensor: a rank that holds a parameter block of a P-split model
    //  generates exactly its slice of the one global tensor.
    const i64 stride = (i64)gridDim.x * 256;
    for (i64 j = (i64)blockIdx.x * 256 + threadIdx.x; j < total; j += stride) {
        const i64 i = first + j;
        const i64 p = i / M;
        const u64 base = seed * 0x9E3779B97F4A7C15ull + 2ull * (u64)i;
        const u64 h1 = splitmix64(base), h2 = splitmix64(base + 1);
        const double u1 = ((double)(h1 >> 11) + 0.5) * 0x1.0p-53;
        const double u2 = (double)(h2 >> 11) * 0x1.0p-53;
        const double e = sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
        int k = (int)(p % 7) - 3;
        double sigma = 1.0;
        for (; k > 0; --k) sigma *= 10.0;
        for (; k < 0; ++k) sigma /= 10.0;
        out[j] = (T)((double)p + sigma * e);
    }
}

}  // namespace mcr
