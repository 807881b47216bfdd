// mcr_diag.hpp -- split R-hat and ESS from the rank-normalised draws z (time order).
//
// Reference: src/mcmc_ref/diagnostics.py:76-85 (_split_chains), :136-151 (_rhat), :154-193 (_ess,
// _autocorr: UNSPLIT chains, per-chain mean of the full chain, (n - lag) normaliser, stop at the
// first negative rho), :196-201 (_variance, ddof=1).
//
// Structure (all fp64 VALU; LDS only as a staging buffer).  The reference walks lag by lag and stops at
// the first negative rho; here the lags are produced in three tiers of growing width, each tier only for
// the (parameter, kind) pairs the previous one left undecided:
//   tier 1  k_acov_seg<FIRST>  every pair, lags 0..63.  grid (segment, chain, parameter x kind).  A workgroup
//           stages 2048 draws of one chain (+ halo) ONCE in LDS (swizzled so that 16-byte reads at 64-byte
//           lane spacing hit distinct banks) and forms the raw lag products P_l = sum_i z_i z_{i+l} with an
//           8 (i) x 8 (lag) register tile per lane: 12 ds_read_b128 feed 64 FMAs, so the loop runs at the fp64
//           FMA rate.  Products are taken on the raw z (no mean yet): the mean correction
//               sum (z_i - m)(z_{i+l} - m) = P_l - m (2S - head_l - tail_l) + (n - l) m^2
//           is applied in the combine step, which makes the pass single-sweep, independent of chain length
//           (any n) and fine-grained enough to fill 256 CUs.  Lag 0 gives sum z^2, so variances cost nothing
//           extra.  (Measured with 32 lags instead: the pass got 8 % shorter -- staging, not the FMAs, is most of
//           it -- while three times as many C1 pairs needed tier 2; 64 stays.  Measured on the matrix cores
//           instead, the lag products being the diagonals of a [16 x Q] x [Q x 80] product of the segment cut into
//           rows of 16: slower, fp64 MFMA has the rate of the fp64 VALU FMA and a fifth of the tile is off the
//           diagonals -- DESIGN.md section 4.)  Everything across lanes is DPP or one pass through the LDS.
//           k_diag_combine: one workgroup per pair, one wave per chain (record sums, head / tail prefix scans),
//           then wave 0: split R-hat, var_hat, the rho terms of lags 1..63; pairs without a negative rho so far
//           are flagged.
//   tier 2  k_acov_seg<!FIRST> + k_diag_combine2: flagged pairs only (the others exit at once), lags 64..255; the
//           segment and a 272-draw halo are staged once for the three 64-lag blocks.
//   tier 3  k_acov_long + k_diag_long_scan, in rounds [256, 16384), [16384, 262144), ...: the pairs still undecided
//           (sticky chains: a random walk truncates after thousands of lags) are compacted into a list, their
//           deviations z - mean are materialised once, and EVERY CU works on them: one workgroup per
//           (256-lag group, listed pair) accumulates the products over all chains and segments in registers.
//           A round costs O(n x lags of the round) per listed pair and two near-empty launches otherwise.
// A chain that is exactly constant (min == max) contributes exactly zero deviations, as in the
// reference when its mean is exact; this keeps the W == 0 branches of _rhat / _ess exact.
#pragma once
#include "mcr_device.hpp"

namespace mcr {

constexpr int kSeg = 2048;      // draws of one chain per k_acov_seg workgroup (1024 for chains of <= 1024 draws)
#ifndef MCR_LAG1
#define MCR_LAG1 64
#endif
constexpr int kLag1 = MCR_LAG1; // tier 1: lags 0 .. kLag1-1 (32 or 64: one lag per lane in k_diag_combine)
constexpr int kSegRec = kLag1 + 12;  // doubles per tier-1 record: the lag products + 11 scalars
constexpr int kMoreBlocks = 3;  // tier 2: lags 64 .. 64 + 64*3 - 1 = 255
constexpr int kLag2 = kLag1 + 64 * kMoreBlocks;   // first lag of tier 3
constexpr int kLongGroup = 256; // lags per k_acov_long workgroup
constexpr int kLongSlots = 2;   // listed pairs a k_acov_long launch works on at a time (its grid: lag groups x this; a launch
                                // with nothing listed must stay cheap: every workgroup of it still has to find a CU with free LDS)
enum SegField { SG_S = kLag1, SG_S0, SG_Q0, SG_S1, SG_Q1, SG_MIN, SG_MAX, SG_MIN0, SG_MAX0, SG_MIN1, SG_MAX1 };   // (MIN / MAX of the rank codes: chain, first half, second half)

__device__ __forceinline__ i64 pos8(i64 j) { return j + ((j >> 3) << 1); }

// Rank code -> z.  The clamp only matters for tensors the call is about to reject (NaN draws break
// the sort's ordering, so some codes may be stale workspace bytes): it keeps the read inside the table.
__device__ __forceinline__ double zdec(const double* __restrict__ ztab, u32 code, i64 M)
{
    const u32 cmax = (u32)(2 * M - 1);
    return ztab[code < cmax ? code : cmax];
}

// One block of LPL * LG lags of raw products for a staged segment, accumulated into the lane's LPL registers.
// A: swizzled segment (>= seglen rounded up to 8 * 64 / LG draws), B: swizzled window that starts `lag base`
// draws later (LPL * LG + 16 draws longer); both zero padded.  Lane (g, ph) = (lane % LG, lane / LG) owns lags
// LPL g .. LPL g + LPL - 1 of the block and draws 8 ph .. 8 ph + 7 of every span of 8 * 64 / LG draws; the waves split
// the spans.  LPL = 8: 12 ds_read_b128 per 64 FMAs; LPL = 16: 16 per 128 (the loop is LDS-bandwidth-bound: four waves
// of a CU share one LDS, 8 cycles per b128 read, against 4 cycles per fp64 FMA on each of the four SIMDs).
template <int NT, int LG, int LPL = 8>
__device__ __forceinline__ void seg_accumulate(const double* __restrict__ A, const double* __restrict__ B,
                                               int seglen, double (&acc)[LPL])
{
    static_assert(LPL == 8 || LPL == 16, "8 or 16 lags per lane");
    constexpr int NW = NT / kWave, PH = 64 / LG, SPAN = PH * 8;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g = lane % LG, ph = lane / LG;
    const int nit = (seglen + SPAN - 1) / SPAN;
    for (int it = w; it < nit; it += NW) {
        const int i0 = it * SPAN + (ph << 3);
        const int s = i0 + g * LPL;
        const double2* pa = reinterpret_cast<const double2*>(A + 10 * (i0 >> 3));
        const double2* pb = reinterpret_cast<const double2*>(B + 10 * (s >> 3));
        double a[8], b[LPL + 8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const double2 v = pa[j]; a[2 * j] = v.x; a[2 * j + 1] = v.y; }
#pragma unroll
        for (int q = 0; q < (LPL + 8) / 8; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) { const double2 v = pb[5 * q + j]; b[8 * q + 2 * j] = v.x; b[8 * q + 2 * j + 1] = v.y; }
#pragma unroll
        for (int li = 0; li < LPL; ++li)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[li] = fma(a[k], b[k + li], acc[li]);
    }
}

// Sums the lanes' registers of one lag block over phases and waves: tot[0 .. LPL * LG).  All NT threads must call it.
template <int NT, int LG, int LPL = 8>
__device__ __forceinline__ void seg_reduce(double (&acc)[LPL], double* tot, double* wred)
{
    constexpr int NW = NT / kWave;
    static_assert(LPL * LG <= 64, "a block is at most 64 lags");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane % LG, ph = lane / LG;
#pragma unroll
    for (int li = 0; li < LPL; ++li) {
#pragma unroll
        for (int o = LG; o < 64; o <<= 1) acc[li] += __shfl_xor(acc[li], o, kWave);
    }
    __syncthreads();  // wred / tot may still be in use
    if (ph == 0) {
#pragma unroll
        for (int li = 0; li < LPL; ++li) wred[w * 64 + g * LPL + li] = acc[li];
    }
    __syncthreads();
    if (tid < LPL * LG) {
        double t = 0.0;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) t += wred[ww * 64 + tid];
        tot[tid] = t;
    }
    __syncthreads();
}

// seg_reduce for the 8 x 8 tile with the cross-lane part on DPP and the LDS: the two phases that share a DPP row are
// added with one row rotation, the four rows of every wave go to `scr` (NT / 64 * 4 * 64 doubles; it may alias the
// staged segment, which is dead by then) and 64 threads add them.  No ds_bpermute.  All NT threads must call it.
template <int NT, int LG = 8>
__device__ __forceinline__ void seg_reduce_rows(double (&acc)[8], double* tot, double* scr)
{
    constexpr int NW = NT / kWave;
    static_assert(LG == 8 || LG == 4, "8 or 4 lag groups of 8 lags");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane & (LG - 1), row = lane >> 4;
#pragma unroll
    for (int li = 0; li < 8; ++li) {
        if (LG == 4) acc[li] += dpp_f64<kDppRor4>(acc[li]);       // four phases share a row: l, l + 4, l + 8, l + 12
        acc[li] += dpp_f64<kDppRor8>(acc[li]);
    }
    __syncthreads();  // scr / tot may still be in use
    if ((lane & 15) < LG) {
#pragma unroll
        for (int li = 0; li < 8; ++li) scr[(w * 4 + row) * 64 + g * 8 + li] = acc[li];
    }
    __syncthreads();
    if (tid < LG * 8) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < NW * 4; ++r) t += scr[r * 64 + tid];
        tot[tid] = t;
    }
    __syncthreads();
}

// FIRST == true : lags 0..kLag1-1 + the segment's sums (record of kSegRec doubles), every pair.
// FIRST == false: lags kLag1..kLag1+64*kMoreBlocks-1 (record of 64*kMoreBlocks doubles), flagged pairs only; the
//                 segment and its 272-draw halo are staged once and serve all three lag blocks.
// grid (nseg, C, 2 * P); blockIdx.z = 2 * p + kind.
template <int NT, int SEG, bool FIRST>
__global__ __launch_bounds__(NT) void k_acov_seg(const u32* __restrict__ zb, const u32* __restrict__ zt,
                                                 const double* __restrict__ ztab, i64 M,
                                                 const i64* __restrict__ off, int C, i64 n, i64 nh, int nseg,
                                                 const unsigned* __restrict__ more, double* __restrict__ rec, int kind_sel)
{
    constexpr int NW = NT / kWave;
    static_assert(SEG % 128 == 0, "spans of the widest tile are 128 draws");
    static_assert(kLag1 == 64 || kLag1 == 32, "tier 1 is one or half a lag block");
    constexpr int WIN = FIRST ? SEG + 80 : SEG + 64 * kMoreBlocks + 80;     // staged window (draws)
    constexpr int LX = WIN / 8 * 10;                                         // its swizzled length
    constexpr int NLD = (WIN + NT - 1) / NT;
    static_assert(LX >= NW * 4 * 64, "the reduction scratch aliases the staged window");
    __shared__ __attribute__((aligned(16))) double sx[LX];
    __shared__ __attribute__((aligned(16))) double scr2[FIRST ? 8 : NW * 4 * 64];
    __shared__ double tot[64];
    __shared__ double wred[NW * 4 * 8];     // (the half windows' min / max go through `tot`, free until the lag products are reduced:
                                            //  512 more bytes here cost the seventh resident workgroup per CU)

    const int tid = threadIdx.x;
    const int seg = blockIdx.x, c = blockIdx.y;
    // kind_sel < 0: grid.z = 2 P, both kinds in one launch; 0 / 1: grid.z = P, this kind only (a lone call runs the bulk
    // half on a second stream under the fold kernel: launch_diag)
    const i64 pk = (kind_sel < 0) ? (i64)blockIdx.z : 2 * (i64)blockIdx.z + kind_sel, p = pk >> 1;
    const int kind = (int)(pk & 1);
    if (!FIRST && more[pk] == 0u) return;
    const u32* zc = (kind ? zt : zb) + p * M + off[c];     // rank codes; z = ztab[code]
    const i64 nc = off[c + 1] - off[c], hc = nc / 2;
    const i64 hspan = (hc > 0) ? hc + nh : 0;
    const i64 nload = (n > hspan) ? n : hspan;          // <= nc
    const i64 s0 = (i64)seg * SEG;
    const int seglen = (int)((n - s0 < 0) ? 0 : ((n - s0 < (i64)SEG) ? n - s0 : (i64)SEG));

    if (FIRST) {
        // ---- stage (all loads of a lane in flight together) + segment sums; all index tests in 32-bit,
        //      relative to the segment start ----
        auto rel = [&](i64 x) -> int { const i64 d = x - s0; return (int)(d < 0 ? 0 : (d > WIN ? WIN : d)); };
        const int r_load = rel(nload);                      // draws [0, r_load) of the window exist
        const int r_n = rel(n);                             // draws [0, r_n) enter the products
        const int own = (r_load < SEG) ? r_load : SEG;    // this workgroup owns window slots [0, own)
        const int own_n = (own < r_n) ? own : r_n;
        auto clip = [&](int x) -> int { return x < own ? x : own; };
        const int a0 = (hc > 0) ? clip(rel(0)) : 0, a1 = (hc > 0) ? clip(rel(nh)) : 0;             // first half
        const int b0 = (hc > 0) ? clip(rel(hc)) : 0, b1 = (hc > 0) ? clip(rel(hc + nh)) : 0;        // second half
        double S = 0.0, S0 = 0.0, Q0 = 0.0, S1 = 0.0, Q1 = 0.0;
        // min / max only tell a constant chain (min == max) from the others, and z = ztab[code] is strictly monotone:
        // they are taken on the integer codes (full-rate v_min_u32 / v_max_u32; the f64 pair is not, and needs its operands
        // canonicalised) and stored as doubles, which hold a u32 exactly
        u32 cmin = 0xFFFFFFFFu, cmax = 0u;
        u32 c0min = 0xFFFFFFFFu, c0max = 0u, c1min = 0xFFFFFFFFu, c1max = 0u;    // the same for the two halves of the chain (split R-hat)
        double v[NLD];
        u32 cd[NLD];
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int j = u * NT + tid;
            cd[u] = (j < r_load) ? zc[s0 + j] : 0u;
            v[u] = (j < r_load) ? zdec(ztab, cd[u], M) : 0.0;
        }
        // The three windows are ranges of j and a wave's 64 slots are consecutive: a wave that lies wholly inside (or
        // outside) a window -- all but a handful -- decides that with scalar compares and adds without per-lane masks.
        const int jw0 = __builtin_amdgcn_readfirstlane(tid & ~63);
#define MCR_IN_WINDOW(lo, hi, BODY)                                                        \
        if (jw >= (lo) && jw + 64 <= (hi)) { BODY }                                        \
        else if (jw + 64 > (lo) && jw < (hi)) { if (j >= (lo) && j < (hi)) { BODY } }
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int jw = u * NT + jw0, j = u * NT + tid;
            const double x = v[u];
            MCR_IN_WINDOW(0, own_n, S += x; cmin = min(cmin, cd[u]); cmax = max(cmax, cd[u]);)
            MCR_IN_WINDOW(a0, a1, S0 += x; Q0 = fma(x, x, Q0); c0min = min(c0min, cd[u]); c0max = max(c0max, cd[u]);)
            MCR_IN_WINDOW(b0, b1, S1 += x; Q1 = fma(x, x, Q1); c1min = min(c1min, cd[u]); c1max = max(c1max, cd[u]);)
            if (j < WIN) sx[pos8(j)] = (j < r_n) ? x : 0.0;
        }
#undef MCR_IN_WINDOW
        // The five sums and min / max: DPP inside the rows of 16 lanes, the 4 NW row results through the LDS, seven
        // lanes finish (one barrier; no cross-row shuffles).
        S = row_sum(S); S0 = row_sum(S0); Q0 = row_sum(Q0); S1 = row_sum(S1); Q1 = row_sum(Q1);
        cmin = row_min_u32(cmin); cmax = row_max_u32(cmax);
        c0min = row_min_u32(c0min); c0max = row_max_u32(c0max); c1min = row_min_u32(c1min); c1max = row_max_u32(c1max);
        static_assert(NW * 4 * 4 <= 64, "the four half-window fields of every DPP row fit `tot`");
        if ((tid & 15) == 0) {
            double* q = wred + (tid >> 4) * 8;
            q[0] = S; q[1] = S0; q[2] = Q0; q[3] = S1; q[4] = Q1; q[5] = (double)cmin; q[6] = (double)cmax;
            double* q2 = tot + (tid >> 4) * 4;
            q2[0] = (double)c0min; q2[1] = (double)c0max; q2[2] = (double)c1min; q2[3] = (double)c1max;
        }
        __syncthreads();
        double* r = rec + ((pk * C + c) * (i64)nseg + seg) * kSegRec;
        if (tid < 11) {
            const double* src = (tid < 7) ? wred + tid : tot + (tid - 7);
            const int stride = (tid < 7) ? 8 : 4;
            double t = src[0];
            for (int w = 1; w < NW * 4; ++w) {
                const double x = src[w * stride];
                t = (tid < 5) ? t + x : (((tid & 1) == 1) ? fmin(t, x) : fmax(t, x));      // 5, 7, 9: minima; 6, 8, 10: maxima
            }
            r[SG_S + tid] = t;        // SG_S, SG_S0, SG_Q0, SG_S1, SG_Q1, SG_MIN, SG_MAX, SG_MIN0 .. SG_MAX1 are consecutive
        }
        double acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.0;
        seg_accumulate<NT, kLag1 / 8, 8>(sx, sx, seglen, acc);
        seg_reduce_rows<NT, kLag1 / 8>(acc, tot, sx);          // the staged window is dead after the barrier inside
        if (tid < kLag1) r[tid] = tot[tid];
    } else {
        double* r = rec + ((pk * C + c) * (i64)nseg + seg) * (64 * kMoreBlocks);
        if (seglen == 0) {   // nothing of [0, n) in this segment: zero record
            for (int j = tid; j < 64 * kMoreBlocks; j += NT) r[j] = 0.0;
            return;
        }
        double v[NLD];
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int j = u * NT + tid;
            v[u] = (j < WIN && s0 + j < n) ? zdec(ztab, zc[s0 + j], M) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int j = u * NT + tid;
            if (j < WIN) sx[pos8(j)] = v[u];
        }
        __syncthreads();
        for (int blk = 0; blk < kMoreBlocks; ++blk) {
            const int lb = kLag1 + 64 * blk;
            double acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = 0.0;
            seg_accumulate<NT, 8, 8>(sx, sx + 10 * (lb >> 3), seglen, acc);
            seg_reduce_rows<NT>(acc, tot, scr2);
            if (tid < 64) r[blk * 64 + tid] = tot[tid];
        }
    }
}

// k_tier3 walks the lags of a listed pair in STAGES of kT3StageGroups groups of 256 lags; per pair and stage one word counts
// the finished groups (low bits) and carries kT3Prev once the stage in front has been scanned without a decision.
constexpr int kT3StageGroups = 8, kT3Stages = 9;          // stages of 2, 6, 8, 8, ... groups: 64 groups cover chains up to 16 384 + 256 draws
constexpr unsigned kT3Prev = 0x80000000u;
// first lag group of stage st: most walks that reach tier 3 at all end within a few hundred lags of it, so the first stage
// is short (lags 256 .. 767), the second takes the rest of the first 2 048, the others 2 048 each
// (chains of up to 2 304 draws -- 8 groups at most, the packaged corpus has 3 -- are ONE stage: with so few items per pair
//  a second stage only adds a hand-over, measured 71 -> 120 us on 13 listed pairs of 10 x 1 000 draws)
__host__ __device__ constexpr int t3_stage_first(int st, int groups)
{
    return groups <= kT3StageGroups ? (st == 0 ? 0 : groups) : (st == 0 ? 0 : (st == 1 ? 2 : kT3StageGroups * (st - 1)));
}

// Per-chain state kept between the combine kernels.
constexpr int kChState = 6;   // mean, S, constant flag, head(kLag1), tail(kLag1), the chain's LEFT-TO-RIGHT mean (NaN until a band lag needs it)
// Per-pair scan state: rho_sum, terms, var_hat, decided (tier 3)
constexpr int kPairState = 4;

// ---- GUARD BAND: no `rho < 0` decision on a value within round-off of zero ----
// The reference stops at the first `rho < 0` (diagnostics.py:171-177) and the number of accumulated terms is an integer
// output.  Every tier forms its autocovariances in another summation order than _autocorr's left-to-right loop (tier 1 / 2:
// FMA'd raw products in a register tile, tree sums over lanes and segments, then the cancelling mean correction,
// ~1e-15 rho_0; tier 3: tree sums ~1e-16 rho_0 or FFTs ~1e-14 rho_0), so a rho that close to zero could take the other sign
// here than in the reference and move the truncation lag by one.  No decision is therefore taken on such a value: every
// tier looks for the first lag whose rho is below +band (kRhoBand, MCR_RHO_BAND); if that rho is above -band, the lag is
// RE-DERIVED THE REFERENCE'S WAY before it is compared with zero (ref_cov_sum below) and the walk resumes behind it when
// the result is not negative.  Lags outside the band keep the value of their tier.
constexpr double kRhoBand = 1e-10;
constexpr int kGuardMax = 64;     // band lags re-derived per pair and kernel (chains up to 32 768 draws; fewer beyond, guard_budget)

// A band lag costs a sequential pass over the chains (one thread per chain adds left to right: ~10 cycles per draw, 45 us
// for 10 000 draws, 4.5 ms for a million).  The budget bounds what an adversarial tensor (rho exactly zero at many lags of
// many long chains) can cost a call: 64 lags up to 32 768 draws, proportionally fewer beyond, never fewer than 4; lags
// beyond the budget are decided on the value their tier has.
__device__ __forceinline__ int guard_budget(i64 n)
{
    const i64 b = (n <= 32768) ? (i64)kGuardMax : ((i64)kGuardMax * 32768) / n;
    return (int)(b < 4 ? 4 : b);
}

constexpr int kGuardChains12 = 4, kGuardChunk12 = 128;     // tiers 1 / 2: 8 KB of LDS in the combine kernels
constexpr int kGuardChains = 8, kGuardChunk = 256;          // tier 3 (inside the union with the products' staging buffers)
template <int CH, int CK>
struct GuardLds {
    double gA[CH][CK], gB[CH][CK];
    double gcov[CH], gmean[CH];
    double gsum;
};

// cov_sum of _autocorr(chains, lag, .) (diagnostics.py:180-193) for one (parameter, kind): per chain
//     mean = sum(chain[:n]) / n                                   left to right
//     cov  = sum_i (chain[i] - mean) * (chain[i + lag] - mean)    left to right, the product rounded before it is added (no FMA)
//     cov /= n - lag ;  cov_sum += cov                             chains in order
// with chain[i] = ztab[code]: the z the other tiers use.  One thread per chain does the additions, all threads of the
// workgroup stage the draws (table look-ups, CK per chain at a time) into the LDS for it.  With z equal to the reference's
// (bit for bit in the central 85 % of the ranks, within 1-2 ulp of the device log in the tails) this IS the reference's
// cov_sum bit for bit -- its mean included: the tiers' S / n is a tree sum and differs from sum(chain) / n by an ulp
// (ADVICE r3), so the left-to-right mean is taken here, once per chain, and kept in chstate[.][5].
// All blockDim.x threads call it with identical arguments; every thread returns the value.
template <int CH, int CK>
__device__ __noinline__ double ref_cov_sum(const u32* __restrict__ z, const double* __restrict__ ztab, i64 M,
                                           const i64* __restrict__ off, int C, i64 n, i64 lag,
                                           double* __restrict__ chst, GuardLds<CH, CK>& G)
{
#pragma clang fp contract(off)      // products and sums round like CPython's
    const int tid = threadIdx.x, NT = (int)blockDim.x;
    __syncthreads();
    if (tid == 0) G.gsum = 0.0;
    const i64 len = n - lag;
    for (int c0 = 0; c0 < C; c0 += CH) {
        const int nc = (C - c0 < CH) ? C - c0 : CH;
        __syncthreads();
        if (tid < nc) G.gmean[tid] = chst[(c0 + tid) * kChState + 5];
        __syncthreads();
        bool need = false;
        for (int c = 0; c < nc; ++c) need = need || (G.gmean[c] != G.gmean[c]);
        if (need) {                                          // the chains' left-to-right means
            double s = 0.0;
            for (i64 i0 = 0; i0 < n; i0 += CK) {
                const int cl = (int)((n - i0 < CK) ? n - i0 : CK);
                __syncthreads();
                for (int e = tid; e < nc * CK; e += NT) {
                    const int c = e / CK, j = e - c * CK;
                    if (j < cl) G.gA[c][j] = zdec(ztab, z[off[c0 + c] + i0 + j], M);
                }
                __syncthreads();
                if (tid < nc)
                    for (int j = 0; j < cl; ++j) s += G.gA[tid][j];
            }
            if (tid < nc) { const double m = s / (double)n; G.gmean[tid] = m; chst[(c0 + tid) * kChState + 5] = m; }
            __syncthreads();
        }
        double cov = 0.0;
        for (i64 i0 = 0; i0 < len; i0 += CK) {
            const int cl = (int)((len - i0 < CK) ? len - i0 : CK);
            __syncthreads();
            for (int e = tid; e < nc * CK; e += NT) {
                const int c = e / CK, j = e - c * CK;
                if (j < cl) {
                    const u32* zc = z + off[c0 + c];
                    const double m = G.gmean[c];
                    G.gA[c][j] = zdec(ztab, zc[i0 + j], M) - m;
                    G.gB[c][j] = zdec(ztab, zc[i0 + j + lag], M) - m;
                }
            }
            __syncthreads();
            if (tid < nc)
                for (int j = 0; j < cl; ++j) cov += G.gA[tid][j] * G.gB[tid][j];
        }
        if (tid < nc) G.gcov[tid] = cov / (double)len;
        __syncthreads();
        if (tid == 0)
            for (int c = 0; c < nc; ++c) G.gsum += G.gcov[c];
    }
    __syncthreads();
    return G.gsum;
}

// The first truly negative rho among the 64 lags of a block (tiers 1 and 2).  Lane l of WAVE 0 holds rho of lag lb + l
// (`valid`: the lag exists and enters the walk); every thread of the workgroup calls this, the other waves only help with
// the re-derivations.  Returns (to every thread) the lane of the first negative rho, 64 if there is none; wave 0's `rho`
// of a re-derived lag is replaced by the reference's value.
template <class F>
__device__ __forceinline__ int first_negative_guarded(double& rho, bool valid, double band, int& budget, F&& rederive,
                                                      int* s_req, unsigned* __restrict__ guard_count)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int from = 0;
    unsigned long long exact = 0ull;          // lanes whose rho has been re-derived
    for (;;) {
        if (w == 0) {
            int req = -1, res = 64;
            for (;;) {
                unsigned long long low = __ballot(valid && rho < band);
                low = (from < 64) ? (low & (~0ull << from)) : 0ull;
                if (!low) break;
                const int first = __ffsll((long long)low) - 1;
                const double rf = readlane_f64(rho, first);
                if (((exact >> first) & 1ull) || rf < -band || budget <= 0) {
                    if (rf < 0.0) { res = first; break; }
                    from = first + 1;
                    continue;
                }
                req = first;
                break;
            }
            if (lane == 0) { s_req[0] = req; s_req[1] = res; }
        }
        __syncthreads();
        const int req = s_req[0];
        if (req < 0) return s_req[1];
        const double r = rederive(req);          // all threads; barriers inside
        if (threadIdx.x == 0) atomicAdd(guard_count, 1u);
        if (w == 0) {
            if (lane == req) rho = r;
            exact |= 1ull << req;
            --budget;
        }
    }
}

// One workgroup per (parameter, kind), one wave per chain (kCombineWaves at most; the chains' record sums, head / tail
// scans and loads are independent, so this divides the latency chain of the kernel by the number of chains), then wave 0
// alone.  grid (P, 2), block 64 * min(C, kCombineWaves).  Also resets the tier-3 list of this call.
constexpr int kCombineWaves = 8;
__global__ __launch_bounds__(64 * kCombineWaves) void k_diag_combine(const u32* __restrict__ zb, const u32* __restrict__ zt,
                                                     const double* __restrict__ ztab, i64 M,
                                                     const i64* __restrict__ off, int C, i64 n, i64 nh,
                                                     int nseg, const double* __restrict__ rec,
                                                     double* __restrict__ res, i64 P, unsigned* __restrict__ more,
                                                     double* __restrict__ state, double* __restrict__ chstate,
                                                     unsigned* __restrict__ long_count, unsigned* __restrict__ pair_done,
                                                     double band, unsigned* __restrict__ guard_count, int kind_sel)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* cm = reinterpret_cast<double*>(smem);   // C   chain means
    double* cq = cm + C;                             // C   sum (z - mean)^2
    double* hm = cq + C;                             // 2C  half means
    double* hq = hm + 2 * C;                         // 2C  half sums of squared deviations
    __shared__ double wcov[kCombineWaves][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, W = (int)(blockDim.x >> 6);
    const i64 p = blockIdx.x;
    const int kind = (kind_sel < 0) ? (int)blockIdx.y : kind_sel;      // grid (P, 2), or (P, 1) for one kind
    const i64 pk = p * 2 + kind;
    const u32* z = (kind ? zt : zb) + p * M;
    const int f_rhat = kind ? R_RHAT_TAIL : R_RHAT_BULK;
    const int f_ess = kind ? R_ESS_TAIL : R_ESS_BULK;
    const int f_lag = kind ? R_LAG_TAIL : R_LAG_BULK;
    if (pk == 0 && threadIdx.x == 0) long_count[0] = 0u;     // k_long_list (a later launch on this stream) sets the real length
    if (pair_done != nullptr && threadIdx.x == 0) {        // k_tier3's per-pair words [kT3Stages + 1][2 P]: one per stage, then "decided"
        const i64 npk = 2 * P;
        pair_done[pk] = kT3Prev;                           // stage 0 has no stage in front of it to wait for
        for (int st = 1; st <= kT3Stages; ++st) pair_done[st * npk + pk] = 0u;
    }

    double covsum = 0.0;   // lane l: sum over (this wave's) chains of sum_i (z_i - m)(z_{i+l} - m)
    for (int c = w; c < C; c += W) {
        const double* R = rec + ((pk * C + c) * (i64)nseg) * kSegRec;
        double Pl = 0.0, S = 0.0, S0 = 0.0, Q0 = 0.0, S1 = 0.0, Q1 = 0.0, Q = 0.0;
        double vmin = INFINITY, vmax = -INFINITY, vmin0 = INFINITY, vmax0 = -INFINITY, vmin1 = INFINITY, vmax1 = -INFINITY;
        for (int sgm = 0; sgm < nseg; ++sgm) {
            const double* r = R + (i64)sgm * kSegRec;
            Pl += (lane < kLag1) ? r[lane] : 0.0; Q += r[0];
            S += r[SG_S]; S0 += r[SG_S0]; Q0 += r[SG_Q0]; S1 += r[SG_S1]; Q1 += r[SG_Q1];
            vmin = fmin(vmin, r[SG_MIN]); vmax = fmax(vmax, r[SG_MAX]);
            vmin0 = fmin(vmin0, r[SG_MIN0]); vmax0 = fmax(vmax0, r[SG_MAX0]);
            vmin1 = fmin(vmin1, r[SG_MIN1]); vmax1 = fmax(vmax1, r[SG_MAX1]);
        }
        const bool constant = !(vmin < vmax);
        // a HALF of the chain can be constant when the chain is not (a few draws, heavy ties: the one odd draw is the last
        // of an odd-length chain, which the split drops): its squared deviations are exactly zero in the reference
        // (diagnostics.py:196-201 on equal values), and Q - S m is that only up to rounding
        const bool const0 = constant || !(vmin0 < vmax0), const1 = constant || !(vmin1 < vmax1);
        const double m = (n > 0) ? S / (double)n : 0.0;
        const u32* zc = z + off[c];
        double th, tt;
        const double head = wave_excl_scan((lane < n) ? zdec(ztab, zc[lane], M) : 0.0, th);            // sum_{i<l} z_i
        const double tail = wave_excl_scan((lane < n) ? zdec(ztab, zc[n - 1 - lane], M) : 0.0, tt);    // sum_{i>=n-l} z_i
        if (!constant && lane < n && lane < kLag1)
            covsum += Pl - m * ((S - tail) + (S - head)) + (double)(n - lane) * m * m;
        const double h32 = (kLag1 < 64) ? __shfl(head, kLag1 & 63, kWave) : th;   // sums of the first / last kLag1 draws
        const double t32 = (kLag1 < 64) ? __shfl(tail, kLag1 & 63, kWave) : tt;
        if (lane == 0) {
            cm[c] = m;
            cq[c] = constant ? 0.0 : Q - S * m;
            const double m0 = (nh > 0) ? S0 / (double)nh : 0.0, m1 = (nh > 0) ? S1 / (double)nh : 0.0;
            hm[2 * c] = m0; hm[2 * c + 1] = m1;
            hq[2 * c] = const0 ? 0.0 : fmax(Q0 - S0 * m0, 0.0);
            hq[2 * c + 1] = const1 ? 0.0 : fmax(Q1 - S1 * m1, 0.0);
            double* cs = chstate + (pk * C + c) * kChState;
            cs[0] = m; cs[1] = S; cs[2] = constant ? 1.0 : 0.0; cs[3] = h32; cs[4] = t32; cs[5] = NAN;
        }
    }
    wcov[w][lane] = covsum;
    __syncthreads();
    if (w == 0)
        for (int ww = 1; ww < W; ++ww) covsum += wcov[ww][lane];      // fixed order
    __shared__ double bc[2];   // var_hat, mode (0: NaN, 1: var_hat == 0, 2: scan)
    __shared__ int s_req[2];
    __shared__ GuardLds<kGuardChains12, kGuardChunk12> G;
    if (threadIdx.x == 0) {
        // ---- split R-hat (diagnostics.py:136-151); chains shorter than 2 draws are skipped ----
        int ms = 0;
        for (int k = 0; k < C; ++k) ms += (off[k + 1] - off[k] >= 2) ? 2 : 0;
        double rhat;
        if (ms < 2 || nh < 2) {
            rhat = NAN;
        } else {
            double st = 0.0;
            for (int k = 0; k < C; ++k)
                if (off[k + 1] - off[k] >= 2) { st += hm[2 * k]; st += hm[2 * k + 1]; }
            const double mt = st / (double)ms;
            double sb = 0.0, sw = 0.0;
            for (int k = 0; k < C; ++k)
                if (off[k + 1] - off[k] >= 2) {
                    const double a = hm[2 * k] - mt, b = hm[2 * k + 1] - mt;
                    sb += a * a; sb += b * b;
                    sw += hq[2 * k] / (double)(nh - 1);
                    sw += hq[2 * k + 1] / (double)(nh - 1);
                }
            const double vb = (double)nh * sb / (double)(ms - 1);
            const double vw = sw / (double)ms;
            const double vh = (double)(nh - 1) / (double)nh * vw + vb / (double)nh;
            rhat = (vw == 0.0) ? ((vb == 0.0) ? 1.0 : INFINITY) : sqrt(vh / vw);
        }
        res[f_rhat * P + p] = rhat;
        // ---- ESS prologue (diagnostics.py:154-169) ----
        double vh = 0.0, mode = 0.0;
        if (!(C == 0 || n < 2)) {
            double st = 0.0;
            for (int k = 0; k < C; ++k) st += cm[k];
            const double mt = st / (double)C;
            double sb = 0.0, sw = 0.0;
            for (int k = 0; k < C; ++k) {
                const double a = cm[k] - mt;
                sb += a * a;
                sw += fmax(cq[k], 0.0) / (double)(n - 1);
            }
            const double vb = (C > 1) ? (double)n * sb / (double)(C - 1) : 0.0;
            const double vw = sw / (double)C;
            vh = (double)(n - 1) / (double)n * vw + vb / (double)n;
            mode = (vh == 0.0) ? 1.0 : 2.0;
        }
        bc[0] = vh; bc[1] = mode;
    }
    __syncthreads();
    const double vh = bc[0];
    const int mode = (int)bc[1];
    // ---- rho terms of lags 1..kLag1-1, one per lane of wave 0; the first negative one stops the sum
    //      (diagnostics.py:171-177), a rho within the guard band of zero being re-derived the reference's way first
    //      (the other waves stay for that).  The prefix is summed with a wave tree instead of left to right.
    double rho = 0.0;
    const bool valid = mode == 2 && lane >= 1 && lane < n && lane < kLag1;
    const double den = (double)C * vh;
    if (valid) rho = (covsum / (double)(n - lane)) / den;
    int budget = guard_budget(n);
    const int first = first_negative_guarded(rho, valid && w == 0, band, budget,
        [&](int l) { return ref_cov_sum(z, ztab, M, off, C, n, (i64)l, chstate + pk * C * kChState, G) / den; },
        s_req, guard_count);                                                  // lag of the first negative rho (64: none)
    if (w != 0) return;
    const double rho_sum = wave_sum((valid && lane < first) ? rho : 0.0);
    const int nvalid = (int)__popcll(__ballot(valid && lane < first));
    if (lane != 0) return;
    unsigned cont = 0u;
    if (mode == 0) {
        res[f_ess * P + p] = NAN;
        res[f_lag * P + p] = 0.0;
    } else if (mode == 1) {
        res[f_ess * P + p] = (double)((i64)C * n);
        res[f_lag * P + p] = 0.0;
    } else {
        cont = (first == 64 && n > kLag1) ? 1u : 0u;
        if (!cont) {
            res[f_ess * P + p] = (double)((i64)C * n) / (1.0 + 2.0 * rho_sum);
            res[f_lag * P + p] = (double)nvalid;
        }
    }
    more[pk] = cont;
    double* stp = state + pk * kPairState;
    stp[0] = rho_sum; stp[1] = (double)nvalid; stp[2] = vh; stp[3] = 1.0;      // [3]: decided, as far as tier 3 is concerned
}

// Tier 2 for flagged pairs: lags kLag1..kLag2-1 from the second k_acov_seg pass.  A pair that is still undecided
// (very sticky chains) gets its deviations z - mean materialised once ([M] doubles per pair: the sort's key
// buffers, free by now) and is marked for tier 3 (state[pk][3] = 0).  grid (P, 2), block 1024.
//
// The tier-3 LIST is built from those marks in ascending pair order by k_long_list (next launch): which pairs the FFT
// slots serve (the first fft.slots list entries, mcr_fft.hpp) is then a function of the data alone, not of the order in
// which workgroups happened to run -- FFT and direct products agree to ~1e-14, not bit for bit, so with a list appended
// to by atomics the same call could return different ESS bits from run to run.  (Measured: building the
// list here, by the workgroup that finishes last behind agent-scope fences, costs 20 ns per workgroup of the grid --
// 41 us on 1000 parameters; a launch of its own costs 4.)
__device__ __forceinline__ void combine2_pair(const u32* __restrict__ zb, const u32* __restrict__ zt,
                                              const double* __restrict__ ztab, i64 M,
                                              const i64* __restrict__ off, int C, i64 n, int nseg,
                                              const double* __restrict__ rec2,
                                              const unsigned* __restrict__ more,
                                              double* __restrict__ state,
                                              double* __restrict__ chstate,
                                              double* __restrict__ res, i64 P,
                                              double* __restrict__ dev_b, double* __restrict__ dev_t,
                                              double* hb, double (*wcov)[64], double* ctl, bool defer_dev,
                                              double band, unsigned* __restrict__ guard_count, int* s_req,
                                              GuardLds<kGuardChains12, kGuardChunk12>& G)
{
    constexpr int NW2 = 16;
    const i64 p = blockIdx.x;
    const int kind = blockIdx.y;
    const i64 pk = p * 2 + kind;
    if (more[pk] == 0u) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const u32* z = (kind ? zt : zb) + p * M;
    const int f_ess = kind ? R_ESS_TAIL : R_ESS_BULK;
    const int f_lag = kind ? R_LAG_TAIL : R_LAG_BULK;
    double rho_sum = state[pk * kPairState + 0];
    i64 terms = (i64)state[pk * kPairState + 1];
    const double vhat = state[pk * kPairState + 2];
    bool stop = false;
    int budget = guard_budget(n);

    // ---- lags kLag1..kLag2-1, block by block ----
    // hb[c] / tb[c]: running sums of the first / last `lb` draws of chain c (head / tail bases)
    double* tb = hb + C;
    for (int c = tid; c < C; c += (int)blockDim.x) { hb[c] = chstate[(pk * C + c) * kChState + 3]; tb[c] = chstate[(pk * C + c) * kChState + 4]; }
    __syncthreads();
    for (int blk = 0; blk < kMoreBlocks && !stop; ++blk) {
        const i64 lb = kLag1 + 64 * blk;
        if (lb >= n) break;
        {   // one wave per chain (the chains' scans and record sums are independent), then wave 0 alone
            double covsum = 0.0;
            for (int c = w; c < C; c += NW2) {
                const double* cs = chstate + (pk * C + c) * kChState;
                const u32* zc = z + off[c];
                double th, tt;
                const double hx = wave_excl_scan((lb + lane < n) ? zdec(ztab, zc[lb + lane], M) : 0.0, th);
                const double tx = wave_excl_scan((lb + lane < n) ? zdec(ztab, zc[n - 1 - (lb + lane)], M) : 0.0, tt);
                const double head = hb[c] + hx, tail = tb[c] + tx;   // sums of the first / last (lb + lane) draws
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) { hb[c] += th; tb[c] += tt; }
                if (cs[2] != 0.0) continue;                       // constant chain: zero deviations
                const double m = cs[0], S = cs[1];
                double Pl = 0.0;
                for (int sgm = 0; sgm < nseg; ++sgm)
                    Pl += rec2[((pk * C + c) * (i64)nseg + sgm) * (64 * kMoreBlocks) + blk * 64 + lane];
                const i64 lag = lb + lane;
                if (lag < n) covsum += Pl - m * ((S - tail) + (S - head)) + (double)(n - lag) * m * m;
            }
            wcov[w][lane] = covsum;
        }
        __syncthreads();
        {
            const double den = (double)C * vhat;
            const i64 lag = lb + lane;
            const bool valid = w == 0 && lag < n;
            double rho = 0.0;
            if (w == 0) {
                double covsum = wcov[0][lane];
                const int nw = (C < NW2) ? C : NW2;
                for (int ww = 1; ww < nw; ++ww) covsum += wcov[ww][lane];     // fixed order
                rho = valid ? (covsum / (double)(n - lag)) / den : 0.0;
            }
            // first negative rho of the block; a rho within the guard band of zero is re-derived the reference's way first
            const int first = first_negative_guarded(rho, valid, band, budget,
                [&](int l) { return ref_cov_sum(z, ztab, M, off, C, n, lb + (i64)l, chstate + pk * C * kChState, G) / den; },
                s_req, guard_count);
            if (w == 0) {
                const double add = wave_sum((valid && lane < first) ? rho : 0.0);
                const int cnt = (int)__popcll(__ballot(valid && lane < first));
                if (lane == 0) { ctl[0] = (first < 64) ? 1.0 : 0.0; ctl[1] = add; ctl[2] = (double)cnt; }
            }
        }
        __syncthreads();
        rho_sum += ctl[1];
        terms += (i64)ctl[2];
        stop = ctl[0] != 0.0;
        __syncthreads();
    }
    if (stop || kLag2 >= n) {
        if (tid == 0) {
            res[f_ess * P + p] = (double)((i64)C * n) / (1.0 + 2.0 * rho_sum);
            res[f_lag * P + p] = (double)terms;
        }
        return;
    }
    // ---- still undecided at lag kLag2: hand the pair to tier 3 ----
    // (long chains: the deviations are written by k_dev_fill, chip-wide, once the list exists -- one workgroup walking
    //  C x n dependent table reads took 0.6 ms per pair at n = 100 000)
    double* dev = (kind ? dev_t : dev_b) + p * M;
    for (int c = 0; c < C && !defer_dev; ++c) {
        const double* cs = chstate + (pk * C + c) * kChState;
        const bool konst = cs[2] != 0.0;
        const double m = cs[0];
        const u32* zc = z + off[c];
        double* dc = dev + off[c];
        for (i64 i = tid; i < n; i += blockDim.x) dc[i] = konst ? 0.0 : zdec(ztab, zc[i], M) - m;
    }
    if (tid == 0) {
        double* stp = state + pk * kPairState;
        stp[0] = rho_sum; stp[1] = (double)terms; stp[3] = 0.0;     // [3] == 0: on the tier-3 list (built below)
    }
}

__global__ __launch_bounds__(1024) void k_diag_combine2(const u32* __restrict__ zb, const u32* __restrict__ zt,
                                                       const double* __restrict__ ztab, i64 M,
                                                       const i64* __restrict__ off, int C, i64 n, int nseg,
                                                       const double* __restrict__ rec2,
                                                       const unsigned* __restrict__ more,
                                                       double* __restrict__ state,
                                                       double* __restrict__ chstate,
                                                       double* __restrict__ res, i64 P,
                                                       double* __restrict__ dev_b, double* __restrict__ dev_t,
                                                       const double* __restrict__ part, int ntiles, int defer_dev,
                                                       double band, unsigned* __restrict__ guard_count)
{
    __shared__ double ctl[3];
    __shared__ double wcov[16][64];
    __shared__ int s_req[2];
    __shared__ GuardLds<kGuardChains12, kGuardChunk12> G;
    extern __shared__ __attribute__((aligned(16))) char smem2[];
    const int tid = threadIdx.x;
    // pooled mean / std from the tile partials and rhat = pymax(bulk, tail) (k_diag_combine wrote both): the work of
    // k_finalize, done here by the first thread of the parameter's first workgroup
    if (blockIdx.y == 0 && tid == 0) finalize_param(part, ntiles, M, P, C, res, blockIdx.x);
    combine2_pair(zb, zt, ztab, M, off, C, n, nseg, rec2, more, state, chstate, res, P, dev_b, dev_t,
                  reinterpret_cast<double*>(smem2), wcov, ctl, defer_dev != 0, band, guard_count, s_req, G);
}

// Deviations z - mean of the listed pairs, in time order, for tier 3: grid (chunks of 4096 pooled draws, slots), block 256.
__global__ __launch_bounds__(256) void k_dev_fill(const u32* __restrict__ zb, const u32* __restrict__ zt,
                                                  const double* __restrict__ ztab, i64 M, const i64* __restrict__ off, int C,
                                                  i64 n, const double* __restrict__ chstate,
                                                  const unsigned* __restrict__ long_count, const unsigned* __restrict__ long_list,
                                                  double* __restrict__ dev_b, double* __restrict__ dev_t)
{
    const unsigned count = *long_count;
    const i64 j0 = (i64)blockIdx.x * 4096;
    for (unsigned slot = blockIdx.y; slot < count; slot += gridDim.y) {
        const i64 pk = long_list[slot], p = pk >> 1;
        const u32* z = ((pk & 1) ? zt : zb) + p * M;
        double* dev = ((pk & 1) ? dev_t : dev_b) + p * M;
        for (i64 j = j0 + threadIdx.x; j < j0 + 4096 && j < M; j += 256) {
            int lo = 0, hi = C;                          // chain of pooled position j: last c with off[c] <= j
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (off[mid] <= j) lo = mid; else hi = mid; }
            const double* cs = chstate + (pk * C + lo) * kChState;
            if (j - off[lo] < n) dev[j] = (cs[2] != 0.0) ? 0.0 : zdec(ztab, z[j], M) - cs[0];
        }
    }
}

// The tier-3 list in ascending pair order: one workgroup compacts the marks of all 2 P pairs.  grid 1, block 1024.
__global__ __launch_bounds__(1024) void k_long_list(const unsigned* __restrict__ more, const double* __restrict__ state, i64 P,
                                                   unsigned* __restrict__ long_count, unsigned* __restrict__ long_list)
{
    __shared__ unsigned s_wtot[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const i64 npk = 2 * P;
    unsigned base = 0;
    for (i64 k0 = 0; k0 < npk; k0 += 1024) {
        const i64 pk = k0 + tid;
        const bool listed = pk < npk && more[pk] != 0u && state[pk * kPairState + 3] == 0.0;
        const unsigned long long bal = __ballot(listed);
        if (lane == 0) s_wtot[w] = (unsigned)__popcll(bal);
        __syncthreads();
        unsigned before = base, total = 0;
        for (int ww = 0; ww < 16; ++ww) { const unsigned t = s_wtot[ww]; if (ww < w) before += t; total += t; }
        if (listed) long_list[before + (unsigned)__popcll(bal & ((1ull << lane) - 1ull))] = (unsigned)pk;
        base += total;
        __syncthreads();
    }
    if (tid == 0) long_count[0] = base;
}

// Tier 3 products.  grid (lag groups of the round, kLongSlots), block NT.  Workgroup (g, s) takes the listed pairs
// s, s + kLongSlots, ... and for each accumulates, over ALL chains and segments, the deviation products of lags
// [L0 + 256 g, L0 + 256 g + 256) in registers (4 blocks of 64 lags: one staging of the segment serves 256 lags),
// then writes acov[pair][lag] = sum_c sum_i d_c[i] d_c[i + lag] once.  Deterministic: fixed summation order.
template <int NT>
struct LongLds {
    static constexpr int SEG = kSeg;
    static constexpr int LA = (SEG + 16) / 8 * 10, LB = (SEG + kLongGroup + 16) / 8 * 10;
    double sA[LA];
    double sB[LB];
    double tot[64];
    double wred[NT / kWave * 64];
};

// One listed pair, one group of 256 lags [lbase, lbase + 256) (clipped to lend).  All NT threads call it.
template <int NT>
__device__ __forceinline__ void acov_long_pair(const double* __restrict__ dev, const i64* __restrict__ off, int C, i64 n,
                                               i64 lbase, i64 lend, double* __restrict__ out, LongLds<NT>& L)
{
    constexpr int SEG = kSeg;
    const int tid = threadIdx.x;
    double acc[4][8];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[b][i] = 0.0;
    for (int c = 0; c < C; ++c) {
        const double* dc = dev + off[c];
        for (i64 s0 = 0; s0 + lbase < n; s0 += SEG) {           // beyond that every product has a factor past the chain
            const int seglen = (int)((n - s0 < (i64)SEG) ? n - s0 : (i64)SEG);
            __syncthreads();
            for (int j = tid; j < SEG + 16; j += NT) { const i64 g = s0 + j; L.sA[pos8(j)] = (g < n) ? dc[g] : 0.0; }
            for (int j = tid; j < SEG + kLongGroup + 16; j += NT) { const i64 g = s0 + lbase + j; L.sB[pos8(j)] = (g < n) ? dc[g] : 0.0; }
            __syncthreads();
#pragma unroll
            for (int b = 0; b < 4; ++b) seg_accumulate<NT, 8>(L.sA, L.sB + 80 * b, seglen, acc[b]);     // 64 draws = 80 swizzled slots
        }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        seg_reduce<NT, 8>(acc[b], L.tot, L.wred);
        const i64 lag = lbase + 64 * b + tid;
        if (tid < 64 && lag < lend) out[lag] = L.tot[tid];
        __syncthreads();
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void k_acov_long(const double* __restrict__ dev_b, const double* __restrict__ dev_t,
                                                  i64 M, const i64* __restrict__ off, int C, i64 n, i64 L0, i64 L1,
                                                  const unsigned* __restrict__ long_count,
                                                  const unsigned* __restrict__ long_list,
                                                  const double* __restrict__ state, double* __restrict__ acov,
                                                  unsigned slot_from)
{
    __shared__ __attribute__((aligned(16))) LongLds<NT> L;
    const unsigned count = *long_count;
    const i64 lend = (L1 < n) ? L1 : n;
    const i64 lbase = L0 + (i64)kLongGroup * blockIdx.x;
    if (lbase >= lend) return;
    for (unsigned slot = slot_from + blockIdx.y; slot < count; slot += gridDim.y) {     // entries below slot_from: served by the FFT tier
        const i64 pk = long_list[slot];
        if (state[pk * kPairState + 3] != 0.0) continue;            // decided in an earlier round
        acov_long_pair<NT>(((pk & 1) ? dev_t : dev_b) + (pk >> 1) * M, off, C, n, lbase, lend, acov + pk * n, L);
    }
}

// Tier 3 scan of one round's lags [L0, min(L1, n)) for the listed pairs: first negative rho, ordered prefix sum.
// grid (kLongSlots), block 256.  Guard band as in tiers 1 and 2 (see ref_cov_sum): the scan looks for the first lag whose
// rho is below +band; inside (-band, +band) the lag is re-derived with _autocorr's own sums before it is compared with
// zero, and the scan resumes behind it when it is not negative.  A band lag costs one sequential pass over the chains
// (~0.15 ms for 32 768 draws, 4.5 ms for a million); at most guard_budget(n) of them are re-derived per pair and round,
// further ones are decided on the value they have.
struct ScanLds {
    double red[4];
    long long sfirst;
    GuardLds<kGuardChains, kGuardChunk> G;
};

// The scan of one listed pair over the lags [L0, lend).  All 256 threads call it.  z: the pair's rank codes (time order);
// chst: its chains' state records.  mark: set state[pk][3] when the pair is decided (the listed route's rounds test it;
// k_tier3 must NOT -- its workgroups build their lists from that word while other workgroups scan: ADVICE r3).
__device__ __forceinline__ bool long_scan_pair(i64 pk, int C, i64 n, i64 L0, i64 lend, double* __restrict__ state,
                                               double* __restrict__ acov, double* __restrict__ res, i64 P,
                                               const u32* __restrict__ z, const double* __restrict__ ztab, i64 M,
                                               double* __restrict__ chst, const i64* __restrict__ off, double band,
                                               unsigned* __restrict__ guard_count, bool mark, ScanLds& S)
{
    const int tid = threadIdx.x;
    double* stp = state + pk * kPairState;
    const double vhat = stp[2];
    double* a = acov + pk * n;
    // The products were written by OTHER workgroups of this launch (k_tier3: in stages, and this CU may have read a cache line
    // that straddles a stage boundary -- acov[pk] is line-aligned only when n is a multiple of 16 -- while scanning the stage in
    // front): they are read at agent scope, past the CU's own L1, whatever it holds.  (The acquire fence in front of a scan
    // invalidates that L1 already; this makes the scan independent of it.)
    auto ld = [&](i64 l) { return __hip_atomic_load(a + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    const double den = (double)C * vhat;
    const double thr = band * den;
    i64 from = L0, first = lend;
    int budget = guard_budget(n);
    for (;;) {
        if (tid == 0) S.sfirst = (long long)lend;
        __syncthreads();
        long long mine = (long long)lend;
        for (i64 l = from + tid; l < lend; l += 256) {
            // rho < band  <=>  acov < band * den * (n - l) (den > 0): no division in the search; where exactly the band's
            // top lies does not matter, every lag it catches is looked at again below
            if (ld(l) < thr * (double)(n - l)) { mine = l; break; }     // the thread's lags ascend: its first one below the band's top
        }
        if (mine < (long long)lend) atomicMin(&S.sfirst, mine);
        __syncthreads();
        first = (i64)S.sfirst;
        if (first >= lend) break;                                  // every remaining lag of the round is clearly positive
        const double rho_f = (ld(first) / (double)(n - first)) / den;
        if (rho_f < -band || budget == 0) {                        // clearly negative (or out of budget: decided as it stands)
            if (!(rho_f < 0.0)) { from = first + 1; __syncthreads(); continue; }
            break;
        }
        --budget;
        // ---- inside the band: the reference's own sum for this lag ----
        const double cov_sum = ref_cov_sum(z, ztab, M, off, C, n, first, chst, S.G);
        const double rho_x = cov_sum / den;
        if (tid == 0) {
            a[first] = cov_sum * (double)(n - first);      // what the prefix sum below adds for this lag (sign kept)
            atomicAdd(guard_count, 1u);
        }
        __syncthreads();
        if (rho_x < 0.0) break;                            // the reference's `if rho < 0: break`
        from = first + 1;                                  // zero or positive: the walk goes on
    }
    __syncthreads();
    double s = 0.0;
    for (i64 l = L0 + tid; l < first; l += 256) s += (ld(l) / (double)(n - l)) / den;
    s = block_sum<256>(s, S.red);
    if (tid == 0) {
        const double rho_sum = stp[0] + s;
        const double terms = stp[1] + (double)(first - L0);
        if (first < lend || lend >= n) {
            const i64 p = pk >> 1;
            const int kind = (int)(pk & 1);
            res[(kind ? R_ESS_TAIL : R_ESS_BULK) * P + p] = (double)((i64)C * n) / (1.0 + 2.0 * rho_sum);
            res[(kind ? R_LAG_TAIL : R_LAG_BULK) * P + p] = terms;
            if (mark) stp[3] = 1.0;
        } else {
            stp[0] = rho_sum; stp[1] = terms;
        }
    }
    __syncthreads();
    return first < lend || lend >= n;          // decided (uniform over the workgroup)
}

__global__ __launch_bounds__(256) void k_diag_long_scan(int C, i64 n, i64 L0, i64 L1,
                                                        const unsigned* __restrict__ long_count,
                                                        const unsigned* __restrict__ long_list,
                                                        double* __restrict__ state, double* __restrict__ acov,
                                                        double* __restrict__ res, i64 P,
                                                        const u32* __restrict__ zb, const u32* __restrict__ zt,
                                                        const double* __restrict__ ztab, double* __restrict__ chstate,
                                                        i64 M, const i64* __restrict__ off, double band,
                                                        unsigned* __restrict__ guard_count)
{
    __shared__ ScanLds S;
    const unsigned count = *long_count;
    const i64 lend = (L1 < n) ? L1 : n;
    for (unsigned slot = blockIdx.x; slot < count; slot += gridDim.x) {
        const i64 pk = long_list[slot];
        if (state[pk * kPairState + 3] != 0.0) continue;
        long_scan_pair(pk, C, n, L0, lend, state, acov, res, P, ((pk & 1) ? zt : zb) + (pk >> 1) * M, ztab, M,
                       chstate + pk * C * kChState, off, band, guard_count, true, S);
    }
}

// Tier 3 in ONE launch for chains of at most 16 384 draws (a single round of lags [256, n), no FFT) and at most
// kTier3MaxPairs pairs per chunk -- what every call of the C1 shape and of the packaged corpus takes, nearly always with
// nothing listed.  grid (lag groups, kLongSlots), block 256.  Every workgroup compacts the tier-3 marks of the 2 P
// pairs into its own LDS in ascending pair order (the list k_long_list would build; with nothing listed it is done
// after one round of loads), takes the products of its 256 lags for the pairs of its slot, and counts itself off per
// pair behind an agent-scope release; the workgroup that finishes a pair's last lag group acquires and runs the scan
// (first negative rho with the guard band, ordered prefix sum).  Three launches of round 2 -- list, products, scan --
// in one; the fences are paid by workgroups that have a listed pair to work on, nobody else.
constexpr int kTier3MaxPairs = 2048;
union Tier3Lds {
    LongLds<256> L;
    ScanLds S;
};
// ONE launch for the whole of tier 3 on chains of at most 16 384 draws and at most kTier3MaxPairs pairs per chunk -- what
// every call of the C1 shape and of the packaged corpus takes, with nothing listed more often than not.  grid: a FIXED number
// of workgroups (MCR_T3_WG), block 256.
//   * Every workgroup compacts the tier-3 marks of the 2 P pairs into its own LDS in ascending pair order (the list
//     k_long_list would build; with nothing listed it is done after one pass over the marks).
//   * The work items are (listed pair, group of 256 lags), handed out round-robin, STAGE by stage: stages of 2, 6, 8, 8, ...
//     groups (lags 256 .. 767, .. 2 303, then 2 048 each), and all items of stage s come before any of stage s + 1.
//   * The reference's walk stops at the first negative rho, so a pair's lags are scanned stage by stage, each scan as
//     soon as (a) the stage's products are complete and (b) the stage in front has been scanned without a decision.  Both
//     events are atomics on the pair's word of that stage -- the finisher of a group adds 1, the scanner of the stage in
//     front ORs in kT3Prev -- and whichever comes second sees the other in the value it gets back and runs the scan:
//     exactly once, nobody waits for anybody.  A scan that finds the first negative rho (or reaches the chain's end)
//     writes the pair's results and sets its `decided` word; items of later stages look at that word first and skip.
//   Round 3 computed all n - 256 lags of every listed pair, two pairs at a time: 122 pairs of an AR(0.99) model at the C1
//   shape -- walks that end by lag 1 673 -- took 12 ms, a pipelined call 6.7 ms; the packaged corpus' 34 listed pairs went
//   17 to a slot (tools/sticky_prof.py, tools/corpus_kprof.py).  The list a launch works on does not change under it
//   (ADVICE r3): it is built from `more` and `state[.][3]`, which nobody writes here.
__global__ __launch_bounds__(256) void k_tier3(const double* __restrict__ dev_b, const double* __restrict__ dev_t, i64 M,
                                               const i64* __restrict__ off, int C, i64 n, const unsigned* __restrict__ more,
                                               double* __restrict__ state, double* __restrict__ acov, double* __restrict__ res,
                                               i64 P, double band, unsigned* __restrict__ guard_count,
                                               unsigned* __restrict__ words, const u32* __restrict__ zb,
                                               const u32* __restrict__ zt, const double* __restrict__ ztab,
                                               double* __restrict__ chstate)
{
    __shared__ __attribute__((aligned(16))) Tier3Lds U;
    __shared__ unsigned short slist[kTier3MaxPairs];
    __shared__ unsigned s_wtot[4], s_flag;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const i64 npk = 2 * P;
    unsigned count = 0;
    for (i64 k0 = 0; k0 < npk; k0 += 256) {
        const i64 pk = k0 + tid;
        const bool listed = pk < npk && more[pk] != 0u && state[pk * kPairState + 3] == 0.0;
        const unsigned long long bal = __ballot(listed);
        if (lane == 0) s_wtot[w] = (unsigned)__popcll(bal);
        __syncthreads();
        unsigned before = count, total = 0;
        for (int ww = 0; ww < 4; ++ww) { const unsigned t = s_wtot[ww]; if (ww < w) before += t; total += t; }
        if (listed) slist[before + (unsigned)__popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)pk;
        count += total;
        __syncthreads();
    }
    if (count == 0) return;
    unsigned* decided = words + (i64)kT3Stages * npk;
    const int groups = (int)((n - kLag2 + kLongGroup - 1) / kLongGroup);            // <= kT3Stages * kT3StageGroups (n <= 16 384)
    int nstages = 1;
    while (t3_stage_first(nstages, groups) < groups) ++nstages;                             // <= kT3Stages
    auto stage_groups = [&](int st) { const int a = t3_stage_first(st, groups), b = t3_stage_first(st + 1, groups); return (b < groups ? b : groups) - a; };
    for (int st = 0; st < nstages; ++st) {
        const unsigned cnt = (unsigned)stage_groups(st);
        for (unsigned item = blockIdx.x; item < count * cnt; item += gridDim.x) {
            const unsigned slot = item / cnt;
            const i64 pk = slist[slot];
            if (tid == 0) s_flag = atomicOr(&decided[pk], 0u);                       // (read at the L2: another workgroup sets it)
            __syncthreads();
            const bool skip = s_flag != 0u;
            __syncthreads();
            if (skip) continue;
            const double* dev = ((pk & 1) ? dev_t : dev_b) + (pk >> 1) * M;
            const i64 lbase = kLag2 + (i64)kLongGroup * (t3_stage_first(st, groups) + (int)(item - slot * cnt));
            acov_long_pair<256>(dev, off, C, n, lbase, n, acov + pk * n, U.L);
            __syncthreads();                                  // every store of this workgroup's lags is issued
            if (tid == 0) {
                __threadfence();                              // agent-scope release of them
                const unsigned old = atomicAdd(&words[(i64)st * npk + pk], 1u);
                s_flag = ((old & 0xFFFFu) + 1u == cnt && (old & kT3Prev)) ? 1u : 0u;   // last group of the stage, and the stage in front is through
                if (s_flag) __threadfence();                  // acquire: the other groups' lags, the scan state of the stage in front
            }
            __syncthreads();
            int sc = st;                                      // this workgroup scans stage sc, and the following ones while they are ready
            while (s_flag) {
                const i64 L0 = kLag2 + (i64)kLongGroup * t3_stage_first(sc, groups);
                const i64 L1 = kLag2 + (i64)kLongGroup * t3_stage_first(sc + 1, groups);
                const i64 lend = (L1 < n) ? L1 : n;
                // (mark = false: state[pk][3] stays 0 for the whole launch, so every workgroup compacts the SAME list)
                const bool done = long_scan_pair(pk, C, n, L0, lend, state, acov, res, P, ((pk & 1) ? zt : zb) + (pk >> 1) * M, ztab, M,
                                                 chstate + pk * C * kChState, off, band, guard_count, false, U.S);
                if (tid == 0) {
                    if (done) { atomicExch(&decided[pk], 1u); s_flag = 0u; }
                    else {                                    // hand the walk to stage sc + 1: scan it now if its products are complete
                        __threadfence();                      // release: the partial sums in state[pk]
                        const unsigned old = atomicOr(&words[(i64)(sc + 1) * npk + pk], kT3Prev);
                        s_flag = ((old & 0xFFFFu) == (unsigned)stage_groups(sc + 1)) ? 1u : 0u;
                        if (s_flag) __threadfence();
                    }
                }
                __syncthreads();
                ++sc;
            }
            __syncthreads();
        }
    }
}

}  // namespace mcr
