// mcr_device.hpp -- device-side building blocks (gfx950 / CDNA4, wave64).
//
// Everything here is written for 64-lane wavefronts and fp64 VALU; nothing is shaped for MFMA
// (the path is sort / scan / reduce work: HBM- and LDS-bound).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mcr {

using i64 = long long;
using u32 = unsigned int;
using u64 = unsigned long long;

constexpr int kWave = 64;

// ---- reductions ----------------------------------------------------------------------------
// Cross-lane traffic goes through DPP (a VALU operand modifier: no LDS crossbar, no waitcnt) wherever the pattern is
// one DPP can express; __shfl_xor compiles to two ds_bpermute_b32 per double and step, which made the seven
// reductions of k_acov_seg's staging phase a quarter of that kernel.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                            __builtin_amdgcn_readlane(__double2loint(v), lane));
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppRor4 = 0x124, kDppRor8 = 0x128;   // quad_perm [1,0,3,2], [2,3,0,1], row_ror:4, :8

// Sum over the 16 lanes of every DPP row (lanes 16 r .. 16 r + 15); every lane of the row gets a row total (lanes of
// one row may differ in the last bit: the rotations associate the four quad sums in rotated order).
__device__ __forceinline__ double row_sum(double v)
{
    v += dpp_f64<kDppXor1>(v); v += dpp_f64<kDppXor2>(v); v += dpp_f64<kDppRor4>(v); v += dpp_f64<kDppRor8>(v);
    return v;
}
template <int CTRL>
__device__ __forceinline__ u32 dpp_u32(u32 v) { return (u32)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ u32 row_min_u32(u32 v)
{
    v = min(v, dpp_u32<kDppXor1>(v)); v = min(v, dpp_u32<kDppXor2>(v));
    v = min(v, dpp_u32<kDppRor4>(v)); v = min(v, dpp_u32<kDppRor8>(v));
    return v;
}
__device__ __forceinline__ u32 row_max_u32(u32 v)
{
    v = max(v, dpp_u32<kDppXor1>(v)); v = max(v, dpp_u32<kDppXor2>(v));
    v = max(v, dpp_u32<kDppRor4>(v)); v = max(v, dpp_u32<kDppRor8>(v));
    return v;
}
// All 64 lanes must be active.  Every lane gets the same value (the four row totals of lanes 0, 16, 32, 48).
__device__ __forceinline__ double wave_sum(double v)
{
    v = row_sum(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
// Exclusive prefix sums over the 64 lanes of a wave, plus the wave total: the six-step DPP scan (row_shr 1, 2, 4, 8
// inside the rows of 16, then row_bcast 15 / 31 carry the row totals into the rows above).  All 64 lanes active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64_or_zero(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);       // lanes without a source (shifted in, masked rows) get +0.0
}
__device__ __forceinline__ double wave_excl_scan(double v, double& total)
{
    double incl = v;
    incl += dpp_f64_or_zero<0x111, 0xf>(incl);     // row_shr:1
    incl += dpp_f64_or_zero<0x112, 0xf>(incl);     // row_shr:2
    incl += dpp_f64_or_zero<0x114, 0xf>(incl);     // row_shr:4
    incl += dpp_f64_or_zero<0x118, 0xf>(incl);     // row_shr:8
    incl += dpp_f64_or_zero<0x142, 0xa>(incl);     // row_bcast:15 -> rows 1, 3
    incl += dpp_f64_or_zero<0x143, 0xc>(incl);     // row_bcast:31 -> rows 2, 3
    total = readlane_f64(incl, 63);
    return incl - v;
}

// The value lane j (0..15) of the caller's row of 16 lanes holds: DPP row_newbcast (gfx90a and later).  The control is an
// immediate, so j must be a constant once the calling loop is unrolled.
__device__ __forceinline__ int row_lane(int v, int j)
{
#define MCR_RL(J) case J: return __builtin_amdgcn_update_dpp(0, v, 0x150 + J, 0xf, 0xf, false);
    switch (j & 15) {
        MCR_RL(0) MCR_RL(1) MCR_RL(2) MCR_RL(3) MCR_RL(4) MCR_RL(5) MCR_RL(6) MCR_RL(7)
        MCR_RL(8) MCR_RL(9) MCR_RL(10) MCR_RL(11) MCR_RL(12) MCR_RL(13) MCR_RL(14)
        default: return __builtin_amdgcn_update_dpp(0, v, 0x15F, 0xf, 0xf, false);
    }
#undef MCR_RL
}

// Integer scans over the 64 lanes of a wave on DPP (all lanes active).  OLD = the identity of the operation, which lanes
// without a source keep.
template <int CTRL, int ROW_MASK, int OLD>
__device__ __forceinline__ int dpp_i32_or(int v) { return __builtin_amdgcn_update_dpp(OLD, v, CTRL, ROW_MASK, 0xf, false); }

// exclusive prefix sums of one u32 per lane (+ the wave total): the six-step DPP scan on integers
__device__ __forceinline__ u32 wave_excl_scan_u32(u32 v, u32& total)
{
    int incl = (int)v;
    incl += dpp_i32_or<0x111, 0xf, 0>(incl); incl += dpp_i32_or<0x112, 0xf, 0>(incl);
    incl += dpp_i32_or<0x114, 0xf, 0>(incl); incl += dpp_i32_or<0x118, 0xf, 0>(incl);      // row_shr 1, 2, 4, 8
    incl += dpp_i32_or<0x142, 0xa, 0>(incl);                                               // row_bcast:15 -> rows 1, 3
    incl += dpp_i32_or<0x143, 0xc, 0>(incl);                                               // row_bcast:31 -> rows 2, 3
    total = (u32)__builtin_amdgcn_readlane(incl, 63);
    return (u32)incl - v;
}

// inclusive prefix maximum (values >= -1); `excl` <- the prefix maximum of the lanes before this one (-1 in lane 0)
__device__ __forceinline__ int wave_prefix_max(int v, int& excl)
{
    v = max(v, dpp_i32_or<0x111, 0xf, -1>(v)); v = max(v, dpp_i32_or<0x112, 0xf, -1>(v));
    v = max(v, dpp_i32_or<0x114, 0xf, -1>(v)); v = max(v, dpp_i32_or<0x118, 0xf, -1>(v));
    v = max(v, dpp_i32_or<0x142, 0xa, -1>(v)); v = max(v, dpp_i32_or<0x143, 0xc, -1>(v));
    excl = dpp_i32_or<0x138, 0xf, -1>(v);                       // wave_shr:1
    return v;
}
// inclusive suffix minimum; `excl` <- the suffix minimum of the lanes after this one (INT_MAX in lane 63).  There is no
// row broadcast towards lower lanes: the rows' minima (their lanes 0) come back as scalars.
__device__ __forceinline__ int wave_suffix_min(int v, int& excl)
{
    constexpr int kMax = 0x7fffffff;
    v = min(v, dpp_i32_or<0x101, 0xf, kMax>(v)); v = min(v, dpp_i32_or<0x102, 0xf, kMax>(v));
    v = min(v, dpp_i32_or<0x104, 0xf, kMax>(v)); v = min(v, dpp_i32_or<0x108, 0xf, kMax>(v));      // row_shl 1, 2, 4, 8
    const int r1 = __builtin_amdgcn_readlane(v, 16), r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    const int row = (threadIdx.x & 63) >> 4;
    const int m23 = min(r2, r3), m123 = min(r1, m23);
    const int above = (row == 0) ? m123 : ((row == 1) ? m23 : ((row == 2) ? r3 : kMax));
    v = min(v, above);
    excl = dpp_i32_or<0x130, 0xf, kMax>(v);                     // wave_shl:1
    return v;
}

// Deterministic block sum (fixed association order): wave shuffle tree, then waves in index order.
// `red` is LDS scratch of NT/64 doubles.  Every thread gets the result.
template <int NT>
__device__ __forceinline__ double block_sum(double v, double* red)
{
    v = wave_sum(v);
    __syncthreads();  // `red` may still be read from a previous call
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
#pragma unroll
    for (int w = 0; w < NT / kWave; ++w) r += red[w];
    return r;
}

// Three deterministic block sums with one barrier pair.  `red` holds 3 * NT/64 doubles.
template <int NT>
__device__ __forceinline__ void block_sum3(double& a, double& b, double& c, double* red)
{
    constexpr int NW = NT / kWave;
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        red[w] = a; red[NW + w] = b; red[2 * NW + w] = c;
    }
    __syncthreads();
    double ra = 0.0, rb = 0.0, rc = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { ra += red[w]; rb += red[NW + w]; rc += red[2 * NW + w]; }
    a = ra; b = rb; c = rc;
    __syncthreads();
}

// ---- AS241 (Wichura 1988) PPND16: z = Phi^-1(p) -----------------------------------------------
// Same evaluation order as statistics.NormalDist.inv_cdf, which the reference calls at
// src/mcmc_ref/diagnostics.py:131.  Contraction is disabled so the central branch
// (|p-0.5| <= 0.425, pure +,*,/) is bit-identical to the CPU evaluation.
__device__ __noinline__ double inv_cdf(double p)
{
#pragma clang fp contract(off)
    double q = p - 0.5, r, num, den, x;
    if (fabs(q) <= 0.425) {
        r = 0.180625 - q * q;
        num = (((((((2.5090809287301226727e+3 * r + 3.3430575583588128105e+4) * r +
                    6.7265770927008700853e+4) * r + 4.5921953931549871457e+4) * r +
                  1.3731693765509461125e+4) * r + 1.9715909503065514427e+3) * r +
                1.3314166789178437745e+2) * r + 3.3871328727963666080e+0) * q;
        den = (((((((5.2264952788528545610e+3 * r + 2.8729085735721942674e+4) * r +
                    3.9307895800092710610e+4) * r + 2.1213794301586595867e+4) * r +
                  5.3941960214247511077e+3) * r + 6.8718700749205790830e+2) * r +
                4.2313330701600911252e+1) * r + 1.0);
        return num / den;
    }
    r = (q <= 0.0) ? p : 1.0 - p;
    r = sqrt(-log(r));
    if (r <= 5.0) {
        r = r - 1.6;
        num = (((((((7.74545014278341407640e-4 * r + 2.27238449892691845833e-2) * r +
                    2.41780725177450611770e-1) * r + 1.27045825245236838258e+0) * r +
                  3.64784832476320460504e+0) * r + 5.76949722146069140550e+0) * r +
                4.63033784615654529590e+0) * r + 1.42343711074968357734e+0);
        den = (((((((1.05075007164441684324e-9 * r + 5.47593808499534494600e-4) * r +
                    1.51986665636164571966e-2) * r + 1.48103976427480074590e-1) * r +
                  6.89767334985100004550e-1) * r + 1.67638483018380384940e+0) * r +
                2.05319162663775882187e+0) * r + 1.0);
    } else {
        r = r - 5.0;
        num = (((((((2.01033439929228813265e-7 * r + 2.71155556874348757815e-5) * r +
                    1.24266094738807843860e-3) * r + 2.65321895265761230930e-2) * r +
                  2.96560571828504891230e-1) * r + 1.78482653991729133580e+0) * r +
                5.46378491116411436990e+0) * r + 6.65790464350110377720e+0);
        den = (((((((2.04426310338993978564e-15 * r + 1.42151175831644588870e-7) * r +
                    1.84631831751005468180e-5) * r + 7.86869131145613259100e-4) * r +
                  1.48753612908506148525e-2) * r + 1.36929880922735805310e-1) * r +
                5.99832206555887937690e-1) * r + 1.0);
    }
    x = num / den;
    return (q < 0.0) ? -x : x;
}

// ---- merge path --------------------------------------------------------------------------------
// Split point of diagonal `diag` for two ascending runs given by accessors a(i), b(j):
// returns ai (and bi = diag - ai) such that merging takes a[0..ai) and b[0..bi) first.  Ties
// take from `a` first.  Comparisons are on doubles, so -0.0 == +0.0 (as Python compares them).
template <class FA, class FB>
__device__ __forceinline__ i64 merge_path(FA a, i64 na, FB b, i64 nb, i64 diag)
{
    i64 lo = diag > nb ? diag - nb : 0;
    i64 hi = diag < na ? diag : na;
    while (lo < hi) {
        i64 mid = (lo + hi) >> 1;
        const auto ak = a(mid), bk = b(diag - 1 - mid);
        if (!(bk < ak)) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// 32-bit variant for searches over LDS (tile-local indices): the loop carries two ints instead of two 64-bit
// values -- half the selects per step of the per-thread partition search, which is ~1/3 of a merge level's VALU work.
template <class FA, class FB>
__device__ __forceinline__ int merge_path32(FA a, int na, FB b, int nb, int diag)
{
    int lo = diag > nb ? diag - nb : 0;
    int hi = diag < na ? diag : na;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const auto ak = a(mid), bk = b(diag - 1 - mid);      // doubles, or packed (key, position) records
        const bool right = !(bk < ak);
        lo = right ? mid + 1 : lo;
        hi = right ? hi : mid;
    }
    return lo;
}

// The same split point found by a whole wave: 64 probes per step instead of one, so a search over
// global memory costs ~log64(n) dependent round trips instead of log2(n).  All 64 lanes must call it
// with identical arguments; every lane returns the result.
template <class FA, class FB>
__device__ __forceinline__ i64 merge_path_wave(FA a, i64 na, FB b, i64 nb, i64 diag)
{
    const int lane = threadIdx.x & 63;
    i64 lo = diag > nb ? diag - nb : 0;
    i64 hi = diag < na ? diag : na;
    while (lo < hi) {
        const i64 span = hi - lo;
        const bool small = span <= 64;
        const i64 mid = small ? lo + lane : lo + (span * (lane + 1)) / 65;
        bool pred = false;
        if (!small || lane < span) pred = !(b(diag - 1 - mid) < a(mid));   // monotone: true ... true false ... false
        const int cnt = __popcll(__ballot(pred));
        if (small) { lo += cnt; hi = lo; }
        else {
            const i64 nlo = (cnt == 0) ? lo : lo + (span * (i64)cnt) / 65 + 1;
            const i64 nhi = (cnt == 64) ? hi : lo + (span * (i64)(cnt + 1)) / 65;
            lo = nlo; hi = nhi;
        }
    }
    return lo;
}

// LDS index swizzle: the low four bits of an element index are XORed with the next four, so that "thread t owns
// elements [16t, 16t+16)" accesses (element i of every thread at once) spread over all banks, and a run of 16
// consecutive elements still occupies its own 16 slots.  One instruction cheaper per access than the pad slot per
// 16 elements it replaced (arrays are still sized for that padding).
__device__ __forceinline__ int pos16(int e) { return e ^ ((e >> 4) & 15); }
// The same for the 2- and 4-byte position arrays that travel with the keys: with elements a quarter (half) as wide the
// chunk owners' lanes wrap around the banks four (two) times less often, so the XOR takes the bits one place higher --
// 2 LDS cycles per 64-lane access of a thread-owned chunk of 8 or 16 slots instead of the 4 that pos16 gives them.
__device__ __forceinline__ int posi(int e) { return e ^ ((e >> 5) & 15); }

// ---- f32 draws as packed sort records ------------------------------------------------------------
// A record is (order-preserving 32-bit image of the float) << 32 | pooled position: ONE 64-bit integer compare orders
// (value, position), one 8-byte LDS slot holds key and payload, and the widened f64 value -- all the statistics are
// computed on -- is recovered exactly from the key.  -0.0f is mapped onto +0.0f first (they tie, as in Python).
__device__ __forceinline__ u32 f32_key(float x)
{
    u32 b = __float_as_uint(x);
    if (x == 0.0f) b = 0u;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ double key_value(u32 k)
{
    const u32 b = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
    return (double)__uint_as_float(b);
}
constexpr u64 kRecPad = ~0ull;      // sorts behind every real record (a NaN's key is below 0xFFFFFFFF)

// Element i of a sorted key array as the f64 value the statistics use: plain doubles, or f32 records.
__device__ __forceinline__ double sorted_key(const double* a, i64 i) { return a[i]; }
__device__ __forceinline__ double sorted_key(const u64* a, i64 i) { return key_value((u32)(a[i] >> 32)); }

// ---- counter-based generator (bench / stress tensor) ------------------------------------------
__host__ __device__ __forceinline__ u64 splitmix64(u64 x)
{
    u64 z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

}  // namespace mcr
