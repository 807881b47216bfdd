// mcr_ext.hpp -- extensions named in the north star but ABSENT from the reference (SURVEY.md 8(a)
// rows X1-X3): two-sample Kolmogorov-Smirnov statistic, Wasserstein-1 distance and the parameter
// covariance matrix.  PARITY UNPINNED by the reference; the tests pin them to scipy.stats.ks_2samp /
// scipy.stats.wasserstein_distance / numpy.cov, whose published definitions are restated here.
#pragma once
#include "mcr_device.hpp"

namespace mcr {

// ------------------------------------------------------------------------------------------------
// KS + Wasserstein-1 from the two ascending samples r[Mr], a[Ma] of one parameter: one merge-path
// pass over the pooled order.  At every pooled position t that ends a run of equal values, with
// i = #(r <= v_t), j = #(a <= v_t):    D = max |i/Mr - j/Ma|     W1 += |i/Mr - j/Ma| (v_{t+1} - v_t)
// (scipy: searchsorted(..., side="right") / n on the pooled values; deltas = diff(sorted pooled)).
// D is formed exactly: max |i*Ma - j*Mr| as an integer, divided once by Mr*Ma (scipy's exact mode rounds
// its statistic to the same rational).
// grid (nblk, P); part[(p*nblk + blk)*2 + {0,1}] = block max, block sum.
// ------------------------------------------------------------------------------------------------
template <int NT, int VT>
__global__ __launch_bounds__(NT) void k_two_sample(const double* __restrict__ rs, i64 Mr,
                                                   const double* __restrict__ as, i64 Ma,
                                                   double* __restrict__ part, int nblk)
{
    constexpr int OB = NT * VT;
    __shared__ double sk[OB + 2 + (OB + 2) / 16 + 1];
    __shared__ i64 sh[2];
    __shared__ double red[2 * NT / 64];
    const int tid = threadIdx.x, blk = blockIdx.x;
    const i64 p = blockIdx.y;
    const double* A = rs + p * Mr;
    const double* B = as + p * Ma;
    const i64 tot = Mr + Ma;
    const i64 d0 = (i64)blk * OB;
    const i64 d1 = (d0 + OB < tot) ? d0 + OB : tot;
    auto GA = [&](i64 i) { return A[i]; };
    auto GB = [&](i64 j) { return B[j]; };
    if (tid < 64) { const i64 r0 = merge_path_wave(GA, Mr, GB, Ma, d0); if (tid == 0) sh[0] = r0; }
    else if (tid < 128) { const i64 r1 = merge_path_wave(GA, Mr, GB, Ma, d1); if (tid == 64) sh[1] = r1; }
    __syncthreads();
    const i64 ai0 = sh[0], ai1 = sh[1], bi0 = d0 - ai0, bi1 = d1 - ai1;
    // pieces + one look-ahead draw each (the successor of the block's last value)
    const int ca = (int)(ai1 - ai0) + (ai1 < Mr ? 1 : 0), cb = (int)(bi1 - bi0) + (bi1 < Ma ? 1 : 0);
    for (int e = tid; e < ca + cb; e += NT) sk[pos16(e)] = (e < ca) ? A[ai0 + e] : B[bi0 + (e - ca)];
    __syncthreads();
    const int total = (int)(d1 - d0);
    const int diag = (tid * VT < total) ? tid * VT : total;
    const int nout = (total - diag < VT) ? total - diag : VT;
    auto LA = [&](int i) { return sk[pos16(i)]; };
    auto LB = [&](int j) { return sk[pos16(ca + j)]; };
    // merge path over the pieces WITHOUT the look-ahead draws (they belong to later blocks)
    const int na = (int)(ai1 - ai0), nb = (int)(bi1 - bi0);
    int ai = merge_path32(LA, na, LB, nb, diag);
    int bi = diag - ai;
    double w = 0.0;
    i64 ksn = 0;   // max |i*Ma - j*Mr|: the KS statistic is this integer over Mr*Ma (an exact rational)
    double ak = (ai < ca) ? sk[pos16(ai)] : 0.0, bk = (bi < cb) ? sk[pos16(ca + bi)] : 0.0;
    for (int i = 0; i < nout; ++i) {
        const bool takeA = (bi >= nb) || (ai < na && !(bk < ak));
        const double v = takeA ? ak : bk;
        if (takeA) { ++ai; ak = (ai < ca) ? sk[pos16(ai)] : 0.0; }
        else       { ++bi; bk = (bi < cb) ? sk[pos16(ca + bi)] : 0.0; }
        // successor in the pooled order (look-ahead draws included); none at the very end
        const bool hasA = ai < ca, hasB = bi < cb;
        const bool last = !hasA && !hasB;
        const double nxt = (!hasB || (hasA && !(bk < ak))) ? ak : bk;
        if (last || nxt != v) {
            const i64 ci = ai0 + ai, cj = bi0 + bi;
            const i64 num = ci * Ma - cj * Mr;
            ksn = max(ksn, num < 0 ? -num : num);
            const double diff = fabs((double)ci / (double)Mr - (double)cj / (double)Ma);
            if (!last) w = fma(diff, nxt - v, w);
        }
    }
    // block reduce
    double ks = (double)ksn;   // exact: the numerator is below 2^53 (checked on the host)
    for (int o = 32; o > 0; o >>= 1) { ks = fmax(ks, __shfl_xor(ks, o, kWave)); w += __shfl_xor(w, o, kWave); }
    if ((tid & 63) == 0) { red[tid >> 6] = ks; red[NT / 64 + (tid >> 6)] = w; }
    __syncthreads();
    if (tid == 0) {
        double m = 0.0, s = 0.0;
        for (int ww = 0; ww < NT / 64; ++ww) { m = fmax(m, red[ww]); s += red[NT / 64 + ww]; }
        part[(p * nblk + blk) * 2] = m;
        part[(p * nblk + blk) * 2 + 1] = s;
    }
}

__global__ void k_two_sample_final(const double* __restrict__ part, int nblk, i64 P, double denom,
                                   double* __restrict__ ks, double* __restrict__ w1)
{
    const i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    double m = 0.0, s = 0.0;
    for (int b = 0; b < nblk; ++b) { m = fmax(m, part[(p * nblk + b) * 2]); s += part[(p * nblk + b) * 2 + 1]; }
    ks[p] = m / denom;   // one correctly rounded division of the exact rational
    w1[p] = s;
}

// Non-finite draws seen by k_tile_sort (third entry of its per-tile partials), summed per parameter.
__global__ void k_bad_count(const double* __restrict__ part, int ntiles, i64 P, double* __restrict__ bad)
{
    const i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    double b = 0.0;
    for (int t = 0; t < ntiles; ++t) b += part[(p * ntiles + t) * 4 + 2];
    bad[p] = b;
}

// ------------------------------------------------------------------------------------------------
// Parameter covariance (population, ddof = 0, like compare.py:63): the one dense contraction of the
// path, so the one place the matrix cores are used.  G = (X - K)(X - K)^T with X [P][M] (K_p = first
// draw of parameter p, a shift against cancellation), v_mfma_f64_16x16x4f64: one wave owns a 16x16
// tile of G and a slice of the draw axis; cov = G/M - (mu - K)(mu - K)^T is applied by the finisher.
// grid (tiles_i * tiles_j, ksplit); block 64.  partial[ks][P16][P16].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_cov_mfma(const double* __restrict__ X, i64 M, i64 P, int tiles,
                                                 i64 kchunk, double* __restrict__ partial)
{
    typedef double v4d __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x;
    const int ti = blockIdx.x / tiles, tj = blockIdx.x % tiles;
    const int ks = blockIdx.y;
    const int r = lane & 15, kq = lane >> 4;               // operand row / k index of this lane
    const i64 pi = (i64)ti * 16 + r, pj = (i64)tj * 16 + r;
    const double* xi = X + (pi < P ? pi : 0) * M;
    const double* xj = X + (pj < P ? pj : 0) * M;
    const double Ki = xi[0], Kj = xj[0];
    const i64 t0 = (i64)ks * kchunk, t1 = (t0 + kchunk < M) ? t0 + kchunk : M;
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    for (i64 t = t0; t < t1; t += 4) {
        const i64 tt = t + kq;
        const double a = (pi < P && tt < t1) ? xi[tt] - Ki : 0.0;    // A[row r][k kq]
        const double b = (pj < P && tt < t1) ? xj[tt] - Kj : 0.0;    // B[k kq][col r]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    // result layout measured on gfx950 (tools/ubench/mfma64_layout.hip): D[(lane/16) + 4*i][lane%16] = acc[i]
    const i64 P16 = (i64)tiles * 16;
    double* out = partial + (i64)ks * P16 * P16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const i64 row = (i64)ti * 16 + kq + 4 * i, col = (i64)tj * 16 + r;
        out[row * P16 + col] = acc[i];
    }
}

__global__ void k_cov_final(const double* __restrict__ partial, int ksplit, int tiles, const double* __restrict__ X,
                            i64 M, i64 P, const double* __restrict__ mean, double* __restrict__ cov)
{
    const i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P * P) return;
    const i64 i = idx / P, j = idx % P;
    const i64 P16 = (i64)tiles * 16;
    double g = 0.0;
    for (int ks = 0; ks < ksplit; ++ks) g += partial[(i64)ks * P16 * P16 + i * P16 + j];
    const double di = mean[i] - X[i * M], dj = mean[j] - X[j * M];
    cov[idx] = g / (double)M - di * dj;
}

}  // namespace mcr
