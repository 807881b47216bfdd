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

// Non-finite draws seen by k_tile_sort (fourth entry of its per-tile partials), summed per parameter.
__global__ void k_bad_count(const double* __restrict__ part, int ntiles, i64 P, double* __restrict__ bad)
{
    const i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    double b = 0.0;
    for (int t = 0; t < ntiles; ++t) b += part[(p * ntiles + t) * kMomRec + 3];
    bad[p] = b;
}

// ------------------------------------------------------------------------------------------------
// Parameter covariance (population, ddof = 0, like compare.py:63): the one dense contraction of the path, so
// the one place the matrix cores are used.  G = (X - mu)(X - mu)^T with X [P][M] row-major (mu_p from the streaming
// moments kernel, so G / M IS the covariance: no correction term), v_mfma_f64_16x16x4f64.
//
// A SYRK-shaped tiling for gfx950: a 256-thread workgroup owns a 64 x 64 tile of G (upper triangle only; the finisher
// mirrors it) and one slice of the draw axis.  Per step of 32 draws the two 64 x 32 operand panels go global -> registers
// (two 8-byte loads per lane, 256 contiguous bytes per row: coalesced) -> centred -> LDS with a row stride of 34 doubles,
// which makes the one-double-per-lane operand fetch of the MFMA (lane 16k + i reads row i, draw k) conflict-free
// (16 rows x 272 bytes hit 16 different 8-byte bank pairs).  Double buffered: the loads of panel s + 1 are in flight while
// the 32 MFMAs of panel s run.  Each of the 4 waves owns a 32 x 32 quadrant = 2 x 2 MFMA tiles, so every operand
// register feeds two MFMAs (4 LDS reads per 4 MFMAs per wave and step of 4 draws).
// grid (nb * nb, ksplit) with nb = ceil(P / 64); workgroups below the diagonal exit at once.  partial[ks][P64][P64].
// ------------------------------------------------------------------------------------------------
constexpr int kCovBM = 64, kCovBK = 32, kCovLd = kCovBK + 2;

template <bool EVEN>     // EVEN: M even and X 16-byte aligned, so every row takes 16-byte loads
__global__ __launch_bounds__(256) void k_cov_mfma(const double* __restrict__ X, const double* __restrict__ mean, i64 M,
                                                  i64 P, int nb, i64 kchunk, double* __restrict__ partial)
{
    typedef double v4d __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) double sA[2][kCovBM * kCovLd];
    __shared__ __attribute__((aligned(16))) double sB[2][kCovBM * kCovLd];
    const int bi = blockIdx.x / nb, bj = blockIdx.x % nb;
    if (bj < bi) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ks = blockIdx.y;
    const i64 t0 = (i64)ks * kchunk, t1 = (t0 + kchunk < M) ? t0 + kchunk : M;

    // staging role of this lane: rows lr + 16 u (u = 0..3) of both panels, draws 2 lc, 2 lc + 1 of the step
    const int lr = tid >> 4, lc = tid & 15;
    const double* pa[4]; const double* pb[4];
    double ma[4], mb[4];
    bool va[4], vb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const i64 ra = (i64)bi * kCovBM + lr + 16 * u, rb = (i64)bj * kCovBM + lr + 16 * u;
        va[u] = ra < P; vb[u] = rb < P;
        pa[u] = X + (va[u] ? ra : 0) * M; pb[u] = X + (vb[u] ? rb : 0) * M;
        ma[u] = va[u] ? mean[ra] : 0.0; mb[u] = vb[u] ? mean[rb] : 0.0;
    }
    double ra_[4][2], rb_[4][2];
    // Loads are unconditional (addresses clamped into the row, invalid rows read row 0) and masked afterwards: no
    // branch sits between a load and the next one, so all 8 of a step are in flight together -- and they stay in
    // flight during the MFMAs of the current panel: `fetch` only issues them, `stash` (after the MFMAs) centres, masks
    // and writes them to the other LDS buffer.
    bool in0 = false, in1 = false;
    auto fetch = [&](i64 t) {
        const i64 c = t + 2 * lc;
        in0 = c < t1; in1 = c + 1 < t1;
        if constexpr (EVEN) {
            const i64 cc = (c + 1 < M) ? c : M - 2;
#pragma unroll
            for (int u = 0; u < 4; ++u) { const double2 v = *reinterpret_cast<const double2*>(pa[u] + cc); ra_[u][0] = v.x; ra_[u][1] = v.y; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const double2 v = *reinterpret_cast<const double2*>(pb[u] + cc); rb_[u][0] = v.x; rb_[u][1] = v.y; }
        } else {
            const i64 c0 = (c < M) ? c : M - 1, c1 = (c + 1 < M) ? c + 1 : M - 1;
#pragma unroll
            for (int u = 0; u < 4; ++u) { ra_[u][0] = pa[u][c0]; ra_[u][1] = pa[u][c1]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { rb_[u][0] = pb[u][c0]; rb_[u][1] = pb[u][c1]; }
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the uses of these registers behind the MFMAs that follow
    };
    auto stash = [&](int buf) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int o = (lr + 16 * u) * kCovLd + 2 * lc;
            *reinterpret_cast<double2*>(&sA[buf][o]) = make_double2((va[u] && in0) ? ra_[u][0] - ma[u] : 0.0,
                                                                    (va[u] && in1) ? ra_[u][1] - ma[u] : 0.0);
            *reinterpret_cast<double2*>(&sB[buf][o]) = make_double2((vb[u] && in0) ? rb_[u][0] - mb[u] : 0.0,
                                                                    (vb[u] && in1) ? rb_[u][1] - mb[u] : 0.0);
        }
    };

    const int wr = (w >> 1) * 32, wc = (w & 1) * 32;        // this wave's quadrant
    const int oi = lane & 15, ok = lane >> 4;                 // operand row / draw of this lane
    v4d acc00 = {0, 0, 0, 0}, acc01 = {0, 0, 0, 0}, acc10 = {0, 0, 0, 0}, acc11 = {0, 0, 0, 0};
    int buf = 0;
    if (t0 < t1) { fetch(t0); stash(0); }
    __syncthreads();
    for (i64 t = t0; t < t1; t += kCovBK) {
        const bool more = t + kCovBK < t1;
        fetch(more ? t + kCovBK : t);                          // in flight during the MFMAs below (the last step re-reads its own panel: no branch)
        const double* A = sA[buf];
        const double* B = sB[buf];
#pragma unroll
        for (int kk = 0; kk < kCovBK / 4; ++kk) {
            const int c = kk * 4 + ok;
            const double a0 = A[(wr + oi) * kCovLd + c], a1 = A[(wr + 16 + oi) * kCovLd + c];
            const double b0 = B[(wc + oi) * kCovLd + c], b1 = B[(wc + 16 + oi) * kCovLd + c];
            acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc11, 0, 0, 0);
        }
        stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // result layout measured on gfx950 (tools/ubench/mfma64_layout.hip): D[(lane/16) + 4*v][lane%16] = acc[v]
    const i64 P64 = (i64)nb * kCovBM;
    double* out = partial + (i64)ks * P64 * P64;
    const i64 r0 = (i64)bi * kCovBM + wr + ok, c0 = (i64)bj * kCovBM + wc + oi;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        out[(r0 + 4 * v) * P64 + c0] = acc00[v];
        out[(r0 + 4 * v) * P64 + c0 + 16] = acc01[v];
        out[(r0 + 16 + 4 * v) * P64 + c0] = acc10[v];
        out[(r0 + 16 + 4 * v) * P64 + c0 + 16] = acc11[v];
    }
}

// cov[i][j] = sum over draw slices of G[min(i,j)-block-ordered entry] / M (only tiles on or above the diagonal exist).
__global__ void k_cov_final(const double* __restrict__ partial, int ksplit, int nb, i64 M, i64 P, double* __restrict__ cov)
{
    const i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P * P) return;
    i64 i = idx / P, j = idx % P;
    if (j / kCovBM < i / kCovBM) { const i64 t = i; i = j; j = t; }       // below the block diagonal: read the mirrored entry
    const i64 P64 = (i64)nb * kCovBM;
    double g = 0.0;
    for (int ks = 0; ks < ksplit; ++ks) g += partial[(i64)ks * P64 * P64 + i * P64 + j];
    cov[idx] = g / (double)M;
}

}  // namespace mcr
