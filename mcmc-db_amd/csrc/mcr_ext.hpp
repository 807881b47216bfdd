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
// A SYRK-shaped tiling for gfx950: a 256-thread workgroup owns a 128 x 128 tile of G (upper triangle only, the grid
// holds no workgroup below the diagonal; the finisher mirrors) and one slice of the draw axis.  Per step of 16 draws the
// two 128 x 16 operand panels go global -> registers (four 16-byte loads per lane and panel, a whole 128-byte line per
// row) -> centred -> LDS with a row stride of 18 doubles, which makes the one-double-per-lane operand fetch of the MFMA
// (lane 16 k + i reads row i, draw k) conflict-free.  Double buffered: the loads of panel s + 1 are in flight while the
// 64 MFMAs of panel s run.  Each of the 4 waves owns a 64 x 64 quadrant = 4 x 4 MFMA tiles (16 accumulators), so one
// operand register feeds four MFMAs: 8 LDS reads per 16 MFMAs, and a panel byte loaded from memory serves 128 rows of
// the other panel -- half the memory traffic per flop of the 64 x 64 tiles of round 2, whose 2 x 2 quadrants waited on
// their loads (35-46 % of the matrix peak).  74 KB of LDS: two workgroups per CU.
// grid (tiles on or above the diagonal, ksplit); partial[ks][P128][P128].
// ------------------------------------------------------------------------------------------------
constexpr int kCovBM = 128, kCovBK = 16, kCovLd = kCovBK + 2;

template <bool EVEN>     // EVEN: M even and X 16-byte aligned, so every row takes 16-byte loads
__global__ __launch_bounds__(256, 2) void k_cov_mfma(const double* __restrict__ X, const double* __restrict__ mean, i64 M,
                                                     i64 P, int nb, i64 kchunk, double* __restrict__ partial)
{
    typedef double v4d __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) double sA[2][kCovBM * kCovLd];
    __shared__ __attribute__((aligned(16))) double sB[2][kCovBM * kCovLd];
    // upper-triangle tile index -> (bi, bj), bi <= bj: row bi holds nb - bi tiles
    int bi = 0, rest = (int)blockIdx.x;
    while (rest >= nb - bi) { rest -= nb - bi; ++bi; }
    const int bj = bi + rest;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ks = blockIdx.y;
    const i64 t0 = (i64)ks * kchunk, t1 = (t0 + kchunk < M) ? t0 + kchunk : M;

    // staging role of this lane: row lr of both panels, draws 8 lh .. 8 lh + 7 of the step
    const int lr = tid >> 1, lh = tid & 1;
    const i64 rowa = (i64)bi * kCovBM + lr, rowb = (i64)bj * kCovBM + lr;
    const bool va = rowa < P, vb = rowb < P;
    const double* pa = X + (va ? rowa : 0) * M;
    const double* pb = X + (vb ? rowb : 0) * M;
    const double ma = va ? mean[rowa] : 0.0, mb = vb ? mean[rowb] : 0.0;
    double ra[8], rb[8];
    i64 cur = 0;
    // Loads are unconditional (addresses clamped into the row, invalid rows read row 0) and masked afterwards: no branch
    // sits between a load and the next one, so all 8 of a step are in flight together -- and they stay in flight during
    // the MFMAs of the current panel: `fetch` only issues them, `stash` (after the MFMAs) centres, masks and writes them to
    // the other LDS buffer.
    auto fetch = [&](i64 t) {
        cur = t + 8 * lh;
        if constexpr (EVEN) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const i64 c = cur + 2 * u, cc = (c + 1 < M) ? c : M - 2;
                const double2 x = *reinterpret_cast<const double2*>(pa + cc), y = *reinterpret_cast<const double2*>(pb + cc);
                ra[2 * u] = x.x; ra[2 * u + 1] = x.y; rb[2 * u] = y.x; rb[2 * u + 1] = y.y;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) { const i64 c = (cur + u < M) ? cur + u : M - 1; ra[u] = pa[c]; rb[u] = pb[c]; }
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the uses of these registers behind the MFMAs that follow
    };
    auto stash = [&](int buf) {
        __builtin_amdgcn_sched_barrier(0);
        double xa[8], xb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool in = cur + u < t1;
            xa[u] = (va && in) ? ra[u] - ma : 0.0;
            xb[u] = (vb && in) ? rb[u] - mb : 0.0;
        }
        const int o = lr * kCovLd + 8 * lh;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            *reinterpret_cast<double2*>(&sA[buf][o + 2 * u]) = make_double2(xa[2 * u], xa[2 * u + 1]);
            *reinterpret_cast<double2*>(&sB[buf][o + 2 * u]) = make_double2(xb[2 * u], xb[2 * u + 1]);
        }
    };

    const int wr = (w >> 1) * 64, wc = (w & 1) * 64;        // this wave's quadrant
    const int oi = lane & 15, ok = lane >> 4;                 // operand row / draw of this lane
    v4d acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = v4d{0, 0, 0, 0};
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
            double av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { av[a] = A[(wr + 16 * a + oi) * kCovLd + c]; bv[a] = B[(wc + 16 * a + oi) * kCovLd + c]; }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // result layout measured on gfx950 (tools/ubench/mfma64_layout.hip): D[(lane/16) + 4*v][lane%16] = acc[v]
    const i64 PB = (i64)nb * kCovBM;
    double* out = partial + (i64)ks * PB * PB;
    const i64 r0 = (i64)bi * kCovBM + wr + ok, c0 = (i64)bj * kCovBM + wc + oi;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int v = 0; v < 4; ++v) out[(r0 + 16 * a + 4 * v) * PB + c0 + 16 * b] = acc[a][b][v];
}

// cov[i][j] = sum over draw slices of G[min(i,j)-block-ordered entry] / M (only tiles on or above the diagonal exist).
__global__ void k_cov_final(const double* __restrict__ partial, int ksplit, int nb, i64 M, i64 P, double* __restrict__ cov)
{
    const i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P * P) return;
    i64 i = idx / P, j = idx % P;
    if (j / kCovBM < i / kCovBM) { const i64 t = i; i = j; j = t; }       // below the block diagonal: read the mirrored entry
    const i64 PB = (i64)nb * kCovBM;
    double g = 0.0;
    for (int ks = 0; ks < ksplit; ++ks) g += partial[(i64)ks * PB * PB + i * PB + j];
    cov[idx] = g / (double)M;
}

}  // namespace mcr
