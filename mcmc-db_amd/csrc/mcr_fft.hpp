// mcr_fft.hpp -- autocovariances of LONG chains by FFT (tier 3 of the ESS lags for chains of more than 16 384 draws).
//
// Spec: _autocorr of src/mcmc_ref/diagnostics.py:180-193 -- for every lag the sum over chains of
// sum_i (z_i - m)(z_{i+lag} - m) -- which the reference walks lag by lag until the first negative rho (:171-177).
// For a sticky chain of n = 100 000 draws that walk is thousands of lags long: the direct products of k_acov_long
// cost O(n x lags) per (parameter, kind) pair, an FFT costs O(n log n) for ALL lags: zero-pad the deviations of each
// chain to N >= 2n (no circular wrap), |FFT|^2 summed over the chains, one more FFT of that real, even spectrum, and
// acov[lag] = Re(result)[lag] / N.  The scan for the first negative rho (k_diag_long_scan) is the same as for the
// direct tiers; the integer truncation lag is exact as long as no rho lies within the FFT's round-off (~1e-14 of
// rho_0) of zero, which the parity tests check on random walks against the CPU restatement of the reference.
//
// Layout (four-step, N = N1 x N2, both powers of two <= 2048, every sub-transform in LDS):
// Two chains share one complex transform (real part: chain a, imaginary part: chain b; see k_fft_cols).
//   FFT 1 (natural in, transposed out):  X[k1 + N1 k2] = sum_n2 W_N^(n2 k1) [sum_n1 x[n1 N2 + n2] W_N1^(n1 k1)] W_N2^(n2 k2)
//     k_fft_cols : per tile of columns n2, length-N1 transforms over n1 (+ twiddle W_N^(n2 k1))  -> A[c][k1][n2]
//     k_fft_rows_power : per row k1, length-N2 transforms over n2, |.|^2 summed over chains       -> S[k1][k2]
//   FFT 2 (transposed in, natural out):   Y[N2 k1 + k2] = sum_n1 W_N^(n1 k2) W_N1^(n1 k1) [sum_n2 S[n1][n2] W_N2^(n2 k2)]
//     k_fft_rows_spec : per row n1 (= the k1 above), length-N2 transform (+ twiddle W_N^(n1 k2))  -> B[n1][k2]
//     k_fft_cols_out : per tile of columns k2, length-N1 transforms over n1 -> acov[N2 k1 + k2] = Re / N
// Only (parameter, kind) pairs on the tier-3 list do any work: grid slot s serves list entry s (the first `slots`
// entries; the kernels read the list length on the device and exit at once beyond it; entries past the slots of a
// launch are served by the direct products of k_acov_long).
#pragma once
#include "mcr_diag.hpp"

namespace mcr {
namespace fft {

constexpr int kMaxSub = 2048;        // longest sub-transform (32 KB of LDS as complex fp64)
constexpr int kColElems = 4096;      // complex elements a column kernel holds in LDS: COLS = kColElems / N1 columns at a time

__device__ __forceinline__ int bitrev(int x, int bits) { return (int)(__brev((unsigned)x) >> (32 - bits)); }

// tw[t] = exp(-2 pi i t / L), t < L / 2
__global__ __launch_bounds__(256) void k_fft_twiddles(double2* __restrict__ tw, int L)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= L / 2) return;
    double s, c;
    sincospi(-2.0 * (double)t / (double)L, &s, &c);
    tw[t] = make_double2(c, s);
}

// F independent forward transforms of length L = 2^logL held in LDS as buf[f * L + i]; radix-2 decimation in
// frequency, in place: X[k] ends up at position bitrev(k).  All NT threads call it; ends with a barrier.
template <int NT>
__device__ __forceinline__ void lds_fft(double2* buf, int F, int logL, const double2* __restrict__ tw)
{
    const int L = 1 << logL, halfL = L >> 1;
    const int total = F * halfL;
    for (int s = 0; s < logL; ++s) {
        const int logh = logL - 1 - s, h = 1 << logh;
        for (int b = threadIdx.x; b < total; b += NT) {
            const int f = b >> (logL - 1), r = b & (halfL - 1);
            const int g = r >> logh, j = r & (h - 1);
            const int i = (f << logL) + (g << (logh + 1)) + j;
            const double2 a = buf[i], c = buf[i + h];
            const double2 w = tw[j << s];                       // W_{2h}^j = W_L^(j L / 2h)
            const double dx = a.x - c.x, dy = a.y - c.y;
            buf[i] = make_double2(a.x + c.x, a.y + c.y);
            buf[i + h] = make_double2(dx * w.x - dy * w.y, dx * w.y + dy * w.x);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ double2 big_twiddle(i64 m, i64 N)      // exp(-2 pi i m / N), m taken mod N
{
    double s, c;
    sincospi(-2.0 * (double)(m & (N - 1)) / (double)N, &s, &c);
    return make_double2(c, s);
}

struct Plan { int log1, log2; };     // N1 = 2^log1, N2 = 2^log2, N = N1 N2

// grid (N2 / COLS, chains of the batch, slots).  dev: deviations of the listed pair (chain c at off[c], n draws).
template <int NT>
__global__ __launch_bounds__(NT) void k_fft_cols(const double* __restrict__ dev_b, const double* __restrict__ dev_t, i64 M,
                                                 const i64* __restrict__ off, int c0, int C, i64 n, Plan pl,
                                                 const double2* __restrict__ tw1, const unsigned* __restrict__ long_count,
                                                 const unsigned* __restrict__ long_list, const double* __restrict__ state,
                                                 double2* __restrict__ A, int CB)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2* buf = reinterpret_cast<double2*>(smem);
    const int N1 = 1 << pl.log1, N2 = 1 << pl.log2;
    const i64 N = (i64)N1 * N2;
    const int COLS = kColElems >> pl.log1;
    const int col0 = blockIdx.x * COLS, cb = blockIdx.y;
    const unsigned slot = blockIdx.z;
    if (slot >= *long_count) return;
    {
        const i64 pk = long_list[slot];
        if (state[pk * kPairState + 3] != 0.0) return;
        // TWO chains per complex transform: z = d_a + i d_b.  |FFT z|^2 = |A|^2 + |B|^2 + (a real, ODD cross term), and the
        // inverse transform of a real odd sequence is purely imaginary, so Re IFFT(|FFT z|^2) = acov_a + acov_b: the
        // sum over chains that the spectrum is summed to anyway, at half the forward transforms and with no untangling.
        const int ca = c0 + 2 * cb;
        const double* dpair = ((pk & 1) ? dev_t : dev_b) + (pk >> 1) * M;
        const double* da = dpair + off[ca];
        const double* db = (ca + 1 < C) ? dpair + off[ca + 1] : nullptr;
        for (int e = threadIdx.x; e < COLS * N1; e += NT) {
            const int n1 = e / COLS, cl = e - n1 * COLS;
            const i64 g = (i64)n1 * N2 + col0 + cl;
            buf[(cl << pl.log1) + n1] = make_double2(g < n ? da[g] : 0.0, (db != nullptr && g < n) ? db[g] : 0.0);
        }
        __syncthreads();
        lds_fft<NT>(buf, COLS, pl.log1, tw1);
        double2* out = A + ((i64)slot * CB + cb) * N;
        for (int e = threadIdx.x; e < COLS * N1; e += NT) {
            const int k1 = e / COLS, cl = e - k1 * COLS;
            const double2 v = buf[(cl << pl.log1) + bitrev(k1, pl.log1)];
            const double2 w = big_twiddle((i64)(col0 + cl) * k1, N);
            out[(i64)k1 * N2 + col0 + cl] = make_double2(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
        }
    }
}

// grid (N1, slots): row k1 of the CB chains of the batch -> S[k1][.] (+)= sum_c |FFT_N2(A_c[k1][.])|^2, kept in the
// bit-reversed position order of the in-place transform (k_fft_rows_spec undoes it).
template <int NT>
__global__ __launch_bounds__(NT) void k_fft_rows_power(const double2* __restrict__ A, Plan pl, const double2* __restrict__ tw2,
                                                       const unsigned* __restrict__ long_count,
                                                       const unsigned* __restrict__ long_list,
                                                       const double* __restrict__ state, double* __restrict__ S, int CB,
                                                       int nchains, int first_batch)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2* buf = reinterpret_cast<double2*>(smem);
    const int N2 = 1 << pl.log2;
    const i64 N = (i64)N2 << pl.log1;
    const int k1 = blockIdx.x;
    constexpr int PER = kMaxSub / NT;
    const unsigned slot = blockIdx.y;
    if (slot >= *long_count) return;
    {
        const i64 pk = long_list[slot];
        if (state[pk * kPairState + 3] != 0.0) return;
        double acc[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) acc[u] = 0.0;
        for (int cb = 0; cb < nchains; ++cb) {
            const double2* row = A + ((i64)slot * CB + cb) * N + (i64)k1 * N2;
            __syncthreads();
            for (int e = threadIdx.x; e < N2; e += NT) buf[e] = row[e];
            __syncthreads();
            lds_fft<NT>(buf, 1, pl.log2, tw2);
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int e = threadIdx.x + u * NT;
                if (e < N2) { const double2 v = buf[e]; acc[u] = fma(v.x, v.x, fma(v.y, v.y, acc[u])); }
            }
        }
        double* srow = S + (i64)slot * N + (i64)k1 * N2;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = threadIdx.x + u * NT;
            if (e < N2) srow[e] = first_batch ? acc[u] : srow[e] + acc[u];
        }
    }
}

// grid (N1, slots): row n1 of the spectrum -> length-N2 transform, twiddle W_N^(n1 k2) -> Bm[n1][k2]
template <int NT>
__global__ __launch_bounds__(NT) void k_fft_rows_spec(const double* __restrict__ S, Plan pl, const double2* __restrict__ tw2,
                                                      const unsigned* __restrict__ long_count,
                                                      const unsigned* __restrict__ long_list,
                                                      const double* __restrict__ state, double2* __restrict__ Bm)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2* buf = reinterpret_cast<double2*>(smem);
    const int N2 = 1 << pl.log2;
    const i64 N = (i64)N2 << pl.log1;
    const int n1 = blockIdx.x;
    const unsigned slot = blockIdx.y;
    if (slot >= *long_count) return;
    {
        const i64 pk = long_list[slot];
        if (state[pk * kPairState + 3] != 0.0) return;
        const double* srow = S + (i64)slot * N + (i64)n1 * N2;
        for (int e = threadIdx.x; e < N2; e += NT) buf[bitrev(e, pl.log2)] = make_double2(srow[e], 0.0);   // position e held k2 = bitrev(e)
        __syncthreads();
        lds_fft<NT>(buf, 1, pl.log2, tw2);
        double2* orow = Bm + (i64)slot * N + (i64)n1 * N2;
        for (int k2 = threadIdx.x; k2 < N2; k2 += NT) {
            const double2 v = buf[bitrev(k2, pl.log2)];
            const double2 w = big_twiddle((i64)n1 * k2, N);
            orow[k2] = make_double2(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
        }
    }
}

// grid (N2 / COLS, slots): columns k2 -> length-N1 transforms over n1 -> acov[pair][N2 k1 + k2] = Re / N for lags < n
template <int NT>
__global__ __launch_bounds__(NT) void k_fft_cols_out(const double2* __restrict__ Bm, Plan pl, const double2* __restrict__ tw1,
                                                     i64 n, const unsigned* __restrict__ long_count,
                                                     const unsigned* __restrict__ long_list,
                                                     const double* __restrict__ state, double* __restrict__ acov)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2* buf = reinterpret_cast<double2*>(smem);
    const int N1 = 1 << pl.log1, N2 = 1 << pl.log2;
    const i64 N = (i64)N1 * N2;
    const int COLS = kColElems >> pl.log1;
    const int col0 = blockIdx.x * COLS;
    const double inv = 1.0 / (double)N;
    const unsigned slot = blockIdx.y;
    if (slot >= *long_count) return;
    {
        const i64 pk = long_list[slot];
        if (state[pk * kPairState + 3] != 0.0) return;
        const double2* in = Bm + (i64)slot * N;
        for (int e = threadIdx.x; e < COLS * N1; e += NT) {
            const int n1 = e / COLS, cl = e - n1 * COLS;
            buf[(cl << pl.log1) + n1] = in[(i64)n1 * N2 + col0 + cl];
        }
        __syncthreads();
        lds_fft<NT>(buf, COLS, pl.log1, tw1);
        double* out = acov + pk * n;
        for (int e = threadIdx.x; e < COLS * N1; e += NT) {
            const int k1 = e / COLS, cl = e - k1 * COLS;
            const i64 lag = (i64)k1 * N2 + col0 + cl;
            if (lag < n) out[lag] = buf[(cl << pl.log1) + bitrev(k1, pl.log1)].x * inv;
        }
    }
}

}  // namespace fft
}  // namespace mcr
