// Microbenchmark: v_mfma_f64_16x16x4f64 issue rate on gfx950, alone and as the autocovariance Gram step
// (4 LDS operand loads + 5 MFMAs per 64 draws), to decide whether the lag products belong on the matrix cores.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters)
{
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = v4d{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        asm volatile("" : "+v"(a), "+v"(b));
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Gram step: per 64 draws, A = z[64 it + lane], B_q = z[64 it + 16 q + lane], q = 0..4 (B_0 = A, B_4 = next A)
__global__ __launch_bounds__(256) void k_gram(double* out, int nit, int reps)
{
    __shared__ double z[2048 + 128];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 2048 + 128; i += 256) z[i] = (i % 97) * 1e-3;
    __syncthreads();
    v4d g0{0,0,0,0}, g1{0,0,0,0}, g2{0,0,0,0}, g3{0,0,0,0}, g4{0,0,0,0};
    for (int r = 0; r < reps; ++r)
    for (int it = w; it < nit; it += 4) {
        const double* p = z + 64 * it + lane;
        const double a = p[0], b1 = p[16], b2 = p[32], b3 = p[48], b4 = p[64];
        g0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, g0, 0, 0, 0);
        g1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, g1, 0, 0, 0);
        g2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b2, g2, 0, 0, 0);
        g3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b3, g3, 0, 0, 0);
        g4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b4, g4, 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += g0[i] + g1[i] + g2[i] + g3[i] + g4[i];
    out[blockIdx.x * 256 + tid] = s;
}

template <typename F> float timeit(F f, int n = 5)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < n; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / n;
}

int main()
{
    double* out; CK(hipMalloc(&out, 256 * 4096 * 8));
    const int blocks = 256 * 8, iters = 2000;
    {
        float ms = timeit([&] { hipLaunchKernelGGL(k_mfma<5>, dim3(blocks), dim3(256), 0, 0, out, iters); });
        double fl = (double)blocks * 4 * iters * 5 * 2048.0;
        printf("mfma f64 16x16x4, 5 independent accumulators: %.1f TFLOP/s  (%.3f ms)\n", fl / ms / 1e9, ms);
    }
    {
        float ms = timeit([&] { hipLaunchKernelGGL(k_mfma<1>, dim3(blocks), dim3(256), 0, 0, out, iters); });
        double fl = (double)blocks * 4 * iters * 1 * 2048.0;
        printf("mfma f64 16x16x4, 1 accumulator (dependent): %.1f TFLOP/s  (%.3f ms)\n", fl / ms / 1e9, ms);
    }
    {
        const int nit = 32, reps = 200;    // 2048 draws per workgroup
        float ms = timeit([&] { hipLaunchKernelGGL(k_gram, dim3(blocks), dim3(256), 0, 0, out, nit, reps); });
        double steps = (double)blocks * nit * reps;         // 64-draw steps
        printf("gram step (4 LDS loads + 5 mfma per 64 draws): %.2f G draws/s x 64 lags, %.1f TFLOP/s issued, %.3f ms\n",
               steps * 64 / ms / 1e6, steps * 5 * 2048 / ms / 1e9, ms);
        printf("  -> 8 M draws (C1 bulk + tail) would take %.1f us\n", 8e6 / (steps * 64 / ms * 1e3) * 1e3 * 1e3);
    }
    return 0;
}
