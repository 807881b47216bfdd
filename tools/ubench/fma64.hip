// Microbenchmarks: fp64 FMA rate, and the acov 8x8 register tile with/without its LDS reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

template<int NACC>
__global__ __launch_bounds__(1024) void k_fma(double* out, int iters, double a, double b)
{
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-9 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(acc[i], a, b);
    }
    double s = 0; for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template<bool LDSREAD, int NT>
__global__ __launch_bounds__(NT) void k_tile(double* out, int nit, int reps)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* d = reinterpret_cast<double*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, NW = NT / 64;
    for (int i = tid; i < 12800; i += NT) d[i] = (i % 97) * 1e-3;
    __syncthreads();
    const int g = lane & 7, ph = lane >> 3;
    double acc[8]; for (int i = 0; i < 8; ++i) acc[i] = 0;
    double a[8], b[16];
    for (int j = 0; j < 8; ++j) a[j] = d[j + lane]; for (int j = 0; j < 16; ++j) b[j] = d[64 + j + lane];
    for (int r = 0; r < reps; ++r)
    for (int it = w; it < nit; it += NW) {
        const int i0 = (it << 6) + (ph << 3);
        const int s = i0 + (g << 3);
        if (LDSREAD) {
            const double2* pa = reinterpret_cast<const double2*>(d + 10 * (i0 >> 3));
            const double2* pb = reinterpret_cast<const double2*>(d + 10 * (s >> 3));
#pragma unroll
            for (int j = 0; j < 4; ++j) { const double2 v = pa[j]; a[2*j] = v.x; a[2*j+1] = v.y; }
#pragma unroll
            for (int j = 0; j < 4; ++j) { const double2 v = pb[j]; b[2*j] = v.x; b[2*j+1] = v.y; }
#pragma unroll
            for (int j = 0; j < 4; ++j) { const double2 v = pb[5+j]; b[8+2*j] = v.x; b[9+2*j] = v.y; }
        } else {
            asm volatile("" : "+v"(a[0]), "+v"(b[0]));
        }
#pragma unroll
        for (int li = 0; li < 8; ++li)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[li] = fma(a[k], b[k + li], acc[li]);
    }
    double s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * NT + tid] = s;
}

template<typename F> float timeit(F f, int n = 5) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < n; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / n;
}

int main() {
    double* out; CK(hipMalloc(&out, 2048 * 1024 * 8));
    {
        const int iters = 4096, blocks = 256 * 2;
        float ms = timeit([&]{ hipLaunchKernelGGL(k_fma<8>, dim3(blocks), dim3(1024), 0, 0, out, iters, 1.0000001, 1e-9); });
        double fl = 2.0 * 8 * iters * 1024.0 * blocks;
        printf("fma64 8 acc, 1024thr x %d blocks: %.3f ms  %.1f TFLOP/s\n", blocks, ms, fl / ms / 1e9);
        ms = timeit([&]{ hipLaunchKernelGGL(k_fma<16>, dim3(blocks), dim3(1024), 0, 0, out, iters, 1.0000001, 1e-9); });
        fl = 2.0 * 16 * iters * 1024.0 * blocks;
        printf("fma64 16 acc: %.3f ms  %.1f TFLOP/s\n", ms, fl / ms / 1e9);
    }
    {
        const int nit = 157, reps = 20, blocks = 256;
        hipFuncSetAttribute((const void*)k_tile<true, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 110 * 1024);
        hipFuncSetAttribute((const void*)k_tile<false, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 110 * 1024);
        hipFuncSetAttribute((const void*)k_tile<true, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 110 * 1024);
        float ms = timeit([&]{ hipLaunchKernelGGL((k_tile<true, 1024>), dim3(blocks), dim3(1024), 110 * 1024, 0, out, nit, reps); });
        double fl = 2.0 * 64 * 64 * nit * reps * blocks;
        printf("tile LDS  1024thr: %.3f ms  => %.2f us per 157-iteration block, %.1f TFLOP/s\n", ms, ms * 1e3 / reps, fl / ms / 1e9);
        ms = timeit([&]{ hipLaunchKernelGGL((k_tile<false, 1024>), dim3(blocks), dim3(1024), 110 * 1024, 0, out, nit, reps); });
        printf("tile noLDS 1024thr: %.3f ms  => %.2f us per block, %.1f TFLOP/s\n", ms, ms * 1e3 / reps, fl / ms / 1e9);
        ms = timeit([&]{ hipLaunchKernelGGL((k_tile<true, 512>), dim3(blocks), dim3(512), 110 * 1024, 0, out, nit, reps); });
        printf("tile LDS   512thr: %.3f ms  => %.2f us per block, %.1f TFLOP/s\n", ms, ms * 1e3 / reps, fl / ms / 1e9);
    }
    return 0;
}
