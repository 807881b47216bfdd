// Probe the operand / result lane layout of v_mfma_f64_16x16x4f64 on gfx950.
// Measured (MI355X, ROCm 7.2): A[i][k] and B[k][j] sit at lane 16k + i (resp. 16k + j);
// D[i][j] is returned in lane 16*(i % 4) + j, register i / 4  (i.e. lane l, register v holds row (l/16) + 4v).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void probe(double* out, int mode)
{
    const int l = threadIdx.x;
    double a, b;
    if (mode == 0) { a = (double)(1 << (l / 16)) * 1000.0 + (l % 16); b = 1.0; }   // D[i][j] = sum_k A[i][k]
    else { a = 1.0; b = (double)(1 << (l / 16)) * 1000.0 + (l % 16); }             // D[i][j] = sum_k B[k][j]
    v4d acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = acc[i];
}
int main()
{
    double* d; hipMalloc(&d, 256 * 8); double h[256];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode); hipMemcpy(h, d, 256 * 8, hipMemcpyDeviceToHost);
        printf("mode %d (operand element (idx, k) at lane 16k+idx  =>  every entry = 15000 + 4*idx)\n", mode);
        for (int l = 0; l < 64; ++l) { printf(" l%2d:", l); for (int i = 0; i < 4; ++i) printf(" %7.0f", h[l * 4 + i]); if (l % 4 == 3) printf("\n"); }
    }
    return 0;
}
