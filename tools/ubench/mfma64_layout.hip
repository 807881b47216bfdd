// Debug: the covariance tile loop with a rank-1 X, printed per (row, col).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void cov_tile(const double* X, long long M, long long P, double* out)
{
    const int lane = threadIdx.x, r = lane & 15, kq = lane >> 4;
    const double* xi = X + (r < P ? r : 0) * M;
    const double Ki = xi[0];
    v4d acc = {0, 0, 0, 0};
    for (long long t = 0; t < M; t += 4) {
        const long long tt = t + kq;
        const double a = (r < P && tt < M) ? xi[tt] - Ki : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) out[(kq + 4 * i) * 16 + r] = acc[i];
}
int main()
{
    const int P = 6, M = 8;
    double hX[P * M], e[M] = {0.3, -1.2, 0.7, 2.1, -0.4, 0.9, -1.6, 0.5};
    for (int i = 0; i < P; ++i) for (int t = 0; t < M; ++t) hX[i * M + t] = (i + 1) * e[t];
    double *dX, *dO; hipMalloc(&dX, sizeof(hX)); hipMalloc(&dO, 256 * 8);
    hipMemcpy(dX, hX, sizeof(hX), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(cov_tile, dim3(1), dim3(64), 0, 0, dX, (long long)M, (long long)P, dO);
    double h[256]; hipMemcpy(h, dO, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int t = 0; t < M; ++t) s += (e[t] - e[0]) * (e[t] - e[0]);
    for (int i = 0; i < 7; ++i) { for (int j = 0; j < 7; ++j) printf(" %8.3f", h[i * 16 + j] / s); printf("\n"); }
    return 0;
}
