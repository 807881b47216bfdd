// Microbenchmark: issue rate of the instruction kinds the sort kernels are made of, per SIMD, at 1 / 2 / 4 / 8 waves per SIMD
// (gfx950).  The question behind it: does a 32-bit VOP2 select (v_cndmask_b32_e32, mask in VCC) issue at one per 2 cycles
// like v_fma_f32, and the VOP3 form (mask in an SGPR pair), the fp64 compare and the DPP move at one per 4?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

template <int KIND>
__global__ __launch_bounds__(256) void k_rate(unsigned* out, int iters)
{
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned b = blockIdx.x + 17;
    double d0 = threadIdx.x * 0.5, d1 = 3.25;
    unsigned long long m = 0x5555555555555555ull ^ (unsigned long long)blockIdx.x;
    if (KIND == 0) asm volatile("s_mov_b64 vcc, %0\n s_nop 4" :: "s"(m) : "vcc");      // the mask is set once; nothing in the loop writes VCC
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {   // v_cndmask_b32_e32 (VOP2, mask = VCC)
            asm volatile("v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e32 %7, %7, %8, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else if (KIND == 1) {   // v_cndmask_b32_e64 (VOP3, mask in an SGPR pair)
            asm volatile("v_cndmask_b32_e64 %0, %0, %9, %8\n v_cndmask_b32_e64 %1, %1, %9, %8\n v_cndmask_b32_e64 %2, %2, %9, %8\n v_cndmask_b32_e64 %3, %3, %9, %8\n"
                         "v_cndmask_b32_e64 %4, %4, %9, %8\n v_cndmask_b32_e64 %5, %5, %9, %8\n v_cndmask_b32_e64 %6, %6, %9, %8\n v_cndmask_b32_e64 %7, %7, %9, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m), "v"(b));
        } else if (KIND == 2) {   // v_add_u32 (VOP2)
            asm volatile("v_add_u32_e32 %0, %0, %8\n v_add_u32_e32 %1, %1, %8\n v_add_u32_e32 %2, %2, %8\n v_add_u32_e32 %3, %3, %8\n"
                         "v_add_u32_e32 %4, %4, %8\n v_add_u32_e32 %5, %5, %8\n v_add_u32_e32 %6, %6, %8\n v_add_u32_e32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else if (KIND == 3) {   // v_cmp_lt_f64 (to SGPR pairs)
            unsigned long long s0, s1, s2, s3;
            asm volatile("v_cmp_lt_f64_e64 %0, %4, %5\n v_cmp_lt_f64_e64 %1, %5, %4\n v_cmp_lt_f64_e64 %2, %4, %5\n v_cmp_lt_f64_e64 %3, %5, %4\n"
                         "v_cmp_lt_f64_e64 %0, %4, %5\n v_cmp_lt_f64_e64 %1, %5, %4\n v_cmp_lt_f64_e64 %2, %4, %5\n v_cmp_lt_f64_e64 %3, %5, %4\n"
                         : "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3) : "v"(d0), "v"(d1));
            m ^= s0 ^ s1 ^ s2 ^ s3;
        } else if (KIND == 4) {   // v_mov_b32 DPP quad_perm
            asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 6) {   // VALU compare into VCC + selects on VCC (VOP2): 1 cmp + 3 cndmask, twice
            asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cndmask_b32_e32 %4, %4, %8, vcc\n"
                         "v_cmp_lt_u32_e32 vcc, %1, %0\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e32 %7, %7, %8, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
        } else if (KIND == 7) {   // the same with the compare into an SGPR pair and VOP3 selects
            unsigned long long s0, s1;
            asm volatile("v_cmp_lt_u32_e64 %8, %0, %1\n v_cndmask_b32_e64 %2, %2, %10, %8\n v_cndmask_b32_e64 %3, %3, %10, %8\n v_cndmask_b32_e64 %4, %4, %10, %8\n"
                         "v_cmp_lt_u32_e64 %9, %1, %0\n v_cndmask_b32_e64 %5, %5, %10, %9\n v_cndmask_b32_e64 %6, %6, %10, %9\n v_cndmask_b32_e64 %7, %7, %10, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&s"(s0), "=&s"(s1) : "v"(b));
        } else if (KIND == 8) {   // fp64 compare into VCC + 6 selects on VCC: one compare-exchange of (f64 key, u32 payload) pairs
            asm volatile("v_cmp_lt_f64_e32 vcc, %9, %10\n v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n"
                         "v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_mov_b32_e32 %6, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(d0), "v"(d1) : "vcc");
        } else if (KIND == 9) {   // the same compare-exchange with the mask in an SGPR pair (what the compiler emits)
            unsigned long long s0;
            asm volatile("v_cmp_lt_f64_e64 %8, %10, %11\n v_cndmask_b32_e64 %0, %0, %9, %8\n v_cndmask_b32_e64 %1, %1, %9, %8\n v_cndmask_b32_e64 %2, %2, %9, %8\n"
                         "v_cndmask_b32_e64 %3, %3, %9, %8\n v_cndmask_b32_e64 %4, %4, %9, %8\n v_cndmask_b32_e64 %5, %5, %9, %8\n v_mov_b32_e32 %6, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&s"(s0) : "v"(b), "v"(d0), "v"(d1));
        } else {                  // v_fma_f32 for reference
            float f0 = __uint_as_float(a0), f1 = __uint_as_float(a1), f2 = __uint_as_float(a2), f3 = __uint_as_float(a3);
            float f4 = __uint_as_float(a4), f5 = __uint_as_float(a5), f6 = __uint_as_float(a6), f7 = __uint_as_float(a7), g = 1.0001f;
            asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                         "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(g));
            a0 = __float_as_uint(f0); a1 = __float_as_uint(f1); a2 = __float_as_uint(f2); a3 = __float_as_uint(f3);
            a4 = __float_as_uint(f4); a5 = __float_as_uint(f5); a6 = __float_as_uint(f6); a7 = __float_as_uint(f7);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (unsigned)m;
}

template <int KIND> int run(const char* name, unsigned* out)
{
    const int iters = 20000;
    for (int wps : {1, 2, 4, 8}) {                      // waves per SIMD: blocks of 256 threads = 4 waves = 1 per SIMD
        const int blocks = 256 * wps;                   // one block per CU and wave-per-SIMD step
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, out, 100); hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double instr_per_simd = (double)iters * 8 * wps;        // wave-instructions issued on one SIMD
        printf("%-22s %d waves/SIMD: %.2f ns per wave-instruction per SIMD  (%.3f ms)\n", name, wps, ms * 1e6 / instr_per_simd, ms);
    }
    return 0;
}

int main()
{
    unsigned* out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    run<5>("v_fma_f32", out); run<2>("v_add_u32_e32", out); run<0>("v_cndmask_b32_e32(vcc)", out); run<1>("v_cndmask_b32_e64(sgpr)", out);
    run<3>("v_cmp_lt_f64_e64", out); run<4>("v_mov_b32_dpp", out);
    run<6>("cmp_u32->vcc + 3 sel e32", out); run<7>("cmp_u32->sgpr + 3 sel e64", out);
    run<8>("CE f64: vcc + 6 sel e32", out); run<9>("CE f64: sgpr + 6 sel e64", out);
    return 0;
}
