// Micro-benchmark: per-SIMD issue cost (cycles per wave64 instruction) of the Float64 ops the dense-SK bulk update is made of, at 8 and
// at 2 waves per SIMD (the SK block kernel runs 2).  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/f64_rates.hip -o tools/ubench/f64_rates.out
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP> __global__ __launch_bounds__(256) void k(double* out, int iters, double seed, unsigned long long* clk)
{
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    double a0 = threadIdx.x * 1e-3 + seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double b0 = 1e-9 * threadIdx.x, b1 = b0 * 2;
    double m = seed > 100 ? 0.5 : 1.0;          // wave-uniform
    uint32_t u0 = threadIdx.x, u1 = u0 * 3;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %2, %8, %9, %2\n v_fma_f64 %3, %8, %9, %3\n v_fma_f64 %4, %8, %9, %4\n v_fma_f64 %5, %8, %9, %5\n v_fma_f64 %6, %8, %9, %6\n v_fma_f64 %7, %8, %9, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));) }
        if (OP == 1) { REP16(asm volatile("v_fmac_f64 %0, %9, %8\n v_fmac_f64 %1, %9, %8\n v_fmac_f64 %2, %9, %8\n v_fmac_f64 %3, %9, %8\n v_fmac_f64 %4, %9, %8\n v_fmac_f64 %5, %9, %8\n v_fmac_f64 %6, %9, %8\n v_fmac_f64 %7, %9, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "s"(m));) }
        if (OP == 2) { REP16(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));) }
        if (OP == 3) { REP16(asm volatile("v_mov_b64 %0, %8\n v_mov_b64 %1, %9\n v_mov_b64 %2, %8\n v_mov_b64 %3, %9\n v_mov_b64 %4, %8\n v_mov_b64 %5, %9\n v_mov_b64 %6, %8\n v_mov_b64 %7, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));) }
        if (OP == 4) { uint32_t s0, s1, s2, s3; REP16(asm volatile("v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %5, 5\n v_readlane_b32 %2, %4, 7\n v_readlane_b32 %3, %5, 9\n v_readlane_b32 %0, %4, 13\n v_readlane_b32 %1, %5, 15\n v_readlane_b32 %2, %4, 17\n v_readlane_b32 %3, %5, 19" : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(u0), "v"(u1));) a0 += s0 + s1 + s2 + s3; }
        if (OP == 5) { REP16(asm volatile("v_pk_fma_f32 %0, %8, %9, %0\n v_pk_fma_f32 %1, %8, %9, %1\n v_pk_fma_f32 %2, %8, %9, %2\n v_pk_fma_f32 %3, %8, %9, %3\n v_pk_fma_f32 %4, %8, %9, %4\n v_pk_fma_f32 %5, %8, %9, %5\n v_pk_fma_f32 %6, %8, %9, %6\n v_pk_fma_f32 %7, %8, %9, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));) }
        if (OP == 7) { REP16(asm volatile("v_fmac_f64_dpp %0, %9, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %9, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %2, %9, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %9, %8 row_newbcast:9 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %4, %9, %8 row_newbcast:11 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %9, %8 row_newbcast:13 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %6, %9, %8 row_newbcast:15 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %7, %9, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));) }
        if (OP == 6) { REP16(asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (blockIdx.x == 0 && threadIdx.x == 0 && clk) { clk[0] = __builtin_amdgcn_s_memtime() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}
template <int OP> void run(const char* name, double* d, int wps)
{
    const int blocks = 256 * wps, iters = 1000;   // wps blocks of 4 waves per CU = wps waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    static unsigned long long* clk = nullptr;
    if (!clk) hipMalloc(&clk, 16);
    k<OP><<<blocks, 256>>>(d, 10, 1, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, iters, 2, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2] = {0, 0};
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0;
    const double insts_per_simd = (double)wps * iters * 128.0;   // waves per SIMD x instrs per wave
    printf("%-28s %d waves/SIMD %8.3f ms -> %.2f cycles per wave-instr per SIMD (%.2f per wave) @ measured %.3f GHz\n", name, wps, ms, ms * 1e6 / insts_per_simd * ghz,
           ms * 1e6 / (iters * 128.0) * ghz, ghz);
}
int main()
{
    double* d; hipMalloc(&d, 256 * 8 * 256 * 8);
    for (int wps : {8, 2, 1}) {
        run<0>("v_fma_f64 (vgpr)", d, wps); run<1>("v_fmac_f64 (sgpr multiplier)", d, wps); run<2>("v_add_f64", d, wps); run<6>("v_mul_f64", d, wps); run<3>("v_mov_b64", d, wps);
        run<4>("v_readlane_b32", d, wps); run<7>("v_fmac_f64_dpp row_newbcast", d, wps); run<5>("v_pk_fma_f32", d, wps);
    }
    return 0;
}
