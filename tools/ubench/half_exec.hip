// Does a wave64 VALU instruction whose upper 32 lanes are masked off (EXEC = 0x00000000ffffffff) issue faster than a full one on gfx950?
// Per-SIMD cycles per wave-instruction, 8 waves per SIMD, independent streams; TW = active lanes (64, 32, 16).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP> __global__ __launch_bounds__(256) void k(double* out, int iters, double seed, int tw, unsigned long long* clk)
{
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    double a0 = threadIdx.x * 1e-3 + seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    double b0 = 1e-9 * threadIdx.x + 1.0, b1 = b0 * 2;
    uint32_t u0 = threadIdx.x * 2654435761u, u1 = u0 ^ 0x9e3779b9u, u2 = u0 + 77, u3 = u1 * 3 + 1, w0 = u0 >> 3, w1 = u1 >> 5;
    if ((int)(threadIdx.x & 63) < tw) {
        for (int i = 0; i < iters; ++i) {
            if (OP == 0) { REP16(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %5\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %5" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w0), "v"(w1));) }
            if (OP == 1) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %5\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %5" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w0), "v"(w1));) }
            if (OP == 2) { REP16(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %5\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) }
            if (OP == 3) { REP16(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %5\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (double)(u0 ^ u1 ^ u2 ^ u3);
    if (blockIdx.x == 0 && threadIdx.x == 0 && clk) { clk[0] = __builtin_amdgcn_s_memtime() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}
template <int OP> void run(const char* name, double* d, int tw)
{
    const int blocks = 256 * 8, iters = 1000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    static unsigned long long* clk = nullptr;
    if (!clk) (void)hipMalloc(&clk, 16);
    k<OP><<<blocks, 256>>>(d, 10, 1, tw, nullptr);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, iters, 2, tw, clk);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2] = {0, 0};
    (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0;
    const double insts_per_simd = (double)blocks * 4 / 1024.0 * iters * 64.0;
    printf("%-14s TW %2d %8.3f ms -> %.2f cycles per wave-instr per SIMD @ %.3f GHz\n", name, tw, ms, ms * 1e6 / insts_per_simd * ghz, ghz);
}
int main()
{
    double* d; (void)hipMalloc(&d, 256 * 8 * 256 * 8);
    for (int tw : {64, 32, 16}) { run<0>("v_xor_b32", d, tw); run<1>("v_mul_lo_u32", d, tw); run<2>("v_add_f64", d, tw); run<3>("v_mul_f64", d, tw); }
    return 0;
}
