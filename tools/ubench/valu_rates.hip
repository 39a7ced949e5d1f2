// Micro-benchmark: per-SIMD issue cost (cycles per wave64 instruction) of the integer VALU ops the sweep kernel
// is made of.  All CUs busy, 8 waves per SIMD, long unrolled dependent-free streams.  Build: hipcc --offload-arch=gfx950 -O3
// The shader clock is MEASURED per run (round 3): thread 0 of workgroup 0 reads s_memtime (shader cycles) and s_memrealtime (the
// constant 100 MHz reference) around its loop, so "cycles" below are cycles of the clock the run actually had — a DVFS-lowered
// clock can no longer pass for a slower pipe.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP> __global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed, unsigned long long* clk)
{
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x9e3779b9u, a2 = a0 + 77, a3 = a1 * 3 + 1;
    uint32_t b0 = a0 >> 3, b1 = a1 >> 5, b2 = a2 >> 7, b3 = a3 >> 9;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %5\n v_xor_b32 %2, %2, %6\n v_xor_b32 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 1) { REP16(asm volatile("v_or3_b32 %0, %0, %4, %5\n v_or3_b32 %1, %1, %5, %6\n v_or3_b32 %2, %2, %6, %7\n v_or3_b32 %3, %3, %7, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 2) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %5\n v_mul_lo_u32 %2, %2, %6\n v_mul_lo_u32 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 3) { REP16(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %5\n v_mul_hi_u32 %2, %2, %6\n v_mul_hi_u32 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 4) {
            uint64_t p0 = a0, p1 = a1, p2 = a2, p3 = a3;
            REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, 0\n v_mad_u64_u32 %1, vcc, %5, %6, 0\n v_mad_u64_u32 %2, vcc, %6, %7, 0\n v_mad_u64_u32 %3, vcc, %7, %4, 0" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");)
            a0 ^= (uint32_t)p0 ^ (uint32_t)(p0 >> 32); a1 ^= (uint32_t)p1; a2 ^= (uint32_t)p2; a3 ^= (uint32_t)(p3 >> 32);
        }
        if (OP == 5) { REP16(asm volatile("v_bfi_b32 %0, %4, %0, %5\n v_bfi_b32 %1, %5, %1, %6\n v_bfi_b32 %2, %6, %2, %7\n v_bfi_b32 %3, %7, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 6) { REP16(asm volatile("v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %5, %6\n v_and_or_b32 %2, %2, %6, %7\n v_and_or_b32 %3, %3, %7, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 7) { REP16(asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %5\n v_mul_u32_u24 %2, %2, %6\n v_mul_u32_u24 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 8) { REP16(asm volatile("v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %5, %6\n v_mad_u32_u24 %2, %2, %6, %7\n v_mad_u32_u24 %3, %3, %7, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 9) { REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 10) { REP16(asm volatile("v_mul_hi_u32_u24 %0, %0, %4\n v_mul_hi_u32_u24 %1, %1, %5\n v_mul_hi_u32_u24 %2, %2, %6\n v_mul_hi_u32_u24 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 11) { REP16(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %6\n v_mov_b32 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 12) { REP16(asm volatile("v_pk_mul_lo_u16 %0, %0, %4\n v_pk_mul_lo_u16 %1, %1, %5\n v_pk_mul_lo_u16 %2, %2, %6\n v_pk_mul_lo_u16 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 13) { REP16(asm volatile("v_mad_u16 %0, %0, %4, %5\n v_mad_u16 %1, %1, %5, %6\n v_mad_u16 %2, %2, %6, %7\n v_mad_u16 %3, %3, %7, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 14) { REP16(asm volatile("v_bitop3_b32 %0, %0, %4, %5 bitop3:0x96\n v_bitop3_b32 %1, %1, %5, %6 bitop3:0x96\n v_bitop3_b32 %2, %2, %6, %7 bitop3:0x96\n v_bitop3_b32 %3, %3, %7, %4 bitop3:0x96" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 15) { REP16(asm volatile("v_alignbit_b32 %0, %0, %4, %5\n v_alignbit_b32 %1, %1, %5, %6\n v_alignbit_b32 %2, %2, %6, %7\n v_alignbit_b32 %3, %3, %7, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 16) { REP16(asm volatile("v_bcnt_u32_b32 %0, %4, %0\n v_bcnt_u32_b32 %1, %5, %1\n v_bcnt_u32_b32 %2, %6, %2\n v_bcnt_u32_b32 %3, %7, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
    if (blockIdx.x == 0 && threadIdx.x == 0 && clk) { clk[0] = __builtin_amdgcn_s_memtime() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}
template <int OP> void run(const char* name, uint32_t* d)
{
    const int blocks = 256 * 8, iters = 2000;   // 8 blocks of 4 waves per CU = 8 waves per SIMD
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
    const double ghz = h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0;          // shader cycles per 10 ns tick of the 100 MHz reference
    const double insts_per_simd = (double)blocks * 4 / 1024.0 * iters * 64.0;   // waves per SIMD x instrs per wave
    printf("%-18s %8.3f ms  -> %.2f ns per wave-instr per SIMD = %.2f cycles @ measured %.3f GHz\n", name, ms, ms * 1e6 / insts_per_simd,
           ms * 1e6 / insts_per_simd * ghz, ghz);
}
int main()
{
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_xor_b32", d); run<1>("v_or3_b32", d); run<2>("v_mul_lo_u32", d); run<3>("v_mul_hi_u32", d);
    run<4>("v_mad_u64_u32", d); run<5>("v_bfi_b32", d); run<6>("v_and_or_b32", d); run<7>("v_mul_u32_u24", d);
    run<8>("v_mad_u32_u24", d); run<9>("v_add_u32", d); run<10>("v_mul_hi_u32_u24", d); run<11>("v_mov_b32", d);
    run<12>("v_pk_mul_lo_u16", d); run<13>("v_mad_u16", d);
    run<14>("v_bitop3_b32", d); run<15>("v_alignbit_b32", d); run<16>("v_bcnt_u32_b32", d);
    return 0;
}
