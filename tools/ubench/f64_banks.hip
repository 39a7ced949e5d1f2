// Micro-benchmark: does the register BANK of the operands of v_fmac_f64_dpp (dst/src2, DPP src0, src1) change its issue cost?
// Explicit registers; eight accumulators per case.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/f64_banks.hip -o tools/ubench/f64_banks.out
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
// accumulators v[32+4i+A : +1], multiplier (dpp) v[16+M : +1], row value v[24+D : +1]; A, M, D in {0, 2} choose the bank pair (0,1) or (2,3)
#define FM(a0, a1, m0, m1, d0, d1) "v_fmac_f64_dpp v[" #a0 ":" #a1 "], v[" #m0 ":" #m1 "], v[" #d0 ":" #d1 "] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
#define CLOB "v16","v17","v18","v19","v24","v25","v26","v27","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63"
template <int CASE> __global__ __launch_bounds__(256) void k(double* out, int iters, double seed)
{
    asm volatile("v_mov_b32 v16, 0\n v_mov_b32 v17, 0\n v_mov_b32 v18, 0\n v_mov_b32 v19, 0\n v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0\n" ::: CLOB);
    for (int i = 0; i < iters; ++i) {
        if (CASE == 0) { REP8(asm volatile(FM(32,33,16,17,24,25) FM(36,37,16,17,24,25) FM(40,41,16,17,24,25) FM(44,45,16,17,24,25) FM(48,49,16,17,24,25) FM(52,53,16,17,24,25) FM(56,57,16,17,24,25) FM(60,61,16,17,24,25) ::: CLOB);) }   // all in banks 0,1
        if (CASE == 1) { REP8(asm volatile(FM(32,33,18,19,24,25) FM(36,37,18,19,24,25) FM(40,41,18,19,24,25) FM(44,45,18,19,24,25) FM(48,49,18,19,24,25) FM(52,53,18,19,24,25) FM(56,57,18,19,24,25) FM(60,61,18,19,24,25) ::: CLOB);) }   // multiplier in 2,3
        if (CASE == 2) { REP8(asm volatile(FM(32,33,16,17,26,27) FM(36,37,16,17,26,27) FM(40,41,16,17,26,27) FM(44,45,16,17,26,27) FM(48,49,16,17,26,27) FM(52,53,16,17,26,27) FM(56,57,16,17,26,27) FM(60,61,16,17,26,27) ::: CLOB);) }   // row value in 2,3
        if (CASE == 3) { REP8(asm volatile(FM(32,33,18,19,26,27) FM(36,37,18,19,26,27) FM(40,41,18,19,26,27) FM(44,45,18,19,26,27) FM(48,49,18,19,26,27) FM(52,53,18,19,26,27) FM(56,57,18,19,26,27) FM(60,61,18,19,26,27) ::: CLOB);) }   // both sources in 2,3, accumulators in 0,1
        if (CASE == 4) { REP8(asm volatile(FM(32,33,16,17,24,25) FM(34,35,16,17,24,25) FM(36,37,16,17,24,25) FM(38,39,16,17,24,25) FM(40,41,16,17,24,25) FM(42,43,16,17,24,25) FM(44,45,16,17,24,25) FM(46,47,16,17,24,25) ::: CLOB);) }   // accumulators alternate, sources 0,1 (the kernel's pattern)
        if (CASE == 5) { REP8(asm volatile(FM(32,33,16,17,26,27) FM(34,35,16,17,26,27) FM(36,37,16,17,26,27) FM(38,39,16,17,26,27) FM(40,41,16,17,26,27) FM(42,43,16,17,26,27) FM(44,45,16,17,26,27) FM(46,47,16,17,26,27) ::: CLOB);) }   // accumulators alternate, sources split
        if (CASE == 6) { REP8(asm volatile(FM(32,33,16,17,24,25) FM(34,35,18,19,24,25) FM(36,37,16,17,26,27) FM(38,39,18,19,26,27) FM(40,41,16,17,24,25) FM(42,43,18,19,24,25) FM(44,45,16,17,26,27) FM(46,47,18,19,26,27) ::: CLOB);) }   // everything varies (like the real kernel)
    }
    double r; asm volatile("v_mov_b64 %0, v[32:33]" : "=v"(r) :: CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + seed;
}
template <int CASE> void run(const char* name, double* d, int wps)
{
    const int blocks = 256 * wps, iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<CASE><<<blocks, 256>>>(d, 10, 1); hipDeviceSynchronize();
    hipEventRecord(e0); k<CASE><<<blocks, 256>>>(d, iters, 2); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %d waves/SIMD: %.2f cycles per instruction and SIMD at 2.4 GHz\n", name, wps, ms * 1e-3 * 2.4e9 / ((double)wps * iters * 64.0));
}
int main()
{
    double* d; hipMalloc(&d, 256 * 8 * 256 * 8);
    for (int wps : {2, 4}) {
        run<0>("acc 0,1  mult 0,1  row 0,1", d, wps); run<1>("acc 0,1  mult 2,3  row 0,1", d, wps); run<2>("acc 0,1  mult 0,1  row 2,3", d, wps);
        run<3>("acc 0,1  mult 2,3  row 2,3", d, wps); run<4>("acc alternating  mult 0,1  row 0,1", d, wps); run<5>("acc alternating  mult 0,1  row 2,3", d, wps);
        run<6>("all alternating", d, wps);
    }
    return 0;
}
