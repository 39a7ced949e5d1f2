// Micro-benchmark: SIMD issue cost of ONE PRODUCER TASK of sweep_kernel<3> (64 slots x 32 replicas: 3 Philox4x32-10 ACCEPT blocks + the
// bit-sliced threshold refinement, the code of sparse_kernels.hpp itself) at 1, 2, 4 and 8 waves per SIMD — what the VALU model of
// tools/valu_model.py prices at 277 ns.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I rrrmc.jl_amd/csrc
#include "sparse_kernels.hpp"
#include <cstdio>
using namespace rrrmc;
struct TaskParams { uint32_t taum[64 * 4]; uint32_t k0, k1; int iters; uint32_t* out; };
template <int NT, int BLOCKS>
__global__ __launch_bounds__(256) void ptask(TaskParams P)
{
    const uint32_t group = blockIdx.x;
    uint64_t g = (uint64_t)threadIdx.x * 1000003ull + 17;
    uint32_t acc = 0;
    for (int it = 0; it < P.iters; ++it) {
        uint32_t lt[NT], eq[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) { lt[n] = 0u; eq[n] = 0xffffffffu; }
#pragma unroll
        for (int b = 0; b < BLOCKS; ++b) refine_block<NT>(lt, eq, accept_planes(P.k0, P.k1, g, group, (uint32_t)b), (uint32_t)b, P.taum);
#pragma unroll
        for (int n = 0; n < NT; ++n) acc ^= lt[n] + eq[n];
        g += 256;
    }
    P.out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int NT, int BLOCKS> void run(int waves_per_simd, uint32_t* d)
{
    TaskParams P{};
    for (int i = 0; i < 256; ++i) P.taum[i] = (i * 2654435761u) & 0x10000u ? ~0u : 0u;
    P.k0 = 0x5eed; P.k1 = 0x1234; P.out = d;
    const int blocks = 256 * waves_per_simd;        // a block of 4 waves = one wave per SIMD of a CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    P.iters = 10;
    ptask<NT, BLOCKS><<<blocks, 256>>>(P);
    hipDeviceSynchronize();
    P.iters = 4000;
    hipEventRecord(e0);
    ptask<NT, BLOCKS><<<blocks, 256>>>(P);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tasks_per_simd = (double)waves_per_simd * P.iters;
    printf("NT=%d blocks=%d  %d waves/SIMD: %8.3f ms -> %.1f ns per task per SIMD (= %.0f cycles @2.35 GHz), %.0f ns per task per wave\n", NT, BLOCKS, waves_per_simd, ms,
           ms * 1e6 / tasks_per_simd, ms * 1e6 / tasks_per_simd * 2.35, ms * 1e6 / P.iters);
}
int main()
{
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 3, 4, 8}) run<2, 3>(w, d);
    for (int w : {1, 4, 8}) run<2, 2>(w, d);
    for (int w : {1, 4, 8}) run<2, 1>(w, d);
    return 0;
}
