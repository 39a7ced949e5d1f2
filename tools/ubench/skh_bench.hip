// Stand-alone timing harness for the dense-SK block kernels (perf experiments only: correctness is tests/test_gpu_sk_*.py through the
// library).  Builds in seconds, unlike the whole library:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DRRRMC_SKB_STAMPS] tools/ubench/skh_bench.hip -o tools/ubench/skh_bench.out
//   ./tools/ubench/skh_bench.out [kernel: h8 | h4 | v8 | v4] [N = 1024] [R = 2048] [iters = 65536] [beta = 1.0]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../rrrmc.jl_amd/csrc/sk_hblock_kernel.hpp"

using namespace rrrmc;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef void (*kfn)(SkBlockParams);

static kfn pick(const char* which, int N, int* nth, int* wgs)
{
    const bool h = which[0] == 'h', half = which[1] == '4';
    if (half) {
        const int spt = (N + 255) / 256;
        *nth = 256; *wgs = 2;
        if (h) return spt == 1 ? sk_hblock_kernel<1, 256, 4> : spt == 2 ? sk_hblock_kernel<2, 256, 4> : spt == 3 ? sk_hblock_kernel<3, 256, 4> : sk_hblock_kernel<4, 256, 4>;
        return spt == 1 ? sk_block_kernel<1, 256, 4> : spt == 2 ? sk_block_kernel<2, 256, 4> : spt == 3 ? sk_block_kernel<3, 256, 4> : sk_block_kernel<4, 256, 4>;
    }
    *wgs = 1;
    if (N <= 256) { *nth = 256; return h ? sk_hblock_kernel<1, 256, 8> : sk_block_kernel<1, 256>; }
    const int spt = (N + 511) / 512;
    *nth = 512;
    if (h) return spt == 1 ? sk_hblock_kernel<1, 512, 8> : spt == 2 ? sk_hblock_kernel<2, 512, 8> : spt == 3 ? sk_hblock_kernel<3, 512, 8> : spt == 4 ? sk_hblock_kernel<4, 512, 8>
                         : spt <= 6 ? sk_hblock_kernel<6, 512, 8> : sk_hblock_kernel<8, 512, 8>;
    return spt == 1 ? sk_block_kernel<1, 512> : spt == 2 ? sk_block_kernel<2, 512> : spt == 3 ? sk_block_kernel<3, 512> : sk_block_kernel<4, 512>;
}

int main(int argc, char** argv)
{
    const char* which = argc > 1 ? argv[1] : "h8";
    const int N = argc > 2 ? atoi(argv[2]) : 1024, R = argc > 3 ? atoi(argv[3]) : 2048;
    const long long iters = argc > 4 ? atoll(argv[4]) : 65536;
    const double beta = argc > 5 ? atof(argv[5]) : 1.0;
    const int G8 = (R + 7) / 8, Rp = G8 * 8, ld = (N + 1023) / 1024 * 1024;
    int nth = 0, wgs = 0;
    kfn fn = pick(which, N, &nth, &wgs);

    std::vector<double> J((size_t)N * N, 0.0), J4((size_t)N * ld, 0.0);
    srand48(12345);
    for (int i = 0; i < N; ++i)
        for (int j = i + 1; j < N; ++j) {
            // Gaussian by Box-Muller, variance 1/N
            const double u1 = drand48() + 1e-300, u2 = drand48();
            const double g = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2) / std::sqrt((double)N);
            J[(size_t)i * N + j] = J[(size_t)j * N + i] = g;
        }
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) J4[(size_t)i * ld + j] = 4.0 * J[(size_t)i * N + j];
    std::vector<uint8_t> sp((size_t)G8 * N);
    for (auto& b : sp) b = (uint8_t)(lrand48() & 0xff);

    double *dJ, *dJ4, *lf, *lfl, *hl, *E, *blkJw;
    uint8_t* dsp; int32_t* ml; int64_t* acc; uint32_t* blkSites;
    const long long nblk = (iters + kSkW - 1) / kSkW;
    CK(hipMalloc(&dJ, sizeof(double) * J.size())); CK(hipMalloc(&dJ4, sizeof(double) * J4.size()));
    CK(hipMalloc(&lf, sizeof(double) * (size_t)Rp * N)); CK(hipMalloc(&lfl, sizeof(double) * (size_t)Rp * N)); CK(hipMalloc(&hl, sizeof(double) * (size_t)Rp * N));
    CK(hipMalloc(&E, sizeof(double) * Rp)); CK(hipMalloc(&acc, sizeof(int64_t) * Rp)); CK(hipMalloc(&ml, sizeof(int32_t) * Rp));
    CK(hipMalloc(&dsp, (sp.size() + 3) / 4 * 4));
    CK(hipMalloc(&blkJw, sizeof(double) * (size_t)nblk * kSkW * kSkW)); CK(hipMalloc(&blkSites, sizeof(uint32_t) * (size_t)(nblk + 1) * kSkW));
    CK(hipMemcpy(dJ, J.data(), sizeof(double) * J.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dJ4, J4.data(), sizeof(double) * J4.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dsp, sp.data(), sp.size(), hipMemcpyHostToDevice));
    CK(hipMemset(acc, 0, sizeof(int64_t) * Rp));
    hipLaunchKernelGGL(sk_fields_kernel, dim3((N + 31) / 32, G8), dim3(256), 0, 0, dJ, dsp, lf, lfl, ml, N);
    hipLaunchKernelGGL(sk_energy_kernel, dim3((Rp + 63) / 64), dim3(64), 0, 0, lf, E, N, Rp);
    CK(hipDeviceSynchronize());

    SkBlockParams P{};
    P.J4 = dJ4; P.blkJw = blkJw; P.blkSites = blkSites; P.lf = lf; P.lfl = lfl; P.hl = hl; P.move_last = ml; P.spins = dsp; P.E_cur = E; P.acc_cur = acc; P.Es = nullptr;
    P.beta = beta; P.sN = std::sqrt((double)N); P.step = 1024; P.k0 = 0x5EED; P.k1 = 0; P.replica0 = 0; P.N = N; P.ldJ = ld; P.iters = iters; P.it_base = 0;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        P.g0 = (uint64_t)rep * iters;
        hipLaunchKernelGGL(sk_block_prep_kernel, dim3((unsigned)(nblk + 1)), dim3(256), 0, 0, dJ4, blkSites, blkJw, P.g0, iters, nblk, P.k0, P.k1, N, ld);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(fn, dim3((unsigned)(G8 * wgs)), dim3((unsigned)nth), 0, 0, P);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<int64_t> a(Rp);
        CK(hipMemcpy(a.data(), acc, sizeof(int64_t) * Rp, hipMemcpyDeviceToHost));
        double s = 0; for (int r = 0; r < R; ++r) s += (double)a[r];
        printf("%s N=%d R=%d iters=%lld: %.3f ms  %.3e attempts/s  accepted/attempt (cumulative) %.4f\n", which, N, R, iters, ms, (double)R * iters / (ms * 1e-3),
               s / R / ((rep + 1.0) * iters));
    }
    return 0;
}
