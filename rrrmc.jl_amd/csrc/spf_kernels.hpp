// gfx950 kernels for the Float64-coupling sparse models GraphRRGNormal / GraphEANormal (SimpleGraph{Float64};
// src/graphs/RRG.jl:503-627, src/graphs/EA.jl:534-680) under standardMC (src/RRRMC.jl:81-127).  SURVEY.md §8f rank 3.
//
// Float64 fields cannot be bit-sliced, so this is the HBM-bound picture of the north star: one lane per replica,
// replica-minor arrays so that a wavefront's access to one site is one coalesced 512-byte line:
//   lf, lfl : [W][N][64] Float64     local fields and lfields_last (the undo copy of update_cache!, RRG.jl:583-593)
//   spins   : [W][ceil(N/32)][64] uint32   lane-private bit words (bit x&31 of word x>>5)
// W = Rpad/64 wavefronts, one per workgroup; occupancy (up to 32 wavefronts per CU) hides the latency of the
// accept path, so the kernel wants many replicas: 64 Ki replicas of N = 4096 are 4.3 GB of fields — 288 GB of HBM
// hold millions.  All replicas attempt the same site (SITE stream, precomputed by spf_sites_kernel), the local field
// of the next D sites is prefetched D iterations ahead and re-read only when an accepted move touched it.
// Every field receives exactly the reference's sequence of IEEE operations (no FMA contraction) and det_exp is the
// fixed-order exp shared with the oracle, so trajectories and energies are bit-identical to the oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "philox.hpp"
#include "sk_kernels.hpp"

namespace rrrmc {

constexpr int kSpfMaxK = 8;
constexpr int kSpfDepth = 8;          // prefetch distance in iterations (even: two iterations share one Philox block)

struct SpfParams {
    const int32_t* A;       // [N][K]
    const double* J;        // [N][K]
    const int32_t* sites;   // [iters] sites of this launch's iterations
    uint32_t* spins;        // [W][NW][64]
    double* lf;             // [W][N][64]
    double* lfl;            // [W][N][64]
    int32_t* move_last;     // [Rpad], -1 = none
    double* E_cur;          // [Rpad]
    int64_t* acc_cur;       // [Rpad]
    double* Es;             // [nsamples][Rpad]; may be null
    double beta;
    uint64_t g0;            // iterations already consumed from the streams
    int64_t iters, step, sample0;
    int64_t it_off;         // iterations of this sampling call done by earlier launches (samples are taken at call-relative k*step)
    uint32_t k0, k1, replica0;
    int N, NW, Rpad;
};

__global__ __launch_bounds__(256) void spf_sites_kernel(int32_t* __restrict__ sites, int64_t n, uint64_t g0, uint32_t k0, uint32_t k1, uint32_t N)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < n) sites[t] = (int32_t)site_of(k0, k1, g0 + 1 + (uint64_t)t, N);
}

// initial spins (INIT stream, same bits as init_spins_kernel) in the lane-private layout. grid (ceil(NW/4), W), block 256
__global__ __launch_bounds__(256) void spf_init_spins_kernel(uint32_t* __restrict__ spins, int N, int NW, uint32_t replica0, uint32_t k0, uint32_t k1)
{
    const int lane = threadIdx.x & 63, q = blockIdx.x * 4 + (threadIdx.x >> 6), w = blockIdx.y;
    if (q >= NW) return;
    const uint32_t replica = replica0 + (uint32_t)(w * 64 + lane);
    uint32_t word = 0u;
    for (int j = 0; j < 32; ++j) {
        const int x = q * 32 + j;
        if (x < N) word |= ((init_spin_word(k0, k1, replica >> 5, (uint64_t)x) >> (replica & 31u)) & 1u) << j;
    }
    spins[((size_t)w * NW + q) * 64 + lane] = word;
}

// energy (RRG.jl:546-574 / EA.jl:584-611): rebuilds lfields, clears lfields_last and move_last. grid W, block 64
template <int K>
__global__ __launch_bounds__(64) void spf_energy_kernel(SpfParams P)
{
    const int lane = threadIdx.x, w = blockIdx.x, N = P.N;
    const int r = w * 64 + lane;
    double* lf = P.lf + (size_t)w * N * 64 + lane;
    double* lfl = P.lfl + (size_t)w * N * 64 + lane;
    const uint32_t* sp = P.spins + (size_t)w * P.NW * 64 + lane;
    double E1 = 0.0;
    for (int x = 0; x < N; ++x) {
        const int sx = 2 * (int)((sp[(size_t)(x >> 5) * 64] >> (x & 31)) & 1u) - 1;
        double f = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int y = P.A[(size_t)x * K + k];
            const int sy = 2 * (int)((sp[(size_t)(y >> 5) * 64] >> (y & 31)) & 1u) - 1;
            f = __dadd_rn(f, -__dmul_rn(__dmul_rn(P.J[(size_t)x * K + k], (double)sx), (double)sy));
        }
        E1 = __dadd_rn(E1, f);
        lf[(size_t)x * 64] = __dmul_rn(2.0, f);
        lfl[(size_t)x * 64] = 0.0;
    }
    P.E_cur[r] = __dmul_rn(E1, 0.5);
    P.move_last[r] = -1;
}

// grid W, block 64
template <int K>
__global__ __launch_bounds__(64) void spf_sweep_kernel(SpfParams P)
{
    constexpr int D = kSpfDepth;
    const int lane = threadIdx.x, w = blockIdx.x, N = P.N;
    const int r = w * 64 + lane;
    double* lf = P.lf + (size_t)w * N * 64 + lane;
    double* lfl = P.lfl + (size_t)w * N * 64 + lane;
    uint32_t* sp = P.spins + (size_t)w * P.NW * 64 + lane;
    const uint32_t replica = P.replica0 + (uint32_t)r;
    double E = P.E_cur[r];
    int64_t nacc = P.acc_cur[r];
    int32_t mlast = P.move_last[r];
    int64_t ns = P.sample0;
    const int64_t iters = P.iters, step = P.step;

    double pre[D];
    int ps[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        ps[d] = d < iters ? P.sites[d] : 0;
        pre[d] = d < iters ? lf[(size_t)ps[d] * 64] : 0.0;
    }
    Philox4 blk = {{0u, 0u, 0u, 0u}};
    uint64_t blk_id = ~0ull;

    for (int64_t base = 0; base < iters; base += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int64_t it = base + d + 1;
            if (it > iters) break;
            const uint64_t g = P.g0 + (uint64_t)it;
            if ((P.it_off + it) % step == 0) {               // sample before the move (RRRMC.jl:104-108)
                if (P.Es) P.Es[(size_t)ns * P.Rpad + r] = E;
                ++ns;
            }
            const int i = __builtin_amdgcn_readfirstlane(ps[d]);
            const double lfi = pre[d];
            const double dE = -lfi;                          // delta_energy: RRG.jl:619-625
            const double x = __dmul_rn(-P.beta, dE);
            if ((g >> 1) != blk_id) {                        // ACCEPT_F64 stream: one block serves iterations 2h, 2h+1
                blk_id = g >> 1;
                blk = philox4x32_10((uint32_t)blk_id, (uint32_t)(blk_id >> 32), replica, TAG_ACCEPT_F64, P.k0, P.k1);
            }
            const uint64_t u = (g & 1u) ? (((uint64_t)blk.w[2] << 32) | blk.w[3]) : (((uint64_t)blk.w[0] << 32) | blk.w[1]);
            const bool acc = x >= 0.0 || (double)(u >> 11) * 0x1.0p-53 < det_exp(x);       // accept: RRRMC.jl:39
            if (__builtin_amdgcn_ballot_w64(acc) != 0ull) {
                // update_cache! (RRG.jl:576-617, EA.jl:613-653) for the accepting lanes; neighbours are common to all lanes
                int y[K];
                double Jk[K], fy[K];
                uint32_t wy[K];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    y[k] = P.A[(size_t)i * K + k];
                    Jk[k] = P.J[(size_t)i * K + k];
                }
                const uint32_t wi = sp[(size_t)(i >> 5) * 64];
                const bool fast = acc && mlast == i;
                if (acc) {
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        wy[k] = sp[(size_t)(y[k] >> 5) * 64];
                        fy[k] = lf[(size_t)y[k] * 64];
                    }
                    const uint32_t snew = ((wi >> (i & 31)) & 1u) ^ 1u;
                    sp[(size_t)(i >> 5) * 64] = wi ^ (1u << (i & 31));
                    if (fast) {                              // exact undo: swap lfields <-> lfields_last on the (unique) neighbours
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            if (k > 0 && y[k] == y[k - 1]) continue;
                            const double t = lfl[(size_t)y[k] * 64];
                            lf[(size_t)y[k] * 64] = t;
                            lfl[(size_t)y[k] * 64] = fy[k];
                        }
                        lf[(size_t)i * 64] = -lfi;
                        lfl[(size_t)i * 64] = -lfl[(size_t)i * 64];
                    } else {
                        double v = 0.0;
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            const bool rep = k > 0 && y[k] == y[k - 1];      // GraphEA with L = 2: two bonds to the same neighbour
                            if (!rep) {
                                v = fy[k];
                                lfl[(size_t)y[k] * 64] = v;
                            }
                            // a flip of the neighbour itself cannot have happened in between: y != i
                            const uint32_t sy = (wy[k] >> (y[k] & 31)) & 1u;
                            const double c = (snew ^ sy) ? -4.0 : 4.0;       // 4 * sigma_xy with the NEW s_x
                            v = __dadd_rn(v, -__dmul_rn(c, Jk[k]));
                            const bool last = k == K - 1 || y[k + 1] != y[k];
                            if (last) lf[(size_t)y[k] * 64] = v;
                        }
                        lfl[(size_t)i * 64] = lfi;
                        lf[(size_t)i * 64] = -lfi;
                        mlast = i;
                    }
                    E = __dadd_rn(E, dE);
                    nacc += 1;
                }
                // prefetched fields that this move may have changed (closed neighbourhood of i): read them again
#pragma unroll
                for (int e = 0; e < D; ++e) {
                    if (e == d) continue;
                    const int64_t ite = e > d ? base + e + 1 : base + D + e + 1;
                    if (ite > iters) continue;
                    const int pe = __builtin_amdgcn_readfirstlane(ps[e]);
                    bool stale = pe == i;
#pragma unroll
                    for (int k = 0; k < K; ++k) stale |= pe == y[k];
                    if (stale) pre[e] = lf[(size_t)pe * 64];
                }
            }
            if (it + D <= iters) {
                ps[d] = P.sites[it + D - 1];
                pre[d] = lf[(size_t)ps[d] * 64];
            }
        }
    }
    P.E_cur[r] = E;
    P.acc_cur[r] = nacc;
    P.move_last[r] = mlast;
}

// lane-private spins -> bit words per replica in BitVector order, out[r][q] (q < NW); grid (ceil(NW/4), W), block 256
__global__ __launch_bounds__(256) void spf_spins_out_kernel(const uint32_t* __restrict__ spins, uint32_t* __restrict__ out, int NW, int NWout)
{
    const int lane = threadIdx.x & 63, q = blockIdx.x * 4 + (threadIdx.x >> 6), w = blockIdx.y;
    if (q >= NW) return;
    out[(size_t)(w * 64 + lane) * NWout + q] = spins[((size_t)w * NW + q) * 64 + lane];
}
__global__ __launch_bounds__(256) void spf_spins_in_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ spins, int NW, int NWin)
{
    const int lane = threadIdx.x & 63, q = blockIdx.x * 4 + (threadIdx.x >> 6), w = blockIdx.y;
    if (q >= NW) return;
    spins[((size_t)w * NW + q) * 64 + lane] = in[(size_t)(w * 64 + lane) * NWin + q];
}

// pm1dot between two lane-private snapshots; grid (W, npairs), block 64
__global__ __launch_bounds__(64) void overlap_lanes_kernel(const uint32_t* const* __restrict__ srcA, const uint32_t* const* __restrict__ srcB,
                                                           int N, int NW, int Rpad, int32_t* __restrict__ out)
{
    const int lane = threadIdx.x, w = blockIdx.x, p = blockIdx.y;
    const uint32_t* a = srcA[p] + (size_t)w * NW * 64 + lane;
    const uint32_t* b = srcB[p] + (size_t)w * NW * 64 + lane;
    int32_t c = 0;
    for (int q = 0; q < NW; ++q) c += __popc(a[(size_t)q * 64] ^ b[(size_t)q * 64]);
    out[(size_t)p * Rpad + w * 64 + lane] = N - 2 * c;
}

}  // namespace rrrmc
