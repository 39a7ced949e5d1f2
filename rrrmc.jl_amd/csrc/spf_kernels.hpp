// gfx950 kernels for the Float64-coupling sparse models GraphRRGNormal / GraphEANormal (SimpleGraph{Float64};
// src/graphs/RRG.jl:503-627, src/graphs/EA.jl:534-680) under standardMC (src/RRRMC.jl:81-127).  SURVEY.md §8f rank 3.
//
// Float64 fields cannot be bit-sliced, so this is the HBM-bound picture of the north star: one lane per replica,
// replica-minor arrays so that a wavefront's access to one site is one coalesced 512-byte line:
//   lf    : [W][N][64] Float64   local fields
//   spins : [W][N] uint64        bit l of word (w, x) = spin x of lane l's replica (one 8-byte broadcast load per site)
//   undo  : [W][K+2][64] Float64 the part of lfields_last that update_cache!'s undo path (RRG.jl:583-593) can ever read:
//           the K neighbour fields and the own field saved by the LAST accepted move of the replica, plus move_last.
//           (All other entries of the reference's lfields_last are dead: the swap happens only when move_last == move.)
// W = Rpad/64 wavefronts, one per workgroup; occupancy hides the latency of the accept path, so the kernel wants many
// replicas: 256 Ki replicas of N = 4096 are 8.6 GB of fields — 288 GB of HBM hold millions.  All replicas attempt the
// same site (SITE stream, precomputed by spf_sites_kernel); the local field of the next D sites is prefetched D
// iterations ahead and re-read only when an accepted move touched it.
// Every field receives exactly the reference's sequence of IEEE operations (no FMA contraction) and det_exp is the
// fixed-order exp shared with the oracle, so trajectories and energies are bit-identical to the oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "philox.hpp"
#include "sk_kernels.hpp"
#include "spf_params.hpp"

namespace rrrmc {

__global__ __launch_bounds__(256) void spf_sites_kernel(int32_t* __restrict__ sites, int64_t n, uint64_t g0, uint32_t k0, uint32_t k1, uint32_t N)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < n) sites[t] = (int32_t)site_of(k0, k1, g0 + 1 + (uint64_t)t, N);
}

__device__ __forceinline__ unsigned long long spf_load_spins(const unsigned long long* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// initial spins (INIT stream, same bits as init_spins_kernel). grid (ceil(N/256), W), block 256
__global__ __launch_bounds__(256) void spf_init_spins_kernel(unsigned long long* __restrict__ spins, int N, uint32_t replica0, uint32_t k0, uint32_t k1)
{
    const int x = blockIdx.x * 256 + threadIdx.x, w = blockIdx.y;
    if (x >= N) return;
    unsigned long long word = 0ull;
    for (int l = 0; l < 64; ++l) {
        const uint32_t replica = replica0 + (uint32_t)(w * 64 + l);
        word |= (unsigned long long)((init_spin_word(k0, k1, replica >> 5, (uint64_t)x) >> (replica & 31u)) & 1u) << l;
    }
    spins[(size_t)w * N + x] = word;
}

// energy (RRG.jl:546-574 / EA.jl:584-611): rebuilds lfields and clears the undo record. grid W, block 64
template <int K>
__global__ __launch_bounds__(64) void spf_energy_kernel(SpfParams P)
{
    const int lane = threadIdx.x, w = blockIdx.x, N = P.N;
    const int r = w * 64 + lane;
    double* lf = P.lf + (size_t)w * N * 64 + lane;
    const unsigned long long* sp = P.spins + (size_t)w * N;
    double E1 = 0.0;
    for (int x = 0; x < N; ++x) {
        const int sx = 2 * (int)((sp[x] >> lane) & 1ull) - 1;
        double f = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int y = P.A[(size_t)x * K + k];
            const int sy = 2 * (int)((sp[y] >> lane) & 1ull) - 1;
            f = __dadd_rn(f, -__dmul_rn(__dmul_rn(P.J[(size_t)x * K + k], (double)sx), (double)sy));
        }
        E1 = __dadd_rn(E1, f);
        lf[(size_t)x * 64] = __dmul_rn(2.0, f);
    }
    P.E_cur[r] = __dmul_rn(E1, 0.5);
    P.move_last[r] = -1;
}

// The same in two steps for calls that are short beside it (round 4: one wavefront per group walking all N sites took 1.0 ms at N = 4096,
// 8 % of a 2^16-iteration call of spf_team_kernel): the fields of all sites side by side, then the energy as the reference's sequential sum.
// lfields[x] = 2 lf is exact, so lf = lfields[x] / 2 is the summand of RRG.jl:546-574 bit for bit.
// grid W * ceil(N / 4) (one dimension: W may exceed the 65 536 blocks of the others), block 256: one wavefront per site
template <int K>
__global__ __launch_bounds__(256) void spf_fields_kernel(SpfParams P)
{
    const int N = P.N, nb = (N + 3) / 4;
    const int lane = threadIdx.x & 63, w = (int)(blockIdx.x / (unsigned)nb), x = (int)(blockIdx.x % (unsigned)nb) * 4 + (int)(threadIdx.x >> 6);
    if (x >= N) return;
    const unsigned long long* sp = P.spins + (size_t)w * N;
    const int sx = 2 * (int)((sp[x] >> lane) & 1ull) - 1;
    double f = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int y = P.A[(size_t)x * K + k];
        const int sy = 2 * (int)((sp[y] >> lane) & 1ull) - 1;
        f = __dadd_rn(f, -__dmul_rn(__dmul_rn(P.J[(size_t)x * K + k], (double)sx), (double)sy));
    }
    P.lf[((size_t)w * N + x) * 64 + lane] = __dmul_rn(2.0, f);
}

// grid W, block 64
__global__ __launch_bounds__(64) void spf_energy_sum_kernel(SpfParams P)
{
    const int lane = threadIdx.x, w = blockIdx.x, N = P.N;
    const int r = w * 64 + lane;
    const double* lf = P.lf + (size_t)w * N * 64 + lane;
    double E1 = 0.0;
    int x = 0;
    // the sum is the reference's sequential one (its order is fixed); what can be wide is the number of lines in flight per wavefront
    for (; x + 32 <= N; x += 32) {
        double v[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) v[q] = lf[(size_t)(x + q) * 64];
#pragma unroll
        for (int q = 0; q < 32; ++q) E1 = __dadd_rn(E1, __dmul_rn(v[q], 0.5));
    }
    for (; x + 8 <= N; x += 8) {
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = lf[(size_t)(x + q) * 64];
#pragma unroll
        for (int q = 0; q < 8; ++q) E1 = __dadd_rn(E1, __dmul_rn(v[q], 0.5));
    }
    for (; x < N; ++x) E1 = __dadd_rn(E1, __dmul_rn(lf[(size_t)x * 64], 0.5));
    P.E_cur[r] = __dmul_rn(E1, 0.5);
    P.move_last[r] = -1;
}

// One wavefront = 64 replicas.  Everything an iteration needs is requested ahead of time:
//   local field of the attempted site          kSpfDepth iterations ahead  (pre[])
//   neighbour table row (A, J: scalar loads)   kSpfNb + 1 iterations ahead (staging SGPRs)
//   neighbour fields / spin words               kSpfNb iterations ahead     (nb slots), speculatively: they are only used
//                                               by accepting lanes, but some lane accepts in almost every iteration
// so the accept path never waits for a dependent load.  An accepted move at site i invalidates what was requested for later
// iterations only inside the closed neighbourhood of i; `pend` keeps the site ids of all outstanding requests one per lane,
// K+1 compares find the intersecting slots and those (rare, ~D K^2 / N) are re-read after the stores, in program order.
// grid W, block 64
template <int K>
__global__ __launch_bounds__(64) void spf_sweep_kernel(SpfParams P)
{
    constexpr int D = kSpfDepth, NB = kSpfNb;
    static_assert(D % NB == 0 && D % 2 == 0 && NB * (K + 1) <= 64 - D, "slot geometry");
    const int lane = threadIdx.x, w = blockIdx.x, N = P.N;
    const int r = w * 64 + lane;
    double* lf = P.lf + (size_t)w * N * 64 + lane;
    double* undo = P.undo + (size_t)w * (K + 1) * 64 + lane;
    unsigned long long* sp = P.spins + (size_t)w * N;
    const uint32_t replica = P.replica0 + (uint32_t)r;
    double E = P.E_cur[r];
    int64_t nacc = P.acc_cur[r];
    int32_t mlast = P.move_last[r];
    double sv[K + 1];
#pragma unroll
    for (int k = 0; k <= K; ++k) sv[k] = undo[(size_t)k * 64];
    int64_t ns = P.sample0;
    const int64_t iters = P.iters, step = P.step;

    // outstanding requests.  Lane f*(K+1)+j of `pend` (j = 0: site, j >= 1: neighbour j-1) describes nb slot f,
    // lane 64-D+e the site of pre slot e; -1 = nothing outstanding.
    int pend = -1;
    double pre[D];
    int ps[D];
    int ny[NB][K];
    double nJ[NB][K], nf[NB][K];
    unsigned long long nw[NB][K], nwi[NB];
    int sy[K];                 // staged neighbour row of the iteration NB+1 ahead
    double sJ[K];

    auto load_row = [&](int site, int (&yy)[K], double (&JJ)[K]) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            yy[k] = P.A[(size_t)site * K + k];
            JJ[k] = P.J[(size_t)site * K + k];
        }
    };
    auto request_nb = [&](int f, int site) {      // (re)issue the speculative loads of nb slot f from its row
        nwi[f] = spf_load_spins(sp + site);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            nw[f][k] = spf_load_spins(sp + ny[f][k]);
            nf[f][k] = lf[(size_t)ny[f][k] * 64];
        }
    };
    auto mark_nb = [&](int f, int site) {
        pend = lane == (f * (K + 1)) ? (site) : pend;
#pragma unroll
        for (int k = 0; k < K; ++k) pend = lane == (f * (K + 1) + 1 + k) ? (ny[f][k]) : pend;
    };

    // prologue: sites and fields of iterations 1..D, neighbour data of iterations 1..NB, staged row of iteration NB+1
    // (P.sites holds iters + 2 D entries)
#pragma unroll
    for (int d = 0; d < D; ++d) {
        ps[d] = __builtin_amdgcn_readfirstlane(P.sites[d]);
        pre[d] = lf[(size_t)ps[d] * 64];
        pend = lane == (64 - D + d) ? (ps[d]) : pend;
    }
#pragma unroll
    for (int f = 0; f < NB; ++f) {
        load_row(ps[f], ny[f], nJ[f]);
        request_nb(f, ps[f]);
        mark_nb(f, ps[f]);
    }
    load_row(ps[NB % D], sy, sJ);
    int snext = P.sites[D];                                       // site of iteration D+1, consumed by the refill of iteration 1
    Philox4 blk = {{0u, 0u, 0u, 0u}};
    uint64_t blk_id = ~0ull;

    for (int64_t base = 0; base < iters; base += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            constexpr int dummy = 0; (void)dummy;
            const int f = d % NB;
            const int64_t it = base + d + 1;
            if (it > iters) break;
            const uint64_t g = P.g0 + (uint64_t)it;
            if ((P.it_off + it) % step == 0) {               // sample before the move (RRRMC.jl:104-108)
                if (P.Es) P.Es[(size_t)ns * P.Rpad + r] = E;
                ++ns;
            }
            const int i = ps[d];
            const double lfi = pre[d];
            const double dE = -lfi;                          // delta_energy: RRG.jl:619-625
            const double x = __dmul_rn(-P.beta, dE);
            if ((g >> 1) != blk_id) {                        // ACCEPT_F64 stream: one block serves iterations 2h, 2h+1
                blk_id = g >> 1;
                blk = philox4x32_10((uint32_t)blk_id, (uint32_t)(blk_id >> 32), replica, TAG_ACCEPT_F64, P.k0, P.k1);
            }
            const uint64_t u = (g & 1u) ? (((uint64_t)blk.w[2] << 32) | blk.w[3]) : (((uint64_t)blk.w[0] << 32) | blk.w[1]);
            const bool acc = x >= 0.0 || (double)(u >> 11) * 0x1.0p-53 < det_exp(x);       // accept: RRRMC.jl:39
            const unsigned long long amask = __builtin_amdgcn_ballot_w64(acc);
            // this iteration's requests are consumed now
            pend = lane == (64 - D + d) ? (-1) : pend;
#pragma unroll
            for (int j = 0; j <= K; ++j) pend = lane == (f * (K + 1) + j) ? (-1) : pend;
            {
                // update_cache! (RRG.jl:576-617, EA.jl:613-653); neighbours are common to all lanes.  No branch around a memory
                // operation: every lane stores to every neighbour line — lanes that do not accept write the value they read back
                // (full 512-byte lines instead of masked partial ones) — so the compiler can count the outstanding requests
                // (s_waitcnt vmcnt(n)) instead of draining them, and the prefetch distances above survive.
                const unsigned long long wi = nwi[f];
                __hip_atomic_store(sp + i, wi ^ amask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);   // same value from every lane
                const uint32_t snew = (uint32_t)((wi >> lane) & 1ull) ^ 1u;
                const bool fast = acc && mlast == i;         // exact undo: swap lfields <-> lfields_last on the (unique) neighbours
                const bool slow = acc && !fast;
                double v = 0.0, svrun = 0.0, frun = 0.0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const bool rep = k > 0 && ny[f][k] == ny[f][k - 1];       // GraphEA with L = 2: two bonds to the same neighbour
                    if (!rep) {                                               // wave-uniform
                        frun = nf[f][k];
                        svrun = sv[k];
                        v = frun;
                        sv[k] = acc ? frun : svrun;                           // lfields_last[y] = lfields[y] (slow) / the swap (fast)
                    }
                    const uint32_t sbit = (uint32_t)((nw[f][k] >> lane) & 1ull);
                    const double c = (snew ^ sbit) ? -4.0 : 4.0;             // 4 * sigma_xy with the NEW s_x
                    v = __dadd_rn(v, -__dmul_rn(c, nJ[f][k]));
                    // the last store of a run of equal neighbours wins (same wavefront, same address: in order)
                    lf[(size_t)ny[f][k] * 64] = fast ? svrun : (slow ? v : frun);
                }
                lf[(size_t)i * 64] = acc ? -lfi : lfi;
                sv[K] = fast ? -sv[K] : (slow ? lfi : sv[K]);
                mlast = slow ? i : mlast;
                E = acc ? __dadd_rn(E, dE) : E;
                nacc += acc ? 1 : 0;
                // outstanding requests inside the closed neighbourhood of i are stale: read them again (after the stores)
                unsigned long long hit = __builtin_amdgcn_ballot_w64(pend == i);
#pragma unroll
                for (int k = 0; k < K; ++k) hit |= __builtin_amdgcn_ballot_w64(pend == ny[f][k]);
                if (amask != 0ull && hit != 0ull) {
#pragma unroll
                    for (int e = 0; e < D; ++e)
                        if (e != d && ((hit >> (64 - D + e)) & 1ull)) pre[e] = lf[(size_t)ps[e] * 64];
#pragma unroll
                    for (int h = 0; h < NB; ++h) {
                        if (h == f) continue;
                        if ((hit >> (h * (K + 1))) & ((1ull << (K + 1)) - 1ull)) {
                            // slot h serves iteration it + ((h - f + NB) % NB): its site is ps[(d + (h - f + NB) % NB) % D]
                            request_nb(h, ps[(d + (h - f + NB) % NB) % D]);
                        }
                    }
                }
            }
            // refill (unconditional: the host pads the site stream with the D following iterations, whose requests are simply
            // never consumed): neighbour data of iteration it+NB from the staged row, the row of iteration it+NB+1, field of it+D
            {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    // the staged row was loaded one iteration ago into VGPRs; it becomes scalar only HERE (the empty asm pins the
                    // value to a VGPR until now, otherwise the compiler scalarises — and waits for — it right after the load)
                    int yv = sy[k];
                    asm volatile("" : "+v"(yv));
                    ny[f][k] = __builtin_amdgcn_readfirstlane(yv);
                    nJ[f][k] = sJ[k];
                }
                request_nb(f, ps[(d + NB) % D]);
                mark_nb(f, ps[(d + NB) % D]);
            }
            load_row(ps[(d + NB + 1) % D], sy, sJ);
            {
                int sv_next = snext;                                  // loaded one iteration ago: no dependent load chain here
                asm volatile("" : "+v"(sv_next));
                ps[d] = __builtin_amdgcn_readfirstlane(sv_next);
                pre[d] = lf[(size_t)ps[d] * 64];
                pend = lane == (64 - D + d) ? (ps[d]) : pend;
                snext = P.sites[it + D];
            }
        }
    }
    P.E_cur[r] = E;
    P.acc_cur[r] = nacc;
    P.move_last[r] = mlast;
#pragma unroll
    for (int k = 0; k <= K; ++k) undo[(size_t)k * 64] = sv[k];
}

// pm1dot between two snapshots in the [W][N] uint64 layout; grid (W, npairs), block 64
__global__ __launch_bounds__(64) void overlap_lanes_kernel(const unsigned long long* const* __restrict__ srcA,
                                                           const unsigned long long* const* __restrict__ srcB,
                                                           int N, int Rpad, int32_t* __restrict__ out)
{
    const int lane = threadIdx.x, w = blockIdx.x, p = blockIdx.y;
    const unsigned long long* a = srcA[p] + (size_t)w * N;
    const unsigned long long* b = srcB[p] + (size_t)w * N;
    int32_t c = 0;
    for (int x0 = 0; x0 < N; x0 += 64) {                       // lane j loads word x0 + j, then every lane picks its bit
        const int x = x0 + lane;
        const unsigned long long v = x < N ? (a[x] ^ b[x]) : 0ull;
        const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
#pragma unroll 8
        for (int j = 0; j < 64; ++j) {
            const uint32_t wl = (uint32_t)__builtin_amdgcn_readlane((int)lo, j), wh = (uint32_t)__builtin_amdgcn_readlane((int)hi, j);
            c += (int32_t)(((lane < 32 ? wl : wh) >> (lane & 31)) & 1u);
        }
    }
    out[(size_t)p * Rpad + w * 64 + lane] = N - 2 * c;
}

}  // namespace rrrmc
