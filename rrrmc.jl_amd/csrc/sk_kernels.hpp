// gfx950 kernels for the dense Gaussian SK model GraphSKNormal (src/graphs/SK.jl:170-297) under standardMC
// (src/RRRMC.jl:81-127).  Float64 couplings and local fields; delta_energy(i) = +lfields[i] (SK.jl:278-284).
//
// Layout: one workgroup of 256 threads owns kSkRB = 8 replicas; thread t holds the local fields (and the undo copy
// lfields_last, SK.jl:247-250) of the sites j = q*256 + t, q < SPT, for its 8 replicas in REGISTERS, plus their spin
// bits.  All replicas attempt the same site (SITE stream); the J row of that site (8 KiB at N = 1024, shared by every
// workgroup, L2-resident) is prefetched one step ahead.  Per step: the site's owner thread publishes the 8 fields,
// 8 lanes decide (rand53 < exp(-beta dE), deterministic exp shared with the oracle), every thread applies
// lf[j] += 4 sigma_ij J_ij for the accepted replicas.  Each field receives exactly the reference's sequence of
// floating-point operations, so trajectories and energies are bit-identical to the oracle, not merely within 1e-6.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "philox.hpp"
#include "det_math.hpp"

namespace rrrmc {

constexpr int kSkThreads = 256;
constexpr int kSkRB = 8;              // replicas per workgroup
constexpr int kSkMaxSPT = 8;          // sites per thread -> N <= 2048
struct SkParams {
    const double* J;        // [N][N]
    double* lf;             // [G][N][kSkRB]  local fields (replica fastest)
    double* lfl;            // [G][N][kSkRB]  lfields_last
    int32_t* move_last;     // [G][kSkRB]     -1 = none
    uint8_t* spins;         // [G][N]         bit r = spin of replica 8*group + r
    double* E_cur;          // [G * kSkRB]
    int64_t* acc_cur;       // [G * kSkRB]
    double* Es;             // [nsamples][G * kSkRB]; may be null
    double beta;
    uint64_t g0;            // iterations already consumed from the streams
    int64_t iters, step, sample0;
    uint32_t k0, k1, replica0;
    int N;
};

// Workgroup barrier for exchanges that go through LDS only: waits for this wavefront's LDS operations, not for its global loads —
// __syncthreads() also drains the vector-memory counter, i.e. it would wait for the J-row prefetch at every barrier of the step.
__device__ __forceinline__ void sk_lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// NTH threads per workgroup (256, 512 or 1024): a wavefront issues an instruction every ~5 cycles whatever the others do, and a step is a
// chain of ~700 of them at 256 threads (most in the field update) — more wavefronts with fewer sites each shorten the chain.
template <int SPT, int NTH>
__global__ __launch_bounds__(NTH) void sk_sweep_kernel(SkParams P)
{
    __shared__ double sh_lfi[2][kSkRB];
    __shared__ uint32_t sh_si[2], sh_acc[2], sh_swap[2];
    const int tid = threadIdx.x, N = P.N;
    const int grp = blockIdx.x, Rp = gridDim.x * kSkRB;
    double lf[SPT][kSkRB], lfl[SPT][kSkRB];
    uint32_t sb[SPT];
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        const int j = q * NTH + tid;
        sb[q] = j < N ? P.spins[(size_t)grp * N + j] : 0u;
#pragma unroll
        for (int r = 0; r < kSkRB; ++r) {
            lf[q][r] = j < N ? P.lf[((size_t)grp * N + j) * kSkRB + r] : 0.0;
            lfl[q][r] = j < N ? P.lfl[((size_t)grp * N + j) * kSkRB + r] : 0.0;
        }
    }
    // lanes 0..7 of wave 0 each own one replica's scalar state
    double E_run = 0.0;
    int64_t A_run = 0;
    int32_t mlast = -1;
    if (tid < kSkRB) { E_run = P.E_cur[grp * kSkRB + tid]; A_run = P.acc_cur[grp * kSkRB + tid]; mlast = P.move_last[grp * kSkRB + tid]; }
    int64_t ns = P.sample0;

    // The SITE stream and the ACCEPT_F64 uniforms do not depend on the state: they are drawn 64 iterations at a time (lane = iteration)
    // by the waves that are not on the decision's critical path — wave 1 the sites, waves 2 and 3 the 8 x 64 uniforms — into LDS, one
    // block ahead; a step then reads its site and its uniform instead of running two Philox blocks between its barriers.
    __shared__ uint32_t sh_site[2][64];
    __shared__ double sh_u[2][64][kSkRB];
    auto prepare = [&](int64_t blk) {          // iterations blk * 64 + 1 .. blk * 64 + 64 -> buffer blk & 1
        const int buf = (int)(blk & 1);
        if (tid >= 64 && tid < 128)
            sh_site[buf][tid - 64] = site_of(P.k0, P.k1, P.g0 + (uint64_t)(blk * 64 + (tid - 64) + 1), (uint32_t)N);
        else if (tid >= 128 && tid < 256) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = (tid - 128) + 128 * k, l = idx >> 3, r = idx & 7;
                sh_u[buf][l][r] = rand53(P.k0, P.k1, P.g0 + (uint64_t)(blk * 64 + l + 1), P.replica0 + (uint32_t)(grp * kSkRB + r));
            }
        }
    };
    prepare(0);
    __syncthreads();
    double Jq[SPT], Jn[SPT];
    uint32_t site = sh_site[0][0];
#pragma unroll
    for (int q = 0; q < SPT; ++q) { const int j = q * NTH + tid; Jq[q] = j < N ? P.J[(size_t)site * N + j] : 0.0; }

    long long next_sample = P.step;          // iterations k * step: a counter instead of a 64-bit modulo per iteration
    double expc[17];
    sk_exp_constants(expc);
#ifdef RRRMC_SK_STAMPS
    uint64_t st[5] = {0, 0, 0, 0, 0}, st2[2] = {0, 0};
#endif
    for (int64_t it = 1; it <= P.iters; ++it) {
#ifdef RRRMC_SK_STAMPS
        const uint64_t t0 = __builtin_amdgcn_s_memtime();
#endif
        const int b = (int)(it & 1);
        const int li = (int)((it - 1) & 63), bi = (int)(((it - 1) >> 6) & 1);        // this iteration in its block
        const int qi = (int)(site / (uint32_t)NTH), owner = (int)(site % (uint32_t)NTH);
        if (tid == owner) {
#pragma unroll
            for (int q = 0; q < SPT; ++q)
                if (q == qi) {
#pragma unroll
                    for (int r = 0; r < kSkRB; ++r) sh_lfi[b][r] = lf[q][r];
                    sh_si[b] = sb[q];
                }
        }
        // What does not depend on the state — the next block of stream values (li = 0; its buffer was last read a step ago, before two
        // barriers) and the prefetch of the next site's J row (it stays in flight across the barriers: sk_lds_barrier does not wait for
        // it) — is done by the deciding wavefront before the barrier and by the others after it, while they would wait for the decision:
        // the barrier then only waits for the owner's publication.
        uint32_t site_n = 0u;
        auto stateless = [&]() {
            if (li == 0) prepare(((it - 1) >> 6) + 1);
            site_n = li == 63 ? sh_site[bi ^ 1][0] : sh_site[bi][li + 1];
#pragma unroll
            for (int q = 0; q < SPT; ++q) { const int j = q * NTH + tid; Jn[q] = j < N ? P.J[(size_t)site_n * N + j] : 0.0; }
        };
        if (tid < 64) stateless();
#ifdef RRRMC_SK_STAMPS
        const uint64_t t1 = __builtin_amdgcn_s_memtime();
#endif
        sk_lds_barrier();
#ifdef RRRMC_SK_STAMPS
        const uint64_t t2 = __builtin_amdgcn_s_memtime();
#endif
        if (tid >= 64) stateless();
        if (tid < 64) {          // the first wave: lanes 0..7 decide, the whole wave ballots
            bool acc = false, swp = false;
            if (tid < kSkRB) {
                if (it == next_sample) { next_sample += P.step;          // sample BEFORE the move (RRRMC.jl:104-108)
                    if (P.Es) P.Es[ns * Rp + grp * kSkRB + tid] = E_run;
                    ns += 1;
                }
                const double dE = sh_lfi[b][tid];                       // delta_energy, SK.jl:278-284
                const double x = -P.beta * dE;
                acc = (x >= 0.0) || (sh_u[bi][li][tid] < det_exp_c(x, expc));    // RRRMC.jl:39
                swp = acc && (mlast == (int32_t)site);                  // undo path of update_cache!, SK.jl:247-250
                if (acc) { E_run += dE; A_run += 1; mlast = (int32_t)site; }
            }
            const unsigned long long ba = __ballot(acc), bs = __ballot(swp);
            if (tid == 0) { sh_acc[b] = (uint32_t)ba; sh_swap[b] = (uint32_t)bs; }
        }
#ifdef RRRMC_SK_STAMPS
        const uint64_t t3 = __builtin_amdgcn_s_memtime();
#endif
        sk_lds_barrier();
#ifdef RRRMC_SK_STAMPS
        const uint64_t t4 = __builtin_amdgcn_s_memtime();
#endif
        const uint32_t accm = sh_acc[b], swpm = sh_swap[b], si_old = sh_si[b];
        const uint32_t normal = accm & ~swpm;
        const uint32_t si_new = si_old ^ accm;
#ifdef RRRMC_SK_STAMPS
        asm volatile("" :: "s"(accm), "s"(swpm), "s"(si_old));
        const uint64_t t4a = __builtin_amdgcn_s_memtime();
#endif

        if (swpm) {          // workgroup-uniform and rare: swap lfields <-> lfields_last of those replicas.  Selects, not a branch per
                             // replica: the branches made the compiler shuttle all the fields through temporaries on the path WITHOUT a swap
#pragma unroll
            for (int r = 0; r < kSkRB; ++r) {
                const bool sw = (swpm >> r) & 1u;
#pragma unroll
                for (int q = 0; q < SPT; ++q) { const double a0 = lf[q][r], b0 = lfl[q][r]; lf[q][r] = sw ? b0 : a0; lfl[q][r] = sw ? a0 : b0; }
            }
        }
        if (normal) {        // workgroup-uniform
            // replica outermost: one scalar branch per replica (8 per step) instead of one per (site, replica) pair
            double d4[SPT];
            uint32_t diff[SPT];
#pragma unroll
            for (int q = 0; q < SPT; ++q) {
                d4[q] = 4.0 * Jq[q];
                diff[q] = si_new ^ sb[q];                               // bit r set: s_i != s_j for replica r -> sigma = -1
            }
#pragma unroll
            for (int r = 0; r < kSkRB; ++r)
                if ((normal >> r) & 1u) {          // (a scalar branch per replica: a third of them move)
#pragma unroll
                    for (int q = 0; q < SPT; ++q) {
                        const double old = lf[q][r];
                        lfl[q][r] = old;
                        // lfields[j] = lfj + 4*J*sigma, SK.jl:256-262; the sign goes into bit 63 of 4 J (x + (-y) is the same operation as the select of -d4)
                        const double dl = __longlong_as_double(__double_as_longlong(d4[q]) ^ ((long long)((diff[q] >> r) & 1u) << 63));
                        lf[q][r] = old + dl;
                    }
                }
#ifdef RRRMC_SK_STAMPS
            st2[1] += __builtin_amdgcn_s_memtime() - t4a;
#endif
            if (tid == owner) {
                // the eight published fields in one go (a read per accepted replica, each behind its own branch, made the owner's
                // wavefront the one everybody waits for at the next barrier; reading them in every thread up front costs more than
                // it hides), then selects
                double lfm[kSkRB];
#pragma unroll
                for (int r = 0; r < kSkRB; ++r) lfm[r] = sh_lfi[b][r];
#pragma unroll
                for (int q = 0; q < SPT; ++q)
                    if (q == qi) {
#pragma unroll
                        for (int r = 0; r < kSkRB; ++r) {
                            const bool on = (normal >> r) & 1u;
                            lfl[q][r] = on ? lfm[r] : lfl[q][r];                                                      // SK.jl:263-264
                            lf[q][r] = on ? -lfm[r] : lf[q][r];
                        }
                    }
            }
        }
        if (tid == owner) {
#pragma unroll
            for (int q = 0; q < SPT; ++q)
                if (q == qi) sb[q] ^= accm;                              // spinflip!, Interface.jl:89-92
        }
        site = site_n;
#pragma unroll
        for (int q = 0; q < SPT; ++q) Jq[q] = Jn[q];
#ifdef RRRMC_SK_STAMPS
        st2[0] += t4a - t4;
        { const uint64_t t5 = __builtin_amdgcn_s_memtime(); st[0] += t1 - t0; st[1] += t2 - t1; st[2] += t3 - t2; st[3] += t4 - t3; st[4] += t5 - t4; }
#endif
    }

#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        const int j = q * NTH + tid;
        if (j < N) {
            P.spins[(size_t)grp * N + j] = (uint8_t)sb[q];
#pragma unroll
            for (int r = 0; r < kSkRB; ++r) {
                P.lf[((size_t)grp * N + j) * kSkRB + r] = lf[q][r];
                P.lfl[((size_t)grp * N + j) * kSkRB + r] = lfl[q][r];
            }
        }
    }
#ifdef RRRMC_SK_STAMPS
    if (grp == 1 && (tid == 0 || tid == 200))
        printf("sk stamps tid %d iters %lld: publish %llu barA %llu decide %llu barB %llu apply %llu\n", tid, (long long)P.iters,
               (unsigned long long)st[0], (unsigned long long)st[1], (unsigned long long)st[2], (unsigned long long)st[3], (unsigned long long)st[4]);
    if (grp == 1 && (tid == 0 || tid == 200)) printf("   update: lds read %llu, fields (when a move was accepted) %llu\n", (unsigned long long)st2[0], (unsigned long long)st2[1]);
#endif
    if (tid < kSkRB) { P.E_cur[grp * kSkRB + tid] = E_run; P.acc_cur[grp * kSkRB + tid] = A_run; P.move_last[grp * kSkRB + tid] = mlast; }
}

// energy(X, C) of SK.jl:212-237 with the reference's summation order: thread = (site i, replica r) sums over j in order;
// then one thread per replica accumulates n -= lf over i in order.  Also resets the undo state.
__global__ __launch_bounds__(256) void sk_fields_kernel(const double* __restrict__ J, const uint8_t* __restrict__ spins,
                                                        double* __restrict__ lf, double* __restrict__ lfl,
                                                        int32_t* __restrict__ move_last, int N)
{
    const int grp = blockIdx.y;
    const int i = blockIdx.x * 32 + (threadIdx.x >> 3), r = threadIdx.x & 7;
    if (i >= N) return;
    const uint8_t* sp = spins + (size_t)grp * N;
    const uint32_t si = (sp[i] >> r) & 1u;
    double acc = 0.0;
    for (int j = 0; j < N; ++j) {
        const uint32_t sj = (sp[j] >> r) & 1u;
        const double Jij = J[(size_t)j * N + i];          // = J[i][j]: symmetric, read transposed for coalescing
        acc += (si ^ sj) ? -Jij : Jij;                    // (1 - 2(si xor sj)) * Ji[j]
    }
    lf[((size_t)grp * N + i) * kSkRB + r] = 2 * acc;
    lfl[((size_t)grp * N + i) * kSkRB + r] = 0.0;
    if (i == 0) move_last[grp * kSkRB + r] = -1;
}

// The same fields with one thread per SITE and the 8 replicas of the group in registers (round 4: the kernel above is one dependent chain of
// N additions per thread and took 1.4 ms at N = 1024, 2048 replicas — a quarter of a 16 384-iteration call of sk_hblock_kernel): eight
// independent chains per thread, one coalesced load of J and one byte of spins (the same for the whole workgroup) per j, the sign applied
// to the high word.  Every (site, replica) still sums over j in the reference's order (SK.jl:212-237).  grid (ceil(N / 256), G8), block 256
__global__ __launch_bounds__(256) void sk_fields8_kernel(const double* __restrict__ J, const uint8_t* __restrict__ spins,
                                                         double* __restrict__ lf, double* __restrict__ lfl,
                                                         int32_t* __restrict__ move_last, int N)
{
    const int grp = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const uint8_t* sp = spins + (size_t)grp * N;
    if (blockIdx.x == 0 && threadIdx.x < kSkRB) move_last[grp * kSkRB + threadIdx.x] = -1;
    if (i >= N) return;
    const uint32_t si = sp[i];
    double acc[kSkRB];
#pragma unroll
    for (int r = 0; r < kSkRB; ++r) acc[r] = 0.0;
#pragma unroll 4
    for (int j = 0; j < N; ++j) {
        const uint32_t x = si ^ (uint32_t)sp[j];                 // bit r: the spins of i and j differ in replica r
        const double Jij = J[(size_t)j * N + i];                 // = J[i][j]: symmetric, read transposed for coalescing
        const unsigned long long jb = (unsigned long long)__double_as_longlong(Jij);
#pragma unroll
        for (int r = 0; r < kSkRB; ++r)
            acc[r] += __longlong_as_double((long long)(jb ^ ((unsigned long long)((x >> r) & 1u) << 63)));      // (1 - 2 (si xor sj)) * Ji[j]
    }
    double* o = lf + ((size_t)grp * N + i) * kSkRB;
    double* ol = lfl + ((size_t)grp * N + i) * kSkRB;
#pragma unroll
    for (int r = 0; r < kSkRB; ++r) { o[r] = 2 * acc[r]; ol[r] = 0.0; }
}

__global__ __launch_bounds__(64) void sk_energy_kernel(const double* __restrict__ lf, double* __restrict__ E_out, int N, int Rp)
{
    const int rep = blockIdx.x * 64 + threadIdx.x;
    if (rep >= Rp) return;
    const int grp = rep / kSkRB, r = rep % kSkRB;
    double n = 0.0;
    for (int i = 0; i < N; ++i) n -= lf[((size_t)grp * N + i) * kSkRB + r] * 0.5;     // n -= lf_half, SK.jl:231 (lfields = 2 lf_half, exact)
    E_out[rep] = n / 2;
}

__global__ __launch_bounds__(256) void sk_init_spins_kernel(uint8_t* __restrict__ spins, int N, uint32_t replica0, uint32_t k0, uint32_t k1)
{
    // INIT stream: bit of replica rho at site x = bit (rho & 31) of word (x & 3) of ctr (x >> 2, 0, rho >> 5, TAG_INIT)
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= N) return;
    const uint32_t rho0 = replica0 + blockIdx.y * kSkRB;                 // 8 consecutive replicas: same 32-group (replica0 % 32 == 0)
    const uint32_t w = init_spin_word(k0, k1, rho0 >> 5, (uint64_t)x);
    spins[(size_t)blockIdx.y * N + x] = (uint8_t)((w >> (rho0 & 31u)) & 0xffu);
}

__global__ __launch_bounds__(256) void transpose_es_f64_kernel(const double* __restrict__ Es, double* __restrict__ out, int64_t nsamp, int Rp, int R)
{
    __shared__ double tile[32][33];
    const int64_t s0 = (int64_t)blockIdx.x * 32;
    const int r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int64_t s = s0 + j;
        tile[j][tx] = (s < nsamp && r0 + tx < Rp) ? Es[s * Rp + r0 + tx] : 0.0;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j;
        const int64_t s = s0 + tx;
        if (r < R && s < nsamp) out[(int64_t)r * nsamp + s] = tile[tx][j];
    }
}

// ---------------------------------------------------------------------------------------------------
// Binary SK model GraphSK (src/graphs/SK.jl:28-165): couplings +-1/sqrt(N) bit-packed, integer cache
// lfields[i] = sqrt(N) * delta_energy(i).  Same workgroup layout as the Gaussian kernel, int32 fields.
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t TAG_SKBITS = 10;

struct SkbParams {
    const uint32_t* Jbits;  // [N][NW] rows of coupling bits (bit j of row i = J_ij in {0,1}; J~ = 2J - 1)
    int32_t* lf;            // [G][N][kSkRB]
    int32_t* lfl;           // [G][N][kSkRB]
    int32_t* move_last;     // [G][kSkRB]
    uint8_t* spins;         // [G][N]
    double* E_cur;          // [G * kSkRB]
    int64_t* acc_cur;
    double* Es;
    double beta, sN;
    uint64_t g0;
    int64_t iters, step, sample0;
    uint32_t k0, k1, replica0;
    int N, NW;
};

// (the structure of the step: streams from LDS, LDS-only barriers, what the first barrier waits for: as in sk_sweep_kernel)
template <int SPT, int NTH>
__global__ __launch_bounds__(NTH) void skb_sweep_kernel(SkbParams P)
{
    __shared__ int32_t sh_lfi[2][kSkRB];
    __shared__ uint32_t sh_si[2], sh_acc[2], sh_swap[2];
    const int tid = threadIdx.x, N = P.N;
    const int grp = blockIdx.x, Rp = gridDim.x * kSkRB;
    int32_t lf[SPT][kSkRB], lfl[SPT][kSkRB];
    uint32_t sb[SPT];
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        const int j = q * NTH + tid;
        sb[q] = j < N ? P.spins[(size_t)grp * N + j] : 0u;
#pragma unroll
        for (int r = 0; r < kSkRB; ++r) {
            lf[q][r] = j < N ? P.lf[((size_t)grp * N + j) * kSkRB + r] : 0;
            lfl[q][r] = j < N ? P.lfl[((size_t)grp * N + j) * kSkRB + r] : 0;
        }
    }
    double E_run = 0.0;
    int64_t A_run = 0;
    int32_t mlast = -1;
    if (tid < kSkRB) { E_run = P.E_cur[grp * kSkRB + tid]; A_run = P.acc_cur[grp * kSkRB + tid]; mlast = P.move_last[grp * kSkRB + tid]; }
    int64_t ns = P.sample0;

    __shared__ uint32_t sh_site[2][64];
    __shared__ double sh_u[2][64][kSkRB];
    auto prepare = [&](int64_t blk) {          // iterations blk * 64 + 1 .. blk * 64 + 64 -> buffer blk & 1
        const int buf = (int)(blk & 1);
        if (tid >= 64 && tid < 128)
            sh_site[buf][tid - 64] = site_of(P.k0, P.k1, P.g0 + (uint64_t)(blk * 64 + (tid - 64) + 1), (uint32_t)N);
        else if (tid >= 128 && tid < 256) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = (tid - 128) + 128 * k, l = idx >> 3, r = idx & 7;
                sh_u[buf][l][r] = rand53(P.k0, P.k1, P.g0 + (uint64_t)(blk * 64 + l + 1), P.replica0 + (uint32_t)(grp * kSkRB + r));
            }
        }
    };
    prepare(0);
    __syncthreads();
    uint32_t Jq[SPT], Jn[SPT];     // coupling bit of (site, j) for this thread's sites
    uint32_t site = sh_site[0][0];
#pragma unroll
    for (int q = 0; q < SPT; ++q) { const int j = q * NTH + tid; Jq[q] = j < N ? (P.Jbits[(size_t)site * P.NW + (j >> 5)] >> (j & 31)) & 1u : 0u; }

    long long next_sample = P.step;          // iterations k * step: a counter instead of a 64-bit modulo per iteration
    double expc[17];
    sk_exp_constants(expc);
    for (int64_t it = 1; it <= P.iters; ++it) {
        const int b = (int)(it & 1);
        const int li = (int)((it - 1) & 63), bi = (int)(((it - 1) >> 6) & 1);        // this iteration in its block
        const int qi = (int)(site / (uint32_t)NTH), owner = (int)(site % (uint32_t)NTH);
        if (tid == owner) {
#pragma unroll
            for (int q = 0; q < SPT; ++q)
                if (q == qi) {
#pragma unroll
                    for (int r = 0; r < kSkRB; ++r) sh_lfi[b][r] = lf[q][r];
                    sh_si[b] = sb[q];
                }
        }
        uint32_t site_n = 0u;
        auto stateless = [&]() {
            if (li == 0) prepare(((it - 1) >> 6) + 1);
            site_n = li == 63 ? sh_site[bi ^ 1][0] : sh_site[bi][li + 1];
#pragma unroll
            for (int q = 0; q < SPT; ++q) { const int j = q * NTH + tid; Jn[q] = j < N ? (P.Jbits[(size_t)site_n * P.NW + (j >> 5)] >> (j & 31)) & 1u : 0u; }
        };
        if (tid < 64) stateless();
        sk_lds_barrier();
        if (tid >= 64) stateless();
        if (tid < 64) {
            bool acc = false, swp = false;
            if (tid < kSkRB) {
                if (it == next_sample) { next_sample += P.step;
                    if (P.Es) P.Es[ns * Rp + grp * kSkRB + tid] = E_run;
                    ns += 1;
                }
                const double dE = (double)sh_lfi[b][tid] / P.sN;            // delta_energy = lfields / sqrt(N), SK.jl:137-140
                const double x = -P.beta * dE;
                acc = (x >= 0.0) || (sh_u[bi][li][tid] < det_exp_c(x, expc));
                swp = acc && (mlast == (int32_t)site);
                if (acc) { E_run += dE; A_run += 1; mlast = (int32_t)site; }
            }
            const unsigned long long ba = __ballot(acc), bs = __ballot(swp);
            if (tid == 0) { sh_acc[b] = (uint32_t)ba; sh_swap[b] = (uint32_t)bs; }
        }
        sk_lds_barrier();
        const uint32_t accm = sh_acc[b], swpm = sh_swap[b], si_old = sh_si[b];
        const uint32_t normal = accm & ~swpm;
        const uint32_t si_new = si_old ^ accm;

        if (swpm) {          // (selects, as in sk_sweep_kernel)
#pragma unroll
            for (int r = 0; r < kSkRB; ++r) {
                const bool sw = (swpm >> r) & 1u;
#pragma unroll
                for (int q = 0; q < SPT; ++q) { const int32_t a0 = lf[q][r], b0 = lfl[q][r]; lf[q][r] = sw ? b0 : a0; lfl[q][r] = sw ? a0 : b0; }
            }
        }
        if (normal) {
            // replica outermost: one scalar branch per replica instead of one per (site, replica) pair
            uint32_t x3[SPT];
#pragma unroll
            for (int q = 0; q < SPT; ++q) x3[q] = si_new ^ sb[q] ^ (Jq[q] ? 0xffu : 0u);   // bit r: s_i xor s_j xor J_ij
#pragma unroll
            for (int r = 0; r < kSkRB; ++r)
                if ((normal >> r) & 1u) {
#pragma unroll
                    for (int q = 0; q < SPT; ++q) {
                        const int32_t old = lf[q][r];
                        lfl[q][r] = old;
                        lf[q][r] = old + (((x3[q] >> r) & 1u) ? 4 : -4);       // lfj + 8*Jsij - 4, SK.jl:118-121
                    }
                }
            if (tid == owner) {
                int32_t lfm[kSkRB];
#pragma unroll
                for (int r = 0; r < kSkRB; ++r) lfm[r] = sh_lfi[b][r];
#pragma unroll
                for (int q = 0; q < SPT; ++q)
                    if (q == qi) {
#pragma unroll
                        for (int r = 0; r < kSkRB; ++r) {
                            const bool on = (normal >> r) & 1u;
                            lfl[q][r] = on ? lfm[r] : lfl[q][r];
                            lf[q][r] = on ? -lfm[r] : lf[q][r];
                        }
                    }
            }
        }
        if (tid == owner) {
#pragma unroll
            for (int q = 0; q < SPT; ++q)
                if (q == qi) sb[q] ^= accm;
        }
        site = site_n;
#pragma unroll
        for (int q = 0; q < SPT; ++q) Jq[q] = Jn[q];
    }
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        const int j = q * NTH + tid;
        if (j < N) {
            P.spins[(size_t)grp * N + j] = (uint8_t)sb[q];
#pragma unroll
            for (int r = 0; r < kSkRB; ++r) {
                P.lf[((size_t)grp * N + j) * kSkRB + r] = lf[q][r];
                P.lfl[((size_t)grp * N + j) * kSkRB + r] = lfl[q][r];
            }
        }
    }
    if (tid < kSkRB) { P.E_cur[grp * kSkRB + tid] = E_run; P.acc_cur[grp * kSkRB + tid] = A_run; P.move_last[grp * kSkRB + tid] = mlast; }
}

// energy(X::GraphSK, C) SK.jl:62-96: lfields[i] = 2(-lf + 2 s_i), lf = -(2 s_i - 1)(N - 1 - 2 sc), sc = popcount(J_i xor s)
__global__ __launch_bounds__(256) void skb_fields_kernel(const uint32_t* __restrict__ Jbits, const uint8_t* __restrict__ spins,
                                                         int32_t* __restrict__ lf, int32_t* __restrict__ lfl,
                                                         int32_t* __restrict__ move_last, int N, int NW)
{
    const int grp = blockIdx.y;
    const int i = blockIdx.x * 32 + (threadIdx.x >> 3), r = threadIdx.x & 7;
    if (i >= N) return;
    const uint8_t* sp = spins + (size_t)grp * N;
    int sc = 0;
    for (int j = 0; j < N; ++j) sc += (int)(((Jbits[(size_t)i * NW + (j >> 5)] >> (j & 31)) ^ (sp[j] >> r)) & 1u);
    const int si = (sp[i] >> r) & 1;
    const int l = -(2 * si - 1) * (N - 1 - 2 * sc);
    lf[((size_t)grp * N + i) * kSkRB + r] = 2 * (-l + 2 * si);
    lfl[((size_t)grp * N + i) * kSkRB + r] = 0;
    if (i == 0) move_last[grp * kSkRB + r] = -1;
}

// n = -2 sum(s) + sum_i lf_i, lf_i = 2 s_i - lfields_i / 2;  E = (n / 2) / sqrt(N)
__global__ __launch_bounds__(64) void skb_energy_kernel(const int32_t* __restrict__ lf, const uint8_t* __restrict__ spins,
                                                        double* __restrict__ E_out, int N, int Rp, double sN)
{
    const int rep = blockIdx.x * 64 + threadIdx.x;
    if (rep >= Rp) return;
    const int grp = rep / kSkRB, r = rep % kSkRB;
    long long n = 0;
    for (int i = 0; i < N; ++i) {
        const int si = (spins[(size_t)grp * N + i] >> r) & 1;
        n += -2 * si + (2 * si - lf[((size_t)grp * N + i) * kSkRB + r] / 2);
    }
    E_out[rep] = (double)(n / 2) / sN;
}

}  // namespace rrrmc
