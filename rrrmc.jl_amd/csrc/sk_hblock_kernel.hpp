// gfx950 kernel for standardMC (src/RRRMC.jl:81-127) on the dense SK models (src/graphs/SK.jl), blocked form, round 4: the bulk update
// without control flow.
//
// sk_block_kernel (round 3, sk_block_kernel.hpp) takes 64 attempts at a time: gather the attempted sites' fields into a window, let one
// wavefront per replica decide the block on the window, then apply the accepted moves to the register-resident fields.  Its apply
// phase tested every (attempt, replica) with a scalar compare and a branch and was bound by exactly that: as many scalar as vector
// instructions, a taken branch per replica that did not accept (profiles/r03/sk_block_summary.txt).  This kernel keeps gather and decide
// and replaces the bulk representation:
//
//   * the registers hold H_j = sigma_j lfields[j] — the field WITHOUT the site's own sign (sigma_j = 2 s_j - 1).  update_cache!
//     (SK.jl:239-276) adds 4 sigma_i' sigma_j J_ij to lfields[j] (sigma_i' = the moved spin after its flip), i.e. sigma_i' 4 J_ij to H_j: the
//     sign is the same for every site, and H_i itself does not change (J_ii = 0; lfields[i] = -lfm is the flip of sigma_i).  IEEE addition
//     is symmetric under negation of both operands, so sigma_j H_j is lfields[j] bit for bit at all times.
//   * per (attempt k, replica r) the deciding wavefront leaves a multiplier m in {+1.0, -1.0, 0.0} (0 = not accepted) in LDS; the bulk
//     update is H[q][r] = fma(row_k[q], m, H[q][r]) — one instruction per site word, no test, no branch, no lane masks, EXEC untouched.  The
//     product with +-1 is exact, so the result is the reference's add; with 0 the field is unchanged (a field that is exactly -0.0 may
//     become +0.0: no comparison or sum can tell).
//   * lfields_last (SK.jl:255-262 copies it on every accepted move) is only ever read by the undo of the NEXT accepted move (the array swap
//     of SK.jl:247-250), by the next block's window and by the store at the end of the launch: the copy is made before the block's last
//     accepted move and before a move that the following accepted move undoes (flagged by the deciding wavefront), nowhere else.
//     It is kept as Hl_j = (H_j before the move); lfields_last[j] = sigma_j Hl_j, with the opposite sign at j = move_last, whose spin has
//     flipped since.  Hl lives in memory (P.hl, [replica][site]: coalesced), not in registers: a copy is a store, so no register changes
//     under a branch (a conditional register copy made the compiler shuttle every field through temporaries on the hot path), and the
//     fields take half the registers.
//   * the spins live in LDS (one byte per site, bit = replica); the deciding wavefronts flip them once per block.
// Everything a caller can see — energies, samples, accepted counts, configurations, both field arrays, move_last — is what sk_block_kernel and
// the oracle produce (tests/emulate_sk_block.py models this bulk phase on the CPU; tests/test_gpu_sk_parity.py, test_gpu_sk_seams.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "sk_block_kernel.hpp"

namespace rrrmc {

#ifdef RRRMC_SKH_ABL_HOTROWS
#define SKH_ROW(s) ((s) & 7u)          /* timing experiment (tools/ubench/skh_bench.hip): always-hot rows, wrong couplings */
#else
#define SKH_ROW(s) (s)
#endif

template <int I, int E, typename F>
__device__ __forceinline__ void sk_static_for(F&& f)
{
    if constexpr (I < E) { f(std::integral_constant<int, I>{}); sk_static_for<I + 1, E>(f); }
}

#ifndef SKH_PF
#define SKH_PF(spt) ((spt) <= 2 ? 4 : (spt) <= 4 ? 2 : 1)
#endif

template <int SPT, int NTH, int RB = kSkRB, bool BIN = false>
__global__ __launch_bounds__(NTH, RB == 4 ? 2 : 1) void sk_hblock_kernel(SkBlockParams P)
{
    static_assert(RB == 8 || RB == 4, "8 or 4 replicas per workgroup");
    static_assert(NTH == 256 || NTH == 512, "256 or 512 threads");
    constexpr int NH = kSkRB / RB;                         // workgroups per group of 8 replicas
    constexpr int NWV = NTH / 64;                          // wavefronts
    constexpr int RPW = NWV >= RB ? 1 : RB / NWV;          // replicas decided per wavefront
    constexpr int PF = SKH_PF(SPT);                       // attempts per group of row registers (two groups)
    constexpr int NJW = kSkW * kSkW / NTH;                 // doubles of the sub-matrix per thread
    constexpr int NIT = kSkW / (2 * PF);                   // iterations of the apply loop
    constexpr int JPI = NJW >= NIT ? NJW / NIT : 1;        // doubles staged per iteration, in the first NJW / JPI iterations
    static_assert(JPI * NIT == NJW || (JPI == 1 && NJW < NIT), "sub-matrix slices");
    // Wave priorities (whole-group builds: two wavefronts per SIMD, w and w + 4).  A SIMD's arbiter prefers its OLDER wavefront, which then finishes
    // a phase a quarter earlier and waits at the barrier for the younger one (profiles/r05/c3_block_budget.md: decide 9.8 K against 12.3 K cycles,
    // apply 15.2 K against 19.7 K).  Decide: the younger one goes first.  Apply: the two take turns, sixteen attempts at a time (up to four sites
    // per thread: -7 % of the launch at N = 1024 and 2048); with eight sites per thread the younger one first throughout (-14 % at N = 4096); with six
    // neither helps (measured: tools/ubench/run_skh.sh, profiles/r05/sk_priorities.txt).
    constexpr int PRIO_APPLY = RB != 8 ? 0 : SPT <= 4 ? 16 : SPT == 8 ? 1 : 0;
    constexpr bool PRIO_DECIDE = RB == 8 && SPT != 6;
    __shared__ double sh_Jw[2][kSkW * kSkW];               // the block's 64 x 64 coupling sub-matrix, double-buffered: the next block's is staged during apply
    __shared__ double sh_wf[kSkW][RB], sh_wfl[kSkW][RB];
    // acceptance uniforms and their logarithms.  Whole-group build: double-buffered, the next block's are drawn by every wavefront behind its
    // own decisions, while it would wait for the slowest one; the split build (two workgroups per compute unit: 160 KiB of LDS for both)
    // keeps one buffer and draws them at the top of the bulk phase
    constexpr int NUB = RB == 8 ? 2 : 1;
    __shared__ double sh_u[NUB][kSkW][RB], sh_L[NUB][kSkW][RB];
    __shared__ __attribute__((aligned(4))) uint8_t sh_need[RB];                              // the next block attempts replica r's move_last (its window needs lfields_last)
    __shared__ __attribute__((aligned(16))) double sh_mult[kSkW][RB];     // per (attempt, replica): the multiplier +1.0 / -1.0 / 0.0
    __shared__ uint32_t sh_acc[2][kSkW];                   // per attempt: bit 8 + r = undo swap, bit 24 + r = copy lfields_last first
    __shared__ uint8_t sh_cslot[kSkW];
    constexpr int NMAX = NTH * SPT >= kSkThreads * kSkMaxSPT ? NTH * SPT : kSkThreads * kSkMaxSPT;      // sites this build covers (2048 at least)
    __shared__ uint8_t sh_canon[NMAX];
    __shared__ uint32_t sh_spinw[NMAX / 4];                       // the spins: byte j = site j, bit = replica of the group
    __shared__ int32_t sh_mlast[RB];
    uint8_t* const sh_spin = reinterpret_cast<uint8_t*>(sh_spinw);
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), N = P.N;
    const int grp = blockIdx.x / NH, r8 = (blockIdx.x % NH) * RB, Rp = (gridDim.x / NH) * kSkRB;

    // The thread's sites: pairs of neighbours (one 16-byte load), the pairs 2 NTH apart — a wavefront's load of a row of J, or of lfields_last,
    // then covers 1024 contiguous bytes.  (SPT contiguous sites per thread put every lane of a 16-byte load into a different 64-byte segment
    // at SPT = 8: the compute unit's L1 then moved four times the bytes the registers received, and the bulk phase at N = 4096 ran at a
    // quarter of its Float64 rate.)
    constexpr int SC = SPT % 2 == 0 ? 2 : 1;
    auto site_of = [&](int q) -> int { return (q / SC) * (SC * NTH) + SC * tid + (q % SC); };
    double H[SPT][RB];
    double* const hl = P.hl + (size_t)(grp * kSkRB + r8) * (size_t)N;      // this workgroup's replicas: hl[r * N + j]
    {
        int32_t ml[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) ml[r] = P.move_last[grp * kSkRB + r8 + r];
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const int j = site_of(q);
            const uint32_t sb = j < N ? P.spins[(size_t)grp * N + j] : 0u;
            if (j < N) sh_spin[j] = (uint8_t)sb;
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const double a = j < N ? P.lf[((size_t)grp * N + j) * kSkRB + r8 + r] : 0.0;
                const double b = j < N ? P.lfl[((size_t)grp * N + j) * kSkRB + r8 + r] : 0.0;
                const bool up = (sb >> (r8 + r)) & 1u;
                H[q][r] = up ? a : -a;
                if (j < N) hl[(size_t)r * N + j] = (up != (j == ml[r])) ? b : -b;
            }
        }
    }
    for (int j = tid; j < NMAX; j += NTH) sh_canon[j] = 0xffu;
    if (tid < RB) sh_mlast[tid] = P.move_last[grp * kSkRB + r8 + tid];
    __syncthreads();
    double E_run[RPW];
    int64_t A_run[RPW];
    int32_t mlast[RPW];
#pragma unroll
    for (int a = 0; a < RPW; ++a) {
        const int r = wv + a * NWV;
        const bool on = r < RB;
        E_run[a] = on ? P.E_cur[grp * kSkRB + r8 + r] : 0.0;
        A_run[a] = on ? P.acc_cur[grp * kSkRB + r8 + r] : 0;
        mlast[a] = on ? P.move_last[grp * kSkRB + r8 + r] : -1;
    }
    int64_t ns[RPW], next_sample[RPW];
#pragma unroll
    for (int a = 0; a < RPW; ++a) { ns[a] = P.it_base / P.step; next_sample[a] = (P.it_base / P.step + 1) * P.step; }

    const int64_t nblk = (P.iters + kSkW - 1) / kSkW;
#ifdef RRRMC_SKB_STAMPS
    uint64_t st[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
    auto draw_uniforms = [&](int64_t b) {
        for (int idx = tid; idx < kSkW * RB; idx += NTH) {
            const int l = idx / RB, r = idx % RB;
            const double u = rand53(P.k0, P.k1, P.g0 + (uint64_t)(b * kSkW + l + 1), P.replica0 + (uint32_t)(grp * kSkRB + r8 + r));
            sh_u[b & (NUB - 1)][l][r] = u;
            // ln u for the verdict's filter only (the decision itself is det_exp's): the hardware's log2 on the Float32 image of u, good to
            // 1.2e-7 |ln u| + 1e-7 — the filter's margin below is sixteen times that.  (The library's log keeps a dozen Float64 constants in
            // vector registers across the whole block loop; u == 0 or a Float32 denormal gives -inf, i.e. a NaN band: det_exp decides.)
            sh_L[b & (NUB - 1)][l][r] = (double)__builtin_amdgcn_logf((float)u) * 0.69314718055994530942;
        }
    };
    uint32_t sv = P.blkSites[lane];                        // sites of the current block, lane = attempt
    if (nblk > 0) {
        for (int e = tid; e < kSkW * kSkW; e += NTH) sh_Jw[0][e] = P.blkJw[e];
        draw_uniforms(0);
        if (wv == 0) {
            sh_canon[sv] = (uint8_t)lane;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            sh_cslot[lane] = sh_canon[sv];
            sh_acc[0][lane] = 0u;
            if (lane < RB) { const int32_t m = sh_mlast[lane]; sh_need[lane] = (m >= 0 && sh_canon[m >= 0 ? m : 0] != 0xffu) ? 1 : 0; }
        }
    }
    double JA[PF][SPT], JB[PF][SPT];
#pragma unroll
    for (int kk = 0; kk < PF; ++kk) {
        const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)sv, kk);
#pragma unroll
        for (int q = 0; q < SPT; ++q) JA[kk][q] = P.J4[(size_t)s0 * P.ldJ + (site_of(q))];
    }
    __syncthreads();

    for (int64_t b = 0; b < nblk; ++b) {
#ifdef RRRMC_SKB_STAMPS
        const uint64_t tA = __builtin_amdgcn_s_memtime();
#endif
        const int pb = (int)(b & 1);
        const int nv = (int)(P.iters - b * kSkW < kSkW ? P.iters - b * kSkW : kSkW);
        const uint32_t sv_next = P.blkSites[(b + 1) * kSkW + lane];
        // ---- gather: the owners publish lfields of the attempted sites (sigma_j H_j).  lfields_last is only read if a replica's FIRST accepted
        //      move of the block undoes its last one (later undos read what the block itself tracked), which takes move_last among the
        //      block's sites: only then is it fetched from memory (the deciding wavefront puts the sign of lfields_last[move_last] right)
        bool need_fl[RB];
        {
            uint32_t nw[RB / 4];
#pragma unroll
            for (int w = 0; w < RB / 4; ++w) nw[w] = (uint32_t)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<const uint32_t*>(sh_need)[w]);
#pragma unroll
            for (int r = 0; r < RB; ++r) need_fl[r] = (nw[r / 4] >> (8 * (r & 3))) & 0xffu;
        }
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const int j = site_of(q);
            const uint32_t c = sh_canon[j];
            if (c != 0xffu) {
                const uint32_t sb = sh_spin[j];
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    const bool up = (sb >> (r8 + r)) & 1u;
                    sh_wf[c][r] = up ? H[q][r] : -H[q][r];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            if (need_fl[r]) {
#pragma unroll
                for (int q = 0; q < SPT; ++q) {
                    const int j = site_of(q);
                    const uint32_t c = sh_canon[j];
                    if (c != 0xffu) {
                        const double hlv = hl[(size_t)r * N + j];
                        sh_wfl[c][r] = ((sh_spin[j] >> (r8 + r)) & 1u) ? hlv : -hlv;
                    }
                }
            }
        }
#ifdef RRRMC_SKB_STAMPS
        const uint64_t tB = __builtin_amdgcn_s_memtime();
#endif
        sk_lds_barrier();
#ifdef RRRMC_SKB_STAMPS
        const uint64_t tC = __builtin_amdgcn_s_memtime();
#endif
        // ---- decide: one wavefront per replica, lane = attempt (sk_block_kernel's loop)
#pragma unroll
        for (int a = 0; a < RPW; ++a) {
            const int r = wv + a * NWV;
            if (r < RB) {
                const uint32_t cs = sh_cslot[lane];
                double f = sh_wf[cs][r], fl = sh_wfl[cs][r];
                if ((int32_t)sv == mlast[a]) fl = -fl;
                uint32_t sp = ((uint32_t)sh_spin[sv] >> (r8 + r)) & 1u;
                const double u = sh_u[pb & (NUB - 1)][lane][r];
                const int64_t it0 = P.it_base + b * kSkW;
                uint32_t accw = 0u;
                const double Lu = sh_L[pb & (NUB - 1)][lane][r];
                const double Lm = 2e-6 - 2e-6 * Lu, Lhi = Lu + Lm, Llo = Lu - Lm;           // (the band: one lane-evaluation in 4e5 falls into it)
                const unsigned long long vm = nv >= kSkW ? ~0ull : (1ull << nv) - 1ull;
                auto verdict = [&](const double x) -> unsigned long long {
                    const unsigned long long ge0 = __builtin_amdgcn_ballot_w64(x >= 0.0), hi = __builtin_amdgcn_ballot_w64(x > Lhi);
                    const unsigned long long lo = __builtin_amdgcn_ballot_w64(x < Llo), fin = __builtin_amdgcn_ballot_w64(x > -700.0);
                    const unsigned long long sure_acc = ge0 | hi, sure_rej = lo & fin;
                    unsigned long long m = vm & sure_acc;
                    unsigned long long need = vm & ~(sure_acc | sure_rej);
                    asm volatile("" : "+s"(need));
                    if (__builtin_expect(need != 0ull, 0)) {              // once in ~10^4 evaluations of a wavefront: the constants are built here, not kept in registers
                        double expc[17];
                        sk_exp_constants_here(expc);
                        m = vm & (ge0 | __builtin_amdgcn_ballot_w64(u < det_exp_v(x, expc)));
                    }
                    return m;
                };
                auto sample_rel = [&]() -> int { const int64_t d = next_sample[a] - it0 - 1; return d < (int64_t)kSkW ? (int)d : kSkW; };
                int ks = sample_rel();
                auto xof = [&](double fv) -> double { if constexpr (BIN) return -P.beta * (fv / P.sN); else return -P.beta * fv; };
                unsigned long long B = verdict(xof(f));
                if constexpr (PRIO_DECIDE) { if (wv >= NWV / 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
                while (B) {
                    const int k = __builtin_ctzll(B);
                    while (__builtin_expect(k >= ks, 0)) {              // sample BEFORE the move (RRRMC.jl:104-108)
                        if (P.Es && lane == 0) P.Es[ns[a] * Rp + grp * kSkRB + r8 + r] = E_run[a];
                        ns[a] += 1; next_sample[a] += P.step;
                        ks = sample_rel();
                    }
                    const double fk = sk_readlane_f64(f, k);
                    const double dE = BIN ? fk / P.sN : fk;             // delta_energy, SK.jl:278-284 (BIN: SK.jl:137-140)
                    const int32_t site_k = __builtin_amdgcn_readlane((int)sv, k);
                    const uint32_t spk = (uint32_t)__builtin_amdgcn_readlane((int)sp, k);
                    const bool swapped = mlast[a] == site_k;            // undo path of update_cache!, SK.jl:247-250
                    E_run[a] += dE; A_run[a] += 1;
                    const bool dup = (int32_t)sv == site_k;
                    const uint32_t aw = 1u | (swapped ? 0x100u : 0u) | (spk << 16);
                    uint32_t m0_keep;
                    asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1" : "+v"(accw), "=&s"(m0_keep) : "s"(aw), "s"(k));
                    if (__builtin_expect(swapped, 0)) {
                        const double t = f; f = fl; fl = t;
                    } else {
                        const double d = sh_Jw[pb][k * kSkW + lane];
                        const uint32_t neg = sp ^ spk ^ 1u;
                        const double dl = __longlong_as_double(__double_as_longlong(d) ^ ((long long)neg << 63));
                        fl = f;
                        const double fn = f + dl;                       // lfields[j] = lfj + 4 sigma J, SK.jl:256-262
                        f = dup ? -fl : fn;                             // lfields[move] = -lfm, SK.jl:263-264
                        mlast[a] = site_k;
                    }
                    sp ^= dup ? 1u : 0u;
                    B = verdict(xof(f)) & ((~0ull << k) << 1);
                }
                while (next_sample[a] <= it0 + nv) {
                    if (P.Es && lane == 0) P.Es[ns[a] * Rp + grp * kSkRB + r8 + r] = E_run[a];
                    ns[a] += 1; next_sample[a] += P.step;
                }
                // what the bulk phase needs, for all 64 attempts at once (lane = attempt): the multiplier, the flags, the spin flips
                const bool acc = accw & 1u, swp = (accw >> 8) & 1u, spk1 = (accw >> 16) & 1u;
                const unsigned long long Am = __builtin_amdgcn_ballot_w64(acc), Sm = __builtin_amdgcn_ballot_w64(swp);
                const unsigned long long later = Am & ((~0ull << lane) << 1);          // the accepted attempts behind this lane's
                const bool next_undoes = later != 0ull && ((Sm >> __builtin_ctzll(later | (1ull << 63))) & 1ull);
                const bool cpy = acc && !swp && (later == 0ull || next_undoes);
                sh_mult[lane][r] = (acc && !swp) ? (spk1 ? -1.0 : 1.0) : 0.0;           // sigma_i' = the moved spin AFTER its flip
                const uint32_t fw = (swp ? (0x100u << r) : 0u) | (cpy ? (0x1000000u << r) : 0u);
                if (fw) atomicOr(&sh_acc[pb][lane], fw);
                if (acc) atomicXor(&sh_spinw[sv >> 2], 1u << (8u * (sv & 3u) + (uint32_t)(r8 + r)));     // spinflip!, Interface.jl:89-92 (an undo flips it back)
                if (lane == 0) sh_mlast[r] = mlast[a];
            }
        }
        if constexpr (NUB == 2) { if (b + 1 < nblk) draw_uniforms(b + 1); }
#ifdef RRRMC_SKB_STAMPS
        const uint64_t tD = __builtin_amdgcn_s_memtime();
#endif
        sk_lds_barrier();
#ifdef RRRMC_SKB_STAMPS
        const uint64_t tE = __builtin_amdgcn_s_memtime();
#endif
        // ---- apply: every attempt of the block on the registers; stage the next block meanwhile
        const uint32_t accv = sh_acc[pb][lane];
        const unsigned long long Fm = __builtin_amdgcn_ballot_w64(accv != 0u);          // attempts with a copy or an undo
        const bool more = b + 1 < nblk;
        if (more) {
            if (wv == 0) {
                sh_canon[sv] = 0xffu;
                asm volatile("" ::: "memory");
                sh_canon[sv_next] = (uint8_t)lane;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                sh_cslot[lane] = sh_canon[sv_next];
                sh_acc[pb ^ 1][lane] = 0u;
                // does the next block attempt a replica's move_last?  (sh_mlast is this block's: written before the barrier above)
                if (lane < RB) { const int32_t m = sh_mlast[lane]; sh_need[lane] = (m >= 0 && sh_canon[m >= 0 ? m : 0] != 0xffu) ? 1 : 0; }
            }
            if constexpr (NUB == 1) draw_uniforms(b + 1);
        }
#ifdef RRRMC_SKB_STAMPS
        const uint64_t tF = __builtin_amdgcn_s_memtime();
#endif
        // The bulk update proper, 16 attempts per iteration.  Lane l holds the multipliers of attempt 16 it + (l & 15) (one register pair per
        // replica, read once per iteration); attempt t of the iteration broadcasts its own to all lanes INSIDE the fused multiply-add
        // (DPP row_newbcast:t, which the Float64 pipe supports at full rate): no instruction is spent on the multiplier.  t must be an
        // immediate, hence the static unrolling.  (Per-step LDS reads of a wave-uniform address move 64 lanes' worth of data each and
        // bound the phase at the LDS port; v_readlane costs what a multiply-add costs — both measured, tools/ubench/f64_rates.hip.)
        auto apply_step = [&](auto T_, int k, const double (&d)[SPT], const double (&mv)[RB]) {
            constexpr int t = decltype(T_)::value;
            if (__builtin_expect((Fm >> k) & 1ull, 0)) {
                const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)accv, k);
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    // (a wave-uniform base and a 32-bit lane offset: no per-lane 64-bit address is kept across the loop)
                    double* const hr = hl + (size_t)__builtin_amdgcn_readfirstlane(r * N);
                    if ((w >> (24 + r)) & 1u) {                          // lfields_last = lfields before this move (SK.jl:255-262)
#pragma unroll
                        for (int q = 0; q < SPT; ++q)
                            if (site_of(q) < N) hr[(uint32_t)(site_of(q))] = H[q][r];
                    }
                    if ((w >> (8 + r)) & 1u) {                           // lfields <-> lfields_last (SK.jl:247-250)
#pragma unroll
                        for (int q = 0; q < SPT; ++q) {
                            if (site_of(q) < N) {
                                double* const px = hr + (uint32_t)(site_of(q));
                                const double tmp = *px; *px = H[q][r]; H[q][r] = tmp;
                            }
                        }
                    }
                }
                asm volatile("s_nop 4");                                 // (EXEC was written under the bounds tests: five wait states before a DPP operation)
            }
#pragma unroll
            for (int q = 0; q < SPT; ++q)
#pragma unroll
                for (int r = 0; r < RB; ++r)
                    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(H[q][r]) : "v"(mv[r]), "v"(d[q]), "n"(t));
        };
        // the next block's sub-matrix: a slice per pair of row groups, global -> registers in one step, -> LDS in the next (one or two doubles
        // in flight instead of the whole thread's share held across the loop)
        const double* const jw_src = P.blkJw + (more ? b + 1 : b) * (kSkW * kSkW) + tid;
        double jw_reg[JPI];
#pragma unroll
        for (int e = 0; e < JPI; ++e) jw_reg[e] = jw_src[e * NTH];
        for (int kb = 0; kb < kSkW; kb += 16) {
            if constexpr (PRIO_APPLY == 1) {
                if (wv >= NWV / 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
            } else if constexpr (PRIO_APPLY > 1) {
                if ((((uint32_t)kb / PRIO_APPLY) ^ (uint32_t)(wv >= NWV / 2)) & 1u) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
            }
            double mv[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) mv[r] = sh_mult[kb + (lane & 15)][r];
            sk_static_for<0, 16 / (2 * PF)>([&](auto G_) {
                constexpr int g = decltype(G_)::value;
                const int k0 = kb + g * 2 * PF;
                if (const int i0 = (k0 / (2 * PF)) * JPI; i0 < NJW) {
#pragma unroll
                    for (int e = 0; e < JPI; ++e) sh_Jw[pb ^ 1][(i0 + e) * NTH + tid] = jw_reg[e];
                    const int i1 = i0 + JPI < NJW ? i0 + JPI : i0;      // (the last slice is re-read once more: no load past the table)
#pragma unroll
                    for (int e = 0; e < JPI; ++e) jw_reg[e] = jw_src[(i1 + e) * NTH];
                }
#pragma unroll
                for (int kk = 0; kk < PF; ++kk) {           // rows of attempts k0 + PF .. k0 + 2 PF - 1
                    const uint32_t stn = SKH_ROW((uint32_t)__builtin_amdgcn_readlane((int)sv, k0 + PF + kk));
#pragma unroll
                    for (int q = 0; q < SPT; ++q) JB[kk][q] = P.J4[(size_t)stn * P.ldJ + (site_of(q))];
                }
                sk_static_for<0, PF>([&](auto K_) { constexpr int kk = decltype(K_)::value; apply_step(std::integral_constant<int, g * 2 * PF + kk>{}, k0 + kk, JA[kk], mv); });
                const uint32_t svn = k0 + 2 * PF < kSkW ? sv : sv_next;
#pragma unroll
                for (int kk = 0; kk < PF; ++kk) {           // rows of attempts k0 + 2 PF .. k0 + 3 PF - 1
                    const uint32_t stn = SKH_ROW((uint32_t)__builtin_amdgcn_readlane((int)svn, (k0 + 2 * PF + kk) & 63));
#pragma unroll
                    for (int q = 0; q < SPT; ++q) JA[kk][q] = P.J4[(size_t)stn * P.ldJ + (site_of(q))];
                }
                sk_static_for<0, PF>([&](auto K_) { constexpr int kk = decltype(K_)::value; apply_step(std::integral_constant<int, g * 2 * PF + PF + kk>{}, k0 + PF + kk, JB[kk], mv); });
            });
        }
#ifdef RRRMC_SKB_STAMPS
        const uint64_t tG = __builtin_amdgcn_s_memtime();
#endif
        sv = sv_next;
        sk_lds_barrier();
#ifdef RRRMC_SKB_STAMPS
        { const uint64_t tH = __builtin_amdgcn_s_memtime();
          st[0] += tB - tA; st[1] += tC - tB; st[2] += tD - tC; st[3] += tE - tD; st[4] += tF - tE; st[5] += tG - tF; st[6] += tH - tG; }
#endif
    }
#ifdef RRRMC_SKB_STAMPS
    if (grp == 1 && lane == 0 && (wv == 0 || wv == NWV - 1))
        printf("skh stamps wave %d blocks %lld: gather %llu bar1 %llu decide %llu bar2 %llu stage %llu apply %llu tail+bar3 %llu (memtime ticks)\n", wv, (long long)nblk,
               (unsigned long long)st[0], (unsigned long long)st[1], (unsigned long long)st[2], (unsigned long long)st[3], (unsigned long long)st[4],
               (unsigned long long)st[5], (unsigned long long)st[6]);
#endif

#pragma unroll
    for (int a = 0; a < RPW; ++a) {
        const int r = wv + a * NWV;
        if (r < RB && lane == 0) {
            P.E_cur[grp * kSkRB + r8 + r] = E_run[a]; P.acc_cur[grp * kSkRB + r8 + r] = A_run[a]; P.move_last[grp * kSkRB + r8 + r] = mlast[a];
            sh_mlast[r] = mlast[a];      // (already there unless the launch had no block)
        }
    }
    __syncthreads();
    {
        int32_t ml[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) ml[r] = sh_mlast[r];
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const int j = site_of(q);
            if (j < N) {
                const uint32_t sb = sh_spin[j];
                const size_t off = (size_t)grp * N + j;
                if constexpr (RB == 8) {
                    P.spins[off] = (uint8_t)sb;
                } else {
                    // the byte is shared with the group's other workgroup: change this one's four bits only, atomically on the enclosing word
                    const uint32_t was = (P.spins[off] >> r8) & 0xfu, now = (sb >> r8) & 0xfu, sh = 8u * (uint32_t)(off & 3u) + (uint32_t)r8;
                    atomicXor(reinterpret_cast<uint32_t*>(P.spins + (off & ~(size_t)3)), (was ^ now) << sh);
                }
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    const bool up = (sb >> (r8 + r)) & 1u;
                    const double hlv = hl[(size_t)r * N + j];
                    P.lf[off * kSkRB + r8 + r] = up ? H[q][r] : -H[q][r];
                    P.lfl[off * kSkRB + r8 + r] = (up != (j == ml[r])) ? hlv : -hlv;
                }
            }
        }
    }
}

}  // namespace rrrmc
