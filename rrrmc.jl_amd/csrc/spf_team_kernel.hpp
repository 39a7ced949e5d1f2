// gfx950 kernel for standardMC (src/RRRMC.jl:81-127) on the Float64-coupling sparse models (GraphRRGNormal / GraphEANormal:
// src/graphs/RRG.jl:503-627, src/graphs/EA.jl:534-680): a TEAM of wavefronts per group of replicas (round 4), re-cut in round 5.
//
// spf_sweep_kernel (spf_kernels.hpp) walks the chain with ONE wavefront per group of 64 replicas: one attempt after the other, ~ 1 us each,
// whatever the machine has idle (8192 replicas = 128 wavefronts on 1024 SIMDs).  The attempts of a chain are not all dependent on each other:
// the SITE stream is known in advance, an attempt at site i reads lfields[i] and the spins of i and its neighbours and writes lfields of i and
// its neighbours (update_cache!, RRG.jl:576-617) — two attempts whose closed neighbourhoods do not meet commute exactly (no field receives
// an addition from both, so no rounding order changes).  This kernel runs such attempts side by side:
//
//   * spf_team_plan_kernel (state-independent, once per launch for all teams) writes one record per attempt: site, neighbours, couplings, the
//     64-bit set of the attempts among the previous 64 (the window) that do NOT commute with it, and the latest earlier attempt at the same site.
//   * NW - 1 executing wavefronts per team take the attempts of the chain in pairs (one Philox block serves iterations 2h, 2h + 1), pair h on
//     wavefront h mod (NW - 1).  An attempt's loads go out once every attempt of its conflict set has REPORTED (stores performed; one LDS read
//     of the slots' flags tells) and everything older than the window has retired; unless the pair's second attempt conflicts with the first,
//     the loads of both go out together and the two are worked off side by side: accept (RRRMC.jl:39 — decided by the hardware's 2^x wherever
//     that is safe, by the oracle's det_exp in a narrow band around it: identical decisions), the K + 1 field updates, the spin word — on the
//     group's [N][64] arrays in HBM / L2, exactly as spf_sweep_kernel does.  A wavefront counts its own accepted moves (a count does not care
//     about order).  Results go into the attempt's SLOT in LDS (slot = iteration mod M), free once attempt it - M has retired.
//   * FUSED PAIRS (teams of 32 or 16 replicas; K <= 6 in sixteen-wavefront teams, K = 7, 8 in eight-wavefront ones): a wavefront has twice the team's lanes and an instruction costs the same with half
//     of them idle, so when the pair's attempts commute, lanes 0..31 work off the first and lanes 32..63 the second FOR THE SAME REPLICAS with
//     one instruction stream — eight loads, one decision, one update, one set of records, one report for the two.  What differs per attempt
//     (site, neighbours, couplings, slot, random number) is per-lane data: the record fields come from the fetched pair by a crossbar read.
//     A pair whose undo path may be due (one in a thousand) has written nothing yet and starts again as two attempts one after the other,
//     like the pairs whose second attempt depends on the first.
//   * one RETIRING wavefront keeps what the reference's loop keeps in chain order: the tracked energy (E += dE is a Float64 running sum:
//     its order is the chain's), the samples (RRRMC.jl:104-108) and, per replica, WHICH attempt its last accepted move was (tl).  It consumes
//     the attempts strictly in order from their slots: {the attempted site's own field, or +0.0 for a replica that did not accept (E - 0.0 == E
//     for every E); the attempt's index, or 0} — per attempt two reads, one Float64 add, one integer max — and publishes the retired prefix.
//   * the undo path (RRG.jl:583-593: a move of the spin that was also the replica's last accepted move swaps lfields <-> lfields_last) needs
//     move_last as of t - 1.  move_last = site(tl): the executing wavefront reads the prefix and tl as retired so far and gathers the sites; if
//     an earlier attempt at the same site has not retired yet, or an accepting replica has site(tl) == site(t), it waits until t - 1 has
//     retired — the retiring wavefront then rests until t itself reports — and reads again; otherwise no attempt in flight can make
//     move_last equal to this site.
//   * the undo record of an accepted attempt a (the K neighbour fields before the move and the own field) stays in a's slot until the slot's
//     next user, attempt a + M, EVACUATES it into its wavefront's keep area — only for the replicas whose last accepted move still is a —
//     then marks the slot (ev[slot] = a + M) and only then overwrites it.  So the record of tl = a is in slot(a) iff ev[slot(a)] == a, else in
//     the keep area of the wavefront that runs attempt a + M: a reader takes the slot's data first and the mark second.  A keep entry is
//     written for a replica only while its last accepted move is the evacuated one, i.e. while its previous keep entry is dead.
// No wavefront ever waits for a later attempt and no wavefront waits while it holds an unreported attempt that the awaited one could need
// (the first attempt of a pair is reported before the second waits for the prefix), so the scheme cannot deadlock; every wait is bounded
// all the same and raises a status word (SpfTeamParams::status).  Everything a caller can see (fields, spins, undo records, move_last,
// energies, samples, accepted counts) is bit-identical to spf_sweep_kernel and the oracle (tests/test_gpu_spf_parity.py runs every case — two
// bonds to the same neighbour included — through the builds of both kernels; tools/ubench/spf_team_bench.hip compares every word).
//
// What bounds it (profiles/r05/spf_team_*): memory.  51 bytes per attempt cross the L2s (the lines of the K + 1 fields are read whole, 34 bytes;
// only the accepting lanes store, 18 bytes); at 8192 replicas (256 teams of 32, fused pairs: about 240 instructions per attempt of a team
// over sixteen wavefronts) the kernel moves 3.6 TB/s, at 262 144 replicas 4.3 TB/s.  The retiring wavefront idles most of its time.
//
// Ordering rests on three hardware facts, all within ONE compute unit (a workgroup never spans two; the kernel must not be built for
// threadgroup-split mode, where a workgroup's waves may sit on different compute units — hipcc's default is off and build.py passes no
// -mtgsplit): LDS performs one wavefront's operations in order; a wavefront's `s_waitcnt vmcnt(0)` returns when its stores have been
// performed at the compute unit's L1 / L2 path, which every other wavefront of the workgroup shares; relaxed LDS atomics separated by
// wavefront-scope fences are not reordered by the compiler.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "philox.hpp"
#include "det_math.hpp"
#include "spf_team_params.hpp"

namespace rrrmc {

#if defined(SPF_TEAM_STAMPS) && !defined(SPF_TEAM_COUNT_ONLY)
#define SPF_STAMP(i) do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_last; st_last = t_; } while (0)
#else
#define SPF_STAMP(i) do { } while (0)
#endif
// -DSPF_TEAM_TRACE (tools/ubench/spf_team_bench.hip only): team 0 writes the clock of eight events of every attempt into the sample buffer
#ifdef SPF_TEAM_TRACE
#define SPF_TRACE(it_, e_) do { if (blockIdx.x == 0 && lane == 0) reinterpret_cast<unsigned long long*>(P.Es)[(size_t)(it_) * 8 + (e_)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SPF_TRACE(it_, e_) do { } while (0)
#endif
#ifndef SPF_TEAM_NAP
#define SPF_TEAM_NAP 1
#endif
#ifndef SPF_TEAM_RETIRE_NAP
#define SPF_TEAM_RETIRE_NAP 1
#endif
#ifndef SPF_TEAM_BATCH
#define SPF_TEAM_BATCH 4
#endif
#ifndef SPF_TEAM_AGE1
#define SPF_TEAM_AGE1 24
#endif
#ifndef SPF_TEAM_AGE2
#define SPF_TEAM_AGE2 44
#endif
#ifndef SPF_TEAM_SCOPE
#define SPF_TEAM_SCOPE "workgroup"
#endif
static_assert(kSpfTeamWindow == 64, "the conflicts of an attempt are one 64-bit word");
__global__ __launch_bounds__(256) void spf_team_plan_kernel(const int32_t* __restrict__ A, const double* __restrict__ J, const int32_t* __restrict__ sites,
                                                            uint32_t* __restrict__ plan, int64_t n, int K)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;          // record t holds iteration t (1-based); 0 and n + 1 are padding
    if (t > n + 1) return;
    uint32_t* o = plan + (size_t)t * spf_plan_stride(K);
    if (t == 0 || t == n + 1) {
        for (int q = 0; q < spf_plan_stride(K); ++q) o[q] = 0u;
        return;
    }
    int mine[kSpfMaxK + 1];
    const int i = sites[t - 1];
    mine[0] = i;
    for (int k = 0; k < K; ++k) mine[1 + k] = A[(size_t)i * K + k];
    int32_t same = 0;
    unsigned long long conf = 0ull;
    for (int c = kSpfTeamWindow; c >= 1; --c) {
        if (c >= t) continue;
        const int j = sites[t - 1 - c];
        bool hit = false;
        for (int a = 0; a <= K; ++a) hit |= mine[a] == j;
        for (int k = 0; k < K; ++k) {
            const int y = A[(size_t)j * K + k];
            for (int a = 0; a <= K; ++a) hit |= mine[a] == y;
        }
        if (hit) conf |= 1ull << (c - 1);
        if (j == i) same = (int32_t)(t - c);
    }
    o[0] = (uint32_t)i;
    o[1] = (uint32_t)same;
    o[2] = (uint32_t)conf;
    o[3] = (uint32_t)(conf >> 32);
    for (int k = 0; k < K; ++k) {
        o[4 + k] = (uint32_t)mine[1 + k];
        const unsigned long long jb = (unsigned long long)__double_as_longlong(J[(size_t)i * K + k]);
        o[4 + K + 2 * k] = (uint32_t)jb;
        o[4 + K + 2 * k + 1] = (uint32_t)(jb >> 32);
    }
}

__device__ __forceinline__ int32_t spf_lds_ld(const int32_t* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void spf_lds_st(int32_t* p, int32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int32_t spf_lds_uniform(const int32_t* p)
{
    return __builtin_amdgcn_readfirstlane(spf_lds_ld(p));
}

// One Philox block serves two attempts; the spin word of a team is the team's TW bits of the group's 64-bit word
template <int TW> struct spf_word;
template <> struct spf_word<64> { typedef unsigned long long type; };
template <> struct spf_word<32> { typedef uint32_t type; };
template <> struct spf_word<16> { typedef uint16_t type; };

// grid = teams = W * (64 / TW), block NW * 64, dynamic LDS spf_team_lds_bytes(K, NW, M, TW).
//   TW = replicas per team (64, 32 or 16): a group of 64 replicas is walked by 64 / TW teams, each on its own lanes' part of the group's
//        [N][64] field lines and spin words (lanes >= TW of a team's wavefronts idle).  A chain's pace is bound by what ONE compute unit moves
//        (about 10 bytes per cycle of 512-byte lines that miss its L1), so with fewer groups than compute units narrower teams on more
//        compute units are faster: the same instructions serve fewer replicas, but each compute unit moves a half or a quarter of the bytes.
//   M  = slots (attempt `it` uses slot (it + c0) mod M, whoever runs it): how far the wavefronts may run ahead of the retired prefix
#ifdef SPF_TEAM_WAVES_PER_EU        // experiment (tools/ubench/spf_team_bench.hip): cap the registers so that two sixteen-wavefront teams share a compute unit
#define SPF_TEAM_OCC __attribute__((amdgpu_waves_per_eu(SPF_TEAM_WAVES_PER_EU, SPF_TEAM_WAVES_PER_EU)))
#else
#define SPF_TEAM_OCC
#endif
template <int K, int NW, int M, int TW>
__global__ __launch_bounds__(NW * 64) SPF_TEAM_OCC void spf_team_kernel(SpfTeamParams TP)
{
    constexpr int NX = NW - 1;
    constexpr int AREAS = spf_team_areas(NW, M);
    constexpr int TPG = 64 / TW;                      // teams per group
#ifdef SPF_TEAM_NO_FUSE
    constexpr bool FUSE = false;
#else
    constexpr bool FUSE = TW <= 32 && (K <= 6 || NW == 8);    // pairs of attempts in the two halves of a wavefront (beyond K = 6 the registers of a sixteen-wavefront workgroup run out)
#endif
    typedef typename spf_word<TW>::type word_t;
    static_assert(M <= kSpfTeamWindow && M < 64 && M >= 2 * NX, "the dependency window covers the attempts in flight; one wavefront read brings all flags; a pair per executing wavefront");
    static_assert(AREAS == M + NX + 1, "areas");
    const SpfParams& P = TP.S;
    extern __shared__ __attribute__((aligned(16))) unsigned char spf_team_lds[];
    typedef double nbr_t[K][TW];
    typedef double own_t[TW];
    typedef uint32_t tag_t[TW];
    nbr_t* const rec = reinterpret_cast<nbr_t*>(spf_team_lds);                                    // [AREAS] the K neighbour fields before the move
    own_t* const vown = reinterpret_cast<own_t*>(spf_team_lds + sizeof(nbr_t) * AREAS);           // [AREAS] the own field (+0.0: not accepted)
    tag_t* const vtag = reinterpret_cast<tag_t*>(spf_team_lds + (sizeof(nbr_t) + sizeof(own_t)) * AREAS);    // [AREAS] the tag
    int32_t* const tl = reinterpret_cast<int32_t*>(spf_team_lds + (sizeof(nbr_t) + sizeof(own_t) + sizeof(tag_t)) * AREAS);   // [TW] last accepted attempt as retired (0: before the launch)
    int32_t* const done = tl + TW;                    // [M] iteration whose results the slot holds
    int32_t* const ev = done + M;                     // [M] iteration that has evacuated the slot's previous records and may be overwriting it
    int32_t* const shP = ev + M;                      // retired prefix
    int32_t* const abortf = shP + 1;                  // set when a wait ran into its limit: nobody waits any more

    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), N = P.N;
    const int w = (int)(blockIdx.x / (unsigned)TPG), part = (int)(blockIdx.x % (unsigned)TPG);       // group of 64 replicas, the team's part of it
    const bool act = lane < TW;                       // lanes beyond the team's width idle (they keep valid addresses: lane dl's)
    const int dl = lane & (TW - 1);
    const int r = w * 64 + part * TW + dl;
    double* const lf = P.lf + (size_t)w * N * 64 + part * TW + dl;
    double* const undo = P.undo + (size_t)w * (K + 1) * 64 + part * TW + dl;
    word_t* const sp = reinterpret_cast<word_t*>(P.spins + (size_t)w * N) + part;        // word of site x: sp[x * TPG]
    const int64_t iters = P.iters;
    const uint64_t g0 = P.g0, hb = (g0 + 1) >> 1;        // first pair of the launch
    const int32_t c0 = (int32_t)((int64_t)g0 - 2 * (int64_t)hb);       // slot of iteration `it` = (it + c0) mod M; c0 = 0 or -1
    // the executing wavefront of iteration t (t >= 1)
    auto owner_of = [&](int32_t t) -> int32_t { return (int32_t)((((uint64_t)t + g0) >> 1) - hb) % NX; };
    const int32_t ml0 = P.move_last[r];                  // move_last before the launch
    // every wait of the kernel is on the retired prefix and bounded: a wait that runs into the limit raises the workgroup's abort flag, after
    // which no wavefront waits any more (the launch drains with void results and says so in *TP.status)
    auto wait_prefix = [&](int32_t need, int nap) {
        int32_t polls = 0;
        while (spf_lds_uniform(shP) < need) {
            if (nap > 1) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(1);
            if (++polls > kSpfTeamSpinLimit) { if (lane == 0) spf_lds_st(abortf, 1); break; }
            if ((polls & 1023) == 0 && spf_lds_uniform(abortf)) break;
        }
    };

    if (wv == NW - 1) {
#pragma unroll
        for (int k = 0; k < K; ++k) { const double u_ = undo[(size_t)k * 64]; if (act) rec[M + NX][k][dl] = u_; }
        { const double u_ = undo[(size_t)K * 64]; if (act) { vown[M + NX][dl] = u_; tl[dl] = 0; } }
        if (lane < M) { done[lane] = 0; ev[lane] = 0; }
        if (lane == 0) { *shP = 0; *abortf = 0; }
    }
    __syncthreads();

    if (wv == NW - 1) {
        // ---- retire: the chain's order.  Only LDS traffic (in order per wavefront): compiler barriers, no waits ----
#ifndef SPF_TEAM_NOPRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        double E = P.E_cur[r];
        int32_t tlr = 0;
        int64_t ns = P.sample0;
        // the launch-relative iteration of the next sample (RRRMC.jl:104-108: before the move of every step-th iteration of the call)
        const int64_t until = P.step - (P.it_off % P.step);
        const int32_t step32 = (int32_t)(P.step > (int64_t)1 << 30 ? (int64_t)1 << 30 : P.step);
        int32_t samp = (int32_t)(until > (int64_t)1 << 30 ? (int64_t)1 << 30 : until);
        constexpr int B = SPF_TEAM_BATCH;                          // attempts retired per look at the flags
        static_assert(B >= 1 && B <= 8, "batch");
#ifdef SPF_TEAM_STAMPS
        unsigned long long rt_rounds = 0, rt_idle = 0; const uint64_t rt_t0 = __builtin_amdgcn_s_memtime();
#endif
        const int32_t n32 = (int32_t)iters;
        int s0 = (int)((g0 + 1) & 1u);                             // slot of iteration 1
        int32_t it = 1;
        const int32_t* const fl = done + (lane < M ? lane : 0);
        // one read brings the flags of all slots; then the data of the attempts that have reported, behind their flags (LDS serves a
        // wavefront's reads in order, so data read after a flag that says `it` is that attempt's); the next look at the flags is already
        // under way while the batch is worked off
        int32_t f = spf_lds_ld(fl), idle = 0;
        while (it <= n32) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");        // compiler barrier: the data reads stay behind the flag read
            if (it == samp) {                                      // sample before the move
#ifndef SPF_TEAM_TRACE
                if (P.Es && act) P.Es[(size_t)ns * P.Rpad + r] = E;
#endif
                ++ns;
                samp += step32;
            }
            // slot `lane` holds the attempt it is due to hold (iteration it + lane - s0 for lane >= s0) <=> flag - lane == it - s0; the attempts
            // that have reported in chain order = the run of set bits from s0 (a batch ends at the ring's end and before the next sample)
            const unsigned long long m = __builtin_amdgcn_ballot_w64(f - lane == it - s0) & ((1ull << M) - 1ull);
            int ready = __builtin_ctzll(~(m >> s0));
            ready = ready > B ? B : ready;
            ready = ready > samp - it ? samp - it : ready;
#ifdef SPF_TEAM_STAMPS
            ++rt_rounds; if (ready == 0) ++rt_idle;
#endif
            if (ready == 0) {
                __builtin_amdgcn_s_sleep(SPF_TEAM_RETIRE_NAP);
                f = spf_lds_ld(fl);
                if (++idle > kSpfTeamSpinLimit || ((idle & 1023) == 0 && spf_lds_uniform(abortf))) {        // see wait_prefix
                    if (lane == 0) { spf_lds_st(abortf, 1); spf_lds_st(shP, n32); if (TP.status) *TP.status = 1; }
                    break;
                }
                continue;
            }
            idle = 0;
            const double* const q0 = &vown[s0][dl];
            const uint32_t* const g0p = &vtag[s0][dl];
            // n attempts, straight-line: n reads, the next look at the flags, then per attempt one add and one max
#define SPF_RETIRE_CASE(n)                                                                                                      \
            case n: {                                                                                                           \
                double q[n];                                                                                                    \
                uint32_t qt[n];                                                                                                 \
                _Pragma("unroll") for (int j = 0; j < n; ++j) { q[j] = q0[j * TW]; qt[j] = g0p[j * TW]; }                       \
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                                          \
                f = spf_lds_ld(fl);                                                                                             \
                _Pragma("unroll") for (int j = 0; j < n; ++j) {                                                                 \
                    E = __dadd_rn(E, -q[j]);                       /* dE = -lfields[i] (RRG.jl:619-625); - (+0.0) changes nothing */ \
                    tlr = (int32_t)qt[j] > tlr ? (int32_t)qt[j] : tlr;                                                          \
                }                                                                                                               \
            } break;
            switch (ready) {
                SPF_RETIRE_CASE(1)
#if SPF_TEAM_BATCH >= 2
                SPF_RETIRE_CASE(2)
#endif
#if SPF_TEAM_BATCH >= 3
                SPF_RETIRE_CASE(3)
#endif
#if SPF_TEAM_BATCH >= 4
                SPF_RETIRE_CASE(4)
#endif
#if SPF_TEAM_BATCH >= 5
                SPF_RETIRE_CASE(5)
#endif
#if SPF_TEAM_BATCH >= 6
                SPF_RETIRE_CASE(6)
#endif
#if SPF_TEAM_BATCH >= 7
                SPF_RETIRE_CASE(7)
#endif
#if SPF_TEAM_BATCH >= 8
                SPF_RETIRE_CASE(8)
#endif
            default: break;
            }
#undef SPF_RETIRE_CASE
            if (act) spf_lds_st(tl + dl, tlr);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                // compiler barrier: LDS performs a wavefront's writes in order
            if (lane == 0) spf_lds_st(shP, it + ready - 1);
#ifdef SPF_TEAM_TRACE
            for (int j = 0; j < ready; ++j) SPF_TRACE(it + j, 7);
#endif
            it += ready;
            s0 = s0 + ready >= M ? 0 : s0 + ready;                 // a batch never wraps: ready <= M - s0
        }
#ifdef SPF_TEAM_STAMPS
        if (w == 0 && lane == 0) {
            unsigned long long* o = reinterpret_cast<unsigned long long*>(P.Es) + (size_t)NX * 8;
            o[0] = rt_rounds; o[1] = rt_idle; o[2] = __builtin_amdgcn_s_memtime() - rt_t0; o[3] = 0; o[4] = 0; o[5] = 0; o[6] = 0;
        }
#endif
        if (TP.status && spf_lds_uniform(abortf) && lane == 0) *TP.status = 1;
        if (act) {
            int r_ = r;
            asm volatile("" : "+v"(r_));            // (the addresses are formed here, not carried from the head of the kernel in registers)
            P.E_cur[r_] = E;
            P.move_last[r_] = tlr ? P.sites[tlr - 1] : ml0;
            // every executing wavefront has reported its last attempt: nothing moves any more
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            int area = M + NX;
            if (tlr) {
                const int sl = (tlr + c0) % M;
                area = spf_lds_ld(ev + sl) == tlr ? sl : M + owner_of(tlr + M);
            }
#pragma unroll
            for (int k = 0; k < K; ++k) undo[(size_t)k * 64] = rec[area][k][dl];
            undo[(size_t)K * 64] = vown[area][dl];
        }
        return;
    }

    // ---- execute ----
    const uint32_t replica = P.replica0 + (uint32_t)r;
    const int x = wv;
    constexpr int S = spf_plan_stride(K);
    static_assert(2 * S <= 64, "the two records of a pair are fetched by one wavefront load");
    // the records of the wavefront's next pair travel one pair ahead of their use
    // launch-relative 32-bit arithmetic on the hot path: pair h (h < 2^18) holds iterations 2 h - c0 and 2 h - c0 + 1
    const int32_t n32 = (int32_t)iters;
    auto fetch_pair = [&](int32_t h) -> uint32_t {
        const int32_t it_even = 2 * h - c0;
        const int32_t rec0 = it_even > n32 ? n32 : it_even;         // it_even >= 0; clamped: a pair beyond the launch is fetched and never used
        return TP.plan[(uint32_t)rec0 * (uint32_t)S + (uint32_t)(lane < 2 * S ? lane : 0)];
    };
    auto load_word = [&](int x_) -> word_t { return __hip_atomic_load(sp + (size_t)x_ * TPG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); };
    auto site_of_tl = [&](int32_t t) -> int32_t {                   // move_last of a lane whose last accepted attempt is t
        const int32_t s = P.sites[(t > 0 ? t : 1) - 1];
        return t > 0 ? s : ml0;
    };
    char* const lfb = reinterpret_cast<char*>(P.lf + (size_t)w * N * 64 + part * TW);                   // wave-uniform bases of the team's lines and words
    char* const spb = reinterpret_cast<char*>(reinterpret_cast<word_t*>(P.spins + (size_t)w * N) + part);
    const uint32_t dl8 = (uint32_t)dl * 8u;
    uint32_t pr_next = fetch_pair(x), pr_next2 = fetch_pair(x + NX);
    int32_t nacc = 0;
#ifdef SPF_TEAM_STAMPS
    uint64_t st_acc[6] = {0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime(), st_n = 0;
    uint64_t gw[6] = {0, 0, 0, 0, 0, 0};             // gate waits by the window (count, cycles) and by a conflict (count, cycles); slot waits (count, cycles)
#endif
    int slot_even = (2 * x) % M;                                    // slot of the pair's first attempt: (2 h) mod M, kept by addition (a 64-bit modulo per pair otherwise)
    for (int32_t h = x;; h += NX, slot_even = slot_even + 2 * NX >= M ? slot_even + 2 * NX - M : slot_even + 2 * NX) {
        const uint64_t blk_id = hb + (uint64_t)h;
        const int32_t it_even = 2 * h - c0;                         // iteration whose stream index g is even
        if (it_even > n32) break;
        const uint32_t pr = pr_next;
        pr_next = pr_next2;
        pr_next2 = fetch_pair(h + 2 * NX);
        const Philox4 blk = philox4x32_10((uint32_t)blk_id, (uint32_t)(blk_id >> 32), replica, TAG_ACCEPT_F64, P.k0, P.k1);
        // The two attempts of the pair, A (e = 0) and B (e = 1).  Their state-independent parts first; then, unless B depends on A, the
        // loads of BOTH go out before either is worked off (one memory latency for the two), and both are reported behind one wait for the stores.
        int32_t itv[2], sl_[2], site[2], same[2];
        unsigned long long conf[2];
        bool valid[2];
        int y[2][K];
        double J[2][K], U[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int32_t it = it_even + e;
            valid[e] = it >= 1 && it <= n32;
            itv[e] = it;
            sl_[e] = slot_even + e >= M ? slot_even + e - M : slot_even + e;
            const int o = e * S;
            same[e] = __builtin_amdgcn_readlane((int)pr, o + 1);
            conf[e] = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)pr, o + 3) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)pr, o + 2);
        }
        // what an attempt works with as wave-uniform values (a fused pair takes them per lane instead and skips this)
        auto scalars = [&]() {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int o = e * S;
                site[e] = __builtin_amdgcn_readlane((int)pr, o + 0);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    y[e][k] = __builtin_amdgcn_readlane((int)pr, o + 4 + k);
                    const uint32_t jl = (uint32_t)__builtin_amdgcn_readlane((int)pr, o + 4 + K + 2 * k), jh = (uint32_t)__builtin_amdgcn_readlane((int)pr, o + 4 + K + 2 * k + 1);
                    J[e][k] = __longlong_as_double((long long)(((unsigned long long)jh << 32) | jl));
                }
                const uint64_t u = e ? (((uint64_t)blk.w[2] << 32) | blk.w[3]) : (((uint64_t)blk.w[0] << 32) | blk.w[1]);
                U[e] = (double)(u >> 11) * 0x1.0p-53;
            }
        };
        SPF_STAMP(0);                                               // state-independent preparation
        double lfi[2], nf[2][K];
        word_t wi[2], nw[2][K];
        int32_t tlv[2], mls[2], pfx[2];
        bool reported[2] = {false, false};

        // the loads of attempt e (not behind its slot: nothing of the slot is touched before finish).  They go out
        //   - behind everything the planner's window does not reach back to (a wavefront's requests may run ahead of the retired prefix by the
        //     slots plus its own pairs), and
        //   - behind every attempt of the window that does not commute with this one: those must have REPORTED (their stores performed), not
        //     retired.  Lane c - 1 looks at iteration j = it - c: the flag of its slot says j, or a later user of the slot, once j has reported.
        // One LDS round trip brings the flags, the retired prefix and, behind it (tl is written before the prefix), move_last as retired so far —
        // enough to rule the undo path out if every earlier attempt at this site has retired (same <= prefix).
        bool reused = false;
        auto gate = [&](auto ec) {
            constexpr int e = decltype(ec)::value;
#ifdef SPF_TEAM_EXP_NOPROTO          // timing experiment (wrong results): no look at flags, prefix or tl at all
          const bool reuse = true;
          if (e == 0) { pfx[0] = spf_lds_uniform(shP); tlv[0] = 0; mls[0] = -1; }
#else
          const bool reuse = e == 1 && valid[0] && conf[1] == 0ull && pfx[0] >= itv[1] - kSpfTeamWindow;         // wave-uniform
#endif
          if (reuse) {
            // the pair's second attempt conflicts with nothing in the window: the first one's look at the prefix and at tl (a moment ago) serves it too
            pfx[1] = pfx[0];
            tlv[1] = tlv[0];
          } else {
            const int32_t j = itv[e] - 1 - lane;
            const bool mine = j >= 1 && ((conf[e] >> lane) & 1ull);
            const int32_t* const fj = done + (mine ? (j + c0) % M : 0);
            int32_t f = 0x7fffffff;
            if (conf[e] != 0ull) f = spf_lds_ld(fj);                // (six attempts in seven conflict with nothing in the window)
            int32_t pv = spf_lds_ld(shP);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            int32_t t0 = spf_lds_ld(tl + dl);
            int32_t pnow = __builtin_amdgcn_readfirstlane(pv);
            if (pnow < itv[e] - kSpfTeamWindow || __builtin_amdgcn_ballot_w64(mine && f < j) != 0ull) {
#ifdef SPF_TEAM_STAMPS
                const uint64_t g_t0 = __builtin_amdgcn_s_memtime(); const bool g_byw = pnow < itv[e] - kSpfTeamWindow;
#endif
                wait_prefix(itv[e] - kSpfTeamWindow, SPF_TEAM_NAP);
                int32_t polls = 0;
                while (__builtin_amdgcn_ballot_w64(mine && spf_lds_ld(fj) < j) != 0ull) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++polls > kSpfTeamSpinLimit) { if (lane == 0) spf_lds_st(abortf, 1); break; }
                    if ((polls & 1023) == 0 && spf_lds_uniform(abortf)) break;
                }
                pnow = spf_lds_uniform(shP);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                t0 = spf_lds_ld(tl + dl);
#ifdef SPF_TEAM_STAMPS
                { const uint64_t g_dt = __builtin_amdgcn_s_memtime() - g_t0; if (g_byw) { ++gw[0]; gw[1] += g_dt; } else { ++gw[2]; gw[3] += g_dt; } }
#endif
            }
            pfx[e] = pnow;
            tlv[e] = t0;
          }
            if (e == 1) reused = reuse;
            SPF_TRACE(itv[e], 0);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, SPF_TEAM_SCOPE);
        };
        auto request = [&](auto ec) {
            constexpr int e = decltype(ec)::value;
            gate(ec);
            const bool reuse = e == 1 && reused;
#ifdef SPF_TEAM_EXP_NOMEM
            // timing experiment (tools/ubench/spf_team_bench.hip; wrong results): no global memory traffic at all — what the protocol alone costs
            lfi[e] = 0.25 * (double)((dl + site[e]) & 7) - 1.0;
            wi[e] = (word_t)0x5555aaaa5555aaaaull;
#pragma unroll
            for (int k = 0; k < K; ++k) { nf[e][k] = 0.5 * k; nw[e][k] = (word_t)(0x3333cccc3333ccccull + (unsigned long long)y[e][k]); }
#else
            lfi[e] = lf[(size_t)site[e] * 64];
            wi[e] = load_word(site[e]);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                nf[e][k] = lf[(size_t)y[e][k] * 64];
                nw[e][k] = load_word(y[e][k]);
            }
#endif
#ifdef SPF_TEAM_EXP_NOPROTO
            mls[e] = -1;
#else
            if (reuse) mls[1] = mls[0]; else mls[e] = site_of_tl(tlv[e]);
#endif
        };
        // the attempt is reported once its field and spin stores have been PERFORMED: a workgroup-scope release does not wait for vector stores
        // (it relies on one compute unit issuing them in order; loads of another SIMD were seen to overtake them)
        auto report = [&](auto ec) {
            constexpr int e = decltype(ec)::value;
            if (!valid[e] || reported[e]) return;
#ifndef SPF_TEAM_EXP_NOPROTO2
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, SPF_TEAM_SCOPE);
            if (lane == 0) spf_lds_st(done + sl_[e], itv[e]);
            SPF_TRACE(itv[e], 6);
            reported[e] = true;
        };
        bool accv[2] = {false, false};
        unsigned long long amaskv[2] = {0ull, 0ull};
        auto decide = [&](auto ec) {
            constexpr int e = decltype(ec)::value;
            SPF_TRACE(itv[e], 1);
            const double dE = -lfi[e];                              // delta_energy: RRG.jl:619-625
            const double xx = __dmul_rn(-P.beta, dE);
            // accept (RRRMC.jl:39): xx >= 0 || U < det_exp(xx).  det_exp is some fifty Float64 instructions in a dependent chain; the hardware's
            // 2^x (v_exp_f32, 1 ulp) on the Float32 image of xx is within 2e-5 of it wherever it is normal (|xx| < 87: 2^-24 |xx| log2(e) of
            // argument rounding twice over, the constant, the instruction's ulp; det_exp itself is good to 1e-14), so outside a band of 2^-13
            // around that estimate the comparison is decided by it, and only a wavefront with a replica INSIDE the band (about one attempt in
            // two hundred) evaluates det_exp.  Below |xx| = 87 the estimate underflows towards 0 with exp(xx) < 2^-126 far under the smallest
            // positive U (2^-53): U > 0 is rejected, U == 0 falls into the band.  The decision is det_exp's in every case.
            {
                const float e32 = __builtin_amdgcn_exp2f(__fmul_rn((float)xx, 1.44269504088896340736f));
                const double est = (double)e32;
                const bool sure_yes = xx >= 0.0 || U[e] < __dmul_rn(est, 1.0 - 0x1.0p-13);
                const bool sure_no = !(xx >= 0.0) && U[e] > __dmul_rn(est, 1.0 + 0x1.0p-13);
                bool a_ = sure_yes;
                if (__builtin_amdgcn_ballot_w64(!sure_yes && !sure_no) != 0ull) a_ = xx >= 0.0 || U[e] < det_exp(xx);
                accv[e] = a_;
            }
            amaskv[e] = __builtin_amdgcn_ballot_w64(accv[e]) & (TW == 64 ? ~0ull : (1ull << (TW & 63)) - 1ull);     // lanes >= TW repeat lane dl
            nacc += accv[e] && act ? 1 : 0;
            SPF_STAMP(2);                                           // loads + decision
            SPF_TRACE(itv[e], 2);
        };
        auto finish = [&](auto ec) {
            constexpr int e = decltype(ec)::value;
            const int32_t it = itv[e];
            const int i = site[e], s = sl_[e];
            const bool acc = accv[e];
            const unsigned long long amask = amaskv[e];

            bool fast = false;
            double sv[K];
#pragma unroll
            for (int k = 0; k < K; ++k) sv[k] = 0.0;
            if (same[e] > pfx[e] || __builtin_amdgcn_ballot_w64(acc && mls[e] == i) != 0ull) {
                // an earlier attempt at this site had not retired when tl was read, or some accepting replica's last retired accepted move is at
                // this site: exact only once everything before this attempt has retired (the retiring wavefront then rests until this attempt
                // reports, so tl and the records of its lanes are stable).  The pair's first attempt is among "everything before" the second:
                // it is reported first
                if (e == 1) report(std::integral_constant<int, 0>{});
                pfx[e] = it - 1;
                wait_prefix(it - 1, 1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, SPF_TEAM_SCOPE);
                tlv[e] = spf_lds_ld(tl + dl);
                mls[e] = site_of_tl(tlv[e]);
                fast = acc && mls[e] == i;
                if (fast) {
                    // the record of attempt tlv: the slot's data first, the slot's mark second (see the head of the file)
                    if (tlv[e] > 0) {
                        const int sl = (tlv[e] + c0) % M;
                        double tmp[K];
#pragma unroll
                        for (int k = 0; k < K; ++k) tmp[k] = __hip_atomic_load(&rec[sl][k][dl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        const bool in_slot = spf_lds_ld(ev + sl) == tlv[e];
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        const int keep = M + owner_of(tlv[e] + M);
#pragma unroll
                        for (int k = 0; k < K; ++k) sv[k] = in_slot ? tmp[k] : __hip_atomic_load(&rec[in_slot ? sl : keep][k][dl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else {
#pragma unroll
                        for (int k = 0; k < K; ++k) sv[k] = rec[M + NX][k][dl];
                    }
                }
            }
            const bool slow = acc && !fast;
            SPF_TRACE(it, 3);

            // the slot: free once its previous use (attempt it - M) has retired.  EVACUATION: that attempt's records that are still some replica's
            // last accepted move go to the wavefront's own keep before the slot is marked and written.  A stale look at tl (the lane has
            // accepted since) costs a wasted copy.
            if (it > M) {
                // tl as read with the request will do if the slot was free by then: a replica whose last accepted move was not it - M at a prefix
                // >= it - M has accepted something later
                if (pfx[e] < it - M) {
                    wait_prefix(it - M, 1);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    tlv[e] = spf_lds_ld(tl + dl);
                }
                SPF_TRACE(it, 4);
                const bool live = act && tlv[e] == it - M;
                if (__builtin_amdgcn_ballot_w64(live) != 0ull) {
                    if (live) {
#pragma unroll
                        for (int k = 0; k < K; ++k) rec[M + x][k][dl] = rec[s][k][dl];
                        vown[M + x][dl] = vown[s][dl];
                    }
                }
            }
#ifndef SPF_TEAM_EXP_NOPROTO2
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (lane == 0) spf_lds_st(ev + s, it);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#endif

#ifdef SPF_TEAM_EXP_NOMEM
            if (amask == 0xdeadbeefull) {
#else
            if (amask != 0ull) {
#endif
                // update_cache! (RRG.jl:576-617, EA.jl:613-653).  Only the accepting lanes store (the memory side skips the sectors nobody wrote:
                // 4 - 7 % of the kernel's time against whole lines written back)
                const uint32_t snew = (uint32_t)((wi[e] >> dl) & 1u) ^ 1u;
                double vrun = 0.0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    // two bonds to the same neighbour (GraphEANormal with L = 2; rows are sorted): the second one continues from the first one's
                    // result (EA.jl:626-640 walks all entries), and its store — same wavefront, same address, in order — is the one that stays
                    const bool rep = k > 0 && y[e][k] == y[e][k - 1];            // wave-uniform
                    const uint32_t sbit = (uint32_t)((nw[e][k] >> dl) & 1u);
                    const double c = (snew ^ sbit) ? -4.0 : 4.0;                 // 4 * sigma_xy with the NEW s_x
                    const double v = __dadd_rn(rep ? vrun : nf[e][k], -__dmul_rn(c, J[e][k]));
                    vrun = v;
                    const double out = fast ? sv[k] : (slow ? v : nf[e][k]);
                    if (act && acc) lf[(size_t)y[e][k] * 64] = out;
                }
                if (act && acc) lf[(size_t)i * 64] = -lfi[e];
                if (lane == 0) __hip_atomic_store(sp + (size_t)i * TPG, (word_t)(wi[e] ^ (word_t)amask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (act) {
#ifndef SPF_TEAM_EXP_NOPROTO2
#pragma unroll
                for (int k = 0; k < K; ++k) rec[s][k][dl] = nf[e][k];           // lfields_last[y] = lfields[y] (slow) / the swap (fast)
#endif
                vown[s][dl] = acc ? lfi[e] : 0.0;
                vtag[s][dl] = acc ? (uint32_t)it : 0u;
            }
            SPF_STAMP(3);                                           // undo test, update, stores issued
            SPF_TRACE(it, 5);
#ifdef SPF_TEAM_STAMPS
            ++st_n;
#endif
        };
        const std::integral_constant<int, 0> eA{};
        const std::integral_constant<int, 1> eB{};
#ifdef SPF_TEAM_NO_PAIRING
        const bool together = false;
#else
        const bool together = valid[0] && valid[1] && !(conf[1] & 1ull);        // B commutes with A
#endif
        if constexpr (FUSE) {
            if (together) {
                // FUSED PAIR (teams of 32 or 16 replicas): the two attempts commute, and a wavefront has twice the team's lanes — lanes 0..31 work
                // attempt A off, lanes 32..63 attempt B for the same replicas, with ONE instruction stream for the loads, the decision, the update
                // and the records.  What differs per attempt (site, neighbours, couplings, slot, random number) is selected per lane.
                gate(eA);
                gate(eB);
#ifndef SPF_TEAM_NO_AGEPRIO
                // oldest attempt first: a SIMD's arbiter prefers its older wavefronts, so the younger ones (higher index) fall behind and the whole
                // team then waits for them at the window; the pair that is closest to the retired prefix takes the higher priority instead
                {
                    const int32_t lag = itv[0] - pfx[0];
                    if (lag < SPF_TEAM_AGE1) __builtin_amdgcn_s_setprio(2);
                    else if (lag < SPF_TEAM_AGE2) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                }
#endif
                const bool hi = lane >= 32;
                const bool fact = (lane & 31) < TW;
                const int32_t itL = hi ? itv[1] : itv[0];
                const int sL = hi ? sl_[1] : sl_[0];
                int32_t tlvL = hi ? tlv[1] : tlv[0];
                const int32_t pmin = pfx[0] < pfx[1] ? pfx[0] : pfx[1];
                // the lane's record of the pair: dword f of attempt A in lanes 0..31, of attempt B in lanes 32..63 (a crossbar read of the fetched pair)
                const int rb = hi ? 4 * S : 0;
                auto field = [&](int f) -> uint32_t { return (uint32_t)__builtin_amdgcn_ds_bpermute(rb + 4 * f, (int)pr); };
                const uint32_t siteL = field(0);
                uint32_t yL[K];
                double JL[K], nfL[K];
                word_t nwL[K];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    yL[k] = field(4 + k);
                    JL[k] = __longlong_as_double((long long)(((unsigned long long)field(4 + K + 2 * k + 1) << 32) | field(4 + K + 2 * k)));
                }
                // 32-bit offsets from the team's wave-uniform bases (N < 2^23 sites of 512 bytes: host_spf.hpp, spf_use_team)
                auto LF = [&](uint32_t site_) -> double* { return reinterpret_cast<double*>(lfb + (site_ * 512u + dl8)); };
                auto SP = [&](uint32_t site_) -> word_t* { return reinterpret_cast<word_t*>(spb + site_ * 8u); };
#ifdef SPF_TEAM_EXP_NOMEM
                const double lfiL = 0.25 * (double)((dl + siteL) & 7) - 1.0;
                const word_t wiL = (word_t)0x5555aaaa5555aaaaull;
#pragma unroll
                for (int k = 0; k < K; ++k) { nfL[k] = 0.5 * k; nwL[k] = (word_t)(0x3333cccc3333ccccull + (unsigned long long)yL[k]); }
#else
                const double lfiL = *LF(siteL);
                const word_t wiL = __hip_atomic_load(SP(siteL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#pragma unroll
                for (int k = 0; k < K; ++k) { nfL[k] = *LF(yL[k]); nwL[k] = __hip_atomic_load(SP(yL[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
#endif
                const int32_t mlsL = site_of_tl(tlvL);
                SPF_TRACE(itv[0], 1); SPF_TRACE(itv[1], 1);
                const uint32_t uh = hi ? blk.w[2] : blk.w[0], ul = hi ? blk.w[3] : blk.w[1];
                const double UL = (double)((((uint64_t)uh << 32) | ul) >> 11) * 0x1.0p-53;
                SPF_STAMP(1);
                bool accL;
                {
                    const double xx = __dmul_rn(-P.beta, -lfiL);
                    const float e32 = __builtin_amdgcn_exp2f(__fmul_rn((float)xx, 1.44269504088896340736f));
                    const double est = (double)e32;
                    const bool sure_yes = xx >= 0.0 || UL < __dmul_rn(est, 1.0 - 0x1.0p-13);
                    const bool sure_no = !(xx >= 0.0) && UL > __dmul_rn(est, 1.0 + 0x1.0p-13);
                    accL = sure_yes;
                    if (__builtin_amdgcn_ballot_w64(!sure_yes && !sure_no) != 0ull) accL = xx >= 0.0 || UL < det_exp(xx);
                }
                const unsigned long long amask = __builtin_amdgcn_ballot_w64(accL && fact);
                SPF_TRACE(itv[0], 2); SPF_TRACE(itv[1], 2);
                nacc += accL && fact ? 1 : 0;
                SPF_STAMP(2);
                // the undo path may be due (about one pair in a thousand): nothing has been written yet, and the pair starts again below as two
                // attempts one after the other
                const bool plain = !(same[0] > pfx[0] || same[1] > pfx[1] || __builtin_amdgcn_ballot_w64(accL && fact && mlsL == (int32_t)siteL) != 0ull);
                if (plain) {
                // the slots: free once attempts itA - M and itB - M have retired; their records that are still a replica's last accepted move go to the keep
                if (itv[1] > M) {
                    if (pmin < itv[1] - M) {
#ifdef SPF_TEAM_STAMPS
                        const uint64_t g_t0 = __builtin_amdgcn_s_memtime();
#endif
                        wait_prefix(itv[1] - M, 1);
#ifdef SPF_TEAM_STAMPS
                        ++gw[4]; gw[5] += __builtin_amdgcn_s_memtime() - g_t0;
#endif
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        tlvL = spf_lds_ld(tl + dl);
                    }
                    const bool live = fact && itL > M && tlvL == itL - M;       // at most one of a replica's two lanes
                    if (__builtin_amdgcn_ballot_w64(live) != 0ull) {
                        if (live) {
#pragma unroll
                            for (int k = 0; k < K; ++k) rec[M + x][k][dl] = rec[sL][k][dl];
                            vown[M + x][dl] = vown[sL][dl];
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                if ((lane & 31) == 0) spf_lds_st(ev + sL, itL);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#ifdef SPF_TEAM_EXP_NOMEM
                if (amask == 0xdeadbeefull) {
#else
                if (amask != 0ull) {
#endif
                    const uint32_t snew = (uint32_t)((wiL >> dl) & 1u) ^ 1u;
                    double vrun = 0.0;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const bool rep = k > 0 && yL[k] == yL[k - 1];
                        const uint32_t sbit = (uint32_t)((nwL[k] >> dl) & 1u);
                        const double c = (snew ^ sbit) ? -4.0 : 4.0;
                        const double v = __dadd_rn(rep ? vrun : nfL[k], -__dmul_rn(c, JL[k]));
                        vrun = v;
                        if (fact && accL) *LF(yL[k]) = v;
                    }
                    if (fact && accL) *LF(siteL) = -lfiL;
                    if ((lane & 31) == 0) __hip_atomic_store(SP(siteL), (word_t)(wiL ^ (word_t)(hi ? amask >> 32 : amask)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if (fact) {
#pragma unroll
                    for (int k = 0; k < K; ++k) rec[sL][k][dl] = nfL[k];
                    vown[sL][dl] = accL ? lfiL : 0.0;
                    vtag[sL][dl] = accL ? (uint32_t)itL : 0u;
                }
                SPF_STAMP(3);
                SPF_TRACE(itv[0], 5); SPF_TRACE(itv[1], 5);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, SPF_TEAM_SCOPE);
                if ((lane & 31) == 0) spf_lds_st(done + sL, itL);
                SPF_TRACE(itv[0], 6); SPF_TRACE(itv[1], 6);
                SPF_STAMP(4);
#ifdef SPF_TEAM_STAMPS
                st_n += 2;
#endif
                continue;
                }
                nacc -= accL && fact ? 1 : 0;
            }
        }
        scalars();
        if constexpr (FUSE) {
            // what is left over by the fused pairs (a second attempt that depends on the first, the ends of a launch, the undo path): one attempt
            // after the other, with no overlap to pay registers for
            if (valid[0]) { request(eA); decide(eA); finish(eA); report(eA); }
            if (valid[1]) { request(eB); decide(eB); finish(eB); report(eB); }
            continue;
        }
        if (valid[0]) request(eA);
        if (valid[1] && (together || !valid[0])) request(eB);
        SPF_STAMP(1);                                               // waiting for the dependencies, requests out
        if (valid[0]) decide(eA);
        if (valid[1] && (together || !valid[0])) decide(eB);
        if (valid[0]) finish(eA);               // (finishing A before deciding B was measured: no difference)
        if (valid[1]) {
            if (valid[0] && !together) {
                report(eA);                                         // B's request waits for A's report
                request(eB);
                decide(eB);
            }
            finish(eB);
        }
        report(eA);
        report(eB);
        SPF_STAMP(4);                                               // stores performed, attempts reported
    }
    // accepted moves: a count, whatever the order (the host clears acc_cur at the start of the call)
    // (a fused pair counts its second attempt in the upper half of the wavefront: two lanes of one replica)
    if ((lane & 31) < TW) atomicAdd(reinterpret_cast<unsigned long long*>(P.acc_cur) + r, (unsigned long long)nacc);
#ifdef SPF_TEAM_STAMPS
    if (w == 0 && lane == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(P.Es) + (size_t)x * 8;      // harness only: the sample buffer is not compared in this build
        for (int q = 0; q < 5; ++q) o[q] = st_acc[q];
        o[5] = st_n;
        unsigned long long* o2 = reinterpret_cast<unsigned long long*>(P.Es) + (size_t)(NX + 1 + x) * 8;
        for (int q = 0; q < 6; ++q) o2[q] = gw[q];
    }
#endif
}

}  // namespace rrrmc
