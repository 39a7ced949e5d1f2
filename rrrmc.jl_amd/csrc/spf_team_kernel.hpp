// gfx950 kernel for standardMC (src/RRRMC.jl:81-127) on the Float64-coupling sparse models (GraphRRGNormal / GraphEANormal:
// src/graphs/RRG.jl:503-627, src/graphs/EA.jl:534-680): a TEAM of wavefronts per group of 64 replicas (round 4), with the chain-ordered
// part cut down to three instructions per attempt (round 5).
//
// spf_sweep_kernel (spf_kernels.hpp) walks the chain with ONE wavefront per group of 64 replicas: one attempt after the other, ~ 1 us each,
// whatever the machine has idle (8192 replicas = 128 wavefronts on 1024 SIMDs).  The attempts of a chain are not all dependent on each other:
// the SITE stream is known in advance, an attempt at site i reads lfields[i] and the spins of i and its neighbours and writes lfields of i and
// its neighbours (update_cache!, RRG.jl:576-617) — two attempts whose closed neighbourhoods do not meet commute exactly (no field receives
// an addition from both, so no rounding order changes).  This kernel runs such attempts side by side:
//
//   * NW - 1 executing wavefronts per group take the attempts of the chain in pairs (one Philox block serves iterations 2h, 2h + 1),
//     pair h on wavefront h mod (NW - 1).  Before it touches memory, the wavefront of attempt t waits until every attempt up to
//     need(t) = max(dep(t), t - M) has RETIRED (M = slots = attempts in flight at most), dep(t) = the latest earlier attempt within the window
//     whose closed neighbourhood meets t's (spf_team_plan_kernel: state-independent, once per launch for all groups).  Then it runs the
//     reference's attempt — accept (RRRMC.jl:39), the K + 1 field updates, the spin word — on the group's [N][64] arrays in HBM / L2, exactly
//     as spf_sweep_kernel does, and counts its own accepted moves (a count does not care about order).
//   * one RETIRING wavefront keeps what the reference's loop keeps in chain order: the tracked energy (E += dE is a Float64 running sum:
//     its order is the chain's), the samples (RRRMC.jl:104-108) and, per replica, WHICH attempt its last accepted move was (tl).  It consumes
//     the attempts strictly in order, one 16-byte read per attempt: {the attempted site's own field, or +0.0 for a lane that did not accept
//     (E - 0.0 == E for every E); the attempt's index, or 0}.  Per attempt that is one read, one Float64 add and one integer max.
//   * the undo path (RRG.jl:583-593: a move of the spin that was also the replica's last accepted move swaps lfields <-> lfields_last) needs
//     move_last as of t - 1.  move_last = site(tl): the executing wavefront reads tl as retired so far and gathers the sites; if an accepting
//     lane has site(tl) == site(t) it waits until t - 1 has retired — the retiring wavefront then rests until t itself reports — and reads
//     again; without such a lane no in-flight attempt can make one (an attempt at the same site is a dependency, so it has retired).
//   * the undo record of an accepted attempt a (the K neighbour fields before the move and the own field) stays in a's slot until the slot's
//     next user, attempt a + M, EVACUATES it into its wavefront's keep area — only for the lanes whose last accepted move still is a —
//     then marks the slot (ev[slot] = a + M) and only then overwrites it.  So the record of tl = a is in slot(a) iff ev[slot(a)] == a, else in
//     the keep area of the wavefront that runs attempt a + M: a reader takes the slot's data first and the mark second.  A keep entry is
//     written for a lane only while that lane's last accepted move is the evacuated one, i.e. while the lane's previous keep entry is dead.
// No wavefront ever waits for a later attempt, every wait is on the retired prefix, so the scheme cannot deadlock; everything a caller can see
// (fields, spins, undo records, move_last, energies, samples, accepted counts) is bit-identical to spf_sweep_kernel and the oracle
// (tests/test_gpu_spf_parity.py runs every case — two bonds to the same neighbour included — through both kernels).
//
// Ordering rests on three hardware facts, all within ONE compute unit (a workgroup never spans two; the kernel must not be built for
// threadgroup-split mode, where a workgroup's waves may sit on different compute units — hipcc's default is off and build.py passes no
// -mtgsplit): LDS performs one wavefront's operations in order; a wavefront's `s_waitcnt vmcnt(0)` returns when its stores have been
// performed at the compute unit's L1 / L2 path, which every other wavefront of the workgroup shares; relaxed LDS atomics separated by
// wavefront-scope fences are not reordered by the compiler.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "spf_kernels.hpp"

namespace rrrmc {

#if defined(SPF_TEAM_STAMPS) && !defined(SPF_TEAM_COUNT_ONLY)
#define SPF_STAMP(i) do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_last; st_last = t_; } while (0)
#else
#define SPF_STAMP(i) do { } while (0)
#endif
#ifndef SPF_TEAM_NAP
#define SPF_TEAM_NAP 1
#endif
#ifndef SPF_TEAM_BATCH
#define SPF_TEAM_BATCH 4
#endif
#ifndef SPF_TEAM_SCOPE
#define SPF_TEAM_SCOPE "workgroup"
#endif
constexpr int kSpfTeamWindow = 64;          // dependency window of spf_team_plan_kernel: >= the slots of every build (attempts in flight)

// The attempts of one launch, state-independent, one record of 2 + 3 K dwords per iteration `it` (record 0 and the records behind the last
// iteration are padding, so that an executing wavefront fetches the two records of a pair with one load):
//   [0] site   [1] dep   [2 .. 2 + K) the neighbours   [2 + K .. 2 + 3 K) their couplings (Float64, low word first)
// dep = the latest earlier iteration within the window whose closed neighbourhood meets this one's (0 = none); neighbourhoods meet <=> the
// sites are at distance <= 2.
__host__ __device__ constexpr int spf_plan_stride(int K) { return 2 + 3 * K; }

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
    int32_t dep = 0;
    for (int c = 1; c <= kSpfTeamWindow && c < t && dep == 0; ++c) {
        const int j = sites[t - 1 - c];
        bool hit = false;
        for (int a = 0; a <= K; ++a) hit |= mine[a] == j;
        for (int k = 0; k < K; ++k) {
            const int y = A[(size_t)j * K + k];
            for (int a = 0; a <= K; ++a) hit |= mine[a] == y;
        }
        if (hit) dep = (int32_t)(t - c);
    }
    o[0] = (uint32_t)i;
    o[1] = (uint32_t)dep;
    for (int k = 0; k < K; ++k) {
        o[2 + k] = (uint32_t)mine[1 + k];
        const unsigned long long jb = (unsigned long long)__double_as_longlong(J[(size_t)i * K + k]);
        o[2 + K + 2 * k] = (uint32_t)jb;
        o[2 + K + 2 * k + 1] = (uint32_t)(jb >> 32);
    }
}

struct SpfTeamParams {
    SpfParams S;
    const uint32_t* plan;       // [iters + 2][2 + 3 K]
    int32_t* status;            // one word per context, may be null: set to 1 by a workgroup whose wait ran into kSpfTeamSpinLimit (a protocol
                                // failure: the launch then runs to its end without waiting and its results are void)
};
constexpr int32_t kSpfTeamSpinLimit = 1 << 22;      // polls of one wait (each >= 64 cycles asleep): ~ 0.3 s, a thousand times the longest legitimate wait

// what the retiring wavefront reads of an attempt, one 16-byte LDS read per lane
struct __attribute__((aligned(16))) SpfVt {
    double v;           // the attempted site's own field before the move (lfields_last[i]), +0.0 for a lane that did not accept
    uint32_t tag;       // the attempt's launch-relative iteration, 0 for a lane that did not accept
    uint32_t pad;
};
typedef uint32_t spf_u32x4 __attribute__((ext_vector_type(4)));

// record areas: [0, M) the slots, M + x the KEEP of executing wavefront x, M + NX the undo records the launch starts with
__host__ __device__ constexpr int spf_team_areas(int NW, int D) { return (2 * D + 1) * (NW - 1) + 1; }
__host__ __device__ constexpr size_t spf_team_lds_bytes(int K, int NW, int D)
{
    return (sizeof(double) * (size_t)K * 64 + sizeof(SpfVt) * 64) * (size_t)spf_team_areas(NW, D) + sizeof(int32_t) * (size_t)(64 + 2 * (2 * D * (NW - 1)) + 4);      // tl, done, ev, prefix + abort flag
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

// grid W, block NW * 64, dynamic LDS spf_team_lds_bytes(K, NW, D).  D = pairs of slots per executing wavefront: how far the wavefronts
// may run ahead of the retired prefix (2 D (NW - 1) attempts)
template <int K, int NW, int D>
__global__ __launch_bounds__(NW * 64) void spf_team_kernel(SpfTeamParams TP)
{
    constexpr int NX = NW - 1;
    constexpr int M = 2 * NX * D;                     // slots = attempts in flight at most
    constexpr int AREAS = spf_team_areas(NW, D);
    static_assert(M <= kSpfTeamWindow && M < 64, "the dependency window covers the attempts in flight; one wavefront read brings all flags");
    static_assert(AREAS == M + NX + 1, "areas");
    const SpfParams& P = TP.S;
    extern __shared__ __attribute__((aligned(16))) unsigned char spf_team_lds[];
    typedef double nbr_t[K][64];
    typedef SpfVt vt_t[64];
    nbr_t* const rec = reinterpret_cast<nbr_t*>(spf_team_lds);                                    // [AREAS] the K neighbour fields before the move
    vt_t* const vt = reinterpret_cast<vt_t*>(spf_team_lds + sizeof(nbr_t) * AREAS);               // [AREAS] own field + tag
    int32_t* const tl = reinterpret_cast<int32_t*>(spf_team_lds + (sizeof(nbr_t) + sizeof(vt_t)) * AREAS);   // [64] last accepted attempt as retired (0: before the launch)
    int32_t* const done = tl + 64;                    // [M] iteration whose results the slot holds
    int32_t* const ev = done + M;                     // [M] iteration that has evacuated the slot's previous records and may be overwriting it
    int32_t* const shP = ev + M;                      // retired prefix
    int32_t* const abortf = shP + 1;                  // set when a wait ran into its limit: nobody waits any more

    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), w = blockIdx.x, N = P.N;
    const int r = w * 64 + lane;
    double* const lf = P.lf + (size_t)w * N * 64 + lane;
    double* const undo = P.undo + (size_t)w * (K + 1) * 64 + lane;
    unsigned long long* const sp = P.spins + (size_t)w * N;
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
        for (int k = 0; k < K; ++k) rec[M + NX][k][lane] = undo[(size_t)k * 64];
        vt[M + NX][lane].v = undo[(size_t)K * 64];
        tl[lane] = 0;
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
                if (P.Es) P.Es[(size_t)ns * P.Rpad + r] = E;
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
                __builtin_amdgcn_s_sleep(1);
                f = spf_lds_ld(fl);
                if (++idle > kSpfTeamSpinLimit || ((idle & 1023) == 0 && spf_lds_uniform(abortf))) {        // see wait_prefix
                    if (lane == 0) { spf_lds_st(abortf, 1); spf_lds_st(shP, n32); if (TP.status) *TP.status = 1; }
                    break;
                }
                continue;
            }
            idle = 0;
            const SpfVt* const q0 = &vt[s0][lane];
            // n attempts, straight-line: n reads, the next look at the flags, then per attempt one add and one max
#define SPF_RETIRE_CASE(n)                                                                                                      \
            case n: {                                                                                                           \
                spf_u32x4 q[n];                                                                                                 \
                _Pragma("unroll") for (int j = 0; j < n; ++j) q[j] = *reinterpret_cast<const spf_u32x4*>(q0 + j * 64); \
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                                          \
                f = spf_lds_ld(fl);                                                                                             \
                _Pragma("unroll") for (int j = 0; j < n; ++j) {                                                                 \
                    const double vj = __longlong_as_double((long long)(((unsigned long long)q[j].y << 32) | q[j].x));           \
                    E = __dadd_rn(E, -vj);                         /* dE = -lfields[i] (RRG.jl:619-625); - (+0.0) changes nothing */ \
                    tlr = (int32_t)q[j].z > tlr ? (int32_t)q[j].z : tlr;                                                        \
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
            spf_lds_st(tl + lane, tlr);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                // compiler barrier: LDS performs a wavefront's writes in order
            if (lane == 0) spf_lds_st(shP, it + ready - 1);
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
        P.E_cur[r] = E;
        P.move_last[r] = tlr ? P.sites[tlr - 1] : ml0;
        {
            // every executing wavefront has reported its last attempt: nothing moves any more
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            int area = M + NX;
            if (tlr) {
                const int sl = (tlr + c0) % M;
                area = spf_lds_ld(ev + sl) == tlr ? sl : M + owner_of(tlr + M);
            }
#pragma unroll
            for (int k = 0; k < K; ++k) undo[(size_t)k * 64] = rec[area][k][lane];
            undo[(size_t)K * 64] = vt[area][lane].v;
        }
        return;
    }

    // ---- execute ----
    const uint32_t replica = P.replica0 + (uint32_t)r;
    const int x = wv;
    constexpr int S = spf_plan_stride(K);
    static_assert(2 * S <= 64, "the two records of a pair are fetched by one wavefront load");
    // the records of the wavefront's next pair travel one pair ahead of their use
    auto fetch_pair = [&](uint64_t h) -> uint32_t {
        const int64_t it_even = (int64_t)(2 * (hb + h) - g0);
        const int64_t rec0 = it_even > iters ? iters : it_even;     // it_even >= 0; clamped: a pair beyond the launch is fetched and never used
        return TP.plan[(size_t)rec0 * S + (lane < 2 * S ? lane : 0)];
    };
    auto site_of_tl = [&](int32_t t) -> int32_t {                   // move_last of a lane whose last accepted attempt is t
        const int32_t s = P.sites[(t > 0 ? t : 1) - 1];
        return t > 0 ? s : ml0;
    };
    uint32_t pr_next = fetch_pair((uint64_t)x), pr_next2 = fetch_pair((uint64_t)x + NX);
    int32_t nacc = 0;
#ifdef SPF_TEAM_STAMPS
    uint64_t st_acc[6] = {0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime(), st_n = 0;
#endif
    for (uint64_t h = (uint64_t)x;; h += NX) {
        const uint64_t blk_id = hb + h;
        const int64_t it_even = (int64_t)(2 * blk_id - g0);         // iteration whose stream index g is even
        if (it_even > iters) break;
        const uint32_t pr = pr_next;
        pr_next = pr_next2;
        pr_next2 = fetch_pair(h + 2 * NX);
        const Philox4 blk = philox4x32_10((uint32_t)blk_id, (uint32_t)(blk_id >> 32), replica, TAG_ACCEPT_F64, P.k0, P.k1);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int64_t it = it_even + e;
            if (it < 1 || it > iters) continue;
            const int s = (int)((2 * h + (uint64_t)e) % (uint64_t)M);
            const int o = e * S;
            const int i = __builtin_amdgcn_readlane((int)pr, o + 0);
            const int32_t dep = __builtin_amdgcn_readlane((int)pr, o + 1);
            int y[K];
            double J[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                y[k] = __builtin_amdgcn_readlane((int)pr, o + 2 + k);
                const uint32_t jl = (uint32_t)__builtin_amdgcn_readlane((int)pr, o + 2 + K + 2 * k), jh = (uint32_t)__builtin_amdgcn_readlane((int)pr, o + 2 + K + 2 * k + 1);
                J[k] = __longlong_as_double((long long)(((unsigned long long)jh << 32) | jl));
            }
            const uint64_t u = e ? (((uint64_t)blk.w[2] << 32) | blk.w[3]) : (((uint64_t)blk.w[0] << 32) | blk.w[1]);
            const double U = (double)(u >> 11) * 0x1.0p-53;
            const int64_t lo = it - M;                              // the slot's previous use; everything older than the dependency window
            const int32_t need = (int32_t)(dep > lo ? dep : lo);
            SPF_STAMP(0);                                           // state-independent preparation
            wait_prefix(need, SPF_TEAM_NAP);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, SPF_TEAM_SCOPE);
            SPF_STAMP(1);                                           // waiting for the dependency / the slot

#ifdef SPF_TEAM_EXP_NOMEM
            // timing experiment (tools/ubench/spf_team_bench.hip; wrong results): no global memory traffic at all — what the protocol alone costs
            const double lfi = 0.25 * (double)((lane + i) & 7) - 1.0;
            const unsigned long long wi = 0x5555aaaa5555aaaaull;
            double nf[K];
            unsigned long long nw[K];
#pragma unroll
            for (int k = 0; k < K; ++k) { nf[k] = 0.5 * k; nw[k] = 0x3333cccc3333ccccull + (unsigned long long)y[k]; }
#else
            const double lfi = lf[(size_t)i * 64];
            const unsigned long long wi = spf_load_spins(sp + i);
            double nf[K];
            unsigned long long nw[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                nf[k] = lf[(size_t)y[k] * 64];
                nw[k] = spf_load_spins(sp + y[k]);
            }
#endif
            int32_t tlv = spf_lds_ld(tl + lane);                    // as retired so far (>= need): later attempts in flight are at other sites
            int32_t mls = site_of_tl(tlv);

            const double dE = -lfi;                                 // delta_energy: RRG.jl:619-625
            const double xx = __dmul_rn(-P.beta, dE);
            const bool acc = xx >= 0.0 || U < det_exp(xx);          // accept: RRRMC.jl:39
            const unsigned long long amask = __builtin_amdgcn_ballot_w64(acc);
            nacc += acc ? 1 : 0;
            SPF_STAMP(2);                                           // loads + decision

            bool fast = false;
            double sv[K];
#pragma unroll
            for (int k = 0; k < K; ++k) sv[k] = 0.0;
            if (__builtin_amdgcn_ballot_w64(acc && mls == i) != 0ull) {
                // some accepting replica's last retired accepted move is at this site: exact only once everything before this attempt has retired
                // (the retiring wavefront then rests until this attempt reports, so tl and the records of its lanes are stable)
                wait_prefix((int32_t)(it - 1), 1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, SPF_TEAM_SCOPE);
                tlv = spf_lds_ld(tl + lane);
                mls = site_of_tl(tlv);
                fast = acc && mls == i;
                if (fast) {
                    // the record of attempt tlv: the slot's data first, the slot's mark second (see the head of the file)
                    if (tlv > 0) {
                        const int sl = (tlv + c0) % M;
                        double tmp[K];
#pragma unroll
                        for (int k = 0; k < K; ++k) tmp[k] = __hip_atomic_load(&rec[sl][k][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        const bool in_slot = spf_lds_ld(ev + sl) == tlv;
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        const int keep = M + owner_of(tlv + M);
#pragma unroll
                        for (int k = 0; k < K; ++k) sv[k] = in_slot ? tmp[k] : __hip_atomic_load(&rec[in_slot ? sl : keep][k][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else {
#pragma unroll
                        for (int k = 0; k < K; ++k) sv[k] = rec[M + NX][k][lane];
                    }
                }
            }
            const bool slow = acc && !fast;

            // EVACUATION: records of this slot's previous use (attempt it - M) that are still some replica's last accepted move go to the
            // wavefront's own keep before the slot is marked and written.  A stale look at tl (the lane has accepted since) costs a wasted copy.
            if (it > (int64_t)M) {
                const bool live = tlv == (int32_t)(it - M);
                if (__builtin_amdgcn_ballot_w64(live) != 0ull) {
                    if (live) {
#pragma unroll
                        for (int k = 0; k < K; ++k) rec[M + x][k][lane] = rec[s][k][lane];
                        vt[M + x][lane].v = vt[s][lane].v;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (lane == 0) spf_lds_st(ev + s, (int32_t)it);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

#ifdef SPF_TEAM_EXP_NOMEM
            if (amask == 0xdeadbeefull) {
#else
            if (amask != 0ull) {
#endif
                // update_cache! (RRG.jl:576-617, EA.jl:613-653).  Full 512-byte lines: lanes that do not accept write back what they read
                const uint32_t snew = (uint32_t)((wi >> lane) & 1ull) ^ 1u;
                double vrun = 0.0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    // two bonds to the same neighbour (GraphEANormal with L = 2; rows are sorted): the second one continues from the first one's
                    // result (EA.jl:626-640 walks all entries), and its store — same wavefront, same address, in order — is the one that stays
                    const bool rep = k > 0 && y[k] == y[k - 1];                  // wave-uniform
                    const uint32_t sbit = (uint32_t)((nw[k] >> lane) & 1ull);
                    const double c = (snew ^ sbit) ? -4.0 : 4.0;                 // 4 * sigma_xy with the NEW s_x
                    const double v = __dadd_rn(rep ? vrun : nf[k], -__dmul_rn(c, J[k]));
                    vrun = v;
                    lf[(size_t)y[k] * 64] = fast ? sv[k] : (slow ? v : nf[k]);
                }
                lf[(size_t)i * 64] = acc ? -lfi : lfi;
                if (lane == 0) __hip_atomic_store(sp + i, wi ^ amask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
#pragma unroll
            for (int k = 0; k < K; ++k) rec[s][k][lane] = nf[k];                // lfields_last[y] = lfields[y] (slow) / the swap (fast)
            vt[s][lane].v = acc ? lfi : 0.0;
            vt[s][lane].tag = acc ? (uint32_t)it : 0u;
            SPF_STAMP(3);                                           // undo test, update, stores issued
            // the field and spin stores must have been PERFORMED before the attempt is reported: a workgroup-scope release does not wait for
            // vector stores (it relies on one compute unit issuing them in order; loads of another SIMD were seen to overtake them)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, SPF_TEAM_SCOPE);
            if (lane == 0) spf_lds_st(done + s, (int32_t)it);
            SPF_STAMP(4);                                           // stores performed, attempt reported
#ifdef SPF_TEAM_STAMPS
            ++st_n;
#endif
        }
    }
    // accepted moves: a count, whatever the order (the host clears acc_cur at the start of the call)
    atomicAdd(reinterpret_cast<unsigned long long*>(P.acc_cur) + r, (unsigned long long)nacc);
#ifdef SPF_TEAM_STAMPS
    if (w == 0 && lane == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(P.Es) + (size_t)x * 8;      // harness only: the sample buffer is not compared in this build
        for (int q = 0; q < 5; ++q) o[q] = st_acc[q];
        o[5] = st_n;
    }
#endif
}

}  // namespace rrrmc
