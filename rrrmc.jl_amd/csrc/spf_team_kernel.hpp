// gfx950 kernel for standardMC (src/RRRMC.jl:81-127) on the Float64-coupling sparse models (GraphRRGNormal / GraphEANormal:
// src/graphs/RRG.jl:503-627, src/graphs/EA.jl:534-680), round 4: a TEAM of wavefronts per group of 64 replicas.
//
// spf_sweep_kernel (spf_kernels.hpp) walks the chain with ONE wavefront per group of 64 replicas: one attempt after the other, ~ 1 us each,
// whatever the machine has idle (8192 replicas = 128 wavefronts on 1024 SIMDs).  The attempts of a chain are not all dependent on each other:
// the SITE stream is known in advance, an attempt at site i reads lfields[i] and the spins of i and its neighbours and writes lfields of i and
// its neighbours (update_cache!, RRG.jl:576-617) — two attempts whose closed neighbourhoods do not meet commute exactly (no field receives
// an addition from both, so no rounding order changes).  This kernel runs such attempts side by side:
//
//   * NW - 1 executing wavefronts per group take the attempts of the chain in pairs (one Philox block serves iterations 2h, 2h + 1),
//     pair h on wavefront h mod (NW - 1).  Before it touches memory, the wavefront of attempt t waits until every attempt up to
//     need(t) = max(dep(t), t - 2 (NW - 1)) has RETIRED, dep(t) = the latest earlier attempt within that window whose closed neighbourhood meets
//     t's (spf_team_plan_kernel: state-independent, once per launch for all groups).  Then it runs the reference's attempt — accept (RRRMC.jl:39),
//     the K + 1 field updates, the spin word — on the group's [N][64] arrays in HBM / L2, exactly as spf_sweep_kernel does.
//   * one RETIRING wavefront keeps what the reference's loop keeps in chain order: the tracked energy (E += dE is a Float64 running sum:
//     its order is the chain's), the accepted count, the samples (RRRMC.jl:104-108), and per replica the site of the last accepted move
//     (move_last) and WHERE its undo record is.  It consumes the attempts strictly in order, one slot per attempt: slot (wavefront, parity)
//     carries the K neighbour fields before the move and the attempted site's own field (NaN for a lane that did not accept).  This
//     wavefront is the chain's serial part: it reads one value and writes two words per attempt and accepting replica.
//   * the undo path (RRG.jl:583-593: a move of the spin that was also the replica's last accepted move swaps lfields <-> lfields_last) needs
//     move_last as of t - 1: a wavefront whose accepting lanes include a replica with move_last == site (as retired so far) waits until
//     t - 1 has retired — the retiring wavefront then rests until t itself reports — and reads move_last again; without such a lane no
//     in-flight attempt can make one (an attempt at the same site is a dependency).  The record is read from the area the retiring
//     wavefront names.
//   * a slot is reused two pairs later; before that its owner EVACUATES the records that are still some replica's last accepted move into
//     its own keep area and swings the replica's source over by compare-and-swap.
// No wavefront ever waits for a later attempt, every wait is on the retired prefix, so the scheme cannot deadlock; everything a caller can see
// (fields, spins, undo records, move_last, energies, samples, accepted counts) is bit-identical to spf_sweep_kernel and the oracle
// (tests/test_gpu_spf_parity.py runs every case — two bonds to the same neighbour included — through both kernels).
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
#define SPF_TEAM_NAP 2
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
};

__host__ __device__ constexpr size_t spf_team_lds_bytes(int K, int NW, int D)
{
    return sizeof(double) * (size_t)((2 * D + 1) * (NW - 1) + 1) * (K + 1) * 64 + sizeof(int32_t) * (64 + 64 + 4 * D * (NW - 1) + 4);
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
    static_assert(M <= kSpfTeamWindow && M < 64, "the dependency window covers the attempts in flight; one wavefront read brings all flags");
    const SpfParams& P = TP.S;
    extern __shared__ __attribute__((aligned(16))) unsigned char spf_team_lds[];
    typedef double slot_t[K + 1][64];
    // record areas: [0, M) the slots (attempt with stream index g: slot (g - 2 hb) mod M), M + x the KEEP of wavefront x, M + NX the launch's initial records
    slot_t* const rec = reinterpret_cast<slot_t*>(spf_team_lds);
    int32_t* const mlr = reinterpret_cast<int32_t*>(spf_team_lds + sizeof(slot_t) * (M + NX + 1));   // [64] move_last as retired
    int32_t* const rsrc = mlr + 64;                   // [64] the area that holds the undo record of the replica's last retired accepted move
    int32_t* const done = rsrc + 64;                  // [M] iteration whose results the slot holds
    int32_t* const ssite = done + M;                  // [M] its site
    int32_t* const shP = ssite + M;                   // retired prefix

    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), w = blockIdx.x, N = P.N;
    const int r = w * 64 + lane;
    double* const lf = P.lf + (size_t)w * N * 64 + lane;
    double* const undo = P.undo + (size_t)w * (K + 1) * 64 + lane;
    unsigned long long* const sp = P.spins + (size_t)w * N;
    const int64_t iters = P.iters;
    const uint64_t g0 = P.g0, hb = (g0 + 1) >> 1;        // first pair of the launch

    if (wv == NW - 1) {
#pragma unroll
        for (int k = 0; k <= K; ++k) rec[M + NX][k][lane] = undo[(size_t)k * 64];
        mlr[lane] = P.move_last[r];
        rsrc[lane] = M + NX;
        if (lane < M) done[lane] = 0;
        if (lane == 0) *shP = 0;
    }
    __syncthreads();

    if (wv == NW - 1) {
        // ---- retire: the chain's order.  Only LDS traffic (in order per wavefront): compiler barriers, no waits ----
#ifndef SPF_TEAM_NOPRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        double E = P.E_cur[r];
        int32_t nacc = 0;
        int64_t ns = P.sample0;
        // the launch-relative iteration of the next sample (RRRMC.jl:104-108: before the move of every step-th iteration of the call)
        const int64_t until = P.step - (P.it_off % P.step);
        const int32_t step32 = (int32_t)(P.step > (int64_t)1 << 30 ? (int64_t)1 << 30 : P.step);
        int32_t samp = (int32_t)(until > (int64_t)1 << 30 ? (int64_t)1 << 30 : until);
        int32_t ml = P.move_last[r];
        constexpr int B = SPF_TEAM_BATCH;                          // attempts retired per look at the flags
#ifdef SPF_TEAM_STAMPS
        unsigned long long rt_rounds = 0, rt_idle = 0, rt_acc[4] = {0, 0, 0, 0}; const uint64_t rt_t0 = __builtin_amdgcn_s_memtime(); uint64_t rt_last = rt_t0;
#endif
        const int32_t n32 = (int32_t)iters;
        int s0 = (int)((g0 + 1) & 1u);                             // slot of iteration 1; the slots follow each other modulo 2 NX
        int32_t it = 1;
        // one read brings the flags of all slots; then the data of the attempts that have reported, behind their flags (LDS serves a
        // wavefront's reads in order, so data read after a flag that says `it` is that attempt's); the next look at the flags is already
        // under way while the batch is worked off
        int32_t f = spf_lds_ld(done + (lane < M ? lane : 0));
        while (it <= n32) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");        // compiler barrier: the data reads stay behind the flag read
            // which slots hold the attempt they are due to hold: one compare over the lanes, then the run of set bits from s0 (with wrap-around)
            int ready, sj[B];
            {
                int d = lane - s0;
                d += d < 0 ? M : 0;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(lane < M && f == it + d && it + d <= n32);
                const unsigned long long rot = ((m >> s0) | (m << (M - s0))) & ((1ull << M) - 1ull);      // bit j: slot (s0 + j) mod M
                ready = __builtin_ctzll(~rot);
                ready = ready > B ? B : ready;
#pragma unroll
                for (int j = 0; j < B; ++j) sj[j] = s0 + j >= M ? s0 + j - M : s0 + j;
            }
#ifdef SPF_TEAM_STAMPS
            ++rt_rounds; if (ready == 0) ++rt_idle;
#endif
            if (ready == 0) {
                __builtin_amdgcn_s_sleep(1);
                f = spf_lds_ld(done + (lane < M ? lane : 0));
                continue;
            }
            // straight-line from here: taken branches are what this wavefront cannot afford.  Slots that have not reported are read all the
            // same and turned into "nobody accepted"
            double v[B];
            int site[B];
#pragma unroll
            for (int j = 0; j < B; ++j) {
                v[j] = __hip_atomic_load(&rec[sj[j]][K][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                site[j] = spf_lds_ld(ssite + sj[j]);               // left in a vector register: making it scalar here would wait for every read in turn
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            f = spf_lds_ld(done + (lane < M ? lane : 0));
#pragma unroll
            for (int j = 0; j < B; ++j) {
                if (j < ready && it + j == samp) {                 // sample before the move
                    if (P.Es) P.Es[(size_t)ns * P.Rpad + r] = E;
                    ++ns;
                    samp += step32;
                }
                const double vj = j < ready ? v[j] : __builtin_nan("");
                if (vj == vj) {                                    // NaN: the lane did not accept
                    E = __dadd_rn(E, -vj);                         // dE = -lfields[i] (RRG.jl:619-625)
                    nacc += 1;
                    ml = site[j];
                    spf_lds_st(mlr + lane, site[j]);
                    spf_lds_st(rsrc + lane, sj[j]);                // the slot keeps the record until its owner moves it to its keep
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                // compiler barrier: LDS performs a wavefront's writes in order
            if (lane == 0) spf_lds_st(shP, it + ready - 1);
            it += ready;
            s0 = s0 + ready >= M ? s0 + ready - M : s0 + ready;
#if defined(SPF_TEAM_STAMPS) && !defined(SPF_TEAM_COUNT_ONLY)
            { const uint64_t t_ = __builtin_amdgcn_s_memtime(); rt_acc[3] += t_ - rt_last; rt_last = t_; }
#endif
        }
#ifdef SPF_TEAM_STAMPS
        if (w == 0 && lane == 0) {
            unsigned long long* o = reinterpret_cast<unsigned long long*>(P.Es) + (size_t)NX * 8;
            o[0] = rt_rounds; o[1] = rt_idle; o[2] = __builtin_amdgcn_s_memtime() - rt_t0; o[3] = rt_acc[0]; o[4] = rt_acc[1]; o[5] = rt_acc[2]; o[6] = rt_acc[3];
        }
#endif
        P.E_cur[r] = E;
        P.acc_cur[r] = P.acc_cur[r] + nacc;
        P.move_last[r] = ml;
        {
            const int32_t src = spf_lds_ld(rsrc + lane);           // every executing wavefront has reported its last attempt: nothing moves any more
#pragma unroll
            for (int k = 0; k <= K; ++k) undo[(size_t)k * 64] = rec[src][k][lane];
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
    uint32_t pr_next = fetch_pair((uint64_t)x), pr_next2 = fetch_pair((uint64_t)x + NX);
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
#ifdef SPF_TEAM_WARM
        // experiment (measured: no gain, 4.00 against 4.11e10 attempts/s): warm the L2 with the field lines the wavefront's NEXT pair will ask
        // for once its turn has come; the values are thrown away (other attempts may still change them)
        {
            double warm = 0.0;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int i = __builtin_amdgcn_readlane((int)pr_next, e * S);
                warm += lf[(size_t)i * 64];
#pragma unroll
                for (int k = 0; k < K; ++k) warm += lf[(size_t)__builtin_amdgcn_readlane((int)pr_next, e * S + 2 + k) * 64];
            }
            asm volatile("" :: "v"(warm));
        }
#endif
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
            while (spf_lds_uniform(shP) < need) __builtin_amdgcn_s_sleep(SPF_TEAM_NAP);
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
            int32_t ml = spf_lds_ld(mlr + lane);
            const int32_t src0 = spf_lds_ld(rsrc + lane);          // may be stale by the time it is used: see the evacuation below

            const double dE = -lfi;                                 // delta_energy: RRG.jl:619-625
            const double xx = __dmul_rn(-P.beta, dE);
            const bool acc = xx >= 0.0 || U < det_exp(xx);          // accept: RRRMC.jl:39
            const unsigned long long amask = __builtin_amdgcn_ballot_w64(acc);
            SPF_STAMP(2);                                           // loads + decision

            bool fast = false;
            double sv[K];
#pragma unroll
            for (int k = 0; k < K; ++k) sv[k] = 0.0;
            if (__builtin_amdgcn_ballot_w64(acc && ml == i) != 0ull) {
                // some accepting replica's last retired accepted move is this site: exact only once everything before this attempt has retired
                // (the retiring wavefront then rests until this attempt reports, so move_last and the records are stable)
                while (spf_lds_uniform(shP) < (int32_t)(it - 1)) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, SPF_TEAM_SCOPE);
                ml = spf_lds_ld(mlr + lane);
                fast = acc && ml == i;
                // the record: from the area the retiring wavefront named, checked against its owner moving it meanwhile (copy, then swing
                // the source, then overwrite: a source that reads the same before and after the data is the data's)
                bool pending = fast;
                while (__builtin_amdgcn_ballot_w64(pending) != 0ull) {
                    const int32_t c1 = spf_lds_ld(rsrc + lane);
                    double tmp[K];
#pragma unroll
                    for (int k = 0; k < K; ++k) tmp[k] = __hip_atomic_load(&rec[c1][k][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const int32_t c2 = spf_lds_ld(rsrc + lane);
                    if (pending && c2 == c1) {
#pragma unroll
                        for (int k = 0; k < K; ++k) sv[k] = tmp[k];
                        pending = false;
                    }
                }
            }
            const bool slow = acc && !fast;

            // EVACUATION: records of this slot's previous use that are still some replica's last accepted move go to the wavefront's own keep
            // before the slot is written.  The keep entry of a replica is only ever written while the replica's source is this wavefront's
            // SLOT, i.e. while the keep entry is dead, so a stale look at the source costs a wasted copy and a failed swing, never a record.
            {
                const bool live = src0 == s;
                if (__builtin_amdgcn_ballot_w64(live) != 0ull) {
                    if (live) {
#pragma unroll
                        for (int k = 0; k <= K; ++k) rec[M + x][k][lane] = rec[s][k][lane];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    if (live) {
                        int32_t expect = s;
                        __hip_atomic_compare_exchange_strong(rsrc + lane, &expect, M + x, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                }
            }

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
            rec[s][K][lane] = acc ? lfi : __builtin_nan("");
            if (lane == 0) spf_lds_st(ssite + s, i);
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
#ifdef SPF_TEAM_STAMPS
    if (w == 0 && lane == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(P.Es) + (size_t)x * 8;      // harness only: the sample buffer is not compared in this build
        for (int q = 0; q < 5; ++q) o[q] = st_acc[q];
        o[5] = st_n;
    }
#endif
}

}  // namespace rrrmc
