// gfx950 kernels for random-site standardMC (src/RRRMC.jl:81-127) on +-J sparse graphs that do NOT fit the LDS-resident
// sweep_kernel (N beyond 32 767, or wherever the planner's LDS buffers do not fit: GraphEA(64, 3) has N = 262 144).  Same chain, same streams, same planner idea as
// sparse_kernels.hpp — the site sequence is cut into chunks, every chunk into dependency levels whose attempts commute —
// but nothing here is sized by N:
//   plan_big_kernel<K>   one workgroup per chunk (<= 4096 attempts): the attempts are SORTED by (site, index) in LDS (bitonic), an
//                        attempt's predecessors — the latest earlier attempt at every site of its closed neighbourhood — are
//                        found by binary search, levels by relaxation as in plan_kernel.
//   big_sweep_kernel<K>  one workgroup per group of 32 bit-sliced replicas, spins in HBM/L2 ([G][N] words, the context's native
//                        layout); a level's attempts are spread over the 1024 threads, one __syncthreads per level.  A thread
//                        does the whole attempt: gathers the K+1 words, counts the unsatisfied bonds bit-sliced, evaluates the
//                        ACCEPT planes lazily for the replicas that need a random number, flips, and adds the accepted moves to
//                        the per-replica energy / accepted counters in LDS.
// Bit-identical to the oracle (and to sweep_kernel where both apply); slower than the LDS kernel by design — the colour-parallel
// sweeps remain the fast path for lattices — but it makes the reference's own dynamics available at any size.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "philox.hpp"
#include "sparse_kernels.hpp"

namespace rrrmc {

constexpr int kBigChunk = 4096;            // attempts per chunk: the index inside the chunk takes 12 bits of a slot word
constexpr int kBigSiteBits = 20;           // sites take the other 20: N <= 2^20
constexpr int kBigThreads = 1024;

inline size_t plan_big_lds_bytes(int K)
{
    return (size_t)kBigChunk * 4 * 2            // sort keys, sites
           + (size_t)kBigChunk * 2              // levels
           + (size_t)kBigChunk * (K + 1) * 2    // predecessors
           + (size_t)(kBigChunk + 2) * 2 * 3;   // level counters
}

//   slots[slot_base + p] = site | (t << 20)   attempts sorted by dependency level (t = index in chunk)
//   vecs [slot_base + l] = first slot of level l + 1 (l = 0 .. nvec-1); chunks[c].nvec = number of levels
template <int K>
__global__ __launch_bounds__(kPlanThreads) void plan_big_kernel(ChunkDesc* __restrict__ chunks, uint32_t* __restrict__ slots,
                                                                uint32_t* __restrict__ vecs, const int32_t* __restrict__ A,
                                                                int N, uint32_t k0, uint32_t k1, uint64_t gbase)
{
    extern __shared__ uint32_t pb_lds[];
    uint32_t* s_key = pb_lds;                                            // [kBigChunk]   site << 12 | t, sorted
    uint32_t* s_site = s_key + kBigChunk;                                // [kBigChunk]
    uint16_t* s_lvl = reinterpret_cast<uint16_t*>(s_site + kBigChunk);   // [kBigChunk]
    uint16_t* s_pred = s_lvl + kBigChunk;                                // [kBigChunk * (K + 1)]
    uint16_t* s_cnt = s_pred + (size_t)kBigChunk * (K + 1);              // [kBigChunk + 2]  slots per level (1-based)
    uint16_t* s_start = s_cnt + kBigChunk + 2;                           // [kBigChunk + 2]
    uint16_t* s_cur = s_start + kBigChunk + 2;                           // [kBigChunk + 2]
    __shared__ uint32_t s_maxlvl, s_changed;

    const ChunkDesc cd = chunks[blockIdx.x];
    const int count = (int)cd.count;
    const int tid = threadIdx.x;
    int cp = 64;
    while (cp < count) cp <<= 1;                                         // sort length: a power of two, padded with ~0

    for (int t = tid; t < cp; t += kPlanThreads) {
        uint32_t key = 0xffffffffu;
        if (t < count) {
            const uint32_t site = site_of(k0, k1, gbase + cd.g0 + (uint64_t)t, (uint32_t)N);
            s_site[t] = site;
            s_lvl[t] = 1;
            key = (site << 12) | (uint32_t)t;
        }
        s_key[t] = key;
    }
    __syncthreads();
    for (int k = 2; k <= cp; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < cp; i += kPlanThreads) {
                const int l = i ^ j;
                if (l > i) {
                    const uint32_t a = s_key[i], b = s_key[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { s_key[i] = b; s_key[l] = a; }
                }
            }
            __syncthreads();
        }
    // predecessors: for every site x of the closed neighbourhood, the latest earlier attempt at x = the key just below (x, t)
    for (int q = tid; q < count * (K + 1); q += kPlanThreads) {
        const int t = q / (K + 1), j = q - t * (K + 1);
        const uint32_t x = j == 0 ? s_site[t] : (uint32_t)A[(size_t)s_site[t] * K + j - 1];
        const uint32_t want = (x << 12) | (uint32_t)t;
        int lo = 0, hi = count;                                          // first index with key >= want
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_key[mid] < want) lo = mid + 1; else hi = mid;
        }
        uint32_t best = kNoPred;
        if (lo > 0 && (s_key[lo - 1] >> 12) == x) best = s_key[lo - 1] & 0xfffu;
        s_pred[q] = (uint16_t)best;
    }
    for (int l = tid; l < count + 2; l += kPlanThreads) { s_cnt[l] = 0; s_cur[l] = 0; }
    __syncthreads();
    // longest-path depths by relaxation (levels only grow; the fixed point is the sequential rule's result)
    for (;;) {
        if (tid == 0) s_changed = 0u;
        __syncthreads();
        bool ch = false;
        for (int t = tid; t < count; t += kPlanThreads) {
            uint32_t l = 0u;
#pragma unroll
            for (int j = 0; j <= K; ++j) {
                const uint32_t pr = s_pred[t * (K + 1) + j];
                const uint32_t w = pr == kNoPred ? 0u : (uint32_t)s_lvl[pr];
                l = w > l ? w : l;
            }
            l += 1u;
            if (l != (uint32_t)s_lvl[t]) { s_lvl[t] = (uint16_t)l; ch = true; }
        }
        if (ch) s_changed = 1u;
        __syncthreads();
        if (s_changed == 0u) break;
        __syncthreads();
    }
    if (tid == 0) s_maxlvl = 0u;
    __syncthreads();
    // 16-bit counters: LDS atomics work on 32-bit words, so count through the aligned word that holds the half
    auto add16 = [](uint16_t* base, uint32_t idx) -> uint32_t {
        uint32_t* w = reinterpret_cast<uint32_t*>(base) + (idx >> 1);
        const uint32_t sh = (idx & 1u) * 16u;
        return (atomicAdd(w, 1u << sh) >> sh) & 0xffffu;
    };
    for (int t = tid; t < count; t += kPlanThreads) { add16(s_cnt, s_lvl[t]); atomicMax(&s_maxlvl, (uint32_t)s_lvl[t]); }
    __syncthreads();
    if (tid == 0) {
        const uint32_t maxlvl = s_maxlvl;
        uint32_t pos = 0;
        for (uint32_t l = 1; l <= maxlvl; ++l) {
            s_start[l] = (uint16_t)pos;
            vecs[cd.slot_base + l - 1] = pos;
            pos += s_cnt[l];
        }
        chunks[blockIdx.x].nvec = maxlvl;
    }
    __syncthreads();
    for (int t = tid; t < count; t += kPlanThreads) {
        const uint32_t l = s_lvl[t];
        const uint32_t pos = (uint32_t)s_start[l] + add16(s_cur, l);
        slots[cd.slot_base + pos] = s_site[t] | ((uint32_t)t << kBigSiteBits);
    }
}

struct BigSweepParams {
    uint32_t* spins;          // [G][N]   bit-sliced configuration
    const int32_t* A;         // [N][K]
    const int8_t* J;          // [N][K]
    const ChunkDesc* chunks;
    const uint32_t* slots;
    const uint32_t* vecs;
    int32_t* Es;              // [nsamples][Rpad]
    int32_t* E_cur;           // [Rpad]
    int64_t* acc_cur;         // [Rpad]
    uint32_t taum[64 * 4];    // threshold bit planes, as SweepParams::taum
    uint32_t always_mask;
    uint32_t k0, k1, group0;
    int64_t sample0;
    uint64_t gbase;           // iterations done before this sampling call (ChunkDesc::g0 is relative to it)
    int N, Rpad, nchunks;
};

template <int K>
__global__ __launch_bounds__(kBigThreads) void big_sweep_kernel(BigSweepParams P)
{
    constexpr int NT = (K + 1) / 2;          // classes with dE > 0: n = 0 .. NT-1 unsatisfied bonds
    __shared__ int32_t s_E[32], s_A[32];
    const int tid = threadIdx.x;
    const uint32_t group = P.group0 + blockIdx.x;
    uint32_t* gsp = P.spins + (size_t)blockIdx.x * P.N;
    if (tid < 32) { s_E[tid] = P.E_cur[blockIdx.x * 32 + tid]; s_A[tid] = 0; }
    int64_t ns = P.sample0;
    __syncthreads();
    for (int c = 0; c < P.nchunks; ++c) {
        const ChunkDesc cd = P.chunks[c];
        if (cd.flags & kChunkSampleBefore) {                 // sample BEFORE the move of iteration k*step (RRRMC.jl:104-108)
            if (tid < 32) P.Es[ns * P.Rpad + blockIdx.x * 32 + tid] = s_E[tid];
            ns += 1;
        }
        for (uint32_t l = 0; l < cd.nvec; ++l) {
            const uint32_t start = P.vecs[cd.slot_base + l];
            const uint32_t end = l + 1 < cd.nvec ? P.vecs[cd.slot_base + l + 1] : cd.count;
            for (uint32_t p = start + (uint32_t)tid; p < end; p += kBigThreads) {
                const uint32_t slot = P.slots[cd.slot_base + p];
                const uint32_t site = slot & ((1u << kBigSiteBits) - 1u);
                const uint64_t g = P.gbase + cd.g0 + (uint64_t)(slot >> kBigSiteBits);
                const uint32_t s = gsp[site];
                // n = number of unsatisfied bonds, bit-sliced (three planes cover K <= 7)
                uint32_t n0 = 0u, n1 = 0u, n2 = 0u;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const uint32_t gk = gsp[P.A[(size_t)site * K + k]] ^ (P.J[(size_t)site * K + k] < 0 ? 0xffffffffu : 0u);
                    const uint32_t u = s ^ gk;               // bond k unsatisfied
                    const uint32_t c0 = n0 & u;
                    n0 ^= u;
                    const uint32_t c1 = n1 & c0;
                    n1 ^= c0;
                    n2 ^= c1;
                }
                // classes n = 0..NT-1 have dE = 2(K - 2n) > 0 and need u < T_n; every other class is accepted (RRRMC.jl:39)
                uint32_t need[NT], lt[NT], eq[NT], any = 0u;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    uint32_t e = (n & 1) ? n0 : ~n0;
                    e &= (n & 2) ? n1 : ~n1;
                    e &= (n & 4) ? n2 : ~n2;
                    need[n] = e;
                    const bool always = (P.always_mask >> n) & 1u;
                    lt[n] = always ? 0xffffffffu : 0u;
                    eq[n] = always ? 0u : e;                 // only the replicas of this class consume planes: lazy evaluation of
                    any |= eq[n];                            // the counter-based ACCEPT stream gives the same bits wherever it stops
                }
                for (uint32_t pb = 0; pb < 16u && any; ++pb) {
                    const Philox4 o = accept_planes(P.k0, P.k1, g, group, pb);
                    any = 0u;
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const uint32_t tm = P.taum[(pb * 4 + j) * 4 + n];
                            const uint32_t e2 = eq[n] & ~(o.w[j] ^ tm);
                            lt[n] |= (eq[n] ^ e2) & tm;
                            eq[n] = e2;
                        }
                        any |= eq[n];
                    }
                }
                uint32_t rej = 0u;
#pragma unroll
                for (int n = 0; n < NT; ++n) rej |= need[n] & ~lt[n];
                const uint32_t acc = ~rej;
                gsp[site] = s ^ acc;                         // spinflip! + update_cache! (Interface.jl:89-92, RRG.jl:191-234)
                for (uint32_t m = acc; m; m &= m - 1u) {     // bookkeeping of the accepted moves: E += dE, accepted += 1
                    const int r = __builtin_ctz(m);
                    const int nr = (int)((n0 >> r) & 1u) + 2 * (int)((n1 >> r) & 1u) + 4 * (int)((n2 >> r) & 1u);
                    atomicAdd(&s_E[r], 2 * (K - 2 * nr));
                    atomicAdd(&s_A[r], 1);
                }
            }
            __syncthreads();                                 // the next level reads what this one wrote
        }
    }
    if (tid < 32) {
        P.E_cur[blockIdx.x * 32 + tid] = s_E[tid];
        P.acc_cur[blockIdx.x * 32 + tid] += (int64_t)s_A[tid];
    }
}

}  // namespace rrrmc
