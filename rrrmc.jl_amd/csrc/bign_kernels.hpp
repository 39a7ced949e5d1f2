// gfx950 kernels for random-site standardMC (src/RRRMC.jl:81-127) on +-J sparse graphs that do NOT fit the LDS-resident
// sweep_kernel (N beyond 32 767, or wherever the planner's LDS buffers do not fit: GraphEA(64, 3) has N = 262 144).  Same chain, same streams, same planner idea as
// sparse_kernels.hpp — the site sequence is cut into chunks, every chunk into dependency levels whose attempts commute —
// but nothing here is sized by N:
//   plan_big_kernel<K>   one workgroup per chunk (<= 4096 attempts): the attempts are SORTED by (site, index) in LDS (bitonic), an
//                        attempt's predecessors — the latest earlier attempt at every site of its closed neighbourhood — are
//                        found by binary search, levels by relaxation as in plan_kernel.  Besides the slot list it writes every slot's
//                        RECORD (slot word + neighbour row, 8 words) in slot order, so that the sweeps stream their inputs.
//   up to 512 replica groups (16 384 replicas), the default:
//   big_mask_kernel<K>   the ACCEPT planes do not depend on the state: every (slot, group) of a batch gets its acceptance masks
//                        ("u < T_n" for the 32 replicas, every class with dE > 0) from the whole device, ahead of the sweep.
//   big_apply_kernel<K>  one workgroup per r = 2^lgr replicas whose spins fit LDS (r = 4 at N = 262 144): the levels in order, the
//                        inputs (records, masks) fetched two rounds ahead, spin bits gathered from LDS, a flip is one LDS atomic XOR.
//   big_merge_kernel     puts the workgroups' LDS images back into the context's [G][N] words.
//   beyond (or with RRRMC_BIG_NO_MASKS=1):
//   big_sweep_kernel<K>  one workgroup per group of 32 bit-sliced replicas, spins in HBM/L2 ([G][N] words, the context's native
//                        layout); a level's attempts are spread over the 1024 threads, one __syncthreads per level.  A thread
//                        does the whole attempt: gathers the K+1 words, counts the unsatisfied bonds bit-sliced, evaluates the
//                        ACCEPT planes (three blocks for all replicas while the gathers are in flight, then lazily for the replicas
//                        still undecided), flips.
//   Both sweeps count the accepted moves in per-thread bit-sliced counters (how many, and the sum of their n) that are added up
//   over the wavefront only at sample points (big_flush): E += 2 K A - 4 N.
// Bit-identical to the oracle (and to sweep_kernel where both apply); the colour-parallel sweeps remain the fast path for lattices,
// this path makes the reference's own dynamics available at any size.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "philox.hpp"
#include "sparse_kernels.hpp"

namespace rrrmc {

constexpr int kBigChunk = 4096;            // attempts per chunk: the index inside the chunk takes 12 bits of a slot word
constexpr int kBigSiteBits = 20;           // sites take the other 20: N <= 2^20
constexpr int kBigThreads = 1024;
// words of a slot's record: the slot word, then the K neighbour words — one 16-byte load up to K = 3, two beyond
__host__ __device__ constexpr int big_rec_words(int K) { return K <= 3 ? 4 : 8; }

inline size_t plan_big_lds_bytes(int K)
{
    return (size_t)kBigChunk * 4 * 2            // sort keys, sites
           + (size_t)kBigChunk * 2              // levels
           + (size_t)kBigChunk * (K + 1) * 2    // predecessors
           + (size_t)(kBigChunk + 2) * 2 * 3;   // level counters
}

// Where a site's r = 2^lgr bits live in big_apply_kernel's LDS image (S = 32 / r sites per word): byte address of the word in bits 0..21,
// position of the bits inside it in bits 22..26 — what the planner writes into the records when the context runs that kernel (lgr >= 0)
__host__ __device__ inline uint32_t big_lds_ref(uint32_t site, int lgr)
{
    const uint32_t lgS = 5u - (uint32_t)lgr;
    return ((site >> lgS) << 2) | (((site & ((1u << lgS) - 1u)) << lgr) << 22);
}

//   slots[slot_base + p] = site | (t << 20)   attempts sorted by dependency level (t = index in chunk)
//   nbrs [(slot_base + p) * W] = the slot word again, [.. + 1 + k] = k-th neighbour of that site | (J < 0) << 31   (the slot's RECORD:
//                                                W = 4 words up to K = 3, 8 beyond: one or two 16-byte loads; with lgr >= 0 the sites
//                                                are written as big_lds_ref(site, lgr), the form big_apply_kernel consumes)
//   vecs [slot_base + l] = first slot of level l + 1 (l = 0 .. nvec-1); chunks[c].nvec = number of levels
template <int K>
__global__ __launch_bounds__(kPlanThreads) void plan_big_kernel(ChunkDesc* __restrict__ chunks, uint32_t* __restrict__ slots,
                                                                uint32_t* __restrict__ nbrs, uint32_t* __restrict__ vecs,
                                                                const int32_t* __restrict__ A, const int8_t* __restrict__ J,
                                                                int N, uint32_t k0, uint32_t k1, uint64_t gbase, int lgr)
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
        nbrs[((size_t)cd.slot_base + pos) * big_rec_words(K)] = lgr < 0 ? s_site[t] | ((uint32_t)t << kBigSiteBits) : big_lds_ref(s_site[t], lgr);
        s_lvl[t] = (uint16_t)pos;                                        // the levels are done: keep the slot of attempt t
    }
    __syncthreads();
    // the neighbour row of every slot, in slot order (the sweep reads it coalesced instead of chasing A and J per attempt)
    for (int q = tid; q < count * K; q += kPlanThreads) {
        const int t = q / K, k = q - t * K;
        const size_t e = (size_t)s_site[t] * K + k;
        nbrs[((size_t)cd.slot_base + s_lvl[t]) * big_rec_words(K) + 1 + k] = (lgr < 0 ? (uint32_t)A[e] : big_lds_ref((uint32_t)A[e], lgr)) | (J[e] < 0 ? 0x80000000u : 0u);
    }
}

struct BigSweepParams {
    uint32_t* spins;          // [G][N]   bit-sliced configuration
    const ChunkDesc* chunks;
    const uint32_t* slots;
    const uint32_t* nbrs;     // [slot][big_rec_words(K)]   slot word, then neighbour | (J < 0) << 31 for the K neighbours: written by the planner in slot order
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

constexpr int kBigUnroll = 4;                // attempts of one level per thread and round: their loads are all in flight together
constexpr int kBigEager = 3;                 // ACCEPT blocks evaluated for all 32 replicas while the gathers are in flight (state-independent)
constexpr int kBigCntA = 8;                  // per-thread bit-sliced counters between two flushes: accepted moves (<= 255 attempts of a thread) ...
constexpr int kBigCntN = 11;                 // ... and the sum of their n (<= 7 each)
constexpr int kBigFlushAttempts = (1 << kBigCntA) - 1;

// c += x over bit-sliced numbers (plane j = bit j of 32 replicas' values); WX <= W planes come in, the carry ripples through the rest
template <int W, int WX>
__device__ __forceinline__ void planes_add(uint32_t (&c)[W], const uint32_t (&x)[WX])
{
    uint32_t carry = 0u;
#pragma unroll
    for (int j = 0; j < W; ++j) {
        if (j == 0) { carry = c[0] & x[0]; c[0] ^= x[0]; }
        else if (j < WX) { const uint32_t a = c[j], b = x[j]; c[j] = bitop3<0x96>(a, b, carry); carry = bitop3<0xe8>(a, b, carry); }
        else { const uint32_t a = c[j]; c[j] = a ^ carry; carry &= a; }
    }
}

// Adds every thread's counters to the workgroup's per-replica totals (s_E += 2 K A - 4 N, s_A += A) and clears them: a butterfly of
// bit-sliced additions over the wavefront (6 steps), then lane r < 32 reads replica r's two numbers out of the planes.
template <int K>
__device__ __forceinline__ void big_flush(uint32_t (&cA)[kBigCntA], uint32_t (&cN)[kBigCntN], int32_t* s_E, int32_t* s_A, int lane)
{
    constexpr int WA = kBigCntA + 6, WN = kBigCntN + 6;
    uint32_t a[WA], n[WN];
#pragma unroll
    for (int j = 0; j < WA; ++j) a[j] = j < kBigCntA ? cA[j] : 0u;
#pragma unroll
    for (int j = 0; j < WN; ++j) n[j] = j < kBigCntN ? cN[j] : 0u;
#pragma unroll
    for (int st = 1; st < 64; st <<= 1) {
        uint32_t oa[WA], on[WN];
#pragma unroll
        for (int j = 0; j < WA; ++j) oa[j] = (uint32_t)__shfl_xor((int)a[j], st);
#pragma unroll
        for (int j = 0; j < WN; ++j) on[j] = (uint32_t)__shfl_xor((int)n[j], st);
        planes_add<WA, WA>(a, oa);
        planes_add<WN, WN>(n, on);
    }
    if (lane < 32) {
        int32_t A = 0, Nn = 0;
#pragma unroll
        for (int j = 0; j < WA; ++j) A |= (int32_t)((a[j] >> lane) & 1u) << j;
#pragma unroll
        for (int j = 0; j < WN; ++j) Nn |= (int32_t)((n[j] >> lane) & 1u) << j;
        if (A) {
            atomicAdd(&s_E[lane], 2 * K * A - 4 * Nn);       // dE = 2 (K - 2 n) per accepted move
            atomicAdd(&s_A[lane], A);
        }
    }
#pragma unroll
    for (int j = 0; j < kBigCntA; ++j) cA[j] = 0u;
#pragma unroll
    for (int j = 0; j < kBigCntN; ++j) cN[j] = 0u;
}

template <int K>
__global__ __launch_bounds__(kBigThreads) void big_sweep_kernel(BigSweepParams P)
{
    constexpr int NT = (K + 1) / 2;          // classes with dE > 0: n = 0 .. NT-1 unsatisfied bonds
    constexpr int U = K <= 6 ? kBigUnroll : kBigUnroll / 2;      // K = 7: four rows in flight do not fit 128 registers
    __shared__ int32_t s_E[32], s_A[32];
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t group = P.group0 + blockIdx.x;
    uint32_t* gsp = P.spins + (size_t)blockIdx.x * P.N;
    if (tid < 32) { s_E[tid] = P.E_cur[blockIdx.x * 32 + tid]; s_A[tid] = 0; }
    // accepted moves since the last flush, per thread and bit-sliced over the 32 replicas: how many (cA) and the sum of their n (cN)
    uint32_t cA[kBigCntA], cN[kBigCntN];
#pragma unroll
    for (int j = 0; j < kBigCntA; ++j) cA[j] = 0u;
#pragma unroll
    for (int j = 0; j < kBigCntN; ++j) cN[j] = 0u;
    uint32_t pending = 0u;                   // attempts the busiest thread (thread 0) has counted since the last flush
    int64_t ns = P.sample0;
    __syncthreads();
    for (int c = 0; c < P.nchunks; ++c) {
        const ChunkDesc cd = P.chunks[c];
        if (cd.flags & kChunkSampleBefore) {                 // sample BEFORE the move of iteration k*step (RRRMC.jl:104-108)
            if (pending) { big_flush<K>(cA, cN, s_E, s_A, lane); pending = 0u; }
            __syncthreads();
            if (tid < 32) P.Es[ns * P.Rpad + blockIdx.x * 32 + tid] = s_E[tid];
            ns += 1;
        }
        const uint32_t* slots = P.slots + cd.slot_base;
        const uint32_t* nbrs = P.nbrs + (size_t)cd.slot_base * big_rec_words(K);      // in-chunk offsets below fit 32 bits
        for (uint32_t l = 0; l < cd.nvec; ++l) {
            const uint32_t start = P.vecs[cd.slot_base + l];
            const uint32_t end = l + 1 < cd.nvec ? P.vecs[cd.slot_base + l + 1] : cd.count;
            const uint32_t rounds = (end - start + (uint32_t)kBigThreads - 1u) / (uint32_t)kBigThreads;
            if (pending + rounds > (uint32_t)kBigFlushAttempts) { big_flush<K>(cA, cN, s_E, s_A, lane); pending = 0u; }
            pending += rounds;
            for (uint32_t base = start; base < end; base += (uint32_t)(U * kBigThreads)) {
                // the attempts of a level commute: everything is loaded before anything is written
                uint32_t slot[U], sg[U], s[U], gk[U][K];
                bool live[U];
                const char* gbytes = reinterpret_cast<const char*>(gsp);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t p = base + (uint32_t)(u * kBigThreads + tid);
                    live[u] = p < end;
                    slot[u] = 0u;
                    sg[u] = 0u;
                    if (live[u]) {
                        slot[u] = slots[p];
#pragma unroll
                        for (int k = 0; k < K; ++k) gk[u][k] = nbrs[p * (uint32_t)big_rec_words(K) + 1u + (uint32_t)k];
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (live[u]) {
                        s[u] = *reinterpret_cast<const uint32_t*>(gbytes + ((slot[u] & ((1u << kBigSiteBits) - 1u)) << 2));
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            const uint32_t w = gk[u][k];
                            sg[u] |= (w >> 31) << k;                                             // coupling signs of the row, one bit per bond
                            gk[u][k] = *reinterpret_cast<const uint32_t*>(gbytes + (w << 2));    // 32-bit byte offset (N <= 2^20): the shift drops the sign
                        }
                    }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (!live[u]) continue;
                    const uint32_t site = slot[u] & ((1u << kBigSiteBits) - 1u);
                    const uint64_t g = P.gbase + cd.g0 + (uint64_t)(slot[u] >> kBigSiteBits);
                    // u < T_n for all 32 replicas and every class with dE > 0 (a counter-based stream: the bits do not depend on
                    // who asks) — independent of the state, so it runs while the gathers are in flight
                    uint32_t lt[NT], eq[NT];
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const bool always = (P.always_mask >> n) & 1u;
                        lt[n] = always ? 0xffffffffu : 0u;
                        eq[n] = always ? 0u : 0xffffffffu;
                    }
                    if (any_set<NT>(eq)) {
#pragma unroll
                        for (int b = 0; b < kBigEager; ++b)
                            refine_block<NT>(lt, eq, accept_planes(P.k0, P.k1, g, group, (uint32_t)b), (uint32_t)b, P.taum);
                    }
                    // n = number of unsatisfied bonds, bit-sliced (three planes cover K <= 7)
                    uint32_t un[K], n0, n1, n2;
#pragma unroll
                    for (int k = 0; k < K; ++k) un[k] = s[u] ^ gk[u][k] ^ (0u - ((sg[u] >> k) & 1u));      // bond k unsatisfied
                    count_planes<K>(un, n0, n1, n2);
                    // classes n = 0..NT-1 have dE = 2(K - 2n) > 0 and need u < T_n; every other class is accepted (RRRMC.jl:39)
                    uint32_t need[NT];
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        uint32_t e = (n & 1) ? n0 : ~n0;
                        e &= (n & 2) ? n1 : ~n1;
                        e &= (n & 4) ? n2 : ~n2;
                        need[n] = e;
                        eq[n] &= e;                          // only the replicas of the class still matter
                    }
                    for (uint32_t pb = (uint32_t)kBigEager; pb < 16u && any_set<NT>(eq); ++pb)
                        refine_block<NT>(lt, eq, accept_planes(P.k0, P.k1, g, group, pb), pb, P.taum);
                    uint32_t rej = 0u;
#pragma unroll
                    for (int n = 0; n < NT; ++n) rej |= need[n] & ~lt[n];
                    const uint32_t acc = ~rej;
                    gsp[site] = s[u] ^ acc;                  // spinflip! + update_cache! (Interface.jl:89-92, RRG.jl:191-234)
                    // bookkeeping of the accepted moves (E += dE, accepted += 1), deferred: count them here, add them up at the next flush
                    const uint32_t xa[1] = {acc};
                    const uint32_t xn[3] = {n0 & acc, n1 & acc, n2 & acc};
                    planes_add<kBigCntA, 1>(cA, xa);
                    planes_add<kBigCntN, 3>(cN, xn);
                }
            }
            __syncthreads();                                 // the next level reads what this one wrote
        }
    }
    if (pending) big_flush<K>(cA, cN, s_E, s_A, lane);
    __syncthreads();
    if (tid < 32) {
        P.E_cur[blockIdx.x * 32 + tid] = s_E[tid];
        P.acc_cur[blockIdx.x * 32 + tid] += (int64_t)s_A[tid];
    }
}

// ---- few replica groups: the random numbers of a batch are turned into acceptance masks by the WHOLE device first ----------------
// big_sweep_kernel gives a replica group one CU, and most of that CU's time goes into the ACCEPT planes — which do not depend on the
// state.  big_mask_kernel<K> evaluates them for every (slot, group) of a batch on all CUs (lt_n = "u < T_n" for the 32 replicas and every
// class with dE > 0, resolved to the last replica) and big_apply_kernel<K> runs the levels with what is left: gathers, the bond count,
// one select per class, the flip.      masks[(group * cap + slot_base + p) * 4 + n]   (cap = slots per batch of the context; one 16-byte load),
// or, where the NT * r mask bits of a workgroup's r replicas fit a word: masks[(group * S + sub) * cap + slot_base + p]
struct BigMaskParams {
    const ChunkDesc* chunks;
    const uint32_t* slots;
    uint32_t* masks;
    uint32_t taum[64 * 4];
    uint32_t always_mask;
    uint32_t k0, k1, group0;
    uint64_t gbase;
    uint32_t cap;
    int lgr;                  // >= 0: compact masks for big_apply_kernel's workgroups of 2^lgr replicas (NT * 2^lgr <= 32), one word per
                              // (slot, workgroup): class n's bits at n * r;  < 0: four words per (slot, group)
};
constexpr int kBigMaskThreads = 256;
// one word holds the masks of a workgroup's r replicas for all NT classes?
__host__ __device__ inline bool big_masks_compact(int K, int lgr) { return lgr <= 4 && (((K + 1) / 2) << lgr) <= 32; }

template <int K>
__global__ __launch_bounds__(kBigMaskThreads) void big_mask_kernel(BigMaskParams M)
{
    constexpr int NT = (K + 1) / 2;
    constexpr int BPC = kBigChunk / kBigMaskThreads;             // blocks per chunk
    const ChunkDesc cd = M.chunks[blockIdx.x / BPC];
    const uint32_t p = (blockIdx.x % BPC) * (uint32_t)kBigMaskThreads + threadIdx.x;
    if (p >= cd.count) return;
    const uint32_t slot = M.slots[cd.slot_base + p];
    const uint64_t g = M.gbase + cd.g0 + (uint64_t)(slot >> kBigSiteBits);
    const uint32_t group = M.group0 + blockIdx.y;
    uint32_t lt[NT], eq[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const bool always = (M.always_mask >> n) & 1u;
        lt[n] = always ? 0xffffffffu : 0u;
        eq[n] = always ? 0u : 0xffffffffu;
    }
    if (any_set<NT>(eq)) {
#pragma unroll
        for (int b = 0; b < kBigEager; ++b) refine_block<NT>(lt, eq, accept_planes(M.k0, M.k1, g, group, (uint32_t)b), (uint32_t)b, M.taum);
        for (uint32_t pb = (uint32_t)kBigEager; pb < 16u && any_set<NT>(eq); ++pb)
            refine_block<NT>(lt, eq, accept_planes(M.k0, M.k1, g, group, pb), pb, M.taum);
    }
    if (M.lgr >= 0) {
        const uint32_t r = 1u << M.lgr, S = 32u >> M.lgr, rm = (1u << r) - 1u;          // r <= 16 here
        for (uint32_t sub = 0; sub < S; ++sub) {
            uint32_t w = 0u;
#pragma unroll
            for (int n = 0; n < NT; ++n) w |= ((lt[n] >> (sub << M.lgr)) & rm) << ((uint32_t)n << M.lgr);
            M.masks[((size_t)blockIdx.y * S + sub) * M.cap + cd.slot_base + p] = w;
        }
        return;
    }
    uint32_t o[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int n = 0; n < NT; ++n) o[n] = lt[n];
    reinterpret_cast<uint4*>(M.masks)[(size_t)blockIdx.y * M.cap + cd.slot_base + p] = make_uint4(o[0], o[1], o[2], o[3]);
}

// big_apply_kernel<K>: the spins of r = 2^lgr replicas of a group live in LDS, r bits per site and S = 32 / r sites per word (r = 4 at
// N = 262 144: 128 KiB) — one workgroup per r replicas, S workgroups per group.  A CU's L1 resolves about one scattered line per
// cycle, which bounds a kernel that gathers the K + 1 spin words of every attempt from L2 at ~14 us per 4096-attempt chunk whatever
// else it does (measured: the same time with and without the random numbers); LDS serves the same gathers an order of magnitude faster,
// and splitting a group over S workgroups fills the device at few replicas (512 replicas = 128 workgroups instead of 16).  A flip is
// one LDS atomic XOR of the accepted bits (other sites share the word); reads never touch a site written in the same level.
// The levels of a chunk are short chains (level table -> slot records -> spin words -> barrier), so the kernel is one flat sequence of
// ROUNDS (<= 2048 attempts of one level, four per thread) with everything that does not depend on the state fetched a round ahead:
// the records of round i + 1 (site, neighbour row, masks) are in flight while round i runs, and the headers of the launch's chunks
// (descriptor + the first levels' boundaries) are copied into LDS up front, so the round iterator never touches memory.
// Start and end: the workgroup extracts its replicas' bits from the context's [G][N] words and leaves its LDS image in HBM, from where
// big_merge_kernel rebuilds the words (no cross-workgroup atomics on the shared words).
inline int big_lds_lgr(int64_t N)
{
    int lgr = 5;                                                 // r = 32: the whole group in one workgroup
    while (lgr > 0 && ((N + (32 >> lgr) - 1) / (32 >> lgr)) * 4 > 128 * 1024) --lgr;
    return lgr;
}
constexpr int kBigApplyChunks = 512;         // chunks per launch of big_apply_kernel: their headers (12 words each) sit in LDS beside the spins
constexpr int kBigHdr = 12;                  // count, slot_base, nvec, flags, first slot of levels 0..7
inline size_t big_lds_bytes(int64_t N, int lgr) { return (size_t)((N + (32 >> lgr) - 1) / (32 >> lgr)) * 4 + (size_t)kBigApplyChunks * kBigHdr * 4 + 64 * 4; }      // image, headers, per-replica E / accepted

#ifndef RRRMC_BIG_APPLY_THREADS
#define RRRMC_BIG_APPLY_THREADS 1024
#endif
#ifndef RRRMC_BIG_ROUND
#define RRRMC_BIG_ROUND 2048                 // attempts per round of big_apply_kernel
#endif
constexpr int kBigApplyThreads = RRRMC_BIG_APPLY_THREADS;      // four wavefronts per SIMD, two attempts per thread and round (512 x 4 measured 12 % slower)

// Workgroup barrier for state that lives in LDS only: waits for this wavefront's LDS operations, not for its global loads —
// __syncthreads() would also drain the vector-memory counter, i.e. the record prefetch that is meant to stay in flight across it.
__device__ __forceinline__ void big_lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}        // two wavefronts per SIMD: 256 registers each, room for three rounds of records

template <int I> struct BigStage { static constexpr int value = I; };       // a compile-time stage index for the generic lambdas below

template <int K, bool CM>
__device__ __forceinline__ void big_apply_body(const BigSweepParams& P, const uint32_t* __restrict__ masks, uint32_t* __restrict__ img, uint32_t cap, int lgr)
{
    constexpr int NT = (K + 1) / 2;
    constexpr int REC = big_rec_words(K);
    constexpr int U = RRRMC_BIG_ROUND / kBigApplyThreads;     // attempts per thread and round
    constexpr int NTH = kBigApplyThreads;
    constexpr uint32_t RS = (uint32_t)(U * NTH);              // slots per round
    // the image starts at LDS address 0 (no static __shared__ in this kernel): a record's reference IS the address its gather uses
    extern __shared__ uint32_t bl_sp[];                       // [ceil(N / S)]  r bits per site; then the chunk headers; then s_E[32], s_A[32]
    const int tid = threadIdx.x, lane = tid & 63;
    // gathers address the image by ABSOLUTE LDS address (through the array's name every gather pays an add of the array's link-time address,
    // which is 0): should the layout ever put something in front of the image, the kernel traps — the launch fails with a HIP error at the
    // next synchronisation instead of leaving stale energies and spins behind
    typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
    if (__builtin_amdgcn_readfirstlane((int)(uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t*)bl_sp) != 0) __builtin_trap();
    auto lds_word_at = [](uint32_t byte_addr) -> uint32_t { return *(lds_cu32*)(size_t)byte_addr; };
    const uint32_t r = 1u << lgr, lgS = 5u - (uint32_t)lgr, S = 1u << lgS, rm = r == 32u ? 0xffffffffu : (1u << r) - 1u;
    const uint32_t grp = blockIdx.x >> lgS, sub = blockIdx.x & (S - 1u), rsh = sub << lgr;      // replicas rsh .. rsh + r - 1 of group grp
    const uint32_t nwords = ((uint32_t)P.N + S - 1u) >> lgS;
    uint32_t* gsp = P.spins + (size_t)grp * P.N;
    // 32-bit byte offsets below: one address register per load
    const char* gmask = reinterpret_cast<const char*>(masks) + (CM ? (size_t)blockIdx.x * cap * 4 : (size_t)grp * cap * 16);
    const char* recs = reinterpret_cast<const char*>(P.nbrs);
    const uint32_t rep0 = grp * 32u + rsh;
    int32_t* s_E = reinterpret_cast<int32_t*>(bl_sp + nwords + (uint32_t)(kBigApplyChunks * kBigHdr));
    int32_t* s_A = s_E + 32;
    if (tid < (int)r) { s_E[tid] = P.E_cur[rep0 + tid]; s_A[tid] = 0; }
    // this workgroup's bits of every site
    for (uint32_t wd = (uint32_t)tid; wd < nwords; wd += (uint32_t)NTH) {
        uint32_t w = 0u;
        for (uint32_t j = 0; j < S; ++j) {
            const uint32_t site = (wd << lgS) + j;
            if (site < (uint32_t)P.N) w |= ((gsp[site] >> rsh) & rm) << (j << lgr);
        }
        bl_sp[wd] = w;
    }
    uint32_t cA[kBigCntA], cN[kBigCntN];
#pragma unroll
    for (int j = 0; j < kBigCntA; ++j) cA[j] = 0u;
#pragma unroll
    for (int j = 0; j < kBigCntN; ++j) cN[j] = 0u;
    uint32_t pending = 0u;
    int64_t ns = P.sample0;

    // the headers of the launch's chunks, in LDS: everything the round iterator reads comes from here, so that the only vector-memory
    // loads in the loop are the records — whatever the compiler has to wait for, it is something the round needs anyway
    uint32_t* hdr = bl_sp + nwords;                           // [nchunks][kBigHdr]
    for (int i = tid; i < P.nchunks * kBigHdr; i += NTH) {
        const int cc = i / kBigHdr, j = i - cc * kBigHdr;
        const ChunkDesc& d = P.chunks[cc];
        uint32_t v;
        if (j == 0) v = d.count;
        else if (j == 1) v = d.slot_base;
        else if (j == 2) v = d.nvec;
        else if (j == 3) v = d.flags;
        else v = (uint32_t)(j - 4) < d.nvec ? P.vecs[d.slot_base + (uint32_t)(j - 4)] : d.count;
        hdr[i] = v;
    }
    __syncthreads();
    // iterator over the rounds (wave-uniform), always at the round whose records are requested next:
    // chunk c = (sb, count, nvec, flags), level l = [.., lend), round = [base, base + RS)
    int c = 0;
    uint32_t count = 0u, sb = 0u, nvec = 0u, flags = 0u;
    auto enter_chunk = [&]() {
        const uint32_t* h = hdr + c * kBigHdr;
        count = (uint32_t)__builtin_amdgcn_readfirstlane((int)h[0]); sb = (uint32_t)__builtin_amdgcn_readfirstlane((int)h[1]);
        nvec = (uint32_t)__builtin_amdgcn_readfirstlane((int)h[2]); flags = (uint32_t)__builtin_amdgcn_readfirstlane((int)h[3]);
    };
    auto lstart = [&](uint32_t l) -> uint32_t {               // first slot of level l of chunk c (l < nvec)
        // (written so that the common case is an LDS read and nothing else: as one conditional expression the two became ONE flat load,
        // whose wait drains the vector-memory counter — i.e. the record prefetch — at every level boundary)
        uint32_t v = hdr[c * kBigHdr + 4 + (int)(l < (uint32_t)(kBigHdr - 5) ? l : (uint32_t)(kBigHdr - 5))];
        if (l >= (uint32_t)(kBigHdr - 4)) {                    // (a chunk with more than 8 levels: rare; the load as assembly keeps the two apart)
            const uint32_t* pv = P.vecs + sb + l;
            asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(pv) : "memory");
        }
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    };
    bool valid = P.nchunks > 0, first = true, cfirst = true;
    if (valid) enter_chunk();
    uint32_t l = 0u, base = 0u, lend = nvec > 1u ? lstart(1u) : count;
    // two rounds of records in registers (stage = round mod 2; every index below is a compile-time constant):
    // <= U * kBigApplyThreads attempts of one level each, plus what the round is (wave-uniform): a round at all / the first of its
    // level / the first of its chunk / that chunk starts with a sample
    // The loads are inline assembly and so is the wait for them (s_waitcnt vmcnt(kLoads) right after the NEXT round's request: the counter
    // is in order, "all but the youngest kLoads" is the round about to be worked on, requested a whole round ago).  Every wavefront issues
    // every load of every round — lanes without an attempt masked off in EXEC, which costs an issue slot and no traffic — so kLoads is a
    // compile-time constant.  Left to the compiler the same loop drained the counter: with the loads under a wave-uniform "any lane live"
    // branch the count differs between wavefronts, and with unconditional C++ loads its register allocation reused stage registers as
    // address temporaries and waited (vmcnt(1)) for the round in flight before requesting the next one.
    // Hand-placed loads and waits are only as safe as the compiler's register allocation lets them be: a stage register that it spilled or
    // copied between the request and the wait would be read before its data has arrived.  The compact-mask build (the one that matters:
    // r <= 8) has no scratch and no such copies — tests/test_kernel_resources.py compiles it for every K and checks — and takes this route;
    // the four-word-mask build (r = 16, 32) spills a few loop invariants at K >= 4 and keeps compiler-managed loads, requested after the
    // gathers as in round 2.
    constexpr bool ASM = CM;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    constexpr int kLoads = U * ((REC == 4 ? 1 : 2) + 1);
    u32x4 q_a[2][U], q_b[2][U], q_mv[2][U];                   // record words 0..3, 4..7 (REC == 8), the four mask words (!CM)
    uint32_t q_mc[2][U];                                      // the compact mask word (CM)
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int u = 0; u < U; ++u) { q_a[T][u] = (u32x4)(0u); q_b[T][u] = (u32x4)(0u); q_mv[T][u] = (u32x4)(0u); q_mc[T][u] = 0u; }
    auto q_slot = [&](int T, int u) -> uint32_t { return q_a[T][u].x; };
    auto q_raw = [&](int T, int u, int k) -> uint32_t {
        return k == 0 ? q_a[T][u].y : k == 1 ? q_a[T][u].z : k == 2 ? q_a[T][u].w : k == 3 ? q_b[T][u].x : k == 4 ? q_b[T][u].y : k == 5 ? q_b[T][u].z : q_b[T][u].w;
    };
    bool q_live[2][U], q_valid[2], q_first[2], q_cfirst[2], q_sample[2];
    // requests the records of the iterator's round and moves the iterator on
    auto fetch = [&](auto stage) {
        constexpr int T = decltype(stage)::value;
        q_valid[T] = valid; q_first[T] = first; q_cfirst[T] = cfirst; q_sample[T] = (flags & kChunkSampleBefore) != 0u;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t p = base + (uint32_t)(u * NTH + tid);
            q_live[T][u] = valid && p < lend;
            if constexpr (!ASM) {
                // compiler-managed loads under a wave-uniform "any lane live" branch (a dead lane re-reads the round's first slot)
                if (__builtin_amdgcn_ballot_w64(q_live[T][u]) != 0ull) {
                    const uint32_t q = sb + (q_live[T][u] ? p : base);
                    if constexpr (REC == 4) q_a[T][u] = *reinterpret_cast<const u32x4*>(recs + (q << 4));
                    else { q_a[T][u] = *reinterpret_cast<const u32x4*>(recs + (q << 5)); q_b[T][u] = *reinterpret_cast<const u32x4*>(recs + (q << 5) + 16u); }
                    q_mv[T][u] = *reinterpret_cast<const u32x4*>(gmask + (q << 4));
                }
                continue;
            }
            const unsigned long long lm = __builtin_amdgcn_ballot_w64(q_live[T][u]);
            const uint32_t q = q_live[T][u] ? sb + p : 0u;
            unsigned long long sv;
            // 16-byte loads (a CU's vector memory path takes a wavefront's load every ~16 cycles whatever its width; what the loop
            // pays for is the bytes: 64 per cycle and CU)
            if constexpr (REC == 4 && CM)
                asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b64 exec, %[lm]\n\tglobal_load_dwordx4 %[a], %[o], %[rb]\n\tglobal_load_dword %[m], %[om], %[mb]\n\ts_mov_b64 exec, %[sv]"
                             : [a] "+v"(q_a[T][u]), [m] "+v"(q_mc[T][u]), [sv] "=&s"(sv)
                             : [o] "v"(q << 4), [om] "v"(q << 2), [rb] "s"(recs), [mb] "s"(gmask), [lm] "s"(lm) : "memory");
            else
                asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b64 exec, %[lm]\n\tglobal_load_dwordx4 %[a], %[o], %[rb]\n\tglobal_load_dwordx4 %[b], %[o], %[rb] offset:16\n\t"
                             "global_load_dword %[m], %[om], %[mb]\n\ts_mov_b64 exec, %[sv]"
                             : [a] "+v"(q_a[T][u]), [b] "+v"(q_b[T][u]), [m] "+v"(q_mc[T][u]), [sv] "=&s"(sv)
                             : [o] "v"(q << 5), [om] "v"(q << 2), [rb] "s"(recs), [mb] "s"(gmask), [lm] "s"(lm) : "memory");
        }
        if (valid) {
            base += RS;
            first = false; cfirst = false;
            if (base >= lend) {
                l += 1u;
                first = true;
                if (l < nvec) {
                    base = lend;
                    lend = l + 1u < nvec ? lstart(l + 1u) : count;
                } else {
                    c += 1;
                    cfirst = true;
                    if (c >= P.nchunks) valid = false;
                    else {
                        enter_chunk();
                        l = 0u; base = 0u;
                        lend = nvec > 1u ? lstart(1u) : count;
                    }
                }
            }
        }
    };
#ifdef RRRMC_BIG_STAMPS
    uint64_t st_fetch = 0, st_pre = 0, st_lds = 0, st_dec = 0, st_rounds = 0, st_first = 0;
    const uint64_t st_begin = __builtin_amdgcn_s_memtime();
#endif
    // waits until at most NL vector-memory loads are outstanding; stage T's registers pass through the statement, so that nothing that
    // reads them can be scheduled above it
    auto wait_stage = [&](auto stage, auto nl) {
        constexpr int T = decltype(stage)::value, NL = decltype(nl)::value;
        (void)q_a; (void)q_b; (void)q_mv; (void)q_mc;          // (named here so that the lambda captures them: asm operands under if constexpr alone do not)
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            if constexpr (REC == 4) asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(q_a[T][u]), "+v"(q_mc[T][u]) : [n] "n"(NL) : "memory");
            else asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(q_a[T][u]), "+v"(q_b[T][u]), "+v"(q_mc[T][u]) : [n] "n"(NL) : "memory");
        }
    };
    // one round: stage A is executed, the next round is requested into stage C (the one that was executed last)
    auto round = [&](auto stA, auto stC) {
        constexpr int A = decltype(stA)::value;
#ifdef RRRMC_BIG_STAMPS
        const uint64_t tf = __builtin_amdgcn_s_memtime();
#endif
        // ---- the next round: request its records first; they arrive while this round is worked on ----
#ifndef RRRMC_BIG_FETCH_LATE
        if constexpr (ASM) {
            fetch(stC);
            wait_stage(stA, BigStage<kLoads>{});
        }
#else
        if constexpr (ASM) wait_stage(stA, BigStage<0>{});
#endif
#ifdef RRRMC_BIG_STAMPS
        const uint64_t t1 = __builtin_amdgcn_s_memtime();
#endif
        // ---- what must happen before the round's attempts ----
        if (q_cfirst[A] && q_sample[A]) {                        // sample BEFORE the move of iteration k*step (RRRMC.jl:104-108)
            if (pending) { big_flush<K>(cA, cN, s_E, s_A, lane); pending = 0u; }
            big_lds_barrier();
            if (tid < (int)r) P.Es[ns * P.Rpad + rep0 + tid] = s_E[tid];
            ns += 1;
        }
        if (pending + (uint32_t)U > (uint32_t)kBigFlushAttempts) { big_flush<K>(cA, cN, s_E, s_A, lane); pending = 0u; }
        pending += (uint32_t)U;
        if (q_first[A]) big_lds_barrier();                       // a new level reads what the previous one wrote
#ifdef RRRMC_BIG_STAMPS
        const uint64_t t2 = __builtin_amdgcn_s_memtime();
#endif
        // ---- the spin bits of this round (records hold big_lds_ref: byte address of the word | position << 22 | (J < 0) << 31) ----
        const char* lbytes = reinterpret_cast<const char*>(bl_sp);
        uint32_t sw[U], gk[U][K];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // every lane gathers, attempt or not (a lane without one follows the references its registers hold from an earlier round — valid
            // addresses — or word 0): no zero-initialisation, no branch
            {
                const uint32_t e0 = q_slot(A, u);
                sw[u] = lds_word_at(e0 & 0x3fffffu) >> ((e0 >> 22) & 31u);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const uint32_t e = q_raw(A, u, k);
                    gk[u][k] = lds_word_at(e & 0x3fffffu) >> ((e >> 22) & 31u);
                }
            }
        }
#ifdef RRRMC_BIG_STAMPS
        uint32_t chk = 0u;
#pragma unroll
        for (int u = 0; u < U; ++u) { chk ^= sw[u]; for (int k = 0; k < K; ++k) chk ^= gk[u][k]; }
        asm volatile("" :: "v"(chk));
        const uint64_t t3 = __builtin_amdgcn_s_memtime();
#endif
#ifdef RRRMC_BIG_FETCH_LATE
        fetch(stC);
#else
        if constexpr (!ASM) fetch(stC);
#endif
#ifdef RRRMC_BIG_STAMPS
        const uint64_t t0 = __builtin_amdgcn_s_memtime();
#endif
        // ---- decide, flip ----
        uint32_t va[U], vn0[U], vn1[U], vn2[U];                 // accepted replicas of the attempt and the planes of their n (zero for a dead lane)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            va[u] = 0u; vn0[u] = 0u; vn1[u] = 0u; vn2[u] = 0u;
            if (!q_live[A][u]) continue;
            uint32_t un[K], n0, n1, n2;
#pragma unroll
            for (int k = 0; k < K; ++k)                          // bond k unsatisfied: s ^ neighbour ^ (J < 0); bits above r are masked at the end
                un[k] = bitop3<0x96>(sw[u], gk[u][k], (uint32_t)((int32_t)q_raw(A, u, k) >> 31));
            count_planes<K>(un, n0, n1, n2);
            uint32_t rej = 0u;                                   // class n (dE = 2 (K - 2n) > 0) needs u < T_n; every other class is accepted
            // mask of class n in the low r bits (what lies above is cut off with the rest at the end)
            auto cmask = [&](int n) -> uint32_t { return CM ? q_mc[A][u] >> ((uint32_t)n << lgr) : (n == 0 ? q_mv[A][u].x : n == 1 ? q_mv[A][u].y : n == 2 ? q_mv[A][u].z : q_mv[A][u].w) >> rsh; };
            if constexpr (NT > 0) rej = bitop3<0xf4>(rej, count_is<0>(n0, n1, n2), cmask(0));
            if constexpr (NT > 1) rej = bitop3<0xf4>(rej, count_is<1>(n0, n1, n2), cmask(1));
            if constexpr (NT > 2) rej = bitop3<0xf4>(rej, count_is<2>(n0, n1, n2), cmask(2));
            if constexpr (NT > 3) rej = bitop3<0xf4>(rej, count_is<3>(n0, n1, n2), cmask(3));
            const uint32_t acc = ~rej & rm;
            const uint32_t e0 = q_slot(A, u);
            if (acc)                                             // spinflip! + update_cache! (Interface.jl:89-92, RRG.jl:191-234)
                atomicXor(reinterpret_cast<uint32_t*>(const_cast<char*>(lbytes) + (e0 & 0x3fffffu)), acc << ((e0 >> 22) & 31u));
            va[u] = acc; vn0[u] = n0 & acc; vn1[u] = n1 & acc; vn2[u] = n2 & acc;
        }
        // ---- count: the round's U attempts are added up first (carry-save), then once into the running counters ----
        {
            uint32_t a0, a1, a2, p0, p1, p2, q0, q1, q2, r0, r1, r2;
            count_planes<U>(va, a0, a1, a2);
            count_planes<U>(vn0, p0, p1, p2);                    // weight 1
            count_planes<U>(vn1, q0, q1, q2);                    // weight 2
            count_planes<U>(vn2, r0, r1, r2);                    // weight 4
            // P + 2 Q + 4 R <= 7 U = 28: five planes
            const uint32_t t1 = p1 ^ q0, c1 = p1 & q0;
            const uint32_t t2 = bitop3<0x96>(p2, q1, c1), c2 = bitop3<0xe8>(p2, q1, c1);
            const uint32_t t3 = q2 ^ c2, t4 = q2 & c2;
            const uint32_t u2 = t2 ^ r0, d2 = t2 & r0;
            const uint32_t u3 = bitop3<0x96>(t3, r1, d2), d3 = bitop3<0xe8>(t3, r1, d2);
            const uint32_t u4 = bitop3<0x96>(t4, r2, d3);
            const uint32_t xa[3] = {a0, a1, a2};
            const uint32_t xn[5] = {p0, t1, u2, u3, u4};
            planes_add<kBigCntA, 3>(cA, xa);
            planes_add<kBigCntN, 5>(cN, xn);
        }
#ifdef RRRMC_BIG_STAMPS
        const uint64_t t4 = __builtin_amdgcn_s_memtime();
        st_fetch += (t0 - t3) + (t1 - tf); st_pre += t2 - t1; st_lds += t3 - t2; st_dec += t4 - t0; st_rounds += 1; st_first += q_first[A] ? 1 : 0;
#endif
    };
    fetch(BigStage<0>{});
    __syncthreads();
    for (;;) {
        if (!q_valid[0]) break;
        round(BigStage<0>{}, BigStage<1>{});
        if (!q_valid[1]) break;
        round(BigStage<1>{}, BigStage<0>{});
    }
#ifdef RRRMC_BIG_STAMPS
    if (blockIdx.x == 3 && tid == 0)
        printf("big_apply stamps: total %llu fetch %llu pre %llu lds %llu decide %llu rounds %llu levels %llu chunks %d\n",
               (unsigned long long)(__builtin_amdgcn_s_memtime() - st_begin), (unsigned long long)st_fetch, (unsigned long long)st_pre,
               (unsigned long long)st_lds, (unsigned long long)st_dec, (unsigned long long)st_rounds, (unsigned long long)st_first, P.nchunks);
#endif
    if (pending) big_flush<K>(cA, cN, s_E, s_A, lane);
    __syncthreads();
    if (tid < (int)r) {
        P.E_cur[rep0 + tid] = s_E[tid];
        P.acc_cur[rep0 + tid] += (int64_t)s_A[tid];
    }
    // the workgroup's image goes to HBM as it is; big_merge_kernel puts the S images of a group back together
    for (uint32_t wd = (uint32_t)tid; wd < nwords; wd += (uint32_t)NTH) img[(size_t)blockIdx.x * nwords + wd] = bl_sp[wd];
}

template <int K>
__global__ __launch_bounds__(kBigApplyThreads) void big_apply_kernel(BigSweepParams P, const uint32_t* __restrict__ masks, uint32_t* __restrict__ img,
                                                                    uint32_t cap, int lgr)
{
    big_apply_body<K, false>(P, masks, img, cap, lgr);
}
template <int K>       // compact masks: one word per (slot, workgroup)
__global__ __launch_bounds__(kBigApplyThreads) void big_applyc_kernel(BigSweepParams P, const uint32_t* __restrict__ masks, uint32_t* __restrict__ img,
                                                                     uint32_t cap, int lgr)
{
    big_apply_body<K, true>(P, masks, img, cap, lgr);
}

// [G][N] words from the [G * S][ceil(N / S)] images big_apply_kernel leaves behind: bits rsh .. rsh + r - 1 of a site's word come from
// image `sub` of its group
__global__ __launch_bounds__(256) void big_merge_kernel(uint32_t* __restrict__ spins, const uint32_t* __restrict__ img, int N, int lgr)
{
    const uint32_t lgS = 5u - (uint32_t)lgr, S = 1u << lgS, r = 1u << lgr, rm = r == 32u ? 0xffffffffu : (1u << r) - 1u;
    const uint32_t nwords = ((uint32_t)N + S - 1u) >> lgS;
    const uint32_t site = blockIdx.x * 256u + threadIdx.x, grp = blockIdx.y;
    if (site >= (uint32_t)N) return;
    uint32_t w = 0u;
    for (uint32_t sub = 0; sub < S; ++sub)
        w |= ((img[((size_t)grp * S + sub) * nwords + (site >> lgS)] >> ((site & (S - 1u)) << lgr)) & rm) << (sub << lgr);
    spins[(size_t)grp * N + site] = w;
}

}  // namespace rrrmc
