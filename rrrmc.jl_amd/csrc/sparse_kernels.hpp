// gfx950 kernels for sparse +-J models (GraphRRG / GraphEA) under standardMC (src/RRRMC.jl:81-127).
//
// Data model (DESIGN.md has the long form):
//  * replicas are bit-sliced: one uint32 holds the spin bit of one site for the 32 replicas of a group
//    (bit b = replica 32*group + b).  One workgroup owns one group and keeps its whole state in LDS.
//  * all replicas attempt the same site at iteration g (SITE stream) with independent acceptance
//    uniforms (ACCEPT stream) — each replica is a faithful standardMC chain.
//  * the cached local field of the reference (LocalFields, src/Common.jl:27-36) is never stored:
//    delta_energy (src/graphs/RRG.jl:236-244) is recomputed from the K neighbour words with a
//    bit-sliced adder, which makes update_cache! (RRG.jl:191-234) a single XOR of the site's word.
//  * a chunk of <= C consecutive iterations is re-ordered into dependency levels by plan_kernel: two
//    attempts commute exactly unless one's site lies in the other's closed neighbourhood, so executing
//    level by level reproduces the sequential chain bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "philox.hpp"

namespace rrrmc {

constexpr int kWave = 64;
constexpr int kSweepThreads = 1024;              // 16 waves: wave 0 consumes, wave 1 tallies, waves 2..15 produce
constexpr int kFirstProducer = 2;
constexpr int kProducerWaves = kSweepThreads / kWave - kFirstProducer;
constexpr int kPlanThreads = 256;
constexpr int kMaxK = 7;                          // 3 bit planes for the unsatisfied-bond count
constexpr int kRows = 4;                          // rows of 64 slots the consumer keeps in flight
constexpr int kBatchSlots = kRows * kWave;
constexpr uint32_t kChunkSampleBefore = 1u;       // chunk flag: an energy sample is due before its first move

struct ChunkDesc {
    uint64_t g0;         // global iteration (1-based) of the chunk's first attempt
    uint32_t count;      // attempts in the chunk (<= C)
    uint32_t slot_base;  // offset of the chunk's slots / vector table in the plan buffers
    uint32_t nvec;       // number of consumer batches, written by plan_kernel
    uint32_t flags;
};

// ---------------------------------------------------------------------------------------------------
// plan_kernel: one workgroup per chunk.
//   slots[slot_base + p] = site | (t << 16)   attempts sorted by dependency level (t = index in chunk)
//   vecs [slot_base + b] = start | ((n-1) << 16)  consumer batches: runs of n <= 256 sorted slots inside one level
// Level rule: L(t) = 1 + max_{x in N[site_t]} W[x], W[site_t] = L(t), W = level of the last attempt AT x.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kPlanThreads) void plan_kernel(ChunkDesc* __restrict__ chunks, uint32_t* __restrict__ slots,
                                                            uint32_t* __restrict__ vecs, const int32_t* __restrict__ A,
                                                            int N, int K, int Cmax, uint32_t k0, uint32_t k1)
{
    extern __shared__ uint16_t plds[];
    uint16_t* s_site = plds;                   // [Cmax]
    uint16_t* s_lvl = s_site + Cmax;           // [Cmax]
    uint16_t* s_rank = s_lvl + Cmax;           // [Cmax]
    uint16_t* s_cnt = s_rank + Cmax;           // [Cmax + 2]  slots per level (levels are 1-based)
    uint16_t* s_start = s_cnt + Cmax + 2;      // [Cmax + 2]
    uint16_t* s_nb = s_start + Cmax + 2;       // [Cmax * K]
    uint16_t* s_W = s_nb + (size_t)Cmax * K;   // [N]

    const ChunkDesc cd = chunks[blockIdx.x];
    const int count = (int)cd.count;
    const int tid = threadIdx.x;

    for (int t = tid; t < count; t += kPlanThreads) {
        const uint32_t site = site_of(k0, k1, cd.g0 + (uint64_t)t, (uint32_t)N);
        s_site[t] = (uint16_t)site;
        for (int k = 0; k < K; ++k) s_nb[t * K + k] = (uint16_t)A[(size_t)site * K + k];
    }
    for (int x = tid; x < N; x += kPlanThreads) s_W[x] = 0;
    for (int l = tid; l < count + 2; l += kPlanThreads) s_cnt[l] = 0;
    __syncthreads();

    if (tid == 0) {
        uint32_t maxlvl = 0;
        for (int t = 0; t < count; ++t) {
            uint32_t l = s_W[s_site[t]];
            for (int k = 0; k < K; ++k) {
                const uint32_t w = s_W[s_nb[t * K + k]];
                l = w > l ? w : l;
            }
            l += 1;
            s_W[s_site[t]] = (uint16_t)l;
            s_lvl[t] = (uint16_t)l;
            s_rank[t] = s_cnt[l];
            s_cnt[l] = (uint16_t)(s_cnt[l] + 1);
            maxlvl = l > maxlvl ? l : maxlvl;
        }
        uint32_t pos = 0, nvec = 0;
        for (uint32_t l = 1; l <= maxlvl; ++l) {
            s_start[l] = (uint16_t)pos;
            uint32_t n = s_cnt[l], q = pos;
            while (n > 0) {      // the consumer works on batches of <= kBatchSlots slots that lie inside one level
                const uint32_t m = n < (uint32_t)kBatchSlots ? n : (uint32_t)kBatchSlots;
                vecs[cd.slot_base + nvec++] = q | ((m - 1u) << 16);
                q += m;
                n -= m;
            }
            pos += s_cnt[l];
        }
        chunks[blockIdx.x].nvec = nvec;
    }
    __syncthreads();
    for (int t = tid; t < count; t += kPlanThreads) {
        const uint32_t pos = (uint32_t)s_start[s_lvl[t]] + s_rank[t];
        slots[cd.slot_base + pos] = (uint32_t)s_site[t] | ((uint32_t)t << 16);
    }
}

// ---------------------------------------------------------------------------------------------------
// sweep_kernel: one workgroup = one group of 32 bit-sliced replicas, whole state in LDS.
//   wave 0      consumer : applies the moves of chunk c-1, level by level (the only state-dependent work)
//   wave 1      tally    : per-replica accepted-move / energy bookkeeping of chunk c-2, emits energy samples
//   waves 2..15 producers: acceptance masks + gather offsets (slot descriptors) of chunk c
// One barrier per chunk; descriptors and tally words are double-buffered in LDS.
// ---------------------------------------------------------------------------------------------------
struct SweepParams {
    uint32_t* spins;          // [G][N]   bit-sliced configuration
    const uint16_t* table;    // [N][TS]  BYTE offsets of the neighbour words in the LDS spin array:
                              //          4*y (J=+1) or 4*(y+N) (J=-1, complemented copy); TS = row stride
    const ChunkDesc* chunks;  // chunks of this launch
    const uint32_t* slots;
    const uint32_t* vecs;
    int32_t* Es;              // [nsamples][Rpad] energies (sample-major); may be null
    int32_t* E_cur;           // [Rpad] running energy of every replica
    int64_t* acc_cur;         // [Rpad] accepted moves of this sampling call
    uint64_t T[4];            // acceptance thresholds, class n = number of unsatisfied bonds (dE = 2(K-2n) > 0)
    uint32_t always_mask;     // bit n: class n is always accepted (exp(-beta dE) >= 1)
    uint32_t k0, k1;          // Philox key
    uint32_t group0;          // global id of this ctx's first group
    int64_t sample0;          // index of the first sample this launch may emit
    int N, C, TS, Rpad, nchunks;
    unsigned long long* stamps;   // diagnostic builds (-DRRRMC_STAMPS): [G][16] busy shader cycles per wave, else unused
};

template <int K> struct SweepCfg {
    static constexpr int NT = (K + 1) / 2;        // classes with dE > 0: n = 0 .. NT-1
    static constexpr int NW = NT + (K + 2) / 2;   // descriptor words: NT 32-bit masks, then K+1 16-bit byte offsets
                                                  // (own word first, then the K neighbour words), two per word
    static constexpr int NQ = (NW + 3) / 4;       // ... stored as NQ uint4 arrays [NQ][C]
    static constexpr int NS = K <= 1 ? 2 : (K <= 3 ? 3 : 4);   // tally streams: A and the planes of n
};

// ---- producers -------------------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ void produce_chunk(const SweepParams& P, const ChunkDesc& cd, uint4* __restrict__ desc,
                                              const uint16_t* __restrict__ tbl, int pw, int lane, uint32_t group)
{
    constexpr int NT = SweepCfg<K>::NT, NQ = SweepCfg<K>::NQ;
    const int C = P.C;
    const int ntask = ((int)cd.count + kWave - 1) / kWave;
    for (int task = pw; task < ntask; task += kProducerWaves) {
        const int p = task * kWave + lane;
        const bool live = p < (int)cd.count;
        uint32_t slot = 0;
        if (live) slot = P.slots[cd.slot_base + p];
        const uint32_t site = slot & 0xffffu;
        const uint64_t g = cd.g0 + (uint64_t)(slot >> 16);

        uint32_t f[NQ * 4];        // descriptor words
#pragma unroll
        for (int q = 0; q < NQ * 4; ++q) f[q] = 0u;
        {   // gather offsets: own word, then the K neighbour words (table row = K uint16 byte offsets)
            uint32_t off[K + 2];
            off[0] = site * 4u;
            off[K + 1] = 0u;
            const uint16_t* row = tbl + (size_t)site * P.TS;
            if constexpr (K <= 4) {
                const uint2 r = *reinterpret_cast<const uint2*>(row);
                const uint32_t e[4] = {r.x & 0xffffu, r.x >> 16, r.y & 0xffffu, r.y >> 16};
#pragma unroll
                for (int k = 0; k < K; ++k) off[1 + k] = e[k];
            } else {
                const uint4 r = *reinterpret_cast<const uint4*>(row);
                const uint32_t e[8] = {r.x & 0xffffu, r.x >> 16, r.y & 0xffffu, r.y >> 16, r.z & 0xffffu, r.z >> 16, r.w & 0xffffu, r.w >> 16};
#pragma unroll
                for (int k = 0; k < K; ++k) off[1 + k] = e[k];
            }
#pragma unroll
            for (int h = 0; h < (K + 2) / 2; ++h) f[NT + h] = off[2 * h] | (off[2 * h + 1] << 16);
        }

        uint32_t lt[NT], eq[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const bool always = (P.always_mask >> n) & 1u;
            lt[n] = always ? 0xffffffffu : 0u;
            eq[n] = (live && !always) ? 0xffffffffu : 0u;
        }
        // u < T_n, bit-sliced over the 32 replicas, most significant plane first, stopping as soon as every
        // lane of the wave is decided (lazy evaluation of a counter-based stream: the result does not depend on
        // where we stop).
        for (uint32_t pb = 0; pb < 16; ++pb) {
            uint32_t undecided = 0;
#pragma unroll
            for (int n = 0; n < NT; ++n) undecided |= eq[n];
            if (!__any(undecided != 0)) break;
            const Philox4 o = accept_planes(P.k0, P.k1, g, group, pb);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t w = o.w[j];
                const int sh = 63 - (int)(pb * 4 + j);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const uint32_t taum = 0u - (uint32_t)((P.T[n] >> sh) & 1ull);   // wave-uniform
                    const uint32_t z = w ^ ~taum;
                    const uint32_t e2 = eq[n] & z;
                    lt[n] |= (eq[n] ^ e2) & taum;
                    eq[n] = e2;
                }
            }
        }
        if (live) {
#pragma unroll
            for (int n = 0; n < NT; ++n) f[n] = lt[n];
#pragma unroll
            for (int q = 0; q < NQ; ++q) desc[q * C + p] = make_uint4(f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]);
        }
    }
}

// ---- consumer --------------------------------------------------------------------------------------
template <int K> struct SlotDesc { uint4 q[SweepCfg<K>::NQ]; };

template <int K>
__device__ __forceinline__ uint32_t desc_word(const SlotDesc<K>& d, int f)
{
    const uint4& v = d.q[f >> 2];
    return (f & 3) == 0 ? v.x : (f & 3) == 1 ? v.y : (f & 3) == 2 ? v.z : v.w;
}
// acceptance mask of class n
template <int K> __device__ __forceinline__ uint32_t desc_mask(const SlotDesc<K>& d, int n) { return desc_word<K>(d, n); }
// byte offset of word f of the attempt (0 = own word, 1..K = neighbour words)
template <int K> __device__ __forceinline__ uint32_t desc_off(const SlotDesc<K>& d, int f)
{
    const uint32_t w = desc_word<K>(d, SweepCfg<K>::NT + (f >> 1));
    return (f & 1) ? (w >> 16) : (w & 0xffffu);
}

__device__ __forceinline__ uint32_t lds_word(const uint32_t* sp, uint32_t byte_off)
{
    return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(sp) + byte_off);
}

// the K+1 spin words one attempt reads: own word first, then the neighbour words
template <int K> struct SlotWords { uint32_t s; uint32_t g[K]; };

template <int K>
__device__ __forceinline__ void gather_words(SlotWords<K>& w, const SlotDesc<K>& d, const uint32_t* __restrict__ sp)
{
    w.s = lds_word(sp, desc_off<K>(d, 0));
#pragma unroll
    for (int k = 0; k < K; ++k) w.g[k] = lds_word(sp, desc_off<K>(d, 1 + k));
}

// accept decision of one slot for the 32 replicas: planes n0..n2 of n = number of unsatisfied bonds, acc = accepted mask
template <int K>
__device__ __forceinline__ void slot_logic(const SlotDesc<K>& d, const SlotWords<K>& w, uint32_t& n0, uint32_t& n1, uint32_t& n2, uint32_t& acc)
{
    constexpr int NT = SweepCfg<K>::NT;
    const uint32_t s = w.s;
    n2 = 0u;
    if constexpr (K == 3) {
        // g_k = neighbour word, already XORed with the coupling sign (complemented copy for J = -1);
        // u_k = s ^ g_k is "bond k unsatisfied"; for odd K the planes of sum(u_k) are self-dual in s
        const uint32_t x12 = w.g[0] ^ w.g[1];
        n0 = x12 ^ w.g[2] ^ s;
        n1 = ((w.g[0] & w.g[1]) | (w.g[2] & x12)) ^ s;
        // n >= 2: dE <= 0, always accepted; n = 1 needs u < T_1; n = 0 needs u < T_0 (T_0 <= T_1: M_0 is a subset of M_1)
        acc = n1 | (desc_mask<K>(d, 1) & (n0 | desc_mask<K>(d, 0)));
    } else {
        n0 = 0u; n1 = 0u;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t u = w.g[k] ^ s;
            const uint32_t c0 = n0 & u;
            n0 ^= u;
            const uint32_t c1 = n1 & c0;
            n1 ^= c0;
            n2 ^= c1;
        }
        // classes n = 0..NT-1 have dE > 0 and need u < T_n; every other class is accepted (RRRMC.jl:39)
        uint32_t rej = 0u;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            uint32_t e = ~desc_mask<K>(d, n);
            e &= (n & 1) ? n0 : ~n0;
            e &= (n & 2) ? n1 : ~n1;
            if (K > 3) e &= (n & 4) ? n2 : ~n2;
            rej |= e;
        }
        acc = ~rej;
    }
}

// One dependency level at a time; inside a level the attempts commute, so up to kRows x 64 of them are in flight
// per step: all descriptor loads, then all gathers, then the logic and the stores (the single consumer wave is
// latency-bound, this is its instruction-level parallelism).  The code is branch-free inside a step so that the
// LDS waits can be counted (s_waitcnt lgkmcnt(N)) instead of drained: rows 0..NR-2 are full; in the last row the
// lanes past the end of the level re-read the level's last slot (a broadcast) and store to private dummy words.
template <int K, int NR>
__device__ __forceinline__ void consume_rows(const uint4* __restrict__ desc, uint32_t* __restrict__ sp, uint4* __restrict__ tal,
                                             int C, int N, int p0, int plast, int lane)
{
    constexpr int NQ = SweepCfg<K>::NQ;
    SlotDesc<K> d[NR];
    SlotWords<K> w[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        int p = p0 + j * kWave;
        if (j == NR - 1) p = p < plast ? p : plast;      // only the last live row can be partial
#pragma unroll
        for (int q = 0; q < NQ; ++q) d[j].q[q] = desc[q * C + p];
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) gather_words<K>(w[j], d[j], sp);
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        uint32_t n0, n1, n2, acc;
        slot_logic<K>(d[j], w[j], n0, n1, n2, acc);
        const uint32_t snew = w[j].s ^ acc;      // spinflip! + update_cache! (Interface.jl:89-92, RRG.jl:191-234)
        uint32_t oa = desc_off<K>(d[j], 0);      // byte offset of the site's word; its complement lives N words further
        uint32_t ob = oa + 4u * (uint32_t)N;
        int pt = p0 + j * kWave;
        if (j == NR - 1) {
            const bool dead = pt > plast;
            oa = dead ? 8u * (uint32_t)N + 4u * (uint32_t)lane : oa;            // dummy words behind the two spin copies
            ob = dead ? 8u * (uint32_t)N + 256u + 4u * (uint32_t)lane : ob;
            pt = dead ? C + lane : pt;                                          // dummy tally entries behind the chunk's
        }
        *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(sp) + oa) = snew;
        *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(sp) + ob) = ~snew;
        tal[pt] = make_uint4(acc, n0, n1, n2);
    }
}

// The planner has already cut every dependency level into batches of <= kRows x 64 slots.  Inside a batch the
// attempts commute, so all its descriptor loads, then all its gathers are issued before the logic and the stores
// (the single consumer wave is latency-bound: this is its instruction-level parallelism).  A batch is branch-free
// so that the LDS waits can be counted instead of drained: rows 0..NR-2 are full; in the last row the lanes past
// the end of the batch re-read its last slot (a broadcast) and store to private dummy words.
template <int K>
__device__ __forceinline__ void consume_chunk(const SweepParams& P, const ChunkDesc& cd, const uint4* __restrict__ desc,
                                              uint32_t* __restrict__ sp, uint4* __restrict__ tal, int lane)
{
    const int C = P.C, N = P.N;
    const uint32_t nb = cd.nvec;
    for (uint32_t v0 = 0; v0 < nb; v0 += kWave) {
        // the batch table is read 64 entries at a time (one coalesced load) and broadcast lane by lane
        const uint32_t myvd = (v0 + lane < nb) ? P.vecs[cd.slot_base + v0 + lane] : 0u;
        const uint32_t vn = (nb - v0) < (uint32_t)kWave ? (nb - v0) : (uint32_t)kWave;
        for (uint32_t v = 0; v < vn; ++v) {
            const uint32_t vd = __builtin_amdgcn_readlane(myvd, v);
            const int start = (int)(vd & 0xffffu), cm1 = (int)(vd >> 16);   // cm1 = slots - 1
            const int p0 = start + lane, plast = start + cm1;
            if (cm1 >= 3 * kWave) consume_rows<K, 4>(desc, sp, tal, C, N, p0, plast, lane);
            else if (cm1 >= 2 * kWave) consume_rows<K, 3>(desc, sp, tal, C, N, p0, plast, lane);
            else if (cm1 >= kWave) consume_rows<K, 2>(desc, sp, tal, C, N, p0, plast, lane);
            else consume_rows<K, 1>(desc, sp, tal, C, N, p0, plast, lane);
        }
    }
}

// ---- tally wave ------------------------------------------------------------------------------------
// Per replica we need A = number of accepted moves and S = sum over accepted moves of n (unsatisfied bonds before
// the flip): dE = 2(K - 2n) (RRG.jl:236-244), so E += 2(K*A - 2*S).  The consumer leaves, per slot, the words
// (acc, n0, n1, n2); lane l of the tally wave owns the slots p = l, l+64, ... and counts the bit columns of
// acc, acc&n0, acc&n1, acc&n2 with bit-sliced carry-save counters (Harley-Seal): eight words at a time go through
// a fixed adder tree into the planes ones/twos/fours, the tree's carry ("eights") feeds a small streaming
// counter.  A flush transposes every plane (32x32 bit butterfly with ds_swizzle) so that lane r holds replica
// r's column, and popcounts it.
constexpr int kTallyHi = 4;        // streaming levels above the tree: weights 8, 16, 32, 64 -> up to 127 inputs per lane

template <int NS> struct TallyState {
    uint32_t lo[NS][3];            // ones, twos, fours
    uint32_t S[NS][kTallyHi];
    uint32_t Pd[NS][kTallyHi];
    uint32_t ngrp;                 // groups of 8 inputs since the last flush (wave-uniform)
};

#define RRRMC_CSA(sum, carry, a, b, c)                      \
    {                                                       \
        const uint32_t u_ = (a) ^ (b);                      \
        carry = ((a) & (b)) | (u_ & (c));                   \
        sum = u_ ^ (c);                                     \
    }

// streaming add of one word of weight 8 when the group count has exactly TZ trailing one bits
template <int NS, int TZ>
__device__ __forceinline__ void tally_hi_tz(TallyState<NS>& t, const uint32_t (&x)[NS])
{
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        uint32_t v = x[s];
#pragma unroll
        for (int l = 0; l < TZ; ++l) {
            uint32_t sum, carry;
            RRRMC_CSA(sum, carry, t.S[s][l], t.Pd[s][l], v)
            t.S[s][l] = sum;
            v = carry;
        }
        t.Pd[s][TZ] = v;
    }
}

// eight input words per stream
template <int NS>
__device__ __forceinline__ void tally_add8(TallyState<NS>& t, const uint32_t (&x)[8][NS])
{
    uint32_t e8[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        uint32_t ones = t.lo[s][0], twos = t.lo[s][1], fours = t.lo[s][2];
        uint32_t t2a, t2b, t4a, t4b;
        RRRMC_CSA(ones, t2a, ones, x[0][s], x[1][s])
        RRRMC_CSA(ones, t2b, ones, x[2][s], x[3][s])
        RRRMC_CSA(twos, t4a, twos, t2a, t2b)
        RRRMC_CSA(ones, t2a, ones, x[4][s], x[5][s])
        RRRMC_CSA(ones, t2b, ones, x[6][s], x[7][s])
        RRRMC_CSA(twos, t4b, twos, t2a, t2b)
        RRRMC_CSA(fours, e8[s], fours, t4a, t4b)
        t.lo[s][0] = ones; t.lo[s][1] = twos; t.lo[s][2] = fours;
    }
    const uint32_t ng = __builtin_amdgcn_readfirstlane(t.ngrp);
    switch (__builtin_ctz(~ng)) {       // wave-uniform
        case 0: tally_hi_tz<NS, 0>(t, e8); break;
        case 1: tally_hi_tz<NS, 1>(t, e8); break;
        case 2: tally_hi_tz<NS, 2>(t, e8); break;
        default: tally_hi_tz<NS, 3>(t, e8); break;
    }
    t.ngrp = ng + 1u;
}

// 32x32 bit-matrix transpose across the 32 lanes of each wave half: on return bit i of lane r is bit r of lane i.
__device__ __forceinline__ uint32_t transpose32(uint32_t a, int lane)
{
#define RRRMC_TSTAGE(J, M)                                                                           \
    {                                                                                                \
        const uint32_t o = (uint32_t)__builtin_amdgcn_ds_swizzle((int)a, ((J) << 10) | 0x1f);        \
        const bool hi = (lane & (J)) != 0;                                                           \
        const uint32_t keep = hi ? ~(M) : (M);                                                       \
        const uint32_t sel = hi ? (o >> (J)) : (o << (J));                                           \
        a = (a & keep) | (sel & ~keep);                                                              \
    }
    RRRMC_TSTAGE(16, 0x0000ffffu)
    RRRMC_TSTAGE(8, 0x00ff00ffu)
    RRRMC_TSTAGE(4, 0x0f0f0f0fu)
    RRRMC_TSTAGE(2, 0x33333333u)
    RRRMC_TSTAGE(1, 0x55555555u)
#undef RRRMC_TSTAGE
    return a;
}

// Fold the counters into per-replica integers: totA / totS of replica (lane & 31), summed over both wave halves.
// Branch-free over the planes (3 tree planes + kTallyHi x {S, Pd}) so that the independent transposes overlap.
template <int NS>
__device__ __forceinline__ void tally_flush(TallyState<NS>& t, int lane, uint32_t& totA, uint32_t& totS)
{
    uint32_t tot[2] = {0u, 0u};
    const uint32_t ng = __builtin_amdgcn_readfirstlane(t.ngrp);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int wsh = s == 0 ? 0 : s - 1;          // stream 0 counts A; streams 1.. are the planes n0, n1, n2 of S
        uint32_t c = 0u;
#pragma unroll
        for (int l = 0; l < 3; ++l) { c += (uint32_t)__popc(transpose32(t.lo[s][l], lane)) << l; t.lo[s][l] = 0u; }
#pragma unroll
        for (int l = 0; l < kTallyHi; ++l) {
            const uint32_t pend = ((ng >> l) & 1u) ? t.Pd[s][l] : 0u;     // pending word is live iff bit l of the group count
            c += ((uint32_t)__popc(transpose32(t.S[s][l], lane)) + (uint32_t)__popc(transpose32(pend, lane))) << (3 + l);
            t.S[s][l] = 0u;
        }
        tot[s == 0 ? 0 : 1] += c << wsh;
    }
    t.ngrp = 0u;
    totA = tot[0] + (uint32_t)__shfl_xor((int)tot[0], 32);
    totS = tot[1] + (uint32_t)__shfl_xor((int)tot[1], 32);
}

template <int NS>
__device__ __forceinline__ void tally_chunk(TallyState<NS>& t, const ChunkDesc& cd, const uint4* __restrict__ tal, int lane)
{
    // every lane adds the same number of words (zeros past the end) so that the counters' shape stays wave-uniform
    for (int base = 0; base < (int)cd.count; base += 8 * kWave) {
        uint32_t x[8][NS];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = base + j * kWave + lane;
            const uint4 w = p < (int)cd.count ? tal[p] : make_uint4(0u, 0u, 0u, 0u);
            x[j][0] = w.x;
            if (NS > 1) x[j][1 % NS] = w.x & w.y;
            if (NS > 2) x[j][2 % NS] = w.x & w.z;
            if (NS > 3) x[j][3 % NS] = w.x & w.w;
        }
        tally_add8<NS>(t, x);
    }
}

template <int K>
__global__ __launch_bounds__(kSweepThreads) void sweep_kernel(SweepParams P)
{
    constexpr int NQ = SweepCfg<K>::NQ, NS = SweepCfg<K>::NS;
    extern __shared__ uint32_t lds[];
    const int N = P.N, C = P.C;
    uint32_t* sp = lds;                                              // [2N]  words, then complements
    uint4* desc = reinterpret_cast<uint4*>(sp + ((2 * N + 128 + 3) & ~3)); // [2][NQ][C]   (128 dummy words behind the spins)
    uint4* tal = desc + 2 * NQ * C;                                  // [2][C + 64]  (64 dummy entries per buffer)
    uint16_t* tbl = reinterpret_cast<uint16_t*>(tal + 2 * (C + kWave));   // [N][TS]

    const int tid = threadIdx.x, lane = tid & 63;
    // the wave index as a SCALAR: role dispatch becomes s_cbranch (and s_setprio below really is per wave)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t group = P.group0 + blockIdx.x;
    uint32_t* gsp = P.spins + (size_t)blockIdx.x * N;

    for (int x = tid; x < N; x += kSweepThreads) {
        const uint32_t w = gsp[x];
        sp[x] = w;
        sp[x + N] = ~w;
    }
    for (int q = tid; q < N * P.TS; q += kSweepThreads) tbl[q] = P.table[q];
    __syncthreads();

    // Software pipeline over chunks, one workgroup barrier per step:  produce(c) | consume(c-1) | tally(c-2).
    // Each role runs its OWN loop (the branch on the scalar wave index is outside the loops), so that a role's
    // registers are not live across the other roles' code; every wave executes exactly nsteps barriers.
    const int nsteps = P.nchunks + 2;
    const int tal_stride = C + kWave;
#ifdef RRRMC_STAMPS
    unsigned long long busy = 0;
#define RRRMC_T0 const unsigned long long t_in = __builtin_amdgcn_s_memtime();
#define RRRMC_T1 busy += __builtin_amdgcn_s_memtime() - t_in;
#else
#define RRRMC_T0
#define RRRMC_T1
#endif
    if (wave == 0) {
        // consumer: single latency-bound wave on the critical path -> let it win the SIMD's issue arbitration
        __builtin_amdgcn_s_setprio(3);
        for (int c = 0; c < nsteps; ++c) {
            RRRMC_T0
#ifndef RRRMC_ABLATE_CONSUME      // timing experiments only (tools/ablate.sh): results are wrong with a role removed
            if (c >= 1 && c - 1 < P.nchunks)
                consume_chunk<K>(P, P.chunks[c - 1], desc + ((c - 1) & 1) * NQ * C, sp, tal + ((c - 1) & 1) * tal_stride, lane);
#endif
            RRRMC_T1
            __syncthreads();
        }
    } else if (wave == 1) {
        // tally: lanes 0..31 own the running energy / accepted count of replica `lane`
        __builtin_amdgcn_s_setprio(2);
        TallyState<NS> ts;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int l = 0; l < 3; ++l) ts.lo[s][l] = 0u;
#pragma unroll
            for (int l = 0; l < kTallyHi; ++l) { ts.S[s][l] = 0u; ts.Pd[s][l] = 0u; }
        }
        ts.ngrp = 0u;
        int32_t E_run = 0;
        int64_t A_run = 0;
        if (lane < 32) { E_run = P.E_cur[blockIdx.x * 32 + lane]; A_run = P.acc_cur[blockIdx.x * 32 + lane]; }
        int64_t ns = P.sample0;
        const uint32_t grp_per_chunk = (uint32_t)((C + 8 * kWave - 1) / (8 * kWave));   // 8-word groups one chunk adds per lane
        for (int c = 0; c < nsteps; ++c) {
            RRRMC_T0
#ifndef RRRMC_ABLATE_TALLY
            if (c >= 2) {
                const ChunkDesc cd = P.chunks[c - 2];
                // an energy sample is due BEFORE this chunk's moves (RRRMC.jl:104-108)
                const bool sample = (cd.flags & kChunkSampleBefore) != 0;
                if (sample || ts.ngrp + grp_per_chunk > (1u << kTallyHi) - 1u) {
                    uint32_t a, sn;
                    tally_flush<NS>(ts, lane, a, sn);
                    E_run += 2 * (K * (int32_t)a - 2 * (int32_t)sn);
                    A_run += a;
                }
                if (sample) {
                    if (lane < 32 && P.Es) P.Es[ns * P.Rpad + blockIdx.x * 32 + lane] = E_run;
                    ns += 1;
                }
                tally_chunk<NS>(ts, cd, tal + ((c - 2) & 1) * tal_stride, lane);
            }
#endif
            RRRMC_T1
            __syncthreads();
        }
        uint32_t a, sn;
        tally_flush<NS>(ts, lane, a, sn);
        E_run += 2 * (K * (int32_t)a - 2 * (int32_t)sn);
        A_run += a;
        if (lane < 32) { P.E_cur[blockIdx.x * 32 + lane] = E_run; P.acc_cur[blockIdx.x * 32 + lane] = A_run; }
    } else {
        for (int c = 0; c < nsteps; ++c) {
            RRRMC_T0
#ifndef RRRMC_ABLATE_PRODUCE
            if (c < P.nchunks) produce_chunk<K>(P, P.chunks[c], desc + (c & 1) * NQ * C, tbl, wave - kFirstProducer, lane, group);
#endif
            RRRMC_T1
            __syncthreads();
        }
    }
#ifdef RRRMC_STAMPS
    if (lane == 0 && P.stamps) P.stamps[blockIdx.x * 16 + wave] = busy;
#endif
#undef RRRMC_T0
#undef RRRMC_T1
    __syncthreads();
    for (int x = tid; x < N; x += kSweepThreads) gsp[x] = sp[x];
}

// ---------------------------------------------------------------------------------------------------
// energy / cache view (setup, not hot): thread = (replica bit, site part).
//   U[r]      = sum over sites of the number of unsatisfied bonds (each bond seen from both ends)
//   nun[r][x] = unsatisfied bonds of site x (optional): lfields[x] = 4 n - 2K (RRG.jl:164-189, A.1 of SURVEY)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void energy_kernel(const uint32_t* __restrict__ spins, const int32_t* __restrict__ A,
                                                     const int8_t* __restrict__ J, int N, int K, int32_t* __restrict__ E_out,
                                                     uint8_t* __restrict__ nun)
{
    __shared__ uint32_t sU[32];
    const int tid = threadIdx.x, r = tid & 31, part = tid >> 5;
    const uint32_t* sp = spins + (size_t)blockIdx.x * N;
    if (tid < 32) sU[tid] = 0;
    __syncthreads();
    uint32_t U = 0;
    for (int x = part; x < N; x += 8) {
        const uint32_t s = (sp[x] >> r) & 1u;
        uint32_t n = 0;
        for (int k = 0; k < K; ++k) {
            const uint32_t sy = (sp[A[(size_t)x * K + k]] >> r) & 1u;
            n += s ^ sy ^ (J[(size_t)x * K + k] < 0 ? 1u : 0u);
        }
        U += n;
        if (nun) nun[((size_t)blockIdx.x * 32 + r) * N + x] = (uint8_t)n;
    }
    atomicAdd(&sU[r], U);
    __syncthreads();
    // E = -(sat - unsat) = 2*unsat - NK/2, unsat = U/2
    if (tid < 32) E_out[blockIdx.x * 32 + tid] = (int32_t)sU[tid] - (N * K) / 2;
}

__global__ __launch_bounds__(256) void init_spins_kernel(uint32_t* __restrict__ spins, int N, uint32_t group0, uint32_t k0, uint32_t k1)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x < N) spins[(size_t)blockIdx.y * N + x] = init_spin_word(k0, k1, group0 + blockIdx.y, (uint64_t)x);
}

// Es[sample][Rpad] int32 -> out[replica][nsamples] int64 (the layout rrrmc_fetch_results hands to the caller)
__global__ __launch_bounds__(256) void transpose_es_kernel(const int32_t* __restrict__ Es, int64_t* __restrict__ out, int64_t nsamp, int Rpad, int R)
{
    __shared__ int32_t tile[32][33];
    const int64_t s0 = (int64_t)blockIdx.x * 32;
    const int r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int64_t s = s0 + j;
        tile[j][tx] = (s < nsamp) ? Es[s * Rpad + r0 + tx] : 0;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j;
        const int64_t s = s0 + tx;
        if (r < R && s < nsamp) out[(int64_t)r * nsamp + s] = (int64_t)tile[tx][j];
    }
}

}  // namespace rrrmc
