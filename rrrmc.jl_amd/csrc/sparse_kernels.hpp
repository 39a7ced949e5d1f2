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
#include "host_plan.hpp"

namespace rrrmc {

constexpr int kWave = 64;
#ifndef RRRMC_SWEEP_THREADS
#define RRRMC_SWEEP_THREADS 1024
#endif
constexpr int kSweepThreads = RRRMC_SWEEP_THREADS;  // 16 waves: consumer, tally, fixer and 13 producers (896 / 960: 11 / 12 producers, timing experiments)
#ifndef RRRMC_FIXER_WAVE
#define RRRMC_FIXER_WAVE 14
#endif
// waves w, w+4, w+8, w+12 share a SIMD, and a step lasts as long as its most loaded SIMD.  The producers' 64-slot tasks are dealt
// round-robin to the producer waves in wave order; with the fixer on wave 14 the SIMDs get 5, 5, 6 (+ the fix task), 6 of config 2's 22
// tasks per chunk next to the consumer (SIMD 0) and the tally wave (SIMD 1).  Merging the fix task into a producer wave was slower
// (it is a serial chain of two or three Philox blocks: as long as a producer task).
#ifndef RRRMC_TALLY_WAVE
#define RRRMC_TALLY_WAVE 1
#endif
constexpr int kConsumerWave = 0, kTallyWave = 1, kFixerWave = RRRMC_FIXER_WAVE;       // spf_fast_kernel keeps (0, 1, kFastFixerWave)
constexpr int kSwTallyWave = RRRMC_TALLY_WAVE;
// producer index of a wave of sweep_kernel: its rank among the waves that are neither consumer, tally nor fixer
__host__ __device__ constexpr int sw_producer_index(int wave)
{
    int pw = 0;
    for (int w = 0; w < wave; ++w) pw += (w != kConsumerWave && w != kSwTallyWave && w != kFixerWave) ? 1 : 0;
    return pw;
}
constexpr int kProducerWaves = kSweepThreads / kWave - 3;
constexpr int kMaxChunkSlots = 2048;              // longest chunk the host may choose (rrrmc_hip.hip: kMaxChunk)
constexpr int kProducerTasksMax = (kMaxChunkSlots / kWave + kProducerWaves - 1) / kProducerWaves;   // 64-slot tasks per producer wave and chunk
constexpr int kProducerBlocks = 3;               // Philox blocks (4 bit planes each) every producer lane computes (spf_fast_kernel)
constexpr int kLeftMax = 128;                    // capacity of the per-chunk list of slots still undecided after that (spf_fast_kernel)
// sweep_kernel: every lane computes kSwBlocks blocks; the slots still open go through a list to 64-slot "fix tasks" (kSwFixBlocks more
// blocks for every lane, then a loop), dealt to the fixer wave (tasks 0, 1) and, when the list can hold more than two tasks, to the
// producer waves with the fewest tasks of their own.  Measured at config 2: (3, 0, 128) 8.78 ms per launch; (2, 2, 384) — a third less
// Philox work in the producers — 9.52 ms: a producer task only gets 12 % shorter (its list appends are no longer rare) and a fix task
// costs more than a producer task, so three blocks per lane and a nearly idle fixer wave stay.
constexpr int kSwBlocks = 3;
constexpr int kSwFixBlocks = 0;
constexpr int kSwLeftMax = 128;                  // list capacity per chunk; lanes that find it full finish on their own
#ifndef RRRMC_PLAN_THREADS
#define RRRMC_PLAN_THREADS 512
#endif
constexpr int kPlanThreads = RRRMC_PLAN_THREADS;
constexpr int kMaxK = 7;                          // 3 bit planes for the unsatisfied-bond count
// rows of 64 slots the consumer keeps in flight = the longest batch the planner cuts a dependency level into.  A batch costs the
// consumer about 200 cycles whatever it holds (a dozen scalar instructions and two LDS round trips), so the first, long levels of a
// chunk should be ONE batch each: as many rows as the register file takes without spilling (8 VGPRs per row at K = 3; measured at
// config 2: 4 rows 8.99, 8 rows 8.73, 12 rows 8.46, 14 rows 8.34 ms per launch; 15 rows spill)
#ifndef RRRMC_ROWS
#define RRRMC_ROWS 14
#endif
// (K = 4: 8 rows, no change against 4; K >= 5: 6 rows measured 2 % slower than 4 — the rows are 15 VGPRs each there)
template <int K> constexpr int sweep_rows() { return K <= 3 ? RRRMC_ROWS : (K == 4 ? (RRRMC_ROWS < 8 ? RRRMC_ROWS : 8) : (RRRMC_ROWS < 4 ? RRRMC_ROWS : 4)); }
// (ChunkDesc, kChunkSampleBefore: host_plan.hpp — HIP-free, shared with the host's chunk planner and its sanitizer build)

// ---------------------------------------------------------------------------------------------------
// plan_kernel<K>: one workgroup per chunk (state-independent: the plan is shared by every replica group).
//   slots[slot_base + p] = site | (t << 16)       attempts sorted by dependency level (t = index in chunk)
//   vecs [slot_base + b] = start | ((n-1) << 16)  consumer batches: runs of n <= 256 sorted slots inside one level
// Level rule: L(t) = 1 + max_{x in N[site_t]} W_t[x], W_t[x] = level of the last attempt AT x before t (0 if none).
// Computed without a sequential pass: the attempts are bucketed by site (counting sort), every attempt looks up, for each
// site x of its closed neighbourhood, the latest earlier attempt at x (its predecessors), and the levels are the longest-path
// depths of that DAG, reached by relaxing L(t) = 1 + max L(pred) until nothing changes (as many rounds as there are levels,
// about ten).  The order of the slots INSIDE a level is arbitrary (they commute), so it is left to LDS atomics.
// ---------------------------------------------------------------------------------------------------
// counter[key] += 1 for every active lane, one LDS atomic per DISTINCT key of the wavefront (a level of a chunk holds up to half of
// its attempts: 64 lanes adding to one LDS word serialise in the LDS unit); returns the lane's slot = old value + its rank among the
// lanes with the same key.  All lanes of the wavefront must call it together.
__device__ __forceinline__ uint32_t wave_count_by_key(uint32_t* counter, uint32_t key, bool active)
{
    uint32_t slot = 0u;
    unsigned long long todo = __ballot(active);
    while (todo) {                                                            // wave-uniform
        const int leader = __builtin_ctzll(todo);
        const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, leader);
        const unsigned long long same = __ballot(active && key == k0);
        uint32_t base = 0u;
        if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(&counter[k0], (uint32_t)__popcll(same));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
        if (active && key == k0) slot = base + (uint32_t)__popcll(same & ((1ull << (threadIdx.x & 63)) - 1ull));
        todo &= ~same;
    }
    return slot;
}

constexpr uint32_t kNoPred = 0xffffu;

template <int K>
__global__ __launch_bounds__(kPlanThreads) void plan_kernel(ChunkDesc* __restrict__ chunks, uint32_t* __restrict__ slots,
                                                            uint32_t* __restrict__ vecs, const int32_t* __restrict__ A,
                                                            int N, int Cmax, uint32_t k0, uint32_t k1, uint64_t gbase, int batch_slots)
{
    extern __shared__ uint32_t plds32[];
    // two phases share the first region: the site buckets (until the predecessors are known), then the level counters
    const int nshare = (N + 1) > 3 * (Cmax + 2) ? (N + 1) : 3 * (Cmax + 2);
    uint32_t* s_bin = plds32;                                   // [N + 1]     attempts per site -> bucket END offsets (bin x = [end(x-1), end(x)))
    uint32_t* s_cnt = plds32;                                   // [Cmax + 2]  slots per level (levels are 1-based)
    uint32_t* s_start = s_cnt + Cmax + 2;                       // [Cmax + 2]
    uint32_t* s_cur = s_start + Cmax + 2;                       // [Cmax + 2]
    uint16_t* s_site = reinterpret_cast<uint16_t*>(plds32 + nshare); // [Cmax]
    uint16_t* s_lvl = s_site + Cmax;                            // [Cmax]
    uint16_t* s_nb = s_lvl + Cmax;                              // [Cmax * K]
    uint16_t* s_item = s_nb + (size_t)Cmax * K;                 // [Cmax]      attempt ids bucketed by site
    uint16_t* s_pred = s_item + Cmax;                           // [Cmax * (K + 1)]
    __shared__ uint32_t s_maxlvl, s_changed, s_part[kPlanThreads / 64];

    const ChunkDesc cd = chunks[blockIdx.x];
    const int count = (int)cd.count;
    const int tid = threadIdx.x;

    for (int x = tid; x <= N; x += kPlanThreads) s_bin[x] = 0u;
    __syncthreads();
    for (int t = tid; t < count; t += kPlanThreads) {
        const uint32_t site = site_of(k0, k1, gbase + cd.g0 + (uint64_t)t, (uint32_t)N);
        s_site[t] = (uint16_t)site;
        s_lvl[t] = 1;
#pragma unroll
        for (int k = 0; k < K; ++k) s_nb[t * K + k] = (uint16_t)A[(size_t)site * K + k];
        atomicAdd(&s_bin[site], 1u);
    }
    __syncthreads();
    // exclusive scan of the site histogram: each thread owns a contiguous range of sites; the 256 partial sums are scanned with
    // wave shuffles (a serial pass over them would cost more than the rest of the kernel)
    {
        const int per = (N + kPlanThreads - 1) / kPlanThreads, x0 = tid * per, x1 = (x0 + per < N) ? x0 + per : N;
        uint32_t sum = 0u;
        for (int x = x0; x < x1; ++x) sum += s_bin[x];
        uint32_t incl = sum;
        const int ln = tid & 63, wv = tid >> 6;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t n = (uint32_t)__shfl_up((int)incl, o);
            if (ln >= o) incl += n;
        }
        if (ln == 63) s_part[wv] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (int q = 0; q < wv; ++q) run += s_part[q];
        for (int x = x0; x < x1; ++x) { const uint32_t v = s_bin[x]; s_bin[x] = run; run += v; }      // start offsets ...
    }
    __syncthreads();
    // ... used as fill cursors: afterwards s_bin[x] is the END of bucket x and its start is the end of bucket x-1
    for (int t = tid; t < count; t += kPlanThreads) s_item[atomicAdd(&s_bin[s_site[t]], 1u)] = (uint16_t)t;
    __syncthreads();
    // predecessors: for every site x of the closed neighbourhood, the latest earlier attempt at x
    for (int q = tid; q < count * (K + 1); q += kPlanThreads) {
        const int t = q / (K + 1), j = q - t * (K + 1);
        const uint32_t x = j == 0 ? (uint32_t)s_site[t] : (uint32_t)s_nb[t * K + j - 1];
        uint32_t best = kNoPred;
        for (uint32_t b = x ? s_bin[x - 1] : 0u; b < s_bin[x]; ++b) {
            const uint32_t tp = s_item[b];
            if (tp < (uint32_t)t && (best == kNoPred || tp > best)) best = tp;
        }
        s_pred[q] = (uint16_t)best;
    }
    __syncthreads();
    for (int l = tid; l < count + 2; l += kPlanThreads) { s_cnt[l] = 0u; s_cur[l] = 0u; }      // the buckets are dead: level counters
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
    {
        uint32_t mymax = 0u;
        for (int t0 = 0; t0 < count; t0 += kPlanThreads) {        // uniform trip count: the aggregation is a wave-wide operation
            const int t = t0 + tid;
            const bool act = t < count;
            const uint32_t l = act ? (uint32_t)s_lvl[t] : 0u;
            wave_count_by_key(s_cnt, l, act);
            mymax = l > mymax ? l : mymax;
        }
        for (int o = 32; o > 0; o >>= 1) { const uint32_t m = (uint32_t)__shfl_xor((int)mymax, o); mymax = m > mymax ? m : mymax; }
        if ((tid & 63) == 0) atomicMax(&s_maxlvl, mymax);
    }
    __syncthreads();
    if (tid == 0) {
        const uint32_t maxlvl = s_maxlvl;
        uint32_t pos = 0, nvec = 0;
        for (uint32_t l = 1; l <= maxlvl; ++l) {
            s_start[l] = pos;
            uint32_t n = s_cnt[l], q = pos;
            while (n > 0) {      // the consumer works on batches of <= batch_slots slots that lie inside one level
                const uint32_t m = n < (uint32_t)batch_slots ? n : (uint32_t)batch_slots;      // the longest batch the consuming kernel takes
                vecs[cd.slot_base + nvec++] = q | ((m - 1u) << 16);
                q += m;
                n -= m;
            }
            pos += s_cnt[l];
        }
        chunks[blockIdx.x].nvec = nvec;
    }
    __syncthreads();
    for (int t0 = 0; t0 < count; t0 += kPlanThreads) {
        const int t = t0 + tid;
        const bool act = t < count;
        const uint32_t l = act ? (uint32_t)s_lvl[t] : 0u;
        const uint32_t pos = wave_count_by_key(s_cur, l, act);
        if (act) slots[cd.slot_base + s_start[l] + pos] = (uint32_t)s_site[t] | ((uint32_t)t << 16);
    }
}

// ---------------------------------------------------------------------------------------------------
// sweep_kernel: one workgroup = one group of 32 bit-sliced replicas, whole state in LDS.
//   wave 0      consumer : applies the moves of chunk c-2, level by level (the only state-dependent work)
//   wave 1      tally    : per-replica accepted-move / energy bookkeeping of chunk c-3, emits energy samples
//   fixer wave           : finishes the slots of chunk c-1 its producers left open
//   the other 13 waves   : producers: acceptance masks + gather offsets (slot descriptors) of chunk c
// One barrier per chunk; descriptors and tally words are double-buffered in LDS.
// ---------------------------------------------------------------------------------------------------
struct SweepParams {
    uint32_t* spins;          // [G][N]   bit-sliced configuration
    const uint16_t* table;    // [N][TS]  offsets of the neighbour words in the LDS spin array: y (J=+1) or y+N (J=-1, complemented
                              //          copy), as BYTE offsets (x4) in MODE 0 / 1 and as WORD indices in MODE 2
                              //          (8192 < N < 32768: byte offsets no longer fit 16 bits; the table then stays in HBM/L2)
    const ChunkDesc* chunks;  // chunks of this launch
    const uint32_t* slots;
    const uint32_t* vecs;
    int32_t* Es;              // [nsamples][Rpad] energies (sample-major); may be null
    int32_t* E_cur;           // [Rpad] running energy of every replica
    int64_t* acc_cur;         // [Rpad] accepted moves of this sampling call
    uint32_t taum[64 * 4];    // acceptance thresholds T_n as bit planes: [plane][n] = bit (63 - plane) of T_n as 0 / ~0;
                              // class n = number of unsatisfied bonds (dE = 2(K-2n) > 0)
    uint32_t always_mask;     // bit n: class n is always accepted (exp(-beta dE) >= 1)
    uint32_t k0, k1;          // Philox key
    uint32_t group0;          // global id of this ctx's first group
    int64_t sample0;          // index of the first sample this launch may emit
    uint64_t gbase;           // iterations done before this sampling call (ChunkDesc::g0 is relative to it)
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

// list of the slots of one chunk that the producers could not decide within kProducerBlocks blocks
struct LeftList {
    uint32_t* count;     // [1]
    uint32_t* slot;      // [cap]      position in the chunk | (index of the attempt in the chunk << 16)
    uint32_t* lt;        // [NT][cap]  masks so far
    uint32_t* eq;        // [NT][cap]  replicas still undecided
};

template <int NT> __device__ __forceinline__ bool any_set(const uint32_t (&eq)[NT])
{
    uint32_t u = 0u;
#pragma unroll
    for (int n = 0; n < NT; ++n) u |= eq[n];
    return u != 0u;
}

// four more bit planes (block pb of the ACCEPT stream) of the comparison u < T_n for the 32 replicas of a word.
// taum[plane * 4 + n] is bit `plane` of T_n (plane 0 = MSB) spread to a full word (0 or ~0), precomputed by the
// host: the index is wave-uniform, so the words are scalar loads and every VALU op below has a scalar operand.
//   replicas still equal keep being equal where the u bit equals the threshold bit; those that leave with a u bit
//   of 0 against a threshold bit of 1 are below T.
template <int NT>
__device__ __forceinline__ void refine_block(uint32_t (&lt)[NT], uint32_t (&eq)[NT], const Philox4& o, uint32_t pb, const uint32_t* __restrict__ taum)
{
    // two three-input functions of (eq, w, tm) per plane and threshold — left to itself the compiler spends 3.5 instructions on
    // the pair, spelled out as v_bitop3_b32 truth tables it is 2 plus half a v_or3_b32:
    //   x  = eq & ~w & tm       leaves with a u bit of 0 against a threshold bit of 1: below T       (a, b, c) = (1, 0, 1) -> 0x20
    //   eq = eq & ~(w ^ tm)     stays equal where the bits agree                                     (1, 0, 0), (1, 1, 1) -> 0x90
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        uint32_t x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t tm = taum[(pb * 4 + j) * 4 + n];
            x[j] = bitop3<0x20>(eq[n], o.w[j], tm);
            eq[n] = bitop3<0x90>(eq[n], o.w[j], tm);
        }
        lt[n] = (lt[n] | x[0] | x[1]) | (x[2] | x[3]);
    }
}

// ---- producers -------------------------------------------------------------------------------------
template <int K, int MODE>
__device__ __forceinline__ void produce_chunk(const SweepParams& P, const ChunkDesc& cd, uint4* __restrict__ desc,
                                              const uint16_t* __restrict__ tbl, const LeftList& left, int pw, int lane, uint32_t group,
                                              const uint32_t (&slots)[kProducerTasksMax])
{
    constexpr int NT = SweepCfg<K>::NT, NQ = SweepCfg<K>::NQ;
    const int C = P.C;
    const int ntask = ((int)cd.count + kWave - 1) / kWave;
#pragma unroll
    for (int j = 0; j < kProducerTasksMax; ++j) {
        const int task = pw + j * kProducerWaves;
        if (task >= ntask) break;
        const int p = task * kWave + lane;
        const bool live = p < (int)cd.count;
        const uint32_t slot = slots[j];            // P.slots[cd.slot_base + p], requested one step ago (0 for dead lanes)
        const uint32_t site = slot & 0xffffu;
        const uint64_t g = P.gbase + cd.g0 + (uint64_t)(slot >> 16);

        uint32_t f[NQ * 4];        // descriptor words
#pragma unroll
        for (int q = 0; q < NQ * 4; ++q) f[q] = 0u;
        {   // gather offsets: own word, then the K neighbour words (table row = K uint16 byte offsets)
            uint32_t off[K + 2];
            off[0] = MODE == 3 ? site : MODE == 2 ? 2u * site : site * 8u;   // the pair {s, ~s} of the site starts at word 2 * site (MODE 3: one word per site)
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
        // u < T_n, bit-sliced over the 32 replicas, most significant plane first (lazy evaluation of a counter-based
        // stream: the result does not depend on where we stop).  Every lane computes kSwBlocks blocks of 4 planes; the lanes that
        // still have an undecided replica hand the slot to the fix tasks through the leftover list, so that a producer's time
        // does not depend on its unluckiest lane.
        // fully unrolled: the block index is a compile-time constant, so the threshold words are loop-invariant scalars
        uint32_t pb = 0;
#pragma unroll
        for (int b = 0; b < kSwBlocks; ++b) refine_block<NT>(lt, eq, accept_planes(P.k0, P.k1, g, group, (uint32_t)b), (uint32_t)b, P.taum);
        pb = (uint32_t)kSwBlocks;
        bool need = any_set<NT>(eq);
        const unsigned long long bal = __ballot(need);
        if (bal != 0ull) {       // wave-uniform
            uint32_t base = 0u;
            if (lane == 0) base = atomicAdd(left.count, (uint32_t)__popcll(bal));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            const uint32_t idx = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            if (need && idx < (uint32_t)kSwLeftMax) {
                left.slot[idx] = (uint32_t)p | (slot & 0xffff0000u);     // position in the chunk | index of the attempt
#pragma unroll
                for (int n = 0; n < NT; ++n) { left.lt[n * kSwLeftMax + idx] = lt[n]; left.eq[n * kSwLeftMax + idx] = eq[n]; }
                need = false;
#pragma unroll
                for (int n = 0; n < NT; ++n) eq[n] = 0u;
            }
            // list full (practically never): finish those lanes here
            if (base + (uint32_t)__popcll(bal) > (uint32_t)kSwLeftMax)
                for (; pb < 16u; ++pb) {
                    if (!__any(any_set<NT>(eq))) break;
                    refine_block<NT>(lt, eq, accept_planes(P.k0, P.k1, g, group, pb), pb, P.taum);
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

// ---- fix tasks: finish the slots the producers left undecided and patch their masks into the descriptors ----
// One task = 64 entries of the chunk's leftover list: kSwFixBlocks more blocks for every lane, then a loop for whatever is still open.
// Task b of a chunk belongs to worker fix_worker(b): 0 = the fixer wave (tasks 0 and 1), 1 + pw = producer pw — dealt from the LAST
// producer downwards, the ones with the fewest 64-slot tasks of their own.
__device__ __forceinline__ int fix_worker(int b)
{
    const int r = b % (kProducerWaves + 2);
    return r < 2 ? 0 : 1 + (kProducerWaves - 1 - (r - 2));
}

template <int K>
__device__ __forceinline__ void fix_tasks(const SweepParams& P, const ChunkDesc& cd, uint4* __restrict__ desc, const LeftList& left,
                                          int lane, uint32_t group, int worker)
{
    constexpr int NT = SweepCfg<K>::NT;
    uint32_t n = *left.count;
    n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
    n = n < (uint32_t)kSwLeftMax ? n : (uint32_t)kSwLeftMax;
    const int ntask = (int)((n + kWave - 1) / kWave);
    for (int b = 0; b < ntask; ++b) {
        if (fix_worker(b) != worker) continue;
        const uint32_t i = (uint32_t)(b * kWave + lane);
        const bool live = i < n;
        uint32_t lt[NT], eq[NT], sl = 0u;
        if (live) sl = left.slot[i];
#pragma unroll
        for (int q = 0; q < NT; ++q) { lt[q] = live ? left.lt[q * kSwLeftMax + i] : 0u; eq[q] = live ? left.eq[q * kSwLeftMax + i] : 0u; }
        const uint64_t g = P.gbase + cd.g0 + (uint64_t)(sl >> 16);
#pragma unroll
        for (int b2 = 0; b2 < kSwFixBlocks; ++b2)
            refine_block<NT>(lt, eq, accept_planes(P.k0, P.k1, g, group, (uint32_t)(kSwBlocks + b2)), (uint32_t)(kSwBlocks + b2), P.taum);
        for (uint32_t pb = kSwBlocks + kSwFixBlocks; pb < 16u; ++pb) {
            if (!__any(any_set<NT>(eq))) break;
            refine_block<NT>(lt, eq, accept_planes(P.k0, P.k1, g, group, pb), pb, P.taum);
        }
        if (live) {
            uint32_t* m = reinterpret_cast<uint32_t*>(desc + (sl & 0xffffu));      // the masks are the first NT words of q[0]
#pragma unroll
            for (int q = 0; q < NT; ++q) m[q] = lt[q];
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

// MODE of the sweep kernel: 0 = neighbour table staged in LDS, byte offsets; 1 = table read from HBM/L2 (longer chunks), byte offsets
// (8 N <= 65536); 2 = table in HBM/L2, word indices (N > 8192: byte offsets into the 2N-word spin array no longer fit 16 bits).
// 3 = table in HBM/L2 and ONE word per site (8192 < N <= 32 767 when that buys clearly longer chunks): a table entry is the neighbour's
// word index with the coupling's sign in bit 15, applied after the gather — three more consumer instructions per neighbour, half the state.
// off = byte offset or word index (MODE 2, 3) into the LDS spin array
template <int MODE>
__device__ __forceinline__ uint32_t lds_word(const uint32_t* sp, uint32_t off)
{
    if constexpr (MODE == 3) return sp[off & 0x7fffu] ^ (uint32_t)((int32_t)(off << 16) >> 31);      // bit 15: J = -1, read as the complement
    else if constexpr (MODE == 2) return sp[off];
    else return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(sp) + off);
}
// the pair {s, ~s} of one site in a single 8-byte store (off = the pair's offset)
template <int MODE>
__device__ __forceinline__ void lds_store_pair(uint32_t* sp, uint32_t off, uint32_t v)
{
    if constexpr (MODE == 3) sp[off] = v;
    else if constexpr (MODE == 2) *reinterpret_cast<uint2*>(sp + off) = make_uint2(v, ~v);
    else *reinterpret_cast<uint2*>(reinterpret_cast<char*>(sp) + off) = make_uint2(v, ~v);
}

// the K+1 spin words one attempt reads: own word first, then the neighbour words
template <int K> struct SlotWords { uint32_t s; uint32_t g[K]; };

template <int K, int MODE>
__device__ __forceinline__ void gather_words(SlotWords<K>& w, const SlotDesc<K>& d, const uint32_t* __restrict__ sp)
{
    w.s = lds_word<MODE>(sp, desc_off<K>(d, 0));
#pragma unroll
    for (int k = 0; k < K; ++k) w.g[k] = lds_word<MODE>(sp, desc_off<K>(d, 1 + k));
}

// planes n0, n1, n2 of n = u[0] + ... + u[K-1] (K <= 7 one-bit inputs per replica, 32 replicas per word) with carry-save adders:
// a full adder of three words is two v_bitop3_b32 (sum 0x96, majority 0xe8), a half adder two plain ops — 8 instructions at K = 6 or 7,
// 6 at K = 4 or 5, 2 at K = 3, where the ripple form (n0 ^= u, carry into n1, carry into n2) takes 5 per input
template <int K>
__device__ __forceinline__ void count_planes(const uint32_t (&u)[K], uint32_t& n0, uint32_t& n1, uint32_t& n2)
{
    static_assert(K >= 1 && K <= 7, "three count planes");
    uint32_t ones[K], twos[4], fours[2];
    int no = K, nt = 0, nf = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) ones[k] = u[k];
    // reduce the weight-1 column to one word
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
        if (no >= 3) {
            const uint32_t a = ones[no - 1], b = ones[no - 2], c = ones[no - 3];
            ones[no - 3] = bitop3<0x96>(a, b, c);
            twos[nt++] = bitop3<0xe8>(a, b, c);
            no -= 2;
        }
    }
    if constexpr (K >= 2)
        if (no == 2) { const uint32_t a = ones[0], b = ones[1]; ones[0] = a ^ b; twos[nt++] = a & b; no = 1; }
    n0 = ones[0];
    // weight-2 column
    if (nt >= 3) {
        const uint32_t a = twos[nt - 1], b = twos[nt - 2], c = twos[nt - 3];
        twos[nt - 3] = bitop3<0x96>(a, b, c);
        fours[nf++] = bitop3<0xe8>(a, b, c);
        nt -= 2;
    }
    if (nt == 2) { const uint32_t a = twos[0], b = twos[1]; twos[0] = a ^ b; fours[nf++] = a & b; nt = 1; }
    n1 = nt ? twos[0] : 0u;
    n2 = nf == 2 ? (fours[0] ^ fours[1]) : (nf == 1 ? fours[0] : 0u);       // n <= 7: no carry out of the weight-4 column
}

// replicas whose count n (planes n0, n1, n2) equals the constant N: one three-input function
template <int NV>
__device__ __forceinline__ uint32_t count_is(uint32_t n0, uint32_t n1, uint32_t n2)
{
    return bitop3<(1u << (((NV & 1) << 2) | (((NV >> 1) & 1) << 1) | ((NV >> 2) & 1)))>(n0, n1, n2);
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
        uint32_t u[K];
#pragma unroll
        for (int k = 0; k < K; ++k) u[k] = w.g[k] ^ s;
        count_planes<K>(u, n0, n1, n2);
        // classes n = 0..NT-1 have dE > 0 and need u < T_n; every other class is accepted (RRRMC.jl:39)
        uint32_t rej = 0u;
        if constexpr (NT > 0) rej = bitop3<0xf2>(rej, desc_mask<K>(d, 0), count_is<0>(n0, n1, n2));        // rej | (~M_n & [n == class])
        if constexpr (NT > 1) rej = bitop3<0xf2>(rej, desc_mask<K>(d, 1), count_is<1>(n0, n1, n2));
        if constexpr (NT > 2) rej = bitop3<0xf2>(rej, desc_mask<K>(d, 2), count_is<2>(n0, n1, n2));
        if constexpr (NT > 3) rej = bitop3<0xf2>(rej, desc_mask<K>(d, 3), count_is<3>(n0, n1, n2));
        acc = ~rej;
    }
}

// One dependency level at a time; inside a level the attempts commute, so up to kRows x 64 of them are in flight
// per step: all descriptor loads, then all gathers, then the logic and the stores (the single consumer wave is bound by its own
// instruction stream and by LDS latency: the rows are its instruction-level parallelism, and every instruction saved per row
// counts).  The code is branch-free inside a step so that the LDS waits can be counted (s_waitcnt lgkmcnt(N)) instead of
// drained: rows 0..NR-2 are full; in the last row the lanes past the end of the level duplicate the level's last slot.
template <int K, int NR, int MODE>
__device__ __forceinline__ void consume_rows(const uint4* __restrict__ desc, uint32_t* __restrict__ sp, uint4* __restrict__ tal,
                                             int C, int p0, int plast)
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
    for (int j = 0; j < NR; ++j) gather_words<K, MODE>(w[j], d[j], sp);
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        // in the last row the lanes past the end of the batch hold a copy of its last slot: they compute and store exactly what
        // that slot's own lane does (same addresses, same values), so they need no special treatment
        int pt = p0 + j * kWave;
        if (j == NR - 1) pt = pt < plast ? pt : plast;
        typedef uint32_t v3u __attribute__((ext_vector_type(3)));
        if constexpr (K == 3) {
            // the consumer's row spelled as seven three-input functions (the compiler's own version of slot_logic + flip has eleven
            // instructions): planes of n = sum_k (s ^ g_k) are self-dual in s for odd K; a0 = M1 & (n0 | M0) is the part of the accept
            // mask that needed a random number, acc = n1 | a0 (n >= 2: dE <= 0) is left to the tally wave, which has the slack
            const uint32_t s = w[j].s;
            const uint32_t n0 = xor3(w[j].g[0], w[j].g[1], w[j].g[2]) ^ s;
            const uint32_t n1 = bitop3<0xe8>(w[j].g[0], w[j].g[1], w[j].g[2]) ^ s;
            const uint32_t a0 = bitop3<0xa8>(n0, desc_mask<K>(d[j], 0), desc_mask<K>(d[j], 1));
            const uint32_t snew = bitop3<0x1e>(s, n1, a0);      // s ^ (n1 | a0): spinflip! + update_cache! (Interface.jl:89-92, RRG.jl:191-234)
            if constexpr (MODE == 3) sp[desc_off<K>(d[j], 0)] = snew;
            else {
                const uint32_t nsnew = bitop3<0xe1>(s, n1, a0);
                if constexpr (MODE == 2) *reinterpret_cast<uint2*>(sp + desc_off<K>(d[j], 0)) = make_uint2(snew, nsnew);
                else *reinterpret_cast<uint2*>(reinterpret_cast<char*>(sp) + desc_off<K>(d[j], 0)) = make_uint2(snew, nsnew);
            }
            *reinterpret_cast<v3u*>(tal + pt) = v3u{a0, n0, n1};
        } else {
            uint32_t n0, n1, n2, acc;
            slot_logic<K>(d[j], w[j], n0, n1, n2, acc);
            const uint32_t snew = w[j].s ^ acc;      // spinflip! + update_cache! (Interface.jl:89-92, RRG.jl:191-234)
            lds_store_pair<MODE>(sp, desc_off<K>(d[j], 0), snew);
            if constexpr (SweepCfg<K>::NS <= 3) {
                // the tally reads three words for K <= 3 (n2 = 0): a 12-byte LDS write issues faster than a 16-byte one
                *reinterpret_cast<v3u*>(tal + pt) = v3u{acc, n0, n1};
            } else {
                tal[pt] = make_uint4(acc, n0, n1, n2);
            }
        }
    }
}

// consume_rows<NR> for the run-time row count `rows` in LO..HI (wave-uniform): binary search over the instantiations
template <int K, int MODE, int LO, int HI>
__device__ __forceinline__ void consume_dispatch(int rows, const uint4* __restrict__ desc, uint32_t* __restrict__ sp, uint4* __restrict__ tal,
                                                 int C, int p0, int plast)
{
    if constexpr (LO >= HI) consume_rows<K, HI, MODE>(desc, sp, tal, C, p0, plast);
    else {
        constexpr int MID = (LO + HI) / 2;
        if (rows <= MID) consume_dispatch<K, MODE, LO, MID>(rows, desc, sp, tal, C, p0, plast);
        else consume_dispatch<K, MODE, MID + 1, HI>(rows, desc, sp, tal, C, p0, plast);
    }
}

// The planner has already cut every dependency level into batches of <= kRows x 64 slots.  Inside a batch the
// attempts commute, so all its descriptor loads, then all its gathers are issued before the logic and the stores
// (the single consumer wave is latency-bound: this is its instruction-level parallelism).  A batch is branch-free
// so that the LDS waits can be counted instead of drained: rows 0..NR-2 are full; in the last row the lanes past
// the end of the batch duplicate its last slot (same loads, same stores).
template <int K, int MODE>
__device__ __forceinline__ void consume_chunk(const SweepParams& P, const ChunkDesc& cd, const uint4* __restrict__ desc,
                                              uint32_t* __restrict__ sp, uint4* __restrict__ tal, int lane, uint32_t first_vd)
{
    const int C = P.C;
    const uint32_t nb = cd.nvec;
    for (uint32_t v0 = 0; v0 < nb; v0 += kWave) {
        // the batch table is read 64 entries at a time (one coalesced load) and broadcast lane by lane; the first 64 entries
        // (all of them, for the usual chunk) were requested one step ago
        uint32_t myvd = first_vd;
        if (v0 > 0) myvd = (v0 + lane < nb) ? P.vecs[cd.slot_base + v0 + lane] : 0u;
        const uint32_t vn = (nb - v0) < (uint32_t)kWave ? (nb - v0) : (uint32_t)kWave;
        for (uint32_t v = 0; v < vn; ++v) {
            const uint32_t vd = __builtin_amdgcn_readlane(myvd, v);
            const int start = (int)(vd & 0xffffu), cm1 = (int)(vd >> 16);   // cm1 = slots - 1
            const int p0 = start + lane, plast = start + cm1;
            // rows of the batch (wave-uniform; kWave = 64).  Most batches are the short tail levels: one row is tested first, the rest is a
            // binary tree — a flat switch over 14 cases costs the structurised code two scalar instructions per case on the way out
            if (cm1 < kWave) consume_rows<K, 1, MODE>(desc, sp, tal, C, p0, plast);
            else consume_dispatch<K, MODE, 2, (sweep_rows<K>() > 2 ? sweep_rows<K>() : 2)>((cm1 >> 6) + 1, desc, sp, tal, C, p0, plast);
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

// carry-save adder of three words: sum = a ^ b ^ c (truth table 0x96), carry = majority (0xe8) — two v_bitop3_b32; left to the plain
// expression the compiler spends five instructions on it (xor, and, and, or, xor)
#define RRRMC_CSA(sum, carry, a, b, c)                      \
    {                                                       \
        const uint32_t a_ = (a), b_ = (b), c_ = (c);        \
        carry = bitop3<0xe8>(a_, b_, c_);                   \
        sum = bitop3<0x96>(a_, b_, c_);                     \
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
        // keeps the four cases of the caller's switch from being merged into one store with a run-time index (which moves the
        // whole array to scratch memory)
        asm volatile("" : "+v"(t.Pd[s][TZ]));
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
// Butterfly over J = 16, 8, 4, 2, 1: exchange with lane ^ J (ds_swizzle), rotate the partner's word so that the
// bits to take land on the positions this lane gives up (the wrapped-around bits fall on positions it keeps and are
// masked by the bit-field insert): 3 instructions per stage.
struct TransposeConsts {
    uint32_t keep[5];   // bit positions this lane keeps at stage s
    uint32_t rot[5];    // right-rotation of the partner's word at stage s
    __device__ __forceinline__ void init(int lane)
    {
        const uint32_t m[5] = {0x0000ffffu, 0x00ff00ffu, 0x0f0f0f0fu, 0x33333333u, 0x55555555u};
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const int J = 16 >> s;
            const bool hi = (lane & J) != 0;
            keep[s] = hi ? ~m[s] : m[s];
            rot[s] = hi ? (uint32_t)J : (uint32_t)(32 - J);
        }
    }
};

__device__ __forceinline__ uint32_t transpose32(uint32_t a, const TransposeConsts& tc)
{
#define RRRMC_TSTAGE(S, J)                                                                           \
    {                                                                                                \
        const uint32_t o = (uint32_t)__builtin_amdgcn_ds_swizzle((int)a, ((J) << 10) | 0x1f);        \
        const uint32_t sel = __builtin_amdgcn_alignbit(o, o, tc.rot[S]);                             \
        a = bitop3<0xe4>(a, sel, tc.keep[S]);      /* keep ? a : sel */                             \
    }
    RRRMC_TSTAGE(0, 16)
    RRRMC_TSTAGE(1, 8)
    RRRMC_TSTAGE(2, 4)
    RRRMC_TSTAGE(3, 2)
    RRRMC_TSTAGE(4, 1)
#undef RRRMC_TSTAGE
    return a;
}

// Column sums of one stream of a counter set: returns, in lane r and r + 32, the count of replica r over this wave
// half's 32 lanes (the caller adds the two halves).  Branch-free over the planes so that the transposes overlap.
template <int NS>
__device__ __forceinline__ uint32_t tally_stream_total(const TallyState<NS>& t, int s, const TransposeConsts& tc)
{
    const uint32_t ng = __builtin_amdgcn_readfirstlane(t.ngrp);
    uint32_t c = 0u;
#pragma unroll
    for (int l = 0; l < 3; ++l) c += (uint32_t)__popc(transpose32(t.lo[s][l], tc)) << l;
#pragma unroll
    for (int l = 0; l < kTallyHi; ++l) {
        const uint32_t pend = ((ng >> l) & 1u) ? t.Pd[s][l] : 0u;     // pending word is live iff bit l of the group count
        c += ((uint32_t)__popc(transpose32(t.S[s][l], tc)) + (uint32_t)__popc(transpose32(pend, tc))) << (3 + l);
    }
    return c;
}

// Deferred flush.  Folding a counter set into per-replica integers costs ~11 transposes per stream; doing all of it in
// the step where an energy sample falls would make the tally wave that step's slowest wave.  Instead the live set
// is moved aside ("drain" set, registers) and one stream per step is folded; the sample is written when the last
// stream is done.  Energies only have to be complete when the kernel ends, so the delay is invisible.
template <int NS, int K> struct TallyDrain {
    TallyState<NS> set;
    uint32_t tot[2];        // partial totals (A, S) of this wave half
    int phase;              // next stream to fold; NS = idle
    bool emit;              // write an energy sample when done
    int64_t sample;         // ... at this sample index

    __device__ __forceinline__ void step(const TransposeConsts& tc)       // fold one stream
    {
        uint32_t c = 0u;
        switch (phase) {        // wave-uniform
            case 0: c = tally_stream_total<NS>(set, 0, tc); break;
            case 1: c = tally_stream_total<NS>(set, 1 % NS, tc); break;
            case 2: c = tally_stream_total<NS>(set, 2 % NS, tc); break;
            default: c = tally_stream_total<NS>(set, 3 % NS, tc); break;
        }
        if (phase == 0) tot[0] += c; else tot[1] += c << (phase - 1);   // stream 0 counts A; streams 1.. are the planes of n
        phase += 1;
    }
    // finished: update the running energy / accepted count and write the pending sample
    __device__ __forceinline__ void finish(const SweepParams& P, int lane, int32_t& E_run, int64_t& A_run)
    {
        const uint32_t a = tot[0] + (uint32_t)__shfl_xor((int)tot[0], 32);
        const uint32_t sn = tot[1] + (uint32_t)__shfl_xor((int)tot[1], 32);
        E_run += 2 * (K * (int32_t)a - 2 * (int32_t)sn);      // sum over accepted moves of dE = 2(K - 2n)
        A_run += a;
        if (emit && lane < 32 && P.Es) P.Es[sample * P.Rpad + blockIdx.x * 32 + lane] = E_run;
        emit = false;
    }
    __device__ __forceinline__ void run_to_end(const SweepParams& P, const TransposeConsts& tc, int lane, int32_t& E_run, int64_t& A_run)
    {
        if (phase >= NS) return;
        while (phase < NS) step(tc);
        finish(P, lane, E_run, A_run);
    }
};

template <int NS>
__device__ __forceinline__ void tally_reset(TallyState<NS>& t)
{
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int l = 0; l < 3; ++l) t.lo[s][l] = 0u;
#pragma unroll
        for (int l = 0; l < kTallyHi; ++l) { t.S[s][l] = 0u; t.Pd[s][l] = 0u; }
    }
    t.ngrp = 0u;
}

template <int NS, int K, bool MASKED>
__device__ __forceinline__ void tally_group(TallyState<NS>& t, const uint4* __restrict__ tal, int base, int count, int lane)
{
    uint32_t x[8][NS];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int p = base + j * kWave + lane;
        uint4 w = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (SweepCfg<K>::NS <= 3) {
            typedef uint32_t v3u __attribute__((ext_vector_type(3)));
            if (!MASKED || p < count) { const v3u v = *reinterpret_cast<const v3u*>(tal + p); w.x = v.x; w.y = v.y; w.z = v.z; }
        } else {
            if (!MASKED || p < count) w = tal[p];
        }
        if constexpr (K == 3) {
            // the consumer leaves (a0, n0, n1) with acc = n1 | a0 (consume_rows): acc & n0 = (a0 | n1) & n0, acc & n1 = n1
            x[j][0] = w.x | w.z;
            x[j][1 % NS] = bitop3<0xa8>(w.x, w.z, w.y);
            x[j][2 % NS] = w.z;
        } else {
            x[j][0] = w.x;
            if (NS > 1) x[j][1 % NS] = w.x & w.y;
            if (NS > 2) x[j][2 % NS] = w.x & w.z;
            if (NS > 3) x[j][3 % NS] = w.x & w.w;
        }
    }
    tally_add8<NS>(t, x);
}

template <int NS, int K>
__device__ __forceinline__ void tally_chunk(TallyState<NS>& t, const ChunkDesc& cd, const uint4* __restrict__ tal, int lane)
{
    // every lane adds the same number of words (zeros past the end) so that the counters' shape stays wave-uniform; only the last
    // group of 8 x 64 slots can be partial and needs the bounds test
    const int count = (int)cd.count;
    int base = 0;
    for (; base + 8 * kWave <= count; base += 8 * kWave) tally_group<NS, K, false>(t, tal, base, count, lane);
    if (base < count) tally_group<NS, K, true>(t, tal, base, count, lane);
}

template <int K, int MODE = 0>
__global__ __launch_bounds__(kSweepThreads) void sweep_kernel(SweepParams P)
{
    constexpr int NQ = SweepCfg<K>::NQ, NS = SweepCfg<K>::NS;
    extern __shared__ uint32_t lds[];
    const int N = P.N, C = P.C;
    uint32_t* sp = lds;                                              // [2N]  pairs {s, ~s} ([N] words in MODE 3)
    uint4* desc = reinterpret_cast<uint4*>(sp + (((MODE == 3 ? 1 : 2) * N + 128 + 3) & ~3)); // [3][NQ][C]   (128 spare words behind the spins)
    uint4* tal = desc + 3 * NQ * C;                                  // [2][C + 64]  (64 dummy entries per buffer)
    uint32_t* leftmem = reinterpret_cast<uint32_t*>(tal + 2 * (C + kWave));   // [2] leftover lists
    // leftover lists: 4 count words (three in rotation: chunk c counts in word c % 3, which is cleared during step c - 1 — after its last
    // readers, the fix tasks of chunk c - 3 in step c - 2, are behind a barrier), then two entry buffers (chunk c uses buffer c & 1)
    constexpr int kLeftBuf = kSwLeftMax * (1 + 2 * SweepCfg<K>::NT);
    constexpr int kLeftAll = 4 + 2 * kLeftBuf;
    // [N][TS] neighbour table: a copy in LDS, or (MODE >= 1) the HBM/L2 original
    const uint16_t* tbl = MODE >= 1 ? P.table : reinterpret_cast<const uint16_t*>(leftmem + kLeftAll);

    const int tid = threadIdx.x, lane = tid & 63;
    // the wave index as a SCALAR: role dispatch becomes s_cbranch (and s_setprio below really is per wave)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t group = P.group0 + blockIdx.x;
    uint32_t* gsp = P.spins + (size_t)blockIdx.x * N;

    for (int x = tid; x < N; x += kSweepThreads) {
        const uint32_t w = gsp[x];
        if constexpr (MODE == 3) sp[x] = w;
        else *reinterpret_cast<uint2*>(sp + 2 * x) = make_uint2(w, ~w);  // word 2x: the spins of site x, word 2x + 1: their complement
    }
    if constexpr (MODE == 0) {
        uint16_t* tbl_w = reinterpret_cast<uint16_t*>(leftmem + kLeftAll);
        for (int q = tid; q < N * P.TS; q += kSweepThreads) tbl_w[q] = P.table[q];
    }
    if (tid < 4) leftmem[tid] = 0u;
    __syncthreads();
    auto left_list = [&](int c) {
        uint32_t* m = leftmem + 4 + (c & 1) * kLeftBuf;
        LeftList l;
        l.count = leftmem + (c % 3); l.slot = m; l.lt = m + kSwLeftMax; l.eq = m + kSwLeftMax * (1 + SweepCfg<K>::NT);
        return l;
    };

    // Software pipeline over chunks, one workgroup barrier per step:  produce(c) | fix(c-1) | consume(c-2) | tally(c-3).
    // Each role runs its OWN loop (the branch on the scalar wave index is outside the loops), so that a role's
    // registers are not live across the other roles' code; every wave executes exactly nsteps barriers.
    const int nsteps = P.nchunks + 3;
    const int tal_stride = C + kWave;
#ifdef RRRMC_STAMPS
    unsigned long long busy = 0;
#define RRRMC_T0 const unsigned long long t_in = __builtin_amdgcn_s_memtime();
#define RRRMC_T1 { const unsigned long long dt_ = __builtin_amdgcn_s_memtime() - t_in; busy += dt_; \
        if (blockIdx.x == 0 && lane == 0 && P.stamps && c < 4096) P.stamps[16 * 65536 + c * 16 + wave] = dt_; }
#else
#define RRRMC_T0
#define RRRMC_T1
#endif
    if (wave == kConsumerWave) {
        // consumer: single latency-bound wave on the critical path -> let it win the SIMD's issue arbitration
        __builtin_amdgcn_s_setprio(3);
        // chunk descriptors are requested two steps, the head of the batch table one step before use (as the producers do)
        ChunkDesc ca{}, cb{}, cc{};
        uint32_t va = 0u, vb = 0u;
        for (int c = 0; c < nsteps; ++c) {
            RRRMC_T0
            ca = cb; va = vb;           // chunk c-2: consumed now
            cb = cc;                    // chunk c-1: its batch table is requested now
            vb = (c >= 1 && c - 1 < P.nchunks && (uint32_t)lane < cb.nvec) ? P.vecs[cb.slot_base + lane] : 0u;
            if (c < P.nchunks) cc = P.chunks[c];
#ifndef RRRMC_ABLATE_CONSUME      // timing experiments only (tools/ablate.sh): results are wrong with a role removed
            if (c >= 2 && c - 2 < P.nchunks)
                consume_chunk<K, MODE>(P, ca, desc + ((c - 2) % 3) * NQ * C, sp, tal + ((c - 2) & 1) * tal_stride, lane, va);
#endif
            RRRMC_T1
            __syncthreads();
        }
    } else if (wave == kSwTallyWave) {
        // tally: lanes 0..31 own the running energy / accepted count of replica `lane`
        __builtin_amdgcn_s_setprio(2);
        TransposeConsts tc;
        tc.init(lane);
        TallyState<NS> ts;
        tally_reset<NS>(ts);
        TallyDrain<NS, K> dr;
        tally_reset<NS>(dr.set);
        dr.tot[0] = 0u; dr.tot[1] = 0u; dr.phase = NS; dr.emit = false; dr.sample = 0;
        int32_t E_run = 0;
        int64_t A_run = 0;
        if (lane < 32) { E_run = P.E_cur[blockIdx.x * 32 + lane]; A_run = P.acc_cur[blockIdx.x * 32 + lane]; }
        int64_t ns = P.sample0;
        const uint32_t grp_per_chunk = (uint32_t)((C + 8 * kWave - 1) / (8 * kWave));   // 8-word groups one chunk adds per lane
        ChunkDesc ta{}, tb{};
        for (int c = 0; c < nsteps; ++c) {
            RRRMC_T0
            ta = tb;                                        // chunk c-3, requested one step ago
            if (c >= 2 && c - 2 < P.nchunks) tb = P.chunks[c - 2];
#ifndef RRRMC_ABLATE_TALLY
            if (c >= 3) {
                const ChunkDesc cd = ta;
                // an energy sample is due BEFORE this chunk's moves (RRRMC.jl:104-108)
                const bool sample = (cd.flags & kChunkSampleBefore) != 0;
                if (sample || ts.ngrp + grp_per_chunk > (1u << kTallyHi) - 1u) {
                    dr.run_to_end(P, tc, lane, E_run, A_run);     // an older set still draining goes first (order matters)
                    dr.set = ts;                                   // move the live counters aside and start afresh
                    tally_reset<NS>(ts);
                    dr.tot[0] = 0u; dr.tot[1] = 0u; dr.phase = 0;
                    dr.emit = sample; dr.sample = ns;
                    if (sample) ns += 1;
                }
                if (dr.phase < NS) {                               // fold one stream of the draining set per step
                    dr.step(tc);
                    if (dr.phase == NS) dr.finish(P, lane, E_run, A_run);
                }
                tally_chunk<NS, K>(ts, cd, tal + ((c - 3) & 1) * tal_stride, lane);
            }
#endif
            RRRMC_T1
            __syncthreads();
        }
        dr.run_to_end(P, tc, lane, E_run, A_run);
        dr.set = ts;
        dr.tot[0] = 0u; dr.tot[1] = 0u; dr.phase = 0; dr.emit = false;
        dr.run_to_end(P, tc, lane, E_run, A_run);
        if (lane < 32) { P.E_cur[blockIdx.x * 32 + lane] = E_run; P.acc_cur[blockIdx.x * 32 + lane] = A_run; }
    } else if (wave == kFixerWave) {
        ChunkDesc fa{}, fb{};
        for (int c = 0; c < nsteps; ++c) {
            RRRMC_T0
            fa = fb;                                        // chunk c-1, requested one step ago
            if (c < P.nchunks) fb = P.chunks[c];
            if (lane == 0) leftmem[(c + 1) % 3] = 0u;          // the count word of the NEXT chunk's list
            if (c >= 1 && c - 1 < P.nchunks) fix_tasks<K>(P, fa, desc + ((c - 1) % 3) * NQ * C, left_list(c - 1), lane, group, 0);
            RRRMC_T1
            __syncthreads();
        }
    } else {
        const int pw = sw_producer_index(wave);     // producer index 0 .. kProducerWaves-1
        // The chunk descriptor (scalar loads) and the slot words (one coalesced global load per task) are requested ahead of
        // time — the descriptor two steps, the slots one step before they are used — so a step starts computing right after the
        // barrier instead of waiting a microsecond for HBM/L2 (that latency was a third of a step).
        ChunkDesc cdp{}, cd0{}, cd1{}, cd2{};
        uint32_t sl0[kProducerTasksMax], sl1[kProducerTasksMax];
#pragma unroll
        for (int j = 0; j < kProducerTasksMax; ++j) { sl0[j] = 0u; sl1[j] = 0u; }
        auto fetch_slots = [&](const ChunkDesc& cd, uint32_t (&sl)[kProducerTasksMax]) {
#pragma unroll
            for (int j = 0; j < kProducerTasksMax; ++j) {
                const int p = (pw + j * kProducerWaves) * kWave + lane;
                sl[j] = p < (int)cd.count ? P.slots[cd.slot_base + p] : 0u;
            }
        };
        if (P.nchunks > 0) { cd1 = P.chunks[0]; fetch_slots(cd1, sl1); }
        if (P.nchunks > 1) cd2 = P.chunks[1];
        for (int c = 0; c < nsteps; ++c) {
            RRRMC_T0
            cdp = cd0;                                      // chunk c-1: its fix tasks
            cd0 = cd1;
#pragma unroll
            for (int j = 0; j < kProducerTasksMax; ++j) sl0[j] = sl1[j];
            cd1 = cd2;
            if (c + 1 < P.nchunks) fetch_slots(cd1, sl1);
            if (c + 2 < P.nchunks) cd2 = P.chunks[c + 2];
#ifndef RRRMC_ABLATE_PRODUCE
            if constexpr (kSwLeftMax > 2 * kWave)
                if (c >= 1 && c - 1 < P.nchunks) fix_tasks<K>(P, cdp, desc + ((c - 1) % 3) * NQ * C, left_list(c - 1), lane, group, 1 + pw);
            if (c < P.nchunks) produce_chunk<K, MODE>(P, cd0, desc + (c % 3) * NQ * C, tbl, left_list(c), pw, lane, group, sl0);
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
    // the write-back loop starts from an opaque copy of the thread id: otherwise its trip count is shared with the staging loop at
    // the top and kept alive (spilled to scratch memory by the register-hungry tally role) across the whole kernel
    int tid_wb = tid;
    asm volatile("" : "+v"(tid_wb));
    for (int x = tid_wb; x < N; x += kSweepThreads) gsp[x] = sp[MODE == 3 ? x : 2 * x];
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

// ---------------------------------------------------------------------------------------------------
// Colour-parallel Metropolis sweeps for graphs that do not fit the LDS-resident kernel (BASELINE.json config 4:
// GraphEA L = 64, D = 3).  NOT in the reference (its standardMC is random-site, src/RRRMC.jl:113); this is the
// build-defined "checkerboard" extension of SURVEY.md §7 hard part 7: given a proper colouring of the graph, one
// sweep visits the colours in order and attempts every site of a colour simultaneously (they do not interact), with
// the reference's energy function, delta_energy and accept rule (src/graphs/EA.jl:195-275, src/RRRMC.jl:39).
// Random numbers: SWEEP stream, bit planes of the acceptance uniforms of replica group `group` at (sweep, site):
//   ctr = (site, lo32(sweep), group, TAG_SWEEP | pb << 8 | bits 32..47 of sweep << 16), plane j = word j & 3 of block j >> 2.
// Spins stay bit-sliced in HBM/L2: spins[G][N].
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t TAG_SWEEP = 7;

struct ColorSweepParams {
    uint32_t* spins;          // [G][N]
    const int32_t* A;         // [N][K]
    const int8_t* J;          // [N][K]
    const uint32_t* recs;     // [nlist][W] the colour's sites as records: site, then neighbour | (J < 0) << 31 for its K neighbours
                              // (W = 4 words up to K = 3, 8 beyond: one or two 16-byte loads instead of 1 + 2K scattered ones)
    uint32_t taum[64 * 4];    // threshold bit planes, class n = unsatisfied bonds (dE = 2(K - 2n) > 0)
    uint32_t always_mask;
    uint32_t k0, k1, group0;
    uint64_t sweep;           // global sweep index (1-based)
    int N, nlist, G;          // G = replica groups of the context
    int64_t* acc_cur;         // [Rpad] accepted moves per replica (COUNT builds only)
};

// COUNT: also add every replica's accepted moves to P.acc_cur (32x32 bit transpose + popcount per wave, LDS atomics per workgroup,
// 32 global atomics per workgroup) — the parity / acceptance-rate build; the plain build leaves the counters alone.
// GPT = replica groups one thread updates (its site's neighbour row and couplings are loaded once, the gathers of all its groups are
// in flight together): grid.y = ceil(G / GPT)
template <int K, bool COUNT = false, int GPT = 1>
__global__ __launch_bounds__(256) void colored_sweep_kernel(ColorSweepParams P)
{
    constexpr int NT = SweepCfg<K>::NT;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const bool live = idx < P.nlist;
    constexpr int W = K <= 3 ? 4 : 8;
    uint32_t rw[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    {
        const uint4* rp = reinterpret_cast<const uint4*>(P.recs) + (size_t)(live ? idx : 0) * (W / 4);
        const uint4 r0 = rp[0];
        rw[0] = r0.x; rw[1] = r0.y; rw[2] = r0.z; rw[3] = r0.w;
        if constexpr (W == 8) { const uint4 r1 = rp[1]; rw[4] = r1.x; rw[5] = r1.y; rw[6] = r1.z; rw[7] = r1.w; }
    }
    const int x = (int)rw[0];
    int nb[K];
    uint32_t sgn[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { nb[k] = (int)(rw[1 + k] & 0x7fffffffu); sgn[k] = (uint32_t)((int32_t)rw[1 + k] >> 31); }
    uint32_t s[GPT], u[GPT][K];
#pragma unroll
    for (int j = 0; j < GPT; ++j) {
        const int gl = (int)blockIdx.y * GPT + j;
        const uint32_t* sp = P.spins + (size_t)(gl < P.G ? gl : 0) * P.N;
        s[j] = sp[x];
#pragma unroll
        for (int k = 0; k < K; ++k) u[j][k] = sp[nb[k]];
    }
    const uint32_t c3hi = (uint32_t)((P.sweep >> 32) & 0xffffu) << 16;
#pragma unroll
    for (int j = 0; j < GPT; ++j) {
        const int gl = (int)blockIdx.y * GPT + j;
        if (gl >= P.G) break;                       // block-uniform
        const uint32_t group = P.group0 + (uint32_t)gl;
        uint32_t n0, n1, n2, uu[K];
#pragma unroll
        for (int k = 0; k < K; ++k) uu[k] = xor3(s[j], u[j][k], sgn[k]);   // bond k unsatisfied
        count_planes<K>(uu, n0, n1, n2);
        uint32_t lt[NT], eq[NT], cls[NT];
        // only the replicas that are in class n need the comparison at all
        if constexpr (NT > 0) cls[0] = count_is<0>(n0, n1, n2);
        if constexpr (NT > 1) cls[1] = count_is<1>(n0, n1, n2);
        if constexpr (NT > 2) cls[2] = count_is<2>(n0, n1, n2);
        if constexpr (NT > 3) cls[3] = count_is<3>(n0, n1, n2);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const bool always = (P.always_mask >> n) & 1u;
            lt[n] = always ? 0xffffffffu : 0u;
            eq[n] = (live && !always) ? cls[n] : 0u;
        }
        for (uint32_t pb = 0; pb < 16u; ++pb) {
            if (!__any(any_set<NT>(eq))) break;
            const Philox4 o = philox4x32_10((uint32_t)x, (uint32_t)P.sweep, group, TAG_SWEEP | (pb << 8) | c3hi, P.k0, P.k1);
            refine_block<NT>(lt, eq, o, pb, P.taum);
        }
        uint32_t rej = 0u;
#pragma unroll
        for (int n = 0; n < NT; ++n) rej = bitop3<0xf2>(rej, lt[n], cls[n]);        // rej | (~lt_n & [n == class])
        if (live) P.spins[(size_t)gl * P.N + x] = s[j] ^ ~rej;
        if constexpr (COUNT) {
            __shared__ uint32_t s_acc[32];
            __syncthreads();
            if (threadIdx.x < 32) s_acc[threadIdx.x] = 0u;
            __syncthreads();
            TransposeConsts tc;
            tc.init((int)(threadIdx.x & 63));
            uint32_t c = (uint32_t)__popc(transpose32(live ? ~rej : 0u, tc));      // lane r (and r + 32): replica r's accepted moves in this wave half
            c += (uint32_t)__shfl_xor((int)c, 32);
            if ((threadIdx.x & 63) < 32 && c) atomicAdd(&s_acc[threadIdx.x & 31], c);
            __syncthreads();
            if (threadIdx.x < 32 && s_acc[threadIdx.x])
                atomicAdd(reinterpret_cast<unsigned long long*>(P.acc_cur) + gl * 32 + threadIdx.x, (unsigned long long)s_acc[threadIdx.x]);
        }
    }
}

// Bit-sliced energy: U[group*32 + r] += sum over sites of the number of unsatisfied bonds of replica r (every bond
// from both ends); E = U - N*K/2 (RRG.jl:164-189).  One lane per site, three count planes, 32x32 bit transpose +
// popcount per wave, one atomic per replica and workgroup (2048 sites).
constexpr int kEnergySitesPerThread = 8;
template <int K>
__global__ __launch_bounds__(256) void energy_bs_kernel(const uint32_t* __restrict__ spins, const int32_t* __restrict__ A,
                                                        const int8_t* __restrict__ J, int N, uint32_t* __restrict__ U)
{
    __shared__ uint32_t part[4][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t* sp = spins + (size_t)blockIdx.y * N;
    TransposeConsts tc;
    tc.init(lane);
    uint32_t tot = 0u;
    // a workgroup covers 256 * kEnergySitesPerThread sites, so that the per-replica sums meet in registers / LDS and only one
    // atomic per replica and workgroup reaches HBM (at N = 262144 the atomics of a wave-per-256-sites layout took 0.2 ms)
    for (int i = 0; i < kEnergySitesPerThread; ++i) {              // uniform trip count: the transpose exchanges data across lanes
        const int x = (blockIdx.x * kEnergySitesPerThread + i) * 256 + (int)threadIdx.x;
        uint32_t n0 = 0u, n1 = 0u, n2 = 0u;
        if (x < N) {
            const uint32_t s = sp[x];
            uint32_t u[K];
#pragma unroll
            for (int k = 0; k < K; ++k) u[k] = xor3(s, sp[A[(size_t)x * K + k]], J[(size_t)x * K + k] < 0 ? 0xffffffffu : 0u);
            count_planes<K>(u, n0, n1, n2);
        }
        tot += (uint32_t)__popc(transpose32(n0, tc)) + ((uint32_t)__popc(transpose32(n1, tc)) << 1) + ((uint32_t)__popc(transpose32(n2, tc)) << 2);
    }
    tot += (uint32_t)__shfl_xor((int)tot, 32);
    if (lane < 32) part[wave][lane] = tot;
    __syncthreads();
    if (threadIdx.x < 32)
        atomicAdd(&U[blockIdx.y * 32 + threadIdx.x], part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// Es[sample][r] = U[r] - N*K/2 (also leaves the value in E_cur) and clears U for the next sample
__global__ __launch_bounds__(256) void energy_bs_finish_kernel(uint32_t* __restrict__ U, int32_t* __restrict__ E_cur, int32_t* __restrict__ Es_row,
                                                               int Rpad, int halfNK)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= Rpad) return;
    const int32_t E = (int32_t)U[r] - halfNK;
    U[r] = 0u;
    if (E_cur) E_cur[r] = E;
    if (Es_row) Es_row[r] = E;
}

}  // namespace rrrmc
