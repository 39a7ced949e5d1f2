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
constexpr int kSweepThreads = 1024;              // 16 waves: wave 0 consumes, waves 1..15 produce
constexpr int kProducerWaves = kSweepThreads / kWave - 1;
constexpr int kPlanThreads = 256;
constexpr int kMaxK = 7;                          // 3 bit planes for the unsatisfied-bond count
constexpr uint32_t kChunkSampleBefore = 1u;       // chunk flag: an energy sample is due before its first move

struct ChunkDesc {
    uint64_t g0;         // global iteration (1-based) of the chunk's first attempt
    uint32_t count;      // attempts in the chunk (<= C)
    uint32_t slot_base;  // offset of the chunk's slots / vector table in the plan buffers
    uint32_t nvec;       // written by plan_kernel
    uint32_t flags;
};

// ---------------------------------------------------------------------------------------------------
// plan_kernel: one workgroup per chunk.
//   slots[slot_base + p] = site | (t << 16)   attempts sorted by dependency level (t = index in chunk)
//   vecs [slot_base + v] = start | (n << 16)  runs of <= 64 sorted slots that lie inside one level
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
            uint32_t n = s_cnt[l];
            uint32_t q = pos;
            while (n > 0) {
                const uint32_t m = n < (uint32_t)kWave ? n : (uint32_t)kWave;
                vecs[cd.slot_base + nvec++] = q | (m << 16);
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
// sweep_kernel
// ---------------------------------------------------------------------------------------------------
struct SweepParams {
    uint32_t* spins;          // [G][N]   bit-sliced configuration
    const uint16_t* table;    // [N][TS]  neighbour word indices: y (J=+1) or y+N (J=-1), TS = table stride
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
};

template <int K> struct SweepCfg {
    static constexpr int NT = (K + 1) / 2;        // classes with dE > 0: n = 0 .. NT-1
    static constexpr int NF = NT + K + 1;         // descriptor fields: NT masks, own address, K neighbour addresses
};

template <int K>
__device__ __forceinline__ void produce_chunk(const SweepParams& P, const ChunkDesc& cd, uint32_t* __restrict__ desc,
                                              const uint16_t* __restrict__ tbl, int pw, int lane, uint32_t group)
{
    constexpr int NT = SweepCfg<K>::NT;
    const int C = P.C;
    const int ntask = ((int)cd.count + kWave - 1) / kWave;
    for (int task = pw; task < ntask; task += kProducerWaves) {
        const int p = task * kWave + lane;
        const bool live = p < (int)cd.count;
        uint32_t slot = 0;
        if (live) slot = P.slots[cd.slot_base + p];
        const uint32_t site = slot & 0xffffu;
        const uint64_t g = cd.g0 + (uint64_t)(slot >> 16);

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
            for (int n = 0; n < NT; ++n) desc[n * C + p] = lt[n];
            desc[NT * C + p] = site;
            const uint16_t* row = tbl + (size_t)site * P.TS;
#pragma unroll
            for (int k = 0; k < K; ++k) desc[(NT + 1 + k) * C + p] = (uint32_t)row[k];
        }
    }
}

template <int K>
__device__ __forceinline__ void consume_chunk(const SweepParams& P, const ChunkDesc& cd, const uint32_t* __restrict__ desc,
                                              uint32_t* __restrict__ sp, uint4* __restrict__ tal, int lane)
{
    constexpr int NT = SweepCfg<K>::NT;
    const int C = P.C;
    const int N = P.N;
    const uint32_t nvec = cd.nvec;
    for (uint32_t v = 0; v < nvec; ++v) {
        const uint32_t vd = P.vecs[cd.slot_base + v];
        const int start = (int)(vd & 0xffffu), cnt = (int)(vd >> 16);
        if (lane < cnt) {
            const int p = start + lane;
            uint32_t M[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) M[n] = desc[n * C + p];
            const uint32_t a0 = desc[NT * C + p];
            const uint32_t s = sp[a0];
            uint32_t n0, n1, n2 = 0;
            if constexpr (K == 3) {
                // u_k = s ^ g_k is the "bond k unsatisfied" word; for odd K the sum's planes are self-dual in s
                const uint32_t g1 = sp[desc[(NT + 1) * C + p]];
                const uint32_t g2 = sp[desc[(NT + 2) * C + p]];
                const uint32_t g3 = sp[desc[(NT + 3) * C + p]];
                n0 = g1 ^ g2 ^ g3 ^ s;
                n1 = ((g1 & g2) | (g3 & (g1 ^ g2))) ^ s;
            } else {
                n0 = 0; n1 = 0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const uint32_t u = sp[desc[(NT + 1 + k) * C + p]] ^ s;
                    const uint32_t c0 = n0 & u;
                    n0 ^= u;
                    const uint32_t c1 = n1 & c0;
                    n1 ^= c0;
                    n2 ^= c1;
                }
            }
            // classes n = 0..NT-1 have dE > 0 and need u < T_n; every other class is accepted (RRRMC.jl:39)
            uint32_t rej = 0;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                uint32_t e = ~M[n];
                e &= (n & 1) ? n0 : ~n0;
                e &= (n & 2) ? n1 : ~n1;
                if (K > 3) e &= (n & 4) ? n2 : ~n2;
                rej |= e;
            }
            const uint32_t acc = ~rej;
            const uint32_t snew = s ^ acc;      // spinflip! + update_cache! (Interface.jl:89-92, RRG.jl:191-234)
            sp[a0] = snew;
            sp[a0 + N] = ~snew;
            tal[p] = make_uint4(acc, acc & n0, acc & n1, acc & n2);
        }
    }
}

// Tally of one chunk, lane = replica: every wave takes a share of the slots.  accA counts accepted moves,
// accS the sum over accepted moves of n (unsatisfied bonds before the flip): dE = 2(K - 2n) (RRG.jl:236-244).
__device__ __forceinline__ void tally_chunk(const ChunkDesc& cd, const uint4* __restrict__ tal, int tid, uint32_t& accA, uint32_t& accS)
{
    const int r = tid & 31;
    const int part = tid >> 5, nparts = kSweepThreads >> 5;
    for (int p = part; p < (int)cd.count; p += nparts) {
        const uint4 w = tal[p];
        accA += (w.x >> r) & 1u;
        accS += ((w.y >> r) & 1u) + 2u * ((w.z >> r) & 1u) + 4u * ((w.w >> r) & 1u);
    }
}

template <int K>
__global__ __launch_bounds__(kSweepThreads) void sweep_kernel(SweepParams P)
{
    constexpr int NF = SweepCfg<K>::NF;
    extern __shared__ uint32_t lds[];
    const int N = P.N, C = P.C;
    uint32_t* sp = lds;                                         // [2N]  words, then complements
    uint32_t* desc = sp + ((2 * N + 3) & ~3);                   // [2][NF][C]  (16-byte aligned)
    uint4* tal = reinterpret_cast<uint4*>(desc + 2 * NF * C);   // [2][C]
    uint32_t* red = reinterpret_cast<uint32_t*>(tal + 2 * C);   // [64] sample-time reduction
    uint16_t* tbl = reinterpret_cast<uint16_t*>(red + 64);      // [N][TS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t group = P.group0 + blockIdx.x;
    uint32_t* gsp = P.spins + (size_t)blockIdx.x * N;

    for (int x = tid; x < N; x += kSweepThreads) {
        const uint32_t w = gsp[x];
        sp[x] = w;
        sp[x + N] = ~w;
    }
    for (int q = tid; q < N * P.TS; q += kSweepThreads) tbl[q] = P.table[q];
    if (tid < 64) red[tid] = 0;
    __syncthreads();

    uint32_t accA = 0, accS = 0;         // per-thread partial tallies since the last sample
    int32_t E_run = 0;                   // threads 0..31: running energy of replica tid
    int64_t A_run = 0;
    if (tid < 32) { E_run = P.E_cur[blockIdx.x * 32 + tid]; A_run = P.acc_cur[blockIdx.x * 32 + tid]; }
    int64_t ns = P.sample0;

    for (int c = 0; c <= P.nchunks; ++c) {
        if (wave == 0) {
            if (c >= 1) consume_chunk<K>(P, P.chunks[c - 1], desc + ((c - 1) & 1) * NF * C, sp, tal + ((c - 1) & 1) * C, lane);
        } else if (c < P.nchunks) {
            produce_chunk<K>(P, P.chunks[c], desc + (c & 1) * NF * C, tbl, wave - 1, lane, group);
        }
        __syncthreads();
        if (c >= 1) tally_chunk(P.chunks[c - 1], tal + ((c - 1) & 1) * C, tid, accA, accS);
        // An energy sample is due before chunk c (RRRMC.jl:104-108); partial tallies are also folded every 256
        // chunks so that the 32-bit partials cannot overflow on very long sample intervals.
        const bool sample = (c < P.nchunks) && (P.chunks[c].flags & kChunkSampleBefore);
        const bool flush = sample || (c == P.nchunks) || ((c & 255) == 255);
        if (flush) {
            atomicAdd(&red[tid & 31], accA);
            atomicAdd(&red[32 + (tid & 31)], accS);
            accA = 0; accS = 0;
            __syncthreads();
            if (tid < 32) {
                const int32_t a = (int32_t)red[tid], sn = (int32_t)red[32 + tid];
                E_run += 2 * (K * a - 2 * sn);        // sum over accepted moves of dE = 2(K - 2n)
                A_run += a;
                red[tid] = 0; red[32 + tid] = 0;
                if (sample && P.Es) P.Es[ns * P.Rpad + blockIdx.x * 32 + tid] = E_run;
            }
            if (sample) ns += 1;
            __syncthreads();
        }
    }
    if (tid < 32) { P.E_cur[blockIdx.x * 32 + tid] = E_run; P.acc_cur[blockIdx.x * 32 + tid] = A_run; }
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
