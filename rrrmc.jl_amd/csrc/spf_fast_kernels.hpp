// gfx950 kernel for the opt-in FAST standardMC (src/RRRMC.jl:81-127) on the Float64-coupling sparse models GraphRRGNormal /
// GraphEANormal (src/graphs/RRG.jl:503-627, src/graphs/EA.jl:534-680), K <= 4, N <= 8192.
//
// The bit-exact kernel for these models (spf_kernels.hpp) is the literal picture of the reference: one Float64 field per spin and
// replica, one lane per replica — bound by the HBM traffic of whole 512-byte lines for a quarter of useful lanes.  This one runs
// the +-J machinery of sparse_kernels.hpp instead — replicas bit-sliced 32 to a word, the state of a replica group resident in
// LDS, the site stream planned into dependency levels (plan_kernel, shared) — on the observation that delta_energy of site i
//   dE_i = 2 sigma_i sum_k J_ik sigma_k = 2 sum_k (u_k ? -|J_ik| : +|J_ik|),      u_k = "bond k unsatisfied",
// takes one of 2^K values per site, and the pattern ~u has exactly -dE: the 2^(K-1) pattern PAIRS c (patterns u and ~u, indexed by
// x_k = u_1 ^ u_k, k = 2..K) each need ONE acceptance threshold T_c(i) = ceil(exp(-beta |dE_c(i)|) 2^64), applied on the side
// (u_1 = 0 or 1) whose dE is positive; the other side is always accepted (src/RRRMC.jl:39).  So:
//   * no local field is stored or updated: update_cache! (RRG.jl:576-617) collapses to the XOR of the site's word;
//   * the producers compare the 32 replicas' uniforms (ACCEPT stream, bit planes) with the NT = 2^(K-1) per-SITE thresholds
//     (threshold words are per lane here, not wave-uniform scalars: one v_bfe_i32 per plane and threshold more than the +-J kernel);
//   * the consumer picks, per replica, the mask of its pattern pair;
//   * the energy is not tracked per move (a Float64 sum per replica cannot be bit-sliced): at every sample point all 16 waves
//     of the workgroup re-evaluate E = sum_bonds |J| (unsatisfied ? +1 : -1) from the LDS-resident spins: bit transposes + host-made
//     tables of the 256 partial sums of every 8 bonds, in a fixed order.
// Parity contract (tests/test_gpu_spf_fast.py): configurations and accepted counts identical to the oracle's restatement of this mode
// (orc_standard_mc_spf_fast: same streams, same thresholds, computed on the host with libm exp by both); energies within 1e-9
// relative of the oracle's tracked Float64 energy (the north star asks 1e-6 for Float64 models).  The mode differs from the default
// kernel in the acceptance stream (bit planes instead of one 53-bit uniform per replica) and in reading dE from the pattern sum
// instead of the incrementally updated cache (last-bit differences): it is opt-in (rrrmc_standard_mc_fast_async).
#pragma once
#include "sparse_kernels.hpp"

namespace rrrmc {

constexpr int kFastMaxK = 4;

// role layout of spf_fast_kernel: consumer (wave 0), tally (1), a dedicated fixer wave and 13 producers
constexpr int kFastFixerWave = 13;
constexpr int kFastProducerWaves = kSweepThreads / kWave - 3;
constexpr int kFastProducerTasksMax = (kMaxChunkSlots / kWave + kFastProducerWaves - 1) / kFastProducerWaves;

template <int K> struct FastCfg {
    static constexpr int NT = 1 << (K - 1);              // pattern pairs = thresholds per site
    static constexpr int NOFF = (K + 2) / 2;             // words holding the K+1 16-bit byte offsets
    static constexpr int NW = NT + NOFF + 1;             // + one word: bit c = the u_1 value of pair c's tested side
    static constexpr int NQ = (NW + 3) / 4;
};

struct FastParams {
    uint32_t* spins32;        // the model's native spin buffer ([W][N] 64-bit words) seen as 32-bit halves: group g, site x at ((g >> 1) N + x) 2 + (g & 1)
    const uint16_t* table;    // [N][TS] byte offsets into the LDS pair array: 8 y (J > 0) or 8 y + 4 (J < 0: the complemented copy)
    const uint32_t* thr_hi;   // [N][NT] bits 63..32 of the thresholds
    const uint32_t* thr_lo;   // [N][NT] bits 31..0 (only the fixer's late planes read them)
    const uint32_t* flags;    // [N] bit c: pair c is accepted on both sides (dE = 0); bit 8 + c: u_1 of the tested side of pair c
    const uint32_t* bond_off; // [nblk * 32] every bond once: byte offset of word s_x | byte offset of the (sign-folded) word of y << 16
    const double* etab;       // [nblk][4][256] sum of +-|J| over 8 consecutive bonds by their unsatisfied-bits byte (the energy phase)
    int nblk;                 // blocks of 32 bonds, a multiple of 32 (padded with zero-weight bonds)
    const ChunkDesc* chunks;
    const uint32_t* slots;
    const uint32_t* vecs;
    double* Es;               // [nsamples][Rpad]
    int64_t* acc_cur;         // [Rpad]
    uint32_t k0, k1, group0;
    int64_t sample0;
    uint64_t gbase;
    int N, C, TS, Rpad, nchunks;
};

// planes 4 pb .. 4 pb + 3 of the comparison u < T_n, thresholds per lane (64-bit words hi:lo)
template <int NT>
__device__ __forceinline__ void refine_block_lane(uint32_t (&lt)[NT], uint32_t (&eq)[NT], const Philox4& o, uint32_t pb,
                                                  const uint32_t (&thi)[NT], const uint32_t (&tlo)[NT])
{
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        uint32_t x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t plane = pb * 4u + (uint32_t)j;          // 0 = most significant bit
            const uint32_t src = plane < 32u ? thi[n] : tlo[n];
            const uint32_t tm = (uint32_t)((int32_t)(src << (plane & 31u)) >> 31);
            x[j] = bitop3<0x20>(eq[n], o.w[j], tm);
            eq[n] = bitop3<0x90>(eq[n], o.w[j], tm);
        }
        lt[n] = (lt[n] | x[0] | x[1]) | (x[2] | x[3]);
    }
}

template <int K>
__device__ __forceinline__ void fast_produce_chunk(const FastParams& P, const ChunkDesc& cd, uint4* __restrict__ desc, const LeftList& left,
                                                   int pw, int lane, uint32_t group, const uint32_t (&slots)[kFastProducerTasksMax])
{
    constexpr int NT = FastCfg<K>::NT, NQ = FastCfg<K>::NQ, NOFF = FastCfg<K>::NOFF;
    const int C = P.C;
    const int ntask = ((int)cd.count + kWave - 1) / kWave;
#pragma unroll
    for (int j = 0; j < kFastProducerTasksMax; ++j) {
        const int task = pw + j * kFastProducerWaves;
        if (task >= ntask) break;
        const int p = task * kWave + lane;
        const bool live = p < (int)cd.count;
        const uint32_t slot = slots[j];
        const uint32_t site = slot & 0xffffu;
        const uint64_t g = P.gbase + cd.g0 + (uint64_t)(slot >> 16);

        uint32_t f[NQ * 4];
#pragma unroll
        for (int q = 0; q < NQ * 4; ++q) f[q] = 0u;
        {
            uint32_t off[2 * NOFF];
#pragma unroll
            for (int q = 0; q < 2 * NOFF; ++q) off[q] = 0u;
            off[0] = site * 8u;
            const uint16_t* row = P.table + (size_t)site * P.TS;
            const uint2 r = *reinterpret_cast<const uint2*>(row);
            const uint32_t e[4] = {r.x & 0xffffu, r.x >> 16, r.y & 0xffffu, r.y >> 16};
#pragma unroll
            for (int k = 0; k < K; ++k) off[1 + k] = e[k];
#pragma unroll
            for (int h = 0; h < NOFF; ++h) f[NT + h] = off[2 * h] | (off[2 * h + 1] << 16);
        }
        const uint32_t fl = P.flags[site];
        f[NT + NOFF] = fl >> 8;
        uint32_t thi[NT], tlo[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) { thi[n] = P.thr_hi[(size_t)site * NT + n]; tlo[n] = 0u; }

        uint32_t lt[NT], eq[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const bool always = (fl >> n) & 1u;
            lt[n] = always ? 0xffffffffu : 0u;
            eq[n] = (live && !always) ? 0xffffffffu : 0u;
        }
#pragma unroll
        for (int b = 0; b < kProducerBlocks; ++b) refine_block_lane<NT>(lt, eq, accept_planes(P.k0, P.k1, g, group, (uint32_t)b), (uint32_t)b, thi, tlo);
        bool need = any_set<NT>(eq);
        const unsigned long long bal = __ballot(need);
        if (bal != 0ull) {       // wave-uniform
            uint32_t base = 0u;
            if (lane == 0) base = atomicAdd(left.count, (uint32_t)__popcll(bal));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            const uint32_t idx = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            if (need && idx < (uint32_t)kLeftMax) {
                left.slot[idx] = (uint32_t)p | (slot & 0xffff0000u);
#pragma unroll
                for (int n = 0; n < NT; ++n) { left.lt[n * kLeftMax + idx] = lt[n]; left.eq[n * kLeftMax + idx] = eq[n]; }
                need = false;
#pragma unroll
                for (int n = 0; n < NT; ++n) eq[n] = 0u;
            }
            if (__any(need)) {       // list full (practically never): finish those lanes here
#pragma unroll
                for (int n = 0; n < NT; ++n) tlo[n] = P.thr_lo[(size_t)site * NT + n];
                for (uint32_t pb = (uint32_t)kProducerBlocks; pb < 16u; ++pb) {
                    if (!__any(any_set<NT>(eq))) break;
                    refine_block_lane<NT>(lt, eq, accept_planes(P.k0, P.k1, g, group, pb), pb, thi, tlo);
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

template <int K>
__device__ __forceinline__ void fast_fix_chunk(const FastParams& P, const ChunkDesc& cd, uint4* __restrict__ desc, const LeftList& left,
                                               int lane, uint32_t group)
{
    constexpr int NT = FastCfg<K>::NT;
    uint32_t n = *left.count;
    n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
    n = n < (uint32_t)kLeftMax ? n : (uint32_t)kLeftMax;
    for (uint32_t base = 0; base < n; base += kWave) {
        const uint32_t i = base + (uint32_t)lane;
        const bool live = i < n;
        uint32_t lt[NT], eq[NT], thi[NT], tlo[NT], sl = 0u;
        if (live) sl = left.slot[i];
        const uint32_t site = live ? (P.slots[cd.slot_base + (sl & 0xffffu)] & 0xffffu) : 0u;
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            lt[q] = live ? left.lt[q * kLeftMax + i] : 0u; eq[q] = live ? left.eq[q * kLeftMax + i] : 0u;
            thi[q] = P.thr_hi[(size_t)site * NT + q]; tlo[q] = P.thr_lo[(size_t)site * NT + q];
        }
        const uint64_t g = P.gbase + cd.g0 + (uint64_t)(sl >> 16);
        for (uint32_t pb = kProducerBlocks; pb < 16u; ++pb) {
            if (!__any(any_set<NT>(eq))) break;
            refine_block_lane<NT>(lt, eq, accept_planes(P.k0, P.k1, g, group, pb), pb, thi, tlo);
        }
        if (live) {
            // the masks are the first NT words of the slot's descriptor: word n of q-array n / 4
            const int C = P.C;
#pragma unroll
            for (int q = 0; q < NT; ++q) reinterpret_cast<uint32_t*>(desc + (q >> 2) * C + (sl & 0xffffu))[q & 3] = lt[q];
        }
    }
    if (lane == 0) *left.count = 0u;
}

// accept decision of one slot for the 32 replicas
template <int K>
__device__ __forceinline__ uint32_t fast_slot_logic(const uint32_t (&f)[FastCfg<K>::NQ * 4], uint32_t s, const uint32_t (&g)[K])
{
    constexpr int NT = FastCfg<K>::NT, NOFF = FastCfg<K>::NOFF;
    // g_k = neighbour word with the coupling's sign folded in (complemented copy for J < 0): u_k = s ^ g_k is "bond k unsatisfied"
    const uint32_t u1 = s ^ g[0];
    uint32_t x[K > 1 ? K - 1 : 1];
#pragma unroll
    for (int k = 1; k < K; ++k) x[k - 1] = g[0] ^ g[k];          // u_1 ^ u_k
    const uint32_t npw = f[NT + NOFF];
    uint32_t acc = 0u;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        uint32_t m = 0xffffffffu;                                 // replicas whose pattern pair is c
#pragma unroll
        for (int k = 1; k < K; ++k) m &= ((c >> (k - 1)) & 1) ? x[k - 1] : ~x[k - 1];
        const uint32_t np = (uint32_t)((int32_t)(npw << (31 - c)) >> 31);      // 0 / ~0: u_1 of the side that needs the test
        acc |= m & ((u1 ^ np) | f[c]);                             // untested side: accepted; tested side: u < T_c
    }
    return acc;
}

template <int K, int NR>
__device__ __forceinline__ void fast_consume_rows(const uint4* __restrict__ desc, uint32_t* __restrict__ sp, uint32_t* __restrict__ tal,
                                                  int C, int p0, int plast)
{
    constexpr int NQ = FastCfg<K>::NQ, NT = FastCfg<K>::NT;
    uint32_t f[NR][NQ * 4];
    uint32_t s[NR], g[NR][K];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        int p = p0 + j * kWave;
        if (j == NR - 1) p = p < plast ? p : plast;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const uint4 v = desc[q * C + p];
            f[j][4 * q] = v.x; f[j][4 * q + 1] = v.y; f[j][4 * q + 2] = v.z; f[j][4 * q + 3] = v.w;
        }
    }
    auto off_of = [&](int j, int i) -> uint32_t { const uint32_t w = f[j][NT + (i >> 1)]; return (i & 1) ? (w >> 16) : (w & 0xffffu); };
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        s[j] = lds_word<1>(sp, off_of(j, 0));
#pragma unroll
        for (int k = 0; k < K; ++k) g[j][k] = lds_word<1>(sp, off_of(j, 1 + k));
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const uint32_t acc = fast_slot_logic<K>(f[j], s[j], g[j]);
        int pt = p0 + j * kWave;
        if (j == NR - 1) pt = pt < plast ? pt : plast;
        lds_store_pair<1>(sp, off_of(j, 0), s[j] ^ acc);
        tal[pt] = acc;
    }
}

#ifndef RRRMC_FAST_ROWS
#define RRRMC_FAST_ROWS 8
#endif
constexpr int kFastRows = RRRMC_FAST_ROWS;          // rows of 64 slots per consumer batch (the planner is told: host_spf_fast.hpp)
template <int K, int LO, int HI>
__device__ __forceinline__ void fast_consume_dispatch(int rows, const uint4* __restrict__ desc, uint32_t* __restrict__ sp, uint32_t* __restrict__ tal,
                                                      int C, int p0, int plast)
{
    if constexpr (LO >= HI) fast_consume_rows<K, HI>(desc, sp, tal, C, p0, plast);
    else {
        constexpr int MID = (LO + HI) / 2;
        if (rows <= MID) fast_consume_dispatch<K, LO, MID>(rows, desc, sp, tal, C, p0, plast);
        else fast_consume_dispatch<K, MID + 1, HI>(rows, desc, sp, tal, C, p0, plast);
    }
}

template <int K>
__device__ __forceinline__ void fast_consume_chunk(const FastParams& P, const ChunkDesc& cd, const uint4* __restrict__ desc,
                                                   uint32_t* __restrict__ sp, uint32_t* __restrict__ tal, int lane, uint32_t first_vd)
{
    const int C = P.C;
    const uint32_t nb = cd.nvec;
    for (uint32_t v0 = 0; v0 < nb; v0 += kWave) {
        uint32_t myvd = first_vd;
        if (v0 > 0) myvd = (v0 + lane < nb) ? P.vecs[cd.slot_base + v0 + lane] : 0u;
        const uint32_t vn = (nb - v0) < (uint32_t)kWave ? (nb - v0) : (uint32_t)kWave;
        for (uint32_t v = 0; v < vn; ++v) {
            const uint32_t vd = __builtin_amdgcn_readlane(myvd, v);
            const int start = (int)(vd & 0xffffu), cm1 = (int)(vd >> 16);
            const int p0 = start + lane, plast = start + cm1;
            // one row first (the short tail levels), the rest by binary search over the instantiations (as consume_chunk, sparse_kernels.hpp)
            if (cm1 < kWave) fast_consume_rows<K, 1>(desc, sp, tal, C, p0, plast);
            else fast_consume_dispatch<K, 2, kFastRows>((cm1 >> 6) + 1, desc, sp, tal, C, p0, plast);
        }
    }
}

// accepted moves per replica: the acc words of a chunk through the bit-sliced counters of the +-J kernel (one stream)
__device__ __forceinline__ void fast_tally_chunk(TallyState<1>& t, const ChunkDesc& cd, const uint32_t* __restrict__ tal, int lane)
{
    for (int base = 0; base < (int)cd.count; base += 8 * kWave) {
        uint32_t x[8][1];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = base + j * kWave + lane;
            x[j][0] = p < (int)cd.count ? tal[p] : 0u;
        }
        tally_add8<1>(t, x);
    }
}

inline size_t fast_lds_bytes(int64_t N, int K, int C)
{
    const int NT = 1 << (K - 1), NQ = (NT + (K + 2) / 2 + 1 + 3) / 4;
    const size_t words = (size_t)((2 * N + 128 + 3) & ~3ll) + (size_t)3 * NQ * 4 * C + (size_t)2 * (C + 64) + (size_t)2 * (4 + kLeftMax * (1 + 2 * NT)) + 4;
    return words * 4 + (size_t)16 * 2 * 32 * 8;          // + the energy partials [16 waves][2 halves][32 replicas] Float64
}

// One workgroup = one group of 32 bit-sliced replicas; roles and pipeline as sweep_kernel (sparse_kernels.hpp):
//   produce(c) | fix(c-1) | consume(c-2) | tally(c-3),  one barrier per step; a step whose consumer chunk carries the sample flag
// starts with the energy phase (all waves, one extra barrier).
template <int K>
__global__ __launch_bounds__(kSweepThreads) void spf_fast_kernel(FastParams P)
{
    constexpr int NQ = FastCfg<K>::NQ, NT = FastCfg<K>::NT;
    extern __shared__ uint32_t lds[];
    const int N = P.N, C = P.C;
    uint32_t* sp = lds;                                              // [2N] pairs {s, ~s}
    uint4* desc = reinterpret_cast<uint4*>(sp + ((2 * N + 128 + 3) & ~3));   // [3][NQ][C]
    uint32_t* tal = reinterpret_cast<uint32_t*>(desc + 3 * NQ * C);  // [2][C + 64]
    uint32_t* leftmem = tal + 2 * (C + kWave);
    constexpr int kLeftWords = 4 + kLeftMax * (1 + 2 * NT);
    double* epart = reinterpret_cast<double*>(leftmem + 2 * kLeftWords + ((2 * kLeftWords) & 1) + 2);   // [16][2][32], 8-byte aligned (see fast_lds_bytes: + 4 words of slack)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t group = P.group0 + blockIdx.x;
    uint32_t* gsp = P.spins32 + ((size_t)(blockIdx.x >> 1) * N) * 2 + (blockIdx.x & 1);

    for (int x = tid; x < N; x += kSweepThreads) {
        const uint32_t w = gsp[(size_t)x * 2];
        *reinterpret_cast<uint2*>(sp + 2 * x) = make_uint2(w, ~w);
    }
    if (tid < 2) leftmem[tid * kLeftWords] = 0u;
    __syncthreads();
    auto left_list = [&](int c) {
        uint32_t* m = leftmem + (c & 1) * kLeftWords;
        LeftList l;
        l.count = m; l.slot = m + 4; l.lt = m + 4 + kLeftMax; l.eq = m + 4 + kLeftMax * (1 + NT);
        return l;
    };

    // ---- energy phase: E = sum over bonds of |J| (unsatisfied ? +1 : -1), every bond once.  A half-wave takes 32 bonds at a time: lane =
    // bond computes the bond's unsatisfied word (32 replicas), a 32 x 32 bit transpose turns it into lane = replica with 32 bond bits,
    // and the four bytes index tables of the 256 possible partial sums of 8 bonds (host-made, L2-resident: the weights do not depend
    // on the state).  Partials to LDS; the tally wave adds the 32 partials of a replica in a fixed order after the barrier ----
    auto energy_partials = [&]() {
        TransposeConsts etc;
        etc.init(lane);
        const int hw = wave * 2 + (lane >> 5), rr = lane & 31;
        double e = 0.0;
        for (int b = hw; b < P.nblk; b += 32) {
            const uint32_t bo = P.bond_off[(size_t)b * 32 + rr];
            const uint32_t u = lds_word<1>(sp, bo & 0xffffu) ^ lds_word<1>(sp, bo >> 16);
            const uint32_t t = transpose32(u, etc);
            const double* tb = P.etab + (size_t)b * 4 * 256;
            e += tb[t & 255u];
            e += tb[256 + ((t >> 8) & 255u)];
            e += tb[512 + ((t >> 16) & 255u)];
            e += tb[768 + (t >> 24)];
        }
        epart[hw * 32 + rr] = e;
    };
    auto energy_emit = [&](int64_t sample) {          // tally wave, after the barrier
        if (lane < 32) {
            double e = 0.0;
            for (int q = 0; q < 32; ++q) e += epart[q * 32 + lane];
            if (P.Es) P.Es[(size_t)sample * P.Rpad + blockIdx.x * 32 + lane] = e;
        }
    };

    const int nsteps = P.nchunks + 3;
    const int tal_stride = C + kWave;
    // flags of the chunk the consumer works on in step c (chunk c - 2): every role needs them for the energy phase
    auto sample_step = [&](int c) -> bool { return c >= 2 && c - 2 < P.nchunks && (P.chunks[c - 2].flags & kChunkSampleBefore) != 0u; };

    if (wave == kConsumerWave) {
        __builtin_amdgcn_s_setprio(3);
        ChunkDesc ca{}, cb{}, cc{};
        uint32_t va = 0u, vb = 0u;
        for (int c = 0; c < nsteps; ++c) {
            ca = cb; va = vb;
            cb = cc;
            vb = (c >= 1 && c - 1 < P.nchunks && (uint32_t)lane < cb.nvec) ? P.vecs[cb.slot_base + lane] : 0u;
            if (c < P.nchunks) cc = P.chunks[c];
            if (sample_step(c)) { energy_partials(); __syncthreads(); }
            if (c >= 2 && c - 2 < P.nchunks)
                fast_consume_chunk<K>(P, ca, desc + ((c - 2) % 3) * NQ * C, sp, tal + ((c - 2) & 1) * tal_stride, lane, va);
            __syncthreads();
        }
    } else if (wave == kTallyWave) {
        __builtin_amdgcn_s_setprio(2);
        TransposeConsts tc;
        tc.init(lane);
        TallyState<1> ts;
        tally_reset<1>(ts);
        int64_t A_run = 0;
        if (lane < 32) A_run = P.acc_cur[blockIdx.x * 32 + lane];
        int64_t ns = P.sample0;
        const uint32_t grp_per_chunk = (uint32_t)((C + 8 * kWave - 1) / (8 * kWave));
        auto flush = [&]() {
            uint32_t a = tally_stream_total<1>(ts, 0, tc);
            a += (uint32_t)__shfl_xor((int)a, 32);
            A_run += a;
            tally_reset<1>(ts);
        };
        ChunkDesc ta{}, tb{};
        for (int c = 0; c < nsteps; ++c) {
            ta = tb;
            if (c >= 2 && c - 2 < P.nchunks) tb = P.chunks[c - 2];
            if (sample_step(c)) { energy_partials(); __syncthreads(); energy_emit(ns); ns += 1; }
            if (c >= 3) {
                if (ts.ngrp + grp_per_chunk > (1u << kTallyHi) - 1u) flush();
                fast_tally_chunk(ts, ta, tal + ((c - 3) & 1) * tal_stride, lane);
            }
            __syncthreads();
        }
        flush();
        if (lane < 32) P.acc_cur[blockIdx.x * 32 + lane] = A_run;
    } else if (wave == kFastFixerWave) {
        ChunkDesc fa{}, fb{};
        for (int c = 0; c < nsteps; ++c) {
            fa = fb;
            if (c < P.nchunks) fb = P.chunks[c];
            if (sample_step(c)) { energy_partials(); __syncthreads(); }
            if (c >= 1 && c - 1 < P.nchunks) fast_fix_chunk<K>(P, fa, desc + ((c - 1) % 3) * NQ * C, left_list(c - 1), lane, group);
            __syncthreads();
        }
    } else {
        const int pw = wave < kFastFixerWave ? wave - 2 : wave - 3;
        ChunkDesc cd0{}, cd1{}, cd2{};
        uint32_t sl0[kFastProducerTasksMax], sl1[kFastProducerTasksMax];
#pragma unroll
        for (int j = 0; j < kFastProducerTasksMax; ++j) { sl0[j] = 0u; sl1[j] = 0u; }
        auto fetch_slots = [&](const ChunkDesc& cd, uint32_t (&sl)[kFastProducerTasksMax]) {
#pragma unroll
            for (int j = 0; j < kFastProducerTasksMax; ++j) {
                const int p = (pw + j * kFastProducerWaves) * kWave + lane;
                sl[j] = p < (int)cd.count ? P.slots[cd.slot_base + p] : 0u;
            }
        };
        if (P.nchunks > 0) { cd1 = P.chunks[0]; fetch_slots(cd1, sl1); }
        if (P.nchunks > 1) cd2 = P.chunks[1];
        for (int c = 0; c < nsteps; ++c) {
            cd0 = cd1;
#pragma unroll
            for (int j = 0; j < kFastProducerTasksMax; ++j) sl0[j] = sl1[j];
            cd1 = cd2;
            if (c + 1 < P.nchunks) fetch_slots(cd1, sl1);
            if (c + 2 < P.nchunks) cd2 = P.chunks[c + 2];
            if (sample_step(c)) { energy_partials(); __syncthreads(); }
            if (c < P.nchunks) fast_produce_chunk<K>(P, cd0, desc + (c % 3) * NQ * C, left_list(c), pw, lane, group, sl0);
            __syncthreads();
        }
    }
    __syncthreads();
    int tid_wb = tid;
    asm volatile("" : "+v"(tid_wb));
    for (int x = tid_wb; x < N; x += kSweepThreads) gsp[(size_t)x * 2] = sp[2 * x];
}

}  // namespace rrrmc
