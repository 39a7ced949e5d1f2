// rrr_quant_wave_kernel: rrrMC(X::DoubleGraph) (src/RRRMC.jl:221-290) on GraphQuant (src/graphs/QT.jl:126-321) over GraphRRG / GraphEA
// slices, ONE WAVEFRONT PER REPLICA, the whole move-selection cache (DeltaECache{Float64,2} over four ArraySets, src/DeltaE.jl:63-295,
// src/ArraySets.jl:19-85) resident in LDS.  BASELINE.json config 5: 128 replicas of (Nk = 1024, M = 32) per GPU.
//
// The chain of a replica is sequential (rand_move depends on the state), so the wavefront executes it wave-uniformly and spends its
// lanes on what is independent INSIDE an iteration:
//   * the two Philox blocks of the RRR stream and their conversions to Float64, 64 iterations at a time (lane = iteration);
//   * the re-classification of the moved spin and of its two Trotter neighbours (lanes 0..2: before and after the flip,
//     QT.jl:86-108, DeltaE.jl:80-86) and the K bonds of the slice graph's delta_energy (lanes 3..3+K-1, RRG.jl:236-244);
//   * staging / write-back of the state.
// What the thread-per-replica kernel (rrr_quant_kernel) kept in memory and this one does not:
//   * DeltaECache.pos (the class of every spin): GraphQT's delta_energy is a function of three spin bits, the cache is consistent
//     between moves (DeltaE.jl:120-136), so the class before a flip is recomputed from the bits before the flip;
//   * T[1..3] of the three classes with weight 1: they are only ever changed by +-1.0, so they equal the set sizes exactly;
//     T[4] (weight exp(-beta fourK)) and z are the reference's running Float64 sums, updated in its order;
//   * the four ArraySet member arrays (8 bytes per spin in HBM): the sets partition the spins, so they live in ONE LDS array of
//     N + slack entries as four segments with gaps between them — push! appends into the gap behind its segment, delete! swaps with
//     the segment's last entry (ArraySets.jl:56-76): the order inside every set is the reference's.  A gap that runs low (rare: the
//     sizes fluctuate by a few hundred) re-spaces the segments through the replica's HBM arrays, between batches of 64 iterations;
//   * exp and the division by M: the slice graph's delta_energy takes 2K+1 values, so delta_energy_residual and det_exp(-beta dE1)
//     are tables; c = z / z' is only formed when the accept test needs it (c >= 1 is z >= z').
// Register discipline: a single wavefront issues one instruction every ~5 cycles whatever its kind, and the compiler moves every value
// it can prove wave-uniform into scalar registers — of which there are about a hundred; a chain with this much uniform state spills
// them into VGPR lanes and reloads them with v_readlane at every use (a quarter of the loop in the first version).  So the
// chain's state is deliberately kept in VECTOR registers (every lane holds the same value: `vz` below is an opaque zero that hides
// the uniformity); scalar registers are used only for short-lived lane indices.
// Float64 sums in the reference's order, det_exp shared with the oracle: bit-identical trajectories, energies, accepted / staged
// counts and final cache.
#pragma once
#include "rrr_kernels.hpp"

namespace rrrmc {

constexpr int kQwTabDoubles = 176;      // sparse slices: header 16, class-pair weights [16][4], exp[16], dE1[16], dE0(k) + dE1(a) [4][16]
constexpr int kQwMinGap = 512;          // smallest slack (entries per gap) the host accepts for this build: 6 pushes x 64 iterations fit

struct QwLayout { size_t off_spos, off_sv, off_A, off_J, off_rng, off_tab, bytes; int cap; };

// table area, in doubles: header [w3 4][fc 4][dE0 4][4 spare], the weights of a class change (k0 -> k1) [16][4], then
// exp[TE], dE1[TE] and, for GraphRRG / GraphEA slices, dE0 + dE1 [4][16]; TE = 16 entries (a = 0 .. 2K) or 2 Nk (binary GraphSK slices:
// the index is u + s_i Nk with u = |{j != i : J_ij xor s_j}|)
inline size_t qw_tab_doubles(int64_t Nk, bool sk) { return sk ? (size_t)(80 + 4 * Nk) : (size_t)kQwTabDoubles; }

// LDS layout for (N, W, Nk, K): cap = N + slack (slack as large as the LDS allows); cap < N + 4 * kQwMinGap means "does not fit".
// sk: binary GraphSK slices (GraphQSKT) — no neighbour table, larger acceptance tables
inline QwLayout qw_layout(int64_t N, int64_t W, int64_t Nk, int64_t K, size_t lds_limit, bool sk = false)
{
    QwLayout L{};
    size_t o = (size_t)W * 4;
    L.off_spos = o; o += (((size_t)N * 2 + 7) & ~(size_t)7);
    const size_t szA = sk ? 0 : (((size_t)Nk * K * 2 + 7) & ~(size_t)7), szJ = sk ? 0 : (((size_t)Nk * K + 7) & ~(size_t)7);
    const size_t fixed_tail = szA + szJ + 64 * 3 * 8 + qw_tab_doubles(Nk, sk) * 8 + 64;
    const size_t room = lds_limit > o + fixed_tail ? lds_limit - o - fixed_tail : 0;
    int64_t cap = (int64_t)(room / 2) & ~(int64_t)3;
    if (cap > 2 * N) cap = 2 * N;
    if (cap > 65532) cap = 65532;           // slots are stored in 16 bits
    L.cap = (int)cap;
    L.off_sv = o; o += (size_t)cap * 2;
    o = (o + 7) & ~(size_t)7;
    L.off_A = o; o += szA;
    L.off_J = o; o += szJ;
    L.off_rng = o; o += 64 * 3 * 8;
    L.off_tab = o; o += qw_tab_doubles(Nk, sk) * 8;
    L.bytes = o;
    return L;
}

struct QwExtra { int cap; uint32_t off_spos, off_sv, off_A, off_J, off_rng, off_tab; unsigned long long* stamps; };

__device__ __forceinline__ int qw_uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ double qw_unid(double x)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// sum of v over the 64 lanes, the same value in every lane: xor-butterflies inside each row of 16 lanes by DPP (quad_perm 1032 / 2301,
// row_half_mirror, row_mirror: no LDS traffic), the four row sums by v_readlane.  (32 same-address LDS atomics, the first version,
// serialise in the LDS unit: 3 200 cycles per standardMC iteration.)
__device__ __forceinline__ int qw_wave_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);      // quad_perm:[1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);      // quad_perm:[2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);     // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);     // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

// class a + 2 up of a spin from its bit and its two Trotter neighbours' bits (QT.jl:86-103, DeltaE.jl:80-86)
__device__ __forceinline__ int qw_class(int sk, int s1, int s2)
{
    const int d = (sk == s1) - (sk != s2);
    const int a = d != 0 ? 1 : 0;
    const int up = d > 0 || (d == 0 && sk == 1);
    return a + kQL * up;
}

// qw_class as a table: entry (sk | s1 << 1 | s2 << 2), two bits each
constexpr int qw_class_c(int sk, int s1, int s2)
{
    const int d = (sk == s1 ? 1 : 0) - (sk != s2 ? 1 : 0);
    return (d != 0 ? 1 : 0) + 2 * ((d > 0 || (d == 0 && sk == 1)) ? 1 : 0);
}
constexpr uint32_t qw_class_lut()
{
    uint32_t t = 0;
    for (int i = 0; i < 8; ++i) t |= (uint32_t)qw_class_c(i & 1, (i >> 1) & 1, (i >> 2) & 1) << (2 * i);
    return t;
}
constexpr uint32_t kQwClassLut = qw_class_lut();
static_assert(kQL == 2, "two levels of allDE(GraphQT)");

// SK = false: GraphRRG / GraphEA slices (neighbour table and couplings in LDS); SK = true: binary GraphSK slices (GraphQSKT, the
// reference's test_QIsing experiment): the slice's delta_energy is a popcount over the slice's words against row i of J (SK.jl:62-96),
// one word per lane, the row read from HBM/L2 as soon as the move is known
template <bool SK>
__global__ __launch_bounds__(kRrrThreads) void rrr_quant_wave_kernel(RrrParams P, QwExtra X)
{
    extern __shared__ uint32_t qw_lds[];
    unsigned char* lds8 = reinterpret_cast<unsigned char*>(qw_lds);
    const int lane = (int)threadIdx.x, r = (int)blockIdx.x;
    // LDS pointers are kept in vector registers too (an opaque zero is added to each: see "Register discipline" above)
    int vz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
    uint32_t* l_sp = reinterpret_cast<uint32_t*>(lds8 + vz);                                          // [W]
    uint16_t* l_spos = reinterpret_cast<uint16_t*>(lds8 + (X.off_spos + (uint32_t)vz));              // [N]
    uint16_t* l_sv = reinterpret_cast<uint16_t*>(lds8 + (X.off_sv + (uint32_t)vz));                  // [CAP] four segments
    uint16_t* l_A = reinterpret_cast<uint16_t*>(lds8 + (X.off_A + (uint32_t)vz));                    // [Nk][K]
    int8_t* l_J = reinterpret_cast<int8_t*>(lds8 + (X.off_J + (uint32_t)vz));                        // [Nk][K]
    double* l_rng = reinterpret_cast<double*>(lds8 + (X.off_rng + (uint32_t)vz));                    // [64][3]: class uniform, member u64 (as bits), accept uniform
    double* l_hdr = reinterpret_cast<double*>(lds8 + (X.off_tab + (uint32_t)vz));
    double* l_w3 = l_hdr;                                     // [4]: class k's share of the running sum T3 (ft1 for k = 3, else 0)
    double* l_fc = l_hdr + 4;                                 // [4]: class weights get_class_f (1, 1, 1, ft1)
    double* l_dE0 = l_hdr + 8;                                // [4]: delta_energy of GraphQT by class: -fourK, -0.0, 0.0, fourK (DeltaE.jl:80-86)
    const int TE = SK ? 2 * P.Nk : 16;
    double* l_tri = l_hdr + 16;                               // [k0 * 4 + k1][4]: w3(k0), w3(k1), f(k1) - f(k0), -; entry 0 is all zeros (an unchanged neighbour)
    double* l_exp = l_hdr + 80;                               // [TE]: det_exp(-beta dE1(a))
    double* l_dE1 = l_exp + TE;                               // [TE]: dE1(a) = delta_energy_residual: 2 (a - K) / M, or ((2 (2 s_i - 1)(Nk - 1 - 2u)) / sqrt(Nk)) / M
    double* l_dEt = l_dE1 + TE;                               // sparse slices only, [4][16]: dE0(k) + dE1(a), the energy change of an accepted move

    uint32_t* g_sp = P.spins + (size_t)r * P.W;
    uint8_t* g_cls = P.cls + (size_t)r * P.N;
    uint16_t* g_sv = static_cast<uint16_t*>(P.sv) + (size_t)r * 4 * P.N;        // the wave build is 16-bit only (N <= 65 535: the host checks)
    uint16_t* g_spos = static_cast<uint16_t*>(P.spos) + (size_t)r * P.N;
    int32_t* g_t = P.st + (size_t)r * 4;

    // ---- stage the replica ------------------------------------------------------------------------------------------------
    // segment bases and ends (base + size) live in ONE VGPR each (lane q = set q): a lookup by a class index is a v_readlane.
    // In LDS a spin's position is the ABSOLUTE slot in the segmented array (base of its set + ArraySet position): a delete / push
    // then needs the two segment ends only.
    int bv = 0, ev = lane < 4 ? g_t[lane] : 0;      // ev holds the sizes until respace() has placed the segments
    auto B_ = [&](int q) -> int { return __builtin_amdgcn_readlane(bv, q); };
    auto E_ = [&](int q) -> int { return __builtin_amdgcn_readlane(ev, q); };
    auto T_ = [&](int q) -> int { return __builtin_amdgcn_readlane(ev, q) - __builtin_amdgcn_readlane(bv, q); };
    auto respace = [&](int t0, int t1, int t2, int t3) {      // equal gaps behind the four segments
        const int gap = (X.cap - P.N) / 4;
        bv = lane == 1 ? t0 + gap : lane == 2 ? t0 + t1 + 2 * gap : lane == 3 ? t0 + t1 + t2 + 3 * gap : 0;
        ev = bv + (lane == 0 ? t0 : lane == 1 ? t1 : lane == 2 ? t2 : lane == 3 ? t3 : 0);
    };
    respace(E_(0), E_(1), E_(2), E_(3));
    for (int i = lane; i < P.W; i += kRrrThreads) l_sp[i] = g_sp[i];
    {
        const int b0 = B_(0), b1 = B_(1), b2 = B_(2), b3 = B_(3);
        for (int i = lane; i < P.N; i += kRrrThreads) {
            const int c = g_cls[i];
            l_spos[i] = (uint16_t)((int)g_spos[i] + (c == 0 ? b0 : c == 1 ? b1 : c == 2 ? b2 : b3));
        }
    }
    for (int k = 0; k < 4; ++k) {
        const int tk = T_(k), bk = B_(k);
        for (int i = lane; i < tk; i += kRrrThreads) l_sv[bk + i] = g_sv[(size_t)k * P.N + i];
    }
    if (lane < 4) {
        l_w3[lane] = lane == 3 ? P.ft1 : 0.0; l_fc[lane] = lane == 3 ? P.ft1 : 1.0;
    }
    if (lane < 16) {
        const int k0 = lane >> 2, k1 = lane & 3;
        const double f0 = k0 == 3 ? P.ft1 : 1.0, f1 = k1 == 3 ? P.ft1 : 1.0;
        l_tri[lane * 4 + 0] = k0 == 3 ? P.ft1 : 0.0;
        l_tri[lane * 4 + 1] = k1 == 3 ? P.ft1 : 0.0;
        l_tri[lane * 4 + 2] = f1 - f0;
        l_tri[lane * 4 + 3] = 0.0;
    }
    if (lane < 4) {
        l_dE0[lane] = lane == 0 ? -0.0 : lane == 1 ? -P.fourK : lane == 2 ? 0.0 : P.fourK;
    }
    uint32_t xge0_s = 0u;                   // sparse slices, bit a: x = -beta dE1(a) >= 0
    if constexpr (SK) {
        for (int idx = lane; idx < 2 * P.Nk; idx += kRrrThreads) {
            const int si = idx >= P.Nk ? 1 : 0, u = idx - si * P.Nk;
            const int d = 2 * (2 * si - 1) * (P.Nk - 1 - 2 * u);              // lfields[i] (SK.jl:62-96), see slice_delta (rrr_kernels.hpp)
            const double dE1 = ((double)d / P.sN) / (double)P.M;              // slice_res: QT.jl:270-281 over SK.jl:139
            l_exp[idx] = det_exp(-P.beta * dE1);
            l_dE1[idx] = dE1;
        }
    } else {
        for (int i = lane; i < P.Nk * P.K; i += kRrrThreads) { l_A[i] = (uint16_t)P.A[i]; l_J[i] = P.J[i]; }
        if (lane <= P.K) {
            const double dE1 = (double)(2 * (2 * lane - P.K)) / (double)P.M;          // slice_res(slice_delta): sum_q +-J = 2 lane - K
            l_exp[2 * lane] = det_exp(-P.beta * dE1);
            l_dE1[2 * lane] = dE1;
            // delta_energy of the move = dE0 (class k: -fourK, -0.0 / 0.0, fourK, DeltaE.jl:80-86) + dE1, added in that order (QT.jl:283-286)
            for (int k = 0; k < 4; ++k) {
                const double dE0 = k == 0 ? -0.0 : k == 1 ? -P.fourK : k == 2 ? 0.0 : P.fourK;
                l_dEt[k * 16 + 2 * lane] = dE0 + dE1;
            }
        }
        for (int q = 0; q <= P.K; ++q) xge0_s |= (-P.beta * ((double)(2 * (2 * q - P.K)) / (double)P.M) >= 0 ? 1u : 0u) << (2 * q);
    }
    __syncthreads();

    // ---- the chain's state, in vector registers (see "Register discipline" above) --------------------------------------------------
    const int N = P.N + vz, Nk = P.Nk + vz, K = P.K + vz;
    const uint32_t xge0 = xge0_s + (uint32_t)vz;
    const uint32_t nk_magic = (uint32_t)((0x100000000ull + (uint32_t)P.Nk - 1u) / (uint32_t)P.Nk) + (uint32_t)vz;
    const uint32_t rep = P.replica0 + (uint32_t)r;
    const int fmask = lane == 0 ? 4 : lane == 1 ? 2 : 1;       // which bit of (sj, s1, s2) is the moved spin, for lanes 0 (nb0), 1 (nb1), 2.. (move)
    double vzd = __longlong_as_double(((long long)vz << 32) | (uint32_t)vz);      // +0.0, opaque
    const double negbeta = -P.beta + vzd;
    const double lambda = P.lambda + vzd, one_m_lambda = (1 - P.lambda) + vzd, staged_thr = P.staged_thr + vzd;
    // T[0..2] == (double)t[0..2] exactly (weights 1.0); T3 and z are running sums (DeltaE.jl:90-103, 258-282)
    double T3 = P.T[(size_t)r * 4 + 3], z = P.zz[r], E = P.E_cur[r], acc_rate = P.acc_rate[r];
    long long accepted = P.stats[(size_t)r * 2], staged_its = P.stats[(size_t)r * 2 + 1];
    long long ns = 0, next_sample = P.samp0;

    auto bit_of = [&](int x) -> int { return (int)((l_sp[x >> 5] >> (x & 31)) & 1u); };
    // every condition of the chain is wave-uniform, but its operands are kept in vector registers on purpose (see the header), so the
    // compiler would wrap each `if` in an exec-mask region (save, branch if empty, restore); uni() turns it back into a scalar branch
    // with all lanes active inside
    // (the raw ballot builtin on the comparison, and the mask made opaque: HIP's __ballot(c) != 0 goes through a vector bool — v_cndmask,
    // v_cmp_ne — before it becomes the scalar condition it was)
    auto uni = [](bool c) -> bool { unsigned long long m = __builtin_amdgcn_ballot_w64(c); asm volatile("" : "+s"(m)); return m != 0ull; };
    // ArraySet delete!(S, j) + push!(D, j) (ArraySets.jl:56-76) on the segmented array; p = the slot of j (absolute), act = the move
    // happens at all (a Trotter neighbour whose class does not change is skipped, DeltaE.jl:257).  Branch-free: an inactive move reads
    // and discards.  Returns the element that took j's place (the old last of S; j itself when j was the last) so that the caller can
    // patch the slots it has cached.  The gap behind segment D has room (checked once per batch of 64 iterations).
    auto set_move = [&](int j, int S, int D, int p, bool act) -> int {
        const int sS = S, sD = D;               // wave-uniform scalars (v_readlane results)
        const int eS = __builtin_amdgcn_readlane(ev, sS) + vz;
        const int last = (int)l_sv[eS - 1];
        const int eD = __builtin_amdgcn_readlane(ev, sD) + vz;
        if (act) {               // wave-uniform (the callers pass ballots); all lanes store the same values to the same addresses
            l_sv[p] = (uint16_t)last;
            l_spos[last] = (uint16_t)p;
            l_sv[eD] = (uint16_t)j;
            l_spos[j] = (uint16_t)eD;
            ev = lane == sS ? eS - 1 : (lane == sD ? eD + 1 : ev);
        }
        return last;
    };
    // floor(u * t / 2^64) for t < 2^32 (rand(1:t) of the member pick, ArraySets.jl:83)
    auto mulhi_u64_u32 = [](unsigned long long u, uint32_t t) -> uint32_t {
        const unsigned long long lo = (unsigned long long)(uint32_t)u * t;
        const unsigned long long hi = (unsigned long long)(uint32_t)(u >> 32) * t + (lo >> 32);
        return (uint32_t)(hi >> 32);
    };

#ifdef RRRMC_STAMPS
    unsigned long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define QW_T(i) { const unsigned long long now_ = __builtin_readcyclecounter(); stamp[i] += now_ - t_last; t_last = now_; }
#else
#define QW_T(i)
#endif

    for (long long base_it = 0; base_it < P.iters; base_it += kRrrThreads) {
        // ---- the RRR draws of the next 64 iterations, one iteration per lane (state independent) -------------------------
        __syncthreads();
        {
            const uint64_t gl = P.g0 + (uint64_t)(base_it + 1 + (long long)lane);
            const Philox4 a = philox4x32_10((uint32_t)gl, (uint32_t)(gl >> 32), rep, TAG_RRR, P.k0, P.k1);
            const Philox4 b = philox4x32_10((uint32_t)gl, (uint32_t)(gl >> 32), rep, TAG_RRR | (1u << 8), P.k0, P.k1);
            l_rng[lane * 3 + 0] = (double)((((uint64_t)a.w[0] << 32) | a.w[1]) >> 11) * 0x1.0p-53;
            l_rng[lane * 3 + 1] = __longlong_as_double((long long)(((uint64_t)a.w[2] << 32) | a.w[3]));
            l_rng[lane * 3 + 2] = (double)((((uint64_t)b.w[0] << 32) | b.w[1]) >> 11) * 0x1.0p-53;
        }
        // A move pushes at most 3 entries into one set (6 with the undo), so a gap of 6 x 64 entries lasts the whole batch: the
        // segments are re-spaced here, outside the hot loop, when a gap has run low (rare: the set sizes fluctuate by a few hundred)
        {
            const int e0 = B_(1) - E_(0), e1 = B_(2) - E_(1), e2 = B_(3) - E_(2), e3 = X.cap - E_(3);
            const int emin = min(min(e0, e1), min(e2, e3));
            if (emin < 6 * kRrrThreads) {
                __syncthreads();
                const int t0 = T_(0), t1 = T_(1), t2 = T_(2), t3 = T_(3);
                for (int k = 0; k < 4; ++k) {
                    const int tk = T_(k), bk = B_(k);
                    for (int i = lane; i < tk; i += kRrrThreads) g_sv[(size_t)k * P.N + i] = l_sv[bk + i];
                }
                __syncthreads();
                respace(t0, t1, t2, t3);
                for (int k = 0; k < 4; ++k) {
                    const int tk = T_(k), bk = B_(k);
                    for (int i = lane; i < tk; i += kRrrThreads) {
                        const int x = g_sv[(size_t)k * P.N + i];
                        l_sv[bk + i] = (uint16_t)x;
                        l_spos[x] = (uint16_t)(bk + i);
                    }
                }
            }
        }
        __syncthreads();
        const int n_it = (int)(base_it + kRrrThreads < P.iters ? kRrrThreads : P.iters - base_it);
        double u_cls_n = l_rng[0], u_mem_n = l_rng[1], u_acc_n = l_rng[2];      // the draws of an iteration are requested one iteration ahead
#ifdef RRRMC_STAMPS
        unsigned long long t_last = __builtin_readcyclecounter();
#endif
        // the next sample point as an index into this batch (-1: not in it): a 32-bit compare per iteration instead of a 64-bit one
        auto sample_li = [&]() -> int { const long long d = next_sample - (base_it + 1); return d >= 0 && d < (long long)n_it ? (int)d : -1; };
        int li_s = sample_li();
        for (int li = 0; li < n_it; ++li) {
            if (li == li_s) {
                next_sample += P.step;
                if (lane == 0) P.Es[ns * P.R + r] = E;
                ns += 1;
                li_s = sample_li();
            }
            const double u_cls = u_cls_n, u_acc = u_acc_n;
            const unsigned long long u_mem = (unsigned long long)__double_as_longlong(u_mem_n);
            {
                const int nx = (li + 1 < n_it ? li + 1 : li) * 3;
                u_cls_n = l_rng[nx]; u_mem_n = l_rng[nx + 1]; u_acc_n = l_rng[nx + 2];
            }
            // rand_move (DeltaE.jl:146-167): class proportional to T (linear scan over the running sums cT), member uniform
            const int tvv = ev - bv;
            const int t0 = __builtin_amdgcn_readlane(tvv, 0) + vz, t1 = __builtin_amdgcn_readlane(tvv, 1) + vz, t2 = __builtin_amdgcn_readlane(tvv, 2) + vz;
            const double rr = u_cls * z;
            const double c0 = (double)t0, c1 = c0 + (double)t1, c2 = c1 + (double)t2, c3 = c2 + T3;
            int k = (rr < c0 ? 0 : 1) + (rr < c1 ? 0 : 1) + (rr < c2 ? 0 : 1);
            if (uni(!(rr < c3))) {      // r >= z by rounding: back off over the empty classes (DeltaE.jl:155-157)
                k = 3;
                if (T3 == 0) { k = 2; if (t2 == 0) { k = 1; if (t1 == 0) k = 0; } }
            }
            const int sk = __builtin_amdgcn_readfirstlane(k);
            const int move = (int)l_sv[(__builtin_amdgcn_readlane(bv, sk) + vz) + (int)mulhi_u64_u32(u_mem, (uint32_t)(__builtin_amdgcn_readlane(tvv, sk) + vz))];
            QW_T(0)

            // ---- lane-parallel part: classes of (nb0, nb1, move) before / after the flip, their set positions, the slice bonds ----
            int nb0 = move - Nk; if (nb0 < 0) nb0 += N;
            int nb1 = move + Nk; if (nb1 >= N) nb1 -= N;
            const int ks = (int)__umulhi((uint32_t)move, nk_magic), is = move - ks * Nk, off = ks * Nk;
            // lanes 0..2: the sites nb0, nb1, move; lanes 3..3+K-1: bond q = lane - 3 of row `is` of the slice graph
            const int myj = lane == 0 ? nb0 : lane == 1 ? nb1 : move;
            int j1 = myj - Nk; if (j1 < 0) j1 += N;
            int j2 = myj + Nk; if (j2 >= N) j2 -= N;
            const int qk = lane - 3;
            const bool isq = !SK && qk >= 0 && qk < K;
            int yq = 0, jq = 0, sk_pc = 0;
            if constexpr (!SK) {
                const int aidx = isq ? is * K + qk : 0;
                yq = (int)l_A[aidx];
                jq = (int)l_J[aidx];
            } else {
                // binary GraphSK slice: lane w holds word w of the slice (funnel-shifted out of the replica's bit vector: slices need not be
                // word aligned) against word w of row `is` of J (HBM/L2: the load is consumed after the re-classification below)
                const int nw = (Nk + 31) >> 5;
                const bool wl = lane < nw;
                const int b0 = off + 32 * (wl ? lane : 0), qw = b0 >> 5, sh = b0 & 31, rem = Nk - 32 * (wl ? lane : 0);
                uint32_t bits = l_sp[qw] >> sh;
                if (sh && 32 * (qw + 1) < N) bits |= l_sp[qw + 1] << (32 - sh);
                if (rem < 32) bits &= (1u << rem) - 1u;
                const uint32_t jw = wl ? P.Jb[(size_t)is * (size_t)P.Wk + (size_t)lane] : 0u;
                sk_pc = wl ? (int)__popc(bits ^ jw) : 0;
            }
            const int sj = bit_of(myj), s1 = bit_of(j1), s2 = bit_of(j2);
            // the class before and after the flip from an 8-entry table in a constant (qw_class over the three bits); which of a lane's
            // three bits is the moved spin is a lane constant: nb0's upper neighbour, nb1's lower neighbour, the spin itself (M > 2)
            const int cidx = sj | (s1 << 1) | (s2 << 2);
            const int my_k0 = (int)((kQwClassLut >> (2 * cidx)) & 3u);
            const int my_k1 = (int)((kQwClassLut >> (2 * (cidx ^ fmask))) & 3u);
            const int my_pos = (int)l_spos[myj];
            const int sy = bit_of(isq ? off + yq : 0);
            const int si = __builtin_amdgcn_readlane(sj, 2) + vz;                 // the moved spin, before the flip
            int ai;
            if constexpr (!SK) {
                // sum_q J_iq sigma_i sigma_y + K = 2 x (bonds with J_iq sigma_i sigma_y > 0): the table index (RRG.jl:236-244)
                ai = 2 * __popcll(__ballot(isq && ((si == sy) == (jq > 0)))) + vz;
            } else {
                // u = |{j != i : J_ij xor s_j}| = popcount - s_i (J_ii = 0: position i contributes s_i); index u + s_i Nk
                const int sc = qw_wave_sum(sk_pc) + vz;
                ai = (sc - si) + si * Nk;
            }
            // (the six classes are short-lived scalars: lane selects of the set moves, table indices)
            const int k0a = __builtin_amdgcn_readlane(my_k0, 0), k1a = __builtin_amdgcn_readlane(my_k1, 0);
            const int k0b = __builtin_amdgcn_readlane(my_k0, 1), k1b = __builtin_amdgcn_readlane(my_k1, 1);
            const int k0m = __builtin_amdgcn_readlane(my_k0, 2), k1m = __builtin_amdgcn_readlane(my_k1, 2);
            int pa = __builtin_amdgcn_readlane(my_pos, 0) + vz, pb = __builtin_amdgcn_readlane(my_pos, 1) + vz, pm = __builtin_amdgcn_readlane(my_pos, 2) + vz;
            // k1m == k0m +- 2 always (the moved spin changes direction, DeltaE.jl:275-276)
            QW_T(1)

            // accept(c, x) (RRRMC.jl:40-44) with c = z / zp, x = -beta dE1 (dE1 = delta_energy_residual, QT.jl:270-281): c >= 1 is z >= zp
            // (both positive), so the quotient is only formed when it is needed
            auto accept = [&](double zz, double zp) -> bool {
                bool xok;
                if constexpr (SK) xok = negbeta * l_dE1[ai] >= 0;      // x = -beta dE1 >= 0
                else xok = (xge0 >> ai) & 1u;
                if (uni(zz >= zp && xok)) return true;
                const double c = zz / zp;
                const double a = c * l_exp[ai];
                return a >= 1 || u_acc < a;
            };
            // class weights: 1.0, except exp(-beta fourK) for class 3; only T3 is a running sum (x - 0.0 and x + 0.0 leave T3 >= +0 as it is)
            auto flip_move = [&]() { l_sp[move >> 5] ^= 1u << (move & 31); };       // every lane: same word, same value
            const bool cha = k0a != k1a, chb = k0b != k1b;                   // scalar conditions

            // Float64 bookkeeping of one apply_move! in the reference's order (first neighbour, second neighbour, moved spin,
            // DeltaE.jl:257-282); a skipped neighbour adds +-0.0, which changes neither T3 (>= +0) nor z' (> 0)
            // (an unchanged neighbour has k0 == k1: its weight difference is +0.0 by itself, and class 0's share of T3 is 0.0)
            const double* ta = l_tri + 4 * (cha ? k0a * 4 + k1a : 0);
            const double* tb = l_tri + 4 * (chb ? k0b * 4 + k1b : 0);
            const double* tm = l_tri + 4 * (k0m * 4 + k1m);
            const double da0 = ta[0], da1 = ta[1], dza = ta[2];
            const double db0 = tb[0], db1 = tb[1], dzb = tb[2];
            const double dm0 = tm[0], dm1 = tm[1], dzm = tm[2];
            bool acc = false;
            if (uni(acc_rate < staged_thr)) {
                // staged branch: step_rrr (RRRMC.jl:131-138) = compute_staged! + compute_reverse_probabilities!, apply_staged! on acceptance
                staged_its += 1;
                double T3p = T3, zp = z;
                T3p -= da0; T3p += da1; zp += dza;
                T3p -= db0; T3p += db1; zp += dzb;
                T3p -= dm0; T3p += dm1; zp += dzm;
                if (uni(accept(z, zp))) {
                    flip_move();
                    const int la = set_move(nb0, k0a, k1a, pa, cha);
                    if (cha && la == nb1) pb = pa;
                    if (cha && la == move) pm = pa;
                    const int lb = set_move(nb1, k0b, k1b, pb, chb);
                    if (chb && lb == move) pm = pb;
                    set_move(move, k0m, k1m, pm, true);
                    T3 = T3p; z = zp;
                    E += SK ? l_dE0[k] + l_dE1[ai] : l_dEt[k * 16 + ai];
                    accepted += 1;
                    acc = true;
                }
                QW_T(2)
            } else {
                // direct branch: apply_move! (DeltaE.jl:232-295), undone by a second apply_move! on rejection
                flip_move();
                double zp = z;
                T3 -= da0; T3 += da1; zp += dza;
                T3 -= db0; T3 += db1; zp += dzb;
                T3 -= dm0; T3 += dm1; zp += dzm;
                {
                    const int la = set_move(nb0, k0a, k1a, pa, cha);
                    if (cha && la == nb1) pb = pa;
                    if (cha && la == move) pm = pa;
                    const int lb = set_move(nb1, k0b, k1b, pb, chb);
                    if (chb && lb == move) pm = pb;
                    set_move(move, k0m, k1m, pm, true);
                }
                QW_T(2)
                const bool ok = accept(z, zp);
                z = zp;
                QW_T(3)
                if (uni(ok)) {
                    E += SK ? l_dE0[k] + l_dE1[ai] : l_dEt[k * 16 + ai];
                    accepted += 1;
                    acc = true;
                } else {
                    // the undo: apply_move! again, from the flipped configuration (classes k1 -> k0, slots re-read)
                    flip_move();
                    double zq = z;
                    T3 -= da1; T3 += da0; zq -= dza;
                    T3 -= db1; T3 += db0; zq -= dzb;
                    T3 -= dm1; T3 += dm0; zq -= dzm;
                    const int my_q = (int)l_spos[myj];
                    int qa = __builtin_amdgcn_readlane(my_q, 0) + vz, qb = __builtin_amdgcn_readlane(my_q, 1) + vz, qm = __builtin_amdgcn_readlane(my_q, 2) + vz;
                    const int la = set_move(nb0, k1a, k0a, qa, cha);
                    if (cha && la == nb1) qb = qa;
                    if (cha && la == move) qm = qa;
                    const int lb = set_move(nb1, k1b, k0b, qb, chb);
                    if (chb && lb == move) qm = qb;
                    set_move(move, k1m, k0m, qm, true);
                    z = zq;
                    QW_T(4)
                }
            }
            acc_rate = acc_rate * one_m_lambda + (acc ? 1.0 : 0.0) * lambda;                 // RRRMC.jl:281
            QW_T(5)
        }
    }
#ifdef RRRMC_STAMPS
    if (r == 0 && lane == 0 && X.stamps) for (int i = 0; i < 8; ++i) X.stamps[i] = stamp[i];
#endif
#undef QW_T

    // ---- write the replica back: spins, sets, positions, sizes, the classes (recomputed: the cache is consistent), scalars --------
    __syncthreads();
    for (int i = lane; i < P.W; i += kRrrThreads) g_sp[i] = l_sp[i];
    {
        const int b0 = B_(0), b1 = B_(1), b2 = B_(2), b3 = B_(3);
        for (int i = lane; i < P.N; i += kRrrThreads) {
            int j1 = i - P.Nk; if (j1 < 0) j1 += P.N;
            int j2 = i + P.Nk; if (j2 >= P.N) j2 -= P.N;
            const int c = qw_class(bit_of(i), bit_of(j1), bit_of(j2));
            g_cls[i] = (uint8_t)c;
            g_spos[i] = (uint16_t)((int)l_spos[i] - (c == 0 ? b0 : c == 1 ? b1 : c == 2 ? b2 : b3));
        }
    }
    for (int k = 0; k < 4; ++k) {
        const int tk = T_(k), bk = B_(k);
        for (int i = lane; i < tk; i += kRrrThreads) g_sv[(size_t)k * P.N + i] = l_sv[bk + i];
    }
    if (lane < 4) g_t[lane] = ev - bv;
    if (lane == 0) {
        P.T[(size_t)r * 4 + 0] = (double)T_(0); P.T[(size_t)r * 4 + 1] = (double)T_(1); P.T[(size_t)r * 4 + 2] = (double)T_(2); P.T[(size_t)r * 4 + 3] = T3;
        P.zz[r] = z; P.E_cur[r] = E; P.acc_rate[r] = acc_rate;
        P.stats[(size_t)r * 2] = accepted; P.stats[(size_t)r * 2 + 1] = staged_its;
    }
}

// ---------------------------------------------------------------------------------------------------
// standardMC (src/RRRMC.jl:81-127) on a GraphQuant, one wavefront per replica — the Metropolis side of the reference's test_QIsing
// experiment (scripts/scripts.jl:766-863), which runs few replicas: one thread per replica (quant_standard_kernel) leaves the chip idle.
// The chain is state independent up to the accept test, so the wave prepares 64 iterations at a time, one per lane: the site (SITE
// stream), the acceptance uniform (ACCEPT_F64 stream) and — binary GraphSK slices — the site's row of J, fetched into LDS.  Then the
// 64 iterations run wave-uniformly: three spin bits for GraphQT's delta_energy, the slice's delta_energy lane-parallel (one word of the
// slice per lane against the row of J, or one bond of the neighbour table per lane), delta_energy and exp(-beta dE) from tables over
// (Trotter term, slice term), the bit flip.  Bit-identical to quant_standard_kernel: the tables hold the values of its own expressions.
// LDS: spins [W], rows [64][Wk] (SK), A / J (sparse slices), dE [3][TE], exp(-beta dE) [3][TE].
// ---------------------------------------------------------------------------------------------------
struct QsLayout { size_t off_rows, off_A, off_J, off_de, off_ex, bytes; int TE; };
inline QsLayout qs_layout(int64_t W, int64_t Nk, int64_t K, int64_t Wk, bool sk)
{
    QsLayout L{};
    size_t o = ((size_t)W * 4 + 7) & ~(size_t)7;
    L.off_rows = o; o += sk ? (size_t)64 * Wk * 4 : 0;
    L.off_A = o; o += sk ? 0 : (((size_t)Nk * K * 2 + 7) & ~(size_t)7);
    L.off_J = o; o += sk ? 0 : (((size_t)Nk * K + 7) & ~(size_t)7);
    L.TE = sk ? (int)(2 * Nk) : 16;
    L.off_de = o; o += (size_t)3 * L.TE * 8;
    L.off_ex = o; o += (size_t)3 * L.TE * 8;
    L.bytes = o;
    return L;
}
struct QsExtra { uint32_t off_rows, off_A, off_J, off_de, off_ex; int TE; };

template <bool SK>
__global__ __launch_bounds__(kRrrThreads) void quant_standard_wave_kernel(RrrParams P, QsExtra X)
{
    extern __shared__ uint32_t qs_lds[];
    unsigned char* lds8 = reinterpret_cast<unsigned char*>(qs_lds);
    const int lane = (int)threadIdx.x, r = (int)blockIdx.x;
    uint32_t* l_sp = qs_lds;                                                     // [W]
    uint32_t* l_rows = reinterpret_cast<uint32_t*>(lds8 + X.off_rows);           // [64][Wk]  (SK)
    uint16_t* l_A = reinterpret_cast<uint16_t*>(lds8 + X.off_A);                 // [Nk][K]   (sparse slices)
    int8_t* l_J = reinterpret_cast<int8_t*>(lds8 + X.off_J);
    double* l_de = reinterpret_cast<double*>(lds8 + X.off_de);                   // [3][TE]: delta_energy for (qt_delta + 1, slice index)
    double* l_ex = reinterpret_cast<double*>(lds8 + X.off_ex);                   // [3][TE]: det_exp(-beta delta_energy)
    const int TE = X.TE, N = P.N, Nk = P.Nk, K = P.K;
    uint32_t* g_sp = P.spins + (size_t)r * P.W;
    for (int i = lane; i < P.W; i += kRrrThreads) l_sp[i] = g_sp[i];
    if constexpr (!SK)
        for (int i = lane; i < Nk * K; i += kRrrThreads) { l_A[i] = (uint16_t)P.A[i]; l_J[i] = P.J[i]; }
    // the tables: exactly quant_standard_kernel's expressions (dE = (double)qt_delta * fourK + residual, x = -beta dE)
    for (int idx = lane; idx < 3 * TE; idx += kRrrThreads) {
        const int d0 = idx / TE - 1, a = idx - (d0 + 1) * TE;
        double res;
        if constexpr (SK) {
            const int si = a >= Nk ? 1 : 0, u = a - si * Nk;
            const int d = 2 * (2 * si - 1) * (Nk - 1 - 2 * u);
            res = ((double)d / P.sN) / (double)P.M;
        } else {
            res = (double)(2 * (a - K)) / (double)P.M;       // a = sum_q J sigma sigma + K (only even a <= 2K occur)
        }
        const double dE = (double)d0 * P.fourK + res;
        l_de[idx] = dE;
        l_ex[idx] = det_exp(-P.beta * dE);
    }
    __syncthreads();

    const uint32_t rep = P.replica0 + (uint32_t)r;
    const uint32_t nk_magic = (uint32_t)((0x100000000ull + (uint32_t)Nk - 1u) / (uint32_t)Nk);
    double E = P.E_cur[r];
    long long accepted = 0, ns = 0, next_sample = P.step;
    auto bit_of = [&](int x) -> int { return (int)((l_sp[x >> 5] >> (x & 31)) & 1u); };
    for (long long base_it = 0; base_it < P.iters; base_it += kRrrThreads) {
        const int n_it = (int)(base_it + kRrrThreads < P.iters ? kRrrThreads : P.iters - base_it);
        // lane j: iteration base_it + 1 + j
        const uint64_t gl = P.g0 + (uint64_t)(base_it + 1 + (long long)lane);
        const int mv_l = (int)site_of(P.k0, P.k1, gl, (uint32_t)N);
        const double u_l = rand53(P.k0, P.k1, gl, rep);
        const uint32_t u_lo = (uint32_t)(unsigned long long)__double_as_longlong(u_l), u_hi = (uint32_t)((unsigned long long)__double_as_longlong(u_l) >> 32);
        if constexpr (SK) {
            // rows of J of the 64 sites: lane w fetches word w of row j, j = 0 .. 63
            const int nw = (Nk + 31) >> 5;
            __syncthreads();
            for (int j = 0; j < n_it; ++j) {
                const int mvj = __builtin_amdgcn_readlane(mv_l, j);
                const int ksj = (int)__umulhi((uint32_t)mvj, nk_magic), isj = mvj - ksj * Nk;
                if (lane < nw) l_rows[j * P.Wk + lane] = P.Jb[(size_t)isj * (size_t)P.Wk + (size_t)lane];
            }
            __syncthreads();
        }
        for (int j = 0; j < n_it; ++j) {
            const long long it = base_it + 1 + j;
            if (it == next_sample) { next_sample += P.step; if (lane == 0) P.Es[ns * P.R + r] = E; ns += 1; }
            const int move = __builtin_amdgcn_readlane(mv_l, j);
            const uint32_t ulo = (uint32_t)__builtin_amdgcn_readlane((int)u_lo, j), uhi = (uint32_t)__builtin_amdgcn_readlane((int)u_hi, j);
            const double u = __longlong_as_double((long long)(((unsigned long long)uhi << 32) | ulo));
            const int j1 = move - Nk + (move < Nk ? N : 0), j2 = move + Nk - (move + Nk >= N ? N : 0);
            const int sk = bit_of(move), s1 = bit_of(j1), s2 = bit_of(j2);
            const int d0 = (sk == s1) - (sk != s2);                               // qt_delta (QT.jl:86-103)
            const int ks = (int)__umulhi((uint32_t)move, nk_magic), is = move - ks * Nk, off = ks * Nk;
            int a;
            if constexpr (SK) {
                const int nw = (Nk + 31) >> 5;
                const bool wl = lane < nw;
                const int b0 = off + 32 * (wl ? lane : 0), qw = b0 >> 5, sh = b0 & 31, rem = Nk - 32 * (wl ? lane : 0);
                uint32_t bits = l_sp[qw] >> sh;
                if (sh && 32 * (qw + 1) < N) bits |= l_sp[qw + 1] << (32 - sh);
                if (rem < 32) bits &= (1u << rem) - 1u;
                const int sc = qw_wave_sum(wl ? (int)__popc(bits ^ l_rows[j * P.Wk + lane]) : 0);
                a = (sc - sk) + sk * Nk;
            } else {
                const bool isq = lane < K;
                const int aidx = isq ? is * K + lane : 0;
                const int yq = (int)l_A[aidx], jq = (int)l_J[aidx];
                const int sy = bit_of(isq ? off + yq : 0);
                a = 2 * __popcll(__ballot(isq && ((sk == sy) == (jq > 0))));
            }
            const int ci = (d0 + 1) * TE + a;
            const double dE = l_de[ci];
            const double x = -P.beta * dE;
            const bool acc = (x >= 0.0) || (u < l_ex[ci]);                        // RRRMC.jl:39
            if (__ballot(acc) != 0ull) {                                          // wave-uniform
                l_sp[move >> 5] ^= 1u << (move & 31);                             // every lane: same word, same value
                E += dE;
                accepted += 1;
            }
        }
    }
    __syncthreads();
    for (int i = lane; i < P.W; i += kRrrThreads) g_sp[i] = l_sp[i];
    if (lane == 0) {
        P.E_cur[r] = E;
        P.stats[(size_t)r * 2] = accepted; P.stats[(size_t)r * 2 + 1] = 0;
    }
}

}  // namespace rrrmc
