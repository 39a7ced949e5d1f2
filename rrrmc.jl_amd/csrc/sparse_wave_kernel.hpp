// rrrMC(X::SingleGraph) (src/RRRMC.jl:149-219) and bklMC (:311-359) on GraphRRG / GraphEA with integer levels, one WAVEFRONT per replica — the reference's own
// experiment (scripts/scripts.jl:test_RRG) runs a handful of chains, and one thread per replica (rrr_sparse_kernel) leaves the chip idle
// while each of its set moves waits on HBM/L2.  Same construction as rrr_quant_wave_kernel (quant_wave_kernel.hpp):
//   * the chain is executed wave-uniformly; the 64 lanes share what is independent inside an iteration: the RRR draws of 64 iterations
//     (one per lane), and the re-classification of the moved spin and its neighbours (lane q = neighbour q, lane K = the spin itself:
//     delta_energy before and after the flip from the K neighbour bits, the class, the set position);
//   * the DeltaECache lives in LDS: the 2L ArraySets as segments of ONE array with gaps (push! appends into the gap, delete! swaps with
//     the segment's last entry: the reference's order inside every set), a spin's position stored as the absolute slot; the segments are
//     re-spaced between 64-iteration batches when a gap runs low; no class array (the class before a flip is recomputed from the bits);
//   * T[k] is one lane-vector of Float64 running sums (lane k), updated in the reference's order (neighbours in table order, then the
//     moved spin, DeltaE.jl:232-295); a skipped neighbour adds nothing;
//   * conditions are scalar (ballots), data stay in vector registers (the "register discipline" of quant_wave_kernel.hpp).
// The cache is built from the spins at the start of every call, in site order (RRRMC.jl:177-178, DeltaE.jl:74-103), as rrr_sparse_kernel does.
// Bit-identical to rrr_sparse_kernel and to the oracle (tests/test_gpu_rrr_parity.py).
#pragma once

#include "quant_wave_kernel.hpp"

namespace rrrmc {

struct SwLayout { size_t off_spos, off_sv, off_A, off_J, off_rng, off_tab, bytes; int cap; };
constexpr int kSwTabDoubles = 32;       // f[8], dE[8] (as doubles), spare

__host__ __device__ inline int sw_min_gap(int64_t K);
// LDS layout for (N, W, K, L): cap = N + slack — four times the smallest gap per segment where the LDS allows it, so that small
// graphs leave room for several replicas per CU
inline SwLayout sw_layout(int64_t N, int64_t W, int64_t K, int64_t Lv, size_t lds_limit)
{
    SwLayout L{};
    size_t o = (size_t)W * 4;
    L.off_spos = o; o += (((size_t)N * 2 + 7) & ~(size_t)7);
    const size_t szA = (((size_t)N * K * 2 + 7) & ~(size_t)7), szJ = (((size_t)N * K + 7) & ~(size_t)7);
    const size_t fixed_tail = szA + szJ + 64 * 3 * 8 + kSwTabDoubles * 8 + 64;
    const size_t room = lds_limit > o + fixed_tail ? lds_limit - o - fixed_tail : 0;
    int64_t cap = (int64_t)(room / 2) & ~(int64_t)3;
    const int64_t want = (N + 2 * Lv * 4 * (int64_t)(64 * 2 * (K + 1)) + 3) & ~(int64_t)3;
    if (cap > want) cap = want;
    if (cap > 65532) cap = 65532;           // slots are stored in 16 bits
    L.cap = (int)cap;
    L.off_sv = o; o += (size_t)cap * 2;
    o = (o + 7) & ~(size_t)7;
    L.off_A = o; o += szA;
    L.off_J = o; o += szJ;
    L.off_rng = o; o += 64 * 3 * 8;
    L.off_tab = o; o += kSwTabDoubles * 8;
    L.bytes = o;
    return L;
}
// entries one set can gain in a batch of 64 iterations: K + 1 pushes per apply_move!, twice with the undo
__host__ __device__ inline int sw_min_gap(int64_t K) { return (int)(64 * 2 * (K + 1)); }

struct SwExtra { int cap; uint32_t off_spos, off_sv, off_A, off_J, off_rng, off_tab; };

// SL = compile-time bound on the levels of allDE (2: GraphRRG K <= 3; 4: K <= 7, GraphEA in three dimensions): 2 SL classes;
// KM = compile-time bound on K (the neighbour loops are unrolled: all table reads of a site go out together, then all bit reads)
template <int SL, int KM>
__global__ __launch_bounds__(kRrrThreads) void rrr_sparse_wave_kernel(RrrSparseParams P, SwExtra X)
{
    constexpr int C2 = 2 * SL;
    extern __shared__ uint32_t sw_lds[];
    unsigned char* lds8 = reinterpret_cast<unsigned char*>(sw_lds);
    const int lane = (int)threadIdx.x, r = (int)blockIdx.x;
    int vz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
    uint32_t* l_sp = reinterpret_cast<uint32_t*>(lds8 + vz);                                          // [W]
    uint16_t* l_spos = reinterpret_cast<uint16_t*>(lds8 + (X.off_spos + (uint32_t)vz));              // [N] absolute slot
    uint16_t* l_sv = reinterpret_cast<uint16_t*>(lds8 + (X.off_sv + (uint32_t)vz));                  // [cap] 2L segments
    uint16_t* l_A = reinterpret_cast<uint16_t*>(lds8 + (X.off_A + (uint32_t)vz));                    // [N][K]
    int8_t* l_J = reinterpret_cast<int8_t*>(lds8 + (X.off_J + (uint32_t)vz));                        // [N][K]
    double* l_rng = reinterpret_cast<double*>(lds8 + (X.off_rng + (uint32_t)vz));                    // [64][3]
    double* l_f = reinterpret_cast<double*>(lds8 + (X.off_tab + (uint32_t)vz));                      // [8] class weights get_class_f (DeltaE.jl:91)
    int* l_dE = reinterpret_cast<int*>(l_f + 8);                                                     // [8] delta_energy of class k, level units

    const int K = P.K + vz, L = P.L + vz, K2 = 2 * P.L + vz;
    uint32_t* g_sp = P.spins + (size_t)r * P.W;
    for (int i = lane; i < P.W; i += kRrrThreads) l_sp[i] = g_sp[i];
    for (int i = lane; i < P.N * P.K; i += kRrrThreads) { l_A[i] = (uint16_t)P.A[i]; l_J[i] = P.J[i]; }
    if (lane < C2) {
        const bool up = lane >= P.L;
        const int a = up ? lane - P.L : lane;
        l_f[lane] = (lane < 2 * P.L && up) ? P.ft[a < kSLmax ? a : 0] : 1.0;
        l_dE[lane] = lane < 2 * P.L ? (up ? P.lv.dElist[a] : -P.lv.dElist[a]) : 0;
    }
    __syncthreads();

    auto uni = [](bool c) -> bool { return __ballot(c) != 0ull; };
    auto bit_of = [&](int x) -> int { return (int)((l_sp[x >> 5] >> (x & 31)) & 1u); };
    // findk (DeltaE.jl:28-60: exact comparison of |dE|) and the class a + L up (DeltaE.jl:80-86)
    int lev[SL];
#pragma unroll
    for (int k = 0; k < SL; ++k) lev[k] = P.lv.dElist[k] + vz;
    auto klass_of = [&](int d, int s) -> int {
        const int ad = d < 0 ? -d : d;
        int a = 0;
#pragma unroll
        for (int k = 0; k < SL; ++k) a = (k < L && lev[k] == ad) ? k : a;
        const int up = (d > 0 || (d == 0 && s == 1)) ? 1 : 0;
        return a + L * up;
    };
    // 2 sigma_j sum_k J_jk sigma_k (RRG.jl:236-244) for the current configuration (d0) and with the bit of `flip` inverted (d1: the
    // configuration after the move; flip < 0: none)
    auto dE_pair = [&](int j, int flip, int& d0, int& d1) {
        int y[KM], c[KM];
#pragma unroll
        for (int q = 0; q < KM; ++q) { const int idx = q < K ? j * K + q : 0; y[q] = (int)l_A[idx]; c[q] = q < K ? (int)l_J[idx] : 0; }
        const int sj0 = bit_of(j), sj1 = sj0 ^ (int)(j == flip);
        int sy[KM];
#pragma unroll
        for (int q = 0; q < KM; ++q) sy[q] = bit_of(y[q]);
        int a0 = 0, a1 = 0;
#pragma unroll
        for (int q = 0; q < KM; ++q) {
            const int s1 = sy[q] ^ (int)(y[q] == flip);
            a0 += (sj0 == sy[q]) ? c[q] : -c[q];
            a1 += (sj1 == s1) ? c[q] : -c[q];
        }
        d0 = 2 * a0; d1 = 2 * a1;
    };

    // the replica's cache in HBM, in the layout of rrr_sparse_kernel (class bytes, positions inside the set, the 2L member arrays): a
    // resumed call (rrrmc_set_resume) reads it back, every call leaves it there (rrrmc_rrr_cache; the next resumed call)
    uint8_t* g_cls = P.cls + (size_t)r * P.N;
    uint16_t* g_spos = static_cast<uint16_t*>(P.spos) + (size_t)r * P.N;
    uint16_t* g_sv = static_cast<uint16_t*>(P.sv) + (size_t)r * 2 * P.L * P.N;
    double* const sf = P.S.sf + (size_t)r * kSmpF;
    long long* const si = P.S.si + (size_t)r * kSmpI;
    const bool resume = P.S.resume != 0;

    // ---- energy(X, C) and gen_DEcache in site order (RRRMC.jl:177-178, DeltaE.jl:74-103) --------------------------------------------
    // pass 1: classes (parked in l_spos) and the energy; pass 2, block of 64 sites by block: stable partition into the segments
    int esum = 0;
    int tcount = 0;                                  // lane c: |set c|
    if (resume) tcount = lane < 2 * P.L ? (int)si[SI_T0 + lane] : 0;
    for (int i0 = 0; !resume && i0 < P.N; i0 += kRrrThreads) {
        const int i = i0 + lane;
        const bool in = i < P.N;
        int k = 0;
        if (in) {
            int d, dx;
            dE_pair(i, -1, d, dx);
            esum -= d / 2;
            k = klass_of(d, bit_of(i));
            l_spos[i] = (uint16_t)k;
        }
#pragma unroll
        for (int c = 0; c < C2; ++c) {
            const int n = (int)__popcll(__ballot(in && k == c));
            tcount += lane == c ? n : 0;
        }
    }
    int E = (resume ? P.E_cur[r] : qw_wave_sum(esum) / 2) + vz;
    int bv = 0, ev = 0;
    auto B_ = [&](int q) -> int { return __builtin_amdgcn_readlane(bv, q); };
    auto respace = [&](int tv) {                     // tv: lane c = size of set c; equal gaps behind the 2L segments
        const int gap = (X.cap - P.N) / (2 * P.L);
        int b = 0;
#pragma unroll
        for (int c = 0; c < C2; ++c) {
            const int tc = __builtin_amdgcn_readlane(tv, c);
            if (lane > c) b += tc + gap;
        }
        bv = lane < 2 * P.L ? b : 0;
        ev = lane < 2 * P.L ? b + tv : 0;
    };
    respace(tcount);
    __syncthreads();
    if (resume) {
        // the sets as the previous call left them: member order is part of the chain's state (rand_move picks v[rand(1:t)], ArraySets.jl:83)
        for (int i = lane; i < P.N; i += kRrrThreads) {
            const int k = (int)g_cls[i];
            int b = 0;
#pragma unroll
            for (int c = 0; c < C2; ++c) { const int bc = B_(c); b = k == c ? bc : b; }      // (v_readlane takes a wave-uniform lane: select per class)
            l_spos[i] = (uint16_t)((int)g_spos[i] + b);
        }
        for (int c = 0; c < 2 * P.L; ++c) {
            const int tc = min(__builtin_amdgcn_readlane(tcount, c), P.N), bc = B_(c);
            for (int i = lane; i < tc; i += kRrrThreads) l_sv[bc + i] = g_sv[(size_t)c * P.N + i];
        }
    } else {
        int fill = bv;                               // lane c: next free slot of segment c
        for (int i0 = 0; i0 < P.N; i0 += kRrrThreads) {
            const int i = i0 + lane;
            const bool in = i < P.N;
            const int k = in ? (int)l_spos[i] : -1;
            int slot = 0;
#pragma unroll
            for (int c = 0; c < C2; ++c) {
                const unsigned long long m = __ballot(in && k == c);
                const int base = __builtin_amdgcn_readlane(fill, c);
                if (k == c) slot = base + (int)__popcll(m & ((1ull << lane) - 1ull));
                fill += lane == c ? (int)__popcll(m) : 0;
            }
            if (in) { l_sv[slot] = (uint16_t)i; l_spos[i] = (uint16_t)slot; }
        }
    }
    __syncthreads();
    // T[k] = t[k] f(k); z = sum over the classes in order
    double Tv = lane < 2 * P.L ? (resume ? sf[SF_T0 + lane] : (double)tcount * l_f[lane < C2 ? lane : 0]) : 0.0;
    double vzd = __longlong_as_double(((long long)vz << 32) | (uint32_t)vz);      // +0.0, opaque
    auto bcast = [&](double x, int c) -> double {
        const unsigned long long u = (unsigned long long)__double_as_longlong(x);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, c), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), c);
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)) + vzd;
    };
    double z = 0.0 + vzd;
#pragma unroll
    for (int c = 0; c < C2; ++c) if (c < 2 * P.L) z += bcast(Tv, c);
    if (resume) z = sf[SF_Z] + vzd;                  // the running sum of the run (DeltaE.jl:258-282), not the sum of T

    const uint32_t rep = P.replica0 + (uint32_t)r;
    const double lambda = P.lambda + vzd, one_m_lambda = (1 - P.lambda) + vzd, staged_thr = P.staged_thr + vzd;
    double acc_rate = (resume ? sf[SF_ACC] : 0.5) + vzd;
    long long accepted = 0, staged_its = 0, ns = 0, next_sample = P.S.samp0;

    auto mulhi_u64_u32 = [](unsigned long long u, uint32_t t) -> uint32_t {
        const unsigned long long lo = (unsigned long long)(uint32_t)u * t;
        const unsigned long long hi = (unsigned long long)(uint32_t)(u >> 32) * t + (lo >> 32);
        return (uint32_t)(hi >> 32);
    };

    // bklMC (mode 1; RRRMC.jl:311-359): every move is applied; `it` advances by the geometric skip rand_skip (DeltaE.jl:141-144) + 1, the
    // moves are numbered m = 1, 2, ... (RRR stream sub 0 for the class / member, sub 2 for the skip), 64 of them prepared at a time
    // A resumed bklMC call continues the run's loop with `iters` more iterations allowed: `it`, `nextstep` and the number of moves made
    // carry on (the move that was pending at the cut is drawn again: same counter, same cache, same draw).
    const bool bkl = P.mode == 1;
    long long it_bkl = 0, nextstep = P.step, m_done = 0, limit = P.iters;
    if (bkl && resume) { it_bkl = si[SI_IT]; nextstep = si[SI_NEXT]; m_done = si[SI_M]; limit += si[SI_LIMIT]; }
    const long long m_first = m_done;
    bool done = bkl && resume && nextstep > limit && nextstep > P.step;          // (a resumed call whose allowance does not reach the run's next sample point makes no move: the reference's loop ends with its last sample, RRRMC.jl:340-343)
    const double Nd = (double)P.N + vzd;
    for (long long base_it = bkl ? m_first : 0; !done && (bkl || base_it < P.iters); base_it += kRrrThreads) {
        __syncthreads();
        {
            const uint64_t gl = P.g0 + (uint64_t)(base_it + 1 + (long long)lane);
            const Philox4 a = philox4x32_10((uint32_t)gl, (uint32_t)(gl >> 32), rep, TAG_RRR, P.k0, P.k1);
            const Philox4 b = philox4x32_10((uint32_t)gl, (uint32_t)(gl >> 32), rep, TAG_RRR | ((bkl ? 2u : 1u) << 8), P.k0, P.k1);
            l_rng[lane * 3 + 0] = (double)((((uint64_t)a.w[0] << 32) | a.w[1]) >> 11) * 0x1.0p-53;
            l_rng[lane * 3 + 1] = __longlong_as_double((long long)(((uint64_t)a.w[2] << 32) | a.w[3]));
            l_rng[lane * 3 + 2] = (double)((((uint64_t)b.w[0] << 32) | b.w[1]) >> 11) * 0x1.0p-53;
        }
        {   // re-space the segments when a gap has run low (through the lanes' registers and LDS: the sets are at most N entries)
            const int nxt = __shfl_down(bv, 1);
            const int gapv = (lane + 1 < 2 * P.L ? nxt : X.cap) - ev;       // lane c: free entries behind segment c
            const bool low = lane < 2 * P.L && gapv < sw_min_gap(P.K);
            if (uni(low)) {
                __syncthreads();
                // compact to the front (segment by segment, ascending: never overwrites unread data), then spread from the back
                const int tv = ev - bv;
                int dst = 0;
                for (int c = 0; c < 2 * P.L; ++c) {
                    const int tc = __builtin_amdgcn_readlane(tv, c), bc = B_(c);
                    for (int i0 = 0; i0 < tc; i0 += kRrrThreads) {
                        const int i = i0 + lane;
                        const uint16_t x = i < tc ? l_sv[bc + i] : (uint16_t)0;
                        __syncthreads();
                        if (i < tc) l_sv[dst + i] = x;
                        __syncthreads();
                    }
                    dst += tc;
                }
                respace(tv);
                int src = P.N;                       // end of the compacted data
                for (int c = 2 * P.L - 1; c >= 0; --c) {
                    const int tc = __builtin_amdgcn_readlane(tv, c), bc = B_(c);
                    src -= tc;
                    // move [src, src + tc) to [bc, bc + tc), bc >= src: from the back in blocks
                    for (int i1 = tc; i1 > 0; i1 -= kRrrThreads) {
                        const int i = i1 - 1 - lane;
                        const uint16_t x = i >= 0 ? l_sv[src + i] : (uint16_t)0;
                        __syncthreads();
                        if (i >= 0) { l_sv[bc + i] = x; l_spos[x] = (uint16_t)(bc + i); }
                        __syncthreads();
                    }
                }
            }
        }
        __syncthreads();
        const int n_it = bkl ? kRrrThreads : (int)(base_it + kRrrThreads < P.iters ? kRrrThreads : P.iters - base_it);
        double u_cls_n = l_rng[0], u_mem_n = l_rng[1], u_acc_n = l_rng[2];
        auto sample_li = [&]() -> int { const long long d = next_sample - (base_it + 1); return d >= 0 && d < (long long)n_it ? (int)d : -1; };
        int li_s = sample_li();
        for (int li = 0; li < n_it; ++li) {
            if (bkl) {
                if (it_bkl >= limit) { done = true; break; }
            } else if (li == li_s) {
                next_sample += P.step;
                if (lane == 0) P.Es[(size_t)ns * P.Rpad + r] = (int32_t)E;
                ns += 1;
                li_s = sample_li();
            }
            const double u_cls = u_cls_n, u1 = u_acc_n;
            const unsigned long long u_mem = (unsigned long long)__double_as_longlong(u_mem_n);
            {
                const int nx = (li + 1 < n_it ? li + 1 : li) * 3;
                u_cls_n = l_rng[nx]; u_mem_n = l_rng[nx + 1]; u_acc_n = l_rng[nx + 2];
            }
            // rand_move (DeltaE.jl:146-167): the first class k with rr < T[0] + .. + T[k], else the last class of non-zero weight
            const double rr = u_cls * z;
            int k = -1;
            {
                double cT = 0.0 + vzd;
#pragma unroll
                for (int c = 0; c < C2; ++c) {
                    if (c < 2 * P.L) { cT += bcast(Tv, c); k = (k < 0 && rr < cT) ? c : k; }
                }
                if (uni(k < 0)) {
                    k = K2 - 1;
#pragma unroll
                    for (int c = C2 - 1; c > 0; --c) if (c < 2 * P.L) k = (k == c && bcast(Tv, c) == 0) ? c - 1 : k;
                }
            }
            const int sk = __builtin_amdgcn_readfirstlane(k);
            const int dE = l_dE[k];
            const int tvv = ev - bv;
            const int move = (int)l_sv[(__builtin_amdgcn_readlane(bv, sk) + vz) + (int)mulhi_u64_u32(u_mem, (uint32_t)(__builtin_amdgcn_readlane(tvv, sk) + vz))];

            long long skip = 0;
            if (bkl) {
                const double skipf = floor(__ddiv_rn(det_log1p(-u1), det_log1p(-__ddiv_rn(z, Nd))));      // rand_skip, DeltaE.jl:141-144
                skip = skipf >= 9.0e18 ? (long long)9.0e18 : (long long)skipf;
                bool out = false;
                while (it_bkl + skip + 1 >= nextstep) {           // the samples the skipped iterations pass (RRRMC.jl:332-350)
                    if (nextstep > limit) { out = true; break; }        // (fewer iterations allowed than `step`: the sample point lies beyond this call — no sample, no move)
                    if (lane == 0) P.Es[(size_t)ns * P.Rpad + r] = (int32_t)E;
                    ns += 1;
                    nextstep += P.step;
                    if (nextstep > limit) { out = true; break; }
                }
                if (out) { done = true; break; }
            }
            // ---- lane-parallel: lane q < K = neighbour q of the move, lane K = the moved spin ----
            const bool isn = lane < K, isme = lane == K;
            const int rowi = move * K + (isn ? lane : 0);
            const int aq = (int)l_A[rowi], aqm = (int)l_A[rowi > 0 ? rowi - 1 : 0], jq = (int)l_J[rowi];
            // neighbors(X, move)[q]: repeats removed (uA, EA.jl:158); zero couplings dropped for a general-level GraphRRG (RRG.jl:133)
            const bool valid = isme || (isn && !(lane > 0 && aq == aqm) && !(P.lv.skip_zero && jq == 0));
            const int myj = isme ? move : (isn ? aq : 0);
            int d0, d1;
            dE_pair(myj, move, d0, d1);
            const int s0 = bit_of(myj);
            const int my_k0 = klass_of(d0, s0), my_k1 = klass_of(d1, s0 ^ (int)(myj == move));
            int my_pos = (int)l_spos[myj];
            const unsigned long long chm = __ballot(valid && my_k0 != my_k1);         // the sites that change class, bit = lane
            // weights and their differences, lane-parallel
            const double my_f0 = l_f[my_k0], my_f1 = l_f[my_k1];

            // one apply_move! in the reference's order (DeltaE.jl:232-295) over the sites that change class: the running sums (WITH_T), the set
            // moves (WITH_SETS), or both in one pass.  The per-site values are short-lived scalars (v_readlane results used at once).
            auto bcast_s = [&](double x, int c) -> double {
                const unsigned long long u = (unsigned long long)__double_as_longlong(x);
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, c), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), c);
                return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
            };
            auto apply = [&](double& Tx, double& zx, bool forward, bool with_T, bool with_sets) {
                for (int q = 0; q <= P.K; ++q) {
                    if (!((chm >> q) & 1ull)) continue;                                 // wave-uniform
                    const int k0 = __builtin_amdgcn_readlane(forward ? my_k0 : my_k1, q), k1 = __builtin_amdgcn_readlane(forward ? my_k1 : my_k0, q);
                    if (with_T) {
                        const double f0 = bcast_s(forward ? my_f0 : my_f1, q), f1 = bcast_s(forward ? my_f1 : my_f0, q);
                        Tx += lane == k0 ? -f0 : (lane == k1 ? f1 : 0.0);
                        zx += f1 - f0;
                    }
                    if (with_sets) {
                        const int j = __builtin_amdgcn_readlane(myj, q) + vz;
                        const int p = __builtin_amdgcn_readlane(my_pos, q) + vz;
                        // ArraySet delete!(S = k0, j) + push!(D = k1, j) (ArraySets.jl:56-76) on the segmented array
                        const int eS = __builtin_amdgcn_readlane(ev, k0) + vz;
                        const int last = (int)l_sv[eS - 1];
                        const int eD = __builtin_amdgcn_readlane(ev, k1) + vz;
                        l_sv[p] = (uint16_t)last;               // all lanes store the same values to the same addresses
                        l_spos[last] = (uint16_t)p;
                        l_sv[eD] = (uint16_t)j;
                        l_spos[j] = (uint16_t)eD;
                        ev = lane == k0 ? eS - 1 : (lane == k1 ? eD + 1 : ev);
                        // a site still to come that sat at the end of the set has been moved into the freed slot
                        my_pos = (lane > q && myj == last) ? p : my_pos;
                    }
                }
            };
            auto flip_move = [&]() { l_sp[move >> 5] ^= 1u << (move & 31); };

            bool acc = false;
            if (bkl) {
                flip_move();
                double zp = z;
                apply(Tv, zp, true, true, true);
                z = zp;
                it_bkl += skip + 1;
                m_done += 1;
                E += dE; accepted += 1;
            } else if (uni(acc_rate < staged_thr)) {
                // staged branch (RRRMC.jl:131-138): T' and z' from copies, the sets only on acceptance
                staged_its += 1;
                double Tp = Tv, zp = z;
                apply(Tp, zp, true, true, false);
                if (uni(z >= zp || u1 < z / zp)) {          // u1 < 1 <= z / z': the quotient is only formed when it can matter
                    flip_move();
                    apply(Tp, zp, true, false, true);
                    Tv = Tp; z = zp;
                    E += dE; accepted += 1; acc = true;
                }
            } else {
                // direct branch: apply_move!, undone by a second apply_move! on rejection
                flip_move();
                double zp = z;
                apply(Tv, zp, true, true, true);
                const bool ok = uni(z >= zp || u1 < z / zp);
                z = zp;
                if (ok) { E += dE; accepted += 1; acc = true; }
                else {
                    flip_move();
                    double zq = z;
                    my_pos = (int)l_spos[myj];
                    apply(Tv, zq, false, true, true);
                    z = zq;
                }
            }
            acc_rate = acc_rate * one_m_lambda + (acc ? 1.0 : 0.0) * lambda;
        }
    }
    __syncthreads();
    for (int i = lane; i < P.W; i += kRrrThreads) g_sp[i] = l_sp[i];
    // the cache, in rrr_sparse_kernel's layout: class = the segment that holds the spin's slot, position relative to the segment
    {
        const int tv = ev - bv;
        for (int i = lane; i < P.N; i += kRrrThreads) {
            const int slot = (int)l_spos[i];
            int k = 0, b = 0;
#pragma unroll
            for (int c = 0; c < C2; ++c) {
                const int bc = B_(c);
                if (c < 2 * P.L && slot >= bc) { k = c; b = bc; }
            }
            g_cls[i] = (uint8_t)k;
            g_spos[i] = (uint16_t)(slot - b);
        }
        for (int c = 0; c < 2 * P.L; ++c) {
            const int tc = min(__builtin_amdgcn_readlane(tv, c), P.N), bc = B_(c);        // (|set| <= N: a bound on the stores whatever happened)
            for (int i = lane; i < tc; i += kRrrThreads) g_sv[(size_t)c * P.N + i] = l_sv[bc + i];
        }
        if (lane < 2 * P.L) { si[SI_T0 + lane] = tv; sf[SF_T0 + lane] = Tv; }
    }
    if (lane == 0) {
        P.E_cur[r] = (int32_t)E;
        P.acc_cur[r] = accepted;
        P.stats[(size_t)r * 3] = accepted; P.stats[(size_t)r * 3 + 1] = bkl ? accepted : staged_its; P.stats[(size_t)r * 3 + 2] = bkl ? it_bkl : P.iters;
        sf[SF_Z] = z; sf[SF_ACC] = acc_rate;
        if (bkl) { si[SI_IT] = it_bkl; si[SI_NEXT] = nextstep; si[SI_M] = m_done; si[SI_LIMIT] = limit; }
    }
}

}  // namespace rrrmc
