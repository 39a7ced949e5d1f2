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
//     T[4] (weight exp(-beta fourK)) and z are the reference's running Float64 sums, updated in its order.
//   * the four ArraySet member arrays (8 bytes per spin in HBM): the sets partition the spins, so they live in ONE LDS array of
//     N + slack entries as four segments with gaps between them — push! appends into the gap behind its segment, delete! swaps with
//     the segment's last entry (ArraySets.jl:56-76): the order inside every set is the reference's.  A full gap (rare: the sizes
//     fluctuate by a few hundred) re-spaces the segments through the replica's HBM arrays.
// Float64 sums in the reference's order, det_exp shared with the oracle: bit-identical trajectories, energies, accepted / staged
// counts and final cache.
#pragma once
#include "rrr_kernels.hpp"

namespace rrrmc {

constexpr int kQwMinGap = 256;          // smallest slack (entries per gap) the host accepts for this build

struct QwLayout { size_t off_spos, off_sv, off_A, off_J, off_rng, off_exp, bytes; int cap; };

// LDS layout for (N, W, Nk, K): returns cap = N + slack (slack as large as the LDS allows), cap < N + 4 * kQwMinGap means "does not fit"
inline QwLayout qw_layout(int64_t N, int64_t W, int64_t Nk, int64_t K, size_t lds_limit)
{
    QwLayout L{};
    size_t o = (size_t)W * 4;
    L.off_spos = o; o += (((size_t)N * 2 + 7) & ~(size_t)7);
    const size_t fixed_tail = (((size_t)Nk * K * 2 + 7) & ~(size_t)7) + (((size_t)Nk * K + 7) & ~(size_t)7) + 64 * 3 * 8 + 16 * 8 + 64;
    const size_t room = lds_limit > o + fixed_tail ? lds_limit - o - fixed_tail : 0;
    int64_t cap = (int64_t)(room / 2) & ~(int64_t)3;
    if (cap > 2 * N) cap = 2 * N;
    L.cap = (int)cap;
    L.off_sv = o; o += (size_t)cap * 2;
    o = (o + 7) & ~(size_t)7;
    L.off_A = o; o += (((size_t)Nk * K * 2 + 7) & ~(size_t)7);
    L.off_J = o; o += (((size_t)Nk * K + 7) & ~(size_t)7);
    L.off_rng = o; o += 64 * 3 * 8;
    L.off_exp = o; o += 16 * 8;
    L.bytes = o;
    return L;
}

struct QwExtra { int cap; uint32_t off_spos, off_sv, off_A, off_J, off_rng, off_exp; };

__device__ __forceinline__ int qw_uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ double qw_unid(double x)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// class a + 2 up of a spin from its bit and its two Trotter neighbours' bits (QT.jl:86-103, DeltaE.jl:80-86)
__device__ __forceinline__ int qw_class(int sk, int s1, int s2)
{
    const int d = (sk == s1) - (sk != s2);
    const int a = d != 0 ? 1 : 0;
    const int up = d > 0 || (d == 0 && sk == 1);
    return a + kQL * up;
}

__global__ __launch_bounds__(kRrrThreads) void rrr_quant_wave_kernel(RrrParams P, QwExtra X)
{
    extern __shared__ uint32_t qw_lds[];
    unsigned char* lds8 = reinterpret_cast<unsigned char*>(qw_lds);
    const int lane = (int)threadIdx.x, r = (int)blockIdx.x;
    const int N = P.N, Nk = P.Nk, K = P.K, W = P.W, CAP = X.cap;
    uint32_t* l_sp = qw_lds;                                                      // [W]
    uint16_t* l_spos = reinterpret_cast<uint16_t*>(lds8 + X.off_spos);            // [N]
    uint16_t* l_sv = reinterpret_cast<uint16_t*>(lds8 + X.off_sv);                // [CAP] four segments
    uint16_t* l_A = reinterpret_cast<uint16_t*>(lds8 + X.off_A);                  // [Nk][K]
    int8_t* l_J = reinterpret_cast<int8_t*>(lds8 + X.off_J);                      // [Nk][K]
    double* l_rng = reinterpret_cast<double*>(lds8 + X.off_rng);                  // [64][3]: class uniform, member u64 (as bits), accept uniform
    double* l_exp = reinterpret_cast<double*>(lds8 + X.off_exp);                  // [2K+1]: det_exp(-beta * (2 a / M)), a = -K..K

    uint32_t* g_sp = P.spins + (size_t)r * W;
    uint8_t* g_cls = P.cls + (size_t)r * N;
    uint16_t* g_sv = P.sv + (size_t)r * 4 * N;
    uint16_t* g_spos = P.spos + (size_t)r * N;
    int32_t* g_t = P.st + (size_t)r * 4;

    // ---- stage the replica ------------------------------------------------------------------------------------------------
    int t[4], base[4];
    for (int k = 0; k < 4; ++k) t[k] = qw_uni(g_t[k]);
    auto respace = [&]() {      // equal gaps behind the four segments
        const int gap = (CAP - N) / 4;
        base[0] = 0;
        for (int k = 1; k < 4; ++k) base[k] = base[k - 1] + t[k - 1] + gap;
    };
    respace();
    for (int i = lane; i < W; i += kRrrThreads) l_sp[i] = g_sp[i];
    for (int i = lane; i < N; i += kRrrThreads) l_spos[i] = g_spos[i];
    for (int k = 0; k < 4; ++k)
        for (int i = lane; i < t[k]; i += kRrrThreads) l_sv[base[k] + i] = g_sv[(size_t)k * N + i];
    for (int i = lane; i < Nk * K; i += kRrrThreads) { l_A[i] = (uint16_t)P.A[i]; l_J[i] = P.J[i]; }
    if (lane <= 2 * K) {
        const double dE1 = (double)(2 * (lane - K)) / (double)P.M;               // slice_res(slice_delta) for sum_k +-J = lane - K
        l_exp[lane] = det_exp(-P.beta * dE1);
    }
    __syncthreads();

    const uint32_t rep = P.replica0 + (uint32_t)r;
    const uint32_t nk_magic = (uint32_t)((0x100000000ull + (uint32_t)Nk - 1u) / (uint32_t)Nk);
    const double ft1 = P.ft1, fourK = P.fourK;
    // T[0..2] == (double)t[0..2] exactly (weights 1.0); T3 and z are running sums (DeltaE.jl:90-103, 258-282)
    double T3 = qw_unid(P.T[(size_t)r * 4 + 3]), z = qw_unid(P.zz[r]), E = qw_unid(P.E_cur[r]), acc_rate = qw_unid(P.acc_rate[r]);
    long long accepted = P.stats[(size_t)r * 2], staged_its = P.stats[(size_t)r * 2 + 1];
    accepted = ((long long)qw_uni((int)(accepted >> 32)) << 32) | (uint32_t)qw_uni((int)accepted);
    staged_its = ((long long)qw_uni((int)(staged_its >> 32)) << 32) | (uint32_t)qw_uni((int)staged_its);
    long long ns = 0, next_sample = P.step;

    auto bit_of = [&](int x) -> int { return (int)((l_sp[x >> 5] >> (x & 31)) & 1u); };

    // ArraySet delete!(k0, j) + push!(k1, j) on the segmented array; p = spos[j] (uniform).  Returns the element that took j's place
    // (the old last of k0; j itself when j was the last) so that the caller can patch cached positions.
    auto set_move = [&](int j, int k0, int k1, int p) -> int {
        int b0 = 0, t0 = 0, b1 = 0, t1 = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { if (q == k0) { b0 = base[q]; t0 = t[q]; } if (q == k1) { b1 = base[q]; t1 = t[q]; } }
        const int last = qw_uni((int)l_sv[b0 + t0 - 1]);
        int lim = CAP;
#pragma unroll
        for (int q = 0; q < 3; ++q) if (q == k1) lim = base[q + 1];
        if (b1 + t1 == lim) {
            // the gap behind segment k1 is used up: write the four segments to the replica's HBM arrays, re-space, read them back
            __syncthreads();
            for (int k = 0; k < 4; ++k)
                for (int i = lane; i < t[k]; i += kRrrThreads) g_sv[(size_t)k * N + i] = l_sv[base[k] + i];
            __syncthreads();
            respace();
            for (int k = 0; k < 4; ++k)
                for (int i = lane; i < t[k]; i += kRrrThreads) l_sv[base[k] + i] = g_sv[(size_t)k * N + i];
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) { if (q == k0) b0 = base[q]; if (q == k1) b1 = base[q]; }
        }
        if (lane == 0) {
            l_sv[b0 + p] = (uint16_t)last;
            l_spos[last] = (uint16_t)p;
            l_sv[b1 + t1] = (uint16_t)j;
            l_spos[j] = (uint16_t)t1;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { if (q == k0) t[q] -= 1; if (q == k1) t[q] += 1; }
        return last;
    };

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
        __syncthreads();
        const long long it_end = base_it + kRrrThreads < P.iters ? base_it + kRrrThreads : P.iters;
        for (long long it = base_it + 1; it <= it_end; ++it) {
            if (it == next_sample) {
                next_sample += P.step;
                if (lane == 0) P.Es[ns * P.R + r] = E;
                ns += 1;
            }
            const int ri = (int)(it - base_it - 1) * 3;
            const double u_cls = qw_unid(l_rng[ri]);
            const unsigned long long u_mem = (unsigned long long)__double_as_longlong(qw_unid(l_rng[ri + 1]));
            // rand_move (DeltaE.jl:146-167): class proportional to T (linear scan), member uniform
            const double rr = u_cls * z;
            const double Tq[4] = {(double)t[0], (double)t[1], (double)t[2], T3};
            int k = 0;
            double cT = 0.0;
            for (k = 0; k < 4; ++k) {
                cT += Tq[k];
                if (rr < cT) break;
            }
            if (k == 4) k = 3;
            if (!(rr < cT)) {
                while ((k == 3 ? T3 : (double)(k == 0 ? t[0] : k == 1 ? t[1] : t[2])) == 0) k -= 1;
            }
            const double dE0 = k == 0 ? -0.0 : k == 1 ? -fourK : k == 2 ? 0.0 : fourK;
            int bk = 0, tk = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) if (q == k) { bk = base[q]; tk = t[q]; }
            const int move = qw_uni((int)l_sv[bk + (int)mulhi64(u_mem, (uint64_t)tk)]);

            // ---- lane-parallel part: classes of (nb0, nb1, move) before / after the flip, their set positions, the slice bonds ----
            int nb0 = move - Nk; if (nb0 < 0) nb0 += N;
            int nb1 = move + Nk; if (nb1 >= N) nb1 -= N;
            const int ks = (int)__umulhi((uint32_t)move, nk_magic), is = move - ks * Nk, off = ks * Nk;
            int my_k0 = 0, my_k1 = 0, my_pos = 0, my_sat = 0;
            if (lane < 3) {
                const int j = lane == 0 ? nb0 : lane == 1 ? nb1 : move;
                int j1 = j - Nk; if (j1 < 0) j1 += N;
                int j2 = j + Nk; if (j2 >= N) j2 -= N;
                const int sj = bit_of(j), s1 = bit_of(j1), s2 = bit_of(j2);
                my_k0 = qw_class(sj, s1, s2);
                my_k1 = qw_class(sj ^ (j == move), s1 ^ (j1 == move), s2 ^ (j2 == move));
                my_pos = (int)l_spos[j];
            } else if (lane < 3 + K) {
                const int q = lane - 3;
                const int y = (int)l_A[is * K + q];
                const int sy = bit_of(off + y), si = bit_of(move);
                my_sat = ((si == sy) == (l_J[is * K + q] > 0)) ? 1 : 0;          // +J_iq sigma_i sigma_y > 0
            }
            const unsigned long long satm = __ballot(my_sat != 0);
            const int npos = __popcll(satm);
            const int asum = 2 * npos - K;                                        // sum_q J_iq sigma_i sigma_y
            const double dE1 = (double)(2 * asum) / (double)P.M;                 // delta_energy_residual, QT.jl:270-281
            const int k0a = __builtin_amdgcn_readlane(my_k0, 0), k1a = __builtin_amdgcn_readlane(my_k1, 0);
            const int k0b = __builtin_amdgcn_readlane(my_k0, 1), k1b = __builtin_amdgcn_readlane(my_k1, 1);
            const int k0m = __builtin_amdgcn_readlane(my_k0, 2), k1m = __builtin_amdgcn_readlane(my_k1, 2);
            int pa = __builtin_amdgcn_readlane(my_pos, 0), pb = __builtin_amdgcn_readlane(my_pos, 1), pm = __builtin_amdgcn_readlane(my_pos, 2);
            // k1m == k0m +- 2 always (the moved spin changes direction, DeltaE.jl:275-276)

            // accept(c, x) (RRRMC.jl:40-44) with x = -beta dE1
            auto accept = [&](double c) -> bool {
                const double x = -P.beta * dE1;
                bool ok = (c >= 1 && x >= 0);
                if (!ok) {
                    const double a = c * qw_unid(l_exp[asum + K]);
                    ok = a >= 1;
                    if (!ok) ok = qw_unid(l_rng[ri + 2]) < a;
                }
                return ok;
            };
            auto fcls = [&](int kk) -> double { return kk == 3 ? ft1 : 1.0; };
            auto flip_move = [&]() { if (lane == 0) l_sp[move >> 5] ^= 1u << (move & 31); };

            bool acc = false;
            if (acc_rate < P.staged_thr) {
                // staged branch: step_rrr (RRRMC.jl:131-138) = compute_staged! + compute_reverse_probabilities!, apply_staged! on acceptance
                staged_its += 1;
                double T3p = T3, zp = z;
                if (k0a != k1a) { const double f0 = fcls(k0a), f1 = fcls(k1a); if (k0a == 3) T3p -= f0; if (k1a == 3) T3p += f1; zp += f1 - f0; }
                if (k0b != k1b) { const double f0 = fcls(k0b), f1 = fcls(k1b); if (k0b == 3) T3p -= f0; if (k1b == 3) T3p += f1; zp += f1 - f0; }
                { const double f0 = fcls(k0m), f1 = fcls(k1m); if (k0m == 3) T3p -= f0; if (k1m == 3) T3p += f1; zp += f1 - f0; }
                const double c = z / zp;
                if (accept(c)) {
                    flip_move();
                    if (k0a != k1a) { const int l = set_move(nb0, k0a, k1a, pa); if (l == nb1) pb = pa; if (l == move) pm = pa; }
                    if (k0b != k1b) { const int l = set_move(nb1, k0b, k1b, pb); if (l == move) pm = pb; }
                    set_move(move, k0m, k1m, pm);
                    T3 = T3p; z = zp;
                    E += dE0 + dE1;
                    accepted += 1;
                    acc = true;
                }
            } else {
                // direct branch: apply_move! (DeltaE.jl:232-295), undone by a second apply_move! on rejection
                flip_move();
                double zp = z;
                const bool cha = k0a != k1a, chb = k0b != k1b;
                if (cha) {
                    const double f0 = fcls(k0a), f1 = fcls(k1a); if (k0a == 3) T3 -= f0; if (k1a == 3) T3 += f1; zp += f1 - f0;
                    const int l = set_move(nb0, k0a, k1a, pa); if (l == nb1) pb = pa; if (l == move) pm = pa;
                }
                if (chb) {
                    const double f0 = fcls(k0b), f1 = fcls(k1b); if (k0b == 3) T3 -= f0; if (k1b == 3) T3 += f1; zp += f1 - f0;
                    const int l = set_move(nb1, k0b, k1b, pb); if (l == move) pm = pb;
                }
                {
                    const double f0 = fcls(k0m), f1 = fcls(k1m); if (k0m == 3) T3 -= f0; if (k1m == 3) T3 += f1; zp += f1 - f0;
                    set_move(move, k0m, k1m, pm);
                }
                const double c = z / zp;
                z = zp;
                if (accept(c)) {
                    E += dE0 + dE1;
                    accepted += 1;
                    acc = true;
                } else {
                    // the undo: apply_move! again, from the flipped configuration (classes k1 -> k0, positions re-read)
                    flip_move();
                    double zq = z;
                    int qa = 0, qb = 0, qm = 0;
                    if (lane < 3) my_pos = (int)l_spos[lane == 0 ? nb0 : lane == 1 ? nb1 : move];
                    qa = __builtin_amdgcn_readlane(my_pos, 0); qb = __builtin_amdgcn_readlane(my_pos, 1); qm = __builtin_amdgcn_readlane(my_pos, 2);
                    if (cha) {
                        const double f0 = fcls(k1a), f1 = fcls(k0a); if (k1a == 3) T3 -= f0; if (k0a == 3) T3 += f1; zq += f1 - f0;
                        const int l = set_move(nb0, k1a, k0a, qa); if (l == nb1) qb = qa; if (l == move) qm = qa;
                    }
                    if (chb) {
                        const double f0 = fcls(k1b), f1 = fcls(k0b); if (k1b == 3) T3 -= f0; if (k0b == 3) T3 += f1; zq += f1 - f0;
                        const int l = set_move(nb1, k1b, k0b, qb); if (l == move) qm = qb;
                    }
                    {
                        const double f0 = fcls(k1m), f1 = fcls(k0m); if (k1m == 3) T3 -= f0; if (k0m == 3) T3 += f1; zq += f1 - f0;
                        set_move(move, k1m, k0m, qm);
                    }
                    z = zq;
                }
            }
            acc_rate = acc_rate * (1 - P.lambda) + (acc ? 1.0 : 0.0) * P.lambda;             // RRRMC.jl:281
        }
    }

    // ---- write the replica back: spins, sets, positions, sizes, the classes (recomputed: the cache is consistent), scalars --------
    __syncthreads();
    for (int i = lane; i < W; i += kRrrThreads) g_sp[i] = l_sp[i];
    for (int i = lane; i < N; i += kRrrThreads) {
        g_spos[i] = l_spos[i];
        int j1 = i - Nk; if (j1 < 0) j1 += N;
        int j2 = i + Nk; if (j2 >= N) j2 -= N;
        g_cls[i] = (uint8_t)qw_class(bit_of(i), bit_of(j1), bit_of(j2));
    }
    for (int k = 0; k < 4; ++k)
        for (int i = lane; i < t[k]; i += kRrrThreads) g_sv[(size_t)k * N + i] = l_sv[base[k] + i];
    if (lane == 0) {
        for (int k = 0; k < 4; ++k) g_t[k] = t[k];
        P.T[(size_t)r * 4 + 0] = (double)t[0]; P.T[(size_t)r * 4 + 1] = (double)t[1]; P.T[(size_t)r * 4 + 2] = (double)t[2]; P.T[(size_t)r * 4 + 3] = T3;
        P.zz[r] = z; P.E_cur[r] = E; P.acc_rate[r] = acc_rate;
        P.stats[(size_t)r * 2] = accepted; P.stats[(size_t)r * 2 + 1] = staged_its;
    }
}

}  // namespace rrrmc
