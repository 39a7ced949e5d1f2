// gfx950 kernels for the reduced-rejection-rate sampler rrrMC(X::DoubleGraph) (src/RRRMC.jl:221-290) on
// GraphQuant (src/graphs/QT.jl:126-321) = M Suzuki-Trotter slices of one GraphRRG{Int,(-1,1),K} disorder coupled
// along the Trotter axis by GraphQT{fourK} (QT.jl:42-122), with the move-selection cache DeltaECache{Float64,2}
// (src/DeltaE.jl:63-295) over ArraySets (src/ArraySets.jl:19-85).
//
// The site choice is state dependent (rand_move), so replicas cannot share a site stream: one THREAD per replica,
// every structure of the reference kept per replica in HBM/L2 (replica-contiguous).  The Float64 running sums
// (T, z, z', E, acc_rate) are updated in the reference's order, the exp is the deterministic one shared with the
// oracle: trajectories are bit-identical to the CPU restatement.  The path is gather/latency bound.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "philox.hpp"
#include "sk_kernels.hpp"   // det_exp

namespace rrrmc {

constexpr uint32_t TAG_RRR = 8;
constexpr int kRrrThreads = 64;
constexpr int kQL = 2;          // levels of allΔE(GraphQT) = (0.0, fourK), QT.jl:111

struct RrrParams {
    // disorder of the slice graph (shared by slices and replicas)
    const int32_t* A;        // [Nk][K]
    const int8_t* J;         // [Nk][K]
    // per-replica state, replica-contiguous
    uint32_t* spins;         // [R][W]       bit x of replica r: word x >> 5, bit x & 31; x = slice * Nk + i
    uint8_t* cls;            // [R][N]       class of every spin: a + 2 * up  (DeltaECache.pos)
    uint16_t* sv;            // [R][4][N]    ArraySet.v of the four classes
    uint16_t* spos;          // [R][N]       position of the spin inside its set
    int32_t* st;             // [R][4]       set sizes
    double* T;               // [R][4]
    double* zz;              // [R]          z
    double* E_cur;           // [R]
    double* acc_rate;        // [R]
    int64_t* stats;          // [R][2]       accepted, staged iterations (this call)
    double* Es;              // [nsamples][R]
    double beta, fourK, ft1, staged_thr, lambda;
    uint64_t g0;
    int64_t iters, step;
    uint32_t k0, k1, replica0;
    int Nk, M, K, N, W, R;
};

struct RrrView {             // one replica's slices of the arrays above
    uint32_t* sp; uint8_t* cls; uint16_t* sv; uint16_t* spos; int32_t* t;
    int N, Nk, M, K;
    const int32_t* A; const int8_t* J;
    double fourK;
};

__device__ __forceinline__ int sbit(const uint32_t* sp, int x) { return (int)((sp[x >> 5] >> (x & 31)) & 1u); }
__device__ __forceinline__ void sflip(uint32_t* sp, int x) { sp[x >> 5] ^= 1u << (x & 31); }

// GraphQT: neighbours along the Trotter axis (QT.jl:105-108) and delta_energy (QT.jl:86-103)
__device__ __forceinline__ void qt_nb(const RrrView& v, int i, int& j1, int& j2)
{
    j1 = i - v.Nk + (i < v.Nk ? v.N : 0);
    j2 = i + v.Nk - (i + v.Nk >= v.N ? v.N : 0);
}
__device__ __forceinline__ int qt_delta(const RrrView& v, int i)      // in units of fourK: -1, 0, +1
{
    int j1, j2;
    qt_nb(v, i, j1, j2);
    const int sk = sbit(v.sp, i), s1 = sbit(v.sp, j1), s2 = sbit(v.sp, j2);
    return (sk == s1) - (sk != s2);
}
// class of spin i for the current configuration: a = index of |dE| in (0, fourK), up = dE > 0 || (dE == 0 && s == 1)
__device__ __forceinline__ int qt_class(const RrrView& v, int i)      // DeltaE.jl:80-86
{
    const int d = qt_delta(v, i);
    const int a = d != 0 ? 1 : 0;
    const int up = d > 0 || (d == 0 && sbit(v.sp, i) == 1);
    return a + kQL * up;
}
// delta_energy of the slice graph (RRG.jl:236-244) recomputed from the slice's spins: 2 sigma_i sum_k J_ik sigma_k
__device__ __forceinline__ int slice_delta(const RrrView& v, int move)
{
    const int k = move / v.Nk, i = move - k * v.Nk, off = k * v.Nk;
    const int si = sbit(v.sp, move);
    int acc = 0;
    for (int q = 0; q < v.K; ++q) {
        const int y = v.A[i * v.K + q];
        const int sy = sbit(v.sp, off + y);
        acc += (si == sy) ? (int)v.J[i * v.K + q] : -(int)v.J[i * v.K + q];
    }
    return 2 * acc;
}
// ArraySet delete! / push! (ArraySets.jl:56-76); one position array serves the four sets (membership is exclusive)
__device__ __forceinline__ void set_move(const RrrView& v, int j, int k0, int k1)
{
    uint16_t* v0 = v.sv + (size_t)k0 * v.N;
    uint16_t* v1 = v.sv + (size_t)k1 * v.N;
    const int p = v.spos[j];
    const int last = v0[v.t[k0] - 1];
    v0[p] = (uint16_t)last;
    v.spos[last] = (uint16_t)p;
    v.t[k0] -= 1;
    v1[v.t[k1]] = (uint16_t)j;
    v.spos[j] = (uint16_t)v.t[k1];
    v.t[k1] += 1;
    v.cls[j] = (uint8_t)k1;
}

__device__ __forceinline__ double class_f(int k, double ft1) { return k == 3 ? ft1 : 1.0; }   // get_class_f, ft = (1, exp(-beta fourK))

__device__ __forceinline__ RrrView rrr_view(const RrrParams& P, int r)
{
    RrrView v;
    v.sp = P.spins + (size_t)r * P.W;
    v.cls = P.cls + (size_t)r * P.N;
    v.sv = P.sv + (size_t)r * 4 * P.N;
    v.spos = P.spos + (size_t)r * P.N;
    v.t = P.st + (size_t)r * 4;
    v.N = P.N; v.Nk = P.Nk; v.M = P.M; v.K = P.K; v.A = P.A; v.J = P.J; v.fourK = P.fourK;
    return v;
}

// energy(X::GraphQuant, C) (QT.jl:185-199) + DeltaECache construction (DeltaE.jl:74-103), one thread per replica
__global__ __launch_bounds__(kRrrThreads) void rrr_init_kernel(RrrParams P)
{
    const int r = blockIdx.x * kRrrThreads + threadIdx.x;
    if (r >= P.R) return;
    const RrrView v = rrr_view(P, r);
    // E = energy0 * fourK / 4 + sum_k energy(X1[k]) / M
    long long n0 = 0;
    for (int i = 0; i < P.Nk; ++i) {
        int sj = sbit(v.sp, i + (P.M - 1) * P.Nk);
        for (int k = 0; k < P.M; ++k) {
            const int sk = sbit(v.sp, i + k * P.Nk);
            n0 -= 1 - 2 * (sk ^ sj);
            sj = sk;
        }
    }
    double E = (double)n0 * P.fourK / 4;
    for (int k = 0; k < P.M; ++k) {
        long long n = 0;                       // RRG.jl:164-189: n = sum_x lf_x / 2, lf_x = -sum_q J sx sy
        for (int i = 0; i < P.Nk; ++i) n -= slice_delta(v, k * P.Nk + i) / 2;     // -(2 sx sum J sy)/2 = lf_x
        n /= 2;
        E += (double)n / (double)P.M;
    }
    P.E_cur[r] = E;
    // cache: classes and sets in site order
    for (int k = 0; k < 4; ++k) v.t[k] = 0;
    for (int i = 0; i < P.N; ++i) {
        const int k = qt_class(v, i);
        v.cls[i] = (uint8_t)k;
        v.sv[(size_t)k * P.N + v.t[k]] = (uint16_t)i;
        v.spos[i] = (uint16_t)v.t[k];
        v.t[k] += 1;
    }
    double z = 0.0;
    for (int k = 0; k < 4; ++k) {
        const double x = (double)v.t[k] * class_f(k, P.ft1);
        z += x;
        P.T[(size_t)r * 4 + k] = x;
    }
    P.zz[r] = z;
    P.acc_rate[r] = 0.5;
    P.stats[(size_t)r * 2] = 0;
    P.stats[(size_t)r * 2 + 1] = 0;
}

__global__ __launch_bounds__(kRrrThreads) void rrr_quant_kernel(RrrParams P)
{
    const int r = blockIdx.x * kRrrThreads + threadIdx.x;
    if (r >= P.R) return;
    const RrrView v = rrr_view(P, r);
    const uint32_t rep = P.replica0 + (uint32_t)r;
    double T[4], z = P.zz[r], E = P.E_cur[r], acc_rate = P.acc_rate[r];
    for (int k = 0; k < 4; ++k) T[k] = P.T[(size_t)r * 4 + k];
    int64_t accepted = P.stats[(size_t)r * 2], staged_its = P.stats[(size_t)r * 2 + 1];
    int64_t ns = 0;
    const double dEl[2] = {0.0, P.fourK};

    for (int64_t it = 1; it <= P.iters; ++it) {
        if (it % P.step == 0) { P.Es[ns * P.R + r] = E; ns += 1; }
        const uint64_t g = P.g0 + (uint64_t)it;
        // rand_move: DeltaE.jl:146-167
        const Philox4 o = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), rep, TAG_RRR, P.k0, P.k1);
        const double rr = (double)((((uint64_t)o.w[0] << 32) | o.w[1]) >> 11) * 0x1.0p-53 * z;
        int k = 0;
        double cT = 0.0;
        for (k = 0; k < 4; ++k) {
            cT += T[k];
            if (rr < cT) break;
        }
        if (k == 4) k = 3;
        if (!(rr < cT)) while (T[k] == 0) k -= 1;
        const double dE0 = k < kQL ? -dEl[k] : dEl[k - kQL];
        const uint64_t u = ((uint64_t)o.w[2] << 32) | o.w[3];
        const int move = v.sv[(size_t)k * P.N + (int)mulhi64(u, (uint64_t)v.t[k])];

        bool acc = false;
        int nb[2];
        qt_nb(v, move, nb[0], nb[1]);
        if (acc_rate < P.staged_thr) {
            // staged branch: step_rrr (RRRMC.jl:131-138) = compute_staged! + compute_reverse_probabilities!
            staged_its += 1;
            int sj[3], s0[3], s1[3], nst = 0;
            sflip(v.sp, move);
            for (int q = 0; q < 2; ++q) {
                const int j = nb[q];
                const int k0 = v.cls[j], k1 = qt_class(v, j);
                if (k0 == k1) continue;
                sj[nst] = j; s0[nst] = k0; s1[nst] = k1; ++nst;
            }
            {
                const int k0 = v.cls[move];
                sj[nst] = move; s0[nst] = k0; s1[nst] = k0 >= kQL ? k0 - kQL : k0 + kQL; ++nst;
            }
            sflip(v.sp, move);
            double Tp[4] = {T[0], T[1], T[2], T[3]}, zp = z;
            for (int q = 0; q < nst; ++q) {
                const double f0 = class_f(s0[q], P.ft1), f1 = class_f(s1[q], P.ft1);
                Tp[s0[q]] -= f0;
                Tp[s1[q]] += f1;
                zp += f1 - f0;
            }
            const double c = z / zp;
            const double dE1 = (double)slice_delta(v, move) / (double)P.M;         // delta_energy_residual, QT.jl:270-281
            const double x = -P.beta * dE1;
            bool ok = (c >= 1 && x >= 0);
            if (!ok) {                                                                 // accept(c, x), RRRMC.jl:40-44
                const double a = c * det_exp(x);
                ok = a >= 1;
                if (!ok) {
                    const Philox4 o2 = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), rep, TAG_RRR | (1u << 8), P.k0, P.k1);
                    ok = (double)((((uint64_t)o2.w[0] << 32) | o2.w[1]) >> 11) * 0x1.0p-53 < a;
                }
            }
            if (ok) {
                sflip(v.sp, move);                                                     // spinflip!(X, C, move)
                for (int q = 0; q < nst; ++q) set_move(v, sj[q], s0[q], s1[q]);        // apply_staged!
                for (int q = 0; q < 4; ++q) T[q] = Tp[q];
                z = zp;
                E += dE0 + dE1;
                accepted += 1;
                acc = true;
            }
        } else {
            // direct branch: apply_move! (DeltaE.jl:232-295), undone by a second apply_move! on rejection
            const double dE1 = (double)slice_delta(v, move) / (double)P.M;
            double c = 0.0;
            for (int pass = 0; pass < 2; ++pass) {
                sflip(v.sp, move);
                double zp = z;
                for (int q = 0; q < 2; ++q) {
                    const int j = nb[q];
                    const int k0 = v.cls[j], k1 = qt_class(v, j);
                    if (k0 == k1) continue;
                    const double f0 = class_f(k0, P.ft1), f1 = class_f(k1, P.ft1);
                    T[k0] -= f0;
                    T[k1] += f1;
                    zp += f1 - f0;
                    set_move(v, j, k0, k1);
                }
                {
                    const int k0 = v.cls[move], k1 = k0 >= kQL ? k0 - kQL : k0 + kQL;
                    const double f0 = class_f(k0, P.ft1), f1 = class_f(k1, P.ft1);
                    T[k0] -= f0;
                    T[k1] += f1;
                    zp += f1 - f0;
                    set_move(v, move, k0, k1);
                }
                const double cc = z / zp;
                z = zp;
                if (pass == 1) break;            // that was the undo
                c = cc;
                const double x = -P.beta * dE1;
                bool ok = (c >= 1 && x >= 0);
                if (!ok) {
                    const double a = c * det_exp(x);
                    ok = a >= 1;
                    if (!ok) {
                        const Philox4 o2 = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), rep, TAG_RRR | (1u << 8), P.k0, P.k1);
                        ok = (double)((((uint64_t)o2.w[0] << 32) | o2.w[1]) >> 11) * 0x1.0p-53 < a;
                    }
                }
                if (ok) { E += dE0 + dE1; accepted += 1; acc = true; break; }
            }
        }
        acc_rate = acc_rate * (1 - P.lambda) + (acc ? 1.0 : 0.0) * P.lambda;             // RRRMC.jl:281
    }
    for (int k = 0; k < 4; ++k) P.T[(size_t)r * 4 + k] = T[k];
    P.zz[r] = z; P.E_cur[r] = E; P.acc_rate[r] = acc_rate;
    P.stats[(size_t)r * 2] = accepted; P.stats[(size_t)r * 2 + 1] = staged_its;
}

// INIT stream for the replica-contiguous layout: word w of replica r holds sites 32w .. 32w+31
__global__ __launch_bounds__(256) void quant_init_spins_kernel(uint32_t* __restrict__ spins, int N, int W, uint32_t replica0, uint32_t k0, uint32_t k1)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= W) return;
    const uint32_t rho = replica0 + blockIdx.y;
    uint32_t word = 0u;
    for (int b = 0; b < 32; ++b) {
        const int x = 32 * w + b;
        if (x < N) word |= ((init_spin_word(k0, k1, rho >> 5, (uint64_t)x) >> (rho & 31u)) & 1u) << b;
    }
    spins[(size_t)blockIdx.y * W + w] = word;
}

}  // namespace rrrmc
