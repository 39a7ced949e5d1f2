// gfx950 kernels for rrrMC(X::DoubleGraph) (src/RRRMC.jl:221-290) on GraphRRGNormalDiscretized / GraphEANormalDiscretized
// (src/graphs/RRG.jl:285-500, src/graphs/EA.jl:311-532): Gaussian couplings split by `discretize` (src/Common.jl:38-72) into
//   dJ  integer levels -> the inner DiscrGraph X0 = GraphRRG{Int,LEV,K} / GraphEA{Int,LEV,2D} that drives DeltaECache{Int,L}
//   rJ  Float64 residuals -> the residual local-field cache that gives delta_energy_residual (RRG.jl:468-476)
// SURVEY.md §8f rank 3.  As for the other reduced-rejection samplers the move is state dependent, so one THREAD per replica and
// replica-contiguous state in HBM/L2.  X0's integer fields are recomputed from the spins (2 sigma_i sum_k dJ_ik sigma_k is exact),
// the residual fields are cached per replica together with the live part of lfields_last (the K+1 values the undo path of
// update_cache_residual! can read, RRG.jl:437-447) — the rejected direct moves of rrrMC are undone through exactly that path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "philox.hpp"
#include "rrr_kernels.hpp"
#include "sk_kernels.hpp"   // det_exp

namespace rrrmc {

constexpr int kDLmax = 8;          // levels of allΔE(X0)
constexpr int kDKmax = 8;          // neighbours per site

struct RrrDblParams {
    const int32_t* A;        // [N][K]
    const int8_t* dJ;        // [N][K]
    const double* rJ;        // [N][K]
    uint32_t* spins;         // [R][W]   BitVector word order
    uint8_t* cls;            // [R][N]
    uint16_t* sv;            // [R][2L][N]
    uint16_t* spos;          // [R][N]
    double* lf;              // [R][N]   residual local fields
    double* undo;            // [R][K+1] saved neighbour fields / own field of the last residual update
    double* E_cur;           // [R]
    int64_t* stats;          // [R][2]   accepted, staged iterations
    double* Es;              // [nsamples][R]
    double ft[kDLmax];
    int dElist[kDLmax];
    double beta, staged_thr, lambda;
    // level units -> Float64: value = (units * lev_mul) / lev_div; (1, 1.0) for Int levels, (g, 1e5) for DFloat64 levels
    // (src/DFloats.jl:11-36: the Int64 t = round(x * 10^5), Float64(t) = t / 10^5; units = t / g, g = gcd of the levels' t)
    long long lev_mul;
    double lev_div;
    uint64_t g0;
    int64_t iters, step;
    uint32_t k0, k1, replica0;
    int N, K, L, W, R, ea_form, energy_only;
    __host__ __device__ __forceinline__ double to_f64(long long units) const { return (double)(units * lev_mul) / lev_div; }
};

struct DblChain {
    const RrrDblParams* P;
    uint32_t* sp; uint8_t* cls; uint16_t* sv; uint16_t* spos;
    double* lf; double* undo;
    int t[2 * kDLmax];
    double T[2 * kDLmax], z;
    int mlast;

    __device__ __forceinline__ int sbit(int x) const { return (int)((sp[x >> 5] >> (x & 31)) & 1u); }
    __device__ __forceinline__ void sflip(int x) { sp[x >> 5] ^= 1u << (x & 31); }
    // delta_energy(X0, C, i) (RRG.jl:236-244 / EA.jl:266-275) recomputed from the spins
    __device__ __forceinline__ int dE0(int i) const
    {
        const int si = sbit(i);
        int acc = 0;
        for (int q = 0; q < P->K; ++q) {
            const int sy = sbit(P->A[(size_t)i * P->K + q]);
            const int j = (int)P->dJ[(size_t)i * P->K + q];
            acc += (si == sy) ? j : -j;
        }
        return 2 * acc;
    }
    __device__ __forceinline__ int klass(int i) const      // findk + L * up: DeltaE.jl:28-60, 80-86
    {
        const int d = dE0(i), ad = d < 0 ? -d : d;
        int a = 0;
        for (int k = 0; k < P->L; ++k) if (P->dElist[k] == ad) a = k;
        const int up = d > 0 || (d == 0 && sbit(i) == 1);
        return a + P->L * up;
    }
    __device__ __forceinline__ double f(int k) const { return k >= P->L ? P->ft[k - P->L] : 1.0; }
    // neighbors(X0, i): GraphRRG keeps the non-zero couplings (RRG.jl:133), GraphEA removes repeats (EA.jl:158)
    __device__ __forceinline__ bool is_nb(int move, int q) const
    {
        if (P->ea_form) return !(q > 0 && P->A[(size_t)move * P->K + q] == P->A[(size_t)move * P->K + q - 1]);
        return P->dJ[(size_t)move * P->K + q] != 0;
    }
    __device__ __forceinline__ void set_move(int j, int k0, int k1)
    {
        uint16_t* v0 = sv + (size_t)k0 * P->N;
        uint16_t* v1 = sv + (size_t)k1 * P->N;
        const int p = spos[j], last = v0[t[k0] - 1];
        v0[p] = (uint16_t)last; spos[last] = (uint16_t)p; t[k0] -= 1;
        v1[t[k1]] = (uint16_t)j; spos[j] = (uint16_t)t[k1]; t[k1] += 1;
        cls[j] = (uint8_t)k1;
    }
    // update_cache_residual! (RRG.jl:430-466, EA.jl:456-496), called after the flip of `move`
    __device__ void res_update(int move)
    {
        const int K = P->K;
        const int32_t* Ax = P->A + (size_t)move * K;
        const double* Jx = P->rJ + (size_t)move * K;
        const double lfm = lf[move];
        if (mlast == move) {
            for (int k = 0; k < K; ++k) {
                if (k > 0 && Ax[k] == Ax[k - 1]) continue;
                const int y = Ax[k];
                const double tmp = lf[y];
                lf[y] = undo[k];
                undo[k] = tmp;
            }
            lf[move] = -lfm;
            undo[K] = -undo[K];
            return;
        }
        const int sx = sbit(move);
        double v = 0.0;
        for (int k = 0; k < K; ++k) {
            const int y = Ax[k];
            if (!(k > 0 && Ax[k] == Ax[k - 1])) {
                v = lf[y];
                undo[k] = v;
            }
            const double c = (sx ^ sbit(y)) ? -4.0 : 4.0;           // 4 * sigma_xy with the new s_x
            v = v - c * Jx[k];
            if (k == K - 1 || Ax[k + 1] != y) lf[y] = v;
        }
        undo[K] = lfm;
        lf[move] = -lfm;
        mlast = move;
    }
    __device__ __forceinline__ void spinflip(int move) { sflip(move); res_update(move); }     // spinflip!(X::DoubleGraph, C, move)
    // apply_move!(X::DoubleGraph, C, move, cache): DeltaE.jl:232-295; returns c = z / z'
    __device__ double apply_move(int move)
    {
        spinflip(move);
        double zp = z;
        const int32_t* Ax = P->A + (size_t)move * P->K;
        for (int q = 0; q <= P->K; ++q) {
            if (q < P->K && !is_nb(move, q)) continue;
            const int j = q < P->K ? Ax[q] : move;
            const int k0 = cls[j];
            const int k1 = q < P->K ? klass(j) : (k0 >= P->L ? k0 - P->L : k0 + P->L);
            if (q < P->K && k0 == k1) continue;
            const double f0 = f(k0), f1 = f(k1);
            T[k0] -= f0; T[k1] += f1; zp += f1 - f0;
            set_move(j, k0, k1);
        }
        const double cc = z / zp;
        z = zp;
        return cc;
    }
};

// accept(c, x): RRRMC.jl:40-44
__device__ __forceinline__ bool dbl_accept(double c, double x, uint64_t g, uint32_t rep, uint32_t k0, uint32_t k1)
{
    if (c >= 1 && x >= 0) return true;
    const double a = c * det_exp(x);
    if (a >= 1) return true;
    const Philox4 o2 = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), rep, TAG_RRR | (1u << 8), k0, k1);
    return (double)((((uint64_t)o2.w[0] << 32) | o2.w[1]) >> 11) * 0x1.0p-53 < a;
}

// energy(X, C) = E0 + E1 (RRG.jl:326-360) — fills the residual fields — and, with `classes`, gen_ΔEcache(X0, C, beta)'s class
// sets in site order (DeltaE.jl:74-103)
__device__ __forceinline__ double dbl_init_chain(DblChain& c, const RrrDblParams& P, int r, bool classes)
{
    const int N = P.N, K2 = 2 * P.L, K = P.K;
    c.P = &P;
    c.sp = P.spins + (size_t)r * P.W; c.cls = P.cls + (size_t)r * N; c.sv = P.sv + (size_t)r * K2 * N; c.spos = P.spos + (size_t)r * N;
    c.lf = P.lf + (size_t)r * N; c.undo = P.undo + (size_t)r * (K + 1);
    c.mlast = -1;
    long long n0 = 0;
    double E1 = 0.0;
    for (int k = 0; k < K2; ++k) c.t[k] = 0;
    for (int i = 0; i < N; ++i) {
        n0 -= c.dE0(i) / 2;                     // lf_x = -sum dJ sx sy
        const int sx = 2 * c.sbit(i) - 1;
        double fl = 0.0;
        for (int q = 0; q < K; ++q) {
            const int sy = 2 * c.sbit(P.A[(size_t)i * K + q]) - 1;
            fl = fl - P.rJ[(size_t)i * K + q] * (double)sx * (double)sy;
        }
        E1 = E1 + fl;
        c.lf[i] = 2.0 * fl;
        if (classes) {
            const int k = c.klass(i);
            c.cls[i] = (uint8_t)k;
            c.sv[(size_t)k * N + c.t[k]] = (uint16_t)i;
            c.spos[i] = (uint16_t)c.t[k];
            c.t[k] += 1;
        }
    }
    E1 = E1 / 2;
    return P.to_f64(n0 / 2) + E1;
}

// standardMC (RRRMC.jl:81-127) on the DoubleGraph: delta_energy = convert(Float64, dE0 + dE1) (RRG.jl:493-497), common site
// (SITE stream), per-replica ACCEPT_F64 uniform; one thread per replica as for rrrMC.
__global__ __launch_bounds__(kRrrThreads) void dbl_standard_kernel(RrrDblParams P)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= P.R) return;
    DblChain c;
    double E = dbl_init_chain(c, P, r, false);
    const uint32_t rep = P.replica0 + (uint32_t)r;
    long long accepted = 0, ns = 0;
    long long next_sample = P.step;          // iterations k * step: a counter instead of a 64-bit modulo per iteration
    for (long long it = 1; it <= P.iters; ++it) {
        if (it == next_sample) { next_sample += P.step; P.Es[(size_t)ns * P.R + r] = E; ns += 1; }
        const uint64_t g = P.g0 + (uint64_t)it;
        const int move = (int)site_of(P.k0, P.k1, g, (uint32_t)P.N);
        const double dE = P.to_f64(c.dE0(move)) + (-c.lf[move]);
        const double x = -P.beta * dE;
        const bool acc = (x >= 0.0) || (rand53(P.k0, P.k1, g, rep) < det_exp(x));        // RRRMC.jl:39
        if (acc) { c.spinflip(move); E += dE; accepted += 1; }
    }
    P.E_cur[r] = E;
    P.stats[(size_t)r * 2] = accepted; P.stats[(size_t)r * 2 + 1] = 0;
}

__global__ __launch_bounds__(kRrrThreads) void rrr_dbl_kernel(RrrDblParams P)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= P.R) return;
    const int N = P.N, L = P.L, K2 = 2 * P.L, K = P.K;
    DblChain c;
    double E = dbl_init_chain(c, P, r, !P.energy_only);
    if (P.energy_only) { P.E_cur[r] = E; return; }
    c.z = 0.0;
    for (int k = 0; k < 2 * kDLmax; ++k) c.T[k] = 0.0;
    for (int k = 0; k < K2; ++k) { const double x = (double)c.t[k] * c.f(k); c.z += x; c.T[k] = x; }

    const uint32_t rep = P.replica0 + (uint32_t)r;
    long long accepted = 0, staged_its = 0, ns = 0;
    double acc_rate = 0.5;
    long long next_sample = P.step;          // iterations k * step: a counter instead of a 64-bit modulo per iteration
    for (long long it = 1; it <= P.iters; ++it) {
        if (it == next_sample) { next_sample += P.step; P.Es[(size_t)ns * P.R + r] = E; ns += 1; }
        const uint64_t g = P.g0 + (uint64_t)it;
        // rand_move: DeltaE.jl:146-167
        const Philox4 o = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), rep, TAG_RRR, P.k0, P.k1);
        const double rr = (double)((((uint64_t)o.w[0] << 32) | o.w[1]) >> 11) * 0x1.0p-53 * c.z;
        int k = 0;
        double cT = 0.0;
        for (k = 0; k < K2; ++k) { cT += c.T[k]; if (rr < cT) break; }
        if (k == K2) k = K2 - 1;
        if (!(rr < cT)) while (c.T[k] == 0) k -= 1;
        const int dE0 = k < L ? -P.dElist[k] : P.dElist[k - L];
        const int move = c.sv[(size_t)k * N + (int)mulhi64(((uint64_t)o.w[2] << 32) | o.w[3], (uint64_t)c.t[k])];
        bool acc = false;
        if (acc_rate < P.staged_thr) {
            // step_rrr(X0, C, cache) (RRRMC.jl:131-138): compute_staged! flips X0 only, and flips it back
            staged_its += 1;
            int sj[kDKmax + 1], s0[kDKmax + 1], s1[kDKmax + 1], nst = 0;
            c.sflip(move);
            const int32_t* Ax = P.A + (size_t)move * K;
            for (int q = 0; q < K; ++q) {
                if (!c.is_nb(move, q)) continue;
                const int j = Ax[q], k0 = c.cls[j], k1 = c.klass(j);
                if (k0 == k1) continue;
                sj[nst] = j; s0[nst] = k0; s1[nst] = k1; ++nst;
            }
            { const int k0 = c.cls[move]; sj[nst] = move; s0[nst] = k0; s1[nst] = k0 >= L ? k0 - L : k0 + L; ++nst; }
            c.sflip(move);
            double Tp[2 * kDLmax], zp = c.z;
            for (int q = 0; q < 2 * kDLmax; ++q) Tp[q] = c.T[q];
            for (int q = 0; q < nst; ++q) { const double f0 = c.f(s0[q]), f1 = c.f(s1[q]); Tp[s0[q]] -= f0; Tp[s1[q]] += f1; zp += f1 - f0; }
            const double cc = c.z / zp;
            const double dE1 = -c.lf[move];                                   // delta_energy_residual
            if (dbl_accept(cc, -P.beta * dE1, g, rep, P.k0, P.k1)) {
                c.spinflip(move);
                for (int q = 0; q < nst; ++q) c.set_move(sj[q], s0[q], s1[q]);   // apply_staged!
                for (int q = 0; q < 2 * kDLmax; ++q) c.T[q] = Tp[q];
                c.z = zp;
                E += P.to_f64(dE0) + dE1;
                accepted += 1; acc = true;
            }
        } else {
            const double dE1 = -c.lf[move];
            const double cc = c.apply_move(move);
            if (dbl_accept(cc, -P.beta * dE1, g, rep, P.k0, P.k1)) { E += P.to_f64(dE0) + dE1; accepted += 1; acc = true; }
            else c.apply_move(move);
        }
        acc_rate = acc_rate * (1 - P.lambda) + (acc ? 1.0 : 0.0) * P.lambda;      // RRRMC.jl:281
    }
    P.E_cur[r] = E;
    P.stats[(size_t)r * 2] = accepted; P.stats[(size_t)r * 2 + 1] = staged_its;
}

}  // namespace rrrmc
