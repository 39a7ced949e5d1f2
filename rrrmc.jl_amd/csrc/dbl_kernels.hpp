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
    void* sv;                // [R][2L][N]  IDX = uint16_t, or uint32_t when N > 65535
    void* spos;              // [R][N]
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
    int32_t* mlast;          // [R] move_last of the residual cache, kept across resumed standardMC calls
    int resume;              // standardMC continues from E_cur / lf / undo / mlast instead of recomputing them (one chain across hook calls)
    SmpState S;              // rrrMC: resumed calls (rrr_kernels.hpp)
    __host__ __device__ __forceinline__ double to_f64(long long units) const { return (double)(units * lev_mul) / lev_div; }
};

// SLM = compile-time bound on the number of levels (4 or 8).  As for SparseChain (rrr_kernels.hpp): the chain keeps COPIES of the few
// parameters it needs (a pointer to the kernel's parameter struct forces the struct into scratch memory) and its per-class arrays
// are indexed only through fully unrolled selects, so that everything stays in registers.
template <int SLM, typename IDX = uint16_t>
struct DblChain {
    struct Cfg { int N, K, L, ea_form; const int32_t* A; const int8_t* dJ; const double* rJ; int dElist[SLM]; double ft[SLM]; };
    Cfg cfg;
    uint32_t* sp; uint8_t* cls; IDX* sv; IDX* spos;
    double* lf; double* undo;
    int t[2 * SLM];
    double T[2 * SLM], z;
    int mlast;
    __device__ __forceinline__ int tg(int k) const { int x = 0;
#pragma unroll
        for (int a = 0; a < 2 * SLM; ++a) x = k == a ? t[a] : x;
        return x; }
    __device__ __forceinline__ void tadd(int k, int d) {
#pragma unroll
        for (int a = 0; a < 2 * SLM; ++a) t[a] = k == a ? t[a] + d : t[a]; }
    __device__ __forceinline__ double Tg(int k) const { double x = 0.0;
#pragma unroll
        for (int a = 0; a < 2 * SLM; ++a) x = k == a ? T[a] : x;
        return x; }
    __device__ __forceinline__ void Tadd(int k, double d) {
#pragma unroll
        for (int a = 0; a < 2 * SLM; ++a) T[a] = k == a ? T[a] + d : T[a]; }
    __device__ __forceinline__ int lev(int a) const { int d = 0;
#pragma unroll
        for (int k = 0; k < SLM; ++k) d = a == k ? cfg.dElist[k] : d;
        return d; }
    // rand_move's class: the first k with rr < T[0] + .. + T[k], else the last class of non-zero weight (DeltaE.jl:146-160)
    __device__ __forceinline__ int pick_class(double rr, int K2) const
    {
        int ksel = -1;
        double cT = 0.0;
#pragma unroll
        for (int a = 0; a < 2 * SLM; ++a)
            if (a < K2 && ksel < 0) { cT += T[a]; if (rr < cT) ksel = a; }
        if (ksel < 0) { ksel = K2 - 1; while (Tg(ksel) == 0) ksel -= 1; }
        return ksel;
    }

    __device__ __forceinline__ int sbit(int x) const { return (int)((sp[x >> 5] >> (x & 31)) & 1u); }
    __device__ __forceinline__ void sflip(int x) { sp[x >> 5] ^= 1u << (x & 31); }
    // delta_energy(X0, C, i) (RRG.jl:236-244 / EA.jl:266-275) recomputed from the spins
    __device__ __forceinline__ int dE0(int i) const
    {
        const int si = sbit(i);
        int acc = 0;
        for (int q = 0; q < cfg.K; ++q) {
            const int sy = sbit(cfg.A[(size_t)i * cfg.K + q]);
            const int j = (int)cfg.dJ[(size_t)i * cfg.K + q];
            acc += (si == sy) ? j : -j;
        }
        return 2 * acc;
    }
    __device__ __forceinline__ int klass(int i) const      // findk + L * up: DeltaE.jl:28-60, 80-86
    {
        const int d = dE0(i), ad = d < 0 ? -d : d;
        int a = 0;
#pragma unroll
        for (int k = 0; k < SLM; ++k) a = (k < cfg.L && cfg.dElist[k] == ad) ? k : a;
        const int up = d > 0 || (d == 0 && sbit(i) == 1);
        return a + cfg.L * up;
    }
    __device__ __forceinline__ double f(int k) const { double x = 1.0;
#pragma unroll
        for (int a = 0; a < SLM; ++a) x = (k - cfg.L == a) ? cfg.ft[a] : x;
        return x; }
    // neighbors(X0, i): GraphRRG keeps the non-zero couplings (RRG.jl:133), GraphEA removes repeats (EA.jl:158)
    __device__ __forceinline__ bool is_nb(int move, int q) const
    {
        if (cfg.ea_form) return !(q > 0 && cfg.A[(size_t)move * cfg.K + q] == cfg.A[(size_t)move * cfg.K + q - 1]);
        return cfg.dJ[(size_t)move * cfg.K + q] != 0;
    }
    __device__ __forceinline__ void set_move(int j, int k0, int k1)
    {
        IDX* v0 = sv + (size_t)k0 * cfg.N;
        IDX* v1 = sv + (size_t)k1 * cfg.N;
        const int p = spos[j], last = v0[tg(k0) - 1];
        v0[p] = (IDX)last; spos[last] = (IDX)p; tadd(k0, -1);
        const int t1 = tg(k1);
        v1[t1] = (IDX)j; spos[j] = (IDX)t1; tadd(k1, 1);
        cls[j] = (uint8_t)k1;
    }
    // update_cache_residual! (RRG.jl:430-466, EA.jl:456-496), called after the flip of `move`
    __device__ void res_update(int move)
    {
        const int K = cfg.K;
        const int32_t* Ax = cfg.A + (size_t)move * K;
        const double* Jx = cfg.rJ + (size_t)move * K;
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
        const int32_t* Ax = cfg.A + (size_t)move * cfg.K;
        for (int q = 0; q <= cfg.K; ++q) {
            if (q < cfg.K && !is_nb(move, q)) continue;
            const int j = q < cfg.K ? Ax[q] : move;
            const int k0 = cls[j];
            const int k1 = q < cfg.K ? klass(j) : (k0 >= cfg.L ? k0 - cfg.L : k0 + cfg.L);
            if (q < cfg.K && k0 == k1) continue;
            const double f0 = f(k0), f1 = f(k1);
            Tadd(k0, -f0); Tadd(k1, f1); zp += f1 - f0;
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
// the chain's view of replica r's arrays (no computation)
template <int SLM, typename IDX>
__device__ __forceinline__ void dbl_bind_chain(DblChain<SLM, IDX>& c, const RrrDblParams& P, int r)
{
    const int N = P.N, K2 = 2 * P.L, K = P.K;
    c.cfg.N = P.N; c.cfg.K = P.K; c.cfg.L = P.L; c.cfg.ea_form = P.ea_form; c.cfg.A = P.A; c.cfg.dJ = P.dJ; c.cfg.rJ = P.rJ;
#pragma unroll
    for (int k = 0; k < SLM; ++k) { c.cfg.dElist[k] = P.dElist[k]; c.cfg.ft[k] = P.ft[k]; }
    c.sp = P.spins + (size_t)r * P.W; c.cls = P.cls + (size_t)r * N; c.sv = static_cast<IDX*>(P.sv) + (size_t)r * K2 * N; c.spos = static_cast<IDX*>(P.spos) + (size_t)r * N;
    c.lf = P.lf + (size_t)r * N; c.undo = P.undo + (size_t)r * (K + 1);
    c.mlast = -1;
}

template <int SLM, typename IDX>
__device__ __forceinline__ double dbl_init_chain(DblChain<SLM, IDX>& c, const RrrDblParams& P, int r, bool classes)
{
    const int N = P.N, K = P.K;
    dbl_bind_chain(c, P, r);
    long long n0 = 0;
    double E1 = 0.0;
#pragma unroll
    for (int k = 0; k < 2 * SLM; ++k) c.t[k] = 0;
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
            const int k = c.klass(i), tk = c.tg(k);
            c.cls[i] = (uint8_t)k;
            c.sv[(size_t)k * N + tk] = (IDX)i;
            c.spos[i] = (IDX)tk;
            c.tadd(k, 1);
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
    DblChain<1> c;           // no class bookkeeping under standardMC
    double E;
    if (P.resume) {
        dbl_bind_chain(c, P, r);
        c.mlast = P.mlast[r];
        E = P.E_cur[r];
    } else {
        E = dbl_init_chain(c, P, r, false);
    }
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
    P.mlast[r] = c.mlast;
    P.stats[(size_t)r * 2] = accepted; P.stats[(size_t)r * 2 + 1] = 0;
}

template <int SLM, typename IDX>
__global__ __launch_bounds__(kRrrThreads) void rrr_dbl_kernel(RrrDblParams P)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= P.R) return;
    const int N = P.N, L = P.L, K2 = 2 * P.L, K = P.K;
    DblChain<SLM, IDX> c;
    double* const sf = P.S.sf ? P.S.sf + (size_t)r * kSmpF : nullptr;
    long long* const sq = P.S.si ? P.S.si + (size_t)r * kSmpI : nullptr;
    const bool resume = P.S.resume != 0 && !P.energy_only;
    double E;
    if (resume) {
        // a resumed call: the class sets (member order), T, z, the residual fields with their undo record and move_last, and the tracked
        // energy are where the previous call left them
        dbl_bind_chain(c, P, r);
        c.mlast = P.mlast[r];
#pragma unroll
        for (int k = 0; k < 2 * SLM; ++k) { c.t[k] = (int)sq[SI_T0 + k]; c.T[k] = sf[SF_T0 + k]; }
        c.z = sf[SF_Z];
        E = sf[SF_E];
    } else {
    E = dbl_init_chain(c, P, r, !P.energy_only);
    if (P.energy_only) { P.E_cur[r] = E; return; }
    c.z = 0.0;
#pragma unroll
    for (int k = 0; k < 2 * SLM; ++k) c.T[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 2 * SLM; ++k)
        if (k < K2) { const double x = (double)c.t[k] * c.f(k); c.z += x; c.T[k] = x; }
    }

    const uint32_t rep = P.replica0 + (uint32_t)r;
    long long accepted = 0, staged_its = 0, ns = 0;
    double acc_rate = resume ? sf[SF_ACC] : 0.5;
    long long next_sample = P.S.samp0;       // iterations k * step of the run: a counter instead of a 64-bit modulo per iteration
    for (long long it = 1; it <= P.iters; ++it) {
        if (it == next_sample) { next_sample += P.step; P.Es[(size_t)ns * P.R + r] = E; ns += 1; }
        const uint64_t g = P.g0 + (uint64_t)it;
        // rand_move: DeltaE.jl:146-167
        const Philox4 o = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), rep, TAG_RRR, P.k0, P.k1);
        const double rr = (double)((((uint64_t)o.w[0] << 32) | o.w[1]) >> 11) * 0x1.0p-53 * c.z;
        const int k = c.pick_class(rr, K2);
        const int dE0 = k < L ? -c.lev(k) : c.lev(k - L);
        const int move = c.sv[(size_t)k * N + (int)mulhi64(((uint64_t)o.w[2] << 32) | o.w[3], (uint64_t)c.tg(k))];
        bool acc = false;
        if (acc_rate < P.staged_thr) {
            // step_rrr(X0, C, cache) (RRRMC.jl:131-138): compute_staged! flips X0 only, and flips it back
            staged_its += 1;
            // staged changes in fixed slots (slot q = neighbour q, slot kDKmax = the moved spin) and fully unrolled loops: registers
            int sj[kDKmax + 1], s0[kDKmax + 1], s1[kDKmax + 1];
            bool live[kDKmax + 1];
            c.sflip(move);
            const int32_t* Ax = P.A + (size_t)move * K;
#pragma unroll
            for (int q = 0; q < kDKmax; ++q) {
                live[q] = false; sj[q] = 0; s0[q] = 0; s1[q] = 0;
                if (q < K && c.is_nb(move, q)) {
                    const int j = Ax[q], k0 = c.cls[j], k1 = c.klass(j);
                    if (k0 != k1) { live[q] = true; sj[q] = j; s0[q] = k0; s1[q] = k1; }
                }
            }
            { const int k0 = c.cls[move]; live[kDKmax] = true; sj[kDKmax] = move; s0[kDKmax] = k0; s1[kDKmax] = k0 >= L ? k0 - L : k0 + L; }
            c.sflip(move);
            double Tp[2 * SLM], zp = c.z;
#pragma unroll
            for (int q = 0; q < 2 * SLM; ++q) Tp[q] = c.T[q];
#pragma unroll
            for (int q = 0; q <= kDKmax; ++q)
                if (live[q]) {
                    const double f0 = c.f(s0[q]), f1 = c.f(s1[q]);
#pragma unroll
                    for (int a = 0; a < 2 * SLM; ++a) Tp[a] = s0[q] == a ? Tp[a] - f0 : Tp[a];
#pragma unroll
                    for (int a = 0; a < 2 * SLM; ++a) Tp[a] = s1[q] == a ? Tp[a] + f1 : Tp[a];
                    zp += f1 - f0;
                }
            const double cc = c.z / zp;
            const double dE1 = -c.lf[move];                                   // delta_energy_residual
            if (dbl_accept(cc, -P.beta * dE1, g, rep, P.k0, P.k1)) {
                c.spinflip(move);
#pragma unroll
                for (int q = 0; q <= kDKmax; ++q)
                    if (live[q]) c.set_move(sj[q], s0[q], s1[q]);                 // apply_staged!
#pragma unroll
                for (int q = 0; q < 2 * SLM; ++q) c.T[q] = Tp[q];
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
    P.mlast[r] = c.mlast;
#pragma unroll
    for (int k = 0; k < 2 * SLM; ++k) { sq[SI_T0 + k] = c.t[k]; sf[SF_T0 + k] = c.T[k]; }
    sf[SF_Z] = c.z; sf[SF_E] = E; sf[SF_ACC] = acc_rate;
}

}  // namespace rrrmc
