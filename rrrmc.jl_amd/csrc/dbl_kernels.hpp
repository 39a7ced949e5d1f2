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
#include <type_traits>

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
    // apply_move! / compute_staged! as a staged gather (the scheme of SparseChain::apply_move, rrr_kernels.hpp; profiles/r06/f8_floor.md):
    // what the flip and the K + 1 set moves read is requested stage by stage — the moved spin's row (A, level couplings, residual
    // couplings), then per neighbour its residual field, class, position, word of spins and row, then the words of the neighbours'
    // neighbours and the ends of the lists that may lose a site — BEFORE the spin is flipped in memory (the moved spin's new value is
    // patched into the gathered words), and the reference's sequence (update_cache_residual!, then the set moves in neighbour order) runs
    // on the gathered values: same values, same stores in the same order.  A staged move that is rejected writes nothing at all (the
    // reference flips X0 and flips it back, RRRMC.jl:131-138).  K <= KM <= 6.
    template <int KM>
    struct Gathered {
        int y[KM];
        double rj[KM], lfm, lfv[KM], un[KM + 1];
        bool first[KM], lastr[KM], live[KM + 1];
        int sx, sy[KM];                                     // the moved spin AFTER its flip; the row's neighbour spins
        int sj[KM + 1], s0[KM + 1], s1[KM + 1], sp_[KM + 1], sl[KM + 1];
    };
    template <int KM>
    __device__ __forceinline__ void gather(int move, Gathered<KM>& G) const
    {
        const int K = cfg.K, L = cfg.L;
        int dj[KM];
#pragma unroll
        for (int q = 0; q < KM; ++q) { const size_t e = (size_t)move * K + (q < K ? q : 0); G.y[q] = cfg.A[e]; dj[q] = (int)cfg.dJ[e]; G.rj[q] = cfg.rJ[e]; }
        G.lfm = lf[move];
        G.sx = 1 - sbit(move);
        bool val[KM];
#pragma unroll
        for (int q = 0; q < KM; ++q) {
            G.first[q] = q < K && !(q > 0 && G.y[q] == G.y[q - 1]);
            G.lastr[q] = q < K && (q == K - 1 || (q + 1 < KM && G.y[q + 1] != G.y[q]));
            val[q] = q < K && (cfg.ea_form ? G.first[q] : dj[q] != 0);                                  // is_nb
        }
        uint32_t wy[KM];
        int yy[KM][KM], cc[KM][KM];
#pragma unroll
        for (int q = 0; q < KM; ++q) {
            const int j = q < K ? G.y[q] : move;
            G.lfv[q] = lf[j]; wy[q] = sp[j >> 5]; G.un[q] = undo[q < K ? q : 0];
            G.sj[q] = j; G.s0[q] = cls[j]; G.sp_[q] = (int)spos[j];
#pragma unroll
            for (int k = 0; k < KM; ++k) { const size_t e = (size_t)j * K + (k < K ? k : 0); yy[q][k] = cfg.A[e]; cc[q][k] = k < K ? (int)cfg.dJ[e] : 0; }
        }
        G.un[KM] = undo[K];
        G.s0[KM] = cls[move]; G.sp_[KM] = (int)spos[move]; G.sj[KM] = move;
        uint32_t wnb[KM][KM];
#pragma unroll
        for (int q = 0; q < KM; ++q)
#pragma unroll
            for (int k = 0; k < KM; ++k) wnb[q][k] = sp[yy[q][k] >> 5];
#pragma unroll
        for (int q = 0; q <= KM; ++q) { const int tq = tg(G.s0[q]); G.sl[q] = (int)sv[(size_t)G.s0[q] * cfg.N + (tq > 0 ? tq - 1 : 0)]; }
        // the neighbours' new classes (klass: DeltaE.jl:28-60, 80-86) from the gathered words, the moved spin taken as flipped
#pragma unroll
        for (int q = 0; q < KM; ++q) {
            const int sjb = (int)((wy[q] >> (G.sj[q] & 31)) & 1u);
            G.sy[q] = sjb;
            int acc = 0;
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                int b = (int)((wnb[q][k] >> (yy[q][k] & 31)) & 1u);
                b = yy[q][k] == move ? G.sx : b;
                acc += (sjb == b) ? cc[q][k] : -cc[q][k];
            }
            const int d = 2 * acc, ad = d < 0 ? -d : d;
            int a = 0;
#pragma unroll
            for (int k = 0; k < SLM; ++k) a = (k < L && cfg.dElist[k] == ad) ? k : a;
            G.s1[q] = a + L * ((d > 0 || (d == 0 && sjb == 1)) ? 1 : 0);
            G.live[q] = val[q] && G.s0[q] != G.s1[q];
        }
        G.live[KM] = true; G.s1[KM] = G.s0[KM] >= L ? G.s0[KM] - L : G.s0[KM] + L;
    }
    // the weights after the move: T' and z' (the running sums of apply_move! / compute_staged!, in slot order)
    template <int KM>
    __device__ __forceinline__ double weights(const Gathered<KM>& G, double* Tp) const
    {
        double zp = z;
#pragma unroll
        for (int q = 0; q <= KM; ++q)
            if (G.live[q]) {
                const double f0 = f(G.s0[q]), f1 = f(G.s1[q]);
#pragma unroll
                for (int a = 0; a < 2 * SLM; ++a) Tp[a] = G.s0[q] == a ? Tp[a] - f0 : Tp[a];
#pragma unroll
                for (int a = 0; a < 2 * SLM; ++a) Tp[a] = G.s1[q] == a ? Tp[a] + f1 : Tp[a];
                zp += f1 - f0;
            }
        return zp;
    }
    // spinflip!(X::DoubleGraph) + the set moves, on the gathered values
    template <int KM>
    __device__ __forceinline__ void commit(int move, Gathered<KM>& G)
    {
        const int K = cfg.K;
        sflip(move);
        // update_cache_residual! (RRG.jl:430-466, EA.jl:456-496)
        if (mlast == move) {
#pragma unroll
            for (int k = 0; k < KM; ++k)
                if (G.first[k]) { lf[G.y[k]] = G.un[k]; undo[k] = G.lfv[k]; }
            lf[move] = -G.lfm;
            undo[K] = -G.un[KM];
        } else {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < KM; ++k)
                if (k < K) {
                    if (G.first[k]) { v = G.lfv[k]; undo[k] = v; }
                    const double c4 = (G.sx ^ G.sy[k]) ? -4.0 : 4.0;           // 4 * sigma_xy with the new s_x
                    v = v - c4 * G.rj[k];
                    if (G.lastr[k]) lf[G.y[k]] = v;
                }
            undo[K] = G.lfm;
            lf[move] = -G.lfm;
            mlast = move;
        }
        unsigned touched = 0u;
#pragma unroll
        for (int q = 0; q <= KM; ++q) {
            if (!G.live[q]) continue;
            const int j = G.sj[q], k0 = G.s0[q], k1 = G.s1[q], p = G.sp_[q];
            IDX* v0 = sv + (size_t)k0 * cfg.N;
            IDX* v1 = sv + (size_t)k1 * cfg.N;
            const int last = ((touched >> k0) & 1u) ? (int)v0[tg(k0) - 1] : G.sl[q];    // ArraySet delete!(k0, j) + push!(k1, j), ArraySets.jl:56-76
            v0[p] = (IDX)last; spos[last] = (IDX)p; tadd(k0, -1);
            const int t1 = tg(k1);
            v1[t1] = (IDX)j; spos[j] = (IDX)t1; tadd(k1, 1);
            cls[j] = (uint8_t)k1;
            touched |= (1u << k0) | (1u << k1);
#pragma unroll
            for (int q2 = 0; q2 <= KM; ++q2)
                if (q2 > q && G.live[q2] && G.sj[q2] == last) G.sp_[q2] = p;
        }
    }
    // apply_move!(X::DoubleGraph, C, move, cache): DeltaE.jl:232-295; returns c = z / z'
    template <int KM>
    __device__ __forceinline__ double apply_move_g(int move)
    {
        Gathered<KM> G;
        gather<KM>(move, G);
        const double zp = weights<KM>(G, T);
        commit<KM>(move, G);
        const double cz = z / zp;
        z = zp;
        return cz;
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
    // apply_move! with its gather sized for the graph's degree (kernel-uniform); always_inline: out of line the chain would live in scratch
    auto apply = [&](int move) __attribute__((always_inline)) -> double {
        if (K <= 3) return c.template apply_move_g<3>(move);
        if (K <= 4) return c.template apply_move_g<4>(move);
        if (K <= 6) return c.template apply_move_g<6>(move);
        return c.apply_move(move);
    };
    long long accepted = 0, staged_its = 0, ns = 0;
    // the staged step on the gathered values: nothing is written unless the move is accepted
    auto staged_km = [&](auto km, int move, uint64_t g, int dE0, double& E, long long& accepted) __attribute__((always_inline)) -> bool {
        constexpr int KM = decltype(km)::value;
        typename DblChain<SLM, IDX>::template Gathered<KM> G;
        c.template gather<KM>(move, G);
        double Tp[2 * SLM];
#pragma unroll
        for (int q = 0; q < 2 * SLM; ++q) Tp[q] = c.T[q];
        const double zp = c.template weights<KM>(G, Tp);
        const double cc = c.z / zp;
        const double dE1 = -G.lfm;                                            // delta_energy_residual
        if (!dbl_accept(cc, -P.beta * dE1, g, rep, P.k0, P.k1)) return false;
        c.template commit<KM>(move, G);                                       // spinflip! + apply_staged!
#pragma unroll
        for (int q = 0; q < 2 * SLM; ++q) c.T[q] = Tp[q];
        c.z = zp;
        E += P.to_f64(dE0) + dE1;
        accepted += 1;
        return true;
    };
    auto staged = [&](int move, uint64_t g, int dE0, double& E, long long& accepted) __attribute__((always_inline)) -> bool {
        if (K <= 3) return staged_km(std::integral_constant<int, 3>{}, move, g, dE0, E, accepted);
        if (K <= 4) return staged_km(std::integral_constant<int, 4>{}, move, g, dE0, E, accepted);
        return staged_km(std::integral_constant<int, 6>{}, move, g, dE0, E, accepted);
    };
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
            if (K <= 6) acc = staged(move, g, dE0, E, accepted);
            else {
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
            }
        } else {
            const double dE1 = -c.lf[move];
            const double cc = apply(move);
            if (dbl_accept(cc, -P.beta * dE1, g, rep, P.k0, P.k1)) { E += P.to_f64(dE0) + dE1; accepted += 1; acc = true; }
            else apply(move);
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
