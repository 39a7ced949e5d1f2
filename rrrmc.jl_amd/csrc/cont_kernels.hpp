// gfx950 kernel for the continuous-energy samplers on the Float64 sparse models GraphRRGNormal / GraphEANormal — the reference's
// second experiment (scripts/scripts.jl:152-281, test_RRGCont):
//   mode 0  rrrMC(X::SingleGraph)  src/RRRMC.jl:149-219   DeltaECacheCont + DynamicSampler (src/DeltaE.jl:297-410, DynamicSamplers.jl)
//   mode 1  bklMC                  src/RRRMC.jl:311-359   same cache, rand_skip (DeltaE.jl:319-325)
//   mode 2  wtmMC                  src/RRRMC.jl:376-426   THeap of next-flip times (src/WaitingTimes.jl), tau = max(1, exp(beta dE))
//   mode 3  extremal_opt           src/RRRMC.jl:474-521   EOCacheCont (src/DeltaE.jl:557-635): the ranking by dE kept sorted move by move
// One thread per replica, replica-contiguous state in HBM/L2 (the move is state dependent, as for the other reduced-rejection
// kernels).  The graph's local-field cache is the one of spf_kernels.hpp (delta_energy = -lfields, update_cache! with the exact
// undo path through the K+1-entry record), exp / log1p are the fixed-order ones shared with the oracle, every Float64 running sum
// (tree nodes, z, E, acc_rate, t) is updated in the reference's order: bit-identical to the CPU restatement.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "philox.hpp"
#include "rrr_kernels.hpp"   // TAG_RRR, TAG_WTM, prior_of, kRrrThreads
#include "sk_kernels.hpp"    // det_exp, det_log1p

namespace rrrmc {

constexpr int kContKmax = 8;

struct ContParams {
    const int32_t* A;        // [N][K]
    const double* J;         // [N][K]   couplings (GraphRRGNormal / GraphEANormal) or the residuals rJ of a discretised DoubleGraph
    const int8_t* dJ;        // [N][K]   DoubleGraph only: the level part, in level units (null otherwise); its delta_energy, promoted to
    long long lev_mul;       //          Float64 as (units * lev_mul) / lev_div, is added to the residual one (RRG.jl:493-497)
    double lev_div;
    uint32_t* spins;         // [R][W]
    double* lf;              // [R][N]   local fields
    double* undo;            // [R][K+1]
    double* dEs;             // [R][N]   DeltaECacheCont.ΔEs          (wtmMC: heap keys)
    double* v;               // [R][N2]  DynamicSampler.v
    double* ps;              // [R][N2]  DynamicSampler.ps (N2 - 1 used)
    void* hid;               // [R][N]   wtmMC: site at heap position (uint16_t; uint32_t when N > 65535)
    void* hpos;              // [R][N]   wtmMC: heap position of site
    double* E_cur;           // [R]
    int64_t* stats;          // [R][3]
    double* Es;              // [nsamples][R]
    double* t_out;           // [R]
    int32_t* status;         // [R]  1 = DynamicSampler's "unrecoverable loss of precision"
    const int8_t* qJ;        // GraphQuant (bklMC / wtmMC over the whole DoubleGraph, DeltaE.jl:315): slice couplings [qNk][K] (A = the slice's table), else null
    double fourK;            //   delta_energy = qt_delta * fourK + slice_delta / qM (QT.jl:283-286); neighbors = Trotter pair, then the slice's (QT.jl:288-321)
    int qNk, qM;
    // GraphQuant over DENSE slices (every other spin of the slice is a neighbour): qkind 2 = binary GraphSK (GraphQSKT), 3 = GraphSKNormal
    // (GraphQSKNormalT); 1 = sparse +-J slices (qJ above), 4 = sparse Float64 slices (GraphQEAT: qJf and the slice caches below); 0 = not a GraphQuant
    int qkind;
    const double* qJf; double* qflf; double* qfundo; int32_t* qfml;   // sparse Float64 slices: couplings [qNk][K], per-replica slice caches (as RrrParams)
    const uint32_t* qJb; int qWk; double qsN;                       // binary: rows of J as 32-bit words, sqrt(Nk)
    const double* qJd; double* qslf; int32_t* qsmv; uint8_t* qscur;  // GraphSKNormal: couplings, per-replica slice caches (as RrrParams)
    const double* ftau;      // [N]  extremal_opt: cumsum(j^-tau)
    uint32_t* cmin;          // [R][W] extremal_opt: configuration of minimum energy
    double beta, staged_thr, lambda, stepf;
    uint64_t g0;
    int64_t iters, step;
    uint32_t k0, k1, replica0, call;
    int N, K, N2, levs, W, R, Rp, ea_form, mode;      // Rp = row stride of Es
    SmpState S;              // resumed calls (rrr_kernels.hpp)
    int64_t samples_before;  // wtmMC: samples the run had taken before this (resumed) call
    double* ps_top;          // cont_wave_kernel: [R][N2 / 16] the LDS part of the sampler's tree between resumed calls
};

// rankshuffle! (DeltaE.jl:611-634) restated with per-move keys: every run of equal dE is ordered by (Philox key of (g, site), site).
// v[p] = dE of the site ranked p, hid[p] = that site, hpos = inverse, key = scratch of n words.
__device__ inline void eo_order_ties_impl(const double* v, uint16_t* hid, uint16_t* hpos, unsigned long long* key, int n, uint64_t g, uint32_t rep,
                                          uint32_t k0, uint32_t k1)
{
    int i0 = 0;
    while (i0 < n) {
        int i1 = i0 + 1;
        while (i1 < n && v[i1] == v[i0]) ++i1;
        if (i1 - i0 > 1) {
            for (int p = i0; p < i1; ++p) {
                const Philox4 o = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), rep, TAG_RRR | (4u << 8) | ((uint32_t)hid[p] << 16), k0, k1);
                key[p] = ((unsigned long long)o.w[0] << 32) | o.w[1];
            }
            for (int p = i0 + 1; p < i1; ++p) {                  // insertion sort of the run by (key, site)
                const unsigned long long kp = key[p];
                const uint16_t sp_ = hid[p];
                int q = p;
                while (q > i0 && (key[q - 1] > kp || (key[q - 1] == kp && hid[q - 1] > sp_))) { key[q] = key[q - 1]; hid[q] = hid[q - 1]; q -= 1; }
                key[q] = kp; hid[q] = sp_;
            }
            for (int p = i0; p < i1; ++p) hpos[hid[p]] = (uint16_t)p;
        }
        i0 = i1;
    }
}

template <typename IDX>
struct ContChain {
    const ContParams* P;
    uint32_t* sp; double* lf; double* undo; double* dEs; double* v; double* ps; IDX* hid; IDX* hpos;
    int mlast;
    double z;
    long long trefresh;
    uint32_t rep;
    uint64_t nd;
    int rloc;                // replica index inside this context (the GraphQuant slice caches are per replica)

    __device__ __forceinline__ int sbit(int x) const { return (int)((sp[x >> 5] >> (x & 31)) & 1u); }
    __device__ __forceinline__ bool is_nb(const int32_t* Ax, int q) const { return !(P->ea_form && q > 0 && Ax[q] == Ax[q - 1]); }
    // spinflip!(X, C, move): bit flip + update_cache! (RRG.jl:576-617, EA.jl:613-653) with the undo record
    __device__ void flip(int move)
    {
        sp[move >> 5] ^= 1u << (move & 31);
        if (P->qkind) {                          // GraphQuant: integer slices cache nothing (delta_energy is recomputed from the spins);
            if (P->qkind >= 3) slice_cache_update(qview(), move);      // GraphSKNormal / sparse Float64 slices keep their Float64 fields (SK.jl:239-276, EA.jl:613-653)
            return;
        }
        const int K = P->K;
        const int32_t* Ax = P->A + (size_t)move * K;
        const double* Jx = P->J + (size_t)move * K;
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
        double vv = 0.0;
        for (int k = 0; k < K; ++k) {
            const int y = Ax[k];
            if (!(k > 0 && Ax[k] == Ax[k - 1])) { vv = lf[y]; undo[k] = vv; }
            const double c = (sx ^ sbit(y)) ? -4.0 : 4.0;
            vv = vv - c * Jx[k];
            if (k == K - 1 || Ax[k + 1] != y) lf[y] = vv;
        }
        undo[K] = lfm;
        lf[move] = -lfm;
        mlast = move;
    }
    // delta_energy(X0, C, i) of the inner DiscrGraph (RRG.jl:236-244 / EA.jl:266-275), recomputed from the spins
    __device__ __forceinline__ long long dE0(int i) const
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
    __device__ __forceinline__ RrrView qview() const
    {
        RrrView v{};
        v.sp = sp; v.N = P->N; v.Nk = P->qNk; v.M = P->qM; v.K = P->K; v.A = P->A; v.J = P->qJ; v.fourK = P->fourK;
        if (P->qkind == 2) { v.Jb = P->qJb; v.Wk = P->qWk; v.sN = P->qsN; }
        if (P->qkind == 3) {
            v.Jd = P->qJd;
            v.slf = P->qslf + (size_t)rloc * 2 * P->qM * P->qNk; v.smv = P->qsmv + (size_t)rloc * P->qM; v.scur = P->qscur + (size_t)rloc * P->qM;
        }
        if (P->qkind == 4) {
            v.Jf = P->qJf;
            v.flf = P->qflf + (size_t)rloc * P->qM * P->qNk; v.fundo = P->qfundo + (size_t)rloc * P->qM * (P->K + 1); v.fml = P->qfml + (size_t)rloc * P->qM;
        }
        v.nk_magic = (uint32_t)((0x100000000ull + (uint32_t)P->qNk - 1u) / (uint32_t)P->qNk);
        v.wide = P->N > 65535 ? 1 : 0;      // the slice of a spin by division (the multiply-high form is exact below 2^16 only)
        return v;
    }
    // neighbors(X, move) in the reference's order; returns their number (at most K + 2)
    __device__ __forceinline__ int nbrs(int move, int* out) const
    {
        int n = 0;
        if (P->qkind) {
            const RrrView v = qview();
            int j1, j2;
            qt_nb(v, move, j1, j2);
            out[n++] = j1; out[n++] = j2;
            const int k = move / P->qNk, x = move - k * P->qNk;
            const int32_t* Ax = P->A + (size_t)x * P->K;
            for (int q = 0; q < P->K; ++q) {
                if (q > 0 && Ax[q] == Ax[q - 1]) continue;                 // uA of the slice graph (EA.jl:158)
                out[n++] = Ax[q] + k * P->qNk;
            }
            return n;
        }
        const int32_t* Ax = P->A + (size_t)move * P->K;
        for (int q = 0; q < P->K; ++q) {
            if (!is_nb(Ax, q)) continue;
            out[n++] = Ax[q];
        }
        return n;
    }
    // neighbors(X, move) as (count, q-th neighbour): sparse graphs go through the small list `nb`, a GraphQuant over dense slices has the
    // Trotter pair followed by the Nk - 1 other spins of the slice in index order (AllButOne, SK.jl:142,297; QT.jl:288-321)
    __device__ __forceinline__ bool dense() const { return P->qkind == 2 || P->qkind == 3; }
    __device__ __forceinline__ int nb_count(int move, int* nb) const { return dense() ? P->qNk + 1 : nbrs(move, nb); }
    __device__ __forceinline__ int nb_at(int move, int q, const int* nb) const
    {
        if (!dense()) return nb[q];
        int j1, j2;
        const RrrView v = qview();
        qt_nb(v, move, j1, j2);
        if (q == 0) return j1;
        if (q == 1) return j2;
        const int k = move / P->qNk, x = move - k * P->qNk, jj = q - 2;
        return k * P->qNk + (jj < x ? jj : jj + 1);
    }
    __device__ __forceinline__ double dE(int i) const          // RRG.jl:619-625; DoubleGraph: convert(Float64, dE0 + dE1), :493-497
    {
        if (P->qkind) { const RrrView v = qview(); return (double)qt_delta(v, i) * P->fourK + any_residual(v, i); }
        if (P->dJ) return (double)(dE0(i) * P->lev_mul) / P->lev_div + (-lf[i]);
        return -lf[i];
    }

    // ---- DynamicSampler (DynamicSamplers.jl:84-176) ----
    __device__ void refresh()
    {
        double zz = 0.0;
        for (int i = 0; i < P->N2; ++i) zz += v[i];
        z = zz;
        for (int k = 0; k < P->N2 - 1; ++k) ps[k] = 0.0;
        for (int i = 0; i < P->N; ++i) {
            const double vi = v[i];
            int k = 0, u = 1 << (P->levs - 1), off = 1;
            for (int lev = 0; lev < P->levs; ++lev) {
                if ((i & u) == 0) { ps[off - 1 + k] += vi; k *= 2; }
                else k = 2 * k + 1;
                u >>= 1; off *= 2;
            }
        }
        trefresh = 0;
    }
    __device__ void set(int i, double x)
    {
        if (trefresh >= (P->N > 100 ? P->N : 100)) refresh();
        trefresh += 1;
        const double d = x - v[i];
        v[i] = x;
        z += d;
        int k = 0, u = 1 << (P->levs - 1), off = 1;
        for (int lev = 0; lev < P->levs; ++lev) {
            if ((i & u) == 0) { ps[off - 1 + k] += d; k *= 2; }
            else k = 2 * k + 1;
            u >>= 1; off *= 2;
        }
    }
    __device__ int getel(double x)
    {
        for (;;) {
            x *= z;
            int k = 0, off = 1;
            for (int lev = 0; lev < P->levs; ++lev) {
                const double p = ps[off - 1 + k];
                k *= 2;
                if (x > p) { x -= p; k += 1; }
                off *= 2;
            }
            if (k >= P->N || v[k] == 0) {
                if (!(trefresh > 0)) return -1;
                refresh();
                continue;
            }
            return k;
        }
    }
    // apply_move!(X, C, move, cache::DeltaECacheCont): DeltaE.jl:376-410; returns c = z / z'
    __device__ double apply_move(int move)
    {
        flip(move);
        const double z0 = z;
        dEs[move] = dE(move);
        set(move, prior_of(P->beta * dEs[move]));
        int nb[kContKmax + 2];
        const int nn = nb_count(move, nb);
        for (int q = 0; q < nn; ++q) {
            const int j = nb_at(move, q, nb);
            dEs[j] = dE(j);
            set(j, prior_of(P->beta * dEs[j]));
        }
        return z0 / z;
    }

    // ---- wtmMC heap (keys in dEs[], by heap position) ----
    __device__ __forceinline__ double uniform()
    {
        const uint64_t n = nd++, blk = n >> 1;
        const Philox4 o = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), rep, TAG_WTM | (P->call << 8), P->k0, P->k1);
        const uint64_t u = (n & 1u) ? (((uint64_t)o.w[2] << 32) | o.w[3]) : (((uint64_t)o.w[0] << 32) | o.w[1]);
        return (double)(u >> 11) * 0x1.0p-53;
    }
    __device__ __forceinline__ double gen_wt(double d)           // gen_wt(tauDE(d)): WaitingTimes.jl:16-22
    {
        const double e = det_exp(P->beta * d);
        return -(e > 1.0 ? e : 1.0) * det_log1p(-uniform());
    }
    __device__ __forceinline__ bool before(double ta, int a, double tb, int b) const { return ta < tb || (ta == tb && a < b); }
    __device__ void sift_down(int pos, int n)
    {
        double* ht = dEs;
        const double t = ht[pos];
        const int id = hid[pos];
        for (;;) {
            int c = 2 * pos + 1;
            if (c >= n) break;
            if (c + 1 < n && before(ht[c + 1], hid[c + 1], ht[c], hid[c])) c += 1;
            if (!before(ht[c], hid[c], t, id)) break;
            ht[pos] = ht[c]; hid[pos] = hid[c]; hpos[hid[c]] = (IDX)pos;
            pos = c;
        }
        ht[pos] = t; hid[pos] = (IDX)id; hpos[id] = (IDX)pos;
    }
    __device__ void sift_up(int pos)
    {
        double* ht = dEs;
        const double t = ht[pos];
        const int id = hid[pos];
        while (pos > 0) {
            const int par = (pos - 1) >> 1;
            if (!before(t, id, ht[par], hid[par])) break;
            ht[pos] = ht[par]; hid[pos] = hid[par]; hpos[hid[par]] = (IDX)pos;
            pos = par;
        }
        ht[pos] = t; hid[pos] = (IDX)id; hpos[id] = (IDX)pos;
    }
    __device__ __forceinline__ void heap_update(int i, double t)
    {
        const int pos = hpos[i];
        const double old = dEs[pos];
        dEs[pos] = t;
        if (t < old) sift_up(pos); else sift_down(pos, P->N);
    }

    // ---- EOCacheCont (DeltaE.jl:557-635): v[p] = dE of the site ranked p (ascending), hid[p] = that site, hpos = inverse ----
    __device__ __forceinline__ void eo_swap(int a, int b)
    {
        const double x = v[a]; v[a] = v[b]; v[b] = x;
        const IDX h = hid[a]; hid[a] = hid[b]; hid[b] = h;
    }
    __device__ __forceinline__ bool eo_lt(int a, int b) const { return v[a] < v[b] || (v[a] == v[b] && hid[a] < hid[b]); }
    __device__ void eo_sift(int root, int end)
    {
        for (;;) {
            int ch = 2 * root + 1;
            if (ch >= end) break;
            if (ch + 1 < end && eo_lt(ch, ch + 1)) ch += 1;
            if (!eo_lt(root, ch)) break;
            eo_swap(root, ch);
            root = ch;
        }
    }
    // rank = sortperm(dEs): ascending (dE, site) — a stable sort of 1:N (DeltaE.jl:568); heapsort, the order is total
    __device__ void eo_sort_all(int n)
    {
        for (int st = n / 2 - 1; st >= 0; --st) eo_sift(st, n);
        for (int e = n - 1; e > 0; --e) { eo_swap(0, e); eo_sift(0, e); }
    }
    // site j takes the value x: slide it to its place among the others (which stay sorted); nties = adjacent equal pairs
    __device__ int eo_reinsert(int j, double x, int nties)
    {
        const int n = P->N;
        int p = hpos[j];
        const double old = v[p];
        if (p > 0 && v[p - 1] == old) nties -= 1;
        if (p < n - 1 && v[p + 1] == old) nties -= 1;
        if (p > 0 && p < n - 1 && v[p - 1] == v[p + 1]) nties += 1;
        if (x > old) {
            while (p < n - 1 && v[p + 1] < x) { v[p] = v[p + 1]; const IDX s = hid[p + 1]; hid[p] = s; hpos[s] = (IDX)p; p += 1; }
        } else {
            while (p > 0 && v[p - 1] > x) { v[p] = v[p - 1]; const IDX s = hid[p - 1]; hid[p] = s; hpos[s] = (IDX)p; p -= 1; }
        }
        if (p > 0 && p < n - 1 && v[p - 1] == v[p + 1]) nties -= 1;
        if (p > 0 && v[p - 1] == x) nties += 1;
        if (p < n - 1 && v[p + 1] == x) nties += 1;
        v[p] = x; hid[p] = (IDX)j; hpos[j] = (IDX)p;
        return nties;
    }
    __device__ void eo_order_ties(uint64_t g) 
    {
        // the tie keys address the site in 16 bits of the Philox tag: extremal_opt stays at N <= 65535 (refused on the host beyond)
        if constexpr (sizeof(IDX) == 2) eo_order_ties_impl(v, hid, hpos, reinterpret_cast<unsigned long long*>(ps), P->N, g, rep, P->k0, P->k1);
    }
};

template <typename IDX>
__global__ __launch_bounds__(kRrrThreads) void cont_sparse_kernel(ContParams P)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= P.R) return;
    const int N = P.N, K = P.K;
    ContChain<IDX> c;
    c.P = &P;
    c.sp = P.spins + (size_t)r * P.W; c.lf = P.lf + (size_t)r * N; c.undo = P.undo + (size_t)r * (K + 1);
    c.dEs = P.dEs + (size_t)r * N; c.v = P.v + (size_t)r * P.N2; c.ps = P.ps + (size_t)r * P.N2;
    c.hid = static_cast<IDX*>(P.hid) + (size_t)r * N; c.hpos = static_cast<IDX*>(P.hpos) + (size_t)r * N;
    c.mlast = -1; c.z = 0.0; c.trefresh = 0; c.nd = 0;
    c.rep = P.replica0 + (uint32_t)r;
    c.rloc = r;
    double* const sf = P.S.sf + (size_t)r * kSmpF;
    long long* const sq = P.S.si + (size_t)r * kSmpI;
    const bool resume = P.S.resume != 0;
    double E = 0.0;
    if (resume) {
        // a resumed call: local fields and their undo record, the slice caches of a GraphQuant, the sampler's tree (or the heap, or the
        // ranking) are where the previous call left them; the scalars of the chain come from the slabs
        c.mlast = (int)sq[SI_MLAST]; c.z = sf[SF_Z]; c.trefresh = sq[SI_TREF]; c.nd = (uint64_t)sq[SI_ND];
        E = sf[SF_E];
    } else {
    // energy(X, C): RRG.jl:546-574 / EA.jl:584-611
    double E1 = 0.0;
    for (int i = 0; i < N && !P.qkind; ++i) {
        const int sx = 2 * c.sbit(i) - 1;
        double fl = 0.0;
        for (int q = 0; q < K; ++q) {
            const int sy = 2 * c.sbit(P.A[(size_t)i * K + q]) - 1;
            fl = fl - P.J[(size_t)i * K + q] * (double)sx * (double)sy;
        }
        E1 = E1 + fl;
        c.lf[i] = 2.0 * fl;
    }
    E = E1 / 2;
    if (P.dJ) {                                  // energy(X::DoubleGraph, C) = convert(Float64, E0 + E1): RRG.jl:326-360
        long long n0 = 0;
        for (int i = 0; i < N; ++i) n0 -= c.dE0(i) / 2;
        E = (double)((n0 / 2) * P.lev_mul) / P.lev_div + E;
    }
    if (P.qkind) {                               // energy(X::GraphQuant, C): QT.jl:185-199 (as rrr_init_kernel)
        const RrrView v = c.qview();
        long long n0 = 0;
        for (int i = 0; i < P.qNk; ++i) {
            int sj = c.sbit(i + (P.qM - 1) * P.qNk);
            for (int k = 0; k < P.qM; ++k) { const int sk = c.sbit(i + k * P.qNk); n0 -= 1 - 2 * (sk ^ sj); sj = sk; }
        }
        E = (double)n0 * P.fourK / 4;
        for (int k = 0; k < P.qM && P.qkind == 4; ++k) {      // sparse Float64 slices: rebuild the slice cache as energy does (EA.jl:584-611)
            double* lf = v.flf + (size_t)k * P.qNk;
            double E1 = 0.0;
            for (int x = 0; x < P.qNk; ++x) {
                const int sx = 2 * c.sbit(k * P.qNk + x) - 1;
                double fl = 0.0;
                for (int q = 0; q < K; ++q) {
                    const int sy = 2 * c.sbit(k * P.qNk + P.A[(size_t)x * K + q]) - 1;
                    fl = fl - P.qJf[(size_t)x * K + q] * (double)sx * (double)sy;
                }
                E1 = E1 + fl;
                lf[x] = 2.0 * fl;
            }
            v.fml[k] = -1;
            for (int q = 0; q <= K; ++q) v.fundo[(size_t)k * (K + 1) + q] = 0.0;
            E += (E1 / 2) / (double)P.qM;
        }
        for (int k = 0; k < P.qM && P.qkind < 3; ++k) {
            long long n = 0;
            for (int i = 0; i < P.qNk; ++i) n -= slice_delta(v, k * P.qNk + i) / 2;
            n /= 2;
            E += P.qkind == 2 ? ((double)n / P.qsN) / (double)P.qM : (double)n / (double)P.qM;      // GraphSK: n / sqrt(Nk) (SK.jl:95)
        }
        for (int k = 0; k < P.qM && P.qkind == 3; ++k) {      // GraphSKNormal slices: rebuild the slice cache as energy does (SK.jl:212-237)
            double* lf = v.slf + ((size_t)0 * P.qM + k) * P.qNk;
            double* ll = v.slf + ((size_t)1 * P.qM + k) * P.qNk;
            double n = 0.0;
            for (int i = 0; i < P.qNk; ++i) {
                const double* Ji = P.qJd + (size_t)i * P.qNk;
                const int si = c.sbit(k * P.qNk + i);
                double lfi = 0.0;
                for (int j = 0; j < P.qNk; ++j) lfi += (double)(1 - 2 * (si ^ c.sbit(k * P.qNk + j))) * Ji[j];
                lf[i] = 2 * lfi;
                ll[i] = 0.0;
                n -= lfi;
            }
            n /= 2;
            v.smv[k] = -1; v.scur[k] = 0;
            E += n / (double)P.qM;
        }
    }
    }
    long long accepted = 0, second = 0, ns = 0, itdone = 0;
    int bad = 0;
    double t = 0.0;

    if (P.mode == 2) {
        const double st = P.stepf / (double)N, tmax = st * (double)(P.samples_before + P.iters);
        double nextstep = st;
        if (resume) { t = sf[SF_TIME]; nextstep = sf[SF_NEXT]; }      // the run's heap, global time and next sample time carry on
        else {
        for (int i = 0; i < N; ++i) { c.dEs[i] = c.gen_wt(c.dE(i)); c.hid[i] = (IDX)i; c.hpos[i] = (IDX)i; }
        for (int pos = N / 2 - 1; pos >= 0; --pos) c.sift_down(pos, N);
        }
        bool out = false;
        while (t < tmax && !out) {
            const double tp = c.dEs[0];
            const int move = c.hid[0];
            while (tp >= nextstep) {
                P.Es[(size_t)ns * P.Rp + r] = E; ns += 1;
                nextstep += st;
                if (nextstep > tmax + 1e-10) { out = true; break; }
            }
            if (out) break;
            t = tp;
            const double d = c.dE(move);
            c.flip(move);
            c.heap_update(move, t + c.gen_wt(-d));
            int nb[kContKmax + 2];
            const int nn = c.nb_count(move, nb);
            for (int q = 0; q < nn; ++q) { const int j = c.nb_at(move, q, nb); c.heap_update(j, t + c.gen_wt(c.dE(j))); }
            E += d;
            accepted += 1;
        }
        for (; ns < P.iters; ++ns) { P.Es[(size_t)ns * P.Rp + r] = E; nextstep += st; }      // not reached: the loop always emits `samples` samples
        second = accepted; itdone = ns;
        sf[SF_TIME] = t; sf[SF_NEXT] = nextstep;
    } else if (P.mode == 3) {
        // extremal_opt: RRRMC.jl:474-521; stats = (iterations, itmin, iterations), t_out = Emin
        uint32_t* cm = P.cmin + (size_t)r * P.W;
        int nties = 0;
        double Emin = E;
        long long itmin = 0;
        if (resume) {                            // the ranking (v, hid, hpos), Emin / Cmin / itmin of the run carry on
            for (int p = 1; p < N; ++p) if (c.v[p - 1] == c.v[p]) nties += 1;
            Emin = sf[SF_EMIN]; itmin = sq[SI_ITMIN];
        } else {
        for (int i = 0; i < N; ++i) { c.v[i] = c.dE(i); c.hid[i] = (IDX)i; }
        c.eo_sort_all(N);
        for (int p = 0; p < N; ++p) { c.hpos[c.hid[p]] = (IDX)p; if (p > 0 && c.v[p - 1] == c.v[p]) nties += 1; }
        for (int w = 0; w < P.W; ++w) cm[w] = c.sp[w];
        }
        const double z = P.ftau[N - 1];
        long long next_sample = P.S.samp0;
        for (long long it = 1; it <= P.iters; ++it) {
            if (it == next_sample) { next_sample += P.step; P.Es[(size_t)ns * P.Rp + r] = E; ns += 1; }
            const uint64_t g = P.g0 + (uint64_t)it;
            const Philox4 o = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), c.rep, TAG_RRR | (3u << 8), P.k0, P.k1);
            const double rr = (1 - (double)((((uint64_t)o.w[0] << 32) | o.w[1]) >> 11) * 0x1.0p-53) * z;      // rand_move: DeltaE.jl:577-590
            int lo = 0, hi = N;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (P.ftau[mid] < rr) lo = mid + 1; else hi = mid; }
            if (lo > N - 1) lo = N - 1;
            const int move = c.hid[lo];
            const double dE = c.v[lo];
            c.flip(move);                                                                                    // apply_move!: :592-609
            nties = c.eo_reinsert(move, c.dE(move), nties);
            int enb[kContKmax + 2];
            const int enn = c.nb_count(move, enb);                                                           // neighbors(X, move): a GraphQuant's too
            for (int q = 0; q < enn; ++q) { const int j = c.nb_at(move, q, enb); nties = c.eo_reinsert(j, c.dE(j), nties); }
            if (nties > 0) c.eo_order_ties(g);
            E += dE;
            if (E < Emin) {
                Emin = E; itmin = P.S.it0 + it;
                for (int w = 0; w < P.W; ++w) cm[w] = c.sp[w];
            }
        }
        accepted = P.iters; second = itmin; itdone = P.iters;
        t = Emin;
        sf[SF_EMIN] = Emin; sq[SI_ITMIN] = itmin;
    } else {
        if (!resume) {
        for (int i = 0; i < P.N2; ++i) c.v[i] = 0.0;
        for (int i = 0; i < N; ++i) { c.dEs[i] = c.dE(i); c.v[i] = prior_of(P.beta * c.dEs[i]); }      // DeltaECacheCont: DeltaE.jl:304-313
        c.refresh();
        }
        if (P.mode == 0) {
            double acc_rate = resume ? sf[SF_ACC] : 0.5;
            long long next_sample = P.S.samp0;       // iterations k * step of the run: a counter instead of a 64-bit modulo per iteration
            for (long long it = 1; it <= P.iters && !bad; ++it) {
                if (it == next_sample) { next_sample += P.step; P.Es[(size_t)ns * P.Rp + r] = E; ns += 1; }
                const uint64_t g = P.g0 + (uint64_t)it;
                const Philox4 o = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), c.rep, TAG_RRR, P.k0, P.k1);
                const double u0 = (double)((((uint64_t)o.w[0] << 32) | o.w[1]) >> 11) * 0x1.0p-53;
                const Philox4 o2 = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), c.rep, TAG_RRR | (1u << 8), P.k0, P.k1);
                const double u1 = (double)((((uint64_t)o2.w[0] << 32) | o2.w[1]) >> 11) * 0x1.0p-53;
                bool acc = false;
                const int move = c.getel(u0);                                   // rand_move: DeltaE.jl:327-333
                if (move < 0) { bad = 1; break; }
                const double dE = c.dEs[move];
                if (acc_rate < P.staged_thr) {
                    second += 1;
                    const double z0 = c.z;
                    c.flip(move);                                               // compute_staged!: DeltaE.jl:357-374
                    int sj[kContKmax + 1], nst = 0;
                    double sdE[kContKmax + 1], spv[kContKmax + 1];
                    sj[nst] = move; sdE[nst] = c.dE(move); spv[nst] = prior_of(P.beta * sdE[nst]); ++nst;
                    const int32_t* Ax = P.A + (size_t)move * K;
                    for (int q = 0; q < K; ++q) {
                        if (!c.is_nb(Ax, q)) continue;
                        sj[nst] = Ax[q]; sdE[nst] = c.dE(Ax[q]); spv[nst] = prior_of(P.beta * sdE[nst]); ++nst;
                    }
                    c.flip(move);
                    double zp = c.z;                                            // compute_reverse_probabilities!: :345-355
                    for (int q = 0; q < nst; ++q) zp += spv[q] - c.v[sj[q]];
                    if (zp < 2.2250738585072014e-308) zp = 2.2250738585072014e-308;
                    if (zp > (double)N) zp = (double)N;
                    if (u1 < z0 / zp) {
                        c.flip(move);
                        for (int q = 0; q < nst; ++q) { c.dEs[sj[q]] = sdE[q]; c.set(sj[q], spv[q]); }      // apply_staged!
                        E += dE; accepted += 1; acc = true;
                    }
                } else {
                    const double cc = c.apply_move(move);
                    if (u1 < cc) { E += dE; accepted += 1; acc = true; }
                    else c.apply_move(move);
                }
                acc_rate = acc_rate * (1 - P.lambda) + (acc ? 1.0 : 0.0) * P.lambda;
            }
            itdone = P.iters;
            sf[SF_ACC] = acc_rate;
        } else {
            // (a resumed call continues the run's loop with `iters` more iterations allowed; the pending move is drawn again: same draw)
            long long it = 0, nextstep = P.step, m = 0, limit = P.iters;
            if (resume) { it = sq[SI_IT]; nextstep = sq[SI_NEXT]; m = sq[SI_M]; limit += sq[SI_LIMIT]; }
            const bool over = resume && nextstep > limit && nextstep > P.step;  // (a resumed call whose allowance does not reach the run's next sample point makes no move: the reference's loop ends with its last sample, RRRMC.jl:340-343)
            while (!over && it < limit) {
                const uint64_t g = P.g0 + (uint64_t)(m + 1);
                const Philox4 o3 = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), c.rep, TAG_RRR | (2u << 8), P.k0, P.k1);
                const double us = (double)((((uint64_t)o3.w[0] << 32) | o3.w[1]) >> 11) * 0x1.0p-53;
                double b = c.z / (double)N;                                     // rand_skip: DeltaE.jl:319-325
                if (b < 2.2250738585072014e-308) b = 2.2250738585072014e-308;
                if (b > 1.0) b = 1.0;
                const double skipf = floor(det_log1p(-us) / det_log1p(-b));
                const long long skip = skipf >= 9.0e18 ? (long long)9.0e18 : (long long)skipf;
                const Philox4 o = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), c.rep, TAG_RRR, P.k0, P.k1);
                const int move = c.getel((double)((((uint64_t)o.w[0] << 32) | o.w[1]) >> 11) * 0x1.0p-53);
                if (move < 0) { bad = 1; break; }
                const double dE = c.dEs[move];
                bool out = false;
                while (it + skip + 1 >= nextstep) {
                    if (nextstep > limit) { out = true; break; }        // (fewer iterations allowed than `step`: the sample point lies beyond this call — no sample, no move)
                    P.Es[(size_t)ns * P.Rp + r] = E; ns += 1;
                    nextstep += P.step;
                    if (nextstep > limit) { out = true; break; }
                }
                if (out) break;
                c.apply_move(move);                                             // apply_step_bkl!: RRRMC.jl:294-295
                m += 1;
                it += skip + 1;
                E += dE;
                accepted += 1;
            }
            second = accepted; itdone = it;
            sq[SI_IT] = it; sq[SI_NEXT] = nextstep; sq[SI_M] = m; sq[SI_LIMIT] = limit;
        }
    }
    P.E_cur[r] = E;
    P.stats[(size_t)r * 3] = accepted; P.stats[(size_t)r * 3 + 1] = second; P.stats[(size_t)r * 3 + 2] = itdone;
    P.t_out[r] = t;
    P.status[r] = bad;
    sf[SF_E] = E; sf[SF_Z] = c.z;
    sq[SI_MLAST] = c.mlast; sq[SI_TREF] = c.trefresh; sq[SI_ND] = (long long)c.nd;
}

// ---------------------------------------------------------------------------------------------------
// extremal_opt with EOCacheCont, one WAVEFRONT per replica (N <= kEoWaveMaxN): the ranking (v, hid, hpos: 12 bytes per spin) lives in
// LDS; the chain itself (draw, flip, local fields) is executed redundantly by the 64 lanes — wave-uniform, so every global access is
// one transaction — and the lanes split what the reference spends its time on, the re-ranking: the new place of a changed spin is
// found by bisection and the entries between its old and new place move by one, 64 per step.  Same results as mode 3 of
// cont_sparse_kernel (which remains for larger N), several tens of times faster: a moved spin's dE changes sign, so it travels across
// most of the ranking at every move.
// ---------------------------------------------------------------------------------------------------
constexpr int kEoWaveMaxN = 13000;
__host__ __device__ inline size_t eo_wave_lds_bytes(int N) { return ((size_t)N * 12 + 15) & ~(size_t)15; }

__device__ __forceinline__ int wave_sum_i32(int x)
{
    for (int o = 32; o; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

__global__ __launch_bounds__(64) void eo_cont_wave_kernel(ContParams P)
{
    extern __shared__ __align__(16) unsigned char eo_lds[];
    const int r = blockIdx.x, lane = threadIdx.x;
    const int N = P.N, K = P.K;
    double* v = reinterpret_cast<double*>(eo_lds);
    uint16_t* hid = reinterpret_cast<uint16_t*>(v + N);
    uint16_t* hpos = hid + N;
    ContChain<uint16_t> c;
    c.P = &P;
    c.sp = P.spins + (size_t)r * P.W; c.lf = P.lf + (size_t)r * N; c.undo = P.undo + (size_t)r * (K + 1);
    c.dEs = P.dEs + (size_t)r * N; c.v = P.v + (size_t)r * P.N2; c.ps = P.ps + (size_t)r * P.N2;
    c.hid = static_cast<uint16_t*>(P.hid) + (size_t)r * N; c.hpos = static_cast<uint16_t*>(P.hpos) + (size_t)r * N;
    c.mlast = -1; c.z = 0.0; c.trefresh = 0; c.nd = 0;
    c.rep = P.replica0 + (uint32_t)r; c.rloc = r;
    unsigned long long* key = reinterpret_cast<unsigned long long*>(c.ps);
    uint32_t* cm = P.cmin + (size_t)r * P.W;
    double* const sf = P.S.sf + (size_t)r * kSmpF;
    long long* const sq = P.S.si + (size_t)r * kSmpI;
    const bool resume = P.S.resume != 0;
    double E = 0.0, Emin = 0.0;
    long long itmin = 0, ns = 0;
    int nties = 0;
    if (resume) {
        // a resumed call: fields and undo record are in HBM; the ranking comes back from the arrays mode 3 of cont_sparse_kernel keeps it in
        // (v[p], hid[p], hpos[site]: the previous call wrote it there), E / Emin / itmin from the slabs
        for (int p = lane; p < N; p += 64) { v[p] = c.v[p]; hid[p] = c.hid[p]; hpos[p] = c.hpos[p]; }
        c.mlast = (int)sq[SI_MLAST];
        E = sf[SF_E]; Emin = sf[SF_EMIN]; itmin = sq[SI_ITMIN];
        __syncthreads();
        int part = 0;
        for (int p = 1 + lane; p < N; p += 64) part += v[p - 1] == v[p] ? 1 : 0;
        nties = wave_sum_i32(part);
    } else {
    // energy(X, C) (RRG.jl:546-574 / EA.jl:584-611): the fields site-parallel, their sum in site order
    for (int i = lane; i < N; i += 64) {
        const int sx = 2 * c.sbit(i) - 1;
        double fl = 0.0;
        for (int q = 0; q < K; ++q) {
            const int sy = 2 * c.sbit(P.A[(size_t)i * K + q]) - 1;
            fl = fl - P.J[(size_t)i * K + q] * (double)sx * (double)sy;
        }
        c.lf[i] = 2.0 * fl;
    }
    __syncthreads();
    double E1 = 0.0;
    for (int base = 0; base < N; base += 64) {
        const int i = base + lane;
        const double h = i < N ? c.lf[i] * 0.5 : 0.0;          // = fl exactly
        const int m = N - base < 64 ? N - base : 64;
        for (int l = 0; l < m; ++l) E1 = E1 + __shfl(h, l);
    }
    E = E1 / 2;
    if (P.dJ) {                                  // energy(X::DoubleGraph, C) = convert(Float64, E0 + E1): RRG.jl:326-360
        int part = 0;
        for (int i = lane; i < N; i += 64) part -= (int)(c.dE0(i) / 2);
        const long long n0 = (long long)wave_sum_i32(part);
        E = (double)((n0 / 2) * P.lev_mul) / P.lev_div + E;
    }
    // rank = sortperm(dEs) (DeltaE.jl:568): the place of a spin = the number of spins before it in the order (dE, site)
    for (int i = lane; i < N; i += 64) { const double x = c.dE(i); v[i] = x; c.dEs[i] = x; }
    __syncthreads();
    for (int i = lane; i < N; i += 64) {
        const double xi = v[i];
        int cnt = 0;
        for (int j = 0; j < N; ++j) { const double xj = v[j]; cnt += (xj < xi || (xj == xi && j < i)) ? 1 : 0; }
        hpos[i] = (uint16_t)cnt;
    }
    __syncthreads();
    for (int i = lane; i < N; i += 64) { const int p = hpos[i]; v[p] = c.dEs[i]; hid[p] = (uint16_t)i; }
    __syncthreads();
    {
        int part = 0;
        for (int p = 1 + lane; p < N; p += 64) part += v[p - 1] == v[p] ? 1 : 0;
        nties = wave_sum_i32(part);
    }
    Emin = E;
    for (int w = lane; w < P.W; w += 64) cm[w] = c.sp[w];
    }
    const double z = P.ftau[N - 1];
    long long next_sample = P.S.samp0;

    // site j takes the value x (ContChain::eo_reinsert, the slide split over the lanes)
    auto reinsert = [&](int j, double x) {
        int p = hpos[j];
        const double old = v[p];
        if (p > 0 && v[p - 1] == old) nties -= 1;
        if (p < N - 1 && v[p + 1] == old) nties -= 1;
        if (p > 0 && p < N - 1 && v[p - 1] == v[p + 1]) nties += 1;
        if (x > old) {
            int lo = p + 1, hi = N;                              // first place in (p, N) whose value is >= x
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (v[mid] < x) lo = mid + 1; else hi = mid; }
            const int q = lo - 1;
            for (int base = p + 1; base <= q; base += 64) {
                const int i = base + lane;
                double xv = 0.0; uint16_t s = 0;
                if (i <= q) { xv = v[i]; s = hid[i]; }
                __syncthreads();
                if (i <= q) { v[i - 1] = xv; hid[i - 1] = s; hpos[s] = (uint16_t)(i - 1); }
                __syncthreads();
            }
            p = q;
        } else if (x < old) {
            int lo = 0, hi = p;                                  // first place in [0, p) whose value is > x
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (v[mid] <= x) lo = mid + 1; else hi = mid; }
            const int q = lo;
            for (int top = p - 1; top >= q; top -= 64) {
                const int i = top - lane;
                double xv = 0.0; uint16_t s = 0;
                if (i >= q) { xv = v[i]; s = hid[i]; }
                __syncthreads();
                if (i >= q) { v[i + 1] = xv; hid[i + 1] = s; hpos[s] = (uint16_t)(i + 1); }
                __syncthreads();
            }
            p = q;
        }
        if (p > 0 && p < N - 1 && v[p - 1] == v[p + 1]) nties -= 1;
        if (p > 0 && v[p - 1] == x) nties += 1;
        if (p < N - 1 && v[p + 1] == x) nties += 1;
        v[p] = x; hid[p] = (uint16_t)j; hpos[j] = (uint16_t)p;
        __syncthreads();
    };

    for (long long it = 1; it <= P.iters; ++it) {
        if (it == next_sample) { next_sample += P.step; if (lane == 0) P.Es[(size_t)ns * P.Rp + r] = E; ns += 1; }
        const uint64_t g = P.g0 + (uint64_t)it;
        const Philox4 o = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), c.rep, TAG_RRR | (3u << 8), P.k0, P.k1);
        const double rr = (1 - (double)((((uint64_t)o.w[0] << 32) | o.w[1]) >> 11) * 0x1.0p-53) * z;      // rand_move: DeltaE.jl:577-590
        int lo = 0, hi = N;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (P.ftau[mid] < rr) lo = mid + 1; else hi = mid; }
        if (lo > N - 1) lo = N - 1;
        const int move = hid[lo];
        const double dE = v[lo];
        __syncthreads();
        c.flip(move);                                                                                    // apply_move!: :592-609
        __syncthreads();
        reinsert(move, c.dE(move));
        const int32_t* Ax = P.A + (size_t)move * K;
        for (int q = 0; q < K; ++q) {
            if (!c.is_nb(Ax, q)) continue;
            reinsert(Ax[q], c.dE(Ax[q]));
        }
        if (nties > 0) { eo_order_ties_impl(v, hid, hpos, key, N, g, c.rep, P.k0, P.k1); __syncthreads(); }
        E += dE;
        if (E < Emin) {
            Emin = E; itmin = P.S.it0 + it;
            for (int w = lane; w < P.W; w += 64) cm[w] = c.sp[w];
        }
    }
    __syncthreads();
    for (int p = lane; p < N; p += 64) { c.v[p] = v[p]; c.hid[p] = hid[p]; c.hpos[p] = hpos[p]; }      // the ranking, where a resumed call finds it
    if (lane == 0) {
        P.E_cur[r] = E;
        P.stats[(size_t)r * 3] = P.iters; P.stats[(size_t)r * 3 + 1] = itmin; P.stats[(size_t)r * 3 + 2] = P.iters;
        P.t_out[r] = Emin;
        P.status[r] = 0;
        sf[SF_E] = E; sf[SF_EMIN] = Emin; sq[SI_ITMIN] = itmin; sq[SI_MLAST] = c.mlast;
        sf[SF_Z] = 0.0; sq[SI_TREF] = 0; sq[SI_ND] = 0;
    }
}

// [W][N] uint64 (bit l of word (w, x) = spin x of replica 64 w + l)  <->  replica-contiguous [R][Wr] 32-bit words (bit = site & 31)
__global__ __launch_bounds__(256) void cont_spins_in_kernel(const unsigned long long* __restrict__ bs, uint32_t* __restrict__ spins, int N, int Wr, int R)
{
    const int w = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (w >= Wr || r >= R) return;
    uint32_t word = 0u;
    for (int b = 0; b < 32; ++b) {
        const int x = 32 * w + b;
        if (x < N) word |= (uint32_t)((bs[(size_t)(r >> 6) * N + x] >> (r & 63)) & 1ull) << b;
    }
    spins[(size_t)r * Wr + w] = word;
}
__global__ __launch_bounds__(256) void cont_spins_out_kernel(const uint32_t* __restrict__ spins, unsigned long long* __restrict__ bs, int N, int Wr, int R)
{
    const int x = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
    if (x >= N) return;
    unsigned long long word = 0ull;
    for (int b = 0; b < 64; ++b) {
        const int r = g * 64 + b;
        if (r < R) word |= (unsigned long long)((spins[(size_t)r * Wr + (x >> 5)] >> (x & 31)) & 1u) << b;
    }
    bs[(size_t)g * N + x] = word;
}

}  // namespace rrrmc
