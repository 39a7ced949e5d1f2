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
#include <type_traits>

#include "philox.hpp"
#include "sk_kernels.hpp"   // det_exp

namespace rrrmc {

constexpr uint32_t TAG_RRR = 8;
constexpr int kRrrThreads = 64;          // most replicas (threads) a workgroup of the thread-per-replica kernels holds; the host picks
                                        // fewer per workgroup when there are few replicas, to spread them over the CUs (rrr_tpb)
constexpr int kQL = 2;          // levels of allΔE(GraphQT) = (0.0, fourK), QT.jl:111

// ---- resumed calls (rrrmc_set_resume) -----------------------------------------------------------------------------------------------
// A reference sampler keeps its chain in local variables from the first to the last iteration, hook calls included (src/RRRMC.jl:175-212
// rrrMC, :322-350 bklMC, :389-416 wtmMC, :486-516 extremal_opt): the cache object, E, acc_rate, `it` / `nextstep`, the heap's global time,
// Emin / itmin.  A library call that RESUMES the previous one finds the arrays of the cache where the previous call left them (HBM) and
// these scalars in two per-replica slabs; the kernels load them instead of running energy(X, C) + gen_ΔEcache, and store them at their end.
// One layout for every kernel (each uses the entries it has):
constexpr int kSmpF = 40, kSmpI = 40;
enum SmpF { SF_Z = 0, SF_ACC = 1, SF_TIME = 2, SF_NEXT = 3, SF_EMIN = 4, SF_E = 5, SF_T0 = 8 /* T[0..15] */, SF_UNDO = 24 /* undo[0..K] of the wave builds */ };
enum SmpI { SI_IT = 0, SI_NEXT = 1, SI_M = 2, SI_ND = 3, SI_TREF = 4, SI_CUR = 5, SI_MLAST = 6, SI_EMIN = 7, SI_ITMIN = 8, SI_LIMIT = 9,
            SI_T0 = 16 /* t[0..15] set sizes */ };
struct SmpState {            // the same four fields in every parameter struct below
    double* sf;              // [R][kSmpF]
    long long* si;           // [R][kSmpI]
    int resume;              // 1: continue the run the slabs (and the cache arrays) describe
    long long samp0;         // the call's first sample is taken before its iteration samp0 (= step - (run iterations so far) % step)
    long long it0;           // iterations of the run before this call (extremal_opt's itmin counts from the start of the run)
};

struct RrrParams {
    // disorder of the slice graph (shared by slices and replicas)
    const int32_t* A;        // [Nk][K]
    const int8_t* J;         // [Nk][K]
    const uint32_t* Jb;      // [Nk][Wk]     binary GraphSK slices (GraphQSKT, QAliases.jl:34-43) instead of (A, J): row i of J as 32-bit words; else null
    double sN;               //              sqrt(Nk) (SK.jl:49), 0 for GraphRRG slices
    int Wk;
    // GraphSKNormal slices (GraphQSKNormalT, QAliases.jl:45-46; test/runtests.jl:80): Float64 couplings shared by the slices, and every
    // slice's own cache — lfields / lfields_last (swapped wholesale by the undo path, SK.jl:247-250) and move_last — per replica
    const double* Jd;        // [Nk][Nk] else null
    double* slf;             // [R][2][M][Nk]   the two field arrays of every slice
    int32_t* smv;            // [R][M]          move_last of every slice (-1 = none)
    uint8_t* scur;           // [R][M]          which of the two arrays is `lfields`
    // sparse Float64 slices (GraphQEAT = GraphQuant{fourK,GraphEANormal{twoD}}, QAliases.jl:50-83; GraphRRGNormal slices alike): couplings on the
    // slice graph's table A, and every slice's own LocalFields{Float64} — lfields, the live part of lfields_last (the K + 1 values the undo
    // path of update_cache! reads, EA.jl:613-653 / RRG.jl:576-617) and move_last — per replica
    const double* Jf;        // [Nk][K] else null
    double* flf;             // [R][M][Nk]
    double* fundo;           // [R][M][K+1]
    int32_t* fml;            // [R][M]          move_last of every slice (-1 = none)
    // per-replica state, replica-contiguous
    uint32_t* spins;         // [R][W]       bit x of replica r: word x >> 5, bit x & 31; x = slice * Nk + i
    uint8_t* cls;            // [R][N]       class of every spin: a + 2 * up  (DeltaECache.pos)
    void* sv;                // [R][4][N]    ArraySet.v of the four classes: uint16_t, or uint32_t when `wide` (N > 65 535)
    void* spos;              // [R][N]       position of the spin inside its set
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
    int wide;                //              sv / spos hold 32-bit entries (thread-per-replica builds only)
    long long samp0;         //              the call's first sample is taken before its iteration samp0 (resumed calls: see SmpState)
};

struct RrrView {             // one replica's slices of the arrays above
    uint32_t* sp; uint8_t* cls; void* sv; void* spos; int32_t* t;      // sv / spos: uint16_t entries, uint32_t when wide
    int N, Nk, M, K, wide;
    uint32_t nk_magic;       // ceil(2^32 / Nk): floor(x / Nk) = mulhi(x, nk_magic) exactly for x < 2^16 (x e < 2^32 with e = Nk nk_magic - 2^32 < Nk)
    const int32_t* A; const int8_t* J;
    const uint32_t* Jb; int Wk; double sN;
    double fourK;
    const double* Jd; double* slf; int32_t* smv; uint8_t* scur;      // GraphSKNormal slices (this replica's arrays)
    const double* Jf; double* flf; double* fundo; int32_t* fml;      // sparse Float64 slices (this replica's arrays)
};

__device__ __forceinline__ int sbit(const uint32_t* sp, int x) { return (int)((sp[x >> 5] >> (x & 31)) & 1u); }
__device__ __forceinline__ void sflip(uint32_t* sp, int x) { sp[x >> 5] ^= 1u << (x & 31); }
// set members / positions: 16-bit entries up to N = 65 535 (and in every LDS build), 32-bit beyond (the branch is kernel-uniform)
__device__ __forceinline__ int idx_get(const void* a, size_t i, int wide) { return wide ? (int)static_cast<const uint32_t*>(a)[i] : (int)static_cast<const uint16_t*>(a)[i]; }
__device__ __forceinline__ void idx_set(void* a, size_t i, int x, int wide) { if (wide) static_cast<uint32_t*>(a)[i] = (uint32_t)x; else static_cast<uint16_t*>(a)[i] = (uint16_t)x; }
// slice of spin x: x / Nk (the multiply-high form is exact for x < 2^16 only)
__device__ __forceinline__ int slice_of(int x, int Nk, uint32_t nk_magic, int wide) { return wide ? x / Nk : (int)__umulhi((uint32_t)x, nk_magic); }

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
// delta_energy of the slice graph recomputed from the slice's spins, as the integer the reference caches:
//   GraphRRG slices (RRG.jl:236-244): 2 sigma_i sum_k J_ik sigma_k;
//   binary GraphSK slices (SK.jl:62-96,137-140): lfields[i] = 2 sigma_i (Nk - 1 - 2 |{j != i : J_ij xor s_j}|) = sqrt(Nk) * delta_energy —
//   popcounts of the slice's words against row i of J (J_ii = 0, so position i contributes s_i, taken out again).
__device__ __forceinline__ int slice_delta(const RrrView& v, int move)
{
    const int k = slice_of(move, v.Nk, v.nk_magic, v.wide), i = move - k * v.Nk, off = k * v.Nk;
    const int si = sbit(v.sp, move);
    if (v.Jb) {
        const uint32_t* Ji = v.Jb + (size_t)i * v.Wk;
        int sc = 0;
        for (int w = 0; 32 * w < v.Nk; ++w) {
            const int b0 = off + 32 * w, q = b0 >> 5, sh = b0 & 31, rem = v.Nk - 32 * w;
            uint32_t bits = v.sp[q] >> sh;
            if (sh && 32 * (q + 1) < v.N) bits |= v.sp[q + 1] << (32 - sh);
            if (rem < 32) bits &= (1u << rem) - 1u;
            sc += __popc(bits ^ Ji[w]);
        }
        return 2 * (2 * si - 1) * (v.Nk - 1 - 2 * (sc - si));
    }
    int acc = 0;
    for (int q = 0; q < v.K; ++q) {
        const int y = v.A[i * v.K + q];
        const int sy = sbit(v.sp, off + y);
        acc += (si == sy) ? (int)v.J[i * v.K + q] : -(int)v.J[i * v.K + q];
    }
    return 2 * acc;
}
// delta_energy_residual (QT.jl:270-281) = delta_energy(X1[k], C1[k], i) / M; for a GraphSK slice delta_energy = lfields[i] / sN (SK.jl:139)
__device__ __forceinline__ double slice_res(const RrrView& v, int d)
{
    return v.Jb ? ((double)d / v.sN) / (double)v.M : (double)d / (double)v.M;
}
// energy(X1[k], C1[k]) / M inside energy(X::GraphQuant, C) (QT.jl:195): n an Int for GraphRRG, n / sN for GraphSK (SK.jl:95)
__device__ __forceinline__ double slice_energy_over_M(const RrrParams& P, long long n)
{
    return P.Jb ? ((double)n / P.sN) / (double)P.M : (double)n / (double)P.M;
}
// GraphSKNormal slices: delta_energy_residual = delta_energy(X1[k], C1[k], i) / M = lfields[i] / M (SK.jl:278-284, QT.jl:270-281)
__device__ __forceinline__ double skn_residual(const RrrView& v, int move)
{
    const int k = slice_of(move, v.Nk, v.nk_magic, v.wide), i = move - k * v.Nk;
    return v.slf[((size_t)v.scur[k] * v.M + k) * v.Nk + i] / (double)v.M;
}
// update_cache! of the slice graph (SK.jl:239-276), called after the bit flip of spinflip!(X::GraphQuant, C, move) (QT.jl:172-183)
__device__ inline void skn_update(const RrrView& v, int move)
{
    const int k = slice_of(move, v.Nk, v.nk_magic, v.wide), i = move - k * v.Nk, off = k * v.Nk;
    const int cur = v.scur[k];
    if (v.smv[k] == i) { v.scur[k] = (uint8_t)(cur ^ 1); return; }          // :247-250: swap the two arrays, move_last stays
    double* lf = v.slf + ((size_t)cur * v.M + k) * v.Nk;
    double* ll = v.slf + ((size_t)(cur ^ 1) * v.M + k) * v.Nk;
    const double* Ji = v.Jd + (size_t)i * v.Nk;
    const int si = sbit(v.sp, move);
    const double lfm = lf[i];
    for (int j = 0; j < v.Nk; ++j) {
        const double Jsij = (double)(1 - 2 * (si ^ sbit(v.sp, off + j))) * Ji[j];
        const double lfj = lf[j];
        ll[j] = lfj;
        lf[j] = lfj + 4 * Jsij;
    }
    ll[i] = lfm;
    lf[i] = -lfm;
    v.smv[k] = i;
}
// sparse Float64 slices: delta_energy_residual = delta_energy(X1[k], C1[k], i) / M = -lfields[i] / M (EA.jl:655-661, QT.jl:270-281)
__device__ __forceinline__ double spf_slice_residual(const RrrView& v, int move)
{
    const int k = slice_of(move, v.Nk, v.nk_magic, v.wide), i = move - k * v.Nk;
    return (-v.flf[(size_t)k * v.Nk + i]) / (double)v.M;
}
// update_cache! of a sparse Float64 slice graph (EA.jl:613-653 / RRG.jl:576-617), after the bit flip: the neighbours' fields move by
// -4 sigma_xy J (a neighbour listed twice — L = 2 — receives both bonds in order), the old values go to the undo record, and a flip of
// the slice's last moved spin takes the record back instead (the exact undo of a rejected rrrMC move)
__device__ inline void spf_slice_update(const RrrView& v, int move)
{
    const int k = slice_of(move, v.Nk, v.nk_magic, v.wide), i = move - k * v.Nk, off = k * v.Nk, K = v.K;
    double* lf = v.flf + (size_t)k * v.Nk;
    double* undo = v.fundo + (size_t)k * (K + 1);
    const int32_t* Ax = v.A + (size_t)i * K;
    const double* Jx = v.Jf + (size_t)i * K;
    const double lfm = lf[i];
    if (v.fml[k] == i) {
        for (int q = 0; q < K; ++q) {
            if (q > 0 && Ax[q] == Ax[q - 1]) continue;
            const int y = Ax[q];
            const double tmp = lf[y];
            lf[y] = undo[q];
            undo[q] = tmp;
        }
        lf[i] = -lfm;
        undo[K] = -undo[K];
        return;
    }
    const int sx = sbit(v.sp, move);
    double vv = 0.0;
    for (int q = 0; q < K; ++q) {
        const int y = Ax[q];
        if (!(q > 0 && Ax[q] == Ax[q - 1])) { vv = lf[y]; undo[q] = vv; }
        const double c = (sx ^ sbit(v.sp, off + y)) ? -4.0 : 4.0;
        vv = vv - c * Jx[q];
        if (q == K - 1 || Ax[q + 1] != y) lf[y] = vv;
    }
    undo[K] = lfm;
    lf[i] = -lfm;
    v.fml[k] = i;
}
// the residual of a move for every kind of slice
__device__ __forceinline__ double any_residual(const RrrView& v, int move)
{
    return v.Jd ? skn_residual(v, move) : v.Jf ? spf_slice_residual(v, move) : slice_res(v, slice_delta(v, move));
}
// the slice graph's update_cache! after the bit flip of spinflip!(X::GraphQuant, C, move) (QT.jl:172-183); integer slices cache nothing here
__device__ __forceinline__ void slice_cache_update(const RrrView& v, int move)
{
    if (v.Jd) skn_update(v, move);
    else if (v.Jf) spf_slice_update(v, move);
}

// ArraySet delete! / push! (ArraySets.jl:56-76); one position array serves the four sets (membership is exclusive)
__device__ __forceinline__ void set_move(const RrrView& v, int j, int k0, int k1)
{
    const size_t b0 = (size_t)k0 * v.N, b1 = (size_t)k1 * v.N;
    const int p = idx_get(v.spos, j, v.wide);
    const int last = idx_get(v.sv, b0 + v.t[k0] - 1, v.wide);
    idx_set(v.sv, b0 + p, last, v.wide);
    idx_set(v.spos, last, p, v.wide);
    v.t[k0] -= 1;
    idx_set(v.sv, b1 + v.t[k1], j, v.wide);
    idx_set(v.spos, j, v.t[k1], v.wide);
    v.t[k1] += 1;
    v.cls[j] = (uint8_t)k1;
}

__device__ __forceinline__ double class_f(int k, double ft1) { return k == 3 ? ft1 : 1.0; }   // get_class_f, ft = (1, exp(-beta fourK))

__device__ __forceinline__ RrrView rrr_view(const RrrParams& P, int r)
{
    RrrView v;
    v.sp = P.spins + (size_t)r * P.W;
    v.cls = P.cls + (size_t)r * P.N;
    v.wide = P.wide;
    v.sv = P.wide ? static_cast<void*>(static_cast<uint32_t*>(P.sv) + (size_t)r * 4 * P.N) : static_cast<void*>(static_cast<uint16_t*>(P.sv) + (size_t)r * 4 * P.N);
    v.spos = P.wide ? static_cast<void*>(static_cast<uint32_t*>(P.spos) + (size_t)r * P.N) : static_cast<void*>(static_cast<uint16_t*>(P.spos) + (size_t)r * P.N);
    v.t = P.st + (size_t)r * 4;
    v.N = P.N; v.Nk = P.Nk; v.M = P.M; v.K = P.K; v.A = P.A; v.J = P.J; v.fourK = P.fourK;
    v.Jb = P.Jb; v.Wk = P.Wk; v.sN = P.sN;
    v.Jd = P.Jd;
    v.slf = P.Jd ? P.slf + (size_t)r * 2 * P.M * P.Nk : nullptr;
    v.smv = P.Jd ? P.smv + (size_t)r * P.M : nullptr;
    v.scur = P.Jd ? P.scur + (size_t)r * P.M : nullptr;
    v.Jf = P.Jf;
    v.flf = P.Jf ? P.flf + (size_t)r * P.M * P.Nk : nullptr;
    v.fundo = P.Jf ? P.fundo + (size_t)r * P.M * (P.K + 1) : nullptr;
    v.fml = P.Jf ? P.fml + (size_t)r * P.M : nullptr;
    v.nk_magic = (uint32_t)((0x100000000ull + (uint32_t)P.Nk - 1u) / (uint32_t)P.Nk);
    return v;
}

// energy(X::GraphQuant, C) (QT.jl:185-199) + DeltaECache construction (DeltaE.jl:74-103), one thread per replica
__global__ __launch_bounds__(kRrrThreads) void rrr_init_kernel(RrrParams P)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
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
        E += slice_energy_over_M(P, n);
    }
    P.E_cur[r] = E;
    // cache: classes and sets in site order
    for (int k = 0; k < 4; ++k) v.t[k] = 0;
    for (int i = 0; i < P.N; ++i) {
        const int k = qt_class(v, i);
        v.cls[i] = (uint8_t)k;
        idx_set(v.sv, (size_t)k * P.N + v.t[k], i, v.wide);
        idx_set(v.spos, i, v.t[k], v.wide);
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

// The same construction — energy(X::GraphQuant, C) and the DeltaECache in site order — by one WORKGROUP per replica: every thread
// classifies a contiguous block of spins, an exclusive scan of the per-class counts over the threads gives each block its
// offsets inside the four member arrays (site order inside a class is what push! in site order produces: DeltaE.jl:79-88), the
// integer parts of the energy are reduced with atomics (exact) and thread 0 combines them in the reference's Float64 order.
// A sequential pass over N = 32 768 spins per replica at every call costs as much as thousands of iterations.
constexpr int kInitThreads = 256;
__global__ __launch_bounds__(kInitThreads) void rrr_init_coop_kernel(RrrParams P)
{
    __shared__ int s_cnt[4][kInitThreads];
    __shared__ int s_tot[4];
    __shared__ long long s_n0;
    extern __shared__ long long s_slice[];             // [M] integer slice energies
    const int r = (int)blockIdx.x, tid = (int)threadIdx.x;
    const RrrView v = rrr_view(P, r);
    const int N = P.N;
    if (tid == 0) s_n0 = 0;
    for (int k = tid; k < P.M; k += kInitThreads) s_slice[k] = 0;
    __syncthreads();
    // energy, integer parts: n0 = -sum over Trotter bonds of sigma sigma' (QT.jl:68-82); slice k: sum_x lf_x with lf_x = -(2 sx sum J sy)/2
    {
        long long n0 = 0;
        for (int x = tid; x < N; x += kInitThreads) {
            const int i = x % P.Nk, k = x / P.Nk;
            const int prev = i + (k == 0 ? P.M - 1 : k - 1) * P.Nk;
            n0 -= 1 - 2 * (sbit(v.sp, x) ^ sbit(v.sp, prev));
            atomicAdd(reinterpret_cast<unsigned long long*>(&s_slice[k]), (unsigned long long)(long long)(-(slice_delta(v, x) / 2)));
        }
        atomicAdd(reinterpret_cast<unsigned long long*>(&s_n0), (unsigned long long)n0);
    }
    // classes of this thread's block of spins
    const int per = (N + kInitThreads - 1) / kInitThreads, x0 = tid * per, x1 = x0 + per < N ? x0 + per : N;
    int cnt[4] = {0, 0, 0, 0};
    for (int x = x0; x < x1; ++x) {
        const int k = qt_class(v, x);
        v.cls[x] = (uint8_t)k;
        cnt[k] += 1;
    }
    for (int k = 0; k < 4; ++k) s_cnt[k][tid] = cnt[k];
    __syncthreads();
    if (tid < 4) {                                     // exclusive scan over the threads, one class per lane (256 additions)
        int run = 0;
        for (int t = 0; t < kInitThreads; ++t) { const int c = s_cnt[tid][t]; s_cnt[tid][t] = run; run += c; }
        s_tot[tid] = run;
    }
    __syncthreads();
    int off[4];
    for (int k = 0; k < 4; ++k) off[k] = s_cnt[k][tid];
    for (int x = x0; x < x1; ++x) {
        const int k = v.cls[x];
        idx_set(v.sv, (size_t)k * N + off[k], x, v.wide);
        idx_set(v.spos, x, off[k], v.wide);
        off[k] += 1;
    }
    if (tid == 0) {
        double E = (double)s_n0 * P.fourK / 4;
        for (int k = 0; k < P.M; ++k) {
            long long n = s_slice[k];
            n /= 2;
            E += slice_energy_over_M(P, n);
        }
        P.E_cur[r] = E;
        double z = 0.0;
        for (int k = 0; k < 4; ++k) {
            v.t[k] = s_tot[k];
            const double x = (double)s_tot[k] * class_f(k, P.ft1);
            z += x;
            P.T[(size_t)r * 4 + k] = x;
        }
        P.zz[r] = z;
        P.acc_rate[r] = 0.5;
        P.stats[(size_t)r * 2] = 0;
        P.stats[(size_t)r * 2 + 1] = 0;
    }
}

// energy(X::GraphQuant, C) (QT.jl:185-199) and the DeltaECache for GraphSKNormal slices, one workgroup per replica: the slice caches are
// rebuilt as skn energy does (SK.jl:212-237: lfields[i] = 2 lf_i with lf_i the sequential sum over j, n -= lf_i in site order, n /= 2;
// lfields_last = 0, move_last = none), the classes as rrr_init_coop_kernel.
__global__ __launch_bounds__(kInitThreads) void rrr_init_skn_kernel(RrrParams P)
{
    __shared__ int s_cnt[4][kInitThreads];
    __shared__ int s_tot[4];
    __shared__ long long s_n0;
    extern __shared__ long long s_slice[];             // [M] slice energies (as doubles)
    double* s_E = reinterpret_cast<double*>(s_slice);
    const int r = (int)blockIdx.x, tid = (int)threadIdx.x;
    const RrrView v = rrr_view(P, r);
    const int N = P.N, Nk = P.Nk, M = P.M;
    if (tid == 0) s_n0 = 0;
    __syncthreads();
    {
        long long n0 = 0;
        for (int x = tid; x < N; x += kInitThreads) {
            const int i = x % Nk, k = x / Nk;
            const int prev = i + (k == 0 ? M - 1 : k - 1) * Nk;
            n0 -= 1 - 2 * (sbit(v.sp, x) ^ sbit(v.sp, prev));
            // lf_i of slice k: sequential in j (SK.jl:218-226); array 0 becomes lfields, array 1 holds lf_i until the slice sums are done
            const double* Ji = P.Jd + (size_t)i * Nk;
            const int si = sbit(v.sp, x);
            double lf = 0.0;
            for (int j = 0; j < Nk; ++j) lf += (double)(1 - 2 * (si ^ sbit(v.sp, k * Nk + j))) * Ji[j];
            v.slf[((size_t)0 * M + k) * Nk + i] = 2 * lf;
            v.slf[((size_t)1 * M + k) * Nk + i] = lf;
        }
        atomicAdd(reinterpret_cast<unsigned long long*>(&s_n0), (unsigned long long)n0);
    }
    __syncthreads();
    for (int k = tid; k < M; k += kInitThreads) {      // n -= lf in site order, n /= 2 (SK.jl:227-232)
        double n = 0.0;
        for (int i = 0; i < Nk; ++i) n -= v.slf[((size_t)1 * M + k) * Nk + i];
        n /= 2;
        s_E[k] = n;
        v.smv[k] = -1;
        v.scur[k] = 0;
    }
    __syncthreads();
    for (int x = tid; x < N; x += kInitThreads) v.slf[(size_t)M * Nk + x] = 0.0;      // lfields_last = 0
    // classes of this thread's block of spins (as rrr_init_coop_kernel)
    const int per = (N + kInitThreads - 1) / kInitThreads, x0 = tid * per, x1 = x0 + per < N ? x0 + per : N;
    int cnt[4] = {0, 0, 0, 0};
    for (int x = x0; x < x1; ++x) {
        const int k = qt_class(v, x);
        v.cls[x] = (uint8_t)k;
        cnt[k] += 1;
    }
    for (int k = 0; k < 4; ++k) s_cnt[k][tid] = cnt[k];
    __syncthreads();
    if (tid < 4) {
        int run = 0;
        for (int t = 0; t < kInitThreads; ++t) { const int c = s_cnt[tid][t]; s_cnt[tid][t] = run; run += c; }
        s_tot[tid] = run;
    }
    __syncthreads();
    int off[4];
    for (int k = 0; k < 4; ++k) off[k] = s_cnt[k][tid];
    for (int x = x0; x < x1; ++x) {
        const int k = v.cls[x];
        idx_set(v.sv, (size_t)k * N + off[k], x, v.wide);
        idx_set(v.spos, x, off[k], v.wide);
        off[k] += 1;
    }
    if (tid == 0) {
        double E = (double)s_n0 * P.fourK / 4;
        for (int k = 0; k < M; ++k) E += s_E[k] / (double)M;
        P.E_cur[r] = E;
        double z = 0.0;
        for (int k = 0; k < 4; ++k) {
            v.t[k] = s_tot[k];
            const double x = (double)s_tot[k] * class_f(k, P.ft1);
            z += x;
            P.T[(size_t)r * 4 + k] = x;
        }
        P.zz[r] = z;
        P.acc_rate[r] = 0.5;
        P.stats[(size_t)r * 2] = 0;
        P.stats[(size_t)r * 2 + 1] = 0;
    }
}

// energy(X::GraphQuant, C) (QT.jl:185-199) and the DeltaECache for sparse Float64 slices (GraphQEAT), one workgroup per replica: every slice's
// cache is rebuilt as its graph's energy does (EA.jl:584-611 / RRG.jl:546-574: lf_x = -sum_q J sx sy in table order, E1 += lf_x in site order,
// lfields[x] = 2 lf_x, E1 / 2; move_last = none), one thread per slice (the sums are sequential by construction); the classes as rrr_init_coop_kernel.
__global__ __launch_bounds__(kInitThreads) void rrr_init_spf_kernel(RrrParams P)
{
    __shared__ int s_cnt[4][kInitThreads];
    __shared__ int s_tot[4];
    __shared__ long long s_n0;
    extern __shared__ long long s_slice[];             // [M] slice energies (as doubles)
    double* s_E = reinterpret_cast<double*>(s_slice);
    const int r = (int)blockIdx.x, tid = (int)threadIdx.x;
    const RrrView v = rrr_view(P, r);
    const int N = P.N, Nk = P.Nk, M = P.M, K = P.K;
    if (tid == 0) s_n0 = 0;
    __syncthreads();
    {
        long long n0 = 0;
        for (int x = tid; x < N; x += kInitThreads) {
            const int i = x % Nk, k = x / Nk;
            const int prev = i + (k == 0 ? M - 1 : k - 1) * Nk;
            n0 -= 1 - 2 * (sbit(v.sp, x) ^ sbit(v.sp, prev));
        }
        atomicAdd(reinterpret_cast<unsigned long long*>(&s_n0), (unsigned long long)n0);
    }
    for (int k = tid; k < M; k += kInitThreads) {
        double* lf = v.flf + (size_t)k * Nk;
        double E1 = 0.0;
        for (int x = 0; x < Nk; ++x) {
            const int sx = 2 * sbit(v.sp, k * Nk + x) - 1;
            double fl = 0.0;
            for (int q = 0; q < K; ++q) {
                const int sy = 2 * sbit(v.sp, k * Nk + P.A[(size_t)x * K + q]) - 1;
                fl = fl - P.Jf[(size_t)x * K + q] * (double)sx * (double)sy;
            }
            E1 = E1 + fl;
            lf[x] = 2.0 * fl;
        }
        s_E[k] = E1 / 2;
        v.fml[k] = -1;
        for (int q = 0; q <= K; ++q) v.fundo[(size_t)k * (K + 1) + q] = 0.0;
    }
    __syncthreads();
    const int per = (N + kInitThreads - 1) / kInitThreads, x0 = tid * per, x1 = x0 + per < N ? x0 + per : N;
    int cnt[4] = {0, 0, 0, 0};
    for (int x = x0; x < x1; ++x) {
        const int k = qt_class(v, x);
        v.cls[x] = (uint8_t)k;
        cnt[k] += 1;
    }
    for (int k = 0; k < 4; ++k) s_cnt[k][tid] = cnt[k];
    __syncthreads();
    if (tid < 4) {
        int run = 0;
        for (int t = 0; t < kInitThreads; ++t) { const int c = s_cnt[tid][t]; s_cnt[tid][t] = run; run += c; }
        s_tot[tid] = run;
    }
    __syncthreads();
    int off[4];
    for (int k = 0; k < 4; ++k) off[k] = s_cnt[k][tid];
    for (int x = x0; x < x1; ++x) {
        const int k = v.cls[x];
        idx_set(v.sv, (size_t)k * N + off[k], x, v.wide);
        idx_set(v.spos, x, off[k], v.wide);
        off[k] += 1;
    }
    if (tid == 0) {
        double E = (double)s_n0 * P.fourK / 4;
        for (int k = 0; k < M; ++k) E += s_E[k] / (double)M;
        P.E_cur[r] = E;
        double z = 0.0;
        for (int k = 0; k < 4; ++k) {
            v.t[k] = s_tot[k];
            const double x = (double)s_tot[k] * class_f(k, P.ft1);
            z += x;
            P.T[(size_t)r * 4 + k] = x;
        }
        P.zz[r] = z;
        P.acc_rate[r] = 0.5;
        P.stats[(size_t)r * 2] = 0;
        P.stats[(size_t)r * 2 + 1] = 0;
    }
}

// bytes of LDS one replica's hot state takes in the LDS-resident build below
inline size_t rrr_quant_lds_bytes(int64_t N, int64_t W, int64_t Nk, int64_t K)
{
    return (size_t)W * 4 + (((size_t)N * 2 + 3) & ~(size_t)3) + (((size_t)N + 3) & ~(size_t)3) + 16 + (size_t)Nk * K * 4 + (((size_t)Nk * K + 3) & ~(size_t)3) + (size_t)kRrrThreads * 8 * 4;
}

// LDS = false: one thread per replica, everything in HBM/L2 (any number of replicas per workgroup).
// LDS = true:  one workgroup (one wavefront) per replica — the launch shape of a few hundred replicas anyway — with the replica's
//   spins, class bytes, set positions and set sizes, and the slice graph's (A, J), staged in LDS (3.1 bytes per spin + 5 K bytes
//   per slice site: 115 KB at config 5) by all 64 lanes; lane 0 then runs the chain.  Only the ArraySet member arrays (sv, 8 bytes
//   per spin) are left in HBM/L2.  Same arithmetic, same order: results are unchanged.  (Worth 10 % at config 5: a chain on one
//   lane is bound by its instruction stream — 5 cycles per instruction — more than by the loads.)
template <bool LDS>
__global__ __launch_bounds__(kRrrThreads) void rrr_quant_kernel(RrrParams P)
{
    extern __shared__ uint32_t q_lds[];
    int r;
    if constexpr (LDS) {
        r = (int)blockIdx.x;
    } else {
        r = blockIdx.x * blockDim.x + threadIdx.x;
        if (r >= P.R) return;
    }
    RrrView v = rrr_view(P, r);
    uint32_t* l_rng = nullptr;
    uint32_t* g_sp = v.sp; uint8_t* g_cls = v.cls; uint16_t* g_spos = static_cast<uint16_t*>(v.spos); int32_t* g_t = v.t;      // (the LDS build is 16-bit only)
    if constexpr (LDS) {
        const int tid = (int)threadIdx.x, nt = (int)blockDim.x;
        uint32_t* l_sp = q_lds;                                               // [W]
        uint16_t* l_spos = reinterpret_cast<uint16_t*>(l_sp + P.W);           // [N]
        uint8_t* l_cls = reinterpret_cast<uint8_t*>(l_spos) + ((2 * P.N + 3) & ~3);   // [N], both padded to a word
        int32_t* l_t = reinterpret_cast<int32_t*>(l_cls + ((P.N + 3) & ~3));  // [4]
        int32_t* l_A = l_t + 4;                                               // [Nk][K]
        int8_t* l_J = reinterpret_cast<int8_t*>(l_A + P.Nk * P.K);            // [Nk][K]
        l_rng = reinterpret_cast<uint32_t*>(l_J + ((P.Nk * P.K + 3) & ~3));   // [64][8]  the RRR draws of the next 64 iterations
        for (int i = tid; i < P.W; i += nt) l_sp[i] = g_sp[i];
        for (int i = tid; i < P.N; i += nt) { l_spos[i] = g_spos[i]; l_cls[i] = g_cls[i]; }
        for (int i = tid; i < P.Nk * P.K; i += nt) { l_A[i] = P.A[i]; l_J[i] = P.J[i]; }
        if (tid < 4) l_t[tid] = g_t[tid];
        __syncthreads();
        v.sp = l_sp; v.spos = l_spos; v.cls = l_cls; v.t = l_t; v.A = l_A; v.J = l_J;
    }
    const bool worker = !LDS || threadIdx.x == 0;
    const uint32_t rep = P.replica0 + (uint32_t)r;
    double T[4], z = P.zz[r], E = P.E_cur[r], acc_rate = P.acc_rate[r];
    for (int k = 0; k < 4; ++k) T[k] = P.T[(size_t)r * 4 + k];
    int64_t accepted = P.stats[(size_t)r * 2], staged_its = P.stats[(size_t)r * 2 + 1];
    int64_t ns = 0;
    const double dEl[2] = {0.0, P.fourK};

    long long next_sample = P.samp0;         // iterations k * step of the run: a counter instead of a 64-bit modulo per iteration
    // LDS build: the chain runs on lane 0 and is bound by its instruction stream, so the state-independent part of an iteration —
    // the two Philox blocks of the RRR stream — is computed for 64 iterations at a time by the whole wavefront (one iteration per lane)
    for (int64_t base = 0; base < P.iters; base += (LDS ? kRrrThreads : P.iters)) {
    if constexpr (LDS) {
        __syncthreads();
        const uint64_t gl = P.g0 + (uint64_t)(base + 1 + (int64_t)threadIdx.x);
        const Philox4 a = philox4x32_10((uint32_t)gl, (uint32_t)(gl >> 32), rep, TAG_RRR, P.k0, P.k1);
        const Philox4 b = philox4x32_10((uint32_t)gl, (uint32_t)(gl >> 32), rep, TAG_RRR | (1u << 8), P.k0, P.k1);
        uint32_t* q = l_rng + threadIdx.x * 8;
        q[0] = a.w[0]; q[1] = a.w[1]; q[2] = a.w[2]; q[3] = a.w[3]; q[4] = b.w[0]; q[5] = b.w[1]; q[6] = b.w[2]; q[7] = b.w[3];
        __syncthreads();
    }
    const int64_t it_end = LDS ? (base + kRrrThreads < P.iters ? base + kRrrThreads : P.iters) : P.iters;
    if (worker)
    for (int64_t it = base + 1; it <= it_end; ++it) {
        if (it == next_sample) { next_sample += P.step; P.Es[ns * P.R + r] = E; ns += 1; }
        const uint64_t g = P.g0 + (uint64_t)it;
        // rand_move: DeltaE.jl:146-167
        Philox4 o;
        if constexpr (LDS) { const uint32_t* q = l_rng + (it - base - 1) * 8; o.w[0] = q[0]; o.w[1] = q[1]; o.w[2] = q[2]; o.w[3] = q[3]; }
        else o = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), rep, TAG_RRR, P.k0, P.k1);
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
        const int move = idx_get(v.sv, (size_t)k * P.N + (size_t)mulhi64(u, (uint64_t)v.t[k]), v.wide);

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
            const double dE1 = any_residual(v, move);                      // delta_energy_residual, QT.jl:270-281
            const double x = -P.beta * dE1;
            bool ok = (c >= 1 && x >= 0);
            if (!ok) {                                                                 // accept(c, x), RRRMC.jl:40-44
                const double a = c * det_exp(x);
                ok = a >= 1;
                if (!ok) {
                    Philox4 o2;
                    if constexpr (LDS) { const uint32_t* q = l_rng + (it - base - 1) * 8 + 4; o2.w[0] = q[0]; o2.w[1] = q[1]; }
                    else o2 = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), rep, TAG_RRR | (1u << 8), P.k0, P.k1);
                    ok = (double)((((uint64_t)o2.w[0] << 32) | o2.w[1]) >> 11) * 0x1.0p-53 < a;
                }
            }
            if (ok) {
                sflip(v.sp, move);                                                     // spinflip!(X, C, move)
                slice_cache_update(v, move);
                for (int q = 0; q < nst; ++q) set_move(v, sj[q], s0[q], s1[q]);        // apply_staged!
                for (int q = 0; q < 4; ++q) T[q] = Tp[q];
                z = zp;
                E += dE0 + dE1;
                accepted += 1;
                acc = true;
            }
        } else {
            // direct branch: apply_move! (DeltaE.jl:232-295), undone by a second apply_move! on rejection
            const double dE1 = any_residual(v, move);
            double c = 0.0;
            for (int pass = 0; pass < 2; ++pass) {
                sflip(v.sp, move);
                slice_cache_update(v, move);                                          // the undo pass takes the swap path (move_last == move)
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
                        Philox4 o2;
                        if constexpr (LDS) { const uint32_t* q = l_rng + (it - base - 1) * 8 + 4; o2.w[0] = q[0]; o2.w[1] = q[1]; }
                        else o2 = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), rep, TAG_RRR | (1u << 8), P.k0, P.k1);
                        ok = (double)((((uint64_t)o2.w[0] << 32) | o2.w[1]) >> 11) * 0x1.0p-53 < a;
                    }
                }
                if (ok) { E += dE0 + dE1; accepted += 1; acc = true; break; }
            }
        }
        acc_rate = acc_rate * (1 - P.lambda) + (acc ? 1.0 : 0.0) * P.lambda;             // RRRMC.jl:281
    }
    }
    if (worker) {
    for (int k = 0; k < 4; ++k) P.T[(size_t)r * 4 + k] = T[k];
    P.zz[r] = z; P.E_cur[r] = E; P.acc_rate[r] = acc_rate;
    P.stats[(size_t)r * 2] = accepted; P.stats[(size_t)r * 2 + 1] = staged_its;
    }
    if constexpr (LDS) {
        __syncthreads();
        const int tid = (int)threadIdx.x, nt = (int)blockDim.x;
        for (int i = tid; i < P.W; i += nt) g_sp[i] = v.sp[i];
        for (int i = tid; i < P.N; i += nt) { g_spos[i] = static_cast<uint16_t*>(v.spos)[i]; g_cls[i] = v.cls[i]; }
        if (tid < 4) g_t[tid] = v.t[tid];
    }
}

// standardMC (src/RRRMC.jl:81-127) on GraphQuant: delta_energy(X, C, move) = delta_energy(X0) + delta_energy_residual (QT.jl:283-286);
// common site (SITE stream), rand53 < exp(-beta dE) (ACCEPT_F64 stream).  Runs after rrr_init_kernel (which leaves energy(X, C) in E_cur).
__global__ __launch_bounds__(kRrrThreads) void quant_standard_kernel(RrrParams P)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= P.R) return;
    const RrrView v = rrr_view(P, r);
    const uint32_t rep = P.replica0 + (uint32_t)r;
    double E = P.E_cur[r];
    int64_t accepted = 0, ns = 0;
    long long next_sample = P.step;          // iterations k * step: a counter instead of a 64-bit modulo per iteration
    for (int64_t it = 1; it <= P.iters; ++it) {
        if (it == next_sample) { next_sample += P.step; P.Es[ns * P.R + r] = E; ns += 1; }
        const uint64_t g = P.g0 + (uint64_t)it;
        const int move = (int)site_of(P.k0, P.k1, g, (uint32_t)P.N);
        const double dE = (double)qt_delta(v, move) * P.fourK + any_residual(v, move);
        const double x = -P.beta * dE;
        const bool acc = (x >= 0.0) || (rand53(P.k0, P.k1, g, rep) < det_exp(x));        // RRRMC.jl:39
        if (acc) { sflip(v.sp, move); slice_cache_update(v, move); E += dE; accepted += 1; }
    }
    P.E_cur[r] = E;
    P.stats[(size_t)r * 2] = accepted; P.stats[(size_t)r * 2 + 1] = 0;
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

// ---------------------------------------------------------------------------------------------------
// rrrMC(X::SingleGraph) (src/RRRMC.jl:149-219) on GraphSKNormal with the continuous-energy cache DeltaECacheCont
// (src/DeltaE.jl:297-410) over a DynamicSampler (src/DynamicSamplers.jl:18-176, Wong-Easton tree of partial sums).
// Every iteration touches all N fields and re-weights all N leaves (every spin is a neighbour of every other), and the
// reference's running sums are order dependent, so each replica is one THREAD that walks j = 0..N-1 in the reference's
// order; arrays are replica-interleaved ([index][replica]) so that the lock-step walks of a wave coalesce.
// ---------------------------------------------------------------------------------------------------
struct RrrSkParams {
    const double* J;         // [N][N]
    double* lfA;             // [N][Rp]  one of the two field arrays (lfields / lfields_last swap by index, SK.jl:247-250)
    double* lfB;             // [N][Rp]
    double* v;               // [N2][Rp]  sampler weights
    double* ps;              // [N2-1][Rp] partial sums (level by level)
    double* dEs;             // [N][Rp]
    double* st_dE;           // [N][Rp]  staged values
    double* st_p;            // [N][Rp]
    uint32_t* spins;         // [W][Rp]
    double* E_cur;           // [Rp]
    int64_t* stats;          // [Rp][2]
    int32_t* status;         // [Rp]   1 = the sampler's "unrecoverable loss of precision" error
    double* Es;              // [nsamples][Rp]
    double* z_out;           // [Rp]
    double beta, staged_thr, lambda;
    uint64_t g0;
    int64_t iters, step;
    uint32_t k0, k1, replica0;
    int N, N2, levs, W, R, Rp;
    int mode;                // 0 = rrrMC(SingleGraph), 1 = bklMC, 2 = wtmMC (iters = samples)
    uint32_t call;           // wtmMC: number of the call (WTM stream)
    double stepf;            // wtmMC: step in sweeps
    double* t_out;           // wtmMC: [Rp] final global time
    const double* ftau;      // extremal_opt: [N] cumsum(j^-tau)
    uint32_t* cmin;          // extremal_opt: [R][W] configuration of minimum energy (replica-contiguous words)
    double sN;               // binary GraphSK (SK.jl:28-165) run as +-1 couplings: delta_energy = lfields[i] / sN, E = n / sN (sN = sqrt(N)); 0 = GraphSKNormal
    SmpState S;
    int64_t samples_before;  // wtmMC: samples the run had taken before this (resumed) call
};

struct SkChain {             // one replica's view
    const RrrSkParams* P;
    int r;
    int cur;                 // which array currently is `lfields`
    int move_last;
    double z;
    long long trefresh;
    __device__ __forceinline__ double* lf() const { return (cur ? P->lfB : P->lfA) + r; }
    // delta_energy(X, C, i) from the cached field: +lfields[i] for GraphSKNormal (SK.jl:278-284), lfields[i] / sN for the binary GraphSK (:137-140)
    __device__ __forceinline__ double dEv(double lfv) const { return P->sN > 0.0 ? lfv / P->sN : lfv; }
    __device__ __forceinline__ double* lfl() const { return (cur ? P->lfA : P->lfB) + r; }
    __device__ __forceinline__ int sbit(int x) const { return (int)((P->spins[(size_t)(x >> 5) * P->Rp + r] >> (x & 31)) & 1u); }
    __device__ __forceinline__ void sflip(int x) { P->spins[(size_t)(x >> 5) * P->Rp + r] ^= 1u << (x & 31); }

    // spinflip!(X, C, move) = bit flip + update_cache! (SK.jl:239-276)
    __device__ void flip(int move)
    {
        sflip(move);
        if (move_last == move) { cur ^= 1; return; }
        const int Rp = P->Rp, N = P->N;
        double* a = lf();
        double* b = lfl();
        const double* Ji = P->J + (size_t)move * N;
        const int si = sbit(move);
        const double lfm = a[(size_t)move * Rp];
        for (int j = 0; j < N; ++j) {
            const double Js = (si ^ sbit(j)) ? -Ji[j] : Ji[j];
            const double lfj = a[(size_t)j * Rp];
            b[(size_t)j * Rp] = lfj;
            a[(size_t)j * Rp] = lfj + 4 * Js;
        }
        b[(size_t)move * Rp] = lfm;
        a[(size_t)move * Rp] = -lfm;
        move_last = move;
    }
    // refresh!: DynamicSamplers.jl:84-98
    __device__ void refresh()
    {
        const int Rp = P->Rp;
        double* v = P->v + r;
        double* ps = P->ps + r;
        double zz = 0.0;
        for (int i = 0; i < P->N2; ++i) zz += v[(size_t)i * Rp];
        z = zz;
        for (int k = 0; k < P->N2 - 1; ++k) ps[(size_t)k * Rp] = 0.0;
        for (int i = 0; i < P->N; ++i) {
            const double vi = v[(size_t)i * Rp];
            int k = 0, u = 1 << (P->levs - 1), off = 1;
            for (int lev = 0; lev < P->levs; ++lev) {
                if ((i & u) == 0) { ps[(size_t)(off - 1 + k) * Rp] += vi; k *= 2; }
                else k = 2 * k + 1;
                u >>= 1; off *= 2;
            }
        }
        trefresh = 0;
    }
    // setindex!: DynamicSamplers.jl:159-176
    __device__ void set(int i, double x)
    {
        const int Rp = P->Rp;
        if (trefresh >= (P->N > 100 ? P->N : 100)) refresh();
        trefresh += 1;
        double* v = P->v + r;
        double* ps = P->ps + r;
        const double d = x - v[(size_t)i * Rp];
        v[(size_t)i * Rp] = x;
        z += d;
        int k = 0, u = 1 << (P->levs - 1), off = 1;
        for (int lev = 0; lev < P->levs; ++lev) {
            if ((i & u) == 0) { ps[(size_t)(off - 1 + k) * Rp] += d; k *= 2; }
            else k = 2 * k + 1;
            u >>= 1; off *= 2;
        }
    }
    // getel: DynamicSamplers.jl:130-152; -1 = unrecoverable loss of precision
    __device__ int getel(double x)
    {
        const int Rp = P->Rp;
        for (;;) {
            x *= z;
            int k = 0, off = 1;
            for (int lev = 0; lev < P->levs; ++lev) {
                const double p = P->ps[(size_t)(off - 1 + k) * Rp + r];
                k *= 2;
                if (x > p) { x -= p; k += 1; }
                off *= 2;
            }
            if (k >= P->N || P->v[(size_t)k * Rp + r] == 0) {
                if (!(trefresh > 0)) return -1;
                refresh();
                continue;
            }
            return k;
        }
    }
};

__device__ __forceinline__ double prior_of(double x) { return x > 0 ? det_exp(-x) : 1.0; }      // DeltaE.jl:297

// energy(X, C) (SK.jl:212-237) + DeltaECacheCont construction (DeltaE.jl:304-313) + the sampling loop
__global__ __launch_bounds__(kRrrThreads) void rrr_skn_kernel(RrrSkParams P)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= P.R) return;
    const int N = P.N, Rp = P.Rp;
    SkChain c;
    c.P = &P; c.r = r; c.cur = 0; c.move_last = -1; c.z = 0.0; c.trefresh = 0;
    double* const sf = P.S.sf + (size_t)r * kSmpF;
    long long* const sq = P.S.si + (size_t)r * kSmpI;
    const bool resume = P.S.resume != 0;
    double E;
    if (resume) {
        // a resumed call: fields (and which array is `lfields`), move_last, the sampler's tree with its refresh countdown and z — or the
        // heap of waiting times — and the tracked energy are where the previous call left them
        c.cur = (int)sq[SI_CUR]; c.move_last = (int)sq[SI_MLAST]; c.z = sf[SF_Z]; c.trefresh = sq[SI_TREF];
        E = sf[SF_E];
    } else {
    // energy: sequential sums in the reference's order
    double n = 0.0;
    for (int i = 0; i < N; ++i) {
        const double* Ji = P.J + (size_t)i * N;
        const int si = c.sbit(i);
        double lfh = 0.0;
        for (int j = 0; j < N; ++j) lfh += (si ^ c.sbit(j)) ? -Ji[j] : Ji[j];
        P.lfA[(size_t)i * Rp + r] = 2 * lfh;
        P.lfB[(size_t)i * Rp + r] = 0.0;
        n -= lfh;
    }
    E = n / 2;
    if (P.sN > 0.0) E = E / P.sN;               // GraphSK: the integer n of SK.jl:62-95 (exact in Float64), then n / sN
    for (int i = 0; i < P.N2; ++i) P.v[(size_t)i * Rp + r] = 0.0;
    for (int i = 0; i < N; ++i) {
        const double dE = c.dEv(P.lfA[(size_t)i * Rp + r]);
        P.dEs[(size_t)i * Rp + r] = dE;
        P.v[(size_t)i * Rp + r] = prior_of(P.beta * dE);
    }
    c.refresh();
    }

    const uint32_t rep = P.replica0 + (uint32_t)r;
    double acc_rate = resume ? sf[SF_ACC] : 0.5;
    long long accepted = 0, staged_its = 0, ns = 0;
    int bad = 0;
    if (P.mode == 2) {
        // wtmMC (RRRMC.jl:376-426, WaitingTimes.jl): every spin carries the time of its next flip, the smallest one moves; a move
        // re-draws the waiting times of the moved spin and of all the others (AllButOne, SK.jl:297) from tau = max(1, exp(beta dE)).
        // Binary min-heap by (time, spin) per replica in the (interleaved) arrays of the unused sampler: keys in dEs, spin at
        // position / position of spin as 16-bit entries in st_dE / st_p.  WTM stream: draw n of the call for the n-th waiting time.
        double* ht = P.dEs + r;
        uint16_t* hid = reinterpret_cast<uint16_t*>(P.st_dE) + r;
        uint16_t* hpos = reinterpret_cast<uint16_t*>(P.st_p) + r;
        uint64_t nd = 0;
        auto uniform = [&]() {
            const uint64_t n = nd++, blk = n >> 1;
            const Philox4 o = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), rep, 11u | (P.call << 8), P.k0, P.k1);       // TAG_WTM
            const uint64_t u = (n & 1u) ? (((uint64_t)o.w[2] << 32) | o.w[3]) : (((uint64_t)o.w[0] << 32) | o.w[1]);
            return (double)(u >> 11) * 0x1.0p-53;
        };
        auto gen_wt = [&](double dE) { const double e = det_exp(P.beta * dE); return -(e > 1.0 ? e : 1.0) * det_log1p(-uniform()); };
        auto before = [](double ta, int a, double tb, int b) { return ta < tb || (ta == tb && a < b); };
        auto sift_down = [&](int pos) {
            const double t = ht[(size_t)pos * Rp];
            const int id = hid[(size_t)pos * Rp];
            for (;;) {
                int ch = 2 * pos + 1;
                if (ch >= N) break;
                if (ch + 1 < N && before(ht[(size_t)(ch + 1) * Rp], hid[(size_t)(ch + 1) * Rp], ht[(size_t)ch * Rp], hid[(size_t)ch * Rp])) ch += 1;
                if (!before(ht[(size_t)ch * Rp], hid[(size_t)ch * Rp], t, id)) break;
                ht[(size_t)pos * Rp] = ht[(size_t)ch * Rp];
                const int cid = hid[(size_t)ch * Rp];
                hid[(size_t)pos * Rp] = (uint16_t)cid; hpos[(size_t)cid * Rp] = (uint16_t)pos;
                pos = ch;
            }
            ht[(size_t)pos * Rp] = t; hid[(size_t)pos * Rp] = (uint16_t)id; hpos[(size_t)id * Rp] = (uint16_t)pos;
        };
        auto sift_up = [&](int pos) {
            const double t = ht[(size_t)pos * Rp];
            const int id = hid[(size_t)pos * Rp];
            while (pos > 0) {
                const int par = (pos - 1) >> 1;
                if (!before(t, id, ht[(size_t)par * Rp], hid[(size_t)par * Rp])) break;
                ht[(size_t)pos * Rp] = ht[(size_t)par * Rp];
                const int pid = hid[(size_t)par * Rp];
                hid[(size_t)pos * Rp] = (uint16_t)pid; hpos[(size_t)pid * Rp] = (uint16_t)pos;
                pos = par;
            }
            ht[(size_t)pos * Rp] = t; hid[(size_t)pos * Rp] = (uint16_t)id; hpos[(size_t)id * Rp] = (uint16_t)pos;
        };
        auto update = [&](int i, double t) {
            const int pos = hpos[(size_t)i * Rp];
            const double old = ht[(size_t)pos * Rp];
            ht[(size_t)pos * Rp] = t;
            if (t < old) sift_up(pos); else sift_down(pos);
        };
        const double st = P.stepf / (double)N, tmax = st * (double)(P.samples_before + P.iters);
        double t = 0.0, nextstep = st;
        if (resume) { t = sf[SF_TIME]; nextstep = sf[SF_NEXT]; nd = (uint64_t)sq[SI_ND]; }      // the run's heap, global time and next sample time
        else {
        for (int i = 0; i < N; ++i) {                 // THeap(X, C, beta): one waiting time per spin, in index order
            ht[(size_t)i * Rp] = gen_wt(c.dEv(P.lfA[(size_t)i * Rp + r]));
            hid[(size_t)i * Rp] = (uint16_t)i; hpos[(size_t)i * Rp] = (uint16_t)i;
        }
        for (int pos = N / 2 - 1; pos >= 0; --pos) sift_down(pos);
        }
        bool out = false;
        while (t < tmax && !out) {
            const double tp = ht[0];
            const int move = hid[0];
            while (tp >= nextstep) {
                P.Es[(size_t)ns * Rp + r] = E; ns += 1;
                nextstep += st;
                if (nextstep > tmax + 1e-10) { out = true; break; }
            }
            if (out) break;
            t = tp;
            const double dE = c.dEv(c.lf()[(size_t)move * Rp]);
            c.flip(move);                             // update_heap!: WaitingTimes.jl:40-52
            const double* a = c.lf();
            update(move, t + gen_wt(c.dEv(a[(size_t)move * Rp])));          // = -dE
            for (int j = 0; j < N; ++j) {
                if (j == move) continue;
                update(j, t + gen_wt(c.dEv(a[(size_t)j * Rp])));
            }
            E += dE;
            accepted += 1;
        }
        for (; ns < P.iters; ++ns) { P.Es[(size_t)ns * Rp + r] = E; nextstep += st; }   // not reached: the loop always emits `samples` samples
        staged_its = accepted;
        P.t_out[r] = t;
        sf[SF_TIME] = t; sf[SF_NEXT] = nextstep; sq[SI_ND] = (long long)nd;
    }
    if (P.mode == 1) {
        // bklMC (RRRMC.jl:311-359): rand_skip (DeltaE.jl:319-325), rand_move, apply_step_bkl! = apply_move! over all spins
        // (a resumed call continues the run's loop with `iters` more iterations allowed; the pending move is drawn again: same draw)
        long long it = 0, nextstep = P.step, m = 0, limit = P.iters;
        if (resume) { it = sq[SI_IT]; nextstep = sq[SI_NEXT]; m = sq[SI_M]; limit += sq[SI_LIMIT]; }
        const bool over = resume && nextstep > limit && nextstep > P.step;      // (a resumed call whose allowance does not reach the run's next sample point makes no move: the reference's loop ends with its last sample, RRRMC.jl:340-343)
        while (!over && it < limit) {
            const uint64_t g = P.g0 + (uint64_t)(m + 1);
            const Philox4 o3 = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), rep, TAG_RRR | (2u << 8), P.k0, P.k1);
            const double us = (double)((((uint64_t)o3.w[0] << 32) | o3.w[1]) >> 11) * 0x1.0p-53;
            double b = c.z / (double)N;
            if (b < 2.2250738585072014e-308) b = 2.2250738585072014e-308;
            if (b > 1.0) b = 1.0;
            const double skipf = floor(det_log1p(-us) / det_log1p(-b));
            const long long skip = skipf >= 9.0e18 ? (long long)9.0e18 : (long long)skipf;
            const Philox4 o = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), rep, TAG_RRR, P.k0, P.k1);
            const int move = c.getel((double)((((uint64_t)o.w[0] << 32) | o.w[1]) >> 11) * 0x1.0p-53);
            if (move < 0) { bad = 1; break; }
            const double dE = P.dEs[(size_t)move * Rp + r];
            bool out = false;
            while (it + skip + 1 >= nextstep) {
                if (nextstep > limit) { out = true; break; }        // (fewer iterations allowed than `step`: the sample point lies beyond this call — no sample, no move)
                P.Es[(size_t)ns * Rp + r] = E; ns += 1;
                nextstep += P.step;
                if (nextstep > limit) { out = true; break; }
            }
            if (out) break;
            c.flip(move);
            const double* a = c.lf();
            {
                const double d = c.dEv(a[(size_t)move * Rp]);
                P.dEs[(size_t)move * Rp + r] = d;
                c.set(move, prior_of(P.beta * d));
            }
            for (int j = 0; j < N; ++j) {
                if (j == move) continue;
                const double d = c.dEv(a[(size_t)j * Rp]);
                P.dEs[(size_t)j * Rp + r] = d;
                c.set(j, prior_of(P.beta * d));
            }
            m += 1;
            it += skip + 1;
            E += dE;
            accepted += 1;
        }
        staged_its = accepted;
        sq[SI_IT] = it; sq[SI_NEXT] = nextstep; sq[SI_M] = m; sq[SI_LIMIT] = limit;
    }
    long long next_sample = P.S.samp0;       // iterations k * step of the run: a counter instead of a 64-bit modulo per iteration
    for (long long it = 1; P.mode == 0 && it <= P.iters && !bad; ++it) {
        if (it == next_sample) { next_sample += P.step; P.Es[(size_t)ns * Rp + r] = E; ns += 1; }
        const uint64_t g = P.g0 + (uint64_t)it;
        const Philox4 o = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), rep, TAG_RRR, P.k0, P.k1);
        const double u0 = (double)((((uint64_t)o.w[0] << 32) | o.w[1]) >> 11) * 0x1.0p-53;
        const Philox4 o2 = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), rep, TAG_RRR | (1u << 8), P.k0, P.k1);
        const double u1 = (double)((((uint64_t)o2.w[0] << 32) | o2.w[1]) >> 11) * 0x1.0p-53;
        bool acc = false;
        if (acc_rate < P.staged_thr) {
            staged_its += 1;
            const double z = c.z;
            const int move = c.getel(u0);
            if (move < 0) { bad = 1; break; }
            const double dE = P.dEs[(size_t)move * Rp + r];
            // compute_staged!: flip, record (dE', p') for move then every other spin in order, flip back
            c.flip(move);
            {
                const double* a = c.lf();
                for (int j = 0; j < N; ++j) {
                    const double d = c.dEv(a[(size_t)j * Rp]);
                    P.st_dE[(size_t)j * Rp + r] = d;
                    P.st_p[(size_t)j * Rp + r] = prior_of(P.beta * d);
                }
            }
            c.flip(move);
            // compute_reverse_probabilities!: z' = z + sum (p' - v) in staged order (move first, then AllButOne)
            double zp = c.z;
            zp += P.st_p[(size_t)move * Rp + r] - P.v[(size_t)move * Rp + r];
            for (int j = 0; j < N; ++j) {
                if (j == move) continue;
                zp += P.st_p[(size_t)j * Rp + r] - P.v[(size_t)j * Rp + r];
            }
            if (zp < 2.2250738585072014e-308) zp = 2.2250738585072014e-308;
            if (zp > (double)N) zp = (double)N;
            const double cc = z / zp;
            if (u1 < cc) {
                c.flip(move);
                P.dEs[(size_t)move * Rp + r] = P.st_dE[(size_t)move * Rp + r];
                c.set(move, P.st_p[(size_t)move * Rp + r]);
                for (int j = 0; j < N; ++j) {
                    if (j == move) continue;
                    P.dEs[(size_t)j * Rp + r] = P.st_dE[(size_t)j * Rp + r];
                    c.set(j, P.st_p[(size_t)j * Rp + r]);
                }
                E += dE;
                accepted += 1;
                acc = true;
            }
        } else {
            const int move = c.getel(u0);
            if (move < 0) { bad = 1; break; }
            const double dE = P.dEs[(size_t)move * Rp + r];
            for (int pass = 0; pass < 2; ++pass) {
                c.flip(move);
                const double z = c.z;
                const double* a = c.lf();
                {
                    const double d = c.dEv(a[(size_t)move * Rp]);
                    P.dEs[(size_t)move * Rp + r] = d;
                    c.set(move, prior_of(P.beta * d));
                }
                for (int j = 0; j < N; ++j) {
                    if (j == move) continue;
                    const double d = c.dEv(a[(size_t)j * Rp]);
                    P.dEs[(size_t)j * Rp + r] = d;
                    c.set(j, prior_of(P.beta * d));
                }
                const double cc = z / c.z;
                if (pass == 1) break;
                if (u1 < cc) { E += dE; accepted += 1; acc = true; break; }
            }
        }
        acc_rate = acc_rate * (1 - P.lambda) + (acc ? 1.0 : 0.0) * P.lambda;
    }
    P.E_cur[r] = E;
    P.stats[(size_t)r * 2] = accepted;
    P.stats[(size_t)r * 2 + 1] = staged_its;
    P.status[r] = bad;
    P.z_out[r] = c.z;
    sf[SF_E] = E; sf[SF_Z] = c.z; sf[SF_ACC] = acc_rate;
    sq[SI_CUR] = c.cur; sq[SI_MLAST] = c.move_last; sq[SI_TREF] = c.trefresh;
}

// layout changes between the SK sweep kernel's spins ([G8][N] bytes, bit = replica & 7) and this kernel's ([W][Rp] words)
// ---------------------------------------------------------------------------------------------------
// extremal_opt (src/RRRMC.jl:474-521) with the generic EOCacheCont (src/DeltaE.jl:557-635) on the dense SK models: every spin is a
// neighbour (AllButOne, SK.jl:142,297), so every delta_energy changes at every flip and the reference re-sorts the whole ranking
// (sortperm! + rankshuffle!).  One wavefront per replica: fields (with the lfields / lfields_last swap of SK.jl:247-250), spins and
// the ranking live in LDS; the chain is executed wave-uniformly, the 64 lanes share the O(N) field update (row `move` of J, coalesced)
// and a bitonic sort of (dE, tie key, site) — tie keys (the restatement of rankshuffle!, see cont_kernels.hpp) only when two values
// are equal: never with Gaussian couplings, always with the binary model's integer fields.
// stats = (iterations, itmin), t_out = Emin, cmin = configuration of minimum energy.
// ---------------------------------------------------------------------------------------------------
constexpr int kEoSkMaxN = 4096;
inline size_t eo_sk_lds_bytes(int N, int N2, int W) { return (size_t)N * 16 + (size_t)N2 * 18 + (size_t)W * 4 + 16; }

__global__ __launch_bounds__(64) void eo_sk_wave_kernel(RrrSkParams P)
{
    extern __shared__ __align__(16) unsigned char esk_lds[];
    const int r = blockIdx.x, lane = threadIdx.x;
    const int N = P.N, N2 = P.N2, Rp = P.Rp, W = P.W;
    double* fa = reinterpret_cast<double*>(esk_lds);                 // lfields
    double* fb = fa + N;                                             // lfields_last
    double* val = fb + N;                                            // [N2] dE by rank
    unsigned long long* key = reinterpret_cast<unsigned long long*>(val + N2);      // [N2]
    uint32_t* sp = reinterpret_cast<uint32_t*>(key + N2);            // [W]
    uint16_t* site = reinterpret_cast<uint16_t*>(sp + W);            // [N2]
    const uint32_t rep = P.replica0 + (uint32_t)r;
    auto sbit = [&](int x) { return (int)((sp[x >> 5] >> (x & 31)) & 1u); };
    auto dEv = [&](double lfv) { return P.sN > 0.0 ? lfv / P.sN : lfv; };
    for (int w = lane; w < W; w += 64) sp[w] = P.spins[(size_t)w * Rp + r];
    __syncthreads();
    double* const sf = P.S.sf + (size_t)r * kSmpF;
    long long* const sq = P.S.si + (size_t)r * kSmpI;
    const bool resume = P.S.resume != 0;
    double E;
    int move_last = -1;
    if (resume) {
        // a resumed call: the two field arrays, move_last, E, Emin / itmin as the previous call left them (lfA = lfields, lfB = lfields_last);
        // the ranking is a function of the fields and of the last move's tie keys: it is re-established below
        for (int i = lane; i < N; i += 64) { fa[i] = P.lfA[(size_t)i * Rp + r]; fb[i] = P.lfB[(size_t)i * Rp + r]; }
        E = sf[SF_E];
        move_last = (int)sq[SI_MLAST];
        __syncthreads();
    } else {
    // energy(X, C) (SK.jl:212-237): a row per lane, the row sums in row order
    for (int i = lane; i < N; i += 64) {
        const int si = sbit(i);
        double lfh = 0.0;
        for (int j = 0; j < N; ++j) { const double Jij = P.J[(size_t)j * N + i]; lfh += (si ^ sbit(j)) ? -Jij : Jij; }     // J symmetric: coalesced
        fa[i] = 2 * lfh;
        fb[i] = 0.0;
    }
    __syncthreads();
    double n = 0.0;
    for (int base = 0; base < N; base += 64) {
        const int i = base + lane;
        const double h = i < N ? fa[i] * 0.5 : 0.0;                  // = lfh exactly
        const int m = N - base < 64 ? N - base : 64;
        for (int l = 0; l < m; ++l) n -= __shfl(h, l);
    }
    E = n / 2;
    if (P.sN > 0.0) E = E / P.sN;
    }

    auto less = [&](int a, int b) {
        const double xa = val[a], xb = val[b];
        if (xa < xb) return true;
        if (xa > xb) return false;
        const unsigned long long ka = key[a], kb = key[b];
        if (ka != kb) return ka < kb;
        return site[a] < site[b];
    };
    auto sort_all = [&]() {                                          // bitonic, ascending (dE, key, site); padding sorts last
        for (int k = 2; k <= N2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = lane; t < (N2 >> 1); t += 64) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                    const bool up = (i & k) == 0;
                    if (less(l, i) == up) {
                        const double x = val[i]; val[i] = val[l]; val[l] = x;
                        const unsigned long long q = key[i]; key[i] = key[l]; key[l] = q;
                        const uint16_t u = site[i]; site[i] = site[l]; site[l] = u;
                    }
                }
                __syncthreads();
            }
    };
    // the ranking after the move of iteration g (g = 0: construction, sortperm without shuffle: DeltaE.jl:568)
    auto rerank = [&](uint64_t g, bool fresh) {
        const bool keys_first = fresh && P.sN > 0.0;                 // integer fields: ties are the rule
        for (int p = lane; p < N2; p += 64) {
            if (p < N) {
                val[p] = dEv(fa[p]); site[p] = (uint16_t)p;
                unsigned long long kq = 0ull;
                if (keys_first) {
                    const Philox4 o = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), rep, TAG_RRR | (4u << 8) | ((uint32_t)p << 16), P.k0, P.k1);
                    kq = ((unsigned long long)o.w[0] << 32) | o.w[1];
                }
                key[p] = kq;
            } else { val[p] = __builtin_huge_val(); key[p] = ~0ull; site[p] = (uint16_t)0xffff; }
        }
        __syncthreads();
        sort_all();
        if (!fresh || keys_first) return;
        int part = 0;
        for (int p = 1 + lane; p < N; p += 64) part += val[p - 1] == val[p] ? 1 : 0;
        for (int o = 32; o; o >>= 1) part += __shfl_xor(part, o);
        if (part == 0) return;
        for (int p = lane; p < N; p += 64) {
            const Philox4 o = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), rep, TAG_RRR | (4u << 8) | ((uint32_t)site[p] << 16), P.k0, P.k1);
            key[p] = ((unsigned long long)o.w[0] << 32) | o.w[1];
        }
        __syncthreads();
        sort_all();
    };
    // (resumed: the ranking after the run's last move, iteration P.g0 of the streams — the same sort of the same values and keys)
    if (resume && P.S.it0 > 0) rerank(P.g0, true); else rerank(0, false);

    uint32_t* cm = P.cmin + (size_t)r * W;
    double Emin = E;
    long long itmin = 0, ns = 0, next_sample = P.S.samp0;
    if (resume) { Emin = sf[SF_EMIN]; itmin = sq[SI_ITMIN]; }
    else for (int w = lane; w < W; w += 64) cm[w] = sp[w];
    const double z = P.ftau[N - 1];
    for (long long it = 1; it <= P.iters; ++it) {
        if (it == next_sample) { next_sample += P.step; if (lane == 0) P.Es[(size_t)ns * Rp + r] = E; ns += 1; }
        const uint64_t g = P.g0 + (uint64_t)it;
        const Philox4 o = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), rep, TAG_RRR | (3u << 8), P.k0, P.k1);
        const double rr = (1 - (double)((((uint64_t)o.w[0] << 32) | o.w[1]) >> 11) * 0x1.0p-53) * z;      // rand_move: DeltaE.jl:577-590
        int lo = 0, hi = N;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (P.ftau[mid] < rr) lo = mid + 1; else hi = mid; }
        if (lo > N - 1) lo = N - 1;
        const int move = site[lo];
        const double dE = val[lo];
        __syncthreads();
        // spinflip!: bit flip + update_cache! (SK.jl:239-276 / :98-135)
        if (lane == 0) sp[move >> 5] ^= 1u << (move & 31);
        __syncthreads();
        if (move_last == move) {
            double* t = fa; fa = fb; fb = t;
        } else {
            const double* Ji = P.J + (size_t)move * N;
            const int si = sbit(move);
            const double lfm = fa[move];
            __syncthreads();
            for (int j = lane; j < N; j += 64) {
                const double Js = (si ^ sbit(j)) ? -Ji[j] : Ji[j];
                const double lfj = fa[j];
                fb[j] = lfj;
                fa[j] = lfj + 4 * Js;
            }
            __syncthreads();
            if (lane == 0) { fb[move] = lfm; fa[move] = -lfm; }
            move_last = move;
        }
        __syncthreads();
        rerank(g, true);
        E += dE;
        if (E < Emin) {
            Emin = E; itmin = P.S.it0 + it;
            for (int w = lane; w < W; w += 64) cm[w] = sp[w];
        }
    }
    __syncthreads();
    for (int w = lane; w < W; w += 64) P.spins[(size_t)w * Rp + r] = sp[w];
    for (int i = lane; i < N; i += 64) { P.lfA[(size_t)i * Rp + r] = fa[i]; P.lfB[(size_t)i * Rp + r] = fb[i]; }
    if (lane == 0) {
        P.E_cur[r] = E;
        P.stats[(size_t)r * 2] = P.iters; P.stats[(size_t)r * 2 + 1] = itmin;
        P.t_out[r] = Emin;
        P.status[r] = 0;
        sf[SF_E] = E; sf[SF_EMIN] = Emin; sq[SI_ITMIN] = itmin; sq[SI_MLAST] = move_last;
    }
}

// binary GraphSK couplings (bit rows, SK.jl:32) as the +-1 matrix the kernel above works on: Jt[i][j] = 2 J_ij - 1, zero diagonal
__global__ __launch_bounds__(256) void skb_dense_kernel(const uint32_t* __restrict__ Jbits, double* __restrict__ Jt, int N, int NW)
{
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= N) return;
    const int b = (int)((Jbits[(size_t)i * NW + (j >> 5)] >> (j & 31)) & 1u);
    Jt[(size_t)i * N + j] = i == j ? 0.0 : (b ? 1.0 : -1.0);
}

__global__ __launch_bounds__(256) void rrsk_spins_in_kernel(const uint8_t* __restrict__ sk_spins, uint32_t* __restrict__ spins, int N, int /*W*/, int Rp)
{
    const int r = blockIdx.x * 256 + threadIdx.x, w = blockIdx.y;
    if (r >= Rp) return;
    uint32_t word = 0u;
    for (int b = 0; b < 32; ++b) {
        const int x = 32 * w + b;
        if (x < N) word |= (uint32_t)((sk_spins[(size_t)(r >> 3) * N + x] >> (r & 7)) & 1u) << b;
    }
    spins[(size_t)w * Rp + r] = word;
}
__global__ __launch_bounds__(256) void rrsk_spins_out_kernel(const uint32_t* __restrict__ spins, uint8_t* __restrict__ sk_spins, int N, int Rp)
{
    const int x = blockIdx.x * 256 + threadIdx.x, g8 = blockIdx.y;
    if (x >= N) return;
    uint32_t byte = 0u;
    for (int b = 0; b < 8; ++b) byte |= ((spins[(size_t)(x >> 5) * Rp + g8 * 8 + b] >> (x & 31)) & 1u) << b;
    sk_spins[(size_t)g8 * N + x] = (uint8_t)byte;
}

// ---------------------------------------------------------------------------------------------------
// rrrMC(X::SingleGraph) (src/RRRMC.jl:149-219) and bklMC (src/RRRMC.jl:311-359) on the DiscrGraphs GraphRRG / GraphEA with the
// integer-level DeltaECache{Int,L} (src/DeltaE.jl:63-295): SURVEY.md §8(f) rank 1.  One thread per replica, as above.
// ---------------------------------------------------------------------------------------------------
constexpr int kSLmax = 8;          // levels of allΔE: K/2 + 1 <= 4 for +-J couplings with K <= 7, up to 8 for general levels
// Level table shared by the integer-level kernels below.  dElist = allΔE(X) in level units (RRG.jl:262-281, EA.jl:293-309);
// skip_zero: GraphRRG's neighbors(X, i) keeps the non-zero couplings only (RRG.jl:133) — set for general-level GraphRRG
// (GraphEA removes repeats instead, EA.jl:158, which every kernel does anyway: a GraphRRG row has none).
struct LevTable {
    int L, skip_zero;
    int dElist[kSLmax];
    __host__ __device__ __forceinline__ int find(int ad) const      // findk: DeltaE.jl:28-60 (exact comparison of |dE|)
    {
        int a = 0;
        for (int k = 0; k < L; ++k) if (dElist[k] == ad) a = k;
        return a;
    }
};
struct RrrSparseParams {
    const int32_t* A;        // [N][K]
    const int8_t* J;         // [N][K]
    uint32_t* spins;         // [R][W]   replica-contiguous words
    uint8_t* cls;            // [R][N]
    void* sv;                // [R][2L][N]  IDX = uint16_t, or uint32_t when N > 65535 (GraphEA(64, 3): N = 262 144)
    void* spos;              // [R][N]
    int32_t* E_cur;          // [Rpad]
    int64_t* acc_cur;        // [Rpad]
    int64_t* stats;          // [R][3]   accepted, staged iterations / true moves, iterations done
    int32_t* Es;             // [nsamples][Rpad]
    double ft[kSLmax];
    double beta, staged_thr, lambda;
    uint64_t g0;
    int64_t iters, step;
    uint32_t k0, k1, replica0;
    LevTable lv;
    int N, K, L, W, R, Rpad, mode;      // mode 0 = rrrMC, 1 = bklMC
    SmpState S;
};

// LDS = true (one replica per workgroup, see rrr_sparse_kernel): spins, classes, positions and the neighbour table (as 16-bit ids)
// are LDS copies
// SLM = compile-time bound on the number of levels (2, 4 or 8): the per-class arrays have 2 SLM entries and must stay in REGISTERS
// (dynamically indexed arrays of 16 doubles end up in scratch memory, a memory round trip per access: 5x slower kernels)
// The last D members of every class list, followed through the pushes and pops of ONE apply_move! in registers.  A set move
// (ArraySet delete! + push!, ArraySets.jl:56-76) needs the LAST member of the list it deletes from; gathered before the set moves start, that
// value is stale as soon as an earlier move of the same iteration touched the list, and re-reading it is a dependent memory round trip in
// the middle of the store sequence — and with sixteen replicas per wavefront one lane that needs it makes all of them wait (a fifth of
// eo_sparse_kernel's time, profiles/r06/f8_floor.md §3).  A push makes the pushed site the last member; a pop exposes the member before
// (known if it was gathered or pushed) and puts the old last member where the deleted site sat — inside the known window or not.  With
// D = 4 a move of a K = 3 graph (four set moves) never reads a list's end again; beyond what is known the read remains.
// NC = compile-time bound on the number of classes; all accesses are unrolled selects (registers).
template <int NC, int D>
struct TailTrack {
    int e[D][NC];            // e[i][c]: the i-th member from the end of list c
    int kn[NC];              // how many of them are known
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int a = 0; a < NC; ++a) { kn[a] = 0;
#pragma unroll
            for (int i = 0; i < D; ++i) e[i][a] = 0; } }
    __device__ __forceinline__ int known(int c) const { int x = 0;
#pragma unroll
        for (int a = 0; a < NC; ++a) x = c == a ? kn[a] : x;
        return x; }
    __device__ __forceinline__ int last(int c) const { int x = 0;
#pragma unroll
        for (int a = 0; a < NC; ++a) x = c == a ? e[0][a] : x;
        return x; }
    struct Ends { int v[D]; };                         // (by value: a reference to a local array would pin it in scratch memory)
    // what the gather read at the end of list c (count n at that time): w.v[i] = member n - 1 - i
    __device__ __forceinline__ void note(int c, int n, Ends w) {
#pragma unroll
        for (int a = 0; a < NC; ++a) {
            kn[a] = c == a ? (n < D ? n : D) : kn[a];
#pragma unroll
            for (int i = 0; i < D; ++i) e[i][a] = c == a ? w.v[i] : e[i][a];
        } }
    // list c (count n before) lost the member at position p; `lastm` (its last member) was moved there
    __device__ __forceinline__ void popped(int c, int n, int p, int lastm)
    {
        const int idx = n - 2 - p;                       // where position p sits from the end of the shorter list (-1: the deleted site was the last)
#pragma unroll
        for (int a = 0; a < NC; ++a) {
            const bool me = c == a;
            int nk = kn[a] > 0 ? kn[a] - 1 : 0;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const int shifted = i + 1 < D ? e[i + 1 < D ? i + 1 : i][a] : e[i][a];
                const int x = (idx == i && i <= nk) ? lastm : shifted;                       // inside the known window, or right behind it
                e[i][a] = me ? x : e[i][a];
            }
            if (idx >= 0 && idx == nk && nk < D) nk += 1;
            kn[a] = me ? nk : kn[a];
        }
    }
    __device__ __forceinline__ void pushed(int c, int j)
    {
#pragma unroll
        for (int a = 0; a < NC; ++a) {
            const bool me = c == a;
#pragma unroll
            for (int i = D - 1; i >= 1; --i) e[i][a] = me ? e[i - 1][a] : e[i][a];
            e[0][a] = me ? j : e[0][a];
            kn[a] = me ? (kn[a] < D ? kn[a] + 1 : D) : kn[a];
        }
    }
};

template <bool LDS, int SLM, typename IDX = uint16_t>
struct SparseChain {
    // copies of the few parameters the chain needs: a pointer to the kernel's parameter struct would force that struct — and every
    // access to it — into scratch memory
    struct Cfg { int N, K, L, skip_zero; const int32_t* A; const int8_t* J; int dEl[SLM]; double ft[SLM]; };
    Cfg cfg;
    uint32_t* sp; uint8_t* cls; IDX* sv; IDX* spos;
    const uint16_t* A16; const int8_t* Jl;          // LDS copies of the neighbour table / couplings (LDS build only)
    __device__ __forceinline__ int nbr(int i, int q) const { if constexpr (LDS) return (int)A16[i * cfg.K + q]; else return (int)cfg.A[(size_t)i * cfg.K + q]; }
    __device__ __forceinline__ int cpl(int i, int q) const { if constexpr (LDS) return (int)Jl[i * cfg.K + q]; else return (int)cfg.J[(size_t)i * cfg.K + q]; }
    __device__ __forceinline__ int lev(int a) const { int d = 0;
#pragma unroll
        for (int k = 0; k < SLM; ++k) d = a == k ? cfg.dEl[k] : d;
        return d; }
    __device__ __forceinline__ int find(int ad) const { int a = 0;          // findk: DeltaE.jl:28-60 (exact comparison of |dE|)
#pragma unroll
        for (int k = 0; k < SLM; ++k) a = (k < cfg.L && cfg.dEl[k] == ad) ? k : a;
        return a; }
    int t[2 * SLM];
    double T[2 * SLM], z;
    // the per-class arrays are only ever indexed through these fully unrolled selects: a dynamically indexed local array would be
    // placed in scratch memory
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
    // delta_energy (RRG.jl:236-244 / EA.jl:266-275) recomputed from the spins: 2 sigma_i sum_k J_ik sigma_k
    __device__ __forceinline__ int dE(int i) const
    {
        const int si = sbit(i);
        int acc = 0;
        for (int q = 0; q < cfg.K; ++q) {
            const int sy = sbit(nbr(i, q));
            acc += (si == sy) ? cpl(i, q) : -cpl(i, q);
        }
        return 2 * acc;
    }
    __device__ __forceinline__ int klass(int i) const      // a + L*up, DeltaE.jl:80-86
    {
        const int d = dE(i), a = find(d < 0 ? -d : d);
        const int up = d > 0 || (d == 0 && sbit(i) == 1);
        return a + cfg.L * up;
    }
    // neighbors(X, move)[q]: repeats removed (uA, EA.jl:158); zero couplings dropped for a general-level GraphRRG (RRG.jl:133)
    __device__ __forceinline__ bool is_nb(int move, int q) const
    {
        if (q > 0 && nbr(move, q) == nbr(move, q - 1)) return false;
        return !(cfg.skip_zero && cpl(move, q) == 0);
    }
    __device__ __forceinline__ double f(int k) const { double x = 1.0;
#pragma unroll
        for (int a = 0; a < SLM; ++a) x = (k - cfg.L == a) ? cfg.ft[a] : x;
        return x; }
    __device__ __forceinline__ void set_move(int j, int k0, int k1)
    {
        IDX* v0 = sv + (size_t)k0 * cfg.N;
        IDX* v1 = sv + (size_t)k1 * cfg.N;
        const int p = (int)spos[j], last = (int)v0[tg(k0) - 1];
        v0[p] = (IDX)last; spos[last] = (IDX)p; tadd(k0, -1);
        const int t1 = tg(k1);
        v1[t1] = (IDX)j; spos[j] = (IDX)t1; tadd(k1, 1);
        cls[j] = (uint8_t)k1;
    }
    // apply_move!: DeltaE.jl:232-295; returns c = z / z'.
    // One thread walks the chain, so what an iteration costs is its DEPENDENT memory round trips (profiles/r06/f8_floor.md: 20 900 cycles per
    // iteration of a 16-replica wavefront at N = 10^4, 6 600 of them instruction issue).  Written as the reference's loop — neighbour after neighbour: class, new class,
    // delete!, push! — every site pays its own chain of three (class byte and table row; the neighbours' bits; position and the set's last member),
    // because the stores of one site's set move keep the compiler from starting the next site's loads.  Here the K + 1 sites are gathered FIRST
    // (fixed slots, fully unrolled: registers): classes, new classes, positions and — speculatively — the last member of every set that loses a
    // site, all of them independent loads; the set moves then run in the reference's order on values that are already there.  A set move changes
    // what a LATER site of the same iteration gathered in two ways only: the last member of a set another move has touched (then it is read again:
    // `touched`), and the position of a site that was moved into a freed slot (patched, as the wave builds do).  Same stores, same order.
    template <int NB, int KM = NB>
    __device__ double apply_move(int move)
    {
        if constexpr (LDS || KM > 6) {
            sflip(move);
            // the LDS build has no memory latency to hide, and eight (or sixteen) slots of as many row entries do not fit the registers: the
            // reference's loop as it stands
            double zp = z;
            for (int q = 0; q <= cfg.K; ++q) {
                if (q < cfg.K && !is_nb(move, q)) continue;
                const int j = q < cfg.K ? nbr(move, q) : move;
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
        } else {
        // K <= KM: slot q < KM = neighbour q, slot KM = the moved spin
        int sj[KM + 1], s0[KM + 1], s1[KM + 1], sp_[KM + 1], sl[KM + 1];
        bool live[KM + 1];
        {
            // The gather, stage by stage, every load of a stage issued without a branch around it (a slot that holds no neighbour reads the
            // moved spin's entries instead: harmless, and the stage stays one basic block, so its loads go out together):
            //   1. the moved spin's row of the table;  2. per neighbour its class byte, its position, its word of spins, its own row;
            //   3. the words of its neighbours' spins -> delta_energy, the new class;  4. the last member of every set that may lose a site.
            const int K = cfg.K;
            // (the flip: its word is requested WITH the row — read on its own, before everything else, it is a round trip of its own)
            const uint32_t wm = sp[move >> 5];
            int y[KM], cj[KM];
#pragma unroll
            for (int q = 0; q < KM; ++q) { const size_t e = (size_t)move * K + (q < K ? q : 0); y[q] = cfg.A[e]; cj[q] = (int)cfg.J[e]; }
            sp[move >> 5] = wm ^ (1u << (move & 31));
            bool val[KM];
#pragma unroll
            for (int q = 0; q < KM; ++q) val[q] = q < K && !(q > 0 && y[q] == y[q - 1]) && !(cfg.skip_zero && cj[q] == 0);      // is_nb
            int yy[KM][KM], cc[KM][KM];
            uint32_t wown[KM];
#pragma unroll
            for (int q = 0; q < KM; ++q) {
                const int j = val[q] ? y[q] : move;
                sj[q] = j;
                s0[q] = cls[j]; sp_[q] = (int)spos[j]; wown[q] = sp[j >> 5];
#pragma unroll
                for (int k = 0; k < KM; ++k) { const size_t e = (size_t)j * K + (k < K ? k : 0); yy[q][k] = cfg.A[e]; cc[q][k] = k < K ? (int)cfg.J[e] : 0; }
            }
            s0[KM] = cls[move]; sp_[KM] = (int)spos[move]; sj[KM] = move;
            uint32_t wnb[KM][KM];
#pragma unroll
            for (int q = 0; q < KM; ++q)
#pragma unroll
                for (int k = 0; k < KM; ++k) wnb[q][k] = sp[yy[q][k] >> 5];
#pragma unroll
            for (int q = 0; q < KM; ++q) {
                const int sjb = (int)((wown[q] >> (sj[q] & 31)) & 1u);
                int acc = 0;
#pragma unroll
                for (int k = 0; k < KM; ++k) { const int sy = (int)((wnb[q][k] >> (yy[q][k] & 31)) & 1u); acc += (sjb == sy) ? cc[q][k] : -cc[q][k]; }
                const int d = 2 * acc, a = find(d < 0 ? -d : d);
                s1[q] = a + cfg.L * ((d > 0 || (d == 0 && sjb == 1)) ? 1 : 0);          // klass(j): DeltaE.jl:80-86
                live[q] = val[q] && s0[q] != s1[q];
            }
            live[KM] = true; s1[KM] = s0[KM] >= cfg.L ? s0[KM] - cfg.L : s0[KM] + cfg.L;
#pragma unroll
            for (int q = 0; q <= KM; ++q) { const int tq = tg(s0[q]); sl[q] = (int)sv[(size_t)s0[q] * cfg.N + (tq > 0 ? tq - 1 : 0)]; }
        }
        // (following the lists' ends in registers instead — TailTrack, as eo_sparse_kernel does, with the re-read behind a real branch — makes the
        // set moves free of loads and waits, and this kernel a third SLOWER: 120 against 92 ms, measured twice; it is at the edge of its
        // registers, 147 scalar ones spilled, with three gather sizes inlined at three call sites: profiles/r06/f8_floor.md)
        double zp = z;
        unsigned touched = 0u;
#pragma unroll
        for (int q = 0; q <= KM; ++q) {
            if (!live[q]) continue;
            const int j = sj[q], k0 = s0[q], k1 = s1[q], p = sp_[q];
            const double f0 = f(k0), f1 = f(k1);
            Tadd(k0, -f0); Tadd(k1, f1); zp += f1 - f0;
            IDX* v0 = sv + (size_t)k0 * cfg.N;
            IDX* v1 = sv + (size_t)k1 * cfg.N;
            const int last = ((touched >> k0) & 1u) ? (int)v0[tg(k0) - 1] : sl[q];      // ArraySet delete!(k0, j) + push!(k1, j), ArraySets.jl:56-76
            v0[p] = (IDX)last; spos[last] = (IDX)p; tadd(k0, -1);
            const int t1 = tg(k1);
            v1[t1] = (IDX)j; spos[j] = (IDX)t1; tadd(k1, 1);
            cls[j] = (uint8_t)k1;
            touched |= (1u << k0) | (1u << k1);
#pragma unroll
            for (int q2 = 0; q2 <= KM; ++q2)
                if (q2 > q && live[q2] && sj[q2] == last) sp_[q2] = p;
        }
        const double cc2 = z / zp;
        z = zp;
        return cc2;
        }
    }
};

// bytes of LDS of the LDS build: spins, positions, classes, the neighbour table as 16-bit ids, the couplings, 64 x 8 draw words
inline size_t rrr_sparse_lds_bytes(int64_t N, int64_t W, int64_t K)
{
    return (size_t)W * 4 + (((size_t)N * 2 + 3) & ~(size_t)3) + (((size_t)N + 3) & ~(size_t)3) + (((size_t)N * K * 2 + 3) & ~(size_t)3) +
           (((size_t)N * K + 3) & ~(size_t)3) + (size_t)kRrrThreads * 8 * 4;
}

// LDS = false: one thread per replica, everything in HBM/L2.  LDS = true: one workgroup (one wavefront) per replica, as for
// rrr_quant_kernel: the replica's spins, class bytes and set positions and the graph (16-bit neighbour ids, couplings) staged in LDS
// by all 64 lanes, the two Philox blocks of an rrrMC iteration computed 64 iterations at a time by the whole wavefront; lane 0
// runs the chain.  The reference's own experiment (scripts/scripts.jl:test_RRG: N = 10^4, K = 3) takes 121 KB.
// NB = compile-time bound on K for the staged path's register-resident change list (8; 16 for the K > 8 graphs: GraphEA with D >= 5)
template <bool LDS, int SLM, typename IDX = uint16_t, int NB = 8>
__global__ __launch_bounds__(kRrrThreads) void rrr_sparse_kernel(RrrSparseParams P)
{
    extern __shared__ uint32_t rs_lds[];
    int r;
    if constexpr (LDS) {
        r = (int)blockIdx.x;
    } else {
        r = blockIdx.x * blockDim.x + threadIdx.x;
        if (r >= P.R) return;
    }
    const int N = P.N, L = P.L, K2 = 2 * P.L;
    static_assert(!LDS || sizeof(IDX) == 2, "the LDS build keeps 16-bit positions");
    SparseChain<LDS, SLM, IDX> c;
    c.cfg.N = P.N; c.cfg.K = P.K; c.cfg.L = P.L; c.cfg.skip_zero = P.lv.skip_zero; c.cfg.A = P.A; c.cfg.J = P.J;
#pragma unroll
    for (int k = 0; k < SLM; ++k) { c.cfg.dEl[k] = P.lv.dElist[k]; c.cfg.ft[k] = P.ft[k]; }
    c.sp = P.spins + (size_t)r * P.W; c.cls = P.cls + (size_t)r * N; c.sv = static_cast<IDX*>(P.sv) + (size_t)r * K2 * N; c.spos = static_cast<IDX*>(P.spos) + (size_t)r * N;
    c.A16 = nullptr; c.Jl = nullptr;
    uint32_t* g_sp = c.sp; uint8_t* g_cls = c.cls; IDX* g_spos = c.spos;
    uint32_t* l_rng = nullptr;
    if constexpr (LDS) {
        const int tid = (int)threadIdx.x, nt = (int)blockDim.x, NK = N * P.K;
        uint32_t* l_sp = rs_lds;                                                          // [W]
        uint16_t* l_spos = reinterpret_cast<uint16_t*>(l_sp + P.W);                       // [N]
        uint8_t* l_cls = reinterpret_cast<uint8_t*>(l_spos) + ((2 * N + 3) & ~3);         // [N]
        uint16_t* l_A = reinterpret_cast<uint16_t*>(l_cls + ((N + 3) & ~3));              // [N][K]
        int8_t* l_J = reinterpret_cast<int8_t*>(l_A) + ((2 * NK + 3) & ~3);               // [N][K]
        l_rng = reinterpret_cast<uint32_t*>(l_J + ((NK + 3) & ~3));                       // [64][8]
        for (int i = tid; i < P.W; i += nt) l_sp[i] = g_sp[i];
        for (int i = tid; i < NK; i += nt) { l_A[i] = (uint16_t)P.A[i]; l_J[i] = P.J[i]; }
        if (P.S.resume)                          // the positions and classes the previous call wrote back
            for (int i = tid; i < N; i += nt) { l_spos[i] = (uint16_t)g_spos[i]; l_cls[i] = g_cls[i]; }
        __syncthreads();
        c.sp = l_sp; c.spos = reinterpret_cast<IDX*>(l_spos); c.cls = l_cls; c.A16 = l_A; c.Jl = l_J;
    }
    const bool worker = !LDS || threadIdx.x == 0;
    long long E = 0, accepted = 0, staged_its = 0, ns = 0, itdone = 0;
    double* const sf = P.S.sf + (size_t)r * kSmpF;
    long long* const si = P.S.si + (size_t)r * kSmpI;
    if (worker && P.S.resume) {
        // a resumed call: the cache is where the previous call left it (member order, T, z: DeltaE.jl:63-73), E is the tracked energy
#pragma unroll
        for (int k = 0; k < 2 * SLM; ++k) { c.t[k] = (int)si[SI_T0 + k]; c.T[k] = sf[SF_T0 + k]; }
        c.z = sf[SF_Z];
        E = P.E_cur[r];
    } else if (worker) {
    // energy(X, C) and gen_ΔEcache in site order (RRRMC.jl:177-178, DeltaE.jl:74-103)
    long long n = 0;
#pragma unroll
    for (int k = 0; k < 2 * SLM; ++k) c.t[k] = 0;
    for (int i = 0; i < N; ++i) {
        n -= c.dE(i) / 2;                       // lf_x = -sum J sx sy
        const int k = c.klass(i), tk = c.tg(k);
        c.cls[i] = (uint8_t)k;
        c.sv[(size_t)k * N + tk] = (IDX)i;
        c.spos[i] = (IDX)tk;
        c.tadd(k, 1);
    }
    E = n / 2;
    c.z = 0.0;
#pragma unroll
    for (int k = 0; k < 2 * SLM; ++k) c.T[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 2 * SLM; ++k)
        if (k < K2) { const double x = (double)c.t[k] * c.f(k); c.z += x; c.T[k] = x; }
    }

    const uint32_t rep = P.replica0 + (uint32_t)r;
    // apply_move! with its gather sized for the graph's degree (kernel-uniform).  always_inline: called out of line, the chain's state would
    // live in scratch (measured: 3.4e8 against 5.2e8 iterations/s)
    auto apply = [&](int move) __attribute__((always_inline)) -> double {
        if constexpr (LDS) return c.template apply_move<NB, NB>(move);
        else {
            if (c.cfg.K <= 3) return c.template apply_move<NB, 3>(move);
            if (c.cfg.K <= 4) return c.template apply_move<NB, 4>(move);
            if (c.cfg.K <= 6) return c.template apply_move<NB, 6>(move);
            return c.template apply_move<NB, NB>(move);
        }
    };
    if (P.mode == 0) {
        double acc_rate = (worker && P.S.resume) ? sf[SF_ACC] : 0.5;
        long long next_sample = P.S.samp0;       // iterations k * step of the run: a counter instead of a 64-bit modulo per iteration
        for (long long base = 0; base < P.iters; base += (LDS ? (long long)kRrrThreads : (long long)P.iters)) {
        if constexpr (LDS) {                     // the draws of the next 64 iterations, one iteration per lane
            __syncthreads();
            const uint64_t gl = P.g0 + (uint64_t)(base + 1 + (long long)threadIdx.x);
            const Philox4 a = philox4x32_10((uint32_t)gl, (uint32_t)(gl >> 32), rep, TAG_RRR, P.k0, P.k1);
            const Philox4 b = philox4x32_10((uint32_t)gl, (uint32_t)(gl >> 32), rep, TAG_RRR | (1u << 8), P.k0, P.k1);
            uint32_t* q = l_rng + threadIdx.x * 8;
            q[0] = a.w[0]; q[1] = a.w[1]; q[2] = a.w[2]; q[3] = a.w[3]; q[4] = b.w[0]; q[5] = b.w[1]; q[6] = b.w[2]; q[7] = b.w[3];
            __syncthreads();
        }
        const long long it_end = LDS ? (base + kRrrThreads < P.iters ? base + kRrrThreads : (long long)P.iters) : (long long)P.iters;
        if (worker)
        for (long long it = base + 1; it <= it_end; ++it) {
            if (it == next_sample) { next_sample += P.step; P.Es[(size_t)ns * P.Rpad + r] = (int32_t)E; ns += 1; }
            const uint64_t g = P.g0 + (uint64_t)it;
            Philox4 o, o2;
            if constexpr (LDS) {
                const uint32_t* q = l_rng + (it - base - 1) * 8;
                o.w[0] = q[0]; o.w[1] = q[1]; o.w[2] = q[2]; o.w[3] = q[3]; o2.w[0] = q[4]; o2.w[1] = q[5];
            } else {
                o = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), rep, TAG_RRR, P.k0, P.k1);
                o2 = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), rep, TAG_RRR | (1u << 8), P.k0, P.k1);
            }
            const double u1 = (double)((((uint64_t)o2.w[0] << 32) | o2.w[1]) >> 11) * 0x1.0p-53;
            // rand_move
            const double rr = (double)((((uint64_t)o.w[0] << 32) | o.w[1]) >> 11) * 0x1.0p-53 * c.z;
            const int k = c.pick_class(rr, K2);
            const int dE = k < L ? -c.lev(k) : c.lev(k - L);
            const int move = c.sv[(size_t)k * N + (int)mulhi64(((uint64_t)o.w[2] << 32) | o.w[3], (uint64_t)c.tg(k))];
            bool acc = false;
            if (acc_rate < P.staged_thr) {
                staged_its += 1;
                // staged changes: slot q = neighbour q (slot kNbMax = the moved spin), live[q] says whether it changes class; fixed slots
                // and fully unrolled loops keep these arrays in registers
                constexpr int kNbMax = NB;
                int sj[kNbMax + 1], s0[kNbMax + 1], s1[kNbMax + 1];
                bool live[kNbMax + 1];
                c.sflip(move);
#pragma unroll
                for (int q = 0; q < kNbMax; ++q) {
                    live[q] = false; sj[q] = 0; s0[q] = 0; s1[q] = 0;
                    if (q < P.K && c.is_nb(move, q)) {
                        const int j = c.nbr(move, q), k0 = c.cls[j], k1 = c.klass(j);
                        if (k0 != k1) { live[q] = true; sj[q] = j; s0[q] = k0; s1[q] = k1; }
                    }
                }
                { const int k0 = c.cls[move]; live[kNbMax] = true; sj[kNbMax] = move; s0[kNbMax] = k0; s1[kNbMax] = k0 >= L ? k0 - L : k0 + L; }
                c.sflip(move);
                double Tp[2 * SLM], zp = c.z;
#pragma unroll
                for (int q = 0; q < 2 * SLM; ++q) Tp[q] = c.T[q];
#pragma unroll
                for (int q = 0; q <= kNbMax; ++q)
                    if (live[q]) {
                        const double f0 = c.f(s0[q]), f1 = c.f(s1[q]);
#pragma unroll
                        for (int a = 0; a < 2 * SLM; ++a) Tp[a] = s0[q] == a ? Tp[a] - f0 : Tp[a];
#pragma unroll
                        for (int a = 0; a < 2 * SLM; ++a) Tp[a] = s1[q] == a ? Tp[a] + f1 : Tp[a];
                        zp += f1 - f0;
                    }
                if (u1 < c.z / zp) {
                    c.sflip(move);
#pragma unroll
                    for (int q = 0; q <= kNbMax; ++q)
                        if (live[q]) c.set_move(sj[q], s0[q], s1[q]);
#pragma unroll
                    for (int q = 0; q < 2 * SLM; ++q) c.T[q] = Tp[q];
                    c.z = zp;
                    E += dE; accepted += 1; acc = true;
                }
            } else {
                const double cc = apply(move);
                if (u1 < cc) { E += dE; accepted += 1; acc = true; }
                else apply(move);
            }
            acc_rate = acc_rate * (1 - P.lambda) + (acc ? 1.0 : 0.0) * P.lambda;
        }
        }
        itdone = P.iters;
        if (worker) sf[SF_ACC] = acc_rate;
    } else if (worker) {
        // a resumed call continues the loop of the run (RRRMC.jl:327-350) with `iters` more iterations allowed: `it`, `nextstep` and the
        // number of moves made carry on; the draw of the move that was pending at the cut is taken again — same counter, same cache, same draw
        long long it = 0, nextstep = P.step, m = 0, limit = P.iters;
        if (P.S.resume) { it = si[SI_IT]; nextstep = si[SI_NEXT]; m = si[SI_M]; limit += si[SI_LIMIT]; }
        const bool over = P.S.resume && nextstep > limit && nextstep > P.step;  // (a resumed call whose allowance does not reach the run's next sample point makes no move: the reference's loop ends with its last sample, RRRMC.jl:340-343)
        while (!over && it < limit) {
            const uint64_t g = P.g0 + (uint64_t)(m + 1);
            const Philox4 o3 = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), rep, TAG_RRR | (2u << 8), P.k0, P.k1);
            const double us = (double)((((uint64_t)o3.w[0] << 32) | o3.w[1]) >> 11) * 0x1.0p-53;
            const double skipf = floor(__ddiv_rn(det_log1p(-us), det_log1p(-__ddiv_rn(c.z, (double)N))));     // rand_skip, DeltaE.jl:141-144
            const long long skip = skipf >= 9.0e18 ? (long long)9.0e18 : (long long)skipf;
            const Philox4 o = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), rep, TAG_RRR, P.k0, P.k1);
            const double rr = (double)((((uint64_t)o.w[0] << 32) | o.w[1]) >> 11) * 0x1.0p-53 * c.z;
            const int k = c.pick_class(rr, K2);
            const int dE = k < L ? -c.lev(k) : c.lev(k - L);
            const int move = c.sv[(size_t)k * N + (int)mulhi64(((uint64_t)o.w[2] << 32) | o.w[3], (uint64_t)c.tg(k))];
            bool out = false;
            while (it + skip + 1 >= nextstep) {
                if (nextstep > limit) { out = true; break; }        // (fewer iterations allowed than `step`: the sample point lies beyond this call — no sample, no move)
                P.Es[(size_t)ns * P.Rpad + r] = (int32_t)E; ns += 1;
                nextstep += P.step;
                if (nextstep > limit) { out = true; break; }
            }
            if (out) break;
            apply(move);
            m += 1;
            it += skip + 1;
            E += dE;
            accepted += 1;
        }
        staged_its = accepted;
        itdone = it;
        si[SI_IT] = it; si[SI_NEXT] = nextstep; si[SI_M] = m; si[SI_LIMIT] = limit;
    }
    if (worker) {
        P.E_cur[r] = (int32_t)E;
        P.acc_cur[r] = accepted;
        P.stats[(size_t)r * 3] = accepted; P.stats[(size_t)r * 3 + 1] = staged_its; P.stats[(size_t)r * 3 + 2] = itdone;
#pragma unroll
        for (int k = 0; k < 2 * SLM; ++k) { si[SI_T0 + k] = c.t[k]; sf[SF_T0 + k] = c.T[k]; }
        sf[SF_Z] = c.z;
    }
    if constexpr (LDS) {
        __syncthreads();
        const int tid = (int)threadIdx.x, nt = (int)blockDim.x;
        for (int i = tid; i < P.W; i += nt) g_sp[i] = c.sp[i];
        for (int i = tid; i < N; i += nt) { g_spos[i] = c.spos[i]; g_cls[i] = c.cls[i]; }
    }
}

// ---------------------------------------------------------------------------------------------------
// wtmMC — waiting-time method (src/RRRMC.jl:376-426, src/WaitingTimes.jl) on GraphRRG / GraphEA: SURVEY.md §8(f) rank 4.
// One thread per replica; the reference's MutableBinaryMinHeap becomes a binary min-heap per replica in HBM/L2, ordered by
// (time, site) — only "top = smallest time" is observable.  WTM stream: uniform n of replica r in call c is the 53-bit word
// (n & 1) of ctr = (lo(n >> 1), hi(n >> 1), r, TAG_WTM | c << 8).
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t TAG_WTM = 11;
struct WtmParams {
    const int32_t* A;        // [N][K]
    const int8_t* J;         // [N][K]
    uint32_t* spins;         // [R][W]   replica-contiguous words
    double* ht;              // [R][N]   heap keys (next-flip times) by heap position
    void* hid;               // [R][N]   site at heap position      (uint16_t, or uint32_t when N > 65535)
    void* hpos;              // [R][N]   heap position of site
    int32_t* E_cur;          // [Rpad]
    int64_t* acc_cur;        // [Rpad]   num_moves
    double* t_out;           // [R]      final global time
    int32_t* Es;             // [samples][Rpad]
    double tau[2 * kSLmax];  // tauDE = max(1, exp(beta dE)) for dE = -dElist[a] (entry a) and +dElist[a] (entry L + a)
    LevTable lv;
    double step;             // already divided by N
    int64_t samples;
    uint32_t k0, k1, replica0, call;
    int N, K, W, R, Rpad;
    SmpState S;
    int64_t samples_before;  // samples the run had taken before this call (a resumed call: tmax = step * (samples_before + samples))
};

template <typename IDX>
struct WtmChain {
    const WtmParams* P;
    uint32_t* sp; double* ht; IDX* hid; IDX* hpos;
    uint32_t rep;
    uint64_t nd;
    __device__ __forceinline__ int sbit(int x) const { return (int)((sp[x >> 5] >> (x & 31)) & 1u); }
    __device__ __forceinline__ void sflip(int x) { sp[x >> 5] ^= 1u << (x & 31); }
    __device__ __forceinline__ int dE(int i) const
    {
        const int si = sbit(i);
        int acc = 0;
        for (int q = 0; q < P->K; ++q) {
            const int sy = sbit(P->A[(size_t)i * P->K + q]);
            acc += (si == sy) ? (int)P->J[(size_t)i * P->K + q] : -(int)P->J[(size_t)i * P->K + q];
        }
        return 2 * acc;
    }
    __device__ __forceinline__ double uniform()
    {
        const uint64_t n = nd++, blk = n >> 1;
        const Philox4 o = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), rep, TAG_WTM | (P->call << 8), P->k0, P->k1);
        const uint64_t u = (n & 1u) ? (((uint64_t)o.w[2] << 32) | o.w[3]) : (((uint64_t)o.w[0] << 32) | o.w[1]);
        return (double)(u >> 11) * 0x1.0p-53;
    }
    // gen_wt(tau) = -tau * log1p(-rand()), tau = tauDE(dE): WaitingTimes.jl:16-22
    __device__ __forceinline__ double gen_wt(int d) { return -P->tau[P->lv.find(d < 0 ? -d : d) + (d > 0 ? P->lv.L : 0)] * det_log1p(-uniform()); }
    __device__ __forceinline__ bool before(double ta, int a, double tb, int b) const { return ta < tb || (ta == tb && a < b); }
    __device__ void sift_down(int pos, int n)
    {
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
    __device__ __forceinline__ void update(int i, double t)     // update!(theap, i, t)
    {
        const int pos = hpos[i];
        const double old = ht[pos];
        ht[pos] = t;
        if (t < old) sift_up(pos); else sift_down(pos, P->N);
    }
};

template <typename IDX>
__global__ __launch_bounds__(kRrrThreads) void wtm_sparse_kernel(WtmParams P)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= P.R) return;
    const int N = P.N;
    WtmChain<IDX> c;
    c.P = &P;
    c.sp = P.spins + (size_t)r * P.W; c.ht = P.ht + (size_t)r * N; c.hid = static_cast<IDX*>(P.hid) + (size_t)r * N; c.hpos = static_cast<IDX*>(P.hpos) + (size_t)r * N;
    c.rep = P.replica0 + (uint32_t)r;
    c.nd = 0;
    double* const sf = P.S.sf + (size_t)r * kSmpF;
    long long* const si = P.S.si + (size_t)r * kSmpI;
    const double step = P.step, tmax = step * (double)(P.samples_before + P.samples);
    double t = 0.0, nextstep = step;
    long long E;
    if (P.S.resume) {
        // a resumed call: the heap, the global time and the next sample time of the run carry on (RRRMC.jl:396-416), with `samples` more
        E = P.E_cur[r]; t = sf[SF_TIME]; nextstep = sf[SF_NEXT]; c.nd = (uint64_t)si[SI_ND];
    } else {
    // E = energy(X, C); theap = THeap(X, C, beta): one waiting time per spin, in site order (WaitingTimes.jl:26-36)
    long long n = 0;
    for (int i = 0; i < N; ++i) {
        const int d = c.dE(i);
        n -= d / 2;
        c.ht[i] = c.gen_wt(d);
        c.hid[i] = (IDX)i;
        c.hpos[i] = (IDX)i;
    }
    for (int pos = N / 2 - 1; pos >= 0; --pos) c.sift_down(pos, N);
    E = n / 2;
    }
    long long moves = 0, ns = 0;
    bool out = false;
    while (t < tmax && !out) {
        const double tp = c.ht[0];                  // pick_next
        const int move = c.hid[0];
        while (tp >= nextstep) {
            P.Es[(size_t)ns * P.Rpad + r] = (int32_t)E; ns += 1;
            nextstep += step;
            if (nextstep > tmax + 1e-10) { out = true; break; }
        }
        if (out) break;
        t = tp;
        // update_heap!: WaitingTimes.jl:40-52
        const int d = c.dE(move);
        c.sflip(move);
        c.update(move, t + c.gen_wt(-d));
        const int32_t* Ax = P.A + (size_t)move * P.K;
        for (int q = 0; q < P.K; ++q) {
            if (q > 0 && Ax[q] == Ax[q - 1]) continue;        // uA: repeats removed (EA.jl:158)
            if (P.lv.skip_zero && P.J[(size_t)move * P.K + q] == 0) continue;        // uA: non-zero couplings (RRG.jl:133)
            const int j = Ax[q];
            c.update(j, t + c.gen_wt(c.dE(j)));
        }
        E += d;
        moves += 1;
    }
    for (; ns < P.samples; ++ns) { P.Es[(size_t)ns * P.Rpad + r] = (int32_t)E; nextstep += step; }     // not reached: the loop always emits `samples` samples
    P.E_cur[r] = (int32_t)E;
    P.acc_cur[r] = moves;
    P.t_out[r] = t;
    sf[SF_TIME] = t; sf[SF_NEXT] = nextstep; si[SI_ND] = (long long)c.nd;
}

// ---------------------------------------------------------------------------------------------------
// extremal_opt — tau-EO (src/RRRMC.jl:474-521) on GraphRRG / GraphEA with EOCache{Int,L} (src/DeltaE.jl:412-555): SURVEY.md §8(f)
// rank 4.  One thread per replica.  Classes = values of dE in ascending order (2L - has_zero of them); a rank is drawn from the
// caller's cumulative table ftau (binary search = searchsortedfirst), the class holding it gives a uniform member, which flips.
// RRR stream sub 3.
// ---------------------------------------------------------------------------------------------------
struct EoParams {
    const int32_t* A;        // [N][K]
    const int8_t* J;         // [N][K]
    const double* ftau;      // [N] cumsum(j^-tau)
    uint32_t* spins;         // [R][W]
    uint32_t* cmin;          // [R][W] configuration of minimum energy
    uint8_t* cls;            // [R][N]
    void* sv;                // [R][K2][N]  (uint16_t, or uint32_t when N > 65535)
    void* spos;              // [R][N]
    int32_t* E_cur;          // [Rpad]
    int64_t* stats;          // [R][3]: Emin, itmin, -
    int32_t* Es;             // [nsamples][Rpad]
    uint64_t g0;
    int64_t iters, step;
    uint32_t k0, k1, replica0;
    LevTable lv;
    int N, K, L, has_zero, W, R, Rpad;
    int ftau_lds;            // the launch carries N doubles of dynamic LDS: the rank table is searched there
    SmpState S;
};

// rank table in LDS up to this N (80 KB at N = 10^4: one workgroup per CU, which is what rrr_tpb launches anyway)
constexpr int kEoFtauLdsMaxN = 16384;
// `E < Emin && (Emin = E; copy!(Cmin, C))` (RRRMC.jl:508-512) is a copy of the whole configuration at every new minimum — every other
// iteration while a run descends from its random start.  Cmin differs from C by the flips made since the last minimum: up to kEoPend of
// them are kept per replica and a new minimum toggles those bits of Cmin (atomics without a return value) instead of copying W words; a longer stretch without a minimum
// (or a resumed call, whose pending flips are not kept) falls back to the copy.  Same Cmin, bit for bit.
constexpr int kEoPend = 96;
inline size_t eo_sparse_lds_bytes(int64_t N, bool ftau_lds, unsigned tpb) { return (ftau_lds ? sizeof(double) * (size_t)N : 0) + sizeof(uint32_t) * kEoPend * tpb; }

// SLM = compile-time bound on the levels of allΔE (2, 4 or 8): the class counters are indexed through unrolled selects of that length
template <typename IDX, int SLM>
__global__ __launch_bounds__(kRrrThreads) void eo_sparse_kernel(EoParams P)
{
    extern __shared__ double eo_ftau_lds[];
    // Cmin follows C lazily: the flips since the last minimum wait here (kEoPend per thread, after the rank table), see below
    uint32_t* const pend = reinterpret_cast<uint32_t*>(eo_ftau_lds + (P.ftau_lds ? P.N : 0)) + threadIdx.x;
    const int pstride = blockDim.x;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (P.ftau_lds) {
        // rand_move's binary search (DeltaE.jl:473-507) is ceil(log2 N) DEPENDENT probes of a table every replica shares: from LDS they cost
        // an LDS round trip each instead of an L2 one (profiles/r06/f8_floor.md §3)
        for (int i = threadIdx.x; i < P.N; i += blockDim.x) eo_ftau_lds[i] = P.ftau[i];
        __syncthreads();
    }
    if (r >= P.R) return;
    const int N = P.N, L = P.L, K = P.K, K2 = 2 * P.L - P.has_zero;
    uint32_t* sp = P.spins + (size_t)r * P.W;
    uint32_t* cm = P.cmin + (size_t)r * P.W;
    uint8_t* cls = P.cls + (size_t)r * N;
    IDX* sv = static_cast<IDX*>(P.sv) + (size_t)r * K2 * N;
    IDX* spos = static_cast<IDX*>(P.spos) + (size_t)r * N;
    constexpr int NC = 2 * SLM;
    int t[NC];
    auto tg = [&](int k) __attribute__((always_inline)) { int x = 0;
#pragma unroll
        for (int a = 0; a < NC; ++a) x = k == a ? t[a] : x;
        return x; };
    auto tadd = [&](int k, int d) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < NC; ++a) t[a] = k == a ? t[a] + d : t[a]; };
    auto sbit = [&](int x) { return (int)((sp[x >> 5] >> (x & 31)) & 1u); };
    auto dE_of = [&](int i) {
        const int si = sbit(i);
        int acc = 0;
        for (int q = 0; q < K; ++q) {
            const int sy = sbit(P.A[(size_t)i * K + q]);
            acc += (si == sy) ? (int)P.J[(size_t)i * K + q] : -(int)P.J[(size_t)i * K + q];
        }
        return 2 * acc;
    };
    // findks (DeltaE.jl:412-421), 0-based
    auto klass = [&](int d) __attribute__((always_inline)) {
        const int ad = d < 0 ? -d : d;
        int a = 0;
#pragma unroll
        for (int k = 0; k < SLM; ++k) a = (k < L && P.lv.dElist[k] == ad) ? k : a;
        const int ak = a + 1;
        return (d >= 0 ? ak + L - P.has_zero : L + 1 - ak) - 1;
    };
    // one set move (ArraySet delete!(k0, j) + push!(k1, j), ArraySets.jl:56-76) with the lists' ends followed in registers
    constexpr int TD = 4;                        // list members followed from the end
    TailTrack<NC, TD> tk;
    auto set_move = [&](int j, int k0, int k1, int p) __attribute__((always_inline)) -> int {
        IDX* v0 = sv + (size_t)k0 * N;
        IDX* v1 = sv + (size_t)k1 * N;
        const int n0 = tg(k0);
        // (a real branch: written as a select the compiler reads the list's end speculatively in EVERY set move, and the wait for it —
        // vmcnt(0): stores count too on gfx9 — also drains the previous set move's stores, one round trip per set move)
        int last = tk.last(k0);
        if (__builtin_expect(tk.known(k0) <= 0, 0)) { last = (int)v0[n0 - 1]; asm volatile("" : "+v"(last)); }
        v0[p] = (IDX)last; spos[last] = (IDX)p; tadd(k0, -1);
        tk.popped(k0, n0, p, last);
        const int n1 = tg(k1);
        v1[n1] = (IDX)j; spos[j] = (IDX)n1; tadd(k1, 1);
        tk.pushed(k1, j);
        cls[j] = (uint8_t)k1;
        return last;
    };
    auto gather_apply = [&](auto km, int move) __attribute__((always_inline)) {
        constexpr int KM = decltype(km)::value;
        int sj[KM + 1], s0[KM + 1], s1[KM + 1], sp_[KM + 1];
        bool live[KM + 1];
        // (the flip: its word is requested WITH the row — read on its own, before everything else, it is a round trip of its own)
        const uint32_t wm = sp[move >> 5];
        int y[KM], cj[KM];
#pragma unroll
        for (int q = 0; q < KM; ++q) { const size_t e = (size_t)move * K + (q < K ? q : 0); y[q] = P.A[e]; cj[q] = (int)P.J[e]; }
        sp[move >> 5] = wm ^ (1u << (move & 31));
        bool val[KM];
#pragma unroll
        for (int q = 0; q < KM; ++q) val[q] = q < K && !(q > 0 && y[q] == y[q - 1]) && !(P.lv.skip_zero && cj[q] == 0);     // uA (EA.jl:158, RRG.jl:133)
        int yy[KM][KM], cc[KM][KM];
        uint32_t wown[KM];
#pragma unroll
        for (int q = 0; q < KM; ++q) {
            const int j = val[q] ? y[q] : move;              // (a slot without a neighbour reads the moved spin's entries: harmless, never used)
            sj[q] = j;
            s0[q] = cls[j]; sp_[q] = (int)spos[j]; wown[q] = sp[j >> 5];
#pragma unroll
            for (int k = 0; k < KM; ++k) { const size_t e = (size_t)j * K + (k < K ? k : 0); yy[q][k] = P.A[e]; cc[q][k] = k < K ? (int)P.J[e] : 0; }
        }
        s0[KM] = cls[move]; sp_[KM] = (int)spos[move]; sj[KM] = move;
        uint32_t wnb[KM][KM];
#pragma unroll
        for (int q = 0; q < KM; ++q)
#pragma unroll
            for (int k = 0; k < KM; ++k) wnb[q][k] = sp[yy[q][k] >> 5];
        // the ends of the lists that may lose a site: their last TD members (independent of the spin words: same trip)
        typename TailTrack<NC, TD>::Ends ee[KM + 1];
        int en[KM + 1];
#pragma unroll
        for (int q = 0; q <= KM; ++q) {
            const int n = tg(s0[q]);
            en[q] = n;
#pragma unroll
            for (int i = 0; i < TD; ++i) ee[q].v[i] = (int)sv[(size_t)s0[q] * N + (n > i ? n - 1 - i : 0)];
        }
#pragma unroll
        for (int q = 0; q < KM; ++q) {
            const int sjb = (int)((wown[q] >> (sj[q] & 31)) & 1u);
            int acc = 0;
#pragma unroll
            for (int k = 0; k < KM; ++k) { const int sy = (int)((wnb[q][k] >> (yy[q][k] & 31)) & 1u); acc += (sjb == sy) ? cc[q][k] : -cc[q][k]; }
            s1[q] = klass(2 * acc);
            live[q] = val[q] && s0[q] != s1[q];
        }
        {   // the moved spin: its delta_energy changed sign, so its class is the mirror of the one it held (dE = 0 keeps its class)
            const int k0 = s0[KM];
            const int a0 = k0 < L ? L - 1 - k0 : k0 - L + P.has_zero;
            s1[KM] = k0 < L ? ((a0 == 0 && P.has_zero) ? k0 : L + a0 - P.has_zero) : L - 1 - a0;
            live[KM] = s1[KM] != k0;
        }
        tk.clear();
#pragma unroll
        for (int q = 0; q <= KM; ++q) tk.note(s0[q], en[q], ee[q]);
#pragma unroll
        for (int q = 0; q <= KM; ++q) {
            if (!live[q]) continue;
            const int p = sp_[q];
            const int last = set_move(sj[q], s0[q], s1[q], p);
#pragma unroll
            for (int q2 = 0; q2 <= KM; ++q2)
                if (q2 > q && live[q2] && sj[q2] == last) sp_[q2] = p;          // a later site of this move that was relocated into the freed slot
        }
    };
    long long* const si = P.S.si + (size_t)r * kSmpI;
    long long E, Emin, itmin, ns = 0;
    int npend = 0;
    bool pend_over = false;                  // more flips since the last minimum than `pend` holds: the next minimum copies
    if (P.S.resume) {
        // a resumed call: the EOCache (classes, member order), E, Emin / Cmin / itmin of the run carry on (RRRMC.jl:486-516)
#pragma unroll
        for (int k = 0; k < NC; ++k) t[k] = k < K2 ? (int)si[SI_T0 + k] : 0;
        E = P.E_cur[r]; Emin = si[SI_EMIN]; itmin = si[SI_ITMIN];
        pend_over = true;
    } else {
    long long n = 0;
#pragma unroll
    for (int k = 0; k < NC; ++k) t[k] = 0;
    for (int i = 0; i < N; ++i) {
        const int d = dE_of(i);
        n -= d / 2;
        const int k = klass(d);
        cls[i] = (uint8_t)k;
        const int tk0 = tg(k);
        sv[(size_t)k * N + tk0] = (IDX)i;
        spos[i] = (IDX)tk0;
        tadd(k, 1);
    }
    E = n / 2; Emin = E; itmin = 0;
    for (int w = 0; w < P.W; ++w) cm[w] = sp[w];
    }
    const double z = P.ftau[N - 1];
    const uint32_t rep = P.replica0 + (uint32_t)r;
    long long next_sample = P.S.samp0;       // iterations k * step of the run: a counter instead of a 64-bit modulo per iteration
    for (long long it = 1; it <= P.iters; ++it) {
        if (it == next_sample) { next_sample += P.step; P.Es[(size_t)ns * P.Rpad + r] = (int32_t)E; ns += 1; }
        const uint64_t g = P.g0 + (uint64_t)it;
        const Philox4 o = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), rep, TAG_RRR | (3u << 8), P.k0, P.k1);
        // rand_move: DeltaE.jl:473-507
        const double rr = (1 - (double)((((uint64_t)o.w[0] << 32) | o.w[1]) >> 11) * 0x1.0p-53) * z;
        int lo = 0, hi = N;
        if (P.ftau_lds) {
            // searchsortedfirst, two levels per LDS round trip: the probe of this level and both candidates of the next are read together
            // (same comparisons against the same entries as one probe per level)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const int ml = (lo + mid) >> 1, mr = (mid + 1 + hi) >> 1;
                const double xm = eo_ftau_lds[mid], xl = eo_ftau_lds[ml < N ? ml : N - 1], xr = eo_ftau_lds[mr < N ? mr : N - 1];
                asm volatile("" :: "v"(xm), "v"(xl), "v"(xr));             // (all three here: sunk into the branches below they are two round trips again)
                if (xm < rr) { lo = mid + 1; if (lo < hi) { if (xr < rr) lo = mr + 1; else hi = mr; } }
                else { hi = mid; if (lo < hi) { if (xl < rr) lo = ml + 1; else hi = ml; } }
            }
        } else { while (lo < hi) { const int mid = (lo + hi) >> 1; if (P.ftau[mid] < rr) lo = mid + 1; else hi = mid; } }
        int rank = lo + 1;
        if (rank > N) rank = N;
        int k = 0, tt = t[0];
#pragma unroll
        for (int a = 1; a < NC; ++a) { const bool more = rank > tt; k = more ? a : k; tt += more ? t[a] : 0; }
        const int a = k < L ? L - 1 - k : k - L + P.has_zero;                        // index into allΔE
        int dEa = 0;
#pragma unroll
        for (int b = 0; b < SLM; ++b) dEa = a == b ? P.lv.dElist[b] : dEa;
        const int dE = k < L ? -dEa : dEa;
        const int move = sv[(size_t)k * N + (int)mulhi64(((uint64_t)o.w[2] << 32) | o.w[3], (uint64_t)tg(k))];
        // apply_move!: DeltaE.jl:509-541
        if (npend < kEoPend) { pend[npend * pstride] = (uint32_t)move; npend += 1; } else pend_over = true;
        if (K > 6) sp[move >> 5] ^= 1u << (move & 31);        // (the gathered forms flip with their first stage)
        if (K <= 6) {
            // the K + 1 sites gathered stage by stage, every load of a stage issued before any is used, then the set moves in the reference's
            // order on the gathered values — SparseChain::apply_move's scheme (above); slot q < KM = neighbour q, slot KM = the moved spin
            if (K <= 3) gather_apply(std::integral_constant<int, 3>{}, move);
            else if (K <= 4) gather_apply(std::integral_constant<int, 4>{}, move);
            else gather_apply(std::integral_constant<int, 6>{}, move);
        } else {
        tk.clear();
        const int32_t* Ax = P.A + (size_t)move * K;
        for (int q = 0; q <= K; ++q) {
            if (q < K && q > 0 && Ax[q] == Ax[q - 1]) continue;                      // uA: repeats removed (EA.jl:158)
            if (q < K && P.lv.skip_zero && P.J[(size_t)move * K + q] == 0) continue;  // uA: non-zero couplings (RRG.jl:133)
            const int j = q < K ? Ax[q] : move;
            const int k0 = cls[j], k1 = klass(dE_of(j));
            if (k0 == k1) continue;
            set_move(j, k0, k1, (int)spos[j]);
        }
        }
        E += dE;
        if (E < Emin) {
            Emin = E; itmin = P.S.it0 + it;
            if (pend_over) { for (int w = 0; w < P.W; ++w) cm[w] = sp[w]; }
            else for (int q = 0; q < npend; ++q) { const uint32_t x = pend[q * pstride]; atomicXor(&cm[x >> 5], 1u << (x & 31)); }    // (no value comes back: nothing waits)
            npend = 0; pend_over = false;
        }
    }
    P.E_cur[r] = (int32_t)E;
    P.stats[(size_t)r * 3] = Emin; P.stats[(size_t)r * 3 + 1] = itmin; P.stats[(size_t)r * 3 + 2] = P.iters;
#pragma unroll
    for (int k = 0; k < NC; ++k) if (k < K2) si[SI_T0 + k] = t[k];
    si[SI_EMIN] = Emin; si[SI_ITMIN] = itmin;
}

// ---------------------------------------------------------------------------------------------------
// standardMC (src/RRRMC.jl:81-127) on a stand-alone GraphRRG{ET,LEV,K} / GraphEA{ET,LEV,2D} with levels other than (-1, 1)
// (ET = Int or DFloat64): no bit-plane threshold table exists for arbitrary levels, so one thread per
// replica like the samplers above — common site (SITE stream), rand53 < exp(-beta dE) (ACCEPT_F64 stream), the energy tracked in
// integer level units (what DFloat64's + does, src/DFloats.jl:29-30); `-beta * dE` promotes to Float64 as (units * mul) / div.
// iters = 0 gives energy(X, C).
// ---------------------------------------------------------------------------------------------------
struct LevStdParams {
    const int32_t* A;        // [N][K]
    const int8_t* J;         // [N][K] level units
    uint32_t* spins;         // [R][W]
    int32_t* E_cur;          // [Rpad]
    int64_t* acc_cur;        // [Rpad]
    int32_t* Es;             // [nsamples][Rpad]
    double beta, lev_div;
    long long lev_mul;
    uint64_t g0;
    int64_t iters, step;
    uint32_t k0, k1, replica0;
    int N, K, W, R, Rpad;
};

__global__ __launch_bounds__(kRrrThreads) void lev_standard_kernel(LevStdParams P)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= P.R) return;
    const int N = P.N, K = P.K;
    uint32_t* sp = P.spins + (size_t)r * P.W;
    auto sbit = [&](int x) { return (int)((sp[x >> 5] >> (x & 31)) & 1u); };
    auto dE_of = [&](int i) {
        const int si = sbit(i);
        int acc = 0;
        for (int q = 0; q < K; ++q) {
            const int sy = sbit(P.A[(size_t)i * K + q]);
            acc += (si == sy) ? (int)P.J[(size_t)i * K + q] : -(int)P.J[(size_t)i * K + q];
        }
        return 2 * acc;
    };
    long long n = 0;
    for (int i = 0; i < N; ++i) n -= dE_of(i) / 2;
    long long E = n / 2, accepted = 0, ns = 0;
    const uint32_t rep = P.replica0 + (uint32_t)r;
    long long next_sample = P.step;          // iterations k * step: a counter instead of a 64-bit modulo per iteration
    for (long long it = 1; it <= P.iters; ++it) {
        if (it == next_sample) { next_sample += P.step; P.Es[(size_t)ns * P.Rpad + r] = (int32_t)E; ns += 1; }
        const uint64_t g = P.g0 + (uint64_t)it;
        const int move = (int)site_of(P.k0, P.k1, g, (uint32_t)N);
        const int d = dE_of(move);
        const double x = -P.beta * ((double)((long long)d * P.lev_mul) / P.lev_div);
        const bool acc = (x >= 0.0) || (rand53(P.k0, P.k1, g, rep) < det_exp(x));        // RRRMC.jl:39
        if (acc) { sp[move >> 5] ^= 1u << (move & 31); E += d; accepted += 1; }
    }
    P.E_cur[r] = (int32_t)E;
    P.acc_cur[r] = accepted;
}

// bit-sliced [G][N] words (bit = replica & 31)  <->  replica-contiguous [R][W] words (bit = site & 31)
__global__ __launch_bounds__(256) void rrsp_spins_in_kernel(const uint32_t* __restrict__ bs, uint32_t* __restrict__ spins, int N, int W, int R)
{
    const int w = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (w >= W || r >= R) return;
    uint32_t word = 0u;
    for (int b = 0; b < 32; ++b) {
        const int x = 32 * w + b;
        if (x < N) word |= ((bs[(size_t)(r >> 5) * N + x] >> (r & 31)) & 1u) << b;
    }
    spins[(size_t)r * W + w] = word;
}
__global__ __launch_bounds__(256) void rrsp_spins_out_kernel(const uint32_t* __restrict__ spins, uint32_t* __restrict__ bs, int N, int W, int R)
{
    const int x = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
    if (x >= N) return;
    uint32_t word = 0u;
    for (int b = 0; b < 32; ++b) {
        const int r = g * 32 + b;
        if (r < R) word |= ((spins[(size_t)r * W + (x >> 5)] >> (x & 31)) & 1u) << b;
    }
    bs[(size_t)g * N + x] = word;
}

}  // namespace rrrmc
