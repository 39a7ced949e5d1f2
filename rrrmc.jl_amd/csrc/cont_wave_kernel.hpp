// rrrMC(X::SingleGraph) (src/RRRMC.jl:149-219) and bklMC (:311-359) on the Float64 sparse models GraphRRGNormal / GraphEANormal over
// DeltaECacheCont + DynamicSampler (src/DeltaE.jl:297-410, src/DynamicSamplers.jl) — ONE WAVEFRONT PER REPLICA (round 4).
//
// cont_sparse_kernel (cont_kernels.hpp) runs one thread per replica: every step of the chain is a dependent global-memory round trip
// of one lane — getel walks ceil(log2 N) tree levels one load at a time, every setindex! is a read-modify-write per level, refresh! is
// N x levels of them — and 4096 replicas are 256 wavefronts of 16 lanes: 25 us per iteration and replica (profiles/r03/f8_kernels_summary.txt).
// Here the chain of a replica is executed wave-uniformly and the lanes take what is independent inside a step:
//   * the partial-sum tree of the DynamicSampler (DynamicSamplers.jl:54-98: node (lev, k) = sum of its LEFT subtree, a running Float64 sum
//     updated by every setindex!) is split: the top levs - 4 levels live in LDS (2^(levs-4) - 1 nodes: 8 KiB at N = 10^4), the bottom four in
//     memory as one 128-byte block of 15 nodes per group of 16 elements (heap order).  getel (:130-152) reads six levels per LDS round trip
//     (lane = node of the subtree below the current node) and walks them through v_readlane; the last four levels are one memory access.
//   * apply_move! (DeltaE.jl:376-410): lane 0 is the moved spin, lane q its q-th neighbour: local fields (update_cache! with the exact undo
//     record, RRG.jl:576-617), delta_energy, prior = min(1, exp(-beta dE)) (the fixed-order exponential, 40 dependent operations — once for
//     all of them), old weights and the tree blocks of all of them are requested together.  The setindex! calls of a move are applied in
//     the reference's order: z and the LDS levels sequentially (LDS executes in order), the memory blocks by lanes (element, level) with
//     elements that share a block resolved in registers — every node receives its additions in the order of the calls.
//   * refresh! (:84-98; every max(N, 100) calls of setindex!, or when getel lands on an empty element): sum(v) and the tree from scratch
//     are sequential Float64 sums by construction — one pass over v, 64 elements per coalesced load; the lanes own the nodes whose subtree
//     lies inside their 1/64 of the elements, the top six levels and z are accumulated wave-uniformly.
// Bit-identical to cont_sparse_kernel and the oracle: every Float64 value (fields, z, tree nodes, E, acc_rate) is produced by the reference's
// sequence of operations.  Scope: modes 0 / 1 on the pure Float64 models, no repeated neighbours in a row of A (GraphEANormal with L = 2
// keeps the thread kernel), 64 <= N <= 2^16 and 32 KiB of LDS per replica at most.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cont_kernels.hpp"
#include "sk_block_kernel.hpp"      // det_exp_v, sk_readlane_f64

namespace rrrmc {

constexpr int kCwBottom = 4;        // tree levels kept in memory blocks (16 elements per block)

inline size_t cw_lds_bytes(int64_t W, int levs) { return (((size_t)W * 4 + 7) & ~(size_t)7) + sizeof(double) * ((size_t)1 << (levs - kCwBottom)); }

// refresh! (DynamicSamplers.jl:84-98): z = sum(v) and the partial-sum tree from scratch — sequential Float64 sums, left to right, by
// construction.  One call per max(N, 100) calls of setindex!: kept out of line in two small functions (the chain's loop stays small, and a
// callee only uses caller-saved registers — every other block of eight above v40 — so a large callee would set the kernel's register
// count and its occupancy); the state they need is handed over explicitly.  Tree layout: levels < levs - 4 in LDS in heap order, the
// others in 16-double blocks in memory.
__device__ __forceinline__ void cw_node_store(double* l_ps, double* __restrict__ gps, int LT, int lev, int k, double x)
{
    if (lev < LT) l_ps[(1 << lev) - 1 + k] = x;
    else { const int w = lev - LT; gps[(size_t)(k >> w) * 16 + ((1 << w) - 1) + (k & ((1 << w) - 1))] = x; }
}
// (a) the levels whose nodes' subtrees lie inside one lane's chunk of C = N2 / 64 elements (lev >= 6): lane t owns elements [t C, (t + 1) C)
__device__ __noinline__ void cw_refresh_low(const double* __restrict__ v, double* l_ps, double* __restrict__ gps, int N, int N2, int levs, int lane)
{
    const int LT = levs - kCwBottom, C = N2 >> 6, lgC = levs - 6, base = lane * C;
    double acc[10];                                       // levels 6 + j, j < lgC <= 10  (N <= 2^16: the host's LDS bound)
#pragma unroll
    for (int j = 0; j < 10; ++j) acc[j] = 0.0;
#pragma unroll 1
    for (int e = 0; e < C; ++e) {
        const int i = base + e;
        const double vi = i < N ? v[i] : 0.0;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            if (j < lgC) {
                const int lev = 6 + j, span = 1 << (levs - lev);              // elements under a node of this level
                if (((i >> (levs - 1 - lev)) & 1) == 0) acc[j] += vi;
                if (((i + 1) & (span - 1)) == 0) { cw_node_store(l_ps, gps, LT, lev, i >> (levs - lev), acc[j]); acc[j] = 0.0; }
            }
        }
    }
}
// (b) z = sum(v) left to right, and the top six levels: wave-uniform, the elements broadcast out of one coalesced load per 64 (elements
//     beyond N are zeros: they change no sum, and every node is stored).  Returns z.
__device__ __noinline__ double cw_refresh_top(const double* __restrict__ v, double* l_ps, double* __restrict__ gps, int N, int N2, int levs, int lane)
{
    const int LT = levs - kCwBottom;
    double zz = 0.0, top[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (levs >= 12) {
        // the bit a top level tests (levs - 1 - l >= 6) is the same for the 64 elements of an aligned batch and a node ends with a batch:
        // per batch one multiplier per level (1.0 = the elements belong to the node's left subtree, 0.0 = they do not: x * 1 + t and
        // x * 0 + t are exact), per element one addition for z and six fused multiply-adds, no test
#pragma unroll 1
        for (int b0 = 0; b0 < N2; b0 += 64) {
            if (b0 < N) {
                const double mine = b0 + lane < N ? v[b0 + lane] : 0.0;
                double m[6];
#pragma unroll
                for (int l = 0; l < 6; ++l) m[l] = ((b0 >> (levs - 1 - l)) & 1) == 0 ? 1.0 : 0.0;
#pragma unroll 4
                for (int t = 0; t < 64; ++t) {
                    const double vi = sk_readlane_f64(mine, t);
                    zz += vi;
#pragma unroll
                    for (int l = 0; l < 6; ++l) top[l] = __builtin_fma(vi, m[l], top[l]);
                }
            }
#pragma unroll
            for (int l = 0; l < 6; ++l)
                if (((b0 + 64) & ((1 << (levs - l)) - 1)) == 0) { if (lane == 0) cw_node_store(l_ps, gps, LT, l, b0 >> (levs - l), top[l]); top[l] = 0.0; }
        }
        return zz;
    }
#pragma unroll 1
    for (int b0 = 0; b0 < N2; b0 += 64) {
        const double mine = b0 + lane < N ? v[b0 + lane] : 0.0;
#pragma unroll 1
        for (int t = 0; t < 64; ++t) {
            const int i = b0 + t;
            const double vi = sk_readlane_f64(mine, t);
            zz += vi;
#pragma unroll
            for (int l = 0; l < 6; ++l) {
                if (((i >> (levs - 1 - l)) & 1) == 0) top[l] += vi;
                if (((i + 1) & ((1 << (levs - l)) - 1)) == 0) { if (lane == 0) cw_node_store(l_ps, gps, LT, l, i >> (levs - l), top[l]); top[l] = 0.0; }
            }
        }
    }
    return zz;
}
__device__ __forceinline__ double cw_refresh(const double* __restrict__ v, double* l_ps, double* __restrict__ gps, int N, int N2, int levs, int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    cw_refresh_low(v, l_ps, gps, N, N2, levs, lane);
    const double zz = cw_refresh_top(v, l_ps, gps, N, N2, levs, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    __syncthreads();
    return zz;
}

__global__ __launch_bounds__(64) void cont_wave_kernel(ContParams P)
{
    extern __shared__ uint32_t cw_lds[];
    const int lane = (int)threadIdx.x, r = (int)blockIdx.x;
    const int N = P.N, K = P.K, levs = P.levs, LT = levs - kCwBottom, N2 = P.N2;
    uint32_t* l_sp = cw_lds;
    double* l_ps = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(cw_lds) + (((size_t)P.W * 4 + 7) & ~(size_t)7));      // heap order: (1 << lev) - 1 + k
    uint32_t* g_sp = P.spins + (size_t)r * P.W;
    double* lf = P.lf + (size_t)r * N;
    double* v = P.v + (size_t)r * N2;
    double* gps = P.ps + (size_t)r * N2;                      // bottom blocks: 16 doubles per group of 16 elements (15 used)
    const uint32_t rep = P.replica0 + (uint32_t)r;
    auto prior = [&](double x) -> double { return prior_of(x); };                            // DeltaE.jl:297 (the constants of the exponential are not
                                                                                             // kept in registers: 34 of them, across the whole loop)
    auto uni = [](bool c) -> bool { return __builtin_amdgcn_ballot_w64(c) != 0ull; };
    auto bcd = [](double x, int l) -> double { return sk_readlane_f64(x, l); };
    auto sbit = [&](int x) -> int { return (int)((l_sp[x >> 5] >> (x & 31)) & 1u); };

    for (int i = lane; i < P.W; i += 64) l_sp[i] = g_sp[i];
    __syncthreads();

    double z = 0.0;
    long long trefresh = 0;
    const long long tlim = N > 100 ? N : 100;

    auto refresh = [&]() __attribute__((always_inline)) {
        z = cw_refresh(v, l_ps, gps, N, N2, levs, lane);
        trefresh = 0;
    };

    // ---- getel (DynamicSamplers.jl:130-152); -1 = "Unrecoverable loss of precision" --------------------------------------------------------
    // walks `w` levels of a subtree held one node per lane in heap order; returns the offset of the leaf reached
    auto walk = [&](double pv, int w, double& x) __attribute__((always_inline)) -> int {
        int cur = 0;
        for (int s = 0; s < w; ++s) {
            const double p = bcd(pv, cur);
            if (x > p) { x -= p; cur = 2 * cur + 2; } else cur = 2 * cur + 1;
        }
        return cur - ((1 << w) - 1);
    };
    const int hj = 31 - __builtin_clz((unsigned)lane + 1u), hm = lane + 1 - (1 << hj);       // this lane's node of a subtree: level hj, index hm
    const bool nbl = lane >= 1 && lane <= K;
    int pf_y = 0;                            // getel's prefetch for the element it returns: lane q's neighbour, its coupling, the element's field
    double pf_J = 0.0, pf_lf0 = 0.0;
    auto getel = [&](double x) __attribute__((always_inline)) -> int {
        x = bcd(x, 0);                       // (every lane computed the same draw: tell the compiler, or every walk step becomes a loop over lanes)
        for (;;) {
            x *= z;
            int k = 0;
            for (int l0 = 0; l0 < LT; l0 += 6) {
                const int w = LT - l0 < 6 ? LT - l0 : 6;
                const double pv = lane < (1 << w) - 1 ? l_ps[(1 << (l0 + hj)) - 1 + (k << hj) + hm] : 0.0;
                k = (k << w) + walk(pv, w, x);
            }
            {
                const double pv = lane < 15 ? gps[(size_t)k * 16 + lane] : 0.0;
                k = (k << kCwBottom) + walk(pv, kCwBottom, x);
            }
            k = __builtin_amdgcn_readfirstlane(k);
            // what the move needs first — its row of the graph, its local field — is requested together with the weight the emptiness
            // test reads: one memory round trip instead of two
            const int kk = k < N ? k : 0;
            const double vk = v[kk];
            pf_y = nbl ? P.A[(size_t)kk * K + (lane - 1)] : kk;
            pf_J = nbl ? P.J[(size_t)kk * K + (lane - 1)] : 0.0;
            pf_lf0 = lf[kk];
            const bool empty = k >= N || vk == 0;
            if (uni(empty)) {
                if (!(trefresh > 0)) return -1;
                refresh();
                continue;
            }
            return k;
        }
    };

    // ---- setindex! of the elements held by lanes q0 .. q1 - 1 (site s, new weight x, old weight vo), in lane order -------------------------
    // (the caller guarantees that no refresh! falls among them)
    auto sets = [&](int q0, int q1, int s, double x, double vo) __attribute__((always_inline)) {
        const bool mine = lane >= q0 && lane < q1;
        const double d = x - vo;
        if (mine) v[s] = x;
        // memory blocks: lane (q, w), w < 4; requested first, resolved below
        const int qb = q0 + (lane >> 2), wb = lane & 3;
        const bool bl = qb < q1;
        const int sb = __shfl(s, bl ? qb : q0);
        const double db = __shfl(d, bl ? qb : q0);
        const int levb = LT + wb;
        const bool actb = bl && ((sb >> (levs - 1 - levb)) & 1) == 0;
        const int addrb = (sb >> kCwBottom) * 16 + ((1 << wb) - 1) + ((sb >> (kCwBottom - wb)) & ((1 << wb) - 1));
        double nodeb = actb ? gps[addrb] : 0.0;
        for (int q = q0; q < q1; ++q) {
            const double dq = bcd(d, q);
            const int sq = __builtin_amdgcn_readlane(s, q);
            z += dq;
            if (lane < LT && ((sq >> (levs - 1 - lane)) & 1) == 0) l_ps[(1 << lane) - 1 + (sq >> (levs - lane))] += dq;      // (LDS executes in order)
        }
        // a node shared by several elements receives their differences in element order; the last of them stores the sum
        bool last = actb;
        for (int q = q0; q < q1; ++q) {
            const int lq = (q - q0) * 4 + wb;                                  // the lane holding (q, this level)
            const int a2 = __shfl(addrb, lq);
            const bool on2 = __shfl((int)actb, lq) != 0;
            const double d2 = __shfl(db, lq);
            if (actb && on2 && a2 == addrb) {
                if (q <= qb) nodeb += d2;
                else last = false;
            }
        }
        if (last) gps[addrb] = nodeb;
        trefresh += q1 - q0;
    };

    // the K + 1 setindex! of a move (lane 0 .. K), refresh! wherever the reference's counter asks for it (setindex! tests it first, :163-165)
    auto sets_of_move = [&](int s, double x, double vo) __attribute__((always_inline)) {
        for (int q0 = 0; q0 <= K;) {
            if (trefresh >= tlim) refresh();
            const long long room = tlim - trefresh;
            const int n = (long long)(K + 1 - q0) < room ? K + 1 - q0 : (int)room;
            sets(q0, q0 + n, s, x, vo);
            q0 += n;
        }
    };

    // ---- the graph's cache: lane 0 = the moved spin, lane q = neighbour q - 1 (RRG.jl:576-617 with the undo record) ------------------------
    int mlast = -1;
    double undo = 0.0;                       // lane q >= 1: the record of neighbour q - 1; lane 0: undo[K]
    // what flip(move) would leave in lf[] for this lane's site (and the record it would keep); commit = also make it the state
    struct Flip { int s; double lf_old, lf_new, undo_new; };
    auto flip_values = [&](int move, bool fresh) __attribute__((always_inline)) -> Flip {       // fresh: pf_lf0 is still lf[move]
        Flip f{};
        const int y = pf_y;
        const double Jq = pf_J;
        f.s = lane <= K ? y : 0;
        f.lf_old = lane == 0 && fresh ? pf_lf0 : lane <= K ? lf[f.s] : 0.0;
        if (mlast == move) {                                                   // the exact undo: swap with the record
            f.lf_new = lane == 0 ? -f.lf_old : undo;
            f.undo_new = lane == 0 ? -undo : f.lf_old;
        } else {
            const int sx = sbit(move) ^ 1;                                     // the moved spin AFTER the flip
            const double c = (sx ^ sbit(f.s)) ? -4.0 : 4.0;
            f.lf_new = lane == 0 ? -f.lf_old : f.lf_old - c * Jq;
            f.undo_new = f.lf_old;
        }
        return f;
    };
    auto commit_flip = [&](int move, const Flip& f) __attribute__((always_inline)) {
        if (lane == 0) l_sp[move >> 5] ^= 1u << (move & 31);
        if (lane <= K) lf[f.s] = f.lf_new;
        undo = f.undo_new;
        mlast = move;
    };
    // apply_move! (DeltaE.jl:376-410); returns c = z / z'
    auto apply_move = [&](int move, bool fresh) __attribute__((always_inline)) -> double {
        const Flip f = flip_values(move, fresh);
        const double vo = lane <= K ? v[f.s] : 0.0;
        commit_flip(move, f);
        const double z0 = z;
        const double dn = -f.lf_new;                                           // delta_energy = -lfields (RRG.jl:619-625); DeltaECacheCont.ΔEs is not
        const double x = prior(P.beta * dn);                                   // stored: it always equals -lfields, which is what is read instead
        sets_of_move(f.s, x, vo);
        return z0 / z;
    };

    double* const sf = P.S.sf + (size_t)r * kSmpF;
    long long* const sq = P.S.si + (size_t)r * kSmpI;
    double* const g_top = P.ps_top + (size_t)r * ((size_t)N2 >> kCwBottom);        // the LDS levels of the tree between resumed calls
    const bool resume = P.S.resume != 0;
    double E = 0.0;
    if (resume) {
        // a resumed call: fields, weights and the bottom blocks of the tree are in HBM where the previous call left them; the LDS levels,
        // the undo record (one entry per lane) and the scalars of the chain come back from the slabs
        for (int i = lane; i < (1 << LT) - 1; i += 64) l_ps[i] = g_top[i];
        z = sf[SF_Z]; trefresh = sq[SI_TREF]; mlast = (int)sq[SI_MLAST];
        undo = lane <= K ? sf[SF_UNDO + lane] : 0.0;
        E = sf[SF_E];
        __syncthreads();
    } else {
    // ---- energy(X, C) (RRG.jl:546-574), DeltaECacheCont (DeltaE.jl:304-313) ---------------------------------------------------------------
    double E1 = 0.0;
    for (int i0 = 0; i0 < N; i0 += 64) {
        const int i = i0 + lane;
        double fl = 0.0;
        if (i < N) {
            const int sx = 2 * sbit(i) - 1;
            for (int q = 0; q < K; ++q) {
                const int sy = 2 * sbit(P.A[(size_t)i * K + q]) - 1;
                fl = fl - P.J[(size_t)i * K + q] * (double)sx * (double)sy;
            }
            lf[i] = 2.0 * fl;
            const double d = -(2.0 * fl);
            v[i] = prior(P.beta * d);
        }
        const int nb = N - i0 < 64 ? N - i0 : 64;
        for (int t = 0; t < nb; ++t) E1 = E1 + bcd(fl, t);
    }
    for (int i = N + lane; i < N2; i += 64) v[i] = 0.0;
    E = E1 / 2;
    refresh();
    }

    long long accepted = 0, second = 0, ns = 0, itdone = 0;
    int bad = 0;
    double acc_rate = resume ? sf[SF_ACC] : 0.5;
    long long bk_it = 0, bk_next = P.step, bk_m = 0, bk_limit = P.iters;
    if (P.mode == 0) {
        long long next_sample = P.S.samp0;
        for (long long it = 1; it <= P.iters; ++it) {
            if (it == next_sample) { next_sample += P.step; if (lane == 0) P.Es[(size_t)ns * P.Rp + r] = E; ns += 1; }
            const uint64_t g = P.g0 + (uint64_t)it;
            const Philox4 o = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), rep, TAG_RRR, P.k0, P.k1);
            const double u0 = (double)((((uint64_t)o.w[0] << 32) | o.w[1]) >> 11) * 0x1.0p-53;
            const Philox4 o2 = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), rep, TAG_RRR | (1u << 8), P.k0, P.k1);
            const double u1 = bcd((double)((((uint64_t)o2.w[0] << 32) | o2.w[1]) >> 11) * 0x1.0p-53, 0);
            bool acc = false;
            const int move = getel(u0);                                         // rand_move: DeltaE.jl:327-333
            if (move < 0) { bad = 1; break; }
            const double dE = -pf_lf0;                                          // = cache.ΔEs[move] (DeltaE.jl:327-333)
            if (acc_rate < P.staged_thr) {
                second += 1;
                const double z0 = z;
                // compute_staged! (DeltaE.jl:357-374): flip, look, flip back — the cache keeps the record of the second flip
                const Flip f = flip_values(move, true);
                const double vo = lane <= K ? v[f.s] : 0.0;
                const double sdE = -f.lf_new, spv = prior(P.beta * sdE);
                const double diff = spv - vo;
                undo = f.lf_new;                                               // after flip + flip back the record holds the flipped values
                mlast = move;
                double zp = z;                                                  // compute_reverse_probabilities!: DeltaE.jl:345-355
                for (int q = 0; q <= K; ++q) zp += bcd(diff, q);
                if (zp < 2.2250738585072014e-308) zp = 2.2250738585072014e-308;
                if (zp > (double)N) zp = (double)N;
                if (u1 < z0 / zp) {
                    // apply_staged!: the third flip takes the record (the same values), then the K + 1 setindex!
                    Flip g2 = f;
                    g2.undo_new = lane == 0 ? -f.lf_new : f.lf_old;
                    commit_flip(move, g2);
                    sets_of_move(f.s, spv, vo);
                    E += dE; accepted += 1; acc = true;
                }
            } else {
                for (int pass = 0; pass < 2; ++pass) {                          // the second apply_move! is the undo of a rejected move
                    const double cc = apply_move(move, pass == 0);
                    if (pass == 0 && u1 < cc) { E += dE; accepted += 1; acc = true; break; }
                }
            }
            acc_rate = acc_rate * (1 - P.lambda) + (acc ? 1.0 : 0.0) * P.lambda;
        }
        itdone = P.iters;
    } else {
        // (a resumed call continues the run's loop with `iters` more iterations allowed; the pending move is drawn again: same draw)
        if (resume) { bk_it = sq[SI_IT]; bk_next = sq[SI_NEXT]; bk_m = sq[SI_M]; bk_limit += sq[SI_LIMIT]; }
        long long it = bk_it, nextstep = bk_next, m = bk_m;
        const long long limit = bk_limit;
        const bool over = resume && nextstep > limit && nextstep > P.step;      // (a resumed call whose allowance does not reach the run's next sample point makes no move: the reference's loop ends with its last sample, RRRMC.jl:340-343)
        while (!over && it < limit) {
            const uint64_t g = P.g0 + (uint64_t)(m + 1);
            const Philox4 o3 = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), rep, TAG_RRR | (2u << 8), P.k0, P.k1);
            const double us = bcd((double)((((uint64_t)o3.w[0] << 32) | o3.w[1]) >> 11) * 0x1.0p-53, 0);
            double b = z / (double)N;                                           // rand_skip: DeltaE.jl:319-325
            if (b < 2.2250738585072014e-308) b = 2.2250738585072014e-308;
            if (b > 1.0) b = 1.0;
            const double skipf = floor(det_log1p(-us) / det_log1p(-b));
            const long long skip = skipf >= 9.0e18 ? (long long)9.0e18 : (long long)skipf;
            const Philox4 o = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), rep, TAG_RRR, P.k0, P.k1);
            const int move = getel((double)((((uint64_t)o.w[0] << 32) | o.w[1]) >> 11) * 0x1.0p-53);
            if (move < 0) { bad = 1; break; }
            const double dE = -pf_lf0;
            bool out = false;
            while (it + skip + 1 >= nextstep) {
                if (nextstep > limit) { out = true; break; }        // (fewer iterations allowed than `step`: the sample point lies beyond this call — no sample, no move)
                if (lane == 0) P.Es[(size_t)ns * P.Rp + r] = E;
                ns += 1;
                nextstep += P.step;
                if (nextstep > limit) { out = true; break; }
            }
            if (out) break;
            apply_move(move, true);                                             // apply_step_bkl!: RRRMC.jl:294-295
            m += 1;
            it += skip + 1;
            E += dE;
            accepted += 1;
        }
        second = accepted; itdone = it;
        bk_it = it; bk_next = nextstep; bk_m = m;
    }
    __syncthreads();
    for (int i = lane; i < P.W; i += 64) g_sp[i] = l_sp[i];
    for (int i = lane; i < (1 << LT) - 1; i += 64) g_top[i] = l_ps[i];
    if (lane <= K) sf[SF_UNDO + lane] = undo;
    if (lane == 0) {
        P.E_cur[r] = E;
        P.stats[(size_t)r * 3] = accepted; P.stats[(size_t)r * 3 + 1] = second; P.stats[(size_t)r * 3 + 2] = itdone;
        P.t_out[r] = 0.0;
        P.status[r] = bad;
        sf[SF_E] = E; sf[SF_Z] = z; sf[SF_ACC] = acc_rate;
        sq[SI_TREF] = trefresh; sq[SI_MLAST] = mlast; sq[SI_ND] = 0;
        sq[SI_IT] = bk_it; sq[SI_NEXT] = bk_next; sq[SI_M] = bk_m; sq[SI_LIMIT] = bk_limit;
    }
}

}  // namespace rrrmc
