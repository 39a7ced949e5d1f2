/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see philox_contract.h).  Never linked into, imported by or
 * called from the product path; the HIP library fails loudly instead of falling back to this.
 *
 * Plain-C, single-thread restatement of the reference's hot path (carlobaldassi/RRRMC.jl v2.2.0).
 * Each function cites the reference file:line it follows (paths relative to /root/reference).
 * Indices are 0-based here (the reference is 1-based); spins are Julia-BitVector chunks:
 * bit of site x = chunk[x >> 6] >> (x & 63) & 1   (src/Common.jl:15-23, Base.get_chunks_id).
 *
 * PARITY: unpinned against Julia's RNG streams (see philox_contract.h); pinned against the
 * reference's RNG-free invariants by tests/test_oracle_*.py: tracked E == energy(X,C) at every sample
 * (test/runtests.jl:12-20), cache == recomputation (src/graphs/RRG.jl:229-231), allΔE tables
 * (RRG.jl:267-281, EA.jl:293), closed-form toy models (graphs/TwoSpin.jl, ThreeSpin.jl, Ising1D.jl)
 * and the exact Boltzmann law for tiny N (src/RRRMC.jl:528-543).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "philox_contract.h"

#define ORC_API __attribute__((visibility("default")))

/* ---------------------------------------------------------------------------------------------
 * Raw generator access (KATs)
 * ------------------------------------------------------------------------------------------- */
ORC_API void orc_philox(const uint32_t *ctr, const uint32_t *key, uint32_t *out) { orc_philox4x32_10(ctr, key, out); }
ORC_API int64_t orc_site_of(uint64_t seed, uint64_t g, int64_t N) { return orc_site(seed, g, N); }
ORC_API uint64_t orc_accept_uniform(uint64_t seed, uint64_t g, uint32_t replica) { return orc_accept_u64(seed, g, replica); }
ORC_API int orc_accept_less(uint64_t seed, uint64_t g, uint32_t replica, uint64_t T) { return orc_accept_lt(seed, g, replica, T); }
ORC_API uint64_t orc_threshold(double p, int *always) { return orc_threshold64(p, always); }

/* Config(N) with random bits: src/Interface.jl:24-28 (rand!(BitVector)), INIT stream. */
ORC_API void orc_init_config(uint64_t seed, uint32_t replica, int64_t N, uint64_t *chunks)
{
    int64_t nch = (N + 63) / 64;
    memset(chunks, 0, (size_t)nch * sizeof(uint64_t));
    for (int64_t x = 0; x < N; ++x)
        if (orc_init_spin(seed, replica, (uint64_t)x)) chunks[x >> 6] |= 1ull << (x & 63);
}

/* ---------------------------------------------------------------------------------------------
 * Hooks.  Every sampler of the reference calls `hook(it, X, C, accepted, E)` at each sample and ends the chain when it returns false
 * (src/RRRMC.jl:104-108 standardMC, :184-188 and :253-257 rrrMC, :339-343 bklMC with `nextstep`, :402-406 wtmMC with the sample's
 * global time and num_moves, :499-503 extremal_opt with (it, X, C, E, Emin)).  The restatement does the same, inside its loops: a test
 * registers a callback and sees, at every sample, the iteration (or sample time), the configuration, the count and the energy — what the
 * HIP library must hand to ITS hooks from a run cut into resumed calls.  No callback: the samplers run to the end as before.
 * ------------------------------------------------------------------------------------------- */
typedef int (*orc_hook_fn)(double it, const uint64_t *chunks, int64_t nchunks, int64_t accepted, double E, double Emin, void *user);
static __thread orc_hook_fn g_hook = 0;
static __thread void *g_hook_user = 0;
ORC_API void orc_set_hook(orc_hook_fn f, void *user) { g_hook = f; g_hook_user = user; }
#define ORC_HOOK(it, acc, E, Emin) (g_hook ? g_hook((double)(it), chunks, (N + 63) / 64, (int64_t)(acc), (double)(E), (double)(Emin), g_hook_user) : 1)

static inline int spin_bit(const uint64_t *chunks, int64_t x) { return (int)((chunks[x >> 6] >> (x & 63)) & 1u); }
/* unsafe_bitflip!: src/Common.jl:15-23 */
static inline void bitflip(uint64_t *chunks, int64_t x) { chunks[x >> 6] ^= 1ull << (x & 63); }

/* ---------------------------------------------------------------------------------------------
 * Graph construction
 * ------------------------------------------------------------------------------------------- */

/* gen_RRG: src/graphs/RRG.jl:26-69.  Bollobas pairing model with whole-attempt restart on a self
 * loop or a repeated edge; A[x] = ascending neighbour list (findall on the adjacency bit row, :64).
 * Draws: GRAPH stream, one per pair (`j = rand(1:(l-1))`, :45).  Returns the number of attempts
 * used (>= 1), or -1 after 100000 failed attempts (:34), -2 on bad arguments (:27-28). */
ORC_API int orc_gen_rrg(int64_t N, int64_t K, uint64_t seed, int32_t *A)
{
    if (K < 1 || N < 1 || ((N * K) & 1)) return -2;
    int64_t NK = N * K;
    int64_t *rp = (int64_t *)malloc((size_t)NK * sizeof(int64_t));
    /* adjacency "bit rows" B (RRG.jl:32) kept as short neighbour lists: same membership test */
    int32_t *adj = (int32_t *)malloc((size_t)NK * sizeof(int32_t));
    int32_t *deg = (int32_t *)malloc((size_t)N * sizeof(int32_t));
    uint64_t ndraw = 0;
    int ok = 0, attempt;
    for (attempt = 1; attempt <= 100000; ++attempt) {
        for (int64_t q = 0; q < NK; ++q) rp[q] = q + 1;   /* rp[:] = 1:NK */
        memset(deg, 0, (size_t)N * sizeof(int32_t));
        int64_t len = NK;
        int again = 0;
        while (len > 0) {
            int64_t l = len;
            /* j = rand(1:(l-1)) */
            int64_t j = 1 + (int64_t)orc_mulhi64(orc_stream_u64(seed, ORC_TAG_GRAPH, ndraw++), (uint64_t)(l - 1));
            int64_t rv1 = rp[--len];                        /* pop! */
            int64_t t = rp[j - 1]; rp[j - 1] = rp[len - 1]; rp[len - 1] = t;
            int64_t rv2 = rp[--len];                        /* pop! */
            int64_t v1 = (rv1 - 1) % N, v2 = (rv2 - 1) % N; /* mod1(rv, N) - 1 */
            int dup = (v1 == v2);
            for (int32_t q = 0; q < deg[v1] && !dup; ++q) dup = (adj[v1 * K + q] == (int32_t)v2);
            if (dup) { again = 1; break; }
            adj[v1 * K + deg[v1]++] = (int32_t)v2;
            adj[v2 * K + deg[v2]++] = (int32_t)v1;
        }
        if (again) continue;
        ok = 1;
        break;
    }
    if (ok) {
        for (int64_t x = 0; x < N; ++x) {                   /* findall(b): ascending */
            int32_t *a = adj + x * K;
            for (int64_t p = 1; p < K; ++p) {
                int32_t v = a[p]; int64_t q = p - 1;
                while (q >= 0 && a[q] > v) { a[q + 1] = a[q]; --q; }
                a[q + 1] = v;
            }
            for (int64_t k = 0; k < K; ++k) A[x * K + k] = a[k];
        }
    }
    free(rp); free(adj); free(deg);
    return ok ? attempt : -1;
}

/* gen_EA: src/graphs/EA.jl:24-43.  Periodic L^D lattice, column-major linear index, each site
 * pushes its +1 neighbour along every dimension on both ends, then every list is sorted (:40).
 * For L == 2 each neighbour therefore appears twice. */
ORC_API int orc_gen_ea(int64_t L, int64_t D, int32_t *A)
{
    if (L < 2 || D < 1) return -2;
    int64_t N = 1; for (int64_t d = 0; d < D; ++d) N *= L;
    int64_t twoD = 2 * D;
    int64_t *cnt = (int64_t *)calloc((size_t)N, sizeof(int64_t));
    for (int64_t x = 0; x < N; ++x) {
        int64_t stride = 1;
        for (int64_t d = 0; d < D; ++d) {
            int64_t c = (x / stride) % L;
            int64_t y = x - c * stride + ((c + 1) % L) * stride;
            A[x * twoD + cnt[x]++] = (int32_t)y;
            A[y * twoD + cnt[y]++] = (int32_t)x;
            stride *= L;
        }
    }
    for (int64_t x = 0; x < N; ++x) {           /* sort! each list (insertion sort, 2D entries) */
        int32_t *a = A + x * twoD;
        for (int64_t p = 1; p < twoD; ++p) {
            int32_t v = a[p]; int64_t q = p - 1;
            while (q >= 0 && a[q] > v) { a[q + 1] = a[q]; --q; }
            a[q + 1] = v;
        }
    }
    free(cnt);
    return 0;
}

/* gen_J: src/graphs/RRG.jl:71-96 and src/graphs/EA.jl:45-71.  One draw `rand(vLEV)` per bond
 * visited from its smaller endpoint, in (x, k) order; the value is stored at J[x][k] and at the first
 * still-unfilled slot of J[y].  COUPLING stream. */
ORC_API int orc_gen_couplings(int64_t N, int64_t K, const int32_t *A, uint64_t seed,
                              int64_t nlev, const int32_t *lev, int32_t *J)
{
    const int32_t sentinel = INT32_MIN;
    for (int64_t q = 0; q < N * K; ++q) J[q] = sentinel;
    uint64_t ndraw = 0;
    for (int64_t x = 0; x < N; ++x)
        for (int64_t k = 0; k < K; ++k) {
            int64_t y = A[x * K + k];
            if (x < y) {
                int32_t Jxy = lev[orc_mulhi64(orc_stream_u64(seed, ORC_TAG_COUPLING, ndraw++), (uint64_t)nlev)];
                if (J[x * K + k] != sentinel) return -1;
                J[x * K + k] = Jxy;
                int64_t l = 0;
                while (l < K && J[y * K + l] != sentinel) ++l;
                if (l == K) return -1;
                J[y * K + l] = Jxy;
            }
        }
    for (int64_t q = 0; q < N * K; ++q) if (J[q] == sentinel) return -1;
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * Sparse integer models GraphRRG / GraphEA  (ET = Int)
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int64_t N, K;
    const int32_t *A, *J;
    int64_t *lfields, *lfields_last;   /* LocalFields: src/Common.jl:27-36 */
    int64_t move_last;                 /* -1 = none (reference: 0) */
    int ea_form;                       /* 0: GraphRRG's update_cache!, 1: GraphEA's (walks the de-duplicated uA) */
} sparse_t;

enum { ORC_FORM_RRG = 0, ORC_FORM_EA = 1 };

/* energy: src/graphs/RRG.jl:164-189 (EA twin src/graphs/EA.jl:195-222). Also (re)builds the cache. */
static int64_t sparse_energy(sparse_t *X, const uint64_t *s)
{
    int64_t n = 0;
    for (int64_t x = 0; x < X->N; ++x) {
        int64_t sx = 2 * spin_bit(s, x) - 1;
        int64_t lf = 0;
        for (int64_t k = 0; k < X->K; ++k) {
            int64_t y = X->A[x * X->K + k];
            int64_t sy = 2 * spin_bit(s, y) - 1;
            lf -= (int64_t)X->J[x * X->K + k] * sx * sy;
        }
        n += lf;
        X->lfields[x] = 2 * lf;
    }
    n /= 2;                              /* every bond was seen from both ends: exact */
    X->move_last = -1;
    memset(X->lfields_last, 0, (size_t)X->N * sizeof(int64_t));
    return n;
}

/* delta_energy: src/graphs/RRG.jl:236-244, src/graphs/EA.jl:266-275 */
static inline int64_t sparse_delta_energy(const sparse_t *X, int64_t move) { return -X->lfields[move]; }

/* uA[move] of GraphEA (src/graphs/EA.jl:158): the sorted neighbour tuple with repeats removed.  A is
 * sorted, so a repeat is an entry equal to its predecessor (for L = 2 every neighbour appears twice). */
static inline int ea_is_repeat(const int32_t *Ax, int64_t k) { return k > 0 && Ax[k] == Ax[k - 1]; }

/* update_cache!, called after the bit flip.
 * GraphRRG form: src/graphs/RRG.jl:191-234.  GraphEA form: src/graphs/EA.jl:224-264 — same arithmetic, but the
 * undo buffer is saved/swapped once per DISTINCT neighbour (uA), which is what makes the L = 2 lattice
 * (two parallel bonds per neighbour) work. */
static void sparse_update_cache(sparse_t *X, const uint64_t *s, int64_t move)
{
    const int32_t *Ax = X->A + move * X->K;
    if (X->move_last == move) {          /* undo fast path: RRG.jl:198-209, EA.jl:231-241 */
        for (int64_t k = 0; k < X->K; ++k) {
            if (X->ea_form && ea_is_repeat(Ax, k)) continue;
            int64_t y = Ax[k];
            int64_t t = X->lfields[y]; X->lfields[y] = X->lfields_last[y]; X->lfields_last[y] = t;
        }
        X->lfields[move] = -X->lfields[move];
        X->lfields_last[move] = -X->lfields_last[move];
        return;
    }
    const int32_t *Jx = X->J + move * X->K;
    int sx = spin_bit(s, move);
    if (X->ea_form)                      /* EA.jl:245-248 */
        for (int64_t k = 0; k < X->K; ++k)
            if (!ea_is_repeat(Ax, k)) X->lfields_last[Ax[k]] = X->lfields[Ax[k]];
    for (int64_t k = 0; k < X->K; ++k) {
        int64_t y = Ax[k];
        int64_t sxy = 1 - 2 * (sx ^ spin_bit(s, y));
        int64_t lfy = X->lfields[y];
        if (!X->ea_form) X->lfields_last[y] = lfy;      /* RRG.jl:219 */
        X->lfields[y] = lfy - 4 * sxy * (int64_t)Jx[k];
    }
    int64_t lfm = X->lfields[move];
    X->lfields_last[move] = lfm;
    X->lfields[move] = -lfm;
    X->move_last = move;
}

ORC_API int64_t orc_sparse_energy(int64_t N, int64_t K, const int32_t *A, const int32_t *J,
                                  const uint64_t *chunks, int64_t *lfields_out)
{
    sparse_t X = {N, K, A, J, NULL, NULL, -1, 0};
    X.lfields = (int64_t *)malloc((size_t)N * sizeof(int64_t));
    X.lfields_last = (int64_t *)malloc((size_t)N * sizeof(int64_t));
    int64_t E = sparse_energy(&X, chunks);
    if (lfields_out) memcpy(lfields_out, X.lfields, (size_t)N * sizeof(int64_t));
    free(X.lfields); free(X.lfields_last);
    return E;
}

/* accept(x) = x >= 0 || rand() < exp(x): src/RRRMC.jl:39, with the ACCEPT stream for rand(). */
static inline int accept_move(double x, uint64_t seed, uint64_t g, uint32_t replica)
{
    if (x >= 0) return 1;
    int always;
    uint64_t T = orc_threshold64(exp(x), &always);
    if (always) return 1;
    return orc_accept_lt(seed, g, replica, T);
}

/*
 * standardMC: src/RRRMC.jl:81-127 for a sparse integer model, one chain.
 *   form            ORC_FORM_RRG (GraphRRG) or ORC_FORM_EA (GraphEA): which update_cache! is followed
 *   chunks  in/out  configuration (C0 is resumed and mutated in place, :93)
 *   it0             iterations already consumed from this seed's streams (the reference continues the
 *                   global RNG when seed <= 0, :89; here the caller passes the stream position)
 *   Es      out     one energy per `step` iterations, sampled BEFORE the move of iteration k*step (:104-108)
 *   sites_out/flips_out (optional, length iters): the attempted site and whether it was accepted
 * Returns the number of samples written.
 */
ORC_API int64_t orc_standard_mc_sparse(int form, int64_t N, int64_t K, const int32_t *A, const int32_t *J,
                                       double beta, int64_t iters, int64_t step,
                                       uint64_t seed, uint64_t it0, uint32_t replica,
                                       uint64_t *chunks, int64_t *Es, int64_t *accepted_out,
                                       int64_t *lfields_out, int32_t *sites_out, uint8_t *flips_out)
{
    sparse_t X = {N, K, A, J, NULL, NULL, -1, form};
    X.lfields = (int64_t *)malloc((size_t)N * sizeof(int64_t));
    X.lfields_last = (int64_t *)malloc((size_t)N * sizeof(int64_t));
    int64_t E = sparse_energy(&X, chunks);           /* :95 */
    int64_t accepted = 0, nsamp = 0;
    for (int64_t it = 1; it <= iters; ++it) {        /* :100-119 */
        if (it % step == 0) Es[nsamp++] = E;         /* :104-108 (hook is the caller's business) */
        uint64_t g = it0 + (uint64_t)it;
        int64_t i = orc_site(seed, g, N);            /* :113 */
        int64_t dE = sparse_delta_energy(&X, i);     /* :114 */
        int acc = accept_move(-beta * (double)dE, seed, g, replica);   /* :115 */
        if (sites_out) sites_out[it - 1] = (int32_t)i;
        if (flips_out) flips_out[it - 1] = (uint8_t)acc;
        if (!acc) continue;
        bitflip(chunks, i);                          /* :116 -> Interface.jl:89-92 */
        sparse_update_cache(&X, chunks, i);
        E += dE;                                     /* :117 */
        accepted += 1;                               /* :118 */
    }
    if (accepted_out) *accepted_out = accepted;
    if (lfields_out) memcpy(lfields_out, X.lfields, (size_t)N * sizeof(int64_t));
    free(X.lfields); free(X.lfields_last);
    return nsamp;
}

/* R independent chains (replica ids replica0 .. replica0+R-1), configurations stored replica-major:
 * chunks[r * nch + c].  Es is [R][nsamples]. */
ORC_API int64_t orc_standard_mc_sparse_batch(int form, int64_t N, int64_t K, const int32_t *A, const int32_t *J,
                                             double beta, int64_t iters, int64_t step,
                                             uint64_t seed, uint64_t it0, uint32_t replica0, int64_t R,
                                             uint64_t *chunks, int64_t *Es, int64_t *accepted)
{
    int64_t nch = (N + 63) / 64, nsamp = iters / step, got = 0;
    for (int64_t r = 0; r < R; ++r)
        got = orc_standard_mc_sparse(form, N, K, A, J, beta, iters, step, seed, it0, replica0 + (uint32_t)r,
                                     chunks + r * nch, Es + r * nsamp, accepted + r, NULL, NULL, NULL);
    return got;
}

/* allΔE for ±J sparse models: src/graphs/RRG.jl:262-281 (sorted set of 2*|sum of K terms ±1|). */
ORC_API int64_t orc_all_delta_e_pm1(int64_t K, int64_t *out)
{
    int64_t n = 0;
    for (int64_t m = (K & 1); m <= K; m += 2) out[n++] = 2 * m;
    return n;
}

/* allΔE for GraphRRG{Int,LEV,K} / GraphEA{Int,LEV,twoD} with any integer levels (RRG.jl:268-281, EA.jl:295-309): the set
 * of sums of K terms +-l, l in LEV, mapped to 2|x|, sorted.  Returns the number of levels (<= cap) or -1. */
ORC_API int64_t orc_all_delta_e(int64_t K, const int32_t *lev, int64_t nlev, int64_t *out, int64_t cap)
{
    int64_t amax = 0;
    for (int64_t l = 0; l < nlev; ++l) { int64_t a = lev[l] < 0 ? -(int64_t)lev[l] : lev[l]; if (a > amax) amax = a; }
    const int64_t span = K * amax;                              /* reachable sums lie in [-span, span] */
    uint8_t *cur = (uint8_t *)calloc((size_t)(2 * span + 1), 1), *nxt = (uint8_t *)calloc((size_t)(2 * span + 1), 1);
    cur[span] = 1;
    for (int64_t n = 0; n < K; ++n) {
        memset(nxt, 0, (size_t)(2 * span + 1));
        for (int64_t v = -span; v <= span; ++v)
            if (cur[v + span])
                for (int64_t l = 0; l < nlev; ++l) {
                    if (v + lev[l] >= -span && v + lev[l] <= span) nxt[v + lev[l] + span] = 1;
                    if (v - lev[l] >= -span && v - lev[l] <= span) nxt[v - lev[l] + span] = 1;
                }
        uint8_t *t = cur; cur = nxt; nxt = t;
    }
    int64_t n = 0;
    for (int64_t a = 0; a <= span; ++a)
        if (cur[span + a] || cur[span - a]) { if (n == cap) { n = -1; break; } out[n++] = 2 * a; }
    free(cur); free(nxt);
    return n;
}

/* Level units.  The reference's level type is either Int or DFloat64 — "an integer in disguise" (src/DFloats.jl:11-36): the
 * Int64 t = round(x * 10^5) whose Float64 value is t / 10^5 (:23,26), with +, -, ==, < and Integer * DFloat64 acting on t.
 * All level arithmetic here is done in integer *units*: value = (units * mul) / div, with (mul, div) = (1, 1.0) for Int levels
 * and (g, 10^5) for DFloat64 levels, g = gcd of the levels' t (so that units fit the narrow coupling tables). */
static inline double lev_to_f64(int64_t units, int64_t mul, double div) { return (double)(units * mul) / div; }
/* Level specification of a stand-alone GraphRRG{ET,LEV,K} / GraphEA{ET,LEV,2D} (RRG.jl:116-162, EA.jl:138-193) with levels other
 * than (-1, 1): lev[nlev] in units, (mul, div) as above.  NULL = the +-J graphs (allΔE by RRG.jl:262-266 / EA.jl:293). */
typedef struct { const int32_t *lev; int64_t nlev; int64_t mul; double div; } levspec_t;

/* discretize(x, LEV): Common.jl:38-49 — the nearest level (the first one on ties) and the residual x - d */
ORC_API void orc_discretize_scaled(const double *x, int64_t n, const int32_t *lev, int64_t nlev, int64_t mul, double div,
                                   int32_t *d_out, double *r_out)
{
    for (int64_t q = 0; q < n; ++q) {
        int32_t d = lev[0];
        double r = x[q] - lev_to_f64(d, mul, div);
        for (int64_t l = 1; l < nlev; ++l) {
            double r1 = x[q] - lev_to_f64(lev[l], mul, div);
            if (fabs(r1) < fabs(r)) { d = lev[l]; r = r1; }
        }
        d_out[q] = d;
        r_out[q] = r;
    }
}
ORC_API void orc_discretize(const double *x, int64_t n, const int32_t *lev, int64_t nlev, int32_t *d_out, double *r_out)
{
    orc_discretize_scaled(x, n, lev, nlev, 1, 1.0, d_out, r_out);
}

/* ---------------------------------------------------------------------------------------------
 * Dense Gaussian SK model GraphSKNormal (ET = Float64): src/graphs/SK.jl:170-297
 * ------------------------------------------------------------------------------------------- */

/* n-th standard normal of the GAUSS stream: Box-Muller on two 53-bit uniforms of draw n >> 1 (cos branch for even n,
 * sin branch for odd n).  [replaces `randn(N)`, SK.jl:171] */
static double gauss_draw(uint64_t seed, uint64_t n)
{
    uint32_t w[4];
    uint64_t blk = n >> 1;
    orc_draw(seed, (uint32_t)blk, (uint32_t)(blk >> 32), 0u, ORC_TAG_GAUSS, w);
    uint64_t a = ((uint64_t)w[0] << 32) | w[1], b = ((uint64_t)w[2] << 32) | w[3];
    double u1 = ((double)(a >> 11) + 1.0) * 0x1.0p-53;       /* (0, 1] */
    double u2 = (double)(b >> 11) * 0x1.0p-53;               /* [0, 1) */
    double rad = sqrt(-2.0 * log(u1)), ang = 6.283185307179586476925286766559 * u2;
    return (n & 1u) ? rad * sin(ang) : rad * cos(ang);
}

ORC_API double orc_gauss(uint64_t seed, uint64_t n) { return gauss_draw(seed, n); }
ORC_API double orc_exp(double x) { return orc_det_exp(x); }
ORC_API double orc_rand53_of(uint64_t seed, uint64_t g, uint32_t replica) { return orc_rand53(seed, g, replica); }

/* gen_J_gauss: SK.jl:170-179.  Row i = randn(N) scaled by 1/sqrt(N) (rmul!), then zero diagonal and the upper
 * triangle copied onto the lower one.  Draw index of J[i][j] before symmetrisation: i*N + j. */
ORC_API void orc_gen_sk_gauss(int64_t N, uint64_t seed, double *J)
{
    const double scale = 1.0 / sqrt((double)N);
    for (int64_t i = 0; i < N; ++i)
        for (int64_t j = 0; j < N; ++j) J[i * N + j] = gauss_draw(seed, (uint64_t)(i * N + j)) * scale;
    for (int64_t i = 0; i < N; ++i) {
        J[i * N + i] = 0.0;
        for (int64_t j = i + 1; j < N; ++j) J[j * N + i] = J[i * N + j];
    }
}

typedef struct {
    int64_t N;
    const double *J;
    double *lfields, *lfields_last;    /* swapped wholesale by the undo path, SK.jl:248 */
    int64_t move_last;
} skn_t;

/* energy: SK.jl:212-237 (sequential sums in j order; also rebuilds the cache) */
static double skn_energy(skn_t *X, const uint64_t *s)
{
    double n = 0.0;
    for (int64_t i = 0; i < X->N; ++i) {
        const double *Ji = X->J + i * X->N;
        int si = spin_bit(s, i);
        double lf = 0.0;
        for (int64_t j = 0; j < X->N; ++j) lf += (double)(1 - 2 * (si ^ spin_bit(s, j))) * Ji[j];
        X->lfields[i] = 2 * lf;
        n -= lf;
    }
    n /= 2;
    X->move_last = -1;
    memset(X->lfields_last, 0, (size_t)X->N * sizeof(double));
    return n;
}

/* update_cache!: SK.jl:239-276, called after the bit flip */
static void skn_update_cache(skn_t *X, const uint64_t *s, int64_t move)
{
    if (X->move_last == move) {          /* SK.jl:247-250: swap the two arrays, move_last stays */
        double *t = X->lfields; X->lfields = X->lfields_last; X->lfields_last = t;
        return;
    }
    const double *Ji = X->J + move * X->N;
    int si = spin_bit(s, move);
    double lfm = X->lfields[move];
    for (int64_t j = 0; j < X->N; ++j) {
        double Jsij = (double)(1 - 2 * (si ^ spin_bit(s, j))) * Ji[j];
        double lfj = X->lfields[j];
        X->lfields_last[j] = lfj;
        X->lfields[j] = lfj + 4 * Jsij;
    }
    X->lfields_last[move] = lfm;
    X->lfields[move] = -lfm;
    X->move_last = move;
}

ORC_API double orc_skn_energy(int64_t N, const double *J, const uint64_t *chunks, double *lfields_out)
{
    skn_t X = {N, J, NULL, NULL, -1};
    X.lfields = (double *)malloc((size_t)N * sizeof(double));
    X.lfields_last = (double *)malloc((size_t)N * sizeof(double));
    double E = skn_energy(&X, chunks);
    if (lfields_out) memcpy(lfields_out, X.lfields, (size_t)N * sizeof(double));
    free(X.lfields); free(X.lfields_last);
    return E;
}

/* standardMC (src/RRRMC.jl:81-127) on GraphSKNormal, one chain; delta_energy = +lfields[move] (SK.jl:278-284). */
ORC_API int64_t orc_standard_mc_skn(int64_t N, const double *J, double beta, int64_t iters, int64_t step,
                                    uint64_t seed, uint64_t it0, uint32_t replica,
                                    uint64_t *chunks, double *Es, int64_t *accepted_out, double *lfields_out)
{
    skn_t X = {N, J, NULL, NULL, -1};
    X.lfields = (double *)malloc((size_t)N * sizeof(double));
    X.lfields_last = (double *)malloc((size_t)N * sizeof(double));
    double E = skn_energy(&X, chunks);
    int64_t accepted = 0, nsamp = 0;
    for (int64_t it = 1; it <= iters; ++it) {
        if (it % step == 0) Es[nsamp++] = E;
        uint64_t g = it0 + (uint64_t)it;
        int64_t i = orc_site(seed, g, N);
        double dE = X.lfields[i];
        double x = -beta * dE;
        int acc = (x >= 0) || (orc_rand53(seed, g, replica) < orc_det_exp(x));      /* RRRMC.jl:39 */
        if (!acc) continue;
        bitflip(chunks, i);
        skn_update_cache(&X, chunks, i);
        E += dE;
        accepted += 1;
    }
    if (accepted_out) *accepted_out = accepted;
    if (lfields_out) memcpy(lfields_out, X.lfields, (size_t)N * sizeof(double));
    free(X.lfields); free(X.lfields_last);
    return nsamp;
}

/* ---------------------------------------------------------------------------------------------
 * Float64-coupling sparse models GraphRRGNormal / GraphEANormal (SimpleGraph{Float64}):
 * src/graphs/RRG.jl:503-627, src/graphs/EA.jl:534-680
 * ------------------------------------------------------------------------------------------- */

/* gen_J(Float64, N, A) do randn() end — RRG.jl:71-96 / EA.jl:45-71 with f = randn: one GAUSS-stream draw per bond
 * visited from its smaller endpoint in (x, k) order, stored at J[x][k] and at the first unfilled slot of J[y]. */
ORC_API int orc_gen_couplings_gauss(int64_t N, int64_t K, const int32_t *A, uint64_t seed, double *J)
{
    uint8_t *filled = (uint8_t *)calloc((size_t)(N * K), 1);
    uint64_t ndraw = 0;
    int rc = 0;
    for (int64_t x = 0; x < N && !rc; ++x)
        for (int64_t k = 0; k < K; ++k) {
            int64_t y = A[x * K + k];
            if (x < y) {
                double Jxy = gauss_draw(seed, ndraw++);
                if (filled[x * K + k]) { rc = -1; break; }
                J[x * K + k] = Jxy; filled[x * K + k] = 1;
                int64_t l = 0;
                while (l < K && filled[y * K + l]) ++l;
                if (l == K) { rc = -1; break; }
                J[y * K + l] = Jxy; filled[y * K + l] = 1;
            }
        }
    for (int64_t q = 0; q < N * K && !rc; ++q) if (!filled[q]) rc = -1;
    free(filled);
    return rc;
}

typedef struct {
    int64_t N, K;
    const int32_t *A;
    const double *J;
    double *lfields, *lfields_last;
    int64_t move_last;
    int ea_form;
    /* DoubleGraph view (Graph{RRG,EA}NormalDiscretized under the continuous-energy samplers): the inner DiscrGraph X0 whose
     * integer delta_energy is added — promoted to Float64 — to the residual one (RRG.jl:493-497); NULL for the plain Float64 graphs */
    sparse_t *X0;
    int64_t lev_mul;
    double lev_div;
    const uint64_t *cur_s;   /* the live configuration (GraphQT's delta_energy reads spins: QT.jl:86-103) */
    void *Q;                 /* a GraphQuant (quant_t *) behind the continuous-energy samplers instead of (A, J): see cont_quant_* */
} spf_t;

/* energy: RRG.jl:546-574 / EA.jl:584-611 */
static double spf_energy(spf_t *X, const uint64_t *s)
{
    double E1 = 0.0;
    for (int64_t x = 0; x < X->N; ++x) {
        int sx = 2 * spin_bit(s, x) - 1;
        double lf = 0.0;
        for (int64_t k = 0; k < X->K; ++k) {
            int sy = 2 * spin_bit(s, X->A[x * X->K + k]) - 1;
            lf -= X->J[x * X->K + k] * (double)sx * (double)sy;
        }
        E1 += lf;
        X->lfields[x] = 2 * lf;
    }
    E1 /= 2;
    X->move_last = -1;
    memset(X->lfields_last, 0, (size_t)X->N * sizeof(double));
    return E1;
}

/* update_cache!: RRG.jl:576-617 (GraphRRGNormal) and EA.jl:613-653 (GraphEANormal: de-duplicated uA), after the flip */
static void spf_update_cache(spf_t *X, const uint64_t *s, int64_t move)
{
    const int32_t *Ax = X->A + move * X->K;
    const double *Jx = X->J + move * X->K;
    const int64_t K = X->K;
    if (X->move_last == move) {
        for (int64_t k = 0; k < K; ++k) {
            if (X->ea_form && ea_is_repeat(Ax, k)) continue;
            int64_t y = Ax[k];
            double t = X->lfields[y]; X->lfields[y] = X->lfields_last[y]; X->lfields_last[y] = t;
        }
        X->lfields[move] = -X->lfields[move];
        X->lfields_last[move] = -X->lfields_last[move];
        return;
    }
    int sx = spin_bit(s, move);
    if (X->ea_form) {
        for (int64_t k = 0; k < K; ++k) if (!ea_is_repeat(Ax, k)) X->lfields_last[Ax[k]] = X->lfields[Ax[k]];
        for (int64_t k = 0; k < K; ++k) {
            int64_t y = Ax[k];
            int sxy = 1 - 2 * (sx ^ spin_bit(s, y));
            X->lfields[y] -= (double)(4 * sxy) * Jx[k];
        }
    } else {
        for (int64_t k = 0; k < K; ++k) {
            int64_t y = Ax[k];
            int sxy = 1 - 2 * (sx ^ spin_bit(s, y));
            double lfy = X->lfields[y];
            X->lfields_last[y] = lfy;
            X->lfields[y] = lfy - (double)(4 * sxy) * Jx[k];
        }
    }
    double lfm = X->lfields[move];
    X->lfields_last[move] = lfm;
    X->lfields[move] = -lfm;
    X->move_last = move;
}

ORC_API double orc_spf_energy(int form, int64_t N, int64_t K, const int32_t *A, const double *J, const uint64_t *chunks, double *lfields_out)
{
    spf_t X = {N, K, A, J, NULL, NULL, -1, form, NULL, 1, 1.0, NULL, NULL};
    X.lfields = (double *)malloc((size_t)N * sizeof(double));
    X.lfields_last = (double *)malloc((size_t)N * sizeof(double));
    double E = spf_energy(&X, chunks);
    if (lfields_out) memcpy(lfields_out, X.lfields, (size_t)N * sizeof(double));
    free(X.lfields); free(X.lfields_last);
    return E;
}

/* standardMC (src/RRRMC.jl:81-127) on GraphRRGNormal / GraphEANormal, one chain; delta_energy = -lfields[move]
 * (RRG.jl:619-625).  SITE stream shared by all replicas, ACCEPT_F64 uniform per replica. */
ORC_API int64_t orc_standard_mc_spf(int form, int64_t N, int64_t K, const int32_t *A, const double *J, double beta, int64_t iters,
                                    int64_t step, uint64_t seed, uint64_t it0, uint32_t replica,
                                    uint64_t *chunks, double *Es, int64_t *accepted_out, double *lfields_out)
{
    spf_t X = {N, K, A, J, NULL, NULL, -1, form, NULL, 1, 1.0, NULL, NULL};
    X.lfields = (double *)malloc((size_t)N * sizeof(double));
    X.lfields_last = (double *)malloc((size_t)N * sizeof(double));
    double E = spf_energy(&X, chunks);
    int64_t accepted = 0, nsamp = 0;
    for (int64_t it = 1; it <= iters; ++it) {
        if (it % step == 0) Es[nsamp++] = E;
        uint64_t g = it0 + (uint64_t)it;
        int64_t i = orc_site(seed, g, N);
        double dE = -X.lfields[i];
        double x = -beta * dE;
        int acc = (x >= 0) || (orc_rand53(seed, g, replica) < orc_det_exp(x));      /* RRRMC.jl:39 */
        if (!acc) continue;
        bitflip(chunks, i);
        spf_update_cache(&X, chunks, i);
        E += dE;
        accepted += 1;
    }
    if (accepted_out) *accepted_out = accepted;
    if (lfields_out) memcpy(lfields_out, X.lfields, (size_t)N * sizeof(double));
    free(X.lfields); free(X.lfields_last);
    return nsamp;
}

/* The pattern form of delta_energy on GraphRRGNormal / GraphEANormal, used by the library's opt-in FAST standardMC (bit-sliced
 * replicas, no stored fields): with u_k = "bond k of site i is unsatisfied" (sigma_i sigma_y J_ik < 0),
 *   delta_energy(i) = 2 sigma_i sum_k J_ik sigma_y = 2 * sum_k (u_k ? -|J_ik| : +|J_ik|),   summed in the order k = 1..K
 * — the value the reference reads from its incrementally updated cache (RRG.jl:619-625), up to that cache's rounding (documented
 * deviation of the fast mode: ~1e-16 relative).  Complementing every u_k negates the value exactly. */
ORC_API double orc_spf_pattern_delta(int64_t K, const double *Jrow, uint32_t u)
{
    double s = 0.0;
    for (int64_t k = 0; k < K; ++k) { const double a = fabs(Jrow[k]); s += ((u >> k) & 1u) ? -a : a; }
    return 2.0 * s;
}

/* standardMC (src/RRRMC.jl:81-127) on GraphRRGNormal / GraphEANormal in the library's FAST mode (one chain): SITE stream, delta_energy
 * in the pattern form above, accept(x) = x >= 0 || u < exp(x) (:39) with the 64-bit uniform of the ACCEPT stream (bit planes, as the
 * +-J models) against ceil(exp(x) 2^64) — libm exp, computed on the host by oracle and library alike.  E is tracked in Float64
 * (E += dE); the library re-evaluates the energy at every sample instead, so its samples agree to rounding (1e-6 relative is the
 * north-star bound for Float64 models), while configurations and accepted counts are identical. */
ORC_API int64_t orc_standard_mc_spf_fast(int form, int64_t N, int64_t K, const int32_t *A, const double *J, double beta, int64_t iters,
                                         int64_t step, uint64_t seed, uint64_t it0, uint32_t replica,
                                         uint64_t *chunks, double *Es, int64_t *accepted_out)
{
    spf_t X = {N, K, A, J, NULL, NULL, -1, form, NULL, 1, 1.0, NULL, NULL};
    X.lfields = (double *)malloc((size_t)N * sizeof(double));
    X.lfields_last = (double *)malloc((size_t)N * sizeof(double));
    double E = spf_energy(&X, chunks);                       /* :95 */
    free(X.lfields); free(X.lfields_last);
    int64_t accepted = 0, nsamp = 0;
    for (int64_t it = 1; it <= iters; ++it) {
        if (it % step == 0) Es[nsamp++] = E;
        uint64_t g = it0 + (uint64_t)it;
        int64_t i = orc_site(seed, g, N);
        uint32_t u = 0;
        const int si = spin_bit(chunks, i);
        for (int64_t k = 0; k < K; ++k) {
            const int sy = spin_bit(chunks, A[i * K + k]);
            const int neg = J[i * K + k] < 0;
            u |= (uint32_t)(si ^ sy ^ neg) << k;               /* sigma_i sigma_y J < 0 */
        }
        const double dE = orc_spf_pattern_delta(K, J + i * K, u);
        if (!accept_move(-beta * dE, seed, g, replica)) continue;
        bitflip(chunks, i);
        E += dE;
        accepted += 1;
    }
    if (accepted_out) *accepted_out = accepted;
    return nsamp;
}

/* =============================================================================================
 * Reduced-rejection-rate path: ArraySet, DeltaECache, GraphQT, GraphQuant, rrrMC(DoubleGraph)
 * ============================================================================================= */

/* RRR stream: the draws of replica r at global iteration g,
 *   ctr = (lo32(g), hi32(g), r, TAG_RRR | sub << 8):
 *   sub 0: words 0,1 -> class uniform (53 bit)  [rand() of rand_move, src/DeltaE.jl:148]
 *          words 2,3 -> 64-bit word for the member index floor(u64 * t / 2^64)  [rand(1:t), src/ArraySets.jl:83]
 *   sub 1: words 0,1 -> acceptance uniform (53 bit)  [rand() of accept(c, x), src/RRRMC.jl:43] */
static void rrr_draw(uint64_t seed, uint64_t g, uint32_t replica, uint32_t sub, uint32_t w[4])
{
    orc_draw(seed, (uint32_t)g, (uint32_t)(g >> 32), replica, (uint32_t)ORC_TAG_RRR | (sub << 8), w);
}
static inline double u53_of(uint32_t hi, uint32_t lo) { return (double)((((uint64_t)hi << 32) | lo) >> 11) * 0x1.0p-53; }

/* ---- ArraySet: src/ArraySets.jl:19-85 (0-based elements; pos stores index + 1, 0 = absent) ---- */
typedef struct { int64_t N; int32_t *v, *pos; int64_t t; } aset_t;
static void aset_init(aset_t *a, int64_t N) { a->N = N; a->v = (int32_t *)calloc((size_t)N, 4); a->pos = (int32_t *)calloc((size_t)N, 4); a->t = 0; }
static void aset_free(aset_t *a) { free(a->v); free(a->pos); }
static void aset_push(aset_t *a, int32_t i) { a->v[a->t] = i; a->t += 1; a->pos[i] = (int32_t)a->t; }             /* :56-65 */
static void aset_delete(aset_t *a, int32_t i)                                                                     /* :66-76 */
{
    int32_t p = a->pos[i];
    a->v[p - 1] = a->v[a->t - 1];
    a->pos[a->v[p - 1]] = p;
    a->pos[i] = 0;
    a->t -= 1;
}

/* ---- the inner DiscrGraph as seen by the cache: GraphQT (src/graphs/QT.jl:42-122) ---- */
typedef struct { int64_t N, M, Nk; double fourK; } qt_t;
static inline void qt_neighbors(const qt_t *X, int64_t i, int64_t *j1, int64_t *j2)      /* QT.jl:105-108 (0-based) */
{
    *j1 = i - X->Nk + (i < X->Nk ? X->N : 0);
    *j2 = i + X->Nk - (i + X->Nk >= X->N ? X->N : 0);
}
static inline double qt_delta_energy(const qt_t *X, const uint64_t *s, int64_t i)        /* QT.jl:86-103 */
{
    int64_t k1, k2;
    qt_neighbors(X, i, &k1, &k2);
    int sk = spin_bit(s, i), s1 = spin_bit(s, k1), s2 = spin_bit(s, k2);
    int d = (sk == s1) - (sk != s2);                   /* (sk xor ~s1) - (sk xor s2) */
    return (double)d * X->fourK;
}
static int64_t qt_energy0(const qt_t *X, const uint64_t *s)                              /* QT.jl:68-82 */
{
    int64_t n = 0;
    for (int64_t i = 0; i < X->Nk; ++i) {
        int sj = spin_bit(s, i + (X->M - 1) * X->Nk);
        for (int64_t k = 0; k < X->M; ++k) {
            int sk = spin_bit(s, i + k * X->Nk);
            n -= 1 - 2 * (sk ^ sj);
            sj = sk;
        }
    }
    return n;
}
static inline double qt_energy(const qt_t *X, const uint64_t *s) { return (double)qt_energy0(X, s) * X->fourK / 4; }   /* QT.jl:84 */

/* ---- DeltaECache{Float64, L = 2} over GraphQT: src/DeltaE.jl:63-295 (classes 0-based: k = a + L*up) ---- */
enum { QL = 2 };
typedef struct {
    int64_t N;
    double dElist[QL], ft[QL], T[2 * QL], Tp[2 * QL], z, zp;
    aset_t as[2 * QL];
    int8_t *pos;
    int64_t staged[3][3];
    int nstaged;
} dec_t;

static inline int dec_findk(const dec_t *c, double dE)           /* findk: DeltaE.jl:28-60 (exact comparison of |dE|) */
{
    double a = fabs(dE);
    for (int k = 0; k < QL; ++k) if (a == c->dElist[k]) return k;
    return -1;
}
static inline double dec_class_f(const dec_t *c, int k) { return k >= QL ? c->ft[k - QL] : 1.0; }     /* get_class_f: :139 */
static inline int dec_class_of(const dec_t *c, double dE, int sbit)                                 /* :80-86, :214-216 */
{
    int a = dec_findk(c, dE);
    int up = dE > 0 || (dE == 0 && sbit == 1);
    return a + QL * up;
}

static void dec_init(dec_t *c, const qt_t *X0, const uint64_t *s, double beta)                      /* DeltaE.jl:74-103 */
{
    c->N = X0->N;
    c->dElist[0] = 0.0; c->dElist[1] = X0->fourK;                /* allΔE(GraphQT) = (0.0, fourK): QT.jl:111 */
    for (int k = 0; k < 2 * QL; ++k) aset_init(&c->as[k], c->N);
    c->pos = (int8_t *)calloc((size_t)c->N, 1);
    for (int64_t i = 0; i < c->N; ++i) {
        int k = dec_class_of(c, qt_delta_energy(X0, s, i), spin_bit(s, i));
        c->pos[i] = (int8_t)k;
        aset_push(&c->as[k], (int32_t)i);
    }
    for (int k = 0; k < QL; ++k) c->ft[k] = orc_det_exp(-beta * c->dElist[k]);      /* exp(-beta dE): :91 (shared deterministic exp) */
    c->z = 0.0;
    for (int k = 0; k < 2 * QL; ++k) {
        double x = (double)c->as[k].t * dec_class_f(c, k);
        c->z += x;
        c->T[k] = x;
    }
    c->zp = c->z;
    c->nstaged = 0;
}
static void dec_free(dec_t *c) { for (int k = 0; k < 2 * QL; ++k) aset_free(&c->as[k]); free(c->pos); }

/* rand_move: DeltaE.jl:146-167 */
static int64_t dec_rand_move(const dec_t *c, uint64_t seed, uint64_t g, uint32_t replica, double *dE)
{
    uint32_t w[4];
    rrr_draw(seed, g, replica, 0, w);
    double r = u53_of(w[0], w[1]) * c->z;
    int k = 0;
    double cT = 0.0;
    for (k = 0; k < 2 * QL; ++k) {
        cT += c->T[k];
        if (r < cT) break;
    }
    if (k == 2 * QL) k = 2 * QL - 1;                 /* `for outer k` leaves k at the last value */
    if (!(r < cT)) while (c->T[k] == 0) k -= 1;      /* :155-157 */
    *dE = k < QL ? -c->dElist[k] : c->dElist[k - QL];
    uint64_t u = ((uint64_t)w[2] << 32) | w[3];
    int64_t idx = (int64_t)orc_mulhi64(u, (uint64_t)c->as[k].t);        /* rand(1:t) - 1 */
    return c->as[k].v[idx];
}

/* compute_staged!: DeltaE.jl:202-230 on X0 = GraphQT (no cache: spinflip! only flips the bit) */
static void dec_compute_staged(dec_t *c, const qt_t *X0, uint64_t *s, int64_t i)
{
    bitflip(s, i);
    c->nstaged = 0;
    int64_t nb[2];
    qt_neighbors(X0, i, &nb[0], &nb[1]);
    for (int q = 0; q < 2; ++q) {
        int64_t j = nb[q];
        int k0 = c->pos[j];
        int k1 = dec_class_of(c, qt_delta_energy(X0, s, j), spin_bit(s, j));
        if (k0 == k1) continue;
        c->staged[c->nstaged][0] = j; c->staged[c->nstaged][1] = k0; c->staged[c->nstaged][2] = k1; c->nstaged++;
    }
    int k0 = c->pos[i];
    int k1 = k0 >= QL ? k0 - QL : k0 + QL;           /* k1 = k0 - L(2(k0 > L) - 1) */
    c->staged[c->nstaged][0] = i; c->staged[c->nstaged][1] = k0; c->staged[c->nstaged][2] = k1; c->nstaged++;
    bitflip(s, i);
}

/* compute_reverse_probabilities!: DeltaE.jl:184-200 */
static double dec_reverse(dec_t *c)
{
    double zp = c->z;
    memcpy(c->Tp, c->T, sizeof c->T);
    for (int q = 0; q < c->nstaged; ++q) {
        int k0 = (int)c->staged[q][1], k1 = (int)c->staged[q][2];
        double f0 = dec_class_f(c, k0), f1 = dec_class_f(c, k1);
        c->Tp[k0] -= f0;
        c->Tp[k1] += f1;
        zp += f1 - f0;
    }
    c->zp = zp;
    return zp;
}

/* apply_staged!: DeltaE.jl:169-182 */
static void dec_apply_staged(dec_t *c)
{
    for (int q = 0; q < c->nstaged; ++q) {
        int32_t j = (int32_t)c->staged[q][0];
        int k0 = (int)c->staged[q][1], k1 = (int)c->staged[q][2];
        aset_delete(&c->as[k0], j);
        aset_push(&c->as[k1], j);
        c->pos[j] = (int8_t)k1;
    }
    double tmp[2 * QL];
    memcpy(tmp, c->T, sizeof tmp); memcpy(c->T, c->Tp, sizeof tmp); memcpy(c->Tp, tmp, sizeof tmp);     /* T, T' = T', T */
    c->z = c->zp;
}

/* binary GraphSK (src/graphs/SK.jl:28-165): defined with its samplers further down, used here as a slice graph */
typedef struct {
    int64_t N, nch;
    double sN;
    const uint64_t *J;
    int64_t *lfields, *lfields_last;
    int64_t move_last;
} skb_t;
static double skb_energy(skb_t *X, const uint64_t *s);
static void skb_update_cache(skb_t *X, const uint64_t *s, int64_t move);

/* ---- GraphQuant over M slices of one GraphRRG disorder (A, J), or of one binary GraphSK disorder (GraphQSKT, src/QAliases.jl:34-43 —
 * the reference's test_QIsing experiment, scripts/scripts.jl:766-864): src/graphs/QT.jl:126-321 ---- */
typedef struct {
    qt_t X0;
    int64_t Nk, M, K;
    sparse_t *X1;            /* M slice graphs sharing A, J; each with its own LocalFields */
    skb_t *S1;               /* or (X1 == NULL) M binary-SK slice graphs sharing J */
    skn_t *G1;               /* or M GraphSKNormal slice graphs sharing J (GraphQSKNormalT, src/QAliases.jl:45-46; test/runtests.jl:80) */
    spf_t *F1;               /* or M GraphEANormal / GraphRRGNormal slice graphs sharing (A, J::Float64) (GraphQEAT, src/QAliases.jl:50-83) */
    uint64_t **C1;           /* M slice configurations (copies of the slice bits) */
} quant_t;

static void quant_init(quant_t *Q, int64_t Nk, int64_t M, int64_t K, const int32_t *A, const int32_t *J, double fourK)
{
    Q->X0.N = Nk * M; Q->X0.M = M; Q->X0.Nk = Nk; Q->X0.fourK = fourK;
    Q->Nk = Nk; Q->M = M; Q->K = K;
    Q->X1 = (sparse_t *)calloc((size_t)M, sizeof(sparse_t));
    Q->C1 = (uint64_t **)calloc((size_t)M, sizeof(uint64_t *));
    /* which update_cache! the slices follow: GraphRRG's (RRG.jl:191-234) unless a row lists a neighbour twice — only a GraphEA does
     * (L = 2, EA.jl:156), and its update_cache! walks the de-duplicated uA in the move_last fast path (EA.jl:224-264); without
     * repeated entries the two forms give the same integers */
    int form = ORC_FORM_RRG;
    for (int64_t x = 0; x < Nk && form == ORC_FORM_RRG; ++x)
        for (int64_t q = 1; q < K; ++q)
            if (A[x * K + q] == A[x * K + q - 1]) { form = ORC_FORM_EA; break; }
    for (int64_t k = 0; k < M; ++k) {
        sparse_t X = {Nk, K, A, J, NULL, NULL, -1, form};
        X.lfields = (int64_t *)calloc((size_t)Nk, 8);
        X.lfields_last = (int64_t *)calloc((size_t)Nk, 8);
        Q->X1[k] = X;
        Q->C1[k] = (uint64_t *)calloc((size_t)((Nk + 63) / 64), 8);
    }
    Q->S1 = NULL; Q->G1 = NULL; Q->F1 = NULL;
}
/* GraphQuant over M sparse Float64 slice graphs sharing (A, J) — GraphQEAT = GraphQuant{fourK,GraphEANormal{twoD}} (src/QAliases.jl:50-83):
 * every slice is a GraphEANormal (form = 1: EA.jl:534-680) or GraphRRGNormal (form = 0: RRG.jl:503-627) with its own LocalFields{Float64} */
static void quant_init_spf(quant_t *Q, int64_t Nk, int64_t M, int64_t K, const int32_t *A, const double *Jf, int form, double fourK)
{
    Q->X0.N = Nk * M; Q->X0.M = M; Q->X0.Nk = Nk; Q->X0.fourK = fourK;
    Q->Nk = Nk; Q->M = M; Q->K = K;
    Q->X1 = NULL; Q->S1 = NULL; Q->G1 = NULL;
    Q->F1 = (spf_t *)calloc((size_t)M, sizeof(spf_t));
    Q->C1 = (uint64_t **)calloc((size_t)M, sizeof(uint64_t *));
    for (int64_t k = 0; k < M; ++k) {
        spf_t X = {Nk, K, A, Jf, NULL, NULL, -1, form, NULL, 1, 1.0, NULL, NULL};
        X.lfields = (double *)calloc((size_t)Nk, 8);
        X.lfields_last = (double *)calloc((size_t)Nk, 8);
        Q->F1[k] = X;
        Q->C1[k] = (uint64_t *)calloc((size_t)((Nk + 63) / 64), 8);
    }
}
static void quant_init_skn(quant_t *Q, int64_t Nk, int64_t M, const double *Jd, double fourK)
{
    Q->X0.N = Nk * M; Q->X0.M = M; Q->X0.Nk = Nk; Q->X0.fourK = fourK;
    Q->Nk = Nk; Q->M = M; Q->K = 0;
    Q->X1 = NULL; Q->S1 = NULL; Q->F1 = NULL;
    Q->G1 = (skn_t *)calloc((size_t)M, sizeof(skn_t));
    Q->C1 = (uint64_t **)calloc((size_t)M, sizeof(uint64_t *));
    for (int64_t k = 0; k < M; ++k) {
        skn_t X = {Nk, Jd, NULL, NULL, -1};
        X.lfields = (double *)calloc((size_t)Nk, 8);
        X.lfields_last = (double *)calloc((size_t)Nk, 8);
        Q->G1[k] = X;
        Q->C1[k] = (uint64_t *)calloc((size_t)((Nk + 63) / 64), 8);
    }
}
static void quant_init_sk(quant_t *Q, int64_t Nk, int64_t M, const uint64_t *Jb, double fourK)
{
    Q->X0.N = Nk * M; Q->X0.M = M; Q->X0.Nk = Nk; Q->X0.fourK = fourK;
    Q->Nk = Nk; Q->M = M; Q->K = 0;
    Q->X1 = NULL;
    Q->G1 = NULL; Q->F1 = NULL;
    Q->S1 = (skb_t *)calloc((size_t)M, sizeof(skb_t));
    Q->C1 = (uint64_t **)calloc((size_t)M, sizeof(uint64_t *));
    for (int64_t k = 0; k < M; ++k) {
        skb_t X = {Nk, (Nk + 63) / 64, sqrt((double)Nk), Jb, NULL, NULL, -1};
        X.lfields = (int64_t *)calloc((size_t)Nk, 8);
        X.lfields_last = (int64_t *)calloc((size_t)Nk, 8);
        Q->S1[k] = X;
        Q->C1[k] = (uint64_t *)calloc((size_t)((Nk + 63) / 64), 8);
    }
}
static void quant_free(quant_t *Q)
{
    for (int64_t k = 0; k < Q->M; ++k) {
        if (Q->X1) { free(Q->X1[k].lfields); free(Q->X1[k].lfields_last); }
        if (Q->S1) { free(Q->S1[k].lfields); free(Q->S1[k].lfields_last); }
        if (Q->G1) { free(Q->G1[k].lfields); free(Q->G1[k].lfields_last); }
        if (Q->F1) { free(Q->F1[k].lfields); free(Q->F1[k].lfields_last); }
        free(Q->C1[k]);
    }
    free(Q->X1); free(Q->S1); free(Q->G1); free(Q->F1); free(Q->C1);
}
/* energy(X1[k], C1[k]) as the Float64 the reference divides: an Int for GraphRRG slices, n / sqrt(Nk) for GraphSK ones (SK.jl:95) */
static inline double quant_slice_energy(quant_t *Q, int64_t k)
{
    if (Q->G1) return skn_energy(&Q->G1[k], Q->C1[k]);
    if (Q->F1) return spf_energy(&Q->F1[k], Q->C1[k]);
    return Q->S1 ? skb_energy(&Q->S1[k], Q->C1[k]) : (double)sparse_energy(&Q->X1[k], Q->C1[k]);
}
/* energy: QT.jl:185-199 — copies the slice bits into C1[k] and (re)builds every slice cache */
static double quant_energy(quant_t *Q, const uint64_t *s)
{
    double E = qt_energy(&Q->X0, s);
    for (int64_t k = 0; k < Q->M; ++k) {
        memset(Q->C1[k], 0, (size_t)((Q->Nk + 63) / 64) * 8);
        for (int64_t i = 0; i < Q->Nk; ++i)
            if (spin_bit(s, k * Q->Nk + i)) Q->C1[k][i >> 6] |= 1ull << (i & 63);
        E += quant_slice_energy(Q, k) / (double)Q->M;
    }
    return E;
}
/* delta_energy_residual: QT.jl:270-281 */
static inline double quant_residual(const quant_t *Q, int64_t move)
{
    int64_t k = move / Q->Nk, i = move % Q->Nk;
    if (Q->G1) return Q->G1[k].lfields[i] / (double)Q->M;                                   /* SK.jl:278-284 */
    if (Q->F1) return (-Q->F1[k].lfields[i]) / (double)Q->M;                                /* delta_energy = -lfields: EA.jl:655-661, RRG.jl:619-625 */
    if (Q->S1) return ((double)Q->S1[k].lfields[i] / Q->S1[k].sN) / (double)Q->M;          /* SK.jl:137-140 */
    return (double)sparse_delta_energy(&Q->X1[k], i) / (double)Q->M;
}
/* spinflip!(X::GraphQuant, C, move): Interface.jl:89-92 + update_cache! QT.jl:172-183 */
static void quant_spinflip(quant_t *Q, uint64_t *s, int64_t move)
{
    bitflip(s, move);
    int64_t k = move / Q->Nk, i = move % Q->Nk;
    bitflip(Q->C1[k], i);
    if (Q->G1) { skn_update_cache(&Q->G1[k], Q->C1[k], i); return; }
    if (Q->F1) { spf_update_cache(&Q->F1[k], Q->C1[k], i); return; }
    if (Q->S1) skb_update_cache(&Q->S1[k], Q->C1[k], i);
    else sparse_update_cache(&Q->X1[k], Q->C1[k], i);
}

/* apply_move!: DeltaE.jl:232-295 with X = GraphQuant, X0 = inner_graph(X) = GraphQT */
static double dec_apply_move(dec_t *c, quant_t *Q, uint64_t *s, int64_t move)
{
    quant_spinflip(Q, s, move);
    const qt_t *X0 = &Q->X0;
    double zp = c->z;
    int64_t nb[2];
    qt_neighbors(X0, move, &nb[0], &nb[1]);
    for (int q = 0; q < 2; ++q) {
        int32_t j = (int32_t)nb[q];
        int k0 = c->pos[j];
        int k1 = dec_class_of(c, qt_delta_energy(X0, s, j), spin_bit(s, j));
        if (k0 == k1) continue;
        double f0 = dec_class_f(c, k0), f1 = dec_class_f(c, k1);
        c->T[k0] -= f0;
        c->T[k1] += f1;
        zp += f1 - f0;
        aset_delete(&c->as[k0], j);
        aset_push(&c->as[k1], j);
        c->pos[j] = (int8_t)k1;
    }
    int k0 = c->pos[move];
    int k1 = k0 >= QL ? k0 - QL : k0 + QL;
    double f0 = dec_class_f(c, k0), f1 = dec_class_f(c, k1);
    c->T[k0] -= f0;
    c->T[k1] += f1;
    zp += f1 - f0;
    aset_delete(&c->as[k0], (int32_t)move);
    aset_push(&c->as[k1], (int32_t)move);
    c->pos[move] = (int8_t)k1;
    double cc = c->z / zp;
    c->z = zp;
    return cc;
}

/* accept(c, x): src/RRRMC.jl:40-44 */
static int accept_cx(double c, double x, uint64_t seed, uint64_t g, uint32_t replica)
{
    if (c >= 1 && x >= 0) return 1;
    double a = c * orc_det_exp(x);
    if (a >= 1) return 1;
    uint32_t w[4];
    rrr_draw(seed, g, replica, 1, w);
    return u53_of(w[0], w[1]) < a;
}

/*
 * rrrMC(X::DoubleGraph, beta, iters; step, staged_thr, staged_thr_fact): src/RRRMC.jl:221-290, X = GraphQuant over
 * M slices of GraphRRG{Int,(-1,1),K}(A, J) with Nk spins each.  One chain.
 *   chunks  in/out  the N = Nk*M spins, slice-major (slice k holds bits k*Nk .. (k+1)*Nk-1)
 *   Es      out     energies sampled before the move of iteration k*step
 *   stats   out     [accepted, staged_its]; cache_out (optional): pos[N] then the four set sizes
 */
static int64_t orc_rrr_mc_quant_impl(quant_t *Q,
                                 double beta, int64_t iters, int64_t step, double staged_thr, double staged_thr_fact,
                                 uint64_t seed, uint64_t it0, uint32_t replica,
                                 uint64_t *chunks, double *Es, int64_t *stats, int32_t *cache_out)
{
    const int64_t Nk = Q->Nk, M = Q->M;
    const double fourK = Q->X0.fourK;
    (void)Nk; (void)M; (void)fourK;
    const int64_t N = Nk * M;
    double E = quant_energy(Q, chunks);                                    /* :237 */
    dec_t cache;
    dec_init(&cache, &Q->X0, chunks, beta);                                  /* :239-240 */
    const double lambda = staged_thr_fact / (double)N;                      /* :243 */
    int64_t staged_its = 0, accepted = 0, nsamp = 0;
    double acc_rate = 0.5;
    for (int64_t it = 1; it <= iters; ++it) {                               /* :249-282 */
        if (it % step == 0) { Es[nsamp++] = E; if (!ORC_HOOK(it, accepted, E, 0)) break; }
        const uint64_t g = it0 + (uint64_t)it;
        int acc = 0;
        if (acc_rate < staged_thr) {
            staged_its += 1;
            double z = cache.z, dE0;
            int64_t move = dec_rand_move(&cache, seed, g, replica, &dE0);   /* step_rrr: :131-138 */
            dec_compute_staged(&cache, &Q->X0, chunks, move);
            double zp = dec_reverse(&cache);
            double c = z / zp;
            double dE1 = quant_residual(Q, move);
            if (accept_cx(c, -beta * dE1, seed, g, replica)) {
                quant_spinflip(Q, chunks, move);
                dec_apply_staged(&cache);
                E += dE0 + dE1;
                accepted += 1;
                acc = 1;
            }
        } else {
            double dE0;
            int64_t move = dec_rand_move(&cache, seed, g, replica, &dE0);
            double dE1 = quant_residual(Q, move);
            double c = dec_apply_move(&cache, Q, chunks, move);
            if (accept_cx(c, -beta * dE1, seed, g, replica)) {
                E += dE0 + dE1;
                accepted += 1;
                acc = 1;
            } else {
                dec_apply_move(&cache, Q, chunks, move);
            }
        }
        acc_rate = acc_rate * (1 - lambda) + (double)acc * lambda;          /* :281 */
    }
    if (stats) { stats[0] = accepted; stats[1] = staged_its; }
    if (cache_out) {
        for (int64_t i = 0; i < N; ++i) cache_out[i] = cache.pos[i];
        for (int k = 0; k < 2 * QL; ++k) cache_out[N + k] = (int32_t)cache.as[k].t;
    }
    /* check_consistency(ΔEcache): DeltaE.jl:120-136, ArraySets.jl:27-42 — returns -1 on violation */
    int64_t bad = 0, total = 0;
    for (int k = 0; k < 2 * QL; ++k) {
        total += cache.as[k].t;
        for (int64_t p = 0; p < cache.as[k].t; ++p) {
            int32_t x = cache.as[k].v[p];
            if (cache.as[k].pos[x] != p + 1 || cache.pos[x] != k) bad = 1;
        }
    }
    for (int64_t i = 0; i < N; ++i) {
        int k = dec_class_of(&cache, qt_delta_energy(&Q->X0, chunks, i), spin_bit(chunks, i));
        if (k != cache.pos[i]) bad = 1;
    }
    if (total != N) bad = 1;
    dec_free(&cache);
    return bad ? -1 : nsamp;
}
ORC_API int64_t orc_rrr_mc_quant(int64_t Nk, int64_t M, int64_t K, const int32_t *A, const int32_t *J, double fourK,
                                 double beta, int64_t iters, int64_t step, double staged_thr, double staged_thr_fact,
                                 uint64_t seed, uint64_t it0, uint32_t replica,
                                 uint64_t *chunks, double *Es, int64_t *stats, int32_t *cache_out)
{
    quant_t Q;
    quant_init(&Q, Nk, M, K, A, J, fourK);
    int64_t r = orc_rrr_mc_quant_impl(&Q, beta, iters, step, staged_thr, staged_thr_fact, seed, it0, replica, chunks, Es, stats, cache_out);
    quant_free(&Q);
    return r;
}
ORC_API int64_t orc_rrr_mc_quant_sk(int64_t Nk, int64_t M, const uint64_t *Jb, double fourK,
                                 double beta, int64_t iters, int64_t step, double staged_thr, double staged_thr_fact,
                                 uint64_t seed, uint64_t it0, uint32_t replica,
                                 uint64_t *chunks, double *Es, int64_t *stats, int32_t *cache_out)
{
    quant_t Q;
    quant_init_sk(&Q, Nk, M, Jb, fourK);
    int64_t r = orc_rrr_mc_quant_impl(&Q, beta, iters, step, staged_thr, staged_thr_fact, seed, it0, replica, chunks, Es, stats, cache_out);
    quant_free(&Q);
    return r;
}

ORC_API int64_t orc_rrr_mc_quant_skn(int64_t Nk, int64_t M, const double *Jd, double fourK,
                                 double beta, int64_t iters, int64_t step, double staged_thr, double staged_thr_fact,
                                 uint64_t seed, uint64_t it0, uint32_t replica,
                                 uint64_t *chunks, double *Es, int64_t *stats, int32_t *cache_out)
{
    quant_t Q;
    quant_init_skn(&Q, Nk, M, Jd, fourK);
    int64_t r = orc_rrr_mc_quant_impl(&Q, beta, iters, step, staged_thr, staged_thr_fact, seed, it0, replica, chunks, Es, stats, cache_out);
    quant_free(&Q);
    return r;
}

/* GraphQEAT (src/QAliases.jl:50-83): GraphQuant over sparse Float64 slices */
ORC_API int64_t orc_rrr_mc_quant_spf(int form, int64_t Nk, int64_t M, int64_t K, const int32_t *A, const double *Jf, double fourK,
                                 double beta, int64_t iters, int64_t step, double staged_thr, double staged_thr_fact,
                                 uint64_t seed, uint64_t it0, uint32_t replica,
                                 uint64_t *chunks, double *Es, int64_t *stats, int32_t *cache_out)
{
    quant_t Q;
    quant_init_spf(&Q, Nk, M, K, A, Jf, form, fourK);
    int64_t r = orc_rrr_mc_quant_impl(&Q, beta, iters, step, staged_thr, staged_thr_fact, seed, it0, replica, chunks, Es, stats, cache_out);
    quant_free(&Q);
    return r;
}

/* standardMC (src/RRRMC.jl:81-127) on GraphQuant: delta_energy = delta_energy(X0) + delta_energy_residual (QT.jl:283-286).
 * SITE stream for the spin, ACCEPT_F64 stream for rand(). */
static int64_t orc_standard_mc_quant_impl(quant_t *Q,
                                      double beta, int64_t iters, int64_t step, uint64_t seed, uint64_t it0, uint32_t replica,
                                      uint64_t *chunks, double *Es, int64_t *accepted_out)
{
    const int64_t Nk = Q->Nk, M = Q->M;
    const double fourK = Q->X0.fourK;
    (void)Nk; (void)M; (void)fourK;
    const int64_t N = Nk * M;
    double E = quant_energy(Q, chunks);
    int64_t accepted = 0, nsamp = 0;
    for (int64_t it = 1; it <= iters; ++it) {
        if (it % step == 0) Es[nsamp++] = E;
        const uint64_t g = it0 + (uint64_t)it;
        const int64_t i = orc_site(seed, g, N);
        const double dE = qt_delta_energy(&Q->X0, chunks, i) + quant_residual(Q, i);
        const double x = -beta * dE;
        const int acc = (x >= 0) || (orc_rand53(seed, g, replica) < orc_det_exp(x));      /* RRRMC.jl:39 */
        if (!acc) continue;
        quant_spinflip(Q, chunks, i);
        E += dE;
        accepted += 1;
    }
    if (accepted_out) *accepted_out = accepted;
    return nsamp;
}
ORC_API int64_t orc_standard_mc_quant(int64_t Nk, int64_t M, int64_t K, const int32_t *A, const int32_t *J, double fourK,
                                      double beta, int64_t iters, int64_t step, uint64_t seed, uint64_t it0, uint32_t replica,
                                      uint64_t *chunks, double *Es, int64_t *accepted_out)
{
    quant_t Q;
    quant_init(&Q, Nk, M, K, A, J, fourK);
    int64_t r = orc_standard_mc_quant_impl(&Q, beta, iters, step, seed, it0, replica, chunks, Es, accepted_out);
    quant_free(&Q);
    return r;
}
ORC_API int64_t orc_standard_mc_quant_sk(int64_t Nk, int64_t M, const uint64_t *Jb, double fourK,
                                      double beta, int64_t iters, int64_t step, uint64_t seed, uint64_t it0, uint32_t replica,
                                      uint64_t *chunks, double *Es, int64_t *accepted_out)
{
    quant_t Q;
    quant_init_sk(&Q, Nk, M, Jb, fourK);
    int64_t r = orc_standard_mc_quant_impl(&Q, beta, iters, step, seed, it0, replica, chunks, Es, accepted_out);
    quant_free(&Q);
    return r;
}

ORC_API int64_t orc_standard_mc_quant_skn(int64_t Nk, int64_t M, const double *Jd, double fourK,
                                      double beta, int64_t iters, int64_t step, uint64_t seed, uint64_t it0, uint32_t replica,
                                      uint64_t *chunks, double *Es, int64_t *accepted_out)
{
    quant_t Q;
    quant_init_skn(&Q, Nk, M, Jd, fourK);
    int64_t r = orc_standard_mc_quant_impl(&Q, beta, iters, step, seed, it0, replica, chunks, Es, accepted_out);
    quant_free(&Q);
    return r;
}
ORC_API int64_t orc_standard_mc_quant_spf(int form, int64_t Nk, int64_t M, int64_t K, const int32_t *A, const double *Jf, double fourK,
                                      double beta, int64_t iters, int64_t step, uint64_t seed, uint64_t it0, uint32_t replica,
                                      uint64_t *chunks, double *Es, int64_t *accepted_out)
{
    quant_t Q;
    quant_init_spf(&Q, Nk, M, K, A, Jf, form, fourK);
    int64_t r = orc_standard_mc_quant_impl(&Q, beta, iters, step, seed, it0, replica, chunks, Es, accepted_out);
    quant_free(&Q);
    return r;
}
ORC_API double orc_quant_energy_spf(int form, int64_t Nk, int64_t M, int64_t K, const int32_t *A, const double *Jf, double fourK, const uint64_t *chunks)
{
    quant_t Q;
    quant_init_spf(&Q, Nk, M, K, A, Jf, form, fourK);
    const double E = quant_energy(&Q, chunks);
    quant_free(&Q);
    return E;
}
ORC_API double orc_quant_energy_skn(int64_t Nk, int64_t M, const double *Jd, double fourK, const uint64_t *chunks)
{
    quant_t Q;
    quant_init_skn(&Q, Nk, M, Jd, fourK);
    const double E = quant_energy(&Q, chunks);
    quant_free(&Q);
    return E;
}

/* energy(X::GraphQuant, C) and its parts, for the tests */
static double orc_quant_energy_impl(quant_t *Q,
                                const uint64_t *chunks, double *qt_part)
{
    const int64_t Nk = Q->Nk, M = Q->M;
    const double fourK = Q->X0.fourK;
    (void)Nk; (void)M; (void)fourK;
    double E = quant_energy(Q, chunks);
    if (qt_part) *qt_part = qt_energy(&Q->X0, chunks);
    return E;
}
ORC_API double orc_quant_energy(int64_t Nk, int64_t M, int64_t K, const int32_t *A, const int32_t *J, double fourK,
                                const uint64_t *chunks, double *qt_part)
{
    quant_t Q;
    quant_init(&Q, Nk, M, K, A, J, fourK);
    double r = orc_quant_energy_impl(&Q, chunks, qt_part);
    quant_free(&Q);
    return r;
}
ORC_API double orc_quant_energy_sk(int64_t Nk, int64_t M, const uint64_t *Jb, double fourK,
                                const uint64_t *chunks, double *qt_part)
{
    quant_t Q;
    quant_init_sk(&Q, Nk, M, Jb, fourK);
    double r = orc_quant_energy_impl(&Q, chunks, qt_part);
    quant_free(&Q);
    return r;
}

/* ---------------------------------------------------------------------------------------------
 * Observables
 * ------------------------------------------------------------------------------------------- */
/* pm1dot(a, b) = (2a-1).(2b-1) = N - 2 popcount(a xor b): scripts/scripts.jl:283-295 */
ORC_API int64_t orc_pm1dot(const uint64_t *a, const uint64_t *b, int64_t N)
{
    int64_t l = N;
    for (int64_t c = 0; c < (N + 63) / 64; ++c) l -= 2 * __builtin_popcountll(a[c] ^ b[c]);
    return l;
}

/* Window statistic of parseovs (scripts/scripts.jl:368-405): over the samples i1 in [i, j-2], j1 in [i+1, j-1]
 * (0-based half-open window [i, j)), q2 = (pm1dot/N)^2; returns mean q2 and sqrt(max(0, <q2^2> - <q2>^2)).
 * Cs = samples x nch chunk rows of ONE chain.  Note the reference's loop ranges include i1 == j1 and both orders. */
ORC_API int orc_q2_window(const uint64_t *Cs, int64_t N, int64_t i, int64_t j, double *mq2_out, double *sq2_out)
{
    const int64_t nch = (N + 63) / 64;
    double mq2 = 0.0, mq4 = 0.0;
    int64_t n = 0;
    for (int64_t i1 = i; i1 <= j - 2; ++i1)
        for (int64_t j1 = i + 1; j1 <= j - 1; ++j1) {
            double q = (double)orc_pm1dot(Cs + i1 * nch, Cs + j1 * nch, N) / (double)N;
            double q2 = q * q;
            mq2 += q2;
            mq4 += q2 * q2;
            n += 1;
        }
    if (n == 0) return 1;
    mq2 /= (double)n;
    mq4 /= (double)n;
    double v = mq4 - mq2 * mq2;
    *mq2_out = mq2;
    *sq2_out = sqrt(v > 0.0 ? v : 0.0);
    return 0;
}

/* GraphQuant observables of one configuration:
 *   transverse_mag (QT.jl:113-122): p = -energy0/N, x = beta*fourK/2, cosh(x) - p sinh(x)
 *   overlaps (QT.jl:213-251): ovs[d-1] = sum over slice pairs at ring distance d of (Nk - 2|s1 xor s2|), normalised by M*Nk
 *                             (by M*Nk/2 for the last entry when M is even)
 *   Qenergy (QT.jl:253-268): -Gamma*transverse_mag + sum_k energy(X1[k], C1[k]) / N, accumulated in slice order
 * Integer parts (energy0, per-slice energies, raw overlap sums) are returned too: they are what the device computes. */
static int orc_quant_observables_impl(quant_t *Q, double beta,
                                  double Gamma, const uint64_t *chunks, double *Qenergy, double *tmag, double *ovs,
                                  int64_t *energy0_out, int64_t *Eslice_out, int64_t *ovs_raw_out)
{
    const int64_t Nk = Q->Nk, M = Q->M;
    const double fourK = Q->X0.fourK;
    (void)Nk; (void)M; (void)fourK;
    const int64_t N = Nk * M;
    (void)quant_energy(Q, chunks);                    /* fills C1[k] */
    const int64_t e0 = qt_energy0(&Q->X0, chunks);
    const double p = -(double)e0 / (double)N;
    const double x = beta * fourK / 2;
    const double tm = cosh(x) - p * sinh(x);
    double E = -Gamma * tm;
    for (int64_t k = 0; k < M; ++k) {
        const double Ek = quant_slice_energy(Q, k);            /* GraphSK slices: n / sqrt(Nk), n the integer returned in Eslice_out */
        if (Eslice_out) Eslice_out[k] = Q->S1 ? (int64_t)llround(Ek * Q->S1[k].sN) : (int64_t)Ek;
        E += Ek / (double)N;
    }
    for (int64_t d = 0; d < M / 2; ++d) { ovs[d] = 0.0; if (ovs_raw_out) ovs_raw_out[d] = 0; }
    for (int64_t k1 = 0; k1 < M - 1; ++k1)
        for (int64_t k2 = k1 + 1; k2 < M; ++k2) {
            int64_t d = k2 - k1 < M + k1 - k2 ? k2 - k1 : M + k1 - k2;
            int64_t o = orc_pm1dot(Q->C1[k1], Q->C1[k2], Nk);
            ovs[d - 1] += (double)o;
            if (ovs_raw_out) ovs_raw_out[d - 1] += o;
        }
    for (int64_t d = 1; d <= (M - 1) / 2; ++d) ovs[d - 1] /= (double)(M * Nk);
    if (M % 2 == 0 && M >= 2) ovs[M / 2 - 1] /= (double)(M * Nk) / 2;
    *Qenergy = E;
    *tmag = tm;
    if (energy0_out) *energy0_out = e0;
    return 0;
}
ORC_API int orc_quant_observables(int64_t Nk, int64_t M, int64_t K, const int32_t *A, const int32_t *J, double fourK, double beta,
                                  double Gamma, const uint64_t *chunks, double *Qenergy, double *tmag, double *ovs,
                                  int64_t *energy0_out, int64_t *Eslice_out, int64_t *ovs_raw_out)
{
    quant_t Q;
    quant_init(&Q, Nk, M, K, A, J, fourK);
    int r = orc_quant_observables_impl(&Q, beta, Gamma, chunks, Qenergy, tmag, ovs, energy0_out, Eslice_out, ovs_raw_out);
    quant_free(&Q);
    return r;
}
ORC_API int orc_quant_observables_sk(int64_t Nk, int64_t M, const uint64_t *Jb, double fourK, double beta,
                                  double Gamma, const uint64_t *chunks, double *Qenergy, double *tmag, double *ovs,
                                  int64_t *energy0_out, int64_t *Eslice_out, int64_t *ovs_raw_out)
{
    quant_t Q;
    quant_init_sk(&Q, Nk, M, Jb, fourK);
    int r = orc_quant_observables_impl(&Q, beta, Gamma, chunks, Qenergy, tmag, ovs, energy0_out, Eslice_out, ovs_raw_out);
    quant_free(&Q);
    return r;
}

/* ---------------------------------------------------------------------------------------------
 * Colour-parallel ("checkerboard") sweeps on a sparse +-J model: the build-defined sampler of BASELINE.json config 4
 * (not in the reference; SURVEY.md §7 hard part 7).  One sweep = for each colour in order, every site of that colour
 * attempts a Metropolis flip (RRRMC.jl:39 accept rule, EA.jl:266-275 delta_energy) against the current spins; sites of
 * one colour do not interact, so visiting them in index order equals updating them simultaneously.
 * SWEEP stream: bit j (MSB first) of replica r's uniform at (sweep, site) = bit (r & 31) of word (j & 3) of
 *   ctr = (site, lo32(sweep), r >> 5, TAG_SWEEP | (j >> 2) << 8 | bits 32..47 of sweep << 16).
 * ------------------------------------------------------------------------------------------- */
static int sweep_accept_lt(uint64_t seed, uint64_t sweep, uint32_t site, uint32_t replica, uint64_t T)
{
    uint32_t w[4];
    const uint32_t grp = replica >> 5, bit = replica & 31u;
    const uint32_t c3hi = (uint32_t)((sweep >> 32) & 0xffffu) << 16;
    for (int j = 0; j < 64; ++j) {
        if ((j & 3) == 0) orc_draw(seed, site, (uint32_t)sweep, grp, (uint32_t)ORC_TAG_SWEEP | ((uint32_t)(j >> 2) << 8) | c3hi, w);
        unsigned ub = (w[j & 3] >> bit) & 1u, tb = (unsigned)((T >> (63 - j)) & 1u);
        if (ub != tb) return ub < tb;
    }
    return 0;
}

/* Es[k] = energy BEFORE sweep (k+1)*step; returns the number of samples.  color[N] must be a proper colouring. */
ORC_API int64_t orc_colored_sweeps_sparse(int64_t N, int64_t K, const int32_t *A, const int32_t *J, const int32_t *color, int32_t ncolors,
                                          double beta, int64_t sweeps, int64_t step, uint64_t seed, uint64_t sweep0, uint32_t replica,
                                          uint64_t *chunks, int64_t *Es, int64_t *accepted_out)
{
    sparse_t X = {N, K, A, J, NULL, NULL, -1, 1};
    X.lfields = (int64_t *)malloc((size_t)N * sizeof(int64_t));
    X.lfields_last = (int64_t *)malloc((size_t)N * sizeof(int64_t));
    int64_t E = sparse_energy(&X, chunks);
    int64_t accepted = 0, nsamp = 0;
    for (int64_t sw = 1; sw <= sweeps; ++sw) {
        if (sw % step == 0) Es[nsamp++] = E;
        const uint64_t gsw = sweep0 + (uint64_t)sw;
        for (int32_t c = 0; c < ncolors; ++c)
            for (int64_t i = 0; i < N; ++i) {
                if (color[i] != c) continue;
                int64_t dE = sparse_delta_energy(&X, i);
                double x = -beta * (double)dE;
                int acc = 1;
                if (!(x >= 0)) {
                    int always;
                    uint64_t T = orc_threshold64(exp(x), &always);
                    acc = always ? 1 : sweep_accept_lt(seed, gsw, (uint32_t)i, replica, T);
                }
                if (!acc) continue;
                bitflip(chunks, i);
                sparse_update_cache(&X, chunks, i);
                E += dE;
                accepted += 1;
            }
    }
    if (accepted_out) *accepted_out = accepted;
    free(X.lfields); free(X.lfields_last);
    return nsamp;
}

/* ---------------------------------------------------------------------------------------------
 * Binary SK model GraphSK (couplings +-1/sqrt(N), bit-packed; integer cache): src/graphs/SK.jl:17-165
 * J is given as N rows of Julia-BitVector chunks (nch = ceil(N/64) words per row).
 * ------------------------------------------------------------------------------------------- */
enum { ORC_TAG_SKBITS = 10 };

/* gen_J: SK.jl:17-26.  Row i = bitrand(N): bit j = bit (j & 31) of word ((j >> 5) & 3) of ctr (j >> 7, i, 0, TAG_SKBITS);
 * then J[i][i] = 0 and the upper triangle is mirrored. */
ORC_API void orc_gen_sk_binary(int64_t N, uint64_t seed, uint64_t *J)
{
    int64_t nch = (N + 63) / 64;
    memset(J, 0, (size_t)(N * nch) * 8);
    for (int64_t i = 0; i < N; ++i)
        for (int64_t j = 0; j < N; ++j) {
            uint32_t w[4];
            orc_draw(seed, (uint32_t)(j >> 7), (uint32_t)i, 0u, ORC_TAG_SKBITS, w);
            if ((w[(j >> 5) & 3] >> (j & 31)) & 1u) J[i * nch + (j >> 6)] |= 1ull << (j & 63);
        }
    for (int64_t i = 0; i < N; ++i) {
        J[i * nch + (i >> 6)] &= ~(1ull << (i & 63));
        for (int64_t j = i + 1; j < N; ++j) {
            uint64_t b = (J[i * nch + (j >> 6)] >> (j & 63)) & 1ull;
            J[j * nch + (i >> 6)] = (J[j * nch + (i >> 6)] & ~(1ull << (i & 63))) | (b << (i & 63));
        }
    }
}

/* energy: SK.jl:62-96 */
static double skb_energy(skb_t *X, const uint64_t *s)
{
    int64_t sum_s = 0;
    for (int64_t c = 0; c < X->nch; ++c) sum_s += __builtin_popcountll(s[c]);
    int64_t n = -2 * sum_s;
    for (int64_t i = 0; i < X->N; ++i) {
        const uint64_t *Ji = X->J + i * X->nch;
        int64_t sc = 0;
        for (int64_t c = 0; c < X->nch; ++c) sc += __builtin_popcountll(Ji[c] ^ s[c]);      /* sum(map!(xor, tmps, Ji, s)) */
        int64_t si = spin_bit(s, i);
        int64_t lf = -(2 * si - 1) * (X->N - 1 - 2 * sc);
        X->lfields[i] = 2 * (-lf + 2 * si);
        n += lf;
    }
    n /= 2;                                   /* @assert n % 2 == 0 */
    X->move_last = -1;
    memset(X->lfields_last, 0, (size_t)X->N * 8);
    return (double)n / X->sN;
}

/* update_cache!: SK.jl:98-135 */
static void skb_update_cache(skb_t *X, const uint64_t *s, int64_t move)
{
    if (X->move_last == move) {
        int64_t *t = X->lfields; X->lfields = X->lfields_last; X->lfields_last = t;
        return;
    }
    const uint64_t *Ji = X->J + move * X->nch;
    int64_t si = spin_bit(s, move);
    int64_t lfm = X->lfields[move];
    for (int64_t j = 0; j < X->N; ++j) {
        int64_t Jsij = si ^ spin_bit(s, j) ^ spin_bit(Ji, j);
        int64_t lfj = X->lfields[j];
        X->lfields_last[j] = lfj;
        X->lfields[j] = lfj + 8 * Jsij - 4;
    }
    X->lfields_last[move] = lfm;
    X->lfields[move] = -lfm;
    X->move_last = move;
}

ORC_API double orc_skb_energy(int64_t N, const uint64_t *J, const uint64_t *chunks, int64_t *lfields_out)
{
    skb_t X = {N, (N + 63) / 64, sqrt((double)N), J, NULL, NULL, -1};
    X.lfields = (int64_t *)malloc((size_t)N * 8);
    X.lfields_last = (int64_t *)malloc((size_t)N * 8);
    double E = skb_energy(&X, chunks);
    if (lfields_out) memcpy(lfields_out, X.lfields, (size_t)N * 8);
    free(X.lfields); free(X.lfields_last);
    return E;
}

/* standardMC (src/RRRMC.jl:81-127) on GraphSK; delta_energy = lfields[move] / sqrt(N) (SK.jl:137-140) */
ORC_API int64_t orc_standard_mc_skb(int64_t N, const uint64_t *J, double beta, int64_t iters, int64_t step,
                                    uint64_t seed, uint64_t it0, uint32_t replica,
                                    uint64_t *chunks, double *Es, int64_t *accepted_out, int64_t *lfields_out)
{
    skb_t X = {N, (N + 63) / 64, sqrt((double)N), J, NULL, NULL, -1};
    X.lfields = (int64_t *)malloc((size_t)N * 8);
    X.lfields_last = (int64_t *)malloc((size_t)N * 8);
    double E = skb_energy(&X, chunks);
    int64_t accepted = 0, nsamp = 0;
    for (int64_t it = 1; it <= iters; ++it) {
        if (it % step == 0) Es[nsamp++] = E;
        uint64_t g = it0 + (uint64_t)it;
        int64_t i = orc_site(seed, g, N);
        double dE = (double)X.lfields[i] / X.sN;
        double x = -beta * dE;
        int acc = (x >= 0) || (orc_rand53(seed, g, replica) < orc_det_exp(x));
        if (!acc) continue;
        bitflip(chunks, i);
        skb_update_cache(&X, chunks, i);
        E += dE;
        accepted += 1;
    }
    if (accepted_out) *accepted_out = accepted;
    if (lfields_out) memcpy(lfields_out, X.lfields, (size_t)N * 8);
    free(X.lfields); free(X.lfields_last);
    return nsamp;
}

/* =============================================================================================
 * Continuous-energy RRR path: DynamicSampler (src/DynamicSamplers.jl:18-176), DeltaECacheCont
 * (src/DeltaE.jl:297-410) and rrrMC(X::SingleGraph) (src/RRRMC.jl:149-219) on GraphSKNormal.
 * ============================================================================================= */

/* ---- DynamicSampler: Wong-Easton binary tree of partial sums.  ps is stored level by level (level lev occupies
 * indices off-1 .. 2*off-2 with off = 2^(lev-1)); node (lev, k) holds the sum of its LEFT subtree.  0-based elements. */
typedef struct {
    double *v, *ps, z;
    int64_t N, N2;
    int levs;
    int64_t trefresh;
} dyns_t;

static void dyns_refresh(dyns_t *d)                                   /* refresh!: DynamicSamplers.jl:84-98 */
{
    double z = 0.0;
    for (int64_t i = 0; i < d->N2; ++i) z += d->v[i];                 /* sum(v), left to right */
    d->z = z;
    for (int64_t k = 0; k < d->N2 - 1; ++k) d->ps[k] = 0.0;
    for (int64_t i = 0; i < d->N; ++i) {
        int64_t k = 0, u = (int64_t)1 << (d->levs - 1), off = 1;     /* the tinds/tpos table of :54-82, walked directly (:178-197) */
        for (int lev = 0; lev < d->levs; ++lev) {
            if ((i & u) == 0) { d->ps[off - 1 + k] += d->v[i]; k *= 2; }
            else k = 2 * k + 1;
            u >>= 1; off *= 2;
        }
    }
    d->trefresh = 0;
}
static void dyns_init(dyns_t *d, int64_t N)                           /* DynamicSampler(v): :34-51 (v filled by the caller) */
{
    d->N = N;
    d->levs = 0;
    while (((int64_t)1 << d->levs) < N) d->levs++;                   /* ceil(log2(N)) */
    d->N2 = (int64_t)1 << d->levs;
    d->v = (double *)calloc((size_t)d->N2, 8);
    d->ps = (double *)calloc((size_t)(d->N2 > 1 ? d->N2 - 1 : 1), 8);
    d->z = 0.0; d->trefresh = 0;
}
static void dyns_free(dyns_t *d) { free(d->v); free(d->ps); }
static void dyns_set(dyns_t *d, int64_t i, double x)                  /* setindex!: :159-176 */
{
    int64_t lim = d->N > 100 ? d->N : 100;
    if (d->trefresh >= lim) dyns_refresh(d);
    d->trefresh += 1;
    double dd = x - d->v[i];
    d->v[i] = x;
    d->z += dd;
    int64_t k = 0, u = (int64_t)1 << (d->levs - 1), off = 1;
    for (int lev = 0; lev < d->levs; ++lev) {
        if ((i & u) == 0) { d->ps[off - 1 + k] += dd; k *= 2; }
        else k = 2 * k + 1;
        u >>= 1; off *= 2;
    }
}
/* getel: :130-152.  Returns the 0-based element, or -1 for the reference's "Unrecoverable loss of precision" error. */
static int64_t dyns_getel(dyns_t *d, double x)
{
    for (;;) {
        x *= d->z;
        int64_t k = 0, off = 1;
        for (int lev = 0; lev < d->levs; ++lev) {
            double p = d->ps[off - 1 + k];
            k *= 2;
            if (x > p) { x -= p; k += 1; }
            off *= 2;
        }
        if (k >= d->N || d->v[k] == 0) {
            if (!(d->trefresh > 0)) return -1;
            dyns_refresh(d);
            continue;                                                  /* `return getel(dynsmp, x)` with the already reduced x */
        }
        return k;
    }
}

static inline double prior_of(double x) { return x > 0 ? orc_det_exp(-x) : 1.0; }      /* prior: DeltaE.jl:297 */

/* GraphSKNormal or the binary GraphSK behind the continuous-energy samplers: both are SimpleGraph{Float64} (SK.jl:28,181), the
 * samplers see delta_energy(X, C, i) (= +lfields[i], SK.jl:278-284, or lfields[i] / sN with integer fields, SK.jl:137-140),
 * spinflip! and energy only. */
typedef struct { skn_t *n; skb_t *b; } skx_t;
static inline double skx_dE(const skx_t *X, int64_t i) { return X->b ? (double)X->b->lfields[i] / X->b->sN : X->n->lfields[i]; }
static inline void skx_update(skx_t *X, const uint64_t *s, int64_t m) { if (X->b) skb_update_cache(X->b, s, m); else skn_update_cache(X->n, s, m); }
static inline double skx_energy(skx_t *X, const uint64_t *s) { return X->b ? skb_energy(X->b, s) : skn_energy(X->n, s); }

/* RRR stream for rrrMC(SingleGraph): sub 0 words 0,1 -> rand() of rand_move (getel); sub 1 words 0,1 -> rand() < c */

/*
 * rrrMC(X::SingleGraph, beta, iters; step, staged_thr = 0.8 for a SimpleGraph, staged_thr_fact): src/RRRMC.jl:149-219
 * with X = GraphSKNormal and gen_ΔEcache = DeltaECacheCont (DeltaE.jl:299-316).  One chain.
 * Returns the number of samples, or -1 on the sampler's precision-loss error.
 */
static int64_t orc_rrr_mc_skn_impl(skx_t *X, int64_t N, double beta, int64_t iters, int64_t step,
                               double staged_thr, double staged_thr_fact, uint64_t seed, uint64_t it0, uint32_t replica,
                               uint64_t *chunks, double *Es, int64_t *stats, double *dE_out, double *z_out)
{
    double E = skx_energy(X, chunks);                                        /* :177 */
    /* DeltaECacheCont: DeltaE.jl:304-313 */
    double *dEs = (double *)malloc((size_t)N * 8);
    dyns_t ds;
    dyns_init(&ds, N);
    for (int64_t i = 0; i < N; ++i) { dEs[i] = skx_dE(X, i); ds.v[i] = prior_of(beta * dEs[i]); }
    dyns_refresh(&ds);
    double *st_dE = (double *)malloc((size_t)N * 8), *st_p = (double *)malloc((size_t)N * 8);
    int64_t *st_j = (int64_t *)malloc((size_t)N * 8);

    const double lambda = staged_thr_fact / (double)N;
    int64_t staged_its = 0, accepted = 0, nsamp = 0, bad = 0;
    double acc_rate = 0.5;
    for (int64_t it = 1; it <= iters && !bad; ++it) {
        if (it % step == 0) { Es[nsamp++] = E; if (!ORC_HOOK(it, accepted, E, 0)) break; }
        const uint64_t g = it0 + (uint64_t)it;
        uint32_t w[4];
        rrr_draw(seed, g, replica, 0, w);
        int acc = 0;
        if (acc_rate < staged_thr) {
            staged_its += 1;
            double z = ds.z;                                                  /* step_rrr: :131-138 */
            int64_t move = dyns_getel(&ds, u53_of(w[0], w[1]));               /* rand_move: DeltaE.jl:327-333 */
            if (move < 0) { bad = 1; break; }
            double dE = dEs[move];
            /* compute_staged!: DeltaE.jl:357-374 */
            bitflip(chunks, move); skx_update(X, chunks, move);
            int64_t ns = 0;
            st_j[ns] = move; st_dE[ns] = skx_dE(X, move); st_p[ns] = prior_of(beta * st_dE[ns]); ns++;
            for (int64_t j = 0; j < N; ++j) {                                 /* AllButOne(N, move): Common.jl:78-92 */
                if (j == move) continue;
                st_j[ns] = j; st_dE[ns] = skx_dE(X, j); st_p[ns] = prior_of(beta * st_dE[ns]); ns++;
            }
            bitflip(chunks, move); skx_update(X, chunks, move);
            /* compute_reverse_probabilities!: DeltaE.jl:345-355 */
            double zp = ds.z;
            for (int64_t q = 0; q < ns; ++q) zp += st_p[q] - ds.v[st_j[q]];
            if (zp < 2.2250738585072014e-308) zp = 2.2250738585072014e-308;   /* clamp(z, floatmin, N) */
            if (zp > (double)N) zp = (double)N;
            double c = z / zp;
            rrr_draw(seed, g, replica, 1, w);
            if (u53_of(w[0], w[1]) < c) {                                     /* :192-198 */
                bitflip(chunks, move); skx_update(X, chunks, move);
                for (int64_t q = 0; q < ns; ++q) { dEs[st_j[q]] = st_dE[q]; dyns_set(&ds, st_j[q], st_p[q]); }     /* apply_staged! */
                E += dE;
                accepted += 1;
                acc = 1;
            }
        } else {
            int64_t move = dyns_getel(&ds, u53_of(w[0], w[1]));
            if (move < 0) { bad = 1; break; }
            double dE = dEs[move];
            double c = 0.0;
            for (int pass = 0; pass < 2; ++pass) {                            /* apply_move!: DeltaE.jl:379-410 */
                bitflip(chunks, move); skx_update(X, chunks, move);
                double z = ds.z;
                dEs[move] = skx_dE(X, move);
                dyns_set(&ds, move, prior_of(beta * dEs[move]));
                for (int64_t j = 0; j < N; ++j) {
                    if (j == move) continue;
                    dEs[j] = skx_dE(X, j);
                    dyns_set(&ds, j, prior_of(beta * dEs[j]));
                }
                double cc = z / ds.z;
                if (pass == 1) break;
                c = cc;
                rrr_draw(seed, g, replica, 1, w);
                if (u53_of(w[0], w[1]) < c) { E += dE; accepted += 1; acc = 1; break; }     /* :202-208 */
            }
        }
        acc_rate = acc_rate * (1 - lambda) + (double)acc * lambda;
    }
    if (stats) { stats[0] = accepted; stats[1] = staged_its; }
    if (dE_out) memcpy(dE_out, dEs, (size_t)N * 8);
    if (z_out) *z_out = ds.z;
    free(dEs); free(st_dE); free(st_p); free(st_j);
    dyns_free(&ds);
    return bad ? -1 : nsamp;
}
ORC_API int64_t orc_rrr_mc_skn(int64_t N, const double *J, double beta, int64_t iters, int64_t step,
                               double staged_thr, double staged_thr_fact, uint64_t seed, uint64_t it0, uint32_t replica,
                               uint64_t *chunks, double *Es, int64_t *stats, double *dE_out, double *z_out)
{
    skn_t Xn = {N, J, NULL, NULL, -1};
    Xn.lfields = (double *)malloc((size_t)N * 8);
    Xn.lfields_last = (double *)malloc((size_t)N * 8);
    skx_t X = {&Xn, NULL};
    int64_t r = orc_rrr_mc_skn_impl(&X, N, beta, iters, step, staged_thr, staged_thr_fact, seed, it0, replica, chunks, Es, stats, dE_out, z_out);
    free(Xn.lfields); free(Xn.lfields_last);
    return r;
}
/* the same sampler on the binary GraphSK (J = bit rows, SK.jl:32) */
ORC_API int64_t orc_rrr_mc_skb(int64_t N, const uint64_t *Jb, double beta, int64_t iters, int64_t step,
                               double staged_thr, double staged_thr_fact, uint64_t seed, uint64_t it0, uint32_t replica,
                               uint64_t *chunks, double *Es, int64_t *stats, double *dE_out, double *z_out)
{
    skb_t Xb = {N, (N + 63) / 64, sqrt((double)N), Jb, NULL, NULL, -1};
    Xb.lfields = (int64_t *)malloc((size_t)N * 8);
    Xb.lfields_last = (int64_t *)malloc((size_t)N * 8);
    skx_t X = {NULL, &Xb};
    int64_t r = orc_rrr_mc_skn_impl(&X, N, beta, iters, step, staged_thr, staged_thr_fact, seed, it0, replica, chunks, Es, stats, dE_out, z_out);
    free(Xb.lfields); free(Xb.lfields_last);
    return r;
}

/* DynamicSampler unit access for the tests: build from v, apply updates (i, x), sample with the given uniforms */
ORC_API int64_t orc_dyns_test(int64_t N, const double *v, int64_t nupd, const int64_t *upd_i, const double *upd_x,
                              int64_t nsmp, const double *xs, int64_t *out, double *z_out, double *ps_out)
{
    dyns_t d;
    dyns_init(&d, N);
    for (int64_t i = 0; i < N; ++i) d.v[i] = v[i];
    dyns_refresh(&d);
    for (int64_t q = 0; q < nupd; ++q) dyns_set(&d, upd_i[q], upd_x[q]);
    for (int64_t q = 0; q < nsmp; ++q) out[q] = dyns_getel(&d, xs[q]);
    if (z_out) *z_out = d.z;
    if (ps_out) memcpy(ps_out, d.ps, (size_t)(d.N2 - 1) * 8);
    int64_t n2 = d.N2;
    dyns_free(&d);
    return n2;
}

/* =============================================================================================
 * rrrMC(X::SingleGraph) (src/RRRMC.jl:149-219) and bklMC (src/RRRMC.jl:311-359) on the DiscrGraphs GraphRRG / GraphEA
 * with DeltaECache{Int, L} (src/DeltaE.jl:63-295): SURVEY.md §8(f) rank 1.
 * ============================================================================================= */
enum { SL_MAX = 8, SK_MAX = 32, ORC_NB_MAX = 2050 };      /* ORC_NB_MAX: neighbours of a spin of a GraphQuant over dense slices (2 + Nk - 1) */
typedef struct {
    int64_t N;
    int L;
    int64_t dElist[SL_MAX];
    double ft[SL_MAX], T[2 * SL_MAX], Tp[2 * SL_MAX], z, zp;
    aset_t as[2 * SL_MAX];
    int8_t *pos;
    int64_t staged[SK_MAX + 1][3];
    int nstaged;
} decs_t;

static inline int decs_findk(const decs_t *c, int64_t dE) { int64_t a = dE < 0 ? -dE : dE; for (int k = 0; k < c->L; ++k) if (a == c->dElist[k]) return k; return -1; }
static inline double decs_f(const decs_t *c, int k) { return k >= c->L ? c->ft[k - c->L] : 1.0; }
static inline int decs_class(const decs_t *c, int64_t dE, int sbit) { return decs_findk(c, dE) + c->L * (dE > 0 || (dE == 0 && sbit == 1)); }

/* dElist == NULL: the +-J table (RRG.jl:262-266, EA.jl:293) */
static void decs_init_scaled(decs_t *c, sparse_t *X, const uint64_t *s, double beta, const int64_t *dElist, int L, int64_t mul, double div)   /* DeltaE.jl:74-103 */
{
    c->N = X->N;
    int64_t tmp[SK_MAX + 1];
    if (dElist) { c->L = L; for (int k = 0; k < L; ++k) tmp[k] = dElist[k]; }
    else c->L = (int)orc_all_delta_e_pm1(X->K, tmp);
    for (int k = 0; k < c->L; ++k) c->dElist[k] = tmp[k];
    for (int k = 0; k < 2 * c->L; ++k) aset_init(&c->as[k], c->N);
    c->pos = (int8_t *)calloc((size_t)c->N, 1);
    for (int64_t i = 0; i < c->N; ++i) {
        int k = decs_class(c, sparse_delta_energy(X, i), spin_bit(s, i));
        c->pos[i] = (int8_t)k;
        aset_push(&c->as[k], (int32_t)i);
    }
    for (int k = 0; k < c->L; ++k) c->ft[k] = orc_det_exp(-beta * lev_to_f64(c->dElist[k], mul, div));
    c->z = 0.0;
    for (int k = 0; k < 2 * c->L; ++k) { double x = (double)c->as[k].t * decs_f(c, k); c->z += x; c->T[k] = x; }
    c->zp = c->z;
    c->nstaged = 0;
}
static void decs_init_levels(decs_t *c, sparse_t *X, const uint64_t *s, double beta, const int64_t *dElist, int L)
{
    decs_init_scaled(c, X, s, beta, dElist, L, 1, 1.0);
}
static void decs_init(decs_t *c, sparse_t *X, const uint64_t *s, double beta) { decs_init_levels(c, X, s, beta, NULL, 0); }
static void decs_free(decs_t *c) { for (int k = 0; k < 2 * c->L; ++k) aset_free(&c->as[k]); free(c->pos); }

static int64_t decs_rand_move(const decs_t *c, uint64_t seed, uint64_t g, uint32_t replica, int64_t *dE)      /* DeltaE.jl:146-167 */
{
    uint32_t w[4];
    rrr_draw(seed, g, replica, 0, w);
    double r = u53_of(w[0], w[1]) * c->z;
    int k = 0, K2 = 2 * c->L;
    double cT = 0.0;
    for (k = 0; k < K2; ++k) { cT += c->T[k]; if (r < cT) break; }
    if (k == K2) k = K2 - 1;
    if (!(r < cT)) while (c->T[k] == 0) k -= 1;
    *dE = k < c->L ? -c->dElist[k] : c->dElist[k - c->L];
    uint64_t u = ((uint64_t)w[2] << 32) | w[3];
    return c->as[k].v[(int64_t)orc_mulhi64(u, (uint64_t)c->as[k].t)];
}
/* neighbors(X, i) = uA[i]: GraphRRG keeps the neighbours with a non-zero coupling (RRG.jl:133,261); GraphEA removes the
 * repeats of the sorted list and keeps zero couplings (EA.jl:158,292). */
static int sparse_neighbors(const sparse_t *X, int64_t i, int64_t *out)
{
    int n = 0;
    const int32_t *Ax = X->A + i * X->K;
    for (int64_t k = 0; k < X->K; ++k) {
        if (X->ea_form) { if (k > 0 && Ax[k] == Ax[k - 1]) continue; }
        else if (X->J[i * X->K + k] == 0) continue;
        out[n++] = Ax[k];
    }
    return n;
}
static void sparse_spinflip(sparse_t *X, uint64_t *s, int64_t i) { bitflip(s, i); sparse_update_cache(X, s, i); }

static void decs_compute_staged(decs_t *c, sparse_t *X, uint64_t *s, int64_t i)                 /* DeltaE.jl:202-230 */
{
    sparse_spinflip(X, s, i);
    c->nstaged = 0;
    int64_t nb[SK_MAX];
    int nn = sparse_neighbors(X, i, nb);
    for (int q = 0; q < nn; ++q) {
        int64_t j = nb[q];
        int k0 = c->pos[j], k1 = decs_class(c, sparse_delta_energy(X, j), spin_bit(s, j));
        if (k0 == k1) continue;
        c->staged[c->nstaged][0] = j; c->staged[c->nstaged][1] = k0; c->staged[c->nstaged][2] = k1; c->nstaged++;
    }
    int k0 = c->pos[i], k1 = k0 >= c->L ? k0 - c->L : k0 + c->L;
    c->staged[c->nstaged][0] = i; c->staged[c->nstaged][1] = k0; c->staged[c->nstaged][2] = k1; c->nstaged++;
    sparse_spinflip(X, s, i);
}
static double decs_reverse(decs_t *c)                                                           /* DeltaE.jl:184-200 */
{
    double zp = c->z;
    memcpy(c->Tp, c->T, sizeof c->T);
    for (int q = 0; q < c->nstaged; ++q) {
        int k0 = (int)c->staged[q][1], k1 = (int)c->staged[q][2];
        double f0 = decs_f(c, k0), f1 = decs_f(c, k1);
        c->Tp[k0] -= f0; c->Tp[k1] += f1; zp += f1 - f0;
    }
    c->zp = zp;
    return zp;
}
static void decs_apply_staged(decs_t *c)                                                        /* DeltaE.jl:169-182 */
{
    for (int q = 0; q < c->nstaged; ++q) {
        int32_t j = (int32_t)c->staged[q][0];
        int k0 = (int)c->staged[q][1], k1 = (int)c->staged[q][2];
        aset_delete(&c->as[k0], j); aset_push(&c->as[k1], j); c->pos[j] = (int8_t)k1;
    }
    double tmp[2 * SL_MAX];
    memcpy(tmp, c->T, sizeof tmp); memcpy(c->T, c->Tp, sizeof tmp); memcpy(c->Tp, tmp, sizeof tmp);
    c->z = c->zp;
}
static double decs_apply_move(decs_t *c, sparse_t *X, uint64_t *s, int64_t move)                /* DeltaE.jl:232-295 */
{
    sparse_spinflip(X, s, move);
    double zp = c->z;
    int64_t nb[SK_MAX];
    int nn = sparse_neighbors(X, move, nb);
    for (int q = 0; q <= nn; ++q) {
        int32_t j = (int32_t)(q < nn ? nb[q] : move);
        int k0 = c->pos[j];
        int k1 = q < nn ? decs_class(c, sparse_delta_energy(X, j), spin_bit(s, j)) : (k0 >= c->L ? k0 - c->L : k0 + c->L);
        if (q < nn && k0 == k1) continue;
        double f0 = decs_f(c, k0), f1 = decs_f(c, k1);
        c->T[k0] -= f0; c->T[k1] += f1; zp += f1 - f0;
        aset_delete(&c->as[k0], j); aset_push(&c->as[k1], j); c->pos[j] = (int8_t)k1;
    }
    double cc = c->z / zp;
    c->z = zp;
    return cc;
}
static int decs_consistent(const decs_t *c, const sparse_t *X, const uint64_t *s)               /* DeltaE.jl:120-136 */
{
    int64_t total = 0;
    for (int k = 0; k < 2 * c->L; ++k) {
        total += c->as[k].t;
        for (int64_t p = 0; p < c->as[k].t; ++p) { int32_t x = c->as[k].v[p]; if (c->as[k].pos[x] != p + 1 || c->pos[x] != k) return 0; }
    }
    for (int64_t i = 0; i < c->N; ++i) if (decs_class(c, sparse_delta_energy(X, i), spin_bit(s, i)) != c->pos[i]) return 0;
    return total == c->N;
}

/* mode 0: rrrMC(X::SingleGraph) RRRMC.jl:149-219 (staged_thr = 0.5 for a DiscrGraph); mode 1: bklMC RRRMC.jl:311-359.
 * Streams: RRR sub 0 = rand_move, sub 1 = `rand() < c`, sub 2 = rand_skip; g counts iterations (rrrMC) or moves (bklMC).
 * stats = [accepted, staged_its or true moves, iterations done]. */
static int64_t rrr_bkl_sparse_impl(int mode, int form, int64_t N, int64_t K, const int32_t *A, const int32_t *J, const levspec_t *ls,
                                   double beta, int64_t iters, int64_t step, double staged_thr, double staged_thr_fact,
                                   uint64_t seed, uint64_t it0, uint32_t replica,
                                   uint64_t *chunks, int64_t *Es, int64_t *stats, int32_t *cache_out)
{
    sparse_t X = {N, K, A, J, NULL, NULL, -1, form};
    X.lfields = (int64_t *)malloc((size_t)N * 8);
    X.lfields_last = (int64_t *)malloc((size_t)N * 8);
    int64_t E = sparse_energy(&X, chunks);
    decs_t c;
    if (ls) {
        int64_t dElist[SL_MAX];
        const int64_t L = orc_all_delta_e(K, ls->lev, ls->nlev, dElist, SL_MAX);
        if (L < 1) { free(X.lfields); free(X.lfields_last); return -2; }
        decs_init_scaled(&c, &X, chunks, beta, dElist, (int)L, ls->mul, ls->div);
    } else decs_init(&c, &X, chunks, beta);
    int64_t accepted = 0, staged_its = 0, nsamp = 0, it = 0;
    if (mode == 0) {
        const double lambda = staged_thr_fact / (double)N;
        double acc_rate = 0.5;
        for (it = 1; it <= iters; ++it) {
            if (it % step == 0) { Es[nsamp++] = E; if (!ORC_HOOK(it, accepted, E, 0)) break; }
            const uint64_t g = it0 + (uint64_t)it;
            int acc = 0;
            uint32_t w[4];
            if (acc_rate < staged_thr) {
                staged_its += 1;
                double z = c.z;
                int64_t dE, move = decs_rand_move(&c, seed, g, replica, &dE);
                decs_compute_staged(&c, &X, chunks, move);
                double cc = z / decs_reverse(&c);
                rrr_draw(seed, g, replica, 1, w);
                if (u53_of(w[0], w[1]) < cc) { sparse_spinflip(&X, chunks, move); decs_apply_staged(&c); E += dE; accepted++; acc = 1; }
            } else {
                int64_t dE, move = decs_rand_move(&c, seed, g, replica, &dE);
                double cc = decs_apply_move(&c, &X, chunks, move);
                rrr_draw(seed, g, replica, 1, w);
                if (u53_of(w[0], w[1]) < cc) { E += dE; accepted++; acc = 1; }
                else decs_apply_move(&c, &X, chunks, move);
            }
            acc_rate = acc_rate * (1 - lambda) + (double)acc * lambda;
        }
        it = iters;
    } else {
        int64_t nextstep = step, m = 0;
        it = 0;
        while (it < iters) {
            m += 1;
            const uint64_t g = it0 + (uint64_t)m;
            uint32_t w[4];
            rrr_draw(seed, g, replica, 2, w);
            /* rand_skip: floor(Int, log1p(-rand()) / log1p(-z / N)), DeltaE.jl:141-144 */
            double skipf = __builtin_floor(orc_det_log1p(-u53_of(w[0], w[1])) / orc_det_log1p(-c.z / (double)N));
            int64_t skip = skipf >= 9.0e18 ? (int64_t)9.0e18 : (int64_t)skipf;
            int64_t dE, move = decs_rand_move(&c, seed, g, replica, &dE);
            int out = 0;
            while (it + skip + 1 >= nextstep) {
                Es[nsamp++] = E; if (!ORC_HOOK(nextstep, accepted, E, 0)) { out = 1; break; }
                nextstep += step;
                if (nextstep > iters) { out = 1; break; }
            }
            if (out) break;
            decs_apply_move(&c, &X, chunks, move);
            it += skip + 1;
            E += dE;
            accepted += 1;
        }
        staged_its = accepted;
    }
    if (stats) { stats[0] = accepted; stats[1] = staged_its; stats[2] = it; }
    if (cache_out) {
        for (int64_t i = 0; i < N; ++i) cache_out[i] = c.pos[i];
        for (int k = 0; k < 2 * c.L; ++k) cache_out[N + k] = (int32_t)c.as[k].t;
    }
    int ok = decs_consistent(&c, &X, chunks);
    decs_free(&c);
    free(X.lfields); free(X.lfields_last);
    return ok ? nsamp : -1;
}
ORC_API int64_t orc_rrr_bkl_sparse(int mode, int form, int64_t N, int64_t K, const int32_t *A, const int32_t *J,
                                   double beta, int64_t iters, int64_t step, double staged_thr, double staged_thr_fact,
                                   uint64_t seed, uint64_t it0, uint32_t replica,
                                   uint64_t *chunks, int64_t *Es, int64_t *stats, int32_t *cache_out)
{
    return rrr_bkl_sparse_impl(mode, form, N, K, A, J, NULL, beta, iters, step, staged_thr, staged_thr_fact, seed, it0, replica, chunks, Es,
                               stats, cache_out);
}
ORC_API int64_t orc_rrr_bkl_sparse_lev(int mode, int form, int64_t N, int64_t K, const int32_t *A, const int32_t *J,
                                       const int32_t *lev, int64_t nlev, int64_t mul, double div,
                                       double beta, int64_t iters, int64_t step, double staged_thr, double staged_thr_fact,
                                       uint64_t seed, uint64_t it0, uint32_t replica,
                                       uint64_t *chunks, int64_t *Es, int64_t *stats, int32_t *cache_out)
{
    const levspec_t ls = {lev, nlev, mul, div};
    return rrr_bkl_sparse_impl(mode, form, N, K, A, J, &ls, beta, iters, step, staged_thr, staged_thr_fact, seed, it0, replica, chunks, Es,
                               stats, cache_out);
}

/* standardMC (src/RRRMC.jl:81-127) on a stand-alone GraphRRG / GraphEA with general levels: ET = Int or DFloat64, the energy is tracked
 * in units (integer arithmetic, as DFloat64's + does: src/DFloats.jl:29-30); `-beta * dE` promotes to Float64 (:42-43).  SITE stream for
 * the site, ACCEPT_F64 stream for rand() (the bit-plane ACCEPT stream belongs to the +-J kernel's threshold table). */
ORC_API int64_t orc_standard_mc_lev(int form, int64_t N, int64_t K, const int32_t *A, const int32_t *J, int64_t mul, double div,
                                    double beta, int64_t iters, int64_t step, uint64_t seed, uint64_t it0, uint32_t replica,
                                    uint64_t *chunks, int64_t *Es, int64_t *accepted_out)
{
    sparse_t X = {N, K, A, J, NULL, NULL, -1, form};
    X.lfields = (int64_t *)malloc((size_t)N * 8);
    X.lfields_last = (int64_t *)malloc((size_t)N * 8);
    int64_t E = sparse_energy(&X, chunks);
    int64_t accepted = 0, nsamp = 0;
    for (int64_t it = 1; it <= iters; ++it) {
        if (it % step == 0) Es[nsamp++] = E;
        const uint64_t g = it0 + (uint64_t)it;
        const int64_t i = orc_site(seed, g, N);
        const int64_t dE = sparse_delta_energy(&X, i);
        const double x = -beta * lev_to_f64(dE, mul, div);
        const int acc = (x >= 0) || (orc_rand53(seed, g, replica) < orc_det_exp(x));      /* RRRMC.jl:39 */
        if (!acc) continue;
        sparse_spinflip(&X, chunks, i);
        E += dE;
        accepted += 1;
    }
    if (accepted_out) *accepted_out = accepted;
    const int ok = E == sparse_energy(&X, chunks);
    free(X.lfields); free(X.lfields_last);
    return ok ? nsamp : -1;
}

/* =============================================================================================
 * rrrMC(X::DoubleGraph) (src/RRRMC.jl:221-290) on GraphRRGNormalDiscretized / GraphEANormalDiscretized with integer levels
 * (src/graphs/RRG.jl:285-500, src/graphs/EA.jl:311-532): Gaussian couplings split by discretize (Common.jl:38-72) into an inner
 * DiscrGraph X0 = GraphRRG{Int,LEV,K}(A, dJ) — which drives the DeltaECache — and a Float64 residual rJ whose local-field
 * cache (with its own lfields_last / move_last) gives delta_energy_residual.  SURVEY.md §8f rank 3.
 * spinflip!(X, C, move) = bit flip + update_cache!(X0) + update_cache_residual!(X): the combined update_cache! (RRG.jl:362-428)
 * is the two separate updates whenever the two move_last agree and falls back to them otherwise (:366-370).
 * ============================================================================================= */
static void dbl_spinflip(sparse_t *X0, spf_t *X1, uint64_t *s, int64_t move)
{
    bitflip(s, move);
    sparse_update_cache(X0, s, move);
    spf_update_cache(X1, s, move);
}
/* apply_move!(X::DoubleGraph, ...): DeltaE.jl:232-295 — decs_apply_move with the full graph's spinflip! */
static double dbl_apply_move(decs_t *c, sparse_t *X0, spf_t *X1, uint64_t *s, int64_t move)
{
    dbl_spinflip(X0, X1, s, move);
    double zp = c->z;
    int64_t nb[SK_MAX];
    int nn = sparse_neighbors(X0, move, nb);
    for (int q = 0; q <= nn; ++q) {
        int32_t j = (int32_t)(q < nn ? nb[q] : move);
        int k0 = c->pos[j];
        int k1 = q < nn ? decs_class(c, sparse_delta_energy(X0, j), spin_bit(s, j)) : (k0 >= c->L ? k0 - c->L : k0 + c->L);
        if (q < nn && k0 == k1) continue;
        double f0 = decs_f(c, k0), f1 = decs_f(c, k1);
        c->T[k0] -= f0; c->T[k1] += f1; zp += f1 - f0;
        aset_delete(&c->as[k0], j); aset_push(&c->as[k1], j); c->pos[j] = (int8_t)k1;
    }
    double cc = c->z / zp;
    c->z = zp;
    return cc;
}

ORC_API double orc_dbl_energy_scaled(int form, int64_t N, int64_t K, const int32_t *A, const int32_t *dJ, const double *rJ, const uint64_t *chunks,
                                     int64_t mul, double div)
{
    sparse_t X0 = {N, K, A, dJ, NULL, NULL, -1, form};
    spf_t X1 = {N, K, A, rJ, NULL, NULL, -1, form, NULL, 1, 1.0, NULL, NULL};
    X0.lfields = (int64_t *)malloc((size_t)N * 8); X0.lfields_last = (int64_t *)malloc((size_t)N * 8);
    X1.lfields = (double *)malloc((size_t)N * 8); X1.lfields_last = (double *)malloc((size_t)N * 8);
    const int64_t E0 = sparse_energy(&X0, chunks);
    const double E1 = spf_energy(&X1, chunks);
    free(X0.lfields); free(X0.lfields_last); free(X1.lfields); free(X1.lfields_last);
    return lev_to_f64(E0, mul, div) + E1;                     /* convert(Float64, E0 + E1): RRG.jl:359 */
}
ORC_API double orc_dbl_energy(int form, int64_t N, int64_t K, const int32_t *A, const int32_t *dJ, const double *rJ, const uint64_t *chunks)
{
    return orc_dbl_energy_scaled(form, N, K, A, dJ, rJ, chunks, 1, 1.0);
}

/* stats = [accepted, staged iterations]; cache_out = pos[N] then the 2L class sizes */
ORC_API int64_t orc_rrr_double_sparse_scaled(int form, int64_t N, int64_t K, const int32_t *A, const int32_t *dJ, const double *rJ,
                                             const int32_t *lev, int64_t nlev, int64_t mul, double div, double beta, int64_t iters, int64_t step,
                                             double staged_thr, double staged_thr_fact, uint64_t seed, uint64_t it0, uint32_t replica,
                                             uint64_t *chunks, double *Es, int64_t *stats, int32_t *cache_out)
{
    int64_t dElist[SL_MAX];
    const int64_t L = orc_all_delta_e(K, lev, nlev, dElist, SL_MAX);
    if (L < 1) return -2;
    sparse_t X0 = {N, K, A, dJ, NULL, NULL, -1, form};
    spf_t X1 = {N, K, A, rJ, NULL, NULL, -1, form, NULL, 1, 1.0, NULL, NULL};
    X0.lfields = (int64_t *)malloc((size_t)N * 8); X0.lfields_last = (int64_t *)malloc((size_t)N * 8);
    X1.lfields = (double *)malloc((size_t)N * 8); X1.lfields_last = (double *)malloc((size_t)N * 8);
    double E = lev_to_f64(sparse_energy(&X0, chunks), mul, div);
    E = E + spf_energy(&X1, chunks);
    decs_t c;
    decs_init_scaled(&c, &X0, chunks, beta, dElist, (int)L, mul, div);
    const double lambda = staged_thr_fact / (double)N;
    double acc_rate = 0.5;
    int64_t accepted = 0, staged_its = 0, nsamp = 0;
    for (int64_t it = 1; it <= iters; ++it) {
        if (it % step == 0) { Es[nsamp++] = E; if (!ORC_HOOK(it, accepted, E, 0)) break; }
        const uint64_t g = it0 + (uint64_t)it;
        int acc = 0;
        if (acc_rate < staged_thr) {
            staged_its += 1;
            double z = c.z;
            int64_t dE0, move = decs_rand_move(&c, seed, g, replica, &dE0);
            decs_compute_staged(&c, &X0, chunks, move);             /* step_rrr(X0, C, cache): flips X0's cache twice */
            double cc = z / decs_reverse(&c);
            double dE1 = -X1.lfields[move];                         /* delta_energy_residual: RRG.jl:468-476 */
            if (accept_cx(cc, -beta * dE1, seed, g, replica)) {
                dbl_spinflip(&X0, &X1, chunks, move);
                decs_apply_staged(&c);
                E += lev_to_f64(dE0, mul, div) + dE1;
                accepted++; acc = 1;
            }
        } else {
            int64_t dE0, move = decs_rand_move(&c, seed, g, replica, &dE0);
            double dE1 = -X1.lfields[move];
            double cc = dbl_apply_move(&c, &X0, &X1, chunks, move);
            if (accept_cx(cc, -beta * dE1, seed, g, replica)) { E += lev_to_f64(dE0, mul, div) + dE1; accepted++; acc = 1; }
            else dbl_apply_move(&c, &X0, &X1, chunks, move);
        }
        acc_rate = acc_rate * (1 - lambda) + (double)acc * lambda;
    }
    if (stats) { stats[0] = accepted; stats[1] = staged_its; }
    if (cache_out) {
        for (int64_t i = 0; i < N; ++i) cache_out[i] = c.pos[i];
        for (int k = 0; k < 2 * c.L; ++k) cache_out[N + k] = (int32_t)c.as[k].t;
    }
    int ok = decs_consistent(&c, &X0, chunks);
    /* the residual cache must equal a recomputation up to rounding (the reference's invariant, runtests.jl:12-20) */
    decs_free(&c);
    free(X0.lfields); free(X0.lfields_last); free(X1.lfields); free(X1.lfields_last);
    return ok ? nsamp : -1;
}
ORC_API int64_t orc_rrr_double_sparse(int form, int64_t N, int64_t K, const int32_t *A, const int32_t *dJ, const double *rJ,
                                      const int32_t *lev, int64_t nlev, double beta, int64_t iters, int64_t step,
                                      double staged_thr, double staged_thr_fact, uint64_t seed, uint64_t it0, uint32_t replica,
                                      uint64_t *chunks, double *Es, int64_t *stats, int32_t *cache_out)
{
    return orc_rrr_double_sparse_scaled(form, N, K, A, dJ, rJ, lev, nlev, 1, 1.0, beta, iters, step, staged_thr, staged_thr_fact, seed, it0,
                                        replica, chunks, Es, stats, cache_out);
}

/* standardMC (src/RRRMC.jl:81-127) on the same DoubleGraphs: delta_energy(X, C, move) = convert(Float64, dE0 + dE1)
 * (RRG.jl:493-497, EA.jl:523-527), spinflip! = bit flip + both caches.  SITE / ACCEPT_F64 streams as for the other Float64 models. */
ORC_API int64_t orc_standard_mc_dbl(int form, int64_t N, int64_t K, const int32_t *A, const int32_t *dJ, const double *rJ,
                                    int64_t mul, double div, double beta, int64_t iters, int64_t step, uint64_t seed, uint64_t it0,
                                    uint32_t replica, uint64_t *chunks, double *Es, int64_t *accepted_out)
{
    sparse_t X0 = {N, K, A, dJ, NULL, NULL, -1, form};
    spf_t X1 = {N, K, A, rJ, NULL, NULL, -1, form, NULL, 1, 1.0, NULL, NULL};
    X0.lfields = (int64_t *)malloc((size_t)N * 8); X0.lfields_last = (int64_t *)malloc((size_t)N * 8);
    X1.lfields = (double *)malloc((size_t)N * 8); X1.lfields_last = (double *)malloc((size_t)N * 8);
    double E = lev_to_f64(sparse_energy(&X0, chunks), mul, div);
    E = E + spf_energy(&X1, chunks);
    int64_t accepted = 0, nsamp = 0;
    for (int64_t it = 1; it <= iters; ++it) {
        if (it % step == 0) Es[nsamp++] = E;
        const uint64_t g = it0 + (uint64_t)it;
        const int64_t i = orc_site(seed, g, N);
        const double dE = lev_to_f64(sparse_delta_energy(&X0, i), mul, div) + (-X1.lfields[i]);
        const double x = -beta * dE;
        const int acc = (x >= 0) || (orc_rand53(seed, g, replica) < orc_det_exp(x));      /* RRRMC.jl:39 */
        if (!acc) continue;
        dbl_spinflip(&X0, &X1, chunks, i);
        E += dE;
        accepted += 1;
    }
    if (accepted_out) *accepted_out = accepted;
    free(X0.lfields); free(X0.lfields_last); free(X1.lfields); free(X1.lfields_last);
    return nsamp;
}

/* =============================================================================================
 * wtmMC (waiting-time method; src/RRRMC.jl:376-426, src/WaitingTimes.jl) on the DiscrGraphs GraphRRG / GraphEA.
 * SURVEY.md §8f rank 4.  Every spin carries the time of its next flip, t_i = t + gen_wt(tau_i) with
 * tau = max(1, exp(beta dE)) and gen_wt(tau) = -tau log1p(-rand()) (WaitingTimes.jl:16-22); the spin with the smallest time moves.
 * The reference keeps the times in DataStructures' MutableBinaryMinHeap (third party, v0.18, not under /root/reference): only its
 * contract "top = smallest key" matters (ties have probability zero; here the lower site index wins), so this restatement
 * simply scans for the minimum.  WTM stream: the n-th uniform of replica r in call `call` is the 53-bit word (n & 1) of
 *   ctr = (lo32(n >> 1), hi32(n >> 1), r, TAG_WTM | call << 8);
 * n = i for the initial time of spin i (THeap(X, C, beta), :26-36), then one draw per updated spin in the order of update_heap!
 * (:40-52: the moved spin, then its neighbours).  exp and log1p are the deterministic ones of philox_contract.h.
 * ============================================================================================= */
enum { ORC_TAG_WTM = 11 };
static double wtm_uniform(uint64_t seed, uint64_t n, uint32_t replica, uint32_t call)
{
    uint32_t w[4];
    uint64_t blk = n >> 1;
    orc_draw(seed, (uint32_t)blk, (uint32_t)(blk >> 32), replica, (uint32_t)ORC_TAG_WTM | (call << 8), w);
    unsigned h = (unsigned)(n & 1u);
    return u53_of(w[2 * h], w[2 * h + 1]);
}
static int64_t wtm_lev_mul = 1;          /* level units of the running orc_wtm_mc_sparse* call (set at entry, single-threaded oracle) */
static double wtm_lev_div = 1.0;
static inline double wtm_tau(double beta, int64_t dE) { double e = orc_det_exp(beta * lev_to_f64(dE, wtm_lev_mul, wtm_lev_div)); return e > 1.0 ? e : 1.0; }   /* tauDE: :16 */
static inline double wtm_gen(double tau, double u) { return -tau * orc_det_log1p(-u); }                                          /* gen_wt: :18-22 */

/* Es[samples] energies at global times k*step/N; stats = [num_moves, samples taken, final tracked energy]; t_out = final global time */
ORC_API int64_t orc_wtm_mc_sparse_lev(int form, int64_t N, int64_t K, const int32_t *A, const int32_t *J, int64_t mul, double div, double beta,
                                      int64_t samples, double step, uint64_t seed, uint32_t call, uint32_t replica,
                                      uint64_t *chunks, int64_t *Es, int64_t *stats, double *t_out)
{
    wtm_lev_mul = mul; wtm_lev_div = div;
    sparse_t X = {N, K, A, J, NULL, NULL, -1, form};
    X.lfields = (int64_t *)malloc((size_t)N * 8);
    X.lfields_last = (int64_t *)malloc((size_t)N * 8);
    int64_t E = sparse_energy(&X, chunks);
    double *tm = (double *)malloc((size_t)N * sizeof(double));
    uint64_t nd = 0;
    for (int64_t i = 0; i < N; ++i) tm[i] = wtm_gen(wtm_tau(beta, sparse_delta_energy(&X, i)), wtm_uniform(seed, nd++, replica, call));
    step /= (double)N;
    const double tmax = step * (double)samples;
    double t = 0.0, nextstep = step;
    int64_t num_moves = 0, nsamp = 0;
    int out = 0;
    while (t < tmax && !out) {
        int64_t move = 0;
        for (int64_t i = 1; i < N; ++i) if (tm[i] < tm[move]) move = i;          /* pick_next: top_with_handle */
        const double tp = tm[move];
        while (tp >= nextstep) {
            Es[nsamp++] = E; if (!ORC_HOOK(nextstep, num_moves, E, 0)) { out = 1; break; }
            nextstep += step;
            if (nextstep > tmax + 1e-10) { out = 1; break; }
        }
        if (out) break;
        t = tp;
        /* update_heap!: WaitingTimes.jl:40-52 */
        const int64_t dE = sparse_delta_energy(&X, move);
        sparse_spinflip(&X, chunks, move);
        tm[move] = t + wtm_gen(wtm_tau(beta, -dE), wtm_uniform(seed, nd++, replica, call));
        int64_t nb[SK_MAX];
        int nn = sparse_neighbors(&X, move, nb);
        for (int q = 0; q < nn; ++q) {
            int64_t j = nb[q];
            tm[j] = t + wtm_gen(wtm_tau(beta, sparse_delta_energy(&X, j)), wtm_uniform(seed, nd++, replica, call));
        }
        E += dE;
        num_moves += 1;
    }
    if (stats) { stats[0] = num_moves; stats[1] = nsamp; stats[2] = E; }
    if (t_out) *t_out = t;
    free(tm); free(X.lfields); free(X.lfields_last);
    return nsamp;
}
ORC_API int64_t orc_wtm_mc_sparse(int form, int64_t N, int64_t K, const int32_t *A, const int32_t *J, double beta, int64_t samples,
                                  double step, uint64_t seed, uint32_t call, uint32_t replica,
                                  uint64_t *chunks, int64_t *Es, int64_t *stats, double *t_out)
{
    return orc_wtm_mc_sparse_lev(form, N, K, A, J, 1, 1.0, beta, samples, step, seed, call, replica, chunks, Es, stats, t_out);
}

/* =============================================================================================
 * extremal_opt (tau-EO; src/RRRMC.jl:474-521) on the DiscrGraphs GraphRRG / GraphEA with EOCache{Int,L} (src/DeltaE.jl:412-555).
 * SURVEY.md §8f rank 4.  Spins are ranked by dE (classes of equal dE, ArraySets as in DeltaECache); a rank i is drawn with
 * probability ~ i^-tau from the cumulative table ftau[i] = sum_{j<=i} j^-tau (computed by the caller: DeltaE.jl:444-445), the class
 * holding that rank is found and a uniform member of it flips — always.  RRR stream sub 3: words 0,1 -> rand() of `r`, words 2,3 ->
 * the member index.  Returns the samples (E at iterations k*step, before the move: the hook's argument), Emin, Cmin, itmin.
 * ============================================================================================= */
typedef struct {
    int64_t N;
    int L, has_zero, K2;
    int64_t dElist[SL_MAX];
    aset_t as[2 * SL_MAX];
    int8_t *pos;
} eoc_t;
static inline int eoc_findks(const eoc_t *c, int64_t dE)                                   /* findks: DeltaE.jl:412-421 (0-based class) */
{
    int64_t a = dE < 0 ? -dE : dE;
    int ak = 0;
    for (int k = 0; k < c->L; ++k) if (c->dElist[k] == a) ak = k + 1;
    return (dE >= 0 ? ak + c->L - c->has_zero : c->L + 1 - ak) - 1;
}

ORC_API int64_t orc_extremal_opt_sparse_lev(int form, int64_t N, int64_t K, const int32_t *A, const int32_t *J, const int32_t *lev, int64_t nlev,
                                            const double *ftau, int64_t iters, int64_t step, uint64_t seed, uint64_t it0, uint32_t replica,
                                            uint64_t *chunks, int64_t *Es, int64_t *Emin_out, uint64_t *Cmin, int64_t *itmin_out)
{
    sparse_t X = {N, K, A, J, NULL, NULL, -1, form};
    X.lfields = (int64_t *)malloc((size_t)N * 8);
    X.lfields_last = (int64_t *)malloc((size_t)N * 8);
    const int64_t nch = (N + 63) / 64;
    int64_t E = sparse_energy(&X, chunks), Emin = E, itmin = 0;
    memcpy(Cmin, chunks, (size_t)nch * 8);
    eoc_t c;
    memset(&c, 0, sizeof c);
    c.N = N;
    int64_t tmp[SK_MAX + 1 > SL_MAX ? SK_MAX + 1 : SL_MAX];
    c.L = lev ? (int)orc_all_delta_e(K, lev, nlev, tmp, SL_MAX) : (int)orc_all_delta_e_pm1(K, tmp);
    if (c.L < 1) { free(X.lfields); free(X.lfields_last); return -2; }
    for (int k = 0; k < c.L; ++k) c.dElist[k] = tmp[k];
    c.has_zero = c.dElist[0] == 0;
    c.K2 = 2 * c.L - c.has_zero;
    for (int k = 0; k < c.K2; ++k) aset_init(&c.as[k], N);
    c.pos = (int8_t *)calloc((size_t)N, 1);
    for (int64_t i = 0; i < N; ++i) {                                     /* EOCache: DeltaE.jl:431-448 */
        int k = eoc_findks(&c, sparse_delta_energy(&X, i));
        c.pos[i] = (int8_t)k;
        aset_push(&c.as[k], (int32_t)i);
    }
    const double z = ftau[N - 1];
    int64_t nsamp = 0;
    for (int64_t it = 1; it <= iters; ++it) {
        if (it % step == 0) { Es[nsamp++] = E; if (!ORC_HOOK(it, 0, E, Emin)) break; }
        const uint64_t g = it0 + (uint64_t)it;
        uint32_t w[4];
        rrr_draw(seed, g, replica, 3, w);
        /* rand_move: DeltaE.jl:473-507 */
        const double r = (1 - u53_of(w[0], w[1])) * z;
        int64_t lo = 0, hi = N;                                           /* searchsortedfirst(ftau, r): first index with ftau >= r */
        while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (ftau[mid] < r) lo = mid + 1; else hi = mid; }
        int64_t rank = lo + 1;
        if (rank > N) rank = N;
        int k = -1;
        int64_t t = 0;
        while (rank > t) { k += 1; t += c.as[k].t; }
        const int64_t dE = k < c.L ? -c.dElist[c.L - 1 - k] : c.dElist[k - c.L + c.has_zero];
        const uint64_t u = ((uint64_t)w[2] << 32) | w[3];
        const int64_t move = c.as[k].v[(int64_t)orc_mulhi64(u, (uint64_t)c.as[k].t)];
        /* apply_move!: DeltaE.jl:509-541 */
        sparse_spinflip(&X, chunks, move);
        int64_t nb[SK_MAX];
        int nn = sparse_neighbors(&X, move, nb);
        for (int q = 0; q <= nn; ++q) {
            int32_t j = (int32_t)(q < nn ? nb[q] : move);
            int k0 = c.pos[j], k1 = eoc_findks(&c, sparse_delta_energy(&X, j));
            if (k0 == k1) continue;
            aset_delete(&c.as[k0], j); aset_push(&c.as[k1], j); c.pos[j] = (int8_t)k1;
        }
        E += dE;
        if (E < Emin) { Emin = E; memcpy(Cmin, chunks, (size_t)nch * 8); itmin = it; }
    }
    int ok = E == sparse_energy(&X, chunks);                              /* tracked energy == energy(X, C) */
    for (int64_t i = 0; i < N && ok; ++i) ok = eoc_findks(&c, sparse_delta_energy(&X, i)) == c.pos[i];
    if (Emin_out) *Emin_out = Emin;
    if (itmin_out) *itmin_out = itmin;
    for (int k = 0; k < c.K2; ++k) aset_free(&c.as[k]);
    free(c.pos); free(X.lfields); free(X.lfields_last);
    return ok ? nsamp : -1;
}
ORC_API int64_t orc_extremal_opt_sparse(int form, int64_t N, int64_t K, const int32_t *A, const int32_t *J, const double *ftau,
                                        int64_t iters, int64_t step, uint64_t seed, uint64_t it0, uint32_t replica,
                                        uint64_t *chunks, int64_t *Es, int64_t *Emin_out, uint64_t *Cmin, int64_t *itmin_out)
{
    return orc_extremal_opt_sparse_lev(form, N, K, A, J, NULL, 0, ftau, iters, step, seed, it0, replica, chunks, Es, Emin_out, Cmin, itmin_out);
}

/* =============================================================================================
 * Continuous-energy samplers on the Float64 sparse models GraphRRGNormal / GraphEANormal — the reference's second experiment
 * (scripts/scripts.jl:152-281 test_RRGCont): rrrMC(X::SingleGraph) (src/RRRMC.jl:149-219, staged_thr = 0.8 for a SimpleGraph),
 * bklMC (:311-359) and wtmMC (:376-426) over DeltaECacheCont + DynamicSampler (src/DeltaE.jl:297-410) / THeap.
 *   mode 0 rrrMC: RRR stream sub 0 (getel uniform), sub 1 (`rand() < c`), g = iteration
 *   mode 1 bklMC: sub 2 (rand_skip), sub 0 (getel); g = move counter
 *   mode 2 wtmMC: WTM stream (n-th uniform of the call), `iters` = samples, `stepf` = step in sweeps
 * neighbors(X, i) = A[i] for GraphRRGNormal (RRG.jl:627), the de-duplicated uA[i] for GraphEANormal (EA.jl:680).
 * stats = [accepted / moves, staged iterations / true moves, iterations done]; t_out = wtmMC's final global time.
 * Returns the number of samples, -1 on DynamicSampler's precision-loss error.
 * ============================================================================================= */
static int spf_neighbors(const spf_t *X, int64_t i, int64_t *out)
{
    int n = 0;
    if (X->Q) {                                        /* neighbors(X::GraphQuant, i): QT.jl:288-321 — the two Trotter neighbours, then the slice's */
        const quant_t *Q = (const quant_t *)X->Q;
        int64_t j1, j2;
        qt_neighbors(&Q->X0, i, &j1, &j2);
        out[n++] = j1; out[n++] = j2;
        const int64_t k = i / Q->Nk, x = i % Q->Nk;
        if (Q->F1) {                                   /* sparse Float64 slices: neighbors(X1[k], x) = A[x] (RRG.jl:627) / uA[x] (EA.jl:680) */
            const spf_t *F = &Q->F1[k];
            const int32_t *Ax = F->A + x * F->K;
            for (int64_t q = 0; q < F->K; ++q) {
                if (F->ea_form && q > 0 && Ax[q] == Ax[q - 1]) continue;
                out[n++] = Ax[q] + k * Q->Nk;
            }
            return n;
        }
        if (!Q->X1) {                                  /* GraphSK / GraphSKNormal slices: AllButOne(Nk, x) (SK.jl:142,297), in index order */
            for (int64_t j = 0; j < Q->Nk; ++j) if (j != x) out[n++] = j + k * Q->Nk;
            return n;
        }
        int64_t nb1[SK_MAX];
        const int nn = sparse_neighbors(&Q->X1[k], x, nb1);
        for (int q = 0; q < nn; ++q) out[n++] = nb1[q] + k * Q->Nk;
        return n;
    }
    const int32_t *Ax = X->A + i * X->K;
    for (int64_t k = 0; k < X->K; ++k) {
        if (X->ea_form && k > 0 && Ax[k] == Ax[k - 1]) continue;
        out[n++] = Ax[k];
    }
    return n;
}
static inline void spf_spinflip(spf_t *X, uint64_t *s, int64_t i)
{
    if (X->Q) { quant_spinflip((quant_t *)X->Q, s, i); return; }
    bitflip(s, i);
    if (X->X0) sparse_update_cache(X->X0, s, i);
    spf_update_cache(X, s, i);
}
static inline double spf_dE(const spf_t *X, int64_t i)                                          /* RRG.jl:619-625; DoubleGraph: :493-497 */
{
    if (X->Q) return qt_delta_energy(&((const quant_t *)X->Q)->X0, X->cur_s, i) + quant_residual((const quant_t *)X->Q, i);   /* QT.jl:283-286 */
    if (X->X0) return lev_to_f64(sparse_delta_energy(X->X0, i), X->lev_mul, X->lev_div) + (-X->lfields[i]);
    return -X->lfields[i];
}
/* apply_move!(X, C, move, cache::DeltaECacheCont): DeltaE.jl:376-410; returns c = z / z' */
static double cont_apply_move(spf_t *X, uint64_t *s, dyns_t *ds, double *dEs, double beta, int64_t move)
{
    spf_spinflip(X, s, move);
    const double z = ds->z;
    dEs[move] = spf_dE(X, move);
    dyns_set(ds, move, prior_of(beta * dEs[move]));
    int64_t nb[ORC_NB_MAX];
    int nn = spf_neighbors(X, move, nb);
    for (int q = 0; q < nn; ++q) {
        int64_t j = nb[q];
        dEs[j] = spf_dE(X, j);
        dyns_set(ds, j, prior_of(beta * dEs[j]));
    }
    return z / ds->z;
}

static int64_t cont_sparse_impl(int mode, int form, int64_t N, int64_t K, const int32_t *A, const double *J,
                                const int32_t *dJ, int64_t mul, double div, double beta,
                                int64_t iters, int64_t step, double stepf, double staged_thr, double staged_thr_fact,
                                uint64_t seed, uint64_t it0, uint32_t call, uint32_t replica,
                                uint64_t *chunks, double *Es, int64_t *stats, double *t_out, quant_t *Q)
{
    spf_t X = {N, K, A, J, NULL, NULL, -1, form, NULL, 1, 1.0, NULL, NULL};
    X.lfields = (double *)malloc((size_t)N * 8);
    X.lfields_last = (double *)malloc((size_t)N * 8);
    sparse_t X0 = {N, K, A, dJ, NULL, NULL, -1, form};
    double E = 0.0;
    if (Q) {                                                   /* a GraphQuant: energy QT.jl:185-199 */
        X.Q = Q; X.cur_s = chunks;
        E = quant_energy(Q, chunks);
    } else if (dJ) {                                                  /* energy(X::DoubleGraph, C) = convert(Float64, E0 + E1): RRG.jl:326-360 */
        X0.lfields = (int64_t *)malloc((size_t)N * 8);
        X0.lfields_last = (int64_t *)malloc((size_t)N * 8);
        X.X0 = &X0; X.lev_mul = mul; X.lev_div = div;
        E = lev_to_f64(sparse_energy(&X0, chunks), mul, div);
        E = E + spf_energy(&X, chunks);
    } else {
        E = spf_energy(&X, chunks);
    }
    int64_t accepted = 0, second = 0, nsamp = 0, itdone = 0, bad = 0;
    double t = 0.0;
    if (mode == 2) {
        /* wtmMC: WaitingTimes.jl with tau = max(1, exp(beta dE)) evaluated per update */
        double *tm = (double *)malloc((size_t)N * 8);
        uint64_t nd = 0;
        for (int64_t i = 0; i < N; ++i) {
            double e = orc_det_exp(beta * spf_dE(&X, i));
            tm[i] = wtm_gen(e > 1.0 ? e : 1.0, wtm_uniform(seed, nd++, replica, call));
        }
        const double st = stepf / (double)N, tmax = st * (double)iters;
        double nextstep = st;
        int out = 0;
        while (t < tmax && !out) {
            int64_t move = 0;
            for (int64_t i = 1; i < N; ++i) if (tm[i] < tm[move]) move = i;
            const double tp = tm[move];
            while (tp >= nextstep) {
                Es[nsamp++] = E; if (!ORC_HOOK(nextstep, accepted, E, 0)) { out = 1; break; }
                nextstep += st;
                if (nextstep > tmax + 1e-10) { out = 1; break; }
            }
            if (out) break;
            t = tp;
            const double dE = spf_dE(&X, move);
            spf_spinflip(&X, chunks, move);
            double e = orc_det_exp(beta * -dE);
            tm[move] = t + wtm_gen(e > 1.0 ? e : 1.0, wtm_uniform(seed, nd++, replica, call));
            int64_t nb[ORC_NB_MAX];
            int nn = spf_neighbors(&X, move, nb);
            for (int q = 0; q < nn; ++q) {
                e = orc_det_exp(beta * spf_dE(&X, nb[q]));
                tm[nb[q]] = t + wtm_gen(e > 1.0 ? e : 1.0, wtm_uniform(seed, nd++, replica, call));
            }
            E += dE;
            accepted += 1;
        }
        free(tm);
        second = accepted; itdone = nsamp;
    } else {
        double *dEs = (double *)malloc((size_t)N * 8);
        dyns_t ds;
        dyns_init(&ds, N);
        for (int64_t i = 0; i < N; ++i) { dEs[i] = spf_dE(&X, i); ds.v[i] = prior_of(beta * dEs[i]); }    /* DeltaECacheCont: DeltaE.jl:304-313 */
        dyns_refresh(&ds);
        if (mode == 0) {
            const double lambda = staged_thr_fact / (double)N;
            double acc_rate = 0.5;
            int64_t st_j[SK_MAX + 1];
            double st_dE[SK_MAX + 1], st_p[SK_MAX + 1];
            for (int64_t it = 1; it <= iters && !bad; ++it) {
                if (it % step == 0) { Es[nsamp++] = E; if (!ORC_HOOK(it, accepted, E, 0)) break; }
                const uint64_t g = it0 + (uint64_t)it;
                uint32_t w[4];
                rrr_draw(seed, g, replica, 0, w);
                int acc = 0;
                int64_t move = dyns_getel(&ds, u53_of(w[0], w[1]));                    /* rand_move: DeltaE.jl:327-333 */
                if (move < 0) { bad = 1; break; }
                const double dE = dEs[move];
                if (acc_rate < staged_thr) {
                    second += 1;
                    const double z = ds.z;
                    spf_spinflip(&X, chunks, move);                                    /* compute_staged!: DeltaE.jl:357-374 */
                    int ns = 0;
                    st_j[ns] = move; st_dE[ns] = spf_dE(&X, move); st_p[ns] = prior_of(beta * st_dE[ns]); ns++;
                    int64_t nb[ORC_NB_MAX];
                    int nn = spf_neighbors(&X, move, nb);
                    for (int q = 0; q < nn; ++q) { st_j[ns] = nb[q]; st_dE[ns] = spf_dE(&X, nb[q]); st_p[ns] = prior_of(beta * st_dE[ns]); ns++; }
                    spf_spinflip(&X, chunks, move);
                    double zp = ds.z;                                                  /* compute_reverse_probabilities!: :345-355 */
                    for (int q = 0; q < ns; ++q) zp += st_p[q] - ds.v[st_j[q]];
                    if (zp < 2.2250738585072014e-308) zp = 2.2250738585072014e-308;
                    if (zp > (double)N) zp = (double)N;
                    const double c = z / zp;
                    rrr_draw(seed, g, replica, 1, w);
                    if (u53_of(w[0], w[1]) < c) {
                        spf_spinflip(&X, chunks, move);
                        for (int q = 0; q < ns; ++q) { dEs[st_j[q]] = st_dE[q]; dyns_set(&ds, st_j[q], st_p[q]); }       /* apply_staged! */
                        E += dE; accepted += 1; acc = 1;
                    }
                } else {
                    const double c = cont_apply_move(&X, chunks, &ds, dEs, beta, move);
                    rrr_draw(seed, g, replica, 1, w);
                    if (u53_of(w[0], w[1]) < c) { E += dE; accepted += 1; acc = 1; }
                    else cont_apply_move(&X, chunks, &ds, dEs, beta, move);
                }
                acc_rate = acc_rate * (1 - lambda) + (double)acc * lambda;
            }
            itdone = iters;
        } else {
            int64_t it = 0, nextstep = step, m = 0;
            while (it < iters) {
                m += 1;
                const uint64_t g = it0 + (uint64_t)m;
                uint32_t w[4];
                rrr_draw(seed, g, replica, 2, w);
                double b = ds.z / (double)N;                                           /* rand_skip: DeltaE.jl:319-325 */
                if (b < 2.2250738585072014e-308) b = 2.2250738585072014e-308;
                if (b > 1.0) b = 1.0;
                double skipf = __builtin_floor(orc_det_log1p(-u53_of(w[0], w[1])) / orc_det_log1p(-b));
                int64_t skip = skipf >= 9.0e18 ? (int64_t)9.0e18 : (int64_t)skipf;
                rrr_draw(seed, g, replica, 0, w);
                int64_t move = dyns_getel(&ds, u53_of(w[0], w[1]));
                if (move < 0) { bad = 1; break; }
                const double dE = dEs[move];
                int out = 0;
                while (it + skip + 1 >= nextstep) {
                    Es[nsamp++] = E; if (!ORC_HOOK(nextstep, accepted, E, 0)) { out = 1; break; }
                    nextstep += step;
                    if (nextstep > iters) { out = 1; break; }
                }
                if (out) break;
                cont_apply_move(&X, chunks, &ds, dEs, beta, move);                    /* apply_step_bkl!: RRRMC.jl:294-295 */
                it += skip + 1;
                E += dE;
                accepted += 1;
            }
            second = accepted; itdone = it;
        }
        free(dEs);
        dyns_free(&ds);
    }
    if (stats) { stats[0] = accepted; stats[1] = second; stats[2] = itdone; }
    if (t_out) *t_out = t;
    free(X.lfields); free(X.lfields_last);
    if (dJ) { free(X0.lfields); free(X0.lfields_last); }
    return bad ? -1 : nsamp;
}
ORC_API int64_t orc_cont_sparse(int mode, int form, int64_t N, int64_t K, const int32_t *A, const double *J, double beta,
                                int64_t iters, int64_t step, double stepf, double staged_thr, double staged_thr_fact,
                                uint64_t seed, uint64_t it0, uint32_t call, uint32_t replica,
                                uint64_t *chunks, double *Es, int64_t *stats, double *t_out)
{
    return cont_sparse_impl(mode, form, N, K, A, J, NULL, 1, 1.0, beta, iters, step, stepf, staged_thr, staged_thr_fact, seed, it0, call,
                            replica, chunks, Es, stats, t_out, NULL);
}
/* bklMC / wtmMC on the discretised DoubleGraphs: for a graph that is not a DiscrGraph the reference builds the continuous-energy
 * caches over the WHOLE graph (gen_ΔEcache(X::AbstractGraph, ...) -> DeltaECacheCont, DeltaE.jl:315; THeap), with
 * delta_energy(X, C, i) = convert(Float64, dE0 + dE1) (RRG.jl:493-497) and neighbors(X, i) = A[i] (RRG.jl:499) / uA[i] (EA.jl:529). */
ORC_API int64_t orc_cont_double(int mode, int form, int64_t N, int64_t K, const int32_t *A, const int32_t *dJ, const double *rJ,
                                int64_t mul, double div, double beta, int64_t iters, int64_t step, double stepf,
                                uint64_t seed, uint64_t it0, uint32_t call, uint32_t replica,
                                uint64_t *chunks, double *Es, int64_t *stats, double *t_out)
{
    return cont_sparse_impl(mode, form, N, K, A, rJ, dJ, mul, div, beta, iters, step, stepf, 0.8, 5.0, seed, it0, call, replica, chunks, Es,
                            stats, t_out, NULL);
}

/* =============================================================================================
 * extremal_opt (src/RRRMC.jl:474-521) with EOCacheCont (src/DeltaE.jl:557-635) on the Float64 sparse graphs (dJ = NULL) and on the
 * discretised DoubleGraphs (dJ = level part): the graphs that are not DiscrGraphs.
 *   cache     dEs[i] = delta_energy(X, C, i); rank = sortperm(dEs) (stable: ties in site order), DeltaE.jl:563-573
 *   rand_move r = (1 - rand()) z, i = searchsortedfirst(ftau, r), move = rank[i], dE = dEs[move] (:577-590); RRR stream sub 3
 *   apply     spinflip!, dEs of the moved spin and of neighbors(X, move) refreshed, then the WHOLE vector re-sorted and every run of
 *             equal values shuffled uniformly (sortperm! + rankshuffle!, :592-634).
 * Neither the sort's tie order nor Julia's shuffle! is pinned, so the uniform shuffle is restated as a random key:
 * in iteration g every site gets the 64-bit key Philox(g, replica, TAG_RRR | 4 << 8 | site << 16) and the ranking after the move is
 * the ascending order of (dE, key, site) — a fresh uniform permutation of every tie run per move, as rankshuffle! gives.  (With
 * Gaussian couplings ties have probability zero and the keys are never looked at.)
 * Returns the number of samples (E before the move of iterations k * step), -1 if the tracked energy drifted from energy(X, C)
 * by more than 1e-9 relative or the ranking is not sorted.
 * ============================================================================================= */
typedef struct { const double *dEs; uint64_t seed, g; uint32_t replica; int fresh; } eocmp_t;
static inline uint64_t eo_tie_key(const eocmp_t *c, int32_t site)
{
    uint32_t w[4];
    orc_draw(c->seed, (uint32_t)c->g, (uint32_t)(c->g >> 32), c->replica, (uint32_t)ORC_TAG_RRR | (4u << 8) | ((uint32_t)site << 16), w);
    return ((uint64_t)w[0] << 32) | w[1];
}
static int eo_less(const eocmp_t *c, int32_t a, int32_t b)
{
    if (c->dEs[a] < c->dEs[b]) return 1;
    if (c->dEs[a] > c->dEs[b]) return 0;
    if (c->fresh) {
        const uint64_t ka = eo_tie_key(c, a), kb = eo_tie_key(c, b);
        if (ka != kb) return ka < kb;
    }
    return a < b;
}
static void eo_sort(const eocmp_t *c, int32_t *rank, int32_t *tmp, int64_t n)           /* top-down merge sort of rank[0..n) */
{
    if (n < 2) return;
    const int64_t h = n / 2;
    eo_sort(c, rank, tmp, h);
    eo_sort(c, rank + h, tmp, n - h);
    int64_t i = 0, j = h, k = 0;
    while (i < h && j < n) tmp[k++] = eo_less(c, rank[j], rank[i]) ? rank[j++] : rank[i++];
    while (i < h) tmp[k++] = rank[i++];
    while (j < n) tmp[k++] = rank[j++];
    memcpy(rank, tmp, (size_t)n * sizeof(int32_t));
}
ORC_API int64_t orc_extremal_opt_cont(int form, int64_t N, int64_t K, const int32_t *A, const double *J, const int32_t *dJ, int64_t mul, double div,
                                      const double *ftau, int64_t iters, int64_t step, uint64_t seed, uint64_t it0, uint32_t replica,
                                      uint64_t *chunks, double *Es, double *Emin_out, uint64_t *Cmin, int64_t *itmin_out)
{
    if (N > 65535) return -2;                                             /* the tie key carries the site in 16 bits */
    spf_t X = {N, K, A, J, NULL, NULL, -1, form, NULL, 1, 1.0, NULL, NULL};
    X.lfields = (double *)malloc((size_t)N * 8);
    X.lfields_last = (double *)malloc((size_t)N * 8);
    sparse_t X0 = {N, K, A, dJ, NULL, NULL, -1, form};
    double E = 0.0;
    if (dJ) {
        X0.lfields = (int64_t *)malloc((size_t)N * 8);
        X0.lfields_last = (int64_t *)malloc((size_t)N * 8);
        X.X0 = &X0; X.lev_mul = mul; X.lev_div = div;
        E = lev_to_f64(sparse_energy(&X0, chunks), mul, div);
        E = E + spf_energy(&X, chunks);
    } else {
        E = spf_energy(&X, chunks);
    }
    const int64_t nch = (N + 63) / 64;
    double Emin = E;
    int64_t itmin = 0, nsamp = 0;
    memcpy(Cmin, chunks, (size_t)nch * 8);
    double *dEs = (double *)malloc((size_t)N * 8);
    int32_t *rank = (int32_t *)malloc((size_t)N * 4), *tmp = (int32_t *)malloc((size_t)N * 4);
    for (int64_t i = 0; i < N; ++i) { dEs[i] = spf_dE(&X, i); rank[i] = (int32_t)i; }
    eocmp_t cmp = {dEs, seed, 0, replica, 0};
    eo_sort(&cmp, rank, tmp, N);                                          /* sortperm(dEs): no shuffle at construction */
    const double z = ftau[N - 1];
    for (int64_t it = 1; it <= iters; ++it) {
        if (it % step == 0) { Es[nsamp++] = E; if (!ORC_HOOK(it, 0, E, Emin)) break; }
        const uint64_t g = it0 + (uint64_t)it;
        uint32_t w[4];
        rrr_draw(seed, g, replica, 3, w);
        const double r = (1 - u53_of(w[0], w[1])) * z;
        int64_t lo = 0, hi = N;
        while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (ftau[mid] < r) lo = mid + 1; else hi = mid; }
        int64_t i = lo + 1;
        if (i > N) i = N;
        const int64_t move = rank[i - 1];
        const double dE = dEs[move];
        spf_spinflip(&X, chunks, move);
        dEs[move] = spf_dE(&X, move);
        int64_t nb[SK_MAX];
        int nn = spf_neighbors(&X, move, nb);
        for (int q = 0; q < nn; ++q) dEs[nb[q]] = spf_dE(&X, nb[q]);
        cmp.g = g; cmp.fresh = 1;
        eo_sort(&cmp, rank, tmp, N);
        E += dE;
        if (E < Emin) { Emin = E; memcpy(Cmin, chunks, (size_t)nch * 8); itmin = it; }
    }
    int ok = 1;
    for (int64_t i = 1; i < N && ok; ++i) ok = dEs[rank[i - 1]] <= dEs[rank[i]];
    for (int64_t i = 0; i < N && ok; ++i) ok = dEs[i] == spf_dE(&X, i);
    {
        double Ex = spf_energy(&X, chunks);
        if (dJ) Ex = lev_to_f64(sparse_energy(&X0, chunks), mul, div) + Ex;
        const double tol = 1e-9 * (fabs(Ex) > 1.0 ? fabs(Ex) : 1.0);
        if (fabs(Ex - E) > tol) ok = 0;
    }
    if (Emin_out) *Emin_out = Emin;
    if (itmin_out) *itmin_out = itmin;
    free(dEs); free(rank); free(tmp); free(X.lfields); free(X.lfields_last);
    if (dJ) { free(X0.lfields); free(X0.lfields_last); }
    return ok ? nsamp : -1;
}

/* extremal_opt (src/RRRMC.jl:474-521) on a GraphQuant over GraphRRG / GraphEA slices: a DoubleGraph is not a DiscrGraph, so gen_EOcache
 * builds the generic EOCacheCont (DeltaE.jl:557-635) over all N = Nk M spins with delta_energy = delta_energy(X0) + residual
 * (QT.jl:283-286) and neighbors(X, i) = the Trotter pair, then the slice graph's (QT.jl:288-321).  Streams and tie rule as
 * orc_extremal_opt_cont. */
static int64_t extremal_opt_quant_impl(quant_t *Qp, int form, int64_t K, const int32_t *A,
                                       const double *ftau, int64_t iters, int64_t step, uint64_t seed, uint64_t it0, uint32_t replica,
                                       uint64_t *chunks, double *Es, double *Emin_out, uint64_t *Cmin, int64_t *itmin_out);
ORC_API int64_t orc_extremal_opt_quant(int form, int64_t Nk, int64_t M, int64_t K, const int32_t *A, const int32_t *J, double fourK,
                                       const double *ftau, int64_t iters, int64_t step, uint64_t seed, uint64_t it0, uint32_t replica,
                                       uint64_t *chunks, double *Es, double *Emin_out, uint64_t *Cmin, int64_t *itmin_out)
{
    if (Nk * M > 65535 || K + 2 > SK_MAX + 2) return -2;
    quant_t Q;
    quant_init(&Q, Nk, M, K, A, J, fourK);
    for (int64_t k = 0; k < M; ++k) Q.X1[k].ea_form = form;
    const int64_t r = extremal_opt_quant_impl(&Q, form, K, A, ftau, iters, step, seed, it0, replica, chunks, Es, Emin_out, Cmin, itmin_out);
    quant_free(&Q);
    return r;
}
ORC_API int64_t orc_extremal_opt_quant_spf(int form, int64_t Nk, int64_t M, int64_t K, const int32_t *A, const double *Jf, double fourK,
                                           const double *ftau, int64_t iters, int64_t step, uint64_t seed, uint64_t it0, uint32_t replica,
                                           uint64_t *chunks, double *Es, double *Emin_out, uint64_t *Cmin, int64_t *itmin_out)
{
    if (Nk * M > 65535) return -2;
    quant_t Q;
    quant_init_spf(&Q, Nk, M, K, A, Jf, form, fourK);
    const int64_t r = extremal_opt_quant_impl(&Q, form, K, A, ftau, iters, step, seed, it0, replica, chunks, Es, Emin_out, Cmin, itmin_out);
    quant_free(&Q);
    return r;
}
ORC_API int64_t orc_extremal_opt_quant_dense(int kind, int64_t Nk, int64_t M, const uint64_t *Jb, const double *Jd, double fourK,
                                             const double *ftau, int64_t iters, int64_t step, uint64_t seed, uint64_t it0, uint32_t replica,
                                             uint64_t *chunks, double *Es, double *Emin_out, uint64_t *Cmin, int64_t *itmin_out)
{
    if (Nk * M > 65535 || Nk + 1 > ORC_NB_MAX) return -2;
    quant_t Q;
    if (kind == 3) quant_init_skn(&Q, Nk, M, Jd, fourK); else quant_init_sk(&Q, Nk, M, Jb, fourK);
    const int64_t r = extremal_opt_quant_impl(&Q, 0, 0, NULL, ftau, iters, step, seed, it0, replica, chunks, Es, Emin_out, Cmin, itmin_out);
    quant_free(&Q);
    return r;
}
static int64_t extremal_opt_quant_impl(quant_t *Qp, int form, int64_t K, const int32_t *A,
                                       const double *ftau, int64_t iters, int64_t step, uint64_t seed, uint64_t it0, uint32_t replica,
                                       uint64_t *chunks, double *Es, double *Emin_out, uint64_t *Cmin, int64_t *itmin_out)
{
    quant_t Q = *Qp;                     /* the caller owns (and frees) the slice arrays */
    const int64_t Nk = Q.Nk, M = Q.M;
    const int64_t N = Nk * M;
    spf_t X = {N, K, A, NULL, NULL, NULL, -1, form, NULL, 1, 1.0, NULL, NULL};
    X.Q = &Q; X.cur_s = chunks;
    double E = quant_energy(&Q, chunks);
    const int64_t nch = (N + 63) / 64;
    double Emin = E;
    int64_t itmin = 0, nsamp = 0;
    memcpy(Cmin, chunks, (size_t)nch * 8);
    double *dEs = (double *)malloc((size_t)N * 8);
    int32_t *rank = (int32_t *)malloc((size_t)N * 4), *tmp = (int32_t *)malloc((size_t)N * 4);
    for (int64_t i = 0; i < N; ++i) { dEs[i] = spf_dE(&X, i); rank[i] = (int32_t)i; }
    eocmp_t cmp = {dEs, seed, 0, replica, 0};
    eo_sort(&cmp, rank, tmp, N);
    const double z = ftau[N - 1];
    for (int64_t it = 1; it <= iters; ++it) {
        if (it % step == 0) { Es[nsamp++] = E; if (!ORC_HOOK(it, 0, E, Emin)) break; }
        const uint64_t g = it0 + (uint64_t)it;
        uint32_t w[4];
        rrr_draw(seed, g, replica, 3, w);
        const double r = (1 - u53_of(w[0], w[1])) * z;
        int64_t lo = 0, hi = N;
        while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (ftau[mid] < r) lo = mid + 1; else hi = mid; }
        int64_t i = lo + 1;
        if (i > N) i = N;
        const int64_t move = rank[i - 1];
        const double dE = dEs[move];
        spf_spinflip(&X, chunks, move);
        dEs[move] = spf_dE(&X, move);
        int64_t nb[ORC_NB_MAX];
        int nn = spf_neighbors(&X, move, nb);
        for (int q = 0; q < nn; ++q) dEs[nb[q]] = spf_dE(&X, nb[q]);
        cmp.g = g; cmp.fresh = 1;
        eo_sort(&cmp, rank, tmp, N);
        E += dE;
        if (E < Emin) { Emin = E; memcpy(Cmin, chunks, (size_t)nch * 8); itmin = it; }
    }
    int ok = 1;
    for (int64_t i = 1; i < N && ok; ++i) ok = dEs[rank[i - 1]] <= dEs[rank[i]];
    for (int64_t i = 0; i < N && ok; ++i) ok = dEs[i] == spf_dE(&X, i);
    {
        uint64_t *cp = (uint64_t *)malloc((size_t)nch * 8);
        memcpy(cp, chunks, (size_t)nch * 8);
        const double Ex = quant_energy(&Q, cp);
        const double tol = 1e-9 * (fabs(Ex) > 1.0 ? fabs(Ex) : 1.0);
        if (fabs(Ex - E) > tol) ok = 0;
        free(cp);
    }
    if (Emin_out) *Emin_out = Emin;
    if (itmin_out) *itmin_out = itmin;
    free(dEs); free(rank); free(tmp);
    return ok ? nsamp : -1;
}

/* extremal_opt with EOCacheCont (see orc_extremal_opt_cont) on the dense SK models: neighbors(X, i) = AllButOne (SK.jl:142,297), so
 * every delta_energy is refreshed and the whole ranking re-sorted at every flip.  Same streams and tie rule as orc_extremal_opt_cont. */
static int64_t extremal_opt_sk_impl(skx_t *X, int64_t N, const double *ftau, int64_t iters, int64_t step, uint64_t seed, uint64_t it0,
                                    uint32_t replica, uint64_t *chunks, double *Es, double *Emin_out, uint64_t *Cmin, int64_t *itmin_out)
{
    if (N > 65535) return -2;
    const int64_t nch = (N + 63) / 64;
    double E = skx_energy(X, chunks), Emin = E;
    int64_t itmin = 0, nsamp = 0;
    memcpy(Cmin, chunks, (size_t)nch * 8);
    double *dEs = (double *)malloc((size_t)N * 8);
    int32_t *rank = (int32_t *)malloc((size_t)N * 4), *tmp = (int32_t *)malloc((size_t)N * 4);
    for (int64_t i = 0; i < N; ++i) { dEs[i] = skx_dE(X, i); rank[i] = (int32_t)i; }
    eocmp_t cmp = {dEs, seed, 0, replica, 0};
    eo_sort(&cmp, rank, tmp, N);
    const double z = ftau[N - 1];
    for (int64_t it = 1; it <= iters; ++it) {
        if (it % step == 0) { Es[nsamp++] = E; if (!ORC_HOOK(it, 0, E, Emin)) break; }
        const uint64_t g = it0 + (uint64_t)it;
        uint32_t w[4];
        rrr_draw(seed, g, replica, 3, w);
        const double r = (1 - u53_of(w[0], w[1])) * z;
        int64_t lo = 0, hi = N;
        while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (ftau[mid] < r) lo = mid + 1; else hi = mid; }
        int64_t i = lo + 1;
        if (i > N) i = N;
        const int64_t move = rank[i - 1];
        const double dE = dEs[move];
        bitflip(chunks, move);
        skx_update(X, chunks, move);
        for (int64_t j = 0; j < N; ++j) dEs[j] = skx_dE(X, j);
        cmp.g = g; cmp.fresh = 1;
        eo_sort(&cmp, rank, tmp, N);
        E += dE;
        if (E < Emin) { Emin = E; memcpy(Cmin, chunks, (size_t)nch * 8); itmin = it; }
    }
    int ok = 1;
    for (int64_t i = 1; i < N && ok; ++i) ok = dEs[rank[i - 1]] <= dEs[rank[i]];
    const double Ex = skx_energy(X, chunks);
    if (fabs(Ex - E) > 1e-9 * (fabs(Ex) > 1.0 ? fabs(Ex) : 1.0)) ok = 0;
    if (Emin_out) *Emin_out = Emin;
    if (itmin_out) *itmin_out = itmin;
    free(dEs); free(rank); free(tmp);
    return ok ? nsamp : -1;
}
ORC_API int64_t orc_extremal_opt_skn(int64_t N, const double *J, const double *ftau, int64_t iters, int64_t step, uint64_t seed, uint64_t it0,
                                     uint32_t replica, uint64_t *chunks, double *Es, double *Emin_out, uint64_t *Cmin, int64_t *itmin_out)
{
    skn_t Xn = {N, J, NULL, NULL, -1};
    Xn.lfields = (double *)malloc((size_t)N * 8);
    Xn.lfields_last = (double *)malloc((size_t)N * 8);
    skx_t X = {&Xn, NULL};
    int64_t r = extremal_opt_sk_impl(&X, N, ftau, iters, step, seed, it0, replica, chunks, Es, Emin_out, Cmin, itmin_out);
    free(Xn.lfields); free(Xn.lfields_last);
    return r;
}
ORC_API int64_t orc_extremal_opt_skb(int64_t N, const uint64_t *Jb, const double *ftau, int64_t iters, int64_t step, uint64_t seed, uint64_t it0,
                                     uint32_t replica, uint64_t *chunks, double *Es, double *Emin_out, uint64_t *Cmin, int64_t *itmin_out)
{
    skb_t Xb = {N, (N + 63) / 64, sqrt((double)N), Jb, NULL, NULL, -1};
    Xb.lfields = (int64_t *)malloc((size_t)N * 8);
    Xb.lfields_last = (int64_t *)malloc((size_t)N * 8);
    skx_t X = {NULL, &Xb};
    int64_t r = extremal_opt_sk_impl(&X, N, ftau, iters, step, seed, it0, replica, chunks, Es, Emin_out, Cmin, itmin_out);
    free(Xb.lfields); free(Xb.lfields_last);
    return r;
}

/* bklMC / wtmMC on a GraphQuant over GraphRRG / GraphEA slices: a DoubleGraph is not a DiscrGraph, so the reference builds the
 * continuous-energy caches over the WHOLE graph (DeltaE.jl:315, WaitingTimes.jl) with delta_energy = delta_energy(X0) + residual
 * (QT.jl:283-286) and neighbors(X, i) = the two Trotter neighbours, then the slice graph's (QT.jl:288-321).  mode 1 / 2 as orc_cont_sparse. */
/* the same over dense slices: kind 2 = binary GraphSK (Jb: bit-packed rows), 3 = GraphSKNormal (Jd: Nk x Nk Float64) */
ORC_API int64_t orc_cont_quant_dense(int mode, int kind, int64_t Nk, int64_t M, const uint64_t *Jb, const double *Jd, double fourK, double beta,
                                     int64_t iters, int64_t step, double stepf, uint64_t seed, uint64_t it0, uint32_t call, uint32_t replica,
                                     uint64_t *chunks, double *Es, int64_t *stats, double *t_out)
{
    if (Nk + 1 > ORC_NB_MAX) return -2;
    quant_t Q;
    if (kind == 3) quant_init_skn(&Q, Nk, M, Jd, fourK); else quant_init_sk(&Q, Nk, M, Jb, fourK);
    int64_t r = cont_sparse_impl(mode, 0, Nk * M, 0, NULL, NULL, NULL, 1, 1.0, beta, iters, step, stepf, 0.8, 5.0, seed, it0, call, replica,
                                 chunks, Es, stats, t_out, &Q);
    quant_free(&Q);
    return r;
}
ORC_API int64_t orc_cont_quant_spf(int mode, int form, int64_t Nk, int64_t M, int64_t K, const int32_t *A, const double *Jf, double fourK, double beta,
                                   int64_t iters, int64_t step, double stepf, uint64_t seed, uint64_t it0, uint32_t call, uint32_t replica,
                                   uint64_t *chunks, double *Es, int64_t *stats, double *t_out)
{
    if (K + 2 > SK_MAX + 2) return -2;
    quant_t Q;
    quant_init_spf(&Q, Nk, M, K, A, Jf, form, fourK);
    int64_t r = cont_sparse_impl(mode, 0, Nk * M, K, NULL, NULL, NULL, 1, 1.0, beta, iters, step, stepf, 0.8, 5.0, seed, it0, call, replica,
                                 chunks, Es, stats, t_out, &Q);
    quant_free(&Q);
    return r;
}
ORC_API int64_t orc_cont_quant(int mode, int form, int64_t Nk, int64_t M, int64_t K, const int32_t *A, const int32_t *J, double fourK, double beta,
                               int64_t iters, int64_t step, double stepf, uint64_t seed, uint64_t it0, uint32_t call, uint32_t replica,
                               uint64_t *chunks, double *Es, int64_t *stats, double *t_out)
{
    if (K + 2 > SK_MAX + 2) return -2;
    quant_t Q;
    quant_init(&Q, Nk, M, K, A, J, fourK);
    for (int64_t k = 0; k < M; ++k) Q.X1[k].ea_form = form;          /* neighbors(X1[k], i): uA of a GraphEA slice (EA.jl:158,292) */
    int64_t r = cont_sparse_impl(mode, 0, Nk * M, K, NULL, NULL, NULL, 1, 1.0, beta, iters, step, stepf, 0.8, 5.0, seed, it0, call, replica,
                                 chunks, Es, stats, t_out, &Q);
    quant_free(&Q);
    return r;
}

/* wtmMC (src/RRRMC.jl:376-426, src/WaitingTimes.jl) on GraphSKNormal: as orc_wtm_mc_sparse with delta_energy = +lfields[i]
 * (SK.jl:278-284) and neighbors(X, i) = AllButOne (every other spin, in index order: SK.jl:297).  WTM stream: spin i's initial
 * time is draw i, then one draw per updated spin in the order of update_heap! (the moved spin, then j = 0..N-1 except it). */
static int64_t orc_wtm_mc_skn_impl(skx_t *X, int64_t N, double beta, int64_t samples, double step, uint64_t seed, uint32_t call,
                               uint32_t replica, uint64_t *chunks, double *Es, int64_t *stats, double *t_out)
{
    double E = skx_energy(X, chunks);
    double *tm = (double *)malloc((size_t)N * sizeof(double));
    uint64_t nd = 0;
    for (int64_t i = 0; i < N; ++i) {
        double e = orc_det_exp(beta * skx_dE(X, i));
        tm[i] = wtm_gen(e > 1.0 ? e : 1.0, wtm_uniform(seed, nd++, replica, call));
    }
    step /= (double)N;
    const double tmax = step * (double)samples;
    double t = 0.0, nextstep = step;
    int64_t num_moves = 0, nsamp = 0;
    int out = 0;
    while (t < tmax && !out) {
        int64_t move = 0;
        for (int64_t i = 1; i < N; ++i) if (tm[i] < tm[move]) move = i;          /* pick_next: top_with_handle */
        const double tp = tm[move];
        while (tp >= nextstep) {
            Es[nsamp++] = E; if (!ORC_HOOK(nextstep, num_moves, E, 0)) { out = 1; break; }
            nextstep += step;
            if (nextstep > tmax + 1e-10) { out = 1; break; }
        }
        if (out) break;
        t = tp;
        const double dE = skx_dE(X, move);
        bitflip(chunks, move);
        skx_update(X, chunks, move);
        for (int64_t q = -1; q < N; ++q) {                                       /* the moved spin first, then every other spin */
            const int64_t j = q < 0 ? move : q;
            if (q == move) continue;
            double e = orc_det_exp(beta * skx_dE(X, j));
            tm[j] = t + wtm_gen(e > 1.0 ? e : 1.0, wtm_uniform(seed, nd++, replica, call));
        }
        E += dE;
        num_moves += 1;
    }
    if (stats) { stats[0] = num_moves; stats[1] = nsamp; }
    if (t_out) *t_out = t;
    const double Echeck = skx_energy(X, chunks);
    int ok = fabs(Echeck - E) < 1e-9 * (1.0 + fabs(E));
    free(tm);
    return ok ? nsamp : -1;
}
ORC_API int64_t orc_wtm_mc_skn(int64_t N, const double *J, double beta, int64_t samples, double step, uint64_t seed, uint32_t call,
                               uint32_t replica, uint64_t *chunks, double *Es, int64_t *stats, double *t_out)
{
    skn_t Xn = {N, J, NULL, NULL, -1};
    Xn.lfields = (double *)malloc((size_t)N * 8);
    Xn.lfields_last = (double *)malloc((size_t)N * 8);
    skx_t X = {&Xn, NULL};
    int64_t r = orc_wtm_mc_skn_impl(&X, N, beta, samples, step, seed, call, replica, chunks, Es, stats, t_out);
    free(Xn.lfields); free(Xn.lfields_last);
    return r;
}
/* the same sampler on the binary GraphSK (J = bit rows, SK.jl:32) */
ORC_API int64_t orc_wtm_mc_skb(int64_t N, const uint64_t *Jb, double beta, int64_t samples, double step, uint64_t seed, uint32_t call,
                               uint32_t replica, uint64_t *chunks, double *Es, int64_t *stats, double *t_out)
{
    skb_t Xb = {N, (N + 63) / 64, sqrt((double)N), Jb, NULL, NULL, -1};
    Xb.lfields = (int64_t *)malloc((size_t)N * 8);
    Xb.lfields_last = (int64_t *)malloc((size_t)N * 8);
    skx_t X = {NULL, &Xb};
    int64_t r = orc_wtm_mc_skn_impl(&X, N, beta, samples, step, seed, call, replica, chunks, Es, stats, t_out);
    free(Xb.lfields); free(Xb.lfields_last);
    return r;
}

/* bklMC (src/RRRMC.jl:311-359) on GraphSKNormal with DeltaECacheCont: rand_skip (DeltaE.jl:319-325),
 * rand_move, apply_step_bkl! = apply_move!(X, C, move, cache, Val{false}) (RRRMC.jl:294-295) over all N - 1 neighbours.
 * RRR stream sub 2 (skip), sub 0 (getel); g counts moves.  stats = [moves, moves, iterations done]. */
static int64_t orc_bkl_mc_skn_impl(skx_t *X, int64_t N, double beta, int64_t iters, int64_t step,
                               uint64_t seed, uint64_t it0, uint32_t replica, uint64_t *chunks, double *Es, int64_t *stats)
{
    double E = skx_energy(X, chunks);
    double *dEs = (double *)malloc((size_t)N * 8);
    dyns_t ds;
    dyns_init(&ds, N);
    for (int64_t i = 0; i < N; ++i) { dEs[i] = skx_dE(X, i); ds.v[i] = prior_of(beta * dEs[i]); }
    dyns_refresh(&ds);
    int64_t it = 0, nextstep = step, m = 0, accepted = 0, nsamp = 0, bad = 0;
    while (it < iters) {
        m += 1;
        const uint64_t g = it0 + (uint64_t)m;
        uint32_t w[4];
        rrr_draw(seed, g, replica, 2, w);
        double b = ds.z / (double)N;
        if (b < 2.2250738585072014e-308) b = 2.2250738585072014e-308;
        if (b > 1.0) b = 1.0;
        double skipf = __builtin_floor(orc_det_log1p(-u53_of(w[0], w[1])) / orc_det_log1p(-b));
        int64_t skip = skipf >= 9.0e18 ? (int64_t)9.0e18 : (int64_t)skipf;
        rrr_draw(seed, g, replica, 0, w);
        int64_t move = dyns_getel(&ds, u53_of(w[0], w[1]));
        if (move < 0) { bad = 1; break; }
        const double dE = dEs[move];
        int out = 0;
        while (it + skip + 1 >= nextstep) {
            Es[nsamp++] = E; if (!ORC_HOOK(nextstep, accepted, E, 0)) { out = 1; break; }
            nextstep += step;
            if (nextstep > iters) { out = 1; break; }
        }
        if (out) break;
        bitflip(chunks, move); skx_update(X, chunks, move);           /* apply_move!: DeltaE.jl:376-410 */
        dEs[move] = skx_dE(X, move);
        dyns_set(&ds, move, prior_of(beta * dEs[move]));
        for (int64_t j = 0; j < N; ++j) {
            if (j == move) continue;
            dEs[j] = skx_dE(X, j);
            dyns_set(&ds, j, prior_of(beta * dEs[j]));
        }
        it += skip + 1;
        E += dE;
        accepted += 1;
    }
    if (stats) { stats[0] = accepted; stats[1] = accepted; stats[2] = it; }
    free(dEs); dyns_free(&ds);
    return bad ? -1 : nsamp;
}
ORC_API int64_t orc_bkl_mc_skn(int64_t N, const double *J, double beta, int64_t iters, int64_t step,
                               uint64_t seed, uint64_t it0, uint32_t replica, uint64_t *chunks, double *Es, int64_t *stats)
{
    skn_t Xn = {N, J, NULL, NULL, -1};
    Xn.lfields = (double *)malloc((size_t)N * 8);
    Xn.lfields_last = (double *)malloc((size_t)N * 8);
    skx_t X = {&Xn, NULL};
    int64_t r = orc_bkl_mc_skn_impl(&X, N, beta, iters, step, seed, it0, replica, chunks, Es, stats);
    free(Xn.lfields); free(Xn.lfields_last);
    return r;
}
/* the same sampler on the binary GraphSK (J = bit rows, SK.jl:32) */
ORC_API int64_t orc_bkl_mc_skb(int64_t N, const uint64_t *Jb, double beta, int64_t iters, int64_t step,
                               uint64_t seed, uint64_t it0, uint32_t replica, uint64_t *chunks, double *Es, int64_t *stats)
{
    skb_t Xb = {N, (N + 63) / 64, sqrt((double)N), Jb, NULL, NULL, -1};
    Xb.lfields = (int64_t *)malloc((size_t)N * 8);
    Xb.lfields_last = (int64_t *)malloc((size_t)N * 8);
    skx_t X = {NULL, &Xb};
    int64_t r = orc_bkl_mc_skn_impl(&X, N, beta, iters, step, seed, it0, replica, chunks, Es, stats);
    free(Xb.lfields); free(Xb.lfields_last);
    return r;
}
