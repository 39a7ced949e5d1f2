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
