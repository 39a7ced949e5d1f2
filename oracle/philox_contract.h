/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it.
 *
 * Philox4x32-10 and the random-stream contract of this build, restated independently of the
 * library's device/host implementation (rrrmc.jl_amd/csrc/philox.hpp) so that the two can be
 * checked against each other.
 *
 * PARITY NOTE ("parity unpinned" w.r.t. Julia's RNG): the reference draws every random number from
 * Julia's global RNG (src/RRRMC.jl:39,89,113; src/Interface.jl:26; src/graphs/RRG.jl:45,155), whose
 * algorithm is not pinned by the reference (Project.toml:21 only says julia >= 1.3; CI runs 1.3, 1.7
 * and nightly, .github/workflows/ci.yml:15-18) and Julia is not installable here.  This build
 * therefore replaces each of those draws by an addressed Philox draw, defined below; parity is
 * oracle <-> HIP on these streams, and oracle <-> reference on RNG-free quantities (energies, caches,
 * class bookkeeping, closed-form toy models).
 *
 * Stream contract (key = (seed & 0xffffffff, seed >> 32); counter = (c0,c1,c2,c3)):
 *   SITE    : the site attempted at global iteration g (1-based, 64 bit), shared by all replicas:
 *             ctr = (lo32(g>>1), hi32(g>>1), 0, TAG_SITE); u64 = (w[2h] << 32) | w[2h+1], h = g & 1;
 *             site = floor(u64 * N / 2^64)            [replaces `rand(1:N)`, src/RRRMC.jl:113]
 *   ACCEPT  : the uniform u in [0,1) of replica r at iteration g is the 64-bit binary fraction whose
 *             bit j (j = 0 is the MSB) is bit (r & 31) of word (j & 3) of
 *             ctr = (lo32(g), hi32(g), r >> 5, TAG_ACCEPT | ((j >> 2) << 8)).
 *             A move with x = -beta*dE < 0 is accepted iff u * 2^-64 < exp(x)
 *                                                     [replaces `rand() < exp(x)`, src/RRRMC.jl:39]
 *             (32 replicas share one Philox word column-wise: replica = bit position.  Every output bit
 *              of Philox is used at most once, so the uniforms are independent; the bits are consumed
 *              lazily, MSB first, which a counter-based generator allows.)
 *   INIT    : initial spin bit of replica r at site x (0-based) = bit (r & 31) of word (x & 3) of
 *             ctr = (x >> 2, 0, r >> 5, TAG_INIT)     [replaces `rand!(BitVector)`, src/Interface.jl:26]
 *   GRAPH   : n-th 64-bit draw of graph construction = words (2h, 2h+1) of
 *             ctr = (lo32(n>>1), hi32(n>>1), 0, TAG_GRAPH), h = n & 1; `rand(1:m)` = 1 + floor(u64*m/2^64)
 *   COUPLING: same with TAG_COUPLING                  [replaces `rand(vLEV)`, src/graphs/RRG.jl:155]
 *   GAUSS   : n-th pair of normals via Box-Muller on two 53-bit uniforms, TAG_GAUSS (SK couplings)
 */
#ifndef RRRMC_ORACLE_PHILOX_CONTRACT_H
#define RRRMC_ORACLE_PHILOX_CONTRACT_H
#include <stdint.h>

enum {
    ORC_TAG_SITE = 1, ORC_TAG_ACCEPT = 2, ORC_TAG_INIT = 3, ORC_TAG_GRAPH = 4,
    ORC_TAG_COUPLING = 5, ORC_TAG_GAUSS = 6, ORC_TAG_SWEEP = 7, ORC_TAG_RRR = 8
};

static inline void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; ++round) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline void orc_draw(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4])
{
    uint32_t ctr[4] = {c0, c1, c2, c3};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    orc_philox4x32_10(ctr, key, out);
}

static inline uint64_t orc_mulhi64(uint64_t a, uint64_t b)
{
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
}

/* n-th 64-bit word of a sequential stream (SITE / GRAPH / COUPLING / ...). */
static inline uint64_t orc_stream_u64(uint64_t seed, uint32_t tag, uint64_t n)
{
    uint32_t w[4];
    uint64_t blk = n >> 1;
    orc_draw(seed, (uint32_t)blk, (uint32_t)(blk >> 32), 0u, tag, w);
    unsigned h = (unsigned)(n & 1u);
    return ((uint64_t)w[2 * h] << 32) | w[2 * h + 1];
}

static inline int64_t orc_site(uint64_t seed, uint64_t g, int64_t N)
{
    return (int64_t)orc_mulhi64(orc_stream_u64(seed, ORC_TAG_SITE, g), (uint64_t)N);
}

/* ceil(p * 2^64) for 0 < p < 1 as u64; *always = 1 when p >= 1 (accept regardless of u). */
static inline uint64_t orc_threshold64(double p, int *always)
{
    *always = 0;
    if (!(p > 0.0)) return 0;          /* never (also NaN) */
    if (p >= 1.0) { *always = 1; return UINT64_MAX; }
    union { double d; uint64_t u; } v; v.d = p;
    int bexp = (int)((v.u >> 52) & 0x7ff);
    uint64_t man = v.u & ((1ull << 52) - 1);
    int e;                              /* p = man * 2^e */
    if (bexp == 0) { e = -1074; } else { man |= (1ull << 52); e = bexp - 1075; }
    int sh = e + 64;                    /* p * 2^64 = man * 2^sh */
    if (sh >= 0) return man << sh;      /* p < 1 => man << sh < 2^64 */
    int s = -sh;
    if (s >= 64) return 1;
    return (man + ((1ull << s) - 1)) >> s;
}

/* Lazy evaluation of [u(g, replica) < T] with the ACCEPT stream. */
static inline int orc_accept_lt(uint64_t seed, uint64_t g, uint32_t replica, uint64_t T)
{
    uint32_t w[4];
    uint32_t grp = replica >> 5, bit = replica & 31u;
    for (int j = 0; j < 64; ++j) {
        if ((j & 3) == 0)
            orc_draw(seed, (uint32_t)g, (uint32_t)(g >> 32), grp,
                     (uint32_t)ORC_TAG_ACCEPT | ((uint32_t)(j >> 2) << 8), w);
        unsigned ub = (w[j & 3] >> bit) & 1u;
        unsigned tb = (unsigned)((T >> (63 - j)) & 1u);
        if (ub != tb) return ub < tb;
    }
    return 0; /* u == T */
}

/* Full 64-bit value of the ACCEPT uniform (used by tests and by Float64 models). */
static inline uint64_t orc_accept_u64(uint64_t seed, uint64_t g, uint32_t replica)
{
    uint32_t w[4];
    uint32_t grp = replica >> 5, bit = replica & 31u;
    uint64_t u = 0;
    for (int j = 0; j < 64; ++j) {
        if ((j & 3) == 0)
            orc_draw(seed, (uint32_t)g, (uint32_t)(g >> 32), grp,
                     (uint32_t)ORC_TAG_ACCEPT | ((uint32_t)(j >> 2) << 8), w);
        u = (u << 1) | ((w[j & 3] >> bit) & 1u);
    }
    return u;
}

static inline int orc_init_spin(uint64_t seed, uint32_t replica, uint64_t x)
{
    uint32_t w[4];
    orc_draw(seed, (uint32_t)(x >> 2), 0u, replica >> 5, ORC_TAG_INIT, w);
    return (int)((w[x & 3] >> (replica & 31u)) & 1u);
}

/* ------------------------------------------------------------------------------------------------
 * Float64-energy models (GraphSKNormal, DoubleGraph residuals): one 53-bit uniform per replica and iteration,
 *   ACCEPT_F64: ctr = (lo32(g>>1), hi32(g>>1), replica, TAG_ACCEPT_F64); u64 = (w[2h] << 32) | w[2h+1], h = g & 1;
 *               rand() = (u64 >> 11) * 2^-53                       [replaces `rand()`, src/RRRMC.jl:39]
 * and a deterministic exp shared (same algorithm, same operation order, no FMA) by the oracle and the device so that
 * `rand() < exp(x)` is decided identically on both sides (SURVEY.md §7 hard part 3).
 * ---------------------------------------------------------------------------------------------- */
enum { ORC_TAG_ACCEPT_F64 = 9 };

static inline double orc_rand53(uint64_t seed, uint64_t g, uint32_t replica)
{
    uint32_t w[4];
    uint64_t blk = g >> 1;
    orc_draw(seed, (uint32_t)blk, (uint32_t)(blk >> 32), replica, ORC_TAG_ACCEPT_F64, w);
    unsigned h = (unsigned)(g & 1u);
    uint64_t u = ((uint64_t)w[2 * h] << 32) | w[2 * h + 1];
    return (double)(u >> 11) * 0x1.0p-53;
}

/* exp(x): Cody-Waite reduction x = k ln2 + r, degree-13 Taylor polynomial in Horner form (separate multiply and
 * add roundings), exact scaling by 2^k.  Relative error ~2e-16; compile with -ffp-contract=off. */
static inline double orc_det_exp(double x)
{
    static const double LOG2E = 1.44269504088896338700e+00;
    static const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    static const double c[14] = {1.0, 1.0, 0.5, 1.0 / 6, 1.0 / 24, 1.0 / 120, 1.0 / 720, 1.0 / 5040, 1.0 / 40320, 1.0 / 362880,
                                 1.0 / 3628800, 1.0 / 39916800, 1.0 / 479001600, 1.0 / 6227020800.0};
    if (x != x) return x;
    if (x < -745.2) return 0.0;
    if (x > 709.7) return __builtin_inf();
    double k = __builtin_floor(x * LOG2E + 0.5);
    double r = (x - k * LN2_HI) - k * LN2_LO;
    double p = c[13];
    for (int n = 12; n >= 0; --n) p = p * r + c[n];
    return __builtin_ldexp(p, (int)k);
}

/* log1p(y) for -1 < y <= 0 with a fixed operation order (used by rand_skip of bklMC, src/DeltaE.jl:141-144): u = 1 + y,
 * log(u) by reduction to m in (sqrt(1/2), sqrt(2)] and the atanh series in s = (m-1)/(m+1), plus the classic
 * correction ((u - 1) - y) / u for the rounding of 1 + y.  Relative error ~3e-16; same code on the device. */
static inline double orc_det_log1p(double y)
{
    static const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    double u = 1.0 + y;
    if (u == 1.0) return y;
    if (!(u > 0.0)) return -__builtin_inf();
    union { double d; uint64_t b; } v; v.d = u;
    int e = (int)((v.b >> 52) & 0x7ff) - 1023;              /* u is normal here (u >= 2^-53) */
    v.b = (v.b & ((1ull << 52) - 1)) | (1023ull << 52);     /* m in [1, 2) */
    double m = v.d;
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0);
    double s2 = s * s;
    double p = 1.0 / 21.0;
    for (int n = 19; n >= 1; n -= 2) p = p * s2 + 1.0 / (double)n;
    double lg = 2.0 * s * p;
    double r = (double)e * LN2_HI + (lg + (double)e * LN2_LO);
    return r - ((u - 1.0) - y) / u;
}

#endif
