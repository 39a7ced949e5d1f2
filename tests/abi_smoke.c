/*
 * abi_smoke.c — the C ABI driven from C.  Compiled by gcc against include/rrrmc_hip.h and linked to librrrmc_hip.so (see
 * tests/test_gpu_abi_smoke.py): a signature drift between header and library is a compile / link error here, where the ctypes
 * and Julia bindings would only misbehave at run time.  It runs small instances of BASELINE configs 2, 3 and 5 through the
 * entry points a reference-side binding uses, the same runs through a two-shard multi-device context (both shards on device 0),
 * and prints what it got as `key v0 v1 ...` lines; the Python test compares them with the oracle.
 *
 *   abi_smoke <seed> <fourK>        (fourK: the Trotter coupling of the GraphQuant case, computed by the caller as QT.jl:165 does)
 */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rrrmc_hip.h"

#define CHECK(call)                                                                                        \
    do {                                                                                                   \
        int32_t rc_ = (call);                                                                              \
        if (rc_ != RRRMC_OK) {                                                                             \
            fprintf(stderr, "%s -> status %d: %s\n", #call, rc_, rrrmc_last_error(ctx));                  \
            return 1;                                                                                      \
        }                                                                                                  \
    } while (0)

static void print_i64(const char *key, const int64_t *v, int64_t n)
{
    printf("%s", key);
    for (int64_t i = 0; i < n; ++i) printf(" %" PRId64, v[i]);
    printf("\n");
}
static void print_f64(const char *key, const double *v, int64_t n)
{
    printf("%s", key);
    for (int64_t i = 0; i < n; ++i) printf(" %a", v[i]);          /* hex floats: exact */
    printf("\n");
}
static void print_u64(const char *key, const uint64_t *v, int64_t n)
{
    printf("%s", key);
    for (int64_t i = 0; i < n; ++i) printf(" %" PRIu64, v[i]);
    printf("\n");
}

/* configs[1] small: GraphRRG(N = 256, K = 3, +-J), standardMC; single context, then two shards of one multi-device context */
static int run_c2(uint64_t seed)
{
    enum { N = 256, K = 3, R = 96, ITERS = 4096, STEP = 512, NS = ITERS / STEP, NCH = (N + 63) / 64 };
    rrrmc_ctx *ctx = NULL;
    static int32_t A[N * K];
    static int8_t J[N * K];
    static int64_t E0[R], Es[R * NS], acc[R], E1[R], Es2[R * NS], acc2[R];
    static uint64_t C1[R * NCH], C2[R * NCH];
    CHECK(rrrmc_gen_rrg(N, K, seed, A));
    CHECK(rrrmc_gen_couplings_pm1(N, K, A, seed, J));
    CHECK(rrrmc_ctx_create(&ctx, RRRMC_MODEL_SPARSE_PM1, N, K, R, 0, 0));
    CHECK(rrrmc_set_graph(ctx, A, J));
    CHECK(rrrmc_seed(ctx, seed));
    CHECK(rrrmc_init_spins_random(ctx));
    CHECK(rrrmc_energy(ctx, E0));
    CHECK(rrrmc_standard_mc(ctx, 1.0, ITERS, STEP, Es, acc));
    CHECK(rrrmc_get_spins(ctx, C1));
    CHECK(rrrmc_energy(ctx, E1));
    rrrmc_ctx_destroy(ctx);
    ctx = NULL;
    print_i64("c2_E0", E0, R);
    print_i64("c2_Es", Es, (int64_t)R * NS);
    print_i64("c2_acc", acc, R);
    print_i64("c2_E1", E1, R);
    print_u64("c2_C1", C1, (int64_t)R * NCH);
    /* the same job as ONE context over two shards (64 + 32 replicas), both on device 0 */
    const int32_t devs[2] = {0, 0};
    CHECK(rrrmc_ctx_create_multi(&ctx, RRRMC_MODEL_SPARSE_PM1, N, K, 0, R, devs, 2, 0));
    CHECK(rrrmc_set_graph(ctx, A, J));
    CHECK(rrrmc_seed(ctx, seed));
    CHECK(rrrmc_init_spins_random(ctx));
    CHECK(rrrmc_standard_mc_async(ctx, 1.0, ITERS, STEP));
    CHECK(rrrmc_sync(ctx));
    CHECK(rrrmc_fetch_results(ctx, Es2, acc2));
    CHECK(rrrmc_get_spins(ctx, C2));
    rrrmc_ctx_destroy(ctx);
    printf("c2_multi_equal %d\n", !memcmp(Es, Es2, sizeof Es) && !memcmp(acc, acc2, sizeof acc) && !memcmp(C1, C2, sizeof C1));
    return 0;
}

/* configs[2] small: GraphSKNormal(N = 64), standardMC, Float64 energies */
static int run_c3(uint64_t seed)
{
    enum { N = 64, R = 16, ITERS = 2048, STEP = 256, NS = ITERS / STEP };
    rrrmc_ctx *ctx = NULL;
    static double Jm[N * N], Es[R * NS], E1[R];
    static int64_t acc[R];
    CHECK(rrrmc_gen_sk_gauss(N, seed, Jm));
    CHECK(rrrmc_ctx_create(&ctx, RRRMC_MODEL_SK_NORMAL, N, 0, R, 0, 0));
    CHECK(rrrmc_set_couplings_dense(ctx, Jm));
    CHECK(rrrmc_seed(ctx, seed));
    CHECK(rrrmc_init_spins_random(ctx));
    CHECK(rrrmc_standard_mc_f64(ctx, 1.0, ITERS, STEP, Es, acc));
    CHECK(rrrmc_energy_f64(ctx, E1));
    rrrmc_ctx_destroy(ctx);
    print_f64("c3_Es", Es, (int64_t)R * NS);
    print_i64("c3_acc", acc, R);
    print_f64("c3_E1", E1, R);
    return 0;
}

/* configs[4] small: GraphQuant(GraphRRG(Nk = 32, K = 3), M = 4) under rrrMC; single context and two shards */
static int run_c5(uint64_t seed, double fourK)
{
    enum { NK = 32, K = 3, M = 4, R = 40, ITERS = 3000, STEP = 500, NS = ITERS / STEP };
    const double beta = 2.0;
    rrrmc_ctx *ctx = NULL;
    static int32_t A[NK * K];
    static int8_t J[NK * K];
    static double Es[R * NS], Es2[R * NS];
    static int64_t acc[R], staged[R], acc2[R], staged2[R];
    CHECK(rrrmc_gen_rrg(NK, K, seed, A));
    CHECK(rrrmc_gen_couplings_pm1(NK, K, A, seed, J));
    for (int pass = 0; pass < 2; ++pass) {
        const int32_t devs[2] = {0, 0};
        if (pass == 0) CHECK(rrrmc_ctx_create_quant(&ctx, NK, K, M, R, 0, 0));
        else CHECK(rrrmc_ctx_create_multi(&ctx, RRRMC_MODEL_QUANT_RRG, NK, K, M, R, devs, 2, 0));
        CHECK(rrrmc_set_graph(ctx, A, J));
        CHECK(rrrmc_quant_set_field(ctx, beta, fourK));
        CHECK(rrrmc_seed(ctx, seed));
        CHECK(rrrmc_init_spins_random(ctx));
        CHECK(rrrmc_rrr_mc_async(ctx, beta, fourK, ITERS, STEP, 0.5, 5.0));
        CHECK(rrrmc_sync(ctx));
        CHECK(rrrmc_fetch_results_f64(ctx, pass ? Es2 : Es, pass ? acc2 : acc));
        CHECK(rrrmc_rrr_stats(ctx, pass ? staged2 : staged));
        rrrmc_ctx_destroy(ctx);
        ctx = NULL;
    }
    print_f64("c5_Es", Es, (int64_t)R * NS);
    print_i64("c5_acc", acc, R);
    print_i64("c5_staged", staged, R);
    printf("c5_multi_equal %d\n", !memcmp(Es, Es2, sizeof Es) && !memcmp(acc, acc2, sizeof acc) && !memcmp(staged, staged2, sizeof staged));
    return 0;
}

int main(int argc, char **argv)
{
    const uint64_t seed = argc > 1 ? strtoull(argv[1], NULL, 0) : 12345u;
    const double fourK = argc > 2 ? strtod(argv[2], NULL) : 1.0;
    rrrmc_ctx *ctx = NULL;          /* (for CHECK's error text before a context exists) */
    (void)ctx;
    printf("version %d devices %d\n", rrrmc_version(), rrrmc_device_count());
    if (rrrmc_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 2; }
    /* errors come back as status codes with a text, never as an abort (the reference raises ArgumentError, src/RRRMC.jl:94) */
    const int32_t rc = rrrmc_ctx_create(&ctx, RRRMC_MODEL_SPARSE_PM1, 0, 3, 8, 0, 0);
    printf("bad_create status %d text_len %d\n", rc, (int)strlen(rrrmc_last_error(NULL)));
    if (run_c2(seed) || run_c3(seed) || run_c5(seed, fourK)) return 1;
    printf("done\n");
    return 0;
}
