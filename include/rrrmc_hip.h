/*
 * rrrmc_hip.h — C ABI of the MI355X (gfx950) Ising Monte Carlo sweep engine.
 *
 * This is the drop-in boundary behind RRRMC.jl's AbstractGraph / standardMC API.
 * In the reference the samplers call the graph through a per-site interface
 * (src/Interface.jl:87-158: energy, delta_energy, spinflip!/update_cache!, neighbors, allΔE);
 * a per-site call across an FFI is useless on a GPU, so the boundary sits one level up: one call
 * runs the WHOLE sampler loop (src/RRRMC.jl:81-127) for a batch of R replicas of one graph.
 *
 * Conventions
 *  - every function returns an int32 status (RRRMC_OK == 0); the text of the last failure is
 *    available from rrrmc_last_error(ctx) (ctx may be NULL for creation failures);
 *    nothing throws or aborts across the ABI (the reference raises ArgumentError, src/RRRMC.jl:94).
 *  - the caller owns every host buffer and keeps it alive for the duration of the call
 *    (Julia: GC.@preserve); the library owns all device memory inside the ctx.
 *  - spins travel in Julia's BitVector chunk layout (src/Interface.jl:21-29, src/Common.jl:15-23):
 *    replica r, site x (0-based) = bit (x & 63) of chunks[r * nchunks + (x >> 6)], nchunks = ceil(N/64);
 *    spin value sigma = 2*bit - 1.  Unused high bits of the last chunk are zero.
 *  - neighbour tables are row-major N x K, 0-BASED int32 (the Julia glue subtracts 1 from
 *    reinterpret(Int64, X.A), src/graphs/RRG.jl:118-119); couplings are N x K int8 (+-1).
 *  - a ctx made by rrrmc_ctx_create is bound to ONE device; rrrmc_ctx_create_multi spreads the replicas of one ctx over several
 *    (one stream and one host thread per device).  No ctx is re-entrant (the reference's graph objects are stateful too:
 *    src/graphs/RRG.jl:121).  A multi-process job instead gives every process its own ctx with a replica offset.
 *  - random numbers: Philox4x32-10 streams addressed by (seed; iteration, replica) — see DESIGN.md
 *    "Random-stream contract"; results do not depend on how replicas are sharded over devices.
 */
#ifndef RRRMC_HIP_H
#define RRRMC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RRRMC_API __attribute__((visibility("default")))

typedef struct rrrmc_ctx rrrmc_ctx;

enum rrrmc_status {
    RRRMC_OK = 0,
    RRRMC_ERR_INVALID_ARG = 1,   /* reference: ArgumentError */
    RRRMC_ERR_STATE = 2,         /* call order violated (e.g. sampling before set_graph) */
    RRRMC_ERR_UNSUPPORTED = 3,   /* model/size outside what the HIP kernels cover */
    RRRMC_ERR_HIP = 4,           /* HIP runtime failure (no device, launch error, ...) */
    RRRMC_ERR_NOMEM = 5
};

enum rrrmc_model {
    RRRMC_MODEL_SPARSE_PM1 = 1,  /* GraphRRG{Int,(-1,1),K} src/graphs/RRG.jl:116 and GraphEA{Int,(-1,1),2D} src/graphs/EA.jl:138 */
    RRRMC_MODEL_SK_NORMAL = 2,   /* GraphSKNormal (Float64 couplings) src/graphs/SK.jl:181-210; K is ignored */
    RRRMC_MODEL_SPARSE_F64 = 5,  /* GraphRRGNormal{K} src/graphs/RRG.jl:503-520 and GraphEANormal{2D} src/graphs/EA.jl:534-552: sparse, Float64 couplings */
    RRRMC_MODEL_SPARSE_DISCRETIZED = 6, /* GraphRRGNormalDiscretized src/graphs/RRG.jl:285-307, GraphEANormalDiscretized src/graphs/EA.jl:311-352 (Int or DFloat64 LEV) */
    RRRMC_MODEL_SPARSE_LEVELS = 7,      /* GraphRRG{ET,LEV,K} src/graphs/RRG.jl:116-162, GraphEA{ET,LEV,2D} src/graphs/EA.jl:138-193 with levels other than (-1,1); ET = Int or DFloat64 */
    RRRMC_MODEL_SK_BINARY = 4,   /* GraphSK (couplings +-1/sqrt(N), bit-packed) src/graphs/SK.jl:28-60; K is ignored; energies Float64 */
    RRRMC_MODEL_QUANT_RRG = 3,   /* GraphQuant over M Suzuki-Trotter slices of one GraphRRG{Int,(-1,1),K} disorder
                                    (src/graphs/QT.jl:126-170 with the shared-disorder pattern of src/QAliases.jl:43-67);
                                    created with rrrmc_ctx_create_quant */
    /* selectors for rrrmc_ctx_create_multi only (the contexts it makes report RRRMC_MODEL_QUANT_RRG): */
    RRRMC_MODEL_QUANT_SK = 8,    /* GraphQuant over binary GraphSK slices (GraphQSKT, src/QAliases.jl:34-43): rrrmc_ctx_create_quant_sk per device */
    RRRMC_MODEL_QUANT_SKN = 9,   /* GraphQuant over GraphSKNormal slices (GraphQSKNormalT, src/QAliases.jl:45-46): rrrmc_ctx_create_quant_skn per device */
    RRRMC_MODEL_QUANT_F64 = 10   /* GraphQuant over sparse Float64 slices (GraphQEAT, src/QAliases.jl:50-83): rrrmc_ctx_create_quant_f64 per device */
};

/* Library ABI version (major*10000 + minor*100 + patch). */
RRRMC_API int32_t rrrmc_version(void);

/* Text of the most recent error on this ctx (or of the last failed rrrmc_ctx_create when ctx == NULL).
 * The pointer stays valid until the next failing call on the same ctx / thread. */
RRRMC_API const char *rrrmc_last_error(const rrrmc_ctx *ctx);

/* Number of visible HIP devices (0 when there is none; never fails). */
RRRMC_API int32_t rrrmc_device_count(void);
/* Measurement helper (bench.py; SURVEY.md §8d "also report a measured device-copy bandwidth"): times `reps` device-to-device copies of
 * `nbytes` with HIP events on a private stream and returns (bytes read + bytes written) per second in GB/s. */
RRRMC_API int32_t rrrmc_device_copy_bandwidth(int32_t device, int64_t nbytes, int32_t reps, double *gbps_out);
/* Page-locked host memory for result buffers (optional).  Every entry point accepts ordinary host memory; a buffer from rrrmc_host_alloc
 * lets the copies of rrrmc_fetch_results / rrrmc_get_spins / rrrmc_standard_mc run at the bus rate instead of through the driver's
 * staging of pageable memory (the reference returns Es and C from every call, src/RRRMC.jl:126: at config 2 that is 64 MiB per call).
 * Free with rrrmc_host_free; both are independent of any context. */
RRRMC_API int32_t rrrmc_host_alloc(int64_t nbytes, void **out);
RRRMC_API int32_t rrrmc_host_free(void *p);

/*
 * Create a context for R replicas (chains) of one graph with N spins and K neighbour slots per spin.
 *   model     rrrmc_model
 *   device    HIP device ordinal
 *   replica0  global id of this ctx's first replica (replica ids address the random streams; a job of
 *             R_total replicas sharded over several devices passes the shard offset here; must be a
 *             multiple of 32)
 * Replaces: constructing the graph object + its LocalFields cache, src/graphs/RRG.jl:122-138.
 */
RRRMC_API int32_t rrrmc_ctx_create(rrrmc_ctx **out, int32_t model, int64_t N, int64_t K, int64_t R,
                                   int32_t device, uint32_t replica0);
/*
 * One context over SEVERAL devices (rrrmc_ctx_create(..., device_ids[], ndev)): what a reference user gets from one
 * standardMC call in one process (src/RRRMC.jl:81-88).  The R replicas are sharded over device_ids[0..ndev-1] in whole 32-replica
 * groups, in global-id order (a device may be named more than once: each entry gets its own shard and stream); replica ids address
 * the random streams, so every result is the one a single-device context gives.  Every function of this header that takes a ctx
 * accepts the multi-device context: per-replica buffers ([R], [R x nsamples], [R x chunks] ...) are the caller's full arrays and
 * arrive gathered; enqueueing calls return when every device has its work queued, rrrmc_sync / fetch / energy calls run one host
 * thread per device.  rrrmc_last_timing / rrrmc_timing_total report the slowest device.
 *   model  any rrrmc_model; RRRMC_MODEL_QUANT_RRG takes (N = Nk, K, M) as rrrmc_ctx_create_quant does, RRRMC_MODEL_QUANT_SK / _SKN take
 *          (N = Nk, M) as rrrmc_ctx_create_quant_sk / _skn do (K ignored), RRRMC_MODEL_QUANT_F64 takes (N = Nk, K, M) as rrrmc_ctx_create_quant_f64;
 *          M is ignored otherwise.
 */
RRRMC_API int32_t rrrmc_ctx_create_multi(rrrmc_ctx **out, int32_t model, int64_t N, int64_t K, int64_t M, int64_t R,
                                         const int32_t *device_ids, int32_t ndev, uint32_t replica0);
RRRMC_API void rrrmc_ctx_destroy(rrrmc_ctx *ctx);

/* Disorder: A[N*K] 0-based neighbours, J[N*K] couplings in {-1,+1}; J[x*K+k] belongs to the bond
 * (x, A[x*K+k]) and must be symmetric (checked).  Replaces GraphRRG{Int,(-1,1),K}(A, J), RRG.jl:122. */
RRRMC_API int32_t rrrmc_set_graph(rrrmc_ctx *ctx, const int32_t *A, const int8_t *J);

/* Random.seed!(seed) (src/RRRMC.jl:89): selects the Philox key and rewinds the iteration counter.
 * Not calling it between two sampling calls continues the streams (the reference's `seed <= 0`). */
RRRMC_API int32_t rrrmc_seed(rrrmc_ctx *ctx, uint64_t seed);

/* Config(N) with random spins for every replica (src/Interface.jl:24-28), INIT stream of the seed. */
RRRMC_API int32_t rrrmc_init_spins_random(rrrmc_ctx *ctx);
/* C0 (src/RRRMC.jl:93): chunks[R * ceil(N/64)] in BitVector layout. */
RRRMC_API int32_t rrrmc_set_spins(rrrmc_ctx *ctx, const uint64_t *chunks);
RRRMC_API int32_t rrrmc_get_spins(rrrmc_ctx *ctx, uint64_t *chunks);

/* energy(X, C) for every replica (src/Interface.jl:105, src/graphs/RRG.jl:164-189): E_out[R] (int64).
 * Also rebuilds the device-side caches, exactly as the reference's `energy` resets LocalFields. */
RRRMC_API int32_t rrrmc_energy(rrrmc_ctx *ctx, int64_t *E_out);

/* Debug/parity view of the cache: lfields_out[R * N] (int64), lfields[x] = -delta_energy(X, C, x)
 * (src/graphs/RRG.jl:236-244), recomputed from the current spins. */
RRRMC_API int32_t rrrmc_get_fields(rrrmc_ctx *ctx, int64_t *lfields_out);

/*
 * standardMC(X, beta, iters; step) for all R replicas (src/RRRMC.jl:81-127).
 *   Es_out        [R * (iters / step)] int64, replica-major: the energy BEFORE the move of iteration
 *                 k*step (src/RRRMC.jl:104-108).  May be NULL.
 *   accepted_out  [R] accepted moves of this call.  May be NULL.
 * The configuration is resumed from the ctx and left updated (C0 is mutated in place, RRRMC.jl:93).
 * Synchronous: returns when the results are on the host.
 */
RRRMC_API int32_t rrrmc_standard_mc(rrrmc_ctx *ctx, double beta, int64_t iters, int64_t step,
                                    int64_t *Es_out, int64_t *accepted_out);

/* Device-resident form of the same call: enqueue on the ctx's stream and return; results stay in HBM
 * until fetched.  rrrmc_sync waits for the stream.  Used by bench.py for the timed region. */
RRRMC_API int32_t rrrmc_standard_mc_async(rrrmc_ctx *ctx, double beta, int64_t iters, int64_t step);
RRRMC_API int32_t rrrmc_sync(rrrmc_ctx *ctx);
/* Results of the last (a)sync sampling call: Es_out[R * nsamples] replica-major, accepted_out[R]. */
RRRMC_API int32_t rrrmc_fetch_results(rrrmc_ctx *ctx, int64_t *Es_out, int64_t *accepted_out);

/* ---- colour-parallel ("checkerboard") sweeps: build-defined extension for large sparse graphs ---------------------
 * The reference's standardMC is random-site (src/RRRMC.jl:113) and its kernel here keeps a replica group's spins in
 * LDS (one or two 4-byte words per site next to the chunk buffers: up to N = 32 767).  For larger graphs (BASELINE.json
 * config 4: GraphEA L=64, D=3) the engine offers sweeps over a
 * proper colouring: one sweep attempts every site once, colour by colour, all sites of a colour at once, with the
 * reference's delta_energy and accept rule (src/graphs/EA.jl:266-275, src/RRRMC.jl:39).  A ctx of
 * RRRMC_MODEL_SPARSE_PM1 whose state does not fit the LDS sweep still runs rrrmc_standard_mc* (random-site, the same chain, through
 * the big-N kernels: N <= 2^20, roughly ten times slower than these sweeps).
 *   color[N] in 0..ncolors-1, adjacent sites must differ (checked).  Call after rrrmc_set_graph: the colouring is stored as per-site
 *   records of that graph, and a later rrrmc_set_graph drops it (rrrmc_colored_sweeps_async then returns RRRMC_ERR_STATE).
 *   rrrmc_colored_sweeps_async: `sweeps` sweeps; an energy sample is taken BEFORE sweep k*step (as RRRMC.jl:104-108
 *   with a sweep as the unit); results through rrrmc_sync + rrrmc_fetch_results (Es [R x sweeps/step]; accepted_out
 *   is not tracked by this sampler and reads -1). */
RRRMC_API int32_t rrrmc_set_coloring(rrrmc_ctx *ctx, const int32_t *color, int32_t ncolors);
RRRMC_API int32_t rrrmc_colored_sweeps_async(rrrmc_ctx *ctx, double beta, int64_t sweeps, int64_t step);
/* on != 0: the colour sweeps also count every replica's accepted moves (fetched like standardMC's accepted counts; a few per cent
 * slower: one 32x32 bit transpose per wavefront and site).  Default off: accepted_out reads -1 after a colour-sweep call. */
RRRMC_API int32_t rrrmc_colored_count_accepted(rrrmc_ctx *ctx, int32_t on);

/* ---- Float64-energy models (RRRMC_MODEL_SK_NORMAL): ET = Float64 (src/graphs/SK.jl:181) --------------------
 * Dense couplings J[N*N] row-major; must be symmetric with a zero diagonal (GraphSKNormal(J; check=true),
 * SK.jl:184-195: violations are RRRMC_ERR_INVALID_ARG).  Replaces GraphSKNormal(J). */
RRRMC_API int32_t rrrmc_set_couplings_dense(rrrmc_ctx *ctx, const double *J);
/* energy / cache / sampler results as Float64 (same meaning as the integer forms above; delta_energy = +lfields,
 * SK.jl:278-284).  rrrmc_standard_mc_async / rrrmc_sync are shared by all models. */
RRRMC_API int32_t rrrmc_energy_f64(rrrmc_ctx *ctx, double *E_out);
RRRMC_API int32_t rrrmc_get_fields_f64(rrrmc_ctx *ctx, double *lfields_out);
RRRMC_API int32_t rrrmc_standard_mc_f64(rrrmc_ctx *ctx, double beta, int64_t iters, int64_t step,
                                        double *Es_out, int64_t *accepted_out);
RRRMC_API int32_t rrrmc_fetch_results_f64(rrrmc_ctx *ctx, double *Es_out, int64_t *accepted_out);
/* RRRMC_MODEL_SK_BINARY: couplings as N BitVector rows of ceil(N/64) chunks (GraphSK(J::Vector{BitVector}), SK.jl:34-48):
 * symmetric, zero diagonal (checked).  The integer cache is read with rrrmc_get_fields (lfields = sqrt(N) * delta_energy,
 * SK.jl:137-140), energies with the _f64 entry points.  rrrmc_gen_sk_binary = gen_J (SK.jl:17-26), SKBITS stream. */
RRRMC_API int32_t rrrmc_set_couplings_bits(rrrmc_ctx *ctx, const uint64_t *J_chunks);
RRRMC_API int32_t rrrmc_gen_sk_binary(int64_t N, uint64_t seed, uint64_t *J_chunks_out);
/* gen_J_gauss (src/graphs/SK.jl:170-179): J_out[N*N], normal(0, 1/N), symmetric, zero diagonal. GAUSS stream. */
RRRMC_API int32_t rrrmc_gen_sk_gauss(int64_t N, uint64_t seed, double *J_out);

/* ---- GraphQuant + rrrMC (RRRMC_MODEL_QUANT_RRG) ---------------------------------------------------------------
 * N = Nk * M spins per replica, slice-major (slice k holds spins k*Nk .. (k+1)*Nk-1, QT.jl:105-108); the slice graph's
 * (A, J) [Nk x K] is given with rrrmc_set_graph.  Energies are Float64: use the _f64 entry points for energy / results.
 * Replaces GraphQuant{fourK,GraphRRG}(...) (QT.jl:139-170).  The slice graph must be simple (no repeated bonds: a GraphEA with L = 2 is
 * refused).  rrrmc_bkl_mc_async / rrrmc_wtm_mc_async run the generic continuous-energy caches over all Nk * M spins (DeltaE.jl:315) with
 * neighbors(X, i) = the two Trotter neighbours, then the slice graph's (QT.jl:288-321); rrrmc_quant_set_field must have been called. */
RRRMC_API int32_t rrrmc_ctx_create_quant(rrrmc_ctx **out, int64_t Nk, int64_t K, int64_t M, int64_t R,
                                         int32_t device, uint32_t replica0);
/* GraphQuant over binary GraphSK slices — GraphQSKT(Nk, M, Gamma, beta) = GraphQuant(Nk, M, Gamma, beta, GraphSK, gen_J(Nk))
 * (src/QAliases.jl:34-43), the graph of the reference's quantum experiment (scripts/scripts.jl:766-864, test_QIsing).  The slice
 * couplings are given with rrrmc_set_couplings_bits (Nk rows of ceil(Nk/64) chunks, as for RRRMC_MODEL_SK_BINARY; rrrmc_gen_sk_binary
 * draws them); everything else is as for rrrmc_ctx_create_quant: rrrmc_quant_set_field, rrrmc_rrr_mc_async, rrrmc_standard_mc_async,
 * rrrmc_energy_f64, rrrmc_quant_observables.  delta_energy_residual = (lfields[i] / sqrt(Nk)) / M (SK.jl:137-140, QT.jl:270-281) with
 * the integer field recomputed by popcounts of the slice's spin words against row i of J. */
RRRMC_API int32_t rrrmc_ctx_create_quant_sk(rrrmc_ctx **out, int64_t Nk, int64_t M, int64_t R, int32_t device, uint32_t replica0);
/* GraphQuant over M GraphSKNormal slices sharing one Gaussian coupling matrix (GraphQSKNormalT, src/QAliases.jl:45-46 — one of the graphs
 * of the reference's own test loop, test/runtests.jl:80): then rrrmc_set_couplings_dense(ctx, J[Nk x Nk]) and rrrmc_quant_set_field.
 * rrrMC (rrrmc_rrr_mc_async) and standardMC are wired (thread per replica, every slice keeps its Float64 lfields / lfields_last / move_last
 * exactly as SK.jl:212-276 updates them); bklMC / wtmMC / extremal_opt and the observables are not (RRRMC_ERR_UNSUPPORTED). */
RRRMC_API int32_t rrrmc_ctx_create_quant_skn(rrrmc_ctx **out, int64_t Nk, int64_t M, int64_t R, int32_t device, uint32_t replica0);
/* GraphQuant over M sparse Float64 slices sharing one (A, J::Float64) — GraphQEAT = GraphQuant{fourK,GraphEANormal{twoD}}
 * (src/QAliases.jl:50-83: GraphQEAT(L, D, M, Gamma, beta), GraphQEAT(fname, M, Gamma, beta), GraphQEAT(X::GraphEANormal, M, Gamma, beta)), and the
 * same over a GraphRRGNormal: then rrrmc_set_graph_f64(ctx, A[Nk x K], J[Nk x K]) (sorted rows, symmetric; a neighbour may repeat: L = 2) and
 * rrrmc_quant_set_field.  Every slice keeps its own LocalFields{Float64}: lfields, the live part of lfields_last (the K + 1 values the undo
 * path of update_cache! reads, src/graphs/EA.jl:613-653 / RRG.jl:576-617) and move_last, per replica.  K <= 8.  Samplers: rrrmc_rrr_mc_async
 * (rrrMC(X::DoubleGraph), src/RRRMC.jl:221-290: delta_energy_residual = -lfields[i] / M, src/graphs/QT.jl:270-281), rrrmc_standard_mc_async, and
 * the generic caches over all Nk * M spins — rrrmc_bkl_mc_async, rrrmc_wtm_mc_async, rrrmc_extremal_opt_async (neighbors = the two Trotter
 * neighbours, then the slice graph's, QT.jl:288-321); one thread per replica.  rrrmc_quant_observables is not wired for these slices. */
RRRMC_API int32_t rrrmc_ctx_create_quant_f64(rrrmc_ctx **out, int64_t Nk, int64_t K, int64_t M, int64_t R, int32_t device, uint32_t replica0);
/* The Trotter coupling fourK (a type parameter of GraphQuant in the reference, QT.jl:126) and the beta it was derived
 * from: needed by rrrmc_energy_f64 before the first rrrMC call, and by rrrmc_standard_mc_async — standardMC on the GraphQuant
 * (src/RRRMC.jl:81-127 with delta_energy = delta_energy(X0) + delta_energy_residual, QT.jl:283-286; SITE + ACCEPT_F64 streams),
 * which takes fourK from here. */
/* GraphQuant over GraphEA slices (ea_form = 1; call before rrrmc_set_graph): the slice graph may list a neighbour twice (L = 2: two bonds
 * to the same site) — delta_energy sums every entry, neighbors() of the continuous-energy caches is the de-duplicated list (EA.jl:158). */
RRRMC_API int32_t rrrmc_quant_slice_form(rrrmc_ctx *ctx, int32_t ea_form);
RRRMC_API int32_t rrrmc_quant_set_field(rrrmc_ctx *ctx, double beta, double fourK);
/* rrrMC(X::DoubleGraph, beta, iters; step, staged_thr, staged_thr_fact) (src/RRRMC.jl:221-290) for all R replicas.
 *   fourK = round(2/beta * log(coth(beta * Gamma / M)), digits = 8)  (QT.jl:165) is computed by the caller.
 * Enqueues on the ctx's stream; rrrmc_sync + rrrmc_fetch_results_f64 return Es [R x iters/step] and accepted [R];
 * rrrmc_rrr_stats adds the number of staged iterations per replica ("frac. staged iters", RRRMC.jl:287). */
RRRMC_API int32_t rrrmc_rrr_mc_async(rrrmc_ctx *ctx, double beta, double fourK, int64_t iters, int64_t step,
                                     double staged_thr, double staged_thr_fact);
RRRMC_API int32_t rrrmc_rrr_stats(rrrmc_ctx *ctx, int64_t *staged_iters_out);
/* rrrmc_rrr_mc_async also serves rrrMC(X::SingleGraph, ...) (src/RRRMC.jl:149-219): on RRRMC_MODEL_SPARSE_PM1 with the
 * integer-level DeltaECache{Int,L} (fourK ignored, results through rrrmc_fetch_results), on RRRMC_MODEL_SK_NORMAL with
 * DeltaECacheCont + DynamicSampler (src/DeltaE.jl:297-410, src/DynamicSamplers.jl; results through rrrmc_fetch_results_f64).
 * On RRRMC_MODEL_SPARSE_F64 (GraphRRGNormal / GraphEANormal) rrrmc_rrr_mc_async, rrrmc_bkl_mc_async and rrrmc_wtm_mc_async run the
 * continuous-energy versions (DeltaECacheCont + DynamicSampler, THeap; results through rrrmc_fetch_results_f64); rrrmc_bkl_mc_async
 * also serves RRRMC_MODEL_SK_NORMAL.  RRRMC_MODEL_SK_BINARY (GraphSK, a SimpleGraph{Float64} too: SK.jl:28) runs rrrMC / bklMC / wtmMC
 * through the same continuous-energy caches over delta_energy = lfields[i] / sqrt(N) (SK.jl:137-140).
 * bklMC(X, beta, iters; step) (src/RRRMC.jl:311-359) on RRRMC_MODEL_SPARSE_PM1: rejection-free Bortz-Kalos-Lebowitz sampler;
 * `iters` counts the skipped rejections too; rrrmc_rrr_stats then returns the number of moves actually made ("true it"). */
RRRMC_API int32_t rrrmc_bkl_mc_async(rrrmc_ctx *ctx, double beta, int64_t iters, int64_t step);
/* wtmMC(X, beta, samples; step) (src/RRRMC.jl:376-426, src/WaitingTimes.jl) on RRRMC_MODEL_SPARSE_PM1: the waiting-time
 * method.  `step` is in sweeps (it is divided by N like RRRMC.jl:391); `samples` energies are taken at global times k*step/N.
 * rrrmc_sync + rrrmc_fetch_results return Es [R x samples] and, as `accepted`, num_moves; rrrmc_wtm_times the final global time
 * of every replica ("global time", RRRMC.jl:421). */
RRRMC_API int32_t rrrmc_wtm_mc_async(rrrmc_ctx *ctx, double beta, int64_t samples, double step);
RRRMC_API int32_t rrrmc_wtm_times(rrrmc_ctx *ctx, double *t_out);
/* extremal_opt(X, tau, iters; step) (src/RRRMC.jl:474-521) on RRRMC_MODEL_SPARSE_PM1, EOCache{Int,L} (src/DeltaE.jl:412-555).
 *   ftau[N] = cumsum(j^-tau, j = 1..N), computed by the caller as the reference does at DeltaE.jl:444-445.
 * rrrmc_sync + rrrmc_fetch_results return the energies the hook would see (E at iterations k*step, before the move; `accepted`
 * is not meaningful); rrrmc_extremal_opt_results returns Emin[R], Cmin (R x ceil(N/64) chunks) and itmin[R]; the final
 * configuration is read with rrrmc_get_spins. */
RRRMC_API int32_t rrrmc_extremal_opt_async(rrrmc_ctx *ctx, const double *ftau, int64_t iters, int64_t step);
RRRMC_API int32_t rrrmc_extremal_opt_results(rrrmc_ctx *ctx, int64_t *Emin_out, uint64_t *Cmin_chunks, int64_t *itmin_out);
/* The same call on the graphs that are not DiscrGraphs — RRRMC_MODEL_SPARSE_F64 (GraphRRGNormal / GraphEANormal) and
 * RRRMC_MODEL_SPARSE_DISCRETIZED (the DoubleGraphs) — runs the reference's generic cache, EOCacheCont (src/DeltaE.jl:557-635): the
 * ranking of all spins by delta_energy is re-established after every flip (the kernel slides the K + 1 changed entries to their
 * places instead of re-sorting) and every run of EQUAL values is put in a fresh uniformly random order per move (rankshuffle!,
 * :611-634; the library's restatement: order by a per-(move, site) Philox key).  N <= 65 535.  RRRMC_MODEL_SK_NORMAL and
 * RRRMC_MODEL_SK_BINARY run it too (every spin a neighbour: the ranking is re-sorted at every flip; N <= 4096).  Energies are Float64:
 * rrrmc_fetch_results_f64, and rrrmc_extremal_opt_results_f64 for (Emin, Cmin, itmin). */
RRRMC_API int32_t rrrmc_extremal_opt_results_f64(rrrmc_ctx *ctx, double *Emin_out, uint64_t *Cmin_chunks, int64_t *itmin_out);
/* parity/debug view of the move-selection cache after the last rrrMC call: pos_out[R * N] = class of every spin
 * (DeltaECache.pos, 0-based a + 2*up), sizes_out[R * 4] = |class k| (DeltaE.jl:63-73).  RRRMC_MODEL_SPARSE_PM1 / _LEVELS (rrrMC and
 * bklMC; class a + L*up) and RRRMC_MODEL_SPARSE_DISCRETIZED: sizes_out[R * 16], class k of replica r at 16 r + k. */
RRRMC_API int32_t rrrmc_rrr_cache(rrrmc_ctx *ctx, int8_t *pos_out, int32_t *sizes_out);

/* Opt-in FAST standardMC for RRRMC_MODEL_SPARSE_F64 (GraphRRGNormal / GraphEANormal, K <= 4, N <= 8192): replicas bit-sliced and
 * resident in LDS as for the +-J models, no stored fields — delta_energy takes 2^K values per site, so the accept test runs against
 * per-site threshold tables.  Same SITE stream as rrrmc_standard_mc_async; the acceptance uniforms come from the ACCEPT bit-plane stream
 * (as the +-J models) instead of ACCEPT_F64, and delta_energy is the pattern sum 2 sum_k +-|J_ik| instead of the incrementally
 * updated cache (last-bit differences), so the chain is NOT the one rrrmc_standard_mc_async produces; it is a faithful standardMC
 * chain with its own oracle restatement (configurations / accepted counts identical to it, energies — re-evaluated at every sample —
 * within 1e-9 relative).  Two orders of magnitude faster.  Results: rrrmc_sync + rrrmc_fetch_results_f64. */
RRRMC_API int32_t rrrmc_standard_mc_fast_async(rrrmc_ctx *ctx, double beta, int64_t iters, int64_t step);

/* Resumed standardMC calls.  A reference call keeps its incrementally updated cache and its tracked energy E from the first to the last
 * iteration, hook calls included (src/RRRMC.jl:95-118); a library call starts, like a fresh reference call, from E = energy(X, C) and a
 * rebuilt cache (:95).  For the integer models the two coincide; for the Float64 models (RRRMC_MODEL_SK_NORMAL / SK_BINARY / SPARSE_F64 /
 * SPARSE_DISCRETIZED, standardMC on a GraphQuant) the last bits differ.  rrrmc_set_resume(ctx, 1): every following standardMC call that
 * finds the state left by a previous standardMC call (no rrrmc_set_spins / rrrmc_init_spins_random / other sampler in between) continues
 * from it — tracked energy, local fields, undo record, move_last — so that a run cut into pieces at the hook points is bit for bit the
 * run made in one call.  rrrmc_tracked_energy_f64 reads that tracked energy (what the reference hands to its hook), R doubles. */
RRRMC_API int32_t rrrmc_set_resume(rrrmc_ctx *ctx, int32_t on);
/* Resumed rrrMC / bklMC / wtmMC / extremal_opt calls — the hook of every sampler (src/RRRMC.jl:152,186 · :224,255 · :314,341 · :379,404 · :477,501).
 * A reference call keeps its whole chain in local variables from the first to the last iteration, hook calls included: the move-selection
 * cache (DeltaECache: the ArraySets in member order, T, z — src/DeltaE.jl:63-73; DeltaECacheCont: the DynamicSampler's tree and its refresh
 * countdown — src/DynamicSamplers.jl:18-33; THeap and the global time — src/WaitingTimes.jl; EOCache / EOCacheCont: the ranking), the graph's
 * own cache, the tracked E, the acc_rate average of rrrMC, `it` / `nextstep` of bklMC, Emin / Cmin / itmin of extremal_opt.  With
 * rrrmc_set_resume(ctx, 1), a call of one of these samplers that finds the RUN left by a previous call of the SAME sampler with the SAME
 * parameters (beta, fourK, staged_thr, staged_thr_fact; bklMC: step; wtmMC: step; extremal_opt: ftau) — and nothing in between that changes
 * the configuration, the disorder, the seed or the caches (rrrmc_set_spins, rrrmc_init_spins_random, rrrmc_seed, rrrmc_set_graph*,
 * rrrmc_energy*, rrrmc_get_fields, any other sampler) — CONTINUES that run instead of starting one (energy(X, C) + a fresh cache):
 *   - the run's iteration counter carries on: samples are taken before the iterations that are multiples of `step` of the RUN's count
 *     (a first call of step - 1 iterations takes none; the next call of `step` iterations takes one before its first move), extremal_opt's
 *     itmin counts from the start of the run;
 *   - bklMC: `iters` more iterations are allowed; `it`, `nextstep` and the pending (skip, move) draw carry on — a call of `step` iterations
 *     ends at the next sample point, with the configuration, E and accepted count the reference hands to its hook there (:336-341);
 *     the reference's loop ends with its LAST SAMPLE (:340-343: the iterations between it and `iters` are never made), so a resumed call
 *     whose allowance does not reach the run's next sample point makes no move;
 *   - wtmMC: `samples` more samples; the heap, the global time and `nextstep` carry on (:399-404);
 *   - rrrmc_fetch_results* / rrrmc_rrr_stats return the samples and the accepted / staged counts of the CALL.
 * A run cut into calls anywhere is bit for bit the run made in one call — energies, configurations, counts and the cache — through every
 * kernel build (the LDS-resident ones write their state back at the end of a call and read it at the start of the next).
 * rrrmc_tracked_energy / _f64 read the tracked E without disturbing the run; rrrmc_get_spins, rrrmc_rrr_cache, rrrmc_wtm_times,
 * rrrmc_extremal_opt_results*, the snapshot and observable calls do not disturb it either.
 * rrrmc_tracked_energy: the integer models' tracked energy (level units), R values. */
RRRMC_API int32_t rrrmc_tracked_energy(rrrmc_ctx *ctx, int64_t *E_out);
/* Debug mode: the reference's latent consistency checks as a switch a user can turn on (its commented-out asserts in update_cache!,
 * src/graphs/RRG.jl:229-231, SK.jl:125-130, 268-273, and its test suite's hook, test/runtests.jl:12-20).  on != 0: after EVERY standardMC
 * call the library recomputes energy(X, C) of every replica from the configuration on the device and compares it with the energy the
 * sampler tracked (exactly for RRRMC_MODEL_SPARSE_PM1; within 1e-10 N for RRRMC_MODEL_SK_NORMAL, where the cached local fields are
 * compared with recomputed ones too); the following rrrmc_sync (or any synchronous call) returns RRRMC_ERR_STATE and names a replica
 * if any value differs.  Costs one energy evaluation per call; off by default. */
RRRMC_API int32_t rrrmc_set_debug_checks(rrrmc_ctx *ctx, int32_t on);
RRRMC_API int32_t rrrmc_tracked_energy_f64(rrrmc_ctx *ctx, double *E_out);

/* Timing of the last sampling call measured with HIP events on the ctx's stream:
 *   total_ms   first planner launch -> last sweep kernel end
 *   sweep_ms   sum of the sweep (dominant) kernel's durations,  sweep_launches = how many launches */
RRRMC_API int32_t rrrmc_last_timing(rrrmc_ctx *ctx, double *total_ms, double *sweep_ms, int32_t *sweep_launches);
/* Accumulated kernel timing over MANY queued async calls (RRRMC_MODEL_SPARSE_PM1 standardMC): rrrmc_timing_accumulate(ctx, 1)
 * gives every sweep launch of every following sampling call its own HIP-event pair (both calls synchronise the stream; the
 * sampling calls in between stay asynchronous); rrrmc_timing_total returns the summed duration of those launches and their count.
 * rrrmc_timing_accumulate(ctx, 0) returns to the per-call bookkeeping of rrrmc_last_timing.  on > 1 additionally creates the event
 * pairs of the first `on` launches right away, so that a timed region of queued calls performs no event creation. */
RRRMC_API int32_t rrrmc_timing_accumulate(rrrmc_ctx *ctx, int32_t on);
RRRMC_API int32_t rrrmc_timing_total(rrrmc_ctx *ctx, double *sweep_ms, int64_t *sweep_launches);

/* The build of spf_team_kernel<K, waves, slots, width> the default standardMC of a RRRMC_MODEL_SPARSE_F64 context launches on its device
 * (teams of `width` replicas, `waves` wavefronts and `slots` in-flight records each); zeros when the one-wavefront kernel runs instead.
 * A reporting helper (bench.py names the kernel its measured traffic belongs to); any output may be NULL. */
RRRMC_API int32_t rrrmc_spf_team_build(rrrmc_ctx *ctx, int32_t *waves_out, int32_t *width_out, int32_t *slots_out);
/* Samples per replica the last sampling call took: the row length of rrrmc_fetch_results*' Es_out.  iters / step for a call that starts a
 * run; a RESUMED call takes a sample before every iteration that is a multiple of `step` of the run's count (see rrrmc_set_resume). */
RRRMC_API int64_t rrrmc_results_samples(const rrrmc_ctx *ctx);
/* Iterations consumed from the current seed's streams so far. */
RRRMC_API int64_t rrrmc_iterations_done(const rrrmc_ctx *ctx);

/* ---- Float64-coupling sparse models (RRRMC_MODEL_SPARSE_F64) ---------------------------
 * GraphRRGNormal / GraphEANormal: SimpleGraph{Float64} on the neighbour table of GraphRRG / GraphEA.  Create with
 * rrrmc_ctx_create(model = RRRMC_MODEL_SPARSE_F64, N, K, R); K <= 8.  A [N x K] 0-based with ascending rows (a neighbour may
 * repeat: EA with L = 2, handled through the de-duplicated list like EA.jl:613-653), J [N x K] Float64, symmetric.
 * Energies / fields / results through the _f64 entry points; sampler: rrrmc_standard_mc_async / rrrmc_standard_mc_f64.
 * Replaces GraphRRGNormal{K}(N) (RRG.jl:503-520), GraphEANormal{twoD}(L, A, J) (EA.jl:539-552). */
RRRMC_API int32_t rrrmc_set_graph_f64(rrrmc_ctx *ctx, const int32_t *A, const double *J);
/* gen_J(Float64, N, A) do randn() end (RRG.jl:71-96, EA.jl:45-71): one GAUSS-stream normal per bond. J_out[N*K]. */
RRRMC_API int32_t rrrmc_gen_couplings_gauss(int64_t N, int64_t K, const int32_t *A, uint64_t seed, double *J_out);

/* ---- DoubleGraphs with discretised Gaussian couplings (RRRMC_MODEL_SPARSE_DISCRETIZED) --------
 * GraphRRGNormalDiscretized{Int,LEV,K} / GraphEANormalDiscretized{Int,LEV,2D}: couplings cJ ~ Normal(0,1) split by
 * discretize (src/Common.jl:38-72) into integer levels dJ (the inner DiscrGraph X0, which drives the DeltaECache) and
 * Float64 residuals rJ.  Create with rrrmc_ctx_create(model = RRRMC_MODEL_SPARSE_DISCRETIZED, N, K, R); N <= 2^28, K <= 8.
 *   lev[nlev]  the levels (distinct integers in -127..127; allΔE(X0) must have <= 8 values)
 *   ea_form    0: GraphRRG conventions (neighbors = non-zero couplings, RRG.jl:133), 1: GraphEA (repeats removed, EA.jl:158)
 * Sampler: rrrmc_rrr_mc_async (rrrMC(X::DoubleGraph), src/RRRMC.jl:221-290; fourK ignored); results through
 * rrrmc_fetch_results_f64 / rrrmc_rrr_stats; rrrmc_rrr_cache returns pos[R*N] and sizes[R*16] (class k of replica r at 16 r + k, k < 2L); energy through rrrmc_energy_f64.
 * rrrmc_bkl_mc_async / rrrmc_wtm_mc_async run bklMC / wtmMC with the continuous-energy caches over the whole graph (a DoubleGraph
 * is not a DiscrGraph: DeltaE.jl:315) and neighbors(X, i) = A[i] (RRG.jl:499) / uA[i] (EA.jl:529).
 * rrrmc_standard_mc_async / rrrmc_standard_mc_f64 run standardMC (src/RRRMC.jl:81-127) with
 * delta_energy = convert(Float64, dE0 + dE1) (RRG.jl:493-497, EA.jl:523-527).
 *
 * Level units.  The reference's level type is Int or DFloat64 (src/DFloats.jl:11-36: the Int64 t = round(x * 10^5) whose
 * Float64 value is t / 10^5; Float64 LEV tuples are mapped to it, RRG.jl:324, EA.jl:357).  lev[] and dJ[] are integer *units*;
 * wherever the reference promotes a level quantity to Float64 the library computes (units * mul) / div, with (mul, div) set by
 * rrrmc_set_level_scale: (1, 1.0) for Int levels (the default), (g, 1e5) for DFloat64 levels with units = t / g, g = gcd of
 * the levels' t.  Call it before rrrmc_energy_f64 / the samplers; rrrmc_discretize_scaled is the matching discretize. */
RRRMC_API int32_t rrrmc_set_graph_discretized(rrrmc_ctx *ctx, const int32_t *A, const int8_t *dJ, const double *rJ,
                                              const int32_t *lev, int32_t nlev, int32_t ea_form);
RRRMC_API int32_t rrrmc_set_level_scale(rrrmc_ctx *ctx, int64_t mul, double div);
/* discretize(cvec, LEV) (src/Common.jl:38-72): d_out[n] levels (units), r_out[n] residuals. */
RRRMC_API int32_t rrrmc_discretize(const double *x, int64_t n, const int32_t *lev, int32_t nlev, int8_t *d_out, double *r_out);
RRRMC_API int32_t rrrmc_discretize_scaled(const double *x, int64_t n, const int32_t *lev, int32_t nlev, int64_t mul, double div,
                                          int8_t *d_out, double *r_out);

/* ---- stand-alone GraphRRG / GraphEA with general levels (RRRMC_MODEL_SPARSE_LEVELS) ----------
 * GraphRRG{ET,LEV,K}(A, J) / GraphEA{ET,LEV,2D}(A, J) with any levels (the reference's tests use (-1,0,1), (-1.0,0.0,1.0), ...:
 * test/runtests.jl:36-60); LEV = (-1, 1) has its own bit-sliced context, RRRMC_MODEL_SPARSE_PM1.  Create with
 * rrrmc_ctx_create(model = RRRMC_MODEL_SPARSE_LEVELS, N, K, R); N <= 2^28, K <= 8, allΔE(X) <= 8 values.
 *   J[N*K], lev[nlev]   integer level units in -127..127 (rrrmc_set_level_scale gives their value: Int levels (1, 1.0), DFloat64
 *                       levels (g, 1e5), see above); ea_form as for rrrmc_set_graph_discretized
 * Samplers: rrrmc_standard_mc_async (SITE stream + ACCEPT_F64 stream: rand53 < exp(-beta dE)), rrrmc_rrr_mc_async (rrrMC(X::SingleGraph),
 * fourK ignored), rrrmc_bkl_mc_async, rrrmc_wtm_mc_async, rrrmc_extremal_opt_async.  Energies (rrrmc_energy, rrrmc_fetch_results,
 * rrrmc_extremal_opt_results) are int64 level units — exactly the reference's Int / DFloat64 payload divided by g. */
RRRMC_API int32_t rrrmc_set_graph_levels(rrrmc_ctx *ctx, const int32_t *A, const int8_t *J, const int32_t *lev, int32_t nlev, int32_t ea_form);
/* gen_J with rand(vLEV) (RRG.jl:71-96, EA.jl:45-71): COUPLING stream, J_out[N*K] in the units of lev[]. */
RRRMC_API int32_t rrrmc_gen_couplings_lev(int64_t N, int64_t K, const int32_t *A, uint64_t seed, const int32_t *lev, int32_t nlev, int8_t *J_out);

/* ---- snapshots and observables -------------------------------------------------------
 * The reference's scripts keep a copy of C.s at every hook call (scripts/scripts.jl:56-66, to_mat :13-21) and later
 * compute time overlaps with pm1dot (:283-295, parseovs :368-405).  Here the copies stay in HBM (a "snapshot" = the
 * context's spin buffer in its native layout) and the popcounts run on the device.
 *   rrrmc_snapshot_reserve  (re)allocates room for nslots snapshots (drops existing ones)
 *   rrrmc_snapshot_store    copies the live configuration into `slot` (device-to-device, on the ctx's stream)
 *   rrrmc_snapshot_get      returns slot as R x ceil(N/64) BitVector chunks, like rrrmc_get_spins (the BitMatrix column dump)
 *   rrrmc_overlaps          q_out[p * R + r] = pm1dot(replica r in slotA[p], replica r in slotB[p]) = N - 2|a xor b|;
 *                           slot -1 names the live configuration */
RRRMC_API int32_t rrrmc_snapshot_reserve(rrrmc_ctx *ctx, int32_t nslots);
RRRMC_API int32_t rrrmc_snapshot_store(rrrmc_ctx *ctx, int32_t slot);
RRRMC_API int32_t rrrmc_snapshot_get(rrrmc_ctx *ctx, int32_t slot, uint64_t *chunks);
RRRMC_API int32_t rrrmc_overlaps(rrrmc_ctx *ctx, int64_t npairs, const int32_t *slotA, const int32_t *slotB, int32_t *q_out);
/* GraphQuant observables of the live configuration of every replica (RRRMC_MODEL_QUANT_RRG):
 *   Qenergy_out[R]   Qenergy(X, C)            (src/graphs/QT.jl:253-268)
 *   tmag_out[R]      transverse_mag(X0, C, beta) (QT.jl:113-122)
 *   ovs_out[R * (M/2)] overlaps(X)            (QT.jl:213-251)
 * Any output may be NULL.  The integer sums run on the device; the Float64 tail follows the reference's operation order. */
RRRMC_API int32_t rrrmc_quant_observables(rrrmc_ctx *ctx, double beta, double Gamma, double *Qenergy_out, double *tmag_out,
                                          double *ovs_out);

/* ---- host-side graph constructors (setup, not hot): the disorder formats of SURVEY.md §8 a7/a8 ---- */
/* gen_RRG (src/graphs/RRG.jl:26-69): A_out[N*K] 0-based, rows ascending. GRAPH stream of `seed`. */
RRRMC_API int32_t rrrmc_gen_rrg(int64_t N, int64_t K, uint64_t seed, int32_t *A_out);
/* gen_EA (src/graphs/EA.jl:24-43): periodic L^D lattice, A_out[L^D * 2D]. */
RRRMC_API int32_t rrrmc_gen_ea(int64_t L, int64_t D, int32_t *A_out);
/* gen_J with LEV = (-1, 1) (src/graphs/RRG.jl:71-96, :154-156): J_out[N*K]. COUPLING stream. */
RRRMC_API int32_t rrrmc_gen_couplings_pm1(int64_t N, int64_t K, const int32_t *A, uint64_t seed, int8_t *J_out);

#ifdef __cplusplus
}
#endif
#endif /* RRRMC_HIP_H */
