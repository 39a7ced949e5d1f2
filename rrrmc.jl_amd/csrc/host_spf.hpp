// Host side of the Float64-coupling sparse models (GraphRRGNormal / GraphEANormal).
// Included by rrrmc_hip.hip inside its anonymous namespace, after the context struct and the common helpers
// (fail, HIP_TRY, free_dev, ensure_state); not a stand-alone translation unit.
// ---- Float64-coupling sparse models (GraphRRGNormal / GraphEANormal) host side ------------------------------------
constexpr int64_t kSpfItersPerLaunch = 1 << 20;

int32_t spf_ctx_create(rrrmc_ctx** out, int64_t N, int64_t K, int64_t R, int32_t device, uint32_t replica0)
{
    if (N < 1 || K < 1 || R < 1) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "N, K, R must be >= 1 (given N=%lld K=%lld R=%lld)", (long long)N, (long long)K, (long long)R);
    if (K > kSpfMaxK) return fail(nullptr, RRRMC_ERR_UNSUPPORTED, "K=%lld: the Float64 sparse kernels cover K <= %d", (long long)K, kSpfMaxK);
    if (N > (int64_t)INT32_MAX / 64) return fail(nullptr, RRRMC_ERR_UNSUPPORTED, "N=%lld is too large", (long long)N);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, RRRMC_ERR_HIP, "no HIP device is visible: this library has no CPU path");
    if (device < 0 || device >= ndev) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "device %d out of range (0..%d)", device, ndev - 1);
    rrrmc_ctx* ctx = new (std::nothrow) rrrmc_ctx();
    if (!ctx) return fail(nullptr, RRRMC_ERR_NOMEM, "out of host memory");
    ctx->model = RRRMC_MODEL_SPARSE_F64; ctx->N = N; ctx->K = K; ctx->R = R;
    ctx->pfW = (R + 63) / 64; ctx->Rpad = ctx->pfW * 64;
    ctx->device = device; ctx->replica0 = replica0;
#define PF_TRY(expr)                                                                                             \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess) {                                                                                  \
            int32_t rc_ = fail(nullptr, e_ == hipErrorOutOfMemory ? RRRMC_ERR_NOMEM : RRRMC_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
            rrrmc_ctx_destroy(ctx);                                                                              \
            return rc_;                                                                                          \
        }                                                                                                        \
    } while (0)
    PF_TRY(hipSetDevice(device));
    PF_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    PF_TRY(hipEventCreate(&ctx->ev_begin));
    PF_TRY(hipEventCreate(&ctx->ev_end));
    const size_t nf = (size_t)ctx->Rpad * (size_t)N;
    const size_t nsw = (size_t)ctx->pfW * (size_t)N;
    PF_TRY(hipMalloc(&ctx->d_A, sizeof(int32_t) * N * K));
    PF_TRY(hipMalloc(&ctx->pf_J, sizeof(double) * N * K));
    PF_TRY(hipMalloc(&ctx->sk_lf, sizeof(double) * nf));
    PF_TRY(hipMalloc(&ctx->pf_undo, sizeof(double) * (size_t)ctx->Rpad * (size_t)(K + 1)));
    PF_TRY(hipMalloc(&ctx->pf_spins, sizeof(unsigned long long) * nsw));
    PF_TRY(hipMalloc(&ctx->pf_sites, sizeof(int32_t) * (kSpfItersPerLaunch + 2 * kSpfDepth)));
    PF_TRY(hipMalloc(&ctx->sk_move_last, sizeof(int32_t) * ctx->Rpad));
    PF_TRY(hipMalloc(&ctx->sk_E, sizeof(double) * ctx->Rpad));
    PF_TRY(hipMalloc(&ctx->d_acc, sizeof(int64_t) * ctx->Rpad));
    PF_TRY(hipMemset(ctx->pf_spins, 0, sizeof(unsigned long long) * nsw));
    PF_TRY(hipMemset(ctx->sk_lf, 0, sizeof(double) * nf));
    PF_TRY(hipMemset(ctx->pf_undo, 0, sizeof(double) * (size_t)ctx->Rpad * (size_t)(K + 1)));
#undef PF_TRY
    *out = ctx;
    return RRRMC_OK;
}

SpfParams spf_params(rrrmc_ctx* ctx)
{
    SpfParams P{};
    P.A = ctx->d_A; P.J = ctx->pf_J; P.sites = ctx->pf_sites; P.spins = ctx->pf_spins; P.lf = ctx->sk_lf; P.undo = ctx->pf_undo;
    P.move_last = ctx->sk_move_last; P.E_cur = ctx->sk_E; P.acc_cur = ctx->d_acc; P.Es = ctx->sk_Es;
    P.k0 = (uint32_t)ctx->seed; P.k1 = (uint32_t)(ctx->seed >> 32); P.replica0 = ctx->replica0;
    P.N = (int)ctx->N; P.Rpad = (int)ctx->Rpad;
    return P;
}

typedef void (*spf_fn)(SpfParams);
spf_fn spf_sweep_for_K(int K) { RRRMC_DISPATCH_UPTO8(K, spf_sweep_kernel) }
spf_fn spf_energy_for_K(int K) { RRRMC_DISPATCH_UPTO8(K, spf_energy_kernel) }
spf_fn spf_fields_for_K(int K) { RRRMC_DISPATCH_UPTO8(K, spf_fields_kernel) }

// spf_team_kernel (spf_team_kernel.hpp, launched through spf_team_api.hpp): teams of wavefronts walk the chains of a group of 64 replicas.
// A chain's pace is bound by what one compute unit moves, so the groups are split into teams of 32 or 16 replicas until there are as many teams
// as compute units; sixteen wavefronts per team (fifteen executing) while the teams are few enough to own a compute unit each and their
// records fit its LDS, eight otherwise.
constexpr int64_t kSpfTeamItersPerLaunch = 1 << 18;     // spf_plan_stride(K) = 4 + 3 K dwords of plan per iteration
struct SpfTeamBuild { int nw, tw; size_t lds; };
inline SpfTeamBuild spf_team_build(const rrrmc_ctx* ctx)
{
    const int K = (int)ctx->K;
    int ncu = 256;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || ncu < 1) ncu = 256;
    int tw = 64;
    while (tw > 16 && ctx->pfW * (64 / tw) < ncu) tw /= 2;
    // tests / timing experiments: RRRMC_SPF_TEAM_WIDTH = 64 | 32 | 16 first — the number of wavefronts follows from the width that will run;
    // RRRMC_SPF_TEAM_WAVES = 8 | 16 (a team narrower than a group exists with sixteen wavefronts only: a requested width below 64 keeps 16)
    bool width_forced = false;
    if (const char* e = std::getenv("RRRMC_SPF_TEAM_WIDTH"); e && (!std::strcmp(e, "64") || !std::strcmp(e, "32") || !std::strcmp(e, "16"))) { tw = std::atoi(e); width_forced = true; }
    int nw = (ctx->pfW * (64 / tw) <= ncu || (width_forced && tw < 64)) ? 16 : 8;
    if (const char* e = std::getenv("RRRMC_SPF_TEAM_WAVES"); e && (!std::strcmp(e, "8") || !std::strcmp(e, "16")) && !(width_forced && tw < 64)) nw = std::atoi(e);
    // K = 7, 8 (GraphEANormal in four dimensions): the fused pairs need more registers than a sixteen-wavefront workgroup leaves per thread, so
    // narrow teams of these degrees run EIGHT wavefronts (256 registers per thread, 42 slots) with fused pairs — 3.5 against 3.0·10¹⁰ attempts/s
    // at 8192 replicas (profiles/r06/spf_sizes.jsonl); RRRMC_SPF_TEAM_WAVES=16 keeps the sixteen-wavefront build without fusion (the tests' cross-check)
    const char* wv_env = std::getenv("RRRMC_SPF_TEAM_WAVES");
    if (K >= 7 && tw < 64 && !(wv_env && !std::strcmp(wv_env, "16"))) nw = 8;
    if (nw == 8 && !(K >= 7 && tw < 64)) tw = 64;          // (for K <= 6 the eight-wavefront build exists for whole groups only)
    size_t lds = spf_team_build_lds(K, nw, tw);
    if (!lds && nw == 16 && tw == 64) { nw = 8; lds = spf_team_build_lds(K, nw, tw); }      // K >= 5: the records of sixteen wavefronts x 64 replicas do not fit
    return SpfTeamBuild{nw, tw, lds};
}
// The team kernel is the default (graphs with two bonds to the same neighbour included: GraphEANormal with L = 2); spf_sweep_kernel (one
// wavefront per group) stays as the cross-check of the tests (RRRMC_SPF_TEAM=0).
inline bool spf_use_team(const rrrmc_ctx* ctx)
{
    // the fused pairs address a team's field lines with 32-bit byte offsets (site * 512 + lane * 8): beyond 2^23 sites the one-wavefront kernel runs
    if (ctx->N >= (int64_t)1 << 23) return false;
    const char* e = std::getenv("RRRMC_SPF_TEAM");
    return !(e && e[0] == '0');
}

int32_t spf_run_energy(rrrmc_ctx* ctx)
{
    if (const char* e = std::getenv("RRRMC_SPF_ENERGY_V1"); e && e[0] == '1') {           // tests: one wavefront per group walks the sites
        hipLaunchKernelGGL(spf_energy_for_K((int)ctx->K), dim3((unsigned)ctx->pfW), dim3(64), 0, ctx->stream, spf_params(ctx));
    } else {
        hipLaunchKernelGGL(spf_fields_for_K((int)ctx->K), dim3((unsigned)(((ctx->N + 3) / 4) * ctx->pfW)), dim3(256), 0, ctx->stream, spf_params(ctx));
        HIP_TRY(ctx, hipGetLastError());
        hipLaunchKernelGGL(spf_energy_sum_kernel, dim3((unsigned)ctx->pfW), dim3(64), 0, ctx->stream, spf_params(ctx));
    }
    HIP_TRY(ctx, hipGetLastError());
    ctx->pf_lf_live = true;
    return RRRMC_OK;
}

// debug mode (rrrmc_set_debug_checks) after a standardMC call on GraphRRGNormal / GraphEANormal: the reference's commented-out check of
// update_cache! (RRG.jl:611-615: energy(X, C) again, maximum(abs.(lfields_bk - lfields)) < 1e-10) and the test suite's tracked E == energy(X, C)
// (test/runtests.jl:12-20), on the device: fields and energy recomputed from the configuration into scratch buffers, compared element-wise.
// Also the cross-check of spf_team_kernel that needs no second kernel: a lost update shows as a field off by a multiple of 4 J.
int32_t spf_debug_check(rrrmc_ctx* ctx)
{
    const size_t nf = (size_t)ctx->Rpad * (size_t)ctx->N;
    if (!ctx->dbg_flag) { HIP_TRY(ctx, hipMalloc(&ctx->dbg_flag, sizeof(int32_t) * 2)); HIP_TRY(ctx, hipMemsetAsync(ctx->dbg_flag, 0, sizeof(int32_t) * 2, ctx->stream)); }
    if (!ctx->dbg_lf) {
        HIP_TRY(ctx, hipMalloc(&ctx->dbg_lf, sizeof(double) * nf));
        HIP_TRY(ctx, hipMalloc(&ctx->dbg_E, sizeof(double) * ctx->Rpad));
        HIP_TRY(ctx, hipMalloc(&ctx->dbg_ml, sizeof(int32_t) * ctx->Rpad));
    }
    hipStream_t st = ctx->stream;
    SpfParams P = spf_params(ctx);
    P.lf = ctx->dbg_lf; P.E_cur = ctx->dbg_E; P.move_last = ctx->dbg_ml;          // nothing of the live state is written
    hipLaunchKernelGGL(spf_fields_for_K((int)ctx->K), dim3((unsigned)(((ctx->N + 3) / 4) * ctx->pfW)), dim3(256), 0, st, P);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(spf_energy_sum_kernel, dim3((unsigned)ctx->pfW), dim3(64), 0, st, P);
    HIP_TRY(ctx, hipGetLastError());
    if (dbg_inject()) hipLaunchKernelGGL(dbg_inject_f64_kernel, dim3(1), dim3(1), 0, st, ctx->dbg_E);                 // (RRRMC_DEBUG_INJECT=1: tests only)
    const double tol = 1e-10 * (double)(ctx->K + 1);             // the reference's 1e-10 per field; the energy is a sum of N of them
    // fields: [W][N][64], replica of element e = (e / (N * 64)) * 64 + e % 64
    hipLaunchKernelGGL(dbg_compare_f64_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, ctx->sk_lf, ctx->dbg_lf, (long long)nf, tol,
                       (long long)ctx->N * 64, 64, ctx->dbg_flag);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(dbg_compare_f64_kernel, dim3((unsigned)((ctx->R + 255) / 256)), dim3(256), 0, st, ctx->sk_E, ctx->dbg_E, (long long)ctx->R, 1e-10 * (double)ctx->N,
                       (long long)ctx->Rpad, (int)ctx->Rpad, ctx->dbg_flag);
    HIP_TRY(ctx, hipGetLastError());
    return RRRMC_OK;
}

int32_t spf_standard_mc_async(rrrmc_ctx* ctx, double beta, int64_t iters, int64_t step)
{
    if (iters < 0) return fail(ctx, RRRMC_ERR_INVALID_ARG, "iters must be >= 0, given %lld", (long long)iters);
    if (step < 1) return fail(ctx, RRRMC_ERR_INVALID_ARG, "step must be >= 1, given %lld", (long long)step);
    if (std::isnan(beta)) return fail(ctx, RRRMC_ERR_INVALID_ARG, "beta is NaN");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->results_valid = false; ctx->last_call_wtm = false; ctx->last_call_eo = false;
    ctx->timing_valid = false;
    ctx->last_call_rrr = false;
    const int64_t nsamp = iters / step;
    const size_t es_need = (size_t)(nsamp > 0 ? nsamp : 1) * ctx->Rpad;
    if (es_need > ctx->sk_Es_cap) {
        free_dev(ctx->sk_Es);
        ctx->sk_Es_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->sk_Es, sizeof(double) * es_need));
        ctx->sk_Es_cap = es_need;
    }
    const int64_t nl = (iters + kSpfTeamItersPerLaunch - 1) / kSpfTeamItersPerLaunch;      // the shorter of the two launch lengths
    while ((int64_t)ctx->ev_sweep.size() < 2 * (nl > 0 ? nl : 1)) {
        hipEvent_t e;
        HIP_TRY(ctx, hipEventCreate(&e));
        ctx->ev_sweep.push_back(e);
    }
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_begin, st));
    // E = energy(X, C) at the start of every call (RRRMC.jl:95) — unless the call resumes the previous one (rrrmc_set_resume)
    if (!(ctx->resume && ctx->std_cache_live && ctx->pf_lf_live)) { const int32_t rc = spf_run_energy(ctx); if (rc) return rc; }
    ctx->std_cache_live = true;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_acc, 0, sizeof(int64_t) * ctx->Rpad, st));
    spf_fn fn = spf_sweep_for_K((int)ctx->K);
    const bool team = spf_use_team(ctx);
    const SpfTeamBuild tb = team ? spf_team_build(ctx) : SpfTeamBuild{0, 0, 0};
    if (team && !tb.lds) return fail(ctx, RRRMC_ERR_UNSUPPORTED, "spf_team_kernel has no build for K=%lld", (long long)ctx->K);
    if (team) {
        if (!ctx->pf_plan) HIP_TRY(ctx, hipMalloc(&ctx->pf_plan, sizeof(uint32_t) * (size_t)(kSpfTeamItersPerLaunch + 2) * (size_t)spf_plan_stride((int)ctx->K)));
        if (!ctx->pf_status) {
            HIP_TRY(ctx, hipMalloc(&ctx->pf_status, sizeof(int32_t)));
            HIP_TRY(ctx, hipMemsetAsync(ctx->pf_status, 0, sizeof(int32_t), ctx->stream));
        }
    }
    const int64_t per_launch = team ? kSpfTeamItersPerLaunch : kSpfItersPerLaunch;
    int64_t done = 0;
    int launches = 0;
    while (done < iters) {
        const int64_t n = std::min<int64_t>(per_launch, iters - done);
        const int64_t nsites = n + 2 * kSpfDepth;       // the kernel requests (and never consumes) data of the iterations just past its end
        hipLaunchKernelGGL(spf_sites_kernel, dim3((unsigned)((nsites + 255) / 256)), dim3(256), 0, st, ctx->pf_sites, nsites, ctx->it_done + (uint64_t)done,
                           (uint32_t)ctx->seed, (uint32_t)(ctx->seed >> 32), (uint32_t)ctx->N);
        HIP_TRY(ctx, hipGetLastError());
        SpfParams P = spf_params(ctx);
        P.beta = beta; P.g0 = ctx->it_done + (uint64_t)done; P.iters = n; P.step = step;
        P.it_off = done; P.sample0 = done / step;
        if (team) {
            HIP_TRY(ctx, spf_team_plan_launch(ctx->d_A, ctx->pf_J, ctx->pf_sites, ctx->pf_plan, n, (int)ctx->K, st));
        }
        HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[2 * launches], st));
        if (team) {
            SpfTeamParams TP{P, ctx->pf_plan, ctx->pf_status};
            HIP_TRY(ctx, spf_team_launch((int)ctx->K, tb.nw, tb.tw, (int)ctx->pfW, st, TP));
        } else {
            hipLaunchKernelGGL(fn, dim3((unsigned)ctx->pfW), dim3(64), 0, st, P);
        }
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[2 * launches + 1], st));
        ++launches;
        done += n;
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev_end, st));
    ctx->sweep_launches = launches;
    ctx->nsamp = nsamp;
    ctx->it_done += (uint64_t)iters;
    ctx->results_valid = true;
    ctx->timing_valid = true;
    return ctx->debug_checks ? spf_debug_check(ctx) : RRRMC_OK;
}

// rrrMC / bklMC / wtmMC (modes 0 / 1 / 2) with the continuous-energy caches and extremal_opt (mode 3, EOCacheCont; `ftau` = its rank table)
// on GraphRRGNormal / GraphEANormal (cont_kernels.hpp)
int32_t spf_cont_async(rrrmc_ctx* ctx, int mode, double beta, int64_t iters, int64_t step, double stepf, double staged_thr, double staged_thr_fact,
                       const double* ftau = nullptr)
{
    if (iters < 0) return fail(ctx, RRRMC_ERR_INVALID_ARG, "iters / samples must be >= 0, given %lld", (long long)iters);
    if (mode != 2 && step < 1) return fail(ctx, RRRMC_ERR_INVALID_ARG, "step must be >= 1, given %lld", (long long)step);
    if (mode == 2 && (!(stepf > 0.0) || !std::isfinite(stepf))) return fail(ctx, RRRMC_ERR_INVALID_ARG, "step must be positive and finite, given %g", stepf);
    if (!std::isfinite(beta)) return fail(ctx, RRRMC_ERR_INVALID_ARG, "beta must be finite, given: %g", beta);
    // extremal_opt addresses its tie-break keys with 16 site bits of the Philox tag (and ranks all N spins per move); the others are 32-bit beyond
    if (mode == 3 && ctx->N > 65535) return fail(ctx, RRRMC_ERR_UNSUPPORTED, "N=%lld: extremal_opt on a continuous-energy graph addresses spins with 16 bits", (long long)ctx->N);
    if (ctx->N > (int64_t)1 << 28) return fail(ctx, RRRMC_ERR_UNSUPPORTED, "N=%lld is beyond the thread-per-replica samplers (N <= 2^28)", (long long)ctx->N);
    if (mode == 3) {
        if (!ftau) return fail(ctx, RRRMC_ERR_INVALID_ARG, "ftau is NULL");
        for (int64_t i = 0; i < ctx->N; ++i)
            if (!(ftau[i] > 0.0) || !std::isfinite(ftau[i]) || (i && ftau[i] < ftau[i - 1]))
                return fail(ctx, RRRMC_ERR_INVALID_ARG, "ftau must be a positive non-decreasing table (cumsum of j^-tau), violated at %lld", (long long)i);
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->results_valid = false; ctx->last_call_wtm = false; ctx->last_call_eo = false;
    ctx->timing_valid = false;
    // the discretised DoubleGraphs (bklMC / wtmMC over the whole graph: DeltaE.jl:315) keep their spins in the kernel's layout already
    const bool quantm = ctx->model == RRRMC_MODEL_QUANT_RRG;          // bklMC / wtmMC over the whole GraphQuant (DeltaE.jl:315): spins in q_spins too
    if (quantm && !(ctx->last_fourK > 0.0)) return fail(ctx, RRRMC_ERR_STATE, "a GraphQuant needs fourK: call rrrmc_quant_set_field first");
    const bool dblm = ctx->model == RRRMC_MODEL_SPARSE_DISCRETIZED || quantm;
    const int64_t N = ctx->N, K = ctx->K, R = ctx->R, W = dblm ? ctx->qW : (N + 31) / 32;
    int levs = 0;
    while (((int64_t)1 << levs) < N) ++levs;
    const int64_t N2 = (int64_t)1 << levs;
    SmpState S{};
    { const int32_t rcs = smp_begin(ctx, mode + 1, beta, mode == 2 ? stepf : staged_thr, staged_thr_fact, quantm ? ctx->last_fourK : 0.0, mode == 2 ? 1 : step, ftau, &S); if (rcs) return rcs; }
    if (!ctx->cs_top) HIP_TRY(ctx, hipMalloc(&ctx->cs_top, sizeof(double) * (size_t)R * (size_t)std::max<int64_t>(N2 >> kCwBottom, 1)));
    if (!ctx->cs_buf) {
        if (!dblm) HIP_TRY(ctx, hipMalloc(&ctx->cs_spins, sizeof(uint32_t) * R * W));
        HIP_TRY(ctx, hipMalloc(&ctx->cs_buf, sizeof(double) * (size_t)R * (size_t)(2 * N + 2 * N2 + K + 1)));
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->cs_u16), (N > 65535 ? sizeof(uint32_t) : sizeof(uint16_t)) * (size_t)R * 2 * N));
        HIP_TRY(ctx, hipMalloc(&ctx->rs_status, sizeof(int32_t) * R));
        if (!ctx->q_stats) HIP_TRY(ctx, hipMalloc(&ctx->q_stats, sizeof(int64_t) * R * 3));
        if (!ctx->wt_time) HIP_TRY(ctx, hipMalloc(&ctx->wt_time, sizeof(double) * R));
    }
    const int64_t nsamp = mode == 2 ? iters : smp_nsamp(ctx, iters, step);
    const size_t es_need = (size_t)(nsamp > 0 ? nsamp : 1) * ctx->Rpad;
    if (es_need > ctx->sk_Es_cap) {
        free_dev(ctx->sk_Es);
        ctx->sk_Es_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->sk_Es, sizeof(double) * es_need));
        ctx->sk_Es_cap = es_need;
    }
    while (ctx->ev_sweep.size() < 2) {
        hipEvent_t e;
        HIP_TRY(ctx, hipEventCreate(&e));
        ctx->ev_sweep.push_back(e);
    }
    ContParams P{};
    P.S = S; P.samples_before = mode == 2 ? ctx->smp_it : 0; P.ps_top = ctx->cs_top;
    if (mode == 3) {
        if (!ctx->eo_cmin) {
            HIP_TRY(ctx, hipMalloc(&ctx->eo_cmin, sizeof(uint32_t) * R * W));
            HIP_TRY(ctx, hipMalloc(&ctx->eo_ftau, sizeof(double) * N));
        }
        ctx->eo_W = W;
        if (!S.resume) {                // (a resumed call: the run's table is on the device already)
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            HIP_TRY(ctx, hipMemcpy(ctx->eo_ftau, ftau, sizeof(double) * N, hipMemcpyHostToDevice));
        }
        P.ftau = ctx->eo_ftau; P.cmin = ctx->eo_cmin;
    }
    double* b = ctx->cs_buf;
    P.A = ctx->d_A; P.J = quantm ? nullptr : dblm ? ctx->db_rJ : ctx->pf_J; P.spins = dblm ? ctx->q_spins : ctx->cs_spins;
    P.dJ = dblm && !quantm ? ctx->db_dJ : nullptr; P.lev_mul = ctx->db_lev_mul; P.lev_div = ctx->db_lev_div;
    if (quantm) {
        P.qJ = ctx->d_J; P.fourK = ctx->last_fourK; P.qNk = (int)ctx->qNk; P.qM = (int)ctx->qM;
        P.qkind = ctx->q_spf ? 4 : ctx->q_skn ? 3 : (ctx->q_sk ? 2 : 1);
        if (ctx->q_spf) { P.qJ = nullptr; P.qJf = ctx->q_Jf; P.qflf = ctx->q_flf; P.qfundo = ctx->q_fundo; P.qfml = ctx->q_fml; }
        if (ctx->q_sk) { P.qJb = ctx->q_Jb; P.qWk = (int)ctx->q_Wk; P.qsN = std::sqrt((double)ctx->qNk); }
        if (ctx->q_skn) { P.qJd = ctx->sk_J; P.qslf = ctx->q_slf; P.qsmv = ctx->q_smv; P.qscur = ctx->q_scur; }
    }
    P.lf = b; b += (size_t)R * N;
    P.dEs = b; b += (size_t)R * N;
    P.v = b; b += (size_t)R * N2;
    P.ps = b; b += (size_t)R * N2;
    P.undo = b;
    P.hid = ctx->cs_u16;
    P.hpos = N > 65535 ? static_cast<void*>(reinterpret_cast<uint32_t*>(ctx->cs_u16) + (size_t)R * N) : static_cast<void*>(ctx->cs_u16 + (size_t)R * N);
    P.E_cur = ctx->sk_E; P.stats = ctx->q_stats; P.Es = ctx->sk_Es; P.t_out = ctx->wt_time; P.status = ctx->rs_status;
    P.beta = beta; P.staged_thr = staged_thr; P.lambda = staged_thr_fact / (double)N; P.stepf = stepf;
    P.g0 = mode == 1 ? ctx->smp_g0 : ctx->it_done;          // bklMC numbers its moves from the start of the RUN
    P.iters = iters; P.step = step;
    P.k0 = (uint32_t)ctx->seed; P.k1 = (uint32_t)(ctx->seed >> 32); P.replica0 = ctx->replica0; P.call = ctx->smp_call & 0xffffffu;
    P.N = (int)N; P.K = (int)K; P.N2 = (int)N2; P.levs = levs; P.W = (int)W; P.R = (int)R; P.Rp = (int)ctx->Rpad;
    P.ea_form = 1;           // de-duplicate repeated neighbours (EA.jl:680); a no-op for GraphRRGNormal tables
    P.mode = mode;
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_begin, st));
    if (!dblm) {
        hipLaunchKernelGGL(cont_spins_in_kernel, dim3((unsigned)((W + 255) / 256), (unsigned)R), dim3(256), 0, st, ctx->pf_spins, ctx->cs_spins, (int)N, (int)W, (int)R);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[0], st));
    const char* no_wave = std::getenv("RRRMC_EO_NO_WAVE");            // tests / timing experiments
    if (mode == 3 && !quantm && N <= kEoWaveMaxN && !(no_wave && no_wave[0] == '1')) {
        // extremal_opt: one wavefront per replica, the ranking in LDS
        const size_t lds = eo_wave_lds_bytes((int)N);
        HIP_TRY(ctx, raise_lds_attr(reinterpret_cast<const void*>(eo_cont_wave_kernel), lds));
        hipLaunchKernelGGL(eo_cont_wave_kernel, dim3((unsigned)R), dim3(64), lds, st, P);
    } else if (const char* nw = std::getenv("RRRMC_CONT_NO_WAVE"); (mode == 0 || mode == 1) && !dblm && !ctx->pf_multi_edge && N >= 64 && K <= kContKmax &&
               cw_lds_bytes(W, levs) <= (size_t)32768 && (S.resume ? ctx->smp_build == 1 : !(nw && nw[0] == '1'))) {
        // rrrMC / bklMC on the pure Float64 models: one wavefront per replica, the top of the sampler's tree in LDS (cont_wave_kernel.hpp)
        // (the two builds keep the tree in different layouts: a resumed call runs the build that holds the run's state)
        ctx->smp_build = 1;
        hipLaunchKernelGGL(cont_wave_kernel, dim3((unsigned)R), dim3(64), cw_lds_bytes(W, levs), st, P);
    } else {
        if (N > 65535) hipLaunchKernelGGL(cont_sparse_kernel<uint32_t>, dim3(rrr_blocks(R)), dim3(rrr_tpb(R)), 0, st, P);
        else hipLaunchKernelGGL(cont_sparse_kernel<uint16_t>, dim3(rrr_blocks(R)), dim3(rrr_tpb(R)), 0, st, P);
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[1], st));
    if (!dblm) {
        hipLaunchKernelGGL(cont_spins_out_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)ctx->pfW), dim3(256), 0, st, ctx->cs_spins, ctx->pf_spins, (int)N, (int)W, (int)R);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev_end, st));
    if (mode == 2) { if (!S.resume) ctx->wtm_calls += 1; } else ctx->it_done += (uint64_t)iters;
    smp_commit(ctx, mode + 1, iters);
    ctx->pf_lf_live = false;
    ctx->db_cache_valid = false;
    if (quantm) ctx->q_cache_valid = false;
    ctx->stats_stride = 3;
    ctx->sweep_launches = 1;
    ctx->nsamp = nsamp;
    ctx->results_valid = true;
    ctx->timing_valid = true;
    ctx->last_call_rrr = true;          // accepted / moves come from q_stats
    ctx->last_call_wtm = mode == 2;
    ctx->last_call_eo = mode == 3;          // Emin in wt_time, itmin in q_stats[.][1], Cmin in eo_cmin
    return RRRMC_OK;
}
