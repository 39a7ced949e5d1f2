// Host side of the discretised DoubleGraphs (Graph{RRG,EA}NormalDiscretized) under rrrMC.
// Included by rrrmc_hip.hip inside its anonymous namespace, after the context struct and the common helpers
// (fail, HIP_TRY, free_dev, ensure_state); not a stand-alone translation unit.
// ---- DoubleGraphs Graph{RRG,EA}NormalDiscretized host side -----------------------------------------------------------
double host_det_exp(double x);
int32_t dbl_ctx_create(rrrmc_ctx** out, int64_t N, int64_t K, int64_t R, int32_t device, uint32_t replica0)
{
    if (N < 1 || K < 1 || R < 1) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "N, K, R must be >= 1 (given N=%lld K=%lld R=%lld)", (long long)N, (long long)K, (long long)R);
    if (K > kDKmax) return fail(nullptr, RRRMC_ERR_UNSUPPORTED, "K=%lld: the DoubleGraph kernel covers K <= %d", (long long)K, kDKmax);
    if (N > (int64_t)1 << 28) return fail(nullptr, RRRMC_ERR_UNSUPPORTED, "N=%lld is beyond the DoubleGraph kernels (N <= 2^28)", (long long)N);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, RRRMC_ERR_HIP, "no HIP device is visible: this library has no CPU path");
    if (device < 0 || device >= ndev) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "device %d out of range (0..%d)", device, ndev - 1);
    rrrmc_ctx* ctx = new (std::nothrow) rrrmc_ctx();
    if (!ctx) return fail(nullptr, RRRMC_ERR_NOMEM, "out of host memory");
    ctx->model = RRRMC_MODEL_SPARSE_DISCRETIZED; ctx->N = N; ctx->K = K; ctx->R = R; ctx->Rpad = R;
    ctx->qW = 2 * ((N + 63) / 64);
    ctx->device = device; ctx->replica0 = replica0;
#define DB_TRY(expr)                                                                                             \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess) {                                                                                  \
            int32_t rc_ = fail(nullptr, e_ == hipErrorOutOfMemory ? RRRMC_ERR_NOMEM : RRRMC_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
            rrrmc_ctx_destroy(ctx);                                                                              \
            return rc_;                                                                                          \
        }                                                                                                        \
    } while (0)
    DB_TRY(hipSetDevice(device));
    DB_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    DB_TRY(hipEventCreate(&ctx->ev_begin));
    DB_TRY(hipEventCreate(&ctx->ev_end));
    DB_TRY(hipMalloc(&ctx->d_A, sizeof(int32_t) * N * K));
    DB_TRY(hipMalloc(&ctx->db_dJ, sizeof(int8_t) * N * K));
    DB_TRY(hipMalloc(&ctx->db_rJ, sizeof(double) * N * K));
    DB_TRY(hipMalloc(&ctx->q_spins, sizeof(uint32_t) * R * ctx->qW));
    DB_TRY(hipMalloc(&ctx->db_cls, (size_t)R * N));
    const size_t idx_bytes = N > 65535 ? sizeof(uint32_t) : sizeof(uint16_t);          // set members / positions: 32-bit beyond 65 535 spins
    DB_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->db_sv), idx_bytes * (size_t)R * 2 * kDLmax * N));
    DB_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->db_spos), idx_bytes * (size_t)R * N));
    DB_TRY(hipMalloc(&ctx->db_lf, sizeof(double) * (size_t)R * N));
    DB_TRY(hipMalloc(&ctx->db_undo, sizeof(double) * (size_t)R * (K + 1)));
    DB_TRY(hipMalloc(&ctx->q_stats, sizeof(int64_t) * R * 3));          // rrrMC / standardMC use a stride of 2, bklMC / wtmMC of 3
    DB_TRY(hipMalloc(&ctx->sk_E, sizeof(double) * R));
    DB_TRY(hipMalloc(&ctx->d_acc, sizeof(int64_t) * R));
    DB_TRY(hipMemset(ctx->q_spins, 0, sizeof(uint32_t) * R * ctx->qW));
    DB_TRY(hipMemset(ctx->q_stats, 0, sizeof(int64_t) * R * 3));
#undef DB_TRY
    *out = ctx;
    return RRRMC_OK;
}

RrrDblParams dbl_params(rrrmc_ctx* ctx, double beta)
{
    RrrDblParams P{};
    P.A = ctx->d_A; P.dJ = ctx->db_dJ; P.rJ = ctx->db_rJ; P.spins = ctx->q_spins; P.cls = ctx->db_cls; P.sv = ctx->db_sv; P.spos = ctx->db_spos;
    P.lf = ctx->db_lf; P.undo = ctx->db_undo; P.E_cur = ctx->sk_E; P.stats = ctx->q_stats; P.Es = ctx->sk_Es;
    P.lev_mul = ctx->db_lev_mul; P.lev_div = ctx->db_lev_div;
    for (int k = 0; k < ctx->db_L; ++k) { P.dElist[k] = ctx->db_dElist[k]; P.ft[k] = host_det_exp(-beta * P.to_f64(ctx->db_dElist[k])); }   // DeltaE.jl:91
    P.beta = beta;
    P.k0 = (uint32_t)ctx->seed; P.k1 = (uint32_t)(ctx->seed >> 32); P.replica0 = ctx->replica0;
    P.N = (int)ctx->N; P.K = (int)ctx->K; P.L = ctx->db_L; P.W = (int)ctx->qW; P.R = (int)ctx->R; P.ea_form = ctx->db_ea_form;
    return P;
}

int32_t dbl_run_energy(rrrmc_ctx* ctx)
{
    RrrDblParams P = dbl_params(ctx, 0.0);
    P.energy_only = 1;
    hipLaunchKernelGGL((rrr_dbl_kernel<4, uint16_t>), dim3(rrr_blocks(ctx->R)), dim3(rrr_tpb(ctx->R)), 0, ctx->stream, P);          // energy only: the level bound and the index width are irrelevant
    HIP_TRY(ctx, hipGetLastError());
    return RRRMC_OK;
}

// standard = false: rrrMC(X::DoubleGraph); true: standardMC on the same graph (staged_thr* unused)
int32_t dbl_mc_async(rrrmc_ctx* ctx, bool standard, double beta, int64_t iters, int64_t step, double staged_thr, double staged_thr_fact)
{
    if (iters < 0) return fail(ctx, RRRMC_ERR_INVALID_ARG, "iters must be >= 0, given %lld", (long long)iters);
    if (step < 1) return fail(ctx, RRRMC_ERR_INVALID_ARG, "step must be >= 1, given %lld", (long long)step);
    if (std::isnan(beta)) return fail(ctx, RRRMC_ERR_INVALID_ARG, "beta is NaN");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->results_valid = false; ctx->last_call_wtm = false; ctx->last_call_eo = false;
    ctx->timing_valid = false;
    SmpState S{};
    if (!standard) { const int32_t rcs = smp_begin(ctx, 1, beta, staged_thr, staged_thr_fact, 0.0, step, nullptr, &S); if (rcs) return rcs; }
    else S.samp0 = step;
    const int64_t nsamp = standard ? iters / step : smp_nsamp(ctx, iters, step);
    const size_t es_need = (size_t)(nsamp > 0 ? nsamp : 1) * ctx->R;
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
    RrrDblParams P = dbl_params(ctx, beta);
    P.S = S;
    P.staged_thr = staged_thr; P.lambda = staged_thr_fact / (double)ctx->N;
    P.g0 = ctx->it_done; P.iters = iters; P.step = step;
    if (!ctx->db_mlast) HIP_TRY(ctx, hipMalloc(&ctx->db_mlast, sizeof(int32_t) * (size_t)ctx->R));
    P.mlast = ctx->db_mlast;
    P.resume = (standard && ctx->resume && ctx->std_cache_live) ? 1 : 0;      // continue from the tracked energy / residual cache
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_begin, st));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[0], st));
    if (standard) hipLaunchKernelGGL(dbl_standard_kernel, dim3(rrr_blocks(ctx->R)), dim3(rrr_tpb(ctx->R)), 0, st, P);
    else if (ctx->N > 65535) {
        if (ctx->db_L <= 4) hipLaunchKernelGGL((rrr_dbl_kernel<4, uint32_t>), dim3(rrr_blocks(ctx->R)), dim3(rrr_tpb(ctx->R)), 0, st, P);
        else hipLaunchKernelGGL((rrr_dbl_kernel<8, uint32_t>), dim3(rrr_blocks(ctx->R)), dim3(rrr_tpb(ctx->R)), 0, st, P);
    }
    else if (ctx->db_L <= 4) hipLaunchKernelGGL((rrr_dbl_kernel<4, uint16_t>), dim3(rrr_blocks(ctx->R)), dim3(rrr_tpb(ctx->R)), 0, st, P);   // per-class arrays sized at compile time: registers
    else hipLaunchKernelGGL((rrr_dbl_kernel<8, uint16_t>), dim3(rrr_blocks(ctx->R)), dim3(rrr_tpb(ctx->R)), 0, st, P);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[1], st));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_end, st));
    ctx->stats_stride = 2;
    ctx->sweep_launches = 1;
    ctx->nsamp = nsamp;
    ctx->it_done += (uint64_t)iters;
    if (!standard) smp_commit(ctx, 1, iters);
    ctx->results_valid = true;
    ctx->timing_valid = true;
    ctx->last_call_rrr = true;          // accepted counts live in q_stats
    ctx->db_cache_valid = !standard;    // the class sets exist only after rrrMC
    ctx->std_cache_live = standard;
    return RRRMC_OK;
}
int32_t dbl_rrr_mc_async(rrrmc_ctx* ctx, double beta, int64_t iters, int64_t step, double staged_thr, double staged_thr_fact)
{
    return dbl_mc_async(ctx, false, beta, iters, step, staged_thr, staged_thr_fact);
}
