// Host side of RRRMC_MODEL_SPARSE_LEVELS: stand-alone GraphRRG{ET,LEV,K} / GraphEA{ET,LEV,2D} with levels other than (-1, 1),
// ET = Int or DFloat64 (src/graphs/RRG.jl:116-162, src/graphs/EA.jl:138-193).
// Included by rrrmc_hip.hip inside its anonymous namespace, after the context struct and the common helpers; not a stand-alone
// translation unit.  Spins live in q_spins / qW (R x W 32-bit words, BitVector order) — the layout the thread-per-replica kernels of
// rrr_kernels.hpp work on, so rrrMC / bklMC / wtmMC / extremal_opt run without the bit-sliced <-> replica-contiguous conversion of
// the +-J contexts.  Energies are integer level units (int32 on the device); rrrmc_set_level_scale gives their Float64 value.
int32_t lev_ctx_create(rrrmc_ctx** out, int64_t N, int64_t K, int64_t R, int32_t device, uint32_t replica0)
{
    if (N < 1 || K < 1 || R < 1) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "N, K, R must be >= 1 (given N=%lld K=%lld R=%lld)", (long long)N, (long long)K, (long long)R);
    if (K > 16) return fail(nullptr, RRRMC_ERR_UNSUPPORTED, "K=%lld: the general-level kernels cover K <= 16", (long long)K);
    if (N > (int64_t)1 << 28) return fail(nullptr, RRRMC_ERR_UNSUPPORTED, "N=%lld is beyond the thread-per-replica kernels (N <= 2^28)", (long long)N);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, RRRMC_ERR_HIP, "no HIP device is visible: this library has no CPU path");
    if (device < 0 || device >= ndev) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "device %d out of range (0..%d)", device, ndev - 1);
    rrrmc_ctx* ctx = new (std::nothrow) rrrmc_ctx();
    if (!ctx) return fail(nullptr, RRRMC_ERR_NOMEM, "out of host memory");
    ctx->model = RRRMC_MODEL_SPARSE_LEVELS; ctx->N = N; ctx->K = K; ctx->R = R;
    ctx->G = (R + 31) / 32; ctx->Rpad = 32 * ctx->G;
    ctx->qW = 2 * ((N + 63) / 64);
    ctx->device = device; ctx->replica0 = replica0;
#define LV_TRY(expr)                                                                                             \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess) {                                                                                  \
            int32_t rc_ = fail(nullptr, e_ == hipErrorOutOfMemory ? RRRMC_ERR_NOMEM : RRRMC_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
            rrrmc_ctx_destroy(ctx);                                                                              \
            return rc_;                                                                                          \
        }                                                                                                        \
    } while (0)
    LV_TRY(hipSetDevice(device));
    LV_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    LV_TRY(hipEventCreate(&ctx->ev_begin));
    LV_TRY(hipEventCreate(&ctx->ev_end));
    LV_TRY(hipMalloc(&ctx->d_A, sizeof(int32_t) * N * K));
    LV_TRY(hipMalloc(&ctx->d_J, sizeof(int8_t) * N * K));
    LV_TRY(hipMalloc(&ctx->q_spins, sizeof(uint32_t) * R * ctx->qW));
    LV_TRY(hipMalloc(&ctx->d_E, sizeof(int32_t) * ctx->Rpad));
    LV_TRY(hipMalloc(&ctx->d_acc, sizeof(int64_t) * ctx->Rpad));
    LV_TRY(hipMemset(ctx->q_spins, 0, sizeof(uint32_t) * R * ctx->qW));
    LV_TRY(hipMemset(ctx->d_E, 0, sizeof(int32_t) * ctx->Rpad));
    LV_TRY(hipMemset(ctx->d_acc, 0, sizeof(int64_t) * ctx->Rpad));
#undef LV_TRY
    *out = ctx;
    return RRRMC_OK;
}

// standardMC; iters = 0 with `energy_only` leaves energy(X, C) in d_E without touching the result bookkeeping
int32_t lev_standard_mc_async(rrrmc_ctx* ctx, double beta, int64_t iters, int64_t step, bool energy_only)
{
    if (iters < 0) return fail(ctx, RRRMC_ERR_INVALID_ARG, "iters must be >= 0, given %lld", (long long)iters);
    if (step < 1) return fail(ctx, RRRMC_ERR_INVALID_ARG, "step must be >= 1, given %lld", (long long)step);
    if (std::isnan(beta)) return fail(ctx, RRRMC_ERR_INVALID_ARG, "beta is NaN");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t nsamp = iters / step;
    if (!energy_only) {
        ctx->results_valid = false; ctx->last_call_wtm = false; ctx->last_call_eo = false;
        ctx->timing_valid = false;
        const size_t es_need = (size_t)(nsamp > 0 ? nsamp : 1) * ctx->Rpad;
        if (es_need > ctx->Es_cap) {
            free_dev(ctx->d_Es);
            ctx->Es_cap = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->d_Es, sizeof(int32_t) * es_need));
            ctx->Es_cap = es_need;
        }
        while (ctx->ev_sweep.size() < 2) {
            hipEvent_t e;
            HIP_TRY(ctx, hipEventCreate(&e));
            ctx->ev_sweep.push_back(e);
        }
    }
    LevStdParams P{};
    P.A = ctx->d_A; P.J = ctx->d_J; P.spins = ctx->q_spins; P.E_cur = ctx->d_E; P.acc_cur = ctx->d_acc; P.Es = ctx->d_Es;
    P.beta = beta; P.lev_mul = ctx->lv_mul; P.lev_div = ctx->lv_div;
    P.g0 = ctx->it_done; P.iters = energy_only ? 0 : iters; P.step = step;
    P.k0 = (uint32_t)ctx->seed; P.k1 = (uint32_t)(ctx->seed >> 32); P.replica0 = ctx->replica0;
    P.N = (int)ctx->N; P.K = (int)ctx->K; P.W = (int)ctx->qW; P.R = (int)ctx->R; P.Rpad = (int)ctx->Rpad;
    hipStream_t st = ctx->stream;
    if (!energy_only) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_begin, st));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[0], st));
    }
    hipLaunchKernelGGL(lev_standard_kernel, dim3(rrr_blocks(ctx->R)), dim3(rrr_tpb(ctx->R)), 0, st, P);
    HIP_TRY(ctx, hipGetLastError());
    if (energy_only) return RRRMC_OK;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[1], st));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_end, st));
    ctx->sweep_launches = 1;
    ctx->nsamp = nsamp;
    ctx->it_done += (uint64_t)iters;
    ctx->results_valid = true;
    ctx->timing_valid = true;
    ctx->last_call_rrr = false;
    ctx->colored_call = false;
    return RRRMC_OK;
}
