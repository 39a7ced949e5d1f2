// Host side of the reduced-rejection samplers: GraphQuant, rrrMC on GraphSKNormal, rrrMC / bklMC on GraphRRG / GraphEA.
// Included by rrrmc_hip.hip inside its anonymous namespace, after the context struct and the common helpers
// (fail, HIP_TRY, free_dev, ensure_state); not a stand-alone translation unit.
// ---- GraphQuant / rrrMC host side -----------------------------------------------------------------------------------
RrrParams quant_params(rrrmc_ctx* ctx, double beta, double fourK)
{
    RrrParams P{};
    P.A = ctx->d_A; P.J = ctx->d_J;
    if (ctx->q_sk) { P.Jb = ctx->q_Jb; P.Wk = (int)ctx->q_Wk; P.sN = std::sqrt((double)ctx->qNk); }
    if (ctx->q_skn) { P.Jd = ctx->sk_J; P.slf = ctx->q_slf; P.smv = ctx->q_smv; P.scur = ctx->q_scur; }
    if (ctx->q_spf) { P.Jf = ctx->q_Jf; P.flf = ctx->q_flf; P.fundo = ctx->q_fundo; P.fml = ctx->q_fml; }
    P.spins = ctx->q_spins; P.cls = ctx->q_cls; P.sv = ctx->q_sv; P.spos = ctx->q_spos; P.st = ctx->q_st;
    P.T = ctx->q_T; P.zz = ctx->q_z; P.E_cur = ctx->sk_E; P.acc_rate = ctx->q_accrate; P.stats = ctx->q_stats; P.Es = ctx->sk_Es;
    P.beta = beta; P.fourK = fourK;
    P.ft1 = 0.0;
    P.k0 = (uint32_t)ctx->seed; P.k1 = (uint32_t)(ctx->seed >> 32); P.replica0 = ctx->replica0;
    P.Nk = (int)ctx->qNk; P.M = (int)ctx->qM; P.K = (int)ctx->K; P.N = (int)ctx->N; P.W = (int)ctx->qW; P.R = (int)ctx->R;
    P.wide = ctx->N > 65535 ? 1 : 0;
    return P;
}

// deterministic exp on the host: the same operation sequence as det_exp on the device / orc_det_exp in the oracle
double host_det_exp(double x)
{
    static const double LOG2E = 1.44269504088896338700e+00;
    static const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    static const double c[14] = {1.0, 1.0, 0.5, 1.0 / 6, 1.0 / 24, 1.0 / 120, 1.0 / 720, 1.0 / 5040, 1.0 / 40320, 1.0 / 362880,
                                 1.0 / 3628800, 1.0 / 39916800, 1.0 / 479001600, 1.0 / 6227020800.0};
    if (x != x) return x;
    if (x < -745.2) return 0.0;
    if (x > 709.7) return HUGE_VAL;
    volatile double t0 = x * LOG2E;
    const double k = std::floor(t0 + 0.5);
    volatile double a = k * LN2_HI, b = k * LN2_LO;
    volatile double r0 = x - a;
    const double r = r0 - b;
    double p = c[13];
    for (int n = 12; n >= 0; --n) { volatile double m = p * r; p = m + c[n]; }
    return std::ldexp(p, (int)k);
}

// energy(X, C) + gen_ΔEcache (RRRMC.jl:237-240): E into sk_E, cache arrays rebuilt
int32_t quant_run_init(rrrmc_ctx* ctx, double beta, double fourK)
{
    RrrParams P = quant_params(ctx, beta, fourK);
    P.ft1 = host_det_exp(-beta * fourK);
    if (ctx->q_skn) {              // GraphSKNormal slices: Float64 slice caches rebuilt in the reference's summation order
        if ((size_t)ctx->qM * sizeof(long long) > 32768) return fail(ctx, RRRMC_ERR_UNSUPPORTED, "M = %lld slices exceed the init kernel's LDS", (long long)ctx->qM);
        hipLaunchKernelGGL(rrr_init_skn_kernel, dim3((unsigned)ctx->R), dim3(kInitThreads), (size_t)ctx->qM * sizeof(long long), ctx->stream, P);
        HIP_TRY(ctx, hipGetLastError());
        return RRRMC_OK;
    }
    if (ctx->q_spf) {              // sparse Float64 slices (GraphQEAT): every slice's LocalFields rebuilt in its graph's summation order
        if ((size_t)ctx->qM * sizeof(long long) > 32768) return fail(ctx, RRRMC_ERR_UNSUPPORTED, "M = %lld slices exceed the init kernel's LDS", (long long)ctx->qM);
        hipLaunchKernelGGL(rrr_init_spf_kernel, dim3((unsigned)ctx->R), dim3(kInitThreads), (size_t)ctx->qM * sizeof(long long), ctx->stream, P);
        HIP_TRY(ctx, hipGetLastError());
        return RRRMC_OK;
    }
    // one workgroup per replica (a parallel construction of the same cache); the thread-per-replica form only for absurd M
    if ((size_t)ctx->qM * sizeof(long long) <= 32768)
        hipLaunchKernelGGL(rrr_init_coop_kernel, dim3((unsigned)ctx->R), dim3(kInitThreads), (size_t)ctx->qM * sizeof(long long), ctx->stream, P);
    else
        hipLaunchKernelGGL(rrr_init_kernel, dim3(rrr_blocks(ctx->R)), dim3(rrr_tpb(ctx->R)), 0, ctx->stream, P);
    HIP_TRY(ctx, hipGetLastError());
    return RRRMC_OK;
}


// rrrMC(X::SingleGraph) on GraphSKNormal (RRRMC.jl:149-219): thread-per-replica kernel over interleaved arrays
// mode 0 = rrrMC(SingleGraph), 1 = bklMC, 2 = wtmMC (iters = samples, stepf = step in sweeps)
// mode 0 rrrMC, 1 bklMC, 2 wtmMC (iters = samples), 3 extremal_opt (ftau = its rank table; beta unused)
int32_t sk_rrr_mc_async(rrrmc_ctx* ctx, double beta, int64_t iters, int64_t step, double staged_thr, double staged_thr_fact, int mode = 0, double stepf = 1.0,
                        const double* ftau = nullptr)
{
    if (iters < 0) return fail(ctx, RRRMC_ERR_INVALID_ARG, "iters must be >= 0, given %lld", (long long)iters);
    if (step < 1) return fail(ctx, RRRMC_ERR_INVALID_ARG, "step must be >= 1, given %lld", (long long)step);
    if (mode == 2 && (!(stepf > 0.0) || !std::isfinite(stepf))) return fail(ctx, RRRMC_ERR_INVALID_ARG, "step must be positive and finite, given %g", stepf);
    if (mode == 2 && ctx->N > 65535) return fail(ctx, RRRMC_ERR_UNSUPPORTED, "N=%lld: the wtmMC heap indexes spins with 16 bits", (long long)ctx->N);
    if (mode == 3) {
        if (ctx->N > kEoSkMaxN) return fail(ctx, RRRMC_ERR_UNSUPPORTED, "N=%lld: extremal_opt on the dense SK models keeps the ranking in LDS, N <= %d", (long long)ctx->N, kEoSkMaxN);
        if (!ftau) return fail(ctx, RRRMC_ERR_INVALID_ARG, "ftau is NULL");
        for (int64_t i = 0; i < ctx->N; ++i)
            if (!(ftau[i] > 0.0) || !std::isfinite(ftau[i]) || (i && ftau[i] < ftau[i - 1]))
                return fail(ctx, RRRMC_ERR_INVALID_ARG, "ftau must be a positive non-decreasing table (cumsum of j^-tau), violated at %lld", (long long)i);
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->results_valid = false; ctx->last_call_wtm = false; ctx->last_call_eo = false;
    ctx->timing_valid = false;
    SmpState S{};
    { const int32_t rcs = smp_begin(ctx, mode + 1, beta, mode == 2 ? stepf : staged_thr, staged_thr_fact, 0.0, mode == 2 ? 1 : step, ftau, &S); if (rcs) return rcs; }
    const int64_t N = ctx->N, Rp = ctx->Rpad;
    int levs = 0;
    while (((int64_t)1 << levs) < N) ++levs;
    const int64_t N2 = (int64_t)1 << levs, W = (N + 31) / 32;
    const size_t per = (size_t)Rp;
    const size_t ndbl = per * (size_t)(5 * N + 2 * N2 + 1);
    if (!ctx->rs_buf) {
        HIP_TRY(ctx, hipMalloc(&ctx->rs_buf, sizeof(double) * ndbl));
        HIP_TRY(ctx, hipMalloc(&ctx->rs_spins, sizeof(uint32_t) * W * per));
        HIP_TRY(ctx, hipMalloc(&ctx->rs_status, sizeof(int32_t) * per));
        HIP_TRY(ctx, hipMalloc(&ctx->q_stats, sizeof(int64_t) * per * 2));
    }
    if ((mode == 2 || mode == 3) && !ctx->wt_time) HIP_TRY(ctx, hipMalloc(&ctx->wt_time, sizeof(double) * per));
    if (mode == 3) {
        if (!ctx->eo_cmin) {
            HIP_TRY(ctx, hipMalloc(&ctx->eo_cmin, sizeof(uint32_t) * ctx->R * W));
            HIP_TRY(ctx, hipMalloc(&ctx->eo_ftau, sizeof(double) * N));
        }
        ctx->eo_W = W;
        if (!S.resume) {                    // (a resumed call: the run's table is on the device already)
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            HIP_TRY(ctx, hipMemcpy(ctx->eo_ftau, ftau, sizeof(double) * N, hipMemcpyHostToDevice));
        }
    }
    const int64_t nsamp = mode == 2 ? iters : smp_nsamp(ctx, iters, step);
    const size_t es_need = (size_t)(nsamp > 0 ? nsamp : 1) * Rp;
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
    hipStream_t st = ctx->stream;
    // The binary GraphSK is a SimpleGraph{Float64} too (SK.jl:28): the same continuous-energy caches, over delta_energy = lfields[i] / sN
    // with integer fields — i.e. this kernel on the +-1 matrix (every field and the energy's n stay exact integers in Float64) with
    // the division by sN = sqrt(N) applied where the reference applies it (SK.jl:95,139).
    const bool bin = ctx->model == RRRMC_MODEL_SK_BINARY;
    if (bin) {
        if (!ctx->sk_J) HIP_TRY(ctx, hipMalloc(&ctx->sk_J, sizeof(double) * N * N));
        hipLaunchKernelGGL(skb_dense_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)N), dim3(256), 0, st, ctx->skb_J, ctx->sk_J, (int)N, (int)ctx->skb_NW);
        HIP_TRY(ctx, hipGetLastError());
    }
    RrrSkParams P{};
    double* b = ctx->rs_buf;
    P.J = ctx->sk_J;
    P.sN = bin ? std::sqrt((double)N) : 0.0;
    P.lfA = b; b += per * N;
    P.lfB = b; b += per * N;
    P.v = b; b += per * N2;
    P.ps = b; b += per * N2;
    P.dEs = b; b += per * N;
    P.st_dE = b; b += per * N;
    P.st_p = b; b += per * N;
    P.z_out = b;
    P.spins = ctx->rs_spins; P.E_cur = ctx->sk_E; P.stats = ctx->q_stats; P.status = ctx->rs_status; P.Es = ctx->sk_Es;
    P.beta = beta; P.staged_thr = staged_thr; P.lambda = staged_thr_fact / (double)N;
    P.g0 = mode == 1 ? ctx->smp_g0 : ctx->it_done;          // bklMC numbers its moves from the start of the RUN
    P.iters = iters; P.step = step;
    P.k0 = (uint32_t)ctx->seed; P.k1 = (uint32_t)(ctx->seed >> 32); P.replica0 = ctx->replica0;
    P.N = (int)N; P.N2 = (int)N2; P.levs = levs; P.W = (int)W; P.R = (int)ctx->R; P.Rp = (int)Rp;
    P.mode = mode;
    P.S = S; P.samples_before = mode == 2 ? ctx->smp_it : 0;
    P.call = ctx->smp_call & 0xffffffu; P.stepf = stepf; P.t_out = ctx->wt_time;
    P.ftau = ctx->eo_ftau; P.cmin = ctx->eo_cmin;
    ctx->stats_stride = 2;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_begin, st));
    hipLaunchKernelGGL(rrsk_spins_in_kernel, dim3((unsigned)((Rp + 255) / 256), (unsigned)W), dim3(256), 0, st, ctx->sk_spins, ctx->rs_spins, (int)N, (int)W, (int)Rp);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[0], st));
    if (mode == 3) {
        const size_t lds = eo_sk_lds_bytes((int)N, (int)N2, (int)W);
        HIP_TRY(ctx, raise_lds_attr(reinterpret_cast<const void*>(eo_sk_wave_kernel), lds));
        hipLaunchKernelGGL(eo_sk_wave_kernel, dim3((unsigned)ctx->R), dim3(64), lds, st, P);
    } else {
        hipLaunchKernelGGL(rrr_skn_kernel, dim3(rrr_blocks(ctx->R)), dim3(rrr_tpb(ctx->R)), 0, st, P);
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[1], st));
    hipLaunchKernelGGL(rrsk_spins_out_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)ctx->G8), dim3(256), 0, st, ctx->rs_spins, ctx->sk_spins, (int)N, (int)Rp);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_end, st));
    ctx->sweep_launches = 1;
    ctx->nsamp = nsamp;
    if (mode == 2) { if (!S.resume) ctx->wtm_calls += 1; } else ctx->it_done += (uint64_t)iters;
    smp_commit(ctx, mode + 1, iters);
    ctx->results_valid = true;
    ctx->timing_valid = true;
    ctx->last_call_rrr = true;
    ctx->last_call_wtm = mode == 2;
    ctx->last_call_eo = mode == 3;          // Emin in wt_time, itmin in q_stats[.][1], Cmin in eo_cmin
    return RRRMC_OK;
}


// The thread-per-replica kernels work on replica-contiguous spin words: the +-J contexts convert from / to their bit-sliced store
// around the kernel, the general-level contexts (RRRMC_MODEL_SPARSE_LEVELS) keep that layout natively (q_spins, qW).
struct RpView { uint32_t* spins; int64_t W; bool convert; };
int32_t rp_prepare(rrrmc_ctx* ctx, RpView* v)
{
    if (ctx->model == RRRMC_MODEL_SPARSE_LEVELS) { v->spins = ctx->q_spins; v->W = ctx->qW; v->convert = false; return RRRMC_OK; }
    v->W = (ctx->N + 31) / 32;
    if (!ctx->rp_spins) HIP_TRY(ctx, hipMalloc(&ctx->rp_spins, sizeof(uint32_t) * ctx->R * v->W));
    v->spins = ctx->rp_spins; v->convert = true;
    return RRRMC_OK;
}
int32_t rp_in(rrrmc_ctx* ctx, const RpView& v, hipStream_t st)
{
    if (!v.convert) return RRRMC_OK;
    hipLaunchKernelGGL(rrsp_spins_in_kernel, dim3((unsigned)((v.W + 255) / 256), (unsigned)ctx->R), dim3(256), 0, st, ctx->d_spins, ctx->rp_spins, (int)ctx->N, (int)v.W, (int)ctx->R);
    HIP_TRY(ctx, hipGetLastError());
    return RRRMC_OK;
}
int32_t rp_out(rrrmc_ctx* ctx, const RpView& v, hipStream_t st)
{
    if (!v.convert) return RRRMC_OK;
    hipLaunchKernelGGL(rrsp_spins_out_kernel, dim3((unsigned)((ctx->N + 255) / 256), (unsigned)ctx->G), dim3(256), 0, st, ctx->rp_spins, ctx->d_spins, (int)ctx->N, (int)v.W, (int)ctx->R);
    HIP_TRY(ctx, hipGetLastError());
    return RRRMC_OK;
}

// rrrMC(SingleGraph) / bklMC on GraphRRG / GraphEA: thread-per-replica kernel over replica-contiguous arrays
int32_t sparse_rrr_bkl_async(rrrmc_ctx* ctx, int mode, double beta, int64_t iters, int64_t step, double staged_thr, double staged_thr_fact)
{
    if (iters < 0) return fail(ctx, RRRMC_ERR_INVALID_ARG, "iters must be >= 0, given %lld", (long long)iters);
    if (step < 1) return fail(ctx, RRRMC_ERR_INVALID_ARG, "step must be >= 1, given %lld", (long long)step);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->results_valid = false; ctx->last_call_wtm = false; ctx->last_call_eo = false;
    ctx->timing_valid = false;
    const int64_t N = ctx->N, K = ctx->K, R = ctx->R;
    const int L = ctx->lv.L;
    // set members / positions are 16-bit spin ids up to N = 65 535, 32-bit beyond (GraphEA(64, 3): N = 262 144)
    const bool wide_idx = N > 65535;
    const size_t idx_bytes = wide_idx ? 4 : 2;
    if (N > (int64_t)1 << 28) return fail(ctx, RRRMC_ERR_UNSUPPORTED, "N=%lld is beyond the rrrMC kernel (N <= 2^28)", (long long)N);
    SmpState S{};
    { const int32_t rcs = smp_begin(ctx, mode + 1, beta, staged_thr, staged_thr_fact, 0.0, step, nullptr, &S); if (rcs) return rcs; }
    RpView rv;
    int32_t rc = rp_prepare(ctx, &rv);
    if (rc) return rc;
    const int64_t W = rv.W;
    if (!ctx->rp_cls) {
        HIP_TRY(ctx, hipMalloc(&ctx->rp_cls, (size_t)R * N));
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->rp_sv), idx_bytes * R * 2 * L * N));
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->rp_spos), idx_bytes * R * N));
        if (!ctx->q_stats) HIP_TRY(ctx, hipMalloc(&ctx->q_stats, sizeof(int64_t) * R * 3));
    }
    ctx->stats_stride = 3;
    const int64_t nsamp = smp_nsamp(ctx, iters, step);
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
    RrrSparseParams P{};
    P.S = S;
    P.A = ctx->d_A; P.J = ctx->d_J; P.spins = rv.spins; P.cls = ctx->rp_cls; P.sv = ctx->rp_sv; P.spos = ctx->rp_spos;
    P.E_cur = ctx->d_E; P.acc_cur = ctx->d_acc; P.stats = ctx->q_stats; P.Es = ctx->d_Es;
    P.lv = ctx->lv;
    for (int k = 0; k < L && k < kSLmax; ++k) P.ft[k] = host_det_exp(-beta * lv_to_f64(ctx, ctx->lv.dElist[k]));     // exp(-beta dE_k), DeltaE.jl:91
    P.beta = beta; P.staged_thr = staged_thr; P.lambda = staged_thr_fact / (double)N;
    P.g0 = mode == 1 ? ctx->smp_g0 : ctx->it_done;          // bklMC numbers its moves from the start of the RUN
    P.iters = iters; P.step = step;
    P.k0 = (uint32_t)ctx->seed; P.k1 = (uint32_t)(ctx->seed >> 32); P.replica0 = ctx->replica0;
    P.N = (int)N; P.K = (int)K; P.L = L; P.W = (int)W; P.R = (int)R; P.Rpad = (int)ctx->Rpad; P.mode = mode;
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_begin, st));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_acc, 0, sizeof(int64_t) * ctx->Rpad, st));
    if ((rc = rp_in(ctx, rv, st))) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[0], st));
    // few replicas: one WAVEFRONT per replica with the whole DeltaECache in LDS (sparse_wave_kernel.hpp)
    const SwLayout swl = sw_layout(N, W, K, L, (size_t)kLdsLimit);
    const char* no_wave = std::getenv("RRRMC_RRR_NO_WAVE");              // tests / timing experiments
    // a replica's workgroup owns its LDS: as many replicas at a time as the CUs hold (256 CUs x workgroups per CU); beyond that the
    // thread-per-replica kernel fills the chip better
    int64_t wave_max_R = 256 * std::max<int64_t>(1, (int64_t)(kLdsLimit / std::max<size_t>(swl.bytes, 1)));
    if (const char* e = std::getenv("RRRMC_RRR_WAVE_MAX_R")) wave_max_R = std::atoll(e);
    int64_t sw_cap = swl.cap;
    if (const char* e = std::getenv("RRRMC_RRR_WAVE_SLACK")) {          // tests: a small slack makes the segments re-space often
        const int64_t want = N + std::atoll(e);
        if (want >= N + 2 * L * sw_min_gap(K) && want < sw_cap) sw_cap = want & ~(int64_t)3;
    }
    const bool wave_ok = !wide_idx && K <= 7 && L <= 4 && R <= wave_max_R && sw_cap >= N + 2 * L * (int64_t)sw_min_gap(K) &&
                         !(no_wave && no_wave[0] == '1');
    if (wave_ok) {
        SwExtra X{};
        X.cap = (int)sw_cap; X.off_spos = (uint32_t)swl.off_spos; X.off_sv = (uint32_t)swl.off_sv; X.off_A = (uint32_t)swl.off_A;
        X.off_J = (uint32_t)swl.off_J; X.off_rng = (uint32_t)swl.off_rng; X.off_tab = (uint32_t)swl.off_tab;
        typedef void (*sw_fn)(RrrSparseParams, SwExtra);
        const sw_fn fn = L <= 2 ? (K <= 3 ? rrr_sparse_wave_kernel<2, 3> : rrr_sparse_wave_kernel<2, 7>) : (K <= 4 ? rrr_sparse_wave_kernel<4, 4> : rrr_sparse_wave_kernel<4, 7>);
        HIP_TRY(ctx, raise_lds_attr(reinterpret_cast<const void*>(fn), swl.bytes));
        hipLaunchKernelGGL(fn, dim3((unsigned)R), dim3(kRrrThreads), swl.bytes, st, P, X);
    } else {
        // few replicas run one per workgroup anyway: stage the replica's hot state and the graph in LDS if they fit
        const size_t lds = rrr_sparse_lds_bytes(N, W, K);
        const char* no_lds = std::getenv("RRRMC_RRR_NO_LDS");            // tests / timing experiments
        const bool big_k = K > 8;                 // the staged path keeps its change list in registers: 8 slots, or 16 for K > 8
        const bool use_lds = !wide_idx && !big_k && rrr_tpb(R) == 1 && lds <= (size_t)kLdsLimit && !(no_lds && no_lds[0] == '1');
        // the per-class arrays are sized at compile time (2, 4 or 8 levels) so that they stay in registers
        typedef void (*rs_fn)(RrrSparseParams);
        const int slm = L <= 2 ? 0 : (L <= 4 ? 1 : 2);
        static const rs_fn lds_fns[3] = {rrr_sparse_kernel<true, 2>, rrr_sparse_kernel<true, 4>, rrr_sparse_kernel<true, 8>};
        static const rs_fn glb_fns[3] = {rrr_sparse_kernel<false, 2>, rrr_sparse_kernel<false, 4>, rrr_sparse_kernel<false, 8>};
        static const rs_fn wide_fns[3] = {rrr_sparse_kernel<false, 2, uint32_t>, rrr_sparse_kernel<false, 4, uint32_t>, rrr_sparse_kernel<false, 8, uint32_t>};
        if (big_k) {
            if (K > 16) return fail(ctx, RRRMC_ERR_UNSUPPORTED, "K=%lld: the rrrMC kernel covers K <= 16", (long long)K);
            if (wide_idx) hipLaunchKernelGGL((rrr_sparse_kernel<false, 8, uint32_t, 16>), dim3(rrr_blocks(R)), dim3(rrr_tpb(R)), 0, st, P);
            else hipLaunchKernelGGL((rrr_sparse_kernel<false, 8, uint16_t, 16>), dim3(rrr_blocks(R)), dim3(rrr_tpb(R)), 0, st, P);
        } else if (wide_idx) {
            hipLaunchKernelGGL(wide_fns[slm], dim3(rrr_blocks(R)), dim3(rrr_tpb(R)), 0, st, P);
        } else if (use_lds) {
            HIP_TRY(ctx, raise_lds_attr(reinterpret_cast<const void*>(lds_fns[slm]), lds));
            hipLaunchKernelGGL(lds_fns[slm], dim3((unsigned)R), dim3(kRrrThreads), lds, st, P);
        } else {
            hipLaunchKernelGGL(glb_fns[slm], dim3(rrr_blocks(R)), dim3(rrr_tpb(R)), 0, st, P);
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[1], st));
    if ((rc = rp_out(ctx, rv, st))) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_end, st));
    ctx->sweep_launches = 1;
    ctx->nsamp = nsamp;
    ctx->it_done += (uint64_t)iters;
    smp_commit(ctx, mode + 1, iters);
    ctx->results_valid = true;
    ctx->timing_valid = true;
    ctx->last_call_rrr = true;
    ctx->colored_call = false;
    return RRRMC_OK;
}

// wtmMC (src/RRRMC.jl:376-426) on GraphRRG / GraphEA: thread-per-replica kernel with a binary heap of next-flip times
int32_t sparse_wtm_async(rrrmc_ctx* ctx, double beta, int64_t samples, double step)
{
    if (samples < 0) return fail(ctx, RRRMC_ERR_INVALID_ARG, "samples must be >= 0, given %lld", (long long)samples);
    if (!(step > 0.0) || !std::isfinite(step)) return fail(ctx, RRRMC_ERR_INVALID_ARG, "step must be positive and finite, given %g", step);
    if (!std::isfinite(beta)) return fail(ctx, RRRMC_ERR_INVALID_ARG, "beta must be finite, given: %g", beta);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->results_valid = false; ctx->last_call_wtm = false; ctx->last_call_eo = false;
    ctx->timing_valid = false;
    const int64_t N = ctx->N, K = ctx->K, R = ctx->R;
    SmpState S{};
    { const int32_t rcs = smp_begin(ctx, 3, beta, step, 0.0, 0.0, 1, nullptr, &S); if (rcs) return rcs; }
    RpView rv;
    int32_t rc = rp_prepare(ctx, &rv);
    if (rc) return rc;
    const int64_t W = rv.W;
    if (!ctx->wt_t) {
        HIP_TRY(ctx, hipMalloc(&ctx->wt_t, sizeof(double) * R * N));
        const size_t idx_bytes = N > 65535 ? sizeof(uint32_t) : sizeof(uint16_t);
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->wt_id), idx_bytes * R * N));
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->wt_pos), idx_bytes * R * N));
        HIP_TRY(ctx, hipMalloc(&ctx->wt_time, sizeof(double) * R));
    }
    const size_t es_need = (size_t)(samples > 0 ? samples : 1) * ctx->Rpad;
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
    WtmParams P{};
    P.A = ctx->d_A; P.J = ctx->d_J; P.spins = rv.spins; P.ht = ctx->wt_t; P.hid = ctx->wt_id; P.hpos = ctx->wt_pos;
    P.E_cur = ctx->d_E; P.acc_cur = ctx->d_acc; P.t_out = ctx->wt_time; P.Es = ctx->d_Es;
    P.lv = ctx->lv;
    for (int a = 0; a < ctx->lv.L; ++a)           // tauDE = max(1, exp(beta dE)), WaitingTimes.jl:16; dE = -+dElist[a]
        for (int sgn = 0; sgn < 2; ++sgn) {
            const double e = host_det_exp(beta * lv_to_f64(ctx, sgn ? ctx->lv.dElist[a] : -ctx->lv.dElist[a]));
            P.tau[a + (sgn ? ctx->lv.L : 0)] = e > 1.0 ? e : 1.0;
        }
    (void)K;
    P.step = step / (double)N; P.samples = samples;
    P.S = S; P.samples_before = ctx->smp_it;
    P.k0 = (uint32_t)ctx->seed; P.k1 = (uint32_t)(ctx->seed >> 32); P.replica0 = ctx->replica0; P.call = ctx->smp_call & 0xffffffu;
    P.N = (int)N; P.K = (int)K; P.W = (int)W; P.R = (int)R; P.Rpad = (int)ctx->Rpad;
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_begin, st));
    if ((rc = rp_in(ctx, rv, st))) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[0], st));
    if (N > 65535) hipLaunchKernelGGL(wtm_sparse_kernel<uint32_t>, dim3(rrr_blocks(R)), dim3(rrr_tpb(R)), 0, st, P);
    else hipLaunchKernelGGL(wtm_sparse_kernel<uint16_t>, dim3(rrr_blocks(R)), dim3(rrr_tpb(R)), 0, st, P);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[1], st));
    if ((rc = rp_out(ctx, rv, st))) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_end, st));
    if (!S.resume) ctx->wtm_calls += 1;
    smp_commit(ctx, 3, samples);
    ctx->sweep_launches = 1;
    ctx->nsamp = samples;
    ctx->results_valid = true;
    ctx->timing_valid = true;
    ctx->last_call_rrr = false;
    ctx->last_call_wtm = true;
    ctx->colored_call = false;
    return RRRMC_OK;
}

// extremal_opt (src/RRRMC.jl:474-521) on GraphRRG / GraphEA: thread-per-replica kernel over the rrrMC class arrays
int32_t sparse_eo_async(rrrmc_ctx* ctx, const double* ftau, int64_t iters, int64_t step)
{
    if (iters < 0) return fail(ctx, RRRMC_ERR_INVALID_ARG, "iters must be >= 0, given %lld", (long long)iters);
    if (step < 1) return fail(ctx, RRRMC_ERR_INVALID_ARG, "step must be >= 1, given %lld", (long long)step);
    if (!ftau) return fail(ctx, RRRMC_ERR_INVALID_ARG, "ftau is NULL");
    const int64_t N = ctx->N, K = ctx->K, R = ctx->R;
    for (int64_t i = 0; i < N; ++i)
        if (!(ftau[i] > 0.0) || !std::isfinite(ftau[i]) || (i && ftau[i] < ftau[i - 1]))
            return fail(ctx, RRRMC_ERR_INVALID_ARG, "ftau must be a positive non-decreasing table (cumsum of j^-tau), violated at %lld", (long long)i);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->results_valid = false; ctx->last_call_wtm = false; ctx->last_call_eo = false;
    ctx->timing_valid = false;
    const int L = ctx->lv.L;
    SmpState S{};
    { const int32_t rcs = smp_begin(ctx, 4, 0.0, 0.0, 0.0, 0.0, step, ftau, &S); if (rcs) return rcs; }
    RpView rv;
    int32_t rc = rp_prepare(ctx, &rv);
    if (rc) return rc;
    const int64_t W = rv.W;
    ctx->eo_W = W;
    if (!ctx->rp_cls) {
        HIP_TRY(ctx, hipMalloc(&ctx->rp_cls, (size_t)R * N));
        const size_t idx_bytes = N > 65535 ? sizeof(uint32_t) : sizeof(uint16_t);
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->rp_sv), idx_bytes * R * 2 * L * N));
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->rp_spos), idx_bytes * R * N));
        if (!ctx->q_stats) HIP_TRY(ctx, hipMalloc(&ctx->q_stats, sizeof(int64_t) * R * 3));
    }
    if (!ctx->eo_cmin) {
        HIP_TRY(ctx, hipMalloc(&ctx->eo_cmin, sizeof(uint32_t) * R * 2 * ((N + 63) / 64)));
        HIP_TRY(ctx, hipMalloc(&ctx->eo_ftau, sizeof(double) * N));
    }
    ctx->stats_stride = 3;
    const int64_t nsamp = smp_nsamp(ctx, iters, step);
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
    hipStream_t st = ctx->stream;
    if (!S.resume) {                    // (a resumed call: the run's table and its Cmin are on the device)
        HIP_TRY(ctx, hipStreamSynchronize(st));
        HIP_TRY(ctx, hipMemcpy(ctx->eo_ftau, ftau, sizeof(double) * N, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemset(ctx->eo_cmin, 0, sizeof(uint32_t) * R * 2 * ((N + 63) / 64)));
    }
    EoParams P{};
    P.S = S;
    P.A = ctx->d_A; P.J = ctx->d_J; P.ftau = ctx->eo_ftau; P.spins = rv.spins; P.cmin = ctx->eo_cmin;
    P.lv = ctx->lv;
    P.cls = ctx->rp_cls; P.sv = ctx->rp_sv; P.spos = ctx->rp_spos; P.E_cur = ctx->d_E; P.stats = ctx->q_stats; P.Es = ctx->d_Es;
    P.g0 = ctx->it_done; P.iters = iters; P.step = step;
    P.k0 = (uint32_t)ctx->seed; P.k1 = (uint32_t)(ctx->seed >> 32); P.replica0 = ctx->replica0;
    P.N = (int)N; P.K = (int)K; P.L = L; P.has_zero = ctx->lv.dElist[0] == 0 ? 1 : 0; P.W = (int)W; P.R = (int)R; P.Rpad = (int)ctx->Rpad;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_begin, st));
    if ((rc = rp_in(ctx, rv, st))) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[0], st));
    {
        const char* no_lds = std::getenv("RRRMC_EO_NO_FTAU_LDS");       // tests / timing experiments
        P.ftau_lds = N <= kEoFtauLdsMaxN && !(no_lds && no_lds[0] == '1') ? 1 : 0;
        const size_t lds = eo_sparse_lds_bytes(N, P.ftau_lds != 0, rrr_tpb(R));
        typedef void (*eo_fn)(EoParams);
        static const eo_fn fns[2][3] = {{eo_sparse_kernel<uint16_t, 2>, eo_sparse_kernel<uint16_t, 4>, eo_sparse_kernel<uint16_t, 8>},
                                        {eo_sparse_kernel<uint32_t, 2>, eo_sparse_kernel<uint32_t, 4>, eo_sparse_kernel<uint32_t, 8>}};
        const eo_fn fn = fns[N > 65535 ? 1 : 0][L <= 2 ? 0 : (L <= 4 ? 1 : 2)];      // (class counters sized at compile time, as rrr_sparse_kernel's)
        HIP_TRY(ctx, raise_lds_attr(reinterpret_cast<const void*>(fn), lds));
        hipLaunchKernelGGL(fn, dim3(rrr_blocks(R)), dim3(rrr_tpb(R)), lds, st, P);
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[1], st));
    if ((rc = rp_out(ctx, rv, st))) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_end, st));
    ctx->sweep_launches = 1;
    ctx->nsamp = nsamp;
    ctx->it_done += (uint64_t)iters;
    smp_commit(ctx, 4, iters);
    ctx->results_valid = true;
    ctx->timing_valid = true;
    ctx->last_call_rrr = false;
    ctx->last_call_eo = true;
    ctx->colored_call = false;
    return RRRMC_OK;
}
