// Host side of the dense SK models (GraphSKNormal, GraphSK): context creation, energy, standardMC orchestration.
// Included by rrrmc_hip.hip inside its anonymous namespace, after the context struct and the common helpers
// (fail, HIP_TRY, free_dev, ensure_state); not a stand-alone translation unit.
// ---- GraphSKNormal (dense Float64) host side -----------------------------------------------------------------------
typedef void (*sk_fn)(SkParams);
sk_fn sk_sweep_for(int spt, int nth)
{
    if (nth == 1024) return spt == 1 ? sk_sweep_kernel<1, 1024> : spt == 2 ? sk_sweep_kernel<2, 1024> : nullptr;
    if (nth == 512) return spt == 1 ? sk_sweep_kernel<1, 512> : spt == 2 ? sk_sweep_kernel<2, 512> : spt == 3 ? sk_sweep_kernel<3, 512> : spt == 4 ? sk_sweep_kernel<4, 512> : nullptr;
    switch (spt) {
        case 1: return sk_sweep_kernel<1, 256>; case 2: return sk_sweep_kernel<2, 256>; case 3: return sk_sweep_kernel<3, 256>; case 4: return sk_sweep_kernel<4, 256>;
        case 5: return sk_sweep_kernel<5, 256>; case 6: return sk_sweep_kernel<6, 256>; case 7: return sk_sweep_kernel<7, 256>; case 8: return sk_sweep_kernel<8, 256>;
        default: return nullptr;
    }
}
// threads per workgroup of sk_sweep_kernel: 512 beyond N = 256 (measured at N = 1024, 2048 replicas: 256 threads 101.7 ms per 65 536
// iterations, 512 threads 82.5, 1024 threads 85.1); RRRMC_SK_THREADS = 256 / 512 / 1024 overrides — the builds are bit-identical, the
// tests compare them
int sk_threads_for(int64_t N)
{
    int nth = N > 256 ? 512 : 256;
    if (const char* e = std::getenv("RRRMC_SK_THREADS")) {
        const int v = std::atoi(e);
        if ((v == 256 || v == 512 || v == 1024) && (N + v - 1) / v <= (v == 256 ? 8 : v == 512 ? 4 : 2)) nth = v;
    }
    return nth;
}

// the blocked kernel (sk_block_kernel.hpp): 256 threads up to N = 256, 512 threads (one to four sites per thread) beyond, 1024 on request;
// nullptr = this shape keeps sk_sweep_kernel
// row stride of the 4J matrix: rows are zero-padded to a multiple of 1024 so that the blocked kernel's row loads need no bounds test
int64_t sk_ldJ(int64_t N) { return (N + 1023) / 1024 * 1024; }
typedef void (*sk_block_fn)(SkBlockParams);
sk_block_fn sk_block_for(int spt, int nth)
{
    if (nth == 256) return spt == 1 ? sk_block_kernel<1, 256> : nullptr;
    if (nth == 512) return spt == 1 ? sk_block_kernel<1, 512> : spt == 2 ? sk_block_kernel<2, 512> : spt == 3 ? sk_block_kernel<3, 512> : spt == 4 ? sk_block_kernel<4, 512> : nullptr;
    if (nth == 1024) return spt == 1 ? sk_block_kernel<1, 1024> : spt == 2 ? sk_block_kernel<2, 1024> : nullptr;
    return nullptr;
}
// the same kernels for the binary model (BIN: fields and the +-4.0 matrix as doubles, dE = lfields / sqrt(N)); 1024 threads are not offered
sk_block_fn skb_block_for(int spt, int nth)
{
    if (nth == 256) return spt == 1 ? sk_block_kernel<1, 256, 8, true> : nullptr;
    if (nth == 512) return spt == 1 ? sk_block_kernel<1, 512, 8, true> : spt == 2 ? sk_block_kernel<2, 512, 8, true> : spt == 3 ? sk_block_kernel<3, 512, 8, true> : spt == 4 ? sk_block_kernel<4, 512, 8, true> : nullptr;
    return nullptr;
}
sk_block_fn skb_block_half_for(int spt)
{
    switch (spt) {
        case 1: return sk_block_kernel<1, 256, 4, true>; case 2: return sk_block_kernel<2, 256, 4, true>;
        case 3: return sk_block_kernel<3, 256, 4, true>; case 4: return sk_block_kernel<4, 256, 4, true>;
        default: return nullptr;
    }
}
constexpr int kSkMaxN = 4096;         // sk_hblock_kernel<8, 512, 8>: eight sites per thread (the round-3 kernels stop at 2048)
// round 4: the same shapes with the branch-free bulk phase (sk_hblock_kernel.hpp); RRRMC_SK_BLOCK_V1 = 1 keeps round 3's sk_block_kernel
// (bit-identical; the tests compare them).  1024 threads per workgroup are not offered by the new kernel (128 registers per thread).
bool sk_block_v1_forced()
{
    const char* e = std::getenv("RRRMC_SK_BLOCK_V1");
    return e && e[0] == '1';
}
template <bool BIN>
sk_block_fn sk_hblock_for(int spt, int nth)
{
    if (nth == 256) return spt == 1 ? sk_hblock_kernel<1, 256, 8, BIN> : nullptr;
    if (nth == 512) return spt == 1 ? sk_hblock_kernel<1, 512, 8, BIN> : spt == 2 ? sk_hblock_kernel<2, 512, 8, BIN> : spt == 3 ? sk_hblock_kernel<3, 512, 8, BIN> : spt == 4 ? sk_hblock_kernel<4, 512, 8, BIN>
                         : spt <= 6 ? sk_hblock_kernel<6, 512, 8, BIN> : spt <= 8 ? sk_hblock_kernel<8, 512, 8, BIN> : nullptr;      // 2048 < N <= 4096 (round 4)
    return nullptr;
}
template <bool BIN>
sk_block_fn sk_hblock_half_for(int spt)
{
    switch (spt) {
        case 1: return sk_hblock_kernel<1, 256, 4, BIN>; case 2: return sk_hblock_kernel<2, 256, 4, BIN>;
        case 3: return sk_hblock_kernel<3, 256, 4, BIN>; case 4: return sk_hblock_kernel<4, 256, 4, BIN>;
        default: return nullptr;
    }
}
// two 4-replica workgroups of 256 threads per group of 8 replicas (co-resident on a compute unit: one decides while the other applies)
sk_block_fn sk_block_half_for(int spt)
{
    switch (spt) {
        case 1: return sk_block_kernel<1, 256, 4>; case 2: return sk_block_kernel<2, 256, 4>;
        case 3: return sk_block_kernel<3, 256, 4>; case 4: return sk_block_kernel<4, 256, 4>;
        default: return nullptr;
    }
}
// Measured (2048 replicas, beta = 1, 65 536 iterations, ms per launch, whole group / split):  N = 128: 30.6 / 19.8,  256: 29.8 / 19.4,
// 512: 26.4 / 21.5,  768: 30.7 / 24.5,  1024: 30.9 / 26.6.  Beyond N = 1024 the split build would need 512 threads and two workgroups of
// 8 wavefronts per compute unit, i.e. 128 registers per thread for a state that takes 112 of them: whole groups there.
// RRRMC_SK_RB = 4 / 8 forces the split / whole-group build (bit-identical; the tests compare them)
// Round 4's kernel (sk_hblock_kernel), same measurement, whole group / split: N = 128: 17.8 / 12.1,  256: 17.5 / 12.0,  512: 14.3 / 12.2,  768: 15.7 / 16.7,
// 1024: 15.7 / 21.9 (the split build loads every row of 4J twice per compute unit, and the bulk phase no longer hides that).
int sk_rb_for(int64_t N, bool v1)
{
    int rb = N <= (v1 ? 1024 : 512) ? 4 : 8;
    if (const char* e = std::getenv("RRRMC_SK_RB")) {
        const int v = std::atoi(e);
        if (v == 8 || (v == 4 && N <= 1024)) rb = v;
    }
    return rb;
}
bool sk_legacy_forced()
{
    const char* e = std::getenv("RRRMC_SK_LEGACY");
    return e && e[0] == '1';
}

typedef void (*skb_fn)(SkbParams);
skb_fn skb_sweep_for(int spt, int nth)
{
    if (nth == 1024) return spt == 1 ? skb_sweep_kernel<1, 1024> : spt == 2 ? skb_sweep_kernel<2, 1024> : nullptr;
    if (nth == 512) return spt == 1 ? skb_sweep_kernel<1, 512> : spt == 2 ? skb_sweep_kernel<2, 512> : spt == 3 ? skb_sweep_kernel<3, 512> : spt == 4 ? skb_sweep_kernel<4, 512> : nullptr;
    switch (spt) {
        case 1: return skb_sweep_kernel<1, 256>; case 2: return skb_sweep_kernel<2, 256>; case 3: return skb_sweep_kernel<3, 256>; case 4: return skb_sweep_kernel<4, 256>;
        case 5: return skb_sweep_kernel<5, 256>; case 6: return skb_sweep_kernel<6, 256>; case 7: return skb_sweep_kernel<7, 256>; case 8: return skb_sweep_kernel<8, 256>;
        default: return nullptr;
    }
}

int32_t sk_ctx_create(rrrmc_ctx** out, int32_t model, int64_t N, int64_t R, int32_t device, uint32_t replica0)
{
    if (N < 1 || R < 1) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "N, R must be >= 1 (given N=%lld R=%lld)", (long long)N, (long long)R);
    if (N > kSkMaxN) return fail(nullptr, RRRMC_ERR_UNSUPPORTED, "N=%lld: the register-resident SK kernels cover N <= %d", (long long)N, kSkMaxN);
    if (replica0 % 32) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "replica0 must be a multiple of 32 (given %u)", replica0);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, RRRMC_ERR_HIP, "no HIP device is visible: this library has no CPU path");
    if (device < 0 || device >= ndev) return fail(nullptr, RRRMC_ERR_INVALID_ARG, "device %d out of range (0..%d)", device, ndev - 1);
    rrrmc_ctx* ctx = new (std::nothrow) rrrmc_ctx();
    if (!ctx) return fail(nullptr, RRRMC_ERR_NOMEM, "out of host memory");
    ctx->model = model; ctx->N = N; ctx->K = 0; ctx->R = R;
    ctx->G8 = (R + kSkRB - 1) / kSkRB; ctx->Rpad = ctx->G8 * kSkRB; ctx->G = 0;
    ctx->device = device; ctx->replica0 = replica0;
#define SK_TRY(expr)                                                                                             \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess) {                                                                                  \
            int32_t rc_ = fail(nullptr, RRRMC_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));           \
            rrrmc_ctx_destroy(ctx);                                                                              \
            return rc_;                                                                                          \
        }                                                                                                        \
    } while (0)
    SK_TRY(hipSetDevice(device));
    SK_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    SK_TRY(hipEventCreate(&ctx->ev_begin));
    SK_TRY(hipEventCreate(&ctx->ev_end));
    const size_t nf = (size_t)ctx->G8 * N * kSkRB;
    if (model == RRRMC_MODEL_SK_BINARY) {
        ctx->skb_NW = (int)(2 * ((N + 63) / 64));
        SK_TRY(hipMalloc(&ctx->skb_J, sizeof(uint32_t) * N * ctx->skb_NW));
        SK_TRY(hipMalloc(&ctx->skb_lf, sizeof(int32_t) * nf));
        SK_TRY(hipMalloc(&ctx->skb_lfl, sizeof(int32_t) * nf));
        SK_TRY(hipMemset(ctx->skb_lf, 0, sizeof(int32_t) * nf));
        SK_TRY(hipMemset(ctx->skb_lfl, 0, sizeof(int32_t) * nf));
    } else {
        SK_TRY(hipMalloc(&ctx->sk_J, sizeof(double) * N * N));
        SK_TRY(hipMalloc(&ctx->sk_J4, sizeof(double) * N * sk_ldJ(N)));
        SK_TRY(hipMalloc(&ctx->sk_lf, sizeof(double) * nf));
        SK_TRY(hipMalloc(&ctx->sk_lfl, sizeof(double) * nf));
        SK_TRY(hipMemset(ctx->sk_lf, 0, sizeof(double) * nf));
        SK_TRY(hipMemset(ctx->sk_lfl, 0, sizeof(double) * nf));
    }
    SK_TRY(hipMalloc(&ctx->sk_move_last, sizeof(int32_t) * ctx->Rpad));
    SK_TRY(hipMalloc(&ctx->sk_spins, ((size_t)ctx->G8 * N + 3) / 4 * 4));       // (whole words: sk_block_kernel<.., 4> updates its bits by word atomics)
    SK_TRY(hipMalloc(&ctx->sk_E, sizeof(double) * ctx->Rpad));
    SK_TRY(hipMalloc(&ctx->d_acc, sizeof(int64_t) * ctx->Rpad));
    SK_TRY(hipMemset(ctx->sk_spins, 0, (size_t)ctx->G8 * N));
#undef SK_TRY
    *out = ctx;
    return RRRMC_OK;
}

// energy(X, C), SK.jl:212-237: rebuilds lfields, zeroes lfields_last, move_last = none; E into sk_E
int32_t sk_run_energy(rrrmc_ctx* ctx)
{
    const dim3 grid((unsigned)((ctx->N + 31) / 32), (unsigned)ctx->G8);
    if (ctx->model == RRRMC_MODEL_SK_BINARY) {
        hipLaunchKernelGGL(skb_fields_kernel, grid, dim3(256), 0, ctx->stream, ctx->skb_J, ctx->sk_spins, ctx->skb_lf, ctx->skb_lfl,
                           ctx->sk_move_last, (int)ctx->N, ctx->skb_NW);
        HIP_TRY(ctx, hipGetLastError());
        hipLaunchKernelGGL(skb_energy_kernel, dim3((unsigned)((ctx->Rpad + 63) / 64)), dim3(64), 0, ctx->stream, ctx->skb_lf, ctx->sk_spins,
                           ctx->sk_E, (int)ctx->N, (int)ctx->Rpad, std::sqrt((double)ctx->N));
        HIP_TRY(ctx, hipGetLastError());
        return RRRMC_OK;
    }
    // one thread per site with the 8 replicas in registers; the debug check below recomputes with the one-chain-per-thread kernel
    hipLaunchKernelGGL(sk_fields8_kernel, dim3((unsigned)((ctx->N + 255) / 256), (unsigned)ctx->G8), dim3(256), 0, ctx->stream, ctx->sk_J, ctx->sk_spins,
                       ctx->sk_lf, ctx->sk_lfl, ctx->sk_move_last, (int)ctx->N);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(sk_energy_kernel, dim3((unsigned)((ctx->Rpad + 63) / 64)), dim3(64), 0, ctx->stream, ctx->sk_lf, ctx->sk_E,
                       (int)ctx->N, (int)ctx->Rpad);
    HIP_TRY(ctx, hipGetLastError());
    return RRRMC_OK;
}

// debug mode (rrrmc_set_debug_checks) after a standardMC call on GraphSKNormal: the reference's commented-out check of update_cache!
// (SK.jl:268-273: energy(X, C) again, maximum(abs.(lfields_bk - lfields)) < 1e-10) and the test suite's tracked E == energy(X, C)
// (test/runtests.jl:12-20), on the device: fields and energy recomputed into scratch buffers, compared element-wise
int32_t sk_debug_check(rrrmc_ctx* ctx)
{
    const size_t nf = (size_t)ctx->G8 * ctx->N * kSkRB;
    if (!ctx->dbg_flag) { HIP_TRY(ctx, hipMalloc(&ctx->dbg_flag, sizeof(int32_t) * 2)); HIP_TRY(ctx, hipMemsetAsync(ctx->dbg_flag, 0, sizeof(int32_t) * 2, ctx->stream)); }
    if (!ctx->dbg_lf) {
        HIP_TRY(ctx, hipMalloc(&ctx->dbg_lf, sizeof(double) * nf));
        HIP_TRY(ctx, hipMalloc(&ctx->dbg_lfl, sizeof(double) * nf));
        HIP_TRY(ctx, hipMalloc(&ctx->dbg_E, sizeof(double) * ctx->Rpad));
        HIP_TRY(ctx, hipMalloc(&ctx->dbg_ml, sizeof(int32_t) * ctx->Rpad));
    }
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(sk_fields_kernel, dim3((unsigned)((ctx->N + 31) / 32), (unsigned)ctx->G8), dim3(256), 0, st, ctx->sk_J, ctx->sk_spins,
                       ctx->dbg_lf, ctx->dbg_lfl, ctx->dbg_ml, (int)ctx->N);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(sk_energy_kernel, dim3((unsigned)((ctx->Rpad + 63) / 64)), dim3(64), 0, st, ctx->dbg_lf, ctx->dbg_E, (int)ctx->N, (int)ctx->Rpad);
    HIP_TRY(ctx, hipGetLastError());
    if (dbg_inject()) hipLaunchKernelGGL(dbg_inject_f64_kernel, dim3(1), dim3(1), 0, st, ctx->dbg_E);                 // (RRRMC_DEBUG_INJECT=1: tests only)
    const double tol = 1e-10 * (double)ctx->N;               // the reference's 1e-10 / 1e-11 at its test sizes, scaled with the length of the sums
    hipLaunchKernelGGL(dbg_compare_f64_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, ctx->sk_lf, ctx->dbg_lf, (long long)nf, tol,
                       (long long)ctx->N * kSkRB, (int)kSkRB, ctx->dbg_flag);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(dbg_compare_f64_kernel, dim3((unsigned)((ctx->R + 255) / 256)), dim3(256), 0, st, ctx->sk_E, ctx->dbg_E, (long long)ctx->R, tol,
                       (long long)ctx->Rpad, (int)ctx->Rpad, ctx->dbg_flag);
    HIP_TRY(ctx, hipGetLastError());
    return RRRMC_OK;
}

int32_t sk_standard_mc_async(rrrmc_ctx* ctx, double beta, int64_t iters, int64_t step)
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
    while (ctx->ev_sweep.size() < 2) {
        hipEvent_t e;
        HIP_TRY(ctx, hipEventCreate(&e));
        ctx->ev_sweep.push_back(e);
    }
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_begin, st));
    // E = energy(X, C) at the start of every call (RRRMC.jl:95) — unless the call resumes the previous one (rrrmc_set_resume): then the
    // tracked energy, both field arrays and move_last continue, as inside one reference call
    if (!(ctx->resume && ctx->std_cache_live)) { const int32_t rc = sk_run_energy(ctx); if (rc) return rc; }
    ctx->std_cache_live = true;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_acc, 0, sizeof(int64_t) * ctx->Rpad, st));
    const bool binary = ctx->model == RRRMC_MODEL_SK_BINARY;
    const double sqrtN = std::sqrt((double)ctx->N);
    const int nth = sk_threads_for(ctx->N), spt = (int)((ctx->N + nth - 1) / nth);
    int blk_nth = binary && nth == 1024 ? 512 : nth;     // threads per workgroup (the binary build has no 1024-thread form)
    const int blk_spt = (int)((ctx->N + blk_nth - 1) / blk_nth);
    const bool v1 = sk_block_v1_forced() || blk_nth == 1024;
    sk_block_fn blockk = sk_legacy_forced() ? nullptr : v1 ? (binary ? skb_block_for(blk_spt, blk_nth) : sk_block_for(spt, nth))
                                                           : (binary ? sk_hblock_for<true>(blk_spt, blk_nth) : sk_hblock_for<false>(blk_spt, blk_nth));
    int blk_wgs = 1;                                      // workgroups per group of 8 replicas
    if (blockk && sk_rb_for(ctx->N, v1) == 4) {
        const int hs = (int)((ctx->N + 255) / 256);
        const sk_block_fn h = v1 ? (binary ? skb_block_half_for(hs) : sk_block_half_for(hs)) : (binary ? sk_hblock_half_for<true>(hs) : sk_hblock_half_for<false>(hs));
        if (h) { blockk = h; blk_nth = 256; blk_wgs = 2; }
    }
    if (!blockk && !sk_legacy_forced() && ctx->N <= 1024) {
        // no whole-group build of this shape (e.g. RRRMC_SK_THREADS = 256 beyond N = 256): the split build covers every N <= 1024
        const int hs = (int)((ctx->N + 255) / 256);
        const sk_block_fn h = v1 ? (binary ? skb_block_half_for(hs) : sk_block_half_for(hs)) : (binary ? sk_hblock_half_for<true>(hs) : sk_hblock_half_for<false>(hs));
        if (h) { blockk = h; blk_nth = 256; blk_wgs = 2; }
    }
    // (still none: the one-attempt-at-a-time kernels below, for both models)

    if (iters == 0) {
        // nothing to sweep: the call is E = energy(X, C) (RRRMC.jl:95) and an empty sample list — the bindings issue it before a hooked run.
        // No sweep kernel is launched (some shapes, 2048 < N <= 4096, have a blocked build only, which needs iterations to plan)
        HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[0], st));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[1], st));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_end, st));
        ctx->sweep_launches = 1;
        ctx->nsamp = 0;
        ctx->results_valid = true;
        ctx->timing_valid = true;
        return RRRMC_OK;
    }
    if (!blockk) {
        // no blocked build covers the shape (RRRMC_SK_LEGACY / RRRMC_SK_BLOCK_V1 / RRRMC_SK_THREADS can ask for one that does not exist)
        const int nthl = sk_threads_for(ctx->N), sptl = (int)((ctx->N + nthl - 1) / nthl);
        if (binary ? skb_sweep_for(sptl, nthl) == nullptr : sk_sweep_for(sptl, nthl) == nullptr)
            return fail(ctx, RRRMC_ERR_UNSUPPORTED, "no dense-SK kernel build covers N=%lld with %d threads per workgroup under the requested RRRMC_SK_* overrides",
                        (long long)ctx->N, nthl);
    }

    if (binary && blockk && iters > 0) {
        // the blocked kernel on the binary model's state as doubles (sk_block_kernel<.., BIN>): the +-4.0 matrix is built from the bit rows
        // once, the integer fields travel int32 -> Float64 -> int32 around the call (exact)
        const size_t nf = (size_t)ctx->G8 * ctx->N * kSkRB;
        const int64_t ld = sk_ldJ(ctx->N);
        if (!ctx->sk_lf) { HIP_TRY(ctx, hipMalloc(&ctx->sk_lf, sizeof(double) * nf)); HIP_TRY(ctx, hipMalloc(&ctx->sk_lfl, sizeof(double) * nf)); }
        if (!ctx->sk_J4) HIP_TRY(ctx, hipMalloc(&ctx->sk_J4, sizeof(double) * ctx->N * ld));
        if (!ctx->skb_J4_valid) {
            hipLaunchKernelGGL(skb_dense4_kernel, dim3((unsigned)((ld + 255) / 256), (unsigned)ctx->N), dim3(256), 0, st, ctx->skb_J, ctx->sk_J4, (int)ctx->N,
                               ctx->skb_NW, (int)ld);
            HIP_TRY(ctx, hipGetLastError());
            ctx->skb_J4_valid = true;
        }
        hipLaunchKernelGGL(skb_fields_to_f64_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, ctx->skb_lf, ctx->skb_lfl, ctx->sk_lf, ctx->sk_lfl, nf);
        HIP_TRY(ctx, hipGetLastError());
    } else if (binary) {
        SkbParams B{};
        B.Jbits = ctx->skb_J; B.lf = ctx->skb_lf; B.lfl = ctx->skb_lfl; B.move_last = ctx->sk_move_last; B.spins = ctx->sk_spins;
        B.E_cur = ctx->sk_E; B.acc_cur = ctx->d_acc; B.Es = ctx->sk_Es;
        B.beta = beta; B.sN = sqrtN; B.g0 = ctx->it_done; B.iters = iters; B.step = step; B.sample0 = 0;
        B.k0 = (uint32_t)ctx->seed; B.k1 = (uint32_t)(ctx->seed >> 32); B.replica0 = ctx->replica0; B.N = (int)ctx->N; B.NW = ctx->skb_NW;
        const int nthb = sk_threads_for(ctx->N), sptb = (int)((ctx->N + nthb - 1) / nthb);
        HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[0], st));
        hipLaunchKernelGGL(skb_sweep_for(sptb, nthb), dim3((unsigned)ctx->G8), dim3((unsigned)nthb), 0, st, B);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[1], st));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_end, st));
        ctx->sweep_launches = 1;
        ctx->nsamp = nsamp;
        ctx->it_done += (uint64_t)iters;
        ctx->results_valid = true;
        ctx->timing_valid = true;
        return RRRMC_OK;
    }
    SkParams P{};
    P.J = ctx->sk_J; P.lf = ctx->sk_lf; P.lfl = ctx->sk_lfl; P.move_last = ctx->sk_move_last; P.spins = ctx->sk_spins;
    P.E_cur = ctx->sk_E; P.acc_cur = ctx->d_acc; P.Es = ctx->sk_Es;
    P.beta = beta; P.g0 = ctx->it_done; P.iters = iters; P.step = step; P.sample0 = 0;
    P.k0 = (uint32_t)ctx->seed; P.k1 = (uint32_t)(ctx->seed >> 32); P.replica0 = ctx->replica0; P.N = (int)ctx->N;
    if (blockk && iters > 0) {
        // blocked kernel: segments of <= kSkSegIters iterations, each with its state-independent block tables (sites, coupling sub-matrices)
        const int64_t nseg = (iters + kSkSegIters - 1) / kSkSegIters;
        const int64_t cap_blk = ((iters < kSkSegIters ? iters : kSkSegIters) + kSkW - 1) / kSkW;
        if ((size_t)cap_blk > ctx->sk_blk_cap) {
            free_dev(ctx->sk_blkJw); free_dev(ctx->sk_blkSites);
            ctx->sk_blk_cap = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->sk_blkJw, sizeof(double) * (size_t)cap_blk * kSkW * kSkW));
            HIP_TRY(ctx, hipMalloc(&ctx->sk_blkSites, sizeof(uint32_t) * (size_t)(cap_blk + 1) * kSkW));
            ctx->sk_blk_cap = (size_t)cap_blk;
        }
        while ((int64_t)ctx->ev_sweep.size() < 2 * nseg) {
            hipEvent_t e;
            HIP_TRY(ctx, hipEventCreate(&e));
            ctx->ev_sweep.push_back(e);
        }
        if (!v1 && !ctx->sk_hl) HIP_TRY(ctx, hipMalloc(&ctx->sk_hl, sizeof(double) * (size_t)ctx->Rpad * (size_t)ctx->N));
        SkBlockParams Bk{};
        Bk.hl = ctx->sk_hl;
        Bk.J4 = ctx->sk_J4; Bk.blkJw = ctx->sk_blkJw; Bk.blkSites = ctx->sk_blkSites;
        Bk.lf = ctx->sk_lf; Bk.lfl = ctx->sk_lfl; Bk.move_last = ctx->sk_move_last; Bk.spins = ctx->sk_spins;
        Bk.E_cur = ctx->sk_E; Bk.acc_cur = ctx->d_acc; Bk.Es = ctx->sk_Es;
        Bk.beta = beta; Bk.sN = sqrtN; Bk.step = step; Bk.k0 = P.k0; Bk.k1 = P.k1; Bk.replica0 = ctx->replica0; Bk.N = (int)ctx->N; Bk.ldJ = (int)sk_ldJ(ctx->N);
        for (int64_t sg = 0; sg < nseg; ++sg) {
            const int64_t base = sg * kSkSegIters, n = iters - base < kSkSegIters ? iters - base : kSkSegIters, nblk = (n + kSkW - 1) / kSkW;
            Bk.g0 = ctx->it_done + (uint64_t)base; Bk.iters = n; Bk.it_base = base;
            hipLaunchKernelGGL(sk_block_prep_kernel, dim3((unsigned)(nblk + 1)), dim3(256), 0, st, ctx->sk_J4, ctx->sk_blkSites, ctx->sk_blkJw,
                               Bk.g0, n, nblk, Bk.k0, Bk.k1, Bk.N, Bk.ldJ);
            HIP_TRY(ctx, hipGetLastError());
            HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[2 * sg], st));
            hipLaunchKernelGGL(blockk, dim3((unsigned)(ctx->G8 * blk_wgs)), dim3((unsigned)blk_nth), 0, st, Bk);
            HIP_TRY(ctx, hipGetLastError());
            HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[2 * sg + 1], st));
        }
        if (binary) {
            const size_t nf = (size_t)ctx->G8 * ctx->N * kSkRB;
            hipLaunchKernelGGL(skb_fields_from_f64_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, ctx->sk_lf, ctx->sk_lfl, ctx->skb_lf, ctx->skb_lfl, nf);
            HIP_TRY(ctx, hipGetLastError());
        }
        HIP_TRY(ctx, hipEventRecord(ctx->ev_end, st));
        ctx->sweep_launches = (int)nseg;
        ctx->nsamp = nsamp;
        ctx->it_done += (uint64_t)iters;
        ctx->results_valid = true;
        ctx->timing_valid = true;
        return ctx->debug_checks && !binary ? sk_debug_check(ctx) : RRRMC_OK;
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[0], st));
    hipLaunchKernelGGL(sk_sweep_for(spt, nth), dim3((unsigned)ctx->G8), dim3((unsigned)nth), 0, st, P);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[1], st));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_end, st));
    ctx->sweep_launches = 1;
    ctx->nsamp = nsamp;
    ctx->it_done += (uint64_t)iters;
    ctx->results_valid = true;
    ctx->timing_valid = true;
    return ctx->debug_checks ? sk_debug_check(ctx) : RRRMC_OK;
}
