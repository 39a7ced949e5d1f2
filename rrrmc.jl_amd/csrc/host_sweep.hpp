// Host side of the headline path: standardMC on the +-J sparse models (RRRMC_MODEL_SPARSE_PM1) — the chunk list of a call, the site-stream
// planner one batch ahead of the sweep on its own stream (two buffer sets, two chunk tables: the planner of a call runs beside the previous
// call's last sweep), the sweep launches and their HIP events.  Included by rrrmc_hip.hip inside its anonymous namespace, after the context
// struct and the common helpers; not a stand-alone translation unit.  The profile JSONs of the headline kernel (profiles/rNN/traffic.json,
// valu_model.json) are stamped over this file, host_plan.hpp, sparse_kernels.hpp and philox.hpp (bench.py: SWEEP_KERNEL_SOURCES): a change
// to HOW the kernel is launched invalidates them like a change to the kernel.
// Chunk list of a random-site sampling call (shared by the +-J kernel and the fast Float64 one): cuts at every multiple of `step`
// and every C moves.  On return ctx->chunks_n / ctx->chunk_batches describe the list, *reuse_out says whether the list already
// on the device is the same one (then nothing needs to be uploaded).
int32_t prepare_chunk_list(rrrmc_ctx* ctx, int64_t iters, int64_t step, int C, bool* reuse_out)
{
    // chunk list: cuts at every multiple of `step` (a sample precedes the move of iteration k*step) and every C moves.
    // ChunkDesc::g0 is relative to the call (the kernels add gbase = it_done), so the list depends on (iters, step, C) only:
    // back-to-back calls of one shape reuse the list already on the device.  A new shape rewrites the pinned staging buffer —
    // only after the previous upload from it has completed (several async calls may be queued behind each other).
    const bool reuse = ctx->chunks_iters == iters && ctx->chunks_step == step && ctx->chunks_C == C;
    if (!reuse) {
        std::vector<ChunkDesc> chunks;
        plan_chunk_list(iters, step, C, kWave, ctx->batch_slots_max, ctx->batch_chunks_max, ctx->batch_first_chunks, chunks, ctx->chunk_batches);
        const size_t nch_all = chunks.size();
        ctx->chunks_iters = -1;                                   // invalid until the new list is staged
        if (ctx->upload_pending) { HIP_TRY(ctx, hipEventSynchronize(ctx->ev_upload)); ctx->upload_pending = false; }
        if (nch_all > ctx->chunks_cap) {
            // the device list may still be read by queued launches of earlier calls
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->plan_stream));
            free_dev(ctx->d_chunks);
            free_dev(ctx->d_chunks_alt);
            ctx->chunks_cap = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->d_chunks, sizeof(ChunkDesc) * nch_all));
            if (ctx->model == RRRMC_MODEL_SPARSE_PM1) HIP_TRY(ctx, hipMalloc(&ctx->d_chunks_alt, sizeof(ChunkDesc) * nch_all));      // (the fast Float64 sampler keeps one table)
            ctx->chunks_cap = nch_all;
        }
        if (nch_all > ctx->h_chunks_cap) {
            if (ctx->h_chunks) { (void)hipHostFree(ctx->h_chunks); ctx->h_chunks = nullptr; }
            ctx->h_chunks_cap = 0;
            HIP_TRY(ctx, hipHostMalloc(&ctx->h_chunks, sizeof(ChunkDesc) * nch_all));
            ctx->h_chunks_cap = nch_all;
        }
        if (nch_all) std::memcpy(ctx->h_chunks, chunks.data(), sizeof(ChunkDesc) * nch_all);
        ctx->chunks_n = nch_all;
    }
    *reuse_out = reuse;
    return RRRMC_OK;
}


// buffers of the random-site path for graphs beyond the LDS kernel, on first use
int32_t ensure_big_buffers(rrrmc_ctx* ctx)
{
    if (!ctx->big_mode || ctx->big_bufs_ready) return RRRMC_OK;
    const int64_t N = ctx->N, K = ctx->K;
    for (int i = 0; i < 2; ++i) {
        if (!ctx->d_nbrs[i]) HIP_TRY(ctx, hipMalloc(&ctx->d_nbrs[i], sizeof(uint32_t) * kMaxSlotsPerBatch * big_rec_words((int)K)));
        if (ctx->big_masks && !ctx->d_masks[i])
            HIP_TRY(ctx, hipMalloc(&ctx->d_masks[i], sizeof(uint32_t) * (ctx->big_cm ? (32 >> ctx->big_lgr) : 4) * ctx->batch_slots_max * ctx->G));
    }
    if (ctx->big_masks) {
        const int64_t S = 32 >> ctx->big_lgr;
        if (!ctx->d_bigimg) HIP_TRY(ctx, hipMalloc(&ctx->d_bigimg, sizeof(uint32_t) * ctx->G * S * ((N + S - 1) / S)));
        HIP_TRY(ctx, raise_lds_attr(reinterpret_cast<const void*>(ctx->big_cm ? big_applyc_for_K((int)K) : big_apply_for_K((int)K)), big_lds_bytes(N, ctx->big_lgr)));
    }
    ctx->big_bufs_ready = true;
    return RRRMC_OK;
}

// rrrmc_standard_mc_async for RRRMC_MODEL_SPARSE_PM1 (arguments checked for the model by the caller)
int32_t pm1_standard_mc_async(rrrmc_ctx* ctx, double beta, int64_t iters, int64_t step)
{
    int32_t rc = RRRMC_OK;
    ctx->colored_call = false;
    if (iters < 0) return fail(ctx, RRRMC_ERR_INVALID_ARG, "iters must be >= 0, given %lld", (long long)iters);
    if (step < 1) return fail(ctx, RRRMC_ERR_INVALID_ARG, "step must be >= 1, given %lld", (long long)step);
    if (std::isnan(beta)) return fail(ctx, RRRMC_ERR_INVALID_ARG, "beta is NaN");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { const int32_t rcb = ensure_big_buffers(ctx); if (rcb) return rcb; }
    const int64_t N = ctx->N, K = ctx->K;
    const int C = ctx->C;
    ctx->results_valid = false; ctx->last_call_wtm = false; ctx->last_call_eo = false;
    ctx->timing_valid = false;

    // thresholds for the classes with dE > 0: class n (unsatisfied bonds) has dE = 2 (K - 2n)
    SweepParams P{};
    const int NT = (int)(K + 1) / 2;
    for (int n = 0; n < NT; ++n) {
        const double dE = 2.0 * (double)(K - 2 * n);
        bool always;
        const uint64_t T = threshold64(std::exp(-beta * dE), &always);
        if (always) P.always_mask |= 1u << n;
        for (int plane = 0; plane < 64; ++plane) P.taum[plane * 4 + n] = ((T >> (63 - plane)) & 1ull) ? ~0u : 0u;
    }

    bool reuse = false;
    rc = prepare_chunk_list(ctx, iters, step, C, &reuse);
    if (rc) return rc;
    // The planner of this call may run beside the previous call's sweeps (two chunk tables, two buffer sets whose turn `plan_parity` tracks).
    // A call that fails half way leaves that bookkeeping between two states: whatever happens, an early return marks the chunk table stale, so
    // that the next call takes the fully ordered path (new upload, its planner behind everything queued so far) with either set.
    struct StaleOnError { rrrmc_ctx* c; bool ok; ~StaleOnError() { if (!ok) c->chunks_iters = -1; } } stale_guard{ctx, false};
    const int64_t nsamp = iters / step;
    const size_t nchunks = ctx->chunks_n;
    const std::vector<rrrmc_ctx::BatchDesc>& batches = ctx->chunk_batches;

    const size_t es_need = (size_t)(nsamp > 0 ? nsamp : 1) * ctx->Rpad;
    if (es_need > ctx->Es_cap) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));          // queued launches of earlier calls still write the old buffer
        free_dev(ctx->d_Es);
        ctx->Es_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_Es, sizeof(int32_t) * es_need));
        ctx->Es_cap = es_need;
    }

    std::vector<hipEvent_t>& EV = ctx->acc_mode ? ctx->ev_pool : ctx->ev_sweep;
    const size_t ebase = ctx->acc_mode ? ctx->ev_pool_used : 0;
    while (EV.size() < ebase + 2 * batches.size()) {
        hipEvent_t e;
        HIP_TRY(ctx, hipEventCreate(&e));
        EV.push_back(e);
    }
    while (ctx->ev_plan.size() < batches.size()) {
        hipEvent_t e;
        HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->ev_plan.push_back(e);
    }

    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_begin, st));
    // E = energy(X, C) at the start of every call (src/RRRMC.jl:95); accepted = 0 (:96)
    rc = run_energy(ctx, nullptr);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_acc, 0, sizeof(int64_t) * ctx->Rpad, st));
    if (!reuse) {
        if (nchunks) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_chunks, ctx->h_chunks, sizeof(ChunkDesc) * nchunks, hipMemcpyHostToDevice, st));
        if (nchunks) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_chunks_alt, ctx->h_chunks, sizeof(ChunkDesc) * nchunks, hipMemcpyHostToDevice, st));
        ctx->chunks_iters = iters; ctx->chunks_step = step; ctx->chunks_C = C;
        ctx->upload_pending = true;
    }
    // a new chunk table orders this call's planner behind its upload (and with it behind every earlier sweep); with the table of the previous call
    // the planner only waits for the sweep that last read the buffer set it is about to write (launch_plan)
    HIP_TRY(ctx, hipEventRecord(ctx->ev_upload, st));
    if (!reuse) HIP_TRY(ctx, hipStreamWaitEvent(ctx->plan_stream, ctx->ev_upload, 0));
    for (int i = 0; i < 2; ++i)
        if (!ctx->ev_set_free[i]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_set_free[i], hipEventDisableTiming));
    const int par = ctx->plan_parity;
    auto set_of = [par](int b) -> int { return (b + par) & 1; };
    // the chunk table of this call (the planner writes the chunks' level counts into it; the previous call's sweeps still read the other copy; the
    // call before that has finished before this call's first planner starts: it waits for a sweep of the previous call or for that call's own)
    ChunkDesc* const d_chunks = (ctx->call_parity ^= 1) ? ctx->d_chunks_alt : ctx->d_chunks;

    P.spins = ctx->d_spins;
    P.table = ctx->d_table;
    P.Es = ctx->d_Es;
    P.E_cur = ctx->d_E;
    P.acc_cur = ctx->d_acc;
    P.k0 = (uint32_t)ctx->seed;
    P.k1 = (uint32_t)(ctx->seed >> 32);
    P.group0 = ctx->replica0 / 32;
    P.gbase = ctx->it_done;
    P.N = (int)N; P.C = C; P.TS = ctx->TS; P.Rpad = (int)ctx->Rpad;
    sweep_fn fn = ctx->lds_mode ? sweep_for_K((int)K, ctx->sweep_mode) : nullptr;
    BigSweepParams PB{};
    BigMaskParams PM{};
    if (ctx->big_masks) {
        std::memcpy(PM.taum, P.taum, sizeof(PM.taum));
        PM.always_mask = P.always_mask; PM.k0 = P.k0; PM.k1 = P.k1; PM.group0 = P.group0; PM.gbase = P.gbase;
        PM.cap = (uint32_t)ctx->batch_slots_max;
        PM.lgr = ctx->big_cm ? ctx->big_lgr : -1;
    }
    if (ctx->big_mode) {
        PB.spins = ctx->d_spins; PB.Es = ctx->d_Es; PB.E_cur = ctx->d_E; PB.acc_cur = ctx->d_acc;
        std::memcpy(PB.taum, P.taum, sizeof(PB.taum));
        PB.always_mask = P.always_mask; PB.k0 = P.k0; PB.k1 = P.k1; PB.group0 = P.group0; PB.gbase = P.gbase;
        PB.N = (int)N; PB.Rpad = (int)ctx->Rpad;
    }
#ifdef RRRMC_STAMPS
    if (!g_stamps) HIP_TRY(ctx, hipMalloc(&g_stamps, sizeof(unsigned long long) * 16 * (65536 + 4096)));
    P.stamps = g_stamps;
#endif
    // The site-stream planner runs on its own stream, one batch ahead of the sweep: plan(b) may start as soon as the sweep that last read its
    // buffer set (sweep(b-2), or a sweep of the previous call) has finished; sweep(b) waits for plan(b).
    const int nb = (int)batches.size();
    auto launch_plan = [&](int b) -> int32_t {
        const rrrmc_ctx::BatchDesc& bt = batches[b];
        const int set = set_of(b);
        if (ctx->set_used[set]) HIP_TRY(ctx, hipStreamWaitEvent(ctx->plan_stream, ctx->ev_set_free[set], 0));
        if (ctx->big_mode) {
            hipLaunchKernelGGL(plan_big_for_K((int)K), dim3((unsigned)bt.n), dim3(kPlanThreads), ctx->plan_lds_bytes, ctx->plan_stream,
                               d_chunks + bt.first, ctx->d_slots[set], ctx->d_nbrs[set], ctx->d_vecs[set], ctx->d_A, ctx->d_J, (int)N, P.k0, P.k1, P.gbase,
                               ctx->big_masks ? ctx->big_lgr : -1);
            if (ctx->big_masks) {
                PM.chunks = d_chunks + bt.first; PM.slots = ctx->d_slots[set]; PM.masks = ctx->d_masks[set];
                hipLaunchKernelGGL(big_mask_for_K((int)K), dim3((unsigned)(bt.n * (kBigChunk / kBigMaskThreads)), (unsigned)ctx->G), dim3(kBigMaskThreads), 0,
                                   ctx->plan_stream, PM);
            }
        }
        else
            hipLaunchKernelGGL(plan_for_K((int)K), dim3((unsigned)bt.n), dim3(kPlanThreads), ctx->plan_lds_bytes, ctx->plan_stream,
                               d_chunks + bt.first, ctx->d_slots[set], ctx->d_vecs[set], ctx->d_A, (int)N, C, P.k0, P.k1, P.gbase,
                               sweep_rows_for_K((int)K) * kWave);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipEventRecord(ctx->ev_plan[b], ctx->plan_stream));
        return RRRMC_OK;
    };
    if (nb > 0) { rc = launch_plan(0); if (rc) return rc; }
    for (int b = 0; b < nb; ++b) {
        const rrrmc_ctx::BatchDesc& bt = batches[b];
        HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_plan[b], 0));
        P.chunks = d_chunks + bt.first;
        P.nchunks = (int)bt.n;
        P.sample0 = bt.sample0;
        const int set = set_of(b);
        P.slots = ctx->d_slots[set];
        P.vecs = ctx->d_vecs[set];
        HIP_TRY(ctx, hipEventRecord(EV[ebase + 2 * b], st));
        if (ctx->big_mode) {
            PB.chunks = P.chunks; PB.nchunks = P.nchunks; PB.sample0 = P.sample0; PB.slots = P.slots; PB.nbrs = ctx->d_nbrs[set]; PB.vecs = P.vecs;
            if (ctx->big_masks) {
                const int S = 32 >> ctx->big_lgr;
                hipLaunchKernelGGL(ctx->big_cm ? big_applyc_for_K((int)K) : big_apply_for_K((int)K), dim3((unsigned)(ctx->G * S)), dim3(kBigApplyThreads), big_lds_bytes(N, ctx->big_lgr), st, PB,
                                   ctx->d_masks[set], ctx->d_bigimg, (uint32_t)ctx->batch_slots_max, ctx->big_lgr);
                hipLaunchKernelGGL(big_merge_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)ctx->G), dim3(256), 0, st, ctx->d_spins, ctx->d_bigimg,
                                   (int)N, ctx->big_lgr);
            }
            else
                hipLaunchKernelGGL(big_sweep_for_K((int)K), dim3((unsigned)ctx->G), dim3(kBigThreads), 0, st, PB);
        } else {
            hipLaunchKernelGGL(fn, dim3((unsigned)ctx->G), dim3(kSweepThreads), ctx->lds_bytes, st, P);
        }
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipEventRecord(EV[ebase + 2 * b + 1], st));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_set_free[set], st));
        ctx->set_used[set] = true;
        ctx->plan_parity = (par + b + 1) & 1;          // the set the next batch (of this call or the next) takes: advanced with the sets, batch by batch
        // the next plan is enqueued AFTER this sweep so that the sweep's workgroups (one per CU, most of the LDS)
        // are placed first and the planner's small workgroups fill in beside them
        if (b + 1 < nb) { rc = launch_plan(b + 1); if (rc) return rc; }
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev_end, st));
    stale_guard.ok = true;
    ctx->sweep_launches = (int)batches.size();
    ctx->last_ev_base = ebase;
    ctx->last_ev_pool = ctx->acc_mode;
    if (ctx->acc_mode) ctx->ev_pool_used += 2 * batches.size();
    ctx->nsamp = nsamp;
    ctx->it_done += (uint64_t)iters;
    ctx->results_valid = true;
    ctx->timing_valid = true;
    if (ctx->debug_checks) { rc = debug_check_pm1(ctx); if (rc) return rc; }
    return RRRMC_OK;
}
