// Host side of the opt-in FAST standardMC on GraphRRGNormal / GraphEANormal (spf_fast_kernels.hpp).
// Included by rrrmc_hip.hip inside its anonymous namespace, after prepare_chunk_list.

typedef void (*fast_fn)(FastParams);
fast_fn spf_fast_for_K(int K)
{
    switch (K) { case 1: return spf_fast_kernel<1>; case 2: return spf_fast_kernel<2>; case 3: return spf_fast_kernel<3>; case 4: return spf_fast_kernel<4>;
                 default: return nullptr; }
}

// delta_energy of site x for the unsatisfied-bond pattern u (bit k = bond k): 2 * sum_k (u_k ? -|J| : +|J|), k in order — the same
// expression as orc_spf_pattern_delta of the oracle
inline double fast_pattern_delta(const double* Jrow, int K, uint32_t u)
{
    double s = 0.0;
    for (int k = 0; k < K; ++k) { const double a = std::fabs(Jrow[k]); s += ((u >> k) & 1u) ? -a : a; }
    return 2.0 * s;
}

// geometry and beta-independent tables (on first use after rrrmc_set_graph_f64)
int32_t fast_setup(rrrmc_ctx* ctx)
{
    const int64_t N = ctx->N, K = ctx->K;
    if (K > kFastMaxK) return fail(ctx, RRRMC_ERR_UNSUPPORTED, "the fast Float64 mode covers K <= %d (2^(K-1) thresholds per site), given K=%lld", kFastMaxK, (long long)K);
    if (N > 8192) return fail(ctx, RRRMC_ERR_UNSUPPORTED, "the fast Float64 mode keeps the replica group's state in LDS with 16-bit byte offsets: N <= 8192, given %lld", (long long)N);
    if (ctx->pff_ready) return RRRMC_OK;
    if (ctx->h_Jf.size() != (size_t)(N * K) || ctx->h_A.size() != (size_t)(N * K)) return fail(ctx, RRRMC_ERR_STATE, "rrrmc_set_graph_f64 has not been called");
    const int TS = 4;
    const int NT = 1 << (K - 1);
    // longest chunk (multiple of 64, at most kMaxChunk and 8 N) whose buffers fit LDS next to the state
    int C = 0;
    for (int c = kMaxChunk; c >= 64; c -= 64)
        if (fast_lds_bytes(N, (int)K, c) <= (size_t)kLdsLimit && plan_lds_bytes(N, (int)K, c) <= (size_t)kLdsLimit) { C = c; break; }
    if (C == 0) return fail(ctx, RRRMC_ERR_UNSUPPORTED, "N=%lld K=%lld: the state does not fit the 160 KiB LDS", (long long)N, (long long)K);
    if ((int64_t)C > 8 * N) C = (int)std::max<int64_t>(64, (8 * N) / 64 * 64);
    ctx->pff_C = C;
    ctx->pff_lds = fast_lds_bytes(N, (int)K, C);
    ctx->pff_plan_lds = plan_lds_bytes(N, (int)K, C);
    std::vector<uint16_t> table((size_t)N * TS, 0);
    std::vector<uint32_t> bond_off;
    std::vector<double> bond_w;
    for (int64_t x = 0; x < N; ++x)
        for (int64_t k = 0; k < K; ++k) {
            const double j = ctx->h_Jf[(size_t)(x * K + k)];
            const int64_t y = ctx->h_A[(size_t)(x * K + k)];
            const uint32_t yoff = (uint32_t)(4 * (2 * y + (j < 0 ? 1 : 0)));      // byte offset of word 2y (s_y) or 2y + 1 (~s_y)
            table[(size_t)(x * TS + k)] = (uint16_t)yoff;
            if (y > x) { bond_off.push_back((uint32_t)(8 * x) | (yoff << 16)); bond_w.push_back(std::fabs(j)); }      // every bond once (two entries for a double bond)
        }
    // the energy phase: blocks of 32 bonds, 32 half-waves -> a multiple of 32 blocks; padding bonds have weight 0
    const int nblk = (int)(((bond_off.size() + 31) / 32 + 31) / 32 * 32);
    bond_off.resize((size_t)nblk * 32, 0u);
    bond_w.resize((size_t)nblk * 32, 0.0);
    std::vector<double> etab((size_t)nblk * 4 * 256);
    for (int b = 0; b < nblk; ++b)
        for (int q = 0; q < 4; ++q)
            for (int v = 0; v < 256; ++v) {
                double sum = 0.0;
                for (int i = 0; i < 8; ++i) { const double w = bond_w[(size_t)b * 32 + q * 8 + i]; sum += ((v >> i) & 1) ? w : -w; }
                etab[((size_t)b * 4 + q) * 256 + v] = sum;
            }
    ctx->pff_nblk = nblk;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // (a failure midway leaves pff_ready false: the next call starts over, so whatever an earlier attempt allocated is released first)
    free_dev(ctx->pff_table); free_dev(ctx->pff_absJ); free_dev(ctx->pff_bond_off); free_dev(ctx->pff_thr_hi); free_dev(ctx->pff_thr_lo); free_dev(ctx->pff_flags);
    HIP_TRY(ctx, hipMalloc(&ctx->pff_table, sizeof(uint16_t) * table.size()));
    HIP_TRY(ctx, hipMalloc(&ctx->pff_absJ, sizeof(double) * etab.size()));
    HIP_TRY(ctx, hipMalloc(&ctx->pff_bond_off, sizeof(uint32_t) * bond_off.size()));
    HIP_TRY(ctx, hipMalloc(&ctx->pff_thr_hi, sizeof(uint32_t) * (size_t)N * NT));
    HIP_TRY(ctx, hipMalloc(&ctx->pff_thr_lo, sizeof(uint32_t) * (size_t)N * NT));
    HIP_TRY(ctx, hipMalloc(&ctx->pff_flags, sizeof(uint32_t) * (size_t)N));
    HIP_TRY(ctx, hipMemcpy(ctx->pff_table, table.data(), sizeof(uint16_t) * table.size(), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->pff_absJ, etab.data(), sizeof(double) * etab.size(), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->pff_bond_off, bond_off.data(), sizeof(uint32_t) * bond_off.size(), hipMemcpyHostToDevice));
    for (int i = 0; i < 2; ++i) {
        if (!ctx->d_slots[i]) HIP_TRY(ctx, hipMalloc(&ctx->d_slots[i], sizeof(uint32_t) * kMaxSlotsPerBatch));
        if (!ctx->d_vecs[i]) HIP_TRY(ctx, hipMalloc(&ctx->d_vecs[i], sizeof(uint32_t) * kMaxSlotsPerBatch));
    }
    if (!ctx->plan_stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->plan_stream, hipStreamNonBlocking));
    if (!ctx->ev_upload) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_upload, hipEventDisableTiming));
    HIP_TRY(ctx, raise_lds_attr(reinterpret_cast<const void*>(spf_fast_for_K((int)K)), ctx->pff_lds));
    HIP_TRY(ctx, raise_lds_attr(reinterpret_cast<const void*>(plan_for_K((int)K)), ctx->pff_plan_lds));
    ctx->pff_ready = true;
    ctx->pff_beta_valid = false;
    return RRRMC_OK;
}

// thresholds of every (site, pattern pair) for this beta: T = ceil(exp(x) 2^64) with x = -beta dE < 0 on the tested side (libm exp,
// as the oracle), both sides accepted when x >= 0 on both (dE = 0)
int32_t fast_thresholds(rrrmc_ctx* ctx, double beta)
{
    if (ctx->pff_beta_valid && ctx->pff_beta == beta) return RRRMC_OK;
    const int64_t N = ctx->N, K = ctx->K;
    const int NT = 1 << (K - 1);
    std::vector<uint32_t> hi((size_t)N * NT), lo((size_t)N * NT), fl((size_t)N);
    for (int64_t x = 0; x < N; ++x) {
        uint32_t f = 0u;
        for (int c = 0; c < NT; ++c) {
            const uint32_t u0 = (uint32_t)c << 1;                          // the pair's pattern with u_1 = 0: u_{k+1} = bit k-1 of c
            const uint32_t u1 = (~u0) & ((1u << K) - 1u);                  // its complement (u_1 = 1)
            const double x0 = -beta * fast_pattern_delta(&ctx->h_Jf[(size_t)(x * K)], (int)K, u0);
            const double x1 = -beta * fast_pattern_delta(&ctx->h_Jf[(size_t)(x * K)], (int)K, u1);
            uint64_t T = ~0ull;
            bool always = true;
            if (x0 < 0) { T = threshold64(std::exp(x0), &always); }                       // tested side: u_1 = 0
            else if (x1 < 0) { T = threshold64(std::exp(x1), &always); f |= 1u << (8 + c); }   // tested side: u_1 = 1
            if (always) f |= 1u << c;
            hi[(size_t)(x * NT + c)] = (uint32_t)(T >> 32);
            lo[(size_t)(x * NT + c)] = (uint32_t)T;
        }
        fl[(size_t)x] = f;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));          // queued launches of earlier calls read the old thresholds
    HIP_TRY(ctx, hipMemcpy(ctx->pff_thr_hi, hi.data(), sizeof(uint32_t) * hi.size(), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->pff_thr_lo, lo.data(), sizeof(uint32_t) * lo.size(), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->pff_flags, fl.data(), sizeof(uint32_t) * fl.size(), hipMemcpyHostToDevice));
    ctx->pff_beta = beta;
    ctx->pff_beta_valid = true;
    return RRRMC_OK;
}

int32_t spf_fast_mc_async(rrrmc_ctx* ctx, double beta, int64_t iters, int64_t step)
{
    if (iters < 0) return fail(ctx, RRRMC_ERR_INVALID_ARG, "iters must be >= 0, given %lld", (long long)iters);
    if (step < 1) return fail(ctx, RRRMC_ERR_INVALID_ARG, "step must be >= 1, given %lld", (long long)step);
    if (std::isnan(beta)) return fail(ctx, RRRMC_ERR_INVALID_ARG, "beta is NaN");
    int32_t rc = fast_setup(ctx);
    if (rc) return rc;
    rc = fast_thresholds(ctx, beta);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->results_valid = false; ctx->last_call_wtm = false; ctx->last_call_eo = false; ctx->last_call_rrr = false;
    ctx->timing_valid = false;
    ctx->pf_lf_live = false; ctx->std_cache_live = false;          // the stored fields / tracked energy no longer describe the configuration
    const int64_t N = ctx->N, K = ctx->K;
    const int C = ctx->pff_C;
    bool reuse = false;
    rc = prepare_chunk_list(ctx, iters, step, C, &reuse);
    if (rc) return rc;
    const int64_t nsamp = iters / step;
    const size_t nchunks = ctx->chunks_n;
    const std::vector<rrrmc_ctx::BatchDesc>& batches = ctx->chunk_batches;
    const size_t es_need = (size_t)(nsamp > 0 ? nsamp : 1) * ctx->Rpad;
    if (es_need > ctx->sk_Es_cap) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        free_dev(ctx->sk_Es);
        ctx->sk_Es_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->sk_Es, sizeof(double) * es_need));
        ctx->sk_Es_cap = es_need;
    }
    while (ctx->ev_sweep.size() < 2 * batches.size() + 2) {
        hipEvent_t e;
        HIP_TRY(ctx, hipEventCreate(&e));
        ctx->ev_sweep.push_back(e);
    }
    while (ctx->ev_plan.size() < batches.size()) {
        hipEvent_t e;
        HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->ev_plan.push_back(e);
    }
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_begin, st));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_acc, 0, sizeof(int64_t) * ctx->Rpad, st));
    if (!reuse) {
        if (nchunks) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_chunks, ctx->h_chunks, sizeof(ChunkDesc) * nchunks, hipMemcpyHostToDevice, st));
        ctx->chunks_iters = iters; ctx->chunks_step = step; ctx->chunks_C = C;
        ctx->upload_pending = true;
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev_upload, st));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->plan_stream, ctx->ev_upload, 0));

    FastParams P{};
    P.spins32 = reinterpret_cast<uint32_t*>(ctx->pf_spins);
    P.table = ctx->pff_table; P.thr_hi = ctx->pff_thr_hi; P.thr_lo = ctx->pff_thr_lo; P.flags = ctx->pff_flags; P.etab = ctx->pff_absJ; P.bond_off = ctx->pff_bond_off; P.nblk = ctx->pff_nblk;
    P.Es = ctx->sk_Es; P.acc_cur = ctx->d_acc;
    P.k0 = (uint32_t)ctx->seed; P.k1 = (uint32_t)(ctx->seed >> 32);
    P.group0 = ctx->replica0 / 32;
    P.gbase = ctx->it_done;
    P.N = (int)N; P.C = C; P.TS = 4; P.Rpad = (int)ctx->Rpad;
    fast_fn fn = spf_fast_for_K((int)K);
    const unsigned G = (unsigned)(ctx->Rpad / 32);
    const int nb = (int)batches.size();
    auto launch_plan = [&](int b) -> int32_t {
        const rrrmc_ctx::BatchDesc& bt = batches[b];
        if (b >= 2) HIP_TRY(ctx, hipStreamWaitEvent(ctx->plan_stream, ctx->ev_sweep[2 * (b - 2) + 1], 0));
        hipLaunchKernelGGL(plan_for_K((int)K), dim3((unsigned)bt.n), dim3(kPlanThreads), ctx->pff_plan_lds, ctx->plan_stream,
                           ctx->d_chunks + bt.first, ctx->d_slots[b & 1], ctx->d_vecs[b & 1], ctx->d_A, (int)N, C, P.k0, P.k1, P.gbase,
                           kFastRows * kWave);      // the longest batch spf_fast_kernel's consumer takes
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipEventRecord(ctx->ev_plan[b], ctx->plan_stream));
        return RRRMC_OK;
    };
    if (nb > 0) { rc = launch_plan(0); if (rc) return rc; }
    for (int b = 0; b < nb; ++b) {
        const rrrmc_ctx::BatchDesc& bt = batches[b];
        HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_plan[b], 0));
        P.chunks = ctx->d_chunks + bt.first;
        P.nchunks = (int)bt.n;
        P.sample0 = bt.sample0;
        P.slots = ctx->d_slots[b & 1];
        P.vecs = ctx->d_vecs[b & 1];
        HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[2 * b], st));
        hipLaunchKernelGGL(fn, dim3(G), dim3(kSweepThreads), ctx->pff_lds, st, P);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipEventRecord(ctx->ev_sweep[2 * b + 1], st));
        if (b + 1 < nb) { rc = launch_plan(b + 1); if (rc) return rc; }
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev_end, st));
    ctx->sweep_launches = nb;
    ctx->last_ev_base = 0; ctx->last_ev_pool = false;
    ctx->nsamp = nsamp;
    ctx->it_done += (uint64_t)iters;
    ctx->results_valid = true;
    ctx->timing_valid = true;
    return RRRMC_OK;
}
