"""GPU edge cases of the C ABI for the headline path: empty / degenerate runs, several plan batches in one call, misuse."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_empty_and_degenerate_runs(pkg, oracle):
    seed = 1
    X = pkg.GraphRRG(64, 3, seed=seed)
    with pkg.Engine(X, 5) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc = eng.standard_mc(1.0, 0, 10)                     # iters = 0
        assert Es.shape == (5, 0) and (acc == 0).all() and eng.get_config() == C0 and eng.iterations_done() == 0
        Es, acc = eng.standard_mc(1.0, 7, 10)                     # step > iters: moves, but no sample
        assert Es.shape == (5, 0) and eng.iterations_done() == 7
        ref = oracle.standard_mc_sparse_batch(X.A, X.J.astype(np.int32), 1.0, 7, 10, seed, C0.s)
        assert (eng.get_config().s == ref[1]).all() and (acc == ref[2]).all()
        Es, acc = eng.standard_mc(1.0, 1, 1)                      # a single iteration, sampled before its move
        ref1 = oracle.standard_mc_sparse_batch(X.A, X.J.astype(np.int32), 1.0, 1, 1, seed, ref[1], it0=7)
        assert (Es == ref1[0]).all() and (eng.get_config().s == ref1[1]).all()


def test_two_spin_model(pkg, oracle):
    """GraphTwoSpin (src/graphs/TwoSpin.jl:26-41) as the N = 2, K = 1 sparse model: E = -J s1 s2."""
    A = np.array([[1], [0]], np.int32)
    J = np.array([[1], [1]], np.int8)
    X = pkg.GraphRRG.from_AJ(A, J)
    with pkg.Engine(X, 33) as eng:
        eng.seed(3)
        eng.init_spins_random()
        C0 = eng.get_config()
        E0 = eng.energy()
        Es, acc = eng.standard_mc(0.7, 500, 1)
        C1 = eng.get_config()
    b = C0.bits().astype(np.int64)
    assert (E0 == -(2 * b[:, 0] - 1) * (2 * b[:, 1] - 1)).all() and set(np.unique(Es)) <= {-1, 1}
    ref = oracle.standard_mc_sparse_batch(A, J.astype(np.int32), 0.7, 500, 1, 3, C0.s)
    assert (Es == ref[0]).all() and (C1.s == ref[1]).all() and (acc == ref[2]).all()


def test_three_spin_model(pkg, oracle):
    """GraphThreeSpin (src/graphs/ThreeSpin.jl:26-47) as the N = 3, K = 2 triangle: closed-form energies, every sampler against the oracle."""
    A = np.array([[1, 2], [0, 2], [0, 1]], np.int32)
    J = np.ones((3, 2), np.int8)
    X = pkg.GraphRRG.from_AJ(A, J)
    J32 = J.astype(np.int32)
    with pkg.Engine(X, 40) as eng:
        eng.seed(5)
        eng.init_spins_random()
        C0 = eng.get_config()
        E0 = eng.energy()
        Es, acc = eng.standard_mc(2.0, 2000, 10)
        C1 = eng.get_config()
        eng.seed(5); eng.set_config(C0)
        Er, ar, st = eng.rrr_mc(2.0, 2000, 10)
        Cr = eng.get_config()
        eng.seed(5); eng.set_config(C0)
        Ew, mw, tw = eng.wtm_mc(2.0, 50, 1.0)
        Cw = eng.get_config()
        eng.seed(5); eng.set_config(C0)
        Ee, Emin, Cmin, itmin = eng.extremal_opt(1.3, 500, 5)
    b = 2 * C0.bits().astype(np.int64) - 1
    assert (E0 == -(b[:, 0] * b[:, 1] + b[:, 1] * b[:, 2] + b[:, 2] * b[:, 0])).all()
    assert set(np.unique(Es)) <= {-3, 1} and (Emin == -3).all()
    ref = oracle.standard_mc_sparse_batch(A, J32, 2.0, 2000, 10, 5, C0.s)
    assert (Es == ref[0]).all() and (C1.s == ref[1]).all() and (acc == ref[2]).all()
    for r in range(40):
        rr = oracle.rrr_sparse(A, J32, 2.0, 2000, 10, 5, C0.s[r], replica=r)
        assert (Er[r] == rr[0]).all() and (Cr.s[r] == rr[1]).all() and ar[r] == rr[2] and st[r] == rr[3]
        w = oracle.wtm_mc_sparse(A, J32, 2.0, 50, 1.0, 5, C0.s[r], replica=r)
        assert (Ew[r] == w[0]).all() and (Cw.s[r] == w[1]).all() and mw[r] == w[2] and tw[r] == w[3]
        e = oracle.extremal_opt_sparse(A, J32, 1.3, 500, 5, 5, C0.s[r], replica=r)
        assert (Ee[r] == e[0]).all() and (Cmin.s[r] == e[3]).all() and itmin[r] == e[4]


def test_several_plan_batches_in_one_call(pkg, oracle):
    """More iterations than one planner batch (2^22 slots), with a step that straddles the batch boundary."""
    seed, N, R = 77, 64, 32
    iters, step = (1 << 22) + 12345, 100003
    X = pkg.GraphRRG(N, 3, seed=seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc = eng.standard_mc(1.0, iters, step)
        C1 = eng.get_config()
        assert eng.last_timing()[2] == 2                              # two sweep launches
    ref = oracle.standard_mc_sparse_batch(X.A, X.J.astype(np.int32), 1.0, iters, step, seed, C0.s)
    assert Es.shape == (R, iters // step)
    assert (Es == ref[0]).all() and (C1.s == ref[1]).all() and (acc == ref[2]).all()


def test_misuse_is_reported_not_crashed(pkg):
    L = pkg.lib()
    X = pkg.GraphRRG(64, 3, seed=5)
    ctx = C.c_void_p()
    assert L.rrrmc_ctx_create(C.byref(ctx), 1, 64, 3, 8, 0, 0) == 0
    try:
        Es = np.zeros(8 * 10, np.int64)
        acc = np.zeros(8, np.int64)
        # sampling before the graph / the configuration exist
        assert L.rrrmc_standard_mc(ctx, 1.0, 100, 10, Es.ctypes.data, acc.ctypes.data) == 2
        assert b"rrrmc_set_graph" in L.rrrmc_last_error(ctx)
        assert L.rrrmc_set_graph(ctx, X.A, X.J) == 0
        assert L.rrrmc_standard_mc(ctx, 1.0, 100, 10, Es.ctypes.data, acc.ctypes.data) == 2
        assert L.rrrmc_seed(ctx, 5) == 0 and L.rrrmc_init_spins_random(ctx) == 0
        # bad arguments
        assert L.rrrmc_standard_mc(ctx, float("nan"), 100, 10, Es.ctypes.data, acc.ctypes.data) == 1
        assert L.rrrmc_standard_mc(ctx, 1.0, -1, 10, Es.ctypes.data, acc.ctypes.data) == 1
        assert L.rrrmc_standard_mc(ctx, 1.0, 100, 0, Es.ctypes.data, acc.ctypes.data) == 1
        assert L.rrrmc_fetch_results_f64(ctx, None, None) == 2          # integer model: wrong entry point
        bad = X.A.copy()
        bad[0, 0] = 64
        assert L.rrrmc_set_graph(ctx, bad, X.J) == 1                     # neighbour out of range
        # the context is still usable afterwards
        assert L.rrrmc_set_graph(ctx, X.A, X.J) == 0
        assert L.rrrmc_standard_mc(ctx, 1.0, 100, 10, Es.ctypes.data, acc.ctypes.data) == 0
    finally:
        L.rrrmc_ctx_destroy(ctx)
    # out-of-range sizes at creation
    assert L.rrrmc_ctx_create(C.byref(ctx), 1, 64, 9, 8, 0, 0) == 3      # K > 7
    assert L.rrrmc_ctx_create(C.byref(ctx), 1, 0, 3, 8, 0, 0) == 1
    assert L.rrrmc_ctx_create(C.byref(ctx), 1, 64, 3, 8, 99, 0) == 1     # no such device
    assert L.rrrmc_ctx_create(C.byref(ctx), 42, 64, 3, 8, 0, 0) == 3     # unknown model


def test_contexts_of_different_sizes_coexist(pkg, oracle):
    """Kernel attributes (the dynamic-LDS bound) belong to the kernel, not to a context: a big context must keep working after a
    small one of the same kernel has been created, and the other way round, with their calls interleaved."""
    seed = 515
    Xbig, Xsmall = pkg.GraphRRG(4096, 3, seed=seed), pkg.GraphRRG(64, 3, seed=seed)
    Qbig = pkg.GraphQuant(pkg.GraphRRG(256, 3, seed=seed), 16, 0.5, 2.0)
    Qsmall = pkg.GraphQuant(pkg.GraphRRG(10, 3, seed=seed), 4, 0.5, 2.0)
    with pkg.Engine(Xbig, 32) as a:
        a.seed(seed); a.init_spins_random()
        Ca = a.get_config()
        with pkg.Engine(Xsmall, 32) as b, pkg.Engine(Qbig, 4) as qa:
            b.seed(seed); b.init_spins_random()
            qa.seed(seed); qa.init_spins_random()
            Cb, Cqa = b.get_config(), qa.get_config()
            with pkg.Engine(Qsmall, 4) as qb:
                qb.seed(seed); qb.init_spins_random()
                Cqb = qb.get_config()
                Ea, _ = a.standard_mc(1.0, 8192, 4096)
                Eb, _ = b.standard_mc(1.0, 8192, 4096)
                Eqa = qa.rrr_mc(2.0, 2000, 100)[0]
                Eqb = qb.rrr_mc(2.0, 2000, 100)[0]
                Ea2, _ = a.standard_mc(1.0, 4096, 4096)
                Eqa2 = qa.rrr_mc(2.0, 1000, 100)[0]
    ra = oracle.standard_mc_sparse_batch(Xbig.A, Xbig.J.astype(np.int32), 1.0, 8192, 4096, seed, Ca.s)
    rb = oracle.standard_mc_sparse_batch(Xsmall.A, Xsmall.J.astype(np.int32), 1.0, 8192, 4096, seed, Cb.s)
    assert (Ea == ra[0]).all() and (Eb == rb[0]).all()
    ra2 = oracle.standard_mc_sparse_batch(Xbig.A, Xbig.J.astype(np.int32), 1.0, 4096, 4096, seed, ra[1], it0=8192)
    assert (Ea2 == ra2[0]).all()
    for r in range(4):
        q = oracle.rrr_mc_quant(Qbig.A, Qbig.J.astype(np.int32), Qbig.M, Qbig.fourK, 2.0, 2000, 100, seed, Cqa.s[r], replica=r)
        assert (Eqa[r] == q[0]).all()
        q2 = oracle.rrr_mc_quant(Qbig.A, Qbig.J.astype(np.int32), Qbig.M, Qbig.fourK, 2.0, 1000, 100, seed, q[1], it0=2000, replica=r)
        assert (Eqa2[r] == q2[0]).all()
        qs = oracle.rrr_mc_quant(Qsmall.A, Qsmall.J.astype(np.int32), Qsmall.M, Qsmall.fourK, 2.0, 2000, 100, seed, Cqb.s[r], replica=r)
        assert (Eqb[r] == qs[0]).all()


def test_create_destroy_cycles(pkg):
    X = pkg.GraphRRG(256, 3, seed=9)
    for k in range(40):
        with pkg.Engine(X, 64) as eng:
            eng.seed(k + 1)
            eng.init_spins_random()
            eng.standard_mc(1.0, 2000, 500)


def test_full_length_config2_chain(pkg, oracle):
    """BASELINE config 2 at its full chain length (2^24 iterations, a sample every 2^12) for one replica group: four planner
    batches, 4096 samples, tally-counter flushes — every energy sample and the final spins equal the oracle's."""
    seed, N, R = 0x5EED, 4096, 32
    iters, step = 1 << 24, 1 << 12
    X = pkg.GraphRRG(N, 3, seed=seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc = eng.standard_mc(1.0, iters, step)
        C1 = eng.get_config()
    ref = oracle.standard_mc_sparse_batch(X.A, X.J.astype(np.int32), 1.0, iters, step, seed, C0.s)
    assert Es.shape == (R, iters // step)
    assert (Es == ref[0]).all() and (C1.s == ref[1]).all() and (acc == ref[2]).all()


def test_full_width_config2_subset_against_oracle(pkg, oracle):
    """BASELINE config 2 at its full width (8192 replicas = 256 workgroups, one per CU): three whole replica groups — the first,
    one in the middle, the last — are checked against the oracle, and the acceptance rate of all replicas is sane."""
    seed, N, R = 0x5EED, 4096, 8192
    iters, step = 1 << 18, 1 << 12
    X = pkg.GraphRRG(N, 3, seed=seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc = eng.standard_mc(1.0, iters, step)
        C1 = eng.get_config()
        E1 = eng.energy()
    A, J = X.A, X.J.astype(np.int32)
    for g in (0, 100, 255):
        sl = slice(32 * g, 32 * g + 32)
        ref = oracle.standard_mc_sparse_batch(A, J, 1.0, iters, step, seed, C0.s[sl], replica0=32 * g)
        assert (Es[sl] == ref[0]).all() and (C1.s[sl] == ref[1]).all() and (acc[sl] == ref[2]).all()
        for r in (32 * g, 32 * g + 31):
            assert E1[r] == oracle.sparse_energy(A, J, C1.s[r])
    a = acc / iters
    assert 0.07 < a.mean() < 0.12 and a.std() < 0.01          # beta = 1, K = 3: about 8-10 % of the attempts are accepted
    assert E1.mean() / N < -1.05                                # well below the random-configuration energy 0


def test_gpu_replicas_follow_the_boltzmann_law(pkg, oracle):
    """The reference's stationarity check (exact `truep`, RRRMC.jl:528-543, on a tiny system): after many sweeps the 8192 GPU
    replicas of a 6-spin GraphRRG are distributed like exp(-beta E) / Z.  (All replicas attempt the same sites, so they are not
    independent samples: the chi-square bound is generous.)"""
    seed, N, beta, R = 11, 6, 0.6, 8192
    X = pkg.GraphRRG(N, 3, seed=seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        eng.standard_mc(beta, 6000, 6000)
        C = eng.get_config()
    counts = np.bincount(C.s[:, 0].astype(np.int64), minlength=2 ** N).astype(np.float64)
    A, J = X.A, X.J.astype(np.int32)
    Es = np.array([oracle.sparse_energy(A, J, np.array([c], np.uint64)) for c in range(2 ** N)], np.float64)
    p = np.exp(-beta * Es)
    p /= p.sum()
    chi2 = float((((counts - R * p) ** 2) / (R * p)).sum())
    assert chi2 < 4 * 2 ** N, chi2                                 # 63 degrees of freedom
    assert abs((counts / R) @ Es - p @ Es) < 0.15                 # mean energy


def test_result_arrays_of_the_caller_must_be_contiguous_and_writeable(pkg):
    """Engine.standard_mc(out=(Es, accepted)) hands raw pointers to the C ABI, which writes R * nsamp contiguous values: a strided view of a
    larger reusable buffer or a read-only array is refused before the call, a fitting slice along the first axis is filled."""
    X = pkg.GraphRRG(64, 3, seed=5)
    with pkg.Engine(X, 32) as eng:
        eng.seed(5)
        eng.init_spins_random()
        big = np.zeros((32, 16), np.int64)
        acc = np.zeros(32, np.int64)
        with pytest.raises(ValueError):
            eng.standard_mc(1.0, 800, 100, out=(big[:, :8], acc))
        ro = np.zeros((32, 8), np.int64)
        ro.setflags(write=False)
        with pytest.raises(ValueError):
            eng.standard_mc(1.0, 800, 100, out=(ro, acc))
        ok = np.zeros((64, 8), np.int64)
        Es, a = eng.standard_mc(1.0, 800, 100, out=(ok[:32], acc))
        assert Es.shape == (32, 8) and (ok[32:] == 0).all() and a.sum() > 0


@pytest.mark.parametrize("N,K,R,iters,step", [(4096, 3, 1024, 1 << 16, 4096), (4096, 3, 96, (1 << 17) + 77, 1000), (216, 6, 64, 50000, 777)])
def test_queued_calls_of_one_shape_equal_calls_made_one_by_one(pkg, oracle, N, K, R, iters, step):
    """Back-to-back asynchronous calls of the same (iters, step): the chunk table stays on the device and the planner of call s + 1 runs beside
    the sweep of call s (its own stream, its own copy of the table and the other set of plan buffers: the planner writes per-chunk level counts
    that the running sweep reads).  Five queued calls must leave exactly what five calls with a sync between them leave, and the last call's
    samples are the oracle's for that piece of the chain."""
    seed, beta, calls = 424242 + N + K, 0.8, 5
    X = pkg.GraphRRG(N, K, seed=seed) if K == 3 else pkg.GraphEA(6, 3, seed=seed)
    outs = []
    for queued in (True, False):
        with pkg.Engine(X, R) as eng:
            eng.seed(seed)
            eng.init_spins_random()
            C0 = eng.get_config().s.copy()
            before_last = None
            for c in range(calls):
                if c == calls - 1 and not queued:
                    before_last = eng.get_config().s.copy()
                eng.standard_mc_async(beta, iters, step)
                if not queued:
                    eng.sync()
            eng.sync()
            Es, acc = eng.fetch_results()
            outs.append((C0, Es, acc, eng.get_config().s.copy(), before_last))
    for u, v in zip(outs[0][:4], outs[1][:4]):
        assert (u == v).all()
    C_before = outs[1][4]
    for r in (0, R // 2, R - 1):
        ref = oracle.standard_mc_sparse(X.A, X.J.astype(np.int32), beta, iters, step, seed, C_before[r], replica=r, it0=(calls - 1) * iters, form="rrg" if K == 3 else "ea")
        assert (outs[0][1][r] == ref[0]).all() and (outs[0][3][r] == ref[1]).all() and outs[0][2][r] == ref[2]


def test_queued_calls_of_alternating_shapes(pkg, oracle):
    """Queued calls whose (iters, step) changes now and then: a new shape uploads both copies of the chunk table behind everything queued so far and
    its planner waits for that upload; calls that repeat the previous shape overlap their planner with the previous sweep again.  Queued and
    synchronised runs must agree after every pattern, several plan batches per call included."""
    seed, beta, R, N = 777001, 0.7, 256, 4096
    X = pkg.GraphRRG(N, 3, seed=seed)
    A_, B_, C_ = (40000, 1000), (1 << 18, 1 << 14), (123457, 99)
    pattern = [A_, A_, B_, B_, B_, A_, C_, C_, A_, A_, A_, B_]
    outs = []
    for queued in (True, False):
        with pkg.Engine(X, R) as eng:
            eng.seed(seed)
            eng.init_spins_random()
            for iters, step in pattern:
                eng.standard_mc_async(beta, iters, step)
                if not queued:
                    eng.sync()
            eng.sync()
            Es, acc = eng.fetch_results()
            outs.append((Es, acc, eng.get_config().s.copy(), eng.energy()))
    for u, v in zip(outs[0], outs[1]):
        assert (u == v).all()


def _queued_vs_synchronised(pkg, X, R, seed, beta, pattern, engine_kw=None, between=None):
    """the calls of `pattern` queued back to back, and with a sync after each: (Es, accepted, configuration, energy, configuration before the last call)"""
    outs = []
    for queued in (True, False):
        with pkg.Engine(X, R, **(engine_kw or {})) as eng:
            eng.seed(seed)
            eng.init_spins_random()
            if between is not None:
                between(eng, None)
            before_last = None
            for c, (iters, step) in enumerate(pattern):
                if c == len(pattern) - 1 and not queued:
                    before_last = eng.get_config().s.copy()
                if between is not None and c > 0:
                    between(eng, c)
                    if not queued:
                        eng.sync()
                eng.standard_mc_async(beta, iters, step)
                if not queued:
                    eng.sync()
            eng.sync()
            Es, acc = eng.fetch_results()
            outs.append((Es, acc, eng.get_config().s.copy(), eng.energy(), before_last))
    for u, v in zip(outs[0][:4], outs[1][:4]):
        assert (u == v).all()
    return outs


def test_queued_calls_through_a_multi_device_context(pkg, oracle):
    """the planner of a call runs beside the previous call's sweep on EVERY shard of rrrmc_ctx_create_multi (each child
    context has its own two chunk tables, buffer sets and plan stream).  Five queued calls == five synchronised ones == the oracle."""
    seed, beta, R, N, iters, step = 515151, 0.8, 128, 4096, 1 << 16, 4096
    X = pkg.GraphRRG(N, 3, seed=seed)
    outs = _queued_vs_synchronised(pkg, X, R, seed, beta, [(iters, step)] * 5, engine_kw={"devices": [0, 0]})
    for r in (0, 63, 64, 127):          # both shards
        ref = oracle.standard_mc_sparse(X.A, X.J.astype(np.int32), beta, iters, step, seed, outs[1][4][r], replica=r, it0=4 * iters)
        assert (outs[0][0][r] == ref[0]).all() and (outs[0][2][r] == ref[1]).all() and outs[0][1][r] == ref[2]


@pytest.mark.parametrize("no_masks", ["0", "1"])
def test_queued_calls_in_big_mode(pkg, oracle, monkeypatch, no_masks):
    """N = 40 000 does not fit the LDS-resident kernel: plan_big_kernel (+ big_mask_kernel) of call s + 1 runs beside
    big_apply_kernel / big_sweep_kernel of call s, through the same two chunk tables and buffer sets."""
    monkeypatch.setenv("RRRMC_BIG_NO_MASKS", no_masks)
    seed, beta, R, N, iters, step = 616161, 0.9, 64, 40000, 200000, 50000
    X = pkg.GraphRRG(N, 3, seed=seed)
    outs = _queued_vs_synchronised(pkg, X, R, seed, beta, [(iters, step)] * 5)
    for r in (0, 33, 63):
        ref = oracle.standard_mc_sparse(X.A, X.J.astype(np.int32), beta, iters, step, seed, outs[1][4][r], replica=r, it0=4 * iters)
        assert (outs[0][0][r] == ref[0]).all() and (outs[0][2][r] == ref[1]).all() and outs[0][1][r] == ref[2]


def test_queued_calls_alternating_with_colour_sweeps(pkg, oracle):
    """a colour-sweep call between two queued standardMC calls of one shape writes the spins the next planner-overlapped
    sweep reads: the stream orders them, the chunk table stays valid across the other sampler's call."""
    seed, beta, R = 717171, 0.8, 96
    X = pkg.GraphEA(8, 3, seed=seed)
    color = pkg.checkerboard_coloring(8, 3)

    def between(eng, c):
        if c is None:
            eng.set_coloring(color)
        else:
            eng.colored_sweeps_async(beta, 3, 1)

    _queued_vs_synchronised(pkg, X, R, seed, beta, [(60000, 5000)] * 5, between=between)
