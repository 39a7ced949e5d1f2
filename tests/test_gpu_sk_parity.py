"""GPU parity for GraphSKNormal (src/graphs/SK.jl:170-297) under standardMC: BASELINE.json asks for energies within
1e-6 relative; the kernel reproduces the reference's floating-point operation order, so we also check bit equality."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REL_TOL = 1e-6     # north_star tolerance for Float64-coupling models


@pytest.mark.parametrize("N,R,beta,iters,step", [
    (10, 8, 2.0, 10000, 100),        # test/runtests.jl:67 GraphSKNormal(10), beta=2, 10^4 iters, step 100
    (64, 19, 1.0, 20000, 250),       # R not a multiple of 8
    (256, 16, 0.5, 8000, 1),         # step = 1
    (300, 8, 1.0, 6000, 500),        # N not a multiple of 256 (two sites per thread, partially filled)
    (1024, 24, 1.0, 6000, 1000),     # BASELINE config 3 size, few replicas
])
def test_skn_standard_mc(pkg, oracle, N, R, beta, iters, step):
    seed = 31337 + N
    X = pkg.GraphSKNormal(N, seed=seed)
    J_ref = oracle.gen_sk_gauss(N, seed)
    assert (X.J == J_ref).all()
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        E0 = eng.energy()
        lf0 = eng.fields()
        Es, acc = eng.standard_mc(beta, iters, step)
        C1 = eng.get_config()
        lf1 = eng.fields()           # the live cache after the run
        E1 = eng.energy()            # recomputed from scratch
    assert (C0.s == oracle.init_configs(seed, 0, R, N)).all()
    Es_ref, ch_ref, acc_ref, lf_ref = oracle.standard_mc_skn_batch(X.J, beta, iters, step, seed, C0.s)
    for r in range(R):
        e0, f0 = oracle.skn_energy(X.J, C0.s[r], want_fields=True)
        assert E0[r] == e0 and (lf0[r] == f0).all()
    # north-star tolerance
    assert np.allclose(Es, Es_ref, rtol=REL_TOL, atol=1e-9)
    assert (C1.s == ch_ref).all() and (acc == acc_ref).all()
    # and in fact bit for bit: same sequence of IEEE operations per field, same exp, same uniforms
    assert (Es == Es_ref).all()
    assert (lf1 == lf_ref).all()
    # the reference's own invariant: tracked E == energy(X, C) (test/runtests.jl:12-20, atol 1e-11 scaled to N)
    E_last_tracked = Es_ref[:, -1] if Es_ref.shape[1] else None
    for r in range(min(R, 4)):
        assert abs(E1[r] - oracle.skn_energy(X.J, C1.s[r])) == 0.0


def test_skn_resume_and_c0(pkg, oracle):
    seed, N, R = 9, 128, 8
    X = pkg.GraphSKNormal(N, seed=seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es1, a1 = eng.standard_mc(0.8, 3000, 100)
        Es2, a2 = eng.standard_mc(0.8, 2500, 100)
        C2 = eng.get_config()
    for r in range(R):
        e1 = oracle.standard_mc_skn(X.J, 0.8, 3000, 100, seed, C0.s[r], replica=r)
        e2 = oracle.standard_mc_skn(X.J, 0.8, 2500, 100, seed, e1[1], it0=3000, replica=r)
        assert (Es1[r] == e1[0]).all() and (Es2[r] == e2[0]).all() and (C2.s[r] == e2[1]).all()
        assert a1[r] == e1[2] and a2[r] == e2[2]


def test_skn_rejects_bad_couplings(pkg):
    J = np.zeros((4, 4))
    J[0, 1] = 1.0
    with pytest.raises(ValueError):
        pkg.GraphSKNormal.from_J(J)                       # not symmetric (SK.jl:191)
    X = pkg.GraphSKNormal(4, seed=1)
    X.J = X.J.copy()
    X.J[2, 2] = 0.5
    with pytest.raises(pkg.RRRMCError) as e:              # checked again at the ABI (SK.jl:189)
        pkg.Engine(X, 8)
    assert e.value.code == 1 and "diagonal" in str(e.value)


@pytest.mark.parametrize("N,R,beta,iters,step", [
    (10, 8, 2.0, 10000, 100),        # test/runtests.jl:66 GraphSK(10)
    (100, 12, 1.0, 20000, 200),
    (1024, 16, 0.7, 5000, 500),
])
def test_binary_sk_standard_mc(pkg, oracle, N, R, beta, iters, step):
    """GraphSK (bit-packed +-1/sqrt(N) couplings, integer cache): src/graphs/SK.jl:28-165"""
    seed = 1234 + N
    X = pkg.GraphSK(N, seed=seed)
    assert (X.J == oracle.gen_sk_binary(N, seed)).all()
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        E0 = eng.energy()
        lf0 = eng.fields()
        Es, acc = eng.standard_mc(beta, iters, step)
        C1 = eng.get_config()
        lf1 = eng.fields()
        E1 = eng.energy()
    for r in range(R):
        e0, f0 = oracle.skb_energy(X.J, C0.s[r], want_fields=True)
        assert E0[r] == e0 and (lf0[r] == f0).all()
        Es_ref, ch_ref, acc_ref, lf_ref = oracle.standard_mc_skb(X.J, beta, iters, step, seed, C0.s[r], replica=r)
        assert np.allclose(Es[r], Es_ref, rtol=REL_TOL, atol=1e-9)
        assert (Es[r] == Es_ref).all() and (C1.s[r] == ch_ref).all() and acc[r] == acc_ref and (lf1[r] == lf_ref).all()
        assert E1[r] == oracle.skb_energy(X.J, C1.s[r])


def test_config3_full_width_subset(pkg, oracle):
    """BASELINE config 3 at its full width (GraphSKNormal N = 1024, 2048 replicas = 256 workgroups of 8): three workgroups'
    replicas against the oracle, bit for bit."""
    seed, N, R, beta, iters, step = 0x5EED, 1024, 2048, 1.0, 4096, 1024
    X = pkg.GraphSKNormal(N, seed=seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc = eng.standard_mc(beta, iters, step)
        C1 = eng.get_config()
    for g in (0, 131, 255):
        sl = slice(8 * g, 8 * g + 8)
        Es_ref, ch_ref, acc_ref, _ = oracle.standard_mc_skn_batch(X.J, beta, iters, step, seed, C0.s[sl], replica0=8 * g)
        assert (Es[sl] == Es_ref).all() and (C1.s[sl] == ch_ref).all() and (acc[sl] == acc_ref).all()
    assert 0.2 < (acc / iters).mean() < 0.5


def _random_sk_cases(n, seed):
    rng = np.random.default_rng(seed)
    return [(int(rng.integers(2, 1400)), int(rng.integers(1, 40)), float(rng.choice([0.0, 0.5, 1.0, 2.0])), int(rng.integers(1, 4000)),
             int(rng.integers(1, 900)), bool(rng.integers(0, 2))) for _ in range(n)]


@pytest.mark.parametrize("N,R,beta,iters,step,binary", _random_sk_cases(16, 777))
def test_sk_randomized_shapes(pkg, oracle, N, R, beta, iters, step, binary):
    """Seeded random sizes of the dense kernels (1 to 6 sites per thread, partially filled last tile, any number of replicas in the last
    group of 8), Gaussian and binary couplings, with a resumed second call."""
    seed = 13 * N + R
    X = pkg.GraphSK(N, seed=seed) if binary else pkg.GraphSKNormal(N, seed=seed)
    run = oracle.standard_mc_skb if binary else oracle.standard_mc_skn
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc = eng.standard_mc(beta, iters, step)
        C1 = eng.get_config()
        Es2, acc2 = eng.standard_mc(beta, iters // 2 + 1, step)
    for r in sorted({0, R - 1}):
        ref = run(X.J, beta, iters, step, seed, C0.s[r], replica=r)
        assert (Es[r] == ref[0]).all() and (C1.s[r] == ref[1]).all() and acc[r] == ref[2]
        ref2 = run(X.J, beta, iters // 2 + 1, step, seed, ref[1], it0=iters, replica=r)
        assert (Es2[r] == ref2[0]).all() and acc2[r] == ref2[2]


@pytest.mark.parametrize("N,R,beta,iters,step,thr", [
    (10, 40, 2.0, 6000, 100, 0.8),            # test/runtests.jl:66 GraphSK(10) under rrrMC (:153), default staged_thr of a SimpleGraph
    (10, 8, 2.0, 3000, 50, 0.0),              # always direct
    (10, 8, 2.0, 3000, 50, 1.0),              # always staged
    (100, 70, 1.0, 3000, 100, 0.8),
    (333, 9, 1.5, 1500, 100, 0.8),
])
def test_binary_sk_rrr_bkl_wtm_bit_exact(pkg, oracle, N, R, beta, iters, step, thr):
    """The binary GraphSK is a SimpleGraph{Float64} (SK.jl:28): rrrMC / bklMC / wtmMC run DeltaECacheCont / THeap over
    delta_energy = lfields[i] / sqrt(N) with integer fields (SK.jl:137-140)."""
    seed = 636000 + N
    X = pkg.GraphSK(N, seed=seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc, staged = eng.rrr_mc(beta, iters, step, staged_thr=thr)
        C1 = eng.get_config()
        E1 = eng.energy()
        Eb, mb = eng.bkl_mc(beta, iters * 4, step * 4)
        C2 = eng.get_config()
        Ew, mw, tw = eng.wtm_mc(beta, 30, step=1.5)
        C3 = eng.get_config()
        Es4, acc4 = eng.standard_mc(beta, 2000, 100)              # the integer-field Metropolis kernel still works afterwards
    for r in range(R if N <= 100 else 3):
        ref = oracle.rrr_mc_skb(X.J, beta, iters, step, seed, C0.s[r], replica=r, staged_thr=thr)
        assert np.allclose(Es[r], ref[0], rtol=1e-6, atol=1e-9)
        assert (Es[r] == ref[0]).all() and (C1.s[r] == ref[1]).all() and acc[r] == ref[2] and staged[r] == ref[3]
        assert E1[r] == oracle.skb_energy(X.J, ref[1])
        rb = oracle.bkl_mc_skb(X.J, beta, iters * 4, step * 4, seed, ref[1], it0=iters, replica=r)
        assert (Eb[r] == rb[0]).all() and (C2.s[r] == rb[1]).all() and mb[r] == rb[2]
        rw = oracle.wtm_mc_skb(X.J, beta, 30, 1.5, seed, rb[1], replica=r)
        assert (Ew[r] == rw[0]).all() and (C3.s[r] == rw[1]).all() and mw[r] == rw[2] and tw[r] == rw[3]


@pytest.mark.parametrize("kind", ["skn", "skb", "rrgn", "ean", "dbl", "quant"])
def test_hooked_run_is_the_unhooked_chain_for_float64_models(pkg, oracle, kind):
    """A run with a hook is ONE reference call (src/RRRMC.jl:95-118): the cache and the tracked E live on across hook calls.  The
    library's hooked standardMC resumes its pieces (rrrmc_set_resume), so for the Float64 models too the energies handed to the hook
    — the tracked E — and the final state equal the un-hooked run bit for bit (round 1 re-entered the sampler at every hook point and
    recomputed E and the fields there: last-bit drift, possibly a forked trajectory)."""
    seed, R, beta, iters, step = 99, 5, 0.8, 4000, 37
    X = {"skn": lambda: pkg.GraphSKNormal(96, seed=seed), "skb": lambda: pkg.GraphSK(96, seed=seed),
         "rrgn": lambda: pkg.GraphRRGNormal(200, 3, seed=seed), "ean": lambda: pkg.GraphEANormal(6, 3, seed=seed),
         "dbl": lambda: pkg.GraphRRGNormalDiscretized(120, 3, (-1, 0, 1), seed=seed),
         "quant": lambda: pkg.GraphQuant(pkg.GraphRRG(24, 3, seed=seed), 5, 0.5, 2.0)}[kind]()
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es_ref, _ = eng.standard_mc(beta, iters, step)
        C_ref = eng.get_config().s.copy()
    seen = []

    def hook(it, X_, C, accepted, E):
        seen.append((it, E.copy(), accepted.copy()))
        return True

    Cfg = pkg.Config(X.N, R)
    Cfg.s[:] = C0.s
    Es, Cend = pkg.standardMC(X, beta, iters, seed=seed, step=step, hook=hook, C0=Cfg, quiet=True, replicas=R)
    assert Es.shape == Es_ref.shape and (Es == Es_ref).all()           # bit for bit, Float64 included
    assert (Cend.s == C_ref).all()
    assert [h[0] for h in seen] == list(range(step, iters + 1, step))
    assert all((h[1] == Es_ref[:, k]).all() for k, h in enumerate(seen))
    assert (np.diff(np.stack([h[2] for h in seen]), axis=0) >= 0).all()


@pytest.mark.parametrize("N,R", [(1024, 16), (700, 8), (2048, 8)])
def test_skn_thread_counts_agree(pkg, monkeypatch, N, R):
    """sk_sweep_kernel with 256 / 512 / 1024 threads per workgroup (RRRMC_SK_THREADS): the same chain bit for bit — energies, accepted
    counts, configurations and the live field cache."""
    seed = 4242 + N
    X = pkg.GraphSKNormal(N, seed=seed)
    outs = []
    for nth in ("256", "512", "1024"):
        monkeypatch.setenv("RRRMC_SK_THREADS", nth)
        with pkg.Engine(X, R) as eng:
            eng.seed(seed)
            eng.init_spins_random()
            Es, acc = eng.standard_mc(1.0, 3000, 100)
            Es2, acc2 = eng.standard_mc(1.0, 1000, 7)      # a second call continues the streams
            outs.append((Es, acc, Es2, acc2, eng.get_config().s, eng.fields()))
    for o in outs[1:]:
        for u, v in zip(outs[0], o):
            assert (u == v).all()


@pytest.mark.parametrize("N,R", [(1024, 16), (700, 9), (300, 8), (64, 5), (2048, 8)])
def test_binary_sk_block_kernel_builds_agree(pkg, oracle, monkeypatch, N, R):
    """The binary model through sk_block_kernel<.., BIN> (whole-group and split builds; the integer fields travel as doubles) against the
    one-attempt-at-a-time skb_sweep_kernel (RRRMC_SK_LEGACY = 1) and the oracle: energies, accepted counts, configurations, the int32 cache."""
    seed = 4001 + N
    X = pkg.GraphSK(N, seed=seed)
    outs = []
    for env in ({"RRRMC_SK_RB": "8"}, {"RRRMC_SK_RB": "4"}, {"RRRMC_SK_LEGACY": "1"}, {"RRRMC_SK_BLOCK_V1": "1", "RRRMC_SK_RB": "8"}, {"RRRMC_SK_BLOCK_V1": "1", "RRRMC_SK_RB": "4"}):
        for k in ("RRRMC_SK_RB", "RRRMC_SK_LEGACY", "RRRMC_SK_BLOCK_V1"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with pkg.Engine(X, R) as eng:
            eng.seed(seed)
            eng.init_spins_random()
            C0 = eng.get_config().s.copy()
            Es, acc = eng.standard_mc(1.2, 3000, 100)
            Es2, acc2 = eng.standard_mc(0.6, 777, 7)
            outs.append((Es, acc, Es2, acc2, eng.get_config().s, eng.fields(), eng.energy()))
    for o in outs[1:]:
        for u, v in zip(outs[0], o):
            assert (u == v).all()
    for r in (0, R - 1):
        ref = oracle.standard_mc_skb(X.J, 1.2, 3000, 100, seed, C0[r], replica=r)
        assert (outs[0][0][r] == ref[0]).all() and outs[0][1][r] == ref[2]


@pytest.mark.parametrize("N,R", [(1024, 24), (1000, 13), (700, 8), (300, 9), (256, 16), (37, 5)])
def test_skn_block_kernel_builds_agree(pkg, oracle, monkeypatch, N, R):
    """sk_block_kernel with one 8-replica workgroup per group of replicas and with two co-resident 4-replica workgroups
    (RRRMC_SK_RB = 8 / 4), and the one-attempt-at-a-time sk_sweep_kernel (RRRMC_SK_LEGACY = 1): the same chain bit for bit — energies,
    accepted counts, configurations (the two halves of a group share each spin byte) and the live field cache; R not a multiple of 4
    leaves padded replicas in the last workgroups."""
    seed = 977 + N
    X = pkg.GraphSKNormal(N, seed=seed)
    outs = []
    for env in ({"RRRMC_SK_RB": "8"}, {"RRRMC_SK_RB": "4"}, {"RRRMC_SK_LEGACY": "1"}, {"RRRMC_SK_BLOCK_V1": "1", "RRRMC_SK_RB": "8"}, {"RRRMC_SK_BLOCK_V1": "1", "RRRMC_SK_RB": "4"}):
        for k in ("RRRMC_SK_RB", "RRRMC_SK_LEGACY", "RRRMC_SK_BLOCK_V1"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with pkg.Engine(X, R) as eng:
            eng.seed(seed)
            eng.init_spins_random()
            Es, acc = eng.standard_mc(1.5, 4000, 100)
            Es2, acc2 = eng.standard_mc(0.7, 1111, 7)      # a second call continues the streams
            outs.append((Es, acc, Es2, acc2, eng.get_config().s, eng.fields(), eng.energy()))
    for o in outs[1:]:
        for u, v in zip(outs[0], o):
            assert (u == v).all()
    # and the oracle, for the split build's first call
    C0 = oracle.init_configs(seed, 0, R, N)
    Es_ref, ch_ref, acc_ref, lf_ref = oracle.standard_mc_skn_batch(X.J, 1.5, 4000, 100, seed, C0)
    assert (outs[1][0] == Es_ref).all() and (outs[1][1] == acc_ref).all()


@pytest.mark.parametrize("binary", [False, True], ids=["gauss", "binary"])
@pytest.mark.parametrize("N", [700, 1500])
def test_sk_threads_override_never_leaves_a_model_without_a_kernel(pkg, monkeypatch, binary, N):
    """RRRMC_SK_THREADS = 256 (documented as a bit-identical override) asks for a workgroup shape the blocked kernels do not have beyond
    N = 256: up to N = 1024 the split build takes over, beyond it the one-attempt-at-a-time kernels — for the binary model too (it used
    to fail with 'internal: no blocked kernel').  Same chain as the default build."""
    seed = 6100 + N
    X = pkg.GraphSK(N, seed=seed) if binary else pkg.GraphSKNormal(N, seed=seed)
    outs = []
    for thr in (None, "256"):
        for k in ("RRRMC_SK_RB", "RRRMC_SK_LEGACY", "RRRMC_SK_BLOCK_V1", "RRRMC_SK_THREADS"):
            monkeypatch.delenv(k, raising=False)
        if thr:
            monkeypatch.setenv("RRRMC_SK_THREADS", thr)
        with pkg.Engine(X, 5) as eng:
            eng.seed(seed)
            eng.init_spins_random()
            Es, acc = eng.standard_mc(1.1, 1500, 50)
            outs.append((Es, acc, eng.get_config().s.copy(), eng.fields()))
    for u, v in zip(outs[0], outs[1]):
        assert (u == v).all()


@pytest.mark.parametrize("binary,N", [(False, 3000), (False, 3585), (False, 4096), (True, 2500)])
def test_dense_sk_beyond_2048_sites(pkg, oracle, binary, N):
    """src/graphs/SK.jl:181-210 has no size limit; round 3 stopped at N = 2048 (eight sites per thread of the one-attempt-at-a-time kernels).
    sk_hblock_kernel<6 | 8, 512, 8> covers N <= 4096: same chain as the oracle, both models, a call that crosses no seam and one that does."""
    seed, R, beta = 8800 + N, 3, 1.0
    X = pkg.GraphSK(N, seed=seed) if binary else pkg.GraphSKNormal(N, seed=seed)
    run = oracle.standard_mc_skb if binary else oracle.standard_mc_skn
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config().s.copy()
        Es, acc = eng.standard_mc(beta, 3000, 100)
        C1, lf1 = eng.get_config().s.copy(), eng.fields()
    for r in (0, R - 1):
        ref = run(X.J, beta, 3000, 100, seed, C0[r], replica=r)
        assert (Es[r] == ref[0]).all() and (C1[r] == ref[1]).all() and acc[r] == ref[2] and (lf1[r] == ref[3]).all()
    # a call of zero iterations (E = energy(X, C) and an empty sample list: what the bindings issue before a hooked run) must not need a
    # sweep build (these sizes have a blocked one only), and the hooked run must be the un-hooked chain
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.set_config(pkg.Config(N, R, C0.copy()))
        Es0, acc0 = eng.standard_mc(beta, 0, 1)
        assert Es0.shape == (R, 0) and (acc0 == 0).all()
        assert (eng.get_config().s == C0).all()
    seen = []
    EsH, CH = pkg.standardMC(X, beta, 3000, seed=seed, step=100, hook=lambda it, X_, C, accd, E: seen.append(it) or True, C0=pkg.Config(N, R, C0.copy()),
                             quiet=True, replicas=R)
    assert seen == list(range(100, 3001, 100))
    assert (np.asarray(EsH) == Es).all() and (CH.s == C1).all()
    with pytest.raises(pkg.RRRMCError):
        pkg.Engine(pkg.GraphSKNormal(4097, seed=1), 8)
