"""GPU parity for rrrMC(X::DoubleGraph) on GraphQuant (src/RRRMC.jl:221-290, src/graphs/QT.jl, src/DeltaE.jl, src/ArraySets.jl).
north_star tolerance for Float64 models is 1e-6 relative; the kernel keeps the reference's operation order and the
deterministic exp, so we additionally require bit equality with the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("Nk,M,Gamma,beta,R,iters,step,thr", [
    (10, 8, 0.5, 2.0, 8, 10000, 100, 0.5),     # test/runtests.jl:78 GraphQuant(10, 8, 0.5, 2.0, GraphRRG, 10, 3); default staged_thr
    (10, 8, 0.5, 2.0, 8, 10000, 100, 0.0),     # runtests.jl:150 staged_thr = 0.0 (always direct)
    (10, 8, 0.5, 2.0, 8, 10000, 100, 1.0),     # runtests.jl:155 staged_thr = 1.0 (always staged)
    (64, 16, 0.3, 1.0, 70, 20000, 250, 0.5),   # more than one 64-thread block, R not a multiple of 64
    (256, 32, 0.5, 2.0, 16, 30000, 1000, 0.5),
    (1024, 32, 0.5, 2.0, 4, 20000, 4096, 0.5), # BASELINE config 5 geometry (Nk=1024, M=32 -> N=32768), few replicas
])
def test_rrr_quant_bit_exact(pkg, oracle, Nk, M, Gamma, beta, R, iters, step, thr):
    seed = 8426732438942 + Nk
    X1 = pkg.GraphRRG(Nk, 3, seed=seed)
    X = pkg.GraphQuant(X1, M, Gamma, beta)
    assert X.fourK == oracle.quant_fourK(beta, Gamma, M)
    A, J = X1.A, X1.J.astype(np.int32)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        E0 = eng.energy()
        Es, acc, staged = eng.rrr_mc(beta, iters, step, staged_thr=thr)
        C1 = eng.get_config()
        pos, sizes = eng.rrr_cache()
        E1 = eng.energy()
    assert (C0.s == oracle.init_configs(seed, 0, R, X.N)).all()
    for r in range(R):
        e0, _ = oracle.quant_energy(A, J, M, X.fourK, C0.s[r])
        assert E0[r] == e0
        Es_ref, ch_ref, acc_ref, st_ref, pos_ref, sizes_ref = oracle.rrr_mc_quant(A, J, M, X.fourK, beta, iters, step, seed, C0.s[r],
                                                                                  replica=r, staged_thr=thr, want_cache=True)
        assert np.allclose(Es[r], Es_ref, rtol=1e-6, atol=1e-9)          # north-star tolerance
        assert (Es[r] == Es_ref).all()                                    # ... and bit for bit
        assert (C1.s[r] == ch_ref).all() and acc[r] == acc_ref and staged[r] == st_ref
        assert (pos[r] == pos_ref).all() and (sizes[r] == sizes_ref).all()
        # the reference's invariant: tracked E ~ energy(X, C) (test/runtests.jl:12-20)
        e1, _ = oracle.quant_energy(A, J, M, X.fourK, C1.s[r])
        assert E1[r] == e1


@pytest.mark.parametrize("Nk,M,Gamma,beta,R,iters,step", [
    (10, 8, 0.5, 2.0, 40, 10000, 100),         # test/runtests.jl:78 under standardMC (:141-143)
    (64, 16, 0.3, 1.0, 70, 20000, 250),
    (1024, 32, 0.5, 2.0, 4, 40000, 4096),      # BASELINE config 5 geometry
])
def test_standard_mc_quant_bit_exact(pkg, oracle, Nk, M, Gamma, beta, R, iters, step):
    """standardMC on GraphQuant: delta_energy = delta_energy(X0) + delta_energy_residual (QT.jl:283-286); two calls continue the streams."""
    seed = 8426732438942 + Nk + 1
    X1 = pkg.GraphRRG(Nk, 3, seed=seed)
    X = pkg.GraphQuant(X1, M, Gamma, beta)
    A, J = X1.A, X1.J.astype(np.int32)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc = eng.standard_mc(beta, iters, step)
        C1 = eng.get_config()
        Es2, acc2 = eng.standard_mc(beta, iters // 2, step)
        C2 = eng.get_config()
        E2 = eng.energy()
        with pytest.raises(pkg.RRRMCError):
            eng.rrr_cache()
    for r in range(R):
        ref = oracle.standard_mc_quant(A, J, M, X.fourK, beta, iters, step, seed, C0.s[r], replica=r)
        assert np.allclose(Es[r], ref[0], rtol=1e-6, atol=1e-9)
        assert (Es[r] == ref[0]).all() and (C1.s[r] == ref[1]).all() and acc[r] == ref[2]
        ref2 = oracle.standard_mc_quant(A, J, M, X.fourK, beta, iters // 2, step, seed, ref[1], it0=iters, replica=r)
        assert (Es2[r] == ref2[0]).all() and (C2.s[r] == ref2[1]).all() and acc2[r] == ref2[2]
        assert E2[r] == oracle.quant_energy(A, J, M, X.fourK, C2.s[r])[0]
    assert 0 < acc.sum() < R * iters


def test_rrr_quant_lds_and_global_builds_agree(pkg, oracle, monkeypatch):
    """Few replicas run one workgroup per replica with the hot state staged in LDS; many replicas (or RRRMC_QUANT_NO_LDS=1) run the
    thread-per-replica build on HBM/L2.  Same chains bit for bit; odd N exercises the LDS carving's padding."""
    seed = 31337
    X = pkg.GraphQuant(pkg.GraphRRG(15, 4, seed=seed), 5, 0.4, 1.2)          # N = 75
    out = []
    for no_lds in ("0", "1"):
        monkeypatch.setenv("RRRMC_QUANT_NO_LDS", no_lds)
        with pkg.Engine(X, 9) as eng:
            eng.seed(seed)
            eng.init_spins_random()
            C0 = eng.get_config()
            a = eng.rrr_mc(1.2, 6000, 100)
            b = eng.rrr_mc(1.2, 3000, 100, staged_thr=1.0)
            out.append((a[0], a[1], a[2], b[0], b[1], b[2], eng.get_config().s, eng.rrr_cache()[0], eng.rrr_cache()[1]))
    for u, v in zip(*out):
        assert (u == v).all()
    for r in range(9):
        ref = oracle.rrr_mc_quant(X.A, X.J.astype(np.int32), X.M, X.fourK, 1.2, 6000, 100, seed, C0.s[r], replica=r)
        assert (out[0][0][r] == ref[0]).all() and out[0][1][r] == ref[2]
    with pkg.Engine(X, 300) as eng:                                            # more than 256 replicas: several per workgroup
        eng.seed(seed)
        eng.init_spins_random()
        Es, acc, st = eng.rrr_mc(1.2, 2000, 100)
        C0b = oracle.init_configs(seed, 0, 300, X.N)
    for r in (0, 8, 255, 256, 299):
        ref = oracle.rrr_mc_quant(X.A, X.J.astype(np.int32), X.M, X.fourK, 1.2, 2000, 100, seed, C0b[r], replica=r)
        assert (Es[r] == ref[0]).all() and acc[r] == ref[2]


def test_rrrMC_front_end(pkg, oracle):
    seed = 4242
    X = pkg.GraphQuant(pkg.GraphRRG(16, 3, seed=seed), 4, 0.7, 1.5)
    Es, C = pkg.rrrMC(X, 1.5, 4000, step=50, seed=seed, quiet=True, replicas=4)
    C0 = pkg.Config(X.N, 4, oracle.init_configs(seed, 0, 4, X.N))
    for r in range(4):
        ref = oracle.rrr_mc_quant(X.A, X.J.astype(np.int32), X.M, X.fourK, 1.5, 4000, 50, seed, C0.s[r], replica=r)
        assert (Es[r] == ref[0]).all() and (C.s[r] == ref[1]).all()
    with pytest.raises(ValueError):
        pkg.GraphQuant(pkg.GraphRRG(16, 3, seed=seed), 2, 0.7, 1.5)       # "M must be greater than 2", QT.jl:47


@pytest.mark.parametrize("N,R,beta,iters,step,thr", [
    (10, 8, 2.0, 10000, 100, None),     # test/runtests.jl:67,144-157 GraphSKNormal(10): default staged_thr (0.8 for a SimpleGraph)
    (10, 8, 2.0, 10000, 100, 0.0),
    (10, 8, 2.0, 10000, 100, 1.0),
    (100, 20, 1.0, 3000, 50, None),     # N not a power of two: the sampler tree is padded to 128 leaves
    (256, 70, 1.5, 600, 100, None),
])
def test_rrr_skn_continuous_cache(pkg, oracle, N, R, beta, iters, step, thr):
    """rrrMC(X::SingleGraph) with DeltaECacheCont + DynamicSampler (src/DeltaE.jl:297-410, src/DynamicSamplers.jl) on GraphSKNormal."""
    seed = 555 + N
    X = pkg.GraphSKNormal(N, seed=seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc, staged = eng.rrr_mc(beta, iters, step, staged_thr=thr)
        C1 = eng.get_config()
        E1 = eng.energy()
    for r in range(R):
        ref = oracle.rrr_mc_skn(X.J, beta, iters, step, seed, C0.s[r], replica=r, staged_thr=0.8 if thr is None else thr)
        assert np.allclose(Es[r], ref[0], rtol=1e-6, atol=1e-9)          # north-star tolerance
        assert (Es[r] == ref[0]).all() and (C1.s[r] == ref[1]).all() and acc[r] == ref[2] and staged[r] == ref[3]
        assert E1[r] == oracle.skn_energy(X.J, C1.s[r])


@pytest.mark.parametrize("kind,R,beta,iters,step,thr", [
    (("rrg", 10, 3), 8, 2.0, 10000, 100, None),     # test/runtests.jl:36,144-157: GraphRRG(10,3) under rrrMC, three staged_thr regimes
    (("rrg", 10, 3), 8, 2.0, 10000, 100, 0.0),
    (("rrg", 10, 3), 8, 2.0, 10000, 100, 1.0),
    (("ea", 2, 3), 8, 2.0, 10000, 100, None),       # runtests.jl:46: doubled neighbours -> uA de-duplication
    (("ea", 3, 2), 8, 2.0, 10000, 100, None),
    (("rrg", 4096, 3), 70, 1.0, 20000, 1000, None), # BASELINE config 2 graph under the RRR sampler
    (("ea", 6, 3), 40, 1.5, 8000, 80, None),        # K = 6: four ΔE levels
])
def test_rrr_and_bkl_on_discrete_graphs(pkg, oracle, kind, R, beta, iters, step, thr):
    """rrrMC(X::SingleGraph) and bklMC on GraphRRG / GraphEA with DeltaECache{Int,L}."""
    seed = 8426732438942 % 2 ** 40 + kind[1]
    X = pkg.GraphRRG(kind[1], kind[2], seed=seed) if kind[0] == "rrg" else pkg.GraphEA(kind[1], kind[2], seed=seed)
    A, J = X.A, X.J.astype(np.int32)
    form = "ea" if kind[0] == "ea" else "rrg"
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc, staged = eng.rrr_mc(beta, iters, step, staged_thr=thr)
        C1 = eng.get_config()
        E1 = eng.energy()
        eng.seed(seed)
        eng.set_config(C0)
        Eb, moves = eng.bkl_mc(beta, iters, step)
        Cb = eng.get_config()
    for r in list(range(min(R, 6))) + [R - 1]:
        ref = oracle.rrr_sparse(A, J, beta, iters, step, seed, C0.s[r], replica=r, staged_thr=0.5 if thr is None else thr, form=form)
        assert (Es[r] == ref[0]).all() and (C1.s[r] == ref[1]).all() and acc[r] == ref[2] and staged[r] == ref[3]
        assert E1[r] == oracle.sparse_energy(A, J, C1.s[r])
        refb = oracle.rrr_sparse(A, J, beta, iters, step, seed, C0.s[r], replica=r, form=form, bkl=True)
        assert (Eb[r] == refb[0]).all() and (Cb.s[r] == refb[1]).all() and moves[r] == refb[2]


def _random_rrr_cases(n, seed):
    rng = np.random.default_rng(seed)
    cases = []
    while len(cases) < n:
        K = int(rng.integers(1, 7))
        N = int(rng.integers(max(6, 2 * K + 2), 500))
        if (N * K) % 2:
            N += 1
        cases.append((N, K, int(rng.integers(1, 150)), float(rng.choice([0.3, 1.0, 2.0])), int(rng.integers(1, 6000)),
                      int(rng.integers(1, 700)), str(rng.choice(["rrr", "bkl", "wtm", "eo"])), int(rng.choice([0, 1, 4, 16, 64]))))
    return cases


@pytest.mark.parametrize("N,K,R,beta,iters,step,sampler,tpb", _random_rrr_cases(20, 9090))
def test_discrete_samplers_randomized(pkg, oracle, monkeypatch, N, K, R, beta, iters, step, sampler, tpb):
    """Seeded random shapes for the thread-per-replica samplers on GraphRRG (K = 1 .. 6: with and without a zero level), with the
    launch shape forced to 1 .. 64 replicas per workgroup (RRRMC_RRR_TPB): results must not depend on it."""
    if tpb:
        monkeypatch.setenv("RRRMC_RRR_TPB", str(tpb))
    seed = 53 * N + K
    X = pkg.GraphRRG(N, K, seed=seed)
    A, J = X.A, X.J.astype(np.int32)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        if sampler == "rrr":
            out = eng.rrr_mc(beta, iters, step)
        elif sampler == "bkl":
            out = eng.bkl_mc(beta, iters, step)
        elif sampler == "wtm":
            out = eng.wtm_mc(beta, max(iters // max(step, 1), 1), step=float(step))
        else:
            out = eng.extremal_opt(1.0 + beta / 4, iters, step)
        C1 = eng.get_config()
    for r in sorted({0, R - 1}):
        if sampler in ("rrr", "bkl"):
            ref = oracle.rrr_sparse(A, J, beta, iters, step, seed, C0.s[r], replica=r, bkl=sampler == "bkl")
            assert (out[0][r] == ref[0]).all() and (C1.s[r] == ref[1]).all() and out[1][r] == ref[2]
        elif sampler == "wtm":
            ref = oracle.wtm_mc_sparse(A, J, beta, max(iters // max(step, 1), 1), float(step), seed, C0.s[r], replica=r)
            assert (out[0][r] == ref[0]).all() and (C1.s[r] == ref[1]).all() and out[1][r] == ref[2] and out[2][r] == ref[3]
        else:
            ref = oracle.extremal_opt_sparse(A, J, 1.0 + beta / 4, iters, step, seed, C0.s[r], replica=r)
            assert (out[0][r] == ref[0]).all() and (C1.s[r] == ref[1]).all() and out[1][r] == ref[2] and (out[2].s[r] == ref[3]).all()


@pytest.mark.parametrize("kind,R,beta,iters,step,thr", [
    (("rrg", 10000, 3), 6, 2.0, 30000, 1000, 0.5),     # the reference's experiment size (scripts.jl:23): 121 KB of LDS per replica
    (("rrg", 333, 4), 40, 1.0, 12000, 100, 0.0),       # even K, always direct
    (("ea", 5, 3), 33, 1.5, 9000, 125, 1.0),           # K = 6, always staged
])
def test_rrr_sparse_lds_and_global_builds_agree(pkg, oracle, monkeypatch, kind, R, beta, iters, step, thr):
    """Few replicas run one workgroup per replica with the state and the graph staged in LDS (and the RRR draws batched over the
    wavefront); RRRMC_RRR_NO_LDS=1 runs the thread-per-replica build on HBM/L2.  Same chains, rrrMC and bklMC."""
    seed = 8800 + kind[1]
    X = pkg.GraphRRG(kind[1], kind[2], seed=seed) if kind[0] == "rrg" else pkg.GraphEA(kind[1], kind[2], seed=seed)
    form = "ea" if kind[0] == "ea" else "rrg"
    out = []
    for no_lds in ("0", "1"):
        monkeypatch.setenv("RRRMC_RRR_NO_LDS", no_lds)
        with pkg.Engine(X, R) as eng:
            eng.seed(seed)
            eng.init_spins_random()
            C0 = eng.get_config()
            a = eng.rrr_mc(beta, iters, step, staged_thr=thr)
            Ca = eng.get_config().s
            b = eng.bkl_mc(beta, iters * 5, step * 5)
            out.append((a[0], a[1], a[2], Ca, b[0], b[1], eng.get_config().s))
    for u, v in zip(*out):
        assert (u == v).all()
    for r in (0, R - 1):
        ref = oracle.rrr_sparse(X.A, X.J.astype(np.int32), beta, iters, step, seed, C0.s[r], replica=r, staged_thr=thr, form=form)
        assert (out[0][0][r] == ref[0]).all() and (out[0][3][r] == ref[1]).all() and out[0][1][r] == ref[2] and out[0][2][r] == ref[3]


@pytest.mark.parametrize("slices,Nk,M,Gamma,beta,R", [
    ("rrg", 10, 8, 0.5, 2.0, 40),            # test/runtests.jl:78 under bklMC / wtmMC (:145-151)
    ("ea3x2", 9, 5, 0.4, 1.0, 9),            # GraphEA(3, 2) slices (K = 4)
    ("ea2x3", 8, 6, 0.4, 1.0, 7),            # GraphEA(2, 3) slices: every neighbour listed twice (two bonds, EA.jl:156); uA de-duplicated
    ("ea2x2", 4, 4, 0.6, 1.5, 5),            # GraphEA(2, 2)
    ("rrg", 64, 16, 0.3, 1.5, 70),
    ("rrg", 1024, 32, 0.5, 2.0, 2),          # BASELINE config 5 geometry
])
def test_bkl_and_wtm_on_graph_quant(pkg, oracle, slices, Nk, M, Gamma, beta, R):
    """A GraphQuant is a DoubleGraph, not a DiscrGraph: bklMC and wtmMC build the continuous-energy caches over all N = Nk M spins
    (DeltaE.jl:315, WaitingTimes.jl) with delta_energy = delta_energy(X0) + residual (QT.jl:283-286) and neighbors = the Trotter pair,
    then the slice graph's (QT.jl:288-321)."""
    seed = 919000 + Nk + M
    X1 = pkg.GraphEA(int(slices[2]), int(slices[4]), seed=seed) if slices.startswith("ea") else pkg.GraphRRG(Nk, 3, seed=seed)
    X = pkg.GraphQuant(X1, M, Gamma, beta)
    A, J = X1.A, X1.J.astype(np.int32)
    form = "ea" if slices.startswith("ea") else "rrg"
    iters, step, samples = (4000, 100, 20) if Nk >= 1024 else (12000, 200, 40)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Eb, mb = eng.bkl_mc(beta, iters, step)
        C1 = eng.get_config()
        E1 = eng.energy()
        eng.seed(seed)
        eng.set_config(C0)
        Ew, mw, tw = eng.wtm_mc(beta, samples, step=2.0)
        C2 = eng.get_config()
        Er, ar, st = eng.rrr_mc(beta, 2000, 100)               # the DoubleGraph rrrMC still works after the other samplers
        C3 = eng.get_config()
    for r in range(R):
        b = oracle.cont_quant("bkl", A, J, M, X.fourK, beta, iters, step, seed, C0.s[r], replica=r, form=form)
        assert np.allclose(Eb[r], b[0], rtol=1e-6, atol=1e-9)
        assert (Eb[r] == b[0]).all() and (C1.s[r] == b[1]).all() and mb[r] == b[2][0]
        assert E1[r] == oracle.quant_energy(A, J, M, X.fourK, C1.s[r])[0]
        w = oracle.cont_quant("wtm", A, J, M, X.fourK, beta, samples, 1, seed, C0.s[r], replica=r, stepf=2.0, form=form)
        assert (Ew[r] == w[0]).all() and (C2.s[r] == w[1]).all() and mw[r] == w[2][0] and tw[r] == w[3]
        ref = oracle.rrr_mc_quant(A, J, M, X.fourK, beta, 2000, 100, seed, w[1], replica=r)
        assert (Er[r] == ref[0]).all() and (C3.s[r] == ref[1]).all() and ar[r] == ref[2]
    if slices.startswith("ea2"):             # standardMC on the same GraphQuant (repeated bonds in delta_energy)
        with pkg.Engine(X, R) as eng:
            eng.seed(seed)
            eng.set_config(C0)
            Es, acc = eng.standard_mc(beta, 3000, 100)
            C4 = eng.get_config()
        for r in range(R):
            ref = oracle.standard_mc_quant(A, J, M, X.fourK, beta, 3000, 100, seed, C0.s[r], replica=r)
            assert (Es[r] == ref[0]).all() and (C4.s[r] == ref[1]).all() and acc[r] == ref[2]


def test_quant_wave_kernel_equals_thread_kernel_and_respaces(pkg, oracle, monkeypatch):
    """rrrMC on a GraphQuant has two builds: one wavefront per replica with the DeltaECache in LDS (few replicas) and one thread per
    replica (many).  Same chain, bit for bit — also when the LDS build is given so little slack between its four set segments
    that it has to re-space them every few batches (a quench from a random start moves thousands of spins between the classes)."""
    Nk, M, R, seed, beta, Gamma = 96, 12, 6, 4242, 2.0, 0.5
    iters, step = 6000, 250
    X = pkg.GraphQuant(pkg.GraphRRG(Nk, 3, seed=seed), M, Gamma, beta)

    def run():
        with pkg.Engine(X, R) as eng:
            eng.seed(seed)
            eng.init_spins_random()
            C0 = eng.get_config()
            Es, acc, staged = eng.rrr_mc(beta, iters, step)
            Es2, acc2, staged2 = eng.rrr_mc(beta, 777, 100, staged_thr=1.0)      # staged branch, continuing the stream
            return C0, Es, acc, staged, Es2, acc2, staged2, eng.get_config().s.copy(), eng.rrr_cache()

    monkeypatch.setenv("RRRMC_QUANT_WAVE_SLACK", "2048")
    wave = run()
    monkeypatch.delenv("RRRMC_QUANT_WAVE_SLACK")
    monkeypatch.setenv("RRRMC_QUANT_NO_WAVE", "1")
    thread = run()
    for a, b in zip(wave[1:8], thread[1:8]):
        assert (np.asarray(a) == np.asarray(b)).all()
    assert (wave[8][0] == thread[8][0]).all() and (wave[8][1] == thread[8][1]).all()
    A, J = X.X1.A, X.X1.J.astype(np.int32)
    ref = oracle.rrr_mc_quant(A, J, M, X.fourK, beta, iters, step, seed, wave[0].s[3], replica=3)
    assert (wave[1][3] == ref[0]).all() and wave[2][3] == ref[2]


@pytest.mark.parametrize("slices,Nk,M,Gamma,beta,R,tau", [
    ("rrg", 10, 8, 0.5, 2.0, 20, 1.3),           # test/runtests.jl:78 under extremal_opt (:153-157)
    ("ea2x3", 8, 6, 0.4, 1.0, 5, 1.8),           # GraphEA(2, 3) slices: repeated bonds
    ("rrg", 32, 8, 0.5, 2.0, 3, 1.2),
])
def test_extremal_opt_on_graph_quant(pkg, oracle, slices, Nk, M, Gamma, beta, R, tau):
    """extremal_opt on a GraphQuant: not a DiscrGraph, so the reference's generic EOCacheCont ranks all Nk M spins by
    delta_energy(X0) + residual and re-sorts after every flip; neighbors = the Trotter pair, then the slice graph's."""
    seed = 77100 + Nk + M
    X1 = pkg.GraphEA(int(slices[2]), int(slices[4]), seed=seed) if slices.startswith("ea") else pkg.GraphRRG(Nk, 3, seed=seed)
    X = pkg.GraphQuant(X1, M, Gamma, beta)
    A, J = X1.A, X1.J.astype(np.int32)
    form = "ea" if slices.startswith("ea") else "rrg"
    iters, step = (800, 100) if Nk >= 32 else (4000, 200)      # integer-valued slice energies: the ranking is mostly ties, re-keyed at every flip
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, Emin, Cmin, itmin = eng.extremal_opt(tau, iters, step)
        C1 = eng.get_config()
        Er, ar, st = eng.rrr_mc(beta, 500, 100)                    # the DoubleGraph rrrMC still works afterwards
    for r in range(R):
        ref = oracle.extremal_opt_quant(A, J, M, X.fourK, tau, iters, step, seed, C0.s[r], replica=r, form=form)
        assert (Es[r] == ref[0]).all() and (C1.s[r] == ref[1]).all()
        assert Emin[r] == ref[2] and (Cmin.s[r] == ref[3]).all() and itmin[r] == ref[4]


@pytest.mark.parametrize("kind,R,beta,iters,step,thr", [
    ("rrg10", 5, 2.0, 5000, 50, 0.5), ("rrg300", 7, 2.0, 8000, 200, 0.0), ("rrg300", 7, 1.0, 8000, 200, 1.0), ("rrg4096", 4, 2.0, 30000, 1000, 0.8),
    ("ea2x3", 6, 1.0, 4000, 40, 0.5), ("ea5x3", 5, 1.5, 9000, 300, 0.8), ("ea8x3", 3, 3.0, 20000, 500, 0.8), ("rrg300lev", 6, 1.5, 8000, 200, 0.8),
    ("rrg500k7", 3, 1.2, 6000, 200, 0.8),
])
def test_rrr_sparse_wave_kernel_equals_thread_kernel(pkg, oracle, kind, R, beta, iters, step, thr, monkeypatch):
    """rrrMC(SingleGraph) on GraphRRG / GraphEA has a wavefront-per-replica build (few replicas: the whole DeltaECache in LDS, the 2L sets
    as segments of one array) next to the thread-per-replica one (RRRMC_RRR_NO_WAVE=1).  Same chains bit for bit — repeated bonds
    (GraphEA L = 2), zero couplings of a level graph, K = 7, always-direct / always-staged, and with so little slack between the segments
    that they are re-spaced every few batches — and equal to the oracle."""
    seed = 515100 + len(kind) + R
    lev = None
    if kind == "rrg10":
        X, form = pkg.GraphRRG(10, 3, seed=seed), "rrg"
    elif kind == "rrg300":
        X, form = pkg.GraphRRG(300, 3, seed=seed), "rrg"
    elif kind == "rrg4096":
        X, form = pkg.GraphRRG(4096, 3, seed=seed), "rrg"
    elif kind == "rrg500k7":
        X, form = pkg.GraphRRG(500, 6, seed=seed), "rrg"
    elif kind == "rrg300lev":
        lev = (-1, 0, 1)
        X, form = pkg.GraphRRG(300, 3, lev, seed=seed), "rrg"
    else:
        X, form = pkg.GraphEA(int(kind[2]), 3, seed=seed), "ea"

    def run():
        with pkg.Engine(X, R) as eng:
            eng.seed(seed)
            eng.init_spins_random()
            C0 = eng.get_config()
            a = eng.rrr_mc(beta, iters, step, staged_thr=thr)
            c1 = eng.get_config().s.copy()
            b = eng.rrr_mc(beta, 333, 10, staged_thr=thr)          # continues the streams, rebuilds the cache
            c2 = eng.get_config().s.copy()
            k = eng.bkl_mc(beta, 20 * iters, 20 * step)            # bklMC through the same builds
            c3 = eng.get_config().s.copy()
            return C0, a[0], a[1], a[2], c1, b[0], b[1], b[2], c2, k[0], k[1], c3

    wave = run()
    monkeypatch.setenv("RRRMC_RRR_WAVE_SLACK", str(2 * 4 * 64 * 2 * (X.K + 1)))       # the smallest slack accepted: re-spacing every few batches
    tight = run()
    monkeypatch.delenv("RRRMC_RRR_WAVE_SLACK")
    monkeypatch.setenv("RRRMC_RRR_NO_WAVE", "1")
    thread = run()
    for x, y, w in zip(wave[1:], thread[1:], tight[1:]):
        assert (np.asarray(x) == np.asarray(y)).all() and (np.asarray(w) == np.asarray(y)).all()
    if lev is None:
        r = R - 1
        ref = oracle.rrr_sparse(X.A, X.J.astype(np.int32), beta, iters, step, seed, wave[0].s[r], replica=r, staged_thr=thr, form=form)
        assert (wave[1][r] == ref[0]).all() and (wave[4][r] == ref[1]).all() and wave[2][r] == ref[2] and wave[3][r] == ref[3]
