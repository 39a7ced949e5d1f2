"""GPU parity for GraphQuant over binary GraphSK slices — GraphQSKT (src/QAliases.jl:34-43), the graph of the reference's quantum
experiment scripts/scripts.jl:766-864 (test_QIsing: standardMC and rrrMC, Qenergy logged at every sample) and of
test/runtests.jl:79.  SURVEY.md §8a rows a9b x a10-a14.  Bit equality with the oracle (north_star: 1e-6 relative for Float64)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("Nk,M,Gamma,beta,R,iters,step,thr", [
    (10, 8, 0.5, 2.0, 8, 10000, 100, 0.5),      # runtests.jl:79 GraphQuant(10, 8, 0.5, 2.0, GraphSK, gen_J(10)); default staged_thr
    (10, 8, 0.5, 2.0, 8, 6000, 100, 0.0),       # always direct (every rejection is undone)
    (10, 8, 0.5, 2.0, 8, 6000, 100, 1.0),       # always staged
    (45, 7, 0.3, 1.0, 70, 12000, 250, 0.5),     # slices not word aligned; more than one 64-thread block
    (64, 16, 0.3, 2.0, 16, 20000, 500, 0.5),    # word-aligned slices
    (1024, 16, 0.3, 2.0, 3, 20000, 4096, 0.5),  # test_QIsing's geometry (N = 1024, M = 16, beta = 2, Gamma = 0.3)
])
def test_rrr_quant_sk_bit_exact(pkg, oracle, Nk, M, Gamma, beta, R, iters, step, thr):
    seed = 5426732438 + Nk
    X = pkg.GraphQSKT(Nk, M, Gamma, beta, seed=seed)
    assert X.fourK == oracle.quant_fourK(beta, Gamma, M) and (X.J == oracle.gen_sk_binary(Nk, seed)).all()
    Jb = X.J
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        E0 = eng.energy()
        Es, acc, staged = eng.rrr_mc(beta, iters, step, staged_thr=thr)
        C1 = eng.get_config()
        pos, sizes = eng.rrr_cache()
        E1 = eng.energy()
        Es2, acc2, staged2 = eng.rrr_mc(beta, iters // 4, step, staged_thr=thr)       # continues the streams
        C2 = eng.get_config()
    for r in range(R):
        assert E0[r] == oracle.quant_sk_energy(Jb, Nk, M, X.fourK, C0.s[r])[0]
        ref = oracle.rrr_mc_quant_sk(Jb, Nk, M, X.fourK, beta, iters, step, seed, C0.s[r], replica=r, staged_thr=thr, want_cache=True)
        assert np.allclose(Es[r], ref[0], rtol=1e-6, atol=1e-9)
        assert (Es[r] == ref[0]).all() and (C1.s[r] == ref[1]).all() and acc[r] == ref[2] and staged[r] == ref[3]
        assert (pos[r] == ref[4]).all() and (sizes[r] == ref[5]).all()
        assert E1[r] == oracle.quant_sk_energy(Jb, Nk, M, X.fourK, C1.s[r])[0]
    for r in (0, R - 1):
        # the second call restarts acc_rate at 0.5 and rebuilds the cache, like a second rrrMC(...; C0 = C) does
        ref2 = oracle.rrr_mc_quant_sk(Jb, Nk, M, X.fourK, beta, iters // 4, step, seed, C1.s[r], it0=iters, replica=r, staged_thr=thr)
        assert (Es2[r] == ref2[0]).all() and (C2.s[r] == ref2[1]).all() and acc2[r] == ref2[2]


@pytest.mark.parametrize("Nk,M,Gamma,beta,R,iters,step", [
    (10, 8, 0.5, 2.0, 40, 10000, 100),          # runtests.jl:79 under standardMC (:141-143)
    (45, 7, 0.3, 1.0, 70, 12000, 250),
    (1024, 16, 0.3, 2.0, 3, 40000, 4096),       # test_QIsing's ":met" leg
])
def test_standard_mc_quant_sk_bit_exact(pkg, oracle, Nk, M, Gamma, beta, R, iters, step):
    seed = 5426732438 + Nk + 1
    X = pkg.GraphQSKT(Nk, M, Gamma, beta, seed=seed)
    Jb = X.J
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc = eng.standard_mc(beta, iters, step)
        C1 = eng.get_config()
        Es2, acc2 = eng.standard_mc(beta, iters // 2, step)
        C2 = eng.get_config()
        E2 = eng.energy()
    for r in range(R):
        ref = oracle.standard_mc_quant_sk(Jb, Nk, M, X.fourK, beta, iters, step, seed, C0.s[r], replica=r)
        assert np.allclose(Es[r], ref[0], rtol=1e-6, atol=1e-9)
        assert (Es[r] == ref[0]).all() and (C1.s[r] == ref[1]).all() and acc[r] == ref[2]
        ref2 = oracle.standard_mc_quant_sk(Jb, Nk, M, X.fourK, beta, iters // 2, step, seed, ref[1], it0=iters, replica=r)
        assert (Es2[r] == ref2[0]).all() and (C2.s[r] == ref2[1]).all() and acc2[r] == ref2[2]
        assert E2[r] == oracle.quant_sk_energy(Jb, Nk, M, X.fourK, C2.s[r])[0]
    assert 0 < acc.sum() < R * iters


def test_quant_sk_lds_and_global_builds_agree(pkg, oracle, monkeypatch):
    seed = 777
    X = pkg.GraphQSKT(33, 5, 0.4, 1.2, seed=seed)          # N = 165: unaligned slices, odd sizes
    out = []
    for no_lds in ("0", "1"):
        monkeypatch.setenv("RRRMC_QUANT_NO_LDS", no_lds)
        with pkg.Engine(X, 9) as eng:
            eng.seed(seed)
            eng.init_spins_random()
            C0 = eng.get_config()
            a = eng.rrr_mc(1.2, 6000, 100)
            out.append((a[0], a[1], a[2], eng.get_config().s, eng.rrr_cache()[0], eng.rrr_cache()[1]))
    for u, v in zip(*out):
        assert (u == v).all()
    for r in range(9):
        ref = oracle.rrr_mc_quant_sk(X.J, 33, 5, X.fourK, 1.2, 6000, 100, seed, C0.s[r], replica=r)
        assert (out[0][0][r] == ref[0]).all() and out[0][1][r] == ref[2]


@pytest.mark.parametrize("Nk,M,R", [(10, 8, 7), (45, 5, 3), (1024, 16, 2)])
def test_quant_sk_observables(pkg, oracle, Nk, M, R):
    """Qenergy / transverse_mag / overlaps (QT.jl:113-122, 213-268): what test_QIsing's hook logs at every sample."""
    seed = 41337 + Nk
    beta, Gamma = 2.0, 0.3
    X = pkg.GraphQSKT(Nk, M, Gamma, beta, seed=seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        eng.rrr_mc(beta, 5000, 5000)
        C = eng.get_config()
        Q, tm, ov = eng.quant_observables()
    for r in range(R):
        Qr, tmr, ovr, *_ = oracle.quant_sk_observables(X.J, Nk, M, X.fourK, beta, Gamma, C.s[r])
        assert Q[r] == Qr and tm[r] == tmr and (ov[r] == ovr).all()


def test_quant_sk_errors(pkg):
    X = pkg.GraphQSKT(10, 8, 0.5, 2.0, seed=3)
    with pkg.Engine(X, 2) as eng:
        A = np.zeros((10, 3), np.int32)
        rc = pkg.lib().rrrmc_set_graph(eng._ctx, A, np.ones((10, 3), np.int8))
        assert rc != 0                                     # SK slices take rrrmc_set_couplings_bits
        bad = X.J.copy()
        bad[0, 0] |= np.uint64(1)                          # J[1][1] = 1: "diagonal entries of J must be 0" (SK.jl:38)
        assert pkg.lib().rrrmc_set_couplings_bits(eng._ctx, bad.reshape(-1)) == 1


@pytest.mark.parametrize("Nk,M,Gamma,beta,R,iters,step,thr", [
    (10, 8, 0.5, 2.0, 40, 12000, 100, None),         # test/runtests.jl:80: GraphQuant(10, 8, 0.5, 2.0, GraphSKNormal, SK.gen_J_gauss(10))
    (10, 8, 0.5, 2.0, 9, 6000, 250, 1.0),            # staged branch only
    (10, 8, 0.5, 2.0, 9, 6000, 250, 0.0),            # direct branch only: apply_move! and its undo through the slice's swap path
    (40, 5, 0.3, 1.2, 70, 5000, 100, None),
    (96, 12, 0.8, 1.0, 3, 3000, 500, None),
])
def test_graph_quant_over_sknormal_slices(pkg, oracle, Nk, M, Gamma, beta, R, iters, step, thr):
    """GraphQuant over GraphSKNormal slices (GraphQSKNormalT): every slice keeps its own Float64 lfields / lfields_last / move_last
    (SK.jl:212-276); rrrMC(X::DoubleGraph) and standardMC bit for bit against the oracle, incl. the DeltaECache state."""
    seed = 31000 + Nk * M
    X = pkg.GraphQSKNormalT(Nk, M, Gamma, beta, seed=seed)
    J = X.J
    kw = {} if thr is None else {"staged_thr": thr}
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        E0 = eng.energy()
        Es, acc, staged = eng.rrr_mc(beta, iters, step, **kw)
        C1 = eng.get_config()
        pos, sizes = eng.rrr_cache()
        E1 = eng.energy()
        Ess, accs = eng.standard_mc(beta, 2000, 100)
        C2 = eng.get_config()
    for r in sorted(set([0, 1, R // 2, R - 1])):
        assert E0[r] == oracle.quant_skn_energy(J, Nk, M, X.fourK, C0.s[r])
        ref = oracle.rrr_mc_quant_skn(J, Nk, M, X.fourK, beta, iters, step, seed, C0.s[r], replica=r, want_cache=True,
                                      **({} if thr is None else {"staged_thr": thr}))
        assert (Es[r] == ref[0]).all() and (C1.s[r] == ref[1]).all() and acc[r] == ref[2] and staged[r] == ref[3]
        assert (pos[r] == ref[4]).all() and (sizes[r] == ref[5]).all()
        assert E1[r] == oracle.quant_skn_energy(J, Nk, M, X.fourK, C1.s[r])
        std = oracle.standard_mc_quant_skn(J, Nk, M, X.fourK, beta, 2000, 100, seed, ref[1], it0=iters, replica=r)
        assert (Ess[r] == std[0]).all() and (C2.s[r] == std[1]).all() and accs[r] == std[2]


@pytest.mark.parametrize("kind,Nk,M,Gamma,beta,R", [
    ("skn", 10, 8, 0.5, 2.0, 12),            # test/runtests.jl:80 under bklMC / wtmMC / extremal_opt (:145-157)
    ("skb", 10, 8, 0.5, 2.0, 12),            # test/runtests.jl:79 (GraphQSKT)
    ("skn", 24, 5, 0.3, 1.0, 5),
    ("skb", 40, 4, 0.6, 1.5, 3),
])
def test_cont_samplers_on_graph_quant_over_dense_slices(pkg, oracle, kind, Nk, M, Gamma, beta, R):
    """bklMC, wtmMC and extremal_opt on a GraphQuant over dense slices (GraphQSKNormalT / GraphQSKT): the generic continuous-energy
    caches over all Nk M spins, every spin with the Trotter pair and the Nk - 1 other spins of its slice as neighbours
    (QT.jl:288-321, SK.jl:142,297)."""
    seed = 52000 + Nk * M
    X = pkg.GraphQSKNormalT(Nk, M, Gamma, beta, seed=seed) if kind == "skn" else pkg.GraphQSKT(Nk, M, Gamma, beta, seed=seed)
    kw = {"Jd": X.J} if kind == "skn" else {"Jb": X.J}
    iters, step, samples, eo_iters = 3000, 100, 12, 1200
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Eb, mb = eng.bkl_mc(beta, iters, step)
        C1 = eng.get_config()
        eng.seed(seed)
        eng.set_config(C0)
        Ew, mw, tw = eng.wtm_mc(beta, samples, step=2.0)
        C2 = eng.get_config()
        eng.seed(seed)
        eng.set_config(C0)
        Ee, Emin, Cmin, itmin = eng.extremal_opt(1.4, eo_iters, 100)
        C3 = eng.get_config()
        Er, ar, st = eng.rrr_mc(beta, 500, 100)               # the DoubleGraph rrrMC still works afterwards
    for r in sorted(set([0, R // 2, R - 1])):
        b = oracle.cont_quant_dense("bkl", Nk, M, X.fourK, beta, iters, step, seed, C0.s[r], replica=r, **kw)
        assert (Eb[r] == b[0]).all() and (C1.s[r] == b[1]).all() and mb[r] == b[2][0]
        w = oracle.cont_quant_dense("wtm", Nk, M, X.fourK, beta, samples, 1, seed, C0.s[r], replica=r, stepf=2.0, **kw)
        assert (Ew[r] == w[0]).all() and (C2.s[r] == w[1]).all() and mw[r] == w[2][0] and tw[r] == w[3]
        e = oracle.extremal_opt_quant_dense(Nk, M, X.fourK, 1.4, eo_iters, 100, seed, C0.s[r], replica=r, **kw)
        assert (Ee[r] == e[0]).all() and (C3.s[r] == e[1]).all() and Emin[r] == e[2] and (Cmin.s[r] == e[3]).all() and itmin[r] == e[4]


@pytest.mark.parametrize("slices,Nk,M,R", [("sk", 10, 8, 5), ("sk", 70, 6, 3), ("sk", 128, 5, 4), ("rrg", 96, 12, 6), ("ea2x3", 8, 6, 5)])
def test_quant_wave_builds_equal_thread_builds(pkg, slices, Nk, M, R, monkeypatch):
    """standardMC and rrrMC on a GraphQuant have two builds each: one wavefront per replica (few replicas; binary GraphSK slices: one
    word of the slice per lane, slices that are not word aligned included) and one thread per replica (RRRMC_QUANT_NO_WAVE=1).
    Same chains bit for bit, hooked (resumed) standardMC calls included."""
    seed, beta, Gamma = 90210 + Nk, 2.0, 0.4
    if slices == "sk":
        X = pkg.GraphQSKT(Nk, M, Gamma, beta, seed=seed)
    elif slices == "rrg":
        X = pkg.GraphQuant(pkg.GraphRRG(Nk, 3, seed=seed), M, Gamma, beta)
    else:
        X = pkg.GraphQuant(pkg.GraphEA(2, 3, seed=seed), M, Gamma, beta)

    def run():
        with pkg.Engine(X, R) as eng:
            eng.seed(seed)
            eng.init_spins_random()
            a = eng.standard_mc(beta, 5000, 100)
            c1 = eng.get_config().s.copy()
            b = eng.standard_mc(beta, 777, 7)               # continues the streams; a batch of 64 iterations ends inside the call
            c2 = eng.get_config().s.copy()
            c = eng.rrr_mc(beta, 3000, 100)
            c3 = eng.get_config().s.copy()
            return a[0], a[1], c1, b[0], b[1], c2, c[0], c[1], c3

    wave = run()
    monkeypatch.setenv("RRRMC_QUANT_NO_WAVE", "1")
    thread = run()
    for x, y in zip(wave, thread):
        assert (np.asarray(x) == np.asarray(y)).all()
    assert wave[1].sum() > 0 and wave[7].sum() > 0
