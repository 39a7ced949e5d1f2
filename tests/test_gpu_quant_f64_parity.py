"""GPU parity for GraphQuant over sparse Float64 slices — GraphQEAT = GraphQuant{fourK,GraphEANormal{twoD}} (src/QAliases.jl:50-83): every
slice a GraphEANormal (src/graphs/EA.jl:534-680) on ONE shared (A, J), its own LocalFields{Float64} with the exact undo path of
update_cache! (:613-653).  rrrMC(X::DoubleGraph) (src/RRRMC.jl:221-290) with delta_energy_residual = -lfields[i] / M (src/graphs/QT.jl:270-281),
standardMC, and the generic caches (bklMC / wtmMC / extremal_opt over all Nk M spins, neighbors = Trotter pair then the slice's, QT.jl:288-321).
Float64 energies: the north star's tolerance is 1e-6 relative; the kernels keep the reference's operation order, so bit equality is required."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 8426732438942


def qeat(pkg, L, D, M, Gamma=0.5, beta=2.0):
    return pkg.GraphQEAT(L, D, M, Gamma, beta, seed=SEED + L)


@pytest.mark.parametrize("L,D,M,R,iters,step,thr", [
    (4, 2, 8, 6, 6000, 100, 0.5),          # VERDICT r5 item 3's sizes
    (4, 2, 8, 6, 6000, 100, 0.0),          # always direct: every rejected move is undone through the slice's record
    (4, 2, 8, 6, 6000, 100, 1.0),          # always staged
    (8, 3, 16, 4, 20000, 500, 0.5),        # N = 8192
    (2, 3, 5, 5, 4000, 100, 0.5),          # L = 2: every neighbour twice in a row of A (two bonds, two couplings)
    (3, 1, 4, 70, 3000, 50, 0.5),          # a chain of 3 (K = 2), more than one 64-thread block of replicas
])
def test_rrr_qeat_bit_exact(pkg, oracle, L, D, M, R, iters, step, thr):
    X = qeat(pkg, L, D, M)
    assert X.f64_slices and X.fourK == oracle.quant_fourK(2.0, 0.5, M)
    A, J = X.X1.A, X.X1.J
    assert (np.abs(J) <= 2).all() and np.unique(np.abs(J)).size > X.X1.N // 2
    with pkg.Engine(X, R) as eng:
        eng.seed(SEED)
        eng.init_spins_random()
        C0 = eng.get_config()
        E0 = eng.energy()
        Es, acc, staged = eng.rrr_mc(2.0, iters, step, staged_thr=thr)
        C1 = eng.get_config()
        pos, sizes = eng.rrr_cache()
        Et = eng.run_energy()
        Es2, acc2, _ = eng.rrr_mc(2.0, iters // 2, step, staged_thr=thr)          # a second call starts from energy(X, C), continues the streams
        C2 = eng.get_config()
    for r in range(R):
        assert E0[r] == oracle.quant_spf_energy(A, J, M, X.fourK, C0.s[r])
        ref = oracle.rrr_mc_quant_spf(A, J, M, X.fourK, 2.0, iters, step, SEED, C0.s[r], replica=r, staged_thr=thr, want_cache=True)
        assert np.allclose(Es[r], ref[0], rtol=1e-6, atol=1e-9) and (Es[r] == ref[0]).all()
        assert (C1.s[r] == ref[1]).all() and acc[r] == ref[2] and staged[r] == ref[3]
        assert (pos[r] == ref[4]).all() and (sizes[r] == ref[5]).all()
        # the reference's invariant (test/runtests.jl:12-20): tracked E ~ energy(X, C)
        assert abs(Et[r] - oracle.quant_spf_energy(A, J, M, X.fourK, C1.s[r])) < 1e-9
        ref2 = oracle.rrr_mc_quant_spf(A, J, M, X.fourK, 2.0, iters // 2, step, SEED, ref[1], it0=iters, replica=r, staged_thr=thr)
        assert (Es2[r] == ref2[0]).all() and (C2.s[r] == ref2[1]).all() and acc2[r] == ref2[2]
    assert 0 < acc.sum() < R * iters or thr == 0.0


@pytest.mark.parametrize("L,D,M,R,iters,step", [(4, 2, 8, 40, 8000, 100), (8, 3, 16, 4, 30000, 1000), (2, 3, 5, 5, 4000, 100)])
def test_standard_mc_qeat_bit_exact(pkg, oracle, L, D, M, R, iters, step):
    X = qeat(pkg, L, D, M)
    A, J = X.X1.A, X.X1.J
    with pkg.Engine(X, R) as eng:
        eng.seed(SEED)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc = eng.standard_mc(2.0, iters, step)
        C1 = eng.get_config()
    for r in range(R):
        ref = oracle.standard_mc_quant_spf(A, J, M, X.fourK, 2.0, iters, step, SEED, C0.s[r], replica=r)
        assert (Es[r] == ref[0]).all() and (C1.s[r] == ref[1]).all() and acc[r] == ref[2]
    assert 0 < acc.sum() < R * iters


@pytest.mark.parametrize("L,D,M", [(4, 2, 8), (2, 3, 5)])
def test_generic_caches_on_qeat(pkg, oracle, L, D, M):
    """bklMC / wtmMC (DeltaECacheCont + DynamicSampler, THeap over all Nk M spins) and extremal_opt (EOCacheCont)"""
    X = qeat(pkg, L, D, M)
    A, J = X.X1.A, X.X1.J
    R, iters, step = 4, 3000, 100
    with pkg.Engine(X, R) as eng:
        eng.seed(SEED)
        eng.init_spins_random()
        C0 = eng.get_config()
        Eb, mb = eng.bkl_mc(2.0, iters, step)
        Cb = eng.get_config().s.copy()
        eng.set_config(C0)
        Ew, mw, tw = eng.wtm_mc(2.0, 20, 50.0)
        Cw = eng.get_config().s.copy()
        eng.set_config(C0)
        Ee, Emin, Cmin, itmin = eng.extremal_opt(1.3, 1500, 100)
        Ce = eng.get_config().s.copy()
    for r in range(R):
        ob = oracle.cont_quant_spf("bkl", A, J, M, X.fourK, 2.0, iters, step, SEED, C0.s[r], replica=r)
        assert (Eb[r] == ob[0]).all() and (Cb[r] == ob[1]).all() and mb[r] == ob[2][0]
        ow = oracle.cont_quant_spf("wtm", A, J, M, X.fourK, 2.0, 20, 1, SEED, C0.s[r], replica=r, stepf=50.0)
        assert (Ew[r] == ow[0]).all() and (Cw[r] == ow[1]).all() and mw[r] == ow[2][0] and tw[r] == ow[3]
        oe = oracle.extremal_opt_quant_spf(A, J, M, X.fourK, 1.3, 1500, 100, SEED, C0.s[r], it0=iters, replica=r)
        assert (Ee[r] == oe[0]).all() and (Ce[r] == oe[1]).all() and Emin[r] == oe[2] and (Cmin.s[r] == oe[3]).all() and itmin[r] == oe[4]


def test_qeat_constructors_and_multi_device(pkg, oracle, tmp_path):
    """GraphQEAT(fname, M, Γ, β) / GraphQEAT(X::GraphEANormal, M, Γ, β) (QAliases.jl:69-83), and the context over two shards"""
    X1 = pkg.GraphEANormal(4, 2, seed=SEED)
    f = tmp_path / "ea.txt"
    with open(f, "w") as fh:
        fh.write("type: test\nsize: 4\nname: t\n")
        for x in range(X1.N):
            for k in range(X1.K):
                if x < X1.A[x, k]:
                    fh.write("%d %d %r\n" % (x + 1, X1.A[x, k] + 1, float(X1.J[x, k])))
    Xa = pkg.GraphQEAT(str(f), 6, 0.7, 1.5)
    Xb = pkg.GraphQEAT(X1, 6, 0.7, 1.5)
    assert (Xa.X1.A == Xb.X1.A).all() and (Xa.X1.J == Xb.X1.J).all() and Xa.fourK == Xb.fourK and Xa.N == 96
    R = 64
    outs = []
    for kw in ({}, {"devices": [0, 0]}):
        with pkg.Engine(Xb, R, **kw) as eng:
            eng.seed(SEED)
            eng.init_spins_random()
            C0 = eng.get_config().s.copy()
            Es, acc, st = eng.rrr_mc(1.5, 3000, 100)
            outs.append((C0, Es, acc, st, eng.get_config().s.copy()))
    for u, v in zip(*outs):
        assert (u == v).all()
    ref = oracle.rrr_mc_quant_spf(Xb.X1.A, Xb.X1.J, 6, Xb.fourK, 1.5, 3000, 100, SEED, outs[0][0][40], replica=40)
    assert (outs[1][1][40] == ref[0]).all() and (outs[1][4][40] == ref[1]).all()
    with pytest.raises(pkg.RRRMCError):
        with pkg.Engine(Xb, 2) as eng:
            eng.seed(1); eng.init_spins_random()
            eng.quant_observables()
