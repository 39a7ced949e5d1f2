"""The thread-per-replica samplers beyond N = 65 535 spins: set members / positions (DeltaECache, EOCache, DeltaECacheCont's heap) are
32-bit there.  One lattice just above the limit, GraphEA-shaped with L = 41, D = 3 (N = 68 921, K = 6), for every graph family
that runs through those kernels: continuous couplings (GraphEANormal: rrrMC, bklMC, wtmMC), a discretised DoubleGraph
(GraphEANormalDiscretized: rrrMC(DoubleGraph), standardMC) and general levels (GraphEA{Int,(-1,0,1)}: rrrMC, bklMC, wtmMC,
extremal_opt).  The +-J graphs at this size and at L = 64: test_gpu_colored_sweeps.py.  Two replicas of each run are compared
with the oracle bit for bit (src/RRRMC.jl:149-219, :221-290, :311-359, :376-426, :474-521)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

L_, D = 41, 3
REPS = (0, 33)


def test_cont_samplers_above_16_bit(pkg, oracle):
    seed = 6841
    X = pkg.GraphEANormal(L_, D, seed=seed)
    assert X.N == 68921
    beta = 1.5
    with pkg.Engine(X, 34) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc, staged = eng.rrr_mc(beta, 4000, 500)
        C1 = eng.get_config()
        eng.seed(seed)
        eng.set_config(C0)
        Eb, mb = eng.bkl_mc(beta, 3000, 500)
        C2 = eng.get_config()
        eng.seed(seed)
        eng.set_config(C0)
        Ew, mw, tw = eng.wtm_mc(beta, 8, 200.0)
        C3 = eng.get_config()
        with pytest.raises(pkg.RRRMCError) as e:            # EOCacheCont's tie keys address the site with 16 bits of the Philox tag
            eng.extremal_opt(1.3, 10, 1)
        assert e.value.code == 3
    for r in REPS:
        a = oracle.cont_sparse("rrr", X.A, X.J, beta, 4000, 500, seed, C0.s[r], replica=r, form="ea")
        assert (Es[r] == a[0]).all() and (C1.s[r] == a[1]).all() and acc[r] == a[2][0] and staged[r] == a[2][1]
        b = oracle.cont_sparse("bkl", X.A, X.J, beta, 3000, 500, seed, C0.s[r], replica=r, form="ea")
        assert (Eb[r] == b[0]).all() and (C2.s[r] == b[1]).all() and mb[r] == b[2][0]
        w = oracle.cont_sparse("wtm", X.A, X.J, beta, 8, 1, seed, C0.s[r], replica=r, stepf=200.0, form="ea")
        assert (Ew[r] == w[0]).all() and (C3.s[r] == w[1]).all() and mw[r] == w[2][0] and tw[r] == w[3]
        assert mw[r] > 100


def test_double_graph_above_16_bit(pkg, oracle):
    seed = 6842
    lev = (-1, 0, 1)
    X = pkg.GraphEANormalDiscretized(L_, D, lev, seed=seed)
    levu, mul, div = oracle.dfloat_units(lev)
    dJ, rJ = X.dJ, X.rJ
    beta = 1.5
    with pkg.Engine(X, 34) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        E0 = eng.energy()
        Es, acc, staged = eng.rrr_mc(beta, 4000, 500, staged_thr=0.5)
        C1 = eng.get_config()
        pos, sizes = eng.rrr_cache()
        eng.seed(seed)
        eng.set_config(C0)
        Es_s, acc_s = eng.standard_mc(beta, 4000, 500)
        C2 = eng.get_config()
    nl = len(oracle.all_delta_e(X.K, levu))
    for r in REPS:
        assert E0[r] == oracle.dbl_energy(X.A, dJ, rJ, C0.s[r], form="ea", mul=mul, div=div)
        ref = oracle.rrr_double_sparse(X.A, dJ, rJ, levu, beta, 4000, 500, seed, C0.s[r], replica=r, staged_thr=0.5, form="ea", mul=mul, div=div)
        assert (Es[r] == ref[0]).all() and (C1.s[r] == ref[1]).all() and acc[r] == ref[2] and staged[r] == ref[3]
        assert (pos[r] == ref[4]).all() and (sizes[r, :2 * nl] == ref[5]).all()
        s = oracle.standard_mc_dbl(X.A, dJ, rJ, beta, 4000, 500, seed, C0.s[r], replica=r, form="ea", mul=mul, div=div)
        assert (Es_s[r] == s[0]).all() and (C2.s[r] == s[1]).all() and acc_s[r] == s[2]


def test_levels_above_16_bit(pkg, oracle):
    seed = 6843
    lev = (-1, 0, 1)
    X = pkg.GraphEA(L_, D, lev, seed=seed)
    units, mul, div = pkg.level_units(lev)
    assert X.model_kind == 7
    beta = 1.4
    with pkg.Engine(X, 34) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc, staged = eng.rrr_mc(beta, 4000, 500)
        C1 = eng.get_config()
        eng.seed(seed)
        eng.set_config(C0)
        Eb, mb = eng.bkl_mc(beta, 4000, 500)
        C2 = eng.get_config()
        eng.seed(seed)
        eng.set_config(C0)
        Ew, mw, tw = eng.wtm_mc(beta, 8, 200.0)
        C3 = eng.get_config()
        eng.seed(seed)
        eng.set_config(C0)
        Ee, Emin, Cmin, itmin = eng.extremal_opt(1.3, 3000, 500)
        C4 = eng.get_config()
        eng.seed(seed)
        eng.set_config(C0)
        Es_s, acc_s = eng.standard_mc(beta, 4000, 500)
        C5 = eng.get_config()
    for r in REPS:
        a = oracle.rrr_sparse(X.A, X.J, beta, 4000, 500, seed, C0.s[r], replica=r, form="ea", lev=units, mul=mul, div=div)
        assert (Es[r] == X.energy_value(a[0])).all() and (C1.s[r] == a[1]).all() and acc[r] == a[2] and staged[r] == a[3]
        b = oracle.rrr_sparse(X.A, X.J, beta, 4000, 500, seed, C0.s[r], replica=r, form="ea", bkl=True, lev=units, mul=mul, div=div)
        assert (Eb[r] == X.energy_value(b[0])).all() and (C2.s[r] == b[1]).all() and mb[r] == b[2]
        w = oracle.wtm_mc_sparse(X.A, X.J, beta, 8, 200.0, seed, C0.s[r], replica=r, form="ea", mul=mul, div=div)
        assert (Ew[r] == X.energy_value(w[0])).all() and (C3.s[r] == w[1]).all() and mw[r] == w[2] and tw[r] == w[3]
        assert mw[r] > 100
        e = oracle.extremal_opt_sparse(X.A, X.J, 1.3, 3000, 500, seed, C0.s[r], replica=r, form="ea", lev=units)
        assert (Ee[r] == X.energy_value(e[0])).all() and (C4.s[r] == e[1]).all()
        assert Emin[r] == X.energy_value(e[2]) and (Cmin.s[r] == e[3]).all() and itmin[r] == e[4]
        s = oracle.standard_mc_lev(X.A, X.J, beta, 4000, 500, seed, C0.s[r], replica=r, form="ea", mul=mul, div=div)
        assert (Es_s[r] == X.energy_value(s[0])).all() and (C5.s[r] == s[1]).all() and acc_s[r] == s[2]


def test_graph_quant_above_16_bit(pkg, oracle):
    """GraphQuant with N = Nk M > 65 535 (Nk = 1100, M = 60: 66 000 spins): the thread-per-replica builds with 32-bit set members /
    positions and the slice of a spin by division; rrrMC (incl. the cache), standardMC and bklMC against the oracle, GraphRRG slices;
    rrrMC over binary GraphSK slices as well."""
    seed, beta, Gamma = 660001, 2.0, 0.5
    Nk, M, R = 1100, 60, 3
    X1 = pkg.GraphRRG(Nk, 3, seed=seed)
    X = pkg.GraphQuant(X1, M, Gamma, beta)
    assert X.N == 66000
    A, J = X1.A, X1.J.astype(np.int32)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        E0 = eng.energy()
        Es, acc, staged = eng.rrr_mc(beta, 4000, 500)
        C1 = eng.get_config()
        pos, sizes = eng.rrr_cache()
        eng.seed(seed)
        eng.set_config(C0)
        Es_s, acc_s = eng.standard_mc(beta, 6000, 500)
        C2 = eng.get_config()
        eng.seed(seed)
        eng.set_config(C0)
        Eb, mb = eng.bkl_mc(beta, 3000, 500)
        C3 = eng.get_config()
    for r in (0, R - 1):
        assert E0[r] == oracle.quant_energy(A, J, M, X.fourK, C0.s[r])[0]
        ref = oracle.rrr_mc_quant(A, J, M, X.fourK, beta, 4000, 500, seed, C0.s[r], replica=r, want_cache=True)
        assert (Es[r] == ref[0]).all() and (C1.s[r] == ref[1]).all() and acc[r] == ref[2] and staged[r] == ref[3]
        assert (pos[r] == ref[4]).all() and (sizes[r] == ref[5]).all()
        s = oracle.standard_mc_quant(A, J, M, X.fourK, beta, 6000, 500, seed, C0.s[r], replica=r)
        assert (Es_s[r] == s[0]).all() and (C2.s[r] == s[1]).all() and acc_s[r] == s[2]
        b = oracle.cont_quant("bkl", A, J, M, X.fourK, beta, 3000, 500, seed, C0.s[r], replica=r)
        assert (Eb[r] == b[0]).all() and (C3.s[r] == b[1]).all() and mb[r] == b[2][0]
    Xs = pkg.GraphQSKT(Nk, M, Gamma, beta, seed=seed)
    with pkg.Engine(Xs, 2) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc, staged = eng.rrr_mc(beta, 3000, 500)
        C1 = eng.get_config()
    ref = oracle.rrr_mc_quant_sk(Xs.J, Nk, M, Xs.fourK, beta, 3000, 500, seed, C0.s[1], replica=1)
    assert (Es[1] == ref[0]).all() and (C1.s[1] == ref[1]).all() and acc[1] == ref[2]
