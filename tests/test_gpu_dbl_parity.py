"""GPU parity for rrrMC(X::DoubleGraph) and standardMC on GraphRRGNormalDiscretized / GraphEANormalDiscretized with Int or DFloat64 levels
(src/RRRMC.jl:221-290, src/graphs/RRG.jl:285-500, src/graphs/EA.jl:311-532).  north_star tolerance for
Float64 models is 1e-6 relative; the kernel keeps the reference's operation order, so we additionally require bit equality."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _graph(pkg, kind, lev, seed):
    if kind == "rrg10":
        return pkg.GraphRRGNormalDiscretized(10, 3, lev, seed=seed), "rrg"      # test/runtests.jl:41
    if kind == "rrg300k3":
        return pkg.GraphRRGNormalDiscretized(300, 3, lev, seed=seed), "rrg"
    if kind == "rrg4096":
        return pkg.GraphRRGNormalDiscretized(4096, 3, lev, seed=seed), "rrg"
    if kind == "ea2x3":
        return pkg.GraphEANormalDiscretized(2, 3, lev, seed=seed), "ea"         # runtests.jl:51 (L = 2: double bonds)
    if kind == "ea3x2":
        return pkg.GraphEANormalDiscretized(3, 2, lev, seed=seed), "ea"         # runtests.jl:61
    if kind == "ea8x3":
        return pkg.GraphEANormalDiscretized(8, 3, lev, seed=seed), "ea"
    raise KeyError(kind)


@pytest.mark.parametrize("kind,lev,R,beta,iters,step,thr", [
    ("rrg10", (-1, 0, 1), 70, 2.0, 6000, 50, 0.5),
    ("rrg10", (-1, 0, 1), 8, 2.0, 3000, 50, 0.0),        # runtests.jl:150 always direct (every rejection goes through the undo path)
    ("rrg10", (-1, 0, 1), 8, 2.0, 3000, 50, 1.0),        # runtests.jl:155 always staged
    ("rrg10", (-1, 1), 8, 1.0, 3000, 50, 0.5),
    ("rrg300k3", (-2, -1, 1, 2), 64, 1.5, 20000, 500, 0.5),
    ("rrg4096", (-1, 0, 1), 6, 2.0, 30000, 1024, 0.5),
    ("ea2x3", (-1, 0, 1), 16, 1.0, 4000, 64, 0.5),
    ("ea3x2", (-1, 0, 1), 33, 1.5, 5000, 100, 0.5),
    ("ea8x3", (-1, 0, 1), 64, 2.5, 20000, 1000, 0.5),
    ("rrg10", (-1.0, 0.0, 1.0), 8, 2.0, 3000, 50, 0.5),                 # DFloat64 levels (RRG.jl:324)
    ("rrg300k3", (-1.5, -0.5, 0.5, 1.5), 64, 1.5, 20000, 500, 0.5),
    ("rrg300k3", (-0.3, 0.3), 16, 3.0, 20000, 500, 0.5),
    ("ea3x2", (-1.25, 0.0, 1.25), 33, 1.5, 5000, 100, 0.5),
    ("ea8x3", (-0.8, 0.0, 0.8), 32, 2.5, 20000, 1000, 0.0),
])
def test_rrr_double_graph_bit_exact(pkg, oracle, kind, lev, R, beta, iters, step, thr):
    seed = 777000 + len(kind) + R
    X, form = _graph(pkg, kind, lev, seed)
    cJ = oracle.gen_couplings_gauss(X.A, seed)
    lev, mul, div = oracle.dfloat_units(lev)            # Int levels: (lev, 1, 1.0); Float64 levels -> DFloat64 units
    assert X.LEV == lev and X.lev_mul == mul and X.lev_div == div
    dJ, rJ = oracle.discretize(cJ, lev, mul, div)
    assert (X.cJ == cJ).all() and (X.dJ == dJ).all() and (X.rJ == rJ).all()
    nl = len(oracle.all_delta_e(X.K, lev))
    sc = dict(mul=mul, div=div)
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
        assert E0[r] == oracle.dbl_energy(X.A, dJ, rJ, C0.s[r], form=form, **sc)
        Es_ref, ch_ref, acc_ref, st_ref, pos_ref, sizes_ref = oracle.rrr_double_sparse(X.A, dJ, rJ, lev, beta, iters, step, seed, C0.s[r],
                                                                                      replica=r, staged_thr=thr, form=form, **sc)
        assert np.allclose(Es[r], Es_ref, rtol=1e-6, atol=1e-9)          # north-star tolerance
        assert (Es[r] == Es_ref).all()                                    # ... and bit for bit
        assert (C1.s[r] == ch_ref).all() and acc[r] == acc_ref and staged[r] == st_ref
        assert (pos[r] == pos_ref).all() and (sizes[r, :2 * nl] == sizes_ref).all() and sizes[r, 2 * nl:].sum() == 0
        assert E1[r] == oracle.dbl_energy(X.A, dJ, rJ, C1.s[r], form=form, **sc)


@pytest.mark.parametrize("kind,lev,R,beta,iters,step", [
    ("rrg10", (-1, 0, 1), 70, 2.0, 6000, 50),
    ("rrg300k3", (-2, -1, 1, 2), 64, 1.5, 20000, 500),
    ("rrg300k3", (-1.5, -0.5, 0.5, 1.5), 33, 1.0, 20000, 500),
    ("rrg4096", (-0.75, 0.0, 0.75), 6, 2.0, 30000, 1024),
    ("ea2x3", (-1, 0, 1), 16, 1.0, 4000, 64),
    ("ea8x3", (-1.25, 0.0, 1.25), 64, 2.5, 20000, 1000),
])
def test_standard_mc_double_graph_bit_exact(pkg, oracle, kind, lev, R, beta, iters, step):
    """standardMC (RRRMC.jl:81-127) on the DoubleGraphs: delta_energy = convert(Float64, dE0 + dE1) (RRG.jl:493-497); two calls in a
    row continue the streams (it0) and re-run energy(X, C) like the reference's sampler does at entry."""
    seed = 555000 + len(kind) + R
    X, form = _graph(pkg, kind, lev, seed)
    lev, mul, div = oracle.dfloat_units(lev)
    sc = dict(mul=mul, div=div)
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
            eng.rrr_cache()                                   # no class sets after standardMC
    for r in range(R):
        Es_ref, ch_ref, acc_ref = oracle.standard_mc_dbl(X.A, X.dJ, X.rJ, beta, iters, step, seed, C0.s[r], replica=r, form=form, **sc)
        assert np.allclose(Es[r], Es_ref, rtol=1e-6, atol=1e-9)          # north-star tolerance
        assert (Es[r] == Es_ref).all() and (C1.s[r] == ch_ref).all() and acc[r] == acc_ref
        Es_ref2, ch_ref2, acc_ref2 = oracle.standard_mc_dbl(X.A, X.dJ, X.rJ, beta, iters // 2, step, seed, ch_ref, it0=iters, replica=r,
                                                            form=form, **sc)
        assert (Es2[r] == Es_ref2).all() and (C2.s[r] == ch_ref2).all() and acc2[r] == acc_ref2
        assert E2[r] == oracle.dbl_energy(X.A, X.dJ, X.rJ, C2.s[r], form=form, **sc)


@pytest.mark.parametrize("kind,lev,R,beta", [
    ("rrg10", (-1, 0, 1), 40, 2.0),                       # runtests.jl:41 under bklMC / wtmMC (:145-151)
    ("rrg300k3", (-1.5, -0.5, 0.5, 1.5), 33, 1.0),        # DFloat64 levels
    ("ea2x3", (-1, 0, 1), 16, 1.0),                       # runtests.jl:51: double bonds, neighbors = uA
    ("ea8x3", (-1.25, 0.0, 1.25), 20, 1.5),
])
def test_bkl_and_wtm_on_double_graphs(pkg, oracle, kind, lev, R, beta):
    """A DoubleGraph is not a DiscrGraph: bklMC and wtmMC build the continuous-energy caches over the whole graph (DeltaE.jl:315,
    WaitingTimes.jl) with delta_energy = convert(Float64, dE0 + dE1) (RRG.jl:493-497)."""
    seed = 888000 + len(kind) + R
    X, form = _graph(pkg, kind, lev, seed)
    units, mul, div = oracle.dfloat_units(lev)
    sc = dict(mul=mul, div=div)
    iters, step, samples = 20000, 500, 40
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
        Er, ar, st = eng.rrr_mc(beta, 2000, 100)             # the DoubleGraph rrrMC still works after the other samplers
    for r in range(R):
        b = oracle.cont_double("bkl", X.A, X.dJ, X.rJ, beta, iters, step, seed, C0.s[r], replica=r, form=form, **sc)
        assert (Eb[r] == b[0]).all() and (C1.s[r] == b[1]).all() and mb[r] == b[2][0]
        assert E1[r] == oracle.dbl_energy(X.A, X.dJ, X.rJ, C1.s[r], form=form, **sc)
        w = oracle.cont_double("wtm", X.A, X.dJ, X.rJ, beta, samples, 1, seed, C0.s[r], replica=r, stepf=2.0, form=form, **sc)
        assert (Ew[r] == w[0]).all() and (C2.s[r] == w[1]).all() and mw[r] == w[2][0] and tw[r] == w[3]
        ref = oracle.rrr_double_sparse(X.A, X.dJ, X.rJ, units, beta, 2000, 100, seed, w[1], replica=r, form=form, **sc)
        assert (Er[r] == ref[0]).all() and ar[r] == ref[2]


def test_dfloat_unit_levels_equal_int_levels(pkg):
    """Float64 levels (-1.0, 0.0, 1.0) -> DFloat64 t = (-10^5, 0, 10^5): every promoted value t / 10^5 is the integer itself, so the
    trajectories equal those of Int levels (-1, 0, 1) bit for bit."""
    seed = 31
    Xi = pkg.GraphRRGNormalDiscretized(200, 3, (-1, 0, 1), seed=seed)
    Xf = pkg.GraphRRGNormalDiscretized(200, 3, (-1.0, 0.0, 1.0), seed=seed)
    assert Xf.lev_mul == 100000 and Xf.lev_div == 100000.0 and Xf.LEV == (-1, 0, 1) and (Xi.dJ == Xf.dJ).all() and (Xi.rJ == Xf.rJ).all()
    out = []
    for X in (Xi, Xf):
        with pkg.Engine(X, 16) as eng:
            eng.seed(seed)
            eng.init_spins_random()
            a = eng.rrr_mc(1.7, 5000, 50)
            b = eng.standard_mc(1.7, 5000, 50)
            out.append((a[0], a[1], a[2], b[0], b[1], eng.get_config().s))
    for u, v in zip(*out):
        assert (u == v).all()


def test_rrrMC_front_end_double_graph(pkg, oracle):
    seed = 99
    X = pkg.GraphRRGNormalDiscretized(64, 3, (-1, 0, 1), seed=seed)
    Es, C = pkg.rrrMC(X, 1.3, 5000, step=100, seed=seed, quiet=True, replicas=5)
    C0 = oracle.init_configs(seed, 0, 5, X.N)
    for r in range(5):
        ref = oracle.rrr_double_sparse(X.A, X.dJ, X.rJ, X.LEV, 1.3, 5000, 100, seed, C0[r], replica=r)
        assert (Es[r] == ref[0]).all() and (C.s[r] == ref[1]).all()
    with pytest.raises(NotImplementedError):
        pkg.GraphRRGNormalDiscretized(10, 3, (-1.00001, 0.0, 1.27), seed=seed)  # DFloat64 units beyond the 8-bit coupling table
    Es, C = pkg.standardMC(X, 1.3, 5000, step=100, seed=seed, quiet=True, replicas=5)
    for r in range(5):
        ref = oracle.standard_mc_dbl(X.A, X.dJ, X.rJ, 1.3, 5000, 100, seed, C0[r], replica=r)
        assert (Es[r] == ref[0]).all() and (C.s[r] == ref[1]).all()
    with pytest.raises(ValueError):
        pkg.GraphRRGNormalDiscretized(10, 3, (-1, 1, 1), seed=seed)
