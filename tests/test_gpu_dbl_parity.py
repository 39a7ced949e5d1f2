"""GPU parity for rrrMC(X::DoubleGraph) on GraphRRGNormalDiscretized / GraphEANormalDiscretized with integer levels
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
])
def test_rrr_double_graph_bit_exact(pkg, oracle, kind, lev, R, beta, iters, step, thr):
    seed = 777000 + len(kind) + R
    X, form = _graph(pkg, kind, lev, seed)
    cJ = oracle.gen_couplings_gauss(X.A, seed)
    dJ, rJ = oracle.discretize(cJ, lev)
    assert (X.cJ == cJ).all() and (X.dJ == dJ).all() and (X.rJ == rJ).all()
    nl = len(oracle.all_delta_e(X.K, lev))
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        E0 = eng.energy()
        Es, acc, staged = eng.rrr_mc(beta, iters, step, staged_thr=thr)
        C1 = eng.get_config()
        pos, sizes = eng.rrr_cache()
        E1 = eng.energy()
        with pytest.raises(pkg.RRRMCError):
            eng.standard_mc(beta, 10, 1)                      # not wired for this DoubleGraph
    assert (C0.s == oracle.init_configs(seed, 0, R, X.N)).all()
    for r in range(R):
        assert E0[r] == oracle.dbl_energy(X.A, dJ, rJ, C0.s[r], form=form)
        Es_ref, ch_ref, acc_ref, st_ref, pos_ref, sizes_ref = oracle.rrr_double_sparse(X.A, dJ, rJ, lev, beta, iters, step, seed, C0.s[r],
                                                                                      replica=r, staged_thr=thr, form=form)
        assert np.allclose(Es[r], Es_ref, rtol=1e-6, atol=1e-9)          # north-star tolerance
        assert (Es[r] == Es_ref).all()                                    # ... and bit for bit
        assert (C1.s[r] == ch_ref).all() and acc[r] == acc_ref and staged[r] == st_ref
        assert (pos[r] == pos_ref).all() and (sizes[r, :2 * nl] == sizes_ref).all() and sizes[r, 2 * nl:].sum() == 0
        assert E1[r] == oracle.dbl_energy(X.A, dJ, rJ, C1.s[r], form=form)


def test_rrrMC_front_end_double_graph(pkg, oracle):
    seed = 99
    X = pkg.GraphRRGNormalDiscretized(64, 3, (-1, 0, 1), seed=seed)
    Es, C = pkg.rrrMC(X, 1.3, 5000, step=100, seed=seed, quiet=True, replicas=5)
    C0 = oracle.init_configs(seed, 0, 5, X.N)
    for r in range(5):
        ref = oracle.rrr_double_sparse(X.A, X.dJ, X.rJ, X.LEV, 1.3, 5000, 100, seed, C0[r], replica=r)
        assert (Es[r] == ref[0]).all() and (C.s[r] == ref[1]).all()
    with pytest.raises(NotImplementedError):
        pkg.GraphRRGNormalDiscretized(10, 3, (-1.5, 0.0, 1.5), seed=seed)      # DFloat64 levels: not covered
    with pytest.raises(ValueError):
        pkg.GraphRRGNormalDiscretized(10, 3, (-1, 1, 1), seed=seed)
