"""GPU parity for stand-alone GraphRRG{ET,LEV,K} / GraphEA{ET,LEV,2D} with levels other than (-1, 1) — Int, Float64 (-> DFloat64) and
rational levels, the graph families of test/runtests.jl:36-60 — under every sampler of the reference's test loop (:140-163):
standardMC, rrrMC (staged_thr 0.5 / 0.0 / 1.0), bklMC, wtmMC, plus extremal_opt.  SURVEY.md §8a rows a7/a8.  Bit-exact against the
oracle: energies are integer level units on both sides."""
from fractions import Fraction

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F = Fraction


def _graph(pkg, kind, lev, seed):
    if kind == "rrg10":
        return pkg.GraphRRG(10, 3, lev, seed=seed), "rrg"            # runtests.jl:37-40
    if kind == "rrg300":
        return pkg.GraphRRG(300, 3, lev, seed=seed), "rrg"
    if kind == "rrg4096k4":
        return pkg.GraphRRG(4096, 4, lev, seed=seed), "rrg"
    if kind == "ea2x3":
        return pkg.GraphEA(2, 3, lev, seed=seed), "ea"               # runtests.jl:47-50 (L = 2: double bonds)
    if kind == "ea3x2":
        return pkg.GraphEA(3, 2, lev, seed=seed), "ea"               # runtests.jl:57-60
    if kind == "circ200k8":                                          # +-J with K > 7: beyond the bit-plane kernels, runs as a level graph
        import oracle as O                                           # (gen_RRG's pairing retries practically never succeed at K = 8: a circulant)
        A = np.sort((np.arange(200)[:, None] + np.array([-4, -3, -2, -1, 1, 2, 3, 4])[None, :]) % 200, axis=1).astype(np.int32)
        return pkg.GraphRRG.from_AJ(A, O.gen_couplings(A, seed, pkg.level_units(lev)[0]), lev), "rrg"
    if kind == "ea4x4":
        return pkg.GraphEA(4, 4, lev, seed=seed), "ea"               # K = 8
    if kind == "ea3x5":
        return pkg.GraphEA(3, 5, lev, seed=seed), "ea"               # K = 10
    if kind == "ea6x3":
        return pkg.GraphEA(6, 3, lev, seed=seed), "ea"
    raise KeyError(kind)


CASES = [
    ("rrg10", (-1, 0, 1), 40), ("rrg10", (-1.0, 0.0, 1.0), 8), ("rrg10", (-1.0, 0, 1), 8), ("rrg10", (F(-1), F(0), F(1)), 8),
    ("ea2x3", (-1, 0, 1), 16), ("ea2x3", (-1.0, 0.0, 1.0), 8), ("ea3x2", (-1, 0, 1), 33), ("ea3x2", (F(-1), F(0), F(1)), 8),
    ("rrg300", (-2, -1, 1, 2), 64), ("rrg300", (-1.5, -0.5, 0.5, 1.5), 32), ("rrg300", (F(-1, 3), F(1, 3), F(1)), 16),
    ("rrg4096k4", (-1, 0, 1), 6), ("ea6x3", (-0.75, 0.0, 0.75), 40), ("rrg300", (-1.0, 1.0), 16),
    ("circ200k8", (-1, 1), 33), ("ea4x4", (-1, 1), 16), ("ea3x5", (-1, 1), 16),
]


def _setup(pkg, oracle, kind, lev, seed):
    X, form = _graph(pkg, kind, lev, seed)
    units, mul, div = pkg.level_units(lev)
    assert X.model_kind == 7 and X.LEV == units and (X.lev_mul, X.lev_div) == (mul, div)
    J = oracle.gen_couplings(X.A, seed, units)
    assert (X.J == J).all()
    return X, form, units, mul, div


@pytest.mark.parametrize("kind,lev,R", CASES)
def test_levels_standard_mc_bit_exact(pkg, oracle, kind, lev, R):
    seed = 424200 + len(kind) + R
    X, form, units, mul, div = _setup(pkg, oracle, kind, lev, seed)
    beta, iters, step = 1.3, 6000, 50
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        E0 = eng.energy()
        Es, acc = eng.standard_mc(beta, iters, step)
        C1 = eng.get_config()
        Es2, acc2 = eng.standard_mc(beta, iters // 3, step)
        C2 = eng.get_config()
        E2 = eng.energy()
    assert Es.dtype == (np.int64 if div == 1.0 else np.float64)
    assert (C0.s == oracle.init_configs(seed, 0, R, X.N)).all()
    for r in range(R):
        assert E0[r] == X.energy_value(oracle.sparse_energy(X.A, X.J.astype(np.int32), C0.s[r]))
        Es_ref, ch, a = oracle.standard_mc_lev(X.A, X.J, beta, iters, step, seed, C0.s[r], replica=r, form=form, mul=mul, div=div)
        assert (Es[r] == X.energy_value(Es_ref)).all() and (C1.s[r] == ch).all() and acc[r] == a
        Es_ref2, ch2, a2 = oracle.standard_mc_lev(X.A, X.J, beta, iters // 3, step, seed, ch, it0=iters, replica=r, form=form, mul=mul, div=div)
        assert (Es2[r] == X.energy_value(Es_ref2)).all() and (C2.s[r] == ch2).all() and acc2[r] == a2
        assert E2[r] == X.energy_value(oracle.sparse_energy(X.A, X.J.astype(np.int32), C2.s[r]))
    assert 0 < acc.sum() < R * iters


@pytest.mark.parametrize("thr", [0.5, 0.0, 1.0])
@pytest.mark.parametrize("kind,lev,R", CASES)
def test_levels_rrr_bit_exact(pkg, oracle, kind, lev, R, thr):
    if thr != 0.5 and R > 16:
        R = 16
    seed = 434300 + len(kind) + R
    X, form, units, mul, div = _setup(pkg, oracle, kind, lev, seed)
    beta, iters, step = 1.6, 5000, 50
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc, staged = eng.rrr_mc(beta, iters, step, staged_thr=thr)
        C1 = eng.get_config()
    for r in range(R):
        ref = oracle.rrr_sparse(X.A, X.J, beta, iters, step, seed, C0.s[r], replica=r, staged_thr=thr, form=form, lev=units, mul=mul, div=div)
        assert (Es[r] == X.energy_value(ref[0])).all() and (C1.s[r] == ref[1]).all() and acc[r] == ref[2] and staged[r] == ref[3]


@pytest.mark.parametrize("kind,lev,R", CASES)
def test_levels_bkl_wtm_eo_bit_exact(pkg, oracle, kind, lev, R):
    R = min(R, 24)
    seed = 444400 + len(kind) + R
    X, form, units, mul, div = _setup(pkg, oracle, kind, lev, seed)
    beta = 1.4
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es_b, mv_b = eng.bkl_mc(beta, 20000, 500)
        C1 = eng.get_config()
        eng.seed(seed)
        eng.set_config(C0)
        Es_w, mv_w, t_w = eng.wtm_mc(beta, 30, step=2.0)
        C2 = eng.get_config()
        eng.seed(seed)
        eng.set_config(C0)
        Es_e, Emin, Cmin, itmin = eng.extremal_opt(1.3, 4000, 100)
        C3 = eng.get_config()
    for r in range(R):
        b = oracle.rrr_sparse(X.A, X.J, beta, 20000, 500, seed, C0.s[r], replica=r, form=form, bkl=True, lev=units, mul=mul, div=div)
        assert (Es_b[r] == X.energy_value(b[0])).all() and (C1.s[r] == b[1]).all() and mv_b[r] == b[2]
        w = oracle.wtm_mc_sparse(X.A, X.J, beta, 30, 2.0, seed, C0.s[r], call=0, replica=r, form=form, mul=mul, div=div)
        assert (Es_w[r] == X.energy_value(w[0])).all() and (C2.s[r] == w[1]).all() and mv_w[r] == w[2] and t_w[r] == w[3]
        e = oracle.extremal_opt_sparse(X.A, X.J, 1.3, 4000, 100, seed, C0.s[r], replica=r, form=form, lev=units)
        assert (Es_e[r] == X.energy_value(e[0])).all() and (C3.s[r] == e[1]).all()
        assert Emin[r] == X.energy_value(e[2]) and (Cmin.s[r] == e[3]).all() and itmin[r] == e[4]


def test_levels_unit_float_equals_int(pkg):
    """DFloat64 levels (-1.0, 0.0, 1.0) have t = (-10^5, 0, 10^5): the same trajectories as Int levels (-1, 0, 1), energies as Float64."""
    seed = 77
    Xi = pkg.GraphRRG(120, 3, (-1, 0, 1), seed=seed)
    Xf = pkg.GraphRRG(120, 3, (-1.0, 0.0, 1.0), seed=seed)
    assert (Xi.J == Xf.J).all() and Xf.lev_mul == 100000 and pkg.all_delta_e(Xi) == (0, 2, 4, 6) and pkg.all_delta_e(Xf) == (0.0, 2.0, 4.0, 6.0)
    out = []
    for X in (Xi, Xf):
        with pkg.Engine(X, 12) as eng:
            eng.seed(seed)
            eng.init_spins_random()
            a = eng.standard_mc(1.1, 3000, 30)
            b = eng.rrr_mc(1.1, 3000, 30)
            c = eng.wtm_mc(1.1, 20)
            out.append((a[0], a[1], b[0], b[1], b[2], c[0], c[1], c[2], eng.get_config().s))
    for u, v in zip(*out):
        assert (np.asarray(u) == np.asarray(v)).all()
    assert out[0][0].dtype == np.int64 and out[1][0].dtype == np.float64


def test_levels_misuse(pkg):
    X = pkg.GraphRRG(10, 3, (-1, 0, 1), seed=3)
    with pytest.raises(ValueError):
        pkg.GraphRRG.from_AJ(X.A, np.full(X.A.shape, 2, np.int8), (-1, 0, 1))        # RRG.jl:130
    with pytest.raises(ValueError):
        pkg.GraphRRG(10, 3, (-1, 1, 1), seed=3)                                        # RRG.jl:100
    Y = pkg.GraphRRG.from_AJ(X.A, X.J, (-1, 0, 1))
    assert Y.model_kind == 7 and (Y.J == X.J).all()
    with pkg.Engine(X, 4) as eng:
        eng.seed(1)
        eng.init_spins_random()
        with pytest.raises(pkg.RRRMCError):
            eng.fields()                                                               # no field cache for this model
        with pytest.raises(pkg.RRRMCError):
            eng.colored_sweeps(1.0, 2)
