"""GPU parity for extremal_opt on the graphs that are not DiscrGraphs (src/RRRMC.jl:474-521 with the generic EOCacheCont,
src/DeltaE.jl:557-635): GraphRRGNormal / GraphEANormal and the discretised DoubleGraphs.  Energy samples, final
configuration, Emin, Cmin, itmin equal the oracle's bit for bit (north_star's Float64 tolerance is 1e-6 relative)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(pkg, oracle, X, form, R, tau, iters, step, seed, J=None, **dbl):
    J = X.J if J is None else J
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, Emin, Cmin, itmin = eng.extremal_opt(tau, iters, step)
        C1 = eng.get_config()
        E1 = eng.energy()
        Es2, Emin2, Cmin2, itmin2 = eng.extremal_opt(tau, iters // 2, step)        # continues the streams from C1
        C2 = eng.get_config()
    for r in range(R):
        ref = oracle.extremal_opt_cont(X.A, J, tau, iters, step, seed, C0.s[r], replica=r, form=form, **dbl)
        assert np.allclose(Es[r], ref[0], rtol=1e-6, atol=1e-9)
        assert (Es[r] == ref[0]).all() and (C1.s[r] == ref[1]).all()
        assert Emin[r] == ref[2] and (Cmin.s[r] == ref[3]).all() and itmin[r] == ref[4]
        ref2 = oracle.extremal_opt_cont(X.A, J, tau, iters // 2, step, seed, ref[1], it0=iters, replica=r, form=form, **dbl)
        assert (Es2[r] == ref2[0]).all() and (C2.s[r] == ref2[1]).all() and Emin2[r] == ref2[2] and itmin2[r] == ref2[4]
    return E1, C1


@pytest.mark.parametrize("kind,R,tau,iters,step", [
    ("rrg10", 70, 1.3, 3000, 50),            # test/runtests.jl:40 x :161-163
    ("rrg300", 64, 1.3, 10000, 500),
    ("rrg4096", 5, 1.2, 4000, 512),
    ("ea2x3", 16, 1.5, 2000, 64),            # runtests.jl:50: double bonds, neighbors = uA
    ("ea5x3", 33, 1.3, 6000, 100),
])
def test_extremal_opt_cont_bit_exact(pkg, oracle, kind, R, tau, iters, step):
    seed = 717000 + len(kind) + R
    if kind.startswith("rrg"):
        X, form = pkg.GraphRRGNormal(int(kind[3:]), 3, seed=seed), "rrg"
    else:
        X, form = pkg.GraphEANormal(int(kind[2]), int(kind[4]), seed=seed), "ea"
    E1, C1 = _check(pkg, oracle, X, form, R, tau, iters, step, seed)
    for r in range(R):
        assert E1[r] == oracle.spf_energy(X.A, X.J, C1.s[r], form=form)


@pytest.mark.parametrize("kind,lev,R", [
    ("rrg10", (-1, 0, 1), 40),                            # runtests.jl:41
    ("rrg300", (-1.5, -0.5, 0.5, 1.5), 33),               # DFloat64 levels
    ("ea2x3", (-1, 0, 1), 16),                            # runtests.jl:51
])
def test_extremal_opt_on_double_graphs(pkg, oracle, kind, lev, R):
    seed = 727000 + len(kind) + R
    if kind.startswith("rrg"):
        X, form = pkg.GraphRRGNormalDiscretized(int(kind[3:]), 3, lev, seed=seed), "rrg"
    else:
        X, form = pkg.GraphEANormalDiscretized(int(kind[2]), int(kind[4]), lev, seed=seed), "ea"
    units, mul, div = oracle.dfloat_units(lev)
    E1, C1 = _check(pkg, oracle, X, form, R, 1.3, 4000, 100, seed, J=X.rJ, dJ=X.dJ, mul=mul, div=div)
    for r in range(R):
        assert E1[r] == oracle.dbl_energy(X.A, X.dJ, X.rJ, C1.s[r], form=form, mul=mul, div=div)


def test_extremal_opt_cont_with_ties(pkg, oracle):
    """Integer-valued Float64 couplings: long runs of equal delta_energy, re-ordered by fresh per-move keys at every move
    (rankshuffle!, DeltaE.jl:611-634)."""
    seed = 737
    A = oracle.gen_rrg(60, 3, seed)
    J = oracle.gen_couplings(A, seed).astype(np.float64)
    X = pkg.GraphRRGNormal.from_AJ(A, J)
    _check(pkg, oracle, X, "rrg", 24, 1.4, 1500, 25, seed)
    A = oracle.gen_ea(3, 2)
    J = oracle.gen_couplings(A, seed, (-1, 0, 1)).astype(np.float64) * 0.5
    X = pkg.GraphEANormal.from_AJ(A, J)
    _check(pkg, oracle, X, "ea", 8, 1.3, 1000, 10, seed)


def test_extremal_opt_cont_front_end(pkg, oracle):
    seed = 31
    X = pkg.GraphRRGNormal(64, 3, seed=seed)
    C, Emin, Cmin, itmin = pkg.extremal_opt(X, 1.3, 3000, step=1000, seed=seed, quiet=True, replicas=4)
    C0 = oracle.init_configs(seed, 0, 4, X.N)
    for r in range(4):
        ref = oracle.extremal_opt_cont(X.A, X.J, 1.3, 3000, 1000, seed, C0[r], replica=r)
        assert (C.s[r] == ref[1]).all() and Emin[r] == ref[2] and (Cmin.s[r] == ref[3]).all() and itmin[r] == ref[4]
        assert abs(Emin[r] - oracle.spf_energy(X.A, X.J, Cmin.s[r])) < 1e-9          # tracked (E += dE) vs recomputed: runtests.jl:12-20


@pytest.mark.parametrize("kind,R,iters", [("rrg700", 70, 3000), ("ea2x3", 16, 1500), ("ties", 24, 1200), ("dbl", 20, 2000)])
def test_extremal_opt_cont_wave_and_thread_builds_agree(pkg, oracle, monkeypatch, kind, R, iters):
    """N <= 13 000 runs one wavefront per replica with the ranking in LDS; RRRMC_EO_NO_WAVE=1 runs the thread-per-replica build (the
    one larger graphs get).  Same walks; replicas 0 and R-1 against the oracle."""
    seed = 747000 + R
    dbl = {}
    if kind == "rrg700":
        X, form = pkg.GraphRRGNormal(700, 3, seed=seed), "rrg"
    elif kind == "ea2x3":
        X, form = pkg.GraphEANormal(2, 3, seed=seed), "ea"
    elif kind == "ties":
        A = oracle.gen_rrg(90, 4, seed)
        X, form = pkg.GraphRRGNormal.from_AJ(A, oracle.gen_couplings(A, seed).astype(np.float64)), "rrg"
    else:
        X, form = pkg.GraphRRGNormalDiscretized(150, 3, (-1.5, -0.5, 0.5, 1.5), seed=seed), "rrg"
        units, mul, div = oracle.dfloat_units((-1.5, -0.5, 0.5, 1.5))
        dbl = dict(dJ=X.dJ, mul=mul, div=div)
    J = X.rJ if dbl else X.J
    out = []
    for no_wave in ("0", "1"):
        monkeypatch.setenv("RRRMC_EO_NO_WAVE", no_wave)
        with pkg.Engine(X, R) as eng:
            eng.seed(seed)
            eng.init_spins_random()
            C0 = eng.get_config()
            Es, Emin, Cmin, itmin = eng.extremal_opt(1.3, iters, 50)
            out.append((Es, Emin, Cmin.s, itmin, eng.get_config().s, eng.energy()))
    for u, v in zip(*out):
        assert (u == v).all()
    for r in (0, R - 1):
        ref = oracle.extremal_opt_cont(X.A, J, 1.3, iters, 50, seed, C0.s[r], replica=r, form=form, **dbl)
        assert (out[0][0][r] == ref[0]).all() and (out[0][4][r] == ref[1]).all() and out[0][1][r] == ref[2] and out[0][3][r] == ref[4]


@pytest.mark.parametrize("binary,N,R,tau,iters,step", [
    (False, 10, 40, 1.3, 3000, 50),           # runtests.jl:67 GraphSKNormal(10) under extremal_opt (:161-164)
    (True, 10, 40, 1.3, 3000, 50),            # runtests.jl:66 GraphSK(10): integer fields, ties at every move
    (False, 100, 33, 1.2, 1500, 100),
    (True, 100, 33, 1.5, 1500, 100),
    (False, 1024, 3, 1.3, 600, 64),           # BASELINE config 3's size
    (True, 333, 4, 1.3, 500, 50),
])
def test_extremal_opt_on_sk_models(pkg, oracle, binary, N, R, tau, iters, step):
    """EOCacheCont on the dense SK models: every spin is a neighbour, the whole ranking is rebuilt at every flip (a bitonic sort per
    move, one wavefront per replica)."""
    seed = 757000 + N + int(binary)
    X = pkg.GraphSK(N, seed=seed) if binary else pkg.GraphSKNormal(N, seed=seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, Emin, Cmin, itmin = eng.extremal_opt(tau, iters, step)
        C1 = eng.get_config()
        E1 = eng.energy()
        Es2, Emin2, Cmin2, itmin2 = eng.extremal_opt(tau, iters // 2, step)
        C2 = eng.get_config()
        Es3, acc3 = eng.standard_mc(1.0, 1000, 100)               # the Metropolis kernel still works afterwards
    energy = (lambda c: oracle.skb_energy(X.J, c)) if binary else (lambda c: oracle.skn_energy(X.J, c))
    for r in range(R):
        ref = oracle.extremal_opt_sk(X.J, tau, iters, step, seed, C0.s[r], replica=r, binary=binary)
        assert np.allclose(Es[r], ref[0], rtol=1e-6, atol=1e-9)
        assert (Es[r] == ref[0]).all() and (C1.s[r] == ref[1]).all()
        assert Emin[r] == ref[2] and (Cmin.s[r] == ref[3]).all() and itmin[r] == ref[4]
        assert E1[r] == energy(ref[1])
        ref2 = oracle.extremal_opt_sk(X.J, tau, iters // 2, step, seed, ref[1], it0=iters, replica=r, binary=binary)
        assert (Es2[r] == ref2[0]).all() and (C2.s[r] == ref2[1]).all() and Emin2[r] == ref2[2] and itmin2[r] == ref2[4]
