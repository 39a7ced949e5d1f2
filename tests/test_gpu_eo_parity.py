"""GPU parity for extremal_opt (tau-EO; src/RRRMC.jl:474-521, EOCache src/DeltaE.jl:412-555) on GraphRRG / GraphEA
(SURVEY.md §8f rank 4): trajectory samples, final configuration, Emin, Cmin and itmin equal the oracle's bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,R,tau,iters,step", [
    ("rrg10", 70, 1.3, 3000, 50),
    ("rrg128", 64, 1.3, 20000, 500),
    ("rrg4096", 6, 1.2, 30000, 1024),
    ("ea2x3", 16, 1.5, 2000, 64),          # double bonds; allΔE has a zero level
    ("ea4x3", 33, 1.3, 10000, 100),
])
def test_extremal_opt_bit_exact(pkg, oracle, kind, R, tau, iters, step):
    seed = 9090 + R
    if kind.startswith("rrg"):
        X, form = pkg.GraphRRG(int(kind[3:]), 3, seed=seed), "rrg"
    else:
        X, form = pkg.GraphEA(int(kind[2]), int(kind[4]), seed=seed), "ea"
    J = X.J.astype(np.int32)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, Emin, Cmin, itmin = eng.extremal_opt(tau, iters, step)
        C1 = eng.get_config()
        E1 = eng.energy()
    for r in range(R):
        Es_ref, ch_ref, Emin_ref, Cmin_ref, itmin_ref = oracle.extremal_opt_sparse(X.A, J, tau, iters, step, seed, C0.s[r], replica=r, form=form)
        assert (Es[r] == Es_ref).all() and (C1.s[r] == ch_ref).all()
        assert Emin[r] == Emin_ref and itmin[r] == itmin_ref and (Cmin.s[r] == Cmin_ref).all()
        assert E1[r] == oracle.sparse_energy(X.A, J, ch_ref)


def test_extremal_opt_front_end(pkg, oracle):
    seed = 3
    X = pkg.GraphRRG(64, 3, seed=seed)
    C, Emin, Cmin, itmin = pkg.extremal_opt(X, 1.3, 5000, step=1000, seed=seed, quiet=True, replicas=4)
    C0 = oracle.init_configs(seed, 0, 4, X.N)
    for r in range(4):
        ref = oracle.extremal_opt_sparse(X.A, X.J.astype(np.int32), 1.3, 5000, 1000, seed, C0[r], replica=r)
        assert (C.s[r] == ref[1]).all() and Emin[r] == ref[2] and (Cmin.s[r] == ref[3]).all() and itmin[r] == ref[4]
        assert Emin[r] == oracle.sparse_energy(X.A, X.J.astype(np.int32), Cmin.s[r])
