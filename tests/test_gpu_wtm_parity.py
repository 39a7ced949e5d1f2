"""GPU parity for wtmMC (waiting-time method; src/RRRMC.jl:376-426, src/WaitingTimes.jl) on GraphRRG / GraphEA (SURVEY.md §8f
rank 4): energies, final configuration, number of moves and the final global time (Float64) equal the oracle's bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,R,beta,samples,step", [
    ("rrg10", 70, 2.0, 500, 1.0),           # test/runtests.jl:36 GraphRRG(10, 3)
    ("rrg128", 64, 1.0, 3000, 1.0),         # BASELINE config 1 geometry
    ("rrg4096", 6, 2.0, 20000, 4.0),        # config 2 geometry
    ("ea2x3", 16, 1.0, 400, 1.0),           # runtests.jl:46 GraphEA(2, 3): double bonds
    ("ea4x3", 33, 1.5, 2000, 2.5),
])
def test_wtm_bit_exact(pkg, oracle, kind, R, beta, samples, step):
    seed = 4040 + R
    if kind.startswith("rrg"):
        X, form = pkg.GraphRRG(int(kind[3:]), 3, seed=seed), "rrg"
    else:
        L, D = int(kind[2]), int(kind[4])
        X, form = pkg.GraphEA(L, D, seed=seed), "ea"
    J = X.J.astype(np.int32)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, moves, t = eng.wtm_mc(beta, samples, step)
        C1 = eng.get_config()
        E1 = eng.energy()
        Es2, moves2, t2 = eng.wtm_mc(beta, 100, step)          # a second call: new waiting times (call index 1)
        C2 = eng.get_config()
    assert Es.shape == (R, samples)
    for r in range(R):
        Es_ref, ch_ref, m_ref, t_ref, Ef = oracle.wtm_mc_sparse(X.A, J, beta, samples, step, seed, C0.s[r], replica=r, form=form)
        assert (Es[r] == Es_ref).all() and (C1.s[r] == ch_ref).all() and moves[r] == m_ref and t[r] == t_ref and E1[r] == Ef
        Es2_ref, ch2_ref, m2_ref, t2_ref, _ = oracle.wtm_mc_sparse(X.A, J, beta, 100, step, seed, ch_ref, call=1, replica=r, form=form)
        assert (Es2[r] == Es2_ref).all() and (C2.s[r] == ch2_ref).all() and moves2[r] == m2_ref and t2[r] == t2_ref


def test_wtmMC_front_end(pkg, oracle):
    seed = 12
    X = pkg.GraphRRG(64, 3, seed=seed)
    Es, C = pkg.wtmMC(X, 1.0, 1000, step=1.0, seed=seed, quiet=True, replicas=3)
    C0 = oracle.init_configs(seed, 0, 3, X.N)
    for r in range(3):
        ref = oracle.wtm_mc_sparse(X.A, X.J.astype(np.int32), 1.0, 1000, 1.0, seed, C0[r], replica=r)
        assert (Es[r] == ref[0]).all() and (C.s[r] == ref[1]).all()
    Eq, Cq = pkg.wtmMC(pkg.GraphQSKT(10, 8, 0.5, 2.0, seed=seed), 1.0, 10, seed=seed, quiet=True)   # dense slices too (parity: test_gpu_quant_sk_parity)
    assert Eq.shape[-1] == 10 and np.isfinite(Eq).all()
