"""GPU parity for the colour-parallel (checkerboard) sweeps on sparse +-J models: bit-exact against the oracle's
sequential restatement, plus size-independent properties at BASELINE.json config 4's full lattice (L = 64, D = 3)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,R,beta,sweeps,step", [
    (("ea", 4, 3), 32, 1.0, 40, 5),
    (("ea", 6, 3), 40, 0.6, 25, 1),
    (("ea", 8, 2), 64, 2.0, 30, 7),
    (("ea", 2, 3), 32, 1.0, 20, 2),          # L = 2: doubled neighbours, still two-colourable
    (("rrg", 100, 3), 32, 1.0, 30, 3),       # any graph with a proper colouring (greedy)
    (("ea", 24, 3), 32, 1.0, 6, 2),          # N = 13824: beyond the 16-bit byte offsets of the normal LDS kernel (WIDE build territory)
])
def test_colored_sweeps_bit_exact(pkg, oracle, kind, R, beta, sweeps, step):
    seed = 777 + kind[1]
    if kind[0] == "ea":
        X = pkg.GraphEA(kind[1], kind[2], seed=seed)
        color = pkg.checkerboard_coloring(kind[1], kind[2])
        assert (color == oracle.checkerboard_coloring(kind[1], kind[2])).all()
    else:
        X = pkg.GraphRRG(kind[1], kind[2], seed=seed)
        color = oracle.greedy_coloring(X.A)
    A, J = X.A, X.J.astype(np.int32)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        eng.set_coloring(color)
        C0 = eng.get_config()
        E0 = eng.energy()
        Es = eng.colored_sweeps(beta, sweeps, step)
        C1 = eng.get_config()
        E1 = eng.energy()
        Es2 = eng.colored_sweeps(beta, 3, 1)              # continues the SWEEP stream
        C2 = eng.get_config()
    for r in list(range(min(R, 6))) + [R - 1]:
        assert E0[r] == oracle.sparse_energy(A, J, C0.s[r])
        Es_ref, ch_ref, _ = oracle.colored_sweeps_sparse(A, J, color, beta, sweeps, step, seed, C0.s[r], replica=r)
        assert (Es[r] == Es_ref).all() and (C1.s[r] == ch_ref).all()
        assert E1[r] == oracle.sparse_energy(A, J, C1.s[r])
        Es2_ref, ch2_ref, _ = oracle.colored_sweeps_sparse(A, J, color, beta, 3, 1, seed, ch_ref, sweep0=sweeps, replica=r)
        assert (Es2[r] == Es2_ref).all() and (C2.s[r] == ch2_ref).all()


def test_improper_coloring_is_rejected(pkg):
    X = pkg.GraphEA(4, 2, seed=1)
    with pkg.Engine(X, 32) as eng:
        with pytest.raises(pkg.RRRMCError) as e:
            eng.set_coloring(np.zeros(16, np.int32))
        assert e.value.code == 1 and "proper" in str(e.value)
    with pytest.raises(ValueError):
        pkg.checkerboard_coloring(3, 3)


def test_config4_lattice_properties(pkg, oracle):
    """GraphEA L = 64, D = 3 (N = 262144): the oracle is too slow for a full comparison, so check one replica group over two
    sweeps against it and size-independent properties on more replicas: energies are even multiples consistent with
    allΔE = (0,4,8,12), the beta = inf limit never raises the energy (random-site standardMC at this size: test_gpu_bign_parity)."""
    L_, D = 64, 3
    seed = 64
    X = pkg.GraphEA(L_, D, seed=seed)
    color = pkg.checkerboard_coloring(L_, D)
    A, J = X.A, X.J.astype(np.int32)
    with pkg.Engine(X, 64) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        eng.set_coloring(color)
        C0 = eng.get_config()
        Es = eng.colored_sweeps(1.0, 2, 1)
        C1 = eng.get_config()
        E1 = eng.energy()
        Ecold = eng.colored_sweeps(1e6, 6, 1)             # zero temperature: energy is non-increasing
        assert (np.diff(Ecold, axis=1) <= 0).all()
        assert (Ecold % 2 == 0).all()
    for r in (0, 33):
        Es_ref, ch_ref, _ = oracle.colored_sweeps_sparse(A, J, color, 1.0, 2, 1, seed, C0.s[r], replica=r)
        assert (Es[r] == Es_ref).all() and (C1.s[r] == ch_ref).all()
        assert E1[r] == oracle.sparse_energy(A, J, C1.s[r])


def test_rrr_and_bkl_on_the_config4_lattice(pkg, oracle):
    """rrrMC(SingleGraph), bklMC, wtmMC and extremal_opt on GraphEA(64, 3) (N = 262 144 > 65 535: the DeltaECache's / EOCache's set
    members and positions and the THeap's ids are 32-bit there) against the oracle, incl. the move counts; and on a lattice just
    above the 16-bit limit."""
    for L_, iters in ((64, 3000), (41, 6000)):            # 41^3 = 68 921
        seed = 6400 + L_
        X = pkg.GraphEA(L_, 3, seed=seed)
        A, J = X.A, X.J.astype(np.int32)
        with pkg.Engine(X, 34) as eng:
            eng.seed(seed)
            eng.init_spins_random()
            C0 = eng.get_config()
            Es, acc, staged = eng.rrr_mc(1.0, iters, 500)
            C1 = eng.get_config()
            eng.seed(seed)
            eng.set_config(C0)
            Eb, mb = eng.bkl_mc(1.0, iters, 500)
            C2 = eng.get_config()
            eng.seed(seed)
            eng.set_config(C0)
            Ew, mw, tw = eng.wtm_mc(1.0, 8, 200.0)
            C3 = eng.get_config()
            eng.seed(seed)
            eng.set_config(C0)
            Ee, Emin, Cmin, itmin = eng.extremal_opt(1.3, iters, 500)
            C4 = eng.get_config()
        for r in (0, 33):
            ref = oracle.rrr_sparse(A, J, 1.0, iters, 500, seed, C0.s[r], replica=r, form="ea")
            assert (Es[r] == ref[0]).all() and (C1.s[r] == ref[1]).all() and acc[r] == ref[2] and staged[r] == ref[3]
            b = oracle.rrr_sparse(A, J, 1.0, iters, 500, seed, C0.s[r], replica=r, form="ea", bkl=True)
            assert (Eb[r] == b[0]).all() and (C2.s[r] == b[1]).all() and mb[r] == b[2]
            w = oracle.wtm_mc_sparse(A, J, 1.0, 8, 200.0, seed, C0.s[r], replica=r, form="ea")
            assert (Ew[r] == w[0]).all() and (C3.s[r] == w[1]).all() and mw[r] == w[2] and tw[r] == w[3]
            assert mw[r] > 100
            e = oracle.extremal_opt_sparse(A, J, 1.3, iters, 500, seed, C0.s[r], replica=r, form="ea")
            assert (Ee[r] == e[0]).all() and (C4.s[r] == e[1]).all() and Emin[r] == e[2] and (Cmin.s[r] == e[3]).all() and itmin[r] == e[4]
