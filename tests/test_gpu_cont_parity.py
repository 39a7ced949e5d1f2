"""GPU parity for the continuous-energy samplers on GraphRRGNormal / GraphEANormal — the reference's second experiment
(scripts/scripts.jl:152-281 test_RRGCont): rrrMC(X::SingleGraph) (src/RRRMC.jl:149-219), bklMC (:311-359), wtmMC (:376-426) over
DeltaECacheCont + DynamicSampler / THeap.  north_star tolerance for Float64 models is 1e-6 relative; we require bit equality."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _graph(pkg, kind, seed):
    if kind == "rrg10":
        return pkg.GraphRRGNormal(10, 3, seed=seed), "rrg"          # test/runtests.jl:40
    if kind == "rrg300":
        return pkg.GraphRRGNormal(300, 3, seed=seed), "rrg"
    if kind == "rrg4096":
        return pkg.GraphRRGNormal(4096, 3, seed=seed), "rrg"
    if kind == "ea2x3":
        return pkg.GraphEANormal(2, 3, seed=seed), "ea"             # runtests.jl:50: double bonds
    if kind == "ea5x3":
        return pkg.GraphEANormal(5, 3, seed=seed), "ea"
    raise KeyError(kind)


@pytest.mark.parametrize("kind,R,beta,iters,step,thr", [
    ("rrg10", 70, 2.0, 5000, 50, 0.8),
    ("rrg10", 8, 2.0, 3000, 50, 0.0),          # always direct: every rejection goes through the undo path
    ("rrg10", 8, 2.0, 3000, 50, 1.0),          # always staged
    ("rrg300", 64, 1.5, 20000, 500, 0.8),
    ("rrg4096", 5, 2.0, 20000, 1024, 0.8),
    ("ea2x3", 16, 1.0, 4000, 64, 0.8),
    ("ea5x3", 33, 1.5, 10000, 100, 0.8),
])
def test_rrr_cont_bit_exact(pkg, oracle, kind, R, beta, iters, step, thr):
    seed = 313000 + len(kind) + R
    X, form = _graph(pkg, kind, seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc, staged = eng.rrr_mc(beta, iters, step, staged_thr=thr)
        C1 = eng.get_config()
        E1 = eng.energy()
        Es2, acc2, staged2 = eng.rrr_mc(beta, 1000, 100, staged_thr=thr)          # continues the streams
        C2 = eng.get_config()
    for r in range(R):
        Es_ref, ch_ref, st, _ = oracle.cont_sparse("rrr", X.A, X.J, beta, iters, step, seed, C0.s[r], replica=r, staged_thr=thr, form=form)
        assert np.allclose(Es[r], Es_ref, rtol=1e-6, atol=1e-9)
        assert (Es[r] == Es_ref).all() and (C1.s[r] == ch_ref).all() and acc[r] == st[0] and staged[r] == st[1]
        assert abs(E1[r] - oracle.spf_energy(X.A, X.J, ch_ref, form=form)) == 0
        Es2_ref, ch2_ref, st2, _ = oracle.cont_sparse("rrr", X.A, X.J, beta, 1000, 100, seed, ch_ref, it0=iters, replica=r, staged_thr=thr, form=form)
        assert (Es2[r] == Es2_ref).all() and (C2.s[r] == ch2_ref).all() and acc2[r] == st2[0]


@pytest.mark.parametrize("kind,mode", [("rrg300", "rrr"), ("rrg4096", "rrr"), ("ea5x3", "rrr"), ("rrg300", "bkl"), ("ea5x3", "bkl")])
def test_cont_wave_kernel_equals_thread_kernel(pkg, oracle, monkeypatch, kind, mode):
    """cont_wave_kernel (one wavefront per replica, the top of the DynamicSampler's tree in LDS, the bottom in 16-element blocks) against
    cont_sparse_kernel (RRRMC_CONT_NO_WAVE = 1: one thread per replica, the tree level by level in memory): energies, counts, configurations
    bit for bit over several refresh! periods (max(N, 100) calls of setindex!, DynamicSamplers.jl:163-165), calls chained."""
    seed = 818000 + len(kind)
    X, form = _graph(pkg, kind, seed)
    R, beta = 7, 2.0
    iters = 30000 if mode == "bkl" else 6000
    outs = []
    for env in (None, "1"):
        if env is None:
            monkeypatch.delenv("RRRMC_CONT_NO_WAVE", raising=False)
        else:
            monkeypatch.setenv("RRRMC_CONT_NO_WAVE", env)
        with pkg.Engine(X, R) as eng:
            eng.seed(seed)
            eng.init_spins_random()
            a = eng.rrr_mc(beta, iters, 100) if mode == "rrr" else eng.bkl_mc(beta, iters, 500)
            b = eng.rrr_mc(beta, 777, 7, staged_thr=1.0) if mode == "rrr" else eng.bkl_mc(beta, 5000, 50)
            outs.append((a, b, eng.get_config().s.copy(), eng.energy()))
    for u, w in zip(outs[0][0] + outs[0][1], outs[1][0] + outs[1][1]):
        assert (np.asarray(u) == np.asarray(w)).all()
    assert (outs[0][2] == outs[1][2]).all() and (outs[0][3] == outs[1][3]).all()


@pytest.mark.parametrize("kind,R,beta,iters,step", [
    ("rrg10", 40, 2.0, 20000, 100), ("rrg300", 64, 2.0, 50000, 1000), ("ea2x3", 16, 1.5, 8000, 64), ("rrg4096", 4, 3.0, 200000, 4096),
])
def test_bkl_cont_bit_exact(pkg, oracle, kind, R, beta, iters, step):
    seed = 515000 + len(kind) + R
    X, form = _graph(pkg, kind, seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, moves = eng.bkl_mc(beta, iters, step)
        C1 = eng.get_config()
    assert Es.shape == (R, iters // step)
    for r in range(R):
        Es_ref, ch_ref, st, _ = oracle.cont_sparse("bkl", X.A, X.J, beta, iters, step, seed, C0.s[r], replica=r, form=form)
        assert len(Es_ref) == iters // step
        assert (Es[r] == Es_ref).all() and (C1.s[r] == ch_ref).all() and moves[r] == st[0]


@pytest.mark.parametrize("kind,R,beta,samples,step", [
    ("rrg10", 40, 2.0, 500, 1.0), ("rrg300", 64, 1.5, 3000, 2.0), ("ea2x3", 16, 1.0, 400, 1.0), ("rrg4096", 4, 2.0, 20000, 4.0),
])
def test_wtm_cont_bit_exact(pkg, oracle, kind, R, beta, samples, step):
    seed = 717000 + len(kind) + R
    X, form = _graph(pkg, kind, seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, moves, t = eng.wtm_mc(beta, samples, step)
        C1 = eng.get_config()
        Es2, moves2, t2 = eng.wtm_mc(beta, 100, step)
    for r in range(R):
        Es_ref, ch_ref, st, t_ref = oracle.cont_sparse("wtm", X.A, X.J, beta, samples, 1, seed, C0.s[r], replica=r, stepf=step, form=form)
        assert (Es[r] == Es_ref).all() and (C1.s[r] == ch_ref).all() and moves[r] == st[0] and t[r] == t_ref
        Es2_ref, _, st2, t2_ref = oracle.cont_sparse("wtm", X.A, X.J, beta, 100, 1, seed, ch_ref, call=1, replica=r, stepf=step, form=form)
        assert (Es2[r] == Es2_ref).all() and moves2[r] == st2[0] and t2[r] == t2_ref


@pytest.mark.parametrize("N,R,beta,iters,step", [(10, 40, 2.0, 20000, 100), (64, 33, 1.5, 30000, 500), (300, 8, 2.0, 60000, 2000)])
def test_bkl_skn_bit_exact(pkg, oracle, N, R, beta, iters, step):
    """bklMC on GraphSKNormal (test/runtests.jl:67 GraphSKNormal(10)): DeltaECacheCont with all N - 1 neighbours per move."""
    seed = 919000 + N
    X = pkg.GraphSKNormal(N, seed=seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, moves = eng.bkl_mc(beta, iters, step)
        C1 = eng.get_config()
        E1 = eng.energy()
    assert Es.shape == (R, iters // step)
    for r in range(R):
        Es_ref, ch_ref, m_ref, _ = oracle.bkl_mc_skn(X.J, beta, iters, step, seed, C0.s[r], replica=r)
        assert (Es[r] == Es_ref).all() and (C1.s[r] == ch_ref).all() and moves[r] == m_ref
        assert E1[r] == oracle.skn_energy(X.J, ch_ref)


@pytest.mark.parametrize("N,R,beta,samples,step", [(10, 40, 2.0, 100, 100.0), (64, 33, 1.5, 60, 5.0), (200, 9, 1.0, 20, 2.0)])
def test_wtm_skn_bit_exact(pkg, oracle, N, R, beta, samples, step):
    """wtmMC on GraphSKNormal (test/runtests.jl:67 x :149-151): a heap of next-flip times, all N - 1 neighbours re-drawn per move;
    a second call continues with the next WTM call number."""
    seed = 929000 + N
    X = pkg.GraphSKNormal(N, seed=seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, moves, t = eng.wtm_mc(beta, samples, step=step)
        C1 = eng.get_config()
        E1 = eng.energy()
        Es2, moves2, t2 = eng.wtm_mc(beta, samples // 2, step=step)
    assert Es.shape == (R, samples)
    for r in range(R):
        Es_ref, ch_ref, m_ref, t_ref = oracle.wtm_mc_skn(X.J, beta, samples, step, seed, C0.s[r], replica=r)
        assert (Es[r] == Es_ref).all() and (C1.s[r] == ch_ref).all() and moves[r] == m_ref and t[r] == t_ref
        assert E1[r] == oracle.skn_energy(X.J, ch_ref)
        Es_ref2, _, m_ref2, t_ref2 = oracle.wtm_mc_skn(X.J, beta, samples // 2, step, seed, ch_ref, call=1, replica=r)
        assert (Es2[r] == Es_ref2).all() and moves2[r] == m_ref2 and t2[r] == t_ref2
    assert moves.sum() > 0
