"""GPU parity for random-site standardMC on +-J graphs that do not fit the LDS-resident kernel (plan_big_kernel, then
big_mask_kernel + big_apply_kernel with the spins of 2^lgr replicas per workgroup in LDS, or big_sweep_kernel with the spins in
HBM/L2): both bit-identical to the oracle, and to the LDS kernel where both apply."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _graph(pkg, kind, seed):
    if kind[0] == "rrg":
        return pkg.GraphRRG(kind[1], kind[2], seed=seed), "rrg"
    return pkg.GraphEA(kind[1], kind[2], seed=seed), "ea"


@pytest.mark.parametrize("kind,R,beta,iters,step", [
    (("rrg", 10, 3), 40, 2.0, 10000, 100),          # test/runtests.jl:36 through the big-N kernels
    (("ea", 2, 3), 33, 1.0, 5000, 50),              # double bonds
    (("rrg", 300, 4), 70, 1.0, 30000, 1000),        # K = 4: a class with dE = 0
    (("ea", 6, 3), 64, 1.5, 20000, 216),            # K = 6
    (("rrg", 4096, 3), 64, 1.0, 3 * 4096 + 77, 4096),   # several chunks per sample step, ragged tail
    (("rrg", 200, 5), 32, 0.7, 20000, 999),         # K = 5: three classes with dE > 0
])
def test_forced_big_path_equals_oracle_and_lds_kernel(pkg, oracle, monkeypatch, kind, R, beta, iters, step):
    seed = 600000 + kind[1] * 7 + kind[2]
    X, form = _graph(pkg, kind, seed)
    out = []
    for big, no_masks in (("1", "0"), ("0", "0"), ("1", "1")):
        monkeypatch.setenv("RRRMC_FORCE_BIG", big)
        monkeypatch.setenv("RRRMC_BIG_NO_MASKS", no_masks)
        with pkg.Engine(X, R) as eng:
            eng.seed(seed)
            eng.init_spins_random()
            C0 = eng.get_config()
            a = eng.standard_mc(beta, iters, step)
            b = eng.standard_mc(beta, iters // 3, step)           # a second call continues the streams
            out.append((a[0], a[1], b[0], b[1], eng.get_config().s, eng.energy()))
    for u, v, w in zip(*out):
        assert (u == v).all() and (u == w).all()
    ref = oracle.standard_mc_sparse_batch(X.A, X.J.astype(np.int32), beta, iters, step, seed, C0.s, form=form)
    assert (out[0][0] == ref[0]).all() and (out[0][1] == ref[2]).all()
    ref2 = oracle.standard_mc_sparse_batch(X.A, X.J.astype(np.int32), beta, iters // 3, step, seed, ref[1], it0=iters, form=form)
    assert (out[0][2] == ref2[0]).all() and (out[0][4] == ref2[1]).all() and (out[0][3] == ref2[2]).all()


@pytest.mark.parametrize("kind,R,beta,iters,step", [
    (("rrg", 20000, 3), 40, 1.0, 60000, 20000),     # beyond the LDS kernel (N > ~17 000)
    (("ea", 30, 3), 33, 1.0, 54000, 27000),         # N = 27 000, K = 6
    (("ea", 64, 3), 32, 1.0, 40000, 10000),         # BASELINE config 4's lattice (N = 262 144) under the reference's own dynamics: 4 replicas per workgroup
    (("rrg", 40000, 4), 70, 1.0, 90000, 40000),     # 16 replicas per workgroup, three groups (the last one padded)
    (("ea", 41, 3), 40, 1.5, 70000, 30000),         # N = 68 921: 8 replicas per workgroup, a ragged last LDS word
    (("rrg", 33000, 3), 33, 1.0, 2300000, 500000),  # 562 chunks: two launches of big_apply_kernel (512 chunk headers fit its LDS), samples across them
])
@pytest.mark.parametrize("no_masks", ["0", "1"])       # big_mask_kernel + big_apply_kernel (default) / big_sweep_kernel
def test_large_graphs_random_site(pkg, oracle, monkeypatch, kind, R, beta, iters, step, no_masks):
    monkeypatch.setenv("RRRMC_BIG_NO_MASKS", no_masks)
    seed = 700000 + kind[1]
    X, form = _graph(pkg, kind, seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc = eng.standard_mc(beta, iters, step)
        C1 = eng.get_config()
        E1 = eng.energy()
    J = X.J.astype(np.int32)
    for r in sorted({0, 1, 9, 17, R // 2, R - 1}):
        ref = oracle.standard_mc_sparse(X.A, J, beta, iters, step, seed, C0.s[r], replica=r, form=form)
        assert (Es[r] == ref[0]).all() and (C1.s[r] == ref[1]).all() and acc[r] == ref[2]
        assert E1[r] == oracle.sparse_energy(X.A, J, C1.s[r])
