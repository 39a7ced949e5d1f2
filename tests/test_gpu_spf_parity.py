"""GPU parity for the Float64-coupling sparse models GraphRRGNormal / GraphEANormal under standardMC
(src/graphs/RRG.jl:503-627, src/graphs/EA.jl:534-680).  north_star tolerance for Float64 models is
1e-6 relative; the kernel keeps the reference's operation order (including the lfields_last undo path) and the
deterministic exp, so we additionally require bit equality with the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# every test of this module runs through the three builds of the sweep: spf_team_kernel with sixteen and with eight wavefronts per group of
# 64 replicas (csrc/spf_team_kernel.hpp; the default picks between them by K and the number of groups), and spf_sweep_kernel, one
# wavefront per group (csrc/spf_kernels.hpp).  GraphEANormal with L = 2 (two bonds to the same neighbour) runs through all of them
SPF_BUILDS = {"team": {}, "team8": {"RRRMC_SPF_TEAM_WAVES": "8"}, "team16": {"RRRMC_SPF_TEAM_WAVES": "16", "RRRMC_SPF_TEAM_WIDTH": "64"},
              "team32w": {"RRRMC_SPF_TEAM_WIDTH": "32"}, "single": {"RRRMC_SPF_TEAM": "0"}}


@pytest.fixture(autouse=True, params=list(SPF_BUILDS))
def spf_build(request, monkeypatch):
    for k in ("RRRMC_SPF_TEAM", "RRRMC_SPF_TEAM_WAVES", "RRRMC_SPF_TEAM_WIDTH"):
        monkeypatch.delenv(k, raising=False)
    for k, v in SPF_BUILDS[request.param].items():
        monkeypatch.setenv(k, v)
    return request.param


def _graph(pkg, kind, seed):
    if kind == "rrg10":
        return pkg.GraphRRGNormal(10, 3, seed=seed), "rrg"          # test/runtests.jl:40
    if kind == "rrg200k4":
        return pkg.GraphRRGNormal(200, 4, seed=seed), "rrg"
    if kind == "rrg4096":
        return pkg.GraphRRGNormal(4096, 3, seed=seed), "rrg"        # config-2 geometry with Gaussian couplings
    if kind == "ea3x2":
        return pkg.GraphEANormal(3, 2, seed=seed), "ea"             # test/runtests.jl:60
    if kind == "ea2x3":
        return pkg.GraphEANormal(2, 3, seed=seed), "ea"             # L = 2: every neighbour appears twice (EA.jl:158)
    if kind == "ea6x3":
        return pkg.GraphEANormal(6, 3, seed=seed), "ea"
    raise KeyError(kind)


@pytest.mark.parametrize("kind,R,beta,iters,step", [
    ("rrg10", 70, 2.0, 5000, 50),        # tiny N: the same site repeats often -> undo path, stale prefetches
    ("rrg200k4", 64, 1.0, 20000, 500),
    ("rrg4096", 130, 1.0, 30000, 1024),
    ("ea3x2", 33, 1.5, 5000, 100),
    ("ea2x3", 16, 0.8, 4000, 64),
    ("ea6x3", 64, 1.0, 20000, 1000),
    ("rrg10", 5, 0.0, 300, 7),           # beta = 0: every move accepted
    ("rrg200k4", 3, 50.0, 3000, 1000),   # frozen
])
def test_spf_standard_mc_bit_exact(pkg, oracle, kind, R, beta, iters, step):
    seed = 20260000 + len(kind) + R
    X, form = _graph(pkg, kind, seed)
    assert (X.J == oracle.gen_couplings_gauss(X.A, seed)).all()
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        E0 = eng.energy()
        lf0 = eng.fields()
        Es, acc = eng.standard_mc(beta, iters, step)
        C1 = eng.get_config()
        lf1 = eng.fields()
        # a second call continues the streams (seed <= 0 semantics) and restarts from energy(X, C) like the reference
        Es2, acc2 = eng.standard_mc(beta, 1000, 100)
        C2 = eng.get_config()
    assert (C0.s == oracle.init_configs(seed, 0, R, X.N)).all()
    for r in range(R):
        e0, f0 = oracle.spf_energy(X.A, X.J, C0.s[r], want_fields=True, form=form)
        assert E0[r] == e0 and (lf0[r] == f0).all()
        Es_ref, ch_ref, acc_ref, lf_ref = oracle.standard_mc_spf(X.A, X.J, beta, iters, step, seed, C0.s[r], replica=r, form=form)
        assert np.allclose(Es[r], Es_ref, rtol=1e-6, atol=1e-9)          # north-star tolerance
        assert (Es[r] == Es_ref).all()                                    # ... and bit for bit
        assert (C1.s[r] == ch_ref).all() and acc[r] == acc_ref
        assert (lf1[r] == lf_ref).all()
        Es2_ref, ch2_ref, acc2_ref, _ = oracle.standard_mc_spf(X.A, X.J, beta, 1000, 100, seed, ch_ref, it0=iters, replica=r, form=form)
        assert (Es2[r] == Es2_ref).all() and (C2.s[r] == ch2_ref).all() and acc2[r] == acc2_ref


def test_spf_multi_launch_and_replica_offset(pkg, oracle, spf_build):
    """More iterations than one launch covers (2^18 for the team kernel, 2^20 for the single-wavefront one) with a step that does not divide the
    launch length, and replica0 != 0."""
    seed, R, iters, step = 424242, 8, (1 << 20) + 5000, 70001
    X = pkg.GraphRRGNormal(64, 3, seed=seed)
    with pkg.Engine(X, R, replica0=96) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc = eng.standard_mc(0.9, iters, step)
        C1 = eng.get_config()
        assert eng.last_timing()[2] == (2 if spf_build == "single" else 5)
    assert Es.shape == (R, iters // step)
    for r in (0, 7):
        assert (C0.s[r] == oracle.init_config(seed, 96 + r, X.N)).all()
        Es_ref, ch_ref, acc_ref, _ = oracle.standard_mc_spf(X.A, X.J, 0.9, iters, step, seed, C0.s[r], replica=96 + r)
        assert (Es[r] == Es_ref).all() and (C1.s[r] == ch_ref).all() and acc[r] == acc_ref


def test_spf_front_end_file_loader_and_errors(pkg, oracle, tmp_path):
    seed = 11
    # the text format of gen_AJ (EA.jl:73-118): D = 2
    L = 4
    Xg = pkg.GraphEANormal(L, 2, seed=seed)
    lines = ["type: EA2D", "size: %d" % L, "name: test"]
    for x in range(Xg.N):
        for k in range(4):
            y = Xg.A[x, k]
            if x < y:
                lines.append("%d %d %r" % (x + 1, y + 1, float(Xg.J[x, k])))
    f = tmp_path / "ea.txt"
    f.write_text("\n".join(lines) + "\n")
    Xf = pkg.GraphEANormal(str(f))
    assert (Xf.A == Xg.A).all() and (Xf.J == Xg.J).all() and Xf.L == L and Xf.D == 2
    Es, C = pkg.standardMC(Xf, 1.0, 3000, step=100, seed=seed, quiet=True, replicas=4)
    C0 = oracle.init_configs(seed, 0, 4, Xf.N)
    for r in range(4):
        Es_ref, ch_ref, _, _ = oracle.standard_mc_spf(Xf.A, Xf.J, 1.0, 3000, 100, seed, C0[r], replica=r, form="ea")
        assert (Es[r] == Es_ref).all() and (C.s[r] == ch_ref).all()
    # hook path == reference's checkenergy invariant (runtests.jl:12-20)
    def hook(it, X, Cfg, acc, E):
        for r in range(Cfg.R):
            assert abs(E[r] - oracle.spf_energy(X.A, X.J, Cfg.s[r], form="ea")) < 1e-11
        return True
    pkg.standardMC(Xf, 1.0, 500, step=100, seed=seed, quiet=True, replicas=3, hook=hook)
    # errors
    J_bad = Xg.J.copy()
    J_bad[0, 0] += 1.0
    with pytest.raises(pkg.RRRMCError):
        pkg.Engine(pkg.GraphEANormal.from_AJ(Xg.A, J_bad), 2)                # asymmetric couplings
    with pytest.raises(pkg.RRRMCError):
        pkg.Engine(pkg.GraphRRGNormal.from_AJ(np.zeros((4, 9), np.int32), np.zeros((4, 9))), 2)   # K > 8


def test_spf_overlaps_and_snapshots(pkg, oracle):
    seed, R, T = 5, 67, 3
    X = pkg.GraphRRGNormal(100, 3, seed=seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        eng.snapshot_reserve(T)
        cfgs = []
        for k in range(T):
            eng.standard_mc(0.5, 300, 300)
            eng.snapshot_store(k)
            cfgs.append(eng.get_config().s.copy())
        assert (eng.snapshot_get(1).s == cfgs[1]).all()
        q = eng.overlaps([0, 0, 1], [1, 2, 2])
    for idx, (a, b) in enumerate([(0, 1), (0, 2), (1, 2)]):
        for r in range(R):
            assert q[idx, r] == oracle.pm1dot(cfgs[a][r], cfgs[b][r], X.N)


def _random_spf_cases(n, seed):
    rng = np.random.default_rng(seed)
    cases = []
    while len(cases) < n:
        K = int(rng.integers(1, 6))
        N = int(rng.integers(max(6, 2 * K + 2), 700))
        if (N * K) % 2:
            N += 1
        cases.append((N, K, int(rng.integers(1, 200)), float(rng.choice([0.0, 0.5, 1.0, 3.0])), int(rng.integers(1, 12000)),
                      int(rng.integers(1, 3000))))
    return cases


@pytest.mark.parametrize("N,K,R,beta,iters,step", _random_spf_cases(16, 4242))
def test_spf_randomized_shapes(pkg, oracle, N, K, R, beta, iters, step):
    """The Float64 kernel speculates (fields requested iterations ahead, re-read when an accepted move lands in their neighbourhood):
    seeded random shapes, small N included (hits in nearly every iteration), against the oracle; a second call checks the resume."""
    seed = 77 * N + K
    X = pkg.GraphRRGNormal(N, K, seed=seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc = eng.standard_mc(beta, iters, step)
        C1 = eng.get_config()
        Es2, acc2 = eng.standard_mc(beta, iters // 2 + 1, step)
        C2 = eng.get_config()
    for r in sorted({0, R // 2, R - 1}):
        ref = oracle.standard_mc_spf(X.A, X.J, beta, iters, step, seed, C0.s[r], replica=r)
        assert (Es[r] == ref[0]).all() and (C1.s[r] == ref[1]).all() and acc[r] == ref[2]
        ref2 = oracle.standard_mc_spf(X.A, X.J, beta, iters // 2 + 1, step, seed, ref[1], it0=iters, replica=r)
        assert (Es2[r] == ref2[0]).all() and (C2.s[r] == ref2[1]).all() and acc2[r] == ref2[2]
