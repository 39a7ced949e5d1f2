"""spf_team_kernel (csrc/spf_team_kernel.hpp): standardMC on the Float64 sparse models with a TEAM of wavefronts per group of 64 replicas —
attempts whose closed neighbourhoods do not meet run side by side, one wavefront retires them in the chain's order.  The reference's loop
(src/RRRMC.jl:100-119 over src/graphs/RRG.jl:576-625) knows no such thing, so everything a caller can see must be what the one-attempt-at-a-time
kernel and the oracle produce, bit for bit: samples, configurations, accepted counts, the field cache — and, through resumed calls, the undo
record and move_last (RRG.jl:583-593), which no accessor shows."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# the default picks the team width by the number of groups (few replicas: teams of 16), the others pin a build
BUILDS = ({}, {"RRRMC_SPF_TEAM_WAVES": "8"}, {"RRRMC_SPF_TEAM_WAVES": "16", "RRRMC_SPF_TEAM_WIDTH": "64"}, {"RRRMC_SPF_TEAM": "0"},
          {"RRRMC_SPF_TEAM_WIDTH": "32"})


def _set_build(monkeypatch, env):
    for k in ("RRRMC_SPF_TEAM", "RRRMC_SPF_TEAM_WAVES", "RRRMC_SPF_TEAM_WIDTH"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)


def _run(pkg, X, R, seed, beta, iters, step, replica0=0):
    with pkg.Engine(X, R, replica0=replica0) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config().s.copy()
        Es, acc = eng.standard_mc(beta, iters, step)
        return C0, Es, acc, eng.get_config().s.copy(), eng.fields(), eng.tracked_energy()


def test_builds_agree_at_full_width(pkg, oracle, monkeypatch):
    """BASELINE's config-2 geometry with Gaussian couplings: 8192 replicas = 128 groups, every group's team on its own compute unit."""
    seed, R, beta, iters, step = 99, 8192, 1.0, 40000, 4096
    X = pkg.GraphRRGNormal(4096, 3, seed=seed)
    outs = []
    for env in (BUILDS[0], BUILDS[3]):
        _set_build(monkeypatch, env)
        outs.append(_run(pkg, X, R, seed, beta, iters, step))
    for u, v in zip(outs[0], outs[1]):
        assert (u == v).all()
    C0, Es, acc, C1, lf1, Et = outs[0]
    for r in (0, 4097, R - 1):
        ref = oracle.standard_mc_spf(X.A, X.J, beta, iters, step, seed, C0[r], replica=r)
        assert (Es[r] == ref[0]).all() and (C1[r] == ref[1]).all() and acc[r] == ref[2] and (lf1[r] == ref[3]).all()


@pytest.mark.parametrize("kind,a,b,beta", [("rrg", 8, 3, 0.0), ("rrg", 10, 3, 0.4), ("rrg", 12, 4, 1.0), ("rrg", 30, 5, 0.2), ("ea", 3, 3, 0.6),
                                           ("rrg", 200, 7, 0.5), ("ea", 3, 4, 1.0)])
def test_undo_path_and_dense_conflicts(pkg, oracle, monkeypatch, kind, a, b, beta):
    """Small graphs (random regular with N = a, K = b; periodic lattices with L = a, D = b, i.e. K = 2 D up to 8): nearly every attempt depends
    on the one before, the same spin is attempted again within a few iterations (at beta = 0 every move is accepted, so every repeat is the
    array-swap undo of RRG.jl:583-593), and the records of a slot are evacuated while other wavefronts read them.  Every replica against the
    oracle, all builds against each other."""
    seed, R, iters, step = 1000 + 10 * a + b, 130, 30000, 777
    X = pkg.GraphRRGNormal(a, b, seed=seed) if kind == "rrg" else pkg.GraphEANormal(a, b, seed=seed)
    outs = []
    for env in BUILDS:
        _set_build(monkeypatch, env)
        outs.append(_run(pkg, X, R, seed, beta, iters, step))
    for o in outs[1:]:
        for u, v in zip(outs[0], o):
            assert (u == v).all()
    C0, Es, acc, C1, lf1, Et = outs[0]
    for r in range(R):
        ref = oracle.standard_mc_spf(X.A, X.J, beta, iters, step, seed, C0[r], replica=r, form=kind)
        assert (Es[r] == ref[0]).all() and (C1[r] == ref[1]).all() and acc[r] == ref[2] and (lf1[r] == ref[3]).all()


def test_launch_boundaries_and_odd_stream_offsets(pkg, oracle, monkeypatch):
    """The team kernel takes the attempts in pairs that share one Philox block (stream indices 2h, 2h + 1) and works in launches of 2^18
    iterations: calls of odd length make the following launches start on an odd index, a call of 2^18 + 3 iterations crosses a launch."""
    seed, R = 31337, 9
    X = pkg.GraphRRGNormal(20, 3, seed=seed)
    calls = [(1, 1), (2, 1), (7, 3), ((1 << 18) + 3, 50001), (1001, 10)]
    for env in BUILDS[:2]:
        _set_build(monkeypatch, env)
        with pkg.Engine(X, R) as eng:
            eng.seed(seed)
            eng.init_spins_random()
            C = eng.get_config().s.copy()
            it0 = 0
            for iters, step in calls:
                Es, acc = eng.standard_mc(0.7, iters, step)
                C1 = eng.get_config().s.copy()
                for r in (0, R - 1):
                    ref = oracle.standard_mc_spf(X.A, X.J, 0.7, iters, step, seed, C[r], it0=it0, replica=r)
                    assert (Es[r] == ref[0]).all() and (C1[r] == ref[1]).all() and acc[r] == ref[2]
                C, it0 = C1, it0 + iters


def test_resumed_pieces_alternate_between_the_builds(pkg, oracle, monkeypatch):
    """rrrmc_set_resume: a run cut into pieces is the un-cut chain only if move_last and the undo record travel across the cuts — here
    every piece runs through a different build, so the records the team kernel leaves behind are read by the single-wavefront kernel and back."""
    N, R, beta, seed = 10, 70, 0.3, 777
    X = pkg.GraphRRGNormal(N, 3, seed=seed + 1)
    pieces = [(4000, 100), (1, 1), (2999, 7), (5000, 5000), (3, 1), (6000, 13)]
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config().s.copy()
        _set_build(monkeypatch, BUILDS[3])
        eng.standard_mc(beta, 0, 1, want_energies=False)          # E = energy(X, C): the start of the reference call
        eng.set_resume(True)
        got = []
        for q, (it, st) in enumerate(pieces):
            _set_build(monkeypatch, BUILDS[q % len(BUILDS)])
            got.append(eng.standard_mc(beta, it, st))
        C1, lf1 = eng.get_config().s.copy(), eng.fields()
    total = sum(p[0] for p in pieces)
    for r in range(R):
        ref = oracle.standard_mc_spf(X.A, X.J, beta, total, 1, seed, C0[r], replica=r)       # sampled every iteration
        base, acc_sum = 0, 0
        for (it, st), (Es, acc) in zip(pieces, got):
            want = [ref[0][base + k * st - 1] for k in range(1, it // st + 1)]
            assert (Es[r] == np.array(want)).all()
            base += it
            acc_sum += int(acc[r])
        assert acc_sum == ref[2] and (C1[r] == ref[1]).all() and (lf1[r] == ref[3]).all()


def test_lattice_with_six_neighbours_and_many_groups(pkg, oracle, monkeypatch):
    """GraphEANormal(6, 3) (K = 6: the records of a slot no longer fit sixteen wavefronts' worth of LDS, the eight-wavefront build runs) and
    more groups than compute units (320 groups of 64 replicas)."""
    seed, R, beta, iters, step = 5, 320 * 64, 1.2, 6000, 500
    X = pkg.GraphEANormal(6, 3, seed=seed)
    outs = []
    for env in (BUILDS[0], BUILDS[3]):
        _set_build(monkeypatch, env)
        outs.append(_run(pkg, X, R, seed, beta, iters, step))
    for u, v in zip(outs[0], outs[1]):
        assert (u == v).all()
    C0, Es, acc, C1, lf1, Et = outs[0]
    for r in (0, 12345, R - 1):
        ref = oracle.standard_mc_spf(X.A, X.J, beta, iters, step, seed, C0[r], replica=r, form="ea")
        assert (Es[r] == ref[0]).all() and (C1[r] == ref[1]).all() and acc[r] == ref[2] and (lf1[r] == ref[3]).all()


@pytest.mark.parametrize("kind,a,b", [("rrg", 4096, 3), ("rrg", 30, 5), ("ea", 2, 3), ("ea", 6, 3), ("rrg", 7, 2)])
def test_two_step_energy_equals_the_walk(pkg, oracle, monkeypatch, kind, a, b):
    """energy (RRG.jl:546-574 / EA.jl:584-611) at the start of every call: the fields of all sites side by side and the reference's
    sequential sum behind them (spf_fields_kernel + spf_energy_sum_kernel) against one wavefront per group walking the sites
    (spf_energy_kernel, RRRMC_SPF_ENERGY_V1=1) and against the oracle — energies and the field cache bit for bit, doubled bonds included."""
    seed, R = 4000 + a + b, 200
    X = pkg.GraphRRGNormal(a, b, seed=seed) if kind == "rrg" else pkg.GraphEANormal(a, b, seed=seed)
    outs = []
    for v1 in ("0", "1"):
        monkeypatch.setenv("RRRMC_SPF_ENERGY_V1", v1)
        with pkg.Engine(X, R) as eng:
            eng.seed(seed)
            eng.init_spins_random()
            C0 = eng.get_config().s.copy()
            outs.append((eng.energy(), eng.fields()))
    assert (outs[0][0] == outs[1][0]).all() and (outs[0][1] == outs[1][1]).all()
    for r in (0, 63, 64, R - 1):
        e0, f0 = oracle.spf_energy(X.A, X.J, C0[r], want_fields=True, form=kind)
        assert outs[0][0][r] == e0 and (outs[0][1][r] == f0).all()


@pytest.mark.parametrize("D,beta", [(2, 0.3), (3, 1.0), (4, 0.6)])
def test_doubled_bonds_through_the_team_kernel(pkg, oracle, monkeypatch, D, beta):
    """GraphEANormal(2, D): every neighbour is listed twice (two bonds to the same site, EA.jl:158).  update_cache! walks all 2 D entries
    (EA.jl:626-640), so the second bond's update continues from the first one's result; with 2^D sites nearly every accepted move is
    followed by an attempt in its neighbourhood, and the undo branch swaps over the UNIQUE neighbours.  All builds, every replica."""
    seed, R, iters, step = 600 + D, 130, 20000, 333
    X = pkg.GraphEANormal(2, D, seed=seed)
    outs = []
    for env in BUILDS:
        _set_build(monkeypatch, env)
        outs.append(_run(pkg, X, R, seed, beta, iters, step))
    for o in outs[1:]:
        for u, v in zip(outs[0], o):
            assert (u == v).all()
    C0, Es, acc, C1, lf1, Et = outs[0]
    for r in range(R):
        ref = oracle.standard_mc_spf(X.A, X.J, beta, iters, step, seed, C0[r], replica=r, form="ea")
        assert (Es[r] == ref[0]).all() and (C1[r] == ref[1]).all() and acc[r] == ref[2] and (lf1[r] == ref[3]).all()


@pytest.mark.parametrize("case", range(12))
def test_random_shapes_team_builds_against_the_single_wavefront_kernel(pkg, monkeypatch, case):
    """Randomised soak: graph size and degree, replica count (so: team width and number of teams), beta, sampling step and the
    lengths of three consecutive resumed calls (odd stream offsets, a launch boundary in some) drawn per case; every team build must leave
    behind exactly what the one-attempt-at-a-time kernel leaves: samples, accepted counts, configurations, field cache, tracked energy."""
    rng = np.random.default_rng(20251004 + case)
    K = int(rng.integers(2, 7))
    N = int(rng.integers(K + 2, 300))
    N += (N * K) % 2
    R = int(rng.integers(65, 700))
    beta = float(rng.choice([0.0, 0.3, 1.0, 2.5]))
    calls = [(int(rng.integers(1, 9000)), int(rng.integers(1, 500))) for _ in range(2)] + [((1 << 18) + int(rng.integers(0, 50)) if case % 4 == 0 else int(rng.integers(1, 3000)), 977)]
    seed = 4242 + case
    X = pkg.GraphRRGNormal(N, K, seed=seed)
    outs = []
    for env in (BUILDS[3], BUILDS[0], BUILDS[1], BUILDS[4]) if case % 2 else (BUILDS[3], BUILDS[0], BUILDS[2]):
        _set_build(monkeypatch, env)
        with pkg.Engine(X, R) as eng:
            eng.seed(seed)
            eng.init_spins_random()
            got = []
            eng.standard_mc(beta, 0, 1, want_energies=False)
            eng.set_resume(True)
            for iters, step in calls:
                Es, acc = eng.standard_mc(beta, iters, step)
                got += [Es, acc]
            got += [eng.get_config().s.copy(), eng.fields(), eng.tracked_energy()]
        outs.append(got)
    for o in outs[1:]:
        for u, v in zip(outs[0], o):
            assert (np.asarray(u) == np.asarray(v)).all()


def _circulant7(pkg, N, seed):
    """a 7-regular graph by construction (the pairing model of gen_RRG rarely finds a simple one at this degree): i +- 1, 2, 3 and the antipode"""
    g = np.random.default_rng(seed)
    A = np.array([sorted({(i + o) % N for o in (1, 2, 3)} | {(i - o) % N for o in (1, 2, 3)} | {(i + N // 2) % N}) for i in range(N)], np.int32)
    bond, J = {}, np.zeros((N, 7))
    for i in range(N):
        for k, j in enumerate(A[i]):
            J[i, k] = bond.setdefault((min(i, int(j)), max(i, int(j))), g.standard_normal())
    return pkg.GraphRRGNormal.from_AJ(A, J)


@pytest.mark.parametrize("kind,a,b,beta", [("rrg", 512, 3, 0.1), ("rrg", 512, 6, 0.5), ("ea", 8, 2, 0.3), ("ea", 5, 3, 1.0), ("rrg", 64, 5, 0.0),
                                           ("circ7", 512, 7, 0.4), ("ea", 4, 4, 0.5), ("ea", 2, 4, 1.0), ("circ7", 66, 7, 0.0)])      # K = 7, 8: fused in eight-wavefront teams
def test_fused_pairs_at_every_width(pkg, oracle, monkeypatch, kind, a, b, beta):
    """Teams of 32 and 16 replicas run the two commuting attempts of a pair in the two halves of a wavefront (K <= 6).  Graphs of a few hundred
    sites at high acceptance: most pairs are fused, a good share of them restarts on the undo test (an accepting replica whose last accepted
    move is at the same site), some have a dependent second attempt; the accepted moves of a replica are counted in two lanes.  Widths 16, 32
    and 64 and the single-wavefront kernel against each other; replicas from both ends of a team against the oracle."""
    seed, R, iters, step = 4000 + 7 * a + b, 1056, 20001, 501
    if kind == "circ7":
        X, kind = _circulant7(pkg, a, seed), "rrg"
    else:
        X = pkg.GraphRRGNormal(a, b, seed=seed) if kind == "rrg" else pkg.GraphEANormal(a, b, seed=seed)
    outs = []
    for env in ({"RRRMC_SPF_TEAM_WIDTH": "16"}, {"RRRMC_SPF_TEAM_WIDTH": "32"}, {"RRRMC_SPF_TEAM_WAVES": "16", "RRRMC_SPF_TEAM_WIDTH": "64"}, {"RRRMC_SPF_TEAM": "0"},
                {"RRRMC_SPF_TEAM_WAVES": "16", "RRRMC_SPF_TEAM_WIDTH": "32"}):          # (the last: sixteen wavefronts — for K = 7, 8 the build WITHOUT fused pairs)
        _set_build(monkeypatch, env)
        outs.append(_run(pkg, X, R, seed, beta, iters, step))
    for o in outs[1:]:
        for u, v in zip(outs[0], o):
            assert (u == v).all()
    C0, Es, acc, C1, lf1, Et = outs[0]
    for r in (0, 15, 16, 31, 32, 63, 64, 1024, R - 1):
        ref = oracle.standard_mc_spf(X.A, X.J, beta, iters, step, seed, C0[r], replica=r, form=kind)
        assert (Es[r] == ref[0]).all() and (C1[r] == ref[1]).all() and acc[r] == ref[2] and (lf1[r] == ref[3]).all()
