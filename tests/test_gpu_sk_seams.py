"""Dense SK standardMC calls LONGER than one launch of the blocked kernel.

sk_block_kernel (csrc/sk_block_kernel.hpp) works in segments of kSkSegIters = 65 536 iterations: at a seam the whole per-replica state
(both field arrays, spins, move_last, tracked energy, accepted count) goes to HBM and comes back, the sample counter continues from
``it_base`` and the block tables are rebuilt.  The reference's loop (src/RRRMC.jl:100-119) has no seams, so a call that crosses one
must equal the oracle's single loop bit for bit: energies sampled on, just before and just after a seam, the configuration, the accepted
count and the live field cache — for the Gaussian (src/graphs/SK.jl:170-297) and the binary model (SK.jl:28-165), for every build of
the kernel (sk_hblock_kernel with two 4-replica workgroups per group and with one 8-replica workgroup, round 3's sk_block_kernel, the
one-attempt-at-a-time legacy kernel)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEG = 1 << 16
BUILDS = ({"RRRMC_SK_RB": "4"}, {"RRRMC_SK_RB": "8"}, {"RRRMC_SK_LEGACY": "1"}, {"RRRMC_SK_BLOCK_V1": "1"})      # sk_hblock_kernel split / whole, sk_sweep_kernel, round 3's sk_block_kernel
N_FOR_ITERS = {SEG: 1024, SEG + 1: 300, 2 * SEG + 1: 64, 200000: 256}


def _set_build(monkeypatch, env):
    for k in ("RRRMC_SK_RB", "RRRMC_SK_LEGACY", "RRRMC_SK_THREADS", "RRRMC_SK_BLOCK_V1"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)


def _graph(pkg, binary, N, seed):
    return pkg.GraphSK(N, seed=seed) if binary else pkg.GraphSKNormal(N, seed=seed)


def _oracle_run(oracle, binary):
    return oracle.standard_mc_skb if binary else oracle.standard_mc_skn


@pytest.mark.parametrize("step", [1, 777, SEG])
@pytest.mark.parametrize("iters", [SEG, SEG + 1, 2 * SEG + 1, 200000])
@pytest.mark.parametrize("binary", [False, True], ids=["gauss", "binary"])
def test_calls_across_segment_seams(pkg, oracle, monkeypatch, binary, iters, step):
    N, R, beta = N_FOR_ITERS[iters], 9, 1.0
    seed = 5000 + N + (7 if binary else 0)
    X = _graph(pkg, binary, N, seed)
    outs = []
    for env in BUILDS:
        _set_build(monkeypatch, env)
        with pkg.Engine(X, R) as eng:
            eng.seed(seed)
            eng.init_spins_random()
            C0 = eng.get_config().s.copy()
            Es, acc = eng.standard_mc(beta, iters, step)
            outs.append((Es, acc, eng.get_config().s.copy(), eng.fields(), eng.tracked_energy()))
    for o in outs[1:]:
        for u, v in zip(outs[0], o):
            assert (u == v).all()
    Es, acc, C1, lf1, Et = outs[0]
    n = iters // step
    assert Es.shape == (R, n)
    for r in (0, R - 1):
        ref = _oracle_run(oracle, binary)(X.J, beta, iters, step, seed, C0[r], replica=r)
        assert (Es[r] == ref[0][:n]).all()
        assert (C1[r] == ref[1]).all() and acc[r] == ref[2] and (lf1[r] == ref[3]).all()


@pytest.mark.parametrize("N,R", [(1500, 5), (2048, 3)])
def test_seams_beyond_the_split_build(pkg, oracle, monkeypatch, N, R):
    """1024 < N <= 2048: whole-group build only (512 threads, three or four sites per thread)."""
    _set_build(monkeypatch, {})
    seed, beta, iters, step = 77 + N, 1.0, SEG + 4097, 4096
    X = pkg.GraphSKNormal(N, seed=seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config().s.copy()
        Es, acc = eng.standard_mc(beta, iters, step)
        C1, lf1 = eng.get_config().s.copy(), eng.fields()
    for r in (0, R - 1):
        ref = oracle.standard_mc_skn(X.J, beta, iters, step, seed, C0[r], replica=r)
        assert (Es[r] == ref[0]).all() and (C1[r] == ref[1]).all() and acc[r] == ref[2] and (lf1[r] == ref[3]).all()


@pytest.mark.parametrize("binary", [False, True], ids=["gauss", "binary"])
def test_resumed_pieces_cut_inside_segments(pkg, oracle, monkeypatch, binary):
    """rrrmc_set_resume: a run cut into pieces (what a hooked standardMC does, engine.py) whose cuts fall INSIDE segments and whose pieces
    cross seams at call-relative 65 536 is the un-cut chain: move_last and lfields_last travel across cuts and seams."""
    N, R, beta, seed = 200, 6, 0.9, 4242
    X = _graph(pkg, binary, N, seed)
    pieces = [(40000, 1000), (100000, 999), (1, 1), (70001, 70001)]          # (iters, step) per call
    for env in BUILDS[:2]:
        _set_build(monkeypatch, env)
        with pkg.Engine(X, R) as eng:
            eng.seed(seed)
            eng.init_spins_random()
            C0 = eng.get_config().s.copy()
            eng.standard_mc(beta, 0, 1, want_energies=False)      # E = energy(X, C): the start of the reference call
            eng.set_resume(True)
            got = [eng.standard_mc(beta, it, st) for it, st in pieces]
            C1, lf1, Et = eng.get_config().s.copy(), eng.fields(), eng.tracked_energy()
        total = sum(p[0] for p in pieces)
        for r in (0, R - 1):
            # the same chain in ONE oracle loop; sample every iteration so that each piece's samples can be looked up
            ref = _oracle_run(oracle, binary)(X.J, beta, total, 1, seed, C0[r], replica=r)
            E_at = ref[0]                                                    # E_at[i - 1] = energy before the move of iteration i
            base, acc_sum = 0, 0
            for (it, st), (Es, acc) in zip(pieces, got):
                want = [E_at[base + k * st - 1] for k in range(1, it // st + 1)]
                assert (Es[r] == np.array(want)).all()
                base += it
                acc_sum += int(acc[r])
            assert acc_sum == ref[2] and (C1[r] == ref[1]).all() and (lf1[r] == ref[3]).all()


def test_hooked_run_with_a_step_beyond_one_segment(pkg, oracle, monkeypatch):
    _set_build(monkeypatch, {})
    N, R, beta, seed, iters, step = 96, 4, 1.1, 909, 250000, 100000
    X = pkg.GraphSKNormal(N, seed=seed)
    seen = []

    def hook(it, X_, C, accepted, E):
        seen.append((it, E.copy(), accepted.copy()))
        return True

    C0 = pkg.Config(N, R)
    C0.s[:] = oracle.init_configs(seed, 0, R, N)
    start = C0.s.copy()
    Es, C1 = pkg.standardMC(X, beta, iters, seed=seed, step=step, hook=hook, C0=C0, quiet=True)
    assert [s[0] for s in seen] == [100000, 200000]
    for r in range(R):
        ref = oracle.standard_mc_skn(X.J, beta, iters, step, seed, start[r], replica=r)
        assert (Es[r] == ref[0]).all() and (C1.s[r] == ref[1]).all()


@pytest.mark.parametrize("binary", [False, True], ids=["gauss", "binary"])
@pytest.mark.parametrize("seam", [SEG, 2 * SEG])
def test_same_site_twice_across_a_seam(pkg, oracle, monkeypatch, binary, seam):
    """update_cache!'s undo path (SK.jl:247-250: two consecutive accepted moves of one site swap lfields <-> lfields_last) with the first move
    the last one of a segment and the second the first of the next: at beta = 0 every attempt is accepted, and the seed is chosen so that
    iterations `seam` and `seam + 1` attempt the same site."""
    N, R, beta = 3, 5, 0.0
    seed = next(s for s in range(1, 500) if oracle.site_of(s, seam, N) == oracle.site_of(s, seam + 1, N)
                and oracle.site_of(s, seam - 1, N) != oracle.site_of(s, seam, N))
    X = _graph(pkg, binary, N, seed)
    iters, step = seam + 700, 1
    outs = []
    for env in BUILDS:
        _set_build(monkeypatch, env)
        with pkg.Engine(X, R) as eng:
            eng.seed(seed)
            eng.init_spins_random()
            C0 = eng.get_config().s.copy()
            Es, acc = eng.standard_mc(beta, iters, step)
            outs.append((Es, acc, eng.get_config().s.copy(), eng.fields()))
    for o in outs[1:]:
        for u, v in zip(outs[0], o):
            assert (u == v).all()
    Es, acc, C1, lf1 = outs[0]
    assert (acc == iters).all()
    for r in range(R):
        ref = _oracle_run(oracle, binary)(X.J, beta, iters, step, seed, C0[r], replica=r)
        assert (Es[r] == ref[0]).all() and (C1[r] == ref[1]).all() and (lf1[r] == ref[3]).all()
