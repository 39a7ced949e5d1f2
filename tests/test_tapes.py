"""RNG-free tapes (tests/golden/tape_*.txt): the oracle pinned from outside.

  * CPU: the plain-Python replay (tests/tape_replay.py — written after the Julia sources, libm exp, no code shared with the C
    oracle) reproduces what the tape holds, i.e. two independent restatements of the reference agree draw for draw; the C oracle run
    again on the tape's inputs reproduces it too (drift check); the Ising1D closed forms (src/graphs/Ising1D.jl) are a known-answer
    test of the tape format's bit layout and of delta_energy bookkeeping.
  * GPU: the HIP library, seeded like the tape, produces the tape's expected results.
  * Anywhere Julia + RRRMC.jl exist: `julia tests/replay_tape.jl` replays the same files through the reference itself."""
import os

import numpy as np
import pytest

import tape_replay as TR

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STD, QNT, QDIR = (os.path.join(GOLD, f) for f in ("tape_rrg_n128.txt", "tape_quant_nk16_m4.txt", "tape_quant_direct.txt"))
# round 3: the Float64 / doubled-bond / undo-path models (where an order-of-operations slip would hide)
EA2, SKN, SKB, RSKN = (os.path.join(GOLD, f) for f in ("tape_ea_l2_d3.txt", "tape_skn_n24.txt", "tape_sk_n10.txt", "tape_rrr_skn_n10.txt"))


def _graph(t, n, K):
    A = np.array([int(v) - 1 for v in t["A"]], np.int32).reshape(n, K)
    J = np.array([int(v) for v in t["J"]], np.int32).reshape(n, K)
    return A, J


def test_python_replay_reproduces_the_standard_mc_tape():
    t = TR.read_tape(STD)
    got = TR.replay_standard_mc(t)
    assert got["Es"] == [int(v) for v in t["expected_Es"]]
    assert got["chunks"] == [int(c, 16) for c in t["expected_chunks"]]
    assert got["accepted"] == int(t["expected_accepted"]) and got["flips"] == [int(v) for v in t["expected_flips"]]
    assert got["min_margin"] > 1e-9          # no decision of the tape hinges on the last bits of exp


@pytest.mark.parametrize("path", [QNT, QDIR])
def test_python_replay_reproduces_the_rrr_quant_tapes(path):
    t = TR.read_tape(path)
    got = TR.replay_rrr_quant(t)
    assert got["chunks"] == [int(c, 16) for c in t["expected_chunks"]]
    assert got["accepted"] == int(t["expected_accepted"]) and got["staged_its"] == int(t["expected_staged_its"])
    assert got["sizes"] == [int(v) for v in t["expected_sizes"]] and got["pos"] == [int(v) for v in t["expected_pos"]]
    np.testing.assert_allclose(got["Es"], [float(v) for v in t["expected_Es"]], rtol=1e-12, atol=1e-12)
    assert got["min_margin"] > 1e-9
    if path == QDIR:
        assert got["staged_its"] == 0 and got["accepted"] < int(t["iters"])      # apply_move! and its undo both happen
    else:
        assert got["staged_its"] > 0


def test_oracle_reproduces_the_tapes(oracle):
    t = TR.read_tape(STD)
    N, K, seed = int(t["N"]), int(t["K"]), int(t["seed"])
    A, J = _graph(t, N, K)
    assert (A == oracle.gen_rrg(N, K, seed)).all() and (J == oracle.gen_couplings(A, seed)).all()
    C0 = np.array([int(c, 16) for c in t["C0"]], np.uint64)
    Es, ch, acc, _lf, sites, flips = oracle.standard_mc_sparse(A, J, float(t["beta"]), int(t["iters"]), int(t["step"]), seed, C0, trace=True)
    assert [int(e) for e in Es] == [int(v) for v in t["expected_Es"]] and acc == int(t["expected_accepted"])
    assert [int(c) for c in ch] == [int(c, 16) for c in t["expected_chunks"]]
    assert [int(s) + 1 for s in sites] == [int(v) for v in t["sites"]] and [int(f) for f in flips] == [int(v) for v in t["expected_flips"]]
    for path in (QNT, QDIR):
        t = TR.read_tape(path)
        Nk, K, M, seed = int(t["Nk"]), int(t["K"]), int(t["M"]), int(t["seed"])
        A, J = _graph(t, Nk, K)
        C0 = np.array([int(c, 16) for c in t["C0"]], np.uint64)
        ref = oracle.rrr_mc_quant(A, J, M, float(t["fourK"]), float(t["beta"]), int(t["iters"]), int(t["step"]), seed, C0,
                                  staged_thr=float(t["staged_thr"]), staged_thr_fact=float(t["staged_thr_fact"]), want_cache=True)
        assert [float(e) for e in ref[0]] == [float(v) for v in t["expected_Es"]]
        assert [int(c) for c in ref[1]] == [int(c, 16) for c in t["expected_chunks"]]
        assert ref[2] == int(t["expected_accepted"]) and ref[3] == int(t["expected_staged_its"])


def test_ising1d_known_answers():
    """GraphIsing1D (src/graphs/Ising1D.jl:35-93): antiferromagnetic ring (J = -1 in the +-1 convention: the reference stores J = trues
    and an energy of +sum sigma sigma') in a unit field, E = sum_i sigma_i sigma_{i+1} - sum_i sigma_i, delta_energy as written there,
    allΔE = (2, 6).  Known answers: the closed-form energies of the uniform and the alternating configurations, delta_energy ==
    energy difference for every flip, |delta_energy| in allΔE + {... field-shifted values} exactly as the reference's formula gives."""
    def energy(s):                                        # Ising1D.jl:35-58 in its commented-out direct form (n0) plus the "!!!" field term
        N = len(s)
        n = sum((2 * 1 - 1) * (2 * s[i] - 1) * (2 * s[(i + 1) % N] - 1) for i in range(N))
        return n + N - 2 * sum(s)

    def energy_popcount(s):                               # the BitVector formula the reference actually evaluates (:46-55)
        N = len(s)
        J = [1] * N
        s1 = s[1:] + s[:1]                                # circshift(s, -1)
        Js = [a & b for a, b in zip(J, s)]
        Js1 = [a & b for a, b in zip(J, s1)]
        ss1 = [a & b for a, b in zip(s, s1)]
        n1 = 8 * sum(a & b for a, b in zip(Js, s1)) - 4 * sum(Js) - 4 * sum(Js1) - 4 * sum(ss1) + 2 * sum(J) + 4 * sum(s) - N
        return n1 + N - 2 * sum(s)

    def delta_energy(s, move):                            # Ising1D.jl:60-88, move 1-based
        N = len(s)
        sg = lambda i: 2 * s[i - 1] - 1
        d = 0
        d -= sg(N) * sg(1) if move == 1 else sg(move - 1) * sg(move)
        d -= sg(N) * sg(1) if move == N else sg(move) * sg(move + 1)
        return 2 * d + 2 * (2 * s[move - 1] - 1)

    N = 12
    assert energy([1] * N) == N - N and energy([0] * N) == N + N            # all up: N bonds - N field; all down: N + N
    assert energy([i & 1 for i in range(N)]) == -N                          # alternating: every bond satisfied, zero magnetisation
    rng = np.random.default_rng(7)
    seen = set()
    for _ in range(50):
        s = [int(b) for b in rng.integers(0, 2, N)]
        assert energy(s) == energy_popcount(s)
        # the BitVector chunk layout of the tapes: site i (1-based) = bit (i-1)&63 of chunk (i-1)>>6
        assert TR.bits_of_chunks(TR.chunks_of_bits(s), N) == s
        for move in range(1, N + 1):
            t = list(s)
            t[move - 1] ^= 1
            d = delta_energy(s, move)
            assert d == energy(t) - energy(s)
            seen.add(abs(d))
    assert seen <= {2, 6} and seen == {2, 6}                                # allΔE(GraphIsing1D) = (2, 6), Ising1D.jl:91


@pytest.mark.gpu
def test_hip_library_reproduces_the_standard_mc_tape(pkg):
    t = TR.read_tape(STD)
    N, K, seed = int(t["N"]), int(t["K"]), int(t["seed"])
    X = pkg.GraphRRG(N, K, seed=seed)
    A, J = _graph(t, N, K)
    assert (X.A == A).all() and (X.J == J).all()
    with pkg.Engine(X, 32) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        assert [int(c) for c in eng.get_config().s[0]] == [int(c, 16) for c in t["C0"]]
        Es, acc = eng.standard_mc(float(t["beta"]), int(t["iters"]), int(t["step"]))
        C1 = eng.get_config()
    assert [int(e) for e in Es[0]] == [int(v) for v in t["expected_Es"]] and int(acc[0]) == int(t["expected_accepted"])
    assert [int(c) for c in C1.s[0]] == [int(c, 16) for c in t["expected_chunks"]]


@pytest.mark.gpu
@pytest.mark.parametrize("path", [QNT, QDIR])
def test_hip_library_reproduces_the_rrr_quant_tapes(pkg, path):
    t = TR.read_tape(path)
    Nk, K, M, seed = int(t["Nk"]), int(t["K"]), int(t["M"]), int(t["seed"])
    X = pkg.GraphQuant(pkg.GraphRRG(Nk, K, seed=seed), M, float(t["Gamma"]), float(t["beta"]))
    assert X.fourK == float(t["fourK"])
    with pkg.Engine(X, 4) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        assert [int(c) for c in eng.get_config().s[0]] == [int(c, 16) for c in t["C0"]]
        Es, acc, staged = eng.rrr_mc(float(t["beta"]), int(t["iters"]), int(t["step"]), staged_thr=float(t["staged_thr"]),
                                     staged_thr_fact=float(t["staged_thr_fact"]))
        C1 = eng.get_config()
        pos, sizes = eng.rrr_cache()
    assert [float(e) for e in Es[0]] == [float(v) for v in t["expected_Es"]]
    assert int(acc[0]) == int(t["expected_accepted"]) and int(staged[0]) == int(t["expected_staged_its"])
    assert [int(c) for c in C1.s[0]] == [int(c, 16) for c in t["expected_chunks"]]
    assert [int(v) + 1 for v in pos[0]] == [int(v) for v in t["expected_pos"]] and [int(v) for v in sizes[0]] == [int(v) for v in t["expected_sizes"]]


# ---- round 3 tapes: GraphEA(2, 3) (doubled bonds + undo fast path), GraphSKNormal(24) and binary GraphSK(10) (whole-array swap), rrrMC on
#      GraphSKNormal(10) through DeltaECacheCont / DynamicSampler (refresh! included) ---------------------------------------------------------
def _floats(v):
    return [float(x) for x in v]


def test_python_replay_reproduces_the_ea_l2_tape():
    t = TR.read_tape(EA2)
    got = TR.replay_standard_mc_ea(t)
    assert got["Es"] == [int(v) for v in t["expected_Es"]] and got["chunks"] == [int(c, 16) for c in t["expected_chunks"]]
    assert got["accepted"] == int(t["expected_accepted"]) and got["flips"] == [int(v) for v in t["expected_flips"]]
    assert got["min_margin"] > 1e-9
    assert got["undos"] >= 5                                   # update_cache!'s undo branch (EA.jl:231-240) is part of the tape
    A = [[int(v) for v in t["A"][x * 6:(x + 1) * 6]] for x in range(8)]
    assert all(a[0] == a[1] and a[2] == a[3] and a[4] == a[5] for a in A)          # every bond doubled (L = 2)


@pytest.mark.parametrize("path", [SKN, SKB])
def test_python_replay_reproduces_the_sk_tapes(path):
    t = TR.read_tape(path)
    got = TR.replay_standard_mc_sk(t)
    assert got["Es"] == _floats(t["expected_Es"])              # the same sequence of IEEE operations: equal bit for bit
    assert got["chunks"] == [int(c, 16) for c in t["expected_chunks"]] and got["accepted"] == int(t["expected_accepted"])
    assert got["lfields"] == _floats(t["expected_lfields"])
    assert got["swaps"] == int(t["expected_swaps"]) >= 1       # a consecutive accepted pair at one site: the array swap of SK.jl:247-250 / :106-109
    assert got["min_margin"] > 1e-9


def test_python_replay_reproduces_the_rrr_sknormal_tape():
    t = TR.read_tape(RSKN)
    got = TR.replay_rrr_single_sk(t)
    assert got["chunks"] == [int(c, 16) for c in t["expected_chunks"]]
    assert got["accepted"] == int(t["expected_accepted"]) and got["staged_its"] == int(t["expected_staged_its"])
    np.testing.assert_allclose(got["Es"], _floats(t["expected_Es"]), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(got["dEs"], _floats(t["expected_dEs"]), rtol=1e-12, atol=1e-12)
    assert abs(got["z"] - float(t["expected_z"])) < 1e-12
    assert got["refreshes"] >= 3 and 0 < got["staged_its"] < int(t["iters"])        # refresh! and both branches of the sampler are on the tape
    assert got["min_margin"] > 1e-9


def test_oracle_reproduces_the_round3_tapes(oracle):
    t = TR.read_tape(EA2)
    N, K, seed = int(t["N"]), int(t["K"]), int(t["seed"])
    A, J = _graph(t, N, K)
    assert (A == oracle.gen_ea(int(t["L"]), int(t["D"]))).all() and (J == oracle.gen_couplings(A, seed)).all()
    C0 = np.array([int(c, 16) for c in t["C0"]], np.uint64)
    o = oracle.standard_mc_sparse(A, J, float(t["beta"]), int(t["iters"]), int(t["step"]), seed, C0, form="ea")
    assert [int(e) for e in o[0]] == [int(v) for v in t["expected_Es"]] and o[2] == int(t["expected_accepted"])
    assert [int(c) for c in o[1]] == [int(c, 16) for c in t["expected_chunks"]]
    for path, run, gen in ((SKN, oracle.standard_mc_skn, oracle.gen_sk_gauss), (SKB, oracle.standard_mc_skb, oracle.gen_sk_binary)):
        t = TR.read_tape(path)
        N, seed = int(t["N"]), int(t["seed"])
        Jx = gen(N, seed)
        C0 = np.array([int(c, 16) for c in t["C0"]], np.uint64)
        o = run(Jx, float(t["beta"]), int(t["iters"]), int(t["step"]), seed, C0)
        assert [float(e) for e in o[0]] == _floats(t["expected_Es"]) and o[2] == int(t["expected_accepted"])
        assert [int(c) for c in o[1]] == [int(c, 16) for c in t["expected_chunks"]] and [float(v) for v in o[3]] == _floats(t["expected_lfields"])
    t = TR.read_tape(RSKN)
    N, seed = int(t["N"]), int(t["seed"])
    C0 = np.array([int(c, 16) for c in t["C0"]], np.uint64)
    o = oracle.rrr_mc_skn(oracle.gen_sk_gauss(N, seed), float(t["beta"]), int(t["iters"]), int(t["step"]), seed, C0,
                          staged_thr=float(t["staged_thr"]), staged_thr_fact=float(t["staged_thr_fact"]), want_cache=True)
    assert [float(e) for e in o[0]] == _floats(t["expected_Es"]) and [int(c) for c in o[1]] == [int(c, 16) for c in t["expected_chunks"]]
    assert o[2] == int(t["expected_accepted"]) and o[3] == int(t["expected_staged_its"]) and o[5] == float(t["expected_z"])


@pytest.mark.gpu
def test_hip_library_reproduces_the_round3_tapes(pkg):
    t = TR.read_tape(EA2)
    seed = int(t["seed"])
    X = pkg.GraphEA(int(t["L"]), int(t["D"]), seed=seed)
    with pkg.Engine(X, 32) as eng:
        eng.seed(seed); eng.init_spins_random()
        assert [int(c) for c in eng.get_config().s[0]] == [int(c, 16) for c in t["C0"]]
        Es, acc = eng.standard_mc(float(t["beta"]), int(t["iters"]), int(t["step"]))
        assert [int(e) for e in Es[0]] == [int(v) for v in t["expected_Es"]] and int(acc[0]) == int(t["expected_accepted"])
        assert [int(c) for c in eng.get_config().s[0]] == [int(c, 16) for c in t["expected_chunks"]]
    for path, G in ((SKN, pkg.GraphSKNormal), (SKB, pkg.GraphSK)):
        t = TR.read_tape(path)
        seed = int(t["seed"])
        with pkg.Engine(G(int(t["N"]), seed=seed), 8) as eng:
            eng.seed(seed); eng.init_spins_random()
            assert [int(c) for c in eng.get_config().s[0]] == [int(c, 16) for c in t["C0"]]
            Es, acc = eng.standard_mc(float(t["beta"]), int(t["iters"]), int(t["step"]))
            assert [float(e) for e in Es[0]] == _floats(t["expected_Es"]) and int(acc[0]) == int(t["expected_accepted"])
            assert [int(c) for c in eng.get_config().s[0]] == [int(c, 16) for c in t["expected_chunks"]]
            assert [float(v) for v in eng.fields()[0]] == _floats(t["expected_lfields"])
    t = TR.read_tape(RSKN)
    seed = int(t["seed"])
    with pkg.Engine(pkg.GraphSKNormal(int(t["N"]), seed=seed), 8) as eng:
        eng.seed(seed); eng.init_spins_random()
        Es, acc, staged = eng.rrr_mc(float(t["beta"]), int(t["iters"]), int(t["step"]), staged_thr=float(t["staged_thr"]),
                                     staged_thr_fact=float(t["staged_thr_fact"]))
        assert [float(e) for e in Es[0]] == _floats(t["expected_Es"])
        assert int(acc[0]) == int(t["expected_accepted"]) and int(staged[0]) == int(t["expected_staged_its"])
        assert [int(c) for c in eng.get_config().s[0]] == [int(c, 16) for c in t["expected_chunks"]]


# ---- tapes of the reduced-rejection-rate and rejection-free samplers on GraphRRG: rrrMC(X::SingleGraph) with
#      DeltaECache{Int,L} (staged and direct branch), bklMC with rand_skip ---------------------------------------------------------------
RRRG, BKLG = (os.path.join(GOLD, f) for f in ("tape_rrr_rrg_n64.txt", "tape_bkl_rrg_n64.txt"))


@pytest.mark.parametrize("path", [RRRG, BKLG])
def test_python_replay_reproduces_the_rrg_sampler_tapes(path):
    t = TR.read_tape(path)
    got = TR.replay_rrr_bkl_rrg(t)
    assert got["Es"] == [int(v) for v in t["expected_Es"]] and got["chunks"] == [int(c, 16) for c in t["expected_chunks"]]
    assert got["accepted"] == int(t["expected_accepted"]) and got["staged_its"] == int(t["expected_staged_its"])
    assert got["iters_done"] == int(t["expected_iters_done"])
    assert got["sizes"] == [int(v) for v in t["expected_sizes"]] and got["pos"] == [int(v) for v in t["expected_pos"]]      # the DeltaECache after the run
    assert got["min_margin"] > 1e-9
    if t["kind"] == "rrrMC_rrg":
        assert 0 < got["staged_its"] < int(t["iters"])         # both branches of RRRMC.jl:186-208 are on the tape
    else:
        assert got["accepted"] < got["iters_done"] // 10       # rand_skip does the work: most iterations are skipped rejections


@pytest.mark.parametrize("path", [RRRG, BKLG])
def test_oracle_reproduces_the_rrg_sampler_tapes(oracle, path):
    t = TR.read_tape(path)
    N, K, seed = int(t["N"]), int(t["K"]), int(t["seed"])
    A, J = _graph(t, N, K)
    assert (A == oracle.gen_rrg(N, K, seed)).all() and (J == oracle.gen_couplings(A, seed)).all()
    C0 = np.array([int(c, 16) for c in t["C0"]], np.uint64)
    Es, ch, acc, staged, its, pos, sizes = oracle.rrr_sparse(A, J, float(t["beta"]), int(t["iters"]), int(t["step"]), seed, C0,
                                                             staged_thr=float(t["staged_thr"]), staged_thr_fact=float(t["staged_thr_fact"]),
                                                             bkl=t["kind"] == "bklMC_rrg", want_cache=True)
    assert [int(e) for e in Es] == [int(v) for v in t["expected_Es"]] and [int(c) for c in ch] == [int(c, 16) for c in t["expected_chunks"]]
    assert (acc, staged, its) == (int(t["expected_accepted"]), int(t["expected_staged_its"]), int(t["expected_iters_done"]))
    assert [int(v) + 1 for v in pos] == [int(v) for v in t["expected_pos"]] and [int(v) for v in sizes] == [int(v) for v in t["expected_sizes"]]


@pytest.mark.gpu
@pytest.mark.parametrize("path", [RRRG, BKLG])
def test_hip_library_reproduces_the_rrg_sampler_tapes(pkg, path):
    t = TR.read_tape(path)
    seed = int(t["seed"])
    X = pkg.GraphRRG(int(t["N"]), int(t["K"]), seed=seed)
    with pkg.Engine(X, 32) as eng:
        eng.seed(seed); eng.init_spins_random()
        assert [int(c) for c in eng.get_config().s[0]] == [int(c, 16) for c in t["C0"]]
        if t["kind"] == "rrrMC_rrg":
            Es, acc, staged = eng.rrr_mc(float(t["beta"]), int(t["iters"]), int(t["step"]), staged_thr=float(t["staged_thr"]),
                                         staged_thr_fact=float(t["staged_thr_fact"]))
            assert int(staged[0]) == int(t["expected_staged_its"])
        else:
            Es, acc = eng.bkl_mc(float(t["beta"]), int(t["iters"]), int(t["step"]))
        exp_Es = [int(v) for v in t["expected_Es"]]
        assert [int(e) for e in Es[0][:len(exp_Es)]] == exp_Es and int(acc[0]) == int(t["expected_accepted"])
        assert [int(c) for c in eng.get_config().s[0]] == [int(c, 16) for c in t["expected_chunks"]]
        assert eng.iterations_done() >= 0          # (the DeltaECache itself is checked by the replay and the oracle: the C ABI exposes it for GraphQuant only)


# ---- wtmMC on GraphRRG: the waiting-time method, every rand() of gen_wt on the tape ----------------------------
WTMG = os.path.join(GOLD, "tape_wtm_rrg_n64.txt")


def test_python_replay_reproduces_the_wtm_tape():
    t = TR.read_tape(WTMG)
    got = TR.replay_wtm_rrg(t)
    assert got["Es"] == [int(v) for v in t["expected_Es"]] and got["chunks"] == [int(c, 16) for c in t["expected_chunks"]]
    assert got["num_moves"] == int(t["expected_num_moves"]) > 5 * int(t["N"])
    assert abs(got["t"] - float(t["expected_t"])) <= 1e-12 * got["t"]          # libm's log1p / exp against the oracle's deterministic ones
    assert got["min_margin"] > 1e-9 and got["draws"] <= len(t["uniforms"])


def test_oracle_reproduces_the_wtm_tape(oracle):
    t = TR.read_tape(WTMG)
    N, K, seed = int(t["N"]), int(t["K"]), int(t["seed"])
    A, J = _graph(t, N, K)
    assert (A == oracle.gen_rrg(N, K, seed)).all() and (J == oracle.gen_couplings(A, seed)).all()
    C0 = np.array([int(c, 16) for c in t["C0"]], np.uint64)
    Es, ch, moves, tt, _ = oracle.wtm_mc_sparse(A, J, float(t["beta"]), int(t["samples"]), float(t["step"]), seed, C0)
    assert [int(e) for e in Es] == [int(v) for v in t["expected_Es"]] and [int(c) for c in ch] == [int(c, 16) for c in t["expected_chunks"]]
    assert moves == int(t["expected_num_moves"]) and tt == float(t["expected_t"])


@pytest.mark.gpu
def test_hip_library_reproduces_the_wtm_tape(pkg):
    t = TR.read_tape(WTMG)
    seed = int(t["seed"])
    X = pkg.GraphRRG(int(t["N"]), int(t["K"]), seed=seed)
    with pkg.Engine(X, 32) as eng:
        eng.seed(seed); eng.init_spins_random()
        assert [int(c) for c in eng.get_config().s[0]] == [int(c, 16) for c in t["C0"]]
        Es, moves, tt = eng.wtm_mc(float(t["beta"]), int(t["samples"]), float(t["step"]))
        assert [int(e) for e in Es[0]] == [int(v) for v in t["expected_Es"]] and int(moves[0]) == int(t["expected_num_moves"])
        assert float(tt[0]) == float(t["expected_t"])
        assert [int(c) for c in eng.get_config().s[0]] == [int(c, 16) for c in t["expected_chunks"]]


# ---- extremal_opt on GraphRRG with EOCache{Int,L} ------------------------------------------------------------------
EOG = os.path.join(GOLD, "tape_eo_rrg_n64.txt")


def test_python_replay_reproduces_the_extremal_opt_tape():
    t = TR.read_tape(EOG)
    got = TR.replay_eo_rrg(t)
    assert got["Es"] == [int(v) for v in t["expected_Es"]] and got["chunks"] == [int(c, 16) for c in t["expected_chunks"]]
    assert got["Emin"] == int(t["expected_Emin"]) and got["itmin"] == int(t["expected_itmin"])
    assert got["Cmin"] == [int(c, 16) for c in t["expected_Cmin"]]
    assert 0 < got["itmin"] < int(t["iters"]) and got["min_margin"] > 1e-9


def test_oracle_reproduces_the_extremal_opt_tape(oracle):
    t = TR.read_tape(EOG)
    N, K, seed = int(t["N"]), int(t["K"]), int(t["seed"])
    A, J = _graph(t, N, K)
    assert (A == oracle.gen_rrg(N, K, seed)).all() and (J == oracle.gen_couplings(A, seed)).all()
    C0 = np.array([int(c, 16) for c in t["C0"]], np.uint64)
    Es, ch, Emin, Cmin, itmin = oracle.extremal_opt_sparse(A, J, float(t["tau"]), int(t["iters"]), int(t["step"]), seed, C0)
    assert [int(e) for e in Es] == [int(v) for v in t["expected_Es"]] and [int(c) for c in ch] == [int(c, 16) for c in t["expected_chunks"]]
    assert (Emin, itmin) == (int(t["expected_Emin"]), int(t["expected_itmin"])) and [int(c) for c in Cmin] == [int(c, 16) for c in t["expected_Cmin"]]


@pytest.mark.gpu
def test_hip_library_reproduces_the_extremal_opt_tape(pkg):
    t = TR.read_tape(EOG)
    seed = int(t["seed"])
    X = pkg.GraphRRG(int(t["N"]), int(t["K"]), seed=seed)
    with pkg.Engine(X, 32) as eng:
        eng.seed(seed); eng.init_spins_random()
        assert [int(c) for c in eng.get_config().s[0]] == [int(c, 16) for c in t["C0"]]
        Es, Emin, Cmin, itmin = eng.extremal_opt(float(t["tau"]), int(t["iters"]), int(t["step"]))
        assert [int(e) for e in Es[0]] == [int(v) for v in t["expected_Es"]]
        assert int(Emin[0]) == int(t["expected_Emin"]) and int(itmin[0]) == int(t["expected_itmin"])
        assert [int(c) for c in Cmin.s[0]] == [int(c, 16) for c in t["expected_Cmin"]]
        assert [int(c) for c in eng.get_config().s[0]] == [int(c, 16) for c in t["expected_chunks"]]


# ---- round 4: standardMC on GraphRRGNormal(16, 3) — Float64 couplings and local fields (RRG.jl:503-609), the model of spf_sweep_kernel and of
#      spf_team_kernel (a team of wavefronts per group of 64 replicas): 263 of the 1032 accepted moves take update_cache!'s undo branch ----------
RRGN = os.path.join(GOLD, "tape_rrgn_n16.txt")


def test_python_replay_reproduces_the_rrg_normal_tape():
    t = TR.read_tape(RRGN)
    got = TR.replay_standard_mc_rrgn(t)
    assert got["Es"] == _floats(t["expected_Es"])              # the same sequence of IEEE operations: equal bit for bit
    assert got["chunks"] == [int(c, 16) for c in t["expected_chunks"]] and got["accepted"] == int(t["expected_accepted"])
    assert got["lfields"] == _floats(t["expected_lfields"])
    assert got["undos"] == int(t["expected_undos"]) >= 5       # update_cache!'s undo branch (RRG.jl:566-577) is part of the tape
    assert got["min_margin"] > 1e-9


def test_oracle_reproduces_the_rrg_normal_tape(oracle):
    t = TR.read_tape(RRGN)
    N, K, seed = int(t["N"]), int(t["K"]), int(t["seed"])
    A = np.array([int(v) - 1 for v in t["A"]], np.int32).reshape(N, K)
    J = np.array(_floats(t["J"])).reshape(N, K)
    assert (A == oracle.gen_rrg(N, K, seed)).all() and (J == oracle.gen_couplings_gauss(A, seed)).all()
    C0 = np.array([int(c, 16) for c in t["C0"]], np.uint64)
    assert [oracle.site_of(seed, g, N) + 1 for g in range(1, 50)] == [int(v) for v in t["sites"][:49]]
    o = oracle.standard_mc_spf(A, J, float(t["beta"]), int(t["iters"]), int(t["step"]), seed, C0)
    assert [float(e) for e in o[0]] == _floats(t["expected_Es"]) and o[2] == int(t["expected_accepted"])
    assert [int(c) for c in o[1]] == [int(c, 16) for c in t["expected_chunks"]] and [float(v) for v in o[3]] == _floats(t["expected_lfields"])


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"RRRMC_SPF_TEAM_WAVES": "8"}, {"RRRMC_SPF_TEAM_WIDTH": "32"}, {"RRRMC_SPF_TEAM": "0"}], ids=["team", "team8", "team32w", "single"])
def test_hip_library_reproduces_the_rrg_normal_tape(pkg, monkeypatch, env):
    for k in ("RRRMC_SPF_TEAM", "RRRMC_SPF_TEAM_WAVES", "RRRMC_SPF_TEAM_WIDTH"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    t = TR.read_tape(RRGN)
    seed = int(t["seed"])
    X = pkg.GraphRRGNormal(int(t["N"]), int(t["K"]), seed=seed)
    assert [float(v) for v in np.asarray(X.J).reshape(-1)] == _floats(t["J"])
    with pkg.Engine(X, 8) as eng:
        eng.seed(seed); eng.init_spins_random()
        assert [int(c) for c in eng.get_config().s[0]] == [int(c, 16) for c in t["C0"]]
        Es, acc = eng.standard_mc(float(t["beta"]), int(t["iters"]), int(t["step"]))
        assert [float(e) for e in Es[0]] == _floats(t["expected_Es"]) and int(acc[0]) == int(t["expected_accepted"])
        assert [int(c) for c in eng.get_config().s[0]] == [int(c, 16) for c in t["expected_chunks"]]
        assert [float(v) for v in eng.fields()[0]] == _floats(t["expected_lfields"])


# ---- round 4: rrrMC(X::SingleGraph) on GraphRRGNormal(64, 3) through DeltaECacheCont + DynamicSampler: the model and size range of
#      cont_wave_kernel (one wavefront per replica, the sampler's tree split between LDS and memory); 127 refresh! calls on the tape --------
RRRGN = os.path.join(GOLD, "tape_rrr_rrgn_n64.txt")


def test_python_replay_reproduces_the_rrr_rrg_normal_tape():
    t = TR.read_tape(RRRGN)
    got = TR.replay_rrr_single_sk(t)
    assert got["chunks"] == [int(c, 16) for c in t["expected_chunks"]]
    assert got["accepted"] == int(t["expected_accepted"]) and got["staged_its"] == int(t["expected_staged_its"])
    np.testing.assert_allclose(got["Es"], _floats(t["expected_Es"]), rtol=1e-12, atol=1e-12)
    assert got["refreshes"] >= 3 and 0 < got["staged_its"] < int(t["iters"])        # refresh! and both branches of the sampler are on the tape
    assert got["min_margin"] > 1e-9


def test_oracle_reproduces_the_rrr_rrg_normal_tape(oracle):
    t = TR.read_tape(RRRGN)
    N, K, seed = int(t["N"]), int(t["K"]), int(t["seed"])
    A = np.array([int(v) - 1 for v in t["A"]], np.int32).reshape(N, K)
    J = np.array(_floats(t["J"])).reshape(N, K)
    assert (A == oracle.gen_rrg(N, K, seed)).all() and (J == oracle.gen_couplings_gauss(A, seed)).all()
    C0 = np.array([int(c, 16) for c in t["C0"]], np.uint64)
    o = oracle.cont_sparse("rrr", A, J, float(t["beta"]), int(t["iters"]), int(t["step"]), seed, C0, staged_thr=float(t["staged_thr"]),
                           staged_thr_fact=float(t["staged_thr_fact"]))
    assert [float(e) for e in o[0]] == _floats(t["expected_Es"]) and [int(c) for c in o[1]] == [int(c, 16) for c in t["expected_chunks"]]
    assert int(o[2][0]) == int(t["expected_accepted"]) and int(o[2][1]) == int(t["expected_staged_its"])


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"RRRMC_CONT_NO_WAVE": "1"}], ids=["wave", "thread"])
def test_hip_library_reproduces_the_rrr_rrg_normal_tape(pkg, monkeypatch, env):
    monkeypatch.delenv("RRRMC_CONT_NO_WAVE", raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    t = TR.read_tape(RRRGN)
    seed = int(t["seed"])
    X = pkg.GraphRRGNormal(int(t["N"]), int(t["K"]), seed=seed)
    with pkg.Engine(X, 8) as eng:
        eng.seed(seed); eng.init_spins_random()
        assert [int(c) for c in eng.get_config().s[0]] == [int(c, 16) for c in t["C0"]]
        Es, acc, staged = eng.rrr_mc(float(t["beta"]), int(t["iters"]), int(t["step"]), staged_thr=float(t["staged_thr"]),
                                     staged_thr_fact=float(t["staged_thr_fact"]))
        assert [float(e) for e in Es[0]] == _floats(t["expected_Es"])
        assert int(acc[0]) == int(t["expected_accepted"]) and int(staged[0]) == int(t["expected_staged_its"])
        assert [int(c) for c in eng.get_config().s[0]] == [int(c, 16) for c in t["expected_chunks"]]


# ---- round 6 tape: rrrMC(X::DoubleGraph) on GraphQEAT(4, 2, 8) = GraphQuant over GraphEANormal slices (src/QAliases.jl:50-83) ----------------------
QEAT = os.path.join(GOLD, "tape_quant_qeat_l4_m8.txt")


def _qeat_graph(t):
    Nk, K = int(t["Nk"]), int(t["K"])
    A = np.array([int(v) - 1 for v in t["A"]], np.int32).reshape(Nk, K)
    J = np.array([float(v) for v in t["J"]], np.float64).reshape(Nk, K)
    return A, J


def test_python_replay_reproduces_the_qeat_tape():
    t = TR.read_tape(QEAT)
    got = TR.replay_rrr_quant(t)
    assert got["chunks"] == [int(c, 16) for c in t["expected_chunks"]]
    assert got["accepted"] == int(t["expected_accepted"]) and got["staged_its"] == int(t["expected_staged_its"])
    assert got["sizes"] == [int(v) for v in t["expected_sizes"]] and got["pos"] == [int(v) for v in t["expected_pos"]]
    assert got["Es"] == [float(v) for v in t["expected_Es"]]          # Float64 slice caches: the same sequence of IEEE operations, bit for bit
    assert got["min_margin"] > 1e-9 and 0 < got["staged_its"] < int(t["iters"]) and got["accepted"] < int(t["iters"])


def test_oracle_reproduces_the_qeat_tape(oracle):
    t = TR.read_tape(QEAT)
    A, J = _qeat_graph(t)
    assert (A == oracle.gen_ea(int(t["L"]), int(t["D"]))).all()
    C0 = np.array([int(c, 16) for c in t["C0"]], np.uint64)
    ref = oracle.rrr_mc_quant_spf(A, J, int(t["M"]), float(t["fourK"]), float(t["beta"]), int(t["iters"]), int(t["step"]), int(t["seed"]), C0,
                                  staged_thr=float(t["staged_thr"]), staged_thr_fact=float(t["staged_thr_fact"]), want_cache=True)
    assert [float(e) for e in ref[0]] == [float(v) for v in t["expected_Es"]] and [int(c) for c in ref[1]] == [int(c, 16) for c in t["expected_chunks"]]
    assert ref[2] == int(t["expected_accepted"]) and ref[3] == int(t["expected_staged_its"])
    assert [int(v) + 1 for v in ref[4]] == [int(v) for v in t["expected_pos"]] and [int(v) for v in ref[5]] == [int(v) for v in t["expected_sizes"]]


@pytest.mark.gpu
def test_hip_library_reproduces_the_qeat_tape(pkg):
    t = TR.read_tape(QEAT)
    A, J = _qeat_graph(t)
    M, seed = int(t["M"]), int(t["seed"])
    X = pkg.GraphQEAT(pkg.GraphEANormal.from_AJ(A, J), M, float(t["Gamma"]), float(t["beta"]))
    assert X.fourK == float(t["fourK"]) and X.f64_slices
    with pkg.Engine(X, 3) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        assert [int(c) for c in eng.get_config().s[0]] == [int(c, 16) for c in t["C0"]]
        Es, acc, staged = eng.rrr_mc(float(t["beta"]), int(t["iters"]), int(t["step"]), staged_thr=float(t["staged_thr"]),
                                     staged_thr_fact=float(t["staged_thr_fact"]))
        C1 = eng.get_config()
        pos, sizes = eng.rrr_cache()
    assert [float(e) for e in Es[0]] == [float(v) for v in t["expected_Es"]]
    assert int(acc[0]) == int(t["expected_accepted"]) and int(staged[0]) == int(t["expected_staged_its"])
    assert [int(c) for c in C1.s[0]] == [int(c, 16) for c in t["expected_chunks"]]
    assert [int(v) + 1 for v in pos[0]] == [int(v) for v in t["expected_pos"]] and [int(v) for v in sizes[0]] == [int(v) for v in t["expected_sizes"]]
