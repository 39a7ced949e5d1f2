"""CPU tests of the oracle's model code against what the reference itself pins: the
tracked-energy invariant of test/runtests.jl:12-20, cache == recomputation, allΔE tables, a closed-form
toy model and the exact Boltzmann law for a tiny system.  No GPU."""
import itertools
import math

import numpy as np
import pytest


def _bits(ch, N):
    x = np.arange(N)
    return ((ch[x >> 6] >> (x & 63).astype(np.uint64)) & np.uint64(1)).astype(np.int64)


def brute_energy(A, J, bits):
    s = 2 * bits - 1
    N, K = A.shape
    return -int(sum(J[x, k] * s[x] * s[A[x, k]] for x in range(N) for k in range(K))) // 2


def test_gen_rrg_is_simple_regular_sorted(oracle):
    for N, K, seed in [(10, 3, 1), (128, 3, 2), (50, 4, 3), (64, 5, 4)]:
        A = oracle.gen_rrg(N, K, seed)
        assert A.shape == (N, K)
        for x in range(N):
            row = list(A[x])
            assert row == sorted(row) and len(set(row)) == K and x not in row       # findall on a bit row: RRG.jl:64
            for y in row:
                assert x in A[y]
    with pytest.raises(ValueError):
        oracle.gen_rrg(5, 3, 1)                                                     # N*K must be even, RRG.jl:28


def test_gen_ea_lattice(oracle):
    A = oracle.gen_ea(4, 3)                 # column-major linear index, periodic, sorted rows (EA.jl:24-43)
    assert A.shape == (64, 6)
    x = 1 + 4 * 2 + 16 * 3                  # (1, 2, 3)
    want = sorted([0 + 8 + 48, 2 + 8 + 48, 1 + 4 + 48, 1 + 12 + 48, 1 + 8 + 32, 1 + 8 + 0])
    assert list(A[x]) == want
    A2 = oracle.gen_ea(2, 3)                # L = 2: every neighbour twice (EA.jl:156)
    assert list(A2[0]) == [1, 1, 2, 2, 4, 4]
    A3 = oracle.gen_ea(3, 2)
    assert list(A3[0]) == [1, 2, 3, 6]


def test_couplings_symmetric_pm1(oracle):
    for A in (oracle.gen_rrg(40, 3, 5), oracle.gen_ea(2, 3), oracle.gen_ea(3, 2)):
        J = oracle.gen_couplings(A, 11)
        assert set(np.unique(J)) <= {-1, 1}
        N, K = A.shape
        used = np.zeros((N, K), bool)
        for x in range(N):
            for k in range(K):
                y = A[x, k]
                ok = False
                for l in range(K):
                    if not used[y, l] and A[y, l] == x and J[y, l] == J[x, k]:
                        used[y, l] = ok = True
                        break
                assert ok


def test_all_delta_e_tables(oracle):
    assert oracle.all_delta_e_pm1(3) == (2, 6)              # GraphRRG K=3: RRG.jl:262-265
    assert oracle.all_delta_e_pm1(6) == (0, 4, 8, 12)       # GraphEA D=3: EA.jl:293
    assert oracle.all_delta_e_pm1(4) == (0, 4, 8)
    assert oracle.all_delta_e_pm1(1) == (2,)                # GraphTwoSpin: TwoSpin.jl:41


@pytest.mark.parametrize("kind", ["rrg10", "ea23", "ea32"])
def test_energy_and_fields_match_brute_force(oracle, kind):
    A = {"rrg10": lambda: oracle.gen_rrg(10, 3, 3), "ea23": lambda: oracle.gen_ea(2, 3), "ea32": lambda: oracle.gen_ea(3, 2)}[kind]()
    J = oracle.gen_couplings(A, 17)
    N, K = A.shape
    for r in range(8):
        ch = oracle.init_config(5, r, N)
        E, lf = oracle.sparse_energy(A, J, ch, want_fields=True)
        bits = _bits(ch, N)
        assert E == brute_energy(A, J, bits)
        for x in range(N):                                  # lfields[x] = -dE(x): flip and recompute
            b2 = bits.copy()
            b2[x] ^= 1
            assert -lf[x] == brute_energy(A, J, b2) - E
        assert set(abs(int(v)) for v in lf) <= set(oracle.all_delta_e_pm1(K))      # Interface.jl:123-125


@pytest.mark.parametrize("kind,form", [("rrg10", "rrg"), ("ea23", "ea"), ("ea32", "ea"), ("rrg10", "ea")])
def test_tracked_energy_equals_recomputed(oracle, kind, form):
    """The reference's only assertion (test/runtests.jl:12-20,125-130): E tracked by the sampler == energy(X, C),
    here at every sample by re-running prefixes; plus cache == recomputation at the end (RRG.jl:229-231)."""
    A = {"rrg10": lambda: oracle.gen_rrg(10, 3, 3), "ea23": lambda: oracle.gen_ea(2, 3), "ea32": lambda: oracle.gen_ea(3, 2)}[kind]()
    J = oracle.gen_couplings(A, 23)
    N = A.shape[0]
    seed, beta, iters, step = 8426732438942, 2.0, 10000, 100
    ch0 = oracle.init_config(seed, 0, N)
    Es, ch1, acc, lf = oracle.standard_mc_sparse(A, J, beta, iters, step, seed, ch0, form=form)
    assert len(Es) == iters // step
    E1, lf1 = oracle.sparse_energy(A, J, ch1, want_fields=True)
    assert (lf == lf1).all()
    for k in (1, 2, 37, 100):                               # sample k is taken BEFORE the move of iteration k*step
        _, chk, _, _ = oracle.standard_mc_sparse(A, J, beta, k * step - 1, step, seed, ch0, form=form)
        assert Es[k - 1] == oracle.sparse_energy(A, J, chk)
    # resumed run (C0 = previous C, stream continued) == one long run
    Es_a, ch_a, acc_a, _ = oracle.standard_mc_sparse(A, J, beta, 6000, step, seed, ch0, form=form)
    Es_b, ch_b, acc_b, _ = oracle.standard_mc_sparse(A, J, beta, 4000, step, seed, ch_a, it0=6000, form=form)
    assert (np.concatenate([Es_a, Es_b]) == Es).all() and (ch_b == ch1).all() and acc_a + acc_b == acc


def test_two_spin_closed_form(oracle):
    """GraphTwoSpin (graphs/TwoSpin.jl:26-41): E = -s1 s2, dE = 2 s1 s2 — as the K = 1 sparse model."""
    A = np.array([[1], [0]], np.int32)
    J = np.array([[1], [1]], np.int32)
    for b0, b1 in itertools.product((0, 1), repeat=2):
        ch = np.array([b0 | (b1 << 1)], np.uint64)
        E, lf = oracle.sparse_energy(A, J, ch, want_fields=True)
        s0, s1 = 2 * b0 - 1, 2 * b1 - 1
        assert E == -s0 * s1 and list(-lf) == [2 * s0 * s1] * 2


def test_three_spin_closed_form(oracle):
    """GraphThreeSpin (graphs/ThreeSpin.jl:26-47): E = -(s1 s2 + s2 s3 + s3 s1), dE(i) = 2 s_i (sum of the other two), allΔE = (0, 4) —
    the K = 2 triangle as a sparse model; also under rrrMC / bklMC / wtmMC / extremal_opt (runtests.jl:27 x :140-163)."""
    A = np.array([[1, 2], [0, 2], [0, 1]], np.int32)
    J = np.ones((3, 2), np.int32)
    assert oracle.all_delta_e_pm1(2) == (0, 4)
    for bits in itertools.product((0, 1), repeat=3):
        ch = np.array([bits[0] | (bits[1] << 1) | (bits[2] << 2)], np.uint64)
        s = [2 * b - 1 for b in bits]
        E, lf = oracle.sparse_energy(A, J, ch, want_fields=True)
        assert E == -(s[0] * s[1] + s[1] * s[2] + s[2] * s[0])
        assert list(-lf) == [2 * s[0] * (s[1] + s[2]), 2 * s[1] * (s[0] + s[2]), 2 * s[2] * (s[0] + s[1])]
    ch = np.array([0b010], np.uint64)
    for bkl in (False, True):
        r = oracle.rrr_sparse(A, J, 2.0, 2000, 10, 11, ch, bkl=bkl)          # beta of runtests.jl:132
        assert set(np.unique(r[0])) <= {-3, 1}
    w = oracle.wtm_mc_sparse(A, J, 2.0, 100, 1.0, 11, ch)
    assert set(np.unique(w[0])) <= {-3, 1} and w[4] == oracle.sparse_energy(A, J, w[1])
    e = oracle.extremal_opt_sparse(A, J, 1.3, 500, 5, 11, ch)
    assert e[2] == -3 and oracle.sparse_energy(A, J, e[3]) == -3


def test_boltzmann_law_small_system(oracle):
    """truep (src/RRRMC.jl:528-543): the chain's stationary law is exp(-beta E)/Z.  N = 8 ring-with-chords graph,
    long single chain of the oracle, total-variation distance to the exact law."""
    N, beta = 8, 0.7
    A = oracle.gen_rrg(N, 3, 12)
    J = oracle.gen_couplings(A, 12)
    p = np.zeros(1 << N)
    for c in range(1 << N):
        p[c] = math.exp(-beta * oracle.sparse_energy(A, J, np.array([c], np.uint64)))
    p /= p.sum()
    seed, step, nsamp = 2024, 8, 60000
    ch = oracle.init_config(seed, 0, N)
    hist = np.zeros(1 << N)
    it0 = 0
    for _ in range(nsamp // 2000):
        for _ in range(2000):
            _, ch, _, _ = oracle.standard_mc_sparse(A, J, beta, step, step + 1, seed, ch, it0=it0)
            it0 += step
            hist[int(ch[0])] += 1
    hist /= hist.sum()
    assert 0.5 * np.abs(hist - p).sum() < 0.04
    # mean energy against the exact value
    Es = np.array([oracle.sparse_energy(A, J, np.array([c], np.uint64)) for c in range(1 << N)])
    assert abs((hist * Es).sum() - (p * Es).sum()) < 0.15


# ---- Float64-coupling sparse models GraphRRGNormal / GraphEANormal ----

def _brute_energy_f64(A, J, bits):
    s = 2 * bits - 1
    N, K = A.shape
    return -sum(J[x, k] * s[x] * s[A[x, k]] for x in range(N) for k in range(K)) / 2


@pytest.mark.parametrize("kind,form", [("rrg", "rrg"), ("ea3", "ea"), ("ea2L2", "ea")])
def test_spf_tracked_energy_and_cache(oracle, kind, form):
    """test/runtests.jl:12-20 on GraphRRGNormal(10,3) / GraphEANormal(3,2) (runtests.jl:40,60) + the L=2 double-bond lattice."""
    seed = 99
    A = {"rrg": lambda: oracle.gen_rrg(10, 3, seed), "ea3": lambda: oracle.gen_ea(3, 2), "ea2L2": lambda: oracle.gen_ea(2, 3)}[kind]()
    J = oracle.gen_couplings_gauss(A, seed)
    N, K = A.shape
    for x in range(N):                       # symmetric: bond (x, y) carries one value from both ends
        for k in range(K):
            y = A[x, k]
            assert J[x, k] in J[y][A[y] == x]
    ch = oracle.init_config(seed, 0, N)
    E0, lf0 = oracle.spf_energy(A, J, ch, want_fields=True, form=form)
    assert math.isclose(E0, _brute_energy_f64(A, J, _bits(ch, N)), rel_tol=1e-12, abs_tol=1e-12)
    Es, ch1, acc, lf = oracle.standard_mc_spf(A, J, 1.5, 4000, 100, seed, ch, form=form)
    assert len(Es) == 40 and Es[0] != Es[-1] and 0 < acc < 4000
    E1, lf1 = oracle.spf_energy(A, J, ch1, want_fields=True, form=form)
    assert np.allclose(lf, lf1, rtol=0, atol=1e-11)                  # incremental cache == recomputation
    # continue the chain sample by sample: tracked energy == energy(X, C) within the reference's 1e-11 (runtests.jl:17)
    ch2 = ch.copy()
    for k in range(1, 6):
        Es_k, ch2, _, _ = oracle.standard_mc_spf(A, J, 1.5, 100, 100, seed, ch2, it0=100 * (k - 1), form=form)
        # Es_k[0] is the energy before iteration 100 of this leg == tracked energy after 99 moves; compare a fresh energy
        assert math.isclose(Es_k[0], Es[k - 1], rel_tol=0, abs_tol=1e-11)


def test_spf_boltzmann_small(oracle):
    """Exact Boltzmann law on a tiny GraphRRGNormal (the reference's truep check, RRRMC.jl:528-543) through many short chains."""
    seed, N, beta = 5, 6, 0.7
    A = oracle.gen_rrg(N, 3, seed)
    J = oracle.gen_couplings_gauss(A, seed)
    counts = np.zeros(2 ** N)
    R, iters = 3000, 400
    for r in range(R):
        ch = oracle.init_config(seed, r, N)
        _, ch1, _, _ = oracle.standard_mc_spf(A, J, beta, iters, iters, seed, ch, replica=r)
        counts[int(ch1[0])] += 1
    Es = np.array([_brute_energy_f64(A, J, np.array([(c >> i) & 1 for i in range(N)])) for c in range(2 ** N)])
    p = np.exp(-beta * Es)
    p /= p.sum()
    # chains share the site stream, so they are not independent; allow 6 sigma of the multinomial error
    assert np.abs(counts / R - p).max() < 6 * np.sqrt(p.max() / R) + 0.01


# ---- DoubleGraphs Graph{RRG,EA}NormalDiscretized under rrrMC ----

def test_all_delta_e_and_discretize(oracle):
    assert oracle.all_delta_e(3, (-1, 1)) == (2, 6)                     # RRG.jl:262-266
    assert oracle.all_delta_e(6, (-1, 1)) == (0, 4, 8, 12)              # EA.jl:293
    assert oracle.all_delta_e(3, (-1, 0, 1)) == (0, 2, 4, 6)
    assert oracle.all_delta_e(2, (-2, 1)) == (0, 2, 4, 6, 8)            # sums of 2 terms +-{1,2}: 0, +-1, +-2, +-3, +-4
    d, r = oracle.discretize(np.array([0.2, -0.7, 0.5, -0.5, 3.0]), (-1, 0, 1))
    assert list(d) == [0, -1, 0, -1, 1]                                 # ties keep the first level (strict <, Common.jl:44)
    assert np.allclose(r, [0.2, 0.3, 0.5, 0.5, 2.0])


def test_dfloat_levels(oracle):
    """DFloat64 (src/DFloats.jl:11-36): t = round(x * 10^5); the oracle works in units t / gcd."""
    assert oracle.dfloat_units((-1, 0, 1)) == ((-1, 0, 1), 1, 1.0)
    assert oracle.dfloat_units((-1.5, -0.5, 0.5, 1.5)) == ((-3, -1, 1, 3), 50000, 100000.0)
    assert oracle.dfloat_units((-0.3, 0.3)) == ((-1, 1), 30000, 100000.0)
    assert oracle.dfloat_units((0.123456, 1.0)) == ((6173, 50000), 2, 100000.0)          # t = (12346, 100000): rounded to 5 digits, gcd 2
    lev, mul, div = oracle.dfloat_units((-1.5, -0.5, 0.5, 1.5))
    x = np.array([0.2, -0.7, 1.0, -1.0, 3.0, 0.0])
    d, r = oracle.discretize(x, lev, mul, div)
    assert list(d) == [1, -1, 1, -3, 3, -1]                # ties (x = 1.0, -1.0, 0.0) keep the first level in LEV order
    assert np.allclose(d * 0.5 + r, x, rtol=0, atol=1e-15)
    # unit levels in DFloat64 form reproduce the Int-level energies bit for bit
    A = oracle.gen_rrg(40, 3, 5)
    cJ = oracle.gen_couplings_gauss(A, 5)
    di, ri = oracle.discretize(cJ, (-1, 0, 1))
    lf, mf, df = oracle.dfloat_units((-1.0, 0.0, 1.0))
    dfl, rfl = oracle.discretize(cJ, lf, mf, df)
    assert (di == dfl).all() and (ri == rfl).all()
    ch = oracle.init_config(5, 0, 40)
    a = oracle.rrr_double_sparse(A, di, ri, (-1, 0, 1), 1.5, 2000, 10, 5, ch)
    b = oracle.rrr_double_sparse(A, dfl, rfl, lf, 1.5, 2000, 10, 5, ch, mul=mf, div=df)
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all()
    a = oracle.standard_mc_dbl(A, di, ri, 1.5, 2000, 10, 5, ch)
    b = oracle.standard_mc_dbl(A, dfl, rfl, 1.5, 2000, 10, 5, ch, mul=mf, div=df)
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and a[2] == b[2] > 0


@pytest.mark.parametrize("form,lev", [("rrg", (-1, 0, 1)), ("rrg", (-1.5, -0.5, 0.5, 1.5)), ("ea", (-0.75, 0.0, 0.75))])
def test_double_graph_standard_mc_tracked_energy(oracle, form, lev):
    """The reference's invariant (test/runtests.jl:12-20) for standardMC on the DoubleGraphs, Int and DFloat64 levels."""
    seed = 321
    A = oracle.gen_rrg(10, 3, seed) if form == "rrg" else oracle.gen_ea(3, 2)
    cJ = oracle.gen_couplings_gauss(A, seed)
    lev, mul, div = oracle.dfloat_units(lev)
    dJ, rJ = oracle.discretize(cJ, lev, mul, div)
    assert np.allclose(dJ * (mul / div) + rJ, cJ, rtol=0, atol=1e-15)
    N = A.shape[0]
    ch = oracle.init_config(seed, 0, N)
    n = 3000
    Es, ch_n, acc = oracle.standard_mc_dbl(A, dJ, rJ, 1.2, n, 1, seed, ch, form=form, mul=mul, div=div)
    assert 0 < acc < n and abs(Es[0] - oracle.dbl_energy(A, dJ, rJ, ch, form=form, mul=mul, div=div)) == 0
    # replay: the tracked energy at sample k equals energy(X, C_k) within 1e-11
    c = ch.copy()
    for k in (1, 10, 500, n):
        _, ck, _ = oracle.standard_mc_dbl(A, dJ, rJ, 1.2, k - 1, 1, seed, ch, form=form, mul=mul, div=div)
        assert abs(Es[k - 1] - oracle.dbl_energy(A, dJ, rJ, ck, form=form, mul=mul, div=div)) < 1e-11
    # the same chain on the undiscretised couplings (GraphRRGNormal with cJ = dJ + rJ) follows the same trajectory here
    Es_f, ch_f, acc_f, _ = oracle.standard_mc_spf(A, cJ, 1.2, n, 1, seed, ch, form=form)
    assert np.allclose(Es_f, Es, rtol=0, atol=1e-9)


@pytest.mark.parametrize("kind,form,lev,thr", [
    ("rrg", "rrg", (-1, 0, 1), 0.5), ("rrg", "rrg", (-1, 1), 0.0), ("rrg", "rrg", (-1, 0, 1), 1.0),
    ("ea3", "ea", (-1, 0, 1), 0.5), ("ea2L2", "ea", (-1, 0, 1), 0.5),
])
def test_double_graph_tracked_energy(oracle, kind, form, lev, thr):
    """test/runtests.jl:41-44,61-64 (GraphRRGNormalDiscretized(10,3,(-1,0,1)), GraphEANormalDiscretized(2,3,...), (3,2,...))
    with the tracked-energy invariant of :12-20 and the staged_thr = 0 / 1 variants of :150-159."""
    seed = 123
    A = {"rrg": lambda: oracle.gen_rrg(10, 3, seed), "ea3": lambda: oracle.gen_ea(3, 2), "ea2L2": lambda: oracle.gen_ea(2, 3)}[kind]()
    cJ = oracle.gen_couplings_gauss(A, seed)
    dJ, rJ = oracle.discretize(cJ, lev)
    assert np.allclose(dJ + rJ, cJ, rtol=0, atol=1e-15) and set(np.unique(dJ)) <= set(lev)
    N = A.shape[0]
    ch = oracle.init_config(seed, 0, N)
    n = 3000
    Es, ch_n, acc, staged, pos, sizes = oracle.rrr_double_sparse(A, dJ, rJ, lev, 1.2, n, 1, seed, ch, staged_thr=thr, form=form)
    assert sizes.sum() == N and 0 < acc <= n
    if thr == 0.0:
        assert staged == 0
    if thr == 1.0:
        assert staged == n
    # energy before iteration k+1 == energy of the configuration after k iterations, replayed chain by chain
    for k in (1, 7, 500, n - 1):
        _, ch_k, *_ = oracle.rrr_double_sparse(A, dJ, rJ, lev, 1.2, k, k, seed, ch, staged_thr=thr, form=form)
        assert abs(Es[k] - oracle.dbl_energy(A, dJ, rJ, ch_k, form=form)) < 1e-9
    # the double graph's energy is the energy of the undiscretized Gaussian model
    assert abs(oracle.dbl_energy(A, dJ, rJ, ch_n, form=form) - oracle.spf_energy(A, cJ, ch_n, form=form)) < 1e-9


# ---- wtmMC (waiting-time method) on GraphRRG / GraphEA ----

@pytest.mark.parametrize("kind,form", [("rrg", "rrg"), ("ea2L2", "ea")])
def test_wtm_tracked_energy_and_sampling_times(oracle, kind, form):
    seed = 31
    A = oracle.gen_rrg(20, 3, seed) if kind == "rrg" else oracle.gen_ea(2, 3)
    J = oracle.gen_couplings(A, seed)
    N = A.shape[0]
    ch = oracle.init_config(seed, 0, N)
    Es, ch1, moves, t, Ef = oracle.wtm_mc_sparse(A, J, 1.2, 400, 1.0, seed, ch, form=form)
    assert len(Es) == 400 and moves > 0
    assert t < 400 / N + 1e-9                                   # global time: `samples` steps of 1/N sweeps (RRRMC.jl:391-393)
    assert Ef == oracle.sparse_energy(A, J, ch1)                # tracked energy == energy(X, C): test/runtests.jl:12-20
    # a different call index draws different waiting times
    Es2, *_ = oracle.wtm_mc_sparse(A, J, 1.2, 400, 1.0, seed, ch, call=1, form=form)
    assert (Es2 != Es).any()


def test_wtm_equilibrium_matches_metropolis(oracle):
    """The waiting-time method samples the same Boltzmann law as Metropolis: compare the time-averaged energy (loose, statistical)."""
    seed, N, beta = 7, 16, 0.8
    A = oracle.gen_rrg(N, 3, seed)
    J = oracle.gen_couplings(A, seed)
    e_wtm, e_met = [], []
    for r in range(20):
        ch = oracle.init_config(seed, r, N)
        Es, *_ = oracle.wtm_mc_sparse(A, J, beta, 40000, 1.0, seed, ch, replica=r)
        e_wtm.append(Es[4000:].mean())
        Em, *_ = oracle.standard_mc_sparse(A, J, beta, 40000, 1, seed, ch, replica=r)
        e_met.append(Em[4000:].mean())
    assert abs(np.mean(e_wtm) - np.mean(e_met)) < 0.35           # both ~ -11; the spread of a 20-chain mean is ~0.1


# ---- extremal_opt (tau-EO) on GraphRRG / GraphEA ----

@pytest.mark.parametrize("kind,form", [("rrg", "rrg"), ("ea", "ea"), ("ea2L2", "ea")])
def test_extremal_opt_invariants(oracle, kind, form):
    seed = 17
    A = {"rrg": lambda: oracle.gen_rrg(40, 3, seed), "ea": lambda: oracle.gen_ea(4, 2), "ea2L2": lambda: oracle.gen_ea(2, 3)}[kind]()
    J = oracle.gen_couplings(A, seed)
    N = A.shape[0]
    ch = oracle.init_config(seed, 0, N)
    E0 = oracle.sparse_energy(A, J, ch)
    Es, ch1, Emin, Cmin, itmin = oracle.extremal_opt_sparse(A, J, 1.3, 3000, 10, seed, ch, form=form)   # also checks cache + tracked E
    assert len(Es) == 300 and Emin <= min(E0, Es.min()) and 0 <= itmin <= 3000
    assert Emin == oracle.sparse_energy(A, J, Cmin)                 # Cmin is the configuration of minimum energy
    # tau-EO is a descent-biased walk: it finds something well below the random start
    assert Emin < E0
    # ground-state check on the tiny ring-like case by brute force
    if N <= 16:
        best = min(brute_energy(A, J, np.array([(c >> i) & 1 for i in range(N)])) for c in range(2 ** N))
        assert Emin >= best


# ---- stand-alone GraphRRG / GraphEA with general levels (test/runtests.jl:36-60 x :140-163) ----------------------------------
@pytest.mark.parametrize("kind,lev,mul,div", [
    ("rrg", (-1, 0, 1), 1, 1.0),                    # GraphRRG(10, 3, (-1,0,1))          runtests.jl:37
    ("rrg", (-1, 0, 1), 100000, 100000.0),          # GraphRRG(10, 3, (-1.0,0.0,1.0))    :38 (DFloat64, t = +-10^5)
    ("rrg", (-3, -1, 1, 3), 50000, 100000.0),       # levels (-1.5,-0.5,0.5,1.5)
    ("ea2", (-1, 0, 1), 1, 1.0),                    # GraphEA(2, 3, (-1,0,1))            :47 (double bonds)
    ("ea3", (-1, 0, 1), 1, 3.0),                    # GraphEA(3, 2, (-1//3,0//1,1//3))   rational levels over the denominator 3
])
def test_general_level_graphs_all_samplers(oracle, kind, lev, mul, div):
    """Every sampler keeps E_tracked == energy(X, C) (runtests.jl:12-20; checked inside the oracle for standardMC / extremal_opt, by the
    DeltaECache consistency check for rrrMC / bklMC), zero couplings drop out of neighbors(X, i) for GraphRRG (RRG.jl:133), and the
    level scale only enters through exp(-beta dE)."""
    seed = 2024
    A = {"rrg": lambda: oracle.gen_rrg(10, 3, seed), "ea2": lambda: oracle.gen_ea(2, 3), "ea3": lambda: oracle.gen_ea(3, 2)}[kind]()
    form = "rrg" if kind == "rrg" else "ea"
    J = oracle.gen_couplings(A, seed, lev)
    assert set(np.unique(J)) <= set(lev) and ((J == 0).any() or 0 not in lev)
    N = A.shape[0]
    ch = oracle.init_config(seed, 0, N)
    sc = dict(mul=mul, div=div)
    E0 = oracle.sparse_energy(A, J, ch)
    Es, c1, acc = oracle.standard_mc_lev(A, J, 0.9, 4000, 1, seed, ch, form=form, **sc)
    assert Es[0] == E0 and 0 < acc < 4000 and len(Es) == 4000
    for thr in (0.5, 0.0, 1.0):
        r = oracle.rrr_sparse(A, J, 0.9, 3000, 1, seed, ch, staged_thr=thr, form=form, lev=lev, **sc)
        assert r[0][0] == E0 and r[2] > 0
    b = oracle.rrr_sparse(A, J, 0.9, 3000, 10, seed, ch, form=form, bkl=True, lev=lev, **sc)
    assert b[2] > 0
    w = oracle.wtm_mc_sparse(A, J, 0.9, 50, 1.0, seed, ch, form=form, **sc)
    assert len(w[0]) == 50 and w[4] == oracle.sparse_energy(A, J, w[1])
    e = oracle.extremal_opt_sparse(A, J, 1.2, 2000, 10, seed, ch, form=form, lev=lev)
    assert e[2] == oracle.sparse_energy(A, J, e[3]) and e[2] <= min(e[0].min(), E0)
    # the scale matters only through beta * dE: (lev, mul, div) and (lev, 1, 1.0) at beta * mul / div give the same chain
    Es2, c2, acc2 = oracle.standard_mc_lev(A, J, 0.9, 4000, 1, seed, ch, form=form, mul=1, div=1.0)
    if mul / div == 1.0:
        assert (Es2 == Es).all() and (c2 == c1).all()


def test_quant_standard_mc_tracked_energy(oracle):
    """standardMC on GraphQuant(10, 8, 0.5, 2.0, GraphRRG, 10, 3) (runtests.jl:78 x :141-143): E_tracked ~ energy(X, C) (:12-20)."""
    seed, Nk, M, beta, Gamma = 99, 10, 8, 2.0, 0.5
    A = oracle.gen_rrg(Nk, 3, seed)
    J = oracle.gen_couplings(A, seed)
    fourK = oracle.quant_fourK(beta, Gamma, M)
    ch = oracle.init_config(seed, 0, Nk * M)
    Es, ch1, acc = oracle.standard_mc_quant(A, J, M, fourK, beta, 5000, 1, seed, ch)
    assert Es[0] == oracle.quant_energy(A, J, M, fourK, ch)[0] and 0 < acc < 5000
    for k in (2, 77, 1234, 5000):
        _, ck, _ = oracle.standard_mc_quant(A, J, M, fourK, beta, k - 1, 1, seed, ch)
        assert abs(Es[k - 1] - oracle.quant_energy(A, J, M, fourK, ck)[0]) < 1e-11


@pytest.mark.parametrize("form,lev", [("rrg", (-1, 0, 1)), ("ea", (-0.75, 0.0, 0.75))])
def test_double_graph_bkl_wtm_tracked_energy(oracle, form, lev):
    """bklMC / wtmMC on the discretised DoubleGraphs (runtests.jl:41,61 x :145-151): the tracked energy stays equal to energy(X, C)."""
    seed = 654
    A = oracle.gen_rrg(10, 3, seed) if form == "rrg" else oracle.gen_ea(3, 2)
    cJ = oracle.gen_couplings_gauss(A, seed)
    units, mul, div = oracle.dfloat_units(lev)
    dJ, rJ = oracle.discretize(cJ, units, mul, div)
    ch = oracle.init_config(seed, 0, A.shape[0])
    sc = dict(mul=mul, div=div)
    for n in (1, 7, 300):
        Es, c1, stats, _ = oracle.cont_double("bkl", A, dJ, rJ, 1.1, n * 50, 50, seed, ch, form=form, **sc)
        assert len(Es) == n
    Es, c1, stats, t = oracle.cont_double("wtm", A, dJ, rJ, 1.1, 60, 1, seed, ch, stepf=1.0, form=form, **sc)
    assert len(Es) == 60 and stats[0] > 0 and t > 0
    # the waiting-time chain's last sample precedes its last moves: replay a prefix that ends on a sample and compare energies
    Es_f, cf, st_f, _ = oracle.cont_sparse("wtm", A, cJ, 1.1, 60, 1, seed, ch, stepf=1.0, form=form)
    assert np.allclose(Es_f, Es, rtol=0, atol=1e-9)                    # same chain as the undiscretised couplings, up to rounding


def test_wtm_skn_tracked_energy(oracle):
    """wtmMC on GraphSKNormal(10) (runtests.jl:67 x :149): the oracle checks E_tracked == energy(X, C) itself; the time-averaged energy
    agrees with Metropolis' (loose, statistical)."""
    seed = 4711
    J = oracle.gen_sk_gauss(10, seed)
    ch = oracle.init_config(seed, 0, 10)
    Es, c1, moves, t = oracle.wtm_mc_skn(J, 1.0, 4000, 1.0, seed, ch)
    assert len(Es) == 4000 and moves > 0 and t > 0
    Em, _, _ = oracle.standard_mc_skn(J, 1.0, 400000, 10, seed, ch)[:3]
    assert abs(Es[500:].mean() - Em[5000:].mean()) < 0.25


# ---- extremal_opt with EOCacheCont (DeltaE.jl:557-635) on the Float64 graphs ---------------------------------------------------
@pytest.mark.parametrize("kind,form", [("rrg", "rrg"), ("ea", "ea"), ("ea2L2", "ea")])
def test_extremal_opt_cont_invariants(oracle, kind, form):
    seed = 23
    A = {"rrg": lambda: oracle.gen_rrg(40, 3, seed), "ea": lambda: oracle.gen_ea(4, 2), "ea2L2": lambda: oracle.gen_ea(2, 3)}[kind]()
    J = oracle.gen_couplings_gauss(A, seed)
    N = A.shape[0]
    ch = oracle.init_config(seed, 0, N)
    E0 = oracle.spf_energy(A, J, ch, form=form)
    # the oracle itself checks: ranking sorted, dEs == delta_energy, tracked E == energy(X, C) (runtests.jl:12-20)
    Es, ch1, Emin, Cmin, itmin = oracle.extremal_opt_cont(A, J, 1.3, 2000, 10, seed, ch, form=form)
    assert len(Es) == 200 and Emin <= min(E0, Es.min()) + 1e-12 and 0 <= itmin <= 2000
    assert abs(Emin - oracle.spf_energy(A, J, Cmin, form=form)) < 1e-9
    assert Emin < E0
    if N <= 16:
        def e_of(c):
            sg = 2 * np.array([(c >> i) & 1 for i in range(N)]) - 1
            return -0.5 * float((J * sg[:, None] * sg[A]).sum())
        assert Emin >= min(e_of(c) for c in range(2 ** N)) - 1e-9


def test_extremal_opt_cont_ties_are_shuffled(oracle):
    """Integer-valued Float64 couplings make long runs of equal dE: the ranking stays sorted (checked inside), the walk differs from
    replica to replica only through the draws, and with tau large the lowest-dE site is taken almost always — so the energy of a
    +-1.0 graph descends exactly like the discrete cache's walk does in distribution (here: it reaches a local minimum region)."""
    seed = 5
    A = oracle.gen_rrg(30, 3, seed)
    J = oracle.gen_couplings(A, seed).astype(np.float64)
    ch = oracle.init_config(seed, 0, 30)
    outs = [oracle.extremal_opt_cont(A, J, 1.5, 400, 1, seed, ch, replica=r) for r in range(4)]
    E0 = oracle.spf_energy(A, J, ch)
    for Es, c1, Emin, Cmin, itmin in outs:
        assert Es[0] == E0 and Emin < E0 and float(Emin).is_integer()
    assert len({tuple(o[0]) for o in outs}) > 1           # different replicas, different tie orders / draws


# ---- GraphQuant over binary GraphSK slices: GraphQSKT (src/QAliases.jl:34-43), the graph of scripts.jl:test_QIsing ---------------
def test_quant_sk_slices_tracked_energy_and_parts(oracle):
    """GraphQuant(10, 8, 0.5, 2.0, GraphSK, gen_J(10)) (runtests.jl:79) under standardMC and rrrMC (:141-143,153-159): the tracked
    energy equals energy(X, C) (:12-20; the rrrMC oracle also runs check_consistency), energy = E_QT + sum_k (n_k / sqrt(Nk)) / M
    with n_k the binary-SK integer energy (SK.jl:62-96)."""
    seed, Nk, M, beta, Gamma = 41, 10, 8, 2.0, 0.5
    Jb = oracle.gen_sk_binary(Nk, seed)
    fourK = oracle.quant_fourK(beta, Gamma, M)
    ch = oracle.init_config(seed, 0, Nk * M)
    E0, qt = oracle.quant_sk_energy(Jb, Nk, M, fourK, ch)
    # parts: slice k's bits against the stand-alone binary-SK energy
    bits = _bits(ch, Nk * M)
    acc = qt
    for k in range(M):
        sl = np.zeros(1, np.uint64)
        for i in range(Nk):
            sl[0] |= np.uint64(int(bits[k * Nk + i]) << i)
        acc += oracle.skb_energy(Jb, sl) / M
    assert E0 == acc
    Es, ch1, nacc = oracle.standard_mc_quant_sk(Jb, Nk, M, fourK, beta, 5000, 1, seed, ch)
    assert Es[0] == E0 and 0 < nacc < 5000
    for k in (2, 77, 1234, 5000):
        _, ck, _ = oracle.standard_mc_quant_sk(Jb, Nk, M, fourK, beta, k - 1, 1, seed, ch)
        assert abs(Es[k - 1] - oracle.quant_sk_energy(Jb, Nk, M, fourK, ck)[0]) < 1e-11
    for thr in (0.5, 0.0, 1.0):
        Er, cr, a, st = oracle.rrr_mc_quant_sk(Jb, Nk, M, fourK, beta, 4000, 1, seed, ch, staged_thr=thr)
        assert Er[0] == E0 and 0 < a < 4000
        for k in (3, 500, 4000):
            _, ck, _, _ = oracle.rrr_mc_quant_sk(Jb, Nk, M, fourK, beta, k - 1, 1, seed, ch, staged_thr=thr)
            assert abs(Er[k - 1] - oracle.quant_sk_energy(Jb, Nk, M, fourK, ck)[0]) < 1e-11
    Q, tm, ovs, e0, Esl, raw = oracle.quant_sk_observables(Jb, Nk, M, fourK, beta, Gamma, ch)
    assert abs(Q - (-Gamma * tm + sum(n / np.sqrt(Nk) / (Nk * M) for n in Esl))) < 1e-12 and len(ovs) == M // 2


def test_binary_sk_under_the_continuous_samplers(oracle):
    """GraphSK(10) (runtests.jl:66) under rrrMC / bklMC / wtmMC (:145-159): a SimpleGraph{Float64}, so DeltaECacheCont over
    delta_energy = lfields[i] / sqrt(N); the tracked energy equals energy(X, C) at the samples (:12-20)."""
    seed, N, beta = 77, 10, 2.0
    Jb = oracle.gen_sk_binary(N, seed)
    ch = oracle.init_config(seed, 0, N)
    E0 = oracle.skb_energy(Jb, ch)
    for thr in (0.8, 0.0, 1.0):
        Es, c1, acc, st = oracle.rrr_mc_skb(Jb, beta, 3000, 1, seed, ch, staged_thr=thr)
        assert Es[0] == E0 and 0 < acc < 3000
        for k in (5, 900, 3000):
            _, ck, _, _ = oracle.rrr_mc_skb(Jb, beta, k - 1, 1, seed, ch, staged_thr=thr)
            assert abs(Es[k - 1] - oracle.skb_energy(Jb, ck)) < 1e-11
    for n in (1, 40):
        Es, c1, moves, itd = oracle.bkl_mc_skb(Jb, beta, n * 50, 50, seed, ch)
        assert len(Es) == n and moves > 0
    Es, c1, moves, t = oracle.wtm_mc_skb(Jb, beta, 50, 1.0, seed, ch)          # checks E == energy(X, C) inside
    assert len(Es) == 50 and moves > 0 and t > 0
    # energies of the binary model are multiples of 2 / sqrt(N) apart
    q = (Es - Es[0]) * np.sqrt(N) / 2
    assert np.allclose(q, np.round(q), atol=1e-9)


@pytest.mark.parametrize("binary", [False, True])
def test_extremal_opt_on_sk_models(oracle, binary):
    """extremal_opt on GraphSK(10) / GraphSKNormal(10) (runtests.jl:66-67 x :161-164): EOCacheCont with every spin a neighbour."""
    seed, N = 55, 12
    J = oracle.gen_sk_binary(N, seed) if binary else oracle.gen_sk_gauss(N, seed)
    energy = (lambda c: oracle.skb_energy(J, c)) if binary else (lambda c: oracle.skn_energy(J, c))
    ch = oracle.init_config(seed, 0, N)
    E0 = energy(ch)
    Es, c1, Emin, Cmin, itmin = oracle.extremal_opt_sk(J, 1.3, 1500, 10, seed, ch, binary=binary)     # checks ranking + tracked E inside
    assert len(Es) == 150 and Emin <= min(E0, Es.min()) + 1e-12 and abs(Emin - energy(Cmin)) < 1e-9
    best = min(energy(np.array([c], np.uint64)) for c in range(2 ** N))
    assert best - 1e-9 <= Emin and Emin < E0
    assert Emin <= best + 1e-9                                       # 1500 tau-EO moves find the ground state of 12 spins


def test_quant_bkl_wtm_tracked_energy(oracle):
    """bklMC / wtmMC on GraphQuant(10, 8, 0.5, 2.0, GraphRRG, 10, 3) (runtests.jl:78 x :145-151): the continuous-energy caches over the
    whole DoubleGraph; E at a sample equals energy(X, C) of the configuration reached by the prefix run."""
    seed, Nk, M, beta, Gamma = 31, 10, 8, 2.0, 0.5
    A = oracle.gen_rrg(Nk, 3, seed)
    J = oracle.gen_couplings(A, seed)
    fourK = oracle.quant_fourK(beta, Gamma, M)
    ch = oracle.init_config(seed, 0, Nk * M)
    E0 = oracle.quant_energy(A, J, M, fourK, ch)[0]
    Es, c1, st, _ = oracle.cont_quant("bkl", A, J, M, fourK, beta, 3000, 1, seed, ch)
    assert len(Es) == 3000 and Es[0] == E0 and 0 < st[0] <= 3000
    for k in (2, 40, 777, 3000):          # a run of k iterations stops at its k-th sample, before the pending move: same chain, same state
        _, ck, _, _ = oracle.cont_quant("bkl", A, J, M, fourK, beta, k, 1, seed, ch)
        assert abs(Es[k - 1] - oracle.quant_energy(A, J, M, fourK, ck)[0]) < 1e-10
    Ew, cw, sw, t = oracle.cont_quant("wtm", A, J, M, fourK, beta, 60, 1, seed, ch, stepf=2.0)
    assert len(Ew) == 60 and sw[0] > 0 and t > 0
    for k in (1, 7, 60):
        _, ck, _, _ = oracle.cont_quant("wtm", A, J, M, fourK, beta, k, 1, seed, ch, stepf=2.0)
        assert abs(Ew[k - 1] - oracle.quant_energy(A, J, M, fourK, ck)[0]) < 1e-10
