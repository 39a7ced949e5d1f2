"""CPU tests of the observables restated in the oracle (scripts/scripts.jl:283-295, 368-405; src/graphs/QT.jl:113-122, 213-268)
against independent numpy formulas, and of the host-side log / BitMatrix helpers.  No GPU."""
import math

import numpy as np


def _bits(ch, N):
    x = np.arange(N)
    return ((ch[x >> 6] >> (x & 63).astype(np.uint64)) & np.uint64(1)).astype(np.int64)


def test_pm1dot_matches_definition(oracle):
    rng = np.random.default_rng(1)
    for N in (1, 10, 63, 64, 65, 200):
        a = oracle.init_config(11, 0, N)
        b = oracle.init_config(11, 1, N)
        sa, sb = 2 * _bits(a, N) - 1, 2 * _bits(b, N) - 1
        assert oracle.pm1dot(a, b, N) == int(sa @ sb)                 # "same as (2a-1) . (2b-1)" scripts.jl:281
        assert oracle.pm1dot(a, a, N) == N
    del rng


def test_q2_window_matches_loops(oracle):
    N, T = 70, 9
    Cs = np.stack([oracle.init_config(5, t, N) for t in range(T)])
    S = np.stack([2 * _bits(c, N) - 1 for c in Cs])
    for i, j in [(0, T), (2, 7), (3, 5)]:
        q2 = [((S[i1] @ S[j1]) / N) ** 2 for i1 in range(i, j - 1) for j1 in range(i + 1, j)]
        m, s = oracle.q2_window(Cs, N, i, j)
        assert math.isclose(m, np.mean(q2), rel_tol=1e-13)
        assert math.isclose(s, math.sqrt(max(0.0, np.mean(np.square(q2)) - np.mean(q2) ** 2)), rel_tol=1e-9, abs_tol=1e-12)
    assert all(math.isnan(v) for v in oracle.q2_window(Cs, N, 3, 4))   # empty window: n == 0


def test_quant_observables_match_definitions(oracle):
    for Nk, M, seed in [(10, 8, 3), (12, 5, 4), (34, 4, 5)]:
        A = oracle.gen_rrg(Nk, 3, seed)
        J = oracle.gen_couplings(A, seed)
        beta, Gamma = 2.0, 0.5
        fourK = oracle.quant_fourK(beta, Gamma, M)
        N = Nk * M
        ch = oracle.init_config(seed, 0, N)
        Q, tm, ovs, e0, Es, raw = oracle.quant_observables(A, J, M, fourK, beta, Gamma, ch)
        s = (2 * _bits(ch, N) - 1).reshape(M, Nk)
        # transverse_mag: QT.jl:113-122 with energy0 = -sum_k s_k . s_{k-1}
        e0_ref = -int(sum(s[k] @ s[k - 1] for k in range(M)))
        assert e0 == e0_ref
        x = beta * fourK / 2
        assert math.isclose(tm, math.cosh(x) + e0_ref / N * math.sinh(x), rel_tol=1e-14)
        # slice energies and Qenergy: QT.jl:253-268
        Es_ref = [-sum(int(J[i, k]) * s[m, i] * s[m, A[i, k]] for i in range(Nk) for k in range(3)) // 2 for m in range(M)]
        assert list(Es) == Es_ref
        assert math.isclose(Q, -Gamma * tm + sum(Es_ref) / N, rel_tol=1e-13)
        # overlaps: the commented-out per-site loop of QT.jl:235-243 is the definition
        ov = np.zeros(M // 2)
        for k1 in range(M - 1):
            for k2 in range(k1 + 1, M):
                ov[min(k2 - k1, M + k1 - k2) - 1] += s[k1] @ s[k2]
        assert list(raw) == list(ov.astype(np.int64))
        for d in range(1, (M - 1) // 2 + 1):
            ov[d - 1] /= M * Nk
        if M % 2 == 0:
            ov[M // 2 - 1] /= M * Nk / 2
        assert np.allclose(ovs, ov, rtol=1e-15)
        # the QT part of energy(X, C) is energy0 * fourK / 4 (QT.jl:84)
        assert oracle.quant_energy(A, J, M, fourK, ch)[1] == e0 * fourK / 4


def test_bitmatrix_chunks_and_ranges(pkg):
    rng = np.random.default_rng(0)
    cols = rng.integers(0, 2, (10, 13)).astype(np.uint8)           # N = 10 spins, 13 samples
    ch = pkg.bitmatrix_chunks(cols)
    assert ch.dtype == np.uint64 and ch.size == (130 + 63) // 64
    for j in range(13):
        for i in range(10):
            lin = i + 10 * j                                           # column-major linear index of a Julia BitMatrix
            assert (int(ch[lin >> 6]) >> (lin & 63)) & 1 == cols[i, j]
    # LogRange(1.0, 20.0, 1.0, 1.5): 1, 2, 3.5, 5.75, 9.125, 14.1875 (scripts.jl:351-358)
    assert list(pkg.log_range(1.0, 20.0)) == [1.0, 2.0, 3.5, 5.75, 9.125, 14.1875]
    ts = [0.1, 0.5, 1.1, 1.9, 2.0, 3.0, 4.5]
    assert pkg.get_ts_range(ts, 1.0) == (2, 4)
    assert pkg.get_ts_range(ts, 4.0) == (6, None)
    assert pkg.get_ts_range(ts, 5.0) == (None, None)


def test_parsets_reads_reference_log_format(pkg, tmp_path):
    f = tmp_path / "output_met.txt"
    f.write_text("#mctime acc E clocktime\n1000 12 -150 0.01\n2000 30 -162 0.025\n")
    assert pkg.parsets(str(f)) == [0.01, 0.025]
