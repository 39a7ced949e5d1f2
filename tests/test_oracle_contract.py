"""CPU tests of the oracle: the random-stream contract and its building blocks (no GPU)."""
import math

import numpy as np
import pytest


# Random123 known-answer vectors for Philox4x32-10 (philox.h kat_vectors)
KATS = [
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0], [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


@pytest.mark.parametrize("ctr,key,out", KATS)
def test_philox_kat(oracle, ctr, key, out):
    assert [int(x) for x in oracle.philox(ctr, key)] == out


def test_threshold_is_exact_ceiling(oracle):
    # accept iff u * 2^-64 < p  <=>  u < ceil(p * 2^64): check against exact rational arithmetic
    from fractions import Fraction
    for p in [math.exp(-2.0), math.exp(-6.0), 0.5, 2.0 ** -70, 1.0 - 2.0 ** -53, 1e-300, 0.1, 3e-20]:
        T, always = oracle.threshold(p)
        assert not always
        exact = -((-Fraction(p) * 2 ** 64) // 1)          # ceil
        assert T == max(1, int(exact)) if p < 2.0 ** -64 else T == int(exact)
    assert oracle.threshold(1.0) == (2 ** 64 - 1, True)
    assert oracle.threshold(1.5)[1] is True
    assert oracle.threshold(0.0) == (0, False)


def test_lazy_compare_equals_full_compare(oracle):
    rng = np.random.default_rng(5)
    seed = 0xABCDEF
    for _ in range(300):
        g = int(rng.integers(1, 2 ** 40))
        r = int(rng.integers(0, 5000))
        u = oracle.accept_uniform(seed, g, r)
        for T in (int(rng.integers(0, 2 ** 63)) * 2 + 1, u, u + 1 if u < 2 ** 64 - 1 else u, max(u - 1, 0), 1, 2 ** 64 - 1):
            assert oracle.accept_less(seed, g, r, T) == (u < T)


def test_accept_uniform_is_bit_transposed_philox(oracle):
    # bit j (MSB first) of replica r's uniform = bit (r & 31) of word (j & 3) of ctr (g_lo, g_hi, r >> 5, 2 | (j >> 2) << 8)
    seed, g, r = 0x1234567811223344, (7 << 32) | 99, 77
    key = [seed & 0xffffffff, seed >> 32]
    u = 0
    for j in range(64):
        w = oracle.philox([g & 0xffffffff, g >> 32, r >> 5, 2 | ((j >> 2) << 8)], key)
        u = (u << 1) | ((int(w[j & 3]) >> (r & 31)) & 1)
    assert u == oracle.accept_uniform(seed, g, r)


def test_uniforms_look_uniform_and_independent(oracle):
    seed = 42
    us = np.array([[oracle.accept_uniform(seed, g, r) for r in range(64)] for g in range(1, 201)], dtype=np.float64) / 2.0 ** 64
    assert abs(us.mean() - 0.5) < 0.01
    assert abs(us.var() - 1 / 12) < 0.005
    c = np.corrcoef(us[:, 3], us[:, 4])[0, 1]          # two replicas of the same 32-group share Philox words, not bits
    assert abs(c) < 0.2


def test_site_stream(oracle):
    seed, N = 99, 37
    s = np.array([oracle.site_of(seed, g, N) for g in range(1, 20001)])
    assert s.min() == 0 and s.max() == N - 1
    counts = np.bincount(s, minlength=N)
    assert abs(counts / len(s) - 1 / N).max() < 0.01
    # word (2h, 2h+1) of ctr (g >> 1, 0, 0, TAG_SITE = 1)
    g = 12345
    w = oracle.philox([g >> 1, 0, 0, 1], [seed, 0])
    u64 = (int(w[2]) << 32 | int(w[3])) if g & 1 else (int(w[0]) << 32 | int(w[1]))
    assert oracle.site_of(seed, g, N) == (u64 * N) >> 64


def test_init_config_layout(oracle):
    seed, N = 7, 131
    ch = oracle.init_config(seed, 40, N)
    assert ch.shape == ((N + 63) // 64,)
    assert int(ch[-1]) >> (N % 64) == 0                 # BitVector invariant: unused bits are zero
    x = 100
    w = oracle.philox([x >> 2, 0, 40 >> 5, 3], [seed, 0])
    assert (int(ch[x >> 6]) >> (x & 63)) & 1 == (int(w[x & 3]) >> (40 & 31)) & 1
