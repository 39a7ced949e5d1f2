"""CPU-side checks of the product package: the C-ABI library loads and exports every declared symbol, the
host-only entry points (graph constructors) agree with the oracle, argument errors surface as status codes, and
the bit-sliced / level-scheduled algorithm of the kernels reproduces the sequential oracle (pure-Python model).
No compute call touches a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import emulate_sweep as EM

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(ROOT, "include", "rrrmc_hip.h")).read()
    declared = sorted(set(re.findall(r"RRRMC_API\s+[\w\s\*]+?\b(rrrmc_\w+)\s*\(", hdr)))
    assert declared == sorted(pkg.SYMBOLS)
    L = pkg.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.rrrmc_version() >= 100


def test_no_device_is_a_loud_error_not_a_fallback(pkg):
    if pkg.lib().rrrmc_device_count() > 0:
        pytest.skip("a HIP device is present")
    X = pkg.GraphRRG(16, 3, seed=1)
    with pytest.raises(pkg.RRRMCError) as e:
        pkg.Engine(X, 4)
    assert e.value.code == 4 and "no CPU path" in str(e.value)
    with pytest.raises(pkg.RRRMCError):
        pkg.standardMC(X, 1.0, 10, quiet=True)


def test_graph_constructors_match_oracle(pkg, oracle):
    for N, K, seed in [(10, 3, 5), (128, 3, 0x5EED), (60, 4, 9)]:
        X = pkg.GraphRRG(N, K, seed=seed)
        A = oracle.gen_rrg(N, K, seed)
        assert (X.A == A).all() and (X.J == oracle.gen_couplings(A, seed)).all()
    for L_, D in [(2, 3), (3, 2), (4, 3)]:
        X = pkg.GraphEA(L_, D, seed=3)
        A = oracle.gen_ea(L_, D)
        assert (X.A == A).all() and (X.J == oracle.gen_couplings(A, 3)).all()
    assert pkg.all_delta_e(pkg.GraphRRG(10, 3, seed=1)) == oracle.all_delta_e_pm1(3) == (2, 6)
    assert pkg.all_delta_e(pkg.GraphEA(2, 3, seed=1)) == (0, 4, 8, 12)
    assert list(pkg.neighbors(pkg.GraphEA(2, 3, seed=1), 0)) == [1, 2, 4]          # uA: EA.jl:158


def test_argument_errors(pkg):
    with pytest.raises(pkg.RRRMCError) as e:
        pkg.GraphRRG(5, 3, seed=1)                       # "N * K must be even", RRG.jl:28
    assert e.value.code == 1 and "even" in str(e.value)
    with pytest.raises(pkg.RRRMCError):
        pkg.GraphEA(1, 3)                                 # "L must be >= 2", EA.jl:25
    with pytest.raises(ValueError):
        pkg.GraphRRG.from_AJ(np.zeros((4, 3), np.int32), np.zeros((4, 3), np.int8))   # J incompatible with levels
    ctx = C.c_void_p()
    L = pkg.lib()
    assert L.rrrmc_ctx_create(C.byref(ctx), 1, 0, 3, 1, 0, 0) == 1          # N < 1
    assert L.rrrmc_ctx_create(C.byref(ctx), 99, 16, 3, 1, 0, 0) == 3        # unknown model
    assert L.rrrmc_ctx_create(C.byref(ctx), 1, 16, 9, 1, 0, 0) == 3         # K out of the kernels' range
    assert L.rrrmc_ctx_create(C.byref(ctx), 1, 16, 3, 1, 0, 5) == 1         # replica0 not a multiple of 32
    assert b"replica0" in L.rrrmc_last_error(None)


def test_config_bit_layout(pkg):
    rng = np.random.default_rng(0)
    bits = rng.integers(0, 2, (5, 131))
    Cfg = pkg.Config.from_bits(bits)
    assert Cfg.s.shape == (5, 3) and (Cfg.bits() == bits).all()
    assert (Cfg.s[:, 2] >> np.uint64(3)).max() == 0      # unused high bits of the last chunk stay zero
    assert Cfg.copy() == Cfg and len(Cfg) == 131


@pytest.mark.parametrize("N,K,beta,iters,step,C", [(16, 3, 1.0, 300, 7, 64), (10, 3, 2.0, 150, 1, 832), (27, 6, 1.0, 200, 10, 64),
                                                   (8, 6, 2.0, 300, 25, 832), (9, 4, 0.5, 200, 3, 832)])
def test_bit_sliced_level_schedule_equals_sequential_chain(oracle, N, K, beta, iters, step, C):
    """The algorithm behind sweep_kernel (tests/emulate_sweep.py mirrors plan_kernel / produce / consume): bit-sliced
    replicas + dependency-level reordering + MSB-first bit-plane acceptance == the oracle's sequential chain."""
    seed = 12345 + N
    if K == 3:
        A, form = oracle.gen_rrg(N, K, seed), "rrg"
    elif N == 27:
        A, form = oracle.gen_ea(3, 3), "ea"
    elif N == 8:
        A, form = oracle.gen_ea(2, 3), "ea"
    else:
        A, form = oracle.gen_ea(3, 2), "ea"
    J = oracle.gen_couplings(A, seed)
    R = 32
    ch = oracle.init_configs(seed, 0, R, N)
    Es, ch2, acc = oracle.standard_mc_sparse_batch(A, J, beta, iters, step, seed, ch, form=form)

    def slice_bits(chunks):
        return [sum(((int(chunks[r, x >> 6]) >> (x & 63)) & 1) << r for r in range(R)) for x in range(N)]

    Es2, sp2, acc2 = EM.sweep(oracle.philox, oracle.site_of, oracle.threshold, A, J, beta, iters, step, seed, slice_bits(ch), C=C)
    assert sp2 == slice_bits(ch2)
    assert (np.array(Es2).T.reshape(Es.shape) == Es).all() if len(Es2) else Es.shape[1] == 0
    assert (np.array(acc2) == acc).all()
