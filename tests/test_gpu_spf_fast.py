"""GPU parity of the opt-in FAST standardMC on GraphRRGNormal / GraphEANormal (bit-sliced replicas, per-site threshold tables,
spf_fast_kernels.hpp) against the oracle's restatement of that mode (orc_standard_mc_spf_fast): configurations and accepted counts
identical, energies (re-evaluated from the spins at every sample on the device, tracked per move by the oracle) within 1e-9 relative
— the north star asks 1e-6 for Float64 models."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make(pkg, kind, seed):
    if kind[0] == "rrg":
        return pkg.GraphRRGNormal(kind[1], kind[2], seed=seed), "rrg"
    return pkg.GraphEANormal(kind[1], kind[2], seed=seed), "ea"


def close(a, b, scale):
    return np.allclose(a, b, rtol=1e-9, atol=1e-9 * scale)


@pytest.mark.parametrize("kind,R,beta,iters,step", [
    (("rrg", 64, 3), 32, 1.0, 6000, 100),
    (("rrg", 200, 3), 70, 0.7, 20000, 333),          # R not a multiple of 64: padded lanes, two groups of the second word half-used
    (("rrg", 1000, 3), 64, 2.0, 50000, 1000),
    (("rrg", 128, 4), 40, 1.0, 12000, 128),          # K = 4: eight thresholds per site
    (("ea", 8, 2), 96, 0.9, 15000, 64),              # GraphEANormal(8, 2): K = 4
    (("ea", 2, 2), 33, 1.0, 3000, 7),                # L = 2: every neighbour twice
    (("rrg", 30, 2), 32, 1.2, 4000, 50),             # K = 2
    (("rrg", 4096, 3), 64, 1.0, 1 << 16, 1 << 12),   # the benchmark geometry
    (("rrg", 96, 3), 32, -0.4, 3000, 100),           # beta < 0 (the reference does not forbid it): the tested side flips
])
def test_fast_mode_matches_its_oracle(pkg, oracle, kind, R, beta, iters, step):
    seed = 5150 + kind[1] + kind[2]
    X, form = make(pkg, kind, seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc = eng.standard_mc_fast(beta, iters, step)
        C1 = eng.get_config()
        E1 = eng.energy()
        Es2, acc2 = eng.standard_mc_fast(beta, 1000, 10)          # continues the streams
        C2 = eng.get_config()
    assert Es.shape == (R, iters // step)
    scale = float(X.N)
    for r in sorted(set([0, 1, R // 2, R - 1])):
        ref = oracle.standard_mc_spf_fast(X.A, X.J, beta, iters, step, seed, C0.s[r], replica=r, form=form)
        assert (C1.s[r] == ref[1]).all() and acc[r] == ref[2]
        assert close(Es[r], ref[0], scale)
        assert abs(E1[r] - oracle.spf_energy(X.A, X.J, C1.s[r], form=form)) <= 1e-9 * scale
        ref2 = oracle.standard_mc_spf_fast(X.A, X.J, beta, 1000, 10, seed, ref[1], it0=iters, replica=r, form=form)
        assert (C2.s[r] == ref2[1]).all() and acc2[r] == ref2[2] and close(Es2[r], ref2[0], scale)


def test_fast_then_exact_kernel_share_the_context(pkg, oracle):
    """After a fast call the default (bit-exact) kernel starts, like a fresh reference call, from energy(X, C) and a rebuilt cache."""
    seed, N, R, beta = 808, 300, 64, 1.0
    X = pkg.GraphRRGNormal(N, 3, seed=seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        eng.standard_mc_fast(beta, 5000, 500)
        Cm = eng.get_config()
        Es, acc = eng.standard_mc(beta, 4000, 200)
        C1 = eng.get_config()
    for r in (0, 33, 63):
        mid = oracle.standard_mc_spf_fast(X.A, X.J, beta, 5000, 500, seed, C0.s[r], replica=r)
        assert (Cm.s[r] == mid[1]).all()
        ref = oracle.standard_mc_spf(X.A, X.J, beta, 4000, 200, seed, mid[1], it0=5000, replica=r)
        assert (Es[r] == ref[0]).all() and (C1.s[r] == ref[1]).all() and acc[r] == ref[2]


def test_fast_mode_limits_are_reported(pkg):
    X = pkg.GraphEANormal(4, 3, seed=1)              # K = 6
    with pkg.Engine(X, 32) as eng:
        eng.seed(1)
        eng.init_spins_random()
        with pytest.raises(pkg.RRRMCError) as e:
            eng.standard_mc_fast_async(1.0, 100, 10)
        assert e.value.code == 3 and "K <= 4" in str(e.value)
    Y = pkg.GraphRRG(64, 3, seed=1)
    with pkg.Engine(Y, 32) as eng:
        eng.seed(1)
        eng.init_spins_random()
        with pytest.raises(pkg.RRRMCError) as e:
            eng.standard_mc_fast_async(1.0, 100, 10)
        assert e.value.code == 3


def test_fast_mode_full_width_subset(pkg, oracle):
    """GraphRRGNormal(4096, 3) at 8192 replicas (256 workgroups): the first, a middle and the last replica group against the oracle,
    and a sane acceptance rate over all replicas."""
    seed, N, R, beta, iters, step = 0x5EED, 4096, 8192, 1.0, 1 << 15, 1 << 12
    X = pkg.GraphRRGNormal(N, 3, seed=seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc = eng.standard_mc_fast(beta, iters, step)
        C1 = eng.get_config()
    for r in (0, 31, 100 * 32 + 7, 8191):
        ref = oracle.standard_mc_spf_fast(X.A, X.J, beta, iters, step, seed, C0.s[r], replica=r)
        assert (C1.s[r] == ref[1]).all() and acc[r] == ref[2] and close(Es[r], ref[0], float(N))
    a = acc / iters
    assert 0.15 < a.mean() < 0.45 and a.std() < 0.02
