"""Debug mode (rrrmc_set_debug_checks): the reference's latent consistency checks — its commented-out asserts in update_cache!
(src/graphs/RRG.jl:229-231, src/graphs/SK.jl:268-273) and its test suite's tracked-E hook (test/runtests.jl:12-20) — as a switch of the
library: after every standardMC call energy(X, C) is recomputed on the device and compared with what the sampler tracked."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("make,R", [(lambda p: p.GraphRRG(512, 3, seed=3), 96), (lambda p: p.GraphEA(8, 3, seed=3), 40),
                                    (lambda p: p.GraphSKNormal(200, seed=3), 24), (lambda p: p.GraphSKNormal(1024, seed=3), 16),
                                    (lambda p: p.GraphRRGNormal(300, 3, seed=3), 130), (lambda p: p.GraphEANormal(5, 3, seed=3), 70)])
def test_debug_checks_pass_and_change_nothing(pkg, make, R):
    X = make(pkg)
    outs = []
    for dbg in (False, True):
        with pkg.Engine(X, R) as eng:
            eng.seed(11); eng.init_spins_random()
            if dbg:
                eng.set_debug_checks(True)
            Es, acc = eng.standard_mc(1.0, 5000, 250)
            Es2, acc2 = eng.standard_mc(0.5, 3000, 100)               # second call: continues the streams, checked again
            outs.append((Es, acc, Es2, acc2, eng.get_config().s, eng.energy()))
    for a, b in zip(*outs):
        assert (a == b).all()


def test_debug_checks_catch_an_inconsistent_state(pkg, monkeypatch):
    """the failing branch: RRRMC_DEBUG_INJECT=1 (a fault injection that exists for this test) shifts one compared value inside the check"""
    for X, R in ((pkg.GraphRRG(256, 3, seed=5), 64), (pkg.GraphSKNormal(64, seed=5), 8), (pkg.GraphRRGNormal(128, 3, seed=5), 70)):
        with pkg.Engine(X, R) as eng:
            eng.seed(5); eng.init_spins_random()
            eng.set_debug_checks(True)
            eng.standard_mc(1.0, 2000, 100)                           # consistent: passes
            monkeypatch.setenv("RRRMC_DEBUG_INJECT", "1")
            with pytest.raises(pkg.RRRMCError) as e:
                eng.standard_mc(1.0, 1000, 100)
            assert e.value.code == 2 and "debug check failed" in str(e.value) and "replica 0" in str(e.value)
            monkeypatch.delenv("RRRMC_DEBUG_INJECT")
            eng.standard_mc(1.0, 1000, 100)                           # the flag is cleared: the context goes on
    with pytest.raises(pkg.RRRMCError) as e:                          # unsupported models say so instead of silently not checking
        with pkg.Engine(pkg.GraphSK(32, seed=1), 8) as e2:
            e2.set_debug_checks(True)
    assert e.value.code == 3
