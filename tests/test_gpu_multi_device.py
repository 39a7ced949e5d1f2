"""rrrmc_ctx_create_multi: ONE context over several devices — here two shards on device 0 (and devices 0, 1 when
the box has two).  Every call is the single-device call, every result is the single-device result (and the oracle's)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def devices_for_test(pkg):
    n = pkg.lib().rrrmc_device_count()
    return [[0, 0], [0, 0, 0]] + ([[0, 1]] if n >= 2 else [])


def test_sparse_pm1_multi_equals_single_and_oracle(pkg, oracle):
    seed, N, K, R, beta, iters, step = 4711, 512, 3, 200, 1.0, 8192, 1024          # 200 replicas: shards of 96 / 104 and 64 / 64 / 72
    X = pkg.GraphRRG(N, K, seed=seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed); eng.init_spins_random()
        C0 = eng.get_config().s.copy()
        E0 = eng.energy(); lf0 = eng.fields()
        Es, acc = eng.standard_mc(beta, iters, step)
        Es2, acc2 = eng.standard_mc(beta, iters // 2, step)                        # a second call continues the streams
        C1 = eng.get_config().s.copy()
    for devs in devices_for_test(pkg):
        with pkg.Engine(X, R, devices=devs) as m:
            m.seed(seed); m.init_spins_random()
            assert (m.get_config().s == C0).all() and (m.energy() == E0).all() and (m.fields() == lf0).all()
            mEs, macc = m.standard_mc(beta, iters, step)
            assert m.iterations_done() == iters
            mEs2, macc2 = m.standard_mc(beta, iters // 2, step)
            assert (mEs == Es).all() and (macc == acc).all() and (mEs2 == Es2).all() and (macc2 == acc2).all()
            assert (m.get_config().s == C1).all()
            tot, sw, nl = m.last_timing()
            assert sw > 0 and nl >= 1
            # set_spins with the caller's full array lands on the right shards
            Cfg = pkg.Config(N, R); Cfg.s[:] = C0
            m.set_config(Cfg)
            assert (m.energy() == E0).all()
    for r in (0, 95, 96, 199):
        ref = oracle.standard_mc_sparse(X.A, X.J.astype(np.int32), beta, iters, step, seed, C0[r], replica=r)
        assert (Es[r] == ref[0]).all() and acc[r] == ref[2]


def test_quant_rrr_multi_equals_single(pkg, oracle):
    seed, R, beta, iters, step = 99, 70, 2.0, 4000, 500
    X = pkg.GraphQuant(pkg.GraphRRG(64, 3, seed=seed), 4, 0.5, beta)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed); eng.init_spins_random()
        C0 = eng.get_config().s.copy()
        Es, acc, staged = eng.rrr_mc(beta, iters, step)
        pos, sizes = eng.rrr_cache()
        obs = eng.quant_observables()
        C1 = eng.get_config().s.copy()
    with pkg.Engine(X, R, devices=[0, 0]) as m:
        m.seed(seed); m.init_spins_random()
        mEs, macc, mst = m.rrr_mc(beta, iters, step)
        mpos, msizes = m.rrr_cache()
        mobs = m.quant_observables()
        assert (mEs == Es).all() and (macc == acc).all() and (mst == staged).all() and (m.get_config().s == C1).all()
        assert (mpos == pos).all() and (msizes == sizes).all()
        assert all((a == b).all() for a, b in zip(obs, mobs))
    for r in (0, 63, 64, 69):
        ref = oracle.rrr_mc_quant(X.X1.A, X.X1.J.astype(np.int32), X.M, X.fourK, beta, iters, step, seed, C0[r], replica=r)
        assert (Es[r] == ref[0]).all() and acc[r] == ref[2] and staged[r] == ref[3]


def test_sk_normal_and_snapshots_multi(pkg):
    seed, N, R = 5, 96, 40
    X = pkg.GraphSKNormal(N, seed=seed)
    outs = []
    for devs in (None, [0, 0]):
        with pkg.Engine(X, R, devices=devs) as eng:
            eng.seed(seed); eng.init_spins_random()
            eng.snapshot_reserve(2)
            eng.snapshot_store(0)
            Es, acc = eng.standard_mc(0.9, 3000, 300)
            eng.snapshot_store(1)
            q = eng.overlaps(np.array([0, 0, 1], np.int32), np.array([1, -1, -1], np.int32))
            outs.append((Es, acc, eng.energy(), eng.fields(), q, eng.snapshot_get(0).s, eng.get_config().s))
    for a, b in zip(*outs):
        assert (a == b).all()


def test_multi_context_errors(pkg):
    X = pkg.GraphRRG(64, 3, seed=1)
    with pytest.raises(pkg.RRRMCError) as e:
        pkg.Engine(X, 64, devices=[0, 99])
    assert e.value.code == 1 and "device" in str(e.value)
    with pkg.Engine(X, 40, devices=[0, 0, 0, 0]) as m:          # two 32-replica groups only: two of the four entries stay idle
        m.seed(1); m.init_spins_random()
        Es, acc = m.standard_mc(1.0, 512, 64)
        assert Es.shape == (40, 8)
    with pkg.Engine(X, 64, devices=[0, 0]) as m:
        with pytest.raises(pkg.RRRMCError) as e:                # an error of a child comes back with the shard it happened on
            m.standard_mc_async(1.0, -5, 1)
        assert "device 0" in str(e.value)


def test_float64_sparse_multi_equals_single_and_oracle(pkg, oracle):
    """GraphRRGNormal through one context over several shards: every shard runs its own teams of wavefronts (csrc/spf_team_kernel.hpp) over
    its own plan table; shards of 64 / 66 and 32 / 32 / 66 replicas, i.e. groups whose upper lanes are padding, in every shard."""
    seed, N, K, R, beta, iters, step = 977, 300, 3, 130, 1.1, 9000, 700
    X = pkg.GraphRRGNormal(N, K, seed=seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed); eng.init_spins_random()
        C0 = eng.get_config().s.copy()
        E0 = eng.energy(); lf0 = eng.fields()
        Es, acc = eng.standard_mc(beta, iters, step)
        Es2, acc2 = eng.standard_mc(beta, iters // 3, step)
        C1, lf1 = eng.get_config().s.copy(), eng.fields()
    for devs in devices_for_test(pkg):
        with pkg.Engine(X, R, devices=devs) as m:
            m.seed(seed); m.init_spins_random()
            assert (m.get_config().s == C0).all() and (m.energy() == E0).all() and (m.fields() == lf0).all()
            mEs, macc = m.standard_mc(beta, iters, step)
            mEs2, macc2 = m.standard_mc(beta, iters // 3, step)
            assert (mEs == Es).all() and (macc == acc).all() and (mEs2 == Es2).all() and (macc2 == acc2).all()
            assert (m.get_config().s == C1).all() and (m.fields() == lf1).all()
    for r in (0, 63, 64, 129):
        ref = oracle.standard_mc_spf(X.A, X.J, beta, iters, step, seed, C0[r], replica=r)
        assert (Es[r] == ref[0]).all() and acc[r] == ref[2]


@pytest.mark.parametrize("dense", ["sk", "skn"])
def test_quant_over_dense_slices_multi_equals_single(pkg, dense):
    """GraphQuant over binary GraphSK / GraphSKNormal slices (GraphQSKT / GraphQSKNormalT, src/QAliases.jl:34-46 — the graph of the reference's
    quantum experiment) through one context over several shards (rrrmc_ctx_create_multi with the RRRMC_MODEL_QUANT_SK / _SKN selectors):
    rrrMC, standardMC, energies and configurations equal the single-device context's, replica by replica."""
    seed, Nk, M, R, beta, Gamma = 606, 24, 6, 100, 1.5, 0.4
    X = (pkg.GraphQSKT if dense == "sk" else pkg.GraphQSKNormalT)(Nk, M, Gamma, beta, seed=seed)
    outs = []
    for devs in [None] + devices_for_test(pkg):
        with pkg.Engine(X, R, devices=devs) as eng:
            eng.seed(seed); eng.init_spins_random()
            E0 = eng.energy()
            Es, acc, staged = eng.rrr_mc(beta, 4000, 200)
            Es2, acc2 = eng.standard_mc(beta, 3000, 500)
            got = [E0, Es, acc, staged, Es2, acc2, eng.get_config().s.copy(), eng.energy()]
            if dense == "sk":
                got += list(eng.quant_observables())
            outs.append(got)
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert (np.asarray(a) == np.asarray(b)).all()
