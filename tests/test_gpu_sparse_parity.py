"""GPU parity: the HIP sweep path (through the C ABI) against the CPU oracle on identical seeds.
Bit-exact for +-J models: energies at every sample, final spins (BitVector chunks), accepted counts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _graph(pkg, kind, seed):
    if kind[0] == "rrg":
        return pkg.GraphRRG(kind[1], kind[2], seed=seed)
    if kind[0] == "circ":
        # K-regular circulant graph (the pairing model practically never yields a simple graph for K = 7):
        # x ~ x +- 1..K//2 and, for odd K, x + N/2; couplings from the library's gen_J restatement
        N, K = kind[1], kind[2]
        offs = [d for j in range(1, K // 2 + 1) for d in (j, -j)] + ([N // 2] if K % 2 else [])
        A = np.sort(np.array([[(x + d) % N for d in offs] for x in range(N)], np.int32), axis=1)
        J = np.zeros((N, K), np.int8)
        from rrrmc_jl_amd._lib import check, lib
        check(lib().rrrmc_gen_couplings_pm1(N, K, A, seed, J))
        return pkg.GraphRRG.from_AJ(A, J)
    return pkg.GraphEA(kind[1], kind[2], seed=seed)


CASES = [
    # kind,            R,   beta, iters, step
    (("rrg", 10, 3),   32,  2.0,  10000, 100),    # test/runtests.jl:36,125-130 (GraphRRG(10,3), beta=2, 10^4 iters, step 100)
    (("ea", 2, 3),     32,  2.0,  10000, 100),    # runtests.jl:46 GraphEA(2,3): L=2 => doubled neighbours
    (("ea", 3, 2),     32,  2.0,  10000, 100),    # runtests.jl:56 GraphEA(3,2)
    (("rrg", 128, 3),  1,   1.0,  20000, 1000),   # BASELINE config 1 shape (1 chain)
    (("rrg", 128, 3),  40,  1.0,  5000,  1),      # step = 1 (reference default), R not a multiple of 32
    (("rrg", 256, 3),  96,  0.5,  30000, 4096),
    (("rrg", 1024, 3), 64,  1.0,  50000, 1024),
    (("rrg", 4096, 3), 64,  1.0,  40000, 4096),   # BASELINE config 2 graph, few replicas
    (("ea", 8, 3),     64,  1.0,  30000, 512),    # K = 6
    (("rrg", 64, 5),   32,  0.7,  8000,  64),     # odd K > 3
    (("rrg", 50, 4),   32,  1.3,  8000,  77),     # even K, step not dividing anything
    (("circ", 100, 7), 32,  0.4,  5000,  500),    # K = 7 (4 classes)
    (("rrg", 64, 3),   32,  0.0,  3000,  100),    # beta = 0: everything accepted
    (("rrg", 64, 3),   32,  50.0, 3000,  100),    # very low temperature: thresholds of ~2^-144 underflow the 64-bit fraction
    (("rrg", 5400, 3), 32,  1.0,  40000, 5000),   # about the largest N whose neighbour table still fits LDS next to a full chunk
    (("rrg", 6000, 3), 32,  1.0,  40000, 5000),   # WIDE build by choice: word offsets, neighbour table in HBM/L2
    (("rrg", 8192, 3), 32,  1.0,  60000, 8192),
    (("rrg", 8200, 3), 32,  1.0,  60000, 8192),   # WIDE build by necessity (byte offsets no longer fit 16 bits)
    (("rrg", 10000, 3), 40, 2.0,  80000, 10000),  # the size of the reference's scripts (scripts/scripts.jl:23 N = 10_000)
    (("rrg", 12000, 4), 32, 1.0,  50000, 5000),   # even K, shorter chunks (LDS)
    (("ea", 24, 3),    32,  1.0,  60000, 6000),   # EA 24^3 = 13824 spins, K = 6
    (("rrg", 16384, 3), 32, 1.0,  70000, 16384),  # one word per site (MODE 3): half the state, chunks three times as long
    (("rrg", 16500, 4), 32, 1.0,  50000, 7000),   # near the end of the two-copy layout's range (MODE 3 by choice)
]


@pytest.mark.parametrize("kind,R,beta,iters,step", CASES)
def test_standard_mc_bit_exact(pkg, oracle, kind, R, beta, iters, step):
    seed = 0xC0FFEE + 7 * kind[1] + R
    X = _graph(pkg, kind, seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        E0 = eng.energy()
        Es, acc = eng.standard_mc(beta, iters, step)
        C1 = eng.get_config()
        E1 = eng.energy()
        lf = eng.fields()
    A, J = X.A, X.J.astype(np.int32)
    assert (C0.s == oracle.init_configs(seed, 0, R, X.N)).all()
    Es_ref, ch_ref, acc_ref = oracle.standard_mc_sparse_batch(A, J, beta, iters, step, seed, C0.s, form="ea" if kind[0] == "ea" else "rrg")
    assert Es.shape == Es_ref.shape
    assert (Es == Es_ref).all()
    assert (C1.s == ch_ref).all()
    assert (acc == acc_ref).all()
    for r in range(min(R, 4)):
        e0 = oracle.sparse_energy(A, J, C0.s[r])
        e1, lf_ref = oracle.sparse_energy(A, J, C1.s[r], want_fields=True)
        assert E0[r] == e0 and E1[r] == e1
        assert (lf[r] == lf_ref).all()


def test_resume_continues_the_streams(pkg, oracle):
    """Two calls without reseeding == one long call (reference: seed <= 0 keeps the RNG going, RRRMC.jl:89)."""
    seed, N, R, beta = 99, 512, 64, 1.0
    X = pkg.GraphRRG(N, 3, seed=seed)
    A, J = X.A, X.J.astype(np.int32)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es1, a1 = eng.standard_mc(beta, 7000, 500)
        assert eng.iterations_done() == 7000
        Es2, a2 = eng.standard_mc(beta, 9001, 300)
        C2 = eng.get_config()
    r1 = oracle.standard_mc_sparse_batch(A, J, beta, 7000, 500, seed, C0.s)
    r2 = oracle.standard_mc_sparse_batch(A, J, beta, 9001, 300, seed, r1[1], it0=7000)
    assert (Es1 == r1[0]).all() and (a1 == r1[2]).all()
    assert (Es2 == r2[0]).all() and (a2 == r2[2]).all()
    assert (C2.s == r2[1]).all()


def test_sharding_invariance(pkg, oracle):
    """A shard with replica0 = 64 reproduces replicas 64..127 of the unsharded job."""
    seed, N, beta, iters, step = 4242, 256, 1.0, 6000, 250
    X = pkg.GraphRRG(N, 3, seed=seed)
    with pkg.Engine(X, 128) as eng:
        eng.seed(seed); eng.init_spins_random()
        Es_all, acc_all = eng.standard_mc(beta, iters, step)
        C_all = eng.get_config()
    with pkg.Engine(X, 64, replica0=64) as eng:
        eng.seed(seed); eng.init_spins_random()
        Es_sh, acc_sh = eng.standard_mc(beta, iters, step)
        C_sh = eng.get_config()
    assert (Es_sh == Es_all[64:]).all() and (acc_sh == acc_all[64:]).all() and (C_sh.s == C_all.s[64:]).all()


def test_set_spins_roundtrip_and_c0(pkg, oracle):
    seed, N, R = 5, 200, 37
    X = pkg.GraphRRG(N, 3, seed=seed)
    rng = np.random.default_rng(3)
    C0 = pkg.Config.from_bits(rng.integers(0, 2, (R, N)))
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.set_config(C0)
        assert eng.get_config() == C0
        Es, acc = eng.standard_mc(1.2, 4000, 40)
        C1 = eng.get_config()
    ref = oracle.standard_mc_sparse_batch(X.A, X.J.astype(np.int32), 1.2, 4000, 40, seed, C0.s)
    assert (Es == ref[0]).all() and (C1.s == ref[1]).all() and (acc == ref[2]).all()


def test_standardMC_front_end_and_hook(pkg, oracle):
    """standardMC(X, beta, iters; step, seed, hook): the reference's own check — tracked E == energy(X, C) at
    every sample (test/runtests.jl:12-20) — plus equality with the hook-free run."""
    seed = 8426732438942 % (2 ** 63)
    X = pkg.GraphRRG(10, 3, seed=seed)
    seen = []

    def hook(it, X_, C, accepted, E):
        for r in range(C.R):
            assert E[r] == oracle.sparse_energy(X_.A, X_.J.astype(np.int32), C.s[r])
        seen.append((it, E.copy()))
        return True

    Es_h, C_h = pkg.standardMC(X, 2.0, 2000, step=100, seed=seed, hook=hook, quiet=True, replicas=32)
    Es, C = pkg.standardMC(X, 2.0, 2000, step=100, seed=seed, quiet=True, replicas=32)
    assert [it for it, _ in seen] == list(range(100, 2001, 100))
    assert (Es_h == Es).all() and C_h == C


@pytest.mark.parametrize("kind,R,beta,iters,step", [
    (("rrg", 300, 3), 40, 1.0, 20000, 300),         # forced at a small size: several levels per chunk, ragged batches
    (("ea", 8, 3), 33, 1.5, 20000, 512),            # K = 6, double-digit levels
    (("rrg", 9000, 5), 32, 0.8, 40000, 9000),       # K = 5: three thresholds
])
def test_single_copy_layout_forced(pkg, oracle, monkeypatch, kind, R, beta, iters, step):
    """MODE 3 of the sweep kernel (one LDS word per site, the coupling's sign in bit 15 of the neighbour index), forced where the
    context would not choose it: same chain as the oracle and as the default layout."""
    seed = 0xBEEF + kind[1]
    X = _graph(pkg, kind, seed)
    out = []
    for single in ("1", "0"):
        monkeypatch.setenv("RRRMC_FORCE_SINGLE", single)
        with pkg.Engine(X, R) as eng:
            eng.seed(seed)
            eng.init_spins_random()
            C0 = eng.get_config()
            a = eng.standard_mc(beta, iters, step)
            b = eng.standard_mc(beta, iters // 2, step)
            out.append((a[0], a[1], b[0], b[1], eng.get_config().s))
    for u, v in zip(*out):
        assert (u == v).all()
    ref = oracle.standard_mc_sparse_batch(X.A, X.J.astype(np.int32), beta, iters, step, seed, C0.s, form="ea" if kind[0] == "ea" else "rrg")
    assert (out[0][0] == ref[0]).all() and (out[0][1] == ref[2]).all()


def test_graphs_up_to_32767_sites_stay_on_the_lds_kernel(pkg, oracle):
    """With one word per site the LDS-resident kernel reaches N = 32 767 (if the planner's buffers fit too); check a size in the new
    range against the oracle (a few replicas: the oracle is sequential)."""
    seed, N = 24242, 24000
    X = pkg.GraphRRG(N, 3, seed=seed)
    with pkg.Engine(X, 32) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc = eng.standard_mc(1.0, 3 * N, N)
        C1 = eng.get_config()
    for r in (0, 31):
        ref = oracle.standard_mc_sparse(X.A, X.J.astype(np.int32), 1.0, 3 * N, N, seed, C0.s[r], replica=r)
        assert (Es[r] == ref[0]).all() and (C1.s[r] == ref[1]).all() and acc[r] == ref[2]


def _random_cases(n, seed):
    rng = np.random.default_rng(seed)
    cases = []
    while len(cases) < n:
        K = int(rng.integers(1, 6))
        N = int(rng.integers(max(8, 2 * K + 2), 2500))
        if (N * K) % 2:
            N += 1
        R = int(rng.integers(1, 100))
        beta = float(rng.choice([0.0, 0.3, 1.0, 2.5, float(rng.uniform(0.1, 3.0))]))
        iters = int(rng.integers(1, 30000))
        step = int(rng.integers(1, iters + 6))
        layout = str(rng.choice(["default", "wide", "single", "big"]))
        cases.append((N, K, R, beta, iters, step, layout))
    return cases


@pytest.mark.parametrize("N,K,R,beta,iters,step,layout", _random_cases(36, 20261003))
def test_randomized_shapes(pkg, oracle, monkeypatch, N, K, R, beta, iters, step, layout):
    """Seeded random (N, K, replicas, beta, iterations, sample step) with every layout of the random-site kernels forced in turn:
    table in LDS / in HBM, one word per site, spins in HBM (big-N kernels).  Ragged chunks, steps longer than the run, single
    replicas, K = 1: everything must equal the oracle."""
    monkeypatch.setenv("RRRMC_FORCE_WIDE", "1" if layout == "wide" else "0")
    if layout == "single":
        monkeypatch.setenv("RRRMC_FORCE_SINGLE", "1")
    if layout == "big":
        monkeypatch.setenv("RRRMC_FORCE_BIG", "1")
    seed = 31 * N + K
    X = pkg.GraphRRG(N, K, seed=seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc = eng.standard_mc(beta, iters, step)
        C1 = eng.get_config()
    ref = oracle.standard_mc_sparse_batch(X.A, X.J.astype(np.int32), beta, iters, step, seed, C0.s)
    assert Es.shape == ref[0].shape and (Es == ref[0]).all() and (C1.s == ref[1]).all() and (acc == ref[2]).all()


def test_hook_can_stop_single_replicas(pkg, oracle):
    """The reference's hook ends ONE chain (`hook(...) || break`, src/RRRMC.jl:107).  A batch hook that returns one flag per replica freezes
    the replicas it says False for: each returns what a reference call stopped there would — its samples up to that point and the
    configuration of that moment — while the others run to the end."""
    N, K, R, beta, iters, step, seed = 64, 3, 6, 1.3, 4000, 100, 4711
    X = pkg.GraphRRG(N, K, seed=seed)
    stop_at = {1: 500, 4: 2300}                       # replica -> iteration at which its hook says stop

    def hook(it, X_, C, accepted, E):
        return np.array([not (r in stop_at and it >= stop_at[r]) for r in range(R)])

    C0 = pkg.Config(N, R)
    C0.s[:] = oracle.init_configs(seed, 0, R, N)
    start = C0.s.copy()
    Es, C1 = pkg.standardMC(X, beta, iters, seed=seed, step=step, hook=hook, C0=C0, quiet=True)
    assert isinstance(Es, list) and len(Es) == R
    Ji = X.J.astype(np.int32)
    for r in range(R):
        n_it = stop_at.get(r, iters)
        # the reference chain stopped by its hook at iteration n_it has made n_it - 1 moves and pushed n_it // step samples
        ref = oracle.standard_mc_sparse(X.A, Ji, beta, n_it - 1 if r in stop_at else iters, step, seed, start[r], replica=r)
        want = oracle.standard_mc_sparse(X.A, Ji, beta, n_it, step, seed, start[r], replica=r)[0][:n_it // step]
        assert len(Es[r]) == n_it // step and (np.asarray(Es[r]) == want).all()
        assert (C1.s[r] == ref[1]).all()
