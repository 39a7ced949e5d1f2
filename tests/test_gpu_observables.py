"""GPU parity of the device-side observables: snapshots, pm1dot overlaps for the three native spin
layouts, the parseovs window statistic and GraphQuant's Qenergy / transverse_mag / overlaps, all against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_with_snapshots(pkg, eng, X, beta, nsnap, step, kind):
    eng.snapshot_reserve(nsnap)
    cfgs = []
    for k in range(nsnap):
        if kind == "rrr":
            eng.rrr_mc(beta, step, step)
        else:
            eng.standard_mc(beta, step, step)
        eng.snapshot_store(k)
        cfgs.append(eng.get_config().s.copy())
    return cfgs


@pytest.mark.parametrize("model,N,R", [
    ("rrg", 200, 70),        # bit-sliced 32 replicas per word, R not a multiple of 32, N not a multiple of 256
    ("rrg", 4096, 64),       # BASELINE config 2 geometry
    ("skn", 100, 13),        # byte-sliced SK layout, R not a multiple of 8
    ("quant", 10 * 8, 5),    # chunk layout
    ("spf", 150, 70),        # GraphRRGNormal: one uint64 per (wavefront, site)
    ("dbl", 90, 9),          # GraphRRGNormalDiscretized: chunk layout
    ("lev", 90, 33),         # GraphRRG with general levels: chunk layout
    ("skb", 70, 11),         # GraphSK (binary couplings): byte-sliced
])
def test_overlaps_match_pm1dot(pkg, oracle, model, N, R):
    seed = 9001 + N
    if model == "rrg":
        X = pkg.GraphRRG(N, 3, seed=seed)
    elif model == "skn":
        X = pkg.GraphSKNormal(N, seed=seed)
    elif model == "spf":
        X = pkg.GraphRRGNormal(N, 3, seed=seed)
    elif model == "dbl":
        X = pkg.GraphRRGNormalDiscretized(N, 3, (-1, 0, 1), seed=seed)
    elif model == "lev":
        X = pkg.GraphRRG(N, 3, (-1.0, 0.0, 1.0), seed=seed)
    elif model == "skb":
        X = pkg.GraphSK(N, seed=seed)
    else:
        X = pkg.GraphQuant(pkg.GraphRRG(10, 3, seed=seed), 8, 0.5, 2.0)
    T = 5
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        cfgs = _run_with_snapshots(pkg, eng, X, 1.0, T, 3 * N, "rrr" if model == "quant" else "std")
        for k in range(T):                                             # BitMatrix column dump == the configuration at that time
            assert (eng.snapshot_get(k).s == cfgs[k]).all()
        pa = [(a, b) for a in range(T) for b in range(T)]
        q = eng.overlaps([p[0] for p in pa], [p[1] for p in pa])
        live = eng.overlaps([-1, 2], [0, -1])
        with pytest.raises(pkg.RRRMCError):
            eng.overlaps([0], [T])                                     # slot out of range
    assert q.shape == (T * T, R)
    for idx, (a, b) in enumerate(pa):
        for r in range(R):
            assert q[idx, r] == oracle.pm1dot(cfgs[a][r], cfgs[b][r], X.N)
    for r in range(R):
        assert live[0, r] == oracle.pm1dot(cfgs[T - 1][r], cfgs[0][r], X.N)
        assert live[1, r] == oracle.pm1dot(cfgs[2][r], cfgs[T - 1][r], X.N)


def test_snapshot_log_and_parseovs(pkg, oracle, tmp_path):
    """The scripts' pipeline end to end: hook -> log file + device snapshots -> q^2(t) windows (scripts/scripts.jl:51-69, 368-405)."""
    seed, N, R, step, nsamp = 77, 128, 6, 500, 12
    X = pkg.GraphRRG(N, 3, seed=seed)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        log = pkg.SnapshotLog(eng, nsamp, prefix=str(tmp_path / "output_met"))
        Es, C = pkg.standardMC(X, 2.0, step * nsamp, step=step, seed=seed, hook=log, quiet=True, engine=eng)
        log.close()
        assert log.count == nsamp and log.mctimes == [step * (k + 1) for k in range(nsamp)]
        # a synthetic clock (the real one is machine-dependent): sample k at t = k + 1
        tsm = [float(k + 1) for k in range(nsamp)]
        mq2, sq2 = pkg.parseovs(eng, tsm, pkg.log_range(1.0, float(nsamp)))
        snaps = [eng.snapshot_get(k).s.copy() for k in range(nsamp)]
        cols, chunks = log.to_mat(2)
    # log format: header + one line per sample, 4 columns, E equals the sampled energy
    lines = (tmp_path / "output_met_r2.txt").read_text().splitlines()
    assert lines[0] == "#mctime acc E clocktime" and len(lines) == 1 + nsamp
    for k, line in enumerate(lines[1:]):
        mct, acc, E, t = line.split()
        assert int(mct) == step * (k + 1) and int(E) == Es[2, k] and float(t) >= 0 and int(acc) >= 0
    assert pkg.parsets(str(tmp_path / "output_met_r2.txt")) == log.clock
    # BitMatrix dump of replica 2
    for k in range(nsamp):
        assert (pkg.Config.from_bits(cols[:, k]).s[0] == snaps[k][2]).all()
    assert chunks.size == (N * nsamp + 63) // 64
    # windows against the oracle's restatement of parseovs, per replica
    w = 0
    for t_st in pkg.log_range(1.0, float(nsamp)):
        i, j = pkg.get_ts_range(tsm, t_st)
        if i is None:
            break
        j = nsamp if j is None else j
        for r in range(R):
            m, s = oracle.q2_window(np.stack([snaps[k][r] for k in range(nsamp)]), N, i, j)
            if np.isnan(m):
                assert np.isnan(mq2[w, r])
            else:
                assert mq2[w, r] == pytest.approx(m, rel=1e-12) and sq2[w, r] == pytest.approx(s, rel=1e-9, abs=1e-12)
        w += 1
    assert w == mq2.shape[0] and w >= 3


@pytest.mark.parametrize("Nk,M,R", [(10, 8, 7), (34, 5, 3), (1024, 32, 2)])
def test_quant_observables(pkg, oracle, Nk, M, R):
    seed = 31337 + Nk
    beta, Gamma = 2.0, 0.5
    X = pkg.GraphQuant(pkg.GraphRRG(Nk, 3, seed=seed), M, Gamma, beta)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        eng.rrr_mc(beta, 5000, 5000)
        C = eng.get_config()
        Q, tm, ov = eng.quant_observables()
    for r in range(R):
        Qr, tmr, ovr, *_ = oracle.quant_observables(X.A, X.J.astype(np.int32), M, X.fourK, beta, Gamma, C.s[r])
        assert Q[r] == Qr and tm[r] == tmr and (ov[r] == ovr).all()
