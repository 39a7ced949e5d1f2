"""Randomised cross-check (GPU box) of the wavefront-per-replica builds against the thread-per-replica builds: rrrMC / bklMC on GraphRRG /
GraphEA (rrr_sparse_wave_kernel), rrrMC / standardMC on GraphQuant over GraphRRG and binary GraphSK slices (rrr_quant_wave_kernel,
quant_standard_wave_kernel).  Random sizes, temperatures, staged thresholds and segment slacks; every pair must agree bit for bit.
python tools/stress_wave.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e  # noqa: E402

pkg = e.load_package()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
WAVE_ENV = ("RRRMC_RRR_NO_WAVE", "RRRMC_QUANT_NO_WAVE")


def run(X, R, seed, calls):
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        out = []
        for name, args in calls:
            res = getattr(eng, name)(*args)
            out += [np.asarray(a).copy() for a in res]
            out.append(eng.get_config().s.copy())
        return out


bad = 0
for case in range(ncases):
    seed = int(rng.integers(1, 1 << 30))
    kind = ["rrg", "ea", "qrrg", "qsk"][case % 4]
    beta = float(rng.choice([0.5, 1.0, 2.0, 4.0]))
    thr = float(rng.choice([0.0, 0.5, 0.8, 1.0]))
    R = int(rng.integers(1, 9))
    for v in WAVE_ENV + ("RRRMC_RRR_WAVE_SLACK", "RRRMC_QUANT_WAVE_SLACK"):
        os.environ.pop(v, None)
    if kind == "rrg":
        K = int(rng.choice([3, 4, 5, 6]))
        N = int(rng.integers(20, 3000)) // 2 * 2
        X = pkg.GraphRRG(N, K, seed=seed)
        iters = int(rng.integers(1000, 30000))
        calls = [("rrr_mc", (beta, iters, max(iters // 7, 1), thr)), ("bkl_mc", (beta, 10 * iters, max(iters, 1)))]
        L = len(pkg.all_delta_e(X))
        if rng.random() < 0.5:
            os.environ["RRRMC_RRR_WAVE_SLACK"] = str(2 * L * 64 * 2 * (K + 1))
        desc = "GraphRRG(%d, %d)" % (N, K)
    elif kind == "ea":
        Lx, D = [(2, 3), (3, 3), (4, 3), (6, 3), (5, 2), (12, 2), (10, 3)][int(rng.integers(0, 7))]
        X = pkg.GraphEA(Lx, D, seed=seed)
        iters = int(rng.integers(1000, 20000))
        calls = [("rrr_mc", (beta, iters, max(iters // 5, 1), thr)), ("bkl_mc", (beta, 10 * iters, max(iters, 1)))]
        if rng.random() < 0.5:
            os.environ["RRRMC_RRR_WAVE_SLACK"] = str(2 * len(pkg.all_delta_e(X)) * 64 * 2 * (2 * D + 1))
        desc = "GraphEA(%d, %d)" % (Lx, D)
    elif kind == "qrrg":
        Nk, M = int(rng.integers(8, 200)) // 2 * 2, int(rng.integers(3, 12))
        X = pkg.GraphQuant(pkg.GraphRRG(Nk, 3, seed=seed), M, float(rng.choice([0.3, 0.5, 1.0])), beta)
        iters = int(rng.integers(1000, 20000))
        calls = [("standard_mc", (beta, iters, max(iters // 9, 1))), ("rrr_mc", (beta, iters, max(iters // 4, 1), thr))]
        if rng.random() < 0.5:
            os.environ["RRRMC_QUANT_WAVE_SLACK"] = "2048"
        desc = "GraphQuant(RRG(%d, 3), M=%d)" % (Nk, M)
    else:
        Nk, M = int(rng.integers(5, 300)), int(rng.integers(3, 10))
        X = pkg.GraphQSKT(Nk, M, float(rng.choice([0.3, 0.5, 1.0])), beta, seed=seed)
        iters = int(rng.integers(1000, 15000))
        calls = [("standard_mc", (beta, iters, max(iters // 9, 1))), ("rrr_mc", (beta, iters, max(iters // 4, 1), thr))]
        desc = "GraphQSKT(%d, %d)" % (Nk, M)
    wave = run(X, R, seed, calls)
    for v in WAVE_ENV:
        os.environ[v] = "1"
    thread = run(X, R, seed, calls)
    ok = all((a == b).all() for a, b in zip(wave, thread))
    bad += 0 if ok else 1
    print("%-3d %-34s R=%d beta=%.1f thr=%.1f slack=%s  %s" % (case, desc, R, beta, thr, os.environ.get("RRRMC_RRR_WAVE_SLACK") or os.environ.get("RRRMC_QUANT_WAVE_SLACK") or "-", "ok" if ok else "MISMATCH"), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
