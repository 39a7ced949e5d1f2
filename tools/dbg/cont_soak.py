#!/usr/bin/env python3
"""Soak: cont_wave_kernel (rrrMC / bklMC on the Float64 sparse models, one wavefront per replica) against cont_sparse_kernel
(RRRMC_CONT_NO_WAVE=1, one thread per replica) over seeded random shapes, temperatures and thresholds — energies, counts, configurations and
the recomputed energy must agree bit for bit.  Not a test of the suite (minutes of GPU time): python tools/dbg/cont_soak.py [cases = 40]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
rng = np.random.default_rng(20261004)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
total = 0
for c in range(cases):
    K = int(rng.choice([3, 3, 4, 5, 6]))
    N = int(rng.integers(64, 6000))
    if (N * K) % 2:
        N += 1
    lattice = c % 5 == 4
    seed = int(rng.integers(1, 1 << 30))
    X = pkg.GraphEANormal(int(rng.integers(4, 12)), 3, seed=seed) if lattice else pkg.GraphRRGNormal(N, K, seed=seed)
    R = int(rng.integers(1, 200))
    beta = float(rng.choice([0.3, 1.0, 2.0, 4.0]))
    mode = "bkl" if c % 3 == 2 else "rrr"
    iters = int(rng.integers(2000, 60000))
    step = int(rng.integers(1, 5000))
    thr = float(rng.choice([0.0, 0.5, 0.8, 1.0]))
    outs = []
    for env in (None, "1"):
        if env is None:
            os.environ.pop("RRRMC_CONT_NO_WAVE", None)
        else:
            os.environ["RRRMC_CONT_NO_WAVE"] = env
        with pkg.Engine(X, R) as eng:
            eng.seed(seed)
            eng.init_spins_random()
            a = eng.rrr_mc(beta, iters, step, staged_thr=thr) if mode == "rrr" else eng.bkl_mc(beta, iters, step)
            b = eng.rrr_mc(beta, iters // 3 + 1, 7, staged_thr=thr) if mode == "rrr" else eng.bkl_mc(beta, iters // 3 + 1, 50)
            outs.append((a, b, eng.get_config().s.copy(), eng.energy()))
    ok = all((np.asarray(u) == np.asarray(w)).all() for u, w in zip(outs[0][0] + outs[0][1], outs[1][0] + outs[1][1]))
    ok = ok and (outs[0][2] == outs[1][2]).all() and (outs[0][3] == outs[1][3]).all()
    total += R * (iters + iters // 3 + 1)
    bad += not ok
    print("%s %s N=%d K=%d R=%d beta=%g iters=%d step=%d thr=%g: %s" % ("ea" if lattice else "rrg", mode, X.N, X.A.shape[1], R, beta, iters, step, thr,
                                                                    "identical" if ok else "MISMATCH"), flush=True)
print("%d cases, %d mismatches, %.3g replica-iterations" % (cases, bad, total))
sys.exit(1 if bad else 0)
