#!/usr/bin/env python3
"""Soak of GraphQuant over Float64 sparse slices (GraphQEAT, src/QAliases.jl:50-83; model RRRMC_MODEL_QUANT_F64): random lattices (L = 2 with
its doubled bonds, chains, 2D, 3D), Trotter numbers, fields, temperatures and all five samplers — a few replicas of every case against the
ORACLE (orc_*_quant_spf), bit for bit.

  python3 tests/soak/qeat_soak.py [cases] [seed]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as e  # noqa: E402
import oracle as O  # noqa: E402

pkg = e.load_package()
CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3131)
bad, t0 = 0, time.time()
for case in range(CASES):
    seed = int(rng.integers(1, 1 << 30))
    L, D = [(2, 2), (2, 3), (3, 1), (7, 1), (3, 2), (4, 2), (6, 2), (3, 3), (4, 3)][int(rng.integers(9))]
    M = int(rng.choice([3, 4, 5, 8, 12]))
    Gamma, betaq = float(rng.choice([0.3, 0.5, 1.2])), float(rng.choice([1.0, 2.0, 3.0]))
    X = pkg.GraphQEAT(L, D, M, Gamma, betaq, seed=seed)
    A, J, fourK = X.X1.A, X.X1.J, X.fourK
    smp = ["std", "rrr", "bkl", "wtm", "eo"][int(rng.integers(5))]
    R = int(rng.choice([3, 66]))
    beta = betaq if rng.integers(2) else float(rng.choice([0.5, 1.5]))
    step = int(rng.choice([1, 10, 100]))
    iters = int(rng.integers(3, 40)) * step
    thr = float(rng.choice([0.0, 0.5, 1.0]))
    with pkg.Engine(X, R) as eng:
        eng.seed(seed); eng.init_spins_random()
        C0 = eng.get_config().s.copy()
        if smp == "std":
            out = eng.standard_mc(beta, iters, step)
        elif smp == "rrr":
            out = eng.rrr_mc(beta, iters, step, staged_thr=thr)
        elif smp == "bkl":
            out = eng.bkl_mc(beta, iters, step)
        elif smp == "wtm":
            out = eng.wtm_mc(beta, max(iters // step, 1), float(step))
        else:
            out = eng.extremal_opt(1.4, iters, step)
        C1 = eng.get_config().s.copy()
    ok = True
    for r in sorted(set([0, R - 1, int(rng.integers(R))])):
        if smp == "std":
            ref = O.standard_mc_quant_spf(A, J, M, fourK, beta, iters, step, seed, C0[r], replica=r)
            ok &= bool((out[0][r] == ref[0]).all() and (C1[r] == ref[1]).all() and out[1][r] == ref[2])
        elif smp == "rrr":
            ref = O.rrr_mc_quant_spf(A, J, M, fourK, beta, iters, step, seed, C0[r], replica=r, staged_thr=thr)
            ok &= bool((out[0][r] == ref[0]).all() and (C1[r] == ref[1]).all() and out[1][r] == ref[2] and out[2][r] == ref[3])
        elif smp == "bkl":
            ref = O.cont_quant_spf("bkl", A, J, M, fourK, beta, iters, step, seed, C0[r], replica=r)
            n = min(out[0].shape[1], len(ref[0]))
            ok &= bool((out[0][r][:n] == ref[0][:n]).all() and (C1[r] == ref[1]).all() and out[1][r] == ref[2][0])
        elif smp == "wtm":
            ref = O.cont_quant_spf("wtm", A, J, M, fourK, beta, max(iters // step, 1), 1, seed, C0[r], replica=r, stepf=float(step))
            ok &= bool((out[0][r] == ref[0]).all() and (C1[r] == ref[1]).all() and out[1][r] == ref[2][0] and out[2][r] == ref[3])
        else:
            ref = O.extremal_opt_quant_spf(A, J, M, fourK, 1.4, iters, step, seed, C0[r], replica=r)
            ok &= bool((out[0][r] == ref[0]).all() and (C1[r] == ref[1]).all() and out[1][r] == ref[2] and (out[2].s[r] == ref[3]).all() and out[3][r] == ref[4])
    bad += 0 if ok else 1
    print(json.dumps({"case": case, "L": L, "D": D, "M": M, "Gamma": Gamma, "beta_quant": betaq, "smp": smp, "R": R, "beta": beta, "iters": iters, "step": step, "thr": thr, "same": ok}), flush=True)
print(json.dumps({"cases": CASES, "mismatches": bad, "seconds": round(time.time() - t0, 1)}), flush=True)
sys.exit(1 if bad else 0)
