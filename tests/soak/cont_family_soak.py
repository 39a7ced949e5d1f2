#!/usr/bin/env python3
"""Soak of the continuous-energy sampler family against the ORACLE on random shapes: rrrMC / bklMC / wtmMC / extremal_opt on GraphRRGNormal
(cont_wave_kernel, cont_sparse_kernel, eo_cont_wave_kernel), GraphSKNormal (rrr_skn_kernel, eo_sk_wave_kernel) and GraphQuant over +-J slices
(rrr_quant_wave_kernel / rrr_quant_kernel, cont_sparse_kernel's GraphQuant hooks), through the wave and the thread builds.  Graph families and
oracle adapters are those of tests/test_gpu_hooks.py with random sizes.

  python3 tests/soak/cont_family_soak.py [cases] [seed]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as e  # noqa: E402
import oracle as O  # noqa: E402
import test_gpu_hooks as T  # noqa: E402

pkg = e.load_package()
CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5151)
KEYS = ("RRRMC_CONT_NO_WAVE", "RRRMC_EO_NO_WAVE", "RRRMC_QUANT_NO_WAVE")
bad, t0 = 0, time.time()
for case in range(CASES):
    fam = ["rrgn", "skn", "quant"][int(rng.integers(3))]
    if fam == "rrgn":
        K = int(rng.choice([3, 4, 5, 6]))
        N = int(rng.choice([64, 100, 257, 600]))
        N += (N * K) % 2
        M = T.RRGNormal(pkg, N, K)
    elif fam == "skn":
        M = T.SKNormal(pkg, int(rng.choice([10, 24, 50, 96])))
    else:
        M = T.Quant(pkg, int(rng.choice([10, 16, 32])), int(rng.choice([3, 4, 8])), float(rng.choice([0.3, 0.5, 1.0])), 2.0)
    X = M.X
    smp = ["rrr", "bkl", "wtm", "eo"][int(rng.integers(4))]
    env = [{}, {"RRRMC_CONT_NO_WAVE": "1"}, {"RRRMC_EO_NO_WAVE": "1"}, {"RRRMC_QUANT_NO_WAVE": "1"}][int(rng.integers(4))]
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    R = int(rng.choice([2, 40, 70]))
    beta = 2.0 if fam == "quant" else float(rng.choice([0.6, 1.2, 2.0]))          # (a GraphQuant's beta is part of the graph)
    tau = float(rng.choice([1.2, 1.6]))
    thr = [None, 0.0, 1.0][int(rng.integers(3))] if smp == "rrr" else None
    step = int(rng.choice([1, 10, 50]))
    nsamp = int(rng.integers(3, 30))
    iters = nsamp * step
    n_arg = nsamp if smp == "wtm" else iters
    seed = T.SEED + case
    with pkg.Engine(X, R) as eng:
        eng.seed(seed); eng.init_spins_random()
        C0 = eng.get_config().s.copy()
        Es, cnt = T.engine_call(eng, smp, n_arg, step, beta, tau, thr)
        C1 = eng.get_config().s.copy()
    ok = True
    for r in sorted(set([0, R - 1, int(rng.integers(R))])):
        oEs, och, ocnt = M.oracle(O, smp, C0[r], r, seed, n_arg, step, beta=beta, tau=tau, thr=thr)
        n = min(np.asarray(Es).shape[1], len(oEs))
        ok &= bool((np.asarray(Es)[r][:n] == np.asarray(oEs)[:n]).all() and (C1[r] == och).all())
        if smp == "eo":
            ok &= bool(cnt[r, 0] == ocnt[0] and cnt[r, 1] == ocnt[2])
        else:
            ok &= bool(np.asarray(cnt)[r] == ocnt)
    bad += 0 if ok else 1
    print(json.dumps({"case": case, "family": fam, "N": int(X.N), "smp": smp, "R": R, "beta": beta, "thr": thr, "step": step, "samples": nsamp, "env": env, "same": ok}), flush=True)
print(json.dumps({"cases": CASES, "mismatches": bad, "seconds": round(time.time() - t0, 1)}), flush=True)
sys.exit(1 if bad else 0)
