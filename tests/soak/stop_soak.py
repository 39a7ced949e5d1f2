#!/usr/bin/env python3
"""Soak of the hook loops (rrrmc.jl_amd/engine.py: rrrMC / bklMC / wtmMC / extremal_opt with hook=): R replicas, each with its OWN random
sample at which its hook returns false — every replica's samples, final configuration (the one its stopping hook saw: the move of that
iteration is not made) and, for extremal_opt, Emin / Cmin / itmin against the ORACLE run with the same stopping hook (the oracle calls a C
hook at the reference's sample points).  Graph families and oracle adapters are those of tests/test_gpu_hooks.py.

  python3 tests/soak/stop_soak.py [cases] [seed]"""
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
CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
SEED = T.SEED
models = {name: cls(pkg) for name, cls in T.MODELS.items()}
bad, t0 = 0, time.time()
for case in range(CASES):
    name = list(models)[int(rng.integers(len(models)))]
    M = models[name]
    X = M.X
    smp = ["rrr", "bkl", "wtm", "eo"][int(rng.integers(4))]
    R = int(rng.integers(1, 7))
    step = int(rng.choice([10, 50, 100]))
    nsamp = int(rng.integers(3, 16))
    iters = nsamp * step
    stops = [None if rng.integers(3) == 0 else int(rng.integers(1, nsamp + 1)) for _ in range(R)]      # replica r stops at its stops[r]-th sample
    calls = [0]

    def hook(it, X_, C, a, b):
        calls[0] += 1
        return np.array([s is None or calls[0] < s for s in stops])

    fn = T.front(pkg, smp)
    arg = 1.3 if smp == "eo" else 2.0
    n_arg = nsamp if smp == "wtm" else iters
    res = fn(X, arg, n_arg, seed=SEED, step=float(step) if smp == "wtm" else step, hook=hook, quiet=True, replicas=R)
    C0 = O.init_configs(SEED, 0, R, X.N)
    ok = True
    for r in range(R):
        oc = [0]

        def ohook(*a, s=stops[r]):
            oc[0] += 1
            return s is None or oc[0] < s

        with O.hooked(ohook):
            oEs, och, ocnt = M.oracle(O, smp, C0[r], r, SEED, n_arg, step, beta=2.0, tau=1.3)
        n = nsamp if stops[r] is None else stops[r]
        if smp == "eo":
            C, Emin, Cmin, itmin = res
            ok &= bool((C.s[r] == och).all() and np.asarray(Emin)[r] == ocnt[0] and (Cmin.s[r] == ocnt[1]).all() and np.asarray(itmin)[r] == ocnt[2])
        else:
            Es, C = res
            er = np.asarray(Es[r]) if R > 1 else np.asarray(Es).reshape(-1)
            ok &= bool(len(er) == n and (er == np.asarray(oEs)[:n]).all() and (C.s[r] == och).all())
    bad += 0 if ok else 1
    print(json.dumps({"case": case, "model": name, "smp": smp, "R": R, "step": step, "samples": nsamp, "stops": stops, "same": ok}), flush=True)
print(json.dumps({"cases": CASES, "mismatches": bad, "seconds": round(time.time() - t0, 1)}), flush=True)
sys.exit(1 if bad else 0)
