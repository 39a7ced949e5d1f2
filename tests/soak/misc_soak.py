#!/usr/bin/env python3
"""Soak of the remaining +-J / Float64 paths against the ORACLE on random shapes:
  colour     colour-parallel sweeps (BASELINE config 4's sampler, src-defined in DESIGN 4b) on random even lattices (checkerboard) and random
             regular graphs (greedy colouring), two consecutive calls (the SWEEP stream continues);
  fast       the opt-in fast Float64 standardMC (spf_fast_kernel, K <= 4) against its own oracle restatement: configurations and accepted counts
             identical, energies within 1e-9 relative;
  overlaps   device snapshots + pm1dot overlaps (scripts/scripts.jl:283-295) of random configurations.

  python3 tests/soak/misc_soak.py [cases] [seed]"""
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
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7171)
bad, t0 = 0, time.time()
for case in range(CASES):
    seed = int(rng.integers(1, 1 << 30))
    what = ["colour", "colour", "fast", "overlaps"][int(rng.integers(4))]
    R = int(rng.choice([32, 40, 96, 200]))
    beta = float(rng.choice([0.5, 1.0, 2.0]))
    ok, desc = True, {}
    try:
        if what == "colour":
            if rng.integers(3):
                L, D = [(2, 3), (4, 2), (4, 3), (6, 2), (6, 3), (8, 2), (10, 2), (2, 4), (4, 4)][int(rng.integers(9))]
                X = pkg.GraphEA(L, D, seed=seed)
                color = pkg.checkerboard_coloring(L, D)
            else:
                K = int(rng.choice([3, 4, 5, 6])); N = int(rng.choice([30, 100, 400])); N += (N * K) % 2
                X = pkg.GraphRRG(N, K, seed=seed)
                color = O.greedy_coloring(X.A)
            A, J = X.A, X.J.astype(np.int32)
            sweeps, step = int(rng.integers(2, 20)), int(rng.choice([1, 2, 5]))
            sweeps -= sweeps % step
            sweeps = max(sweeps, step)
            with pkg.Engine(X, R) as eng:
                eng.seed(seed); eng.init_spins_random()
                eng.set_coloring(color)
                C0 = eng.get_config().s.copy()
                Es = eng.colored_sweeps(beta, sweeps, step)
                C1 = eng.get_config().s.copy()
                Es2 = eng.colored_sweeps(beta, 2, 1)
                C2 = eng.get_config().s.copy()
            for r in sorted(set([0, R - 1, int(rng.integers(R))])):
                a = O.colored_sweeps_sparse(A, J, color, beta, sweeps, step, seed, C0[r], replica=r)
                b = O.colored_sweeps_sparse(A, J, color, beta, 2, 1, seed, a[1], sweep0=sweeps, replica=r)
                ok &= bool((Es[r] == a[0]).all() and (C1[r] == a[1]).all() and (Es2[r] == b[0]).all() and (C2[r] == b[1]).all())
            desc = {"graph": type(X).__name__, "N": int(X.N), "K": int(X.K), "colours": int(max(color)) + 1, "sweeps": sweeps, "step": step}
        elif what == "fast":
            if rng.integers(2):
                L, D = [(4, 2), (6, 2), (10, 2)][int(rng.integers(3))]; X, form = pkg.GraphEANormal(L, D, seed=seed), "ea"
            else:
                K = int(rng.choice([3, 4])); N = int(rng.choice([64, 500, 2000])); N += (N * K) % 2
                X, form = pkg.GraphRRGNormal(N, K, seed=seed), "rrg"
            step = int(rng.choice([16, 100, 1000]))
            iters = int(rng.integers(2, 20)) * step
            with pkg.Engine(X, R) as eng:
                eng.seed(seed); eng.init_spins_random()
                C0 = eng.get_config().s.copy()
                Es, acc = eng.standard_mc_fast(beta, iters, step)
                C1 = eng.get_config().s.copy()
            for r in sorted(set([0, R - 1, int(rng.integers(R))])):
                a = O.standard_mc_spf_fast(X.A, X.J, beta, iters, step, seed, C0[r], replica=r, form=form)
                ok &= bool((C1[r] == a[1]).all() and acc[r] == a[2] and np.allclose(Es[r], a[0], rtol=1e-9, atol=1e-9))
            desc = {"graph": type(X).__name__, "N": int(X.N), "K": int(X.K), "iters": iters, "step": step}
        else:
            N = int(rng.choice([10, 64, 65, 1000, 4097]))
            X = pkg.GraphRRG(N + N % 2, 3, seed=seed)
            with pkg.Engine(X, R) as eng:
                eng.seed(seed); eng.init_spins_random()
                eng.snapshot_reserve(2)
                eng.snapshot_store(0)
                Ca = eng.get_config().s.copy()
                eng.standard_mc(beta, 5 * X.N, 5 * X.N, want_energies=False)
                eng.snapshot_store(1)
                Cb = eng.get_config().s.copy()
                ov = np.asarray(eng.overlaps(0, 1))[0]
            for r in sorted(set([0, R - 1, int(rng.integers(R))])):
                ok &= bool(int(ov[r]) == O.pm1dot(Ca[r], Cb[r], X.N))
            desc = {"N": int(X.N)}
    except pkg.RRRMCError as err:
        print(json.dumps({"case": case, "what": what, "skipped": str(err)[:100]}), flush=True)
        continue
    bad += 0 if ok else 1
    print(json.dumps(dict(desc, case=case, what=what, R=R, beta=beta, same=ok)), flush=True)
print(json.dumps({"cases": CASES, "mismatches": bad, "seconds": round(time.time() - t0, 1)}), flush=True)
sys.exit(1 if bad else 0)
