#!/usr/bin/env python3
"""Soak of the headline path's launch code (csrc/host_sweep.hpp, host_plan.hpp: a call's planner runs beside the previous call's sweep; two
chunk tables, two sets of plan buffers): random +-J graphs (K = 3 .. 6, lattices), replica counts, shard layouts and random SEQUENCES of
asynchronous standardMC calls of changing (iters, step) — queued back to back against the same calls with a sync after each (everything
compared), and the last call of some replicas against the oracle.

  python3 tests/soak/queue_soak.py [cases] [seed]"""
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
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 9090)
bad, t0 = 0, time.time()
for case in range(CASES):
    seed = int(rng.integers(1, 1 << 30))
    if rng.integers(3) == 0:
        L, D = [(4, 2), (6, 2), (4, 3), (6, 3), (16, 2)][int(rng.integers(5))]
        X, form = pkg.GraphEA(L, D, seed=seed), "ea"
    else:
        K = int(rng.choice([3, 3, 4, 5, 6]))
        N = int(rng.choice([64, 500, 4096, 9000, 20000, 40000]))         # (40 000: beyond the LDS-resident kernel — plan_big / big_apply)
        N += (N * K) % 2
        X, form = pkg.GraphRRG(N, K, seed=seed), "rrg"
    R = int(rng.choice([32, 96, 256, 1000]))
    if X.N > 10000:
        R = min(R, 256)
    shards = bool(R >= 64 and rng.integers(4) == 0)
    beta = float(rng.choice([0.3, 0.8, 1.5]))
    shapes = [(int(rng.integers(1, 40)) * int(s), int(s)) for s in rng.choice([1, 7, 64, 1000, 4096], size=int(rng.integers(1, 4)))]
    pattern = [shapes[int(rng.integers(len(shapes)))] for _ in range(int(rng.integers(2, 9)))]
    outs = []
    for queued in (True, False):
        with (pkg.Engine(X, R, devices=[0, 0]) if shards else pkg.Engine(X, R)) as eng:
            eng.seed(seed); eng.init_spins_random()
            before_last, done = None, 0
            for c, (iters, step) in enumerate(pattern):
                if c == len(pattern) - 1 and not queued:
                    before_last = eng.get_config().s.copy()
                eng.standard_mc_async(beta, iters, step)
                if not queued:
                    eng.sync()
                done += iters
            eng.sync()
            Es, acc = eng.fetch_results()
            outs.append((np.asarray(Es), np.asarray(acc), eng.get_config().s.copy(), np.asarray(eng.energy()), before_last))
    same = all(a.shape == b.shape and (a == b).all() for a, b in zip(outs[0][:4], outs[1][:4]))
    it0 = sum(i for i, _ in pattern[:-1])
    iters, step = pattern[-1]
    for r in sorted(set([0, R - 1, int(rng.integers(R))])):
        ref = O.standard_mc_sparse(X.A, X.J.astype(np.int32), beta, iters, step, seed, outs[1][4][r], replica=r, it0=it0, form=form)
        same &= bool((outs[0][0][r] == ref[0]).all() and (outs[0][2][r] == ref[1]).all() and outs[0][1][r] == ref[2])
    bad += 0 if same else 1
    print(json.dumps({"case": case, "graph": type(X).__name__, "N": int(X.N), "K": int(X.K), "R": R, "shards": shards, "beta": beta, "pattern": pattern, "same": bool(same)}), flush=True)
print(json.dumps({"cases": CASES, "mismatches": bad, "seconds": round(time.time() - t0, 1)}), flush=True)
sys.exit(1 if bad else 0)
