#!/usr/bin/env python3
"""Soak of the kernels whose apply_move! was rewritten as a staged gather this round (rrr_sparse_kernel, eo_sparse_kernel with its list ends
followed in registers, rrr_dbl_kernel with its gathered staged step): random graphs (+-J and general levels, K = 3 .. 6, lattices with doubled
bonds), temperatures, staged thresholds, kernel builds — a few replicas of every case against the ORACLE (tests/ infrastructure: this tool
is a test, it ships nothing).

  python3 tests/soak/gather_soak.py [cases] [seed]     -> one line per case, a summary line; exit code 1 on a mismatch"""
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
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 606)
KEYS = ("RRRMC_RRR_NO_WAVE", "RRRMC_RRR_NO_LDS", "RRRMC_EO_NO_FTAU_LDS")
LEVS = [None, None, (-1, 0, 1), (-2, -1, 1, 2), (-1.5, -0.5, 0.5, 1.5)]


def sparse_graph(seed):
    lev = LEVS[int(rng.integers(len(LEVS)))]
    if rng.integers(3) == 0:
        L, D = [(2, 3), (2, 2), (4, 2), (3, 3), (5, 2), (4, 3)][int(rng.integers(6))]
        X = pkg.GraphEA(L, D, seed=seed) if lev is None else pkg.GraphEA(L, D, lev, seed=seed)
        return X, "ea", lev
    K = int(rng.choice([3, 4, 5, 6]))
    N = int(rng.choice([12, 40, 130, 400]))
    N += (N * K) % 2
    X = pkg.GraphRRG(N, K, seed=seed) if lev is None else pkg.GraphRRG(N, K, lev, seed=seed)
    return X, "rrg", lev


bad, t0 = 0, time.time()
for case in range(CASES):
    seed = int(rng.integers(1, 1 << 30))
    what = ["rrr", "bkl", "eo", "dbl"][int(rng.integers(4))]
    env = [{"RRRMC_RRR_NO_WAVE": "1", "RRRMC_RRR_NO_LDS": "1"}, {"RRRMC_RRR_NO_WAVE": "1", "RRRMC_RRR_NO_LDS": "1"}, {}, {"RRRMC_EO_NO_FTAU_LDS": "1"}][int(rng.integers(4))]
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    R = int(rng.choice([5, 70, 300]))
    beta, tau = float(rng.choice([0.4, 1.0, 1.6, 2.5])), float(rng.choice([1.1, 1.5, 2.2]))
    thr = float(rng.choice([0.0, 0.5, 1.0]))
    iters, step = int(rng.integers(500, 6000)), int(rng.choice([1, 7, 100]))
    reps = sorted(set([0, R - 1, int(rng.integers(R))]))
    ok = True
    try:
        if what == "dbl":
            lev = [(-1, 0, 1), (-2, -1, 1, 2), (-0.8, 0.0, 0.8)][int(rng.integers(3))]
            if rng.integers(2):
                X, form = pkg.GraphRRGNormalDiscretized(int(rng.choice([20, 100, 300])), int(rng.choice([3, 4])), lev, seed=seed), "rrg"
            else:
                L, D = [(2, 3), (4, 2), (3, 3)][int(rng.integers(3))]
                X, form = pkg.GraphEANormalDiscretized(L, D, lev, seed=seed), "ea"
            units, mul, div = O.dfloat_units(lev)
            with pkg.Engine(X, R) as eng:
                eng.seed(seed); eng.init_spins_random()
                C0 = eng.get_config()
                Es, acc, st = eng.rrr_mc(beta, iters, step, staged_thr=thr)
                C1 = eng.get_config()
                pos, sizes = eng.rrr_cache()
            for r in reps:
                ref = O.rrr_double_sparse(X.A, X.dJ, X.rJ, units, beta, iters, step, seed, C0.s[r], replica=r, staged_thr=thr, form=form, mul=mul, div=div)
                ok &= bool((Es[r] == ref[0]).all() and (C1.s[r] == ref[1]).all() and acc[r] == ref[2] and st[r] == ref[3] and (pos[r] == ref[4]).all())
            desc = {"graph": type(X).__name__, "N": int(X.N), "K": int(X.K), "lev": list(lev)}
        else:
            X, form, lev = sparse_graph(seed)
            kw = {}
            if lev is not None:
                units, mul, div = pkg.level_units(lev)
                kw = dict(lev=units, mul=mul, div=div)
            val = (lambda x: X.energy_value(x)) if lev is not None else (lambda x: x)
            J = X.J if lev is not None else X.J.astype(np.int32)
            with pkg.Engine(X, R) as eng:
                eng.seed(seed); eng.init_spins_random()
                C0 = eng.get_config()
                if what == "rrr":
                    Es, acc, st = eng.rrr_mc(beta, iters, step, staged_thr=thr)
                elif what == "bkl":
                    Es, acc = eng.bkl_mc(beta, iters, step)
                else:
                    Es, Emin, Cmin, itmin = eng.extremal_opt(tau, iters, step)
                C1 = eng.get_config()
                cache = eng.rrr_cache() if what != "eo" else None
            for r in reps:
                if what == "eo":
                    ekw = {"lev": kw["lev"]} if kw else {}
                    ref = O.extremal_opt_sparse(X.A, J, tau, iters, step, seed, C0.s[r], replica=r, form=form, **ekw)
                    ok &= bool((Es[r] == val(ref[0])).all() and (C1.s[r] == ref[1]).all() and Emin[r] == val(ref[2]) and (Cmin.s[r] == ref[3]).all() and itmin[r] == ref[4])
                else:
                    ref = O.rrr_sparse(X.A, J, beta, iters, step, seed, C0.s[r], replica=r, staged_thr=thr, form=form, bkl=what == "bkl", want_cache=True, **kw)
                    n = min(Es.shape[1], len(ref[0]))
                    ok &= bool((Es[r][:n] == val(ref[0])[:n]).all() and (C1.s[r] == ref[1]).all() and acc[r] == ref[2] and (cache[0][r] == ref[5]).all())
            desc = {"graph": type(X).__name__, "form": form, "N": int(X.N), "K": int(X.K), "lev": None if lev is None else list(lev)}
    except pkg.RRRMCError as err:
        print(json.dumps({"case": case, "what": what, "skipped": str(err)[:120]}), flush=True)
        continue
    bad += 0 if ok else 1
    print(json.dumps(dict(desc, case=case, what=what, R=R, beta=beta, tau=tau, thr=thr, iters=iters, step=step, env=env, same=ok)), flush=True)
print(json.dumps({"cases": CASES, "mismatches": bad, "seconds": round(time.time() - t0, 1)}), flush=True)
sys.exit(1 if bad else 0)
