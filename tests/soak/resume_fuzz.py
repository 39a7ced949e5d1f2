#!/usr/bin/env python3
"""Differential fuzz of the resume contract (include/rrrmc_hip.h, rrrmc_set_resume): between two resumed calls of a sampler,
  (a) operations that only READ (configuration, tracked energy, cache, snapshots, timing, observables) must change nothing:
      the run continues exactly as without them;
  (b) operations that END a run (energy, set_config, init_spins_random + set_config, another sampler, changed parameters, seed) must make the
      next call start a fresh run — i.e. behave exactly as the same call with resume switched off.
Random graph families, samplers, replica counts.  GPU against GPU; one line per case, exit code 1 on a mismatch.

  python3 tests/soak/resume_fuzz.py [cases] [seed]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as e  # noqa: E402

pkg = e.load_package()
CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 77)


def graph(kind, seed):
    return {"rrg": lambda: pkg.GraphRRG(int(rng.choice([12, 60, 200])), 3, seed=seed),
            "ea": lambda: pkg.GraphEA(2, 3, seed=seed),
            "levels": lambda: pkg.GraphRRG(30, 3, (-1, 0, 1), seed=seed),
            "rrgn": lambda: pkg.GraphRRGNormal(int(rng.choice([64, 150])), 3, seed=seed),
            "disc": lambda: pkg.GraphRRGNormalDiscretized(40, 3, (-1, 0, 1), seed=seed),
            "skn": lambda: pkg.GraphSKNormal(int(rng.choice([12, 40])), seed=seed),
            "sk": lambda: pkg.GraphSK(16, seed=seed),
            "quant": lambda: pkg.GraphQuant(pkg.GraphRRG(12, 3, seed=seed), 4, 0.5, 2.0),
            "qeat": lambda: pkg.GraphQEAT(4, 2, 4, 0.5, 2.0, seed=seed)}[kind]()


def call(eng, smp, n, step, beta, tau):
    if smp == "rrr":
        Es, acc, st = eng.rrr_mc(beta, n, step)
        return [np.asarray(Es), np.asarray(acc), np.asarray(st)]
    if smp == "bkl":
        Es, mv = eng.bkl_mc(beta, n, step)
        return [np.asarray(Es), np.asarray(mv)]
    if smp == "wtm":
        Es, mv, t = eng.wtm_mc(beta, max(n // step, 1), float(step))
        return [np.asarray(Es), np.asarray(mv), np.asarray(t)]
    if smp == "std":
        Es, acc = eng.standard_mc(beta, n - n % step, step)
        return [np.asarray(Es), np.asarray(acc)]
    Es, Emin, Cmin, itmin = eng.extremal_opt(tau, n, step)
    return [np.asarray(Es), np.asarray(Emin), np.asarray(Cmin.s), np.asarray(itmin)]


READS = ["get_config", "run_energy", "rrr_cache", "last_timing", "snapshot", "iterations_done"]
ENDS = ["energy", "set_config", "other_sampler", "new_beta", "seed"]


def do_read(eng, op, X, smp):
    if op == "get_config":
        eng.get_config()
    elif op == "run_energy":
        eng.run_energy()
    elif op == "rrr_cache":
        if smp in ("rrr", "bkl") and X.model_kind in (1, 7) or (smp == "rrr" and X.model_kind in (3, 6)):
            eng.rrr_cache()
    elif op == "last_timing":
        eng.last_timing()
    elif op == "snapshot":
        eng.snapshot_reserve(2); eng.snapshot_store(0); eng.snapshot_store(1); eng.overlaps(0, 1)
    else:
        eng.iterations_done()


def scenario(X, R, smp, n1, n2, step, beta, tau, seed, mid, fresh_second):
    """call 1 (a fresh run), `mid`, call 2 with resume on — or, fresh_second, with resume off"""
    with pkg.Engine(X, R) as eng:
        eng.seed(seed); eng.init_spins_random()
        eng.set_resume(True)
        a = call(eng, smp, n1, step, beta, tau)
        beta2 = beta
        if mid in READS:
            do_read(eng, mid, X, smp)
        elif mid == "energy":
            eng.energy()
        elif mid == "set_config":
            eng.set_config(eng.get_config())
        elif mid == "other_sampler":
            other = "std" if smp != "std" else "rrr"
            eng.set_resume(False); call(eng, other, 5 * step, step, beta, tau); eng.set_resume(True)
        elif mid == "new_beta":
            beta2 = beta * 1.25
        elif mid == "seed":
            eng.seed(seed + 1)
        if fresh_second:
            eng.set_resume(False)
        b = call(eng, smp, n2, step, beta2, tau + (0.1 if mid == "new_beta" else 0.0))
        eng.set_resume(False)
        return a + b + [eng.get_config().s.copy(), np.asarray(eng.run_energy())]


kinds = ["rrg", "ea", "levels", "rrgn", "disc", "skn", "sk", "quant", "qeat"]
bad, t0 = 0, time.time()
for case in range(CASES):
    kind = kinds[int(rng.integers(len(kinds)))]
    smp = ["rrr", "bkl", "wtm", "eo", "std"][int(rng.integers(5))]
    seed = int(rng.integers(1, 1 << 30))
    X = graph(kind, seed)
    R = int(rng.choice([1, 40, 70, 200]))
    step = int(rng.choice([1, 5, 20]))
    n1, n2 = int(rng.integers(2, 30)) * step, int(rng.integers(2, 30)) * step
    beta, tau = float(rng.choice([0.7, 1.5, 2.0])), float(rng.choice([1.2, 1.8]))
    if rng.integers(2):
        mid, kindof = READS[int(rng.integers(len(READS)))], "read"
        ref_mid, ref_fresh = None, False                    # the same two calls with nothing in between
    else:
        # (standardMC's run is the tracked energy and the graph's cache: a new beta or seed does not end it, by its contract)
        ends = ENDS[:3] if smp == "std" else ENDS
        mid, kindof = ends[int(rng.integers(len(ends)))], "end"
        ref_mid, ref_fresh = mid, True                      # the same sequence with the second call's resume switched off
    try:
        got = scenario(X, R, smp, n1, n2, step, beta, tau, seed, mid, False)
        ref = scenario(X, R, smp, n1, n2, step, beta, tau, seed, ref_mid, ref_fresh)
    except pkg.RRRMCError as err:
        print(json.dumps({"case": case, "kind": kind, "smp": smp, "skipped": str(err)[:100]}), flush=True)
        continue
    same = len(got) == len(ref) and all(a.shape == b.shape and (a == b).all() for a, b in zip(got, ref))
    bad += 0 if same else 1
    rec = {"case": case, "kind": kind, "N": int(X.N), "smp": smp, "R": R, "step": step, "n": [n1, n2], "between": mid, "expect": kindof, "same": bool(same)}
    if not same:
        rec["differs"] = [i for i, (a, b) in enumerate(zip(got, ref)) if a.shape != b.shape or not (a == b).all()]
    print(json.dumps(rec), flush=True)
print(json.dumps({"cases": CASES, "mismatches": bad, "seconds": round(time.time() - t0, 1)}), flush=True)
sys.exit(1 if bad else 0)
