#!/usr/bin/env python3
"""Soak of the resumed sampler calls (include/rrrmc_hip.h: rrrmc_set_resume): random graphs, samplers, replica counts, kernel builds and CUT
POINTS — a run made in one call against the same run cut into resumed calls of random lengths (not multiples of `step`; standardMC: multiples, its contract), compared bit
for bit: samples, final configuration, tracked energy, counts, and the DeltaECache where the model has one.  GPU against GPU (the oracle
is not involved: tests/test_gpu_hooks.py pins both to it on fixed cases).

  python3 tests/soak/hook_soak.py [cases] [seed]        -> one line per case, a summary line at the end; exit code 1 on a mismatch"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as e  # noqa: E402

pkg = e.load_package()
CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20260)
BUILD_KEYS = ("RRRMC_RRR_NO_WAVE", "RRRMC_RRR_NO_LDS", "RRRMC_RRR_WAVE_SLACK", "RRRMC_CONT_NO_WAVE", "RRRMC_EO_NO_WAVE", "RRRMC_QUANT_NO_WAVE",
              "RRRMC_EO_NO_FTAU_LDS")


def graph(kind, seed):
    if kind == "rrg":
        return pkg.GraphRRG(int(rng.choice([10, 40, 150, 600])), int(rng.choice([3, 4])), seed=seed)
    if kind == "ea":
        L, D = [(2, 3), (4, 2), (3, 3), (6, 2)][int(rng.integers(4))]
        return pkg.GraphEA(L, D, seed=seed)
    if kind == "levels":
        return pkg.GraphRRG(int(rng.choice([10, 60])), 3, (-1, 0, 1), seed=seed)
    if kind == "rrgn":
        return pkg.GraphRRGNormal(int(rng.choice([64, 100, 400])), 3, seed=seed)
    if kind == "ean":
        return pkg.GraphEANormal(4, 2, seed=seed)
    if kind == "disc":
        return pkg.GraphRRGNormalDiscretized(int(rng.choice([20, 100])), 3, (-1, 0, 1), seed=seed)
    if kind == "skn":
        return pkg.GraphSKNormal(int(rng.choice([10, 24, 70])), seed=seed)
    if kind == "sk":
        return pkg.GraphSK(int(rng.choice([10, 33])), seed=seed)
    if kind == "quant":
        return pkg.GraphQuant(pkg.GraphRRG(int(rng.choice([10, 32])), 3, seed=seed), int(rng.choice([4, 8])), 0.5, 2.0)
    if kind == "qskt":
        return pkg.GraphQSKT(10, 4, 0.5, 2.0, seed=seed)
    if kind == "qeat":
        return pkg.GraphQEAT(4, 2, 4, 0.5, 2.0, seed=seed)
    raise ValueError(kind)


def call(eng, smp, n, step, beta, tau, thr):
    if smp == "rrr":
        Es, acc, st = eng.rrr_mc(beta, n, step, staged_thr=thr)
        return np.asarray(Es), np.stack([acc, st], 1).astype(np.float64)
    if smp == "bkl":
        Es, mv = eng.bkl_mc(beta, n, step)
        return np.asarray(Es), np.asarray(mv, np.float64)[:, None]
    if smp == "wtm":
        Es, mv, t = eng.wtm_mc(beta, n, float(step))
        return np.asarray(Es), np.asarray(mv, np.float64)[:, None]
    if smp == "std":
        Es, acc = eng.standard_mc(beta, n, step)
        return np.asarray(Es), np.asarray(acc, np.float64)[:, None]
    Es, Emin, Cmin, itmin = eng.extremal_opt(tau, n, step)
    return np.asarray(Es), np.concatenate([np.asarray(Emin, np.float64)[:, None], np.asarray(itmin, np.float64)[:, None], np.asarray(Cmin.s, np.float64)], 1)


def run(X, R, smp, pieces, step, beta, tau, thr, seed, env, shards=False):
    for k in BUILD_KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    with (pkg.Engine(X, R, devices=[0, 0]) if shards else pkg.Engine(X, R)) as eng:          # two shards of one multi-device context on this GPU
        eng.seed(seed)
        eng.init_spins_random()
        Es, cnt = [], None
        for i, n in enumerate(pieces):
            eng.set_resume(i > 0)
            a, c = call(eng, smp, n, step, beta, tau, thr)
            Es.append(a)
            cnt = c if (cnt is None or smp == "eo") else cnt + c          # counts are per call; extremal_opt's results are the run's
        eng.set_resume(False)
        out = [np.concatenate(Es, 1), cnt, eng.get_config().s.copy(), np.asarray(eng.run_energy())]
        if smp in ("rrr", "bkl") and X.model_kind in (1, 7) or (smp == "rrr" and X.model_kind in (3, 6)):
            out += list(eng.rrr_cache())
    return out


kinds = ["rrg", "ea", "levels", "rrgn", "ean", "disc", "skn", "sk", "quant", "qskt", "qeat"]
envs = [{}, {"RRRMC_RRR_NO_WAVE": "1"}, {"RRRMC_RRR_NO_WAVE": "1", "RRRMC_RRR_NO_LDS": "1"}, {"RRRMC_RRR_WAVE_SLACK": "8"}, {"RRRMC_CONT_NO_WAVE": "1"},
        {"RRRMC_EO_NO_WAVE": "1"}, {"RRRMC_QUANT_NO_WAVE": "1"}, {"RRRMC_EO_NO_FTAU_LDS": "1"}]
bad, t0, units = 0, time.time(), 0
for case in range(CASES):
    kind = kinds[int(rng.integers(len(kinds)))]
    smp = ["rrr", "bkl", "wtm", "eo", "std"][int(rng.integers(5))]
    seed = int(rng.integers(1, 1 << 30))
    X = graph(kind, seed)
    R = int(rng.choice([1, 2, 33, 64, 65, 130, 300]))
    step = int(rng.choice([1, 3, 7, 10, 50]))
    env = envs[int(rng.integers(len(envs)))]
    beta, tau = float(rng.choice([0.5, 1.0, 2.0])), float(rng.choice([1.2, 1.8]))
    thr = [None, 0.0, 1.0][int(rng.integers(3))] if smp == "rrr" else None
    if smp == "wtm":
        total = int(rng.integers(3, 12))                                     # samples
        pieces, left = [], total
        while left:
            n = int(rng.integers(1, left + 1)); pieces.append(n); left -= n
    elif smp == "std":                                                       # standardMC's contract: cuts at the hook points (multiples of `step`)
        nst = int(rng.integers(2, 40))
        total = nst * step
        cuts = sorted(set(int(c) * step for c in rng.integers(0, nst + 1, size=int(rng.integers(1, 5)))) - {0, total})
        pieces = [b - a for a, b in zip([0] + cuts, cuts + [total])]
    else:
        total = int(rng.integers(20, 1500))
        cuts = sorted(set(int(c) for c in rng.integers(0, total + 1, size=int(rng.integers(1, 6)))) - {0, total})
        pieces = [b - a for a, b in zip([0] + cuts, cuts + [total])]
    shards = bool(R >= 64 and rng.integers(4) == 0)
    try:
        one = run(X, R, smp, [total], step, beta, tau, thr, seed, env)
        cut = run(X, R, smp, pieces, step, beta, tau, thr, seed, env, shards)          # (and the sharded run against the single context)
    except pkg.RRRMCError as err:                                            # a combination the library refuses (both ways alike)
        print(json.dumps({"case": case, "kind": kind, "smp": smp, "skipped": str(err)[:100]}), flush=True)
        continue
    same = len(one) == len(cut) and all(a.shape == b.shape and (a == b).all() for a, b in zip(one, cut))
    units += R * total
    if not same:
        bad += 1
        names = ["Es", "counts", "config", "E", "cache_pos", "cache_sizes"]
        print(json.dumps({"case": case, "differs": [names[i] for i, (a, b) in enumerate(zip(one, cut)) if a.shape != b.shape or not (a == b).all()],
                          "shapes": [[list(a.shape), list(b.shape)] for a, b in zip(one, cut)][:2]}), flush=True)
        if one[0].shape == cut[0].shape:
            rr, cc = np.nonzero(one[0] != cut[0])
            if len(rr):
                print(json.dumps({"first_Es_diff": [int(rr[0]), int(cc[0])], "one": float(one[0][rr[0], cc[0]]), "cut": float(cut[0][rr[0], cc[0]]),
                                  "n_diff": int(len(rr))}), flush=True)
    print(json.dumps({"case": case, "kind": kind, "N": int(X.N), "smp": smp, "R": R, "step": step, "pieces": pieces, "beta": beta, "thr": thr,
                      "env": env, "shards": shards, "same": bool(same)}), flush=True)
print(json.dumps({"cases": CASES, "mismatches": bad, "replica_units": units, "seconds": round(time.time() - t0, 1)}), flush=True)
sys.exit(1 if bad else 0)
