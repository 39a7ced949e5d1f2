#!/usr/bin/env python3
"""Soak of standardMC (src/RRRMC.jl:81-127) across the model kinds against the ORACLE on random shapes: general-level GraphRRG / GraphEA
(lev_standard_kernel), GraphRRGNormal / GraphEANormal (spf_team_kernel through its team widths and the single-wavefront kernel, K up to 8),
the discretised DoubleGraphs (dbl_standard_kernel), GraphSKNormal / GraphSK (sk_hblock_kernel builds), GraphQuant over +-J, binary-SK and
Gaussian-SK slices (quant_standard_kernel) — energies, final configuration and accepted count of a few replicas, then a SECOND call that
continues the streams.

  python3 tests/soak/std_family_soak.py [cases] [seed]"""
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
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 6161)
KEYS = ("RRRMC_SPF_TEAM", "RRRMC_SPF_TEAM_WAVES", "RRRMC_SPF_TEAM_WIDTH", "RRRMC_SK_BLOCK_V1", "RRRMC_SK_LEGACY", "RRRMC_QUANT_NO_WAVE")
LEVS = [(-1, 0, 1), (-2, -1, 1, 2), (-1.5, -0.5, 0.5, 1.5)]


def lattice():
    return [(2, 3), (4, 2), (3, 3), (6, 2), (2, 4), (3, 4)][int(rng.integers(6))]


bad, t0 = 0, time.time()
for case in range(CASES):
    seed = int(rng.integers(1, 1 << 30))
    kind = ["lev", "spf", "dbl", "skn", "sk", "quant", "qsk", "qskn"][int(rng.integers(8))]
    env = {}
    try:
        if kind == "lev":
            lev = LEVS[int(rng.integers(len(LEVS)))]
            if rng.integers(2):
                L, D = lattice(); X, form = pkg.GraphEA(L, D, lev, seed=seed), "ea"
            else:
                K = int(rng.choice([3, 4, 5])); N = int(rng.choice([20, 130, 700])); N += (N * K) % 2
                X, form = pkg.GraphRRG(N, K, lev, seed=seed), "rrg"
            units, mul, div = pkg.level_units(lev)
            ref_fn = lambda C, r, it, n, st, it0: O.standard_mc_lev(X.A, X.J, beta, n, st, seed, C, it0=it0, replica=r, form=form, mul=mul, div=div)
            val = X.energy_value
        elif kind == "spf":
            if rng.integers(2):
                L, D = lattice(); X, form = pkg.GraphEANormal(L, D, seed=seed), "ea"
            else:
                K = int(rng.choice([3, 4, 5, 6, 7])); N = int(rng.choice([64, 300, 1000])); N += (N * K) % 2
                X, form = pkg.GraphRRGNormal(N, K, seed=seed), "rrg"
            env = [{}, {"RRRMC_SPF_TEAM_WAVES": "8"}, {"RRRMC_SPF_TEAM_WIDTH": "32"}, {"RRRMC_SPF_TEAM_WIDTH": "16"}, {"RRRMC_SPF_TEAM_WAVES": "16", "RRRMC_SPF_TEAM_WIDTH": "64"},
                   {"RRRMC_SPF_TEAM": "0"}][int(rng.integers(6))]
            ref_fn = lambda C, r, it, n, st, it0: O.standard_mc_spf(X.A, X.J, beta, n, st, seed, C, it0=it0, replica=r, form=form)[:3]
            val = lambda x: x
        elif kind == "dbl":
            lev = [(-1, 0, 1), (-2, -1, 1, 2), (-0.8, 0.0, 0.8)][int(rng.integers(3))]
            if rng.integers(2):
                L, D = [(2, 3), (4, 2), (3, 3)][int(rng.integers(3))]; X, form = pkg.GraphEANormalDiscretized(L, D, lev, seed=seed), "ea"
            else:
                X, form = pkg.GraphRRGNormalDiscretized(int(rng.choice([20, 100, 400])), int(rng.choice([3, 4])), lev, seed=seed), "rrg"
            units, mul, div = O.dfloat_units(lev)
            ref_fn = lambda C, r, it, n, st, it0: O.standard_mc_dbl(X.A, X.dJ, X.rJ, beta, n, st, seed, C, it0=it0, replica=r, form=form, mul=mul, div=div)
            val = lambda x: x
        elif kind in ("skn", "sk"):
            N = int(rng.choice([10, 65, 200, 513]))
            X = pkg.GraphSKNormal(N, seed=seed) if kind == "skn" else pkg.GraphSK(N, seed=seed)
            env = [{}, {}, {"RRRMC_SK_BLOCK_V1": "1"}, {"RRRMC_SK_LEGACY": "1"}][int(rng.integers(4))]
            f = O.standard_mc_skn if kind == "skn" else O.standard_mc_skb
            ref_fn = lambda C, r, it, n, st, it0: f(X.J, beta, n, st, seed, C, it0=it0, replica=r)[:3]
            val = lambda x: x
        else:
            Mq, Gam = int(rng.choice([3, 4, 8])), float(rng.choice([0.3, 0.5, 1.0]))
            env = [{}, {"RRRMC_QUANT_NO_WAVE": "1"}][int(rng.integers(2))]
            if kind == "quant":
                X1 = pkg.GraphRRG(int(rng.choice([10, 32, 64])), 3, seed=seed) if rng.integers(2) else pkg.GraphEA(*[(2, 3), (4, 2)][int(rng.integers(2))], seed=seed)
                X = pkg.GraphQuant(X1, Mq, Gam, 2.0)
                A_, J_ = X1.A, X1.J.astype(np.int32)
                ref_fn = lambda C, r, it, n, st, it0: O.standard_mc_quant(A_, J_, Mq, X.fourK, beta, n, st, seed, C, it0=it0, replica=r)[:3]
            elif kind == "qsk":
                Nk = int(rng.choice([10, 33]))
                X = pkg.GraphQSKT(Nk, Mq, Gam, 2.0, seed=seed)
                ref_fn = lambda C, r, it, n, st, it0: O.standard_mc_quant_sk(X.X1.J, Nk, Mq, X.fourK, beta, n, st, seed, C, it0=it0, replica=r)[:3]
            else:
                Nk = int(rng.choice([10, 24]))
                X = pkg.GraphQSKNormalT(Nk, Mq, Gam, 2.0, seed=seed)
                ref_fn = lambda C, r, it, n, st, it0: O.standard_mc_quant_skn(X.X1.J, Nk, Mq, X.fourK, beta, n, st, seed, C, it0=it0, replica=r)[:3]
            val = lambda x: x
    except pkg.RRRMCError as err:
        print(json.dumps({"case": case, "kind": kind, "skipped": str(err)[:100]}), flush=True)
        continue
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    R = int(rng.choice([3, 70, 200]))
    beta = 2.0 if kind in ("quant", "qsk", "qskn") else float(rng.choice([0.5, 1.0, 2.0]))
    step = int(rng.choice([1, 7, 64, 500]))
    iters = int(rng.integers(2, 30)) * step + int(rng.integers(0, step))
    iters2 = int(rng.integers(1, 10)) * step
    try:
        with pkg.Engine(X, R) as eng:
            eng.seed(seed); eng.init_spins_random()
            C0 = eng.get_config().s.copy()
            Es, acc = eng.standard_mc(beta, iters, step)
            C1 = eng.get_config().s.copy()
            Es2, acc2 = eng.standard_mc(beta, iters2, step)
            C2 = eng.get_config().s.copy()
    except pkg.RRRMCError as err:
        print(json.dumps({"case": case, "kind": kind, "skipped": str(err)[:100]}), flush=True)
        continue
    ok = True
    for r in sorted(set([0, R - 1, int(rng.integers(R))])):
        a = ref_fn(C0[r], r, 0, iters, step, 0)
        ok &= bool((np.asarray(Es)[r] == val(np.asarray(a[0]))).all() and (C1[r] == a[1]).all() and acc[r] == a[2])
        b = ref_fn(a[1], r, 0, iters2, step, iters)
        ok &= bool((np.asarray(Es2)[r] == val(np.asarray(b[0]))).all() and (C2[r] == b[1]).all() and acc2[r] == b[2])
    bad += 0 if ok else 1
    print(json.dumps({"case": case, "kind": kind, "graph": type(X).__name__, "N": int(X.N), "K": int(getattr(X, "K", 0)), "R": R, "beta": beta, "iters": [iters, iters2], "step": step,
                      "env": env, "same": ok}), flush=True)
print(json.dumps({"cases": CASES, "mismatches": bad, "seconds": round(time.time() - t0, 1)}), flush=True)
sys.exit(1 if bad else 0)
