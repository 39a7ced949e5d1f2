#!/usr/bin/env python3
"""The reference's `scripts/scripts.jl:test_RRG` experiment on the MI355X engine: the four samplers of the paper
(Metropolis, BKL, RRR, waiting-time) on one GraphRRG(N, K) instance, a batch of replicas each, with the per-sample log files
(`#mctime acc E clocktime`), device-side configuration snapshots and the time-overlap analysis of `parseovs`.
With --cont the couplings are Gaussian (GraphRRGNormal) and the samplers use the continuous-energy caches: `test_RRGCont`.

  python examples/test_rrg.py [--N 10000] [--K 3] [--beta 2.0] [--samples 50] [--step 10000] [--replicas 64] [--out out_rrg]

Differences from the script: a batch of replicas runs in lockstep (one log / overlap curve per replica), randomness comes from
the engine's Philox streams, and the snapshots stay in HBM — the BitMatrix dump (`to_mat`) is written for replica 0 only.
Every sampler is called ONCE with the reference's `hook` keyword (the log / snapshot hook of scripts.jl:51-69), as the script does.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=10_000)          # scripts.jl:23
    ap.add_argument("--K", type=int, default=3)
    ap.add_argument("--beta", type=float, default=2.0)
    ap.add_argument("--samples", type=int, default=50)
    ap.add_argument("--step", type=int, default=10_000)
    ap.add_argument("--replicas", type=int, default=64)
    ap.add_argument("--seedx", type=int, default=8370000274)  # graph seed, scripts.jl:28
    ap.add_argument("--seed", type=int, default=6540000789)   # sampler seed, scripts.jl:29
    ap.add_argument("--out", default="output_RRG")
    ap.add_argument("--cont", action="store_true", help="Gaussian couplings: GraphRRGNormal, scripts.jl:152-281 test_RRGCont")
    # work per sample of the other samplers relative to RRR, scripts.jl:34-37 (beta = 2)
    ap.add_argument("--met-factor", type=float, default=3.7)
    ap.add_argument("--bkl-factor", type=float, default=94.9)
    ap.add_argument("--wtm-factor", type=float, default=53.0)
    args = ap.parse_args(argv)

    pkg = entry.load_package()
    os.makedirs(args.out, exist_ok=True)
    X = (pkg.GraphRRGNormal if args.cont else pkg.GraphRRG)(args.N, args.K, seed=args.seedx)
    R = args.replicas
    curves = {}
    for alg in ("met", "bkl", "rrr", "wtm"):
        with pkg.Engine(X, R) as eng:
            eng.seed(args.seed)
            eng.init_spins_random()
            log = pkg.SnapshotLog(eng, args.samples, prefix=os.path.join(args.out, "output_%s_sx%d_s%d" % (alg, args.seedx, args.seed)))
            t0 = time.time()
            # the reference's calls, hook included (scripts.jl:90-148): one run per sampler, the hook called at every sample — the library cuts
            # the run there and resumes it, so this is ONE chain per replica, as in the reference
            kw = dict(seed=args.seed, hook=log, engine=eng, quiet=True)
            if alg == "met":
                n = round(args.step * args.met_factor)
                pkg.standardMC(X, args.beta, n * args.samples, step=n, **kw)
            elif alg == "bkl":
                n = round(args.step * args.bkl_factor)
                pkg.bklMC(X, args.beta, n * args.samples, step=n, **kw)
            elif alg == "rrr":
                pkg.rrrMC(X, args.beta, args.step * args.samples, step=args.step, **kw)
            else:
                pkg.wtmMC(X, args.beta, args.samples, step=args.step * args.wtm_factor, **kw)        # rtstep, scripts.jl:134-137
            log.close()
            wall = time.time() - t0
            mq2, sq2 = pkg.parseovs(eng, log.clock, pkg.log_range(log.clock[0], log.clock[-1], st0=log.clock[0]))
            curves[alg] = mq2
            cols, chunks = log.to_mat(0)
            np.save(os.path.join(args.out, "Cs_%s_sx%d_s%d.npy" % (alg, args.seedx, args.seed)), chunks)
            E = eng.energy()
            last = mq2[-1][~np.isnan(mq2[-1])] if len(mq2) else np.zeros(0)
            print("%-3s  %6.2f s   <E>/N = %+.4f   <q^2> over the last window = %s" % (
                alg, wall, E.mean() / args.N, ("%.4f" % last.mean()) if last.size else "n/a"))
    return curves


if __name__ == "__main__":
    main()
