#!/usr/bin/env python3
"""The reference's quantum experiment `scripts/scripts.jl:test_QIsing` (:766-864) on the MI355X engine: Metropolis and RRR on one
GraphQSKT(N, M, Γ, β) instance — a transverse-field SK model, M Suzuki-Trotter slices of one binary SK disorder — a batch of
replicas each, logging `#mctime acc QE clocktime` with QE = Qenergy(X, C) (src/graphs/QT.jl:253-268) and keeping the sampled
configurations as device snapshots.

  python examples/test_qising.py [--N 1024] [--M 16] [--beta 2.0] [--Gamma 0.3] [--samples 50] [--step 10000] [--replicas 128]

As in the script the Metropolis leg does `met_factor` (15.74) iterations per RRR iteration so that both spend similar time per
sample; randomness comes from the engine's Philox streams, the BitMatrix dump (`to_mat`) is written for replica 0 only.
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
    ap.add_argument("--N", type=int, default=1024)             # scripts.jl:766-770
    ap.add_argument("--M", type=int, default=16)
    ap.add_argument("--beta", type=float, default=2.0)
    ap.add_argument("--Gamma", type=float, default=0.3)
    ap.add_argument("--samples", type=int, default=50)
    ap.add_argument("--step", type=int, default=10_000)
    ap.add_argument("--replicas", type=int, default=128)
    ap.add_argument("--seedx", type=int, default=8370000274)   # graph seed, scripts.jl:772
    ap.add_argument("--seed", type=int, default=6540000789)    # sampler seed, scripts.jl:773
    ap.add_argument("--met-factor", type=float, default=15.74)  # scripts.jl:778
    ap.add_argument("--rrr-factor", type=float, default=1.0)
    ap.add_argument("--algs", default="met,rrr")
    ap.add_argument("--out", default=None)
    args = ap.parse_args(argv)

    pkg = entry.load_package()
    out = args.out or "output_QIsing_N%d_M%d_beta%s_Gamma%s_step%d" % (args.N, args.M, args.beta, args.Gamma, args.step)
    os.makedirs(out, exist_ok=True)
    X = pkg.GraphQSKT(args.N, args.M, args.Gamma, args.beta, seed=args.seedx)
    R = args.replicas
    summary = {}
    for alg in args.algs.split(","):
        assert alg in ("met", "rrr")                           # scripts.jl:785
        rstep = round(args.step * (args.met_factor if alg == "met" else args.rrr_factor))
        with pkg.Engine(X, R) as eng:
            eng.seed(args.seed)
            eng.init_spins_random()
            log = pkg.SnapshotLog(eng, args.samples, prefix=os.path.join(out, "output_%s_sx%d_s%d" % (alg, args.seedx, args.seed)),
                                  energy_label="QE")
            t0 = time.time()
            last = {}

            def hook(it, X_, C, acc, E):                       # the hook of scripts.jl:803-809: QE instead of E
                QE, tmag, _ = eng.quant_observables()          # (reads the live configuration; does not disturb the resumed run)
                last.update(QE=QE, tmag=tmag, acc=np.array(acc))
                return log(it, X_, C, acc, QE)

            # one call per sampler with the reference's hook keyword (scripts.jl:815-835): the library cuts the run at the hook points and resumes it
            fn = pkg.standardMC if alg == "met" else pkg.rrrMC
            fn(X, args.beta, rstep * args.samples, step=rstep, seed=args.seed, hook=hook, engine=eng, quiet=True)
            QE, tmag, accepted = last["QE"], last["tmag"], last["acc"]
            log.close()
            wall = time.time() - t0
            cols, chunks = log.to_mat(0)
            np.save(os.path.join(out, "Cs_%s_sx%d_s%d.npy" % (alg, args.seedx, args.seed)), chunks)
            summary[alg] = (wall, float(QE.mean()), float(tmag.mean()), float(accepted.mean()) / (args.samples * rstep))
            print("%-3s  %7.2f s   <QE> = %+.5f   <transverse mag> = %.5f   acceptance = %.4f   (%d replicas x %d samples x %d iterations)" % (
                alg, wall, summary[alg][1], summary[alg][2], summary[alg][3], R, args.samples, rstep))
    return summary


if __name__ == "__main__":
    main()
