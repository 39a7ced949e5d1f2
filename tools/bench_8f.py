#!/usr/bin/env python3
"""Evidence runs for the thread-per-replica kernels of SURVEY.md §8f (one launch of each at the reference's experiment sizes,
scripts/scripts.jl:23 test_RRG N = 10^4 K = 3, :152 test_RRGCont, :766 test_QIsing), meant to run under rocprofv3:

  rocprofv3 --kernel-trace --stats -d out -- python3 tools/bench_8f.py [R]

Prints one JSON line per sampler (kernel ms, moves or iterations per second).  No new kernels: measurements only."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e  # noqa: E402

pkg = e.load_package()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096          # enough replicas that the thread-per-replica builds (not the wave builds) run
SEED = 0x5EED


def report(name, kernel, eng, t0, units, what):
    tot, sw, n = eng.last_timing()
    print(json.dumps({"sampler": name, "kernel": kernel, "replicas": R, "kernel_ms": sw, "wall_s": time.perf_counter() - t0,
                      what + "_per_s_kernel": units / (sw * 1e-3)}), flush=True)


X = pkg.GraphRRG(10000, 3, seed=SEED)                         # scripts.jl:23
with pkg.Engine(X, R) as eng:
    eng.seed(1); eng.init_spins_random()
    eng.standard_mc(2.0, 200000, 200000)                      # a short quench first, as the reference's runs start from random spins too
    t0 = time.perf_counter(); out = eng.rrr_mc(2.0, 20000, 5000)
    report("rrrMC GraphRRG(1e4,3) beta=2", "rrr_sparse_kernel", eng, t0, R * 20000, "iterations")
    t0 = time.perf_counter(); out = eng.bkl_mc(2.0, 400000, 100000)
    report("bklMC GraphRRG(1e4,3) beta=2", "rrr_sparse_kernel (bkl)", eng, t0, R * float(out[1].mean()), "moves")
    t0 = time.perf_counter(); out = eng.wtm_mc(2.0, 4, step=0.5)
    report("wtmMC GraphRRG(1e4,3) beta=2", "wtm_sparse_kernel", eng, t0, R * float(out[1].mean()), "moves")
    t0 = time.perf_counter(); out = eng.extremal_opt(1.3, 20000, 5000)
    report("extremal_opt GraphRRG(1e4,3) tau=1.3", "eo_sparse_kernel", eng, t0, R * 20000, "iterations")

Xn = pkg.GraphRRGNormal(10000, 3, seed=SEED)                  # scripts.jl:152 (test_RRGCont)
with pkg.Engine(Xn, R) as eng:
    eng.seed(1); eng.init_spins_random()
    eng.standard_mc(2.0, 100000, 100000)
    t0 = time.perf_counter(); out = eng.rrr_mc(2.0, 5000, 2500)
    report("rrrMC GraphRRGNormal(1e4,3) beta=2", "cont_wave_kernel (rrr)", eng, t0, R * 5000, "iterations")
    t0 = time.perf_counter(); out = eng.bkl_mc(2.0, 100000, 50000)
    report("bklMC GraphRRGNormal(1e4,3) beta=2", "cont_wave_kernel (bkl)", eng, t0, R * float(out[1].mean()), "moves")

Xd = pkg.GraphRRGNormalDiscretized(10000, 3, (-1, 0, 1), seed=SEED)     # the model family rrrMC(DoubleGraph) was designed for
with pkg.Engine(Xd, R) as eng:
    eng.seed(1); eng.init_spins_random()
    t0 = time.perf_counter(); out = eng.rrr_mc(2.0, 20000, 5000)
    report("rrrMC GraphRRGNormalDiscretized(1e4,3) beta=2", "rrr_dbl_kernel", eng, t0, R * 20000, "iterations")
