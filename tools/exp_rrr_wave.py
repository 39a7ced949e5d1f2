#!/usr/bin/env python3
"""rrrMC(X::SingleGraph) on GraphRRG(10^4, 3) (scripts/scripts.jl:23) — the thread-per-replica kernel against
rrr_sparse_wave_kernel forced beyond its default replica limit (RRRMC_RRR_WAVE_MAX_R), at several replica counts.  Measurements only:

  python3 tools/exp_rrr_wave.py [iters]        -> one JSON line per (replicas, build)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e  # noqa: E402

pkg = e.load_package()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
X = pkg.GraphRRG(10000, 3, seed=0x5EED)
for R in (256, 1024, 4096):
    ref = None
    for build, env in (("default", {}), ("wave", {"RRRMC_RRR_WAVE_MAX_R": "1000000"}), ("thread", {"RRRMC_RRR_NO_WAVE": "1"})):
        for k in ("RRRMC_RRR_WAVE_MAX_R", "RRRMC_RRR_NO_WAVE"):
            os.environ.pop(k, None)
        os.environ.update(env)
        with pkg.Engine(X, R) as eng:
            eng.seed(1); eng.init_spins_random()
            eng.standard_mc(2.0, 200000, 200000, want_energies=False)          # a short quench, as bench.py's f8_rrr_rrg_1e4
            eng.rrr_mc(2.0, iters // 4, iters, want_energies=False)
            t0 = time.perf_counter()
            Es, acc, st = eng.rrr_mc(2.0, iters, iters // 4)
            wall = time.perf_counter() - t0
            _, k_ms, _ = eng.last_timing()
            sig = (Es.tobytes(), acc.tobytes(), eng.get_config().s.tobytes())
        same = ref is None or sig == ref
        ref = ref or sig
        print(json.dumps({"replicas": R, "build": build, "kernel_ms": round(k_ms, 3), "iterations_per_s_kernel": R * iters / (k_ms * 1e-3),
                          "iterations_per_s_wall": R * iters / wall, "same_chains_as_default": same}), flush=True)
