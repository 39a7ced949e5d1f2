#!/usr/bin/env python3
"""rrrMC(X::SingleGraph) on GraphRRG(10^4, 3): threads (= replicas) per workgroup of the thread-per-replica kernel (RRRMC_RRR_TPB overrides
rrr_tpb's choice), at several replica counts.  Measurements only:

  python3 tools/exp_rrr_tpb.py [iters]        -> one JSON line per (replicas, threads per workgroup)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e  # noqa: E402

pkg = e.load_package()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
X = pkg.GraphRRG(10000, 3, seed=0x5EED)
os.environ["RRRMC_RRR_NO_WAVE"] = "1"
for R in (512, 1024, 4096, 16384):
    ref = None
    for tpb in (0, 1, 2, 4, 8, 16, 32, 64):
        os.environ.pop("RRRMC_RRR_TPB", None)
        if tpb:
            os.environ["RRRMC_RRR_TPB"] = str(tpb)
        if tpb == 1 and R > 1024:
            continue
        with pkg.Engine(X, R) as eng:
            eng.seed(1); eng.init_spins_random()
            eng.standard_mc(2.0, 200000, 200000, want_energies=False)
            eng.rrr_mc(2.0, iters // 4, iters, want_energies=False)
            Es, acc, st = eng.rrr_mc(2.0, iters, iters // 4)
            _, k_ms, _ = eng.last_timing()
            sig = (Es.tobytes(), acc.tobytes(), eng.get_config().s.tobytes())
        same = ref is None or sig == ref
        ref = ref or sig
        print(json.dumps({"replicas": R, "tpb": tpb or "default", "kernel_ms": round(k_ms, 3), "iterations_per_s_kernel": round(R * iters / (k_ms * 1e-3)),
                          "same_chains": same}), flush=True)
