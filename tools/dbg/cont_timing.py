"""One-off (GPU box): where the time of cont_sparse_kernel (rrrMC on GraphRRGNormal(10^4, 3), scripts/scripts.jl:152) goes — per-call set-up
(energy, DeltaECacheCont, refresh!) against per-iteration cost, by calls of different lengths.  python tools/dbg/cont_timing.py [R]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as e
pkg = e.load_package()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Xn = pkg.GraphRRGNormal(10000, 3, seed=0x5EED)
with pkg.Engine(Xn, R) as eng:
    eng.seed(1); eng.init_spins_random()
    eng.standard_mc(2.0, 100000, 100000)
    for iters in (1, 1000, 2500, 5000, 10000):
        out = eng.rrr_mc(2.0, iters, max(iters, 1))
        tot, sw, n = eng.last_timing()
        print(json.dumps({"mode": "rrr", "replicas": R, "iters": iters, "kernel_ms": sw, "acc": float(out[1].mean()) / iters}), flush=True)
    for iters in (1000, 100000):
        out = eng.bkl_mc(2.0, iters, iters)
        tot, sw, n = eng.last_timing()
        print(json.dumps({"mode": "bkl", "replicas": R, "iters": iters, "kernel_ms": sw, "moves": float(out[1].mean())}), flush=True)
