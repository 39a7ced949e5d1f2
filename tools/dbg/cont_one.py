"""One-off (GPU box): one rrrMC call of the continuous sampler on GraphRRGNormal(10^4, 3) for rocprofv3.  python tools/dbg/cont_one.py [R] [iters]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as e
pkg = e.load_package()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
Xn = pkg.GraphRRGNormal(10000, 3, seed=0x5EED)
with pkg.Engine(Xn, R) as eng:
    eng.seed(1); eng.init_spins_random()
    eng.standard_mc(2.0, 100000, 100000)
    out = eng.rrr_mc(2.0, iters, iters)
    tot, sw, n = eng.last_timing()
    print(json.dumps({"replicas": R, "iters": iters, "kernel_ms": sw, "iterations_per_s": R * iters / (sw * 1e-3)}))
