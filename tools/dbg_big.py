"""Timing probe for the big-N random-site path: kernel time against the number of sweeps (fixed cost vs per-chunk cost)."""
import faulthandler, sys, time, os
faulthandler.dump_traceback_later(50, repeat=False, file=sys.stderr)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
R = int(sys.argv[1])
X = pkg.GraphEA(64, 3, seed=0x5EED)
eng = pkg.Engine(X, R)
eng.seed(1); eng.init_spins_random()
eng.standard_mc_async(1.0, X.N, X.N); eng.sync()
for sw in (1, 2, 4, 8, 16):
    t0 = time.perf_counter()
    eng.standard_mc_async(1.0, sw * X.N, X.N); eng.sync()
    dt = time.perf_counter() - t0
    print("sweeps", sw, "wall_ms %.3f" % (dt * 1e3), "timing", eng.last_timing(), flush=True)
eng.close()
