"""Diagnostic (GPU box): sweep-kernel time per chunk as a function of the sample step (all-full vs ragged chunks).
Set CH to the chunk length the context will choose (for the printed per-chunk figures only)."""
import sys, os, json
sys.path.insert(0, os.getcwd())
import __graft_entry__ as e
pkg = e.load_package()
X = pkg.GraphRRG(4096, 3, seed=0x5EED)
eng = pkg.Engine(X, 8192)
eng.seed(0x5EED); eng.init_spins_random()
C = int(os.environ.get("CH", "832"))
for step in (C * 5, 4096, C, C * 20):
    iters = step * (1 << 20) // step
    eng.standard_mc_async(1.0, iters, step); eng.sync()
    eng.standard_mc_async(1.0, iters, step); eng.sync()
    tot, sw, n = eng.last_timing()
    nchunks = (iters // step) * ((step + C - 1) // C)
    print("C=%d step=%d iters=%d sweep_ms=%.3f  ns/slot=%.3f  us/chunk=%.3f" % (C, step, iters, sw, sw * 1e6 / iters, sw * 1e3 / nchunks))
