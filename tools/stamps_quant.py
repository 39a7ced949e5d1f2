"""Diagnostic (GPU box): per-segment shader cycles of one iteration of rrr_quant_wave_kernel (replica 0), from a -DRRRMC_STAMPS build
(tools/ablate/zz_stamps.so).  Segments: 0 class + member pick, 1 lane-parallel re-classification + broadcasts, 2 T/z updates + the
three set moves, 3 accept test, 4 undo (rejected moves only), 5 bookkeeping tail."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RRRMC_HIP_LIB"] = os.path.abspath("tools/ablate/zz_stamps.so")
import __graft_entry__ as e
import numpy as np
pkg = e.load_package()
X = pkg.GraphQuant(pkg.GraphRRG(1024, 3, seed=0x5EED), 32, 0.5, 2.0)
eng = pkg.Engine(X, 128)
eng.seed(0x5EED); eng.init_spins_random()
iters = 1 << 16
Es, acc, st = eng.rrr_mc(2.0, iters, 4096)
tot, sw, n = eng.last_timing()
L = pkg.lib()
L.rrrmc_debug_stamps.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
out = np.zeros(16, np.uint64)
L.rrrmc_debug_stamps(eng._ctx, 0, out.ctypes.data)
per = out[:8].astype(np.float64) / iters
print("kernel ms %.2f -> %.0f ns per iteration; acceptance %.3f" % (sw, sw * 1e6 / iters, acc.mean() / iters))
print("cycles per iteration by segment:", [int(x) for x in per], "sum", int(per.sum()))
