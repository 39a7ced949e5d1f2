"""Diagnostic (GPU box): per-wave busy cycles of the sweep kernel, from a -DRRRMC_STAMPS build."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RRRMC_HIP_LIB"] = os.path.abspath("tools/ablate/zz_stamps.so")
import __graft_entry__ as e
import numpy as np
pkg = e.load_package()
X = pkg.GraphRRG(4096, 3, seed=0x5EED)
eng = pkg.Engine(X, 8192)
eng.seed(0x5EED); eng.init_spins_random()
iters = 1 << 20
eng.standard_mc_async(1.0, iters, 4096); eng.sync()
eng.standard_mc_async(1.0, iters, 4096); eng.sync()
tot, sw, n = eng.last_timing()
L = pkg.lib()
L.rrrmc_debug_stamps.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
nch = iters // 4096 * 3          # chunks per launch: 4096 iterations between samples = 3 balanced chunks at C = 1472
for g in (0, 100, 255):
    out = np.zeros(16, np.uint64)
    L.rrrmc_debug_stamps(eng._ctx, g, out.ctypes.data)
    print("group", g, "busy cycles per chunk by wave:", (out / nch).astype(int).tolist())
print("sweep ms", sw, "-> per chunk us", sw * 1e3 / nch, " (~100MHz memtime? see ratio)")

tr = np.zeros((4096, 16), np.uint64)
L.rrrmc_debug_step_trace.argtypes = [C.c_void_p, C.c_void_p]
L.rrrmc_debug_step_trace(eng._ctx, tr.ctypes.data)
tr = tr[2:1200].astype(np.float64)
cons, tal, prod = tr[:, 0], tr[:, 1], tr[:, 2:]
pmax = prod.max(axis=1)
step = np.maximum(np.maximum(cons, tal), pmax)
print("per-step means: consumer %.0f tally %.0f producer-mean %.0f producer-max %.0f  max-of-all %.0f" % (cons.mean(), tal.mean(), prod.mean(), pmax.mean(), step.mean()))
print("consumer slowest in %.0f%% of steps; tally slowest in %.0f%%" % (100 * (cons >= step).mean(), 100 * (tal >= step).mean()))
print("consumer pct 10/50/90: %s   producer-max pct: %s  tally pct: %s" % (np.percentile(cons, [10, 50, 90]).astype(int), np.percentile(pmax, [10, 50, 90]).astype(int), np.percentile(tal, [10, 50, 90]).astype(int)))
