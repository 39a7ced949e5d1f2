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
nch = iters // 4096 * 5
for g in (0, 100, 255):
    out = np.zeros(16, np.uint64)
    L.rrrmc_debug_stamps(eng._ctx, g, out.ctypes.data)
    print("group", g, "busy cycles per chunk by wave:", (out / nch).astype(int).tolist())
print("sweep ms", sw, "-> per chunk us", sw * 1e3 / nch, " (~100MHz memtime? see ratio)")
