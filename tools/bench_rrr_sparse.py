import sys, time, json
sys.path.insert(0, ".")
import __graft_entry__ as e
pkg = e.load_package()
X = pkg.GraphRRG(10000, 3, seed=0x5EED)
for R in (64, 1024, 16384):
    for name in ("rrr", "bkl", "wtm"):
        eng = pkg.Engine(X, R)
        eng.seed(1); eng.init_spins_random()
        iters = 200000 if R <= 1024 else 20000
        t0 = time.perf_counter()
        if name == "rrr": out = eng.rrr_mc(2.0, iters, iters // 4)
        elif name == "bkl": out = eng.bkl_mc(2.0, iters * 20, iters * 5)
        else: out = eng.wtm_mc(2.0, 4, step=iters / 10000 / 4 * 10)
        dt = time.perf_counter() - t0
        tot, sw, n = eng.last_timing()
        moves = float(out[1].mean())
        print(json.dumps({"sampler": name, "R": R, "kernel_ms": sw, "wall_s": dt, "moves_per_replica": moves, "moves_per_s_kernel": R * moves / (sw * 1e-3)}), flush=True)
        eng.close()
