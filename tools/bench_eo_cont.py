import sys, time
sys.path.insert(0, '/root/repo')
import __graft_entry__ as e
pkg = e.load_package()
for N, R, iters in [(1000, 4096, 20000), (10000, 1024, 4000)]:
    X = pkg.GraphRRGNormal(N, 3, seed=5)
    with pkg.Engine(X, R) as eng:
        eng.seed(5); eng.init_spins_random()
        eng.extremal_opt(1.3, 200, 100)
        t0 = time.perf_counter()
        eng.extremal_opt(1.3, iters, iters)
        dt = time.perf_counter() - t0
        print("EO cont RRGNormal N=%d R=%d: %.3e moves/s (%.1f ms)" % (N, R, R * iters / dt, dt * 1e3), flush=True)
for name, X, R, iters in [("GraphSKNormal(1024)", pkg.GraphSKNormal(1024, seed=5), 2048, 2000), ("GraphSK(1024)", pkg.GraphSK(1024, seed=5), 2048, 2000)]:
    with pkg.Engine(X, R) as eng:
        eng.seed(5); eng.init_spins_random()
        eng.extremal_opt(1.3, 100, 100)
        t0 = time.perf_counter()
        eng.extremal_opt(1.3, iters, iters)
        dt = time.perf_counter() - t0
        print("EO cont %s R=%d: %.3e moves/s (%.1f ms)" % (name, R, R * iters / dt, dt * 1e3), flush=True)
