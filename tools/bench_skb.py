"""Binary GraphSK under standardMC (GPU box): python tools/bench_skb.py [N] [R]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
R = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
X = pkg.GraphSK(N, seed=0x5EED)
eng = pkg.Engine(X, R)
eng.seed(0x5EED); eng.init_spins_random()
iters = 1 << 16
eng.standard_mc_async(1.0, iters // 4, 1 << 10); eng.sync()
t0 = time.perf_counter()
eng.standard_mc_async(1.0, iters, 1 << 10); eng.sync()
dt = time.perf_counter() - t0
tot, sw, nl = eng.last_timing()
Es, acc = eng.fetch_results()
print(json.dumps({"model": "GraphSK (binary)", "N": N, "replicas": R, "attempts_per_s": R * iters / dt, "kernel_ms": sw, "acceptance": float(acc.mean()) / iters}))
eng.close()
