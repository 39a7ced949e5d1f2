"""rrrMC / standardMC on GraphQSKT(N = 1024, M = 16, beta = 2, Gamma = 0.3) — the reference's test_QIsing geometry (scripts.jl:766-775)."""
import json
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import __graft_entry__ as e  # noqa: E402

pkg = e.load_package()
X = pkg.GraphQSKT(1024, 16, 0.3, 2.0, seed=8370000274)
for R in (128, 1024, 4096):
    with pkg.Engine(X, R) as eng:
        eng.seed(6540000789)
        eng.init_spins_random()
        eng.rrr_mc(2.0, 2000, 1000)
        for name, iters in (("rrrMC", 1 << 17), ("standardMC", 1 << 19)):
            t0 = time.perf_counter()
            out = eng.rrr_mc(2.0, iters, 1 << 12) if name == "rrrMC" else eng.standard_mc(2.0, iters, 1 << 12)
            dt = time.perf_counter() - t0
            print(json.dumps({"model": "GraphQSKT(1024, 16, 0.3, 2.0)", "sampler": name, "replicas": R, "iters_per_replica": iters,
                              "iterations_per_s": R * iters / dt, "acceptance": float(out[1].mean()) / iters, "wall_ms": dt * 1e3}), flush=True)
