"""standardMC on GraphRRG(N, K, +-J) across graph sizes, degrees and temperatures (GPU box): the headline kernel away from the
benchmark point.  8192 replicas; one JSON line per case with attempts/s end to end and the SURVEY.md §8d roofline fraction
(B = 1 + a(3 + 3K) bytes per attempt against 8 TB/s).  python tools/bench_sizes.py > profiles/rNN/size_sweep.jsonl"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e  # noqa: E402

pkg = e.load_package()
R, SEED = 8192, 0x5EED
CASES = [(128, 3, 1.0), (1024, 3, 1.0), (2048, 3, 1.0), (4096, 3, 0.5), (4096, 3, 1.0), (4096, 3, 2.0), (4096, 4, 1.0), (4096, 6, 1.0),
         (8192, 3, 1.0), (10000, 3, 1.0), (16384, 3, 1.0), (24000, 3, 1.0), (32000, 3, 1.0)]
if os.environ.get("RRRMC_SIZES"):          # e.g. RRRMC_SIZES="1024,3,1.0;4096,6,1.0"
    CASES = [(int(a), int(b), float(c)) for a, b, c in (x.split(",") for x in os.environ["RRRMC_SIZES"].split(";"))]
for N, K, beta in CASES:
    X = pkg.GraphRRG(N, K, seed=SEED)
    eng = pkg.Engine(X, R)
    eng.seed(SEED)
    eng.init_spins_random()
    iters, step = 1 << 21, N
    eng.standard_mc_async(beta, iters // 4, step); eng.sync()
    t0 = time.perf_counter()
    eng.standard_mc_async(beta, iters, step); eng.sync()
    dt = time.perf_counter() - t0
    tot, sw, n = eng.last_timing()
    Es, acc = eng.fetch_results()
    a = float(acc.mean()) / iters
    B = 1 + a * (3 + 3 * K)
    print(json.dumps({"N": N, "K": K, "beta": beta, "replicas": R, "iters": iters, "sample_step": step, "attempts_per_s": R * iters / dt,
                      "sweep_kernel_ms": sw, "acceptance": a, "bytes_per_attempt": B, "roofline_frac": B * R * iters / (sw * 1e-3) / 8e12,
                      "energy_per_spin": float(Es[:, -1].mean()) / N}), flush=True)
    eng.close()
