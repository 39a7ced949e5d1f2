#!/usr/bin/env python3
"""Secondary benchmarks (not the headline metric): throughput of the other model kernels on one GPU.
  python tools/bench_models.py sk      GraphSKNormal N=1024, 2048 replicas (BASELINE.json configs[2])
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402


def bench_sk(N=1024, R=2048, beta=1.0, iters=1 << 16, step=1 << 10, seed=0x5EED):
    if len(sys.argv) > 2:                                    # python tools/bench_models.py sk N [R] [binary]
        N = int(sys.argv[2]); R = int(sys.argv[3]) if len(sys.argv) > 3 else R
    binary = len(sys.argv) > 4 and sys.argv[4] == "binary"
    pkg = entry.load_package()
    X = pkg.GraphSK(N, seed=seed) if binary else pkg.GraphSKNormal(N, seed=seed)
    eng = pkg.Engine(X, R)
    eng.seed(seed)
    eng.init_spins_random()
    eng.standard_mc_async(beta, iters // 4, step); eng.sync()
    t0 = time.perf_counter()
    eng.standard_mc_async(beta, iters, step); eng.sync()
    dt = time.perf_counter() - t0
    total_ms, sweep_ms, _ = eng.last_timing()
    Es, acc = eng.fetch_results()
    a = float(acc.mean()) / iters
    attempts = float(R) * iters
    bytes_per_attempt = 8 + a * (17 * N + 2)                 # SURVEY.md §8d, dense SK Float64
    out = {"model": "GraphSK" if binary else "GraphSKNormal", "N": N, "replicas": R, "beta": beta, "iters": iters, "attempts_per_s": attempts / dt,
           "kernel_ms": sweep_ms, "acceptance": a, "energy_per_spin": float(Es[:, -1].mean()) / N,
           "algorithmic_bytes_per_attempt": bytes_per_attempt,
           "algorithmic_GBps_kernel": bytes_per_attempt * attempts / (sweep_ms * 1e-3) / 1e9}
    print(json.dumps(out))
    eng.close()


def bench_ea(L=64, D=3, R=512, beta=1.0, sweeps=64, step=16, seed=0x5EED):
    """BASELINE.json configs[3] on ONE GPU's share: GraphEA L=64 D=3, 512 of the 4096 replicas, checkerboard sweeps."""
    pkg = entry.load_package()
    X = pkg.GraphEA(L, D, seed=seed)
    eng = pkg.Engine(X, R)
    eng.seed(seed)
    eng.init_spins_random()
    eng.set_coloring(pkg.checkerboard_coloring(L, D))
    eng.colored_sweeps_async(beta, 8, 8); eng.sync()
    t0 = time.perf_counter()
    eng.colored_sweeps_async(beta, sweeps, step); eng.sync()
    dt = time.perf_counter() - t0
    total_ms, sweep_ms, _ = eng.last_timing()
    Es, _ = eng.fetch_results()
    attempts = float(R) * sweeps * X.N
    out = {"model": "GraphEA checkerboard", "L": L, "D": D, "replicas": R, "beta": beta, "sweeps": sweeps,
           "attempts_per_s": attempts / dt, "device_ms": sweep_ms, "energy_per_spin": float(Es[:, -1].mean()) / X.N}
    print(json.dumps(out))
    eng.close()


def bench_quant(Nk=1024, M=32, R=128, beta=2.0, Gamma=0.5, iters=1 << 16, step=1 << 12, seed=0x5EED):
    """BASELINE.json configs[4] on ONE GPU's share: GraphQuant(GraphRRG(1024,3), M=32) under rrrMC, 128 of the 1024 replicas."""
    pkg = entry.load_package()
    X = pkg.GraphQuant(pkg.GraphRRG(Nk, 3, seed=seed), M, Gamma, beta)
    for R in ([int(a) for a in sys.argv[2:]] or [R]):        # the path scales with the replicas: python tools/bench_models.py quant 128 1024 8192
        it = max(step, iters * 128 // max(R, 128))
        eng = pkg.Engine(X, R)
        eng.seed(seed)
        eng.init_spins_random()
        t0 = time.perf_counter()
        Es, acc, staged = eng.rrr_mc(beta, it, step)
        dt = time.perf_counter() - t0
        total_ms, sweep_ms, _ = eng.last_timing()
        out = {"model": "GraphQuant rrrMC", "Nk": Nk, "M": M, "replicas": R, "beta": beta, "Gamma": Gamma, "iters": it,
               "iterations_per_s": float(R) * it / dt, "kernel_iterations_per_s": float(R) * it / (sweep_ms * 1e-3), "kernel_ms": sweep_ms,
               "acceptance": float(acc.mean()) / it, "staged_frac": float(staged.mean()) / it, "energy_per_spin": float(Es[:, -1].mean()) / X.N}
        print(json.dumps(out), flush=True)
        eng.close()


def bench_spf(N=4096, K=3, R=65536, beta=1.0, iters=1 << 16, step=1 << 12, seed=0x5EED):
    """GraphRRGNormal (Float64 sparse): config-2 geometry with Gaussian couplings; the lane-per-replica
    kernel hides its accept-path latency with occupancy, so it is measured at several replica counts."""
    pkg = entry.load_package()
    K = int(os.environ.get("BENCH_SPF_K", K))            # (K = 7, 8: the degrees without fused pairs)
    X = pkg.GraphRRGNormal(N, K, seed=seed)
    model = "GraphRRGNormal"
    if os.environ.get("BENCH_SPF_EA"):                   # "L,D": GraphEANormal(L, D) instead (K = 2 D; L = 8, D = 4 has the N of the default)
        L_, D_ = (int(a) for a in os.environ["BENCH_SPF_EA"].split(","))
        X = pkg.GraphEANormal(L_, D_, seed=seed)
        N, K, model = X.N, 2 * D_, "GraphEANormal(L=%d,D=%d)" % (L_, D_)
    for R in ([int(a) for a in sys.argv[2:]] or [8192, 65536, 262144]):
        eng = pkg.Engine(X, R)
        eng.seed(seed)
        eng.init_spins_random()
        eng.standard_mc_async(beta, iters // 4, step); eng.sync()
        t0 = time.perf_counter()
        eng.standard_mc_async(beta, iters, step); eng.sync()
        dt = time.perf_counter() - t0
        total_ms, sweep_ms, _ = eng.last_timing()
        Es, acc = eng.fetch_results()
        a = float(acc.mean()) / iters
        attempts = float(R) * iters
        bpa = 8 + a * (10 + 17 * K)          # SURVEY.md §8d widths: field 8 B, spin 1 B; lfields_last excluded
        out = {"model": model, "N": N, "K": K, "replicas": R, "beta": beta, "iters": iters, "attempts_per_s": attempts / dt,
               "build": {k: os.environ[k] for k in ("RRRMC_SPF_TEAM", "RRRMC_SPF_TEAM_WAVES", "RRRMC_SPF_TEAM_WIDTH") if k in os.environ} or "default",
               "kernel_ms": sweep_ms, "acceptance": a, "energy_per_spin": float(Es[:, -1].mean()) / N,
               "algorithmic_bytes_per_attempt": bpa, "algorithmic_GBps_kernel": bpa * attempts / (sweep_ms * 1e-3) / 1e9}
        print(json.dumps(out), flush=True)
        eng.close()


def bench_spf_fast(N=4096, K=3, beta=1.0, iters=1 << 20, step=1 << 12, seed=0x5EED):
    """GraphRRGNormal under the opt-in fast standardMC (bit-sliced replicas, per-site thresholds): python tools/bench_models.py spf_fast 8192"""
    pkg = entry.load_package()
    X = pkg.GraphRRGNormal(N, K, seed=seed)
    for R in ([int(a) for a in sys.argv[2:]] or [8192]):
        for st in (step, 16 * step):
            eng = pkg.Engine(X, R)
            eng.seed(seed)
            eng.init_spins_random()
            eng.standard_mc_fast_async(beta, iters, st); eng.sync()      # warm-up of the same shape (buffers, chunk list, thresholds)
            t0 = time.perf_counter()
            eng.standard_mc_fast_async(beta, iters, st); eng.sync()
            dt = time.perf_counter() - t0
            total_ms, sweep_ms, nl = eng.last_timing()
            Es, acc = eng.fetch_results()
            a = float(acc.mean()) / iters
            attempts = float(R) * iters
            bpa = 8 + a * (10 + 17 * K)
            print(json.dumps({"model": "GraphRRGNormal fast standardMC", "N": N, "K": K, "replicas": R, "beta": beta, "iters": iters, "step": st,
                              "attempts_per_s": attempts / dt, "kernel_ms": sweep_ms, "launches": nl, "acceptance": a,
                              "energy_per_spin": float(Es[:, -1].mean()) / N, "algorithmic_bytes_per_attempt": bpa,
                              "algorithmic_GBps_kernel": bpa * attempts / (sweep_ms * 1e-3) / 1e9}), flush=True)
            eng.close()


def bench_dbl(N=4096, K=3, R=8192, beta=2.0, iters=1 << 14, step=1 << 12, seed=0x5EED):
    """GraphRRGNormalDiscretized(N, K, (-1,0,1)) under rrrMC(X::DoubleGraph): thread-per-replica kernel."""
    pkg = entry.load_package()
    X = pkg.GraphRRGNormalDiscretized(N, K, (-1, 0, 1), seed=seed)
    for R in ([int(a) for a in sys.argv[2:]] or [R]):
        eng = pkg.Engine(X, R)
        eng.seed(seed)
        eng.init_spins_random()
        t0 = time.perf_counter()
        Es, acc, staged = eng.rrr_mc(beta, iters, step)
        dt = time.perf_counter() - t0
        total_ms, sweep_ms, _ = eng.last_timing()
        out = {"model": "GraphRRGNormalDiscretized rrrMC", "N": N, "K": K, "replicas": R, "beta": beta, "iters": iters,
               "iterations_per_s": float(R) * iters / dt, "kernel_ms": sweep_ms, "acceptance": float(acc.mean()) / iters,
               "staged_frac": float(staged.mean()) / iters, "energy_per_spin": float(Es[:, -1].mean()) / X.N}
        print(json.dumps(out), flush=True)
        eng.close()


def bench_ea_random(L=64, D=3, beta=1.0, sweeps=8, seed=0x5EED):
    """GraphEA(64, 3) (BASELINE.json configs[3]'s lattice, N = 262 144) under the reference's own random-site standardMC: the big-N
    kernels (spins in HBM/L2).  python tools/bench_models.py ea_random 512 4096"""
    pkg = entry.load_package()
    X = pkg.GraphEA(L, D, seed=seed)
    for R in ([int(a) for a in sys.argv[2:]] or [512, 4096]):
        eng = pkg.Engine(X, R)
        eng.seed(seed)
        eng.init_spins_random()
        iters = sweeps * X.N
        eng.standard_mc_async(beta, X.N, X.N); eng.sync()
        t0 = time.perf_counter()
        eng.standard_mc_async(beta, iters, X.N); eng.sync()
        dt = time.perf_counter() - t0
        total_ms, sweep_ms, nl = eng.last_timing()
        Es, acc = eng.fetch_results()
        print(json.dumps({"model": "GraphEA random-site standardMC (big-N kernels)", "L": L, "D": D, "replicas": R, "beta": beta,
                          "iters": iters, "attempts_per_s": float(R) * iters / dt, "sweep_kernel_ms": sweep_ms, "launches": nl,
                          "acceptance": float(acc.mean()) / iters, "energy_per_spin": float(Es[:, -1].mean()) / X.N}), flush=True)
        eng.close()


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "sk"
    {"sk": bench_sk, "ea": bench_ea, "quant": bench_quant, "spf": bench_spf, "spf_fast": bench_spf_fast, "dbl": bench_dbl, "ea_random": bench_ea_random}[which]()
