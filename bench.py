#!/usr/bin/env python3
"""Headline benchmark: spin-flip attempts per second of standardMC on GraphRRG(N=4096, K=3, +-J), beta = 1,
8192 replicas per MI355X (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W
One "step" = one sampling call (rrrmc_standard_mc_async) of ITERS iterations for every replica with the
configuration already resident in HBM; results (energy samples, accepted counts) stay in HBM inside the
timed region.  For N > 1 launch under torch.distributed.run: one process per GPU, replicas sharded by
global replica id (no data-path collective; one RCCL all_gather of observables after the timed region).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

N_SITES, K_DEG, BETA = 4096, 3, 1.0
REPLICAS_PER_GPU = 8192
ITERS = 1 << 22          # iterations per replica per step (1024 lattice sweeps)
SAMPLE_STEP = 1 << 12    # energy sample every N iterations
SEED = 0x5EED
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(O, X, seconds_target=15.0):
    """Single-thread CPU oracle (port of the reference loop) on a bounded sample of the same workload."""
    A, J = X.A, X.J.astype(np.int32)
    iters, R = 1 << 22, 1
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    pinned, old_aff = None, None
    try:      # one pinned core, as SURVEY.md §8d prescribes (taskset -c <first allowed core>)
        old_aff = os.sched_getaffinity(0)
        pinned = min(old_aff)
        os.sched_setaffinity(0, {pinned})
    except (AttributeError, OSError):
        pinned = None
    ch = O.init_configs(SEED, 0, R, N_SITES)
    t0 = time.perf_counter()
    O.standard_mc_sparse_batch(A, J, BETA, iters, SAMPLE_STEP, SEED, ch)
    dt = time.perf_counter() - t0
    R = max(1, min(128, int(seconds_target / max(dt, 1e-3))))
    ch = O.init_configs(SEED, 0, R, N_SITES)
    t0 = time.perf_counter()
    O.standard_mc_sparse_batch(A, J, BETA, iters, SAMPLE_STEP, SEED, ch)
    dt = time.perf_counter() - t0
    if old_aff is not None and pinned is not None:
        try:
            os.sched_setaffinity(0, old_aff)
        except OSError:
            pass
    return {"value": R * iters / dt, "unit": "attempts/s", "cores": 1, "kind": "port",
            "sample": "%d replicas x 2^22 iterations of the same graph/beta, single thread, oracle/rrrmc_oracle.c (%.1f s)" % (R, dt),
            "cpu_model": cpu_model, "host_cores": os.cpu_count(), "pinned_core": pinned}


def device_copy_bandwidth(pkg, device, nbytes=1 << 30, reps=10):
    """Measured device-to-device copy rate (read + write bytes per second, GB/s): the practical HBM ceiling of this box
    that SURVEY.md §8d asks to report beside the nominal 8 TB/s.  Timed inside the library (HIP events on its own stream)."""
    import ctypes
    out = ctypes.c_double(0.0)
    rc = pkg.lib().rrrmc_device_copy_bandwidth(int(device), int(nbytes), int(reps), ctypes.byref(out))
    if rc != 0:
        raise RuntimeError("rrrmc_device_copy_bandwidth: status %d" % rc)
    return out.value


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--iters", type=int, default=ITERS)
    ap.add_argument("--replicas", type=int, default=REPLICAS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # RRRMC_BENCH_BACKEND=gloo is a debugging aid only: it lets several ranks share one GPU (RCCL refuses that) so that the
    # multi-rank code path can be exercised on a 1-GPU box; the driver's multi-GPU runs use the default, RCCL ("nccl").
    backend = os.environ.get("RRRMC_BENCH_BACKEND", "nccl")
    # RRRMC_BENCH_FORCE_DIST=1: run the distributed code path (process group, RCCL all_reduce / all_gather) with a single rank too
    if world > 1 or os.environ.get("RRRMC_BENCH_FORCE_DIST") == "1":
        import torch
        import torch.distributed as dist
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend)
    n_gpus = world

    entry.build() if rank == 0 and not os.path.exists(os.path.join(ROOT, "rrrmc.jl_amd", "lib", "librrrmc_hip.so")) else None
    pkg = entry.load_package()
    X = pkg.GraphRRG(N_SITES, K_DEG, seed=SEED)
    R = args.replicas
    r0, r_local = pkg.shard_bounds(R * world, world, rank)      # weak scaling: R replicas per GPU, ids 0 .. R*world-1
    assert r_local == R
    eng = pkg.Engine(X, R, device=local_rank, replica0=r0)
    eng.seed(SEED)
    eng.init_spins_random()

    def barrier():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()
        eng.sync()

    for _ in range(args.warmup):
        eng.standard_mc_async(BETA, args.iters, SAMPLE_STEP)
    barrier()
    sweep_ms, launches = 0.0, 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.standard_mc_async(BETA, args.iters, SAMPLE_STEP)
        if args.steps <= 64:      # HIP-event bookkeeping of each call (forces that call's completion)
            _, s, n = eng.last_timing()
            sweep_ms += s
            launches += n
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    Es, acc = eng.fetch_results()
    acc_rate = float(acc.mean()) / args.iters
    e_mean = float(Es[:, -1].mean()) / N_SITES if Es.shape[1] else float("nan")
    if dist is not None:   # the only exchange of the job: gather the per-replica observables over RCCL
        import torch
        dev = torch.device("cuda", local_rank) if backend == "nccl" else None
        E_last = pkg.gather_replica_major(Es[:, -1].copy(), R * world, dist, device=dev)
        acc_all = pkg.gather_replica_major(acc, R * world, dist, device=dev)
        acc_rate = float(acc_all.mean()) / args.iters
        e_mean = float(E_last.mean()) / N_SITES

    if rank == 0:
        attempts = float(R) * args.iters * args.steps * n_gpus
        value = attempts / dt
        bytes_per_attempt = 1.0 + acc_rate * (3 + 3 * K_DEG)          # SURVEY.md §8d fixed-width accounting
        out = {
            "metric": "spin-flip attempts/sec (whole node), GraphRRG N=4096 K=3 +-J standardMC beta=1.0",
            "value": value, "unit": "attempts/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "GraphRRG(N=4096,K=3,+-J) standardMC beta=1.0, %d replicas per GPU, %d iterations per replica per step, energy sample every %d"
                                   % (R, args.iters, SAMPLE_STEP),
                       "replicas_total": R * n_gpus, "acceptance": acc_rate, "energy_per_spin": e_mean,
                       "parallelism": "replicas sharded x%d, no data-path collective" % n_gpus},
        }
        if launches:
            per_launch_attempts = float(R) * args.iters * args.steps / launches
            avg_ms = sweep_ms / launches
            achieved = bytes_per_attempt * per_launch_attempts / (avg_ms * 1e-3) / 1e9
            traffic = None     # HBM bytes per sweep launch from the committed PMC passes (same workload), if present
            tf = os.path.join(ROOT, "profiles", "r01", "traffic.json")
            if os.path.exists(tf) and R == REPLICAS_PER_GPU and args.iters == ITERS:
                traffic = json.load(open(tf))["hbm_bytes_per_launch"]
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                               "algorithmic_bytes_per_launch": bytes_per_attempt * per_launch_attempts,
                               "kernel": "sweep_kernel<3, 1>", "avg_launch_ms": avg_ms, "launches": launches,
                               "algorithmic_bytes_per_attempt": bytes_per_attempt}
            try:      # after the timed region: the box's own copy bandwidth, for reference only (peak stays the nominal figure)
                bw = device_copy_bandwidth(pkg, local_rank)
                out["roofline"]["measured_copy_GBps"] = bw
                out["roofline"]["frac_of_measured_copy"] = achieved / bw
            except Exception as e:      # never let the side measurement hide the bench line
                out["roofline"]["measured_copy_GBps"] = None
                sys.stderr.write("device copy bandwidth not measured: %r\n" % (e,))
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (rank 0's host cores)
            out["cpu_baseline"] = cpu_baseline(entry.load_oracle(), X)
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
