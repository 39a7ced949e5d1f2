#!/usr/bin/env python3
"""Headline benchmark: spin-flip attempts per second of standardMC on GraphRRG(N=4096, K=3, +-J), beta = 1,
8192 replicas per MI355X (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W

One "step" = one sampling call (rrrmc_standard_mc_async) of ITERS iterations for every replica with the
configuration already resident in HBM; results (energy samples, accepted counts) stay in HBM inside the
timed region.  Replicas are sharded by global replica id, one process per GPU, no data-path collective; the
only exchange is one all_gather of observables after the timed region (RCCL).

N > 1 runs either way:
  * under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or
  * stand-alone: `python bench.py --gpus N` spawns N fresh rank processes itself (before anything touches the GPU in
    the parent) and relays rank 0's line.  Fewer than N visible devices is an error, never a silent 1-GPU run.
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import socket
import subprocess
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
HOST_GAP_LIMIT_MS = 0.3  # ms_per_step - kernel ms per step above this fails the run (exit 4): the timed region must be the kernel
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PROFILE_DIRS = [os.path.join(ROOT, "profiles", d) for d in ("r06", "r05", "r04", "r03", "r02")]       # newest first
# what traffic.json / valu_model.json describe: the kernel AND the code that launches it (chunk list, planner overlap, events) — a change to
# how the kernel is launched moves its time like a change to the kernel, and invalidates the committed figures the same way
SWEEP_KERNEL_SOURCES = ["rrrmc.jl_amd/csrc/sparse_kernels.hpp", "rrrmc.jl_amd/csrc/philox.hpp", "rrrmc.jl_amd/csrc/host_sweep.hpp",
                        "rrrmc.jl_amd/csrc/host_plan.hpp"]
LINE_LIMIT_BYTES = 5000  # the printed line: numbers and short keys (README.md "The bench line" is the legend); the full record goes to stderr / a file


# ----------------------------------------------------------------------------------------------------------------
# stand-alone multi-GPU launcher (the parent never initialises HIP: it only counts devices and spawns children)
# ----------------------------------------------------------------------------------------------------------------
def visible_gpu_count():
    """Devices the rank processes will see, counted WITHOUT any HIP call in this process: the visibility variables if set, else the KFD
    topology (every node with SIMDs is a GPU); torch's counter only as a last resort (it may fall back to hipGetDeviceCount)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        vis = os.environ.get(var)
        if vis is not None:
            return len([v for v in vis.split(",") if v.strip()])
    topo = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(topo):
            for line in open(os.path.join(topo, node, "properties")):
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
        if n:
            return n
    except (OSError, ValueError):
        pass
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:
        return 0


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv, child_cmd=None, env_extra=None, timeout=None):
    """Start n rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), wait for all of them and
    return (exit code, rank 0's stdout).  Other ranks' stdout goes to our stderr.  A failing rank ends the job."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        if env_extra:
            env.update(env_extra)
        cmd = (child_cmd or [sys.executable, os.path.abspath(__file__)]) + list(argv)
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True))
    outs, rc = [""] * n, 0
    deadline = None if timeout is None else time.time() + timeout
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                try:
                    o, _ = procs[r].communicate(timeout=0.2)
                except subprocess.TimeoutExpired:
                    continue
                outs[r] = o or ""
                pending.discard(r)
                if procs[r].returncode != 0:
                    rc = rc or procs[r].returncode
                    sys.stderr.write("bench.py: rank %d exited with status %d\n" % (r, procs[r].returncode))
            if rc and pending:               # one rank failed: the others would wait for it at the barrier forever
                break
            if deadline is not None and time.time() > deadline:
                rc = rc or 124
                sys.stderr.write("bench.py: ranks still running after %.0f s\n" % timeout)
                break
    finally:
        for p in procs:                      # exact PIDs of the children we started
            if p.poll() is None:
                p.kill()
                try:
                    p.communicate(timeout=10)
                except Exception:
                    pass
    for r in range(1, n):
        if outs[r].strip():
            sys.stderr.write("[rank %d stdout] %s\n" % (r, outs[r].strip()))
    return rc, outs[0]


def launch(args, argv):
    n = args.gpus
    share = os.environ.get("RRRMC_BENCH_SHARE_GPU") == "1"      # debugging aid: several ranks on one device (forces gloo)
    have = visible_gpu_count()
    if have < n and not share:
        sys.stderr.write("bench.py: --gpus %d requested but only %d GPU(s) are visible; refusing to run a smaller job under that "
                         "label (RRRMC_BENCH_SHARE_GPU=1 lets ranks share devices for debugging)\n" % (n, have))
        return 2
    entry.build_only()                       # compile once, before the ranks start (no GPU use)
    extra = {"RRRMC_BENCH_BACKEND": "gloo"} if share else None
    rc, out0 = spawn_ranks(n, argv, env_extra=extra)
    line = [ln for ln in out0.splitlines() if ln.startswith("{")]
    if rc != 0 or not line:
        sys.stderr.write("bench.py: multi-rank run failed (status %d)\n%s\n" % (rc, out0))
        return rc or 1
    print(line[-1])
    return 0


# ----------------------------------------------------------------------------------------------------------------
# CPU baseline (oracle = test infrastructure; used here only as the reported baseline, after the timed region)
# ----------------------------------------------------------------------------------------------------------------
def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


class pinned_core:
    """One pinned core, as SURVEY.md §8d prescribes (taskset -c <first allowed core>)."""

    def __enter__(self):
        self.old, self.core = None, None
        try:
            self.old = os.sched_getaffinity(0)
            self.core = min(self.old)
            os.sched_setaffinity(0, {self.core})
        except (AttributeError, OSError):
            self.core = None
        return self

    def __exit__(self, *a):
        if self.old is not None and self.core is not None:
            try:
                os.sched_setaffinity(0, self.old)
            except OSError:
                pass


def cpu_baseline(O, X, seconds_target=15.0):
    """Single-thread CPU oracle (port of the reference loop) on a bounded sample of the same workload."""
    A, J = X.A, X.J.astype(np.int32)
    iters, R = 1 << 22, 1
    with pinned_core() as pc:
        ch = O.init_configs(SEED, 0, R, N_SITES)
        t0 = time.perf_counter()
        O.standard_mc_sparse_batch(A, J, BETA, iters, SAMPLE_STEP, SEED, ch)
        dt = time.perf_counter() - t0
        R = max(1, min(128, int(seconds_target / max(dt, 1e-3))))
        ch = O.init_configs(SEED, 0, R, N_SITES)
        t0 = time.perf_counter()
        O.standard_mc_sparse_batch(A, J, BETA, iters, SAMPLE_STEP, SEED, ch)
        dt = time.perf_counter() - t0
    return {"value": R * iters / dt, "unit": "attempts/s", "cores": 1, "kind": "port",
            "sample": "%d replicas x 2^22 iterations of the same graph/beta, single thread, oracle/rrrmc_oracle.c (%.1f s)" % (R, dt),
            "sample_short": "%d replicas x 2^22 its, 1 thread, %.0f s" % (R, dt),
            "build": "gcc -O3 -march=%s -ffp-contract=off" % O.flavour,
            "cpu_model": cpu_model_name(), "host_cores": os.cpu_count(), "pinned_core": pc.core}


def timed_oracle(fn, unit_per_call, min_seconds=1.0, max_calls=64):
    """Repeat an oracle call until at least `min_seconds` have been timed (SURVEY.md §8d asks for a sample that is not noise);
    returns (units per second, calls, seconds)."""
    n, t0 = 0, time.perf_counter()
    while True:
        fn(n)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= min_seconds or n >= max_calls:
            return n * unit_per_call / dt, n, dt


def timed_oracle_all_cores(fn, args, kwargs, chunks_arg, chunks, unit_per_call, chains, min_seconds=1.5):
    """SURVEY.md §8d's optional third figure: `chains` INDEPENDENT oracle chains — O.<fn>(*args, replica=c, it0=k * unit_per_call, **kwargs) with
    argument `chunks_arg` replaced by chunks[c] — one PROCESS each (tools/oracle_worker.py; threads of one process were seen to share cores on
    the pool's virtualised hosts), every process repeating its call for `min_seconds`.  chains = min(the workload's replicas, the host cores this
    process may use).  Returns (units per second of all chains together, chains, seconds of the slowest)."""
    import pickle
    import tempfile
    job = {"fn": fn, "args": list(args), "kwargs": dict(kwargs), "chunks_arg": chunks_arg, "chunks": [np.asarray(c) for c in chunks],
           "it0_stride": int(unit_per_call), "seconds": float(min_seconds)}
    with tempfile.NamedTemporaryFile(suffix=".pkl", delete=False) as f:
        pickle.dump(job, f)
        path = f.name
    try:
        env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")          # the workers are CPU-only
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "oracle_worker.py"), path, str(c)], stdout=subprocess.PIPE, text=True, env=env)
                 for c in range(chains)]
        total, slowest = 0.0, 0.0
        for p_ in procs:
            o, _ = p_.communicate(timeout=300)
            if p_.returncode != 0:
                raise RuntimeError("oracle worker failed")
            n, dt = o.split()
            total += int(n) * unit_per_call / float(dt)
            slowest = max(slowest, float(dt))
    finally:
        os.remove(path)
    return total, chains, slowest


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return os.cpu_count() or 1


CPU_WORKERS_MAX = int(os.environ.get("RRRMC_BENCH_CPU_WORKERS", "64"))       # a one-GPU box of the pool is a SHARE of a 256-core host


def cpu_all_cores_entry(out, one_core_value, unit, measured, what):
    """cpu_all_cores = what `workers` concurrent oracle processes delivered on this box (its effective parallelism is printed: the box is a
    share of its host), cpu_host_projection = one core x every core the OS shows — the whole-host figure of SURVEY.md §8d where the box
    cannot be asked for it.  Baselines, never targets."""
    v, workers, dt = measured
    nc = host_cores()
    out["cpu_all_cores"] = {"value": v, "unit": unit, "cores": workers, "kind": "port", "effective_cores": v / one_core_value,
                            "sample": "%d independent chains (min(replicas, host cores, %d)) %s for %.1f s, one process each, oracle" % (workers, CPU_WORKERS_MAX, what, dt)}
    out["cpu_host_projection"] = {"value": one_core_value * nc, "unit": unit, "cores": nc, "kind": "projection: cpu_one_core x the cores the OS shows (not measured)"}
    out["gpu_over_cpu_all_cores"] = out["value"] / v
    out["gpu_in_host_cores"] = out["value"] / one_core_value


def source_stamp(files=None):
    """sha1 over the sources of the kernel a committed profile describes: a profile JSON whose stamp differs was measured on another build."""
    import hashlib
    h = hashlib.sha1()
    for f in files or SWEEP_KERNEL_SOURCES:
        try:
            h.update(open(os.path.join(ROOT, f), "rb").read())
        except OSError:
            h.update(b"missing:" + f.encode())
    return h.hexdigest()[:16]


def git_head():
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL, text=True).strip()
    except Exception:
        return None


def verify_timed_region(O, X, C_start, C_end, Es_last, acc_last, it0, steps, iters, replicas, replica0):
    """Replay the WHOLE timed region (steps x iters iterations from the configuration it started with) of a few replicas through the
    CPU oracle and compare what the timed run left behind: final configuration, the last step's energy samples and accepted count.
    The oracle is the checker here, after the timed region; one thread per replica (ctypes releases the GIL)."""
    import threading
    A, J = X.A, X.J.astype(np.int32)
    res = {}

    def one(r):
        ch = C_start[r].copy()
        if steps > 1:
            ch = O.standard_mc_sparse(A, J, BETA, iters * (steps - 1), SAMPLE_STEP, SEED, ch, it0=it0, replica=replica0 + r)[1]
        o = O.standard_mc_sparse(A, J, BETA, iters, SAMPLE_STEP, SEED, ch, it0=it0 + iters * (steps - 1), replica=replica0 + r)
        res[r] = bool((o[1] == C_end[r]).all() and (o[0] == Es_last[r]).all() and o[2] == acc_last[r])

    t0 = time.perf_counter()
    th = [threading.Thread(target=one, args=(r,)) for r in replicas]
    [t.start() for t in th]
    [t.join() for t in th]
    return {"replicas": [int(replica0 + r) for r in replicas], "ok": [res.get(r, False) for r in replicas],
            "iterations_replayed_per_replica": int(iters) * int(steps), "seconds": time.perf_counter() - t0,
            "compared": "final configuration, energy samples of the last step, accepted count of the last step (bit-exact)"}


def device_copy_bandwidth(pkg, device, nbytes=1 << 30, reps=10):
    """Measured device-to-device copy rate (read + write bytes per second, GB/s): the practical HBM ceiling of this box
    that SURVEY.md §8d asks to report beside the nominal 8 TB/s.  Timed inside the library (HIP events on its own stream)."""
    import ctypes
    out = ctypes.c_double(0.0)
    rc = pkg.lib().rrrmc_device_copy_bandwidth(int(device), int(nbytes), int(reps), ctypes.byref(out))
    if rc != 0:
        raise RuntimeError("rrrmc_device_copy_bandwidth: status %d" % rc)
    return out.value


def load_profile_json(name):
    """Newest committed profile JSON of that name, with its provenance: (dict, 'profiles/rNN/name', stamp_state) where stamp_state is
    'match' (measured on these kernel sources), 'stale' (another build: the caller drops the derived figures) or 'unstamped'."""
    for d in PROFILE_DIRS:
        p = os.path.join(d, name)
        if os.path.exists(p):
            try:
                j = json.load(open(p))
            except ValueError:
                continue
            st = j.get("source_stamp")
            # a profile names the sources it describes ("source_files"; the headline kernel's when absent)
            state = "unstamped" if not st else ("match" if st == source_stamp(j.get("source_files")) else "stale")
            return j, os.path.relpath(p, ROOT), state
    return None, None, None


# ----------------------------------------------------------------------------------------------------------------
# secondary configurations (BASELINE.json configs[2..4] at ONE GPU's share), after the headline's timed region
# ----------------------------------------------------------------------------------------------------------------
def secondary_c3(pkg, O, device):
    """configs[2]: GraphSKNormal N=1024, 2048 replicas, standardMC beta=1 (sk_block_kernel)."""
    N, R, beta, iters, step = 1024, 2048, 1.0, 1 << 20, 1 << 10      # SURVEY.md §8d's C3 shape: 16 segments of the blocked kernel per call
    X = pkg.GraphSKNormal(N, seed=SEED)
    with pkg.Engine(X, R, device=device) as eng:
        eng.seed(SEED)
        eng.init_spins_random()
        eng.standard_mc_async(beta, iters // 16, step); eng.sync()
        t0 = time.perf_counter()
        eng.standard_mc_async(beta, iters, step); eng.sync()
        dt = time.perf_counter() - t0                                # end to end: energy(X, C), block tables of every segment, the 16 launches
        tot_ms, k_ms, nl = eng.last_timing()
        _, acc = eng.fetch_results(want_energies=False)
    a = float(acc.mean()) / iters
    bpa = 8 + a * (17 * N + 2)                                   # SURVEY.md §8d, dense SK Float64
    out = {"workload": "GraphSKNormal(N=1024) standardMC beta=1.0, 2048 replicas, 2^20 iterations per replica, energy sample every 1024", "value": R * iters / dt,
           "unit": "attempts/s", "kernel": "sk_sweep_kernel" if os.environ.get("RRRMC_SK_LEGACY") == "1" else ("sk_block_kernel" if os.environ.get("RRRMC_SK_BLOCK_V1") == "1" else "sk_hblock_kernel<2, 512, 8>"),
           "avg_launch_ms": k_ms / max(nl, 1), "launches": nl, "call_ms": tot_ms, "acceptance": a,
           "algorithmic_bytes_per_attempt": bpa, "achieved_GBps": bpa * R * iters / (k_ms * 1e-3) / 1e9,
           "note": "algorithmic bytes of the Float64-field picture (SURVEY.md §8d) over the kernel time: the fields live in registers (8 replicas "
                   "per workgroup, one workgroup per compute unit), the only traffic is the 8 KiB row of 4J per attempt and workgroup from L2 / Infinity Cache; "
                   "see DESIGN.md 4c for the phase budget (decide chain, Float64 multiply-add issue)"}
    out["frac"] = out["achieved_GBps"] / HBM_PEAK_GBS
    # FP64 VALU view: the useful work is a * N Float64 adds per attempt and replica (the field update); MI355X vector FP64 = 78.6 TFLOP/s as FMAs
    out["fp64_adds_per_s"] = a * N * R * iters / (k_ms * 1e-3)
    out["fp64_valu_frac"] = out["fp64_adds_per_s"] / (78.6e12 / 2)
    out["bound"] = "instruction issue, two wavefronts per SIMD: the serial decide chain (56 instructions per accepted move) + the fp64 bulk update (profiles/r06/c3_floor.md)"
    out["bound_frac"] = out["fp64_valu_frac"]
    out["bound_frac_meaning"] = ("useful Float64 field adds per second over the vector FP64 add rate (39.3e12/s); against the issue floor of its own "
                                 "design the block is at 0.77 (profiles/r06/c3_floor.md)")
    if O is not None:
        with pinned_core():
            ch = O.init_configs(SEED, 0, 1, N)[0]
            it1 = 1 << 16
            v, n, dt1 = timed_oracle(lambda k: O.standard_mc_skn(X.J, beta, it1, step, SEED, ch, it0=k * it1), it1)
            out["cpu_one_core"] = {"value": v, "unit": "attempts/s", "kind": "port", "build": O.flavour,
                                   "sample": "1 replica x %d x 2^16 iterations (%.1f s), oracle" % (n, dt1)}
    return out


def secondary_c4(pkg, O, device):
    """configs[3] at one GPU's share: GraphEA L=64 D=3, 512 of the 4096 replicas, colour-parallel (checkerboard) sweeps."""
    L, D, R, beta, sweeps, step = 64, 3, 512, 1.0, 256, 16
    X = pkg.GraphEA(L, D, seed=SEED)
    with pkg.Engine(X, R, device=device) as eng:
        eng.seed(SEED)
        eng.init_spins_random()
        eng.set_coloring(pkg.checkerboard_coloring(L, D))
        eng.colored_sweeps_async(beta, 16, 16); eng.sync()
        t0 = time.perf_counter()
        eng.colored_sweeps_async(beta, sweeps, step); eng.sync()
        dt = time.perf_counter() - t0
        _, dev_ms, nl = eng.last_timing()
        eng.colored_count_accepted(True)          # acceptance from a short counted call (the timed build does not count)
        eng.colored_sweeps_async(beta, 16, 16); eng.sync()
        _, acc = eng.fetch_results(want_energies=False)
    attempts = float(R) * sweeps * X.N
    a = float(acc.mean()) / (16 * X.N)
    bpa = 1 + a * (3 + 3 * X.K)                                  # 1 + 21 a for the 6-neighbour lattice
    out = {"workload": "GraphEA(L=64,D=3,+-J) checkerboard sweeps beta=1.0, 512 replicas (one GPU's share of 4096), 256 sweeps, sample every 16",
           "value": attempts / dt, "unit": "attempts/s", "kernel": "colored_sweep_kernel<6>", "device_ms": dev_ms, "launches": nl,
           "acceptance": a, "algorithmic_bytes_per_attempt": bpa, "achieved_GBps": bpa * attempts / (dev_ms * 1e-3) / 1e9}
    out["frac"] = out["achieved_GBps"] / HBM_PEAK_GBS
    # measured traffic of a colour launch (profiles/r04/c4_summary.txt): 42 MB per 512-replica colour = 0.64 B per attempt (bit planes)
    out["bound"] = "hbm / L2 (bit-plane spin words of the colour and its neighbours, measured 0.64 B per attempt)"
    out["bound_frac"] = 0.64 * attempts / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    out["bound_frac_meaning"] = "measured bytes per attempt (profiles/r04/c4_summary.txt) at this run's rate over the 8 TB/s peak"
    if O is not None:
        with pinned_core():
            ch = O.init_configs(SEED, 0, 1, X.N)[0]
            Ji, col = X.J.astype(np.int32), pkg.checkerboard_coloring(L, D)
            v, n, dt1 = timed_oracle(lambda k: O.colored_sweeps_sparse(X.A, Ji, col, beta, 8, 8, SEED, ch, sweep0=8 * k), 8 * X.N)
            out["cpu_one_core"] = {"value": v, "unit": "attempts/s", "kind": "port", "build": O.flavour,
                                   "sample": "1 replica x %d x 8 sweeps (%.1f s), oracle" % (n, dt1)}
    return out


def secondary_c4_random(pkg, O, device):
    """configs[3]'s lattice under the reference's OWN sampler (random-site standardMC, src/RRRMC.jl:81-127) at one GPU's share:
    plan_big_kernel / big_mask_kernel / big_apply_kernel (spins of 4 replicas per workgroup in LDS)."""
    L, D, R, beta, sweeps = 64, 3, 512, 1.0, 16
    X = pkg.GraphEA(L, D, seed=SEED)
    with pkg.Engine(X, R, device=device) as eng:
        eng.seed(SEED)
        eng.init_spins_random()
        eng.standard_mc_async(beta, X.N, X.N); eng.sync()
        t0 = time.perf_counter()
        eng.standard_mc_async(beta, sweeps * X.N, X.N); eng.sync()
        dt = time.perf_counter() - t0
        _, dev_ms, nl = eng.last_timing()
        _, acc = eng.fetch_results(want_energies=False)
    attempts = float(R) * sweeps * X.N
    a = float(acc.mean()) / (sweeps * X.N)
    bpa = 1 + a * (3 + 3 * X.K)
    out = {"workload": "GraphEA(L=64,D=3,+-J) random-site standardMC beta=1.0, 512 replicas (one GPU's share of 4096), 16 sweeps, sample every sweep",
           "value": attempts / dt, "unit": "attempts/s", "kernel": "big_apply_kernel<6>", "device_ms": dev_ms, "launches": nl,
           "acceptance": a, "algorithmic_bytes_per_attempt": bpa, "achieved_GBps": bpa * attempts / (dev_ms * 1e-3) / 1e9,
           "note": "the spins live in LDS (4 replicas per workgroup); the kernel streams 36 bytes of plan records and masks per attempt "
                   "and workgroup, which is what bounds it (DESIGN.md 4g)"}
    out["frac"] = out["achieved_GBps"] / HBM_PEAK_GBS
    out["bound"] = "hbm / L2 stream of plan records and masks (36 B per attempt and workgroup of 4 replicas = 9 B per replica-attempt)"
    out["bound_frac"] = 9.0 * attempts / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    out["bound_frac_meaning"] = "streamed bytes per replica-attempt at this run's rate over the 8 TB/s peak (profiles/r04/c4rand_summary.txt)"
    if O is not None:
        with pinned_core():
            ch = O.init_configs(SEED, 0, 1, X.N)[0]
            Ji = X.J.astype(np.int32)
            v, n, dt1 = timed_oracle(lambda k: O.standard_mc_sparse(X.A, Ji, beta, 8 * X.N, X.N, SEED, ch, it0=8 * X.N * k, replica=0, form="ea"), 8 * X.N)
            out["cpu_one_core"] = {"value": v, "unit": "attempts/s", "kind": "port", "build": O.flavour,
                                   "sample": "1 replica x %d x 8 sweeps (%.1f s), oracle" % (n, dt1)}
    return out


def secondary_c5(pkg, O, device):
    """configs[4] at one GPU's share: GraphQuant(GraphRRG(1024,3), M=32) under rrrMC, 128 of the 1024 replicas."""
    Nk, M, R, beta, Gamma, iters, step = 1024, 32, 128, 2.0, 0.5, 1 << 17, 1 << 12
    X = pkg.GraphQuant(pkg.GraphRRG(Nk, 3, seed=SEED), M, Gamma, beta)
    with pkg.Engine(X, R, device=device) as eng:
        eng.seed(SEED)
        eng.init_spins_random()
        eng.rrr_mc(beta, iters // 8, step, want_energies=False)
        t0 = time.perf_counter()
        _, acc, staged = eng.rrr_mc(beta, iters, step, want_energies=False)
        dt = time.perf_counter() - t0
        _, k_ms, nl = eng.last_timing()
    a = float(acc.mean()) / iters
    out = {"workload": "GraphQuant(GraphRRG(1024,3), M=32, Gamma=0.5) rrrMC beta=2.0, 128 replicas (one GPU's share of 1024), 2^17 iterations per replica",
           "value": R * iters / dt, "unit": "iterations/s", "kernel": eng_kernel_name_quant(), "avg_launch_ms": k_ms / max(nl, 1),
           "launches": nl, "acceptance": a, "staged_frac": float(staged.mean()) / iters}
    # ~ 6 + a' (12 + 8 + 40 q) bytes per iteration, q = fraction of neighbours changing class (not measured: q = 1 bound)
    bpa_lo, bpa_hi = 6 + a * (12 + 8), 6 + a * (12 + 8 + 40)
    out["algorithmic_bytes_per_iteration_range"] = [bpa_lo, bpa_hi]
    out["achieved_GBps_range"] = [b * R * iters / (k_ms * 1e-3) / 1e9 for b in (bpa_lo, bpa_hi)]
    out["frac"] = out["achieved_GBps_range"][1] / HBM_PEAK_GBS
    # one wavefront per chain: 2413 cycles per iteration measured against an issue + LDS round-trip floor of ~ 2000 (profiles/r04/c5_floor.md)
    out["bound"] = "latency of ONE wavefront per chain (322 issued instructions + 7 dependent LDS round trips per iteration); 128 chains light 128 of 1024 SIMDs"
    out["bound_frac"] = 2000.0 / max(k_ms * 1e-3 * 2.1e9 / iters, 1.0)
    out["bound_frac_meaning"] = "the floor of c5_floor.md (2000 cycles per iteration) over this run's cycles per iteration at 2.1 GHz"
    if O is not None:
        Ji = X.X1.J.astype(np.int32)
        with pinned_core():
            ch = O.init_configs(SEED, 0, 1, X.N)[0]
            it1 = 1 << 22
            v, n, dt1 = timed_oracle(lambda k: O.rrr_mc_quant(X.X1.A, Ji, M, X.fourK, beta, it1, step, SEED, ch, it0=k * it1), it1)
            out["cpu_one_core"] = {"value": v, "unit": "iterations/s", "kind": "port", "build": O.flavour,
                                   "sample": "1 replica x %d x 2^22 iterations (%.1f s), oracle" % (n, dt1)}
        nc = min(R, host_cores(), CPU_WORKERS_MAX)
        chs = O.init_configs(SEED, 0, nc, X.N)
        it2 = 1 << 20
        cpu_all_cores_entry(out, out["cpu_one_core"]["value"], "iterations/s",
                            timed_oracle_all_cores("rrr_mc_quant", (X.X1.A, Ji, M, X.fourK, beta, it2, step, SEED, None), {}, 8, chs, it2, nc), "x 2^20-iteration calls")
    return out


def secondary_f64_fast(pkg, O, device):
    """Float64 sparse model in the opt-in fast mode: GraphRRGNormal(N=4096, K=3) at config 2's geometry (8192 replicas, sample every N)."""
    N, K, R, beta, iters, step = 4096, 3, 8192, 1.0, 1 << 20, 1 << 12
    X = pkg.GraphRRGNormal(N, K, seed=SEED)
    with pkg.Engine(X, R, device=device) as eng:
        eng.seed(SEED)
        eng.init_spins_random()
        eng.standard_mc_fast_async(beta, iters, step); eng.sync()
        t0 = time.perf_counter()
        eng.standard_mc_fast_async(beta, iters, step); eng.sync()
        dt = time.perf_counter() - t0
        _, k_ms, nl = eng.last_timing()
        _, acc = eng.fetch_results(want_energies=False)
    a = float(acc.mean()) / iters
    bpa = 8 + a * (10 + 17 * K)                                  # SURVEY.md §8d widths: field 8 B, spin 1 B
    out = {"workload": "GraphRRGNormal(N=4096,K=3) standardMC (fast mode) beta=1.0, 8192 replicas, 2^20 iterations per replica, energy sample every 4096",
           "value": R * iters / dt, "unit": "attempts/s", "kernel": "spf_fast_kernel<3>", "avg_launch_ms": k_ms / max(nl, 1), "launches": nl,
           "acceptance": a, "algorithmic_bytes_per_attempt": bpa, "achieved_GBps": bpa * R * iters / (k_ms * 1e-3) / 1e9,
           "note": "algorithmic bytes of the Float64-field picture (SURVEY.md §8d) over the kernel time: the state lives in LDS, this is a throughput normalisation",
           "bound": "valu issue (bit-sliced replicas in LDS, per-lane thresholds; same machinery as the headline kernel)", "bound_frac": None}
    out["frac"] = out["achieved_GBps"] / HBM_PEAK_GBS
    if O is not None:
        with pinned_core():
            ch = O.init_configs(SEED, 0, 1, N)[0]
            it1 = 1 << 24
            v, n, dt1 = timed_oracle(lambda k: O.standard_mc_spf(X.A, X.J, beta, it1, step, SEED, ch, it0=k * it1), it1)
            out["cpu_one_core"] = {"value": v, "unit": "attempts/s", "kind": "port", "build": O.flavour,
                                   "sample": "1 replica x %d x 2^24 iterations of the reference's cached-field loop (%.1f s), oracle" % (n, dt1)}
    return out


SPF_TEAM_SOURCES = ["rrrmc.jl_amd/csrc/spf_team_kernel.hpp", "rrrmc.jl_amd/csrc/spf_team_params.hpp", "rrrmc.jl_amd/csrc/spf_team_tu.hip",
                    "rrrmc.jl_amd/csrc/host_spf.hpp", "rrrmc.jl_amd/csrc/spf_kernels.hpp"]      # what spf_traffic.json describes


def spf_team_kernel_name(eng, K=3):
    """the build the library launches for this context, asked of the library (rrrmc_spf_team_build) — not a copy of host_spf.hpp's rule"""
    nw, tw, m = eng.spf_team_build()
    return "spf_team_kernel<%d, %d, %d, %d>" % (K, nw, m, tw) if nw else "spf_sweep_kernel<%d>" % K


def spf_traffic(key=None):
    """The committed rocprofv3 traffic measurement of spf_team_kernel, or (None, why) when it is absent, unstamped or was taken on other sources."""
    tf, tf_path, state = load_profile_json("spf_traffic.json")
    if not tf:
        return None, "no committed spf_traffic.json"
    if state != "match":
        return None, "%s is %s (measured on other kernel sources): traffic figures dropped" % (tf_path, state)
    sub = tf.get(key) if key else tf
    if not sub:
        return None, "%s has no entry %s" % (tf_path, key)
    return (sub, tf, tf_path), None


def secondary_f64_exact(pkg, O, device):
    """The DEFAULT (bit-exact) Float64 sparse path: GraphRRGNormal(N=4096, K=3), spf_team_kernel — the reference's cached-field loop
    (src/graphs/RRG.jl:504-627) with the fields in HBM, teams of sixteen wavefronts running the attempts whose neighbourhoods do not meet
    side by side; the measured HBM traffic per attempt comes from the committed rocprofv3 pass."""
    N, K, R, beta, iters, step = 4096, 3, 8192, 1.0, 1 << 16, 1 << 12
    X = pkg.GraphRRGNormal(N, K, seed=SEED)
    with pkg.Engine(X, R, device=device) as eng:
        eng.seed(SEED)
        eng.init_spins_random()
        eng.standard_mc_async(beta, iters, step); eng.sync()
        t0 = time.perf_counter()
        eng.standard_mc_async(beta, iters, step); eng.sync()
        dt = time.perf_counter() - t0
        _, k_ms, nl = eng.last_timing()
        _, acc = eng.fetch_results(want_energies=False)
        kname = spf_team_kernel_name(eng, K)
    a = float(acc.mean()) / iters
    bpa = 8 + a * (10 + 17 * K)                                  # SURVEY.md §8d widths: field 8 B, spin 1 B
    out = {"workload": "GraphRRGNormal(N=4096,K=3) standardMC (exact mode) beta=1.0, 8192 replicas, 2^16 iterations per replica, energy sample every 4096",
           "value": R * iters / dt, "unit": "attempts/s", "kernel": kname, "avg_launch_ms": k_ms / max(nl, 1), "launches": nl,
           "acceptance": a, "algorithmic_bytes_per_attempt": bpa, "achieved_GBps": bpa * R * iters / (k_ms * 1e-3) / 1e9}
    out["frac"] = out["achieved_GBps"] / HBM_PEAK_GBS
    # what binds it at this replica count: the memory system.  The kernel reads the lines of the Float64 fields whole and stores the accepting lanes'
    # fields (51 measured bytes per attempt, Infinity Cache included); one team of 32 replicas per compute unit, two attempts per wavefront instruction
    out["bound"] = "hbm / memory system (measured traffic: the 256-byte lines of a team's Float64 fields read whole, the accepting lanes' fields written; see f64_sparse_exact_262144)"
    got, why = spf_traffic()
    if got:
        tf, _, tf_path = got
        gbps = tf["measured_bytes_per_attempt"] * R * iters / (k_ms * 1e-3) / 1e9
        out["traffic"] = {"measured_bytes_per_attempt": tf["measured_bytes_per_attempt"], "ratio_to_algorithmic": tf["traffic_ratio"],
                          "hbm_GBps_at_this_rate": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBS,
                          "source": "%s (committed rocprofv3 FETCH_SIZE x2 + WRITE_SIZE pass of the same workload, commit %s, stamp matches these sources; not measured in this run)" % (tf_path, tf.get("git_commit"))}
        out["bound_frac"] = gbps / HBM_PEAK_GBS
        out["bound_frac_meaning"] = "measured bytes per attempt (committed pass) at this run's rate over the 8 TB/s peak"
        out["wave_issue_frac"] = tf.get("wave_issue_frac")
    else:
        out["traffic"] = None
        out["traffic_note"] = why
        out["bound_frac"] = None
    if O is not None:
        with pinned_core():
            ch = O.init_configs(SEED, 0, 1, N)[0]
            it1 = 1 << 24
            v, n, dt1 = timed_oracle(lambda k: O.standard_mc_spf(X.A, X.J, beta, it1, step, SEED, ch, it0=k * it1), it1)
            out["cpu_one_core"] = {"value": v, "unit": "attempts/s", "kind": "port", "build": O.flavour,
                                   "sample": "1 replica x %d x 2^24 iterations (%.1f s), oracle" % (n, dt1)}
    return out


def secondary_f64_exact_k8(pkg, O, device):
    """The same path at K = 8 (GraphEANormal(L = 8, D = 4), N = 4096, 8192 replicas): the degrees 7 and 8 fuse their pairs of attempts in
    eight-wavefront teams (the registers of a sixteen-wavefront workgroup do not suffice)."""
    L, D, R, beta, iters, step = 8, 4, 8192, 1.0, 1 << 16, 1 << 12
    X = pkg.GraphEANormal(L, D, seed=SEED)
    with pkg.Engine(X, R, device=device) as eng:
        eng.seed(SEED)
        eng.init_spins_random()
        eng.standard_mc_async(beta, iters // 4, step); eng.sync()
        t0 = time.perf_counter()
        eng.standard_mc_async(beta, iters, step); eng.sync()
        dt = time.perf_counter() - t0
        _, k_ms, nl = eng.last_timing()
        _, acc = eng.fetch_results(want_energies=False)
        kname = spf_team_kernel_name(eng, 2 * D)
    a = float(acc.mean()) / iters
    bpa = 8 + a * (10 + 17 * 2 * D)
    out = {"workload": "GraphEANormal(L=8,D=4: N=4096,K=8) standardMC (exact mode) beta=1.0, 8192 replicas, 2^16 iterations per replica",
           "value": R * iters / dt, "unit": "attempts/s", "kernel": kname, "avg_launch_ms": k_ms / max(nl, 1), "launches": nl,
           "acceptance": a, "algorithmic_bytes_per_attempt": bpa, "achieved_GBps": bpa * R * iters / (k_ms * 1e-3) / 1e9,
           "bound": "the executing wavefronts' instruction streams (seven per team: fused pairs in eight-wavefront teams, DESIGN.md 8.8)", "bound_frac": None}
    out["frac"] = out["achieved_GBps"] / HBM_PEAK_GBS
    return out


def secondary_f64_exact_big(pkg, O, device):
    """The same kernel where it is bound by HBM: 262 144 replicas of GraphRRGNormal(N=4096, K=3) — 8.6 GB of Float64 fields, every attempt
    reads and writes the K + 1 field lines of its site whole.  The traffic per attempt is the committed rocprofv3 measurement of this shape."""
    N, K, R, beta, iters, step = 4096, 3, 262144, 1.0, 1 << 14, 1 << 12
    X = pkg.GraphRRGNormal(N, K, seed=SEED)
    with pkg.Engine(X, R, device=device) as eng:
        eng.seed(SEED)
        eng.init_spins_random()
        eng.standard_mc_async(beta, iters // 4, step); eng.sync()
        t0 = time.perf_counter()
        eng.standard_mc_async(beta, iters, step); eng.sync()
        dt = time.perf_counter() - t0
        _, k_ms, nl = eng.last_timing()
        _, acc = eng.fetch_results(want_energies=False)
        kname = spf_team_kernel_name(eng, K)
    a = float(acc.mean()) / iters
    bpa = 8 + a * (10 + 17 * K)
    out = {"workload": "GraphRRGNormal(N=4096,K=3) standardMC (exact mode) beta=1.0, 262144 replicas, 2^14 iterations per replica, energy sample every 4096",
           "value": R * iters / dt, "unit": "attempts/s", "kernel": kname, "avg_launch_ms": k_ms / max(nl, 1), "launches": nl,
           "acceptance": a, "algorithmic_bytes_per_attempt": bpa, "achieved_GBps": bpa * R * iters / (k_ms * 1e-3) / 1e9}
    out["frac"] = out["achieved_GBps"] / HBM_PEAK_GBS
    out["bound"] = "hbm / memory system (measured traffic: random 512-byte lines read whole, the accepting lanes' fields written)"
    got, why = spf_traffic("at_262144_replicas")
    if got:
        big, tf, tf_path = got
        gbps = big["measured_bytes_per_attempt"] * R * iters / (k_ms * 1e-3) / 1e9
        out["traffic"] = {"measured_bytes_per_attempt": big["measured_bytes_per_attempt"], "hbm_GBps_at_this_rate": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBS,
                          "source": "%s at_262144_replicas (committed rocprofv3 FETCH_SIZE x2 + WRITE_SIZE pass of this shape, commit %s, stamp matches these sources; not measured in this run)" % (tf_path, tf.get("git_commit"))}
        out["bound_frac"] = gbps / HBM_PEAK_GBS
        out["bound_frac_meaning"] = "measured HBM traffic at this run's rate over the 8 TB/s peak"
    else:
        out["traffic"] = None
        out["traffic_note"] = why
        out["bound_frac"] = None
    return out


def secondary_f8_rrr(pkg, O, device):
    """SURVEY.md §8f rank 1 at the reference's experiment size (scripts/scripts.jl:23 test_RRG): rrrMC(X::SingleGraph) on GraphRRG(10^4, 3),
    beta = 2, thread-per-replica kernel (rrr_sparse_kernel: DeltaECache{Int,2} + ArraySets per replica in HBM/L2)."""
    N, K, R, beta, iters, step = 10000, 3, 4096, 2.0, 20000, 5000
    X = pkg.GraphRRG(N, K, seed=SEED)
    with pkg.Engine(X, R, device=device) as eng:
        eng.seed(SEED)
        eng.init_spins_random()
        eng.standard_mc(beta, 20 * N, 20 * N, want_energies=False)          # a short quench: the reference's runs start from random spins too
        eng.rrr_mc(beta, iters // 4, step, want_energies=False)
        t0 = time.perf_counter()
        _, acc, staged = eng.rrr_mc(beta, iters, step, want_energies=False)
        dt = time.perf_counter() - t0
        _, k_ms, nl = eng.last_timing()
        C1 = eng.get_config().s[0].copy()
    out = {"workload": "GraphRRG(N=10000,K=3,+-J) rrrMC beta=2.0, 4096 replicas, 20000 iterations per replica (after a 20-sweep Metropolis quench)",
           "value": R * iters / dt, "unit": "iterations/s", "kernel": "rrr_sparse_kernel<false, 2, unsigned short, 8>", "avg_launch_ms": k_ms / max(nl, 1),
           "launches": nl, "acceptance": float(acc.mean()) / iters, "staged_frac": float(staged.mean()) / iters,
           "note": "one thread per replica, every structure of the reference (ArraySet v/pos, T, z) per replica in HBM: apply_move! gathers its K + 1 "
                   "sites stage by stage; profiles/r06/f8_kernels_summary.txt holds the rocprofv3 trace and SQ counters"}
    if O is not None:
        with pinned_core():
            it1, Ji = 1 << 20, X.J.astype(np.int32)
            v, n, dt1 = timed_oracle(lambda k: O.rrr_sparse(X.A, Ji, beta, it1, it1, SEED, C1, it0=k * it1), it1)
            out["cpu_one_core"] = {"value": v, "unit": "iterations/s", "kind": "port", "build": O.flavour,
                                   "sample": "1 replica x %d x 2^20 iterations from a quenched configuration (%.1f s), oracle" % (n, dt1)}
        nc = min(R, host_cores(), CPU_WORKERS_MAX)
        it2, Ji = 1 << 18, X.J.astype(np.int32)
        cpu_all_cores_entry(out, out["cpu_one_core"]["value"], "iterations/s",
                            timed_oracle_all_cores("rrr_sparse", (X.A, Ji, beta, it2, it2, SEED, None), {}, 6, [C1], it2, nc), "x 2^18-iteration calls from the quenched configuration")
    out["bound"] = "latency: one thread per replica, 4 dependent HBM round trips + 1 237 issued instructions per iteration (profiles/r06/f8_floor.md)"
    out["bound_frac"] = 8550.0 / max(k_ms * 1e-3 * 2.4e9 / iters, 1.0)
    out["bound_frac_meaning"] = "the floor of f8_floor.md §1 (8 550 cycles per iteration of a wavefront) over this run's cycles per iteration at 2.4 GHz"
    return out


def secondary_f8_eo(pkg, O, device):
    """SURVEY.md §8f rank 4 at the reference's experiment size: extremal_opt (src/RRRMC.jl:474-521, EOCache{Int,2}: src/DeltaE.jl:412-555) on
    GraphRRG(10^4, 3), tau = 1.3, from random spins — eo_sparse_kernel, one thread per replica (rank table in LDS, staged gather, lazy Cmin)."""
    N, K, R, tau, iters, step = 10000, 3, 4096, 1.3, 20000, 5000
    X = pkg.GraphRRG(N, K, seed=SEED)
    with pkg.Engine(X, R, device=device) as eng:
        eng.seed(SEED)
        eng.init_spins_random()
        C0 = eng.get_config().s[0].copy()
        t0 = time.perf_counter()
        eng.extremal_opt(tau, iters, step)
        dt = time.perf_counter() - t0
        _, k_ms, nl = eng.last_timing()
    out = {"workload": "GraphRRG(N=10000,K=3,+-J) extremal_opt tau=1.3, 4096 replicas, 20000 iterations per replica from random spins",
           "value": R * iters / dt, "unit": "iterations/s", "kernel": "eo_sparse_kernel<unsigned short, 2>", "avg_launch_ms": k_ms / max(nl, 1), "launches": nl,
           "bound": "latency: one thread per replica (profiles/r06/f8_floor.md §3)", "bound_frac": None}
    if O is not None:
        with pinned_core():
            it1, Ji = 1 << 18, X.J.astype(np.int32)
            v, n, dt1 = timed_oracle(lambda k: O.extremal_opt_sparse(X.A, Ji, tau, it1, it1, SEED, C0, it0=k * it1), it1)
            out["cpu_one_core"] = {"value": v, "unit": "iterations/s", "kind": "port", "build": O.flavour,
                                   "sample": "1 replica x %d x 2^18 iterations from random spins, each call a fresh run (%.1f s), oracle" % (n, dt1)}
    return out


def secondary_f8_dbl(pkg, O, device):
    """SURVEY.md §8f rank 3 (second part) at the same size: rrrMC(X::DoubleGraph) (src/RRRMC.jl:221-290) on GraphRRGNormalDiscretized(10^4, 3,
    (-1, 0, 1)), beta = 2 — rrr_dbl_kernel, one thread per replica; the model family rrrMC(DoubleGraph) was designed for."""
    N, K, R, beta, iters, step = 10000, 3, 4096, 2.0, 20000, 5000
    X = pkg.GraphRRGNormalDiscretized(N, K, (-1, 0, 1), seed=SEED)
    with pkg.Engine(X, R, device=device) as eng:
        eng.seed(SEED)
        eng.init_spins_random()
        eng.rrr_mc(beta, iters // 4, step, want_energies=False)
        C1 = eng.get_config().s[0].copy()
        t0 = time.perf_counter()
        _, acc, staged = eng.rrr_mc(beta, iters, step, want_energies=False)
        dt = time.perf_counter() - t0
        _, k_ms, nl = eng.last_timing()
    out = {"workload": "GraphRRGNormalDiscretized(N=10000,K=3,levels -1 0 1) rrrMC beta=2.0, 4096 replicas, 20000 iterations per replica",
           "value": R * iters / dt, "unit": "iterations/s", "kernel": "rrr_dbl_kernel<4, unsigned short>", "avg_launch_ms": k_ms / max(nl, 1), "launches": nl,
           "acceptance": float(acc.mean()) / iters, "staged_frac": float(staged.mean()) / iters,
           "bound": "latency / issue: one thread per replica, the staged step on gathered values (profiles/r06/f8_floor.md §3)", "bound_frac": None}
    if O is not None:
        with pinned_core():
            it1 = 1 << 17
            units, mul, div = O.dfloat_units((-1, 0, 1))
            v, n, dt1 = timed_oracle(lambda k: O.rrr_double_sparse(X.A, X.dJ, X.rJ, units, beta, it1, it1, SEED, C1, it0=k * it1, mul=mul, div=div), it1)
            out["cpu_one_core"] = {"value": v, "unit": "iterations/s", "kind": "port", "build": O.flavour,
                                   "sample": "1 replica x %d x 2^17 iterations (%.1f s), oracle" % (n, dt1)}
    return out


def secondary_f8_cont(pkg, O, device):
    """SURVEY.md §8f rank 4 at the reference's experiment size (scripts/scripts.jl:152 test_RRGCont): rrrMC(X::SingleGraph) on
    GraphRRGNormal(10^4, 3) — DeltaECacheCont + DynamicSampler (src/DeltaE.jl:299-410, src/DynamicSamplers.jl) — cont_wave_kernel, one
    wavefront per replica with the top of the sampler's tree in LDS."""
    N, K, R, beta, iters, step = 10000, 3, 4096, 2.0, 5000, 2500
    X = pkg.GraphRRGNormal(N, K, seed=SEED)
    with pkg.Engine(X, R, device=device) as eng:
        eng.seed(SEED)
        eng.init_spins_random()
        eng.standard_mc(beta, 10 * N, 10 * N, want_energies=False)          # a short quench first
        eng.rrr_mc(beta, iters // 4, step, want_energies=False)
        t0 = time.perf_counter()
        _, acc, staged = eng.rrr_mc(beta, iters, step, want_energies=False)
        dt = time.perf_counter() - t0
        _, k_ms, nl = eng.last_timing()
        C1 = eng.get_config().s[0].copy()
    out = {"workload": "GraphRRGNormal(N=10000,K=3) rrrMC beta=2.0, 4096 replicas, 5000 iterations per replica (after a 10-sweep Metropolis quench)",
           "value": R * iters / dt, "unit": "iterations/s", "kernel": "cont_wave_kernel", "avg_launch_ms": k_ms / max(nl, 1), "launches": nl,
           "acceptance": float(acc.mean()) / iters,
           "bound": "issue: one wavefront per replica, four per SIMD; 1 090 issued instructions per move, the sampler's tree updated in the reference's "
                    "order of Float64 additions (profiles/r06/f8_floor.md)",
           "bound_frac": 4710.0 / max(k_ms * 1e-3 * 2.4e9 * 1024.0 / (R * iters), 1.0),
           "bound_frac_meaning": "the issue cycles of f8_floor.md §2 (4 710 per move of a wavefront) over this run's SIMD cycles per move at 2.4 GHz"}
    if O is not None:
        with pinned_core():
            it1 = 1 << 16
            v, n, dt1 = timed_oracle(lambda k: O.cont_sparse("rrr", X.A, X.J, beta, it1, it1, SEED, C1, it0=k * it1), it1)
            out["cpu_one_core"] = {"value": v, "unit": "iterations/s", "kind": "port", "build": O.flavour,
                                   "sample": "1 replica x %d x 2^16 iterations from a quenched configuration (%.1f s), oracle" % (n, dt1)}
        nc = min(R, host_cores(), CPU_WORKERS_MAX)
        it2 = 1 << 14
        cpu_all_cores_entry(out, out["cpu_one_core"]["value"], "iterations/s",
                            timed_oracle_all_cores("cont_sparse", ("rrr", X.A, X.J, beta, it2, it2, SEED, None), {}, 7, [C1], it2, nc), "x 2^14-iteration calls")
    return out


def eng_kernel_name_quant():
    return "rrr_quant_wave_kernel" if os.environ.get("RRRMC_QUANT_NO_WAVE") != "1" else "rrr_quant_kernel<true>"


def secondary(pkg, O, device):
    out = {}
    for name, fn in (("c3_sk_normal", secondary_c3), ("c4_ea_checkerboard", secondary_c4), ("c4_ea_random_site", secondary_c4_random),
                     ("c5_quant_rrr", secondary_c5),
                     ("f64_sparse_exact", secondary_f64_exact), ("f64_sparse_exact_262144", secondary_f64_exact_big), ("f64_sparse_exact_k8", secondary_f64_exact_k8), ("f64_sparse_fast", secondary_f64_fast), ("f8_rrr_rrg_1e4", secondary_f8_rrr),
                     ("f8_rrr_rrgn_1e4", secondary_f8_cont), ("f8_eo_rrg_1e4", secondary_f8_eo), ("f8_rrr_disc_1e4", secondary_f8_dbl)):
        t0 = time.perf_counter()
        try:
            out[name] = fn(pkg, O, device)
            out[name]["wall_s"] = time.perf_counter() - t0
        except Exception as e:      # a secondary figure never hides the headline line
            out[name] = {"error": repr(e)}
    return out


# ----------------------------------------------------------------------------------------------------------------
def _sig(x, n=5):
    """n significant digits (the line is read by people and parsers alike: 4.1832e12, not 4183219876543.21)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, (list, tuple)):
        return [_sig(v, n) for v in x]
    try:
        return float("%.*g" % (n, float(x)))
    except (TypeError, ValueError):
        return x


def compact_line(full):
    """The ONE line bench.py prints: the contract's keys, `roofline` and `cpu_baseline` as numbers, and one short record per secondary
    workload (all five BASELINE configs fit the tail a driver keeps).  Prose — what a figure means, where a committed measurement came from,
    what a sample was — lives in README.md ("The bench line") and in the full record (stderr, gpurun_out/bench_detail.json)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    out = {k: _sig(full[k], 6) for k in keep if k in full}
    cfg = full.get("config", {})
    out["config"] = {"workload": cfg.get("workload"), "replicas_total": cfg.get("replicas_total"), "acceptance": _sig(cfg.get("acceptance")),
                     "parallelism": cfg.get("parallelism")}
    r = full.get("roofline")
    if r:
        v = r.get("valu") or {}
        out["roofline"] = {"bound": r.get("bound"), "achieved": _sig(r.get("achieved")), "peak": r.get("peak"), "unit": r.get("unit"), "frac": _sig(r.get("frac")),
                           "traffic": _sig(r.get("traffic")), "kernel": r.get("kernel"), "avg_launch_ms": _sig(r.get("avg_launch_ms")), "launches": r.get("launches"),
                           "bytes_per_attempt": _sig(r.get("algorithmic_bytes_per_attempt")), "valu_frac": _sig(v.get("frac")),
                           "valu_frac_2cyc": _sig(v.get("frac_vs_guide_2cycle")), "copy_GBps": _sig(r.get("measured_copy_GBps")),
                           "stamp": r.get("kernel_source_stamp"), "profile": r.get("profile_dir")}
    c = full.get("cpu_baseline")
    if c:
        out["cpu_baseline"] = {"value": _sig(c.get("value")), "unit": c.get("unit"), "cores": c.get("cores"), "kind": c.get("kind"),
                               "sample": c.get("sample_short") or c.get("sample")}
    for k in ("host_gap_ms_per_step", "sync_value", "sync_value_pageable", "ms_per_step_per_rank"):
        if k in full:
            out[k] = _sig(full[k])
    out["verified"] = full.get("verified")
    sec = full.get("secondary")
    if sec:
        short = {}
        for name, e in sec.items():
            if "error" in e:
                short[name] = {"error": str(e["error"])[:80]}
                continue
            rec = {"value": _sig(e.get("value")), "unit": e.get("unit"), "kernel": e.get("kernel"), "frac": _sig(e.get("frac")),
                   "bound_frac": _sig(e.get("bound_frac")), "acc": _sig(e.get("acceptance"), 3)}
            if e.get("cpu_one_core"):
                rec["cpu1"] = _sig(e["cpu_one_core"].get("value"))
            if e.get("cpu_all_cores"):
                rec["cpuN"] = _sig(e["cpu_all_cores"].get("value"))
                rec["cores"] = e["cpu_all_cores"].get("cores")
            if e.get("traffic"):
                rec["B_meas"] = _sig(e["traffic"].get("measured_bytes_per_attempt"))
            short[name] = {k: v for k, v in rec.items() if v is not None}
        out["secondary"] = short
    out["detail"] = full.get("detail_file")
    return out


def emit(full):
    """print the compact line on stdout (exactly one line); the full record on stderr and, when it can be written, in a file"""
    path = os.environ.get("RRRMC_BENCH_DETAIL", os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
        full["detail_file"] = os.path.relpath(path, ROOT)
    except OSError:
        full["detail_file"] = None
    sys.stderr.write("bench.py full record: %s\n" % json.dumps(full))
    line = json.dumps(compact_line(full))
    if len(line) > LINE_LIMIT_BYTES:
        sys.stderr.write("bench.py: the printed line is %d bytes (limit %d)\n" % (len(line), LINE_LIMIT_BYTES))
    print(line)
    sys.stdout.flush()


# ----------------------------------------------------------------------------------------------------------------
def run_rank(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:       # every rank leaves (non-zero): nobody is left waiting at a barrier, and no line is printed under a wrong label
        sys.stderr.write("bench.py: rank %d: --gpus %d but WORLD_SIZE=%d (one rank per GPU): refusing to run\n" % (rank, args.gpus, world))
        return 2
    dist = None
    # RRRMC_BENCH_BACKEND=gloo is a debugging aid only: it lets several ranks share one GPU (RCCL refuses that) so that the
    # multi-rank code path can be exercised on a 1-GPU box; the driver's multi-GPU runs use the default, RCCL ("nccl").
    backend = os.environ.get("RRRMC_BENCH_BACKEND", "nccl")
    # RRRMC_BENCH_FORCE_DIST=1: run the distributed code path (process group, RCCL all_reduce / all_gather) with a single rank too
    if world > 1 or os.environ.get("RRRMC_BENCH_FORCE_DIST") == "1":
        import torch
        import torch.distributed as dist
        if backend == "nccl":
            if torch.cuda.device_count() <= local_rank:
                sys.stderr.write("bench.py: rank %d has no device %d (%d visible)\n" % (rank, local_rank, torch.cuda.device_count()))
                return 2
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend)
    n_gpus = world

    # rank 0 (re)builds if the in-tree library is stale, everybody else waits for it
    if rank == 0:
        entry.build_only()
    if dist is not None:
        dist.barrier()
    pkg = entry.load_package()
    if pkg.lib().rrrmc_device_count() <= local_rank:
        sys.stderr.write("bench.py: device %d is not visible to the HIP library\n" % local_rank)
        return 2
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

    verify = rank == 0 and not args.no_verify
    if verify:
        eng.snapshot_reserve(1)          # device buffer for the configuration the timed region starts from (allocated before the warm-up)
    for _ in range(args.warmup):
        eng.standard_mc_async(BETA, args.iters, SAMPLE_STEP)
    if verify:
        eng.snapshot_store(0)            # device-to-device copy queued behind the warm-up: no host round trip, the GPU never idles before t0
    # one HIP-event pair per sweep launch of the timed region, read after it (no sync inside); the events are created here, not in the region
    eng.timing_accumulate(True, reserve_launches=4 * args.steps)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.standard_mc_async(BETA, args.iters, SAMPLE_STEP)
    barrier()
    dt_local = time.perf_counter() - t0
    dt = dt_local
    sweep_ms, launches = eng.timing_total()
    eng.timing_accumulate(False)
    per_rank_ms = [1e3 * dt_local / max(args.steps, 1)]
    if dist is not None:
        import torch
        dev = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([dt_local], dtype=torch.float64, device=dev)
        ts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(ts, t)
        per_rank_ms = [1e3 * float(x.item()) / max(args.steps, 1) for x in ts]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    Es, acc = eng.fetch_results()
    C_end = eng.get_config().s.copy() if verify else None
    C_start = eng.snapshot_get(0).s.copy() if verify else None    # what the timed region started from, read back after it
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
            "ms_per_step": 1e3 * dt / max(args.steps, 1), "ms_per_step_per_rank": per_rank_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
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
            default_shape = R == REPLICAS_PER_GPU and args.iters == ITERS
            tf, tf_path, tf_state = load_profile_json("traffic.json") if default_shape else (None, None, None)
            vm, vm_path, vm_state = load_profile_json("valu_model.json") if default_shape else (None, None, None)
            if tf_state == "stale":
                tf = None        # measured on other kernel sources: do not report it for this build
            if vm_state == "stale":
                vm = None
            roof = {
                # The replica state never leaves LDS, so the kernel is bound by VALU issue, not by HBM: `achieved` / `frac` are the
                # SURVEY.md §8d fixed-width ALGORITHMIC bytes over the kernel time (a throughput normalisation against the 8 TB/s
                # the north star names), `traffic` is what really crosses HBM, and `valu` is the roofline that binds.
                "bound": "valu", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "achieved_is": "algorithmic bytes (SURVEY.md §8d: 1 + a(3+3K) per attempt) / kernel time — not measured traffic",
                "traffic": tf["hbm_bytes_per_launch"] if tf else None,
                "traffic_source": ("%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE of this workload, committed; not measured in this run; "
                                   "source stamp %s, commit %s)" % (tf_path, tf_state, tf.get("git_commit"))) if tf
                                  else ("dropped: %s was measured on other kernel sources" % tf_path if tf_state == "stale" else None),
                "kernel_source_stamp": source_stamp(), "git_head": git_head(), "profile_dir": os.path.dirname(tf_path) if tf else None,
                "algorithmic_bytes_per_launch": bytes_per_attempt * per_launch_attempts,
                "kernel": "sweep_kernel<3, 1>", "avg_launch_ms": avg_ms, "launches": int(launches),
                "algorithmic_bytes_per_attempt": bytes_per_attempt}
            if vm:
                # cycle-weighted VALU utilisation: wave-instructions per launch (PMC, committed) x the mean issue cost of the
                # kernel's instruction mix (ISA histogram x tools/ubench/valu_rates.hip) over SIMDs x kernel cycles of THIS run
                cyc = avg_ms * 1e-3 * vm["clock_hz"]
                roof["valu"] = {"wave_insts_per_launch": vm["valu_insts_per_launch"], "mean_issue_cycles": vm["mean_issue_cycles"],
                                "simds": vm["simds"], "clock_hz": vm["clock_hz"], "kernel_cycles": cyc,
                                "achieved": vm["valu_insts_per_launch"] * vm["mean_issue_cycles"] / cyc, "peak": vm["simds"],
                                "unit": "busy SIMDs", "frac": vm["valu_insts_per_launch"] * vm["mean_issue_cycles"] / (vm["simds"] * cyc),
                                # the same count priced at the guide's 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md), whatever the opcode
                                "frac_vs_guide_2cycle": vm["valu_insts_per_launch"] * 2.0 / (vm["simds"] * cyc),
                                "source": "%s (SQ_INSTS_VALU per launch, ISA histogram, ubench issue costs; committed; source stamp %s, commit %s)"
                                          % (vm_path, vm_state, vm.get("git_commit"))}
                if vm.get("occupancy_factor_4_waves"):
                    # the kernel runs 4 waves per SIMD (16-wave workgroups, one per CU): the producer task measured in isolation issues
                    # 7 % slower there than at the 8 waves per SIMD the per-opcode costs were taken at (tools/ubench/producer_task.hip)
                    roof["valu"]["frac_at_4_waves_per_simd"] = roof["valu"]["frac"] * vm["occupancy_factor_4_waves"]
            out["roofline"] = roof
            # what the timed region spent outside the dominant kernel, per step: energy(X, C) + planner at the head of every call, launch gaps
            out["host_gap_ms_per_step"] = 1e3 * dt / max(args.steps, 1) - sweep_ms / max(args.steps, 1)
            try:      # after the timed region: the box's own copy bandwidth, for reference only (peak stays the nominal figure)
                bw = device_copy_bandwidth(pkg, local_rank)
                roof["measured_copy_GBps"] = bw
            except Exception as e:      # never let the side measurement hide the bench line
                roof["measured_copy_GBps"] = None
                sys.stderr.write("device copy bandwidth not measured: %r\n" % (e,))
    if rank == 0 and world == 1 and not args.no_secondary:
        # The reference-style SYNCHRONOUS call (src/RRRMC.jl:126 returns Es and C per call): rrrmc_standard_mc + rrrmc_get_spins, i.e. the
        # sweep plus the transposes and the PCIe transfers of the energy samples (R x 1024 Int64) and the configuration.  Never `value`.
        nsync = 5
        res = (pkg.pinned_empty((R, args.iters // SAMPLE_STEP), np.int64), pkg.pinned_empty((R,), np.int64))      # caller-owned result buffers,
        cfg = pkg.Config(N_SITES, R, s=pkg.pinned_empty((R, (N_SITES + 63) // 64), np.uint64))                   # page-locked (rrrmc_host_alloc)
        eng.standard_mc(BETA, args.iters, SAMPLE_STEP, out=res); eng.get_config(cfg)                              # (first touch of the buffers)
        t1 = time.perf_counter()
        for _ in range(nsync):
            eng.standard_mc(BETA, args.iters, SAMPLE_STEP, out=res)
            eng.get_config(cfg)
        out["sync_value"] = float(R) * args.iters * nsync / (time.perf_counter() - t1)
        t1 = time.perf_counter()
        for _ in range(2):
            eng.standard_mc(BETA, args.iters, SAMPLE_STEP)
            eng.get_config()
        out["sync_value_pageable"] = float(R) * args.iters * 2 / (time.perf_counter() - t1)
        out["sync_value_is"] = ("attempts/s of %d synchronous rrrmc_standard_mc + rrrmc_get_spins calls into page-locked result buffers (rrrmc_host_alloc), "
                                "results on the host after every call: PCIe inclusive; sync_value_pageable = the same with fresh ordinary numpy arrays per call" % nsync)
    eng.close()
    rc = 0
    if rank == 0:
        if not args.no_cpu_baseline or verify:
            os.environ.setdefault("RRRMC_ORACLE_NATIVE", "1")        # the baseline build is -O3 -march=native, made on this host
        if verify:
            # the line checks what it timed: replicas {first, middle, last} of this rank's shard, the whole timed region, against the oracle
            O = entry.load_oracle()
            v = verify_timed_region(O, X, C_start, C_end, Es, acc, args.warmup * args.iters, args.steps, args.iters,
                                    sorted({0, R // 2 - 1 if R > 1 else 0, R - 1}), r0)
            out["verified"] = all(v["ok"])
            out["verification"] = v
            if not out["verified"]:
                sys.stderr.write("bench.py: the timed run does NOT reproduce the oracle's chains: %r\n" % (v,))
                rc = 3
        else:
            out["verified"] = None
        if out.get("host_gap_ms_per_step", 0.0) > HOST_GAP_LIMIT_MS and R == REPLICAS_PER_GPU and args.iters == ITERS and world == 1:
            sys.stderr.write("bench.py: %.3f ms per step of the timed region lie outside the sweep kernel (limit %.1f ms): the line does not "
                             "measure the kernel\n" % (out["host_gap_ms_per_step"], HOST_GAP_LIMIT_MS))
            rc = rc or 4
        if world == 1 and not args.no_cpu_baseline:      # reported at N = 1 only (rank 0's host cores)
            O = entry.load_oracle()
            out["cpu_baseline"] = cpu_baseline(O, X)
        if world == 1 and not args.no_secondary:
            out["secondary"] = secondary(pkg, None if args.no_cpu_baseline else entry.load_oracle(), local_rank)
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)       # 200 x 9.8 ms: a timed region of ~2 s
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--iters", type=int, default=ITERS)
    ap.add_argument("--replicas", type=int, default=REPLICAS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle replay of the timed region (3 replicas; ~10 s at the default size)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch(args, sys.argv[1:])
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
