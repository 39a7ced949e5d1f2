"""GPU tests of the host orchestration around the sweep kernels: queued async calls (no sync in between), two contexts driven
concurrently in one process (one per device when the box has two), per-launch timing accumulation, full-width launch shapes of
BASELINE.json configs 4 and 5, and two rank PROCESSES running the HIP path on their shards."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_queued_async_calls_match_one_long_chain(pkg, oracle):
    """Several rrrmc_standard_mc_async calls queued WITHOUT a sync in between (the shape bench.py's timed region has) continue one
    chain: same shape twice (the device chunk list is reused), then two other shapes (the pinned staging buffer is rewritten while
    earlier calls are still queued).  The final state equals the oracle's after the same total, and the last call's energies equal
    the oracle's for that stretch."""
    seed, N, R = 0xA5, 1024, 96
    X = pkg.GraphRRG(N, 3, seed=seed)
    calls = [(1 << 17, 1 << 10), (1 << 17, 1 << 10), (50000, 777), (1 << 16, 1 << 9), (1 << 16, 1 << 9), (3000, 64)]
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        for iters, step in calls:
            eng.standard_mc_async(1.0, iters, step)
        eng.sync()
        Es, acc = eng.fetch_results()
        C1 = eng.get_config()
        assert eng.iterations_done() == sum(c[0] for c in calls)
    A, J = X.A, X.J.astype(np.int32)
    ch, it0 = C0.s, 0
    for iters, step in calls:
        ref = oracle.standard_mc_sparse_batch(A, J, 1.0, iters, step, seed, ch, it0=it0)
        ch, it0 = ref[1], it0 + iters
    assert (C1.s == ch).all()
    assert (Es == ref[0]).all() and (acc == ref[2]).all()


def test_timing_accumulation_counts_every_launch(pkg):
    X = pkg.GraphRRG(512, 3, seed=3)
    with pkg.Engine(X, 64) as eng:
        eng.seed(3)
        eng.init_spins_random()
        eng.timing_accumulate(True)
        for _ in range(5):
            eng.standard_mc_async(1.0, 1 << 14, 1 << 10)
        ms, n = eng.timing_total()
        assert n == 5 and ms > 0.0
        _, last_ms, nl = eng.last_timing()
        assert nl == 1 and 0.0 < last_ms <= ms
        eng.timing_accumulate(False)
        with pytest.raises(pkg.RRRMCError):
            eng.timing_total()
        eng.standard_mc_async(1.0, 1 << 14, 1 << 10)
        eng.sync()
        assert eng.last_timing()[2] == 1


def test_two_contexts_driven_concurrently(pkg, oracle):
    """Two contexts in ONE process — shards [0, R) and [R, 2R) of one job — fed alternately with async calls: on device 0 and 1 when
    the box has two GPUs (the per-device kernel attributes are raised on both), both on device 0 otherwise.  Each equals the
    oracle's chains for its global replica ids; N = 4096 needs the > 64 KiB dynamic-LDS attribute."""
    seed, N, R = 0x5EED, 4096, 64
    X = pkg.GraphRRG(N, 3, seed=seed)
    dev1 = 1 if pkg.lib().rrrmc_device_count() > 1 else 0
    iters, step = 1 << 15, 1 << 12
    with pkg.Engine(X, R, device=0, replica0=0) as e0, pkg.Engine(X, R, device=dev1, replica0=R) as e1:
        for e in (e0, e1):
            e.seed(seed)
            e.init_spins_random()
        C0 = [e0.get_config(), e1.get_config()]
        for _ in range(3):
            e0.standard_mc_async(1.0, iters, step)
            e1.standard_mc_async(1.0, iters, step)
        e1.sync()
        e0.sync()
        res = [e0.fetch_results(), e1.fetch_results()]
        C1 = [e0.get_config(), e1.get_config()]
    A, J = X.A, X.J.astype(np.int32)
    assert (C0[1].s == oracle.init_configs(seed, R, R, N)).all()
    for k in (0, 1):
        ch, it0 = C0[k].s, 0
        for _ in range(3):
            ref = oracle.standard_mc_sparse_batch(A, J, 1.0, iters, step, seed, ch, it0=it0, replica0=k * R)
            ch, it0 = ref[1], it0 + iters
        assert (C1[k].s == ch).all() and (res[k][0] == ref[0]).all() and (res[k][1] == ref[2]).all()


_RANK_SCRIPT = r"""
import os, sys, json
sys.path.insert(0, {root!r})
import numpy as np
import torch.distributed as dist
import __graft_entry__ as g
pkg = g.load_package()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
seed, N, Rtot, iters, step = 77, 512, 160, 1 << 14, 1 << 10
X = pkg.GraphRRG(N, 3, seed=seed)
r0, rl = pkg.shard_bounds(Rtot, world, rank)
dev = rank % max(pkg.lib().rrrmc_device_count(), 1)
with pkg.Engine(X, rl, device=dev, replica0=r0) as eng:
    eng.seed(seed); eng.init_spins_random()
    Es, acc = eng.standard_mc(1.0, iters, step)
    C1 = eng.get_config()
E_all = pkg.gather_replica_major(Es, Rtot, dist)
acc_all = pkg.gather_replica_major(acc, Rtot, dist)
s_all = pkg.gather_replica_major(C1.s, Rtot, dist)
if rank == 0:
    np.savez({out!r}, Es=E_all, acc=acc_all, s=s_all)
dist.barrier()
dist.destroy_process_group()
"""


def test_two_rank_processes_run_the_hip_path_on_their_shards(pkg, oracle, tmp_path):
    """The multi-GPU layout end to end on hardware: two rank PROCESSES (gloo rendezvous; device = rank when the box has two GPUs,
    both on device 0 otherwise), each running the HIP sampler on its shard (global replica ids), results gathered replica-major.
    The gathered job equals the oracle's 160 chains — i.e. the one-process job."""
    sys.path.insert(0, ROOT)
    import bench
    out = str(tmp_path / "gathered.npz")
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT.format(root=ROOT, out=out))
    rc, _ = bench.spawn_ranks(2, [], child_cmd=[sys.executable, str(script)], timeout=600)
    assert rc == 0
    got = np.load(out)
    seed, N, Rtot, iters, step = 77, 512, 160, 1 << 14, 1 << 10
    X = pkg.GraphRRG(N, 3, seed=seed)
    ref = oracle.standard_mc_sparse_batch(X.A, X.J.astype(np.int32), 1.0, iters, step, seed, oracle.init_configs(seed, 0, Rtot, N))
    assert (got["Es"] == ref[0]).all() and (got["s"] == ref[1]).all() and (got["acc"] == ref[2]).all()


def test_bench_gpus2_standalone_launcher_on_hardware():
    """`python bench.py --gpus 2` WITHOUT torchrun: spawns two ranks and reports n_gpus = 2 with both shards' replicas.  On a 1-GPU
    box the two ranks share the device (RRRMC_BENCH_SHARE_GPU=1: gloo rendezvous); without that switch the request is refused."""
    env = dict(os.environ)
    import torch
    two = torch.cuda.device_count() >= 2
    if not two:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 2 and "only 1 GPU" in r.stderr and not r.stdout.strip()
        env["RRRMC_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--iters", str(1 << 18),
                        "--replicas", "2048", "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["replicas_total"] == 4096 and len(line["ms_per_step_per_rank"]) == 2
    assert line["value"] > 0 and 0.05 < line["config"]["acceptance"] < 0.15
    assert line["roofline"]["launches"] == 3


def test_full_width_config4_three_groups_against_oracle(pkg, oracle):
    """BASELINE config 4 at one GPU's launch shape (GraphEA L = 64, D = 3, 512 replicas = 16 replica groups): the first, a middle
    and the last group's first / last replicas against the oracle over 2 sweeps, incl. the per-replica accepted counts."""
    L_, D, R, seed = 64, 3, 512, 0x5EED
    X = pkg.GraphEA(L_, D, seed=seed)
    color = pkg.checkerboard_coloring(L_, D)
    A, J = X.A, X.J.astype(np.int32)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        eng.set_coloring(color)
        eng.colored_count_accepted(True)
        C0 = eng.get_config()
        eng.colored_sweeps_async(1.0, 2, 1)
        eng.sync()
        Es, acc = eng.fetch_results()
        C1 = eng.get_config()
        eng.colored_count_accepted(False)
        eng.colored_sweeps_async(1.0, 1, 1)
        eng.sync()
        assert (eng.fetch_results()[1] == -1).all()
    for r in (0, 31, 7 * 32 + 5, 511):
        Es_ref, ch_ref, acc_ref = oracle.colored_sweeps_sparse(A, J, color, 1.0, 2, 1, seed, C0.s[r], replica=r)
        assert (Es[r] == Es_ref).all() and (C1.s[r] == ch_ref).all() and acc[r] == acc_ref
    a = acc / (2.0 * X.N)
    assert 0.1 < a.mean() < 0.5 and a.std() < 0.01


def test_full_width_config5_replicas_against_oracle(pkg, oracle):
    """BASELINE config 5 at one GPU's launch shape (GraphQuant(GraphRRG(1024, 3), M = 32), 128 replicas, one workgroup per
    replica, state in LDS): replicas 0, 64 and 127 against the oracle, incl. the DeltaECache classes after the run."""
    Nk, M, R, seed, beta, Gamma = 1024, 32, 128, 0x5EED, 2.0, 0.5
    iters, step = 1 << 13, 1 << 10
    X = pkg.GraphQuant(pkg.GraphRRG(Nk, 3, seed=seed), M, Gamma, beta)
    with pkg.Engine(X, R) as eng:
        eng.seed(seed)
        eng.init_spins_random()
        C0 = eng.get_config()
        Es, acc, staged = eng.rrr_mc(beta, iters, step)
        C1 = eng.get_config()
        pos, sizes = eng.rrr_cache()
    A, J = X.X1.A, X.X1.J.astype(np.int32)
    for r in (0, 64, 127):
        ref = oracle.rrr_mc_quant(A, J, M, X.fourK, beta, iters, step, seed, C0.s[r], replica=r, want_cache=True)
        assert (Es[r] == ref[0]).all() and (C1.s[r] == ref[1]).all() and acc[r] == ref[2] and staged[r] == ref[3]
        assert (pos[r] == ref[4]).all() and (sizes[r] == ref[5]).all()
