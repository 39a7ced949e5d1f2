"""The N > 1 path on CPU: two gloo ranks shard a replica job by global replica id, each computes its shard (with the
oracle standing in for the device, which is absent here), and the product's gather assembles the observables.
The assembled result must equal the unsharded run: sharding invariance of the random-stream contract plus the
collective plumbing bench.py uses under RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_and_align(pkg):
    for R, world in [(8192, 8), (8192, 2), (100, 2), (33, 4), (1, 2), (64, 3)]:
        spans = [pkg.shard_bounds(R, world, r) for r in range(world)]
        assert sum(n for _, n in spans) == R
        pos = 0
        for r0, n in spans:
            assert r0 % 32 == 0
            if n:
                assert r0 == pos
                pos += n
    with pytest.raises(ValueError):
        pkg.shard_bounds(10, 2, 2)


def _worker(rank, world, port, R_total, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    import __graft_entry__ as entry
    import oracle as O
    pkg = entry.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seed, N, beta, iters, step = 77, 64, 1.0, 3000, 100
    A = O.gen_rrg(N, 3, seed)
    J = O.gen_couplings(A, seed)
    r0, n = pkg.shard_bounds(R_total, world, rank)
    ch0 = O.init_configs(seed, r0, n, N)
    Es, ch1, acc = O.standard_mc_sparse_batch(A, J, beta, iters, step, seed, ch0, replica0=r0)
    Es_all = pkg.gather_replica_major(Es, R_total, dist)
    acc_all = pkg.gather_replica_major(acc, R_total, dist)
    ch_all = pkg.gather_replica_major(ch1, R_total, dist)
    if rank == 0:
        np.savez(out_path, Es=Es_all, acc=acc_all, ch=ch_all)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_job_equals_unsharded(oracle, tmp_path):
    import torch.multiprocessing as mp
    R_total = 80                                    # 3 groups: ranks get 64 and 16 replicas
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "gathered.npz")
    mp.spawn(_worker, args=(2, port, R_total, out), nprocs=2, join=True)
    got = np.load(out)
    seed, N = 77, 64
    A = oracle.gen_rrg(N, 3, seed)
    J = oracle.gen_couplings(A, seed)
    ch0 = oracle.init_configs(seed, 0, R_total, N)
    Es, ch1, acc = oracle.standard_mc_sparse_batch(A, J, 1.0, 3000, 100, seed, ch0)
    assert (got["Es"] == Es).all() and (got["acc"] == acc).all() and (got["ch"] == ch1).all()
