"""Multi-GPU: replicas are independent chains, so a job shards by replica id with no data-path collective
(SURVEY.md §8e).  One process per GPU; the only exchange is the final gather of observables over
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Random streams are addressed by the GLOBAL replica id, so results do not depend on the number of shards;
shard boundaries are multiples of 32 replicas (one bit-sliced group = one workgroup).
"""
import numpy as np

GROUP = 32


def shard_bounds(R_total, world, rank):
    """Contiguous shard [replica0, replica0 + R_local) of rank `rank`: groups of 32 replicas dealt as evenly as possible."""
    if not (0 <= rank < world):
        raise ValueError("rank %d out of range for world size %d" % (rank, world))
    groups = (R_total + GROUP - 1) // GROUP
    base, extra = divmod(groups, world)
    g0 = rank * base + min(rank, extra)
    g1 = g0 + base + (1 if rank < extra else 0)
    r0, r1 = g0 * GROUP, min(g1 * GROUP, R_total)
    return r0, max(r1 - r0, 0)


def gather_replica_major(local, R_total, dist, device=None):
    """All-gather a per-replica array (first axis = this rank's replicas) into [R_total, ...] on every rank."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    local = np.ascontiguousarray(local)
    tail = local.shape[1:]
    counts = [shard_bounds(R_total, world, r)[1] for r in range(world)]
    assert local.shape[0] == counts[rank], "local array does not match this rank's shard"
    width = max(counts)
    pad = np.zeros((width,) + tail, local.dtype)
    pad[: local.shape[0]] = local
    # moved as raw bytes: the collectives do not take every dtype (e.g. the uint64 BitVector chunks)
    t = torch.from_numpy(pad.reshape(width, -1).view(np.uint8))
    if device is not None:
        t = t.to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)                        # the single collective of a sampling job
    parts = [np.ascontiguousarray(o.cpu().numpy()).view(local.dtype).reshape((width,) + tail)[: counts[r]]
             for r, o in enumerate(outs)]
    return np.concatenate(parts, axis=0)
