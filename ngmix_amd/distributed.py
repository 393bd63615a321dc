"""
Multi-GPU: objects are independent, so the batch is sharded by contiguous
blocks of objects, one process per GPU (torch.distributed; backend "nccl" is
RCCL on ROCm, over xGMI inside a node).  There is no data-path collective; the
only exchange is the all-gather of fixed-size per-object result records
(32 B for loglike, 584 B for admom, ...) named by north_star.
On the fully connected xGMI mesh an all-gather of B total bytes moves B/8 over
each link in one hop: 1M admom records = 584 MB -> ~0.5 ms.

The helpers work on any backend, so the N > 1 path is covered on CPU with
gloo (tests/test_distributed_gloo.py).
"""
import numpy as np


def shard_bounds(n_objects, rank, world_size):
    """contiguous block [lo, hi) of rank: ceil(N/world) objects per rank,
    the tail ranks possibly short or empty"""
    per = -(-int(n_objects) // int(world_size))
    lo = min(rank * per, n_objects)
    hi = min(lo + per, n_objects)
    return lo, hi


def shard_sizes(n_objects, world_size):
    return [b - a for a, b in (shard_bounds(n_objects, r, world_size)
                               for r in range(world_size))]


def allgather_records(local, n_objects=None, group=None, out=None, async_op=False):
    """
    Gather per-object result records from every rank, in object order.

    local: (n_local, ...) tensor of this rank's records (its shard_bounds
    block).  Equal shards use one all_gather_into_tensor; uneven shards are
    padded to the largest shard and trimmed after the gather.  Returns the
    (n_objects, ...) tensor (and the work handle when async_op).
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if n_objects is None:
        n_objects = local.shape[0] * world
    sizes = shard_sizes(n_objects, world)
    assert local.shape[0] == sizes[rank], (local.shape[0], sizes[rank])
    per = max(sizes)
    tail = tuple(local.shape[1:])
    even = all(s == per for s in sizes)
    if even:
        if out is None:
            out = torch.empty((n_objects,) + tail, dtype=local.dtype,
                              device=local.device)
        work = dist.all_gather_into_tensor(out, local.contiguous(), group=group,
                                           async_op=async_op)
        return (out, work) if async_op else out
    padded = torch.zeros((per,) + tail, dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    buf = torch.empty((world * per,) + tail, dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(buf, padded, group=group, async_op=False)
    pieces = [buf[r * per:r * per + sizes[r]] for r in range(world)]
    full = torch.cat(pieces, dim=0)
    if out is not None:
        out.copy_(full)
        full = out
    return (full, work) if async_op else full


def init_from_env(backend=None):
    """initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (as set
    by torch.distributed.run); returns (rank, world_size, local_rank)"""
    import os
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        # testing hook: NGMIX_DIST_BACKEND=gloo runs several ranks on one GPU
        # (RCCL refuses two ranks per device), local_rank wraps around then
        backend = os.environ.get("NGMIX_DIST_BACKEND", backend)
        if torch.cuda.is_available():
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, **kw)
    return rank, world, local_rank


def gather_object_results(compute_shard, n_objects, record_shape, dtype=None,
                          device=None, group=None):
    """
    Run compute_shard(lo, hi) -> (hi-lo, *record_shape) tensor on this rank's
    block of objects and all-gather the records so every rank holds all
    n_objects results.  Single-process (no process group): just computes.
    """
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return compute_shard(0, n_objects)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(n_objects, rank, world)
    local = compute_shard(lo, hi)
    if local is None or local.shape[0] == 0:
        local = torch.empty((0,) + tuple(record_shape),
                            dtype=dtype or torch.float64, device=device)
    return allgather_records(local, n_objects=n_objects, group=group)
