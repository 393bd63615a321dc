"""
The N > 1 path on CPU: two gloo processes shard a batch of objects by
contiguous blocks, compute per-object records, and all-gather them
(ngmix_amd.distributed, the same helpers bench.py uses over RCCL).
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_objects, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from ngmix_amd import distributed as nd
    r, w, _ = nd.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)

    def compute(lo, hi):
        # a stand-in for a per-object kernel result: a record that is a pure
        # function of the global object index
        idx = torch.arange(lo, hi, dtype=torch.float64)
        return torch.stack([idx, idx ** 2, -idx, idx * 0 + rank], dim=1)

    full = nd.gather_object_results(compute, n_objects, (4,))
    lo, hi = nd.shard_bounds(n_objects, rank, world)
    # async form on an even split
    even_n = world * 5
    lo2, hi2 = nd.shard_bounds(even_n, rank, world)
    out, work = nd.allgather_records(compute(lo2, hi2), n_objects=even_n,
                                     async_op=True)
    work.wait()
    np.save(os.path.join(outdir, "full_%d.npy" % rank), full.numpy())
    np.save(os.path.join(outdir, "even_%d.npy" % rank), out.numpy())
    np.save(os.path.join(outdir, "bounds_%d.npy" % rank), np.array([lo, hi]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_objects", [10, 7, 1])
def test_shard_and_allgather_world2(tmp_path, n_objects):
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_objects, str(tmp_path)), nprocs=world,
             join=True)
    idx = np.arange(n_objects, dtype="f8")
    from ngmix_amd.distributed import shard_bounds
    owner = np.zeros(n_objects)
    covered = np.zeros(n_objects, dtype=int)
    for r in range(world):
        lo, hi = np.load(tmp_path / ("bounds_%d.npy" % r))
        assert (lo, hi) == shard_bounds(n_objects, r, world)
        owner[lo:hi] = r
        covered[lo:hi] += 1
    assert np.all(covered == 1)          # a partition: every object exactly once
    expect = np.stack([idx, idx ** 2, -idx, owner], axis=1)
    for r in range(world):
        full = np.load(tmp_path / ("full_%d.npy" % r))
        np.testing.assert_array_equal(full, expect)   # same on every rank
        even = np.load(tmp_path / ("even_%d.npy" % r))
        assert even.shape == (10, 4)
        np.testing.assert_array_equal(even[:, 0], np.arange(10.0))


def test_shard_bounds_properties():
    from ngmix_amd.distributed import shard_bounds, shard_sizes
    for n in (0, 1, 7, 8, 100000, 1000003):
        for w in (1, 2, 4, 8):
            sizes = shard_sizes(n, w)
            assert sum(sizes) == n and len(sizes) == w
            assert max(sizes) - min(s for s in sizes if s or True) <= -(-n // w)
            prev = 0
            for r in range(w):
                lo, hi = shard_bounds(n, r, w)
                assert lo == prev and hi >= lo
                prev = hi
            assert prev == n
