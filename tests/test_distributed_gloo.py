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
    # async form on an uneven split: wait() also trims the padding
    odd_n = world * 5 - 1
    lo3, hi3 = nd.shard_bounds(odd_n, rank, world)
    odd, work3 = nd.allgather_records(compute(lo3, hi3), n_objects=odd_n, async_op=True)
    work3.wait()
    assert odd.shape == (odd_n, 4)
    assert torch.equal(odd[:, 0], torch.arange(odd_n, dtype=torch.float64))
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


_RANK_SCRIPT = """
import json, os, sys, time
sys.path.insert(0, %(root)r)
mode = sys.argv[1]
rank = int(os.environ["RANK"])
if mode == "fail" and rank == 1:
    sys.exit(7)
if mode == "fail":
    time.sleep(120)          # would hang in a collective; the launcher ends it
import torch
import torch.distributed as dist
from ngmix_amd import distributed as nd
r, w, lr = nd.init_from_env(backend="gloo")
assert r == rank == lr and w == int(os.environ["WORLD_SIZE"])
n = 2 * w + 1
lo, hi = nd.shard_bounds(n, r, w)
rec = torch.arange(lo, hi, dtype=torch.float64)[:, None] * torch.ones(1, 3, dtype=torch.float64)
full = nd.gather_object_results(lambda a, b: rec, n, (3,))
if r == 0:
    print(json.dumps({"n_gpus": w, "sum": float(full.sum()), "backend": dist.get_backend()}))
    sys.stdout.flush()
dist.barrier()
dist.destroy_process_group()
"""


def test_launch_local_ranks(tmp_path, capfd):
    """the launcher bench.py --gpus N uses when it is not started by
    torch.distributed.run: N child processes with the rendezvous environment,
    rank 0's line on stdout, exit status 0"""
    import json
    from ngmix_amd import distributed as nd
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT % {"root": ROOT})
    env_before = dict(os.environ)
    os.environ.pop("MASTER_PORT", None)
    os.environ["NGMIX_DIST_BACKEND"] = "gloo"
    try:
        rc = nd.launch_local_ranks(str(script), ["ok"], 3, need_gpus=False, timeout=120)
    finally:
        os.environ.clear()
        os.environ.update(env_before)
    assert rc == 0
    lines = [ln for ln in capfd.readouterr().out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["backend"] == "gloo"
    assert d["sum"] == 3 * sum(range(7))


def test_launch_local_ranks_failure_propagates(tmp_path, capfd):
    """one rank dying ends the job with its status instead of leaving the
    others in a collective"""
    import time
    from ngmix_amd import distributed as nd
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT % {"root": ROOT})
    t0 = time.time()
    rc = nd.launch_local_ranks(str(script), ["fail"], 2, need_gpus=False, timeout=100)
    assert rc == 7
    assert time.time() - t0 < 60
    assert "exited with status 7" in capfd.readouterr().err


def test_bench_gpus_flag_is_honoured_or_refused():
    """python bench.py --gpus 2 must never measure one GPU and label it two:
    here (no GPU) it has to stop with a clear message and a non-zero status"""
    import subprocess
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the launch would really run")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                        "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=280, cwd=ROOT)
    assert p.returncode != 0
    assert "GPU" in p.stderr and "2 ranks asked for" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def _columns_worker(rank, world, port, n_objects, outdir):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ngmix_amd.distributed import allgather_columns, shard_bounds
        lo, hi = shard_bounds(n_objects, rank, world)
        idx = np.arange(lo, hi)
        cols = {"pars": np.stack([idx * 1.5, -idx * 1.0, idx ** 2 * 1.0], axis=1),
                "flags": (idx % 3).astype(np.int64),
                "ok": (idx % 2 == 0),
                "cov": (idx[:, None, None] * np.ones((1, 2, 2))).astype("f8")}
        full = allgather_columns(cols, n_objects=n_objects)
        np.savez(os.path.join(outdir, "cols_%d.npz" % rank), **full)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_objects,world", [(11, 2), (5, 3), (1, 2)])
def test_allgather_columns_is_one_collective_for_many_records(tmp_path, n_objects, world):
    """distributed.allgather_columns: several per-object arrays of different
    widths and dtypes (the psf / guess / fit records of a pipeline) packed into
    ONE all-gather, uneven and empty shards included, dtypes restored"""
    import torch.multiprocessing as mp
    mp.spawn(_columns_worker, args=(world, _free_port(), n_objects, str(tmp_path)),
             nprocs=world, join=True)
    idx = np.arange(n_objects)
    for r in range(world):
        got = np.load(tmp_path / ("cols_%d.npz" % r))
        np.testing.assert_array_equal(got["pars"], np.stack([idx * 1.5, -idx * 1.0,
                                                             idx ** 2 * 1.0], axis=1))
        assert got["flags"].dtype == np.int64 and got["ok"].dtype == np.bool_
        np.testing.assert_array_equal(got["flags"], idx % 3)
        np.testing.assert_array_equal(got["ok"], idx % 2 == 0)
        assert got["cov"].shape == (n_objects, 2, 2)
        np.testing.assert_array_equal(got["cov"][:, 1, 0], idx * 1.0)
    # without a process group: the columns themselves
    from ngmix_amd.distributed import allgather_columns
    one = allgather_columns({"a": np.arange(3), "b": np.ones((3, 2))})
    assert one["a"].dtype == np.arange(3).dtype and one["b"].shape == (3, 2)
