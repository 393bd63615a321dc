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
    work = dist.all_gather_into_tensor(buf, padded, group=group, async_op=async_op)
    if out is None:
        out = torch.empty((n_objects,) + tail, dtype=local.dtype, device=local.device)

    def trim():
        lo = 0
        for r in range(world):
            out[lo:lo + sizes[r]] = buf[r * per:r * per + sizes[r]]
            lo += sizes[r]

    if async_op:
        return out, _TrimAfterWait(work, trim)
    trim()
    return out


class _TrimAfterWait:
    """work handle of an uneven async all-gather: wait() completes the
    collective and then copies the shards out of the padded buffer"""

    def __init__(self, work, trim):
        self._work, self._trim = work, trim

    def wait(self, *a, **kw):
        r = self._work.wait(*a, **kw)
        if self._trim is not None:
            self._trim()
            self._trim = None
        return r

    def is_completed(self):
        return self._work.is_completed()


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


def visible_gpu_count():
    """number of GPUs this process could use, WITHOUT initialising the HIP
    runtime (torch.cuda.device_count() only counts)"""
    import torch
    return int(torch.cuda.device_count())


def launch_local_ranks(script, argv, nproc, cwd=None, timeout=None, need_gpus=True):
    """
    Start `nproc` ranks of `script` on this node, one child process per GPU,
    with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set as
    torch.distributed.run would, and wait for them.  The caller must not have
    initialised the GPU (children are fork+exec'ed).  Rank 0's stdout is this
    process's stdout.  Returns 0 when every rank exited 0; otherwise the first
    non-zero status, after terminating the ranks that were still running (they
    would wait in a collective forever).

    Refuses (returns 2, message on stderr) when the node has fewer GPUs than
    ranks and the backend is RCCL: RCCL needs one device per rank, and a job
    silently run on fewer GPUs than asked for would be mislabelled.
    NGMIX_DIST_BACKEND=gloo (testing hook) lifts that restriction.
    """
    import os
    import socket
    import subprocess
    import sys
    import time
    backend = os.environ.get("NGMIX_DIST_BACKEND", "nccl")
    if need_gpus:
        ndev = visible_gpu_count()
        if ndev < 1:
            sys.stderr.write("%s: %d ranks asked for but no GPU is visible (the HIP "
                             "kernels are the product; there is no CPU path)\n"
                             % (os.path.basename(script), nproc))
            return 2
        if backend == "nccl" and ndev < nproc:
            sys.stderr.write(
                "%s: %d ranks asked for but only %d GPU(s) are visible; RCCL needs one "
                "device per rank.  Refusing to measure fewer GPUs than asked for.\n"
                % (os.path.basename(script), nproc, ndev))
            return 2
    port = os.environ.get("MASTER_PORT")
    if not port:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
        s.close()
    procs = []
    for r in range(nproc):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nproc),
                   LOCAL_WORLD_SIZE=str(nproc), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env,
                                      cwd=cwd))
    if timeout is None:
        timeout = float(os.environ.get("NGMIX_LAUNCH_TIMEOUT", "3000"))
    deadline = time.time() + timeout
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:
                    q.terminate()
        if pending and time.time() > deadline:
            for q in pending:
                q.kill()
            sys.stderr.write("%s: ranks timed out after %.0f s\n"
                             % (os.path.basename(script), timeout))
            return 3
        time.sleep(0.02)
    if rc:
        sys.stderr.write("%s: a rank exited with status %d\n"
                         % (os.path.basename(script), rc))
    return rc


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
        # an empty shard (n_objects < world, or a short tail): the placeholder
        # must live where the backend communicates -- the current GPU for RCCL
        if device is None and dist.get_backend(group) == "nccl":
            device = torch.device("cuda", torch.cuda.current_device())
        local = torch.empty((0,) + tuple(record_shape),
                            dtype=dtype or torch.float64, device=device)
    return allgather_records(local, n_objects=n_objects, group=group)


def allgather_columns(columns, n_objects=None, group=None):
    """
    ONE all-gather for several per-object arrays of a pipeline (the psf, guess
    and fit records of bootstrap_batch, say) instead of one collective per
    stage: on the point-to-point xGMI mesh a collective costs its launch and
    ring set-up whatever its size, and these records are small.

    columns: dict name -> (n_local, ...) array or tensor of this rank's block
    of objects (shard_bounds); integer and boolean arrays ride as float64
    (exact below 2^53) and come back in their own dtype.  Returns a dict of
    (n_objects, ...) numpy arrays, the same on every rank.  Without a process
    group: the columns themselves, as numpy.
    """
    import torch
    import torch.distributed as dist
    names = sorted(columns)
    arrs, metas = [], []
    for k in names:
        a = columns[k]
        a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
        if a.dtype.kind not in "fiub" or a.dtype.itemsize > 8:
            raise TypeError("allgather_columns: %s has dtype %s" % (k, a.dtype))
        if a.dtype.kind in "iu" and a.size and np.abs(a).max() >= 2 ** 53:
            raise ValueError("allgather_columns: %s does not fit a float64" % k)
        metas.append((k, a.dtype, a.shape[1:]))
        width = int(np.prod(a.shape[1:], dtype=np.int64))
        arrs.append(a.reshape(a.shape[0], width).astype(np.float64))
    if not names:
        return {}
    nloc = arrs[0].shape[0]
    if any(a.shape[0] != nloc for a in arrs):
        raise ValueError("allgather_columns: the columns have different lengths")
    if not (dist.is_available() and dist.is_initialized()):
        return {k: np.asarray(a.reshape((nloc,) + tuple(sh)), dtype=dt)
                for a, (k, dt, sh) in zip(arrs, metas)}
    packed = torch.from_numpy(np.concatenate(arrs, axis=1))
    if dist.get_backend(group) == "nccl":
        packed = packed.to(torch.device("cuda", torch.cuda.current_device()))
    full = allgather_records(packed, n_objects=n_objects, group=group).cpu().numpy()
    out, c = {}, 0
    for a, (k, dt, sh) in zip(arrs, metas):
        w = a.shape[1]
        out[k] = full[:, c:c + w].reshape((full.shape[0],) + tuple(sh)).astype(dt)
        c += w
    return out


def bootstrap_sharded(make_shard, n_objects, keys=None, group=None, **boot_kw):
    """
    pipeline.bootstrap_batch over this rank's block of objects, its per-object
    results gathered to every rank by ONE collective (allgather_columns).

    make_shard(lo, hi) -> (stamps, psf_stamps, kwargs): the StampBatch pair of
    objects lo..hi-1 and keyword arguments for bootstrap_batch (stamp_obj /
    stamp_band relative to lo, ...).  keys: the result arrays to gather
    (default: the per-object records -- flags, nfev, pars, pars_err, psf_T,
    psf_g, psf_flags, guess, guess_flags, s2n, chi2per, lnprob).
    """
    import torch.distributed as dist
    from .pipeline import bootstrap_batch
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(n_objects, rank, world)
    if keys is None:
        keys = ("flags", "nfev", "pars", "pars_err", "psf_T", "psf_g", "psf_flags",
                "guess", "guess_flags", "s2n", "chi2per", "lnprob")
    if hi > lo:
        stamps, psf_stamps, kw = make_shard(lo, hi)
        kw = dict(kw or {})
        kw.update(boot_kw)
        res = bootstrap_batch(stamps, psf_stamps, **kw)
        cols = {k: np.asarray(res[k]) for k in keys if k in res}
        shapes = {k: (v.dtype, v.shape[1:]) for k, v in cols.items()}
    else:
        cols, shapes = None, None
    if world > 1:
        # an empty shard learns the record layout from its neighbours
        layouts = [None] * world
        dist.all_gather_object(layouts, shapes, group=group)
        known = next(s for s in layouts if s is not None)
        if cols is None:
            cols = {k: np.zeros((0,) + tuple(sh), dtype=dt) for k, (dt, sh) in known.items()}
    return allgather_columns(cols, n_objects=n_objects, group=group)
