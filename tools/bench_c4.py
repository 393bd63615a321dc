"""config 4 of BASELINE.json: adaptive moments + em_run over 32x32 stamps,
sharded over the GPUs of one node (one process per GPU), the 584-byte admom
records and the EM mixtures all-gathered over RCCL after each stage.

    python tools/bench_c4.py [--nstamps 125000] [--reps 5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
        --master-addr 127.0.0.1 --master-port 29511 tools/bench_c4.py

Weak scaling: every rank holds --nstamps stamps (125k x 8 = the 1M of config 4).
Rank 0 prints one JSON line per stage."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from ngmix_amd import distributed as nd  # noqa: E402
from ngmix_amd.batch import StampBatch, GMixBatch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nstamps", type=int, default=125000)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--dim", type=int, default=32)
args = ap.parse_args()

rank, world, local_rank = nd.init_from_env(backend="nccl")
torch.cuda.set_device(local_rank)
dev = torch.device("cuda", local_rank)
n, dim, scale = args.nstamps, args.dim, 0.263
rng = np.random.RandomState(5 + rank)
pars = np.zeros((n, 6))
pars[:, 0:2] = rng.uniform(-0.5, 0.5, size=(n, 2)) * scale
pars[:, 2:4] = rng.normal(scale=0.05, size=(n, 2))
pars[:, 4] = rng.uniform(0.3, 0.9, size=n) + 0.27
pars[:, 5] = rng.uniform(50, 200, size=n)
gm_true, _ = GMixBatch.from_pars(pars, "gauss", device=dev)
jac = np.array([(dim - 1) / 2, (dim - 1) / 2, scale, 0, 0, scale, scale ** 2, scale])
d_jac = torch.from_numpy(np.tile(jac, (n, 1))).to(dev)
off = np.arange(n, dtype=np.int64) * dim * dim
geom = StampBatch(None, None, d_jac, np.full(n, dim), np.full(n, dim), off, True)
truth, _ = geom.render(gm_true)
gen = torch.Generator(device=dev)
gen.manual_seed(1 + rank)
sky = 0.05
val = truth + 0.01 * torch.randn(truth.shape, generator=gen, device=dev, dtype=torch.float64)
ierr = torch.full_like(val, 100.0)
sb = StampBatch(val, ierr, d_jac, np.full(n, dim), np.full(n, dim), off, True)
sb_em = StampBatch(val + sky, ierr, d_jac, np.full(n, dim), np.full(n, dim), off, True)

guess = np.zeros((n, 6))
guess[:, 4] = pars[:, 4] * rng.uniform(0.9, 1.1, size=n)
guess[:, 5] = 1.0
wt0, _ = GMixBatch.from_pars(guess, "gauss", device=dev)
emguess = pars.copy()
emguess[:, 4] = (pars[:, 4] - 0.27) * rng.uniform(0.9, 1.1, size=n)
emguess[:, 5] = pars[:, 5] * scale ** 2 * rng.uniform(0.9, 1.1, size=n)
gm0, _ = GMixBatch.from_pars(emguess, "gauss", device=dev)
psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (n, 1)), "gauss",
                             device=dev)


def admom_stage():
    res, status = sb.admom(wt0.clone())           # (n, 73) doubles = 584-byte records
    if world > 1:
        res = nd.allgather_records(res, n_objects=world * n)
    return res


def em_stage():
    g = gm0.clone()
    out, status, _ = sb_em.em(g, psf, sky=sky)
    rec = torch.cat([g.data.reshape(n, -1)[:, :6], out], dim=1)  # mixture + (numiter, fdiff, sky)
    if world > 1:
        rec = nd.allgather_records(rec.contiguous(), n_objects=world * n)
    return rec


for name, stage, nbytes in (("admom", admom_stage, 584), ("em_run", em_stage, 72)):
    stage()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        rec = stage()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({
            "stage": name, "n_gpus": world, "stamps_per_gpu": n, "dim": dim,
            "objects_per_s": world * n / dt, "ms_per_pass": dt * 1e3,
            "gathered_record_bytes": nbytes, "records_on_rank0": int(rec.shape[0]),
            "scaling": "weak", "data": "synthetic"}))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
