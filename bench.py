#!/usr/bin/env python
"""
bench.py -- BASELINE.json's metric on its configs[1] workload:

    pixel-Gaussian evals/sec (render+loglike), 48x48x6-gauss stamps

One "step" = one pass of the hot path over one batch resident in HBM: a
render (accumulating into a model image, render_nb.py:9-36) followed by a
get_loglike (gmix_nb.py:824-874) of every stamp, one kernel launch each.
Weak scaling: every rank holds its own --nstamps stamps; objects are
independent, so there is no data-path collective -- only the all-gather of the
32-byte per-stamp result records named by north_star, overlapped on a side
stream.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C4|C5]

--gpus N > 1 without a torch.distributed.run environment: this process starts
the N ranks itself (one child process per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* set) before anything touches the GPU, relays rank 0's JSON line and
exits with the worst child status.  Under torch.distributed.run (WORLD_SIZE
already set) it is one of the ranks.  Either way the job refuses to run unless
world size == --gpus and (on GPUs) the backend is nccl (= RCCL).
Rank 0 prints ONE JSON line.

Other configs of BASELINE.json, same contract, own rooflines:
  --config C3   complete LM fits of the C2 stamps (lock-step batched driver)
  --config C4   admom + em_run over 32x32 stamps (fp64-VALU bound)
  --config C5   10 epochs x 64x64, 16-gaussian 'bdf' loglike (HBM bound)
A default (C2, N = 1) run appends a short C4 and C5 measurement under
"other_configs" so that the driver's record carries them too.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SCALE = 0.263
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s HBM3E spec
FP64_VALU_PEAK_TF = 78.6  # MI355X_MICROARCH.md: fp64 vector peak (FMA = 2 flop)

# ---- C2: 48x48 stamps, 6 gaussians
NROW = NCOL = 48
NGAUSS = 6
NPIX = NROW * NCOL
PAIRS_PER_STAMP = NPIX * NGAUSS  # 13,824 pixel-gaussian evaluations
# algorithmic bytes per stamp evaluation (SURVEY.md 8d / BASELINE.md section 3)
LOGLIKE_BYTES = 16 * NPIX + 64 + 48 + 32   # 37,008
RENDER_BYTES = 16 * NPIX                   # 36,864 (8 read + 8 written)

# ---- C4: flop model of the iterative kernels,
# counted from the kernel source, FMA = 2 flop, per pixel:
#   admom centroid pass   31  (chi2 8, gate+fexp+apod-free weight 15, 3 sums 5, ...)
#   admom moments pass    45  (7 moment sums + wsum; no covariance in the loop)
#   admom covariance pass 96  (once per object: 28 unique w^2 var F_i F_j sums)
#   em_run pixel pass     48  (1 object gaussian (x) 1 psf gaussian: offsets 2,
#                              squares 3, chi2/2 5, hard-cut fexp + norm 15,
#                              K - y 1, gtot 1, one reciprocal by v_rcp + a
#                              Newton step 5, factor 1, w = gi factor 1, eight
#                              accumulators 14.  The reference's own loop is
#                              ~60; round 1's kernel executed 73.)
# the fused get_loglike kernel the library launches (seven waves per SIMD unless
# NGMIX_LOGLIKE_6WAVES selects the six-wave build for A/B)
LOGLIKE_KERNEL = ("pixpass_wave_kernel<0," if os.environ.get("NGMIX_LOGLIKE_6WAVES")
                  else "pixpass_wave_kernel7<0,")
ADMOM_FLOP_ITER_PX = 31 + 45
ADMOM_FLOP_ONCE_PX = 96
# round 3: the logL term (K - y and its accumulate: 2 flop) is executed only in
# the iterations whose elogL the convergence test can see -- the last two of
# the 40 a config-4 fit runs -- so the executed count is 46 + 2 * 2 / 40
EM_FLOP_ITER_PX = 46.1


# --------------------------------------------------------------------------
# launching
# --------------------------------------------------------------------------

def launch_ranks(args, argv):
    """--gpus N > 1 and no WORLD_SIZE in the environment: start the N ranks as
    child processes (ngmix_amd.distributed.launch_local_ranks).  Nothing in
    this (parent) process initialises the GPU."""
    from ngmix_amd import distributed as nd
    return nd.launch_local_ranks(os.path.abspath(__file__), argv, args.gpus, cwd=ROOT)


def init_rank(args):
    """one process per GPU; "nccl" is RCCL on ROCm (xGMI inside the node).
    Returns (rank, world, device, backend); refuses a world size other than
    --gpus."""
    import torch
    import torch.distributed as dist
    from ngmix_amd import distributed as nd
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP kernels are the product)")
    rank, world, local_rank = nd.init_from_env(backend="nccl")
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the process group has %d rank(s); "
                         "refusing to report a mislabelled number" % (args.gpus, world))
    backend = None
    if world > 1:
        backend = dist.get_backend()
        allowed = ("nccl", os.environ.get("NGMIX_DIST_BACKEND", "nccl"))
        if backend not in allowed or dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: backend %r / world %d is not the RCCL job "
                             "that was asked for" % (backend, dist.get_world_size()))
    torch.cuda.set_device(local_rank)
    return rank, world, torch.device("cuda", local_rank), backend


class PeerFailure(RuntimeError):
    """another rank failed in this leg (every rank records it and moves on)"""


# One leg of a run at N > 1.  A failure that is local to one rank (an
# allocation that fails on one shard) must neither leave the other ranks
# blocked in the leg's collectives nor cost the headline line: the ranks agree
# on an ok flag when they enter the steps and again after the leg, and a rank
# that fails INSIDE its steps keeps the leg's collectives matched with empty
# records until the leg is over (Gather.drain).
LEG = {"entered": False, "settled": False, "error": None}


def agree(failed, device):
    """MAX over the ranks of a failure flag (one small all-reduce)"""
    import torch
    import torch.distributed as dist
    t = torch.tensor([1 if failed else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(int(t.item()))


def inject_failure(where):
    """testing knob NGMIX_BENCH_FAIL = "<config>:<rank>:<build|setup|stepN>"
    (build: before the leg's workload exists; setup: at the entry of its steps) """
    spec = os.environ.get("NGMIX_BENCH_FAIL")
    if spec:
        cfg, rk, at = spec.split(":")
        if (cfg, int(rk), at) == (LEG.get("config"), int(os.environ.get("RANK", "0")), where):
            raise RuntimeError("injected failure (%s)" % spec)


def run_leg(name, fn, world, device, *a, **kw):
    """fn(*a, **kw) as one leg; returns (result, error).  At N > 1 the error is
    the same kind on every rank: the failing rank's own, PeerFailure elsewhere."""
    LEG.update(entered=False, settled=False, error=None, config=name)
    out = err = None
    try:
        inject_failure("build")
        out = fn(*a, **kw)
    except Exception as e:
        err = e
    if world > 1:
        if not LEG["entered"]:
            # failed before its steps: the others are at the entry agreement
            agree(True, device)
        elif not LEG["settled"]:
            if agree(err is not None, device) and err is None:
                out, err = None, PeerFailure("a peer rank failed in leg %s" % name)
    return out, err


def timed_steps(step, args, distributed, device, gat=None):
    """settle + W warm-up steps untimed, then exactly K steps between barrier +
    synchronize on both sides; returns the max over ranks of the elapsed time"""
    import torch
    import torch.distributed as dist
    if distributed:
        LEG["entered"] = True
        try:
            inject_failure("setup")
            mine = None
        except Exception as e:
            mine = e
        if agree(mine is not None, device):
            # (every rank knows: no second agreement after this leg)
            LEG["settled"] = True
            raise mine or PeerFailure("a peer rank failed while setting leg %s up"
                                      % LEG.get("config"))
    count = [0]

    def run(i):
        done = 0
        if LEG["error"] is None:
            before = gat.calls if gat is not None else 0
            try:
                inject_failure("step%d" % count[0])
                count[0] += 1
                step(i)
                return
            except Exception as e:
                if not distributed:
                    raise
                LEG["error"] = e
                done = (gat.calls - before) if gat is not None else 0
        if gat is not None:
            gat.drain(done)

    for _ in range(max(args.settle_steps, 0)):
        run(None)
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        run(None)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        run(i)
    torch.cuda.synchronize()
    own = time.perf_counter() - t0          # this rank's K steps, before the barrier
    if distributed:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    PER_RANK_MS[:] = [own / args.steps * 1e3]
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank's own time per step (the skew an 8-GPU run would show)
        mine = torch.tensor([own / args.steps * 1e3], dtype=torch.float64, device=device)
        every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(every, mine)
        PER_RANK_MS[:] = [float(v.item()) for v in every]
    if LEG["error"] is not None:
        raise LEG["error"]
    return elapsed


# ms per step of every rank's own K steps in the last timed_steps() call
PER_RANK_MS = []


class Gather:
    """north_star's all-gather of per-object result records, on a side stream
    so that it overlaps the next step's first kernel.  pattern: the gathers of
    one step in order, [(key, shape, dtype)] -- what drain() replays with empty
    records for a rank whose step failed."""

    def __init__(self, world, device, pattern=()):
        import torch
        self.world = world
        self.device = device
        self.on = world > 1
        self.side = torch.cuda.Stream(device=device) if self.on else None
        self.pending = None
        self.bufs = {}
        self.pattern = list(pattern)
        self.calls = 0

    def drain(self, done):
        """the gathers a failed step still owes its peers (all but the first
        `done` of the pattern), with zero records"""
        import torch
        if not self.on:
            return
        for key, shape, dtype in self.pattern[done:]:
            self.wait_consumed()
            self.gather(key, torch.zeros(shape, dtype=dtype, device=self.device))

    def wait_consumed(self):
        """the previous gather must have read its source before the kernel
        that rewrites it is launched"""
        import torch
        if self.pending is not None:
            torch.cuda.current_stream().wait_event(self.pending)
            self.pending = None

    def gather(self, key, records):
        import torch
        from ngmix_amd import distributed as nd
        self.calls += 1
        if not self.on:
            return records
        n = records.shape[0]
        if key not in self.bufs:
            self.bufs[key] = torch.empty((self.world * n,) + tuple(records.shape[1:]),
                                         dtype=records.dtype, device=records.device)
        done = torch.cuda.Event()
        done.record()
        with torch.cuda.stream(self.side):
            self.side.wait_event(done)
            nd.allgather_records(records, n_objects=self.world * n, out=self.bufs[key])
            self.pending = torch.cuda.Event()
            self.pending.record()
        return self.bufs[key]


def _events(k):
    import torch
    return [torch.cuda.Event(enable_timing=True) for _ in range(k)]


def _mean_ms(a, b):
    return float(np.mean([x.elapsed_time(y) for x, y in zip(a, b)]))


def kernel_symbols(fragments):
    """the device symbols of libngmix_hip.so that contain every fragment of one
    of the given tuples (so that the JSON line names the kernel as rocprofv3
    will, not a made-up label)"""
    from ngmix_amd import _lib
    try:
        out = subprocess.run(["nm", "-D", "--defined-only", "-C", _lib.LIB_PATH],
                             capture_output=True, text=True, timeout=30).stdout
    except Exception:
        return {}
    found = {}
    for key, frags in fragments.items():
        for ln in out.splitlines():
            name = ln.split(" ", 2)[-1]
            if all(f in name for f in frags) and "__device_stub__" in name:
                found[key] = name.split("__device_stub__")[-1].split("(")[0].strip()
                break
    return found


# --------------------------------------------------------------------------
# C2 (the metric's config)
# --------------------------------------------------------------------------

def make_workload(nstamps, seed, device):
    """per-stamp 'exp' model x gaussian psf T=0.27, noise
    sigma = 0.01*flux/100, uniform weight; images rendered on the device by the
    render kernel itself (truth + N(0, sigma^2))"""
    import torch
    from ngmix_amd.batch import StampBatch, GMixBatch
    rng = np.random.RandomState(seed)
    pars = np.zeros((nstamps, 6))
    pars[:, 0:2] = rng.uniform(-0.5, 0.5, size=(nstamps, 2)) * SCALE
    g = rng.normal(scale=0.1, size=(nstamps, 2))
    gmag = np.sqrt((g ** 2).sum(axis=1))
    g *= np.where(gmag > 0.7, 0.7 / np.maximum(gmag, 1e-30), 1.0)[:, None]
    pars[:, 2:4] = g
    pars[:, 4] = rng.uniform(0.3, 1.5, size=nstamps)
    pars[:, 5] = rng.uniform(50.0, 500.0, size=nstamps)
    psfpars = np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (nstamps, 1))

    gm0, st0 = GMixBatch.from_pars(pars, "exp", device=device)
    psf, _ = GMixBatch.from_pars(psfpars, "gauss", device=device)
    gm, _ = gm0.convolve(psf)
    assert int(gm.set_norms().abs().sum()) == 0 and int(st0.abs().sum()) == 0

    jac = np.array([23.5, 23.5, SCALE, 0.0, 0.0, SCALE, SCALE ** 2, SCALE])
    geom = StampBatch(None, None,
                      torch.from_numpy(np.tile(jac, (nstamps, 1))).to(device),
                      np.full(nstamps, NROW), np.full(nstamps, NCOL),
                      np.arange(nstamps, dtype=np.int64) * NPIX, True)
    truth, _ = geom.render(gm, fast_exp=True)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    sigma = torch.from_numpy(0.01 * pars[:, 5] / 100.0).to(device)
    val = truth.reshape(nstamps, NPIX)
    val += torch.randn(val.shape, generator=gen, device=device,
                       dtype=torch.float64) * sigma[:, None]
    ierr = (1.0 / sigma)[:, None].expand(nstamps, NPIX).contiguous()
    sb = StampBatch(val.reshape(-1), ierr.reshape(-1), geom.jac,
                    np.full(nstamps, NROW), np.full(nstamps, NCOL),
                    np.arange(nstamps, dtype=np.int64) * NPIX, True)
    # evaluate at a perturbed parameter set, as an LM iteration would
    pert = pars.copy()
    pert[:, 4] *= 1.02
    pert[:, 5] *= 0.99
    gmp0, _ = GMixBatch.from_pars(pert, "exp", device=device)
    gmp, _ = gmp0.convolve(psf)
    gmp.set_norms()
    return sb, gmp, pars


def run_c2(args, rank, world, device, backend):
    import torch
    n = args.nstamps or 100000
    distributed = world > 1
    sb, gm, _ = make_workload(n, seed=1000 + rank, device=device)
    image = torch.zeros(sb.total_pix, dtype=torch.float64, device=device)
    out = torch.empty((n, 4), dtype=torch.float64, device=device)
    status = torch.empty(n, dtype=torch.int32, device=device)
    gat = Gather(world, device, [("loglike", (n, 4), torch.float64)])
    ev_r0, ev_r1 = _events(args.steps), _events(args.steps)
    ev_l0, ev_l1 = _events(args.steps), _events(args.steps)

    def step(i):
        if i is not None:
            ev_r0[i].record()
        sb.render(gm, image=image, fast_exp=True, status=status, exact=args.exact)
        if i is not None:
            ev_r1[i].record()
        gat.wait_consumed()      # before the events: only the kernel is timed
        if i is not None:
            ev_l0[i].record()
        sb.loglike(gm, out=out, status=status, exact=args.exact)
        if i is not None:
            ev_l1[i].record()
        gat.gather("loglike", out)

    elapsed = timed_steps(step, args, distributed, device, gat)
    bad = int((status != 0).sum().item())
    render_ms = _mean_ms(ev_r0, ev_r1)
    loglike_ms = _mean_ms(ev_l0, ev_l1)
    # outside the timed region: the write-only form of the render (what
    # GMix.make_image / StampBatch.render(image=None) run: the model is stored
    # without the image being read -- NGMIX_BATCH_RENDER_OVERWRITE), 20 launches
    fe0, fe1 = _events(20), _events(20)
    for k in range(25):
        if k >= 5:
            fe0[k - 5].record()
        fresh, _ = sb.render(gm, image=None, fast_exp=True, status=status, exact=args.exact)
        if k >= 5:
            fe1[k - 5].record()
    torch.cuda.synchronize()
    fresh_ms = _mean_ms(fe0, fe1)
    del fresh
    if rank != 0:
        return None

    pairs_per_step = 2.0 * world * n * PAIRS_PER_STAMP
    value = pairs_per_step * args.steps / elapsed
    dominant = "loglike" if loglike_ms >= render_ms else "render"
    dom_ms = max(loglike_ms, render_ms)
    dom_bytes = (LOGLIKE_BYTES if dominant == "loglike" else RENDER_BYTES) * n
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
    if args.exact:
        syms = kernel_symbols({"loglike": ("pixpass_grid_kernel<0,",),
                               "render": ("pixpass_grid_kernel<2,",)})
    else:
        syms = kernel_symbols({"loglike": (LOGLIKE_KERNEL, "false"),
                               "render": ("pixpass_wave_kernel<2,", "false")})
    traffic_bytes, traffic_source = load_traffic(dominant, n)
    line = {
        "metric": baseline_metric(),
        "value": value,
        "unit": "pixel-gaussian evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "C2: %d stamps/GPU, 48x48 px, 6-gaussian 'exp' (x) "
                        "gaussian psf; one step = render + get_loglike of "
                        "every stamp" % n,
            "stamps_per_gpu": n,
            "parallelism": "stamps sharded across %d rank(s); all-gather of "
                           "32-B result records" % world,
        },
        "roofline": {
            "bound": "hbm",
            "dominant": dominant,
            "kernel": syms.get(dominant, "pixpass_%s_kernel (%s)" % (
                "grid" if args.exact else "wave", dominant)),
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic_bytes,
            "traffic_source": traffic_source,
            "algorithmic_bytes_per_launch": dom_bytes,
            "avg_launch_ms": dom_ms,
            **load_issue("pixpass_wave_kernel7<0, false, 8>" if dominant == "loglike"
                         else "pixpass_wave_kernel<2, false, 16>"),
        },
        "kernels_ms": {"render": render_ms, "loglike": loglike_ms},
        # the fresh (write-only) render, not part of `value`: 18,432 B written per stamp
        "fresh_render": {"ms": fresh_ms, "stamp_renders_per_s_per_gpu": n / (fresh_ms * 1e-3),
                         "hbm_frac": (RENDER_BYTES / 2) * n / (fresh_ms * 1e-3) / 1e9 /
                         HBM_PEAK_GBS},
        "loglike_stamp_evals_per_s_per_gpu": n / (loglike_ms * 1e-3),
        "render_stamp_evals_per_s_per_gpu": n / (render_ms * 1e-3),
        "loglike_hbm_frac": LOGLIKE_BYTES * n / (loglike_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "render_hbm_frac": RENDER_BYTES * n / (render_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "bad_status": bad,
        "settle_steps": max(args.settle_steps, 0),
        "kernel_mode": "exact" if args.exact else "fused",
    }
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(sb, gm)
    return line


# --------------------------------------------------------------------------
# C3: batched Levenberg-Marquardt fits
# --------------------------------------------------------------------------

def run_c3(args, rank, world, device, backend, nstamps=None, steps=None):
    """one step = a complete maximum-likelihood fit ('exp' (x) gaussian psf,
    analytic jacobian, lmder semantics) of every stamp of the C2 batch from
    guess = truth x U(0.9, 1.1): LMBatchFitter.go end to end (lock-step rounds on
    the device, packaging, statistics, download); the per-object result
    records (flags, nfev, pars, errors: 112 B) are all-gathered"""
    import torch
    from ngmix_amd.batch import GMixBatch
    from ngmix_amd.lm_batch import LMBatchFitter
    n = nstamps or args.nstamps or 100000
    distributed = world > 1
    sb, _, pars = make_workload(n, seed=1000 + rank, device=device)
    rng = np.random.RandomState(7 + rank)
    guess = pars * rng.uniform(0.9, 1.1, size=pars.shape)
    guess[:, 0:2] = pars[:, 0:2] + rng.uniform(-0.05, 0.05, size=(n, 2))
    guess[:, 2:4] = pars[:, 2:4] + rng.uniform(-0.03, 0.03, size=(n, 2))
    psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (n, 1)), "gauss",
                                 device=device)
    import copy
    a2 = copy.copy(args)
    a2.steps = steps or args.steps
    if steps is not None:
        # (a short leg of a default C2 run: the GPU has idled through the CPU
        # legs before it, and the clock governor needs ~0.4 s of uninterrupted
        # launches to reach the state the timed steps should see)
        a2.warmup, a2.settle_steps = 2, 70
    else:
        a2.warmup = min(args.warmup, 5)
        a2.settle_steps = min(max(args.settle_steps, 0), 200)
    fitter = LMBatchFitter("exp")
    fitter.time_kernels = True     # HIP events around every lm_eval launch
    gat = Gather(world, device, [("lm", (n, 14), torch.float64)])
    K = steps or args.steps
    state = {"loop": 0.0, "rounds": 0, "bad": 0, "nfev": 0.0, "eval_ms": 0.0,
             "eval_stamps": 0.0, "launches": 0, "kernels": {}, "launched": 0}

    # The steps run as a software pipeline (LMBatchFitter.go_stream): every
    # batch's device work -- set-up, rounds, finalize, pack, downloads -- is
    # queued without the host waiting for anything, and the next batch is
    # queued before the host turns to this one's arrays (--no-pipeline: one
    # synchronous go() per step).  ONE pipeline runs through the untimed and
    # the timed steps, so the timed region sees its steady state at both ends:
    # the batch after the last timed one is queued inside the region -- work
    # for a batch that is not counted, and the closing synchronisation waits
    # for it.
    pipelined = not getattr(args, "no_pipeline", False)
    streams = {}

    def batches(count):
        for _ in range(count):
            yield sb, guess, {"psf": psf}

    def next_result(i):
        if not pipelined:
            return fitter.go(sb, guess, psf=psf)
        if "all" not in streams:
            count = a2.warmup + max(a2.settle_steps, 0) + a2.steps + 1
            streams["all"] = fitter.go_stream(batches(count))
        return next(streams["all"])

    def step(i):
        res = next_result(i)
        if i is not None:
            state["loop"] += fitter.loop_seconds
            state["eval_ms"] += fitter.eval_ms_total
            state["eval_stamps"] += fitter.eval_stamps_total
            state["launches"] += fitter.eval_launch_count
            state["launched"] += fitter.rounds_launched
            for k, v in getattr(fitter, "kernel_ms", {}).items():
                state["kernels"][k] = state["kernels"].get(k, 0.0) + v
            for k, v in (fitter.host_ms or {}).items():
                state.setdefault("host", {})[k] = state.setdefault("host", {}).get(k, 0.0) + v
            state["nsplit"] = fitter.nsplit_used
            state["rounds"] = fitter.rounds
            state["bad"] = int((res["flags"] != 0).sum())
            state["nfev"] = float(np.mean(res["nfev"]))
        if gat.on:
            rec = np.concatenate([res["flags"][:, None].astype("f8"),
                                  res["nfev"][:, None].astype("f8"), res["pars"],
                                  res["pars_err"]], axis=1)
            gat.wait_consumed()
            # (through pinned memory, asynchronously: a copy from pageable memory
            # synchronises the stream, i.e. would make this rank's host wait for
            # the batch it has just queued)
            # Two pinned buffers alternate; a buffer is rewritten only after the
            # event behind its last upload has completed on the HOST (the upload
            # is queued behind a whole batch of kernels: ordering the stream
            # alone would let this write overtake it).
            if "pins" not in state:
                state["pins"] = [[torch.empty(rec.shape, dtype=torch.float64,
                                              pin_memory=True), None] for _ in range(2)]
                state["pin_turn"] = 0
            slot = state["pins"][state["pin_turn"]]
            state["pin_turn"] ^= 1
            if slot[1] is not None:
                slot[1].synchronize()
            slot[0].numpy()[:] = rec
            up = slot[0].to(device, non_blocking=True)
            slot[1] = torch.cuda.Event()
            slot[1].record()
            gat.gather("lm", up)

    # what one synchronous call spends in its lock-step loop (kernels + the
    # counter read-backs), measured before the pipeline starts
    fitter.go(sb, guess, psf=psf)
    fitter.go(sb, guess, psf=psf)
    sync_loop_ms = fitter.loop_seconds * 1e3
    elapsed = timed_steps(step, a2, distributed, device, gat)
    if rank != 0:
        return None
    # (inside the pipeline a batch's loop time is host time and spans the
    # other batch's host work: the figure of a synchronous call is reported)
    loop_ms = sync_loop_ms if pipelined else state["loop"] / K * 1e3
    rounds = max(state["rounds"], 1)
    # the dominant kernel, lm_eval_kernel: one pixel pass per round producing
    # value + 5 derivative images per pixel in registers and the 28 sums
    eval_bytes = LOGLIKE_BYTES + 28 * 8
    # over the whole fit: the stamps every launch still had to evaluate (fits
    # that have converged leave the lock-step rounds; the fitter counts the
    # running ones before each launch) / the HIP-event time of all the launches
    nl = max(state["launches"], 1)
    eval_ms = state["eval_ms"] / nl
    stamps_per_launch = state["eval_stamps"] / nl
    achieved = eval_bytes * state["eval_stamps"] / (state["eval_ms"] * 1e-3) / 1e9
    # HBM bytes of the lm_eval launches of ONE fit from the committed PMC pass
    # (tools/run_prof_all.sh: FETCH_SIZE / WRITE_SIZE passes over tools/bench_lm.py,
    # the same 100k stamps and guesses), spread over this run's launches per fit
    # -- per launch, like `achieved`
    fit_traffic, _ = load_traffic("c3_lm_eval", n, key="c3_nstamps", field="_hbm_bytes_per_fit")
    return {
        "metric": "LM fits/sec ('exp' (x) gaussian psf, 48x48 stamps), 1/2/4/8 GPU",
        "value": world * n * K / elapsed,
        "unit": "fits/s",
        "n_gpus": world, "steps": K, "warmup": a2.warmup,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": "C3: %d stamps/GPU, 48x48 px, 'exp' (x) gaussian psf, guess = truth x "
                        "U(0.9,1.1); one step = a complete lock-step LM fit of every stamp" % n,
            "stamps_per_gpu": n,
            "parallelism": "stamps sharded across %d rank(s); all-gather of 112-B fit "
                           "records" % world,
        },
        "roofline": {
            "bound": "hbm", "kernel": "ngmix::lm_eval_kernel<true, true>",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": fit_traffic / (nl / K) if fit_traffic else None,
            "algorithmic_bytes_per_launch": eval_bytes * stamps_per_launch,
            "avg_launch_ms": eval_ms, "stamps_per_launch": stamps_per_launch,
            "launches_per_fit": nl / K, "pieces": state.get("nsplit", 1),
            **load_issue("lm_eval_kernel<true, true>"),
            "note": "the kernel evaluates value + 5 derivative images per pixel-gaussian "
                    "pair and is VALU-issue bound (profiles/*_pmc_summary.txt), not HBM "
                    "bound; bytes and time are summed over the launches of whole fits, "
                    "each launch counted with the stamps still being fitted",
        },
        # HIP-event time of every kernel of a fit, summed per fit (mean over the
        # timed steps): what ms_per_step is to be read against
        "kernels_ms": {k: v / K for k, v in sorted(state["kernels"].items())},
        "kernels_ms_sum": sum(state["kernels"].values()) / K,
        # what the host did per step: queueing, blocked on the downloads (> 0: the
        # GPU is the bottleneck), packaging
        "host_ms_per_step": {k: v / K for k, v in sorted(state.get("host", {}).items())},
        "rounds_launched": state["launched"] / float(K),
        "device_loop_ms": loop_ms, "rounds": rounds, "pipelined": pipelined,
        "fits_per_s_device_loop": n / (loop_ms * 1e-3) if loop_ms > 0 else None,
        "mean_nfev": state["nfev"], "bad_status": state["bad"],
        "settle_steps": a2.settle_steps,
    }


def run_prepsf(device, n=100000, steps=4):
    """the Fourier-space pre-psf moments (ngmix/prepsfmom.py PGaussMom /
    KSigmaMom) over a catalogue resident in HBM: n 33x33 stamps with their psf
    stamps, padded to 132x132, as one batch (ngmix_amd.prepsfmom.measure_arrays).
    Synchronous steps on rank 0 at N = 1."""
    import time
    import torch
    from ngmix_amd.prepsfmom import PGaussMom, KSigmaMom
    dim, scale = 33, 0.2
    gen = torch.Generator(device=device)
    gen.manual_seed(5)
    ax = torch.arange(dim, dtype=torch.float64, device=device) - 16.0
    r2 = (ax[:, None] ** 2 + ax[None, :] ** 2) * scale ** 2
    psf = torch.exp(-0.5 * r2 / 0.15) * (scale ** 2 / (2 * np.pi * 0.15))
    obj = 50.0 * torch.exp(-0.5 * r2 / 0.40) * (scale ** 2 / (2 * np.pi * 0.40))
    images = obj[None] + 0.02 * torch.randn((n, dim, dim), dtype=torch.float64, device=device,
                                            generator=gen)
    pimages = psf[None] + 1.0e-4 * torch.randn((n, dim, dim), dtype=torch.float64, device=device,
                                               generator=gen)
    weights = torch.full((n, dim, dim), 1.0 / 0.02 ** 2, dtype=torch.float64, device=device)
    rng = np.random.RandomState(3)
    cen = np.tile([16.0, 16.0], (n, 1)) + rng.uniform(-0.3, 0.3, size=(n, 2))
    deriv = (scale, 0.0, 0.0, scale)
    out = {"metric": "pre-psf Fourier moments, stamps/sec (33x33 stamps + psf stamps padded to "
                     "132x132), 1 GPU", "unit": "stamps/s", "steps": steps,
           "config": {"workload": "%d stamps of 33x33 px with 33x33 psf stamps resident in HBM; "
                                  "measure_arrays: transform at the kernel's modes, deconvolution, "
                                  "centre phases, moment and covariance sums" % n}}
    for name, fitter in (("PGaussMom", PGaussMom(1.2)), ("KSigmaMom", KSigmaMom(2.0))):
        fitter.measure_arrays(images, weights, cen, deriv, pimages, cen)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            mom, cov, kernels, padded = fitter.measure_arrays(images, weights, cen, deriv, pimages,
                                                              cen)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out[name] = {"value": n / dt, "ms_per_step": dt * 1e3,
                     "modes_kept": int(kernels["plan"]["irow"].size), "padded_dim": int(padded),
                     "mean_flux": float(np.mean(mom[:, 5])), "finite": bool(np.isfinite(mom[:, 2:]).all())}
    out["value"] = out["PGaussMom"]["value"]
    return out


def run_c3_prior(device, n=100000, steps=6):
    """C3 with the reference's separable joint prior, as a caller of the
    reference builds it (ngmix_amd.joint_prior.PriorSimpleSep of CenPrior,
    GPriorBA, TwoSidedErf terms): the prior's rows evaluated by the prior kernel
    inside the lock-step device loop.  Synchronous steps on rank 0 at N = 1,
    inputs resident in HBM; next to it the same stamps without a prior."""
    import time
    import torch
    from ngmix_amd import priors, joint_prior
    from ngmix_amd.batch import GMixBatch
    from ngmix_amd.lm_batch import LMBatchFitter
    sb, _, pars = make_workload(n, seed=1000, device=device)
    rng = np.random.RandomState(7)
    guess = pars * rng.uniform(0.9, 1.1, size=pars.shape)
    guess[:, 0:2] = pars[:, 0:2] + rng.uniform(-0.05, 0.05, size=(n, 2))
    guess[:, 2:4] = pars[:, 2:4] + rng.uniform(-0.03, 0.03, size=(n, 2))
    psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (n, 1)), "gauss",
                                 device=device)
    prng = np.random.RandomState(11)
    prior = joint_prior.PriorSimpleSep(
        priors.CenPrior(0.0, 0.0, SCALE, SCALE, rng=prng), priors.GPriorBA(0.3, rng=prng),
        priors.TwoSidedErf(-0.1, 0.03, 1.0e3, 1.0, rng=prng),
        priors.TwoSidedErf(-10.0, 1.0, 1.0e6, 1.0e3, rng=prng))

    def clock(fitter):
        fitter.go(sb, guess, psf=psf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = fitter.go(sb, guess, psf=psf)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps, res
    with_prior = LMBatchFitter("exp", prior=prior)
    dt, res = clock(with_prior)
    dt0, res0 = clock(LMBatchFitter("exp"))
    return {
        "metric": "LM fits/sec with a separable joint prior ('exp' (x) gaussian psf, 48x48 "
                  "stamps), 1 GPU",
        "value": n / dt, "unit": "fits/s", "steps": steps, "ms_per_step": dt * 1e3,
        "prior_path": with_prior.prior_path, "mean_nfev": float(res["nfev"].mean()),
        "bad_status": int((res["flags"] != 0).sum()),
        "without_prior": {"value": n / dt0, "ms_per_step": dt0 * 1e3,
                          "mean_nfev": float(res0["nfev"].mean())},
        "config": {"workload": "C3 stamps (%d x %dx%d px), PriorSimpleSep(CenPrior, GPriorBA, "
                               "TwoSidedErf T, TwoSidedErf flux) built from ngmix_amd.priors; "
                               "synchronous LMBatchFitter.go steps" % (n, NROW, NCOL)},
    }


def run_c3_host(device, n=100000, steps=4):
    """C3 from HOST-RESIDENT arrays: one step = the stamps of a catalogue held
    in pinned host memory (images, weight maps, jacobian records, psf records,
    guesses) -> StampBatch.from_stacked (the PCIe transfer of 16 B per pixel, the
    weight -> ierr pass, the count of listed pixels) -> a complete lock-step
    LM fit of every stamp -> the result arrays back on the host.  Synchronous
    steps on rank 0 at N = 1; never the headline value (inputs resident in
    HBM is the contract): what a caller whose data is NOT on the device gets"""
    import time
    import torch
    from ngmix_amd import _lib
    from ngmix_amd.batch import StampBatch
    from ngmix_amd.lm_batch import LMBatchFitter
    sb, _, pars = make_workload(n, seed=1000, device=device)
    rng = np.random.RandomState(7)
    guess = pars * rng.uniform(0.9, 1.1, size=pars.shape)
    guess[:, 0:2] = pars[:, 0:2] + rng.uniform(-0.05, 0.05, size=(n, 2))
    guess[:, 2:4] = pars[:, 2:4] + rng.uniform(-0.03, 0.03, size=(n, 2))
    nrow, ncol = int(sb.nrow[0]), int(sb.ncol[0])

    def pinned(t):
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        h.copy_(t)
        return h
    h_img = pinned(sb.val.reshape(n, nrow, ncol))
    h_wt = pinned((sb.ierr * sb.ierr).reshape(n, nrow, ncol))
    h_jac = pinned(sb.jac)
    del sb
    torch.cuda.empty_cache()
    psf = np.zeros((n, 1), dtype=_lib.GAUSS2D_DTYPE)
    from ngmix_amd.gmix import GMixModel
    psf[:, 0] = GMixModel([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], "gauss")._data[0]
    fitter = LMBatchFitter("exp")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    up_ms = fit_ms = 0.0
    bad = 0

    def step(timed):
        nonlocal up_ms, fit_ms, bad
        ev[0].record()
        stamps = StampBatch.from_stacked(h_img, h_wt, h_jac, device=device)
        ev[1].record()
        res = fitter.go(stamps, guess, psf=psf)
        ev[2].record()
        torch.cuda.synchronize()
        if timed:
            up_ms += ev[0].elapsed_time(ev[1])
            fit_ms += ev[1].elapsed_time(ev[2])
            bad = int((res["flags"] != 0).sum())
        return res
    for _ in range(2):
        step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    elapsed = time.perf_counter() - t0
    nbytes = 2.0 * n * nrow * ncol * 8 + n * 64
    # the same catalogue held as FLOAT32 stamps (what survey postage stamps are
    # stored as): 8 B per pixel over the link, widened exactly on the device
    f32 = None
    bad64 = bad
    try:
        res64 = step(False)
        h_img, h_wt = pinned(h_img.to(torch.float32)), pinned(h_wt.to(torch.float32))
        up0, fit0 = up_ms, fit_ms
        for _ in range(2):
            step(False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            res32 = step(True)
        el32 = time.perf_counter() - t1
        f32 = {"value": n * steps / el32, "unit": "fits/s", "ms_per_step": el32 / steps * 1e3,
               "upload_ms": (up_ms - up0) / steps, "fit_ms": (fit_ms - fit0) / steps,
               "bytes_per_step": nbytes / 2 + n * 32, "bad_status": bad,
               "converged_both": int(((res64["flags"] == 0) & (res32["flags"] == 0)).sum()),
               "note": "float32 images and weights in pinned memory; the fits are float64 on "
                       "the widened values (the reference's np.array(image, dtype='f8'))"}
        up_ms, fit_ms = up0, fit0
    except Exception as e:      # noqa: BLE001 -- a reported extra, never the leg itself
        f32 = {"error": repr(e)}
    return {
        "float32_stamps": f32,
        "metric": "LM fits/sec from host-resident (pinned) arrays ('exp' (x) gaussian psf, "
                  "48x48 stamps), 1 GPU",
        "value": n * steps / elapsed, "unit": "fits/s", "steps": steps,
        "ms_per_step": elapsed / steps * 1e3,
        "upload_ms": up_ms / steps, "fit_ms": fit_ms / steps,
        "pcie_GBps": nbytes / (up_ms / steps * 1e-3) / 1e9,
        "bytes_per_step": nbytes, "bad_status": bad64,
        "config": {"workload": "C3 stamps (%d x %dx%d px) in pinned host memory -> "
                               "StampBatch.from_stacked -> LMBatchFitter.go -> host result "
                               "arrays; synchronous steps" % (n, nrow, ncol)},
        "note": "upload_ms: host -> device copies of val and weight (16 B per pixel) plus "
                "the weight -> ierr pass and the count of listed pixels, HIP events; the "
                "rate is bound by the host link, not by the kernels (fit_ms)",
    }


# --------------------------------------------------------------------------
# C4: admom + em_run over 32x32 stamps
# --------------------------------------------------------------------------

def make_c4(n, seed, device, dim=32):
    """round-ish gaussian (x) gaussian-psf objects on 32x32
    stamps; admom guess T = T_true U(0.9,1.1); 1-gaussian EM guess"""
    import torch
    from ngmix_amd.batch import StampBatch, GMixBatch
    rng = np.random.RandomState(seed)
    pars = np.zeros((n, 6))
    pars[:, 0:2] = rng.uniform(-0.5, 0.5, size=(n, 2)) * SCALE
    pars[:, 2:4] = rng.normal(scale=0.05, size=(n, 2))
    pars[:, 4] = rng.uniform(0.3, 0.9, size=n) + 0.27
    pars[:, 5] = rng.uniform(50, 200, size=n)
    gm_true, _ = GMixBatch.from_pars(pars, "gauss", device=device)
    c = (dim - 1) / 2
    jac = np.array([c, c, SCALE, 0, 0, SCALE, SCALE ** 2, SCALE])
    d_jac = torch.from_numpy(np.tile(jac, (n, 1))).to(device)
    shape = np.full(n, dim)
    off = np.arange(n, dtype=np.int64) * dim * dim
    geom = StampBatch(None, None, d_jac, shape, shape, off, True)
    truth, _ = geom.render(gm_true)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    sky = 0.05
    val = truth + 0.01 * torch.randn(truth.shape, generator=gen, device=device,
                                     dtype=torch.float64)
    ierr = torch.full_like(val, 100.0)
    sb = StampBatch(val, ierr, d_jac, shape, shape, off, True)
    sb_em = StampBatch(val + sky, ierr, d_jac, shape, shape, off, True)
    guess = np.zeros((n, 6))
    guess[:, 4] = pars[:, 4] * rng.uniform(0.9, 1.1, size=n)
    guess[:, 5] = 1.0
    wt0, _ = GMixBatch.from_pars(guess, "gauss", device=device)
    emguess = pars.copy()
    emguess[:, 4] = (pars[:, 4] - 0.27) * rng.uniform(0.9, 1.1, size=n)
    emguess[:, 5] = pars[:, 5] * SCALE ** 2 * rng.uniform(0.9, 1.1, size=n)
    gm0, _ = GMixBatch.from_pars(emguess, "gauss", device=device)
    psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (n, 1)),
                                 "gauss", device=device)
    return dict(sb=sb, sb_em=sb_em, wt0=wt0, gm0=gm0, psf=psf, sky=sky, dim=dim, n=n)


def run_c4(args, rank, world, device, backend, nstamps=None, steps=None, quiet=False):
    """one step = adaptive moments of every stamp, then em_run of every stamp
    (one launch each); the 584-byte admom records and the 72-byte EM records
    are all-gathered on the side stream"""
    import torch
    n = nstamps or args.nstamps or 125000
    w = make_c4(n, 5 + rank, device)
    sb, sb_em, dim = w["sb"], w["sb_em"], w["dim"]
    npx = dim * dim
    distributed = world > 1
    gat = Gather(world, device, [("admom", (n, 73), torch.float64),
                                 ("em", (n, 9), torch.float64)])
    wt = w["wt0"].clone()
    gm = w["gm0"].clone()
    conv, _ = gm.convolve(w["psf"])
    res = torch.zeros((n, 73), dtype=torch.float64, device=device)
    st_a = torch.empty(n, dtype=torch.int32, device=device)
    out_e = torch.empty((n, 3), dtype=torch.float64, device=device)
    st_e = torch.empty(n, dtype=torch.int32, device=device)
    rec_e = torch.empty((n, 9), dtype=torch.float64, device=device)
    K = steps or args.steps
    ev = [_events(K) for _ in range(4)]

    def step(i):
        # the guesses are copied back in (device-to-device, untimed by the
        # kernel events but inside the step: it is part of the job)
        wt.data.copy_(w["wt0"].data)
        gm.data.copy_(w["gm0"].data)
        gat.wait_consumed()
        res.zero_()
        if i is not None:
            ev[0][i].record()
        sb.admom(wt, res=res, status=st_a)
        if i is not None:
            ev[1][i].record()
        gat.gather("admom", res)
        conv_i, _ = gm.convolve(w["psf"])
        if i is not None:
            ev[2][i].record()
        sb_em.em(gm, w["psf"], conv=conv_i, sky=w["sky"], out=out_e, status=st_e)
        if i is not None:
            ev[3][i].record()
        rec_e[:, :6] = gm.data.reshape(n, -1)[:, :6]
        rec_e[:, 6:] = out_e
        gat.gather("em", rec_e)

    import copy
    a2 = copy.copy(args)
    a2.steps = K
    if steps is not None:       # the short leg of a C2 run
        a2.warmup, a2.settle_steps = 2, 30   # ~0.45 s of load before the clock starts
    elapsed = timed_steps(step, a2, distributed, device, gat)
    admom_ms = _mean_ms(ev[0], ev[1])
    em_ms = _mean_ms(ev[2], ev[3])
    # iteration counts of the last pass: the flop model's multiplier
    numiter = res.view(torch.int32).reshape(n, -1)[:, 1].double()
    flags_bad = int((res.view(torch.int32).reshape(n, -1)[:, 0] != 0).sum().item())
    it_admom = float(numiter.mean().item())
    it_em = float(out_e[:, 0].mean().item())
    bad = int((st_a != 0).sum().item()) + int((st_e != 0).sum().item())
    if rank != 0:
        return None
    fl_admom = n * npx * (it_admom * ADMOM_FLOP_ITER_PX + ADMOM_FLOP_ONCE_PX)
    fl_em = n * npx * it_em * EM_FLOP_ITER_PX
    syms = kernel_symbols({"admom": ("admom_grid_kernel<64, 16>",),
                           "em": ("em_wave_kernel<64, 16, 0, 1, 1>",)})

    def roof(ms, flop, key, nbytes):
        tf = flop / (ms * 1e-3) / 1e12
        return {"bound": "fp64_valu", "kernel": syms.get(key, key),
                "achieved": tf, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
                "frac": tf / FP64_VALU_PEAK_TF, "traffic": None,
                "flop_per_launch": flop, "avg_launch_ms": ms,
                "objects_per_s": n / (ms * 1e-3),
                "hbm_algorithmic_GBs": nbytes * n / (ms * 1e-3) / 1e9}

    r_admom = roof(admom_ms, fl_admom, "admom", 16 * npx + 64 + 48 + 584)
    r_em = roof(em_ms, fl_em, "em", 16 * npx + 64 + 6 * 8 * 2 + 24)
    r_admom.update(load_issue("admom_grid_kernel<64, 16>", fl_admom / n))
    r_em.update(load_issue("em_wave_kernel<64, 16, 0, 1, 1>", fl_em / n))
    dom = r_em if em_ms >= admom_ms else r_admom
    line = {
        "metric": "objects/sec (admom + em_run), 32x32 stamps, 1/2/4/8 GPU",
        "value": world * n * K / elapsed,
        "unit": "objects/s",
        "n_gpus": world, "steps": K, "warmup": a2.warmup,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": "C4: %d stamps/GPU, 32x32 px, gaussian (x) gaussian psf; one "
                        "step = run_admom + 1-gaussian em_run of every stamp" % n,
            "stamps_per_gpu": n,
            "parallelism": "stamps sharded across %d rank(s); all-gather of 584-B "
                           "admom and 72-B EM records" % world,
        },
        "roofline": dom,
        "rooflines": {"admom": r_admom, "em_run": r_em},
        "mean_numiter": {"admom": it_admom, "em_run": it_em},
        "kernels_ms": {"admom": admom_ms, "em_run": em_ms},
        "bad_status": bad, "admom_flags_nonzero": flags_bad,
        "settle_steps": max(a2.settle_steps, 0),
    }
    return line


# --------------------------------------------------------------------------
# C5: multi-epoch 'bdf' loglike
# --------------------------------------------------------------------------

def make_c5(nobj, seed, device, nepoch=10, dim=64):
    """objects of `nepoch` 64x64 epochs, 16-gaussian 'bdf'
    (x) gaussian psf, per-epoch sub-pixel jacobian offsets
    (ngmix/tests/_sims.py:150-159); returns (StampBatch, GMixBatch, obj_start)"""
    import torch
    from ngmix_amd.batch import StampBatch, GMixBatch
    ns = nobj * nepoch
    rng = np.random.RandomState(seed)
    pars = np.zeros((nobj, 7))
    pars[:, 0:2] = rng.uniform(-0.3, 0.3, size=(nobj, 2)) * SCALE
    pars[:, 2:4] = rng.normal(scale=0.08, size=(nobj, 2))
    pars[:, 4] = rng.uniform(0.5, 2.0, size=nobj)
    pars[:, 5] = rng.uniform(0.2, 0.8, size=nobj)
    pars[:, 6] = rng.uniform(100, 400, size=nobj)
    spars = np.repeat(pars, nepoch, axis=0)
    jac = np.zeros((ns, 8))
    jac[:, 0] = (dim - 1) / 2 + rng.uniform(-0.5, 0.5, size=ns)
    jac[:, 1] = (dim - 1) / 2 + rng.uniform(-0.5, 0.5, size=ns)
    jac[:, 2] = jac[:, 5] = jac[:, 7] = SCALE
    jac[:, 6] = SCALE ** 2
    gm0, _ = GMixBatch.from_pars(spars, "bdf", device=device)
    psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (ns, 1)),
                                 "gauss", device=device)
    gm, _ = gm0.convolve(psf)
    gm.set_norms()
    jt = torch.from_numpy(jac).to(device)
    shape = np.full(ns, dim)
    off = np.arange(ns, dtype=np.int64) * dim * dim
    geom = StampBatch(None, None, jt, shape, shape, off, True)
    truth, _ = geom.render(gm)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed - 2)
    val = truth
    val += 0.05 * torch.randn(truth.shape, generator=gen, device=device,
                              dtype=torch.float64)
    ierr = torch.full_like(val, 20.0)
    sb = StampBatch(val, ierr, jt, shape, shape, off, True)
    return sb, gm, np.arange(nobj + 1) * nepoch


def run_c5(args, rank, world, device, backend, nobj=None, steps=None):
    """one step = the joint loglike of every object: 10 epochs x 64x64 pixels,
    16-gaussian 'bdf' (x) gaussian psf, float64, summed over the object's
    epochs on the device; 32-byte per-object records all-gathered"""
    import torch
    nobj = nobj or args.nstamps or 20000
    nepoch, dim = 10, 64
    ns = nobj * nepoch
    sb, gm, obj_start = make_c5(nobj, 3 + rank, device, nepoch, dim)
    out = torch.empty((ns, 4), dtype=torch.float64, device=device)
    status = torch.empty(ns, dtype=torch.int32, device=device)
    distributed = world > 1
    gat = Gather(world, device, [("c5", (nobj, 4), torch.float64)])
    K = steps or args.steps
    ev0, ev1, ev2 = _events(K), _events(K), _events(K)
    holder = {}

    def step(i):
        gat.wait_consumed()
        if i is not None:
            ev0[i].record()
        sb.loglike(gm, out=out, status=status)
        if i is not None:
            ev1[i].record()
        per_obj = sb.sum_over_epochs(out, obj_start)
        if i is not None:
            ev2[i].record()
        holder["per_obj"] = per_obj
        gat.gather("c5", per_obj)

    import copy
    a2 = copy.copy(args)
    a2.steps = K
    if steps is not None:
        a2.warmup, a2.settle_steps = 5, 160   # ~0.45 s of load before the clock starts
    elapsed = timed_steps(step, a2, distributed, device, gat)
    ll_ms = _mean_ms(ev0, ev1)
    red_ms = _mean_ms(ev1, ev2)
    bad = int((status != 0).sum().item())
    if rank != 0:
        return None
    epoch_bytes = 16 * dim * dim + 64 + 48 + 32      # 65,680 B
    achieved = epoch_bytes * ns / (ll_ms * 1e-3) / 1e9
    syms = kernel_symbols({"loglike": (LOGLIKE_KERNEL, "false")})
    c5_traffic = load_traffic("c5_loglike", ns, key="c5_nstamps")
    return {
        "metric": "object loglikes/sec, 10 epochs x 64x64, 16-gaussian 'bdf', 1/2/4/8 GPU",
        "value": world * nobj * K / elapsed,
        "unit": "object loglikes/s",
        "n_gpus": world, "steps": K, "warmup": a2.warmup,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": "C5: %d objects/GPU x %d epochs of 64x64 px, 16-gaussian 'bdf' "
                        "(x) gaussian psf; one step = joint loglike of every object"
                        % (nobj, nepoch),
            "objects_per_gpu": nobj,
            "parallelism": "objects (all their epochs) sharded across %d rank(s); "
                           "all-gather of 32-B per-object records" % world,
        },
        "roofline": {
            "bound": "hbm", "kernel": syms.get("loglike", "pixpass_wave_kernel (loglike)"),
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": c5_traffic[0],
            "traffic_source": c5_traffic[1],
            "algorithmic_bytes_per_launch": epoch_bytes * ns, "avg_launch_ms": ll_ms,
            # (on this shape the kernel is instruction-issue bound, not HBM bound)
            **load_issue("pixpass_wave_kernel7<0, false, 8>", c5=True),
        },
        "kernels_ms": {"loglike": ll_ms, "epoch_reduce": red_ms},
        "pixel_gaussian_evals_per_s": world * ns * dim * dim * 16 * K / elapsed,
        "bad_status": bad, "settle_steps": max(a2.settle_steps, 0),
    }


# --------------------------------------------------------------------------
# CPU baseline (the checker, timed beside the product; never the product)
# --------------------------------------------------------------------------

def host_threads():
    """(threads this process may actually use, description): the smaller of
    the affinity mask and the cgroup CPU quota -- omp_get_max_threads() alone
    reported the node's 128 cores for a pod that is given far fewer (round-2
    verdict, weak 7)"""
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:           # cgroup v2
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:                                                # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = float(f.read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    usable = aff if quota is None else max(1, min(aff, int(np.ceil(quota))))
    return usable, {"affinity": aff, "cgroup_quota_cpus": quota,
                    "os_cpu_count": os.cpu_count()}


def _rate(fn, nunits, budget):
    """units/s of fn() over about `budget` seconds (after one untimed call)"""
    fn()
    t0 = time.perf_counter()
    fn()
    t1 = time.perf_counter() - t0
    reps = int(max(1, min(1000, budget / max(t1, 1e-6))))
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return nunits * reps / (time.perf_counter() - t0), reps


def _best_team(run, usable, probe=0.4):
    """the thread count that gives the highest rate on this (possibly shared
    or quota-limited) host: run(threads) -> units/s measured over `probe`
    seconds, tried for the usable count and its halvings down to 8"""
    teams, t = [], usable
    while t >= 8:
        teams.append(t)
        t //= 2
    if not teams:
        teams = [usable]
    rates = {t: run(t, probe) for t in teams}
    best = max(rates, key=rates.get)
    return best, rates


def cpu_baseline(sb, gm, target_seconds=12.0):
    """the CPU oracle (a port of the numba loops: oracle/ngmix_oracle.c) timed
    on this box's host cores on a bounded sample of the same workload"""
    from oracle import oracle as ora
    usable, hostinfo = host_threads()
    S = min(sb.n, 64 * max(usable, 1), 4096)
    gmh = gm.to_numpy()[:S]
    gm_all = np.zeros((S, NGAUSS), dtype=ora.GAUSS2D_DTYPE)
    for name in ora.GAUSS2D_DTYPE.names:
        gm_all[name] = gmh[name]
    val = sb.val[:S * NPIX].cpu().numpy().reshape(S, NROW, NCOL)
    ierr = sb.ierr[:S * NPIX].cpu().numpy().reshape(S, NROW, NCOL)
    jac = np.zeros(1, dtype=ora.JACOBIAN_DTYPE)
    jac[0] = tuple(sb.jac[0].cpu().numpy())
    pixels = np.zeros((S, NPIX), dtype=ora.PIXEL_DTYPE)
    coords = np.zeros((S, NPIX), dtype=ora.COORD_DTYPE)
    for i in range(S):
        ora.fill_pixels(pixels[i], val[i], ierr[i] ** 2, jac, True)
        ora.fill_coords(coords[i], NROW, NCOL, jac)
    images = np.zeros((S, NPIX))

    def run(threads, budget):
        m = S if threads > 1 else min(S, 32)
        r, _ = _rate(lambda: ora.render_loglike_batch(gm_all[:m], pixels[:m], coords[:m],
                                                      images[:m], threads), m, budget)
        return 2.0 * r * PAIRS_PER_STAMP

    best, probe = _best_team(run, usable)
    t0 = time.perf_counter()
    value = run(best, target_seconds)
    dt = time.perf_counter() - t0
    single = run(1, 2.0)
    return {
        "value": value,
        "unit": "pixel-gaussian evals/s",
        "cores": int(best),
        "kind": "port",
        "sample": "%d stamps, render+loglike (48x48x6) repeated for %.0f s, OpenMP "
                  "over stamps, C port of the numba loops, -O2 no-FMA" % (S, target_seconds),
        "seconds": dt,
        "single_core_value": single,
        # threads this process may use (affinity mask and cgroup quota) and the
        # rates of the team sizes tried: `cores` is the best of them.  The GPU
        # box's host is shared with the node's other pods, so the all-thread
        # figure is a lower bound and the single-core one is the stable number
        "host": hostinfo,
        "usable_threads": usable,
        "team_probe": {str(k): v for k, v in probe.items()},
        "shared_host": True,
    }


class _CpuWorkloads(object):
    """small CPU-side stand-ins of the C1/C2/C4/C5 workloads for the C port
    (oracle/ngmix_oracle.c): the reference's AoS pixel arrays, its mixtures"""

    def __init__(self, seed=1):
        from oracle import oracle as ora
        self.ora = ora
        self.rng = np.random.RandomState(seed)

    def jac(self, dim, dr=0.0, dc=0.0):
        j = np.zeros(1, dtype=self.ora.JACOBIAN_DTYPE)
        c = (dim - 1) / 2.0
        j[0] = (c + dr, c + dc, SCALE, 0.0, 0.0, SCALE, SCALE * SCALE, SCALE)
        return j

    def mixture(self, pars, model, psf_T=0.27):
        ora = self.ora
        ng = {"gauss": 1, "exp": 6, "bdf": 16}[model]
        gm = np.zeros(ng, dtype=ora.GAUSS2D_DTYPE)
        if model == "bdf":
            ora.gmix_fill(gm, np.asarray(pars, dtype="f8"), "bdf")
        else:
            ora.gmix_fill(gm, np.asarray(pars, dtype="f8"), model)
        psf = np.zeros(1, dtype=ora.GAUSS2D_DTYPE)
        ora.gmix_fill(psf, np.array([0.0, 0.0, 0.0, 0.0, psf_T, 1.0]), "gauss")
        out = np.zeros(ng, dtype=ora.GAUSS2D_DTYPE)
        ora.gmix_convolve_fill(out, gm, psf)
        ora.gmix_set_norms(out)
        return gm, psf, out

    def stamps(self, n, dim, model):
        ora, rng = self.ora, self.rng
        gms, pix = [], np.zeros((n, dim * dim), dtype=ora.PIXEL_DTYPE)
        coords_all = np.zeros((n, dim * dim), dtype=ora.COORD_DTYPE)
        pars_all = []
        for i in range(n):
            # (config 5: sub-pixel offsets per epoch, _sims.py:150-159)
            off = rng.uniform(-0.5, 0.5, size=2) if model == "bdf" else (0.0, 0.0)
            j = self.jac(dim, off[0], off[1])
            coords = ora.make_coords((dim, dim), j)
            pars = [rng.uniform(-0.5, 0.5) * SCALE, rng.uniform(-0.5, 0.5) * SCALE,
                    rng.normal(scale=0.05), rng.normal(scale=0.05),
                    rng.uniform(0.3, 0.9)]
            if model == "bdf":
                pars += [rng.uniform(0.2, 0.8)]
            pars += [rng.uniform(50, 200)]
            gm0, psf, gm = self.mixture(pars, model)
            im = np.zeros(dim * dim)
            ora.render(gm, coords, im, fast_exp=1)
            im += 0.01 * rng.normal(size=im.size)
            ora.fill_pixels(pix[i], im.reshape(dim, dim), np.full((dim, dim), 1.0e4), j, True)
            gms.append(gm)
            coords_all[i] = coords
            pars_all.append(pars)
        return np.array(gms), pix, coords_all, np.array(pars_all)


def cpu_baseline_c4(budget=1.5, usable=None, teams=None, verbose=False):
    """config-4 cpu_baseline legs: run_admom and a 1-gaussian em_run over 32x32
    stamps with the C port, one core and the best team of host threads"""
    w = _CpuWorkloads(4)
    ora = w.ora
    if usable is None:
        usable, _ = host_threads()
    n4 = max(128, 16 * usable)
    gm, pix, _, pars = w.stamps(min(n4, 1024), 32, "gauss")
    n4 = pix.shape[0]
    conf = np.zeros(1, dtype=ora.ADMOM_CONF_DTYPE)
    conf["maxiter"], conf["shiftmax"], conf["etol"], conf["Ttol"] = 200, 5.0, 1e-5, 1e-3
    info = {}

    def admom_rate(threads, seconds):
        m = n4 if threads > 1 else 32
        wt0 = np.zeros(m, dtype=ora.GAUSS2D_DTYPE)
        for i in range(m):
            g = np.zeros(1, dtype=ora.GAUSS2D_DTYPE)
            ora.gmix_fill(g, np.array([0.0, 0.0, 0.0, 0.0, pars[i, 4] + 0.27, 1.0]), "gauss")
            wt0[i] = g[0]
        res = np.zeros(m, dtype=ora.ADMOM_RESULT_DTYPE)
        dt, n = 0.0, 0
        while dt < seconds:
            wt = wt0.copy()
            t0 = time.perf_counter()
            ora.admom_batch(conf, wt, pix[:m], res, threads)
            dt += time.perf_counter() - t0
            n += m
        assert np.all(res["flags"] == 0)
        info["admom_numiter"] = float(np.mean(res["numiter"]))
        return n / dt

    econf = np.zeros(1, dtype=ora.EM_CONF_DTYPE)
    econf["tol"], econf["maxiter"], econf["miniter"], econf["sky"] = 1e-5, 500, 40, 0.05
    g0 = np.zeros((n4, 1), dtype=ora.GAUSS2D_DTYPE)
    psf = np.zeros((n4, 1), dtype=ora.GAUSS2D_DTYPE)
    conv = np.zeros((n4, 1), dtype=ora.GAUSS2D_DTYPE)
    for i in range(n4):
        p = pars[i].copy()
        p[5] *= SCALE * SCALE
        g0[i], psf[i], conv[i] = w.mixture(p, "gauss")
    px0 = pix.copy()
    px0["val"] += 0.05

    def em_rate(threads, seconds):
        m = n4 if threads > 1 else 32
        dt, n = 0.0, 0
        while dt < seconds:
            a, b, c, px = g0[:m].copy(), psf[:m].copy(), conv[:m].copy(), px0[:m].copy()
            t0 = time.perf_counter()
            numiter, status = ora.em_batch(econf, px, a, b, c, threads)
            dt += time.perf_counter() - t0
            n += m
        assert np.all(status == 0)
        info["em_numiter"] = float(np.mean(numiter))
        return n / dt

    out = {}
    for name, fn in (("admom", admom_rate), ("em_run", em_rate)):
        if teams is None:
            best, probe = _best_team(fn, usable, probe=0.25)
        else:
            probe = {t: fn(t, budget) for t in teams if t > 1}
            best = max(probe, key=probe.get) if probe else 1
        allc = fn(best, budget) if teams is None else probe.get(best, fn(1, budget))
        one = fn(1, budget)
        out[name] = {"value": allc, "unit": "objects/s", "cores": int(best), "kind": "port",
                     "single_core_value": one,
                     "team_probe": {str(k): v for k, v in probe.items()},
                     "mean_numiter": info[name.split("_")[0] + "_numiter"],
                     "sample": "%d 32x32 stamps repeated for %.1f s, C port of %s" % (
                         n4, budget, "admom_nb.admom" if name == "admom" else
                         "em_nb.em_run (1 gaussian (x) 1-gaussian psf)")}
        if verbose:
            print("C4 %s 32x32: 1 thread %.3g objects/s; %s" % (name, one, ", ".join(
                "%s threads %.3g" % kv for kv in sorted(out[name]["team_probe"].items(),
                                                        key=lambda kv: int(kv[0])))))
    return out


def c3_lmder_fit(ora, pix, psf, guess, work=None):
    """ONE complete 'exp' (x) psf LM fit of a stamp's pixel list on this core,
    the reference's algorithm: scipy's MINPACK lmder (what run_leastsq calls
    with Dfun = FitModel.calc_jacobian, fitters.py:93-104) at DEFAULT_LM_PARS
    (ftol = xtol = 1e-5, maxfev 4000: defaults.py:17) around the C port of
    fill_fdiff and of deriv_images (results.py:439-570, derivs_nb.py:40-127).
    Returns leastsq's full output.  (tests/test_oracle_lm.py holds it to the
    reference's own fits, nfev for nfev.)"""
    from scipy.optimize import leastsq
    npix = pix.shape[0]
    ng = 6 * psf.size
    if work is None:
        work = {}
    if work.get("npix") != npix or work.get("ng") != ng:
        work.update(npix=npix, ng=ng, gm0=np.zeros(6, dtype=ora.GAUSS2D_DTYPE),
                    gm=np.zeros(ng, dtype=ora.GAUSS2D_DTYPE), fd=np.zeros(npix),
                    dimg=np.zeros((6, npix)), jac=np.zeros((npix, 6)))
    gm0, gm, fd, dimg, jac = (work[k] for k in ("gm0", "gm", "fd", "dimg", "jac"))
    vv = np.ascontiguousarray(pix["v"])
    uu = np.ascontiguousarray(pix["u"])
    area = np.ascontiguousarray(pix["area"])
    ierr = np.ascontiguousarray(pix["ierr"])
    dcov_de = np.array([[-1.0, 0.0, 1.0], [0.0, 1.0, 0.0]])

    def fill(p):
        # out of range -> GMixRangeError in the reference
        if p[2] * p[2] + p[3] * p[3] >= 1.0 or p[4] <= 0.0 or p[5] == 0.0:
            return False
        ora.gmix_fill(gm0, p, "exp")
        ora.gmix_convolve_fill(gm, gm0, psf)
        ora.gmix_set_norms(gm)
        return True

    def resid(p):
        if not fill(p):
            return np.full(npix, -9.999e9)      # LOWVAL (results.py:461-463)
        ora.fill_fdiff(gm, pix, fd, 0)
        return fd.copy()

    def dfun(p):
        # FitModel.calc_jacobian: composed gaussians + d cov / d(g1, g2, T)
        # (results.py:955-1010), deriv_images, rows scaled by ierr
        if not fill(p):
            return np.zeros((npix, 6))
        g1, g2, T, flux = p[2:6]
        gpars = np.stack([gm[k] for k in ("p", "row", "col", "irr", "irc", "icc")], axis=1)
        mcov = np.repeat(np.stack([gm0[k] for k in ("irr", "irc", "icc")], axis=1),
                         psf.size, axis=0)
        half_tk = 0.5 * (mcov[:, 0] + mcov[:, 2])
        gsq = g1 * g1 + g2 * g2
        f = 2.0 / (1.0 + gsq)
        g = np.array([g1, g2])
        jac_e = f * np.eye(2) + (2.0 * g[:, None] * g[None, :]) * (-f / (1.0 + gsq))
        dcov = np.empty((ng, 3, 3))
        dcov[:, 0:2, :] = half_tk[:, None, None] * (jac_e @ dcov_de)[None, :, :]
        dcov[:, 2, :] = mcov / T
        dimg[:] = 0.0       # deriv_images accumulates (derivs_nb.py:107-125)
        ora.deriv_images(gpars, dcov, vv, uu, area, dimg)
        for k in range(5):
            jac[:, k] = dimg[1 + k] * ierr
        jac[:, 5] = dimg[0] * (ierr / flux)
        return jac
    return leastsq(resid, guess, Dfun=dfun, full_output=1, ftol=1.0e-5, xtol=1.0e-5,
                   maxfev=4000)


def _c3_cpu_fits(budget, seed, nstamps=16):
    """Complete 'exp' (x) gaussian-psf LM fits of 48x48 stamps on THIS core
    for `budget` seconds (c3_lmder_fit: the reference's algorithm).  Returns
    (fits, seconds, sum of nfev, fits that ended with ier 1..4)."""
    w = _CpuWorkloads(seed)
    ora = w.ora
    _, pix, _, pars = w.stamps(nstamps, 48, "exp")
    psf = np.zeros(1, dtype=ora.GAUSS2D_DTYPE)
    ora.gmix_fill(psf, np.array([0.0, 0.0, 0.0, 0.0, 0.27, 1.0]), "gauss")
    rng = np.random.RandomState(33 + seed)
    stat = {"nfev": 0, "ok": 0}
    work = {}

    def fit(i):
        guess = pars[i] * rng.uniform(0.9, 1.1, size=6)
        guess[0:2] = pars[i][0:2] + rng.uniform(-0.05, 0.05, size=2)
        guess[2:4] = pars[i][2:4] + rng.uniform(-0.03, 0.03, size=2)
        out = c3_lmder_fit(ora, pix[i], psf, guess, work)
        stat["nfev"] += out[2]["nfev"]
        stat["ok"] += int(1 <= out[4] <= 4)

    fit(0)
    stat["nfev"] = stat["ok"] = 0
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < budget:
        fit(n % nstamps)
        n += 1
    return n, time.perf_counter() - t0, stat["nfev"], stat["ok"]


def cpu_baseline_c3(budget=1.5, usable=None, verbose=False):
    """config-3 cpu_baseline leg, the reference's own algorithm on the host
    cores: MINPACK lmder with the analytic jacobian at ftol = xtol = 1e-5
    (_c3_cpu_fits), on one core and on every core the process may use -- one
    worker PROCESS per core (MINPACK's callbacks hold the GIL, as in the
    reference), started as fresh interpreters that never touch the GPU."""
    if usable is None:
        usable, _ = host_threads()
    n1, dt1, nfev1, ok1 = _c3_cpu_fits(budget, 3)
    single = n1 / dt1
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-c3-worker", "%g" % budget]
    procs = [subprocess.Popen(cmd + [str(100 + k)], stdout=subprocess.PIPE, text=True, cwd=ROOT)
             for k in range(usable)]
    rates, nfits, nfev, ok = 0.0, 0, 0, 0
    for pr in procs:
        out, _ = pr.communicate(timeout=120 + 4 * budget)
        rec = json.loads(out.strip().splitlines()[-1])
        rates += rec["n"] / rec["seconds"]
        nfits += rec["n"]
        nfev += rec["nfev"]
        ok += rec["ok"]
    mean_nfev = nfev / float(max(nfits, 1))
    if verbose:
        print("C3 'exp' LM fit 48x48 (scipy MINPACK lmder + C fill_fdiff / deriv_images, "
              "ftol = xtol = 1e-5): 1 core %.3g fits/s (mean nfev %.2f); %d worker processes "
              "%.3g fits/s (mean nfev %.2f, %d of %d converged)" % (
                  single, nfev1 / float(max(n1, 1)), usable, rates, mean_nfev, ok, nfits))
    return {"value": rates, "unit": "fits/s", "cores": int(usable), "kind": "port",
            "single_core_value": single, "mean_nfev": mean_nfev,
            "single_core_mean_nfev": nfev1 / float(max(n1, 1)),
            "converged": ok, "fits": nfits,
            "jacobian": "analytic (C port of deriv_images through Dfun, as the reference)",
            "sample": "%d fits of 48x48 'exp' (x) gaussian-psf stamps in %.1f s on %d worker "
                      "processes (one per usable core), and %d in %.1f s on one core: scipy "
                      "MINPACK lmder at ftol = xtol = 1e-5 (the reference's DEFAULT_LM_PARS) "
                      "around the C port of fill_fdiff / deriv_images / gmix_fill / "
                      "gmix_convolve_fill" % (nfits, budget, usable, n1, dt1)}


def cpu_baseline_c5(budget=1.5, usable=None, nepoch=10, verbose=False):
    """config-5 cpu_baseline leg: the joint loglike of objects with 10 epochs
    of 64x64 pixels under a 16-gaussian 'bdf' (x) gaussian-psf mixture"""
    w = _CpuWorkloads(5)
    ora = w.ora
    if usable is None:
        usable, _ = host_threads()
    nobj = max(4, min(32, usable // 4))
    gm, pix, _, _ = w.stamps(nobj * nepoch, 64, "bdf")
    gm = np.ascontiguousarray(gm)

    def rate(threads, seconds):
        m = nobj * nepoch if threads > 1 else 2 * nepoch
        r, _ = _rate(lambda: ora.loglike_batch(gm[:m], pix[:m], threads), m / nepoch, seconds)
        return r

    best, probe = _best_team(rate, usable, probe=0.25)
    allc = rate(best, budget)
    one = rate(1, budget)
    if verbose:
        print("C5 joint loglike 10 x 64x64 x 16: 1 thread %.3g objects/s; %s" % (
            one, ", ".join("%s threads %.3g" % kv for kv in sorted(
                probe.items(), key=lambda kv: int(kv[0])))))
    return {"value": allc, "unit": "object loglikes/s", "cores": int(best), "kind": "port",
            "single_core_value": one,
            "team_probe": {str(k): v for k, v in probe.items()},
            "sample": "%d objects x %d epochs of 64x64, 16 gaussians, repeated for %.1f s, "
                      "C port of gmix_nb.get_loglike" % (nobj, nepoch, budget)}


def cpu_baseline_configs(budget=4.0):
    """the cpu_baseline legs of SURVEY.md section 8(d) on their own: the C
    port of the numba loops (oracle/ngmix_oracle.c, -O2, no FMA contraction)
    on this box's host cores, one core and teams of threads up to what the
    process may use, for C1 (one 48x48x6 stamp), C2 (render + loglike), C4
    (admom and 1-gaussian em_run over 32x32 stamps) and C5.  No GPU is used:
    python bench.py --cpu-baselines [seconds per leg]"""
    from oracle import oracle as ora
    usable, hostinfo = host_threads()
    print("host threads usable: %d (%s; omp_get_max_threads %d)" % (
        usable, hostinfo, ora.num_threads()))
    w = _CpuWorkloads(1)
    n2 = max(64, 32 * usable)
    gm, pix, coords, _ = w.stamps(min(n2, 2048), 48, "exp")
    n2 = pix.shape[0]
    images = np.zeros((n2, 48 * 48))
    teams, t = [1], usable
    while t > 1:
        teams.append(t)
        t //= 2
    for threads in sorted(set(teams)):
        m = n2 if threads > 1 else 16
        r, _ = _rate(lambda: ora.render_loglike_batch(gm[:m], pix[:m], coords[:m], images[:m],
                                                      threads), m, budget)
        print("C2 render+loglike 48x48x6: %3d thread(s): %.3g stamp evals/s (x2 kernels) = "
              "%.3g pixel-gaussian evals/s" % (threads, r, 2 * r * 48 * 48 * 6))
        if threads == 1:
            print("C1 one stamp, render + loglike: %.1f us" % (1e6 / r))
    cpu_baseline_c3(budget, usable, verbose=True)
    cpu_baseline_c4(budget, usable, verbose=True)
    cpu_baseline_c5(budget, usable, verbose=True)


def baseline_metric():
    """BASELINE.json's metric string, verbatim"""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except (OSError, KeyError, ValueError):
        return ("pixel-Gaussian evals/sec (render+loglike), 48×48×6-gauss stamps, "
                "1/2/4/8 GPU")


def load_traffic(kernel, nstamps, key="nstamps", field="_hbm_bytes_per_launch"):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC pass
    (counters cannot be read from inside the timed run: rocprofv3 --pmc is a
    separate, serialising pass).  Returns (bytes or None, source or None)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None, None
    if t.get(key) != nstamps:
        return None, None
    v = t.get(kernel + field)
    if v is None:
        return None, None
    return v, "profiles/pmc_traffic.json (%s)" % t.get("source", "rocprofv3 --pmc pass")


def load_issue(kernel_fragment, flop_per_stamp=None, c5=False):
    """instruction-issue figures of a VALU-bound kernel from the committed
    rocprofv3 PMC pass (profiles/pmc_valu.json, written by
    tools/make_profiles.py): the fraction of the SIMDs' issue slots the kernel
    fills (valu_busy), its VALU wave-instructions per stamp, and -- with a flop
    model -- the flops one lane-instruction carries.  A bare roofline fraction
    of 0.26 reads differently next to '87 % of the issue slots, 0.6 flop per
    slot'.  {} when the file or the kernel is missing."""
    path = os.path.join(ROOT, "profiles", "pmc_valu.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return {}
    for name, fig in table.items():
        if not isinstance(fig, dict) or kernel_fragment not in name:
            continue
        if c5 != name.startswith("c5:"):
            continue
        out = {"valu_busy": fig["valu_busy"],
               "valu_insts_per_stamp": fig["valu_insts_per_stamp"],
               "issue_source": table.get("source")}
        if "salu_insts_per_stamp" in fig:
            out["salu_insts_per_stamp"] = fig["salu_insts_per_stamp"]
        if flop_per_stamp:
            out["flop_per_lane_inst"] = flop_per_stamp / (fig["valu_insts_per_stamp"] * 64.0)
        return out
    return {}


def main():
    if len(sys.argv) >= 3 and sys.argv[1] == "--cpu-c3-worker":
        # one worker of cpu_baseline_c3: a fresh interpreter, no GPU, no torch
        n, dt, nfev, ok = _c3_cpu_fits(float(sys.argv[2]),
                                       int(sys.argv[3]) if len(sys.argv) > 3 else 1)
        print(json.dumps({"n": n, "seconds": dt, "nfev": nfev, "ok": ok}))
        return 0
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", default="C2", choices=["C2", "C3", "C4", "C5"])
    ap.add_argument("--settle-steps", type=int, default=None,
                    help="untimed steps run before the warm-up steps: the GPU's "
                         "clock governor needs ~50-100 ms of load to reach its "
                         "steady state (DESIGN.md section 5); 0 disables "
                         "(default: 100 for C2, 10 for C4 / C5, 3 for C3)")
    ap.add_argument("--nstamps", type=int, default=None,
                    help="stamps (C5: objects) per GPU (weak scaling); default "
                         "100000 (C2), 125000 (C4), 20000 (C5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="C3: one synchronous LMBatchFitter.go() per step instead of the "
                         "software pipeline over the steps (go_stream)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="C2 at N = 1 only: skip the short C4 / C5 legs")
    ap.add_argument("--cpu-baselines", type=float, nargs="?", const=4.0, default=None,
                    metavar="SECONDS",
                    help="only time the CPU port on configs C1 / C2 / C4 (single core "
                         "and all cores; no GPU needed) and exit")
    ap.add_argument("--exact", action="store_true",
                    help="time the exact (no-FMA, bit-identical) kernels "
                         "instead of the default fused ones")
    args = ap.parse_args()
    if args.cpu_baselines is not None:
        cpu_baseline_configs(args.cpu_baselines)
        return 0
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.settle_steps is None:
        # (~0.45 s of uninterrupted load before the clock starts: what the
        # clock governor needs to reach its steady state)
        args.settle_steps = {"C2": 100, "C3": 70, "C4": 30, "C5": 160}[args.config]
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, sys.argv[1:])

    import torch.distributed as dist
    rank, world, device, backend = init_rank(args)
    run = {"C2": run_c2, "C3": run_c3, "C4": run_c4, "C5": run_c5}[args.config]
    line, err = run_leg(args.config, run, world, device, args, rank, world, device, backend)
    if err is not None:
        raise err
    if rank == 0:
        line["rccl_ranks"] = dist.get_world_size() if world > 1 else 1
        line["backend"] = backend or "none (single process)"
        line["per_rank_ms_per_step"] = list(PER_RANK_MS)
    other = {}
    if args.config == "C2" and not args.no_other_configs:
        # a default run carries a short leg of every other config, so that ONE
        # driver record holds C2-C5 -- at N > 1 too: every rank runs the legs
        # (each shards its config as its own --config run does and gathers its
        # records), rank 0 keeps the lines with every rank's own ms per step
        import torch
        torch.cuda.empty_cache()
        # (short legs, but long enough for the clock governor: each runs its
        # own settle + warm-up steps, ~0.45 s of load, before the timed ones)
        # (--nstamps, a testing knob, caps the legs' sizes too)
        cap = args.nstamps or (1 << 30)
        for name, fn, kw in (("C3", run_c3, dict(nstamps=min(100000, cap), steps=30)),
                             ("C4", run_c4, dict(nstamps=min(125000, cap), steps=16)),
                             ("C5", run_c5, dict(nobj=min(20000, max(cap // 10, 8)), steps=60))):
            # (a failure local to one rank is agreed on by all of them, run_leg:
            # every rank records it and goes on to the next leg)
            o, err = run_leg(name, fn, world, device, args, rank, world, device, backend, **kw)
            if err is not None:   # never lose the headline line
                other[name] = {"error": repr(err)}
            elif rank == 0:
                other[name] = {k: o[k] for k in (
                    "metric", "value", "unit", "config", "roofline", "kernels_ms",
                    "bad_status") if k in o}
                for k in ("rooflines", "mean_numiter", "device_loop_ms", "rounds",
                          "fits_per_s_device_loop", "mean_nfev", "ms_per_step",
                          "kernels_ms_sum", "host_ms_per_step", "rounds_launched",
                          "pipelined", "steps",
                          "settle_steps", "n_gpus"):
                    if k in o:
                        other[name][k] = o[k]
                other[name]["per_rank_ms_per_step"] = list(PER_RANK_MS)
            torch.cuda.empty_cache()
        if world == 1 and "error" not in other.get("C3", {"error": 1}):
            # the same fits from host-resident arrays (PCIe inclusive: never `value`)
            try:
                other["C3_host"] = run_c3_host(device, n=min(100000, cap))
            except Exception as e:   # never lose the headline line
                other["C3_host"] = {"error": repr(e)}
            torch.cuda.empty_cache()
            try:
                other["prepsfmom"] = run_prepsf(device, n=min(100000, cap))
            except Exception as e:   # never lose the headline line
                other["prepsfmom"] = {"error": repr(e)}
            torch.cuda.empty_cache()
            # and with the reference's joint prior on the parameters
            try:
                other["C3_prior"] = run_c3_prior(device, n=min(100000, cap))
            except Exception as e:   # never lose the headline line
                other["C3_prior"] = {"error": repr(e)}
            torch.cuda.empty_cache()
    if rank == 0:
        if other:
            if world == 1 and not args.no_cpu_baseline:
                # the C port on this box's host cores next to
                # the GPU figures, bounded to ~2 s per leg
                try:
                    usable, _ = host_threads()
                    other.setdefault("C3", {})["cpu_baseline"] = cpu_baseline_c3(
                        budget=2.0, usable=usable)
                    c4 = cpu_baseline_c4(budget=1.0, usable=usable)
                    if "rooflines" in other.get("C4", {}):
                        other["C4"]["cpu_baseline"] = c4
                    other.setdefault("C5", {})["cpu_baseline"] = cpu_baseline_c5(
                        budget=1.0, usable=usable)
                except Exception as e:
                    other["cpu_baseline_error"] = repr(e)
            line["other_configs"] = other
        print(json.dumps(line))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
