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

    python bench.py [--gpus N] [--steps K] [--warmup W] [--nstamps S]

N > 1 is launched by torch.distributed.run (one process per GPU, RCCL).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NROW = NCOL = 48
NGAUSS = 6
NPIX = NROW * NCOL
PAIRS_PER_STAMP = NPIX * NGAUSS  # 13,824 pixel-gaussian evaluations
SCALE = 0.263
# algorithmic bytes per stamp evaluation (SURVEY.md 8d / BASELINE.md section 3)
LOGLIKE_BYTES = 16 * NPIX + 64 + 48 + 32   # 37,008
RENDER_BYTES = 16 * NPIX                   # 36,864 (8 read + 8 written)
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: 8.0 TB/s spec


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


def cpu_baseline(sb, gm, target_seconds=12.0):
    """the CPU oracle (a port of the numba loops: oracle/ngmix_oracle.c) timed
    on this box's host cores on a bounded sample of the same workload"""
    from oracle import oracle as ora
    nthreads = ora.num_threads()
    S = min(sb.n, 64 * max(nthreads, 1), 4096)
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
    ora.render_loglike_batch(gm_all[:8], pixels[:8], coords[:8], images[:8],
                             nthreads)  # warm up threads / pages
    t0 = time.perf_counter()
    ora.render_loglike_batch(gm_all, pixels, coords, images, nthreads)
    t1 = time.perf_counter() - t0
    reps = int(max(1, min(200, target_seconds / max(t1, 1e-6))))
    t0 = time.perf_counter()
    for _ in range(reps):
        ora.render_loglike_batch(gm_all, pixels, coords, images, nthreads)
    dt = time.perf_counter() - t0
    pairs = 2.0 * S * PAIRS_PER_STAMP * reps
    # and one core on a slice of the sample (SURVEY 8d asks for both)
    S1 = min(S, 32)
    t0 = time.perf_counter()
    ora.render_loglike_batch(gm_all[:S1], pixels[:S1], coords[:S1], images[:S1], 1)
    t1 = time.perf_counter() - t0
    reps1 = int(max(1, min(50, 2.0 / max(t1, 1e-6))))
    t0 = time.perf_counter()
    for _ in range(reps1):
        ora.render_loglike_batch(gm_all[:S1], pixels[:S1], coords[:S1], images[:S1], 1)
    single = 2.0 * S1 * PAIRS_PER_STAMP * reps1 / (time.perf_counter() - t0)
    return {
        "value": pairs / dt,
        "unit": "pixel-gaussian evals/s",
        "cores": int(nthreads),
        "kind": "port",
        "sample": "%d stamps x %d passes of render+loglike (48x48x6), OpenMP "
                  "over stamps, C port of the numba loops, -O2 no-FMA" % (S, reps),
        "seconds": dt,
        "single_core_value": single,
    }


def cpu_baseline_configs(budget=4.0):
    """the cpu_baseline leg for the other configs of SURVEY.md section 8(d):
    the C port of the numba loops (oracle/ngmix_oracle.c, -O2, no FMA
    contraction) on this box's host cores, one core / one thread per core /
    all hardware threads, for C1 (one 48x48x6 stamp), C2 (render + loglike)
    and C4 (admom and 1-gaussian em_run over 32x32 stamps).  No GPU is used:
    python bench.py --cpu-baselines [seconds per leg]"""
    from oracle import oracle as ora
    nth = ora.num_threads()
    scale = SCALE


    def jac(dim):
        j = np.zeros(1, dtype=ora.JACOBIAN_DTYPE)
        c = (dim - 1) / 2.0
        j[0] = (c, c, scale, 0.0, 0.0, scale, scale * scale, scale)
        return j


    def mixture(pars, model, psf_T=0.27):
        ng = {"gauss": 1, "exp": 6}[model]
        gm = np.zeros(ng, dtype=ora.GAUSS2D_DTYPE)
        ora.gmix_fill(gm, np.asarray(pars, dtype="f8"), model)
        psf = np.zeros(1, dtype=ora.GAUSS2D_DTYPE)
        ora.gmix_fill(psf, np.array([0.0, 0.0, 0.0, 0.0, psf_T, 1.0]), "gauss")
        out = np.zeros(ng, dtype=ora.GAUSS2D_DTYPE)
        ora.gmix_convolve_fill(out, gm, psf)
        ora.gmix_set_norms(out)
        return gm, psf, out


    def stamps(n, dim, model, rng):
        j = jac(dim)
        coords = ora.make_coords((dim, dim), j)
        gms, pix = [], np.zeros((n, dim * dim), dtype=ora.PIXEL_DTYPE)
        pars_all = []
        for i in range(n):
            pars = [rng.uniform(-0.5, 0.5) * scale, rng.uniform(-0.5, 0.5) * scale,
                    rng.normal(scale=0.05), rng.normal(scale=0.05),
                    rng.uniform(0.3, 0.9), rng.uniform(50, 200)]
            gm0, psf, gm = mixture(pars, model)
            im = np.zeros(dim * dim)
            ora.render(gm, coords, im, fast_exp=1)
            im += 0.01 * rng.normal(size=im.size)
            ora.fill_pixels(pix[i], im.reshape(dim, dim), np.full((dim, dim), 1.0e4), j, True)
            gms.append(gm)
            pars_all.append(pars)
        return np.array(gms), pix, np.tile(coords, (n, 1)), np.array(pars_all)


    def timed(fn, nunits):
        fn()
        t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter() - t0
        reps = int(max(1, min(1000, budget / max(t1, 1e-6))))
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        dt = time.perf_counter() - t0
        return nunits * reps / dt


    rng = np.random.RandomState(1)
    print("host threads available: %d" % nth)
    # all hardware threads and (SMT boxes) one thread per core
    TEAMS = sorted({1, max(1, nth // 2), nth})

    # ---- C1 / C2
    n2 = max(64, 32 * nth)
    gm, pix, coords, _ = stamps(n2, 48, "exp", rng)
    images = np.zeros((n2, 48 * 48))
    for threads in TEAMS:
        m = min(n2, 32 * threads) if threads > 1 else 16
        r = timed(lambda: ora.render_loglike_batch(gm[:m], pix[:m], coords[:m], images[:m],
                                                   threads), m)
        print("C2 render+loglike 48x48x6: %3d thread(s): %.3g stamp evals/s (x2 kernels) = "
              "%.3g pixel-gaussian evals/s" % (threads, r, 2 * r * 48 * 48 * 6))
        if threads == 1:
            print("C1 one stamp, render + loglike: %.1f us" % (1e6 / r))

    # ---- C4
    n4 = max(128, 64 * nth)
    gm, pix, _, pars = stamps(n4, 32, "gauss", rng)
    conf = np.zeros(1, dtype=ora.ADMOM_CONF_DTYPE)
    conf["maxiter"], conf["shiftmax"], conf["etol"], conf["Ttol"] = 200, 5.0, 1e-5, 1e-3
    for threads in TEAMS:
        m = min(n4, 64 * threads) if threads > 1 else 32

        def run_admom():
            wt = np.zeros(m, dtype=ora.GAUSS2D_DTYPE)
            for i in range(m):
                g = np.zeros(1, dtype=ora.GAUSS2D_DTYPE)
                ora.gmix_fill(g, np.array([0.0, 0.0, 0.0, 0.0, pars[i, 4] + 0.27, 1.0]), "gauss")
                wt[i] = g[0]
            res = np.zeros(m, dtype=ora.ADMOM_RESULT_DTYPE)
            t0 = time.perf_counter()
            ora.admom_batch(conf, wt, pix[:m], res, threads)
            run_admom.dt += time.perf_counter() - t0
            run_admom.n += m
            run_admom.iters = float(np.mean(res["numiter"]))
            assert np.all(res["flags"] == 0)
        run_admom.dt, run_admom.n = 0.0, 0
        run_admom()
        run_admom.dt, run_admom.n = 0.0, 0
        while run_admom.dt < budget:
            run_admom()
        print("C4 admom 32x32: %3d thread(s): %.3g objects/s (mean numiter %.1f)" % (
            threads, run_admom.n / run_admom.dt, run_admom.iters))

    econf = np.zeros(1, dtype=ora.EM_CONF_DTYPE)
    econf["tol"], econf["maxiter"], econf["miniter"], econf["sky"] = 1e-5, 500, 40, 0.05
    for threads in TEAMS:
        m = min(n4, 64 * threads) if threads > 1 else 32

        def run_em():
            g0 = np.zeros((m, 1), dtype=ora.GAUSS2D_DTYPE)
            psf = np.zeros((m, 1), dtype=ora.GAUSS2D_DTYPE)
            conv = np.zeros((m, 1), dtype=ora.GAUSS2D_DTYPE)
            for i in range(m):
                p = pars[i].copy()
                p[5] *= scale * scale
                a, b, c = mixture(p, "gauss")
                g0[i], psf[i], conv[i] = a, b, c
            px = pix[:m].copy()
            px["val"] += 0.05
            t0 = time.perf_counter()
            numiter, status = ora.em_batch(econf, px, g0, psf, conv, threads)
            run_em.dt += time.perf_counter() - t0
            run_em.n += m
            run_em.iters = float(np.mean(numiter))
            assert np.all(status == 0)
        run_em.dt, run_em.n = 0.0, 0
        run_em()
        run_em.dt, run_em.n = 0.0, 0
        while run_em.dt < budget:
            run_em()
        print("C4 em_run 32x32, 1 gaussian: %3d thread(s): %.3g objects/s (mean numiter %.1f)" % (
            threads, run_em.n / run_em.dt, run_em.iters))



def baseline_metric():
    """BASELINE.json's metric string, verbatim"""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except (OSError, KeyError, ValueError):
        return ("pixel-Gaussian evals/sec (render+loglike), 48\u00d748\u00d76-gauss stamps, "
                "1/2/4/8 GPU")


def load_traffic():
    """HBM bytes per loglike launch from the committed rocprofv3 PMC pass
    (profiles/), if one exists for this workload size"""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(path):
        try:
            return json.load(open(path))
        except Exception:
            return None
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--settle-steps", type=int, default=100,
                    help="untimed steps run before the warm-up steps: the GPU's "
                         "clock governor needs ~50-100 ms of load to reach its "
                         "steady state (DESIGN.md section 5); 0 disables")
    ap.add_argument("--nstamps", type=int, default=100000,
                    help="stamps per GPU (weak scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
        return

    import torch
    import torch.distributed as dist
    from ngmix_amd import distributed as nd

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP kernels are the product)")
    # one process per GPU; "nccl" is RCCL on ROCm (xGMI inside the node)
    rank, world, local_rank = nd.init_from_env(backend="nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    distributed = world > 1

    n = args.nstamps
    sb, gm, _ = make_workload(n, seed=1000 + rank, device=device)
    image = torch.zeros(sb.total_pix, dtype=torch.float64, device=device)
    out = torch.empty((n, 4), dtype=torch.float64, device=device)
    status = torch.empty(n, dtype=torch.int32, device=device)
    gathered = None
    side = None
    if distributed:
        gathered = torch.empty((world * n, 4), dtype=torch.float64, device=device)
        side = torch.cuda.Stream(device=device)

    ev_r0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev_r1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev_l1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    pending = None

    def step(i=None):
        nonlocal pending
        if i is not None:
            ev_r0[i].record()
        sb.render(gm, image=image, fast_exp=True, status=status, exact=args.exact)
        if i is not None:
            ev_r1[i].record()
        if pending is not None:
            # the previous step's gather must have consumed `out`
            torch.cuda.current_stream().wait_event(pending)
        sb.loglike(gm, out=out, status=status, exact=args.exact)
        if i is not None:
            ev_l1[i].record()
        if distributed:
            # north_star's all-gather of per-object result records, on a side
            # stream so it overlaps the next step's render
            done = torch.cuda.Event()
            done.record()
            with torch.cuda.stream(side):
                side.wait_event(done)
                nd.allgather_records(out, n_objects=world * n, out=gathered)
                pending = torch.cuda.Event()
                pending.record()

    # steady-state clocks first (untimed, like the warm-up steps that follow)
    # (a fixed count, so that every rank issues the same collectives)
    for _ in range(max(args.settle_steps, 0)):
        step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    bad = int((status != 0).sum().item())
    render_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev_r0, ev_r1)]))
    loglike_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev_r1, ev_l1)]))

    if rank == 0:
        pairs_per_step = 2.0 * world * n * PAIRS_PER_STAMP
        value = pairs_per_step * args.steps / elapsed
        dominant = "loglike" if loglike_ms >= render_ms else "render"
        dom_ms = max(loglike_ms, render_ms)
        dom_bytes = (LOGLIKE_BYTES if dominant == "loglike" else RENDER_BYTES) * n
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        traffic = load_traffic()
        traffic_bytes = None
        if traffic and traffic.get("nstamps") == n:
            traffic_bytes = traffic.get(dominant + "_hbm_bytes_per_launch")
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
                "kernel": "pixpass_%s_kernel<%s>" % (
                    "grid" if args.exact else "wave", dominant),
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic_bytes,
                "algorithmic_bytes_per_launch": dom_bytes,
                "avg_launch_ms": dom_ms,
            },
            "kernels_ms": {"render": render_ms, "loglike": loglike_ms},
            "loglike_stamp_evals_per_s_per_gpu": n / (loglike_ms * 1e-3),
            "render_stamp_evals_per_s_per_gpu": n / (render_ms * 1e-3),
            "bad_status": bad,
            "settle_steps": max(args.settle_steps, 0),
            "kernel_mode": "exact" if args.exact else "fused",
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sb, gm)
        print(json.dumps(line))
        sys.stdout.flush()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
