"""CPU baselines of SURVEY.md section 8(d): the C port of the numba loops
(oracle/ngmix_oracle.c, -O2, no FMA contraction) on this box's host cores,
single core and all cores, for

    C1  one 48x48x6 stamp: get_loglike + fast render latency
    C2  render + get_loglike over 48x48x6 stamps
    C4  admom and 1-gaussian em_run over 32x32 stamps

No GPU is used.  python tools/cpu_baseline.py [seconds_per_leg]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as ora  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
nth = ora.num_threads()
scale = 0.263


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
