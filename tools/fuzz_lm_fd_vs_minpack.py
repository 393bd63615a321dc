"""The forward-difference (lmdif) fits of the lock-step LM driver that the
lmder fuzz (fuzz_lm_vs_minpack.py) does not draw -- co-elliptical psf fits with
1-5 gaussians (CoellipFitter: 6-14 parameters, the register and the team form
of the step) and the 'bdf' / 'bd' galaxy models over 1-2 bands -- against
scipy's MINPACK lmdif, fit by fit: the same stamps and guesses through
LMBatchFitter (normal equations accumulated inside the pixel pass, Cholesky
step) and through Fitter(batched=False) / CoellipFitter(batched=False) (MINPACK
calling the fdiff seam kernel once per evaluation, QR of the jacobian).

Per class of fit: how many fits agree in flags and in ier, the histogram of
nfev(driver) - nfev(MINPACK) in units of one jacobian (n + 1 evaluations), the
converged fraction on BOTH routes, and the largest parameter difference among
the fits both routes converged, in units of the quoted error.

usage: python tools/fuzz_lm_fd_vs_minpack.py [seconds] [seed] [classes]
       classes: comma list out of coellip1..coellip5,bdf,bd   (default all)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("NGMIX_FITTER_BATCHED", "0")
import test_gpu_lm_batch as tb  # noqa: E402
import ngmix_amd as ngmix  # noqa: E402
from ngmix_amd.batch import StampBatch  # noqa: E402
from ngmix_amd.lm_batch import LMBatchFitter  # noqa: E402

CLASSES = ["coellip1", "coellip2", "coellip3", "coellip4", "coellip5", "bdf", "bd"]

PSF_PARS = {"maxfev": 4000, "ftol": 1.0e-5, "xtol": 1.0e-5}   # what psf runners configure


def coellip_case(rng, ngauss):
    """a turbulent psf on a small stamp fitted by ngauss co-elliptical
    gaussians from a guess scattered around a sensible start (the workload of
    tools/lm_advance_share.py, randomised)"""
    n = int(rng.randint(3, 8))
    dim = int(rng.choice([21, 25, 33]))
    scale = 0.263
    cen = (dim - 1) / 2.0
    noise = 10.0 ** rng.uniform(-4.3, -3.0)
    images, jacs, guesses = [], [], []
    T = 0.3 * np.array([0.3, 0.7, 1.5, 3.0, 6.0])[:ngauss]
    F = np.array([0.25, 0.35, 0.25, 0.1, 0.05])[:ngauss]
    if ngauss == 1:
        T = np.array([0.3])
    F = F / F.sum()
    for i in range(n):
        jac = ngmix.DiagonalJacobian(row=cen + rng.uniform(-0.4, 0.4),
                                     col=cen + rng.uniform(-0.4, 0.4), scale=scale)
        gm = ngmix.GMixModel([0.0, 0.0, rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05),
                              rng.uniform(0.25, 0.4), 1.0], "turb")
        im = gm.make_image((dim, dim), jacobian=jac)
        images.append(im + noise * rng.normal(size=im.shape))
        jacs.append(jac)
        g0 = np.concatenate([[0.0, 0.0, 0.0, 0.0], T, F])
        g = g0 * rng.uniform(0.92, 1.08, size=g0.size)
        g[0:2] = rng.uniform(-0.02, 0.02, size=2)
        g[2:4] = rng.uniform(-0.03, 0.03, size=2)
        guesses.append(g)
    weights = [np.full((dim, dim), 1.0 / noise ** 2)] * n
    fit_pars = PSF_PARS if rng.uniform() < 0.5 else None
    obs = [ngmix.Observation(images[i], weight=weights[i], jacobian=jacs[i]) for i in range(n)]
    sb = StampBatch.from_observations(obs)
    res = LMBatchFitter("coellip", ngauss=ngauss, fit_pars=fit_pars).go(sb, np.array(guesses))
    ones = [ngmix.fitting.CoellipFitter(ngauss=ngauss, fit_pars=fit_pars, batched=False).go(
        obs=obs[i], guess=guesses[i]) for i in range(n)]
    return res, ones, 4 + 2 * ngauss


def galaxy_case(rng, model):
    nband = int(rng.randint(1, 3))
    nobj = int(rng.randint(2, 5))
    npsf = int(rng.randint(1, 4))
    dim = int(rng.choice([32, 40, 48]))
    sobj = np.repeat(np.arange(nobj), nband).astype(np.int32)
    sband = np.tile(np.arange(nband), nobj).astype(np.int32)
    ns = sobj.size
    psf_rows, psf = tb._multi_gauss_psf(ns, npsf, bool(rng.randint(2)), rng)
    if model == "bdf":
        extra = np.repeat(rng.uniform(0.1, 0.9, size=nobj), nband)[:, None]
    else:
        extra = np.stack([np.repeat(rng.uniform(-0.3, 0.3, size=nobj), nband),
                          np.repeat(rng.uniform(0.1, 0.9, size=nobj), nband)], axis=1)
    pars, guess, images, weights, jobj, sb = tb._objects_with_psf(
        ns, model, psf, rng, dim=dim, extra=extra, noise=float(rng.choice([0.003, 0.01, 0.03])))
    nshape = 6 if model == "bdf" else 7
    first = np.arange(nobj) * nband
    flux = guess[:, nshape].reshape(nobj, nband)
    g2 = np.concatenate([guess[first, :nshape], flux], axis=1)
    res = LMBatchFitter(model).go(sb, g2, psf=psf, stamp_obj=sobj, stamp_band=sband)
    ones = []
    for o in range(nobj):
        mb = ngmix.MultiBandObsList()
        for b in range(nband):
            ol = ngmix.ObsList()
            s = o * nband + b
            pobs = ngmix.Observation(np.zeros((5, 5)), jacobian=jobj,
                                     gmix=ngmix.GMix(pars=psf_rows[s]))
            ol.append(ngmix.Observation(images[s], weight=weights[s], jacobian=jobj, psf=pobs))
            mb.append(ol)
        ones.append(ngmix.fitting.Fitter(model=model, batched=False).go(obs=mb, guess=g2[o]))
    return res, ones, g2.shape[1]


def one_case(seed, classes=CLASSES):
    """the case of `seed`: (class, n parameters, result of the batch, the
    per-object MINPACK results)"""
    rng = np.random.RandomState(seed)
    cls = classes[int(rng.randint(len(classes)))]
    if cls.startswith("coellip"):
        res, ones, n = coellip_case(rng, int(cls[7:]))
    else:
        res, ones, n = galaxy_case(rng, cls)
    return cls, n, res, ones


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    master = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
    classes = sys.argv[3].split(",") if len(sys.argv) > 3 else CLASSES
    run(budget, master, classes)


def run(budget, master, classes):
    stats = {c: dict(n=0, flags=0, ier=0, conv_drv=0, conv_mp=0, both=0, worst=0.0, worst_seed=None,
                     hist=np.zeros(6, dtype=np.int64), maxd=0, odd=[]) for c in classes}
    EDGES = ["0", "<=1 jac", "<=2 jac", "<=4 jac", "<=8 jac", ">8 jac"]
    t0 = time.time()
    while time.time() - t0 < budget:
        seed = int(master.randint(1 << 30))
        cls, n, res, ones = one_case(seed, classes)
        st = stats[cls]
        for o, one in enumerate(ones):
            st["n"] += 1
            fl, ie, nf = int(res["flags"][o]), int(res["ier"][o]), int(res["nfev"][o])
            st["flags"] += int((one["flags"] == 0) == (fl == 0))
            st["ier"] += int(one["ier"] == ie)
            st["conv_drv"] += int(fl == 0)
            st["conv_mp"] += int(one["flags"] == 0)
            d = abs(nf - int(one["nfev"]))
            st["maxd"] = max(st["maxd"], d)
            jac = n + 1
            k = 0 if d == 0 else 1 if d <= jac else 2 if d <= 2 * jac else 3 if d <= 4 * jac \
                else 4 if d <= 8 * jac else 5
            st["hist"][k] += 1
            if (one["flags"] == 0) != (fl == 0) and len(st["odd"]) < 8:
                st["odd"].append((seed, o, int(one["flags"]), fl, int(one["ier"]), ie,
                                  int(one["nfev"]), nf))
            if one["flags"] == 0 and fl == 0:
                st["both"] += 1
                w = float(np.max(np.abs(res["pars"][o] - one["pars"]) / one["pars_err"]))
                if w > st["worst"]:
                    st["worst"], st["worst_seed"] = w, (seed, o)
    print("fuzz_lm_fd_vs_minpack: %.0f s" % (time.time() - t0))
    for c in classes:
        st = stats[c]
        if not st["n"]:
            continue
        print("%-9s %6d fits | converged: driver %.4f MINPACK %.4f | pass/fail agrees %d (%.4f), ier equal "
              "%d | |dnfev| %s max %d | both converged %d: worst |dpars|/err %.2e %s"
              % (c, st["n"], st["conv_drv"] / st["n"], st["conv_mp"] / st["n"], st["flags"],
                 st["flags"] / st["n"], st["ier"],
                 " ".join("%s:%d" % (e, h) for e, h in zip(EDGES, st["hist"])), st["maxd"],
                 st["both"], st["worst"], st["worst_seed"]))
        for rec in st["odd"]:
            print("      pass/fail differs: seed %d fit %d: flags %d / %d, ier %d / %d, nfev %d / %d "
                  "(MINPACK / driver)" % rec)


if __name__ == "__main__":
    main()
