"""What lm_advance costs the fits with 11-14 parameters (the generic
private-memory step): per fit, the HIP-event time of the advance
launches beside the pixel passes, for 'exp' over 6-9 bands (lmder, 11-14
parameters), 'bdf' over 5-7 bands (lmdif, 11-13) and co-elliptical psf fits
with 4 / 5 gaussians (lmdif, 12 / 14).
usage: python tools/lm_advance_share.py [nobj]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
import ngmix_amd as ngmix  # noqa: E402
from ngmix_amd.batch import GMixBatch, StampBatch  # noqa: E402
from ngmix_amd.lm_batch import LMBatchFitter  # noqa: E402

nobj = int(sys.argv[1]) if len(sys.argv) > 1 else 10000


def report(tag, f, res):
    k = f.kernel_ms
    tot = sum(k.values())
    print("%-22s rounds %3d  lm_advance %.3f ms/launch, %.2f of %.2f ms of kernels (%.0f %%); "
          "lm_eval %.2f ms; flags==0: %.3f; median nfev %d"
          % (tag, f.rounds_launched, k["lm_advance"] / max(f.rounds_launched, 1), k["lm_advance"],
             tot, 100.0 * k["lm_advance"] / tot, k["lm_eval"], float(np.mean(res["flags"] == 0)),
             int(np.median(res["nfev"]))))
    sys.stdout.flush()


def multiband(model, nband, analytic):
    ns = nobj * nband
    sb, _, pars = bench.make_workload(ns, 1000, "cuda")
    rng = np.random.RandomState(7)
    shape = pars[::nband, :5]
    flux = pars[:, 5].reshape(nobj, nband)
    if model == "bdf":
        shape = np.concatenate([shape, np.full((nobj, 1), 0.1)], axis=1)
    guess = np.concatenate([shape, flux], axis=1)
    guess = guess * rng.uniform(0.97, 1.03, size=guess.shape)
    psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (ns, 1)), "gauss")
    sobj = np.repeat(np.arange(nobj), nband)
    sband = np.tile(np.arange(nband), nobj)
    f = LMBatchFitter(model, analytic_jacobian=analytic,
                      fit_pars={"maxfev": 200, "ftol": 1e-5, "xtol": 1e-5})
    f.time_kernels = True
    for _ in range(2):
        res = f.go(sb, guess, psf=psf, stamp_obj=sobj, stamp_band=sband)
    torch.cuda.synchronize()
    report("%s x %d bands (n=%d)" % (model, nband, guess.shape[1]), f, res)


def coellip(ngauss):
    # a turbulent psf on 25x25 stamps, fitted by ngauss co-elliptical gaussians
    rng = np.random.RandomState(5)
    dim, scale = 25, 0.263
    jac = ngmix.DiagonalJacobian(row=12.0, col=12.0, scale=scale)
    gm = ngmix.GMixModel([0.0, 0.0, 0.02, -0.01, 0.3, 1.0], "turb")
    im0 = gm.make_image((dim, dim), jacobian=jac)
    n = nobj
    images = im0[None] + 2.0e-4 * rng.normal(size=(n, dim, dim))
    weights = np.full((n, dim, dim), 1.0 / 2.0e-4 ** 2)
    sb = StampBatch.from_images(images, weights, jac)
    T = 0.3 * np.array([0.3, 0.7, 1.5, 3.0, 6.0])[:ngauss]
    F = np.array([0.25, 0.35, 0.25, 0.1, 0.05])[:ngauss]
    F = F / F.sum()
    g0 = np.concatenate([[0.0, 0.0, 0.02, -0.01], T, F])
    guess = g0[None] * rng.uniform(0.95, 1.05, size=(n, g0.size))
    guess[:, :2] = rng.uniform(-0.01, 0.01, size=(n, 2))
    f = LMBatchFitter("coellip", ngauss=ngauss,
                      fit_pars={"maxfev": 300, "ftol": 1e-5, "xtol": 1e-5})
    f.time_kernels = True
    for _ in range(2):
        res = f.go(sb, guess)
    torch.cuda.synchronize()
    report("coellip-%d (n=%d)" % (ngauss, g0.size), f, res)


print("generic=%s nobj=%d" % (os.environ.get("NGMIX_LM_GENERIC", "0"), nobj))
for nband in (5, 6, 8, 9):
    multiband("exp", nband, True)
for nband in (3, 5, 7):
    multiband("bdf", nband, False)
for ng in (3, 4, 5):
    coellip(ng)
