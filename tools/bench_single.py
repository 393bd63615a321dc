"""the per-object reference API on the GPU (one Observation at a time through
the seam): microseconds per call.  python tools/bench_single.py"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngmix_amd as ngmix  # noqa: E402

rng = np.random.RandomState(1)
jac = ngmix.DiagonalJacobian(row=23.5, col=23.5, scale=0.263)
pgm = ngmix.GMixModel([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], "gauss")
gm = ngmix.GMixModel([0.1, -0.05, 0.1, 0.05, 0.6, 100.0], "exp").convolve(pgm)
im = gm.make_image((48, 48), jacobian=jac, fast_exp=True) + 0.01 * rng.normal(size=(48, 48))
wt = np.full((48, 48), 1e4)
pobs = ngmix.Observation(pgm.make_image((25, 25), jacobian=ngmix.DiagonalJacobian(row=12, col=12, scale=0.263)),
                         jacobian=ngmix.DiagonalJacobian(row=12, col=12, scale=0.263), gmix=pgm)
obs = ngmix.Observation(im, weight=wt, jacobian=jac, psf=pobs)


def t(name, fn, n=200):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    print("%-44s %8.1f us" % (name, (time.perf_counter() - t0) / n * 1e6))


t("Observation(image, weight, jacobian)", lambda: ngmix.Observation(im, weight=wt, jacobian=jac))
t("GMixModel(pars, 'exp').convolve(psf)", lambda: ngmix.GMixModel([0.1, -0.05, 0.1, 0.05, 0.6, 100.0], "exp").convolve(pgm))
t("gm.get_loglike(obs)", lambda: gm.get_loglike(obs))
t("gm.make_image((48, 48), fast_exp=True)", lambda: gm.make_image((48, 48), jacobian=jac, fast_exp=True))
t("gm.fill_fdiff(obs, fdiff)", lambda: gm.fill_fdiff(obs, np.zeros(2304)))
t("GaussMom(1.2).go(obs)", lambda: ngmix.GaussMom(fwhm=1.2).go(obs))
t("run_admom(obs, 0.6)", lambda: ngmix.admom.run_admom(obs, 0.6, rng=rng), n=100)
t("run_em(psf obs, 1 gaussian)", lambda: ngmix.em.run_em(pobs, ngmix.GMixModel([0.0, 0.0, 0.0, 0.0, 0.3, 1.0], "gauss")), n=50)
guess = np.array([0.1, -0.05, 0.1, 0.05, 0.6, 100.0]) * 1.03
fb = ngmix.fitting.Fitter(model="exp", batched=True)
t("Fitter('exp', batched=True).go(obs, guess)", lambda: fb.go(obs=obs, guess=guess), n=200)


def first_fit():
    # an Observation the device has not seen: its pixels are uploaded by the fit
    o = ngmix.Observation(im, weight=wt, jacobian=jac, psf=pobs)
    fb.go(obs=o, guess=guess)


t("  (first fit of a new Observation)", first_fit, n=200)
t("  (a new Fitter per call)", lambda: ngmix.fitting.Fitter(model="exp", batched=True).go(obs=obs, guess=guess), n=100)
t("Fitter('exp', batched=False).go(obs, guess)", lambda: ngmix.fitting.Fitter(model="exp", batched=False).go(obs=obs, guess=guess), n=50)
