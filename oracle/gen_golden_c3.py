#!/usr/bin/env python
"""
Golden vectors for the batched lock-step LM driver on config-3-shaped stamps,
by running the REFERENCE ITSELF (ngmix.fitting.Fitter, MINPACK lmder with the
analytic jacobian, DEFAULT_LM_PARS) under the numba shim on NFIT independent
48x48 'exp' (x) gaussian-psf stamps: the direct link between the driver's
results and the reference's, object by object (round-3 review: the full-size
test goes through this package's own per-object Fitter).  Build container
only; tests/golden/lm_c3.npz is committed.  TEST INFRASTRUCTURE ONLY.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_c3.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "shim"), "/root/reference"]

import numpy as np  # noqa: E402
import ngmix  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "lm_c3.npz")
NFIT = 40
DIM = 48
SCALE = 0.263


def main():
    rng = np.random.RandomState(20240)
    psf_pars = np.array([0.0, 0.0, 0.0, 0.0, 0.27, 1.0])
    psf_gm = ngmix.GMixModel(psf_pars, "gauss")
    images = np.zeros((NFIT, DIM, DIM))
    sigmas = np.zeros(NFIT)
    jacs = []
    truths = np.zeros((NFIT, 6))
    guesses = np.zeros((NFIT, 6))
    keys_s = ("flags", "nfev", "ier", "lnprob", "chi2per", "dof", "s2n", "npix")
    keys_a = ("pars", "pars_err", "pars_cov", "pars_cov0")
    res_s = {k: [] for k in keys_s}
    res_a = {k: [] for k in keys_a}
    for i in range(NFIT):
        truth = np.array([rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1),
                          rng.uniform(-0.25, 0.25), rng.uniform(-0.25, 0.25),
                          rng.uniform(0.25, 0.9), rng.uniform(50.0, 400.0)])
        # stamp centres off the pixel grid, two of the jacobians sheared
        if i % 5 == 4:
            jac = ngmix.Jacobian(row=23.5 + rng.uniform(-0.5, 0.5), col=23.5 + rng.uniform(-0.5, 0.5),
                                 dvdrow=SCALE * 0.99, dvdcol=0.004, dudrow=-0.003, dudcol=SCALE)
        else:
            jac = ngmix.DiagonalJacobian(row=23.5 + rng.uniform(-0.5, 0.5),
                                         col=23.5 + rng.uniform(-0.5, 0.5), scale=SCALE)
        gm = ngmix.GMixModel(truth, "exp").convolve(psf_gm)
        im = gm.make_image((DIM, DIM), jacobian=jac, fast_exp=True)
        sigma = truth[5] / rng.uniform(600.0, 3000.0)
        im = im + sigma * rng.normal(size=im.shape)
        wt = np.full(im.shape, 1.0 / sigma ** 2)
        pobs = ngmix.Observation(np.zeros((5, 5)), jacobian=jac, gmix=psf_gm.copy())
        obs = ngmix.Observation(im, weight=wt, jacobian=jac, psf=pobs)
        # config 3's guess: truth x U(0.9, 1.1), centres and shapes shifted
        guess = truth * rng.uniform(0.9, 1.1, size=6)
        guess[0:2] = truth[0:2] + rng.uniform(-0.05, 0.05, size=2)
        guess[2:4] = truth[2:4] + rng.uniform(-0.03, 0.03, size=2)
        res = ngmix.fitting.Fitter(model="exp").go(obs=obs, guess=guess)
        images[i], sigmas[i] = im, sigma
        jacs.append(jac.get_data().copy())
        truths[i], guesses[i] = truth, guess
        for k in keys_s:
            res_s[k].append(res[k] if k in res else -9999)
        for k in keys_a:
            res_a[k].append(np.array(res[k]))
        print(i, res["flags"], res["nfev"], res["ier"], res["pars"])
    out = {"psf_pars": psf_gm.get_full_pars(), "images": images, "sigma": sigmas,
           "jac": np.concatenate(jacs), "truth": truths, "guess": guesses}
    for k in keys_s:
        out[k] = np.array(res_s[k])
    for k in keys_a:
        out[k] = np.array(res_a[k])
    np.savez_compressed(OUT, **out)
    print("wrote %s (%.1f kB)" % (OUT, os.path.getsize(OUT) / 1e3))


if __name__ == "__main__":
    main()
