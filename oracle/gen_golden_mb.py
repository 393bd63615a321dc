#!/usr/bin/env python
"""
Golden vectors for the lock-step LM driver on MULTI-BAND objects, by running
the REFERENCE ITSELF (ngmix.fitting.Fitter on MultiBandObsLists: MINPACK lmder,
analytic jacobian, DEFAULT_LM_PARS) under the numba shim: objects with 1-7
bands (6-12 parameters) and one or two epochs per band, 'exp' (x) a gaussian or
a three-gaussian psf, 32x32 stamps off the pixel grid.  The direct link between
the reference's fits and the driver's for the parameter counts its team form
of the lmder step serves (9 and up).  Build container only; tests/golden/
lm_mb.npz is committed.  TEST INFRASTRUCTURE ONLY.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_mb.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "shim"), "/root/reference"]

import numpy as np  # noqa: E402
import ngmix  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "lm_mb.npz")
DIM = 32
SCALE = 0.263
NBANDS = (1, 2, 3, 4, 5, 6, 7)
PER = 3          # objects per band count


def main():
    rng = np.random.RandomState(515151)
    out = {"nbands": np.array(NBANDS), "per": np.array(PER)}
    psf_choices = [ngmix.GMixModel([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], "gauss"),
                   ngmix.GMixModel([0.0, 0.0, 0.02, -0.01, 0.30, 1.0], "turb")]
    for nb in NBANDS:
        for k in range(PER):
            tag = "b%d_o%d_" % (nb, k)
            truth = np.concatenate([[rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1),
                                     rng.uniform(-0.25, 0.25), rng.uniform(-0.25, 0.25),
                                     rng.uniform(0.25, 0.8)], rng.uniform(50.0, 400.0, size=nb)])
            nep = rng.randint(1, 3, size=nb)
            ipsf = int(rng.randint(2))
            psf_gm = psf_choices[ipsf]
            mb = ngmix.MultiBandObsList()
            images, sigmas, jacs, bands = [], [], [], []
            for b in range(nb):
                ol = ngmix.ObsList()
                for e in range(nep[b]):
                    jac = ngmix.DiagonalJacobian(row=15.5 + rng.uniform(-0.5, 0.5),
                                                 col=15.5 + rng.uniform(-0.5, 0.5), scale=SCALE)
                    pars_b = np.concatenate([truth[:5], [truth[5 + b]]])
                    gm = ngmix.GMixModel(pars_b, "exp").convolve(psf_gm)
                    im = gm.make_image((DIM, DIM), jacobian=jac, fast_exp=True)
                    sigma = truth[5 + b] / rng.uniform(30.0, 600.0)
                    im = im + sigma * rng.normal(size=im.shape)
                    pobs = ngmix.Observation(np.zeros((5, 5)), jacobian=jac, gmix=psf_gm.copy())
                    ol.append(ngmix.Observation(im, weight=np.full(im.shape, 1.0 / sigma ** 2),
                                                jacobian=jac, psf=pobs))
                    images.append(im)
                    sigmas.append(sigma)
                    jacs.append(jac.get_data().copy())
                    bands.append(b)
                mb.append(ol)
            guess = truth * rng.uniform(0.8, 1.25, size=truth.size)
            guess[4] = truth[4] * rng.uniform(0.6, 1.6)
            guess[0:2] = truth[0:2] + rng.uniform(-0.05, 0.05, size=2)
            guess[2:4] = truth[2:4] + rng.uniform(-0.1, 0.1, size=2)
            # the last object of each band count at tolerances far below the
            # default's 1e-5: more rounds of the step before MINPACK stops
            fit_pars = {"ftol": 1.0e-10, "xtol": 1.0e-10, "maxfev": 4000} if k == PER - 1 else None
            res = ngmix.fitting.Fitter(model="exp", fit_pars=fit_pars).go(obs=mb, guess=guess)
            out[tag + "tol"] = np.array(1.0e-10 if fit_pars else 1.0e-5)
            out[tag + "images"] = np.array(images)
            out[tag + "sigma"] = np.array(sigmas)
            out[tag + "jac"] = np.concatenate(jacs)
            out[tag + "band"] = np.array(bands)
            out[tag + "psf_pars"] = psf_gm.get_full_pars()
            out[tag + "truth"], out[tag + "guess"] = truth, guess
            for key in ("flags", "nfev", "ier", "lnprob", "chi2per", "dof", "s2n", "npix"):
                out[tag + key] = np.array(res[key] if key in res else -9999)
            for key in ("pars", "pars_err", "pars_cov"):
                out[tag + key] = np.array(res[key])
            print(tag, res["flags"], res["nfev"], res["ier"], len(images), "stamps")
            sys.stdout.flush()
    np.savez_compressed(OUT, **out)
    print("wrote %s (%.1f kB)" % (OUT, os.path.getsize(OUT) / 1e3))


if __name__ == "__main__":
    main()
