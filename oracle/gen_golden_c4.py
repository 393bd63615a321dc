#!/usr/bin/env python
"""
Golden vectors for config 4 (adaptive moments + em_run on 32x32 stamps) from the
REFERENCE ITSELF under the numba shim: thirty-two objects of bench.py's C4 shape
(gaussian (x) gaussian psf T = 0.27, fluxes 50-200, noise 0.01, centres off the
pixel grid, the admom guess T = T_true U(0.9, 1.1), the one-gaussian EM guess
and sky = 0.05 of make_c4) run through ngmix.admom.admom_nb.admom with
AdmomFitter's default configuration and through ngmix.em.em_nb.em_run with the
configuration StampBatch.em passes (tol 1e-5, miniter 40, maxiter 500): the
direct link between the reference and ONE batch of the kernels at config 4's
shape, as tests/golden/lm_c3.npz is for config 3.  Build container only;
tests/golden/c4.npz is committed.  TEST INFRASTRUCTURE ONLY.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_c4.py
"""
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "shim"), "/root/reference"]

import numpy as np  # noqa: E402
import ngmix  # noqa: E402
from ngmix.admom import admom as admom_mod  # noqa: E402
from ngmix.admom.admom_nb import admom as admom_nb  # noqa: E402
from ngmix.em import em as em_mod  # noqa: E402
from ngmix.em import em_nb  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "c4.npz")
N, DIM, SCALE, SKY, NOISE, TPSF = 32, 32, 0.263, 0.05, 0.01, 0.27
FIELDS = ("p", "row", "col", "irr", "irc", "icc")


def six(gm):
    d = gm.get_data() if hasattr(gm, "get_data") else gm
    return np.array([[g[f] for f in FIELDS] for g in d])


def gm_in2_row(row, rng):
    """the first EM guess moved: centre by up to 0.4 pixel, sizes x U(0.5, 2)"""
    p, r, c, irr, irc, icc = row
    f = rng.uniform(0.5, 2.0)
    return [p * rng.uniform(0.7, 1.4), r + rng.uniform(-0.4, 0.4) * SCALE,
            c + rng.uniform(-0.4, 0.4) * SCALE, irr * f, irc * f, icc * f]


def main():
    rng = np.random.RandomState(404)
    images = np.zeros((N, DIM, DIM))
    jacs = np.zeros((N, 8))
    wt_in, gm_in, gm_in2 = np.zeros((N, 1, 6)), np.zeros((N, 1, 6)), np.zeros((N, 1, 6))
    am = {k: [] for k in ("flags", "numiter", "npix", "wsum", "sums", "sums_cov", "pars", "wt_out")}
    em = {k: [] for k in ("numiter", "frac_diff", "sky", "gmix_out")}
    em2 = {k: [] for k in ("numiter", "frac_diff", "sky", "gmix_out")}
    psf = ngmix.GMixModel([0.0, 0.0, 0.0, 0.0, TPSF, 1.0], "gauss")
    fitter = admom_mod.AdmomFitter(maxiter=200, shiftmax=5.0, etol=1.0e-5, Ttol=1.0e-3)
    efit = em_mod.EMFitter(tol=1.0e-5, miniter=40, maxiter=500)
    # (and with the stopping rule deciding, not miniter: a poorer guess, tol 1e-6
    # from the fifth iteration on)
    efit2 = em_mod.EMFitter(tol=1.0e-6, miniter=5, maxiter=500)
    for i in range(N):
        pars = [rng.uniform(-0.5, 0.5) * SCALE, rng.uniform(-0.5, 0.5) * SCALE,
                rng.normal(scale=0.05), rng.normal(scale=0.05), rng.uniform(0.3, 0.9),
                rng.uniform(50, 200)]
        jac = ngmix.DiagonalJacobian(row=(DIM - 1) / 2 + rng.uniform(-0.5, 0.5),
                                     col=(DIM - 1) / 2 + rng.uniform(-0.5, 0.5), scale=SCALE)
        im = ngmix.GMixModel(pars, "gauss").convolve(psf).make_image(
            (DIM, DIM), jacobian=jac, fast_exp=True)
        im = im + NOISE * rng.normal(size=im.shape)
        images[i] = im
        jacs[i] = np.array(jac.get_data().tolist()[0])
        weight = np.full(im.shape, 1.0 / NOISE ** 2)
        # ---- adaptive moments (admom.py:325-403 hands these to admom_nb.admom)
        obs = ngmix.Observation(im, weight=weight, jacobian=jac)
        guess = ngmix.GMixModel([0.0, 0.0, 0.0, 0.0, (pars[4] + TPSF) * rng.uniform(0.9, 1.1), 1.0],
                                "gauss")
        wt_in[i] = six(guess)
        ares = fitter._get_am_result()
        wt = guess._data
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            admom_nb(fitter.conf, wt, obs.pixels, ares)
        for k in ("flags", "numiter", "npix", "wsum", "sums", "sums_cov", "pars"):
            am[k].append(np.array(ares[k][0]))
        am["wt_out"].append(six(wt))
        # ---- em_run on image + sky (em.py:228-317 hands these to em_nb.em_run)
        obs_e = ngmix.Observation(im + SKY, weight=weight, jacobian=jac)
        eg = ngmix.GMixModel([pars[0], pars[1], pars[2], pars[3], pars[4] * rng.uniform(0.9, 1.1),
                              pars[5] * SCALE ** 2 * rng.uniform(0.9, 1.1)], "gauss")
        gm_in[i] = six(eg)
        gmc = eg.convolve(psf)
        conf = efit._make_conf(obs_e)
        conf["sky"] = SKY
        sums = efit._make_sums(len(eg))
        pixels = obs_e.pixels.copy()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            numiter, fdiff, skyout = em_nb.em_run(conf, pixels, sums, eg.get_data(),
                                                  psf.get_data(), gmc.get_data())
        em["numiter"].append(numiter)
        em["frac_diff"].append(fdiff)
        em["sky"].append(skyout)
        em["gmix_out"].append(six(eg))
        eg2 = ngmix.GMix(pars=gm_in2_row(gm_in[i, 0], rng))
        gm_in2[i] = six(eg2)
        gmc2 = eg2.convolve(psf)
        conf2 = efit2._make_conf(obs_e)
        conf2["sky"] = SKY
        pixels = obs_e.pixels.copy()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            numiter2, fdiff2, skyout2 = em_nb.em_run(conf2, pixels, efit2._make_sums(1),
                                                     eg2.get_data(), psf.get_data(),
                                                     gmc2.get_data())
        em2["numiter"].append(numiter2)
        em2["frac_diff"].append(fdiff2)
        em2["sky"].append(skyout2)
        em2["gmix_out"].append(six(eg2))
        print("object %2d: admom flags %d numiter %d; em numiter %d frac_diff %.3g; second em "
              "numiter %d frac_diff %.3g" % (i, ares["flags"][0], ares["numiter"][0], numiter,
                                             fdiff, numiter2, fdiff2))
        sys.stdout.flush()
    out = dict(images=images, jac=jacs, noise=np.array(NOISE), sky=np.array(SKY),
               psf=six(psf), admom_wt_in=wt_in, em_gmix_in=gm_in,
               admom_conf=np.array([200, 5.0, 1.0e-5, 1.0e-3]),
               em_conf=np.array([1.0e-5, 40, 500]), em2_gmix_in=gm_in2,
               em2_conf=np.array([1.0e-6, 5, 500]))
    for k, v in em2.items():
        out["em2_" + k] = np.array(v)
    for k, v in am.items():
        out["admom_" + k] = np.array(v)
    for k, v in em.items():
        out["em_" + k] = np.array(v)
    np.savez_compressed(OUT, **out)
    print("wrote %s (%.1f kB)" % (OUT, os.path.getsize(OUT) / 1e3))


if __name__ == "__main__":
    main()
