#!/usr/bin/env python
"""
Golden vectors for the wrappers added after the first fixture set: GaussMom
(ngmix/gaussmom.py) and PSFFluxFitter (ngmix/fitting/fitters.py:144-181), by
running the REFERENCE ITSELF under the numba shim.  Build container only; the
fixture tests/golden/extra.npz is committed.  TEST INFRASTRUCTURE ONLY.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_extra.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "shim"), "/root/reference"]

import numpy as np  # noqa: E402
import ngmix  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "extra.npz")
SCALE = 0.263


def main():
    rng = np.random.RandomState(8321)
    out = {}
    dims = (33, 31)
    jac = ngmix.Jacobian(row=15.7, col=15.2, dvdrow=SCALE * 1.01, dvdcol=0.006,
                         dudrow=-0.004, dudcol=SCALE * 0.98)
    psf_gm = ngmix.GMixModel([0.0, 0.0, 0.01, -0.02, 0.27, 1.0], "turb")
    gm = ngmix.GMixModel([0.05, -0.03, 0.08, 0.04, 0.55, 130.0], "exp").convolve(psf_gm)
    image = gm.make_image(dims, jacobian=jac, fast_exp=True)
    image += 0.02 * rng.normal(size=dims)
    weight = np.full(dims, 1.0 / 0.02 ** 2)
    weight[4, 5] = 0.0
    weight[20, 7:10] = 0.0
    psf_im = psf_gm.make_image((25, 25), jacobian=ngmix.DiagonalJacobian(
        row=12.0, col=12.0, scale=SCALE), fast_exp=True)
    psf_obs = ngmix.Observation(psf_im, jacobian=ngmix.DiagonalJacobian(
        row=12.0, col=12.0, scale=SCALE), gmix=psf_gm)
    obs = ngmix.Observation(image, weight=weight, jacobian=jac, psf=psf_obs)
    out.update(image=image, weight=weight, jac=jac.get_data().copy(),
               psf_image=psf_im, psf_pars=psf_gm.get_full_pars(),
               psf_jac=psf_obs.jacobian.get_data().copy())

    for tag, hi in (("gm6", False), ("gm17", True)):
        res = ngmix.gaussmom.GaussMom(fwhm=1.2, with_higher_order=hi).go(obs)
        out[tag + "_flags"] = res["flags"]
        for k in ("flux", "flux_err", "T", "T_err", "s2n", "e1", "e2", "wsum",
                  "sums_norm"):
            out[tag + "_" + k] = res[k]
        for k in ("pars", "sums", "sums_cov", "e_err", "sums_err"):
            out[tag + "_" + k] = np.array(res[k])
        out[tag + "_npix"] = res["npix"]

    for tag, kw in (("pf", {}), ("pf_nonorm", {"normalize_psf": False})):
        res = ngmix.fitting.PSFFluxFitter(**kw).go(obs)
        for k in ("flags", "chi2per", "dof", "flux", "flux_err"):
            out[tag + "_" + k] = res[k]
    # template flux with the object's own gmix
    obs2 = ngmix.Observation(image, weight=weight, jacobian=jac, gmix=gm)
    res = ngmix.fitting.PSFFluxFitter(do_psf=False).go(obs2)
    out["tf_gmix_pars"] = gm.get_full_pars()
    for k in ("flags", "chi2per", "dof", "flux", "flux_err"):
        out["tf_" + k] = res[k]
    # ---- noise-power sandwich covariance (ngmix/fitting/noise_cov.py)
    from scipy.ndimage import uniform_filter
    ol = ngmix.ObsList()
    truth = [0.03, -0.02, 0.06, -0.04, 0.5, 90.0]
    for e in range(2):
        ejac = ngmix.DiagonalJacobian(row=15.0 + 0.3 * e, col=15.5 - 0.2 * e, scale=SCALE)
        egm = ngmix.GMixModel(truth, "exp").convolve(psf_gm)
        sigma = 0.03
        # stationary correlated noise: smoothed white noise, same for the
        # image's noise and the attached independent realisation
        nz = [uniform_filter(rng.normal(size=(32, 32)), size=3, mode="wrap") * sigma * 3
              for _ in range(2)]
        im = egm.make_image((32, 32), jacobian=ejac, fast_exp=True) + nz[0]
        wt = np.full((32, 32), 1.0 / sigma ** 2)
        eobs = ngmix.Observation(im, weight=wt, jacobian=ejac, psf=psf_obs, noise=nz[1])
        ol.append(eobs)
        out["nc_e%d_image" % e] = im
        out["nc_e%d_weight" % e] = wt
        out["nc_e%d_noise" % e] = nz[1]
        out["nc_e%d_jac" % e] = ejac.get_data().copy()
    guess = np.array(truth) * (1.0 + 0.05 * rng.uniform(-1, 1, size=6))
    out["nc_guess"] = guess
    for tag, uni in (("nc_plain", False), ("nc_sandwich", True)):
        res = ngmix.fitting.Fitter(model="exp", use_noise_image=uni).go(obs=ol, guess=guess)
        for k in ("flags", "nfev", "ier"):
            out[tag + "_" + k] = res[k]
        for k in ("pars", "pars_err", "pars_cov", "pars_cov0"):
            out[tag + "_" + k] = np.array(res[k])
    # the same sandwich with central-difference derivative images
    from ngmix.fitting import noise_cov as ncmod
    fm = ngmix.fitting.results.FitModel(obs=ol, model="exp", guess=guess)
    fd = ncmod._dmodel_images(fit_model=fm, pars=out["nc_sandwich_pars"], band=0,
                              obs=ol[0], kpars=list(range(6)), force_fd=True)
    out["nc_fd_images_e0"] = np.array(fd)
    np.savez_compressed(OUT, **out)
    print("wrote %s (%.1f kB)" % (OUT, os.path.getsize(OUT) / 1e3))


if __name__ == "__main__":
    main()
