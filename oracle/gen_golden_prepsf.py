#!/usr/bin/env python
"""
Golden vectors for ngmix_amd/prepsfmom.py from the REFERENCE's
ngmix.prepsfmom (PGaussMom, KSigmaMom, PrePSFMom) under the numba shim:
ngmix-rendered 'exp' (x) psf stamps with noise, the moments of every kernel
with and without a psf, smoothing, apodisation off, a noise image, a sheared
jacobian, even and odd stamps, a psf stamp smaller than the image, a padding
factor that is not an integer; the k-space kernels of one case; the error
cases.  -> tests/golden/prepsf.npz.  Build container only.  TEST
INFRASTRUCTURE ONLY.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_prepsf.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "shim"), "/root/reference"]

import numpy as np  # noqa: E402
import ngmix  # noqa: E402
from ngmix import prepsfmom  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "prepsf.npz")

RESULT_KEYS = ("flags", "flux", "flux_err", "flux_flags", "T", "T_err", "T_flags", "s2n", "e1", "e2",
               "e", "e_err", "e_cov", "sums", "sums_cov", "sums_norm", "pars", "npix", "wsum")


class _WCS(object):
    """what get_galsim_wcs must return for two jacobians to compare: the
    reference only tests == between the psf's and the image's"""

    def __init__(self, j):
        self.t = (j.dudcol, j.dudrow, j.dvdcol, j.dvdrow)

    def __eq__(self, other):
        return self.t == other.t

    def __ne__(self, other):
        return self.t != other.t


ngmix.Jacobian.get_galsim_wcs = lambda self: _WCS(self)


def scene(rng, dim, pdim, jac_kw, noise=0.02, psf_T=0.3):
    cen = (dim - 1) / 2
    jac = ngmix.Jacobian(row=cen + rng.uniform(-0.3, 0.3), col=cen + rng.uniform(-0.3, 0.3), **jac_kw)
    pcen = (pdim - 1) / 2
    pjac = ngmix.Jacobian(row=pcen + rng.uniform(-0.2, 0.2), col=pcen + rng.uniform(-0.2, 0.2), **jac_kw)
    psf_gm = ngmix.GMixModel([0.0, 0.0, 0.02, -0.01, psf_T, 1.0], "turb")
    gm = ngmix.GMixModel([0.0, 0.0, 0.15, -0.1, 0.5, 50.0], "exp").convolve(psf_gm)
    im = gm.make_image((dim, dim), jacobian=jac) + noise * rng.normal(size=(dim, dim))
    pim = psf_gm.make_image((pdim, pdim), jacobian=pjac) + 1e-4 * rng.normal(size=(pdim, pdim))
    wt = np.full((dim, dim), 1.0 / noise ** 2)
    wt[3, 5] = 0.0
    nim = noise * rng.normal(size=(dim, dim))
    return im, wt, jac, pim, pjac, nim


def main():
    out = {}
    rng = np.random.RandomState(1701)
    diag = dict(dvdrow=0.2, dvdcol=0.0, dudrow=0.0, dudcol=0.2)
    shear = dict(dvdrow=0.21, dvdcol=0.012, dudrow=-0.009, dudcol=0.19)
    cases = [
        ("pgauss", dict(kernel="pgauss", fwhm=1.2), 33, 33, diag, {}),
        ("ksigma", dict(kernel="ksigma", fwhm=2.0), 33, 33, diag, {}),
        ("gauss_alias", dict(kernel="gauss", fwhm=1.2), 32, 25, diag, {}),
        ("pgauss_smooth", dict(kernel="pgauss", fwhm=1.2, fwhm_smooth=0.8), 33, 33, diag, {}),
        ("ksigma_smooth", dict(kernel="ksigma", fwhm=2.0, fwhm_smooth=0.8), 32, 32, diag, {}),
        ("pgauss_noap", dict(kernel="pgauss", fwhm=1.2, ap_rad=0), 33, 33, diag, {}),
        ("pgauss_pad", dict(kernel="pgauss", fwhm=1.2, pad_factor=3.5), 33, 41, diag, {}),
        ("pgauss_shear", dict(kernel="pgauss", fwhm=1.2), 35, 35, shear, {}),
        ("ksigma_shear", dict(kernel="ksigma", fwhm=2.2), 35, 27, shear, {}),
        ("pgauss_nopsf", dict(kernel="pgauss", fwhm=1.2), 33, 33, diag, {"no_psf": True}),
        ("pgauss_noiseim", dict(kernel="pgauss", fwhm=1.2, use_noise_image=True), 33, 33, diag, {}),
    ]
    for tag, kw, dim, pdim, jkw, gokw in cases:
        im, wt, jac, pim, pjac, nim = scene(rng, dim, pdim, jkw)
        psf = ngmix.Observation(pim, jacobian=pjac)
        obs = ngmix.Observation(im, weight=wt, jacobian=jac, psf=psf, noise=nim)
        fitter = prepsfmom.PrePSFMom(**kw)
        res = fitter.go(obs, **gokw)
        for k, v in dict(im=im, wt=wt, jac=jac.get_data(), pim=pim, pjac=pjac.get_data(), nim=nim).items():
            out["%s_in_%s" % (tag, k)] = np.asarray(v)
        for k in RESULT_KEYS:
            if k in res:
                out["%s_%s" % (tag, k)] = np.asarray(res[k])
        out[tag + "_keys"] = np.array(sorted(res.keys()))
        print(tag, res["flags"], res["flux"], res["T"], res["e1"], res["e2"], res["s2n"])
    # the unpacked kernels of one case
    im, wt, jac, pim, pjac, nim = scene(rng, 33, 33, diag)
    obs = ngmix.Observation(im, weight=wt, jacobian=jac, psf=ngmix.Observation(pim, jacobian=pjac))
    for name, cls, fwhm in (("kern_pgauss", prepsfmom.PGaussMom, 1.2), ("kern_ksigma", prepsfmom.KSigmaMom, 2.0)):
        res = cls(fwhm).go(obs, return_kernels=True)
        for k, v in res["kernels"].items():
            out["%s_%s" % (name, k)] = np.asarray(v)
    out["kern_in_im"], out["kern_in_wt"], out["kern_in_pim"] = im, wt, pim
    out["kern_in_jac"], out["kern_in_pjac"] = jac.get_data(), pjac.get_data()
    # error cases: kernel too big for the stamp, not square, no psf, a different wcs, unknown kernel
    errs = {}
    for name, f in (
        ("too_big", lambda: prepsfmom.PGaussMom(30.0).go(obs)),
        ("not_square", lambda: prepsfmom.PGaussMom(1.2).go(ngmix.Observation(np.zeros((10, 12))), no_psf=True)),
        ("no_psf_set", lambda: prepsfmom.PGaussMom(1.2).go(ngmix.Observation(im, weight=wt, jacobian=jac))),
        ("wcs_differs", lambda: prepsfmom.PGaussMom(1.2).go(ngmix.Observation(
            im, weight=wt, jacobian=jac,
            psf=ngmix.Observation(pim, jacobian=ngmix.DiagonalJacobian(row=16, col=16, scale=0.3))))),
        ("bad_kernel", lambda: prepsfmom.PrePSFMom(1.2, "blah")),
        ("not_obs", lambda: prepsfmom.PGaussMom(1.2).go(3)),
        ("noise_missing", lambda: prepsfmom.PGaussMom(1.2, use_noise_image=True).go(obs)),
    ):
        try:
            f()
            errs[name] = "None"
        except Exception as e:      # noqa: BLE001
            errs[name] = type(e).__name__
    out["error_names"] = np.array(sorted(errs))
    out["error_types"] = np.array([errs[k] for k in sorted(errs)])
    print(errs)
    np.savez_compressed(OUT, **out)
    print("wrote %s (%d arrays, %.1f kB)" % (OUT, len(out), os.path.getsize(OUT) / 1e3))


if __name__ == "__main__":
    main()
