#!/usr/bin/env python
"""
A random sweep for ngmix_amd/prepsfmom.py: 36 configurations drawn over both
kernels, kernel sizes, stamp sizes 21-52 (even and odd), psf stamps of other
sizes, padding factors, apodisation widths, smoothing, diagonal and sheared
jacobians, centre offsets, with / without a psf -- each measured by the
REFERENCE's PrePSFMom (under the numba shim).  Inputs are stored as float32
(both sides widen the same numbers), results as float64.
-> tests/golden/prepsf_sweep.npz.  Build container only.  TEST INFRASTRUCTURE ONLY.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_prepsf_sweep.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "shim"), "/root/reference"]

import numpy as np  # noqa: E402
import ngmix  # noqa: E402
from ngmix import prepsfmom  # noqa: E402
from gen_golden_prepsf import _WCS  # noqa: E402,F401  (installs get_galsim_wcs)

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "prepsf_sweep.npz")
KEYS = ("flags", "flux", "flux_err", "T", "T_err", "s2n", "e1", "e2", "e_err", "sums", "sums_cov")


def main():
    rng = np.random.RandomState(2024)
    out = {}
    n = 0
    while n < 36:
        kernel = str(rng.choice(["pgauss", "ksigma"]))
        dim = int(rng.randint(21, 53))
        pdim = int(rng.choice([dim, dim, int(rng.randint(17, 45))]))
        scale = rng.uniform(0.15, 0.3)
        if rng.uniform() < 0.5:
            jkw = dict(dvdrow=scale, dvdcol=0.0, dudrow=0.0, dudcol=scale)
        else:
            jkw = dict(dvdrow=scale * rng.uniform(0.95, 1.05), dvdcol=scale * rng.uniform(-0.08, 0.08),
                       dudrow=scale * rng.uniform(-0.08, 0.08), dudcol=scale * rng.uniform(0.95, 1.05))
        fwhm = rng.uniform(1.0, 1.8) if kernel == "pgauss" else rng.uniform(1.6, 2.6)
        kw = dict(kernel=kernel, fwhm=float(fwhm), pad_factor=float(rng.choice([4, 4, 3.5, 5, 4.25])),
                  ap_rad=float(rng.choice([1.5, 1.5, 0.0, 1.0, 2.2])),
                  fwhm_smooth=float(rng.choice([0.0, 0.0, 0.6, 1.0])))
        no_psf = bool(rng.uniform() < 0.2)
        cen = (dim - 1) / 2
        jac = ngmix.Jacobian(row=cen + rng.uniform(-0.6, 0.6), col=cen + rng.uniform(-0.6, 0.6), **jkw)
        pcen = (pdim - 1) / 2
        pjac = ngmix.Jacobian(row=pcen + rng.uniform(-0.4, 0.4), col=pcen + rng.uniform(-0.4, 0.4), **jkw)
        psf_gm = ngmix.GMixModel([0.0, 0.0, rng.uniform(-0.03, 0.03), rng.uniform(-0.03, 0.03),
                                  rng.uniform(0.2, 0.4), 1.0], "turb")
        gm = ngmix.GMixModel([0.0, 0.0, rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3),
                              rng.uniform(0.2, 0.9), rng.uniform(20, 200)], "exp").convolve(psf_gm)
        noise = rng.uniform(0.005, 0.05)
        im = (gm.make_image((dim, dim), jacobian=jac) + noise * rng.normal(size=(dim, dim))).astype("f4")
        pim = (psf_gm.make_image((pdim, pdim), jacobian=pjac)
               + 1e-4 * rng.normal(size=(pdim, pdim))).astype("f4")
        wt = np.full((dim, dim), 1.0 / noise ** 2, dtype="f4")
        obs = ngmix.Observation(im.astype("f8"), weight=wt.astype("f8"), jacobian=jac,
                                psf=ngmix.Observation(pim.astype("f8"), jacobian=pjac))
        try:
            res = prepsfmom.PrePSFMom(**kw).go(obs, no_psf=no_psf)
        except ngmix.gexceptions.FFTRangeError:
            continue
        tag = "c%02d" % n
        out[tag + "_im"], out[tag + "_pim"], out[tag + "_wt0"] = im, pim, np.array(wt[0, 0])
        out[tag + "_jac"], out[tag + "_pjac"] = jac.get_data(), pjac.get_data()
        out[tag + "_conf"] = np.array([{"pgauss": 0, "ksigma": 1}[kernel], kw["fwhm"], kw["pad_factor"],
                                       kw["ap_rad"], kw["fwhm_smooth"], float(no_psf)])
        for k in KEYS:
            out["%s_%s" % (tag, k)] = np.asarray(res[k])
        print(tag, kernel, dim, pdim, kw, no_psf, res["flags"], float(res["flux"]), float(res["T"]))
        n += 1
    np.savez_compressed(OUT, **out)
    print("wrote %s (%d arrays, %.1f kB)" % (OUT, len(out), os.path.getsize(OUT) / 1e3))


if __name__ == "__main__":
    main()
