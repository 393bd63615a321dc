#!/usr/bin/env python
"""
Golden vectors for the small host helpers added in round 5 -- moments.
regularize_mom_shapes, shape.e1e2_to_eta1eta2, shape.dgs_by_dgo_jacob -- by
running the REFERENCE ITSELF under the numba shim (ngmix/moments.py:578-640,
ngmix/shape.py:350-393,443-468).  Build container only; the fixture
tests/golden/host5.npz is committed.  TEST INFRASTRUCTURE ONLY.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_r5.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "shim"), "/root/reference"]

import numpy as np  # noqa: E402
import ngmix  # noqa: E402
from ngmix import moments, shape  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "host5.npz")
NUMERIC = ("flags", "flux", "flux_err", "flux_flags", "T", "T_err", "T_flags", "s2n",
           "e1", "e2", "e", "e_err", "e_cov", "sums", "sums_cov", "pars")


def main():
    rng = np.random.RandomState(515)
    out = {}
    # moment sums of a plausible object, and degenerate ones: NaN centroid sums
    # (pre-psf fitters), a negative flux sum, a negative T sum
    cases = []
    for k in range(4):
        A = rng.normal(size=(6, 6))
        cov = A @ A.T * 1.0e-3
        sums = np.array([0.01, -0.02, 0.3, -0.2, 1.5, 2.0]) * rng.uniform(0.8, 1.2, size=6)
        if k == 1:
            sums[:2] = np.nan
        if k == 2:
            sums[5] = -0.5
        if k == 3:
            sums[4] = -0.2
        cases.append((sums, cov))
    out["ncase"] = np.array(len(cases))
    out["fwhm_reg"] = np.array([0.0, 0.6, 1.2])
    for k, (sums, cov) in enumerate(cases):
        out["reg%d_sums" % k] = sums
        out["reg%d_cov" % k] = cov
        res = moments.make_mom_result(sums.copy(), cov.copy())
        for f, fwhm in enumerate(out["fwhm_reg"]):
            reg = moments.regularize_mom_shapes(dict(res), float(fwhm))
            for key in NUMERIC:
                if key in reg:
                    out["reg%d_f%d_%s" % (k, f, key)] = np.asarray(reg[key], dtype="f8")
            out["reg%d_f%d_flagstr" % (k, f)] = np.array(reg["flagstr"])
    # shapes
    e = rng.uniform(-0.65, 0.65, size=(200, 2))
    e[0] = 0.0
    out["e"] = e
    eta1, eta2 = shape.e1e2_to_eta1eta2(e[:, 0].copy(), e[:, 1].copy())
    out["eta"] = np.stack([eta1, eta2], axis=1)
    s1, s2 = shape.e1e2_to_eta1eta2(0.3, -0.4)
    out["eta_scalar"] = np.array([s1, s2])
    g = rng.uniform(-0.6, 0.6, size=(200, 2))
    s = rng.uniform(-0.1, 0.1, size=(200, 2))
    out["g"], out["s"] = g, s
    out["jacob"] = shape.dgs_by_dgo_jacob(g[:, 0], g[:, 1], s[:, 0], s[:, 1])
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(out), "arrays")


if __name__ == "__main__":
    main()
