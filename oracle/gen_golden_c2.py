#!/usr/bin/env python
"""
Golden vectors for config 2 (render + loglike of 48x48 stamps, 6 gaussians) from
the REFERENCE ITSELF under the numba shim: eight stamps of config 2's shape
(tests/helpers/c2_inputs.py: numpy-only inputs, rebuilt by the tests), the
'exp' model (x) a gaussian psf through GMix.get_loglike(obs, more=True),
GMix.fill_fdiff and GMix._fill_image (the accumulate-into render, fast exp) at
the generating parameters and at a moved set.  Build container only;
tests/golden/c2.npz (outputs only) is committed.  TEST INFRASTRUCTURE ONLY.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_c2.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "shim"), "/root/reference", os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402
import ngmix  # noqa: E402
from helpers import c2_inputs as c2  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "c2.npz")


def main():
    pars, moved, jac, images, sigma, base = c2.stamps()
    psf = ngmix.GMixModel([0.0, 0.0, 0.0, 0.0, c2.TPSF, 1.0], "gauss")
    out = {"image_sums": images.sum(axis=(1, 2)), "base_sums": base.sum(axis=(1, 2))}
    for tag, pp in (("truth", pars), ("moved", moved)):
        ll = np.zeros((c2.N, 4))
        fdiff = np.zeros((c2.N, c2.DIM * c2.DIM))
        rendered = np.zeros((c2.N, c2.DIM, c2.DIM))
        for i in range(c2.N):
            r = jac[i]
            j = ngmix.Jacobian(row=r[0], col=r[1], dvdrow=r[2], dvdcol=r[3], dudrow=r[4],
                               dudcol=r[5])
            obs = ngmix.Observation(images[i], weight=np.full(images[i].shape, 1.0 / sigma[i] ** 2),
                                    jacobian=j)
            gm = ngmix.GMixModel(pp[i], "exp").convolve(psf)
            d = gm.get_loglike(obs, more=True)
            ll[i] = [d["loglike"], d["s2n_numer"], d["s2n_denom"], d["npix"]]
            gm.fill_fdiff(obs, fdiff[i])
            im = base[i].copy()
            gm._fill_image(im, jacobian=j, fast_exp=True)
            rendered[i] = im
            print(tag, i, ll[i])
            sys.stdout.flush()
        out[tag + "_loglike"] = ll
        out[tag + "_fdiff"] = fdiff
        out[tag + "_rendered"] = rendered
    np.savez_compressed(OUT, **out)
    print("wrote %s (%.1f kB)" % (OUT, os.path.getsize(OUT) / 1e3))


if __name__ == "__main__":
    main()
