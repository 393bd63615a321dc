#!/usr/bin/env python
"""
Golden vectors for config 5 (the joint loglike of multi-epoch objects) from the
REFERENCE ITSELF under the numba shim: six objects of ten 64x64 epochs each
(tests/helpers/c5_inputs.py: numpy-only inputs, rebuilt by the tests), the
7-parameter 'bdf' model (16 gaussians) (x) a gaussian psf evaluated by
GMix.get_loglike(obs, more=True) at the generating parameters and at a moved
set -- per epoch and summed over the object's epochs, as bench.py's C5 step
sums them.  Build container only; tests/golden/c5.npz (outputs only) is
committed.  TEST INFRASTRUCTURE ONLY.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_c5.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "shim"), "/root/reference", os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402
import ngmix  # noqa: E402
from helpers import c5_inputs as c5  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "c5.npz")


def main():
    pars, moved, jac, images = c5.objects()
    psf = ngmix.GMixModel([0.0, 0.0, 0.0, 0.0, c5.TPSF, 1.0], "gauss")
    weight = np.full((c5.DIM, c5.DIM), 1.0 / c5.NOISE ** 2)
    out = {"image_sums": images.sum(axis=(2, 3))}   # (a check that the tests rebuilt the inputs)
    for tag, pp in (("truth", pars), ("moved", moved)):
        per_epoch = np.zeros((c5.NOBJ, c5.NEPOCH, 4))
        for o in range(c5.NOBJ):
            gm = ngmix.GMixModel(pp[o], "bdf").convolve(psf)
            for e in range(c5.NEPOCH):
                r = jac[o, e]
                j = ngmix.Jacobian(row=r[0], col=r[1], dvdrow=r[2], dvdcol=r[3], dudrow=r[4],
                                   dudcol=r[5])
                obs = ngmix.Observation(images[o, e], weight=weight, jacobian=j)
                d = gm.get_loglike(obs, more=True)
                per_epoch[o, e] = [d["loglike"], d["s2n_numer"], d["s2n_denom"], d["npix"]]
            print(tag, "object", o, per_epoch[o].sum(axis=0))
            sys.stdout.flush()
        out[tag + "_per_epoch"] = per_epoch
        out[tag + "_per_object"] = per_epoch.sum(axis=1)
    np.savez_compressed(OUT, **out)
    print("wrote %s (%.1f kB)" % (OUT, os.path.getsize(OUT) / 1e3))


if __name__ == "__main__":
    main()
