#!/usr/bin/env python
"""
Golden vectors for ngmix_amd.gmix_ndim.GMixND from the REFERENCE's GMixND
(under the numba shim): densities (scalar, array, per component, ln and
linear) of 1-, 2- and 3-dimensional mixtures at points in and far outside
them, seeded samples, and a seeded sklearn fit.  -> tests/golden/gmixnd.npz.
Build container only.  TEST INFRASTRUCTURE ONLY.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_gmixnd.py
"""
import contextlib
import io
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "shim"), "/root/reference"]

import numpy as np  # noqa: E402
import ngmix  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "gmixnd.npz")


def mixture(ndim, ngauss, rng):
    w = rng.uniform(0.2, 1.0, size=ngauss)
    means = rng.normal(scale=2.0, size=(ngauss, ndim))
    cov = np.zeros((ngauss, ndim, ndim))
    for i in range(ngauss):
        a = rng.normal(size=(ndim, ndim))
        cov[i] = a @ a.T + 0.3 * np.eye(ndim)
    return w / w.sum(), means, cov


def main():
    out = {}
    rng = np.random.RandomState(909)
    for ndim, ngauss in ((1, 1), (1, 3), (2, 2), (3, 4)):
        tag = "d%dg%d" % (ndim, ngauss)
        w, m, c = mixture(ndim, ngauss, rng)
        if ndim == 1:
            gm = ngmix.GMixND(w, m[:, 0], c[:, 0, 0], rng=np.random.RandomState(5))
        else:
            gm = ngmix.GMixND(w, m, c, rng=np.random.RandomState(5))
        out[tag + "_w"], out[tag + "_m"], out[tag + "_c"] = w, m, c
        pts = rng.normal(scale=3.0, size=(40, ndim))
        pts[0] = 60.0                      # far outside: ln p hugely negative
        out[tag + "_pts"] = pts
        arg = pts[:, 0] if ndim == 1 else pts
        out[tag + "_lnp"] = gm.get_lnprob_array(arg)
        out[tag + "_p"] = gm.get_prob_array(arg)
        out[tag + "_lnp_scalar"] = np.array([gm.get_lnprob_scalar(p) for p in pts])
        out[tag + "_p_scalar"] = np.array([gm.get_prob_scalar(p) for p in pts])
        k = ngauss - 1
        out[tag + "_lnp_comp"] = gm.get_lnprob_array(arg, component=k)
        out[tag + "_p_comp"] = gm.get_prob_array(arg, component=k)
        out[tag + "_norms"], out[tag + "_log_pnorms"], out[tag + "_icovars"] = \
            gm.norms, gm.log_pnorms, gm.icovars
        out[tag + "_sample_one"] = np.atleast_1d(gm.sample())
        out[tag + "_sample_7"] = gm.sample(7)
    # a seeded fit
    data = np.concatenate([np.random.RandomState(1).normal(size=(300, 2)) * [1.0, 0.3] + [2.0, 0.0],
                           np.random.RandomState(2).normal(size=(200, 2)) * [0.4, 0.8] - [1.0, 1.5]])
    gm = ngmix.GMixND(rng=np.random.RandomState(77))
    with contextlib.redirect_stdout(io.StringIO()):
        gm.fit(data, 2, n_iter=500)
    out["fit_data"] = data
    out["fit_w"], out["fit_m"], out["fit_c"] = gm.weights, gm.means, gm.covars
    out["fit_converged"] = np.array(gm.converged)
    out["fit_sample"] = gm.sample(5)
    np.savez_compressed(OUT, **out)
    print("wrote %s (%d arrays, %.1f kB)" % (OUT, len(out), os.path.getsize(OUT) / 1e3))


if __name__ == "__main__":
    main()
