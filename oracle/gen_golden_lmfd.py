#!/usr/bin/env python
"""
Golden vectors for LM fits of the models WITHOUT analytic derivatives (the
reference runs MINPACK lmdif for them: turb, bdf, bd; fitters.py:93-104), and
for 'dev' with analytic_jacobian=False, by running the REFERENCE ITSELF under
the numba shim.  Build container only; tests/golden/lmfd.npz is committed.
TEST INFRASTRUCTURE ONLY.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_lmfd.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "shim"), "/root/reference"]

import numpy as np  # noqa: E402
import ngmix  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "lmfd.npz")
SCALE = 0.263

CASES = {
    # model: truth (shape pars..., flux per band)
    "turb": [0.03, -0.04, 0.10, -0.06, 0.45, 70.0, 110.0],
    "bdf": [0.02, 0.03, -0.12, 0.08, 0.60, 0.35, 90.0, 140.0],
    "bd": [-0.03, 0.02, 0.08, 0.10, 0.55, 0.3, 0.4, 80.0, 120.0],
    "dev": [0.01, -0.02, 0.15, 0.05, 0.50, 100.0, 150.0],
}


def main():
    rng = np.random.RandomState(1618)
    out = {"models": np.array(sorted(CASES))}
    psf_gm = ngmix.GMixModel([0.0, 0.0, -0.01, 0.02, 0.26, 1.0], "gauss")
    out["psf_pars"] = psf_gm.get_full_pars()
    dim, nband = 30, 2
    for model in sorted(CASES):
        truth = np.array(CASES[model])
        nshape = truth.size - nband
        mb = ngmix.MultiBandObsList()
        for b in range(nband):
            ol = ngmix.ObsList()
            jac = ngmix.Jacobian(row=14.3 + 0.3 * b, col=14.8 - 0.2 * b, dvdrow=SCALE * 0.99,
                                 dvdcol=0.005, dudrow=-0.003, dudcol=SCALE)
            bp = list(truth[:nshape]) + [truth[nshape + b]]
            gm = ngmix.GMixModel(bp, model).convolve(psf_gm)
            im = gm.make_image((dim, dim), jacobian=jac, fast_exp=True)
            im += 0.02 * rng.normal(size=im.shape)
            wt = np.full(im.shape, 1.0 / 0.02 ** 2)
            pobs = ngmix.Observation(np.zeros((5, 5)), jacobian=jac, gmix=psf_gm.copy())
            ol.append(ngmix.Observation(im, weight=wt, jacobian=jac, psf=pobs))
            mb.append(ol)
            out["%s_image%d" % (model, b)] = im
            out["%s_weight%d" % (model, b)] = wt
            out["%s_jac%d" % (model, b)] = jac.get_data().copy()
        guess = truth * (1.0 + 0.04 * rng.uniform(-1, 1, size=truth.size))
        guess[0:2] = truth[0:2] + 0.02 * rng.uniform(-1, 1, size=2)
        out[model + "_guess"] = guess
        res = ngmix.fitting.Fitter(model=model, analytic_jacobian=False).go(obs=mb, guess=guess)
        for k in ("flags", "nfev", "ier", "lnprob", "chi2per", "dof", "s2n"):
            out["%s_%s" % (model, k)] = res[k]
        for k in ("pars", "pars_err", "pars_cov"):
            out["%s_%s" % (model, k)] = np.array(res[k])
        print(model, res["flags"], res["nfev"], res["ier"], res["pars"])
    # ---- CoellipFitter (the psf fitter of the LM psf runners): co-elliptical
    # gaussians fitted to a psf image, lmdif
    cjac = ngmix.DiagonalJacobian(row=12.2, col=11.7, scale=SCALE)
    truth_psf = ngmix.GMixCoellip([0.01, -0.02, 0.03, 0.02, 0.12, 0.32, 0.9, 0.6, 0.3, 0.1])
    cim = truth_psf.make_image((25, 25), jacobian=cjac)
    cim += 2.0e-4 * rng.normal(size=cim.shape)
    cobs = ngmix.Observation(cim, weight=np.full(cim.shape, 1.0 / 2.0e-4 ** 2), jacobian=cjac)
    out["coellip_image"] = cim
    out["coellip_jac"] = cjac.get_data().copy()
    for ng, guess in ((2, [0.0, 0.0, 0.0, 0.0, 0.15, 0.6, 0.7, 0.3]),
                      (3, [0.0, 0.0, 0.01, 0.01, 0.1, 0.3, 0.8, 0.55, 0.35, 0.1])):
        guess = np.array(guess)
        res = ngmix.fitting.CoellipFitter(ngauss=ng).go(obs=cobs, guess=guess)
        pre = "coellip%d_" % ng
        out[pre + "guess"] = guess
        for k in ("flags", "nfev", "ier", "lnprob", "chi2per"):
            out[pre + k] = res[k]
        for k in ("pars", "pars_err", "pars_cov"):
            out[pre + k] = np.array(res[k])
        out[pre + "gmix_pars"] = res.get_gmix().get_full_pars()
        print(pre, res["flags"], res["nfev"], res["ier"], res["pars"])
    np.savez_compressed(OUT, **out)
    print("wrote %s (%.1f kB)" % (OUT, os.path.getsize(OUT) / 1e3))


if __name__ == "__main__":
    main()
