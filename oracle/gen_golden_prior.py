#!/usr/bin/env python
"""
Golden vectors for LM fits WITH the reference's separable joint prior
(ngmix/joint_prior.py PriorSimpleSep built from ngmix/priors CenPrior, GPriorBA
and TwoSidedErf), by running the REFERENCE ITSELF under the numba shim: one-
and two-band objects, lmder (analytic jacobian) and lmdif.  Build container
only; the fixture tests/golden/prior.npz is committed.  TEST INFRASTRUCTURE ONLY.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_prior.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "shim"), "/root/reference"]

import numpy as np  # noqa: E402
import ngmix  # noqa: E402
from ngmix import priors, joint_prior  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "prior.npz")
SCALE = 0.263

# the prior's parameters (tests rebuild the batch prior from these)
CEN_SIGMA = 0.05
G_SIGMA = 0.2
T_ERF = (-0.05, 0.03, 3.0, 0.3)
F_ERF = (-1.0, 0.5, 500.0, 20.0)


# the bounded case ("bb"): Normal T and flux priors carrying leastsqbound
# bounds, two-sided and one-sided
T_NORMAL = (0.5, 0.3, (0.05, 2.0))
F_NORMAL = (70.0, 40.0, (1.0, None))


def make_prior(nband, rng, bounded=False):
    if bounded:
        Tp = priors.Normal(T_NORMAL[0], T_NORMAL[1], rng=rng, bounds=T_NORMAL[2])
        Fp = [priors.Normal(F_NORMAL[0], F_NORMAL[1], rng=rng, bounds=F_NORMAL[2])
              for _ in range(nband)]
    else:
        Tp = priors.TwoSidedErf(*T_ERF, rng=rng)
        Fp = [priors.TwoSidedErf(*F_ERF, rng=rng) for _ in range(nband)]
    return joint_prior.PriorSimpleSep(
        priors.CenPrior(0.0, 0.0, CEN_SIGMA, CEN_SIGMA, rng=rng),
        priors.GPriorBA(G_SIGMA, rng=rng), Tp, Fp if nband > 1 else Fp[0])


def main():
    rng = np.random.RandomState(2718)
    out = dict(cen_sigma=CEN_SIGMA, g_sigma=G_SIGMA, T_erf=np.array(T_ERF),
               F_erf=np.array(F_ERF), T_normal=np.array(T_NORMAL[:2]),
               T_bounds=np.array(T_NORMAL[2]), F_normal=np.array(F_NORMAL[:2]),
               F_lower_bound=F_NORMAL[2][0])
    psf_gm = ngmix.GMixModel([0.0, 0.0, 0.01, -0.02, 0.27, 1.0], "gauss")
    dim = 28
    for tag, nband in (("b1", 1), ("b2", 2), ("bb", 1)):
        truth = np.array([0.04, -0.03, 0.25, -0.15, 0.45] + [60.0, 95.0][:nband])
        mb = ngmix.MultiBandObsList()
        for b in range(nband):
            ol = ngmix.ObsList()
            jac = ngmix.Jacobian(row=13.4 + 0.2 * b, col=13.7 - 0.1 * b, dvdrow=SCALE,
                                 dvdcol=0.004, dudrow=-0.006, dudcol=SCALE * 1.01)
            bp = list(truth[:5]) + [truth[5 + b]]
            gm = ngmix.GMixModel(bp, "exp").convolve(psf_gm)
            im = gm.make_image((dim, dim), jacobian=jac, fast_exp=True)
            im += 0.03 * rng.normal(size=im.shape)
            wt = np.full(im.shape, 1.0 / 0.03 ** 2)
            pobs = ngmix.Observation(np.zeros((5, 5)), jacobian=jac, gmix=psf_gm.copy())
            ol.append(ngmix.Observation(im, weight=wt, jacobian=jac, psf=pobs))
            mb.append(ol)
            out["%s_image%d" % (tag, b)] = im
            out["%s_weight%d" % (tag, b)] = wt
            out["%s_jac%d" % (tag, b)] = jac.get_data().copy()
        out[tag + "_psf_pars"] = psf_gm.get_full_pars()
        guess = truth * (1.0 + 0.05 * rng.uniform(-1, 1, size=truth.size))
        guess[0:2] = truth[0:2] + 0.02 * rng.uniform(-1, 1, size=2)
        out[tag + "_guess"] = guess
        prior = make_prior(nband, rng, bounded=(tag == "bb"))
        if tag == "bb":
            assert prior.bounds is not None
            # (scipy >= 1.15 returns a 0-based ipvt; the reference's `ipvt - 1`
            # then scrambles pars_cov of bounded fits -- pars / nfev / ier and
            # lnprob are unaffected and are what the tests use)
            out["bb_scipy_version"] = np.array(__import__("scipy").__version__)
        # the prior itself at a few points (rows and ln p)
        pts = np.array([guess, truth, truth * 1.1])
        rows = np.zeros((pts.shape[0], 4 + nband))
        lnp = np.zeros(pts.shape[0])
        for i, p in enumerate(pts):
            f = np.zeros(5 + nband)
            n = prior.fill_fdiff(p, f)
            rows[i] = f[:n]
            lnp[i] = prior.get_lnprob_scalar(p)
        out[tag + "_prior_pts"] = pts
        out[tag + "_prior_rows"] = rows
        out[tag + "_prior_lnp"] = lnp
        for mode, analytic in (("lmder", True), ("lmdif", False)):
            res = ngmix.fitting.Fitter(model="exp", prior=prior,
                                       analytic_jacobian=analytic).go(obs=mb, guess=guess)
            pre = "%s_%s_" % (tag, mode)
            for k in ("flags", "nfev", "ier", "lnprob", "chi2per", "dof", "s2n"):
                out[pre + k] = res[k]
            for k in ("pars", "pars_err", "pars_cov", "pars_cov0"):
                out[pre + k] = np.array(res[k])
            print(pre, res["flags"], res["nfev"], res["pars"])
    np.savez_compressed(OUT, **out)
    print("wrote %s (%.1f kB)" % (OUT, os.path.getsize(OUT) / 1e3))


if __name__ == "__main__":
    main()
