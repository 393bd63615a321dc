#!/usr/bin/env python
"""
Round-2 golden vectors, by running the REFERENCE ITSELF under the numba shim
(build container only; tests/golden/api2.npz is committed.  TEST
INFRASTRUCTURE ONLY):

  * run_em(fixcov=True) through the public API (ngmix/em/em.py:370-394) on the
    em observation and guess of tests/golden/api.npz;
  * FitModel.calc_fdiff / calc_jacobian / calc_lnprob WITH prior rows
    (ngmix/fitting/results.py:142-210,439-466,480-625) on api.npz's 2-band x
    2-epoch object, and the Fitter result with that prior; the prior is
    tests/helpers/rows_prior.RowsPrior (closed-form rows shared by generator
    and test: the reference's priors package is out of scope);
  * get_model_deriv_data (results.py:955-1010) and FitModel statistics keys.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_r2.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "shim"), "/root/reference"]
sys.path.append(os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import ngmix  # noqa: E402
from ngmix.gexceptions import GMixRangeError  # noqa: E402
from ngmix.fitting import Fitter  # noqa: E402
from ngmix.fitting.results import FitModel, get_model_deriv_data  # noqa: E402
from helpers.rows_prior import RowsPrior  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "api2.npz")


def _jac(rec):
    r = rec[0] if getattr(rec, "ndim", 0) else rec
    return ngmix.Jacobian(row=float(r["row0"]), col=float(r["col0"]),
                          dvdrow=float(r["dvdrow"]), dvdcol=float(r["dvdcol"]),
                          dudrow=float(r["dudrow"]), dudcol=float(r["dudcol"]))


def _obs(g, prefix):
    psf = None
    if prefix + "_psf_image" in g:
        psf = ngmix.Observation(g[prefix + "_psf_image"], weight=g[prefix + "_psf_weight"],
                                jacobian=_jac(g[prefix + "_psf_jac"]))
        if prefix + "_psf_gmix_pars" in g:
            psf.set_gmix(ngmix.GMix(pars=g[prefix + "_psf_gmix_pars"]))
    return ngmix.Observation(g[prefix + "_image"], weight=g[prefix + "_weight"],
                             jacobian=_jac(g[prefix + "_jac"]), psf=psf)


def main():
    with np.load(os.path.join(ROOT, "tests", "golden", "api.npz")) as f:
        g = {k: f[k] for k in f.files}
    out = {}

    # ---- run_em(fixcov=True)
    obs_em = _obs(g, "em")
    guess = ngmix.GMix(pars=g["em_guess_pars"])
    r = ngmix.em.run_em(obs_em, guess, miniter=10, maxiter=80, fixcov=True)
    out["em_fixcov_numiter"] = np.array(r["numiter"])
    out["em_fixcov_fdiff"] = np.array(r["fdiff"])
    out["em_fixcov_sky"] = np.array(r["sky"])
    out["em_fixcov_flags"] = np.array(r["flags"])
    out["em_fixcov_pars"] = r.get_gmix().get_full_pars()
    out["em_fixcov_image"] = r.make_image()
    print("em fixcov numiter=%d flags=%d" % (r["numiter"], r["flags"]))

    # ---- FitModel with prior rows
    mb = ngmix.MultiBandObsList()
    for b in range(int(g["lm_nband"])):
        ol = ngmix.ObsList()
        for e in range(int(g["lm_nepoch"])):
            ol.append(_obs(g, "lm_b%d_e%d" % (b, e)))
        mb.append(ol)
    prior = RowsPrior(2, GMixRangeError)
    fm = FitModel(obs=mb, model="exp", guess=g["lm_guess"], prior=prior)
    out["lmp_n_prior_pars"] = np.array(fm.n_prior_pars)
    out["lmp_fdiff_size"] = np.array(fm.fdiff_size)
    for tag in ("guess", "truth"):
        p = g["lm_" + tag]
        out["lmp_fdiff_" + tag] = fm.calc_fdiff(p)
        out["lmp_jac_" + tag] = fm.calc_jacobian(p)
        ln = fm.calc_lnprob(p, more=True)
        out["lmp_lnprob_" + tag] = np.array(
            [ln["lnprob"], ln["s2n_numer"], ln["s2n_denom"], ln["npix"]])
        out["lmp_lnprob_scalar_" + tag] = np.array(fm.calc_lnprob(p))
    bad = g["lm_bad_pars"]
    out["lmp_fdiff_bad"] = fm.calc_fdiff(bad)
    out["lmp_jac_bad"] = fm.calc_jacobian(bad)
    out["lmp_lnprob_bad"] = np.array(fm.calc_lnprob(bad))
    # a point where only the PRIOR's forward step leaves its domain
    # (|g| + step >= 1 is not reachable with a model in range; T is): skip
    res = Fitter(model="exp", prior=prior).go(obs=mb, guess=g["lm_guess"])
    for k in ("flags", "nfev", "ier", "pars", "pars_err", "pars_cov0", "pars_cov",
              "lnprob", "s2n_numer", "s2n_denom", "npix", "chi2per", "dof", "s2n",
              "g", "g_cov", "g_err", "T", "T_err", "flux", "flux_cov", "flux_err"):
        out["lmp_fit_" + k] = np.array(res[k])
    print("LM with prior: flags=%d nfev=%d ier=%d" % (res["flags"], res["nfev"], res["ier"]))

    # ---- get_model_deriv_data for the analytic models, with and without psf
    psf3 = ngmix.GMix(pars=g["lm_b0_e0_psf_gmix_pars"])
    for model, pars in (("exp", g["lm_truth"][:6]), ("dev", g["lm_truth"][:6] * 1.1),
                        ("gauss", np.array([0.1, -0.2, 0.3, -0.1, 0.7, 3.0]))):
        for psf_tag, psf in (("nopsf", None), ("psf3", psf3)):
            pars = np.asarray(pars, dtype="f8")
            gm0 = ngmix.GMixModel(pars, model)
            gmc = gm0 if psf is None else gm0.convolve(psf)
            gpars, dcov = get_model_deriv_data(gm0, gmc, pars[2], pars[3], pars[4])
            out["dd_%s_%s_pars" % (model, psf_tag)] = np.asarray(pars, dtype="f8")
            out["dd_%s_%s_gpars" % (model, psf_tag)] = gpars
            out["dd_%s_%s_dcov" % (model, psf_tag)] = dcov
    # ---- PSFFluxFitter: template images and a two-epoch ObsList
    # (ngmix/fitting/results.py:677-914), inputs from tests/golden/extra.npz
    with np.load(os.path.join(ROOT, "tests", "golden", "extra.npz")) as f:
        x = {k: f[k] for k in f.files}
    for tag, kw in (("pft", {}), ("pft_nonorm", {"normalize_psf": False})):
        psf_obs = ngmix.Observation(x["psf_image"], jacobian=_jac(x["psf_jac"]))
        # the object's stamp and the psf template must have the same shape
        # for a template fit: use the object's image shape
        tmpl = ngmix.GMix(pars=x["psf_pars"]).make_image(
            x["image"].shape, jacobian=_jac(x["jac"]), fast_exp=True) * 3.7
        psf_obs.template = tmpl
        obs = ngmix.Observation(x["image"], weight=x["weight"], jacobian=_jac(x["jac"]),
                                psf=psf_obs)
        res = ngmix.fitting.PSFFluxFitter(**kw).go(obs)
        out["pft_template"] = tmpl
        for k in ("flags", "chi2per", "dof", "flux", "flux_err"):
            out[tag + "_" + k] = np.array(res[k])
        print(tag, res["flags"], res["flux"], res["flux_err"])
    psf = ngmix.Observation(x["psf_image"], jacobian=_jac(x["psf_jac"]),
                            gmix=ngmix.GMix(pars=x["psf_pars"]))
    ol = ngmix.ObsList()
    for e in range(2):
        pre = "nc_e%d_" % e
        ol.append(ngmix.Observation(x[pre + "image"], weight=x[pre + "weight"],
                                    jacobian=_jac(x[pre + "jac"]), psf=psf))
    for tag, kw in (("pfol", {}), ("pfol_nonorm", {"normalize_psf": False})):
        res = ngmix.fitting.PSFFluxFitter(**kw).go(ol)
        for k in ("flags", "chi2per", "dof", "flux", "flux_err"):
            out[tag + "_" + k] = np.array(res[k])
        print(tag, res["flags"], res["flux"], res["flux_err"])
    # ---- noise-power sandwich covariance for the models WITHOUT analytic
    # derivative images (central differences, ngmix/fitting/noise_cov.py:140-224)
    eol = ngmix.ObsList()
    for e in range(2):
        pre = "nc_e%d_" % e
        eol.append(ngmix.Observation(x[pre + "image"], weight=x[pre + "weight"],
                                     jacobian=_jac(x[pre + "jac"]), psf=psf,
                                     noise=x[pre + "noise"]))
    g0 = x["nc_guess"]
    for model, guess in (("turb", g0), ("bdf", np.array(list(g0[:5]) + [0.3, g0[5]]))):
        res = ngmix.fitting.Fitter(model=model, use_noise_image=True).go(obs=eol,
                                                                         guess=guess)
        pre = "ncfd_%s_" % model
        out[pre + "guess"] = guess
        for k in ("flags", "nfev", "ier"):
            out[pre + k] = np.array(res[k])
        for k in ("pars", "pars_err", "pars_cov", "pars_cov0"):
            out[pre + k] = np.array(res[k])
        print(pre, res["flags"], res["nfev"], res["pars"])
    # ---- CoellipFitter with four and five gaussians (fitters.py:120-141; the
    # reference's psf guessers go up to five, guessers.py:795-797)
    rng = np.random.RandomState(9142)
    cjac = ngmix.DiagonalJacobian(row=16.3, col=15.8, scale=0.263)
    Ts = [0.08, 0.2, 0.5, 1.2, 3.0]
    Fs = [0.35, 0.3, 0.2, 0.1, 0.05]
    for ng in (4, 5):
        truth = np.array([0.01, -0.02, 0.04, 0.03] + Ts[:ng] + Fs[:ng])
        cim = ngmix.GMixCoellip(truth).make_image((33, 33), jacobian=cjac)
        cim += 2.0e-5 * rng.normal(size=cim.shape)
        cobs = ngmix.Observation(cim, weight=np.full(cim.shape, 1.0 / 2.0e-5 ** 2),
                                 jacobian=cjac)
        guess = truth.copy()
        guess[4:] *= 1.0 + 0.04 * rng.uniform(-1, 1, size=2 * ng)
        guess[0:4] += 0.005 * rng.uniform(-1, 1, size=4)
        res = ngmix.fitting.CoellipFitter(ngauss=ng).go(obs=cobs, guess=guess)
        pre = "coellip%d_" % ng
        out[pre + "image"] = cim
        out[pre + "jac"] = cjac.get_data().copy()
        out[pre + "truth"] = truth
        out[pre + "guess"] = guess
        for k in ("flags", "nfev", "ier", "lnprob", "chi2per"):
            out[pre + k] = np.array(res[k])
        for k in ("pars", "pars_err", "pars_cov"):
            out[pre + k] = np.array(res[k])
        out[pre + "gmix_pars"] = res.get_gmix().get_full_pars()
        print(pre, res["flags"], res["nfev"], res["ier"], res["pars"], res["pars_err"])
    np.savez_compressed(OUT, **out)
    print("wrote %s (%.1f kB)" % (OUT, os.path.getsize(OUT) / 1e3))


if __name__ == "__main__":
    main()
