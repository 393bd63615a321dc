#!/usr/bin/env python
"""
The result-dict surface of the reference's entry points on the hot path: for
each of Fitter (every simple model, one band and three), CoellipFitter,
PSFFluxFitter, GaussMom, run_admom / AdmomFitter, the EM fitters and the
template-flux fitter, the REFERENCE ITSELF (under the numba shim) is run on a
small scene and the key set of its result -- with each value's kind and shape
-- is written to tests/golden/result_keys.json.  The -m gpu test compares the
dicts ngmix_amd returns (per-object and the many-object forms) key by key.
Build container only.  TEST INFRASTRUCTURE ONLY.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_keys.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "shim"), "/root/reference"]

import numpy as np  # noqa: E402
import ngmix  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "result_keys.json")


def describe(res):
    out = {}
    for k in res.keys():
        v = res[k]
        if isinstance(v, str):
            out[k] = "str"
        elif isinstance(v, dict):
            out[k] = "dict"
        elif isinstance(v, (bool, np.bool_)):
            out[k] = "bool"
        else:
            a = np.asarray(v)
            out[k] = "%s%s" % (a.dtype.kind, list(a.shape))
    return out


def scene(seed, model="exp", nband=1, dim=25, psf=True):
    rng = np.random.RandomState(seed)
    jac = ngmix.DiagonalJacobian(row=(dim - 1) / 2 + 0.1, col=(dim - 1) / 2 - 0.2, scale=0.263)
    pars = [0.02, -0.03, 0.08, -0.05, 0.5, 100.0]
    psf_gm = ngmix.GMixModel([0.0, 0.0, 0.01, -0.01, 0.27, 1.0], "gauss")
    gm = ngmix.GMixModel(pars, "gauss" if model in ("bdf", "bd", "coellip") else model)

    def one():
        g = gm.convolve(psf_gm) if psf else gm
        im = g.make_image((dim, dim), jacobian=jac) + 0.05 * rng.normal(size=(dim, dim))
        pim = psf_gm.make_image((dim, dim), jacobian=jac) + 1e-4 * rng.normal(size=(dim, dim))
        pobs = ngmix.Observation(pim, weight=np.full(pim.shape, 1e8), jacobian=jac, gmix=psf_gm.copy())
        return ngmix.Observation(im, weight=np.full(im.shape, 400.0), jacobian=jac,
                                 psf=pobs if psf else None)
    if nband == 1:
        return one(), np.array(pars)
    mb = ngmix.MultiBandObsList()
    for _ in range(nband):
        ol = ngmix.ObsList()
        ol.append(one())
        mb.append(ol)
    return mb, np.array(pars[:5] + [pars[5]] * nband)


def main():
    out = {}
    for model in ("gauss", "exp", "dev", "turb"):
        for nband in (1, 3):
            obs, guess = scene(3, model, nband)
            res = ngmix.fitting.Fitter(model=model).go(obs=obs, guess=guess)
            assert res["flags"] == 0
            out["Fitter_%s_%d" % (model, nband)] = describe(res)
    obs, guess = scene(4, "bdf", 1)
    res = ngmix.fitting.Fitter(model="bdf").go(obs=obs, guess=np.array(list(guess[:5]) + [0.5, 100.0]))
    out["Fitter_bdf_1"] = describe(res)
    res = ngmix.fitting.Fitter(model="bd").go(
        obs=obs, guess=np.array(list(guess[:5]) + [0.0, 0.5, 100.0]))
    out["Fitter_bd_1"] = describe(res)
    obs, guess = scene(5, "coellip", 1, psf=False)
    res = ngmix.fitting.CoellipFitter(ngauss=2).go(
        obs=obs, guess=np.array([0.0, 0.0, 0.05, 0.0, 0.3, 0.6, 40.0, 60.0]))
    assert res["flags"] == 0
    out["CoellipFitter_2"] = describe(res)

    obs, guess = scene(6, "exp", 1)
    res = ngmix.fitting.PSFFluxFitter().go(obs=obs)
    out["PSFFluxFitter"] = describe(res)
    res = ngmix.fitting.PSFFluxFitter(do_psf=False).go(obs=obs.psf)
    out["PSFFluxFitter_do_psf_False"] = describe(res)
    res = ngmix.gaussmom.GaussMom(fwhm=1.2).go(obs=obs)
    out["GaussMom"] = describe(res)
    res = ngmix.gaussmom.GaussMom(fwhm=1.2, with_higher_order=True).go(obs=obs)
    out["GaussMom_higher"] = describe(res)
    rng = np.random.RandomState(7)
    res = ngmix.admom.run_admom(obs=obs, guess=0.5, rng=rng)
    assert res["flags"] == 0
    out["run_admom"] = describe(res)
    res = ngmix.admom.AdmomFitter(rng=rng).go(obs=obs, guess=0.5)
    out["AdmomFitter"] = describe(res)
    gm_guess = ngmix.GMixModel([0.0, 0.0, 0.0, 0.0, 0.6, 1.0], "gauss")
    res = ngmix.em.run_em(obs=obs, guess=gm_guess)
    assert res["flags"] == 0
    out["run_em"] = describe(res)
    res = ngmix.em.EMFitterFixCen().go(obs=obs, guess=gm_guess.copy())
    out["EMFitterFixCen"] = describe(res)
    res = ngmix.em.EMFitterFluxOnly().go(obs=obs, guess=gm_guess.copy())
    out["EMFitterFluxOnly"] = describe(res)
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", OUT, len(out), "entries")
    for k in sorted(out):
        print(k, sorted(out[k]))


if __name__ == "__main__":
    main()
