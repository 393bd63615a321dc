"""
The result-dict surface, key by key: tests/golden/result_keys.json holds the
key set (and each value's kind and shape) of what the REFERENCE's entry points
return on a small scene (oracle/gen_golden_keys.py ran them); the same calls
through ngmix_amd must return the same keys with the same kinds and shapes --
per object, and element i of the many-object forms.
"""
import json
import os

import numpy as np
import pytest

import ngmix_amd as ngmix

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "result_keys.json")) as f:
    REF = json.load(f)


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
    """oracle/gen_golden_keys.py's scene"""
    rng = np.random.RandomState(seed)
    jac = ngmix.DiagonalJacobian(row=(dim - 1) / 2 + 0.1, col=(dim - 1) / 2 - 0.2, scale=0.263)
    pars = [0.02, -0.03, 0.08, -0.05, 0.5, 100.0]
    psf_gm = ngmix.GMixModel([0.0, 0.0, 0.01, -0.01, 0.27, 1.0], "gauss")
    gm = ngmix.GMixModel(pars, "gauss" if model in ("bdf", "bd", "coellip") else model)

    def one():
        g = gm.convolve(psf_gm) if psf else gm
        im = g.make_image((dim, dim), jacobian=jac) + 0.05 * rng.normal(size=(dim, dim))
        pim = psf_gm.make_image((dim, dim), jacobian=jac) + 1e-4 * rng.normal(size=(dim, dim))
        pobs = ngmix.Observation(pim, weight=np.full(pim.shape, 1e8), jacobian=jac,
                                 gmix=psf_gm.copy())
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


def same_surface(name, res, ignore=()):
    ref = REF[name]
    got = describe(res)
    for k in ignore:
        got.pop(k, None)
    assert set(got) == set(ref), (name, sorted(set(got) ^ set(ref)))
    for k in ref:
        # an integer the reference holds as a python int and a numpy integer
        # are the same kind; shapes must agree exactly
        assert got[k] == ref[k], (name, k, got[k], ref[k])


@pytest.mark.parametrize("batched", [False, True])
@pytest.mark.parametrize("nband", [1, 3])
@pytest.mark.parametrize("model", ["gauss", "exp", "dev", "turb"])
def test_fitter_surface(model, nband, batched):
    obs, guess = scene(3, model, nband)
    fitter = ngmix.fitting.Fitter(model=model, batched=batched)
    res = fitter.go(obs=obs, guess=guess)
    assert res["flags"] == 0
    same_surface("Fitter_%s_%d" % (model, nband), res)
    if batched:
        many = fitter.go_many([obs] * 3, np.tile(guess, (3, 1)))
        same_surface("Fitter_%s_%d" % (model, nband), many[1])


@pytest.mark.parametrize("batched", [False, True])
def test_bdf_bd_coellip_surface(batched):
    obs, guess = scene(4, "bdf", 1)
    gb = np.array(list(guess[:5]) + [0.5, 100.0])
    f = ngmix.fitting.Fitter(model="bdf", batched=batched)
    same_surface("Fitter_bdf_1", f.go(obs=obs, guess=gb))
    gd = np.array(list(guess[:5]) + [0.0, 0.5, 100.0])
    f2 = ngmix.fitting.Fitter(model="bd", batched=batched)
    same_surface("Fitter_bd_1", f2.go(obs=obs, guess=gd))
    cobs, _ = scene(5, "coellip", 1, psf=False)
    gc = np.array([0.0, 0.0, 0.05, 0.0, 0.3, 0.6, 40.0, 60.0])
    f3 = ngmix.fitting.CoellipFitter(ngauss=2, batched=batched)
    res = f3.go(obs=cobs, guess=gc)
    assert res["flags"] == 0
    same_surface("CoellipFitter_2", res)
    if batched:
        same_surface("Fitter_bdf_1", f.go_many([obs] * 2, np.tile(gb, (2, 1)))[1])
        same_surface("Fitter_bd_1", f2.go_many([obs] * 2, np.tile(gd, (2, 1)))[0])
        same_surface("CoellipFitter_2", f3.go_many([cobs] * 2, np.tile(gc, (2, 1)))[1])


def test_moment_and_flux_surfaces():
    obs, _ = scene(6, "exp", 1)
    same_surface("PSFFluxFitter", ngmix.fitting.PSFFluxFitter().go(obs=obs))
    same_surface("PSFFluxFitter_do_psf_False",
                 ngmix.fitting.PSFFluxFitter(do_psf=False).go(obs=obs.psf))
    same_surface("GaussMom", ngmix.gaussmom.GaussMom(fwhm=1.2).go(obs=obs))
    same_surface("GaussMom_higher",
                 ngmix.gaussmom.GaussMom(fwhm=1.2, with_higher_order=True).go(obs=obs))
    same_surface("GaussMom", ngmix.gaussmom.GaussMom(fwhm=1.2).go_many([obs] * 3)[2])
    rng = np.random.RandomState(7)
    res = ngmix.admom.run_admom(obs=obs, guess=0.5, rng=rng)
    assert res["flags"] == 0
    same_surface("run_admom", res)
    same_surface("AdmomFitter", ngmix.admom.AdmomFitter(rng=rng).go(obs=obs, guess=0.5))
    same_surface("run_admom", ngmix.admom.run_admom_many([obs] * 3, guess=0.5, rng=rng)[1])


def test_em_surfaces():
    obs, _ = scene(6, "exp", 1)
    gm_guess = ngmix.GMixModel([0.0, 0.0, 0.0, 0.0, 0.6, 1.0], "gauss")
    res = ngmix.em.run_em(obs=obs, guess=gm_guess)
    assert res["flags"] == 0
    same_surface("run_em", res)
    same_surface("EMFitterFixCen", ngmix.em.EMFitterFixCen().go(obs=obs, guess=gm_guess.copy()))
    same_surface("EMFitterFluxOnly",
                 ngmix.em.EMFitterFluxOnly().go(obs=obs, guess=gm_guess.copy()))
    same_surface("run_em", ngmix.em.run_em_many([obs] * 3, [gm_guess.copy() for _ in range(3)])[1])
