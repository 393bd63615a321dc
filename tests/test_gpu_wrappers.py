"""
GPU tests of GaussMom / GaussMomBatch / PSFFluxFitter against outputs of the
reference's own classes (tests/golden/extra.npz, oracle/gen_golden_extra.py).
"""
import numpy as np
import pytest

import ngmix_amd as ngmix

pytestmark = pytest.mark.gpu


def _jac(rec):
    r = rec[0] if getattr(rec, "ndim", 0) else rec
    return ngmix.Jacobian(row=float(r["row0"]), col=float(r["col0"]),
                          dvdrow=float(r["dvdrow"]), dvdcol=float(r["dvdcol"]),
                          dudrow=float(r["dudrow"]), dudcol=float(r["dudcol"]))


def _obs(g, with_gmix=None):
    psf = ngmix.Observation(g["psf_image"], jacobian=_jac(g["psf_jac"]),
                            gmix=ngmix.GMix(pars=g["psf_pars"]))
    return ngmix.Observation(g["image"], weight=g["weight"], jacobian=_jac(g["jac"]),
                             psf=psf, gmix=with_gmix)


def _check_mom(res, g, tag):
    assert res["flags"] == int(g[tag + "_flags"]) == 0
    assert res["npix"] == int(g[tag + "_npix"])
    for k in ("flux", "flux_err", "T", "T_err", "s2n", "e1", "e2", "wsum",
              "sums_norm"):
        np.testing.assert_allclose(res[k], float(g[tag + "_" + k]), rtol=1e-11,
                                   err_msg=k)
    for k in ("pars", "sums", "sums_err", "e_err"):
        np.testing.assert_allclose(res[k], g[tag + "_" + k], rtol=1e-11,
                                   atol=1e-13 * np.abs(g[tag + "_" + k]).max(),
                                   err_msg=k)
    ref = g[tag + "_sums_cov"]
    np.testing.assert_allclose(res["sums_cov"], ref, rtol=1e-11,
                               atol=1e-13 * np.abs(ref).max())


@pytest.mark.parametrize("tag,hi", [("gm6", False), ("gm17", True)])
def test_gaussmom(golden, tag, hi):
    g = golden("extra")
    res = ngmix.GaussMom(fwhm=1.2, with_higher_order=hi).go(_obs(g))
    _check_mom(res, g, tag)


def test_gaussmom_batch(golden):
    from ngmix_amd.batch import StampBatch
    g = golden("extra")
    rng = np.random.RandomState(2)
    images = np.stack([g["image"], g["image"] + 0.01 * rng.normal(size=g["image"].shape)])
    weights = np.stack([g["weight"], g["weight"]])
    jac = np.tile(g["jac"].view("f8").reshape(1, 8), (2, 1))
    sb = StampBatch.from_images(images, weights, jac)
    out = ngmix.GaussMomBatch(fwhm=1.2).go(sb)
    _check_mom(out[0], g, "gm6")
    one = ngmix.GaussMom(fwhm=1.2).go(
        ngmix.Observation(images[1], weight=weights[1], jacobian=_jac(g["jac"])))
    for k in ("flux", "T", "e1", "e2", "s2n"):
        np.testing.assert_allclose(out[1][k], one[k], rtol=1e-12)


def test_psf_flux(golden):
    g = golden("extra")
    for tag, kw in (("pf", {}), ("pf_nonorm", {"normalize_psf": False})):
        res = ngmix.PSFFluxFitter(**kw).go(_obs(g))
        assert res["flags"] == int(g[tag + "_flags"])
        for k in ("chi2per", "dof", "flux", "flux_err"):
            np.testing.assert_allclose(res[k], float(g[tag + "_" + k]), rtol=1e-9,
                                       err_msg=tag + k)
    obs2 = _obs(g, with_gmix=ngmix.GMix(pars=g["tf_gmix_pars"]))
    res = ngmix.PSFFluxFitter(do_psf=False).go(obs2)
    assert res["flags"] == int(g["tf_flags"])
    for k in ("chi2per", "dof", "flux", "flux_err"):
        np.testing.assert_allclose(res[k], float(g["tf_" + k]), rtol=1e-9)
