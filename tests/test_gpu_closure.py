"""
Closure tests in the style of the reference's own suite:
simulate with the package's renderer, measure / fit through the reference API,
recover the truth.  The reference's versions draw truth images with galsim
(ngmix/tests/test_admom.py, test_ml_fitting_*.py, test_em.py); here the truth is
rendered by GMix.make_image itself ("use ngmix to make the image to make sure
there are no pixelization effects", test_em.py:29-30), through unit, diagonal
and sheared WCS jacobians.
"""
import numpy as np
import pytest

import ngmix_amd as ngmix

pytestmark = pytest.mark.gpu

SCALE = 0.263


def _jacobians(dim, rng):
    cen = (dim - 1) / 2.0
    off = rng.uniform(-0.4, 0.4, size=2)
    yield "diag", ngmix.DiagonalJacobian(row=cen + off[0], col=cen + off[1], scale=SCALE)
    # a rotated + sheared wcs
    th = 0.6
    c, s = np.cos(th), np.sin(th)
    yield "sheared", ngmix.Jacobian(
        row=cen + off[0], col=cen + off[1],
        dvdrow=SCALE * c * 1.02, dvdcol=-SCALE * s * 0.97,
        dudrow=SCALE * s * 1.01, dudcol=SCALE * c * 0.99)


@pytest.mark.parametrize("jname", ["diag", "sheared"])
def test_fdiff_matches_the_analytic_gaussian(jname):
    """test_gmix.py::test_gmix_loglike_fdiff: fill_fdiff against
    pnorm exp(-chi2/2) area inside +-2 sigma, start in {0, 13}: pins the
    pixel order loc = start + r*ncol + c, the area factor and ierr"""
    rng = np.random.RandomState(883)
    dims = (13, 15)
    jac = dict(_jacobians(14, rng))[jname]
    jac.set_cen(row=6.1, col=7.3)
    pars = np.array([0.02, -0.03, 0.08, 0.05, 0.8, 3.0])
    gm = ngmix.GMixModel(pars, "gauss")
    image = np.zeros(dims)
    weight = rng.uniform(0.5, 2.0, size=dims)
    obs = ngmix.Observation(image, weight=weight, jacobian=jac)
    d = gm.get_data()[0]
    gm.set_norms()
    rows, cols = np.mgrid[0:dims[0], 0:dims[1]]
    v, u = jac(rows, cols)
    vd, ud = v - d["row"], u - d["col"]
    chi2 = d["dcc"] * vd ** 2 + d["drr"] * ud ** 2 - 2 * d["drc"] * vd * ud
    expected = (d["pnorm"] * np.exp(-0.5 * chi2) * jac.area) * np.sqrt(weight)
    for start in (0, 13):
        fdiff = np.zeros(start + image.size)
        gm.fill_fdiff(obs, fdiff, start=start)
        got = fdiff[start:].reshape(dims)
        w = chi2 < 4.0
        np.testing.assert_allclose(got[w], expected[w], rtol=4e-5)
        assert np.all(fdiff[:start] == 0.0)


@pytest.mark.parametrize("jname", ["diag", "sheared"])
def test_admom_recovers_a_sheared_gaussian(jname):
    """test_admom.py:12-103 in small: g to the convergence tolerance (etol = 1e-5
    on the ellipticity, i.e. ~5e-6 on g), T to 1e-5 relative, rho4 -> 2"""
    rng = np.random.RandomState(31415)
    dim = 71
    for trial in range(4):
        jac = dict(_jacobians(dim, rng))[jname]
        g1, g2 = rng.uniform(-0.2, 0.2, size=2)
        T = rng.uniform(0.8, 1.4)
        cen = rng.uniform(-0.3, 0.3, size=2) * SCALE
        gm = ngmix.GMixModel([cen[0], cen[1], g1, g2, T, 100.0], "gauss")
        im = gm.make_image((dim, dim), jacobian=jac)           # exact exp
        obs = ngmix.Observation(im, weight=np.full((dim, dim), 1e6), jacobian=jac)
        res = ngmix.admom.run_admom(obs=obs, guess=T * rng.uniform(0.9, 1.1), rng=rng)
        assert res["flags"] == 0
        fit = res.get_gmix()
        fg1, fg2, fT = fit.get_g1g2T()
        assert abs(fg1 - g1) < 6e-6 and abs(fg2 - g2) < 6e-6
        assert abs(fT / T - 1) < 1e-5
        assert abs(res["rho4"] - 2.0) < 1e-5
        # the iteration stops on the shape and size; the centre lags by < 0.01 pixel
        np.testing.assert_allclose(res["pars"][0:2], cen, atol=2e-3)


def _randomize(rng, gm):
    d = gm.get_data()
    for g in d:
        g["p"] *= rng.uniform(0.9, 1.1)
        g["row"] += rng.uniform(-SCALE, SCALE)
        g["col"] += rng.uniform(-SCALE, SCALE)
        g["irr"] += 0.1 * SCALE ** 2 * rng.uniform(-1, 1)
        g["irc"] += 0.1 * SCALE ** 2 * rng.uniform(-1, 1)
        g["icc"] += 0.1 * SCALE ** 2 * rng.uniform(-1, 1)


@pytest.mark.parametrize("with_psf", [False, True])
@pytest.mark.parametrize("noise", [0.0, 0.05])
def test_em_recovers_one_gaussian(noise, with_psf):
    """test_em.py:21-120: a 1-gaussian object (optionally psf-convolved) from a
    randomised guess: fractional 1e-3, centre to a tenth of a pixel, and the
    reconstructed image within 0.001/scale^2 + 5 noise"""
    rng = np.random.RandomState(42587)
    dim = 25
    jac = ngmix.DiagonalJacobian(row=12.1, col=11.8, scale=SCALE)
    Tpsf = 0.27
    pars = np.array([100.0, 0.03, -0.02, 0.35 * 1.1, 0.02, 0.35 * 0.9])  # p,row,col,irr,irc,icc
    gm = ngmix.GMix(pars=pars)
    psf_gm = ngmix.GMixModel([0.0, 0.0, 0.0, 0.0, Tpsf, 1.0], "gauss")
    truth = gm.convolve(psf_gm) if with_psf else gm
    im = truth.make_image((dim, dim), jacobian=jac) + noise * rng.normal(size=(dim, dim))
    psf_obs = None
    if with_psf:
        psf_obs = ngmix.Observation(psf_gm.make_image((dim, dim), jacobian=jac),
                                    jacobian=jac, gmix=psf_gm)
    obs = ngmix.Observation(im, jacobian=jac, psf=psf_obs)
    guess = gm.copy()
    _randomize(rng, guess)
    res = ngmix.em.run_em(obs=obs, guess=guess)
    assert res["flags"] == 0
    fit = res.get_gmix().get_full_pars()
    if noise == 0.0:
        assert abs(fit[0] / pars[0] - 1) < 1e-3
        assert abs(fit[1] - pars[1]) < SCALE / 10 and abs(fit[2] - pars[2]) < SCALE / 10
        for k in (3, 5):
            assert abs(fit[k] / pars[k] - 1) < 1e-3
    imfit = res.make_image()
    assert np.all(np.abs(imfit - obs.image) < 0.001 / SCALE ** 2 + noise * 5)


@pytest.mark.parametrize("model", ["gauss", "exp", "dev"])
@pytest.mark.parametrize("jname", ["diag", "sheared"])
def test_lm_recovers_a_noiseless_object(model, jname):
    """test_ml_fitting_exp_obj_gauss_psf.py / test_ml_fitting_gauss.py: the LM
    fitter recovers a noiseless psf-convolved object through a sheared wcs:
    g to 1e-5, T and flux to 5e-4 relative"""
    rng = np.random.RandomState(9911)
    dim = 53
    jac = dict(_jacobians(dim, rng))[jname]
    truth = np.array([0.01, -0.02, 0.12, -0.07, 0.9, 400.0])
    psf_gm = ngmix.GMixModel([0.0, 0.0, 0.01, -0.01, 0.27, 1.0], "gauss")
    gm = ngmix.GMixModel(truth, model).convolve(psf_gm)
    im = gm.make_image((dim, dim), jacobian=jac, fast_exp=True)
    psf_obs = ngmix.Observation(psf_gm.make_image((25, 25)), gmix=psf_gm)
    obs = ngmix.Observation(im, weight=np.full((dim, dim), 1e8), jacobian=jac, psf=psf_obs)
    guess = truth * rng.uniform(0.9, 1.1, size=6)
    guess[0:2] = truth[0:2] + rng.uniform(-0.05, 0.05, size=2)
    res = ngmix.fitting.Fitter(model=model).go(obs=obs, guess=guess)
    assert res["flags"] == 0
    assert np.all(np.abs(res["g"] - truth[2:4]) < 1e-5)
    assert abs(res["T"] / truth[4] - 1) < 5e-4
    assert abs(res["flux"] / truth[5] - 1) < 5e-4
    # and the batched driver lands on the same solution
    from ngmix_amd.batch import StampBatch, GMixBatch
    sb = StampBatch.from_observations([obs])
    psf = GMixBatch.from_numpy(psf_gm.get_data()[None, :])
    bres = ngmix.LMBatchFitter(model).go(sb, guess[None, :], psf=psf)
    assert bres["flags"][0] == 0
    np.testing.assert_allclose(bres["pars"][0], res["pars"], rtol=1e-6, atol=1e-8)


def test_runner_is_deterministic():
    """test_runners.py:79-81,172-175: the same seed twice gives bit-identical
    parameters (no atomics, fixed-order reductions)"""
    pars_seen = []
    for _ in range(2):
        rng = np.random.RandomState(77)
        dim = 32
        jac = ngmix.DiagonalJacobian(row=15.3, col=15.9, scale=SCALE)
        psf_gm = ngmix.GMixModel([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], "gauss")
        gm = ngmix.GMixModel([0.02, 0.01, 0.1, 0.05, 0.7, 150.0], "exp").convolve(psf_gm)
        im = gm.make_image((dim, dim), jacobian=jac, fast_exp=True)
        im += 0.01 * rng.normal(size=im.shape)
        psf_obs = ngmix.Observation(psf_gm.make_image((25, 25)), gmix=psf_gm)
        obs = ngmix.Observation(im, weight=np.full(im.shape, 1e4), jacobian=jac,
                                psf=psf_obs)

        class Guesser(object):
            def __call__(self, obs):
                return np.array([0.0, 0.0, 0.08, 0.03, 0.6, 140.0]) * (
                    1.0 + 0.01 * rng.uniform(-1, 1, size=6))

        runner = ngmix.runners.Runner(fitter=ngmix.fitting.Fitter(model="exp"),
                                      guesser=Guesser(), ntry=2)
        res = runner.go(obs=obs)
        assert res["flags"] == 0
        pars_seen.append(res["pars"].copy())
    assert np.array_equal(pars_seen[0], pars_seen[1])
