"""
Fitter.go has two routes: the one-object batch of the lock-step driver
(batched=True -- what a user gets by default, NGMIX_FITTER_BATCHED unset) and
MINPACK on the host calling back into one kernel per evaluation
(batched=False -- what tests/conftest.py makes the suite's default, as the
independent check).  Here both routes run the SAME fits and their result dicts
are compared key by key: the forward-difference models (turb / bdf / bd), fits
that fail (maxfev reached), psf mixtures of different sizes (the batched route
hands those to MINPACK), and the flows built on Fitter -- runners, bootstrap --
under either value of the environment switch.

Reference: ngmix/fitting/fitters.py:64-112 (Fitter.go), leastsqbound.py:33-155
(run_leastsq's flags and defaults), runners.py:116-223, bootstrap.py:67-154.
"""
import numpy as np
import pytest

import ngmix_amd as ngmix

pytestmark = pytest.mark.gpu

SCALE = 0.263


def _object(model, rng, dim=40, noise=0.005, psf_ngauss=1):
    """one observation of `model` (x) psf with its psf observation"""
    jac = ngmix.DiagonalJacobian(row=(dim - 1) / 2 + 0.2, col=(dim - 1) / 2 - 0.3, scale=SCALE)
    if psf_ngauss == 1:
        psf_gm = ngmix.GMixModel([0.0, 0.0, 0.01, -0.01, 0.27, 1.0], "gauss")
    else:
        psf_gm = ngmix.GMixModel([0.0, 0.0, 0.01, -0.01, 0.27, 1.0], "turb")
    truth = {
        "gauss": [0.02, -0.03, 0.08, 0.03, 0.5, 120.0],
        "exp": [0.02, -0.03, 0.08, 0.03, 0.6, 120.0],
        "dev": [0.02, -0.03, 0.08, 0.03, 0.8, 220.0],
        "turb": [0.02, -0.03, 0.08, 0.03, 0.5, 120.0],
        "bdf": [0.02, -0.03, 0.08, 0.03, 0.7, 0.4, 160.0],
        "bd": [0.02, -0.03, 0.08, 0.03, 0.7, 0.1, 0.4, 160.0],
    }[model]
    truth = np.array(truth)
    gm0 = ngmix.GMixModel(truth, model)
    im = gm0.convolve(psf_gm).make_image((dim, dim), jacobian=jac, fast_exp=True)
    im = im + noise * rng.normal(size=im.shape)
    pdim = 25
    pjac = ngmix.DiagonalJacobian(row=(pdim - 1) / 2, col=(pdim - 1) / 2, scale=SCALE)
    # (a psf image with noise and its weight: a noiseless image with unit
    # weights fits with chi2 = 0 and a covariance that fails run_leastsq's
    # checks -- on both routes and in the reference alike)
    pnoise = 1.0e-4
    pim = psf_gm.make_image((pdim, pdim), jacobian=pjac) + pnoise * rng.normal(size=(pdim, pdim))
    psf_obs = ngmix.Observation(pim, weight=np.full(pim.shape, 1.0 / pnoise ** 2),
                                jacobian=pjac, gmix=psf_gm)
    obs = ngmix.Observation(im, weight=np.full(im.shape, 1.0 / noise ** 2), jacobian=jac,
                            psf=psf_obs)
    return obs, truth


def _compare(a, b, tol_sigma, rtol_err):
    """two successful result dicts: the same keys; integers equal where the
    algorithm pins them; values within a fraction of the quoted errors"""
    assert set(a.keys()) == set(b.keys()), set(a.keys()) ^ set(b.keys())
    assert a["flags"] == b["flags"] == 0
    assert a["npix"] == b["npix"] and a["dof"] == b["dof"]
    err = b["pars_err"]
    assert np.all(np.abs(a["pars"] - b["pars"]) <= tol_sigma * err), (a["pars"], b["pars"])
    np.testing.assert_allclose(a["pars_err"], err, rtol=rtol_err)
    for k in ("lnprob", "chi2per", "s2n"):
        np.testing.assert_allclose(a[k], b[k], rtol=1e-5, err_msg=k)
    for k in ("g", "g_err", "T", "T_err", "flux", "flux_err"):
        np.testing.assert_allclose(a[k], b[k], rtol=max(rtol_err, 1e-4), atol=tol_sigma * 1e-2,
                                   err_msg=k)
    assert np.shape(a["g_cov"]) == np.shape(b["g_cov"]) == (2, 2)
    assert np.shape(a["pars_cov"]) == np.shape(b["pars_cov"])


@pytest.mark.parametrize("model", ["dev", "turb", "bdf", "bd"])
def test_both_routes_one_result_dict(model):
    """the same fit through the lock-step driver and through MINPACK: 'dev'
    (lmder, analytic jacobian: nfev equal) and the forward-difference models
    (lmdif on both routes: the iterates differ at the rounding of the
    factorisation, the solutions by a small fraction of their errors)"""
    rng = np.random.RandomState({"dev": 5, "turb": 6, "bdf": 7, "bd": 8}[model])
    obs, truth = _object(model, rng)
    guess = truth * (1.0 + 0.02 * rng.uniform(-1, 1, size=truth.size))
    guess[0:2] = truth[0:2] + 0.01 * rng.uniform(-1, 1, size=2)
    fb = ngmix.fitting.Fitter(model=model, batched=True)
    fm = ngmix.fitting.Fitter(model=model, batched=False)
    a = fb.go(obs=obs, guess=guess)
    b = fm.go(obs=obs, guess=guess)
    assert fb._batch_fitter is not None and fm._batch_fitter is None
    if model == "dev":
        assert a["nfev"] == b["nfev"] and a["ier"] == b["ier"]
        _compare(a, b, 1e-4, 1e-4)
    else:
        # bd's logTratio / fracdev valley is nearly flat: compare in sigmas
        _compare(a, b, 5e-2 if model == "bd" else 2e-2, 5e-2 if model == "bd" else 2e-2)
    # both recover the truth
    nsig = 6.0
    assert np.all(np.abs(a["pars"] - truth) <= nsig * a["pars_err"] + 1e-3)
    # model accessors work off either
    assert len(a.get_gmix()) == len(b.get_gmix())
    assert a.make_image().shape == obs.image.shape


@pytest.mark.parametrize("model", ["exp", "bdf"])
def test_failed_fit_is_the_same_failure_on_both_routes(model):
    """maxfev reached (ier 5 -> flags 2**0, leastsqbound.py:118-122): the same
    flags and ier, no statistics keys, the same key set on both routes"""
    rng = np.random.RandomState(11)
    obs, truth = _object(model, rng)
    guess = truth * np.where(np.arange(truth.size) >= 4, 1.7, 1.0)
    guess[2:4] = 0.3, -0.2
    pars = {"maxfev": 3, "ftol": 1.0e-5, "xtol": 1.0e-5}
    a = ngmix.fitting.Fitter(model=model, fit_pars=pars, batched=True).go(obs=obs, guess=guess)
    b = ngmix.fitting.Fitter(model=model, fit_pars=pars, batched=False).go(obs=obs, guess=guess)
    assert b["flags"] != 0
    assert a["flags"] == b["flags"], (a["flags"], b["flags"], a["ier"], b["ier"])
    assert a["ier"] == b["ier"] == 5
    assert set(a.keys()) == set(b.keys()), set(a.keys()) ^ set(b.keys())
    assert "lnprob" not in a and "s2n" not in a
    assert isinstance(a["errmsg"], str) and a["errmsg"]
    assert a["pars"].shape == b["pars"].shape and a["pars_cov"].shape == b["pars_cov"].shape
    # a failing fit followed by a good one through the same Fitter object
    f = ngmix.fitting.Fitter(model=model, batched=True)
    good = f.go(obs=obs, guess=truth)
    assert good["flags"] == 0


def test_psf_mixtures_of_different_sizes_go_the_minpack_way():
    """two epochs whose psf mixtures have 1 and 3 gaussians: outside what the
    lock-step driver's rectangular psf table holds -- Fitter(batched=True) runs
    the fit through MINPACK and says so by leaving _batch_fitter unset"""
    rng = np.random.RandomState(21)
    o1, truth = _object("exp", rng, psf_ngauss=1)
    o2, _ = _object("exp", rng, psf_ngauss=3)
    ol = ngmix.ObsList()
    ol.append(o1)
    ol.append(o2)
    f = ngmix.fitting.Fitter(model="exp", batched=True)
    a = f.go(obs=ol, guess=truth)
    assert f._batch_fitter is None
    b = ngmix.fitting.Fitter(model="exp", batched=False).go(obs=ol, guess=truth)
    assert a["flags"] == b["flags"] == 0 and a["nfev"] == b["nfev"]
    np.testing.assert_array_equal(a["pars"], b["pars"])
    # the same sizes: the driver takes it
    ol2 = ngmix.ObsList()
    ol2.append(o1)
    ol2.append(_object("exp", rng, psf_ngauss=1)[0])
    f2 = ngmix.fitting.Fitter(model="exp", batched=True)
    c = f2.go(obs=ol2, guess=truth)
    assert f2._batch_fitter is not None and c["flags"] == 0


@pytest.fixture(params=["1", "0"], ids=["batched", "minpack"])
def route(request, monkeypatch):
    """the environment switch a user has: the default of Fitter(batched=None)"""
    monkeypatch.setenv("NGMIX_FITTER_BATCHED", request.param)
    return request.param == "1"


def test_runner_and_bootstrap_on_either_route(route):
    """runners.Runner / PSFRunner / bootstrap (runners.py:116-223,
    bootstrap.py:67-154) over Fitter and CoellipFitter built WITHOUT a batched
    argument, under either value of NGMIX_FITTER_BATCHED: psf fit stored in the
    psf observation, object fit retried until flags == 0, the truth recovered;
    the two routes agree to a small fraction of the errors"""
    results = {}
    rng = np.random.RandomState(31)
    obs, truth = _object("exp", rng)
    # the psf observation carries no mixture: the psf runner has to fit one
    obs.psf.set_gmix(None)
    assert not obs.psf.has_gmix()

    class Guesser(object):
        def __init__(self, pars, seed):
            self.pars, self.rng = np.asarray(pars), np.random.RandomState(seed)

        def __call__(self, obs):
            return self.pars * (1.0 + 0.01 * self.rng.uniform(-1, 1, size=self.pars.size))

    fitter = ngmix.fitting.Fitter(model="exp")
    psf_fitter = ngmix.fitting.CoellipFitter(ngauss=1)
    assert fitter.batched is route and psf_fitter.batched is route
    runner = ngmix.runners.Runner(fitter=fitter, guesser=Guesser(truth, 1), ntry=2)
    psf_runner = ngmix.runners.PSFRunner(
        fitter=psf_fitter, guesser=Guesser([0.0, 0.0, 0.01, -0.01, 0.27, 1.0], 2), ntry=2)
    res = ngmix.bootstrap.bootstrap(obs, runner, psf_runner=psf_runner)
    assert res["flags"] == 0
    assert (fitter._batch_fitter is not None) == route
    assert obs.psf.has_gmix() and obs.psf.meta["result"]["flags"] == 0
    np.testing.assert_allclose(obs.psf.gmix.get_T(), 0.27, rtol=2e-2)
    assert np.all(np.abs(res["pars"] - truth) <= 6.0 * res["pars_err"] + 1e-3)
    results[route] = res
    # the other route on the same data
    other = ngmix.fitting.Fitter(model="exp", batched=not route).go(
        obs=obs, guess=Guesser(truth, 1)(obs))
    assert other["flags"] == 0 and other["nfev"] == res["nfev"]
    assert np.all(np.abs(other["pars"] - res["pars"]) <= 1e-4 * res["pars_err"])


@pytest.mark.parametrize("model", ["exp", "turb", "bdf", "bd"])
def test_both_routes_with_a_host_joint_prior(model):
    """Fitter(prior=<joint_prior of priors.py terms>): the batched route puts
    the prior in its kernel form and keeps the fit inside the device loop; the
    MINPACK route calls prior.fill_fdiff on the host -- one result dict.  A
    prior object the kernel has no form for stays on the MINPACK route."""
    from ngmix_amd import priors, joint_prior
    rng = np.random.RandomState({"exp": 21, "turb": 22, "bdf": 23, "bd": 24}[model])
    obs, truth = _object(model, rng, noise=0.02)
    guess = truth * (1.0 + 0.02 * rng.uniform(-1, 1, size=truth.size))
    guess[0:2] = truth[0:2] + 0.01 * rng.uniform(-1, 1, size=2)
    prng = np.random.RandomState(3)
    cen = priors.CenPrior(0.0, 0.0, SCALE, SCALE, rng=prng)
    g = priors.GPriorBA(0.3, rng=prng)
    T = priors.LogNormal(0.6, 0.4, rng=prng)
    F = priors.TwoSidedErf(-10.0, 1.0, 1.0e4, 100.0, rng=prng)
    fd = priors.Normal(0.5, 0.2, rng=prng, bounds=(0.0, 1.0))
    if model == "bdf":
        prior = joint_prior.PriorBDFSep(cen, g, T, fd, F)
    elif model == "bd":
        prior = joint_prior.PriorBDSep(cen, g, T, priors.Normal(0.0, 0.3, rng=prng), fd, F)
    else:
        prior = joint_prior.PriorSimpleSep(cen, g, T, F)
    fb = ngmix.fitting.Fitter(model=model, prior=prior, batched=True)
    fm = ngmix.fitting.Fitter(model=model, prior=prior, batched=False)
    a = fb.go(obs=obs, guess=guess)
    b = fm.go(obs=obs, guess=guess)
    assert fb._batch_fitter is not None and fb._batch_fitter.prior_path == "kernel"
    assert fm._batch_fitter is None
    assert a["ier"] == b["ier"]
    if model == "exp":
        assert a["nfev"] == b["nfev"]
        _compare(a, b, 1e-4, 1e-4)
    else:
        _compare(a, b, 5e-2 if model == "bd" else 2e-2, 5e-2 if model == "bd" else 2e-2)
    # the prior's rows are not part of chi2 / dof on either route
    assert a["dof"] == a["npix"] - truth.size

    class Mine(priors.LogNormal):
        pass
    other = joint_prior.PriorSimpleSep(cen, g, Mine(0.6, 0.4, rng=prng), F)
    if model == "exp":
        f = ngmix.fitting.Fitter(model=model, prior=other, batched=True)
        c = f.go(obs=obs, guess=guess)
        assert f._batch_fitter is None and c["flags"] == 0
        np.testing.assert_allclose(c["pars"], b["pars"], rtol=1e-9, atol=1e-12)
