"""
ngmix_amd.priors / ngmix_amd.joint_prior against the reference:
tests/golden/priors.npz holds what the REFERENCE's priors returned for the
script of calls in tests/helpers/prior_cases.py (oracle/gen_golden_priors.py
ran it); the same script on ngmix_amd's classes must give the same numbers --
densities and residuals to the bit, the same exception classes, the same
seeded draws and the same generator state afterwards.
"""
import os

import numpy as np
import pytest

import ngmix_amd as ngmix
from helpers import prior_cases

GOLD = os.path.join(os.path.dirname(__file__), "golden", "priors.npz")


@pytest.fixture(scope="module")
def both():
    want = dict(np.load(GOLD, allow_pickle=False))
    got = prior_cases.run(ngmix.priors, ngmix.joint_prior, guessers=ngmix.guessers)
    return want, got


def test_same_entries(both):
    want, got = both
    assert set(want) == set(got), sorted(set(want) ^ set(got))[:20]
    assert len(want) > 900


def test_every_entry_is_the_references(both):
    want, got = both
    bad = []
    for k in sorted(want):
        w, g = want[k], got[k]
        if w.dtype.kind == "U" or g.dtype.kind == "U":
            if str(w) != str(g):
                bad.append((k, str(w), str(g)))
        elif w.shape != g.shape or not np.array_equal(w, g, equal_nan=True):
            bad.append((k, w, g))
    assert not bad, (len(bad), bad[:8])


def test_fitmodel_rows_are_the_joint_priors():
    """the seam FitModel uses: fill_fdiff at the head of the residual vector,
    bounds for leastsqbound, GMixRangeError out of the shape prior"""
    rng = np.random.RandomState(5)
    p = ngmix.joint_prior.PriorSimpleSep(
        ngmix.priors.CenPrior(0.0, 0.0, 0.1, 0.1, rng=rng), ngmix.priors.GPriorBA(0.3, rng=rng),
        ngmix.priors.Normal(0.5, 0.2, rng=rng, bounds=(0.01, 5.0)),
        [ngmix.priors.FlatPrior(-10.0, 1e6, rng=rng)] * 2)
    assert p.nband == 2 and len(p.bounds) == 7 and p.bounds[4] == (0.01, 5.0)
    assert p.bounds[5] == (None, None)
    fdiff = np.zeros(20)
    pars = np.array([0.1, -0.1, 0.3, 0.0, 0.7, 10.0, 20.0])
    assert p.fill_fdiff(pars, fdiff) == 6
    np.testing.assert_allclose(fdiff[:2], [1.0, 1.0], rtol=1e-14)
    assert fdiff[3] == pytest.approx(1.0) and fdiff[4] == 0.0 and fdiff[5] == 0.0
    assert np.isclose(-0.5 * (fdiff[:6] ** 2).sum(), p.get_lnprob_scalar(pars))
    with pytest.raises(ngmix.GMixRangeError):
        p.fill_fdiff(np.array([0, 0, 0.8, 0.8, 0.7, 1.0, 1.0]), fdiff)
    s = p.sample(100)
    assert s.shape == (100, 7) and np.all(np.hypot(s[:, 2], s[:, 3]) < 1.0)


def _joint_cases():
    from ngmix_amd import priors, joint_prior

    def rs(k):
        return np.random.RandomState(900 + k)
    cen = priors.CenPrior(0.01, -0.02, 0.05, 0.07, rng=rs(0))
    g = priors.GPriorBA(0.25, rng=rs(1))
    T_erf = priors.TwoSidedErf(-0.05, 0.03, 3.0, 0.3, rng=rs(2))
    T_ln = priors.LogNormal(0.6, 0.3, rng=rs(3))
    T_lns = priors.LogNormal(0.6, 0.3, rng=rs(3), shift=-0.1)
    T_tg = priors.TruncatedGaussian(0.5, 0.3, 0.05, 2.0, rng=rs(4))
    T_nb = priors.Normal(0.5, 0.3, rng=rs(5), bounds=(0.01, 4.0))
    F = [priors.TwoSidedErf(-1.0, 0.5, 500.0, 20.0, rng=rs(6)),
         priors.FlatPrior(-10.0, 400.0, rng=rs(7))]
    fracdev = priors.Normal(0.5, 0.1, rng=rs(8), bounds=(0.0, 1.0))
    fd_tg = priors.TruncatedGaussian(0.5, 0.1, 0.0, 1.0, rng=rs(9))
    ltr = priors.Normal(0.0, 0.5, rng=rs(10))
    return [
        ("simple_erf", joint_prior.PriorSimpleSep(cen, g, T_erf, F), 7, True),
        ("simple_lognormal", joint_prior.PriorSimpleSep(cen, g, T_ln, F[0]), 6, False),
        ("simple_lognormal_shift", joint_prior.PriorSimpleSep(cen, g, T_lns, F), 7, False),
        ("simple_truncgauss", joint_prior.PriorGalsimSimpleSep(cen, g, T_tg, F[1]), 6, False),
        ("bdf", joint_prior.PriorBDFSep(cen, g, T_nb, fracdev, F), 8, False),
        ("bdf_tg", joint_prior.PriorBDFSep(cen, g, T_ln, fd_tg, F[0]), 7, False),
        ("bd", joint_prior.PriorBDSep(cen, g, T_erf, ltr, fracdev, F), 9, False),
    ]


@pytest.mark.parametrize("case", range(7))
def test_batch_form_of_a_host_joint_prior(case):
    """prior_batch.as_batch_prior: the rows, range errors, ln p and bounds of
    the batch form (torch, here on the CPU) against the host joint prior it was
    made from, point by point -- including points outside every term's range"""
    import torch
    from ngmix_amd import prior_batch as pb
    tag, jp, npars, kernel = _joint_cases()[case]
    bp = pb.as_batch_prior(jp)
    assert isinstance(bp, pb.PriorSepBatch) and not isinstance(bp, pb.PriorBatchAdapter)
    assert bp.bounds == jp.bounds
    # every one of them has a kernel form; ngmix_simple_sep_prior_eval is the
    # code that kernel runs, on the host
    from ngmix_amd import _lib
    desc = bp.descriptor()
    assert desc is not None
    L = _lib.lib()
    rng = np.random.RandomState(77 + case)
    pts = np.zeros((400, npars))
    pts[:, 0:2] = rng.normal(scale=0.05, size=(400, 2))
    pts[:, 2:4] = rng.normal(scale=0.35, size=(400, 2))
    pts[:, 4] = rng.uniform(-0.3, 2.5, size=400)
    pts[:, 5:] = rng.uniform(-0.3, 1.4, size=(400, npars - 5))
    pts[:, -1] = rng.uniform(-30.0, 450.0, size=400)
    rows, bad = bp.fill_fdiff_batch(torch.from_numpy(pts))
    lnp = bp.get_lnprob_batch(torch.from_numpy(pts)).numpy()
    rows, bad = rows.numpy(), bad.numpy()
    nbad = 0
    krows, klnp = np.zeros(12), np.zeros(1)
    for i, p in enumerate(pts):
        buf = np.zeros(npars + 2)
        k = L.ngmix_simple_sep_prior_eval(_lib.ptr(desc), _lib.ptr(p.copy()), _lib.ptr(krows),
                                          _lib.ptr(klnp))
        try:
            n = jp.fill_fdiff(p, buf)
        except ngmix.GMixRangeError:
            assert bad[i], (tag, i)
            assert lnp[i] == -np.inf
            assert k == -1
            nbad += 1
            continue
        assert not bad[i], (tag, i, p)
        assert k == n
        kfin = np.isfinite(buf[:n])
        np.testing.assert_array_equal(np.isfinite(krows[:n]), kfin)
        np.testing.assert_allclose(krows[:n][kfin], buf[:n][kfin], rtol=4e-15, atol=1e-300)
        if kfin.all():
            np.testing.assert_allclose(klnp[0], jp.get_lnprob_scalar(p), rtol=1e-14, atol=1e-15)
        assert n == rows.shape[1]
        fin = np.isfinite(buf[:n])
        np.testing.assert_array_equal(np.isfinite(rows[i]), fin)
        np.testing.assert_allclose(rows[i][fin], buf[:n][fin], rtol=1e-12, atol=1e-14)
        want = jp.get_lnprob_scalar(p)
        if np.isfinite(want):
            np.testing.assert_allclose(lnp[i], want, rtol=1e-12, atol=1e-13)
        else:
            assert lnp[i] == want
    assert 20 < nbad < 380


@pytest.mark.parametrize("model", ["exp", "bdf", "bd"])
def test_pipeline_prior_guess_is_the_prior_drawing_psf_flux_guessers(model):
    """bootstrap_batch(guesser='psfflux', prior=<host joint prior>) builds its
    guesses as TPSFFluxAndPriorGuesser / BDFPSFFluxGuesser do (whose draws are
    pinned to the reference's above): for one object and the same generator
    states the two give the same vector; for many objects every guess is one
    the prior accepts"""
    from ngmix_amd import priors, joint_prior, pipeline, guessers

    def make(seed, tight=False):
        rs = [np.random.RandomState(seed + k) for k in range(8)]
        cen = priors.CenPrior(0.0, 0.0, 0.05, 0.05, rng=rs[0])
        g = priors.GPriorBA(0.2, rng=rs[1])
        T = priors.FlatPrior(0.47, 0.53, rng=rs[2]) if tight else \
            priors.TwoSidedErf(-0.05, 0.03, 3.0, 0.3, rng=rs[2])
        F = [priors.TwoSidedErf(-1.0, 0.5, 500.0, 20.0, rng=rs[3 + b]) for b in range(2)]
        fd = priors.Normal(0.5, 0.1, rng=rs[5], bounds=(0.0, 1.0))
        if model == "exp":
            return joint_prior.PriorSimpleSep(cen, g, T, F)
        if model == "bdf":
            return joint_prior.PriorBDFSep(cen, g, T, fd, F)
        return joint_prior.PriorBDSep(cen, g, T, priors.Normal(0.0, 0.5, rng=rs[6]), fd, F)

    flux = np.array([[120.0, 80.0]])
    nshape = pipeline.MODEL_NLOC[model] - 1
    for tight in (False, True):
        for seed in (10, 20, 30):
            if model == "exp":
                ref = guessers.TPSFFluxAndPriorGuesser(np.random.RandomState(seed + 99), 0.5,
                                                       make(seed, tight))
            else:
                ref = guessers.BDFPSFFluxGuesser(0.5, make(seed, tight))
                if model == "bd":
                    ref.first_flux = 7
            ref._get_psf_fluxes = lambda obs: flux[0]
            want = ref(obs=None, nrand=1)
            got = pipeline._psfflux_guess(model, 1, 2, 0.5, flux,
                                          np.random.RandomState(seed + 99), make(seed, tight))
            np.testing.assert_array_equal(got[0], want)
    prior = make(77, tight=True)
    many = pipeline._psfflux_guess(model, 500, 2, 0.5, np.tile(flux, (500, 1)),
                                   np.random.RandomState(5), prior)
    assert many.shape == (500, nshape + 2)
    assert all(np.isfinite(prior.get_lnprob_scalar(p)) for p in many)
    assert np.abs(many[:, 0]).max() > 0.05        # centres from the prior, not +-0.01
    with pytest.raises(ValueError):
        pipeline._psfflux_guess(model, 3, 3, 0.5, np.ones((3, 3)), np.random.RandomState(5), prior)


@pytest.mark.parametrize("case", range(7))
@pytest.mark.parametrize("mode", ["analytic", "fd"])
def test_prior_normal_sums_are_those_of_the_full_difference_jacobian(case, mode):
    """the prior kernel evaluates, per parameter, only the row that parameter
    belongs to; its sums [J^T J | J^T r | r.r] must be those of the full
    (rows x parameters) one-sided difference jacobian -- n + 1 evaluations of
    every row -- to the BIT.  ngmix_lm_prior_sums_host is the kernel's code on
    the host; the full jacobian is built here from ngmix_simple_sep_prior_eval
    rows with the reference's step rule (forward, backward where the forward
    point is out of range; lmdif: the state's own points)."""
    from ngmix_amd import _lib, prior_batch as pb
    tag, jp, npars, _ = _joint_cases()[case]
    desc = pb.as_batch_prior(jp).descriptor()
    L = _lib.lib()
    rng = np.random.RandomState(300 + case)
    nfit = 60
    st = np.zeros(nfit, dtype=_lib.LM_STATE_DTYPE)
    st["n"] = npars
    st["mode"] = _lib.LM_MODE_FD if mode == "fd" else 0
    x = np.zeros((nfit, npars))
    x[:, 0:2] = rng.normal(scale=0.05, size=(nfit, 2))
    x[:, 2:4] = rng.normal(scale=0.3, size=(nfit, 2))
    x[:, 4] = rng.uniform(0.01, 2.2, size=nfit)
    x[:, 5:] = rng.uniform(0.02, 0.98, size=(nfit, npars - 5))
    x[:, -1] = rng.uniform(-5.0, 420.0, size=nfit)
    x[5, 2:4] = (0.70710678, 0.70710677)      # the forward step leaves |g| < 1
    x[6, 4] = 1.9999999999 if "truncgauss" in tag else x[6, 4]
    st["xt"][:, :npars] = x
    h = 1.0e-8 * np.maximum(1.0, np.abs(x)) * rng.choice([1.0, -1.0], size=x.shape)
    st["hstep"][:, :npars] = h
    st["xstep"][:, :npars] = x + h
    nsum = npars * (npars + 1) // 2 + npars + 1
    got = np.full((nfit, nsum), 7.0)
    assert L.ngmix_lm_prior_sums_host(_lib.ptr(st), nfit, _lib.ptr(desc), 1.0e-8,
                                      _lib.ptr(got)) == 0

    def rows_at(p):
        r = np.zeros(12)
        k = L.ngmix_simple_sep_prior_eval(_lib.ptr(desc), _lib.ptr(np.ascontiguousarray(p)),
                                          _lib.ptr(r), None)
        return (None if k < 0 else r[:k].copy())
    nin = 0
    for f in range(nfit):
        r0 = rows_at(x[f])
        want = np.zeros(nsum)
        if r0 is None:
            want[-1] = np.inf
            np.testing.assert_array_equal(got[f], want)
            continue
        nin += 1
        k = r0.size
        J = np.zeros((k, npars))
        for j in range(npars):
            p = x[f].copy()
            if mode == "fd":
                step = h[f, j]
                p[j] = st["xstep"][f, j]
                rj = rows_at(p)
            else:
                step = 1.0e-8 * max(1.0, abs(x[f, j]))
                p[j] = x[f, j] + step
                rj = rows_at(p)
                if rj is None:
                    step = -step
                    p[j] = x[f, j] + step
                    rj = rows_at(p)
            for i in range(k):
                good = rj is not None and np.isfinite(r0[i]) and np.isfinite(rj[i])
                J[i, j] = (rj[i] - r0[i]) / step if good else 0.0
        t = 0
        for a in range(npars):
            for b in range(a, npars):
                acc = 0.0
                for i in range(k):
                    acc += J[i, a] * J[i, b]
                want[t] = acc
                t += 1
            acc = 0.0
            for i in range(k):
                if np.isfinite(r0[i]):
                    acc += J[i, a] * r0[i]
            want[npars * (npars + 1) // 2 + a] = acc
        ff = 0.0
        for i in range(k):
            ff += r0[i] * r0[i]
        want[-1] = ff
        np.testing.assert_array_equal(got[f], want, err_msg="%s fit %d" % (tag, f))
    assert nin >= 30


@pytest.mark.parametrize("ndim,ngauss", [(1, 1), (1, 3), (2, 2), (3, 4)])
def test_gmixnd_vs_reference(ndim, ngauss):
    """ngmix_amd.GMixND against the reference's (tests/golden/gmixnd.npz,
    oracle/gen_golden_gmixnd.py): norms and inverse covariances exact;
    densities (array / scalar / one component, ln and linear) to 1e-13 -- one
    vectorised pass here, the reference's loop per point there, the same sums
    in the same order; seeded samples through sklearn's sampler exact"""
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "gmixnd.npz")))
    tag = "d%dg%d" % (ndim, ngauss)
    w, m, c = g[tag + "_w"], g[tag + "_m"], g[tag + "_c"]
    if ndim == 1:
        gm = ngmix.GMixND(w, m[:, 0], c[:, 0, 0], rng=np.random.RandomState(5))
    else:
        gm = ngmix.GMixND(w, m, c, rng=np.random.RandomState(5))
    assert gm.ndim == ndim and gm.ngauss == ngauss
    for name in ("norms", "log_pnorms", "icovars"):
        np.testing.assert_array_equal(getattr(gm, name), g["%s_%s" % (tag, name)])
    pts = g[tag + "_pts"]
    arg = pts[:, 0] if ndim == 1 else pts
    tol = dict(rtol=1e-13, atol=1e-300)
    np.testing.assert_allclose(gm.get_lnprob_array(arg), g[tag + "_lnp"], **tol)
    np.testing.assert_allclose(gm.get_prob_array(arg), g[tag + "_p"], **tol)
    np.testing.assert_allclose([gm.get_lnprob_scalar(p) for p in pts], g[tag + "_lnp_scalar"],
                               **tol)
    np.testing.assert_allclose([gm.get_prob_scalar(p) for p in pts], g[tag + "_p_scalar"], **tol)
    k = ngauss - 1
    np.testing.assert_allclose(gm.get_lnprob_array(arg, component=k), g[tag + "_lnp_comp"], **tol)
    np.testing.assert_allclose(gm.get_prob_array(arg, component=k), g[tag + "_p_comp"], **tol)
    np.testing.assert_array_equal(np.atleast_1d(gm.sample()), g[tag + "_sample_one"])
    np.testing.assert_array_equal(gm.sample(7), g[tag + "_sample_7"])
    with pytest.raises(AssertionError):
        gm.get_lnprob_scalar(pts[0], component=ngauss)


def test_gmixnd_fit_and_arguments():
    import contextlib
    import io
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "gmixnd.npz")))
    gm = ngmix.GMixND(rng=np.random.RandomState(77))
    with contextlib.redirect_stdout(io.StringIO()):
        gm.fit(g["fit_data"], 2, n_iter=500)
    assert gm.converged == bool(g["fit_converged"])
    np.testing.assert_allclose(gm.weights, g["fit_w"], rtol=1e-10)
    np.testing.assert_allclose(gm.means, g["fit_m"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(gm.covars, g["fit_c"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(gm.sample(5), g["fit_sample"], rtol=1e-9, atol=1e-10)
    with pytest.raises(RuntimeError):
        ngmix.GMixND(weights=[1.0])
    assert ngmix.gmix_ndim.GMixND is ngmix.GMixND
