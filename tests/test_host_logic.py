"""
CPU tests of host orchestration that involves no pixel arithmetic: runners and
bootstrap control flow with stub fitters (mirrors the behaviours pinned by the
reference's test_runners.py / test_bootstrap.py), Observation container
semantics, flags, scalar shape / moment helpers and admom.get_result branches.
"""
import numpy as np
import pytest

import ngmix_amd as ngmix
from ngmix_amd import flags
from ngmix_amd.admom import get_result
from ngmix_amd.bootstrap import bootstrap, remove_failed_psf_obs
from ngmix_amd.runners import Runner, PSFRunner, run_fitter


class StubResult(dict):
    def get_gmix(self):
        return ngmix.GMix(pars=[1.0, 0.0, 0.0, 1.0, 0.0, 1.0])


class StubFitter(object):
    """returns the scripted flags in order"""

    def __init__(self, flag_sequence):
        self.seq = list(flag_sequence)
        self.calls = []

    def go(self, obs, guess=None):
        self.calls.append((obs, guess))
        return StubResult(flags=self.seq.pop(0))


def _obs(with_psf=True):
    # store_pixels=False: no device work in this CPU-only module
    psf = ngmix.Observation(np.ones((5, 5)), store_pixels=False) if with_psf else None
    return ngmix.Observation(np.ones((8, 8)), psf=psf, store_pixels=False)


def test_run_fitter_retries_until_success():
    f = StubFitter([4, 2, 0, 0])
    res = run_fitter(_obs(), f, guesser=lambda obs: np.zeros(6), ntry=5)
    assert res["flags"] == 0 and len(f.calls) == 3
    f = StubFitter([4, 2])
    res = Runner(f, guesser=lambda obs: 1.0, ntry=2).go(_obs())
    assert res["flags"] == 2 and len(f.calls) == 2
    f = StubFitter([0])
    run_fitter(_obs(), f)            # no guesser: go(obs=obs) only
    assert f.calls[0][1] is None


def test_psf_runner_sets_result_and_gmix():
    mb = ngmix.MultiBandObsList()
    for _ in range(2):
        ol = ngmix.ObsList()
        ol.append(_obs())
        ol.append(_obs())
        mb.append(ol)
    f = StubFitter([0, 8, 0, 0])
    res = PSFRunner(f, guesser=lambda obs: 1.0).go(mb)
    assert [[r["flags"] for r in rl] for rl in res] == [[0, 8], [0, 0]]
    assert mb[0][0].psf.has_gmix() and not mb[0][1].psf.has_gmix()
    assert mb[0][1].psf.meta["result"]["flags"] == 8
    # fits the observation itself when it has no psf
    o = _obs(with_psf=False)
    PSFRunner(StubFitter([0])).go(o)
    assert o.has_gmix() and o.meta["result"]["flags"] == 0
    with pytest.raises(ValueError):
        PSFRunner(StubFitter([0])).go("not an obs")


def test_bootstrap_drops_failed_psf_epochs():
    mb = ngmix.MultiBandObsList()
    ol = ngmix.ObsList()
    for _ in range(3):
        ol.append(_obs())
    mb.append(ol)
    psf_runner = PSFRunner(StubFitter([0, 16, 0]))
    obj = StubFitter([0])
    res = bootstrap(mb, Runner(obj), psf_runner=psf_runner)
    assert res["flags"] == 0
    fitted = obj.calls[0][0]
    assert isinstance(fitted, ngmix.MultiBandObsList) and len(fitted[0]) == 2
    # all psf fits of a band failed
    ol2 = ngmix.ObsList()
    ol2.append(_obs())
    PSFRunner(StubFitter([32])).go(ol2)
    with pytest.raises(ngmix.BootPSFFailure):
        remove_failed_psf_obs(ol2)
    with pytest.raises(ValueError):
        remove_failed_psf_obs(3)
    # ignore_failed_psf=False keeps everything
    obj = StubFitter([0])
    bootstrap(ol2, Runner(obj), psf_runner=PSFRunner(StubFitter([32])),
              ignore_failed_psf=False)
    assert len(obj.calls[0][0]) == 1


def test_observation_containers():
    o = _obs()
    assert o.has_psf() and not o.has_gmix() and not o.has_bmask()
    assert o.pixels is None            # store_pixels=False
    o.set_gmix(ngmix.GMix(pars=[1.0, 0.0, 0.0, 1.0, 0.0, 1.0]))
    g1 = o.gmix
    g1.get_data()["p"] = 5.0           # a copy: the stored gmix is untouched
    assert o.gmix.get_flux() == 1.0
    with pytest.raises(AssertionError):
        o.set_image(np.ones((3, 3)))   # shape must not change
    with pytest.raises(TypeError):
        o.meta = 3
    o2 = o.copy()
    assert o2 == o
    o2.meta["x"] = 1
    assert not (o2 == o)
    with pytest.raises(ValueError):
        o == 3
    ol = ngmix.ObsList()
    with pytest.raises(AssertionError):
        ol.append(3)
    ol.append(o)
    assert ngmix.get_mb_obs(o)[0][0] is o and ngmix.get_mb_obs(ol)[0] is ol
    Isum, Vsum, Npix = ngmix.get_mb_obs(ol).get_s2n_sums()
    assert (Isum, Vsum, Npix) == (64.0, 64.0, 64)
    assert ol.get_s2n() == 8.0
    with pytest.raises(ValueError):
        ngmix.get_mb_obs(3)
    j = o.jacobian
    with pytest.raises(ValueError):
        j._data["row0"] = 3.0          # read-only view of the obs's jacobian


def test_flags_and_helpers():
    assert flags.get_flags_str(0) == ""
    assert flags.get_flags_str(flags.MAXITER | flags.LOW_DET) == \
        "determinant near zero|max iterations reached"
    assert flags.get_flags_str(2 ** 20) == "bit 2**20"
    with pytest.raises(ValueError):
        flags.get_flags_str(-1)
    assert flags.EM_RANGE_ERROR == 2 ** 7 and flags.EM_MAXITER == 2 ** 5
    e1, e2 = ngmix.shape.g1g2_to_e1e2(0.2, -0.1)
    g1, g2 = ngmix.shape.e1e2_to_g1g2(e1, e2)
    np.testing.assert_allclose([g1, g2], [0.2, -0.1], rtol=1e-14)
    with pytest.raises(ngmix.GMixRangeError):
        ngmix.shape.g1g2_to_e1e2(0.9, 0.9)
    T = ngmix.moments.fwhm_to_T(1.2)
    np.testing.assert_allclose(ngmix.moments.T_to_fwhm(T), 1.2, rtol=1e-14)
    irr, irc, icc = ngmix.moments.get_sheared_moments(0.3, 0.02, 0.35, 0.0, 0.0)
    np.testing.assert_allclose([irr, irc, icc], [0.3, 0.02, 0.35], rtol=1e-12)
    with pytest.raises(ValueError):
        ngmix.GMix()
    with pytest.raises(ValueError):
        ngmix.GMix(pars=[1.0] * 5)
    with pytest.raises(ValueError):
        ngmix.GMixModel([1.0] * 5, "exp")
    with pytest.raises(ValueError):
        ngmix.GMixModel([1.0] * 6, "nomodel")
    with pytest.raises(TypeError):
        ngmix.GMixModel([0, 0, 0, 0, 1.0, 1.0], "gauss").convolve(3)
    gm = ngmix.GMixModel([0.1, 0.2, 0.0, 0.0, 1.0, 2.0], "exp")
    c = gm.copy()
    assert c == gm and len(c) == 6
    gm.set_flux(4.0)
    assert gm.copy().get_flux() != 4.0   # GMixModel.copy rebuilds from _pars
    gm.set_cen(1.0, 2.0)
    np.testing.assert_allclose(gm.copy().get_cen(), (1.0, 2.0))
    with pytest.raises(ngmix.GMixRangeError):
        ngmix.GMixModel([0, 0, 0.9, 0.9, 1.0, 1.0], "gauss")
    g0 = ngmix.GMix(ngauss=2)
    with pytest.raises(ngmix.GMixRangeError):
        g0.set_norms()
    cm = ngmix.GMixCM(0.3, 1.7, [0.1, -0.05, 0.1, 0.05, 0.6, 100.0])
    assert len(cm) == 16 and cm.copy() == cm


def test_admom_get_result_branches():
    """hand-built records, as the reference's test_admom.py:109-120"""
    ares = np.zeros(1, dtype=ngmix._lib.ADMOM_RESULT_DTYPE)
    ares["sums_cov"] = np.nan
    res = get_result(ares, 1.0, 1.0)
    assert res["flags"] & flags.NONPOS_VAR
    assert np.isnan(res["e1"]) and res["flagstr"] != ""
    ares = np.zeros(1, dtype=ngmix._lib.ADMOM_RESULT_DTYPE)
    ares["sums_cov"][0] = np.eye(7)
    ares["sums"][0][5] = 1.0
    ares["wsum"] = 1.0
    ares["pars"][0][4] = -1.0          # negative T
    res = get_result(ares, 1.0, 1.0)
    assert res["flags"] == flags.NONPOS_SIZE
    assert res["flux_flags"] == flags.NONPOS_SIZE
    ares["flags"] = flags.MAXITER
    res = get_result(ares, 1.0, 1.0)
    assert res["flux_flags"] == flags.MAXITER and res["T_flags"] == flags.MAXITER


def test_fitting_host_pieces():
    from ngmix_amd.fitting import (get_band_pars, get_lm_n_prior_pars,
                                   _BoundsTransform, leastsqbound)
    pars = np.arange(8.0)
    np.testing.assert_array_equal(get_band_pars("exp", pars, 2),
                                  [0, 1, 2, 3, 4, 7])
    np.testing.assert_array_equal(get_band_pars("bdf", pars, 1),
                                  [0, 1, 2, 3, 4, 5, 7])
    assert get_lm_n_prior_pars("exp", 3) == 8 and get_lm_n_prior_pars("bd", 1) == 7
    with pytest.raises(ValueError):
        get_lm_n_prior_pars("coellip", 1)
    tr = _BoundsTransform([(None, None), (0.0, None), (None, 2.0), (-1.0, 1.0)])
    x = np.array([0.3, 0.5, 1.5, 0.25])
    np.testing.assert_allclose(tr.i2e(tr.e2i(x)), x, rtol=1e-13)
    # bounded fit of a line: slope bounded to [0, 1.5] while the truth is 2
    t = np.linspace(0, 1, 20)
    y = 2.0 * t + 0.1

    def func(p):
        return p[0] * t + p[1] - y

    xfit, cov, info, mesg, ier = leastsqbound(func, np.array([1.0, 0.0]),
                                              bounds=[(0.0, 1.5), (None, None)],
                                              full_output=1)
    assert ier in (1, 2, 3, 4) and xfit[0] <= 1.5 + 1e-12
    xfree, _ = leastsqbound(func, np.array([1.0, 0.0]))
    np.testing.assert_allclose(xfree, [2.0, 0.1], atol=1e-8)


def test_run_leastsq_error_mapping():
    """run_leastsq's exception handling (leastsqbound.py:33-155, pinned by the
    reference's test_leastsqbound.py:187-208): a ValueError mentioning NaNs or
    infs -> LM_FUNC_NOTFINITE, ZeroDivisionError -> DIV_ZERO, any other
    ValueError propagates; maxfev exhaustion -> flags 2**(ier-5) with default
    pars; zero degrees of freedom -> ZERO_DOF"""
    from ngmix_amd.fitting import run_leastsq
    from ngmix_amd import flags as fl
    from ngmix_amd.defaults import PDEF, CDEF
    t = np.linspace(0, 1, 12)
    y = 3.0 * t + 0.5

    def good(p):
        return p[0] * t + p[1] - y

    res = run_leastsq(good, np.array([1.0, 0.0]), 0)
    assert res["flags"] == 0 and res["nfev"] > 0
    np.testing.assert_allclose(res["pars"], [3.0, 0.5], atol=1e-8)
    assert res["pars_cov"].shape == (2, 2) and np.all(np.isfinite(res["pars_err"]))

    def nan_error(p):
        raise ValueError("array must not contain infs or NaNs")

    res = run_leastsq(nan_error, np.array([1.0, 0.0]), 0)
    assert res["flags"] == fl.LM_FUNC_NOTFINITE and res["nfev"] == -1
    assert np.all(res["pars"] == PDEF) and np.all(res["pars_cov"] == CDEF)

    def zero_div(p):
        raise ZeroDivisionError("division by zero")

    res = run_leastsq(zero_div, np.array([1.0, 0.0]), 0)
    assert res["flags"] == fl.DIV_ZERO and res["nfev"] == -1

    def other(p):
        raise ValueError("something else")

    with pytest.raises(ValueError):
        run_leastsq(other, np.array([1.0, 0.0]), 0)

    # maxfev exhausted: ier 5 -> flags 2**0, default pars
    def rosen(p):
        return np.array([10.0 * (p[1] - p[0] ** 2), 1.0 - p[0], 0.0])

    res = run_leastsq(rosen, np.array([-1.2, 1.0]), 0, maxfev=4)
    assert res["ier"] == 5 and res["flags"] == 2 ** 0
    assert np.all(res["pars"] == PDEF)

    # as many residuals as parameters: zero degrees of freedom
    def square(p):
        return np.array([p[0] - 1.0, p[1] + 2.0])

    res = run_leastsq(square, np.array([0.0, 0.0]), 0)
    assert res["flags"] & fl.ZERO_DOF


# ---------------------------------------------------------------------------
# host functions of the LM shell against the reference's own outputs
# (tests/golden/api2.npz, oracle/gen_golden_r2.py)

@pytest.mark.parametrize("model", ["exp", "dev", "gauss"])
@pytest.mark.parametrize("psf_tag", ["nopsf", "psf3"])
def test_get_model_deriv_data_vs_reference(golden, model, psf_tag):
    """results.py:955-1010: composed gaussians and d(cov)/d(g1, g2, T)"""
    from ngmix_amd.fitting import get_model_deriv_data
    g = golden("api2")
    api = golden("api")
    pre = "dd_%s_%s_" % (model, psf_tag)
    pars = g[pre + "pars"]
    gm0 = ngmix.GMixModel(pars, model)
    gmc = gm0
    if psf_tag == "psf3":
        gmc = gm0.convolve(ngmix.GMix(pars=api["lm_b0_e0_psf_gmix_pars"]))
    gpars, dcov = get_model_deriv_data(gm0, gmc, pars[2], pars[3], pars[4])
    assert gpars.shape == g[pre + "gpars"].shape and dcov.shape == g[pre + "dcov"].shape
    # the model fill goes through tanh / atanh (libm ulps); the derivative
    # algebra on top of it is the reference's to the last bit
    np.testing.assert_allclose(gpars, g[pre + "gpars"], rtol=1e-14, atol=1e-16)
    np.testing.assert_allclose(dcov, g[pre + "dcov"], rtol=1e-14, atol=1e-16)
    # exactness of the algebra itself: feed it the reference's composed pars
    class Fixed(object):
        def __init__(self, full):
            self._full = full

        def get_full_pars(self):
            return self._full.ravel().copy()
    ng0 = len(gm0)
    ref_g = g[pre + "gpars"]
    npsf = ref_g.shape[0] // ng0
    # model covariances = composed - psf part; recover them from the golden dcov
    modcov = g[pre + "dcov"][::npsf, 2, :] * pars[4]
    m0 = np.zeros((ng0, 6))
    m0[:, 3:6] = modcov
    _, dc = get_model_deriv_data(Fixed(m0), Fixed(ref_g), pars[2], pars[3], pars[4])
    np.testing.assert_allclose(dc, g[pre + "dcov"], rtol=4e-16, atol=0)


def test_prior_jacobian_rows_vs_reference(golden):
    """results.py:572-625 (one-sided differences of prior.fill_fdiff) against
    the prior rows of the reference's calc_jacobian, bit for bit: the prior
    (tests/helpers/rows_prior.py) is the same code on both sides"""
    from helpers.rows_prior import RowsPrior
    from ngmix_amd.fitting import FitModel, get_lm_n_prior_pars
    g = golden("api2")
    api = golden("api")

    class Shell(object):
        _prior_rows = FitModel._prior_rows
        _fill_prior_jacobian = FitModel._fill_prior_jacobian

    fm = Shell()
    fm.prior = RowsPrior(2, ngmix.GMixRangeError)
    fm.npars = 7
    fm.n_prior_pars = get_lm_n_prior_pars("exp", 2)
    assert fm.n_prior_pars == int(g["lmp_n_prior_pars"]) == 7
    for tag in ("guess", "truth"):
        pars = api["lm_" + tag]
        jac = np.full((10, 7), 99.0)
        n = fm._fill_prior_jacobian(pars, jac)
        assert n == 6
        np.testing.assert_array_equal(jac[:6], g["lmp_jac_" + tag][:6])
        assert np.all(jac[6:] == 99.0)
    # a prior undefined at the point: the error propagates (calc_jacobian
    # turns it into a zero jacobian)
    with pytest.raises(ngmix.GMixRangeError):
        fm._fill_prior_jacobian(api["lm_bad_pars"], np.zeros((10, 7)))
    # forward step outside the domain -> backward difference
    class Wall(object):
        def fill_fdiff(self, p, f):
            if p[4] > 1.0:
                raise ngmix.GMixRangeError("beyond the wall")
            f[0] = p[4] ** 2
            return 1
    fm.prior = Wall()
    pars = np.array([0.0, 0.0, 0.0, 0.0, 1.0, 1.0, 1.0])
    jac = np.zeros((7, 7))
    assert fm._fill_prior_jacobian(pars, jac) == 1
    np.testing.assert_allclose(jac[0, 4], 2.0, rtol=1e-7)
    assert np.all(jac[0, [0, 1, 2, 3, 5, 6]] == 0.0)
    # a row that is not finite gets zeros
    class Inf(object):
        def fill_fdiff(self, p, f):
            f[0] = np.inf
            f[1] = p[0]
            return 2
    fm.prior = Inf()
    jac = np.zeros((7, 7))
    assert fm._fill_prior_jacobian(pars, jac) == 2
    assert np.all(jac[0] == 0.0) and jac[1, 0] == pytest.approx(1.0)


def test_noise_cov_steps():
    """results.py:929-952"""
    from ngmix_amd.noise_cov import get_step
    pars = np.array([0.1, -0.2, 0.3, 0.05, 0.8, 0.4, 2.0e3, 1.0e-9])
    got = [get_step(pars, i, 2) for i in range(pars.size)]
    assert got == [1e-3, 1e-3, 1e-4, 1e-4, 8e-4, 4e-4, 2.0, 1e-6]
    assert get_step(np.array([0, 0, 0, 0, 1e-3, 5.0]), 4, 1) == 1e-4


@pytest.mark.parametrize("nm", [6, 17])
def test_make_mom_result_batch_equals_the_per_object_routine(nm):
    """moments.make_mom_result_batch (one vectorised pass over N objects)
    against make_mom_result object by object, every flag path included: flags
    and ratios identical, propagated errors to the last bit or two"""
    from ngmix_amd import moments
    rng = np.random.RandomState(nm)
    N = 600
    sums = rng.normal(size=(N, nm))
    sums[:, 5] = rng.normal(1.0, 1.0, size=N)      # some non-positive fluxes
    sums[:, 4] = rng.normal(0.5, 0.6, size=N)      # some non-positive sizes
    A = rng.normal(size=(N, nm, nm))
    cov = np.einsum("nij,nkj->nik", A, A) * 0.01
    cov[::7, 5, 5] = -1.0                          # NONPOS_VAR on the flux
    cov[1::11, 4, 4] = 0.0                         # ... on T
    cov[2::13, 2, 2] = -0.5                        # ... on a shape moment
    norm = rng.uniform(1, 2, size=N)
    b = moments.make_mom_result_batch(sums, cov, norm)
    assert len(np.unique(b["flags"])) >= 3 and len(np.unique(b["T_flags"])) >= 2
    for i in range(N):
        r = moments.make_mom_result(sums[i].copy(), cov[i].copy(), norm[i])
        for k, v in b.items():
            if k == "pars" and "pars" not in r:
                assert np.all(np.isnan(v[i]))
                continue
            rv = np.asarray(r[k], dtype="f8")
            vv = np.asarray(v[i], dtype="f8")
            if k == "sums_err" and rv.size != vv.size:
                vv = vv[:rv.size]                  # the scalar routine's 6 nans
            if k.endswith("_err") or k == "e_cov":
                np.testing.assert_allclose(vv, rv, rtol=4e-15, atol=0, equal_nan=True,
                                           err_msg=k)
            else:
                assert np.array_equal(vv, rv, equal_nan=True), (k, i)


def test_lm_batch_result_lazy_keys_are_ordinary_keys():
    """LMBatchResult keeps some arrays on the device until they are read; every
    whole-mapping view (items, values, iteration, len, dict(), {**}, copy,
    pickle) must see them like any other key (round-2 advice: a retry merge
    iterated items() and left a stale pars_cov0)"""
    import copy
    import pickle
    from ngmix_amd.lm_batch import LMBatchResult

    def make():
        calls = []
        r = LMBatchResult(a=np.arange(3))
        r.set_lazy("c", lambda _: calls.append(1) or np.ones(3))
        return r, calls

    r, calls = make()
    assert len(r) == 2 and list(r) == ["a", "c"] and "c" in r and not calls
    assert [k for k, _ in r.items()] == ["a", "c"] and len(calls) == 1
    r["c"]
    assert len(calls) == 1                      # fetched once
    for view in (lambda x: dict(x), lambda x: {**x}, lambda x: x.copy(), copy.copy,
                 copy.deepcopy, lambda x: pickle.loads(pickle.dumps(x))):
        r, calls = make()
        out = view(r)
        assert set(out) == {"a", "c"} and np.all(out["c"] == 1.0)
    r, calls = make()
    assert len(r.values()) == 2
    r, calls = make()
    r["c"] = 7                                  # overwriting drops the fetcher
    assert r["c"] == 7 and not calls and len(r) == 2
    r, calls = make()
    assert np.all(r.pop("c") == 1.0) and "c" not in r and len(r) == 1
    r, calls = make()
    del r["c"]
    assert "c" not in r and not calls
    with pytest.raises(KeyError):
        r["c"]
    # a fetcher is handed the result (no closure over it, hence no reference
    # cycle: dropping the result frees what its arrays hold at once)
    import gc
    import weakref
    r, calls = make()
    r.set_lazy("d", lambda res: res["a"] * 2)
    w = weakref.ref(r)
    assert np.all(r["d"] == 2 * np.arange(3))
    gc.disable()
    try:
        del r
        assert w() is None
    finally:
        gc.enable()


def test_lm_small_batch_arena_layout():
    """lm_batch._carve: the views of a small batch's one allocation are
    16-byte aligned, disjoint, of the requested shapes and types, and the three
    outputs are contiguous in the order the single download assumes"""
    import torch
    from ngmix_amd.lm_batch import _carve
    pieces = [("states", (3, 1000), torch.uint8), ("sums", (5, 28), torch.float64),
              ("status", (5,), torch.int32), ("sstats", (5, 2), torch.float64),
              ("flat", (3 * 24,), torch.float64), ("tri", (3, 21), torch.float64),
              ("rec", (3, 88), torch.float64)]
    buf, views, offs = _carve(torch, "cpu", pieces)
    base = buf.data_ptr()
    spans = []
    for name, shape, dt in pieces:
        v = views[name]
        assert tuple(v.shape) == tuple(shape) and v.dtype == dt and v.is_contiguous()
        a, nb = offs[name]
        assert v.data_ptr() == base + a and a % 16 == 0
        assert nb == v.numel() * v.element_size()
        spans.append((a, a + nb))
    assert all(e0 <= s1 for (_, e0), (s1, _) in zip(spans, spans[1:]))
    assert spans[-1][1] <= buf.numel()
    # writes through one view never show in another
    for name, _, _ in pieces:
        views[name].zero_()
    views["tri"].fill_(7.0)
    assert float(views["flat"].sum()) == 0.0 and float(views["rec"].sum()) == 0.0
    assert offs["flat"][0] < offs["tri"][0] < offs["rec"][0]


def test_observation_pickles_without_its_device_copies():
    """an Observation sent to a worker process (pickle) carries its host arrays;
    the device-resident stamp and batch are dropped and made again on use"""
    import pickle
    import ngmix_amd as ngmix
    rng = np.random.RandomState(5)
    obs = ngmix.Observation(rng.normal(size=(7, 9)), weight=np.full((7, 9), 4.0),
                            jacobian=ngmix.DiagonalJacobian(row=3.0, col=4.0, scale=0.2),
                            meta={"id": 3})
    obs._stamp = object.__new__(type("DeviceThing", (), {"__reduce__": lambda s: 1 / 0}))
    obs._stamp_batch = obs._stamp
    back = pickle.loads(pickle.dumps(obs))
    assert back._stamp is None and back._stamp_batch is None
    assert obs._stamp is not None                       # the original keeps its copies
    assert np.array_equal(back.image, obs.image) and np.array_equal(back.weight, obs.weight)
    assert back.meta["id"] == 3 and back.jacobian.get_scale() == obs.jacobian.get_scale()


def test_small_host_helpers_vs_reference(golden):
    """moments.regularize_mom_shapes, shape.e1e2_to_eta1eta2 and
    shape.dgs_by_dgo_jacob against the REFERENCE's own outputs
    (tests/golden/host5.npz, oracle/gen_golden_r5.py), and the typed list
    containers of ngmix/gmix/gmix_lists.py"""
    from ngmix_amd import moments, shape
    g = golden("host5")
    numeric = ("flags", "flux", "flux_err", "flux_flags", "T", "T_err", "T_flags", "s2n",
               "e1", "e2", "e", "e_err", "e_cov", "sums", "sums_cov", "pars")
    for k in range(int(g["ncase"])):
        res = moments.make_mom_result(g["reg%d_sums" % k].copy(), g["reg%d_cov" % k].copy())
        for f, fwhm in enumerate(g["fwhm_reg"]):
            reg = moments.regularize_mom_shapes(dict(res), float(fwhm))
            if fwhm == 0:
                assert set(reg) == set(res)
            for key in numeric:
                name = "reg%d_f%d_%s" % (k, f, key)
                if name in g:
                    np.testing.assert_allclose(np.asarray(reg[key], dtype="f8"), g[name],
                                               rtol=1e-13, atol=0, equal_nan=True,
                                               err_msg=name)
                else:
                    assert key not in reg, name
            assert reg["flagstr"] == str(g["reg%d_f%d_flagstr" % (k, f)])
            # the size and the flux are the unregularised ones
            assert reg["T"] == res["T"] or (np.isnan(reg["T"]) and np.isnan(res["T"]))
    e = g["e"]
    eta1, eta2 = shape.e1e2_to_eta1eta2(e[:, 0].copy(), e[:, 1].copy())
    np.testing.assert_allclose(np.stack([eta1, eta2], axis=1), g["eta"], rtol=1e-14, atol=0)
    s1, s2 = shape.e1e2_to_eta1eta2(0.3, -0.4)
    assert np.ndim(s1) == 0
    np.testing.assert_allclose([s1, s2], g["eta_scalar"], rtol=1e-14)
    with pytest.raises(ngmix.GMixRangeError):
        shape.e1e2_to_eta1eta2(0.8, 0.7)
    gg, ss = g["g"], g["s"]
    np.testing.assert_array_equal(
        shape.dgs_by_dgo_jacob(gg[:, 0], gg[:, 1], ss[:, 0], ss[:, 1]), g["jacob"])
    # typed lists
    gl = ngmix.GMixList()
    gm = ngmix.GMix(pars=[1.0, 0.0, 0.0, 1.0, 0.0, 1.0])
    gl.append(gm)
    gl[0] = gm
    with pytest.raises(AssertionError):
        gl.append("not a gmix")
    with pytest.raises(AssertionError):
        gl[0] = 3
    mb = ngmix.MultiBandGMixList()
    mb.append(gl)
    with pytest.raises(AssertionError):
        mb.append([gm])
    assert len(mb) == 1 and len(mb[0]) == 1
    assert ngmix.fitting.PSFFluxFitter is ngmix.PSFFluxFitter


@pytest.mark.parametrize("nm", [6, 17])
def test_make_mom_result_batch_on_torch_tensors_is_the_numpy_pass(nm):
    """moments.make_mom_result_batch is written in operations numpy and torch
    share: handed tensors (GaussMomBatch: the kernel's records still on the
    device) it returns tensors whose every value is the numpy pass's -- flags
    and ratios bit for bit, whatever holds a square root to an ulp -- every
    flag path included"""
    import torch
    from ngmix_amd import moments
    rng = np.random.RandomState(40 + nm)
    N = 500
    sums = rng.normal(size=(N, nm))
    sums[:, 5] = rng.normal(1.0, 1.0, size=N)
    sums[:, 4] = rng.normal(0.5, 0.6, size=N)
    A = rng.normal(size=(N, nm, nm))
    cov = np.einsum("nij,nkj->nik", A, A) * 0.01
    cov[::7, 5, 5] = -1.0
    cov[1::11, 4, 4] = 0.0
    cov[2::13, 2, 2] = -0.5
    norm = rng.uniform(1, 2, size=N)
    a = moments.make_mom_result_batch(sums, cov, norm)
    b = moments.make_mom_result_batch(torch.from_numpy(sums), torch.from_numpy(cov),
                                      torch.from_numpy(norm))
    assert set(a) == set(b)
    assert len(np.unique(a["flags"])) >= 3
    for k, v in a.items():
        t = b[k]
        assert isinstance(t, torch.Tensor), k
        got = t.numpy()
        assert got.shape == v.shape, k
        if k.endswith("_err") or k in ("s2n", "e_cov"):
            # (a square root in them: torch's and numpy's differ by an ulp at most)
            np.testing.assert_allclose(got, v, rtol=5e-16, atol=0, equal_nan=True, err_msg=k)
        else:
            assert np.array_equal(got.astype(v.dtype), v, equal_nan=True), k
    # strided views of a record matrix, as GaussMomBatch hands them over
    rec = torch.from_numpy(np.concatenate([norm[:, None], sums, cov.reshape(N, -1)], axis=1))
    c = moments.make_mom_result_batch(rec[:, 1:1 + nm], rec[:, 1 + nm:].reshape(N, nm, nm),
                                      rec[:, 0])
    for k in ("flags", "T", "T_err", "e_err", "pars", "sums_err", "MT_err"):
        np.testing.assert_allclose(c[k].numpy().astype("f8"), np.asarray(a[k], dtype="f8"),
                                   rtol=5e-16, atol=0, equal_nan=True, err_msg=k)


# ---- round 6: the many-object entry points and the batched bootstrap's host logic
def test_many_results_are_the_per_object_dicts():
    """fitting.ManyResults: element i carries Fitter.go's keys for object i --
    the statistics only where flags == 0 (results.py:45-72)"""
    from ngmix_amd.fitting import ManyResults
    n, npars = 3, 7
    rng = np.random.RandomState(2)
    cov = rng.uniform(size=(n, npars, npars))
    arrays = {"flags": np.array([0, 1 << 12, 0]), "nfev": np.array([4, 9, 5]),
              "ier": np.array([1, 5, 2]), "pars": rng.uniform(size=(n, npars)),
              "pars_err": rng.uniform(size=(n, npars)), "pars_cov0": cov, "pars_cov": cov * 2,
              "lnprob": np.array([-1.0, np.nan, -3.0]), "s2n": np.array([10.0, np.nan, 30.0]),
              "npix": np.array([100, 100, 90]), "dof": np.array([93, 93, 83]),
              "chi2per": np.array([1.0, np.nan, 1.1]), "s2n_w": np.array([10.0, np.nan, 30.0]),
              "s2n_numer": np.ones(n), "s2n_denom": np.ones(n),
              "g": rng.uniform(size=(n, 2)), "g_err": rng.uniform(size=(n, 2)),
              "g_cov": cov[:, 2:4, 2:4], "T": np.ones(n), "T_err": np.ones(n),
              "flux": rng.uniform(size=(n, 2)), "flux_err": rng.uniform(size=(n, 2)),
              "flux_cov": cov[:, 5:, 5:]}
    res = ManyResults(arrays, "exp", 2)
    assert len(res) == 3 and len(list(res)) == 3 and len(res[0:2]) == 2
    ok, bad = res[0], res[1]
    assert ok["flags"] == 0 and ok["errmsg"] == "" and ok["model"] == "exp"
    assert isinstance(ok["nfev"], int) and isinstance(ok["lnprob"], float)
    assert ok["npix"] == 100 and ok["flux"].shape == (2,) and ok["flux_cov"].shape == (2, 2)
    np.testing.assert_array_equal(ok["pars_cov"], arrays["pars_cov"][0])
    assert bad["flags"] == 1 << 12 and "ier 5" in bad["errmsg"]
    assert "lnprob" not in bad and "flux" not in bad and bad["pars"].shape == (npars,)
    assert res[-1]["nfev"] == 5
    with pytest.raises(IndexError):
        res[3]


def test_run_fitter_many_retries_only_the_failures_with_a_stub():
    from ngmix_amd.fitting import ManyResults
    from ngmix_amd.runners import run_fitter_many

    class ManyStub(object):
        """object k fails on its first k attempts"""

        def __init__(self):
            self.attempt = {}
            self.batches = []

        def go_many(self, obs, guess):
            self.batches.append([o.meta["k"] for o in obs])
            flags = []
            for o in obs:
                a = self.attempt.get(o.meta["k"], 0)
                self.attempt[o.meta["k"]] = a + 1
                flags.append(0 if a >= o.meta["k"] else 1)
            n = len(obs)
            z = np.zeros((n, 6))
            return ManyResults({"flags": np.array(flags), "nfev": np.ones(n, dtype=int),
                                "ier": np.ones(n, dtype=int), "pars": np.asarray(guess),
                                "pars_err": z, "pars_cov0": np.zeros((n, 6, 6)),
                                "pars_cov": np.zeros((n, 6, 6))}, "exp", 1)
    obs = []
    for k in (0, 2, 1, 0, 5):
        o = _obs(with_psf=False)
        o.meta["k"] = k
        obs.append(o)
    calls = []

    def guesser(obs):
        calls.append(obs.meta["k"])
        return np.full(6, float(len(calls)))
    fitter = ManyStub()
    res = run_fitter_many(obs, fitter, guesser, ntry=3)
    assert fitter.batches == [[0, 2, 1, 0, 5], [2, 1, 5], [2, 5]]
    assert [r["ntry"] for r in res] == [1, 3, 2, 1, 3]
    assert [r["flags"] for r in res] == [0, 0, 0, 0, 1]     # (the last one never passes)
    # every attempt drew a fresh guess, for the objects still failing only
    assert calls == [0, 2, 1, 0, 5, 2, 1, 5, 2, 5]
    assert res[1]["pars"][0] == 9.0        # the guess of its third attempt


def test_bootstrap_batch_host_pieces():
    """the bookkeeping of pipeline.bootstrap_batch that needs no device"""
    from ngmix_amd import pipeline as P
    # caller's guesses: (n, npars) is one attempt
    assert P._tries(np.zeros((4, 6)), "guess").shape == (1, 4, 6)
    assert P._tries(np.zeros((2, 4, 6)), "guess").shape == (2, 4, 6)
    with pytest.raises(ValueError):
        P._tries(np.zeros(6), "guess")
    # results of the fitted objects laid out over all objects
    res = {"flags": np.array([0, 3]), "pars": np.ones((2, 6)), "nfev": np.array([4, 7]),
           "model": "exp"}
    full = P._scatter(res, np.array([0, 2]), 4, 6)
    np.testing.assert_array_equal(full["flags"], [0, 0, 3, 0])
    np.testing.assert_array_equal(full["nfev"], [4, 0, 7, 0])
    assert np.all(np.isnan(full["pars"][[1, 3]])) and np.all(full["pars"][[0, 2]] == 1.0)
    assert full["model"] == "exp"
    empty = P._scatter({"model": "exp"}, np.array([], dtype=int), 3, 6)
    assert empty["pars"].shape == (3, 6) and np.all(empty["nfev"] == 0)
    # TPSFFluxGuesser's recipe for every object at once
    rng = np.random.RandomState(1)
    flux = np.array([[10.0, 20.0], [30.0, 40.0], [50.0, 60.0]])
    for model, nshape in (("exp", 5), ("bdf", 6), ("bd", 7)):
        g = P._psfflux_guess(model, 3, 2, 0.5, flux, rng)
        assert g.shape == (3, nshape + 2)
        assert np.all(np.abs(g[:, :2]) <= 0.01) and np.all(np.abs(g[:, 2:4]) <= 0.02)
        assert np.all(np.abs(g[:, 4] / 0.5 - 1) <= 0.1)
        assert np.all(np.abs(g[:, nshape:] / flux - 1) <= 0.1)
    assert np.all((P._psfflux_guess("bdf", 3, 2, 0.5, flux, rng)[:, 5] >= 0.4))
    # e -> g of the moments' ellipticities
    g1, g2 = P._e1e2_to_g1g2(np.array([0.0, 0.3]), np.array([0.0, -0.4]))
    assert g1[0] == 0.0 and abs(np.hypot(g1[1], g2[1]) - np.tanh(0.5 * np.arctanh(0.5))) < 1e-15
    with pytest.raises(ValueError):
        P.bootstrap_batch(type("S", (), {"n": 1})(), type("S", (), {"n": 1})(), model="spergel")
    assert P.BOOT_PSF_FAILURE == 1 << 30


# ---------------------------------------------------------------------------
# the reference's public names, module by module (tests/golden/api_surface.json,
# read off the reference by oracle/gen_golden_surface.py)

# what ngmix_amd does NOT offer, and why (SURVEY.md section 8 "out of scope"):
_SURFACE_OUT_OF_SCOPE = {
    # galsim-backed fitters / objects: galsim is not the pixel hot path
    "fitting": {"GalsimFitModel", "GalsimFitter", "GalsimMoffatFitModel", "GalsimMoffatFitter",
                "GalsimPSFFitModel", "GalsimPSFFluxFitter", "GalsimSpergelFitModel",
                "GalsimSpergelFitter"},
    "guessers": {"R50NuFluxGuesser"},            # the Spergel (galsim) fitter's guesser
    "joint_prior": {"PriorSpergelSep"},          # and its joint prior
    # the older polynomial variants: no kernel of the reference calls them
    # (fexp = exp5_smooth, fastexp_nb.py:265)
    "fastexp_nb": {"exp3", "exp4", "exp5"},
    # k-space observations feed the galsim fitters only
    "observation": {"KMultiBandObsList", "KObsList", "KObservation", "get_kmb_obs",
                    "make_iilist", "make_kobs"},
}
_ATTRS_OUT_OF_SCOPE = {"make_galsim_object", "get_galsim_wcs"}
_TOP_OUT_OF_SCOPE = {"metacal",             # galsim's image shearing
                     "NumbaExperimentalFeatureWarning", "warnings"}


def test_public_surface_covers_the_reference():
    import importlib
    import os
    import json
    import ngmix_amd
    with open(os.path.join(os.path.dirname(__file__), "golden", "api_surface.json")) as f:
        ref = json.load(f)
    missing = []
    for mod_name, names in ref.items():
        if mod_name == "__top__":
            missing += ["ngmix." + n for n in names
                        if not hasattr(ngmix_amd, n) and n not in _TOP_OUT_OF_SCOPE]
            continue
        mod = importlib.import_module("ngmix_amd." + mod_name)
        skip = _SURFACE_OUT_OF_SCOPE.get(mod_name, set())
        for n, attrs in names.items():
            if n in skip:
                continue
            if not hasattr(mod, n):
                missing.append("%s.%s" % (mod_name, n))
                continue
            if attrs != "function":
                obj = getattr(mod, n)
                missing += ["%s.%s.%s" % (mod_name, n, a) for a in attrs
                            if not hasattr(obj, a) and a not in _ATTRS_OUT_OF_SCOPE]
    assert not missing, missing
    # and the out-of-scope list is not stale: none of it quietly exists
    for mod_name, names in _SURFACE_OUT_OF_SCOPE.items():
        mod = importlib.import_module("ngmix_amd." + mod_name)
        assert not [n for n in names if hasattr(mod, n)]


def _param_names(f):
    import inspect
    try:
        ps = inspect.signature(f).parameters.values()
    except (ValueError, TypeError):
        return None, False
    names = [p.name for p in ps if p.name != "self"]
    catch_all = any(p.kind == p.VAR_KEYWORD for p in ps)
    return names, catch_all


def test_call_signatures_take_the_reference_argument_names():
    """every function, constructor and method the reference defines in these
    modules takes, here, every parameter NAME the reference's takes
    (tests/golden/api_signatures.json), so a call written against the reference
    with keywords binds; extra batch-only keywords are allowed"""
    import importlib
    import inspect
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "api_signatures.json")) as f:
        ref = json.load(f)
    bad = []
    for qual, want in ref.items():
        mod_name, name = qual.split(".")
        if name in _SURFACE_OUT_OF_SCOPE.get(mod_name, set()):
            continue
        obj = getattr(importlib.import_module("ngmix_amd." + mod_name), name)
        if isinstance(want, list):
            cases = [(qual, obj, want)]
        elif want is None:
            continue
        else:
            cases = []
            for attr, w in want.items():
                if w is None or attr in _ATTRS_OUT_OF_SCOPE:
                    continue
                f = obj.__init__ if attr == "__init__" else inspect.getattr_static(obj, attr)
                if isinstance(f, (staticmethod, classmethod)):
                    f = f.__func__
                if not inspect.isfunction(f):
                    continue                     # a property here, a method there: names only
                cases.append(("%s.%s" % (qual, attr), f, w))
        for label, f, w in cases:
            have, catch_all = _param_names(f)
            if have is None or catch_all:
                continue
            miss = [a for a in w if a not in have and a not in ("args", "kw", "kwargs", "keys")]
            if miss:
                bad.append((label, miss))
    assert not bad, bad


def test_host6_behaviours_found_by_the_side_by_side_audit():
    """tests/golden/host6.json (oracle/gen_golden_host6.py): format_pars
    strings, Jacobian's exception for a missing keyword, the mixture summary
    getters to the bit"""
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "host6.json")) as f:
        g = json.load(f)
    for case in g["format_pars"]:
        kw = {} if case["fmt"] is None else {"fmt": case["fmt"]}
        assert ngmix.util.format_pars(np.array(case["pars"]), **kw) == case["expected"], case
    for case in g["jacobian_errors"]:
        try:
            ngmix.Jacobian(**case["kw"])
            got = None
        except Exception as e:      # noqa: BLE001
            got = type(e).__name__
        assert got == case["expected"], (case, got)
    # ---- found by the reference's own tests run against the package
    import io
    import logging
    sh = ngmix.Shape(0.3, -0.4)
    assert float(sh.g).hex() == g["shape_g"]
    sh.set_g1g2(0.1, 0.2)
    assert float(sh.g).hex() == g["shape_g_after_set"]
    try:
        ngmix.Shape(0.1, 0.2).get_sheared(0.1)
        got = None
    except Exception as e:      # noqa: BLE001
        got = type(e).__name__
    assert got == g["shape_get_sheared_one_arg"] == "ValueError"
    buf = io.StringIO()
    ngmix.print_pars(None, stream=buf)
    ngmix.print_pars([1.0, 2.0], front="x:", stream=buf)
    assert buf.getvalue() == g["print_pars_stream"]
    rec = []
    handler = logging.Handler()
    handler.emit = lambda r: rec.append([r.levelname, r.getMessage()])
    lg = logging.getLogger("host6-test")
    lg.setLevel(logging.DEBUG)
    lg.addHandler(handler)
    ngmix.print_pars([1.0, 2.0], logger=lg)
    assert rec == [list(x) for x in g["print_pars_logger"]]
    for case in g["moms_to_e1e2"]:
        args = [float.fromhex(a) for a in case["args"]]
        try:
            got = [float(v).hex() for v in ngmix.moments.moms_to_e1e2(*args)]
        except Exception as e:      # noqa: BLE001
            got = type(e).__name__
        assert got == case["expected"], (case, got)
    with pytest.raises(ngmix.GMixRangeError):
        ngmix.moments.moms_to_e1e2(np.array([0.1]), np.array([0.2]), np.array([-0.1]))
    assert g["moms_to_e1e2_array_bad"] == "GMixRangeError"
    for case in g["getters"]:
        gm = ngmix.GMix(pars=np.array([float.fromhex(p) for p in case["pars"]]))
        for name, want in case["expected"].items():
            got = [float(v).hex() for v in np.atleast_1d(getattr(gm, name)())]
            assert got == want, (name, got, want)
