"""
ngmix_amd/guessers.py against the REFERENCE's own guessers: tests/golden/
guess.npz (oracle/gen_golden_guess.py) holds the draws of ngmix/guessers.py
from seeded RandomStates; the same seeds here give the same numbers to the
bit (the draws are additions and multiplications of the same deviates in the
same order).  CPU tests: the guessers that read only the image sum and the
jacobian; the ones that run kernels (weighted moments, psf template fluxes)
are marked gpu.
"""
import numpy as np
import pytest

import ngmix_amd as ngmix
from ngmix_amd import guessers as G
from ngmix_amd.gexceptions import GMixRangeError


def _psf_obs(g):
    j = g["psf_jac"]
    j = j[0] if j.ndim else j
    jac = ngmix.Jacobian(row=float(j["row0"]), col=float(j["col0"]), dvdrow=float(j["dvdrow"]),
                         dvdcol=float(j["dvdcol"]), dudrow=float(j["dudrow"]),
                         dudcol=float(j["dudcol"]))
    im = g["psf_image"]
    return ngmix.Observation(im, weight=np.full(im.shape, 1.0 / float(g["psf_sigma"]) ** 2),
                             jacobian=jac)


def test_object_guessers_reproduce_the_reference_stream(golden):
    g = golden("guess")
    t = G.TFluxGuesser(np.random.RandomState(11), 0.45, 120.0)
    np.testing.assert_array_equal(np.array([t() for _ in range(3)]), g["tflux1"])
    t = G.TFluxGuesser(np.random.RandomState(12), 0.45, [120.0, 80.0, 33.0])
    np.testing.assert_array_equal(np.array([t() for _ in range(3)]), g["tflux3"])
    np.testing.assert_array_equal(t(nrand=4), g["tflux3_n4"])
    r = G.R50FluxGuesser(np.random.RandomState(13), 0.7, [50.0, 60.0])
    np.testing.assert_array_equal(np.array([r() for _ in range(2)]), g["r50"])
    np.testing.assert_array_equal(r(nrand=3), g["r50_n3"])
    p = G.ParsGuesser(np.random.RandomState(14), g["pars_in"])
    np.testing.assert_array_equal(np.array([p() for _ in range(3)]), g["pars_scalar"])
    np.testing.assert_array_equal(p(nrand=5), g["pars_n5"])
    p = G.ParsGuesser(np.random.RandomState(15), g["pars_in"], widths=g["pars_widths"])
    np.testing.assert_array_equal(p(nrand=4), g["pars_w_n4"])
    with pytest.raises(GMixRangeError):
        G.R50FluxGuesser(np.random.RandomState(1), -1.0, 1.0)


def test_guess_many_is_the_stream_of_the_per_object_calls(golden):
    """guess_many(n objects) consumes the RandomState exactly as n calls do"""
    g = golden("guess")
    t = G.TFluxGuesser(np.random.RandomState(12), 0.45, [120.0, 80.0, 33.0])
    many = t.guess_many([None] * 3)
    np.testing.assert_array_equal(many, g["tflux3"])
    # ... and leaves the generator where the calls leave it
    np.testing.assert_array_equal(t(nrand=4), g["tflux3_n4"])


def test_psf_guessers_from_the_image_sum(golden):
    g = golden("guess")
    obs = _psf_obs(g)
    for ng in range(1, 6):
        gg = G.GMixPSFGuesser(np.random.RandomState(100 + ng), ng)
        got = np.array([gg(obs).get_full_pars() for _ in range(2)])
        np.testing.assert_array_equal(got, g["gmixpsf_m0_ng%d" % ng])
        cg = G.CoellipPSFGuesser(np.random.RandomState(200 + ng), ng)
        assert cg.npars == 4 + 2 * ng
        np.testing.assert_array_equal(np.array([cg(obs) for _ in range(2)]),
                                      g["coellip_m0_ng%d" % ng])
    sg = G.SimplePSFGuesser(np.random.RandomState(300))
    np.testing.assert_array_equal(np.array([sg(obs) for _ in range(3)]), g["simplepsf_m0"])
    with pytest.raises(ValueError):
        G.GMixPSFGuesser(np.random.RandomState(1), 6)


class _StubPrior(object):
    """a joint prior's three members, deterministic: sample() hands out rows
    of a table, points with |cen1| > 1 are out of range"""

    def __init__(self, npars, rng):
        self.npars = npars
        self.cen_prior = type("C", (), {"rng": rng})()
        self.k = 0

    def sample(self, nrand=None):
        n = 1 if nrand is None else nrand
        out = np.zeros((n, self.npars))
        for i in range(n):
            out[i] = 0.001 * (self.k + 1) * np.arange(1, self.npars + 1)
            self.k += 1
        return out[0] if nrand is None else out

    def get_lnprob_scalar(self, pars):
        if abs(pars[0]) > 1.0:
            raise GMixRangeError("cen")
        return -np.inf if pars[4] < 0 else 0.0


def test_prior_guessers_logic():
    rng = np.random.RandomState(5)
    prior = _StubPrior(7, rng)
    g = G.TFluxAndPriorGuesser(rng, 0.5, [10.0, 20.0], prior)
    x = g(nrand=3)
    # cen / g columns are the prior's samples, T and fluxes scatter by 10 %
    np.testing.assert_array_equal(x[:, 0], 0.001 * np.arange(1, 4))
    assert np.all(np.abs(x[:, 4] / 0.5 - 1.0) <= 0.1)
    assert np.all(np.abs(x[:, 5] / 10.0 - 1.0) <= 0.1) and np.all(np.abs(x[:, 6] / 20.0 - 1.0) <= 0.1)
    assert g().shape == (7,)
    b = G.BDFGuesser(0.5, [10.0], _StubPrior(7, np.random.RandomState(6)))
    y = b(nrand=4)
    assert np.all((y[:, 5] >= 0.4) & (y[:, 5] <= 0.6)) and np.all(np.abs(y[:, 6] / 10.0 - 1) <= 0.1)
    d = G.BDGuesser(0.5, [10.0], _StubPrior(8, np.random.RandomState(6)))
    z = d(nrand=2)
    assert np.all((z[:, 5] >= 0.4) & (z[:, 5] <= 0.6)) and np.all(np.abs(z[:, 7] / 10.0 - 1) <= 0.1)
    # a guess the prior rejects is replaced by a sample of it
    bad = np.array([[5.0, 0, 0, 0, 0.5, 1.0, 1.0]])
    G._fix_guess(bad, _StubPrior(7, rng))
    assert abs(bad[0, 0]) < 1.0
    keep = np.array([[0.3, 0.1, 0.2, 0.0, -0.5, 1.0, 1.0]])
    G._fix_guess(keep, _StubPrior(7, rng), keep_shape=True)
    np.testing.assert_array_equal(keep[0, :4], [0.3, 0.1, 0.2, 0.0])
    assert keep[0, 4] > 0
    assert G.PriorGuesser(_StubPrior(6, rng))(nrand=2).shape == (2, 6)


@pytest.mark.gpu
def test_psf_guessers_from_weighted_moments(golden):
    g = golden("guess")
    obs = _psf_obs(g)
    for ng in range(1, 6):
        gg = G.GMixPSFGuesser(np.random.RandomState(100 + ng), ng, guess_from_moms=True)
        got = np.array([gg(obs).get_full_pars() for _ in range(2)])
        # (the moments come out of the weighted-sums kernel: true exp, rounding)
        np.testing.assert_allclose(got, g["gmixpsf_m1_ng%d" % ng], rtol=1e-10, atol=1e-13)
        cg = G.CoellipPSFGuesser(np.random.RandomState(200 + ng), ng, guess_from_moms=True)
        np.testing.assert_allclose(np.array([cg(obs) for _ in range(2)]),
                                   g["coellip_m1_ng%d" % ng], rtol=1e-10, atol=1e-13)
    sg = G.SimplePSFGuesser(np.random.RandomState(300), guess_from_moms=True)
    np.testing.assert_allclose(np.array([sg(obs) for _ in range(3)]), g["simplepsf_m1"],
                               rtol=1e-10, atol=1e-13)


def _mbobs(g):
    psf_gm = ngmix.GMix(pars=g["mb_psf_pars"])
    mb = ngmix.MultiBandObsList()
    lists = {}
    for i, b in enumerate(g["mb_band"]):
        j = g["mb_jac"][i]
        jac = ngmix.Jacobian(row=float(j["row0"]), col=float(j["col0"]), dvdrow=float(j["dvdrow"]),
                             dvdcol=float(j["dvdcol"]), dudrow=float(j["dudrow"]),
                             dudcol=float(j["dudcol"]))
        im = g["mb_images"][i]
        p = ngmix.Observation(np.zeros((5, 5)), jacobian=jac, gmix=psf_gm.copy())
        o = ngmix.Observation(im, weight=np.full(im.shape, 1.0 / g["mb_sigma"][i] ** 2),
                              jacobian=jac, psf=p)
        lists.setdefault(int(b), ngmix.ObsList()).append(o)
    for b in sorted(lists):
        mb.append(lists[b])
    return mb


@pytest.mark.gpu
def test_psf_flux_guesser_vs_reference(golden):
    g = golden("guess")
    mb = _mbobs(g)
    t = G.TPSFFluxGuesser(np.random.RandomState(21), 0.5)
    got = np.array([t(obs=mb) for _ in range(3)])
    np.testing.assert_allclose(t._psf_fluxes, g["tpsfflux_fluxes"], rtol=1e-10)
    np.testing.assert_allclose(got, g["tpsfflux"], rtol=1e-10, atol=1e-13)
    # the whole-catalogue form: the same fluxes by one batch, the stream of the calls
    t2 = G.TPSFFluxGuesser(np.random.RandomState(21), 0.5)
    many = t2.guess_many([mb, _mbobs(g), mb])
    np.testing.assert_allclose(many, g["tpsfflux"], rtol=1e-10, atol=1e-13)
    # ... and through the runner: a Bootstrapper-style object fit from it
    fitter = ngmix.fitting.Fitter(model="exp")
    res = ngmix.runners.run_fitter_many([mb, _mbobs(g)], fitter,
                                        G.TPSFFluxGuesser(np.random.RandomState(3), 0.5), ntry=2)
    assert all(r["flags"] == 0 for r in res)
    one = ngmix.runners.Runner(fitter, G.TPSFFluxGuesser(np.random.RandomState(3), 0.5),
                               ntry=2).go(mb)
    np.testing.assert_allclose(res[0]["pars"], one["pars"], rtol=1e-6, atol=1e-9)
