"""
Guess generators for the runners (reference API: ngmix/guessers.py): callables
that hand a fitter its starting point, one fresh random draw per attempt.

Host-only.  Every guesser draws from its RandomState in the reference's order
(a seeded run sees the same stream: tests/golden/guess.npz holds the
reference's own draws), so Runner / PSFRunner / Bootstrapper code written
against the reference runs unchanged.  The draws are expressed as RECIPES --
an ordered list of (column, centre, half width, relative?) terms -- which one
routine turns into numbers; the same routine serves `guess_many`, the
whole-catalogue form run_fitter_many uses: it consumes the stream exactly as
the per-object calls would, object after object.

Priors are duck-typed (sample, get_lnprob_scalar, cen_prior.rng): the priors
package itself is outside this repository's scope.  The spergel guesser
(R50NuFluxGuesser) belongs to the galsim fitters and is not provided.
"""
import logging

import numpy as np

from . import moments
from .defaults import LOWVAL
from .gexceptions import GMixRangeError, PSFFluxFailure
from .gmix import GMix, GMixModel, get_coellip_npars
from .shape import Shape
from .util import print_pars, srandu

LOGGER = logging.getLogger(__name__)

__all__ = ["TFluxGuesser", "TPSFFluxGuesser", "TPSFFluxAndPriorGuesser",
           "TFluxAndPriorGuesser", "BDFPSFFluxGuesser", "BDFGuesser", "BDGuesser",
           "ParsGuesser", "R50FluxGuesser", "PriorGuesser", "GMixPSFGuesser",
           "SimplePSFGuesser", "CoellipPSFGuesser", "get_shape_guess", "get_psf_fluxes"]


# --------------------------------------------------------------------------
# the draw engine
# --------------------------------------------------------------------------
def _uniform_columns(rng, nrand, terms, ncol):
    """
    terms: [(column, centre, low, high, relative)] in DRAW ORDER.  Each term
    takes nrand uniform deviates in [low, high) -- rng.uniform(low, high, nrand),
    the reference's call -- and writes centre * u (relative) or centre + u.
    """
    out = np.zeros((nrand, ncol))
    for col, centre, low, high, relative in terms:
        u = rng.uniform(low=low, high=high, size=nrand)
        out[:, col] = centre * u if relative else centre + u
    return out


def _uniform_rows(rng, nobj, terms, ncol, centres=None):
    """the same terms drawn OBJECT AFTER OBJECT, one deviate per term and object
    -- the stream nobj separate calls with nrand = 1 consume -- as one block
    draw: rng.uniform(low, high) is low + (high - low) * random_sample(), so a
    (nobj, nterm) block of random_sample() scaled term by term is bit for bit
    the loop's.  centres: optional (nobj, nterm) per-object centres"""
    nt = len(terms)
    block = rng.random_sample(size=(nobj, nt))
    out = np.zeros((nobj, ncol))
    for k, (col, centre, low, high, relative) in enumerate(terms):
        u = low + (high - low) * block[:, k]
        c = centre if centres is None else centres[:, k]
        out[:, col] = c * u if relative else c + u
    return out


def _maybe_scalar(guess, nrand):
    return guess[0, :] if nrand == 1 else guess


def _fix_guess(guess, prior, ntry=4, keep_shape=False):
    """rows the prior rejects (ln p <= LOWVAL or a range error) are replaced by
    a sample of it -- keep_shape: only their size and flux columns (guessers.py
    _fix_guess / _fix_guess_TFlux)"""
    for j in range(guess.shape[0]):
        for _ in range(ntry):
            try:
                bad = prior.get_lnprob_scalar(guess[j, :]) <= LOWVAL
            except GMixRangeError:
                bad = True
            if not bad:
                break
            print_pars(guess[j, :], front="bad guess:", logger=LOGGER)
            if keep_shape:
                guess[j, 4:] = prior.sample()[4:]
            else:
                guess[j, :] = prior.sample()


def _tight_terms(T, fluxes, first_flux=5):
    """cen within 0.01, g within 0.02, T and the fluxes within 10 %"""
    terms = [(0, 0.0, -0.01, 0.01, False), (1, 0.0, -0.01, 0.01, False),
             (2, 0.0, -0.02, 0.02, False), (3, 0.0, -0.02, 0.02, False),
             (4, T, 0.9, 1.1, True)]
    terms += [(first_flux + b, f, 0.9, 1.1, True) for b, f in enumerate(fluxes)]
    return terms


# --------------------------------------------------------------------------
# psf fluxes (guessers.py:205-262)
# --------------------------------------------------------------------------
def get_psf_fluxes(rng, obs):
    """template fluxes of the psf per band (PSFFluxFitter on each band's
    observations); a band whose fit is flagged or not finite takes the mean of
    the good ones times 1 + U(-0.1, 0.1); none good: PSFFluxFailure"""
    from .observation import get_mb_obs
    from .psfflux import PSFFluxFitter
    mbobs = get_mb_obs(obs)
    nband = len(mbobs)
    flux, flux_err = np.zeros(nband), np.zeros(nband)
    flags = np.zeros(nband, dtype="i4")
    fitter = PSFFluxFitter()
    for b, obslist in enumerate(mbobs):
        res = fitter.go(obs=obslist)
        flags[b], flux[b], flux_err[b] = res["flags"], res["flux"], res["flux_err"]
    good = (flags == 0) & np.isfinite(flux)
    if not good.all():
        if not good.any():
            raise PSFFluxFailure("no good psf fluxes")
        nbad = int((~good).sum())
        flux[~good] = flux[good].mean() * (1.0 + rng.uniform(low=-0.1, high=0.1, size=nbad))
    return {"flags": flags, "flux": flux, "flux_err": flux_err}


_get_psf_fluxes = get_psf_fluxes


class _PSFFluxMixin(object):
    """the psf fluxes of the observation last seen (one PSFFluxFitter run per
    observation, however many attempts ask)"""
    _id_last = None
    _psf_fluxes = None

    def _get_psf_fluxes(self, obs):
        if id(obs) != self._id_last:
            self._id_last = id(obs)
            self._psf_fluxes = get_psf_fluxes(rng=self.rng, obs=obs)["flux"]
        return self._psf_fluxes


# --------------------------------------------------------------------------
# object guessers
# --------------------------------------------------------------------------
class TFluxGuesser(object):
    """cen and shape tightly around zero, T and fluxes within 10 % of the
    given ones (guessers.py:14-76)"""

    def __init__(self, rng, T, flux, prior=None):
        self.rng, self.T, self.prior = rng, T, prior
        self.fluxes = np.array(flux, dtype="f8", ndmin=1)

    def __call__(self, nrand=1, obs=None):
        guess = _uniform_columns(self.rng, nrand, _tight_terms(self.T, self.fluxes),
                                 5 + self.fluxes.size)
        if self.prior is not None:
            _fix_guess(guess, self.prior)
        return _maybe_scalar(guess, nrand)

    def guess_many(self, obs):
        """one guess per object of `obs`: the stream of len(obs) calls"""
        guess = _uniform_rows(self.rng, len(obs), _tight_terms(self.T, self.fluxes),
                              5 + self.fluxes.size)
        if self.prior is not None:
            _fix_guess(guess, self.prior)
        return guess


class TPSFFluxGuesser(_PSFFluxMixin):
    """TFluxGuesser with the fluxes taken from psf (template) fits of the
    observation (guessers.py:78-145)"""

    def __init__(self, rng, T, prior=None):
        self.rng, self.T, self.prior = rng, T, prior

    def __call__(self, obs, nrand=1):
        fluxes = self._get_psf_fluxes(obs=obs)
        guess = _uniform_columns(self.rng, nrand, _tight_terms(self.T, fluxes), 5 + fluxes.size)
        if self.prior is not None:
            _fix_guess(guess, self.prior)
        return _maybe_scalar(guess, nrand)

    def guess_many(self, obs):
        """one guess per object: the template fluxes of ALL objects by one batch
        (PSFFluxBatch); the draws are those of the per-object calls unless a band
        of some object needs its flux replaced (those deviates are then drawn
        before the batch's guesses, not between two objects')"""
        from .batch import flatten_observations, GMixBatch
        from .psfflux import PSFFluxBatch
        stamps, sobj, sband, nband, psf = flatten_observations(obs)
        if psf is None:
            raise ValueError("psf-flux guesses need the observations' psf mixtures")
        nobj = len(obs)
        key = sobj.astype(np.int64) * nband + sband
        res = PSFFluxBatch().go(stamps, GMixBatch.from_numpy(psf, device=stamps.device),
                                stamp_obj=key, nobj=nobj * nband)
        flux = res["flux"].reshape(nobj, nband).copy()
        good = (res["flags"].reshape(nobj, nband) == 0) & np.isfinite(flux)
        terms = _tight_terms(self.T, np.ones(nband))
        centres = np.ones((nobj, len(terms)))
        centres[:, 4] = self.T
        for k in range(4):
            centres[:, k] = 0.0
        block_rng = self.rng
        bad_o, bad_b = np.nonzero(~good)
        if bad_o.size:
            ngood = good.sum(axis=1)
            if np.any(ngood == 0):
                raise PSFFluxFailure("no good psf fluxes")
            mean = np.where(good, flux, 0.0).sum(axis=1) / ngood
            flux[bad_o, bad_b] = mean[bad_o] * (1.0 + block_rng.uniform(-0.1, 0.1,
                                                                        size=bad_o.size))
        centres[:, 5:] = flux
        guess = _uniform_rows(block_rng, nobj, terms, 5 + nband, centres=centres)
        if self.prior is not None:
            _fix_guess(guess, self.prior)
        return guess


class TFluxAndPriorGuesser(object):
    """cen and shape sampled from the joint prior, T and fluxes within 10 % of
    the given ones (guessers.py:264-322)"""

    first_flux = 5

    def __init__(self, rng, T, flux, prior):
        self.T, self.prior = T, prior
        self.fluxes = np.array(flux, dtype="f8", ndmin=1)

    def _fluxes(self, obs):
        return self.fluxes

    def _extras(self, rng, guess, nrand):
        pass

    def __call__(self, nrand=1, obs=None):
        rng = self.prior.cen_prior.rng
        fluxes = self._fluxes(obs)
        guess = self.prior.sample(nrand)
        guess[:, 4] = self.T * (1.0 + rng.uniform(low=-0.1, high=0.1, size=nrand))
        self._extras(rng, guess, nrand)
        for b in range(fluxes.size):
            r = rng.uniform(low=-0.1, high=0.1, size=nrand)
            guess[:, self.first_flux + b] = fluxes[b] * (1.0 + r)
        self._fix(guess)
        return _maybe_scalar(guess, nrand)

    def _fix(self, guess):
        _fix_guess(guess, self.prior, keep_shape=True)


class BDFGuesser(TFluxAndPriorGuesser):
    """'bdf': as TFluxAndPriorGuesser with fracdev in [0.4, 0.6]
    (guessers.py:379-430)"""

    first_flux = 6

    def __init__(self, T, flux, prior):
        super().__init__(None, T, flux, prior)

    def _extras(self, rng, guess, nrand):
        guess[:, 5] = rng.uniform(low=0.4, high=0.6, size=nrand)

    def _fix(self, guess):
        _fix_guess(guess, self.prior)


class BDGuesser(BDFGuesser):
    """'bd' (guessers.py:432-487; as there, the uniform draw lands in column 5
    and the fluxes start at column 7)"""

    first_flux = 7


class TPSFFluxAndPriorGuesser(_PSFFluxMixin, TFluxAndPriorGuesser):
    """TFluxAndPriorGuesser with psf fluxes (guessers.py:147-203: the fluxes
    scatter by U(0.9, 1.1))"""

    def __init__(self, rng, T, prior):
        self.rng, self.T, self.prior = rng, T, prior

    def __call__(self, obs, nrand=1):
        rng = self.rng
        fluxes = self._get_psf_fluxes(obs=obs)
        guess = self.prior.sample(nrand)
        guess[:, 4] = self.T * (1.0 + rng.uniform(low=-0.1, high=0.1, size=nrand))
        for b in range(fluxes.size):
            guess[:, 5 + b] = fluxes[b] * rng.uniform(low=0.9, high=1.1, size=nrand)
        _fix_guess(guess, self.prior, keep_shape=True)
        return _maybe_scalar(guess, nrand)


class BDFPSFFluxGuesser(_PSFFluxMixin, BDFGuesser):
    """BDFGuesser with psf fluxes (guessers.py:325-377)"""

    def __init__(self, T, prior):
        self.T, self.prior = T, prior
        self.rng = prior.cen_prior.rng

    def _fluxes(self, obs):
        return self._get_psf_fluxes(obs=obs)

    def __call__(self, obs, nrand=1):
        return TFluxAndPriorGuesser.__call__(self, nrand=nrand, obs=obs)


class R50FluxGuesser(object):
    """TFluxGuesser's layout with a half-light radius in the size column and
    symmetric deviates (guessers.py:602-664)"""

    def __init__(self, rng, r50, flux, prior=None):
        if r50 < 0.0:
            raise GMixRangeError("r50 <= 0: %g" % r50)
        self.rng, self.r50, self.prior = rng, r50, prior
        self.fluxes = np.array(flux, dtype="f8", ndmin=1)

    def __call__(self, nrand=1, obs=None):
        rng = self.rng
        nband = self.fluxes.size
        guess = np.zeros((nrand, 5 + nband))
        for col, width in ((0, 0.01), (1, 0.01), (2, 0.02), (3, 0.02)):
            guess[:, col] = width * srandu(nrand, rng=rng)
        guess[:, 4] = self.r50 * (1.0 + 0.1 * srandu(nrand, rng=rng))
        for b in range(nband):
            guess[:, 5 + b] = self.fluxes[b] * (1.0 + 0.1 * srandu(nrand, rng=rng))
        if self.prior is not None:
            _fix_guess(guess, self.prior)
        return _maybe_scalar(guess, nrand)


class PriorGuesser(object):
    """samples of a joint prior (guessers.py:667-686)"""

    def __init__(self, prior):
        self.prior = prior

    def __call__(self, obs=None, nrand=None):
        return self.prior.sample(nrand)


def get_shape_guess(rng, g1, g2, nrand, width, max=0.99):
    """nrand shapes around (g1, g2): the shape (shrunk to |g| <= max) sheared by
    small random offsets, redrawn until the result is a valid shape
    (guessers.py:570-599)"""
    g = np.sqrt(g1 ** 2 + g2 ** 2)
    if g > max:
        g1, g2 = g1 * (max / g), g2 * (max / g)
    base = Shape(g1, g2)
    out = np.zeros((nrand, 2))
    for i in range(nrand):
        while True:
            try:
                off1 = width[0] * srandu(rng=rng)
                off2 = width[1] * srandu(rng=rng)
                s = base.get_sheared(off1, off2)
                break
            except GMixRangeError:
                continue
        out[i] = s.g1, s.g2
    return out


class ParsGuesser(object):
    """guesses scattered around a parameter vector: absolute widths for cen and
    g, relative ones for the rest (guessers.py:489-567)"""

    def __init__(self, rng, pars, prior=None, widths=None):
        self.rng, self.prior = rng, prior
        self.pars = np.array(pars)
        self.np = self.pars.size
        if widths is None:
            widths = self.pars * 0 + 0.1
            widths[0:2] = 0.02
        self.widths = widths

    def __call__(self, nrand=None, obs=None):
        rng, pars, w = self.rng, self.pars, self.widths
        scalar = nrand is None
        n = 1 if scalar else nrand
        guess = np.zeros((n, self.np))
        guess[:, 0] = pars[0] + w[0] * srandu(n, rng=rng)
        guess[:, 1] = pars[1] + w[1] * srandu(n, rng=rng)
        guess[:, 2:4] = get_shape_guess(rng=rng, g1=pars[2], g2=pars[3], nrand=n,
                                        width=w[2:4], max=0.8)
        for i in range(4, self.np):
            guess[:, i] = pars[i] * (1.0 + w[i] * srandu(n, rng=rng))
        if self.prior is not None:
            _fix_guess(guess, self.prior)
        return guess[0, :] if scalar else guess


# --------------------------------------------------------------------------
# psf guessers
# --------------------------------------------------------------------------
# flux fractions and size factors of the full-mixture guesses (EM / admom) and
# of the co-elliptical ones, by number of gaussians (guessers.py:1025-1051,
# 1217-1242); the 4- and 5-gaussian mixtures repeat their third size factor
_EM_TABLES = {
    2: ([0.596510042804182, 0.4034898268889178],
        [0.5793612389470884, 1.621860687127999]),
    3: ([0.596510042804182, 0.4034898268889178, 1.303069003078001e-07],
        [0.5793612389470884, 1.621860687127999, 7.019347162356363]),
    4: ([0.596510042804182, 0.4034898268889178, 1.303069003078001e-07, 1.0e-8],
        [0.5793612389470884, 1.621860687127999, 7.019347162356363, 16.0]),
    5: ([0.59453032, 0.35671819, 0.03567182, 0.01189061, 0.00118906],
        [0.5, 1.0, 3.0, 10.0, 20.0]),
}
_COELLIP_TABLES = {
    2: ([0.5, 0.5], [0.48955064, 1.50658978]),
    3: ([0.27559669, 0.55817131, 0.166232], [0.36123609, 0.8426139, 2.58747785]),
    4: ([0.44534, 0.366951, 0.10506, 0.0826497], [0.541019, 1.19701, 0.282176, 3.51086]),
    5: ([0.57874897, 0.32273483, 0.03327272, 0.0341253, 0.03111819],
        [0.27831284, 0.9959897, 5.86989779, 5.63590429, 4.17285878]),
}


class GMixPSFGuesser(object):
    """a full gaussian mixture for a psf fit (EM, adaptive moments): flux and
    size from the image sum and 3.5 pixels of fwhm, or from weighted moments
    (guess_from_moms), split over ngauss gaussians (guessers.py:767-1023)"""

    def __init__(self, rng, ngauss, guess_from_moms=False):
        if not 1 <= ngauss <= 5:
            raise ValueError("bad ngauss: %d" % ngauss)
        self.rng, self.ngauss, self.guess_from_moms = rng, ngauss, guess_from_moms

    def __call__(self, obs):
        return self._get_guess(obs=obs)

    def _get_guess(self, obs):
        T, flux = self._get_T_flux(obs=obs)
        return self._split(flux=flux, T=T)

    # ---- the scale of the guess
    def _get_T_flux(self, obs):
        if self.guess_from_moms:
            return self._get_T_flux_from_moms(obs=obs)
        return self._get_T_flux_default(obs=obs)

    def _get_T_flux_default(self, obs):
        return moments.fwhm_to_T(obs.jacobian.scale * 3.5), obs.image.sum()

    def _get_T_flux_from_moms(self, obs):
        scale = obs.jacobian.scale
        Tweight = moments.fwhm_to_T(scale * 3.5)
        wt = GMixModel([0.0, 0.0, 0.0, 0.0, Tweight, 1.0], "gauss")
        res = wt.get_weighted_moments(obs=obs, maxrad=1.0e9)
        if res["flags"] != 0:
            return self._get_T_flux_default(obs=obs)
        Tmeas = res["T"]
        if moments.T_to_fwhm(Tmeas) < scale:
            return self._get_T_flux_default(obs=obs)
        # deweighted as if the profile were a gaussian; fluxes per unit area
        T = 1.0 / (1 / Tmeas - 1 / Tweight)
        flux = res["flux"] * np.pi * (Tweight + T) / scale ** 2
        if T < 0:
            T, flux = res["T"], res["flux"]
        return T, flux

    # ---- the mixture
    def _split(self, flux, T):
        rng = self.rng
        sigma2 = T / 2
        ng = self.ngauss
        u = rng.uniform
        if ng == 1:
            pars = [flux * u(low=0.9, high=1.1), u(low=-0.1, high=0.1), u(low=-0.1, high=0.1),
                    sigma2 * (1.0 + u(low=-0.1, high=0.1)),
                    u(low=-0.2 * sigma2, high=0.2 * sigma2),
                    sigma2 * (1.0 + u(low=-0.1, high=0.1))]
            return GMix(pars=np.array(pars))
        frac, fac = _EM_TABLES[ng]
        pars = []
        for k in range(ng):
            kk = min(k, 2) if ng >= 4 else k      # (see the tables' comment)
            if ng == 2:
                p, irc = frac[k] * flux, None
            else:
                p = flux * frac[kk] * (1.0 + u(low=-0.1, high=0.1))
            row, col = u(low=-0.1, high=0.1), u(low=-0.1, high=0.1)
            irr = fac[kk] * sigma2 * (1.0 + u(low=-0.1, high=0.1))
            irc = 0.0 if ng == 2 else u(low=-0.01, high=0.01)
            icc = fac[kk] * sigma2 * (1.0 + u(low=-0.1, high=0.1))
            pars += [p, row, col, irr, irc, icc]
        return GMix(pars=np.array(pars))


class SimplePSFGuesser(GMixPSFGuesser):
    """[cen1, cen2, g1, g2, T, flux] for a simple-model psf fit
    (guessers.py:1054-1103)"""

    def __init__(self, rng, guess_from_moms=False):
        self.rng, self.guess_from_moms, self.npars = rng, guess_from_moms, 6

    def _head(self):
        guess = np.zeros(self.npars)
        guess[0:2] += self.rng.uniform(low=-0.01, high=0.01, size=2)
        guess[2:4] += self.rng.uniform(low=-0.05, high=0.05, size=2)
        return guess

    def _get_guess(self, obs):
        T, flux = self._get_T_flux(obs=obs)
        guess = self._head()
        guess[4] = T * self.rng.uniform(low=0.9, high=1.1)
        guess[5] = flux * self.rng.uniform(low=0.9, high=1.1)
        return guess


class CoellipPSFGuesser(SimplePSFGuesser):
    """[cen1, cen2, g1, g2, T_1.., F_1..] for a co-elliptical psf fit
    (guessers.py:1106-1243)"""

    def __init__(self, rng, ngauss, guess_from_moms=False):
        if not 1 <= ngauss <= 5:
            raise ValueError("bad ngauss: %d" % ngauss)
        self.rng, self.ngauss, self.guess_from_moms = rng, ngauss, guess_from_moms
        self.npars = get_coellip_npars(ngauss)

    def _get_guess(self, obs):
        if self.ngauss == 1:
            return SimplePSFGuesser._get_guess(self, obs)
        T, flux = self._get_T_flux(obs=obs)
        guess = self._head()
        frac, fac = _COELLIP_TABLES[self.ngauss]
        ng = self.ngauss
        for k in range(ng):
            guess[4 + k] = T * fac[k] * self.rng.uniform(low=0.99, high=1.01)
        for k in range(ng):
            guess[4 + ng + k] = flux * frac[k] * self.rng.uniform(low=0.99, high=1.01)
        return guess
