"""
Levenberg-Marquardt model fitting (reference API: ngmix/fitting/fitters.py,
results.py, leastsqbound.py).

Orchestration stays on the host, as north_star asks: scipy's MINPACK drives the
iteration and calls back into FitModel.  What changes is the callback body:
all observations of the object (bands x epochs) live in ONE device-resident
StampBatch, and each evaluation is a single kernel launch

    calc_fdiff    -> ngmix_fill_fdiff_batch   (residuals written straight into
                                               the LM vector layout)
    calc_jacobian -> ngmix_deriv_images_batch (6 derivative images per obs)
    calc_lnprob   -> ngmix_loglike_batch

instead of the reference's Python loop of one njit call per observation
(results.py:176-189, 456-461, 533-565).  The O(ngauss) model fill / psf
convolution / norms per evaluation go through the C ABI's host entry points.
"""
import copy
import logging

import os

import numpy as np

from . import _lib
from . import gmix as gmix_mod
from .defaults import PDEF, CDEF, LOWVAL, BIGVAL, DEFAULT_LM_PARS
from .flags import (
    ZERO_DOF, DIV_ZERO, LM_SINGULAR_MATRIX, LM_NEG_COV_EIG, LM_NEG_COV_DIAG,
    EIG_NOTFINITE, LM_FUNC_NOTFINITE,
)
from .gexceptions import GMixRangeError
from .observation import get_mb_obs
from .util import print_pars

__all__ = ["Fitter", "CoellipFitter", "FitModel", "CoellipFitModel",
           "run_leastsq", "leastsqbound"]

LOGGER = logging.getLogger(__name__)

SIMPLE_ANALYTIC_MODELS = ("gauss", "exp", "dev")
# relative step for differencing the (smooth) prior rows of the jacobian
STEP_PRIOR = 1.0e-8  # results.py:935


# ---------------------------------------------------------------------------
# bounded least squares (reference: ngmix/fitting/leastsqbound.py)
# ---------------------------------------------------------------------------

class _Sentinels(object):
    """the 'no value' arrays of a failed fit (defaults.py: PDEF for parameters,
    CDEF for covariances and errors)"""

    def __init__(self, npars):
        self.pars = np.full(npars, PDEF)
        self.cov = np.full((npars, npars), CDEF)
        self.err = np.full(npars, CDEF)


def _get_def_stuff(npars):
    d = _Sentinels(npars)
    return d.pars, d.cov, d.err


def _test_cov(pcov):
    """flag bits for a covariance with negative eigenvalues / diagonal, or
    one whose eigenvalues cannot be computed"""
    try:
        eigvals = np.linalg.eigvals(pcov)
    except np.linalg.LinAlgError:
        return EIG_NOTFINITE
    flags = 0
    if (eigvals < 0).any():
        flags |= LM_NEG_COV_EIG
    if (pcov.diagonal() < 0).any():
        flags |= LM_NEG_COV_DIAG
    return flags


class _BoundsTransform(object):
    """
    MINUIT-style maps between external (bounded) and internal (free)
    parameters, per parameter kind: none / lower only / upper only / both
    (leastsqbound.py:183-264).
    """

    def __init__(self, bounds):
        self.bounds = [tuple(b) for b in bounds]

    def i2e(self, xi):
        xe = np.empty_like(xi)
        for i, (v, (lo, hi)) in enumerate(zip(xi, self.bounds)):
            if lo is None and hi is None:
                xe[i] = v
            elif hi is None:
                xe[i] = lo - 1.0 + np.sqrt(v * v + 1.0)
            elif lo is None:
                xe[i] = hi + 1.0 - np.sqrt(v * v + 1.0)
            else:
                xe[i] = lo + ((hi - lo) / 2.0) * (np.sin(v) + 1.0)
        return xe

    def e2i(self, xe):
        xi = np.empty_like(xe)
        for i, (v, (lo, hi)) in enumerate(zip(xe, self.bounds)):
            if lo is None and hi is None:
                xi[i] = v
            elif hi is None:
                xi[i] = np.sqrt((v - lo + 1.0) ** 2 - 1)
            elif lo is None:
                xi[i] = np.sqrt((hi - v + 1.0) ** 2 - 1)
            else:
                xi[i] = np.arcsin((2.0 * (v - lo) / (hi - lo)) - 1.0)
        return xi

    def grad(self, xi):
        g = np.empty_like(xi)
        for i, (v, (lo, hi)) in enumerate(zip(xi, self.bounds)):
            if lo is None and hi is None:
                g[i] = 1.0
            elif hi is None:
                g[i] = v / np.sqrt(v * v + 1.0)
            elif lo is None:
                g[i] = -v / np.sqrt(v * v + 1.0)
            else:
                g[i] = (hi - lo) * np.cos(v) / 2.0
        return g


def leastsqbound(func, x0, args=(), bounds=None, Dfun=None, full_output=0,
                 col_deriv=0, ftol=1.49012e-8, xtol=1.49012e-8, gtol=0.0,
                 maxfev=0, epsfcn=None, factor=100, diag=None):
    """
    scipy.optimize.leastsq (MINPACK lmdif / lmder) with optional (min, max)
    bounds per parameter enforced through the MINUIT internal<->external
    transformation; the returned covariance is in external parameters.
    """
    from scipy.optimize import leastsq
    if bounds is None:
        return leastsq(func, x0, args, Dfun, full_output, col_deriv, ftol, xtol,
                       gtol, maxfev, epsfcn, factor, diag)

    tr = _BoundsTransform(bounds)
    x0 = np.asarray(x0, dtype="f8").flatten()
    n = len(x0)
    if len(bounds) != n:
        raise ValueError("length of x0 != length of bounds")
    if not isinstance(args, tuple):
        args = (args,)
    i0 = tr.e2i(x0)

    def wfunc(x, *a):
        return func(tr.i2e(x), *a)

    wDfun = None
    if Dfun is not None:
        def wDfun(x, *a):
            scale = tr.grad(x)
            if col_deriv == 1:
                scale = scale.reshape(len(x), 1)
            return Dfun(tr.i2e(x), *a) * scale

    xi, _, infodict, mesg, ier = leastsq(
        wfunc, i0, args, wDfun, 1, col_deriv, ftol, xtol, gtol, maxfev, epsfcn,
        factor, diag)
    x = tr.i2e(xi)
    if not full_output:
        return x, ier
    # R of the final jacobian's QR, rescaled from internal to external pars
    grad = tr.grad(xi)
    # MINPACK's ipvt is 1-based from the Fortran library and 0-based from the
    # C translation scipy ships since 1.15; the reference subtracts 1
    # unconditionally (leastsqbound.py:536-543), which under the newer scipy
    # rotates the covariance by one parameter.  The permutation itself says
    # which base it has.
    ipvt0 = infodict["ipvt"] - infodict["ipvt"].min()
    infodict["fjac"] = (infodict["fjac"].T / np.take(grad, ipvt0)).T
    cov_x = None
    if ier in (1, 2, 3, 4):
        perm = np.take(np.eye(n), ipvt0, 0)
        r = np.triu(np.transpose(infodict["fjac"])[:n, :])
        R = np.dot(r, perm)
        try:
            cov_x = np.linalg.inv(np.dot(np.transpose(R), R))
        except (np.linalg.LinAlgError, ValueError):
            pass
    return x, cov_x, infodict, mesg, ier


def _aborted(npars, flags, errmsg):
    """result of a fit whose objective could not be evaluated: no pars_err,
    no ier, nfev = -1 (leastsqbound.py:127-153)"""
    LOGGER.debug(errmsg)
    d = _Sentinels(npars)
    return {"flags": flags, "nfev": -1, "errmsg": errmsg, "pars": d.pars,
            "pars_cov0": d.cov, "pars_cov": d.cov}


def _scaled_covariance(func, pars, cov0, n_prior_pars, k_space):
    """(flags, pars_cov, pars_err, errmsg or None) of a converged fit: cov0
    scaled by chi^2 / dof of the pixel rows (leastsqbound.py:92-116)"""
    npars = pars.size
    none = _Sentinels(npars)
    resid = func(pars)
    ndata = resid.size - n_prior_pars
    if k_space:
        ndata //= 2       # real and imaginary parts are one datum
    dof = ndata - npars
    if dof == 0:
        return ZERO_DOF, none.cov, none.err, None
    pix = resid[n_prior_pars:]
    cov = cov0 * ((pix ** 2).sum() / dof)
    bad = _test_cov(cov)
    if bad:
        return bad, cov, none.err, "bad covariance matrix"
    return 0, cov, np.sqrt(cov.diagonal()), None


def run_leastsq(func, guess, n_prior_pars, **keys):
    """
    One (bounded) MINPACK fit, packaged as the reference's run_leastsq does
    (leastsqbound.py:33-155).  Outcomes, by what leastsq reports:

      ier 1-4, cov_x given   flags from the chi^2-scaled covariance's sanity
                             checks (0 when fine), ZERO_DOF when no freedom
      ier 1-4, cov_x None    LM_SINGULAR_MATRIX, parameters kept
      ier > 4                2**(ier-5), sentinel parameters
      ier 0                  ValueError(errmsg)
      NaN / inf residuals    LM_FUNC_NOTFINITE (scipy's ValueError)
      ZeroDivisionError      DIV_ZERO
    """
    npars = np.size(guess)
    k_space = keys.pop("k_space", False)
    try:
        pars, cov0, info, errmsg, ier = leastsqbound(func, guess, full_output=1, **keys)
        if ier == 0:
            raise ValueError(errmsg)
        none = _Sentinels(npars)
        if ier > 4:
            LOGGER.debug(errmsg)
            flags, pars, cov, err = 2 ** (ier - 5), none.pars, none.cov, none.err
        elif cov0 is None:
            errmsg = "singular covariance"
            LOGGER.debug(errmsg)
            print_pars(pars, front="    pars at singular:", logger=LOGGER)
            flags, cov, err = LM_SINGULAR_MATRIX, none.cov, none.err
        else:
            flags, cov, err, msg = _scaled_covariance(func, pars, cov0, n_prior_pars,
                                                      k_space)
            if msg is not None:
                errmsg = msg
                LOGGER.debug(errmsg)
        return {"flags": flags, "nfev": info["nfev"], "ier": ier, "errmsg": errmsg,
                "pars": pars, "pars_err": err, "pars_cov0": cov0, "pars_cov": cov}
    except ZeroDivisionError:
        return _aborted(npars, DIV_ZERO, "zero division")
    except ValueError as err:
        if not any(word in str(err) for word in ("NaNs", "infs")):
            raise
        return _aborted(npars, LM_FUNC_NOTFINITE, "not finite")


# ---------------------------------------------------------------------------
# model bookkeeping
# ---------------------------------------------------------------------------

def get_band_pars(model, pars, band):
    """shared shape parameters + the flux of `band` (results.py:1013-1047)"""
    num = gmix_mod.get_model_npars(model)
    band_pars = np.zeros(num)
    assert model != "coellip"
    nshared = {"bd": 7, "bdf": 6}.get(model, 5)
    band_pars[0:nshared] = pars[0:nshared]
    band_pars[nshared] = pars[nshared + band]
    return band_pars


def get_lm_n_prior_pars(model, nband):
    """number of residual rows reserved for the prior (results.py:1050-1078)"""
    if model == "bd":
        return 6 + nband
    if model == "bdf":
        return 5 + nband
    if model in ("exp", "dev", "gauss", "turb"):
        return 5 + nband
    raise ValueError("bad model: %s" % model)


# d(irr, irc, icc) per unit of (e1, e2) at T_k / 2 = 1:
# Sigma_k = (T_k/2) [[1 - e1, e2], [e2, 1 + e1]]
_DCOV_DE = np.array([[-1.0, 0.0, 1.0],
                     [0.0, 1.0, 0.0]])


def get_model_deriv_data(gm0, gmc, g1, g2, T):
    """
    For a model whose components share one shape (gauss / exp / dev), the
    composed gaussians as rows [p, v, u, irr, irc, icc] and, per composed
    gaussian, d(irr, irc, icc)/d(g1, g2, T) (results.py:955-1010).  The psf
    part of a composed covariance does not depend on the model, so only the
    model component's Sigma_k = (T_k/2) [[1-e1, e2], [e2, 1+e1]],
    e = 2 g / (1 + |g|^2), is differentiated: chain rule through
    J = d(e1, e2)/d(g1, g2), and d Sigma_k / dT = Sigma_k / T.
    """
    composed = gmc.get_full_pars().reshape(-1, 6)
    model_cov = gm0.get_full_pars().reshape(-1, 6)[:, 3:6]
    # convolve() orders the composed gaussians model-major
    model_cov = np.repeat(model_cov, composed.shape[0] // model_cov.shape[0], axis=0)
    half_Tk = 0.5 * (model_cov[:, 0] + model_cov[:, 2])

    g = np.array([g1, g2])
    gsq = g1 * g1 + g2 * g2
    f = 2.0 / (1.0 + gsq)
    jac_e = f * np.eye(2) + (2.0 * g[:, None] * g[None, :]) * (-f / (1.0 + gsq))

    dcov = np.empty((composed.shape[0], 3, 3))
    dcov[:, 0:2, :] = half_Tk[:, None, None] * (jac_e @ _DCOV_DE)[None, :, :]
    dcov[:, 2, :] = model_cov / T
    return composed, dcov


class _RaggedGMix(object):
    """device mixtures with a per-stamp gaussian count (duck-types GMixBatch
    for the StampBatch operations)"""

    def __init__(self, data, ngauss):
        self.data = data
        self.ngauss = np.asarray(ngauss, dtype=np.int64)
        self.n = self.ngauss.size


class FitModel(dict):
    """
    The LM objective of one object (all its bands and epochs), and after the
    fit the result dict with statistics and model accessors.
    """

    def __init__(self, obs, model, guess, prior=None):
        self.prior = prior
        self.model = gmix_mod.get_model_num(model)
        self.model_name = gmix_mod.get_model_name(self.model)
        self["model"] = self.model_name
        self.obs = get_mb_obs(obs)
        self.nband = len(self.obs)
        self._flat_obs = [o for obslist in self.obs for o in obslist]
        self.nimage = len(self._flat_obs)
        self._set_npars()
        self._set_n_prior_pars()
        self._set_bounds()
        self._setup_device()
        self.fdiff_size = self.totpix + self.n_prior_pars
        self._setup_fit(guess)

    # ---- sizes
    def _set_npars(self):
        self.npars = gmix_mod.get_model_npars(self.model) + self.nband - 1

    def _set_n_prior_pars(self):
        if self.prior is None:
            self.n_prior_pars = 0
        else:
            self.n_prior_pars = get_lm_n_prior_pars(model=self.model_name,
                                                    nband=self.nband)

    def _set_bounds(self):
        self._bounds = None
        if self.prior is not None and hasattr(self.prior, "bounds"):
            self._bounds = self.prior.bounds

    @property
    def bounds(self):
        return copy.deepcopy(self._bounds)

    # ---- device state
    def _setup_device(self):
        """one StampBatch for every observation of the object"""
        from .batch import StampBatch
        if len(self._flat_obs) == 1:
            # (kept on the Observation: see Observation._device_batch)
            self._batch = self._flat_obs[0]._device_batch()
        else:
            self._batch = StampBatch.from_observations(self._flat_obs)
        self._kept = self._batch.npix_kept.astype(np.int64)
        self.totpix = int(self._kept.sum())
        self._pix_start = np.concatenate([[0], np.cumsum(self._kept)[:-1]]).astype(np.int64)
        self._ierr_cache = None

    @property
    def _ierr_host(self):
        """sqrt(weight) of each listed pixel, in residual order (host copy: the
        MINPACK route's calc_jacobian scales its rows by it; made when first
        asked for -- the batched route never does)"""
        if self._ierr_cache is None:
            ierr = []
            for o in self._flat_obs:
                w = np.asarray(o.weight, dtype="f8").ravel()
                if o.ignore_zero_weight:
                    w = w[w > 0.0]
                ierr.append(np.sqrt(np.where(w > 0.0, w, 0.0)))
            self._ierr_cache = np.concatenate(ierr) if ierr else np.zeros(0)
        return self._ierr_cache

    def _setup_fit(self, guess):
        guess = np.array(guess, dtype="f8")
        assert guess.size == self.npars, (
            "guess has npars=%d, expected %d" % (guess.size, self.npars))
        self.dopsf = self.obs[0][0].has_psf_gmix()
        self._psf_list = []
        self._band_of = []
        for band, obslist in enumerate(self.obs):
            for o in obslist:
                self._psf_list.append(o.psf.gmix if self.dopsf else None)
                self._band_of.append(band)
        ng0 = self._model_ngauss()
        self._ngauss_per_obs = np.array(
            [ng0 * (len(p) if p is not None else 1) for p in self._psf_list],
            dtype=np.int64)
        self._gm_off = np.concatenate([[0], np.cumsum(self._ngauss_per_obs)[:-1]])
        self._gm_host = np.zeros(int(self._ngauss_per_obs.sum()),
                                 dtype=_lib.GAUSS2D_DTYPE)
        self._gm0 = [self._make_model(self.get_band_pars(guess, b))
                     for b in range(self.nband)]
        try:
            self._fill_gmix_all(guess)
        except ZeroDivisionError:
            raise GMixRangeError("got zero division")

    def _model_ngauss(self):
        return gmix_mod.get_model_ngauss(self.model)

    def _make_model(self, band_pars):
        return gmix_mod.make_gmix_model(band_pars, self.model)

    def get_band_pars(self, pars, band):
        return get_band_pars(model=self.model_name, pars=pars, band=band)

    def _fill_gmix_all(self, pars):
        """model fill per band, psf convolution per observation, norms; the
        composed mixtures land back to back in self._gm_host
        (results.py:306-347).  Raises GMixRangeError like the reference."""
        L = _lib.lib()
        for band in range(self.nband):
            self._gm0[band]._fill(self.get_band_pars(pars, band))
        for i, psf in enumerate(self._psf_list):
            gm0 = self._gm0[self._band_of[i]]
            lo = int(self._gm_off[i])
            seg = self._gm_host[lo:lo + int(self._ngauss_per_obs[i])]
            if psf is None:
                seg[:] = gm0._data
            else:
                st = L.ngmix_convolve_fill(_lib.ptr(seg), _lib.ptr(gm0._data),
                                           len(gm0), _lib.ptr(psf._data), len(psf))
                _lib.check(st, "ngmix_convolve_fill")
            st = L.ngmix_set_norms(_lib.ptr(seg), seg.size)
            _lib.check(st, "ngmix_set_norms")

    def _device_gmix(self):
        import torch
        flat = self._gm_host.view(np.float64).reshape(-1, 13)
        return _RaggedGMix(torch.from_numpy(flat.copy()).to(self._batch.device),
                           self._ngauss_per_obs)

    # ---- objective
    def _get_priors(self, pars):
        if self.prior is None:
            return 0.0
        return self.prior.get_lnprob_scalar(pars)

    def _fill_priors(self, pars, fdiff):
        if self.prior is None:
            return 0
        return self.prior.fill_fdiff(pars, fdiff)

    def calc_lnprob(self, pars, more=False):
        """sum of get_loglike over all observations + ln prior
        (results.py:142-210): one kernel launch"""
        try:
            ln_priors = self._get_priors(pars)
            self._fill_gmix_all(pars)
            out, status = self._batch.loglike(self._device_gmix(),
                                              exact=gmix_mod.get_exact_kernels())
            st = status.cpu().numpy()
            if np.any(st != 0):
                _lib.check(int(st[st != 0][0]), "get_loglike")
            o = out.cpu().numpy()
            lnprob = float(np.sum(o[:, 0])) + ln_priors
            s2n_numer = float(np.sum(o[:, 1]))
            s2n_denom = float(np.sum(o[:, 2]))
            npix = int(np.sum(o[:, 3]))
        except GMixRangeError:
            lnprob = LOWVAL
            s2n_numer = 0.0
            s2n_denom = BIGVAL
            npix = 0
        if more:
            return {"lnprob": lnprob, "s2n_numer": s2n_numer,
                    "s2n_denom": s2n_denom, "npix": npix}
        return lnprob

    def calc_fdiff(self, pars):
        """[prior rows | (model-data)/err of obs0 | obs1 ...]
        (results.py:439-466): one kernel launch for all observations"""
        import torch
        fdiff = np.zeros(self.fdiff_size)
        try:
            self._fill_gmix_all(pars)
            start = self._fill_priors(pars=pars, fdiff=fdiff)
            dfd = torch.empty(self.totpix, dtype=torch.float64,
                              device=self._batch.device)
            _, status = self._batch.fill_fdiff(
                self._device_gmix(), fdiff=dfd, fdiff_start=self._pix_start,
                exact=gmix_mod.get_exact_kernels())
            st = status.cpu().numpy()
            if np.any(st != 0):
                _lib.check(int(st[st != 0][0]), "fill_fdiff")
            fdiff[start:start + self.totpix] = dfd.cpu().numpy()
        except GMixRangeError:
            fdiff[:] = LOWVAL
        return fdiff

    def calc_jacobian(self, pars):
        """d calc_fdiff / d pars for gauss / exp / dev (results.py:487-570):
        one deriv_images launch for all observations"""
        if self.model_name not in SIMPLE_ANALYTIC_MODELS:
            raise ValueError("analytic jacobian is not available for model %s"
                             % self.model_name)
        jac = np.zeros((self.fdiff_size, self.npars))
        try:
            start = self._fill_prior_jacobian(pars=pars, jac=jac)
            self._fill_gmix_all(pars)
            gpars_all, dcov_all, fluxes = [], [], []
            for i, psf in enumerate(self._psf_list):
                band = self._band_of[i]
                band_pars = self.get_band_pars(pars=pars, band=band)
                g1, g2, T, flux = band_pars[2:6]
                if T == 0.0 or flux == 0.0:
                    raise GMixRangeError("zero T or flux")
                lo = int(self._gm_off[i])
                gmc = _HostMix(self._gm_host[lo:lo + int(self._ngauss_per_obs[i])])
                gm0 = self._gm0[band]
                gp, dc = get_model_deriv_data(gm0=gm0, gmc=gmc, g1=g1, g2=g2, T=T)
                gpars_all.append(gp)
                dcov_all.append(dc)
                fluxes.append(flux)
            gpars = np.concatenate(gpars_all)
            dcov = np.concatenate(dcov_all)
            out = self._batch.deriv_images(gpars, dcov, self._ngauss_per_obs)
            out = out.cpu().numpy()
            for i in range(self.nimage):
                nk = int(self._kept[i])
                o = out[6 * int(self._pix_start[i]):6 * int(self._pix_start[i]) + 6 * nk]
                o = o.reshape(6, nk)
                ierr = self._ierr_host[int(self._pix_start[i]):int(self._pix_start[i]) + nk]
                sl = slice(start, start + nk)
                for k in range(5):
                    jac[sl, k] = o[1 + k] * ierr
                jac[sl, 5 + self._band_of[i]] = o[0] * (ierr / fluxes[i])
                start += nk
        except GMixRangeError:
            jac[:] = 0.0
        return jac

    def _prior_rows(self, pars):
        """the prior's residual rows at pars, or None where it is undefined"""
        rows = np.zeros(self.n_prior_pars)
        try:
            n = self.prior.fill_fdiff(pars, rows)
        except GMixRangeError:
            return None
        return rows[:n]

    def _fill_prior_jacobian(self, pars, jac):
        """
        The prior rows of the jacobian by one-sided differences of
        prior.fill_fdiff with relative step STEP_PRIOR, forward unless the
        forward point is outside the prior's domain (results.py:572-625).  A
        row that is not finite at either point gets zeros.  Returns the number
        of prior rows.
        """
        if self.prior is None:
            return 0
        base = np.zeros(self.n_prior_pars)
        n = self.prior.fill_fdiff(pars, base)   # a range error here propagates
        base = base[:n]
        steps = STEP_PRIOR * np.maximum(1.0, np.abs(pars))
        shifted = np.empty((n, self.npars))
        for j in range(self.npars):
            for h in (steps[j], -steps[j]):
                trial = np.array(pars, dtype="f8")
                trial[j] = pars[j] + h
                rows = self._prior_rows(trial)
                if rows is not None:
                    break
            else:
                raise GMixRangeError(
                    "prior not evaluable within a step of parameter %d" % j)
            steps[j] = h
            shifted[:, j] = rows
        usable = np.isfinite(base)[:, None] & np.isfinite(shifted)
        with np.errstate(invalid="ignore"):
            slope = (shifted - base[:, None]) / steps[None, :]
        jac[:n, :] = np.where(usable, slope, 0.0)
        return n

    # ---- results
    def _flux_start(self):
        """index of the first flux parameter"""
        return {"bd": 7, "bdf": 6}.get(self["model"], 5)

    def _fit_statistics(self):
        """lnprob / s2n / chi2 of the solution, and the shape, size and flux
        blocks of pars / pars_cov / pars_err under their own keys
        (results.py:45-72, 398-408, 1079-1109)"""
        pars, cov, err = self["pars"], self["pars_cov"], self["pars_err"]
        st = self.calc_lnprob(pars, more=True)
        s2n = st["s2n_numer"] / np.sqrt(st["s2n_denom"]) if st["s2n_denom"] > 0 else 0.0
        dof = st["npix"] - self.npars
        st.update(chi2per=st["lnprob"] / (-0.5) / dof, dof=dof, s2n_w=s2n, s2n=s2n)
        shape = slice(2, 4)
        st.update(g=pars[shape].copy(), g_cov=cov[shape, shape].copy(),
                  g_err=err[shape].copy(), T=pars[4], T_err=np.sqrt(cov[4, 4]))
        return st

    def set_fit_result(self, result):
        self.update(result)
        if self["flags"] != 0:
            return
        self.update(self._fit_statistics())
        self._set_flux()

    def _set_flux(self):
        first = self._flux_start()
        if self.nband > 1:
            block = self["pars_cov"][first:, first:]
            self.update(flux=self["pars"][first:], flux_cov=block,
                        flux_err=np.sqrt(block.diagonal()))
        else:
            self.update(flux=self["pars"][first],
                        flux_err=np.sqrt(self["pars_cov"][first, first]))

    def get_gmix(self, band=0):
        pars = self.get_band_pars(pars=self["pars"], band=band)
        return gmix_mod.make_gmix_model(pars, self.model)

    def get_convolved_gmix(self, band=0, obsnum=0):
        gm = self.get_gmix(band)
        obs = self.obs[band][obsnum]
        if obs.has_psf_gmix():
            gm = gm.convolve(obs.psf.gmix)
        return gm

    def make_image(self, band=0, obsnum=0):
        gm = self.get_convolved_gmix(band=band, obsnum=obsnum)
        obs = self.obs[band][obsnum]
        return gm.make_image(obs.image.shape, jacobian=obs.jacobian)


class _HostMix(object):
    """minimal read-only view so get_model_deriv_data can take a record slice"""

    def __init__(self, data):
        self._data = data

    def get_full_pars(self):
        gm = self._data
        pars = np.zeros(6 * gm.size)
        for k, name in enumerate(("p", "row", "col", "irr", "irc", "icc")):
            pars[k::6] = gm[name]
        return pars


class CoellipFitModel(FitModel):
    """co-elliptical gaussians, single band (results.py:628-674)"""

    def __init__(self, obs, ngauss, guess, prior=None):
        self._ngauss = ngauss
        super().__init__(obs=obs, model="coellip", guess=guess, prior=prior)

    def _model_ngauss(self):
        return self._ngauss

    def _set_flux(self):
        pass

    def _set_n_prior_pars(self):
        assert self.nband == 1, "Coellip can only fit one band"
        self.n_prior_pars = 0 if self.prior is None else 3 + 2 * self._ngauss

    def _set_npars(self):
        self.npars = 4 + 2 * self._ngauss

    def get_band_pars(self, pars, band):
        return np.array(pars, dtype="f8").copy()


class Fitter(object):
    """
    Maximum-likelihood fit of `model` with Levenberg-Marquardt.  .go(obs,
    guess) returns the FitModel (a dict) with flags, nfev, ier, errmsg, pars,
    pars_err, pars_cov0, pars_cov and, when flags == 0, lnprob, s2n, chi2per,
    dof, g, g_cov, g_err, T, T_err, flux, flux_err.
    """

    def __init__(self, model, prior=None, fit_pars=None, use_noise_image=False,
                 analytic_jacobian=True, batched=None):
        self.prior = prior
        self.model = gmix_mod.get_model_num(model)
        self.model_name = gmix_mod.get_model_name(self.model)
        self.use_noise_image = use_noise_image
        self.analytic_jacobian = analytic_jacobian
        self.fit_pars = (fit_pars.copy() if fit_pars is not None
                         else DEFAULT_LM_PARS.copy())
        # batched: run the fit as a one-object batch of the lock-step driver
        # (lm_batch.LMBatchFitter: the whole lmder iteration on the device, ~15
        # launches and two small downloads) instead of MINPACK on the host
        # calling back into one kernel per evaluation (~100 launch + PCIe round
        # trips for an 'exp' fit).  Same algorithm, same nfev / ier; iterates
        # agree to the rounding of the normal-equation factorisation.  None:
        # NGMIX_FITTER_BATCHED (default on); False: the MINPACK path -- what
        # the tests use as the independent check of the batched driver.
        if batched is None:
            batched = os.environ.get("NGMIX_FITTER_BATCHED", "1") not in ("0", "")
        self.batched = bool(batched)
        self._batch_fitter = None
        self._kernel_prior = None

    def go(self, obs, guess):
        guess = np.asarray(guess, dtype="f8")
        fit_model = self._make_fit_model(obs=obs, guess=guess)
        if self.batched and self._go_batched(fit_model, guess):
            return fit_model
        if self.analytic_jacobian and self.model_name in SIMPLE_ANALYTIC_MODELS:
            dfun = fit_model.calc_jacobian
        else:
            dfun = None
        result = run_leastsq(fit_model.calc_fdiff, guess=guess,
                             n_prior_pars=fit_model.n_prior_pars,
                             bounds=fit_model.bounds, Dfun=dfun, **self.fit_pars)
        if self.use_noise_image:
            # noise-power sandwich covariance (fitters.py:108-109)
            from .noise_cov import apply_noise_cov
            apply_noise_cov(fit_model=fit_model, result=result)
        fit_model.set_fit_result(result)
        return fit_model

    def _batched_model(self):
        """(model name, ngauss) for LMBatchFitter, or None when this fit is not
        one it runs"""
        return (self.model_name, None) if self.model_name in (
            "gauss", "exp", "dev", "turb", "bdf", "bd") else None

    def _go_batched(self, fm, guess):
        """the fit as a one-object batch (every epoch / band of the object a
        stamp); fills fm as set_fit_result would and returns True, or returns
        False when the fit is outside what the lock-step driver runs (a prior
        object the prior kernel has no form for, the noise-image covariance,
        psf mixtures of different sizes, more parameters than its state
        holds)"""
        from .lm_batch import LMBatchFitter
        spec = self._batched_model()
        if spec is None or self.use_noise_image or \
                fm.npars > _lib.LM_NPMAX or guess.size != fm.npars:
            return False
        bprior = None
        if self.prior is not None:
            # a joint prior the prior kernel evaluates (joint_prior.py's
            # separable priors of priors.py terms) rides along; any other
            # prior object keeps the fit on the MINPACK route
            if self._kernel_prior is None:
                from .prior_batch import as_batch_prior
                try:
                    cand = as_batch_prior(self.prior)
                except TypeError:
                    cand = None
                has = cand is not None and getattr(cand, "descriptor", lambda: None)() is not None
                self._kernel_prior = cand if has else False
            if self._kernel_prior is False:
                return False
            bprior = self._kernel_prior
            if len(bprior.terms) + 4 != fm.npars:
                return False
        psf = None
        if fm.dopsf:
            if len({len(p) for p in fm._psf_list}) != 1:
                return False
            # (host records: they ride in the driver's one upload per fit)
            psf = np.stack([p._data for p in fm._psf_list])
        if self._batch_fitter is None:
            self._batch_fitter = LMBatchFitter(
                spec[0], fit_pars=self.fit_pars, ngauss=spec[1],
                analytic_jacobian=self.analytic_jacobian, prior=bprior)
        res = self._batch_fitter.go(
            fm._batch, guess[None, :], psf=psf,
            stamp_obj=np.zeros(fm.nimage, dtype=np.int32),
            stamp_band=np.asarray(fm._band_of, dtype=np.int32))
        flags = int(res["flags"][0])
        ier = int(res["ier"][0])
        fm.update(flags=flags, nfev=int(res["nfev"][0]), ier=ier,
                  errmsg="" if flags == 0 else "lmder/lmdif ier %d, flags %d" % (ier, flags),
                  pars=res["pars"][0].copy(), pars_err=res["pars_err"][0].copy(),
                  pars_cov0=res["pars_cov0"][0].copy(), pars_cov=res["pars_cov"][0].copy())
        if flags != 0:
            return True
        pars, cov, err = fm["pars"], fm["pars_cov"], fm["pars_err"]
        fm.update(lnprob=float(res["lnprob"][0]), s2n_numer=float(res["s2n_numer"][0]),
                  s2n_denom=float(res["s2n_denom"][0]), npix=int(res["npix"][0]),
                  chi2per=float(res["chi2per"][0]), dof=int(res["npix"][0]) - fm.npars,
                  s2n_w=float(res["s2n_w"][0]), s2n=float(res["s2n"][0]),
                  g=pars[2:4].copy(), g_cov=cov[2:4, 2:4].copy(), g_err=err[2:4].copy(),
                  T=pars[4], T_err=np.sqrt(cov[4, 4]))
        fm._set_flux()
        return True

    def _make_fit_model(self, obs, guess):
        return FitModel(obs=obs, model=self.model, guess=guess, prior=self.prior)

    def go_many(self, obs, guess):
        """
        The fits of MANY objects as ONE batch of the lock-step driver: what a
        caller of the reference writes as a loop of fitter.go(obs=o, guess=g)
        over a catalogue (runners.py:116-150), without a Python step per
        object between the observations and the device.

        obs: a sequence of Observation / ObsList / MultiBandObsList (one per
            object; every object with the same number of bands)
        guess: (nobj, npars)

        Returns a ManyResults: a sequence of per-object result dicts with the
        keys Fitter.go sets (made when an element is read), over the arrays
        of the batch (.arrays: the LMBatchFitter result).
        """
        from .lm_batch import LMBatchFitter
        from .batch import flatten_observations
        spec = self._batched_model()
        if spec is None or self.use_noise_image:
            raise ValueError("go_many runs the models of LMBatchFitter without the "
                             "noise-image covariance")
        guess = np.ascontiguousarray(np.atleast_2d(guess), dtype="f8")
        stamps, sobj, sband, nband, psf = flatten_observations(obs)
        if guess.shape[0] != len(obs):
            raise ValueError("one guess per object")
        fitter = LMBatchFitter(spec[0], fit_pars=self.fit_pars, ngauss=spec[1],
                               analytic_jacobian=self.analytic_jacobian, prior=self.prior)
        trivial = stamps.n == len(obs) and nband == 1
        res = fitter.go(stamps, guess, psf=psf,
                        stamp_obj=None if trivial else sobj,
                        stamp_band=None if trivial else sband)
        return ManyResults(res, self.model_name, nband)


class ManyResults(object):
    """the per-object result dicts of Fitter.go_many, made on access from the
    batch's arrays (.arrays).  Element i holds what Fitter.go returns for
    object i: flags, nfev, ier, errmsg, pars, pars_err, pars_cov0, pars_cov and
    -- when flags == 0 -- lnprob, s2n_numer, s2n_denom, npix, chi2per, dof,
    s2n_w, s2n, g, g_cov, g_err, T, T_err, flux, flux_err (flux_cov with
    several bands), as FitModel.set_fit_result leaves them (results.py:45-72)"""

    _ALWAYS = ("flags", "nfev", "ier", "pars", "pars_err", "pars_cov0", "pars_cov")
    _STATS = ("lnprob", "s2n_numer", "s2n_denom", "npix", "chi2per", "dof", "s2n_w", "s2n",
              "g", "g_cov", "g_err", "T", "T_err", "flux", "flux_err", "flux_cov")

    def __init__(self, arrays, model, nband):
        self.arrays = arrays
        self.model = model
        self.nband = nband
        self._n = len(arrays["flags"])

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        a = self.arrays
        flags, ier = int(a["flags"][i]), int(a["ier"][i])
        out = {"model": self.model, "flags": flags, "nfev": int(a["nfev"][i]), "ier": ier,
               "errmsg": "" if flags == 0 else "lmder/lmdif ier %d, flags %d" % (ier, flags)}
        for k in ("pars", "pars_err", "pars_cov0", "pars_cov"):
            out[k] = np.array(a[k][i])
        if flags != 0:
            return out
        for k in self._STATS:
            if k in a:
                v = a[k][i]
                out[k] = np.array(v) if isinstance(v, np.ndarray) and v.ndim else v.item()
        return out

    def __iter__(self):
        return (self[i] for i in range(self._n))


class CoellipFitter(Fitter):
    """LM fit of ngauss co-elliptical gaussians"""

    def __init__(self, ngauss, prior=None, fit_pars=None, batched=None):
        self._ngauss = ngauss
        super().__init__(model="coellip", prior=prior, fit_pars=fit_pars, batched=batched)

    def _batched_model(self):
        return ("coellip", self._ngauss) if 4 + 2 * self._ngauss <= _lib.LM_NPMAX else None

    def _make_fit_model(self, obs, guess):
        return CoellipFitModel(obs=obs, ngauss=self._ngauss, guess=guess,
                               prior=self.prior)


# the reference keeps the psf-flux (template amplitude) fitter in its fitting
# package (ngmix/fitting/fitters.py:146-..., results.py:677-914)
from .psfflux import PSFFluxFitter, PSFFluxFitModel  # noqa: E402,F401
