"""
Expectation-maximisation fits (reference API: ngmix/em/em.py).  The iteration
runs in the EM HIP kernel; this module prepares the image sky, the psf
normalisation and the guess convolution as EMFitter.go does, and maps the
kernel status onto EM_RANGE_ERROR / EM_MAXITER.
"""
import logging

import numpy as np

from . import _lib
from .flags import EM_RANGE_ERROR, EM_MAXITER
from .gmix import GMix, GMixModel
from .observation import Observation

__all__ = ["run_em", "run_em_many", "prep_image", "prep_obs", "EMResult", "EMManyResults", "EMFitter",
           "EMFitterFixCen", "EMFitterFixCov", "EMFitterFluxOnly", "fit_em"]

logger = logging.getLogger(__name__)

DEFAULT_TOL = 1.0e-5

_em_conf_dtype = _lib.EM_CONF_DTYPE
_STATUS_MESSAGES = {
    _lib.ERR_DET_TOO_LOW: "'det too low'",
    _lib.ERR_T_TOO_LOW: "'T too low'",
    _lib.ERR_GTOT_ZERO: "'gtot == 0'",
    _lib.ERR_ELOGL_ZERO: "'elogL == 0'",
    _lib.ERR_ZERO_DIV: "division by zero",
}


def run_em(obs, guess, sky=None, fixcen=False, fixcov=False, fluxonly=False,
           **kws):
    """EM fit of one Observation from a GMix guess (pre-psf)"""
    if fixcen:
        fitter = EMFitterFixCen(**kws)
    elif fixcov:
        fitter = EMFitterFixCov(**kws)
    elif fluxonly:
        fitter = EMFitterFluxOnly(**kws)
    else:
        fitter = EMFitter(**kws)
    return fitter.go(obs=obs, guess=guess, sky=sky)


fit_em = run_em


def run_em_many(obs, guess, sky=None, fixcen=False, fixcov=False, fluxonly=False, **kws):
    """run_em over a sequence of Observations as ONE batch (the loop over a
    catalogue the reference's callers write around run_em, em.py:30-107);
    guess: a sequence of GMix (pre-psf), one per observation, all of one size;
    sky: as run_em's (a number or one per observation: the images are taken as
    they are; None: prep_obs' sky per image)"""
    if fixcen:
        fitter = EMFitterFixCen(**kws)
    elif fixcov:
        fitter = EMFitterFixCov(**kws)
    elif fluxonly:
        fitter = EMFitterFluxOnly(**kws)
    else:
        fitter = EMFitter(**kws)
    return fitter.go_many(obs=obs, guess=guess, sky=sky)


def prep_image(im0):
    """shift the image so its minimum is 0.001*(max-min); returns (image, sky)"""
    im = im0.copy()
    im_min = im.min()
    sky = 0.001 * (im.max() - im_min) - im_min
    im += sky
    return im, sky


def prep_obs(obs):
    """copy of the observation with the prep_image sky added; (newobs, sky)"""
    imsky, sky = prep_image(obs.image)
    newobs = obs.copy()
    newobs.image = imsky
    return newobs, sky


class EMResult(dict):
    """EM fit result: flags, numiter, fdiff, sky, message (+ the mixtures)"""

    def __init__(self, obs, result, gm=None, gm_conv=None):
        self._obs = obs
        self.update(result)
        if gm is not None and gm_conv is not None:
            self._gm = gm
            self._gm_conv = gm_conv

    def has_gmix(self):
        return hasattr(self, "_gm")

    def get_gmix(self):
        if not self.has_gmix():
            raise RuntimeError("no gmix set")
        return self._gm.copy()

    def get_convolved_gmix(self):
        if not self.has_gmix():
            raise RuntimeError("no gmix set")
        return self._gm_conv.copy()

    def make_image(self):
        gm = self.get_convolved_gmix()
        return gm.make_image(self._obs.image.shape, jacobian=self._obs.jacobian)


class EMFitter(object):
    """full EM: centres, covariances and fluxes vary"""

    _kind = 0

    def __init__(self, miniter=40, maxiter=500, tol=DEFAULT_TOL, vary_sky=False):
        self.miniter = miniter
        self.maxiter = maxiter
        self.tol = tol
        self.vary_sky = vary_sky

    def _make_conf(self, sky):
        conf = np.zeros(1, dtype=_em_conf_dtype)[0]
        conf["tol"] = self.tol
        conf["miniter"] = self.miniter
        conf["maxiter"] = self.maxiter
        conf["vary_sky"] = self.vary_sky
        conf["sky"] = sky
        return conf

    def go(self, obs, guess, sky=None):
        if not isinstance(obs, Observation):
            raise ValueError("input obs must be an instance of Observation")
        if sky is None:
            obs_sky, sky = prep_obs(obs)
        else:
            obs_sky = obs

        if not obs_sky.has_psf() or not obs_sky.psf.has_gmix():
            logger.debug("NO PSF SET")
            gmix_psf = GMixModel([0.0, 0.0, 0.0, 0.0, 0.0, 1.0], "gauss")
        else:
            gmix_psf = obs_sky.psf.gmix  # a copy
            gmix_psf.set_flux(1.0)

        conf = self._make_conf(sky)
        gm_to_fit = guess.copy()
        gm_conv_to_fit = gm_to_fit.convolve(gmix_psf)
        # zero-weight pixels kept in the list are filled with the model
        fill_zero_weight = (not obs_sky.ignore_zero_weight) and \
            bool(np.any(obs_sky.weight <= 0.0))

        status, numiter, fdiff, sky = obs_sky._device_stamp().em_single(
            self._kind, conf, gm_to_fit.get_data(), gmix_psf.get_data(),
            gm_conv_to_fit.get_data(), fill_zero_weight)

        if status == 0:
            gm = GMix(pars=gm_to_fit.get_full_pars())
            gm_conv = GMix(pars=gm_conv_to_fit.get_full_pars())
            if numiter >= self.maxiter:
                flags, message = EM_MAXITER, "maxit"
            else:
                flags, message = 0, "OK"
            result = {"flags": flags, "numiter": numiter, "fdiff": fdiff,
                      "sky": sky, "message": message}
        elif status in _STATUS_MESSAGES:
            # (GMixRangeError, ZeroDivisionError) caught in em.py:307-315
            gm = gm_conv = None
            message = _STATUS_MESSAGES[status]
            logger.info(message)
            result = {"flags": EM_RANGE_ERROR, "message": message}
        else:
            _lib.check(status, "em_run")
        return EMResult(obs=obs, result=result, gm=gm, gm_conv=gm_conv)


def _em_go_many(self, obs, guess, sky=None):
    """EMFitter.go for MANY Observations by one launch of the batch kernel
    (ngmix_em_batch): prep_obs' sky per stamp, the psf mixtures normalised to
    unit flux (or a delta function where an observation has none), the guess
    convolved, the run kind of this fitter.  Returns an EMManyResults: element i
    is the EMResult go(obs[i], guess[i]) returns"""
    from .batch import StampBatch, GMixBatch
    n = len(obs)
    for o in obs:
        if not isinstance(o, Observation):
            raise ValueError("input obs must be an instance of Observation")
    if len(guess) != n:
        raise ValueError("one guess per observation")
    ng = len(guess[0])
    if any(len(g) != ng for g in guess):
        raise ValueError("the guesses of a batch need one size")
    stamps = StampBatch.from_observations(list(obs))
    if sky is None:
        skyb, sky = stamps.prep_em()
    else:
        skyb = stamps
        sky = np.broadcast_to(np.asarray(sky, dtype="f8"), (n,)).copy()
    grec = np.stack([g._data for g in guess]).reshape(n, ng)
    has = [o.has_psf() and o.psf.has_gmix() for o in obs]
    if any(has) and not all(has):
        raise ValueError("either every observation of a batch carries a psf mixture or none")
    if all(has):
        prec = np.stack([o._psf._gmix._data for o in obs])
        if prec.ndim != 2:
            raise ValueError("the psf mixtures of a batch need one size")
        prec = prec.copy()
        # gmix_psf.set_flux(1.0) (em.py:259; gmix.py set_flux: p *= flux / psum, norms unset)
        psum = prec["p"].sum(axis=1)
        prec["p"] *= (1.0 / psum)[:, None]
        prec["norm_set"] = 0
    else:
        delta = GMixModel([0.0, 0.0, 0.0, 0.0, 0.0, 1.0], "gauss")._data
        prec = np.tile(delta, (n, 1))
    gm = GMixBatch.from_numpy(grec.copy(), device=stamps.device)
    psf = GMixBatch.from_numpy(prec, device=stamps.device)
    fill = bool(np.any([(not o.ignore_zero_weight) and bool(np.any(o.weight <= 0.0))
                        for o in obs]))
    out, status, conv = skyb.em(gm, psf, sky=sky, kind=self._kind, miniter=self.miniter,
                                maxiter=self.maxiter, tol=self.tol, vary_sky=self.vary_sky,
                                fill_zero_weight=fill)
    out = out.cpu().numpy()
    status = status.cpu().numpy()
    for st in np.unique(status):
        if st != 0 and int(st) not in _STATUS_MESSAGES:
            _lib.check(int(st), "em_run")
    numiter = out[:, 0].astype(np.int64)
    flags = np.where(numiter >= self.maxiter, EM_MAXITER, 0)
    return EMManyResults(list(obs), flags, numiter, out[:, 1], out[:, 2], status,
                         gm.to_numpy(), conv.to_numpy(), self.maxiter)


EMFitter.go_many = _em_go_many


class EMManyResults(object):
    """the EMResults of EMFitter.go_many, made on access from the batch's
    arrays"""

    def __init__(self, obs, flags, numiter, fdiff, sky, status, gm, conv, maxiter):
        self._obs = obs
        self.flags, self.numiter, self.fdiff, self.sky = flags, numiter, fdiff, sky
        self._status, self._gm, self._conv = status, gm, conv

    def __len__(self):
        return len(self._obs)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        st = int(self._status[i])
        if st == 0:
            flags = int(self.flags[i])
            result = {"flags": flags, "numiter": int(self.numiter[i]),
                      "fdiff": float(self.fdiff[i]), "sky": float(self.sky[i]),
                      "message": "OK" if flags == 0 else "maxit"}
            # (through full parameters, as go() builds them: det recomputed, norms unset)
            gm = GMix(pars=_full_pars(self._gm[i]))
            gm_conv = GMix(pars=_full_pars(self._conv[i]))
            return EMResult(obs=self._obs[i], result=result, gm=gm, gm_conv=gm_conv)
        message = _STATUS_MESSAGES[st]
        return EMResult(obs=self._obs[i], result={"flags": EM_RANGE_ERROR, "message": message})

    def __iter__(self):
        return (self[i] for i in range(len(self)))


def _full_pars(recs):
    out = np.empty(6 * recs.size)
    for k, name in enumerate(("p", "row", "col", "irr", "irc", "icc")):
        out[k::6] = recs[name]
    return out


class EMFitterFixCen(EMFitter):
    """centres held fixed"""
    _kind = 1


class EMFitterFixCov(EMFitter):
    """covariances held fixed"""
    _kind = 2


class EMFitterFluxOnly(EMFitter):
    """only the fluxes vary (default miniter 20)"""
    _kind = 3

    def __init__(self, miniter=20, maxiter=500, tol=DEFAULT_TOL, vary_sky=False):
        super().__init__(miniter=miniter, maxiter=maxiter, tol=tol,
                         vary_sky=vary_sky)
