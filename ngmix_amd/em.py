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

__all__ = ["run_em", "prep_image", "prep_obs", "EMResult", "EMFitter",
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
