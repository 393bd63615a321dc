"""
Gaussian-aperture fluxes of fitted models (reference: ngmix/gaussap.py:1-170,
GMix.get_gaussap_flux, gmix.py:325-390): the flux of each object's (pre-psf)
model seen through a round gaussian weight of the given fwhm.

The reference builds one GMixModel per object and band and inverts two 2x2
matrices per component in Python.  Here the mixtures of all objects are filled
in one launch (ngmix_fill_model_batch) and the aperture factor of a component
with covariance C under the weight W = sigma^2 I is the closed form of what
those inversions compute,

    sqrt(det((C^-1 + W^-1)^-1) / det C) = sigma^2 / sqrt(det(C + W)),

capped at 1 and taken as 1 for a component of vanishing determinant; the flux
is sum_i p_i * factor_i.
"""
import logging

import numpy as np

from . import _lib
from .flags import NO_ATTEMPT, GMIX_RANGE_ERROR
from .moments import fwhm_to_sigma

__all__ = ["get_gaussap_flux", "DEFAULT_FLUX"]

DEFAULT_FLUX = np.nan
GMIX_LOW_DETVAL = 1.0e-200
logger = logging.getLogger(__name__)


def _band_npars(model):
    """parameters of one band's model vector"""
    return 7 if model == "bdf" else 6


def get_gaussap_flux(pars, model, weight_fwhm, fracdev=None, TdByTe=None, mask=None,
                     verbose=True):
    """
    pars: (nobj, npars) fit parameters, the fluxes of all bands at the end
    model: 'gauss', 'exp', 'dev', 'turb', 'bdf', or 'cm' (fracdev and TdByTe,
        one per object, then required)
    weight_fwhm: fwhm of the aperture, in the units of the parameters
    mask: optional (nobj,) bool, False = do not process (flags NO_ATTEMPT)

    Returns (flux (nobj, nband), flags (nobj, nband)): NaN and
    GMIX_RANGE_ERROR where the parameters are not a valid model (|g| >= 1).
    The size is floored at 1e-4 as in the reference.
    """
    from .batch import GMixBatch
    pars = np.array(pars, dtype="f8", ndmin=2)
    nobj = pars.shape[0]
    if mask is not None:
        mask = np.array(mask, dtype=bool, ndmin=1)
        assert mask.shape[0] == nobj, "mask and pars must be same length"
    else:
        mask = np.ones(nobj, dtype=bool)
    extra = None
    if model == "cm":
        fracdev = np.array(fracdev, dtype="f8", ndmin=1)
        TdByTe = np.array(TdByTe, dtype="f8", ndmin=1)
        assert fracdev.size == nobj, "fracdev/pars must be same size"
        assert TdByTe.size == nobj, "TdByTe/pars must be same length"
        extra = np.zeros((nobj, 3))
        extra[:, 0], extra[:, 1] = fracdev, TdByTe
        import ctypes
        tf = ctypes.c_double()
        L = _lib.lib()
        for i in np.nonzero(mask)[0]:
            _lib.check(L.ngmix_get_cm_Tfactor(float(fracdev[i]), float(TdByTe[i]),
                                              ctypes.byref(tf)), "ngmix_get_cm_Tfactor")
            extra[i, 2] = tf.value
    nloc = _band_npars(model)
    nband = pars.shape[1] - nloc + 1
    flags = np.zeros((nobj, nband), dtype="i4")
    flux = np.full((nobj, nband), DEFAULT_FLUX)
    flags[~mask, :] = NO_ATTEMPT
    use = np.nonzero(mask)[0]
    if verbose:
        logger.info("gaussian aperture fluxes of %d objects, %d band(s)" % (use.size, nband))
    if use.size == 0:
        return flux, flags
    sigma2 = fwhm_to_sigma(weight_fwhm) ** 2
    band_pars = np.zeros((use.size, nloc))
    band_pars[:, :nloc - 1] = pars[use, :nloc - 1]
    band_pars[:, 4] = band_pars[:, 4].clip(min=0.0001)
    for band in range(nband):
        band_pars[:, -1] = pars[use, nloc - 1 + band]
        gm, status = GMixBatch.from_pars(band_pars, model,
                                         cm_extra=None if extra is None else extra[use])
        rec = gm.to_numpy()
        bad = status.cpu().numpy() != 0
        det_sum = (rec["irr"] + sigma2) * (rec["icc"] + sigma2) - rec["irc"] ** 2
        with np.errstate(invalid="ignore", divide="ignore"):
            fac = np.minimum(sigma2 / np.sqrt(det_sum), 1.0)
        fac = np.where(rec["det"] > GMIX_LOW_DETVAL, fac, 1.0)
        ap = (rec["p"] * fac).sum(axis=1)
        flux[use, band] = np.where(bad, DEFAULT_FLUX, ap)
        flags[use, band] = np.where(bad, GMIX_RANGE_ERROR, 0)
    return flux, flags
