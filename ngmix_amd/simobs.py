"""
Simulated observations from a mixture (reference: ngmix/simobs.py:1-301): the
model rendered onto the pixel grid of an existing observation (through its psf
mixture unless told otherwise), with gaussian noise drawn from the weight map.
Used for noise images and for checks of a fit against its own model.  The
render is the HIP render kernel (GMix.make_image); the noise is host numpy, in
the reference's order of draws.

    simulate_obs(gmix, obs, add_noise=True, rng=None, add_all=True,
                 noise_factor=None, use_raw_weight=True, convolve_psf=True)

obs may be an Observation, an ObsList (the same mixture for every epoch) or a
MultiBandObsList (gmix: a list, one mixture per band); gmix None gives pure
noise.  The result has the container type of obs; every new Observation keeps
the jacobian, a copy of the weight (scaled by 1 / noise_factor^2) and of the
psf, and carries .noise_image.
"""
import copy
import logging

import numpy as np

from .observation import Observation, ObsList, MultiBandObsList
from .gmix import GMix

__all__ = ["simulate_obs", "get_noise_image", "BIGNOISE"]

LOGGER = logging.getLogger(__name__)
BIGNOISE = 1.0e15


def simulate_obs(gmix, obs, add_noise=True, rng=None, add_all=True, noise_factor=None,
                 use_raw_weight=True, convolve_psf=True):
    opts = dict(add_noise=add_noise, rng=rng, add_all=add_all, noise_factor=noise_factor,
                use_raw_weight=use_raw_weight, convolve_psf=convolve_psf)
    if isinstance(obs, MultiBandObsList):
        return _simulate_bands(gmix, obs, opts)
    if gmix is not None and not isinstance(gmix, GMix):
        raise ValueError("input gmix must be a gaussian mixture")
    if isinstance(obs, ObsList):
        return _simulate_epochs(gmix, obs, opts)
    if isinstance(obs, Observation):
        return _simulate_one(gmix, obs, **opts)
    raise ValueError("obs should be an Observation, ObsList, or MultiBandObsList")


def _simulate_bands(gmix_list, mbobs, opts):
    if gmix_list is not None:
        if not isinstance(gmix_list, list):
            raise ValueError("for simulating MultiBandObsLists, the input must be a list of "
                             "gaussian mixtures")
        if not isinstance(gmix_list[0], GMix):
            raise ValueError("input must be gaussian mixtures")
        if len(gmix_list) != len(mbobs):
            raise ValueError("len(mbobs)==%d but len(gmix_list)==%d" % (len(mbobs),
                                                                        len(gmix_list)))
    out = MultiBandObsList()
    for band, obslist in enumerate(mbobs):
        out.append(_simulate_epochs(None if gmix_list is None else gmix_list[band], obslist,
                                    opts))
    return out


def _simulate_epochs(gmix, obslist, opts):
    out = ObsList()
    for o in obslist:
        # (through the front door, as the reference does: each element is
        # type-checked again)
        out.append(simulate_obs(gmix, o, **opts))
    return out


def _simulate_one(gmix, obs, add_noise, rng, add_all, noise_factor, use_raw_weight,
                  convolve_psf):
    image = _model_image(gmix, obs, convolve_psf)
    noise_image = None
    if add_noise:
        # a fit may run on a weight map edited to mask neighbours; the noise
        # belongs to the unedited one when the observation carries it
        weight = obs.weight_raw if (use_raw_weight and hasattr(obs, "weight_raw")) \
            else obs.weight
        noise_image = get_noise_image(weight=weight, rng=rng, add_all=add_all,
                                      noise_factor=noise_factor)
        image = image + noise_image
    psf = copy.deepcopy(obs.psf) if obs.has_psf() else None
    weight = obs.weight.copy()
    if noise_factor is not None:
        LOGGER.debug("Modding weight with noise factor: %s" % noise_factor)
        weight *= 1.0 / noise_factor ** 2
    new_obs = Observation(image, weight=weight, jacobian=obs.jacobian, psf=psf)
    new_obs.noise_image = noise_image
    return new_obs


def _model_image(gmix, obs, convolve_psf):
    if gmix is None:
        return np.zeros(obs.image.shape)
    if convolve_psf:
        if not obs.has_psf():
            raise RuntimeError("You requested to convolve by the psf, but the observation "
                               "has no psf observation set")
        psf = obs.get_psf()
        if not psf.has_gmix():
            raise RuntimeError("You requested to convolve by the psf, but the observation "
                               "has no psf gmix set")
        gmix = gmix.convolve(psf.gmix)
    return gmix.make_image(obs.image.shape, jacobian=obs.jacobian)


def get_noise_image(weight, rng, add_all=True, noise_factor=None):
    """
    unit normal deviates scaled by 1 / sqrt(weight).  add_all: pixels of zero
    weight get the median error of the others (False: no noise there); every
    weight zero: BIGNOISE everywhere; noise_factor scales the errors.
    """
    if rng is None:
        raise ValueError('you must send an rng to get_noise_image')
    noise_image = rng.normal(loc=0.0, scale=1.0, size=weight.shape)
    err = np.zeros(weight.shape)
    good = weight > 0
    if good.any():
        err[good] = np.sqrt(1.0 / weight[good])
        if add_all and not good.all():
            err[~good] = np.median(err[good])
        if noise_factor is not None:
            LOGGER.debug("Adding noise factor: %s" % noise_factor)
            err *= noise_factor
    else:
        LOGGER.debug("All weight is zero!  Setting noise to %s" % BIGNOISE)
        err[:, :] = BIGNOISE
    noise_image *= err
    return noise_image
