"""
Noise-power sandwich covariance for LM fits under stationary correlated noise
(reference: ngmix/fitting/noise_cov.py:1-224, used by
Fitter(use_noise_image=True), fitters.py:108-109):

    Cov = A^-1 B A^-1,   A^-1 = pars_cov0,
    B_ab = sum_epochs sum_q conj(G_a) G_b |n~(q)|^2 / N^2,
    G_a = fft2(weight * dmodel/dp_a),  n~ = fft2(noise image)

The derivative images of every epoch come from ONE deriv_images launch
(derivs_nb.py:40-127, the same kernel the LM jacobian uses) over full-frame
stamps -- every pixel, masked or not, as the reference evaluates them on
np.mgrid -- or from central differences of fast renders for the models without
analytic derivatives.  The FFTs are O(npix log npix) host numpy, as in the
reference.
"""
import numpy as np

from . import gmix as gmix_mod
from .gexceptions import GMixRangeError

__all__ = ["calc_noise_cov", "apply_noise_cov"]

# absolute floors of the central-difference steps (results.py:929-936)
STEP_CEN = 1.0e-3
STEP_SHAPE = 1.0e-4
STEP_STRUCT_MIN = 1.0e-4
STEP_FLUX_MIN = 1.0e-6
STEP_FRAC = 1.0e-3


def get_step(pars, ipar, nband):
    """results.py:939-952"""
    npars = pars.size
    nshape = npars - nband
    if ipar < 2:
        return STEP_CEN
    elif ipar < 4:
        return STEP_SHAPE
    elif ipar < nshape:
        return max(STEP_STRUCT_MIN, STEP_FRAC * abs(pars[ipar]))
    else:
        return max(STEP_FLUX_MIN, STEP_FRAC * abs(pars[ipar]))


def apply_noise_cov(fit_model, result):
    """replace the chi^2-scaled covariance of a successful fit with the
    sandwich covariance (noise_cov.py:36-88); result is modified in place"""
    from .fitting import _test_cov, _get_def_stuff
    if result["flags"] != 0:
        return
    pcov0 = result.get("pars_cov0")
    if pcov0 is None or not np.all(np.isfinite(pcov0)):
        return
    npars = result["pars"].size
    try:
        cov = calc_noise_cov(fit_model=fit_model, pars=result["pars"], pars_cov0=pcov0)
    except GMixRangeError:
        cov = np.full((npars, npars), np.nan)
    if not np.all(np.isfinite(cov)):
        cflags = _test_cov(np.diag(np.full(npars, -1.0)))
    else:
        cflags = _test_cov(cov)
    if cflags != 0:
        result["flags"] |= cflags
        result["errmsg"] = "bad noise covariance matrix"
        _, result["pars_cov"], result["pars_err"] = _get_def_stuff(npars)
    else:
        result["pars_cov"] = cov
        result["pars_err"] = np.sqrt(np.diag(cov))


def calc_noise_cov(fit_model, pars, pars_cov0):
    """pars_cov0 B pars_cov0 with B from the per-mode noise power of every
    epoch's attached noise image (noise_cov.py:91-137)"""
    npars = pars.size
    nband = fit_model.nband
    nshape = npars - nband
    all_images = _dmodel_images_all(fit_model, pars)
    B = np.zeros((npars, npars))
    i = 0
    for band in range(nband):
        kpars = list(range(nshape)) + [nshape + band]
        for obs in fit_model.obs[band]:
            dimages = all_images[i]
            i += 1
            if not obs.has_noise():
                raise ValueError("use_noise_image needs a noise image in every "
                                 "observation")
            kernels = [np.fft.fft2(obs.weight * dim) for dim in dimages]
            p = np.abs(np.fft.fft2(obs.noise)) ** 2
            n = obs.image.size
            for ia in range(len(kpars)):
                for ib in range(ia, len(kpars)):
                    val = np.sum(np.conj(kernels[ia]) * kernels[ib] * p).real / n ** 2
                    B[kpars[ia], kpars[ib]] += val
                    if ib != ia:
                        B[kpars[ib], kpars[ia]] += val
    return pars_cov0 @ B @ pars_cov0


def _dmodel_images_all(fit_model, pars, force_fd=False):
    """for every observation (bands outer, epochs inner) the derivative images
    of the convolved model with respect to [shape pars..., this band's flux]"""
    from .fitting import SIMPLE_ANALYTIC_MODELS, get_model_deriv_data
    from .batch import StampBatch
    nband = fit_model.nband
    nshape = pars.size - nband
    flat = [(band, obs) for band in range(nband) for obs in fit_model.obs[band]]
    if force_fd or fit_model.model_name not in SIMPLE_ANALYTIC_MODELS:
        return [[_dmodel(fit_model, pars, a, band, obs)
                 for a in list(range(nshape)) + [nshape + band]] for band, obs in flat]

    # analytic: one deriv_images launch over full-frame stamps (no mask)
    gpars_all, dcov_all, ngs, fluxes = [], [], [], []
    for band, obs in flat:
        band_pars = fit_model.get_band_pars(pars=pars, band=band)
        g1, g2, T, flux = band_pars[2:6]
        gm0 = gmix_mod.make_gmix_model(band_pars, fit_model.model)
        gmc = gm0.convolve(obs.psf.gmix) if obs.has_psf_gmix() else gm0
        gp, dc = get_model_deriv_data(gm0=gm0, gmc=gmc, g1=g1, g2=g2, T=T)
        gpars_all.append(gp)
        dcov_all.append(dc)
        ngs.append(gp.shape[0])
        fluxes.append(flux)
    geom = StampBatch.from_observations_geometry([obs for _, obs in flat])
    out = geom.deriv_images(np.concatenate(gpars_all), np.concatenate(dcov_all),
                            np.array(ngs, dtype=np.int64)).cpu().numpy()
    images, start = [], 0
    for (band, obs), flux in zip(flat, fluxes):
        dims = obs.image.shape
        npix = dims[0] * dims[1]
        o = out[start:start + 6 * npix].reshape(6, npix)
        start += 6 * npix
        ims = [o[k].reshape(dims) for k in (1, 2, 3, 4, 5)]
        if flux != 0.0:
            ims.append(o[0].reshape(dims) / flux)
        else:
            ims.append(_dmodel(fit_model, pars, nshape + band, band, obs))
        images.append(ims)
    return images


def _dmodel(fit_model, pars, ipar, band, obs):
    """central difference derivative image of the convolved model with respect
    to one parameter (noise_cov.py:200-224): two fast renders"""
    step = get_step(pars=pars, ipar=ipar, nband=fit_model.nband)
    ims = []
    for sign in (1, -1):
        p = pars.copy()
        p[ipar] += sign * step
        band_pars = fit_model.get_band_pars(pars=p, band=band)
        gm = gmix_mod.make_gmix_model(band_pars, fit_model.model)
        if obs.has_psf_gmix():
            gm = gm.convolve(obs.psf.gmix)
        ims.append(gm.make_image(obs.image.shape, jacobian=obs.jacobian, fast_exp=True))
    return (ims[0] - ims[1]) / (2 * step)
