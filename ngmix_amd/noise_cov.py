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

__all__ = ["calc_noise_cov", "apply_noise_cov", "calc_noise_cov_batch",
           "apply_noise_cov_batch"]

# absolute floors of the central-difference steps (results.py:929-936)
STEP_CEN = 1.0e-3
STEP_SHAPE = 1.0e-4
STEP_STRUCT_MIN = 1.0e-4
STEP_FLUX_MIN = 1.0e-6
STEP_FRAC = 1.0e-3


def get_step(pars, ipar, nband):
    """central-difference step of parameter ipar (results.py:939-952):
    fixed for the centre and the shape, 1e-3 of the value with a floor for
    the structural parameters (T, fracdev ...) and for the nband fluxes"""
    if ipar < 4:
        return (STEP_CEN, STEP_CEN, STEP_SHAPE, STEP_SHAPE)[ipar]
    floor = STEP_FLUX_MIN if ipar >= pars.size - nband else STEP_STRUCT_MIN
    return max(floor, STEP_FRAC * abs(pars[ipar]))


def apply_noise_cov(fit_model, result):
    """replace the chi^2-scaled covariance of a successful fit with the
    sandwich covariance (noise_cov.py:36-88); result is modified in place.
    A fit that failed, or whose curvature is not finite, is left alone; a
    sandwich that cannot be evaluated (a central-difference step left the
    model's domain) or is not finite is reported through the covariance flags
    exactly as a negative-definite one is, with the default errors."""
    from .fitting import _test_cov, _get_def_stuff
    curvature = result.get("pars_cov0") if result["flags"] == 0 else None
    if curvature is None or not np.isfinite(curvature).all():
        return
    npars = result["pars"].size
    cov = _sandwich_or_none(fit_model, result["pars"], curvature)
    bits = _test_cov(cov if cov is not None else -np.eye(npars))
    if bits:
        result["flags"] |= bits
        result["errmsg"] = "bad noise covariance matrix"
        result["pars_cov"], result["pars_err"] = _get_def_stuff(npars)[1:]
        return
    result["pars_cov"] = cov
    result["pars_err"] = np.sqrt(cov.diagonal())


def _sandwich_or_none(fit_model, pars, curvature):
    try:
        cov = calc_noise_cov(fit_model=fit_model, pars=pars, pars_cov0=curvature)
    except GMixRangeError:
        return None
    return cov if np.isfinite(cov).all() else None


def calc_noise_cov(fit_model, pars, pars_cov0):
    """pars_cov0 B pars_cov0 (noise_cov.py:91-137).  B sums, over every epoch,
    the noise power spectrum P = |FFT(noise image)|^2 weighted by the transforms
    K_a = FFT(weight * d model / d p_a) of the epoch's derivative images:
    B_ab += Re sum_k conj(K_a) K_b P / npix^2, a and b running over the shape
    parameters and the epoch's own band flux."""
    npars = pars.size
    nband = fit_model.nband
    nshape = npars - nband
    per_obs = iter(_dmodel_images_all(fit_model, pars))
    B = np.zeros((npars, npars))
    for band in range(nband):
        idx = np.r_[np.arange(nshape), nshape + band]
        for obs in fit_model.obs[band]:
            dimages = next(per_obs)
            if not obs.has_noise():
                raise ValueError("use_noise_image needs a noise image in every "
                                 "observation")
            K = np.fft.fft2(obs.weight[None, :, :] * np.asarray(dimages), axes=(1, 2))
            power = np.abs(np.fft.fft2(obs.noise)) ** 2
            # the upper triangle, mirrored: B stays exactly symmetric
            block = np.einsum("aij,bij,ij->ab", K.conj(), K, power).real
            block = np.triu(block) + np.triu(block, 1).T
            B[np.ix_(idx, idx)] += block / float(obs.image.size) ** 2
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
    """d(convolved model image) / d(parameter ipar) by a central difference of
    two fast renders -- the exponential and apodised truncation of the fit's own
    objective (noise_cov.py:200-224); the step is get_step's"""
    h = get_step(pars=pars, ipar=ipar, nband=fit_model.nband)

    def render_at(offset):
        shifted = np.array(pars, dtype="f8")
        shifted[ipar] += offset
        mix = gmix_mod.make_gmix_model(fit_model.get_band_pars(pars=shifted, band=band),
                                       fit_model.model)
        if obs.has_psf_gmix():
            mix = mix.convolve(obs.psf.gmix)
        return mix.make_image(obs.image.shape, jacobian=obs.jacobian, fast_exp=True)

    return (render_at(h) - render_at(-h)) / (2 * h)


# ---------------------------------------------------------------------------
# the same for a device-resident batch of fits (LMBatchFitter results)
# ---------------------------------------------------------------------------

_MODEL_NSHAPE = {"gauss": 5, "turb": 5, "exp": 5, "dev": 5, "bdf": 6, "bd": 7}


def _central_difference_images(geom, model, bp, psf_chunk, nshape):
    """
    The batched form of _dmodel (noise_cov.py:200-224): for every stamp of the
    chunk and every local parameter [shape..., this band's flux] the central
    difference of two fast renders, steps as get_step (results.py:929-952).
    geom: full-frame StampBatch of the chunk; bp: (m, nloc) device tensor of
    band parameters; psf_chunk: GMixBatch or None.
    Returns (D (m, nloc, npix), bad (m,) bool: a shifted model out of range).
    """
    import torch
    from .batch import GMixBatch
    m, nloc = bp.shape
    npix = geom.total_pix // max(m, 1)
    D = torch.empty((m, nloc, npix), dtype=torch.float64, device=bp.device)
    bad = torch.zeros(m, dtype=torch.bool, device=bp.device)
    for k in range(nloc):
        if k < 2:
            step = torch.full((m,), STEP_CEN, dtype=torch.float64, device=bp.device)
        elif k < 4:
            step = torch.full((m,), STEP_SHAPE, dtype=torch.float64, device=bp.device)
        else:
            floor = STEP_STRUCT_MIN if k < nshape else STEP_FLUX_MIN
            step = torch.clamp(STEP_FRAC * bp[:, k].abs(), min=floor)
        ims = []
        for sign in (1.0, -1.0):
            p = bp.clone()
            p[:, k] += sign * step
            gm, st = GMixBatch.from_pars(p, model, device=bp.device)
            bad |= st != 0
            if psf_chunk is not None:
                gm, _ = gm.convolve(psf_chunk)
            im, st2 = geom.render(gm, fast_exp=True)
            bad |= st2 != 0
            ims.append(im.reshape(m, npix))
        D[:, k] = (ims[0] - ims[1]) / (2.0 * step)[:, None]
    return D, bad


def calc_noise_cov_batch(stamps, noise, model, pars, pars_cov0, psf=None,
                         stamp_obj=None, stamp_band=None, chunk_stamps=4096,
                         force_fd=False):
    """
    calc_noise_cov for N fits at once.

    stamps: StampBatch of every epoch of every object (its weights are the
        `obs.weight` of the reference's kernels)
    noise: flat float64 device tensor laid out like stamps.val: each stamp's
        noise image (Observation.noise)
    model: 'gauss' | 'exp' | 'dev' (analytic derivative images, one
        deriv_images launch per chunk) or 'turb' | 'bdf' | 'bd' (central
        differences of fast renders, two launches per parameter and chunk, as
        the reference's _dmodel); force_fd=True takes the second path for
        every model
    pars (nobj, nshape + nband), pars_cov0 (nobj, npars, npars): the solutions
        and their unscaled covariances (LMBatchFitter's 'pars' / 'pars_cov0')
    psf: GMixBatch, one mixture per stamp, or None

    Then rocFFT (torch.fft) of weight x derivative images and of the noise
    images, the per-mode sums as one contraction, and A^-1 B A^-1 per object.
    Returns the (nobj, npars, npars) sandwich covariances as a numpy array
    (NaN where the model is out of range or a flux is zero).
    """
    import torch
    from .batch import GMixBatch, StampBatch, _as_device_f64
    from .fitting import SIMPLE_ANALYTIC_MODELS
    if model not in _MODEL_NSHAPE:
        raise ValueError("calc_noise_cov_batch: model must be one of %s"
                         % (tuple(_MODEL_NSHAPE),))
    analytic = model in SIMPLE_ANALYTIC_MODELS and not force_fd
    dev = stamps.device
    ns = stamps.n
    pars = np.ascontiguousarray(pars, dtype="f8")
    nobj, npars = pars.shape
    nshape = _MODEL_NSHAPE[model]
    nloc = nshape + 1
    nband = npars - nshape
    sobj = (np.arange(ns, dtype=np.int64) if stamp_obj is None
            else np.ascontiguousarray(stamp_obj, dtype=np.int64))
    sband = (np.zeros(ns, dtype=np.int64) if stamp_band is None
             else np.ascontiguousarray(stamp_band, dtype=np.int64))

    # every stamp's band parameters at its object's solution
    bp = np.empty((ns, nloc))
    bp[:, :nshape] = pars[sobj, :nshape]
    bp[:, nshape] = pars[sobj, nshape + sband]
    d_bp = _as_device_f64(bp, dev)
    flux = d_bp[:, nshape]
    gm0, st0 = GMixBatch.from_pars(bp, model, device=dev)
    bad = (st0 != 0) | (flux == 0.0)
    npsf = psf.ngauss if psf is not None else 0
    if analytic:
        gmc = gm0
        if psf is not None:
            gmc, _ = gm0.convolve(psf)
        ng0, G = gm0.ngauss, gmc.ngauss
        G0 = gm0.data.reshape(ns, ng0, 13)
        GC = gmc.data.reshape(ns, G, 13)
        gpars = GC[:, :, 0:6].contiguous()
        modcov = G0[:, :, 3:6].repeat_interleave(G // ng0, dim=1)          # (ns, G, 3)
        # d(irr, irc, icc) / d(g1, g2, T) of each component (results.py:955-1010)
        g1, g2, T = d_bp[:, 2], d_bp[:, 3], d_bp[:, 4]
        gsq = g1 * g1 + g2 * g2
        f = 2.0 / (1.0 + gsq)
        dfac = -f / (1.0 + gsq)
        de1 = torch.stack([f + 2.0 * g1 * g1 * dfac, 2.0 * g1 * g2 * dfac], dim=1)
        de2 = torch.stack([2.0 * g1 * g2 * dfac, f + 2.0 * g2 * g2 * dfac], dim=1)
        Tk = modcov[:, :, 0] + modcov[:, :, 2]
        dcov = torch.zeros((ns, G, 3, 3), dtype=torch.float64, device=dev)
        for i in range(2):
            dcov[:, :, i, 0] = -0.5 * Tk * de1[:, i, None]
            dcov[:, :, i, 1] = 0.5 * Tk * de2[:, i, None]
            dcov[:, :, i, 2] = 0.5 * Tk * de1[:, i, None]
        dcov[:, :, 2, :] = modcov / T[:, None, None]

    wt = stamps.ierr * stamps.ierr
    d_sobj = torch.from_numpy(sobj).to(dev)
    # local parameter k of a stamp -> column of its object's matrix
    cols = np.concatenate([np.tile(np.arange(nshape), (ns, 1)),
                           (nshape + sband)[:, None]], axis=1)
    d_cols = torch.from_numpy(cols).to(dev)
    B = torch.zeros((nobj, npars, npars), dtype=torch.float64, device=dev)

    shapes = np.stack([stamps.nrow, stamps.ncol], axis=1)
    for shp in np.unique(shapes, axis=0):
        members = np.nonzero((shapes == shp).all(axis=1))[0]
        nrow, ncol = int(shp[0]), int(shp[1])
        npix = nrow * ncol
        for a in range(0, members.size, chunk_stamps):
            idx = members[a:a + chunk_stamps]
            m = idx.size
            d_idx = torch.from_numpy(idx).to(dev)
            geom = StampBatch(None, None, stamps.jac[d_idx], np.full(m, nrow),
                              np.full(m, ncol), np.arange(m, dtype=np.int64) * npix, True)
            if analytic:
                out = geom.deriv_images(gpars[d_idx].reshape(-1, 6),
                                        dcov[d_idx].reshape(-1, 3, 3), G)
                out = out.reshape(m, 6, nrow, ncol)
                # [cen1, cen2, g1, g2, T, flux]: the flux derivative is value / flux
                D = torch.cat([out[:, 1:6],
                               (out[:, 0] / flux[d_idx, None, None])[:, None]], dim=1)
            else:
                psf_chunk = None
                if psf is not None:
                    pdata = psf.data.reshape(ns, npsf, 13)[d_idx]
                    psf_chunk = GMixBatch(pdata.reshape(-1, 13).contiguous(), m, npsf)
                D, cbad = _central_difference_images(geom, model, d_bp[d_idx], psf_chunk,
                                                     nshape)
                D = D.reshape(m, nloc, nrow, ncol)
                bad[d_idx] |= cbad
            pix = (torch.from_numpy(stamps.pix_off[idx]).to(dev)[:, None] +
                   torch.arange(npix, device=dev)[None, :])
            W = wt[pix].reshape(m, 1, nrow, ncol)
            K = torch.fft.fft2(W * D)
            P = torch.fft.fft2(noise[pix].reshape(m, nrow, ncol)).abs() ** 2
            Bs = torch.einsum("naxy,nbxy,nxy->nab", K.conj(), K,
                              P.to(K.dtype)).real / float(npix) ** 2
            # scatter the nloc x nloc blocks into the objects' matrices
            oi = d_sobj[d_idx][:, None, None].expand(m, nloc, nloc)
            ci = d_cols[d_idx]
            B.index_put_((oi, ci[:, :, None].expand(m, nloc, nloc),
                          ci[:, None, :].expand(m, nloc, nloc)), Bs, accumulate=True)
    cov0 = _as_device_f64(np.ascontiguousarray(pars_cov0, dtype="f8"), dev)
    cov = cov0 @ B @ cov0
    obad = torch.zeros(nobj, dtype=torch.bool, device=dev)
    obad.index_put_((d_sobj,), bad, accumulate=True)
    cov[obad] = float("nan")
    return cov.cpu().numpy()


def apply_noise_cov_batch(res, stamps, noise, model, psf=None, stamp_obj=None,
                          stamp_band=None):
    """apply_noise_cov for an LMBatchFitter result dict (modified in place):
    the fits with flags == 0 get the sandwich covariance, or the covariance
    sanity flags and default errors when it is not positive"""
    from .defaults import CDEF
    from .flags import LM_NEG_COV_EIG, LM_NEG_COV_DIAG
    ok = res["flags"] == 0
    if not np.any(ok):
        return res
    cov0 = np.where(ok[:, None, None], res["pars_cov0"], 0.0)
    pars = res["pars"].copy()
    pars[~ok] = 1.0  # any in-range point: their rows are not used
    pars[~ok, 2:4] = 0.0
    if model == "bdf":
        pars[~ok, 5] = 0.5
    if model == "bd":
        pars[~ok, 5] = 0.0
        pars[~ok, 6] = 0.5
    cov = calc_noise_cov_batch(stamps, noise, model, pars, cov0, psf=psf,
                               stamp_obj=stamp_obj, stamp_band=stamp_band)
    finite = np.all(np.isfinite(cov), axis=(1, 2))
    sym = 0.5 * (cov + np.transpose(cov, (0, 2, 1)))
    eig = np.linalg.eigvalsh(np.where(finite[:, None, None], sym, np.eye(cov.shape[1])))
    diag = np.diagonal(cov, axis1=1, axis2=2)
    cflags = np.where(~finite | (eig.min(axis=1) < 0), LM_NEG_COV_EIG, 0)
    cflags |= np.where(~finite | np.any(diag < 0, axis=1), LM_NEG_COV_DIAG, 0)
    good = ok & (cflags == 0)
    failed = ok & (cflags != 0)
    pc = np.array(res["pars_cov"])
    pe = np.array(res["pars_err"])
    pc[good] = cov[good]
    with np.errstate(invalid="ignore"):
        pe[good] = np.sqrt(diag[good])
    pc[failed] = CDEF
    pe[failed] = CDEF
    res["flags"] = res["flags"] | np.where(failed, cflags, 0)
    res["pars_cov"], res["pars_err"] = pc, pe
    return res
