"""
A device-resident psf -> guess -> object-fit pipeline over N objects: the
batched counterpart of the reference's bootstrap (ngmix/bootstrap.py:24-154:
run the psf runner on every psf observation, store the mixtures of the fits
that passed, DROP the epochs whose psf fit failed, then run the object runner
from a guess; runners.py:116-223 for the attempts of either runner).

    1. the psf fit of every psf stamp, up to psf_ntry attempts each
         'admom'    one gaussian from the adaptive moments (ngmix_admom_batch)
         'em'       psf_ngauss free gaussians by EM       (ngmix_em_batch)
         'coellip'  psf_ngauss co-elliptical gaussians    (LMBatchFitter, lmdif)
         'gauss' | 'turb'  a model fit, Fitter(model=...) (LMBatchFitter)
    2. stamps whose psf fit failed leave the object's fit; an object with a
       band that has no stamp left is flagged BOOT_PSF_FAILURE and not fitted
       (remove_failed_psf_obs raises BootPSFFailure there)
    3. the guess: the caller's (guess=...), or
         'admom'    centre / shape / size from the object's adaptive moments,
                    flux from the pixel sum
         'psfflux'  TPSFFluxGuesser / BDFPSFFluxGuesser (guessers.py:78-145,
                    325-377): the template flux of the fitted psf per band
                    (PSFFluxBatch), a size the caller names, centre and shape
                    near zero
    4. lock-step Levenberg-Marquardt over the kept stamps (LMBatchFitter), up
       to ntry attempts per object

Nothing but the guesses and the result records crosses PCIe.
"""
import numpy as np

from .batch import GMixBatch
from .flags import EM_MAXITER, EM_RANGE_ERROR
from .lm_batch import LMBatchFitter, MODEL_NLOC

__all__ = ["bootstrap_batch", "bootstrap_many", "BOOT_PSF_FAILURE", "BOOT_PSF_FLUX_FAILURE"]

# an object one of whose bands lost every epoch to failed psf fits
# (BootPSFFailure, bootstrap.py:118-154)
BOOT_PSF_FAILURE = 1 << 30
# 'psfflux' guesses: no band of the object has a usable template flux
# (PSFFluxFailure, guessers.py:244-249); the fit still runs, from flux 1
BOOT_PSF_FLUX_FAILURE = 1 << 29

LM_PSF_MODELS = ("gauss", "turb", "coellip")


def _to_device(index, device):
    import torch
    return torch.from_numpy(np.ascontiguousarray(index, dtype=np.int64)).to(device)


def _e1e2_to_g1g2(e1, e2):
    e = np.sqrt(e1 ** 2 + e2 ** 2)
    e = np.minimum(e, 0.999999)
    with np.errstate(invalid="ignore", divide="ignore"):
        fac = np.where(e > 0, np.tanh(0.5 * np.arctanh(e)) / np.where(e > 0, e, 1.0), 0.0)
    return e1 * fac, e2 * fac


def _admom_launch(stamps, Tguess, rng, **conf):
    """start admom on every stamp from a round guess of size Tguess (conf: the
    tolerances / maxiter of StampBatch.admom); nothing is waited for"""
    n = stamps.n
    guess = np.zeros((n, 6))
    guess[:, 0:2] = rng.uniform(-0.1, 0.1, size=(n, 2)) * np.sqrt(Tguess / 2.0)
    guess[:, 4] = Tguess
    guess[:, 5] = 1.0
    wt, _ = GMixBatch.from_pars(guess, "gauss", device=stamps.device)
    res, status = stamps.admom(wt, **conf)
    return wt, res, status


def _admom_collect(launched):
    """the converged weight gaussians (a dict of host arrays row, col, irr,
    irc, icc), the result flags and the kernel status.  Only those columns
    leave the device, not the 584-byte result records."""
    import torch
    wt, res, status = launched
    cols = wt.data[:, 1:6].cpu().numpy()  # row, col, irr, irc, icc of the record
    w = {"row": cols[:, 0], "col": cols[:, 1], "irr": cols[:, 2], "irc": cols[:, 3],
         "icc": cols[:, 4]}
    flags = res[:, 0].contiguous().view(torch.int32)[::2].cpu().numpy()
    return w, {"flags": flags}, status.cpu().numpy()


def _admom_gaussians(stamps, Tguess, rng, **conf):
    return _admom_collect(_admom_launch(stamps, Tguess, rng, **conf))


# starting mixtures of the EM psf fit: flux fractions and size factors relative
# to the adaptive-moments T (a core and wings, as GMixPSFGuesser's em2 / em3
# tables do in guessers.py:899-950)
_EM_PSF_GUESS = {
    1: ([1.0], [1.0]),
    2: ([0.6, 0.4], [0.58, 1.62]),
    3: ([0.55, 0.35, 0.10], [0.5, 1.3, 3.0]),
}


def _normalise(gm, n, ngauss):
    """set_flux(1.0) of a stored psf mixture (em.py:129, and what
    gmix_convolve_fill's division by the psf's p sum amounts to, gmix_nb.py):
    p /= sum p per stamp, in place"""
    import torch
    data = gm.data.reshape(n, ngauss, 13)  # a view: edited in place
    psum = data[:, :, 0].sum(dim=1, keepdim=True)
    psum = torch.where(psum != 0, psum, torch.ones_like(psum))
    data[:, :, 0] /= psum
    gm.set_norms()
    return gm


def _tries(guess, what):
    """a caller's guesses as (ntry, n, npars)"""
    g = np.asarray(guess, dtype="f8")
    if g.ndim == 2:
        g = g[None]
    if g.ndim != 3:
        raise ValueError("%s must be (n, npars) or (ntry, n, npars)" % what)
    return g


def _em_psf_once(psf_stamps, ngauss, full, em_pars):
    """one EM attempt from the full-parameter guesses `full` (n, 6 ngauss)"""
    n = psf_stamps.n
    gm, _ = GMixBatch.from_pars(np.ascontiguousarray(full.reshape(n, -1)), "full",
                                device=psf_stamps.device, ngauss=ngauss)
    delta = np.zeros((n, 6))
    delta[:, 5] = 1.0
    nopsf, _ = GMixBatch.from_pars(delta, "gauss", device=psf_stamps.device)
    skyb, sky = psf_stamps.prep_em()
    pars = dict(miniter=40, maxiter=500, tol=1.0e-5)
    pars.update(em_pars or {})
    out, status, _ = skyb.em(gm, nopsf, sky=sky, **pars)
    out = out.cpu().numpy()
    status = status.cpu().numpy()
    flags = np.where(status != 0, EM_RANGE_ERROR,
                     np.where(out[:, 0] >= pars["maxiter"], EM_MAXITER, 0))
    return gm, flags, out[:, 0].astype(np.int64)


def _em_guess(ngauss, T0, cen, rng):
    n = T0.size
    frac, fac = _EM_PSF_GUESS[ngauss]
    full = np.zeros((n, ngauss, 6))
    for i in range(ngauss):
        sig2 = 0.5 * T0 * fac[i]
        full[:, i, 0] = frac[i] * rng.uniform(0.9, 1.1, size=n)
        full[:, i, 1:3] = cen + rng.uniform(-0.02, 0.02, size=(n, 2)) * np.sqrt(T0)[:, None]
        full[:, i, 3] = sig2 * (1.0 + rng.uniform(-0.1, 0.1, size=n))
        full[:, i, 4] = rng.uniform(-0.05, 0.05, size=n) * sig2
        full[:, i, 5] = sig2 * (1.0 + rng.uniform(-0.1, 0.1, size=n))
    return full.reshape(n, -1)


def _psf_attempts(psf_stamps, ngauss, ntry, first_guess, next_guess, once):
    """PSFRunner's attempts (runners.py:152-223, run_fitter's loop :116-150)
    for a whole batch: every stamp is fitted from first_guess; the stamps
    whose fit ended with flags != 0 are fitted again, as a batch of their own,
    from next_guess(t, index), up to ntry attempts in all.
    once(stamps, guess) -> (GMixBatch, flags, nfev).  Returns the mixtures
    (every stamp's LAST attempt), flags, nfev and the attempts used."""
    n = psf_stamps.n
    gm, flags, nfev = once(psf_stamps, first_guess)
    flags, nfev = flags.copy(), nfev.copy()
    tries = np.ones(n, dtype=np.int64)
    for t in range(1, int(ntry)):
        again = np.nonzero(flags != 0)[0]
        if again.size == 0:
            break
        sub, sflags, snfev = once(psf_stamps.select(again), next_guess(t, again))
        d_idx = _to_device(again, gm.data.device)
        gm.data.reshape(n, ngauss, 13)[d_idx] = sub.data.reshape(again.size, ngauss, 13)
        flags[again], nfev[again] = sflags, snfev
        tries[again] += 1
    return gm, flags, nfev, tries


def _lm_psf_guess(kind, ngauss, T0, cen, g, flux, rng):
    """the starting point of an LM psf fit from the adaptive moments (centre,
    shape, size) and the pixel sum, perturbed as the reference's psf guessers
    perturb theirs (SimplePSFGuesser / CoellipPSFGuesser, guessers.py:1054-1243)"""
    n = T0.size
    if kind != "coellip":
        guess = np.zeros((n, 6))
        guess[:, 0:2], guess[:, 2:4] = cen, g
        guess[:, 4] = T0 * (1.0 + rng.uniform(-0.05, 0.05, size=n))
        guess[:, 5] = flux * (1.0 + rng.uniform(-0.05, 0.05, size=n))
        return guess
    frac, fac = _EM_PSF_GUESS[ngauss]
    guess = np.zeros((n, 4 + 2 * ngauss))
    guess[:, 0:2] = cen
    guess[:, 2:4] = g
    for i in range(ngauss):
        guess[:, 4 + i] = T0 * fac[i] * (1.0 + rng.uniform(-0.05, 0.05, size=n))
        guess[:, 4 + ngauss + i] = flux * frac[i]
    return guess


def _pixel_sums(stamps):
    """sum of every stamp's pixels (host array)"""
    import torch
    n = stamps.n
    npix = stamps.npix.astype(np.int64)
    if n and np.all(npix == npix[0]) and stamps._packed():
        return stamps.val.reshape(n, -1).sum(dim=1).cpu().numpy()
    vals = stamps.val
    if not stamps._packed():
        # (a selection shares its parent's pixel arrays: gather its own pixels)
        start = np.concatenate([[0], np.cumsum(npix)[:-1]])
        flat = np.repeat(stamps.pix_off - start, npix) + np.arange(int(npix.sum()))
        vals = vals[torch.from_numpy(flat).to(stamps.device)]
    return torch.segment_reduce(
        vals, "sum", lengths=torch.from_numpy(npix).to(stamps.device)).cpu().numpy()


def _lm_psf_once(kind, ngauss, fit_pars):
    """once() of _psf_attempts for the LM psf fitters: Fitter(model=kind) /
    CoellipFitter(ngauss) on the psf stamps, no psf of their own"""
    def once(stamps, guess):
        fitter = LMBatchFitter(kind, ngauss=ngauss if kind == "coellip" else None,
                               fit_pars=fit_pars)
        res = fitter.go(stamps, guess)
        return fitter.gmix, res["flags"], res["nfev"]
    return once


def _mixture_moments(gm, n, ngauss):
    """T and (g1, g2) of each stamp's whole mixture (GMix.get_T / get_g1g2T:
    flux-weighted second moments about the mixture's centre are not needed
    here -- the gaussians of a psf fit share or nearly share their centre)"""
    d = gm.data.reshape(n, ngauss, 13)
    p = d[:, :, 0]
    psum = p.sum(dim=1)
    irr = (p * d[:, :, 3]).sum(dim=1) / psum
    irc = (p * d[:, :, 4]).sum(dim=1) / psum
    icc = (p * d[:, :, 5]).sum(dim=1) / psum
    irr, irc, icc = (t.cpu().numpy() for t in (irr, irc, icc))
    T = irr + icc
    with np.errstate(invalid="ignore", divide="ignore"):
        e1, e2 = (icc - irr) / T, 2.0 * irc / T
    bad = ~np.isfinite(e1) | ~np.isfinite(e2) | ~(T > 0)
    e1, e2 = np.where(bad, 0.0, e1), np.where(bad, 0.0, e2)
    g1, g2 = _e1e2_to_g1g2(e1, e2)
    return T, g1, g2


def _psfflux_guess(model, nobj, nband, Tguess, flux, rng, prior=None):
    """TPSFFluxGuesser.__call__ (guessers.py:107-145) for every object at
    once; for 'bdf' / 'bd' the extra columns as BDFPSFFluxGuesser / BDGuesser
    fill them (guessers.py:344-377, 451-487: fracdev in [0.4, 0.6]).

    prior: a host joint prior (joint_prior.py: it has .sample): the guesses
    are TPSFFluxAndPriorGuesser's (guessers.py:147-203) -- BDFPSFFluxGuesser's
    for 'bdf' / 'bd' -- for every object at once: centre and shape (for the
    bulge+disk models every parameter) drawn from the prior, the size within
    10 % of Tguess, the fluxes within 10 % of the psf fluxes, and a guess the
    prior gives no probability replaced by a draw from it.  Without one the
    centre and shape start near zero."""
    nshape = MODEL_NLOC[model] - 1
    if prior is not None:
        return _psfflux_prior_guess(model, nobj, nband, nshape, Tguess, flux, rng, prior)
    guess = np.zeros((nobj, nshape + nband))
    guess[:, 0] = rng.uniform(-0.01, 0.01, size=nobj)
    guess[:, 1] = rng.uniform(-0.01, 0.01, size=nobj)
    guess[:, 2] = rng.uniform(-0.02, 0.02, size=nobj)
    guess[:, 3] = rng.uniform(-0.02, 0.02, size=nobj)
    guess[:, 4] = Tguess * rng.uniform(0.9, 1.1, size=nobj)
    if model == "bdf":
        guess[:, 5] = rng.uniform(0.4, 0.6, size=nobj)
    elif model == "bd":
        guess[:, 5] = rng.uniform(-0.1, 0.1, size=nobj)     # log10(Tdev / Texp)
        guess[:, 6] = rng.uniform(0.4, 0.6, size=nobj)
    for b in range(nband):
        guess[:, nshape + b] = flux[:, b] * rng.uniform(0.9, 1.1, size=nobj)
    return guess


def _psfflux_prior_guess(model, nobj, nband, nshape, Tguess, flux, rng, prior):
    bulge_disk = model in ("bdf", "bd")
    if bulge_disk:
        rng = prior.cen_prior.rng            # BDFPSFFluxGuesser draws from the prior's
    guess = np.array(prior.sample(nobj), dtype="f8")
    if guess.shape != (nobj, nshape + nband):
        raise ValueError("the prior samples %d parameters, model '%s' with %d band(s) has %d"
                         % (guess.shape[1], model, nband, nshape + nband))
    guess[:, 4] = Tguess * (1.0 + rng.uniform(low=-0.1, high=0.1, size=nobj))
    if bulge_disk:
        # column 5 for both models, as the reference's BDGuesser inherits it
        guess[:, 5] = rng.uniform(low=0.4, high=0.6, size=nobj)
    for b in range(nband):
        if bulge_disk:
            guess[:, nshape + b] = flux[:, b] * (1.0 + rng.uniform(low=-0.1, high=0.1, size=nobj))
        else:
            guess[:, nshape + b] = flux[:, b] * rng.uniform(low=0.9, high=1.1, size=nobj)
    # _fix_guess / _fix_guess_TFlux: the rows the prior rejects, found for all
    # objects at once through the batch form, repaired one by one
    import torch
    from .prior_batch import as_batch_prior
    from .guessers import _fix_guess
    lnp = as_batch_prior(prior).get_lnprob_batch(torch.from_numpy(guess)).numpy()
    for j in np.nonzero(~(lnp > -np.inf))[0]:
        _fix_guess(guess[j:j + 1], prior, keep_shape=not bulge_disk)
    return guess


def psf_fluxes(stamps, psf_gm, sobj, sband, nobj, nband, rng):
    """_get_psf_fluxes (guessers.py:205-262) for every object at once: the
    template flux of the psf mixtures per (object, band) over that band's
    stamps (PSFFluxFitter on the band's ObsList); a band whose fit is flagged
    or not finite takes the mean of the object's good bands times 1 + U(-0.1,
    0.1); an object with no good band keeps flux 1 and is reported in `none`.
    Returns (flux (nobj, nband), flags (nobj, nband), none (nobj,))"""
    from .psfflux import PSFFluxBatch
    key = sobj * nband + sband
    res = PSFFluxBatch().go(stamps, psf_gm, stamp_obj=key, nobj=nobj * nband)
    flux = res["flux"].reshape(nobj, nband).copy()
    flags = res["flags"].reshape(nobj, nband)
    # (a band the caller's maps leave without a stamp has msq == 0: DIV_ZERO)
    good = (flags == 0) & np.isfinite(flux)
    ngood = good.sum(axis=1)
    mean = np.where(good, flux, 0.0).sum(axis=1) / np.maximum(ngood, 1)
    bad_o, bad_b = np.nonzero(~good)
    fac = 1.0 + rng.uniform(-0.1, 0.1, size=bad_o.size)
    flux[bad_o, bad_b] = np.where(ngood[bad_o] > 0, mean[bad_o] * fac, 1.0)
    return flux, flags, ngood == 0


def bootstrap_batch(stamps, psf_stamps, model="exp", psf_Tguess=0.3, Tguess=None,
                    fit_pars=None, rng=None, psf_ngauss=1, em_pars=None, prior=None,
                    stamp_obj=None, stamp_band=None, ntry=1, psf_fitter=None,
                    psf_ntry=1, guess_admom=None, psf_guess=None, psf_fit_pars=None,
                    guess=None, guesser="admom", drop_failed_psf=True):
    """
    stamps, psf_stamps: StampBatch of the object images and of their psf images
        (stamp i of one belongs to stamp i of the other)
    model: 'gauss' | 'exp' | 'dev' (lmder), 'turb' | 'bdf' | 'bd' (lmdif): any
        model LMBatchFitter fits, as the reference's Bootstrapper takes any
        fitter (bootstrap.py:24-66)
    psf_fitter: 'admom' (one gaussian: the adaptive moments themselves), 'em'
        (psf_ngauss free gaussians; em_pars: miniter / maxiter / tol), 'coellip'
        (CoellipFitter(psf_ngauss)), 'gauss' | 'turb' (Fitter(model=...));
        psf_fit_pars: the LM psf fitters' fit_pars.  Default: 'admom' for one
        gaussian, 'em' for more
    psf_ntry: attempts per psf fit (PSFRunner's ntry)
    psf_guess: (nstamps, npars) or (ntry, nstamps, npars) starting points of
        the psf fits, one per attempt -- the fitter's own parameters ('em':
        full [p, row, col, irr, irc, icc] per gaussian).  Default: from the
        adaptive moments of the psf stamp (psf_Tguess: their starting size)
    drop_failed_psf: stamps whose psf fit failed leave the object's fit, an
        object with an empty band is flagged BOOT_PSF_FAILURE and not fitted
        (the reference's ignore_failed_psf=True, bootstrap.py:105-154).  False:
        the object fit runs on every stamp (a failed psf keeps its last
        mixture) and the object is flagged BOOT_PSF_FAILURE
    guess: (nobj, npars) or (ntry, nobj, npars) starting points of the object
        fits, one per attempt; default: guesser
    guesser: 'admom' (Tguess: the starting size of the moments, default 2 x
        psf_Tguess; guess_admom: etol / Ttol / maxiter of that stage) or
        'psfflux' (Tguess: the size the guesses scatter around, as
        TPSFFluxGuesser's T; with a host joint prior the guesses are
        TPSFFluxAndPriorGuesser's / BDFPSFFluxGuesser's: centre and shape
        drawn from the prior)
    ntry: fits that end with flags != 0 are repeated from the next guess (a
        perturbed one without `guess`) up to ntry times in all, as Runner does
        object by object (runners.py:95-150); 'ntry' of the result counts the
        attempts
    prior: the joint prior of the object fits: a joint_prior.PriorSimpleSep
        (...) of priors.py terms as a caller of the reference builds it, or a
        batch prior (prior_batch.PriorSimpleSepBatch ...)
    stamp_obj / stamp_band: as for LMBatchFitter.go -- objects with several
        epochs and bands (a MultiBandObsList each).  Every stamp gets its own
        psf fit.

    Returns a dict: the LMBatchFitter result arrays per object (objects that
    were not fitted: flags = BOOT_PSF_FAILURE, nfev 0, NaN), plus per stamp
    'psf_flags', 'psf_nfev', 'psf_ntry', 'psf_T', 'psf_g', 'kept', 'psf_gmix'
    and per object 'guess' (the last attempt's start), 'guess_flags',
    'psf_flux' (guesser='psfflux'), 'ntry', 'rounds'.
    """
    assert stamps.n == psf_stamps.n
    if rng is None:
        rng = np.random.RandomState(0)
    if model not in MODEL_NLOC:
        raise ValueError("bootstrap_batch fits %s" % (tuple(MODEL_NLOC),))
    if psf_fitter is None:
        psf_fitter = "admom" if psf_ngauss == 1 else "em"
    if psf_fitter == "em" and psf_ngauss == 1 and psf_guess is None:
        psf_fitter = "admom"   # (one free gaussian by EM = the adaptive moments' job)
    if psf_fitter not in ("admom", "em") + LM_PSF_MODELS:
        raise ValueError("psf_fitter: 'admom', 'em', 'coellip', 'gauss' or 'turb'")
    if psf_fitter in ("gauss", "turb"):
        psf_ngauss = {"gauss": 1, "turb": 3}[psf_fitter]
    if psf_fitter == "admom" and psf_guess is not None:
        raise ValueError("the adaptive-moments psf takes no psf_guess")
    n = stamps.n
    if Tguess is None:
        Tguess = 2.0 * psf_Tguess
    if stamp_obj is None:
        sobj = np.arange(n, dtype=np.int64)
        sband = np.zeros(n, dtype=np.int64)
    else:
        sobj = np.ascontiguousarray(stamp_obj, dtype=np.int64)
        sband = (np.zeros(n, dtype=np.int64) if stamp_band is None
                 else np.ascontiguousarray(stamp_band, dtype=np.int64))
    nobj = int(sobj.max()) + 1
    nband = int(sband.max()) + 1
    nshape = MODEL_NLOC[model] - 1
    if guess is not None:
        guess = _tries(guess, "guess")
        if guess.shape[1:] != (nobj, nshape + nband):
            raise ValueError("guess must be (ntry, %d, %d)" % (nobj, nshape + nband))
    use_admom_guess = guess is None and guesser == "admom"
    if guess is None and guesser not in ("admom", "psfflux"):
        raise ValueError("guesser: 'admom' or 'psfflux'")

    # the guess stage's tolerances: it only has to put the fit inside its basin
    # -- ten times the measurement defaults stop the iteration two or three
    # passes earlier and the LM needs no more rounds for it
    gconf = dict(etol=1.0e-4, Ttol=1.0e-2, no_cov=True)   # (only the weight is read)
    gconf.update(guess_admom or {})

    # ---- 1. the psf fits
    # With a one-gaussian adaptive-moments psf nothing between here and the
    # object's adaptive moments draws random numbers or needs the psf result,
    # so that launch (and the pixel sums of the flux guess) are queued right
    # behind: the device works through them while the host turns the psf
    # moments into mixtures
    o_launched = None
    d_flux = None
    need_moments = psf_guess is None
    pst = np.zeros(n, dtype=np.int64)
    pflags_admom = np.zeros(n, dtype=np.int64)
    if need_moments:
        p_launched = _admom_launch(psf_stamps, psf_Tguess, rng)
        if psf_fitter == "admom" and use_admom_guess:
            o_launched = _admom_launch(stamps, Tguess, rng, **gconf)
            if np.all(stamps.npix == stamps.npix[0]) and stamps._packed():
                d_flux = stamps.val.reshape(n, -1).sum(dim=1)
        pw, prec, pst = _admom_collect(p_launched)
        pflags_admom = prec["flags"]
        psf_bad = (pflags_admom != 0) | (pst != 0)
        psf_T = np.where(psf_bad, psf_Tguess, pw["irr"] + pw["icc"])
        pe1 = np.where(psf_bad, 0.0, (pw["icc"] - pw["irr"]) / psf_T)
        pe2 = np.where(psf_bad, 0.0, 2.0 * pw["irc"] / psf_T)
        pg1, pg2 = _e1e2_to_g1g2(pe1, pe2)
        cen = np.where(psf_bad[:, None], 0.0, np.stack([pw["row"], pw["col"]], axis=1))
    fit_flags = np.zeros(n, dtype=np.int64)
    psf_nfev = np.zeros(n, dtype=np.int64)
    psf_tries = np.ones(n, dtype=np.int64)
    if psf_fitter == "admom":
        psf_pars = np.zeros((n, 6))
        psf_pars[:, 2], psf_pars[:, 3], psf_pars[:, 4], psf_pars[:, 5] = pg1, pg2, psf_T, 1.0
        psf_gm, _ = GMixBatch.from_pars(psf_pars, "gauss", device=stamps.device)
    else:
        if psf_guess is not None:
            pguess = _tries(psf_guess, "psf_guess")
            if pguess.shape[1] != n:
                raise ValueError("psf_guess needs one row per stamp")
            first = pguess[0]
            nxt = lambda t, idx: pguess[min(t, pguess.shape[0] - 1)][idx]  # noqa: E731
        elif psf_fitter == "em":
            first = _em_guess(psf_ngauss, psf_T, cen, rng)
            nxt = lambda t, idx: _em_guess(psf_ngauss, psf_T[idx], cen[idx], rng)  # noqa: E731
        else:
            area = (psf_stamps.jac[:, 7] ** 2).cpu().numpy()
            pflux = _pixel_sums(psf_stamps)
            pflux = np.where(pflux > 0, pflux * area, 1.0)
            gg = np.stack([pg1, pg2], axis=1)
            first = _lm_psf_guess(psf_fitter, psf_ngauss, psf_T, cen, gg, pflux, rng)
            nxt = lambda t, idx: _lm_psf_guess(  # noqa: E731
                psf_fitter, psf_ngauss, psf_T[idx], cen[idx], gg[idx], pflux[idx], rng)
        if psf_fitter == "em":
            once = lambda s, g: _em_psf_once(s, psf_ngauss, g, em_pars)  # noqa: E731
        else:
            once = _lm_psf_once(psf_fitter, psf_ngauss, psf_fit_pars)
        psf_gm, fit_flags, psf_nfev, psf_tries = _psf_attempts(
            psf_stamps, psf_ngauss, psf_ntry, first, nxt, once)
        _normalise(psf_gm, n, psf_ngauss)
        if need_moments:
            psf_bad = psf_bad | (fit_flags != 0)
        else:
            psf_bad = fit_flags != 0
            psf_T, pg1, pg2 = _mixture_moments(psf_gm, n, psf_ngauss)
            psf_T = np.where(psf_bad | ~np.isfinite(psf_T), psf_Tguess, psf_T)

    # ---- 2. the epochs whose psf fit failed leave the fit
    # (remove_failed_psf_obs, bootstrap.py:105-154)
    keep = np.ones(n, dtype=bool)
    boot_failed = np.zeros(nobj, dtype=bool)
    if drop_failed_psf and psf_bad.any():
        keep = ~psf_bad
        left = np.bincount(sobj * nband + sband, weights=keep.astype("f8"),
                           minlength=nobj * nband).reshape(nobj, nband)
        have = np.bincount(sobj * nband + sband, minlength=nobj * nband).reshape(nobj, nband)
        boot_failed = np.any((left == 0) & (have > 0), axis=1)
        keep &= ~boot_failed[sobj]
    all_kept = bool(keep.all())
    fit_obj = np.nonzero(~boot_failed)[0]          # the objects that are fitted

    # ---- 3. the guess
    guess_flags = np.zeros(n, dtype=np.int64)
    psf_flux = None
    out_guess = np.full((nobj, nshape + nband), np.nan)
    if guess is not None:
        first_guess = guess[0]
        next_guess = lambda t, objs: guess[min(t, guess.shape[0] - 1)][objs]  # noqa: E731
    elif guesser == "psfflux":
        kidx = np.nonzero(keep)[0]
        ks = stamps if all_kept else stamps.select(kidx)
        kp = psf_gm if all_kept else psf_gm.select(kidx)
        psf_flux, pf_flags, none_good = psf_fluxes(ks, kp, sobj[kidx], sband[kidx],
                                                  nobj, nband, rng)
        # (a host joint prior also shapes the guesses, as the reference's
        # prior-drawing psf-flux guessers do)
        gprior = prior if hasattr(prior, "sample") else None
        first_guess = _psfflux_guess(model, nobj, nband, Tguess, psf_flux, rng, gprior)
        next_guess = lambda t, objs: _psfflux_guess(  # noqa: E731
            model, nobj, nband, Tguess, psf_flux, rng, gprior)[objs]
    else:
        # adaptive moments of the object stamps, psf size taken out
        if o_launched is None:
            o_launched = _admom_launch(stamps, Tguess, rng, **gconf)
        ow, orec, ost = _admom_collect(o_launched)
        guess_flags = np.where(ost != 0, -1, orec["flags"])
        gbad = (orec["flags"] != 0) | (ost != 0)
        T_obs = np.where(gbad, 2.0 * psf_T, ow["irr"] + ow["icc"])
        e1 = np.where(gbad, 0.0, (ow["icc"] - ow["irr"]) / T_obs)
        e2 = np.where(gbad, 0.0, 2.0 * ow["irc"] / T_obs)
        g1, g2 = _e1e2_to_g1g2(e1, e2)
        flux = d_flux.cpu().numpy() if d_flux is not None else _pixel_sums(stamps)
        # (only the stamps that stay in the fit speak)
        wk = keep.astype("f8")
        good = np.where(gbad, 0.0, wk)
        ngood = np.bincount(sobj, weights=good, minlength=nobj)

        def obj_mean(x, default):
            tot = np.bincount(sobj, weights=np.where(gbad, 0.0, x) * wk, minlength=nobj)
            return np.where(ngood > 0, tot / np.maximum(ngood, 1.0), default)
        Tdiff = np.maximum(T_obs - psf_T, 0.1 * psf_T)
        first_guess = np.zeros((nobj, nshape + nband))
        first_guess[:, 0] = obj_mean(ow["row"], 0.0)
        first_guess[:, 1] = obj_mean(ow["col"], 0.0)
        first_guess[:, 2], first_guess[:, 3] = obj_mean(g1, 0.0), obj_mean(g2, 0.0)
        # (stamps whose moments failed carry the neutral size 2 T_psf - T_psf)
        first_guess[:, 4] = np.bincount(sobj, weights=Tdiff * wk, minlength=nobj) / \
            np.maximum(np.bincount(sobj, weights=wk, minlength=nobj), 1)
        if model == "bdf":
            first_guess[:, 5] = 0.5
        elif model == "bd":
            first_guess[:, 5], first_guess[:, 6] = 0.0, 0.5
        key = sobj * nband + sband
        fsum = np.bincount(key, weights=flux * wk, minlength=nobj * nband)
        fcnt = np.bincount(key, weights=wk, minlength=nobj * nband)
        fmean = (fsum / np.maximum(fcnt, 1)).reshape(nobj, nband)
        first_guess[:, nshape:] = np.where(fmean > 0, fmean, 1.0)

        def next_guess(t, objs):
            g0 = out_guess[objs]
            g2_ = g0 * (1.0 + 0.1 * rng.uniform(-1, 1, size=g0.shape))
            g2_[:, 0:2] = g0[:, 0:2] + 0.05 * rng.uniform(-1, 1, size=(objs.size, 2)) * \
                np.sqrt(g0[:, 4:5])
            return g2_

    # ---- 4. the fits, on the stamps that are left
    fitter = LMBatchFitter(model, fit_pars=fit_pars, prior=prior)
    multi = stamp_obj is not None
    if all_kept:
        fstamps, fpsf, fsobj, fsband = stamps, psf_gm, sobj, sband
        sidx_all = np.arange(n)
    else:
        sidx_all = np.nonzero(keep)[0]
        fstamps, fpsf = stamps.select(sidx_all), psf_gm.select(sidx_all)
        # (objects renumbered without the ones that are not fitted)
        fsobj = np.searchsorted(fit_obj, sobj[sidx_all])
        fsband = sband[sidx_all]
        multi = True
    out_guess[fit_obj] = first_guess[fit_obj]
    res = None
    tries = np.zeros(nobj, dtype=np.int64)
    if fit_obj.size:
        res = fitter.go(fstamps, first_guess[fit_obj], psf=fpsf,
                        stamp_obj=fsobj.astype(np.int32) if multi else None,
                        stamp_band=fsband.astype(np.int32) if multi else None)
        tries[fit_obj] = 1
    for t in range(1, int(ntry)):
        if res is None:
            break
        redo = np.nonzero(res["flags"] != 0)[0]     # (positions among the fitted)
        if redo.size == 0:
            break
        # the failed objects' stamps as a batch of their own
        member = np.isin(fsobj, redo)
        sidx = np.nonzero(member)[0]
        sub_obj = np.searchsorted(redo, fsobj[sidx]).astype(np.int32)
        g2_ = np.ascontiguousarray(next_guess(t, fit_obj[redo]))
        sub = fitter.go(fstamps.select(sidx), g2_, psf=fpsf.select(sidx), stamp_obj=sub_obj,
                        stamp_band=fsband[sidx].astype(np.int32))
        tries[fit_obj[redo]] += 1
        # (items() reads through the keys an LMBatchResult keeps on the
        # device, pars_cov0 among them: the retried objects take every array
        # of their new fit)
        for k, v in sub.items():
            if isinstance(v, np.ndarray) and v.shape[:1] == (redo.size,) and k in res:
                if not res[k].flags.writeable:
                    res[k] = np.array(res[k])
                res[k][redo] = v
        out_guess[fit_obj[redo]] = g2_
    if res is None:
        res = {"model": model}
    if boot_failed.any():
        res = _scatter(res, fit_obj, nobj, nshape + nband)
    res["ntry"] = tries
    if drop_failed_psf:
        res["flags"] = np.where(boot_failed, BOOT_PSF_FAILURE, res["flags"])
    else:
        obj_psf_bad = np.bincount(sobj, weights=psf_bad.astype("f8"), minlength=nobj) > 0
        res["flags"] = res["flags"] | np.where(obj_psf_bad, BOOT_PSF_FAILURE, 0)
    if psf_flux is not None:
        res["flags"] = res["flags"] | np.where(none_good & ~boot_failed,
                                               BOOT_PSF_FLUX_FAILURE, 0)
        res["psf_flux"] = psf_flux
        res["psf_flux_flags"] = pf_flags
    res["psf_T"] = psf_T
    res["psf_g"] = np.stack([pg1, pg2], axis=1)
    # per stamp: the psf fit's flags (-1: the adaptive moments' kernel refused
    # the stamp), the evaluations / iterations and the attempts it took
    res["psf_flags"] = np.where(pst != 0, -1, pflags_admom | fit_flags) \
        if need_moments else fit_flags
    res["psf_em_flags"] = fit_flags
    res["psf_nfev"] = psf_nfev
    res["psf_ntry"] = psf_tries
    res["psf_gmix"] = psf_gm
    res["kept"] = keep
    res["boot_failed"] = boot_failed
    res["guess"] = out_guess
    res["guess_flags"] = guess_flags
    res["rounds"] = fitter.rounds if fit_obj.size else 0
    return res


def bootstrap_many(obs, model="exp", set_psf_results=False, **kw):
    """
    bootstrap_batch for a catalogue of reference-style objects: `obs` is a
    sequence of Observation / ObsList / MultiBandObsList whose observations
    carry their psf OBSERVATIONS (obs.psf with an image) -- what the reference's
    Bootstrapper.go takes one at a time (bootstrap.py:24-103).  The object and
    psf stamps are packed into two batches, **kw goes to bootstrap_batch (psf
    fitter, guesser, attempts, prior, fit_pars ...).

    set_psf_results: store each psf fit's flags in obs.psf.meta['result'] and,
    where it passed, its mixture in obs.psf.gmix, as PSFRunner does
    (runners.py:205-215).

    Returns a fitting.ManyResults: element i is the result dict of object i
    (flags = BOOT_PSF_FAILURE and NaN parameters where the reference raises
    BootPSFFailure); .arrays holds bootstrap_batch's arrays.
    """
    from .batch import flatten_observations, StampBatch
    from .fitting import ManyResults
    from .observation import get_mb_obs
    stamps, sobj, sband, nband, _ = flatten_observations(obs)
    flat = [e for o in obs for ol in get_mb_obs(o) for e in ol]
    for e in flat:
        if not e.has_psf():
            raise ValueError("bootstrap_many: every observation needs its psf observation")
    psf_stamps = StampBatch.from_observations([e.psf for e in flat])
    res = bootstrap_batch(stamps, psf_stamps, model=model, stamp_obj=sobj, stamp_band=sband, **kw)
    if set_psf_results:
        recs = res["psf_gmix"].to_numpy()
        from .gmix import GMix
        for i, e in enumerate(flat):
            flags = int(res["psf_flags"][i])
            e.psf.meta["result"] = {"flags": flags, "nfev": int(res["psf_nfev"][i]),
                                    "ntry": int(res["psf_ntry"][i])}
            if flags == 0:
                gm = GMix(ngauss=recs[i].size)
                gm._data[:] = recs[i]
                e.psf.set_gmix(gm)
    return ManyResults(res, model, nband)


def _scatter(res, fit_obj, nobj, npars):
    """the result arrays of the fitted objects laid out over all nobj objects:
    the others hold 0 (integers) / NaN (floats)"""
    out = {}
    nfit = fit_obj.size
    for k in (res.keys() if hasattr(res, "keys") else ()):
        v = res[k]
        if isinstance(v, np.ndarray) and v.shape[:1] == (nfit,) and nfit != nobj:
            full = np.zeros((nobj,) + v.shape[1:], dtype=v.dtype)
            if v.dtype.kind == "f":
                full[...] = np.nan
            full[fit_obj] = v
            out[k] = full
        else:
            out[k] = v
    if "flags" not in out:
        # nothing was fitted at all
        out["flags"] = np.zeros(nobj, dtype=np.int64)
        out["nfev"] = np.zeros(nobj, dtype=np.int64)
        out["ier"] = np.zeros(nobj, dtype=np.int64)
        out["pars"] = np.full((nobj, npars), np.nan)
        out["pars_err"] = np.full((nobj, npars), np.nan)
        out["pars_cov0"] = np.full((nobj, npars, npars), np.nan)
        out["pars_cov"] = np.full((nobj, npars, npars), np.nan)
    return out
