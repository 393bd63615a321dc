"""
A device-resident psf -> guess -> object-fit pipeline over N objects: the
batched counterpart of the reference's bootstrap (ngmix/bootstrap.py:67-154:
fit obs.psf, store its mixture, then run the object fitter from a guess).

    1. adaptive moments of every psf stamp        (ngmix_admom_batch)
       -> one gaussian psf mixture per stamp; with psf_ngauss > 1 an EM fit
       of psf_ngauss gaussians from that size     (ngmix_em_batch)
    2. adaptive moments of every object stamp     (ngmix_admom_batch)
       -> centre / shape / size guess, flux guess from the pixel sum
    3. lock-step Levenberg-Marquardt              (LMBatchFitter)

Nothing but the guesses and the result records crosses PCIe.  Objects whose
psf or guess measurement fails keep a neutral guess (centre 0, round,
T = T_psf); objects whose psf fit fails are flagged BOOT_PSF_FAILURE the way
remove_failed_psf_obs / BootPSFFailure drop them in the reference.
"""
import numpy as np

from .batch import GMixBatch
from .lm_batch import LMBatchFitter

__all__ = ["bootstrap_batch", "BOOT_PSF_FAILURE"]

BOOT_PSF_FAILURE = 1 << 30


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

EM_MAXITER = 2 ** 1   # em.py's flag values
EM_RANGE_ERROR = 2 ** 0


def _em_psf(psf_stamps, ngauss, T0, cen, rng, em_pars):
    """EM fit of ngauss free gaussians to every psf stamp (runners.py
    PSFRunner + EMFitter on prep_obs images); returns the flux-normalised
    mixtures (GMixBatch) and per-stamp flags"""
    import torch
    n = psf_stamps.n
    frac, fac = _EM_PSF_GUESS[ngauss]
    full = np.zeros((n, ngauss, 6))
    for i in range(ngauss):
        sig2 = 0.5 * T0 * fac[i]
        full[:, i, 0] = frac[i] * rng.uniform(0.9, 1.1, size=n)
        full[:, i, 1:3] = cen + rng.uniform(-0.02, 0.02, size=(n, 2)) * np.sqrt(T0)[:, None]
        full[:, i, 3] = sig2 * (1.0 + rng.uniform(-0.1, 0.1, size=n))
        full[:, i, 4] = rng.uniform(-0.05, 0.05, size=n) * sig2
        full[:, i, 5] = sig2 * (1.0 + rng.uniform(-0.1, 0.1, size=n))
    gm, _ = GMixBatch.from_pars(full.reshape(n, -1), "full", device=psf_stamps.device,
                                ngauss=ngauss)
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
    # set_flux(1.0) of the stored psf mixture (em.py:129): p /= sum p per stamp
    data = gm.data.reshape(n, ngauss, 13)  # a view: edited in place
    psum = data[:, :, 0].sum(dim=1, keepdim=True)
    psum = torch.where(psum > 0, psum, torch.ones_like(psum))
    data[:, :, 0] /= psum
    gm.set_norms()
    return gm, flags


def _coellip_psf(psf_stamps, ngauss, T0, cen, g, rng, ntry=1):
    """lock-step LM fit of ngauss co-elliptical gaussians to every psf stamp
    (PSFRunner + CoellipFitter, runners.py:152-223, fitters.py:120-141) from
    the adaptive-moments centre, shape and size; the fits that end with flags
    != 0 are repeated from a freshly perturbed guess, up to ntry attempts in
    all (PSFRunner's retry loop, runners.py:176-199).  Returns the
    flux-normalised mixtures and the fit flags"""
    gm, flags = _coellip_psf_once(psf_stamps, ngauss, T0, cen, g, rng)
    for _ in range(1, int(ntry)):
        again = np.nonzero(flags != 0)[0]
        if again.size == 0:
            break
        import torch
        sub, sflags = _coellip_psf_once(psf_stamps.select(again), ngauss, T0[again],
                                        cen[again], g[again], rng)
        d_idx = torch.from_numpy(again).to(gm.data.device)
        gm.data.reshape(psf_stamps.n, ngauss, 13)[d_idx] = sub.data.reshape(
            again.size, ngauss, 13)
        flags[again] = sflags
    return gm, flags


def _coellip_psf_once(psf_stamps, ngauss, T0, cen, g, rng):
    n = psf_stamps.n
    frac, fac = _EM_PSF_GUESS[ngauss]
    npix = psf_stamps.npix.astype(np.int64)
    if np.all(npix == npix[0]):
        flux = psf_stamps.val.reshape(n, -1).sum(dim=1).cpu().numpy()
    else:
        import torch
        flux = torch.segment_reduce(
            psf_stamps.val, "sum",
            lengths=torch.from_numpy(npix).to(psf_stamps.device)).cpu().numpy()
    area = (psf_stamps.jac[:, 7] ** 2).cpu().numpy()
    flux = np.where(flux > 0, flux * area, 1.0)
    guess = np.zeros((n, 4 + 2 * ngauss))
    guess[:, 0:2] = cen
    guess[:, 2:4] = g
    for i in range(ngauss):
        guess[:, 4 + i] = T0 * fac[i] * (1.0 + rng.uniform(-0.05, 0.05, size=n))
        guess[:, 4 + ngauss + i] = flux * frac[i]
    fitter = LMBatchFitter("coellip", ngauss=ngauss)
    res = fitter.go(psf_stamps, guess)
    gm = fitter.gmix
    data = gm.data.reshape(n, ngauss, 13)  # a view: edited in place
    import torch
    psum = data[:, :, 0].sum(dim=1, keepdim=True)
    psum = torch.where(psum > 0, psum, torch.ones_like(psum))
    data[:, :, 0] /= psum
    gm.set_norms()
    return gm, res["flags"].copy()


def bootstrap_batch(stamps, psf_stamps, model="exp", psf_Tguess=0.3, Tguess=None,
                    fit_pars=None, rng=None, psf_ngauss=1, em_pars=None, prior=None,
                    stamp_obj=None, stamp_band=None, ntry=1, psf_fitter="em",
                    psf_ntry=1, guess_admom=None):
    """
    stamps, psf_stamps: StampBatch of the object images and of their psf images
        (stamp i of one belongs to stamp i of the other)
    model: 'gauss' | 'exp' | 'dev'
    psf_Tguess / Tguess: starting sizes for the adaptive moments (arcsec^2);
        Tguess defaults to 2 * psf_Tguess
    psf_ngauss: 1: the psf is its adaptive-moments gaussian; 2 or 3: an EM fit
        of that many free gaussians started from the adaptive-moments size
        (em_pars: miniter / maxiter / tol of that fit), or with
        psf_fitter='coellip' a lock-step LM fit of that many co-elliptical
        gaussians (the reference's CoellipFitter psf runners)
    prior: a batch prior for the object fits (prior_batch.PriorSimpleSepBatch ...)
    psf_ntry: attempts per psf fit (psf_fitter='coellip'; PSFRunner's ntry)
    guess_admom: dict of etol / Ttol / maxiter for the adaptive moments of the
        guess stage (defaults 1e-4 / 1e-2: a starting point, not a measurement)
    ntry: fits that end with flags != 0 are repeated from a perturbed guess up
        to ntry times in all, as Runner does object by object
        (runners.py:95-150); 'ntry' of the result counts the attempts
    stamp_obj / stamp_band: as for LMBatchFitter.go -- objects with several
        epochs and bands (a MultiBandObsList each).  Every stamp gets its own
        psf fit; the guess is the mean of the object's stamps' adaptive
        moments and, per band, of their pixel sums.  An object one of whose
        psf fits failed is flagged BOOT_PSF_FAILURE (the reference drops such
        epochs and fails only when none is left, bootstrap.py:118-154).

    Returns a dict: the LMBatchFitter result arrays, plus 'psf_T', 'psf_flags',
    'psf_g', 'guess' (the LM starting points) and 'guess_flags'.
    """
    assert stamps.n == psf_stamps.n
    if rng is None:
        rng = np.random.RandomState(0)
    n = stamps.n
    if Tguess is None:
        Tguess = 2.0 * psf_Tguess

    # the guess stage's tolerances: it only has to put the fit inside its basin
    # -- ten times the measurement defaults stop the iteration two or three
    # passes earlier and the LM needs no more rounds for it
    gconf = dict(etol=1.0e-4, Ttol=1.0e-2, no_cov=True)   # (only the weight is read)
    gconf.update(guess_admom or {})

    # 1. psf: one gaussian per stamp from its adaptive moments.  With a
    # one-gaussian psf nothing between here and the object's adaptive moments
    # draws random numbers or needs the psf result, so that launch (and the pixel
    # sums of the flux guess) are queued right behind: the device works through
    # them while the host turns the psf moments into mixtures
    p_launched = _admom_launch(psf_stamps, psf_Tguess, rng)
    o_launched = _admom_launch(stamps, Tguess, rng, **gconf) if psf_ngauss == 1 else None
    same_size = bool(np.all(stamps.npix == stamps.npix[0]))
    d_flux = stamps.val.reshape(n, -1).sum(dim=1) if same_size else None
    pw, prec, pst = _admom_collect(p_launched)
    psf_bad = (prec["flags"] != 0) | (pst != 0)
    psf_T = np.where(psf_bad, psf_Tguess, pw["irr"] + pw["icc"])
    pe1 = np.where(psf_bad, 0.0, (pw["icc"] - pw["irr"]) / psf_T)
    pe2 = np.where(psf_bad, 0.0, 2.0 * pw["irc"] / psf_T)
    pg1, pg2 = _e1e2_to_g1g2(pe1, pe2)
    psf_pars = np.zeros((n, 6))
    psf_pars[:, 2], psf_pars[:, 3], psf_pars[:, 4], psf_pars[:, 5] = pg1, pg2, psf_T, 1.0
    psf_gm, _ = GMixBatch.from_pars(psf_pars, "gauss", device=stamps.device)
    em_flags = np.zeros(n, dtype=np.int64)
    if psf_ngauss > 1:
        cen = np.where(psf_bad[:, None], 0.0, np.stack([pw["row"], pw["col"]], axis=1))
        if psf_fitter == "coellip":
            psf_gm, em_flags = _coellip_psf(psf_stamps, psf_ngauss, psf_T, cen,
                                            np.stack([pg1, pg2], axis=1), rng,
                                            ntry=psf_ntry)
        else:
            psf_gm, em_flags = _em_psf(psf_stamps, psf_ngauss, psf_T, cen, rng, em_pars)
        psf_bad = psf_bad | (em_flags != 0)

    # 2. guess: adaptive moments of the object, psf size taken out
    if o_launched is None:
        o_launched = _admom_launch(stamps, Tguess, rng, **gconf)
    ow, orec, ost = _admom_collect(o_launched)
    gbad = (orec["flags"] != 0) | (ost != 0)
    T_obs = np.where(gbad, 2.0 * psf_T, ow["irr"] + ow["icc"])
    e1 = np.where(gbad, 0.0, (ow["icc"] - ow["irr"]) / T_obs)
    e2 = np.where(gbad, 0.0, 2.0 * ow["irc"] / T_obs)
    g1, g2 = _e1e2_to_g1g2(e1, e2)
    npix = stamps.npix.astype(np.int64)
    if d_flux is not None:
        flux = d_flux.cpu().numpy()
    else:
        # ragged stamps: a segmented sum on the device
        import torch
        vals = stamps.val
        if not stamps._packed():
            # (a selection shares its parent's pixel arrays: gather its own pixels)
            start = np.concatenate([[0], np.cumsum(npix)[:-1]])
            flat = np.repeat(stamps.pix_off - start, npix) + np.arange(int(npix.sum()))
            vals = vals[torch.from_numpy(flat).to(stamps.device)]
        flux = torch.segment_reduce(
            vals, "sum", lengths=torch.from_numpy(npix).to(stamps.device)).cpu().numpy()
    if stamp_obj is None:
        sobj = np.arange(n, dtype=np.int64)
        sband = np.zeros(n, dtype=np.int64)
    else:
        sobj = np.ascontiguousarray(stamp_obj, dtype=np.int64)
        sband = (np.zeros(n, dtype=np.int64) if stamp_band is None
                 else np.ascontiguousarray(stamp_band, dtype=np.int64))
    nobj = int(sobj.max()) + 1
    nband = int(sband.max()) + 1
    good = (~gbad).astype("f8")
    ngood = np.bincount(sobj, weights=good, minlength=nobj)

    def obj_mean(x, default):
        tot = np.bincount(sobj, weights=np.where(gbad, 0.0, x), minlength=nobj)
        return np.where(ngood > 0, tot / np.maximum(ngood, 1.0), default)
    Tdiff = np.maximum(T_obs - psf_T, 0.1 * psf_T)
    guess = np.zeros((nobj, 5 + nband))
    guess[:, 0] = obj_mean(ow["row"], 0.0)
    guess[:, 1] = obj_mean(ow["col"], 0.0)
    guess[:, 2], guess[:, 3] = obj_mean(g1, 0.0), obj_mean(g2, 0.0)
    # (stamps whose moments failed carry the neutral size 2 T_psf - T_psf)
    guess[:, 4] = np.bincount(sobj, weights=Tdiff, minlength=nobj) / \
        np.maximum(np.bincount(sobj, minlength=nobj), 1)
    key = sobj * nband + sband
    fsum = np.bincount(key, weights=flux, minlength=nobj * nband)
    fcnt = np.bincount(key, minlength=nobj * nband)
    fmean = (fsum / np.maximum(fcnt, 1)).reshape(nobj, nband)
    guess[:, 5:] = np.where(fmean > 0, fmean, 1.0)

    # 3. the fits
    fitter = LMBatchFitter(model, fit_pars=fit_pars, prior=prior)
    res = fitter.go(stamps, guess, psf=psf_gm,
                    stamp_obj=None if stamp_obj is None else sobj.astype(np.int32),
                    stamp_band=None if stamp_obj is None else sband.astype(np.int32))
    tries = np.ones(nobj, dtype=np.int64)
    for _ in range(1, int(ntry)):
        redo = np.nonzero(res["flags"] != 0)[0]
        if redo.size == 0:
            break
        # the failed objects' stamps as a batch of their own
        member = np.isin(sobj, redo)
        sidx = np.nonzero(member)[0]
        sub_obj = np.searchsorted(redo, sobj[sidx]).astype(np.int32)
        g2_ = guess[redo] * (1.0 + 0.1 * rng.uniform(-1, 1, size=guess[redo].shape))
        g2_[:, 0:2] = guess[redo, 0:2] + 0.05 * rng.uniform(-1, 1, size=(redo.size, 2)) * \
            np.sqrt(guess[redo, 4:5])
        sub_psf = GMixBatch(psf_gm.data.reshape(n, psf_gm.ngauss, 13)[
            _to_device(sidx, stamps.device)].reshape(-1, 13).contiguous(), sidx.size,
            psf_gm.ngauss)
        sub = fitter.go(stamps.select(sidx), g2_, psf=sub_psf, stamp_obj=sub_obj,
                        stamp_band=sband[sidx].astype(np.int32))
        tries[redo] += 1
        # (items() reads through the keys an LMBatchResult keeps on the
        # device, pars_cov0 among them: the retried objects take every array
        # of their new fit)
        for k, v in sub.items():
            if isinstance(v, np.ndarray) and v.shape[:1] == (redo.size,) and k in res:
                if not res[k].flags.writeable:
                    res[k] = np.array(res[k])
                res[k][redo] = v
        guess[redo] = g2_
    res["ntry"] = tries
    obj_psf_bad = np.bincount(sobj, weights=psf_bad.astype("f8"), minlength=nobj) > 0
    res["flags"] = res["flags"] | np.where(obj_psf_bad, BOOT_PSF_FAILURE, 0)
    res["psf_T"] = psf_T
    res["psf_g"] = np.stack([pg1, pg2], axis=1)
    res["psf_flags"] = np.where(pst != 0, -1, prec["flags"])
    res["psf_em_flags"] = em_flags
    res["psf_gmix"] = psf_gm
    res["guess"] = guess
    res["guess_flags"] = np.where(ost != 0, -1, orec["flags"])
    res["rounds"] = fitter.rounds
    return res
