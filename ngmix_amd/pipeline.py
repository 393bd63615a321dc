"""
A device-resident psf -> guess -> object-fit pipeline over N objects: the
batched counterpart of the reference's bootstrap (ngmix/bootstrap.py:67-154:
fit obs.psf, store its mixture, then run the object fitter from a guess).

    1. adaptive moments of every psf stamp        (ngmix_admom_batch)
       -> one gaussian psf mixture per stamp
    2. adaptive moments of every object stamp     (ngmix_admom_batch)
       -> centre / shape / size guess, flux guess from the pixel sum
    3. lock-step Levenberg-Marquardt              (LMBatchFitter)

Nothing but the guesses and the result records crosses PCIe.  Objects whose
psf or guess measurement fails keep a neutral guess (centre 0, round,
T = T_psf); objects whose psf fit fails are flagged BOOT_PSF_FAILURE the way
remove_failed_psf_obs / BootPSFFailure drop them in the reference.
"""
import numpy as np

from . import _lib
from .batch import GMixBatch, records_to_numpy
from .lm_batch import LMBatchFitter

__all__ = ["bootstrap_batch", "BOOT_PSF_FAILURE"]

BOOT_PSF_FAILURE = 1 << 30


def _e1e2_to_g1g2(e1, e2):
    e = np.sqrt(e1 ** 2 + e2 ** 2)
    e = np.minimum(e, 0.999999)
    with np.errstate(invalid="ignore", divide="ignore"):
        fac = np.where(e > 0, np.tanh(0.5 * np.arctanh(e)) / np.where(e > 0, e, 1.0), 0.0)
    return e1 * fac, e2 * fac


def _admom_gaussians(stamps, Tguess, rng):
    """admom on every stamp from a round guess of size Tguess; returns the
    converged weight gaussians (records) and the result records"""
    n = stamps.n
    guess = np.zeros((n, 6))
    guess[:, 0:2] = rng.uniform(-0.1, 0.1, size=(n, 2)) * np.sqrt(Tguess / 2.0)
    guess[:, 4] = Tguess
    guess[:, 5] = 1.0
    wt, _ = GMixBatch.from_pars(guess, "gauss", device=stamps.device)
    res, status = stamps.admom(wt)
    rec = records_to_numpy(res, _lib.ADMOM_RESULT_DTYPE)
    return wt.to_numpy()[:, 0], rec, status.cpu().numpy()


def bootstrap_batch(stamps, psf_stamps, model="exp", psf_Tguess=0.3, Tguess=None,
                    fit_pars=None, rng=None):
    """
    stamps, psf_stamps: StampBatch of the object images and of their psf images
        (stamp i of one belongs to stamp i of the other)
    model: 'gauss' | 'exp' | 'dev'
    psf_Tguess / Tguess: starting sizes for the adaptive moments (arcsec^2);
        Tguess defaults to 2 * psf_Tguess

    Returns a dict: the LMBatchFitter result arrays, plus 'psf_T', 'psf_flags',
    'psf_g', 'guess' (the LM starting points) and 'guess_flags'.
    """
    assert stamps.n == psf_stamps.n
    if rng is None:
        rng = np.random.RandomState(0)
    n = stamps.n
    if Tguess is None:
        Tguess = 2.0 * psf_Tguess

    # 1. psf: one gaussian per stamp from its adaptive moments
    pw, prec, pst = _admom_gaussians(psf_stamps, psf_Tguess, rng)
    psf_bad = (prec["flags"] != 0) | (pst != 0)
    psf_T = np.where(psf_bad, psf_Tguess, pw["irr"] + pw["icc"])
    pe1 = np.where(psf_bad, 0.0, (pw["icc"] - pw["irr"]) / psf_T)
    pe2 = np.where(psf_bad, 0.0, 2.0 * pw["irc"] / psf_T)
    pg1, pg2 = _e1e2_to_g1g2(pe1, pe2)
    psf_pars = np.zeros((n, 6))
    psf_pars[:, 2], psf_pars[:, 3], psf_pars[:, 4], psf_pars[:, 5] = pg1, pg2, psf_T, 1.0
    psf_gm, _ = GMixBatch.from_pars(psf_pars, "gauss", device=stamps.device)

    # 2. guess: adaptive moments of the object, psf size taken out
    ow, orec, ost = _admom_gaussians(stamps, Tguess, rng)
    gbad = (orec["flags"] != 0) | (ost != 0)
    T_obs = np.where(gbad, 2.0 * psf_T, ow["irr"] + ow["icc"])
    e1 = np.where(gbad, 0.0, (ow["icc"] - ow["irr"]) / T_obs)
    e2 = np.where(gbad, 0.0, 2.0 * ow["irc"] / T_obs)
    g1, g2 = _e1e2_to_g1g2(e1, e2)
    npix = stamps.npix.astype(np.int64)
    if np.all(npix == npix[0]):
        flux = stamps.val.reshape(n, -1).sum(dim=1).cpu().numpy()
    else:
        cs = np.concatenate([[0], np.cumsum(npix)])
        v = stamps.val.cpu().numpy()
        flux = np.array([v[cs[i]:cs[i + 1]].sum() for i in range(n)])
    guess = np.zeros((n, 6))
    guess[:, 0] = np.where(gbad, 0.0, ow["row"])
    guess[:, 1] = np.where(gbad, 0.0, ow["col"])
    guess[:, 2], guess[:, 3] = g1, g2
    guess[:, 4] = np.maximum(T_obs - psf_T, 0.1 * psf_T)
    guess[:, 5] = np.where(flux > 0, flux, 1.0)

    # 3. the fits
    fitter = LMBatchFitter(model, fit_pars=fit_pars)
    res = fitter.go(stamps, guess, psf=psf_gm)
    res["flags"] = res["flags"] | np.where(psf_bad, BOOT_PSF_FAILURE, 0)
    res["psf_T"] = psf_T
    res["psf_g"] = np.stack([pg1, pg2], axis=1)
    res["psf_flags"] = np.where(pst != 0, -1, prec["flags"])
    res["guess"] = guess
    res["guess_flags"] = np.where(ost != 0, -1, orec["flags"])
    res["rounds"] = fitter.rounds
    return res
