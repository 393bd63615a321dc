"""
Adaptive moments (reference API: ngmix/admom/admom.py).  The iteration runs in
the admom HIP kernel (one work-group per stamp, whole iteration on the
device); this module builds the guess, maps the kernel's status / flags onto
the reference's exceptions and flag bits, and derives the summary statistics
from the 584-byte result record.
"""
import numpy as np
from numpy import diag

from . import _lib
from . import flags as ngflags
from .gexceptions import GMixRangeError
from .gmix import GMix, GMixModel, GMIX_LOW_DETVAL
from .moments import fwhm_to_T
from .observation import Observation
from .shape import e1e2_to_g1g2
from .util import get_ratio_error

__all__ = ["run_admom", "run_admom_many", "find_cen_admom", "AdmomFitter", "AdmomResult",
           "AdmomManyResults"]

DEFAULT_MAXITER = 200
DEFAULT_SHIFTMAX = 5.0  # pixels
DEFAULT_ETOL = 1.0e-5
DEFAULT_TTOL = 1.0e-3

_admom_result_dtype = _lib.ADMOM_RESULT_DTYPE
_admom_conf_dtype = _lib.ADMOM_CONF_DTYPE


def run_admom(obs, guess, maxiter=DEFAULT_MAXITER, shiftmax=DEFAULT_SHIFTMAX,
              etol=DEFAULT_ETOL, Ttol=DEFAULT_TTOL, cenonly=False, rng=None):
    """adaptive moments of one Observation; guess is a GMix or a T value"""
    am = AdmomFitter(maxiter=maxiter, shiftmax=shiftmax, etol=etol, Ttol=Ttol,
                     cenonly=cenonly, rng=rng)
    return am.go(obs=obs, guess=guess)


def run_admom_many(obs, guess, maxiter=DEFAULT_MAXITER, shiftmax=DEFAULT_SHIFTMAX,
                   etol=DEFAULT_ETOL, Ttol=DEFAULT_TTOL, cenonly=False, rng=None):
    """run_admom over a sequence of Observations as ONE batch (the loop over a
    catalogue the reference's callers write around run_admom, admom.py:20-71);
    guess: one T value for all, a sequence of T values, or a sequence of GMix"""
    am = AdmomFitter(maxiter=maxiter, shiftmax=shiftmax, etol=etol, Ttol=Ttol,
                     cenonly=cenonly, rng=rng)
    return am.go_many(obs=obs, guess=guess)


def find_cen_admom(obs, fwhm=None, gmix=None, maxiter=DEFAULT_MAXITER,
                   shiftmax=DEFAULT_SHIFTMAX, etol=DEFAULT_ETOL,
                   Ttol=DEFAULT_TTOL, ntry=1, rng=None):
    """centroid with a fixed-size weight (cenonly adaptive moments); the result
    gets a 'cen' entry (offset from the jacobian centre, or NaNs on failure)"""
    if ntry > 1 and rng is None:
        raise ValueError(
            "send a random number generator rng= when trying more than once "
            "this facilitates generating a new guess for the center")
    if gmix is not None:
        wt = gmix.copy()
    elif fwhm is not None:
        wt = GMixModel([0.0, 0.0, 0.0, 0.0, fwhm_to_T(fwhm), 1.0], "gauss")
    else:
        raise ValueError("send gmix= or fwhm=")
    scale = obs.jacobian.scale
    am = AdmomFitter(maxiter=maxiter, shiftmax=shiftmax, etol=etol, Ttol=Ttol,
                     cenonly=True)
    for itry in range(ntry):
        res = am.go(obs=obs, guess=wt)
        if res["flags"] == 0:
            break
        if ntry > 1:
            drow, dcol = rng.uniform(low=-scale / 2, high=scale / 2, size=2)
            wt.set_cen(row=drow, col=dcol)
    if res["flags"] == 0:
        res["cen"] = res.get_gmix().get_cen()
    else:
        res["cen"] = np.zeros(2) + np.nan
    return res


class AdmomResult(dict):
    """dict of adaptive-moments results with get_gmix() / make_image()"""

    def __init__(self, obs, result):
        self._obs = obs
        self.update(result)

    def get_gmix(self):
        """the fitted gaussian, unit flux: the moment ratios M1/T, M2/T are an
        (e1, e2) ellipticity, which the mixture wants as a reduced shear"""
        if self["flags"]:
            raise RuntimeError("cannot create gmix, fit failed")
        row, col, m1, m2, T = self["pars"][:5]
        g1, g2 = e1e2_to_g1g2(m1 / T, m2 / T)
        return GMixModel(np.array([row, col, g1, g2, T, 1.0]), "gauss")

    def make_image(self):
        if self["flags"] != 0:
            raise RuntimeError("cannot create image, fit failed")
        obs = self._obs
        gm = self.get_gmix()
        gm.set_flux(obs.image.sum())
        return gm.make_image(obs.image.shape, jacobian=obs.jacobian)


class AdmomManyResults(object):
    """the AdmomResults of AdmomFitter.go_many, made on access from the batch's
    584-byte records (.records)"""

    def __init__(self, obs, records, wgt_norms):
        self._obs = obs
        self.records = records
        self._norms = wgt_norms

    def __len__(self):
        return len(self._obs)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        o = self._obs[i]
        return AdmomResult(obs=o, result=get_result(self.records[i:i + 1].copy(),
                                                    o.jacobian.area, self._norms[i]))

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class AdmomFitter(object):
    """adaptive moments fitter; .go(obs, guess) returns an AdmomResult"""

    kind = "am"

    def __init__(self, maxiter=DEFAULT_MAXITER, shiftmax=DEFAULT_SHIFTMAX,
                 etol=DEFAULT_ETOL, Ttol=DEFAULT_TTOL, cenonly=False, rng=None):
        conf = np.zeros(1, dtype=_admom_conf_dtype)
        conf["maxiter"] = maxiter
        conf["shiftmax"] = shiftmax
        conf["etol"] = etol
        conf["Ttol"] = Ttol
        conf["cenonly"] = cenonly
        self.conf = conf
        self.rng = rng

    def go(self, obs, guess):
        if not isinstance(obs, Observation):
            raise ValueError("input obs must be an Observation")
        guess_gmix = self._get_guess(obs=obs, guess=guess)
        ares = self._get_am_result()
        wt_gmix = guess_gmix._data  # the guess IS the weight, mutated in place
        status = obs._device_stamp().admom_single(wt_gmix, self.conf, ares)
        if status in (_lib.ERR_DET_TOO_LOW, _lib.ERR_T_TOO_LOW):
            # GMixRangeError inside admom(): admom.py:355-356
            ares["flags"] = ngflags.GMIX_RANGE_ERROR
        elif status != 0:
            _lib.check(status, "admom")
        result = get_result(ares, obs.jacobian.area, wt_gmix["norm"][0])
        return AdmomResult(obs=obs, result=result)

    def go_many(self, obs, guess):
        """the adaptive moments of MANY Observations by one launch of the batch
        kernel (ngmix_admom_batch); returns an AdmomManyResults: element i is
        the AdmomResult go(obs[i], guess[i]) returns.  Guesses given as T values
        are drawn as go() draws them, object after object, from the same
        stream (admom.py:376-404)"""
        from .batch import StampBatch, GMixBatch
        n = len(obs)
        for o in obs:
            if not isinstance(o, Observation):
                raise ValueError("input obs must be an Observation")
        if np.ndim(guess) == 0 and not isinstance(guess, GMix):
            guess = np.full(n, float(guess))
        if len(guess) != n:
            raise ValueError("one guess per observation")
        if isinstance(guess[0], GMix):
            recs = np.stack([g._data for g in guess]).reshape(n)
        else:
            # _generate_guess for every object: five draws each, in its order
            u = self._get_rng().uniform(size=(n, 5))
            half = 0.5 * np.array([o._jacobian._data["scale"][0] for o in obs])
            pars = np.zeros((n, 6))
            pars[:, 0:2] = -half[:, None] + (2.0 * half[:, None]) * u[:, 0:2]
            pars[:, 2:4] = -0.3 + 0.6 * u[:, 2:4]
            pars[:, 4] = np.asarray(guess, dtype="f8") * (1.0 + (-0.1 + 0.2 * u[:, 4]))
            pars[:, 5] = 1.0
            recs = np.zeros(n, dtype=_lib.GAUSS2D_DTYPE)
            for i in range(n):
                # (the host fill of GMixModel: the same arithmetic as go()'s guess)
                recs[i:i + 1] = GMixModel(pars[i], "gauss")._data
        stamps = StampBatch.from_observations(list(obs))
        wt = GMixBatch.from_numpy(recs.reshape(n, 1), device=stamps.device)
        c = self.conf[0]
        res, status = stamps.admom(wt, maxiter=int(c["maxiter"]), shiftmax=float(c["shiftmax"]),
                                   etol=float(c["etol"]), Ttol=float(c["Ttol"]),
                                   cenonly=bool(c["cenonly"]))
        from .batch import records_to_numpy
        ares = records_to_numpy(res, _admom_result_dtype)
        status = status.cpu().numpy()
        norms = wt.to_numpy()["norm"][:, 0]
        bad = (status == _lib.ERR_DET_TOO_LOW) | (status == _lib.ERR_T_TOO_LOW)
        ares["flags"][bad] = ngflags.GMIX_RANGE_ERROR
        other = (status != 0) & ~bad
        if other.any():
            _lib.check(int(status[other][0]), "admom")
        return AdmomManyResults(list(obs), ares, norms)

    def _get_guess(self, obs, guess):
        if isinstance(guess, GMix):
            return guess
        return self._generate_guess(obs=obs, Tguess=guess)

    def _get_am_result(self):
        return np.zeros(1, dtype=_admom_result_dtype)

    def _get_rng(self):
        if self.rng is None:
            self.rng = np.random.RandomState()
        return self.rng

    def _generate_guess(self, obs, Tguess):
        """a random round-ish unit-flux gaussian around the canonical centre.
        The ORDER of the draws is part of the contract (a seeded reference run
        must see the same stream, admom.py:398-400): the centre pair within
        half a pixel, the shape pair within 0.3, then T within 10 %."""
        draw = self._get_rng().uniform
        half_pixel = 0.5 * obs.jacobian.get_scale()
        cen = draw(-half_pixel, half_pixel, 2)
        shape = draw(-0.3, 0.3, 2)
        T = Tguess * (1.0 + draw(-0.1, 0.1))
        return GMixModel(np.concatenate([cen, shape, [T, 1.0]]), "gauss")


def get_result(ares, jac_area, wgt_norm):
    """
    result record -> dict with flux, T, e and their errors and flag strings
    (reference: admom.py:406-568).  fnorm = jac_area * wgt_norm * wsum turns
    the weighted flux sum into surface-brightness flux units.
    """
    if isinstance(ares, np.ndarray):
        ares = ares[0]
        names = ares.dtype.names
    else:
        names = list(ares.keys())

    res = {}
    for n in names:
        if n == "sums":
            res[n] = ares[n].copy()
        elif n == "sums_cov":
            res[n] = ares[n].reshape((7, 7)).copy()
        else:
            res[n] = ares[n]
    res["sums_norm"] = ares["wsum"]

    nan = np.nan
    res.update({
        "flagstr": "", "flux_flags": 0, "flux_flagstr": "", "T_flags": 0,
        "T_flagstr": "", "rho4_flags": 0, "rho4_flagstr": "",
        "flux": nan, "flux_mean": nan, "flux_err": nan, "T": nan, "T_err": nan,
        "rho4": nan, "rho4_err": nan, "s2n": nan, "e1": nan, "e2": nan,
        "e1err": nan, "e2err": nan,
        "e": np.array([nan, nan]), "e_err": np.array([nan, nan]),
        "e_cov": np.diag([nan, nan]),
    })
    sums, cov = res["sums"], res["sums_cov"]

    if res["flags"] == 0:
        res["T"] = res["pars"][4]
        res["rho4"] = ares["rho4"]
        res["flux_mean"] = sums[5] / res["wsum"]
        res["pars"][5] = res["flux_mean"]

    # flux
    if res["flags"] == 0:
        if res["T"] > GMIX_LOW_DETVAL:
            fnorm = jac_area * wgt_norm * res["wsum"]
            res["flux"] = sums[5] / fnorm
            if cov[5, 5] > 0:
                res["flux_err"] = np.sqrt(cov[5, 5]) / fnorm
                res["s2n"] = res["flux"] / res["flux_err"]
            else:
                res["flux_flags"] |= ngflags.NONPOS_VAR
        else:
            res["flux_flags"] |= ngflags.NONPOS_SIZE
    else:
        res["flux_flags"] |= res["flags"]

    # T and rho4: the sums include the weight, hence the factor 4
    if res["flags"] == 0:
        for name, ind in (("T", 4), ("rho4", 6)):
            if cov[ind, ind] > 0 and cov[5, 5] > 0:
                if sums[5] > 0:
                    if name == "rho4":
                        res["rho4"] = sums[6] / sums[5]
                    res[name + "_err"] = 4 * get_ratio_error(
                        sums[ind], sums[5], cov[ind, ind], cov[5, 5], cov[ind, 5])
                else:
                    res[name + "_flags"] |= ngflags.NONPOS_FLUX
            else:
                res[name + "_flags"] |= ngflags.NONPOS_VAR
    else:
        res["T_flags"] |= res["flags"]
        res["rho4_flags"] |= res["flags"]

    # overall flags and shapes
    if not np.all(np.diagonal(cov[2:, 2:]) > 0):
        res["flags"] |= ngflags.NONPOS_VAR

    if res["flags"] == 0:
        if sums[5] > 0:
            if res["T"] > 0.0:
                res["e"][:] = res["pars"][2:4] / res["T"]
                res["e1"], res["e2"] = res["e"]
                res["e1err"] = 2 * get_ratio_error(
                    sums[2], sums[4], cov[2, 2], cov[4, 4], cov[2, 4])
                res["e2err"] = 2 * get_ratio_error(
                    sums[3], sums[4], cov[3, 3], cov[4, 4], cov[3, 4])
                if not np.isfinite(res["e1err"]) or not np.isfinite(res["e2err"]):
                    res["e1err"] = nan
                    res["e2err"] = nan
                    res["e_err"] = np.array([nan, nan])
                    res["e_cov"] = diag([nan, nan])
                    res["flags"] |= ngflags.NONPOS_SHAPE_VAR
                else:
                    res["e_cov"] = diag([res["e1err"] ** 2, res["e2err"] ** 2])
                    res["e_err"] = np.array([res["e1err"], res["e2err"]])
            else:
                res["flags"] |= ngflags.NONPOS_SIZE
        else:
            res["flags"] |= ngflags.NONPOS_FLUX

    for key in ("flags", "flux_flags", "T_flags", "rho4_flags"):
        strkey = "flagstr" if key == "flags" else key[:-1] + "str"
        res[strkey] = ngflags.get_flags_str(res[key])
    return res
