"""
Batched Levenberg-Marquardt fitting: N independent objects advance in lock
step, two kernel launches per LM step for all of them.

Reference (per object): Fitter.go -> run_leastsq -> scipy leastsq / MINPACK
lmder, calling back into FitModel.calc_fdiff / calc_jacobian once per
evaluation (ngmix/fitting/fitters.py:64-112, leastsqbound.py:33-155,
results.py:439-570).  Here the same iteration (csrc/lm_core.hpp: lmder's
decision logic, tested against MINPACK in tests/test_lm_core.py) runs on the
device for every object at once:

    ngmix_lm_eval_batch     residuals + analytic jacobian at each object's
                            trial point, reduced on chip to J^T J, J^T f, |f|^2
    ngmix_lm_advance_batch  one lmder step per object

and the results are packaged exactly as run_leastsq / FitModel.set_fit_result
package one fit: flags, nfev, ier, pars, pars_err, pars_cov0, pars_cov, and for
flags == 0 lnprob, s2n_numer, s2n_denom, npix, chi2per, dof, s2n_w, s2n, g,
g_cov, g_err, T, T_err, flux, flux_err (flux_cov when nband > 1).

Scope: the simple models with an analytic jacobian (gauss, exp, dev:
results.py SIMPLE_ANALYTIC_MODELS), no prior (prior=None is legal in the
reference: zero prior rows, no bounds, results.py:354-357).
"""
import ctypes

import numpy as np

from . import _lib
from .batch import GMixBatch, _dptr, _stream, _torch
from .defaults import PDEF, CDEF, DEFAULT_LM_PARS
from .flags import (
    ZERO_DOF, LM_SINGULAR_MATRIX, LM_NEG_COV_EIG, LM_NEG_COV_DIAG, EIG_NOTFINITE,
    LM_FUNC_NOTFINITE,
)
from .gmix import get_model_num

__all__ = ["LMBatchFitter"]


def _upper_inverse(R):
    """inverse of a batch (N, n, n) of upper triangular matrices"""
    n = R.shape[1]
    X = np.zeros_like(R)
    for j in range(n):
        X[:, j, j] = 1.0 / R[:, j, j]
        for i in range(j - 1, -1, -1):
            acc = np.einsum("nk,nk->n", R[:, i, i + 1:j + 1], X[:, i + 1:j + 1, j])
            X[:, i, j] = -acc / R[:, i, i]
    return X


def _has_negative_pivot(S):
    """for a batch of symmetric matrices: does LDL^T (no pivoting) meet a
    negative pivot, i.e. does the matrix have a negative eigenvalue"""
    A = S.copy()
    n = A.shape[1]
    neg = np.zeros(A.shape[0], dtype=bool)
    with np.errstate(all="ignore"):
        for k in range(n):
            d = A[:, k, k]
            neg |= d < 0
            safe = np.where(d != 0, d, 1.0)
            col = A[:, k + 1:, k] / safe[:, None]
            A[:, k + 1:, k + 1:] -= col[:, :, None] * A[:, None, k, k + 1:]
    return neg

SIMPLE_ANALYTIC_MODELS = ("gauss", "exp", "dev")


class LMBatchFitter(object):
    """
    fitter = LMBatchFitter(model='exp')
    res = fitter.go(stamps, guess, psf=psf_gmixes)

    model: 'gauss' | 'exp' | 'dev'
    fit_pars: dict with maxfev / ftol / xtol as for Fitter (defaults
        DEFAULT_LM_PARS, ngmix/defaults.py:17)
    """

    def __init__(self, model, fit_pars=None):
        if model not in SIMPLE_ANALYTIC_MODELS:
            raise ValueError("LMBatchFitter supports %s" % (SIMPLE_ANALYTIC_MODELS,))
        self.model = model
        self.fit_pars = dict(DEFAULT_LM_PARS if fit_pars is None else fit_pars)

    def go(self, stamps, guess, psf=None, stamp_obj=None, stamp_band=None,
           check_every=1):
        """
        stamps: StampBatch -- every observation (epoch / band) of every object
        guess: (nobj, 5 + nband) starting parameters
            [cen1, cen2, g1, g2, T, flux_band0, ...]
        psf: GMixBatch with one mixture per stamp, or None
        stamp_obj: (nstamps,) object index of each stamp, non-decreasing;
            None: stamp i is object i
        stamp_band: (nstamps,) band of each stamp; None: band 0

        returns a dict of arrays indexed by object
        """
        torch = _torch()
        L = _lib.lib()
        dev = stamps.device
        guess = np.ascontiguousarray(np.atleast_2d(guess), dtype="f8")
        nobj, npars = guess.shape
        nband = npars - 5
        if nband < 1 or npars > _lib.LM_NPMAX:
            raise ValueError("guess must have 5 + nband (1..%d) columns"
                             % (_lib.LM_NPMAX - 5))
        ns = stamps.n
        if stamp_obj is None:
            if ns != nobj:
                raise ValueError("stamp_obj is needed when objects have several stamps")
            sobj = np.arange(ns, dtype=np.int32)
        else:
            sobj = np.ascontiguousarray(stamp_obj, dtype=np.int32)
            if sobj.shape != (ns,) or np.any(np.diff(sobj) < 0):
                raise ValueError("stamp_obj must be (nstamps,) and non-decreasing")
            if sobj.min() < 0 or sobj.max() >= nobj:
                raise ValueError("stamp_obj out of range")
        if stamp_band is None:
            sband = np.zeros(ns, dtype=np.int32)
        else:
            sband = np.ascontiguousarray(stamp_band, dtype=np.int32)
            if sband.shape != (ns,) or sband.min() < 0 or sband.max() >= nband:
                raise ValueError("stamp_band out of range")
        obj_start = np.searchsorted(sobj, np.arange(nobj + 1)).astype(np.int64)
        if np.any(np.diff(obj_start) == 0):
            raise ValueError("every object needs at least one stamp")
        npsf = 0
        if psf is not None:
            assert psf.n == ns, "one psf mixture per stamp"
            npsf = psf.ngauss

        fp = self.fit_pars
        states = np.zeros(nobj, dtype=_lib.LM_STATE_DTYPE)
        _lib.check(L.ngmix_lm_init(
            _lib.ptr(states), nobj, npars, _lib.ptr(guess),
            float(fp.get("ftol", 1.49012e-8)), float(fp.get("xtol", 1.49012e-8)),
            float(fp.get("gtol", 0.0)), int(fp.get("maxfev", 100 * (npars + 1))),
            float(fp.get("factor", 100.0))), "ngmix_lm_init")
        maxfev = int(states["maxfev"][0])

        d_states = torch.from_numpy(states.view(np.uint8).reshape(nobj, -1)).to(dev)
        d_sobj = torch.from_numpy(sobj).to(dev)
        d_sband = torch.from_numpy(sband).to(dev)
        d_start = torch.from_numpy(obj_start).to(dev)
        d_sums = torch.zeros((ns, _lib.LM_NSUM), dtype=torch.float64, device=dev)
        d_status = torch.zeros(ns, dtype=torch.int32, device=dev)
        d_nact = torch.zeros(1, dtype=torch.int32, device=dev)
        b = stamps._batch(1)
        modnum = get_model_num(self.model)
        rounds = 0
        import time
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        with torch.cuda.device(dev):
            while True:
                _lib.check(L.ngmix_lm_eval_batch(
                    ctypes.byref(b), modnum, _dptr(d_states), _dptr(d_sobj),
                    _dptr(d_sband), _dptr(psf.data) if psf is not None else None,
                    npsf, _dptr(d_sums), _dptr(d_status), _stream()),
                    "ngmix_lm_eval_batch")
                _lib.check(L.ngmix_lm_advance_batch(
                    _dptr(d_states), nobj, _dptr(d_start), _dptr(d_sband),
                    _dptr(d_sums), _dptr(d_nact), _stream()),
                    "ngmix_lm_advance_batch")
                rounds += 1
                if rounds % check_every == 0 or rounds > maxfev:
                    if int(d_nact.item()) == 0:
                        break
                if rounds > maxfev + 2:
                    raise RuntimeError("batched LM did not terminate")
        torch.cuda.synchronize(dev)
        # seconds in the lock-step loop (kernels + one 4-byte readback per round)
        self.loop_seconds = time.perf_counter() - t0
        states = d_states.cpu().numpy().reshape(-1).view(_lib.LM_STATE_DTYPE)
        self.rounds = rounds
        res = self._package(states, npars, stamps, obj_start)
        self._add_stats(res, stamps, psf, sobj, sband, obj_start, nband)
        return res

    # ------------------------------------------------------------------
    def _package(self, st, npars, stamps, obj_start):
        """run_leastsq's packaging (leastsqbound.py:33-155), vectorised"""
        nobj = st.size
        n = npars
        ier = st["info"].astype(np.int64)
        pars = st["x"][:, :n].copy()
        flags = np.zeros(nobj, dtype=np.int64)
        pcov0 = np.full((nobj, n, n), CDEF)
        pcov = np.full((nobj, n, n), CDEF)
        perr = np.full((nobj, n), CDEF)

        notfinite = ier == 0                     # no finite starting residual
        flags[notfinite] |= LM_FUNC_NOTFINITE
        hard = ier > 4
        flags[hard] |= 2 ** (ier[hard] - 5)
        ok = ~(notfinite | hard)

        # cov_x as scipy.optimize.leastsq forms it from fjac / ipvt:
        # inv((R P^T)^T (R P^T)) = P R^-1 R^-T P^T, by back substitution
        R = st["R"][:, :n, :n]
        diagR = np.diagonal(R, axis1=1, axis2=2)
        singular = ok & ((diagR == 0.0).any(axis=1) | ~np.isfinite(R).all(axis=(1, 2)))
        good = ok & ~singular
        with np.errstate(all="ignore"):
            Rinv = _upper_inverse(np.where(good[:, None, None], R, np.eye(n)))
            cov_piv = Rinv @ np.transpose(Rinv, (0, 2, 1))
        ipvt = st["ipvt"][:, :n].astype(np.int64)
        rows = np.arange(nobj)[:, None, None]
        inv = np.empty_like(cov_piv)
        inv[rows, ipvt[:, :, None], ipvt[:, None, :]] = cov_piv
        badinv = good & ~np.isfinite(inv).all(axis=(1, 2))
        singular |= badinv
        good &= ~badinv
        flags[singular] |= LM_SINGULAR_MATRIX
        pcov0[good] = inv[good]

        # pars_cov = pars_cov0 * sum(fdiff^2) / dof at the solution
        npix_obj = np.add.reduceat(stamps.npix_kept.astype(np.int64), obj_start[:-1])
        dof = npix_obj - n
        zero_dof = good & (dof == 0)
        flags[zero_dof] |= ZERO_DOF
        good &= ~zero_dof
        with np.errstate(all="ignore"):
            s_sq = st["fnorm"] ** 2 / dof
            pc = inv * s_sq[:, None, None]
        # _test_cov (leastsqbound.py:158-184): a negative eigenvalue of the
        # symmetric matrix <=> a negative pivot of its LDL^T (inertia)
        cflags = np.zeros(nobj, dtype=np.int64)
        finite = np.isfinite(pc).all(axis=(1, 2))
        tmp = np.where((good & finite)[:, None, None], pc, np.eye(n))
        cflags[good & ~finite] |= EIG_NOTFINITE
        cflags[good & finite & _has_negative_pivot(tmp)] |= LM_NEG_COV_EIG
        d = np.diagonal(tmp, axis1=1, axis2=2)
        cflags[good & finite & (d < 0).any(axis=1)] |= LM_NEG_COV_DIAG
        flags |= np.where(good, cflags, 0)
        pcov[good] = pc[good]
        goodcov = good & (cflags == 0)
        with np.errstate(invalid="ignore"):
            perr[goodcov] = np.sqrt(d)[goodcov]

        bad_pars = notfinite | hard
        pars[bad_pars] = PDEF
        return {
            "model": self.model,
            "flags": flags,
            "nfev": np.where(notfinite, -1, st["nfev"].astype(np.int64)),
            "njev": st["njev"].astype(np.int64),
            "ier": ier,
            "pars": pars,
            "pars_err": perr,
            "pars_cov0": pcov0,
            "pars_cov": pcov,
            "npix": npix_obj,
            "dof": dof,
        }

    def _add_stats(self, res, stamps, psf, sobj, sband, obj_start, nband):
        """FitModel.set_fit_result (results.py:45-72, 398-408, 1079-1109) for
        the fits with flags == 0: one batched get_loglike at the solutions"""
        torch = _torch()
        nobj = res["flags"].size
        ok = res["flags"] == 0
        pars = res["pars"]
        usable = np.where(ok[:, None], pars, 0.0)
        # a harmless model for failed fits (their statistics are not reported)
        usable[~ok, 4] = 1.0
        usable[~ok, 5:] = 1.0
        band_pars = np.empty((stamps.n, 6))
        band_pars[:, :5] = usable[sobj, :5]
        band_pars[:, 5] = usable[sobj, 5 + sband]
        gm0, st0 = GMixBatch.from_pars(band_pars, self.model, device=stamps.device)
        gm = gm0
        if psf is not None:
            gm, _ = gm0.convolve(psf)
        out, st1 = stamps.loglike(gm)
        out = out.cpu().numpy()
        lnprob = np.add.reduceat(out[:, 0], obj_start[:-1])
        s2n_numer = np.add.reduceat(out[:, 1], obj_start[:-1])
        s2n_denom = np.add.reduceat(out[:, 2], obj_start[:-1])
        npix = np.add.reduceat(out[:, 3], obj_start[:-1]).astype(np.int64)
        nan = np.full(nobj, np.nan)
        with np.errstate(all="ignore"):
            s2n = np.where(s2n_denom > 0, s2n_numer / np.sqrt(s2n_denom), 0.0)
            dof = npix - pars.shape[1]
            res["lnprob"] = np.where(ok, lnprob, nan)
            res["s2n_numer"] = np.where(ok, s2n_numer, nan)
            res["s2n_denom"] = np.where(ok, s2n_denom, nan)
            res["npix"] = npix
            res["dof"] = dof
            res["chi2per"] = np.where(ok, lnprob / (-0.5) / dof, nan)
            res["s2n_w"] = np.where(ok, s2n, nan)
            res["s2n"] = res["s2n_w"]
            pc = res["pars_cov"]
            res["g"] = pars[:, 2:4].copy()
            res["g_cov"] = pc[:, 2:4, 2:4].copy()
            res["g_err"] = res["pars_err"][:, 2:4].copy()
            res["T"] = pars[:, 4].copy()
            res["T_err"] = np.sqrt(pc[:, 4, 4])
            if nband == 1:
                res["flux"] = pars[:, 5].copy()
                res["flux_err"] = np.sqrt(pc[:, 5, 5])
            else:
                res["flux"] = pars[:, 5:].copy()
                res["flux_cov"] = pc[:, 5:, 5:].copy()
                res["flux_err"] = np.sqrt(np.diagonal(res["flux_cov"], axis1=1,
                                                      axis2=2))
        del torch
