"""
Priors for the batched LM driver (LMBatchFitter(prior=...)).

The reference evaluates a joint prior once per object and evaluation:
prior.fill_fdiff(pars, fdiff) writes sqrt(-2 ln p) rows at the head of the LM
residual vector (ngmix/joint_prior.py:86-120, results.py:454,468-484) and
prior.bounds feeds leastsqbound's parameter transform (results.py:389-396).
A lock-step batch wants the rows of all N objects at once, so a batch prior is
any object with

    bounds                        None or [(lo, hi)] * npars, None = unbounded
    fill_fdiff_batch(pars)        (N, npars) float64 tensor -> (rows, bad):
                                  rows (N, k) = sqrt(-2 ln p) per prior term,
                                  bad (N,) bool = the reference would raise
                                  GMixRangeError there (whole fdiff = -inf)
    get_lnprob_batch(pars)        (N,) ln p, -inf where bad

evaluated with torch on the device the fit runs on.  PriorSimpleSepBatch is the
separable joint prior of the reference's PriorSimpleSep built from the
elementwise terms below; PriorBatchAdapter serves any reference-style prior
object (fill_fdiff / get_lnprob_scalar / bounds) by looping over the objects
on the host.
"""
import math

import numpy as np

from .gexceptions import GMixRangeError

__all__ = ["GaussianCen", "GPriorBA", "Flat", "TwoSidedErf", "Normal", "LogNormal",
           "TruncatedGaussian", "PriorSepBatch", "PriorSimpleSepBatch", "PriorBatchAdapter",
           "as_batch_prior", "prior_normal_sums"]


def _torch():
    import torch
    return torch


def _root_of_lnprob(lnp):
    """sqrt(-2 ln p) with ln p > 0 counted as 0: the residual row of a term
    whose density is not a gaussian in the parameter"""
    torch = _torch()
    return torch.sqrt(torch.clamp(-2.0 * lnp, min=0.0))


class GaussianCen(object):
    """CenPrior (priors/multivariate.py:24-71): independent gaussians on the
    two centre offsets"""

    def __init__(self, cen1, cen2, sigma1, sigma2):
        self.cen1, self.cen2 = float(cen1), float(cen2)
        self.sinv1, self.sinv2 = 1.0 / float(sigma1), 1.0 / float(sigma2)
        self.s2inv1, self.s2inv2 = 1.0 / float(sigma1) ** 2, 1.0 / float(sigma2) ** 2

    def lnprob_sep(self, x1, x2):
        d1 = self.cen1 - x1
        d2 = self.cen2 - x2
        return -0.5 * d1 * d1 * self.s2inv1, -0.5 * d2 * d2 * self.s2inv2

    def fdiff(self, x1, x2):
        """CenPrior.get_fdiff: signed (x - cen) / sigma"""
        return (x1 - self.cen1) * self.sinv1, (x2 - self.cen2) * self.sinv2


class GPriorBA(object):
    """Bernstein & Armstrong shape prior (priors/shape.py:446-564):
    ln p = 2 ln(1 - g^2) - g^2 / (2 sigma^2); g^2 >= 1 is a range error"""

    def __init__(self, sigma):
        self.sig2inv = 1.0 / float(sigma) ** 2

    def lnprob2d(self, g1, g2):
        torch = _torch()
        gsq = g1 * g1 + g2 * g2
        omgsq = 1.0 - gsq
        bad = omgsq <= 0.0
        safe = torch.where(bad, torch.ones_like(omgsq), omgsq)
        return 2.0 * torch.log(safe) - 0.5 * gsq * self.sig2inv, bad

    def fdiff(self, g1, g2):
        lnp, bad = self.lnprob2d(g1, g2)
        return _root_of_lnprob(lnp), bad


class Flat(object):
    """FlatPrior (priors/priors.py:49-100): ln p = 0 inside [minval, maxval],
    a range error outside; bounds, if given, go to leastsqbound"""

    def __init__(self, minval, maxval, bounds=None):
        self.minval, self.maxval = float(minval), float(maxval)
        self.bounds = bounds

    def lnprob(self, x):
        torch = _torch()
        return torch.zeros_like(x), (x < self.minval) | (x > self.maxval)

    fdiff = lnprob          # the row is 0 inside, a range error outside


class TwoSidedErf(object):
    """TwoSidedErf (priors/priors.py:186-251): flat between two error-function
    edges; p <= 0 gives ln p = -inf (a zero-probability row, not an error)"""

    def __init__(self, minval, width_at_min, maxval, width_at_max, bounds=None):
        self.minval, self.width_at_min = float(minval), float(width_at_min)
        self.maxval, self.width_at_max = float(maxval), float(width_at_max)
        self.bounds = bounds

    def lnprob(self, x):
        torch = _torch()
        p = 0.5 * torch.erf((self.maxval - x) / self.width_at_max) + \
            0.5 * torch.erf((x - self.minval) / self.width_at_min)
        pos = p > 0.0
        lnp = torch.where(pos, torch.log(torch.where(pos, p, torch.ones_like(p))),
                          torch.full_like(p, -math.inf))
        return lnp, torch.zeros_like(pos)

    def fdiff(self, x):
        lnp, bad = self.lnprob(x)
        return _root_of_lnprob(lnp), bad


class Normal(object):
    """Normal (priors/priors.py:395-434): ln p = -(x - mean)^2 / (2 sigma^2);
    bounds, if given, go to leastsqbound"""

    def __init__(self, mean, sigma, bounds=None):
        self.mean, self.sigma = float(mean), float(sigma)
        self.sinv = 1.0 / self.sigma
        self.s2inv = 1.0 / self.sigma ** 2
        self.bounds = bounds

    def lnprob(self, x):
        torch = _torch()
        diff = self.mean - x
        return -0.5 * diff * diff * self.s2inv, torch.zeros_like(x, dtype=torch.bool)

    def fdiff(self, x):
        torch = _torch()
        return (x - self.mean) * self.sinv, torch.zeros_like(x, dtype=torch.bool)


class LogNormal(object):
    """LogNormal (priors/priors.py:674-800): mean and sigma of the variate,
    optional shift; ln p = 0 at the mode; x - shift <= 0 is a range error"""

    def __init__(self, mean, sigma, shift=None, bounds=None):
        if mean <= 0:
            raise ValueError("mean %s is < 0" % mean)
        self.mean, self.sigma = float(mean), float(sigma)
        self.shift = None if shift is None else float(shift)
        spread = 1 + self.sigma ** 2 / self.mean ** 2
        self.logmean = math.log(self.mean) - 0.5 * math.log(spread)
        self.logvar = math.log(spread)
        self.logivar = 1.0 / self.logvar
        log_mode = self.logmean - self.logvar
        self.lnprob_max = -0.5 * self.logivar * (log_mode - self.logmean) ** 2 - log_mode
        self.bounds = bounds

    def lnprob(self, x):
        torch = _torch()
        if self.shift is not None:
            x = x - self.shift
        bad = ~(x > 0.0)
        logx = torch.log(torch.where(bad, torch.ones_like(x), x))
        chi2 = self.logivar * (logx - self.logmean) ** 2
        return -0.5 * chi2 - logx - self.lnprob_max, bad

    def fdiff(self, x):
        lnp, bad = self.lnprob(x)
        return _root_of_lnprob(lnp), bad


class TruncatedGaussian(object):
    """TruncatedGaussian (priors/priors.py:1046-1110): a gaussian on
    [minval, maxval], a range error outside"""

    def __init__(self, mean, sigma, minval, maxval, bounds=None):
        self.mean, self.sigma = float(mean), float(sigma)
        self.sinv = 1.0 / self.sigma
        self.ivar = 1.0 / self.sigma ** 2
        self.minval, self.maxval = float(minval), float(maxval)
        self.bounds = bounds

    def lnprob(self, x):
        diff = x - self.mean
        return -0.5 * diff * diff * self.ivar, (x < self.minval) | (x > self.maxval)

    def fdiff(self, x):
        return (x - self.mean) * self.sinv, (x < self.minval) | (x > self.maxval)


class PriorSepBatch(object):
    """
    A separable joint prior over a batch: a centre term (two parameters, two
    rows), a shape term (two parameters, one row) and one 1-d term per
    remaining parameter, in parameter order -- the row of terms
    joint_prior.py's classes are made of.  rows_from_lnprob: every row is
    sqrt(clip(-2 ln p, 0)) of its term (PriorSimpleSep, joint_prior.py:86-120)
    or each term's own residual (PriorBDSep / PriorBDFSep: signed
    (x - mean) / sigma for the gaussian terms, joint_prior.py:341-378).
    """

    def __init__(self, cen_prior, g_prior, terms, rows_from_lnprob=True):
        self.cen_prior = cen_prior
        self.g_prior = g_prior
        self.terms = list(terms)
        self.rows_from_lnprob = bool(rows_from_lnprob)
        bounds = [(None, None)] * 4
        some = False
        for p in self.terms:
            b = getattr(p, "bounds", None)
            if b is not None:
                some = True
                bounds.append((b[0], b[1]))
            else:
                bounds.append((None, None))
        self.bounds = bounds if some else None

    # how many 1-d terms stand between T and the fluxes (set by as_batch_prior
    # for the bulge+disk priors; 0: [T, F_band...])
    nmid = 0

    def descriptor(self):
        """the ngmix_simple_sep_prior record of this prior, for the kernel
        that evaluates the prior rows of all fits in one launch
        (ngmix_lm_prior_sums_batch); None when a term is not one of the
        kinds the kernel knows or there are more terms than it holds, the
        torch path then serves"""
        from . import _lib
        nband = len(self.terms) - 1 - self.nmid
        if not isinstance(self.cen_prior, GaussianCen) or \
                not isinstance(self.g_prior, GPriorBA) or nband < 1 or \
                nband > _lib.PRIOR_MAXBAND or not 0 <= self.nmid <= _lib.PRIOR_MAXMID:
            return None
        d = np.zeros(1, dtype=_lib.SIMPLE_SEP_PRIOR_DTYPE)
        d["cen1"], d["cen2"] = self.cen_prior.cen1, self.cen_prior.cen2
        d["cen_s2inv1"], d["cen_s2inv2"] = self.cen_prior.s2inv1, self.cen_prior.s2inv2
        d["cen_sinv1"], d["cen_sinv2"] = self.cen_prior.sinv1, self.cen_prior.sinv2
        d["g_sig2inv"] = self.g_prior.sig2inv
        d["nband"], d["nmid"] = nband, self.nmid
        d["rows_mode"] = _lib.PRIOR_ROWS_LNPROB if self.rows_from_lnprob else _lib.PRIOR_ROWS_FDIFF

        def term(p):
            if isinstance(p, TwoSidedErf):
                return _lib.PRIOR_TWO_SIDED_ERF, [p.minval, p.width_at_min, p.maxval,
                                                  p.width_at_max]
            if isinstance(p, Flat):
                return _lib.PRIOR_FLAT, [p.minval, p.maxval, 0.0, 0.0]
            if isinstance(p, Normal):
                return _lib.PRIOR_NORMAL, [p.mean, p.sigma, 0.0, 0.0]
            if isinstance(p, LogNormal):
                return _lib.PRIOR_LOGNORMAL, [p.logmean, p.logivar, p.lnprob_max,
                                              0.0 if p.shift is None else p.shift]
            if isinstance(p, TruncatedGaussian):
                return _lib.PRIOR_TRUNCATED_GAUSSIAN, [p.mean, p.sigma, p.minval, p.maxval]
            return None, None
        kinds = [term(p) for p in self.terms]
        if any(k is None for k, _ in kinds):
            return None
        d["T_kind"], d["T_par"] = kinds[0]
        for m in range(self.nmid):
            d["mid_kind"][0, m], d["mid_par"][0, m] = kinds[1 + m]
        for b in range(nband):
            d["F_kind"][0, b], d["F_par"][0, b] = kinds[1 + self.nmid + b]
        return d

    def _lnprobs(self, pars):
        torch = _torch()
        l1, l2 = self.cen_prior.lnprob_sep(pars[:, 0], pars[:, 1])
        lg, bad = self.g_prior.lnprob2d(pars[:, 2], pars[:, 3])
        cols = [l1, l2, lg]
        for i, p in enumerate(self.terms):
            lp, bp = p.lnprob(pars[:, 4 + i])
            bad = bad | bp
            cols.append(lp)
        return torch.stack(cols, dim=1), bad

    def fill_fdiff_batch(self, pars):
        torch = _torch()
        if self.rows_from_lnprob:
            lnp, bad = self._lnprobs(pars)
            return _root_of_lnprob(lnp), bad
        r1, r2 = self.cen_prior.fdiff(pars[:, 0], pars[:, 1])
        rg, bad = self.g_prior.fdiff(pars[:, 2], pars[:, 3])
        cols = [r1, r2, rg]
        for i, p in enumerate(self.terms):
            rp, bp = p.fdiff(pars[:, 4 + i])
            bad = bad | bp
            cols.append(rp)
        return torch.stack(cols, dim=1), bad

    def get_lnprob_batch(self, pars):
        torch = _torch()
        lnp, bad = self._lnprobs(pars)
        tot = lnp.sum(dim=1)
        return torch.where(bad, torch.full_like(tot, -math.inf), tot)


class PriorSimpleSepBatch(PriorSepBatch):
    """
    PriorSimpleSep (joint_prior.py:10-120) over a batch: rows
    [cen1, cen2, g, T, F_band...] = sqrt(clip(-2 ln p, 0)) and the bounds of
    the T and flux terms.
    """

    def __init__(self, cen_prior, g_prior, T_prior, F_prior):
        self.T_prior = T_prior
        self.F_priors = list(F_prior) if isinstance(F_prior, (list, tuple)) else [F_prior]
        self.nband = len(self.F_priors)
        super().__init__(cen_prior, g_prior, [T_prior] + self.F_priors, rows_from_lnprob=True)


class PriorBatchAdapter(object):
    """
    a reference-style joint prior (fill_fdiff(pars, fdiff) -> nrows,
    get_lnprob_scalar(pars), optional .bounds) evaluated object by object on
    the host: correct for any prior, at one Python call per object and
    evaluation.  max_rows: the length of the buffer handed to fill_fdiff (its
    return value says how many rows it wrote); by default two more than there
    are parameters, which holds every joint prior of joint_prior.py.
    """

    def __init__(self, prior, max_rows=None):
        self.prior = prior
        self.max_rows = None if max_rows is None else int(max_rows)
        self.bounds = getattr(prior, "bounds", None)

    def fill_fdiff_batch(self, pars):
        torch = _torch()
        p = pars.detach().cpu().numpy()
        n = p.shape[0]
        max_rows = self.max_rows if self.max_rows is not None else p.shape[1] + 2
        rows = np.zeros((n, max_rows))
        bad = np.zeros(n, dtype=bool)
        buf = np.zeros(max_rows)
        nrows = max_rows
        for i in range(n):
            buf[:] = 0.0
            try:
                nrows = self.prior.fill_fdiff(p[i], buf)
                rows[i] = buf
            except GMixRangeError:
                bad[i] = True
        return (torch.from_numpy(rows[:, :nrows].copy()).to(pars.device),
                torch.from_numpy(bad).to(pars.device))

    def get_lnprob_batch(self, pars):
        torch = _torch()
        p = pars.detach().cpu().numpy()
        out = np.empty(p.shape[0])
        for i in range(p.shape[0]):
            try:
                out[i] = self.prior.get_lnprob_scalar(p[i])
            except GMixRangeError:
                out[i] = -np.inf
        return torch.from_numpy(out).to(pars.device)


def _batch_term(p):
    """the batch form of a 1-d host prior of priors.py, or None"""
    from . import priors as P
    kind = type(p)
    if kind is P.FlatPrior:
        return Flat(p.minval, p.maxval, bounds=p.bounds)
    if kind is P.TwoSidedErf:
        return TwoSidedErf(p.minval, p.width_at_min, p.maxval, p.width_at_max, bounds=p.bounds)
    if kind is P.Normal:
        return Normal(p.mean, p.sigma, bounds=p.bounds)
    if kind is P.LogNormal:
        return LogNormal(p.mean, p.sigma, shift=p.shift, bounds=p.bounds)
    if kind is P.TruncatedGaussian:
        return TruncatedGaussian(p.mean, p.sigma, p.minval, p.maxval, bounds=p.bounds)
    return None


def as_batch_prior(prior):
    """
    What a caller passed as prior=, in the form the lock-step driver takes
    (fill_fdiff_batch / get_lnprob_batch / bounds):

      None or a batch prior        unchanged
      joint_prior.PriorSimpleSep,  of a CenPrior, a GPriorBA and FlatPrior /
        PriorGalsimSimpleSep,      TwoSidedErf / Normal / LogNormal /
        PriorBDFSep, PriorBDSep    TruncatedGaussian terms: the batch prior of
                                   the same densities, evaluated for all fits
                                   at once on the device, in one kernel inside
                                   the device loop (up to three bands; torch
                                   ops beyond)
      any other object with        PriorBatchAdapter: fill_fdiff /
        fill_fdiff                 get_lnprob_scalar per object on the host
    """
    if prior is None or hasattr(prior, "fill_fdiff_batch"):
        return prior
    from . import priors as P
    from . import joint_prior as J
    simple = type(prior) in (J.PriorSimpleSep, J.PriorGalsimSimpleSep)
    if (simple or type(prior) in (J.PriorBDFSep, J.PriorBDSep)) and \
            type(prior.cen_prior) is P.CenPrior and type(prior.g_prior) is P.GPriorBA and \
            isinstance(prior.F_priors, list):
        terms = [_batch_term(p) for p in prior._scalar_terms()]
        if all(t is not None for t in terms):
            c = prior.cen_prior
            cen = GaussianCen(c.cen1, c.cen2, c.sigma1, c.sigma2)
            g = GPriorBA(prior.g_prior.sigma)
            if simple:
                return PriorSimpleSepBatch(cen, g, terms[0],
                                           terms[1:] if len(terms) > 2 else terms[1])
            batch = PriorSepBatch(cen, g, terms, rows_from_lnprob=False)
            batch.nmid = len(prior._middle)
            return batch
    if not hasattr(prior, "fill_fdiff"):
        raise TypeError("prior must offer fill_fdiff_batch or fill_fdiff, got %r" % (prior,))
    return PriorBatchAdapter(prior)


def bounds_arrays(bounds, npars):
    """(lo, hi) float64 arrays with -inf / +inf for None"""
    lo = np.full(npars, -np.inf)
    hi = np.full(npars, np.inf)
    if bounds is not None:
        if len(bounds) != npars:
            raise ValueError("length of bounds != number of parameters")
        for i, (a, b) in enumerate(bounds):
            if a is not None:
                lo[i] = a
            if b is not None:
                hi[i] = b
    return lo, hi


def prior_normal_sums(prior, xt, xstep=None, hstep=None, step_rel=1.0e-8):
    """
    The prior rows of N objects at their trial points xt (N, n), reduced to
    the normal-equation sums ngmix_lm_advance_batch takes as obj_sums:
    (N, n(n+1)/2 + n + 1) = [J^T J upper triangle | J^T r | r.r].

    The jacobian of the rows is by differences, as in the reference:
    analytic mode (xstep None): forward steps step_rel * max(1, |x_j|), backward
    where the forward point is out of range (results.py:572-625); rows that are
    not finite get zero derivatives.  Forward-difference mode: the state's own
    fdjac2 points (column j of xt replaced by xstep[:, j], divided by
    hstep[:, j]), which is what lmdif does to the whole residual vector.

    Returns (sums, ff) with ff (N,) = r.r (+inf where the prior is out of range).
    """
    torch = _torch()
    N, n = xt.shape
    r0, bad0 = prior.fill_fdiff_batch(xt)
    k = r0.shape[1]
    fin0 = torch.isfinite(r0)
    J = torch.zeros((N, k, n), dtype=torch.float64, device=xt.device)
    for j in range(n):
        xp = xt.clone()
        if xstep is None:
            step = step_rel * torch.clamp(xt[:, j].abs(), min=1.0)
            xp[:, j] = xt[:, j] + step
            rj, badj = prior.fill_fdiff_batch(xp)
            if bool(badj.any()):
                xm = xt.clone()
                xm[:, j] = xt[:, j] - step
                rm, badm = prior.fill_fdiff_batch(xm)
                rj = torch.where(badj[:, None], rm, rj)
                step = torch.where(badj, -step, step)
                badj = badj & badm
        else:
            step = hstep[:, j]
            xp[:, j] = xstep[:, j]
            rj, badj = prior.fill_fdiff_batch(xp)
        good = fin0 & torch.isfinite(rj) & ~badj[:, None] & ~bad0[:, None]
        d = (rj - r0) / step[:, None]
        J[:, :, j] = torch.where(good, d, torch.zeros_like(d))
    rz = torch.where(fin0, r0, torch.zeros_like(r0))
    A = torch.einsum("nka,nkb->nab", J, J)
    g = torch.einsum("nka,nk->na", J, rz)
    ff = (r0 * r0).sum(dim=1)
    ff = torch.where(bad0, torch.full_like(ff, math.inf), ff)
    iu = torch.triu_indices(n, n, device=xt.device)
    sums = torch.cat([A[:, iu[0], iu[1]], g, ff[:, None]], dim=1).contiguous()
    return sums, ff
