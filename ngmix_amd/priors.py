"""
Priors on the parameters of the LM fits -- the host objects a caller hands to
Fitter(prior=...), to the guessers and to bootstrap_batch, with the
reference's names, arguments, numbers and random-number call order
(ngmix/priors/priors.py, shape.py, multivariate.py, random.py, kde.py): every
ln p peaks at 0 so that sqrt(-2 ln p) can stand in for (data - model) / err in
the residual vector of a least-squares fit.

These are scalar host objects (numpy); the lock-step LM driver evaluates the
same densities for all fits of a batch on the device -- prior_batch.py turns a
joint prior built from CenPrior, GPriorBA and FlatPrior / TwoSidedErf / Normal
terms into its batch form (prior_batch.as_batch_prior), anything else is
served object by object through PriorBatchAdapter.

All samplers draw from the RandomState the prior was built with, in the
reference's order, so seeded guesses agree draw for draw
(tests/golden/priors.npz, oracle/gen_golden_priors.py).
"""
import math

import numpy as np

from .gexceptions import GMixRangeError
from .defaults import LOWVAL

__all__ = [
    "make_rng", "srandu", "PriorBase", "FlatPrior", "TwoSidedErf", "Normal", "LMBounds",
    "Bounded1D", "LimitPDF", "LogNormal", "Sinh", "TruncatedGaussian", "GPriorBase",
    "GPriorGauss", "GPriorBA", "ZDisk2D", "CenPrior", "SimpleGauss2D", "KDE",
]


def make_rng(rng=None):
    """rng, or a RandomState seeded from numpy's global generator
    (priors/random.py:4-13)"""
    if rng is None:
        rng = np.random.RandomState(np.random.randint(0, 2 ** 30))
    return rng


def srandu(nrand=None, *, rng):
    """uniform deviates in [-1, 1) (priors/random.py:16-32)"""
    return rng.uniform(low=-1.0, high=1.0, size=nrand)


def _count(nrand):
    """(number of draws, whether the caller wants a scalar back)"""
    return (1, True) if nrand is None else (nrand, False)


def _accept_reject(nrand, propose):
    """
    Fill nrand slots by repeated proposals: propose(nleft) returns (candidates,
    keep-mask) for as many candidates as are still missing; kept ones are
    stored in the order drawn.  One proposal per pass, sized by what is left,
    is what fixes the sequence of generator calls.
    """
    out = None
    ngood = 0
    while ngood < nrand:
        cand, keep = propose(nrand - ngood)
        cols = cand if isinstance(cand, tuple) else (cand,)
        if out is None:
            out = tuple(np.zeros(nrand) for _ in cols)
        idx = np.flatnonzero(keep)
        for dst, src in zip(out, cols):
            dst[ngood:ngood + idx.size] = src[idx]
        ngood += idx.size
    return out if len(out) > 1 else out[0]


def _clipped_root(lnp):
    """sqrt(-2 ln p), with ln p > 0 (round-off at the mode) counted as 0"""
    chi2 = -2 * lnp
    if chi2 < 0.0:
        chi2 = 0.0
    return np.sqrt(chi2)


class PriorBase(object):
    """rng is required; bounds (lo, hi) or None go to leastsqbound through the
    joint prior (priors/priors.py:12-46)"""

    def __init__(self, rng, bounds=None):
        assert rng is not None, 'rng is a required argument'
        self.bounds = bounds
        self.rng = make_rng(rng=rng)

    def has_bounds(self):
        return getattr(self, "bounds", None) is not None


class FlatPrior(PriorBase):
    """p = 1 on [minval, maxval]; a value outside is a GMixRangeError
    (priors/priors.py:49-187)"""

    def __init__(self, minval, maxval, rng):
        super().__init__(rng=rng)
        self.minval = minval
        self.maxval = maxval

    def _check_scalar(self, val):
        if val < self.minval or val > self.maxval:
            raise GMixRangeError("value %s out of range: [%s,%s]" % (val, self.minval,
                                                                      self.maxval))

    def _check_array(self, vals):
        if np.any((vals < self.minval) | (vals > self.maxval)):
            raise GMixRangeError("values were out of range: [%s,%s]" % (self.minval,
                                                                        self.maxval))

    def get_prob_scalar(self, val):
        self._check_scalar(val)
        return 1.0

    def get_lnprob_scalar(self, val):
        self._check_scalar(val)
        return 0.0

    def get_prob_array(self, vals):
        self._check_array(vals)
        return vals * 0 + 1.0

    def get_lnprob_array(self, vals):
        # a scalar zero, whatever the shape of vals (it broadcasts in the sums
        # of the joint priors)
        self._check_array(vals)
        return 0.0

    def get_fdiff(self, val):
        self._check_scalar(val)
        return 0.0

    def sample(self, nrand=None):
        n, scalar = _count(nrand)
        vals = self.minval + (self.maxval - self.minval) * self.rng.uniform(size=n)
        return vals[0] if scalar else vals


class TwoSidedErf(PriorBase):
    """flat between two error-function edges: p = erf((maxval - x) /
    width_at_max) / 2 + erf((x - minval) / width_at_min) / 2; p <= 0 is
    ln p = -inf, not an error (priors/priors.py:190-388)"""

    def __init__(self, minval, width_at_min, maxval, width_at_max, rng):
        super().__init__(rng=rng)
        self.minval = minval
        self.width_at_min = width_at_min
        self.maxval = maxval
        self.width_at_max = width_at_max

    def get_prob_scalar(self, val):
        upper = 0.5 * math.erf((self.maxval - val) / self.width_at_max)
        lower = 0.5 * math.erf((val - self.minval) / self.width_at_min)
        return upper + lower

    def get_lnprob_scalar(self, val):
        p = self.get_prob_scalar(val)
        return np.log(p) if p > 0.0 else LOWVAL

    def get_prob_array(self, vals):
        vals = np.asarray(vals, dtype="f8").reshape(-1)
        return np.array([self.get_prob_scalar(v) for v in vals], dtype="f8")

    def get_lnprob_array(self, vals):
        p = self.get_prob_array(vals)
        lnp = np.full(p.size, LOWVAL)
        pos = p > 0.0
        lnp[pos] = np.log(p[pos])
        return lnp

    def get_fdiff(self, val):
        if isinstance(val, np.ndarray):
            flat = np.asarray(val, dtype="f8").reshape(-1)
            return np.array([self._get_fdiff_scalar(v) for v in flat], dtype="f8")
        return self._get_fdiff_scalar(val)

    def _get_fdiff_scalar(self, val):
        return _clipped_root(self.get_lnprob_scalar(val))

    def sample(self, nrand=None):
        """rejection under the curve over five widths either side"""
        n, scalar = _count(nrand)
        lo = self.minval - 5.0 * self.width_at_min
        hi = self.maxval + 5.0 * self.width_at_max
        rng = self.rng

        def propose(nleft):
            x = rng.uniform(low=lo, high=hi, size=nleft)
            p = self.get_prob_array(x)
            return x, rng.uniform(size=nleft) < p
        vals = _accept_reject(n, propose)
        return vals[0] if scalar else vals


class Normal(PriorBase):
    """ln p = -(mean - x)^2 / (2 sigma^2); bounds go to leastsqbound
    (priors/priors.py:391-505)"""

    def __init__(self, mean, sigma, rng, bounds=None):
        super().__init__(rng=rng, bounds=bounds)
        self.mean = mean
        self.sigma = sigma
        self.sinv = 1.0 / sigma
        self.s2inv = 1.0 / sigma ** 2
        self.ndim = 1

    def get_lnprob(self, val):
        diff = self.mean - val
        return -0.5 * diff * diff * self.s2inv

    get_lnprob_scalar = get_lnprob
    get_lnprob_array = get_lnprob

    def get_prob(self, val):
        return np.exp(self.get_lnprob(val))

    get_prob_array = get_prob

    def get_prob_scalar(self, val):
        return math.exp(self.get_lnprob(val))

    def get_fdiff(self, val):
        return (val - self.mean) * self.sinv

    def sample(self, nrand=None, size=None):
        if size is None and nrand is not None:
            size = nrand
        return self.rng.normal(loc=self.mean, scale=self.sigma, size=size)


class LMBounds(PriorBase):
    """no density at all (fdiff = 0): only the bounds, for leastsqbound
    (priors/priors.py:508-569)"""

    def __init__(self, minval, maxval, rng):
        super().__init__(rng)
        self.bounds = (minval, maxval)
        self.mean = (minval + maxval) / 2.0
        self.sigma = (maxval - minval) * 0.28      # ~ 1 / sqrt(12)

    def get_fdiff(self, val):
        return 0.0 * val

    def sample(self, nrand=None):
        return self.rng.uniform(low=self.bounds[0], high=self.bounds[1], size=nrand)


class Bounded1D(PriorBase):
    """another pdf's samples restricted to the open interval `bounds`
    (priors/priors.py:572-667); takes no rng of its own"""

    def __init__(self, pdf, bounds):
        self.pdf = pdf
        self.set_limits(bounds)

    def set_limits(self, limits):
        try:
            two = len(limits) == 2
        except TypeError:
            two = False
        if not two:
            raise ValueError("expected bounds to be 2-element sequence, got %s" % (limits,))
        if limits[0] >= limits[1]:
            raise ValueError("bounds[0] must be less than bounds[1], got: %s" % (limits,))
        self.limits = limits
        self.bounds = limits

    def sample(self, nrand=None, size=None):
        if size is None and nrand is not None:
            size = nrand
        lo, hi = self.bounds

        def propose(nleft):
            x = self.pdf.sample(nleft)
            return x, (x > lo) & (x < hi)
        vals = _accept_reject(1 if size is None else size, propose)
        return vals[0] if size is None else vals


LimitPDF = Bounded1D


class LogNormal(PriorBase):
    """log-normal with the given mean and sigma of the VARIATE (not of its
    log), optionally shifted; ln p is 0 at the mode; val <= shift is a
    GMixRangeError (priors/priors.py:674-972)"""

    def __init__(self, mean, sigma, rng, shift=None):
        super().__init__(rng=rng)
        if mean <= 0:
            raise ValueError("mean %s is < 0" % mean)
        self.shift = shift
        self.mean = mean
        self.sigma = sigma
        spread = 1 + self.sigma ** 2 / self.mean ** 2
        self.logmean = np.log(self.mean) - 0.5 * np.log(spread)
        self.logvar = np.log(spread)
        self.logsigma = np.sqrt(self.logvar)
        self.logivar = 1.0 / self.logvar
        self.log_mode = self.logmean - self.logvar
        self.mode = np.exp(self.log_mode)
        chi2 = self.logivar * (self.log_mode - self.logmean) ** 2
        self.lnprob_max = -0.5 * chi2 - self.log_mode

    def _lnprob_of_log(self, logval):
        chi2 = self.logivar * (logval - self.logmean) ** 2
        return -0.5 * chi2 - logval - self.lnprob_max

    def get_lnprob_scalar(self, val):
        if self.shift is not None:
            val = val - self.shift
        if val <= 0:
            raise GMixRangeError("values of val must be > 0")
        return self._lnprob_of_log(np.log(val))

    def get_lnprob_array(self, vals):
        vals = np.asarray(vals, dtype="f8")
        if self.shift is not None:
            vals = vals - self.shift
        if np.any(vals <= 0):
            raise GMixRangeError("values must be > 0")
        return self._lnprob_of_log(np.log(vals))

    def get_prob_scalar(self, val):
        return np.exp(self.get_lnprob_scalar(val))

    def get_prob_array(self, vals):
        return np.exp(self.get_lnprob_array(vals))

    def get_fdiff(self, val):
        return _clipped_root(self.get_lnprob_scalar(val))

    def sample(self, nrand=None):
        r = np.exp(self.logmean + self.logsigma * self.rng.normal(size=nrand))
        if self.shift is not None:
            r += self.shift
        return r

    def sample_brute(self, nrand=None, maxval=None):
        """rejection under the curve on [shift, shift + maxval]"""
        rng = self.rng
        if maxval is None:
            maxval = self.mean + 10 * self.sigma
        n, scalar = _count(nrand)

        def propose(nleft):
            x = maxval * rng.rand(nleft)
            if self.shift is not None:
                x += self.shift
            h = rng.uniform(size=nleft)
            return x, h < self.get_prob_array(x)
        vals = _accept_reject(n, propose)
        return vals[0] if scalar else vals

    def _calc_fdiff(self, pars):
        try:
            trial = LogNormal(pars[0], pars[1], rng=self.rng)
            model = trial.get_prob_array(self._fitx) * pars[2]
        except (GMixRangeError, ValueError):
            return self._fity * 0 - np.inf
        return model - self._fity

    def fit(self, x, y):
        """least-squares (mean, sigma, amplitude) of a log-normal to y(x)"""
        from .fitting import run_leastsq
        self._fitx = x
        self._fity = y
        for _ in range(4):
            f1, f2, f3 = 1.0 + self.rng.uniform(low=0.1, high=0.1, size=3)
            guess = np.array([x.mean() * f1, x.std() * f2, y.mean() * f3])
            res = run_leastsq(self._calc_fdiff, guess, 0)
            if res["flags"] == 0:
                break
        return res


class Sinh(PriorBase):
    """fdiff = sinh((x - mean) / scale): nearly flat inside, steep outside
    (priors/priors.py:975-1043)"""

    def __init__(self, mean, scale, rng):
        super().__init__(rng=rng)
        self.mean = mean
        self.scale = scale

    def get_fdiff(self, val):
        return np.sinh((val - self.mean) / self.scale)

    def sample(self, nrand=None):
        n, scalar = _count(nrand)
        vals = self.rng.uniform(low=self.mean - self.scale, high=self.mean + self.scale, size=n)
        return vals[0] if scalar else vals


class TruncatedGaussian(PriorBase):
    """gaussian on [minval, maxval]; outside is a GMixRangeError for scalars
    and ln p = -inf in arrays (priors/priors.py:1046-1169)"""

    def __init__(self, mean, sigma, minval, maxval, rng):
        super().__init__(rng=rng)
        self.mean = mean
        self.sigma = sigma
        self.ivar = 1.0 / sigma ** 2
        self.sinv = 1.0 / sigma
        self.minval = minval
        self.maxval = maxval

    def _check(self, val):
        if val < self.minval or val > self.maxval:
            raise GMixRangeError("value out of range")

    def get_lnprob_scalar(self, val):
        self._check(val)
        diff = val - self.mean
        return -0.5 * diff * diff * self.ivar

    def get_lnprob_array(self, val):
        lnp = np.full(val.size, -np.inf)
        inside = (val > self.minval) & (val < self.maxval)
        diff = val[inside] - self.mean
        lnp[inside] = -0.5 * diff * diff * self.ivar
        return lnp

    def get_fdiff(self, val):
        self._check(val)
        return (val - self.mean) * self.sinv

    def sample(self, nrand=None):
        n, scalar = _count(nrand)
        rng = self.rng

        def propose(nleft):
            x = rng.normal(loc=self.mean, scale=self.sigma, size=nleft)
            return x, (x > self.minval) & (x < self.maxval)
        vals = _accept_reject(n, propose)
        return vals[0] if scalar else vals


# ---------------------------------------------------------------------------
# shapes

class GPriorBase(PriorBase):
    """
    A prior on the reduced shear (g1, g2) that depends on |g| only.  A
    subclass supplies the scalar densities and the fill_* array forms; this
    class samples |g| by rejection under p(|g|) (its maximum found once with
    scipy.optimize.minimize) and spreads the position angle uniformly
    (priors/shape.py:18-367).
    """

    def __init__(self, pars, rng):
        PriorBase.__init__(self, rng=rng)
        self.pars = np.array(pars, dtype="f8")
        self.gmax = 1.0

    def _abstract(self, *args, **kw):
        raise RuntimeError("over-ride me")

    fill_prob_array1d = _abstract
    fill_lnprob_array2d = _abstract
    fill_prob_array2d = _abstract
    get_lnprob_scalar2d = _abstract
    get_prob_scalar2d = _abstract
    get_prob_scalar1d = _abstract

    def get_lnprob_array2d(self, g1arr, g2arr):
        g1arr = np.asarray(g1arr, dtype="f8")
        g2arr = np.asarray(g2arr, dtype="f8")
        output = np.zeros(g1arr.size) + LOWVAL
        self.fill_lnprob_array2d(g1arr, g2arr, output)
        return output

    def get_prob_array2d(self, g1arr, g2arr):
        g1arr = np.asarray(g1arr, dtype="f8")
        g2arr = np.asarray(g2arr, dtype="f8")
        output = np.zeros(g1arr.size)
        self.fill_prob_array2d(g1arr, g2arr, output)
        return output

    def get_prob_array1d(self, garr):
        garr = np.asarray(garr, dtype="f8")
        output = np.zeros(garr.size)
        self.fill_prob_array1d(garr, output)
        return output

    def sample1d(self, nrand, maxguess=0.1):
        """|g| on [0, gmax - 1e-4) under p(|g|), against 1.1 x its maximum"""
        rng = self.rng
        if not hasattr(self, "maxval1d"):
            self.set_maxval1d(maxguess=maxguess)
        ceiling = self.maxval1d * 1.1
        gtop = self.gmax - 1.0e-4

        def propose(nleft):
            g = gtop * rng.uniform(size=nleft)
            h = ceiling * rng.uniform(size=nleft)
            return g, h < self.get_prob_array1d(g)
        return _accept_reject(nrand, propose)

    def sample2d(self, nrand=None, maxguess=0.1):
        n, scalar = _count(nrand)
        g = self.sample1d(n, maxguess=maxguess)
        twotheta = 2 * (self.rng.uniform(size=n) * 2 * np.pi)
        g1 = g * np.cos(twotheta)
        g2 = g * np.sin(twotheta)
        return (g1[0], g2[0]) if scalar else (g1, g2)

    def sample2d_brute(self, nrand):
        """rejection in the (g1, g2) square under p(0, 0)"""
        rng = self.rng
        top = self.get_prob_scalar2d(0.0, 0.0)

        def propose(nleft):
            g1 = srandu(nleft, rng=rng)
            g2 = srandu(nleft, rng=rng)
            h = top * rng.uniform(size=nleft)
            return (g1, g2), h < self.get_prob_array2d(g1, g2)
        return _accept_reject(nrand, propose)

    def set_maxval1d(self, maxguess=0.1):
        import scipy.optimize
        res = scipy.optimize.minimize(self.get_prob_scalar1d_neg, maxguess)
        if res["status"] != 0:
            raise RuntimeError("failed to find min, flags: %d" % res["status"])
        self.maxval1d = -res["fun"]
        self.maxval1d_loc = res["x"]

    def get_prob_scalar1d_neg(self, g, *args):
        return -self.get_prob_scalar1d(g)

    def fit(self, xdata, ydata, guess=None, show=False):
        """fit the 1-d density to a histogram (x, counts) by least squares"""
        import logging
        from .fitting import run_leastsq
        from .util import print_pars
        logger = logging.getLogger(__name__)
        keep = ydata > 0
        self.xdata = xdata[keep]
        self.ydata = ydata[keep]
        self.ierr = 1.0 / np.sqrt(self.ydata)
        if guess is None:
            guess = self._get_guess(self.ydata.sum())
        res = run_leastsq(self._calc_fdiff, guess, 0, maxfev=4000)
        self.fit_pars = res["pars"]
        self.fit_pars_cov = res["pars_cov"]
        self.fit_perr = res["pars_err"]
        print("flags:", res["flags"], "\nnfev:", res["nfev"])
        print_pars(res["pars"], front="pars: ", logger=logger)
        print_pars(res["pars_err"], front="perr: ", logger=logger)
        print("pars list:", "[" + ", ".join("%g" % p for p in res["pars"]) + "]")

    def _calc_fdiff(self, pars):
        self.set_pars(pars)
        return (self.get_prob_array1d(self.xdata) - self.ydata) * self.ierr

    dofit = fit


class GPriorGauss(GPriorBase):
    """round gaussian in (g1, g2) cut at |g| < gmax - 1e-4: sampling only
    (priors/shape.py:370-443)"""

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.sigma = float(self.pars)

    def sample1d(self, nrand=None, **kw):
        raise NotImplementedError("no 1d for gauss")

    def sample2d(self, nrand=None, **kw):
        n, scalar = _count(nrand)
        rng = self.rng
        gtop = self.gmax - 1.0e-4

        def propose(nleft):
            g1 = rng.normal(size=nleft, scale=self.sigma)
            g2 = rng.normal(size=nleft, scale=self.sigma)
            return (g1, g2), np.sqrt(g1 ** 2 + g2 ** 2) < gtop
        g1, g2 = _accept_reject(n, propose)
        return (g1[0], g2[0]) if scalar else (g1, g2)


class GPriorBA(GPriorBase):
    """Bernstein & Armstrong (2014): p(g1, g2) = A (1 - g^2)^2 exp(-g^2 / (2
    sigma^2)); g^2 >= 1 is a GMixRangeError for the scalar ln p
    (priors/shape.py:446-662)"""

    def __init__(self, sigma, rng, A=1.0):
        PriorBase.__init__(self, rng=rng)
        self.set_pars([A, sigma])
        self.gmax = 1.0

    def set_pars(self, pars):
        self.A = pars[0]
        self.set_sigma(pars[1])

    def set_sigma(self, sigma):
        self.sigma = sigma
        self.sig2 = self.sigma ** 2
        self.sig4 = self.sigma ** 4
        self.sig2inv = 1.0 / self.sig2
        self.sig4inv = 1.0 / self.sig4

    def sample1d(self, nrand, maxguess=None):
        if maxguess is None:
            # one deviate per call, whether or not the maximum is known yet
            maxguess = self.sigma + 0.0001 * srandu(rng=self.rng)
        return super().sample1d(nrand, maxguess=maxguess)

    # |g| enters only squared, so sqrt(-2 ln p) serves as a residual
    def get_fdiff(self, g1, g2):
        if isinstance(g1, np.ndarray):
            chi2 = -2 * self.get_lnprob_array2d(g1, g2)
            return np.sqrt(chi2.clip(min=0.0))
        return _clipped_root(self.get_lnprob_scalar2d(g1, g2))

    def get_lnprob_scalar2d(self, g1, g2):
        gsq = g1 * g1 + g2 * g2
        omgsq = 1.0 - gsq
        if omgsq <= 0.0:
            raise GMixRangeError("g^2 too big: %s" % gsq)
        return 2 * np.log(omgsq) - 0.5 * gsq * self.sig2inv

    def get_prob_scalar2d(self, g1, g2):
        gsq = g1 * g1 + g2 * g2
        omgsq = 1.0 - gsq
        p = 0.0
        if omgsq > 0.0:
            p = (omgsq * omgsq) * np.exp(-0.5 * gsq * self.sig2inv)
        return self.A * p

    def fill_prob_array2d(self, g1arr, g2arr, output):
        gsq = g1arr * g1arr + g2arr * g2arr
        omgsq = 1.0 - gsq
        ok = omgsq > 0.0
        output[ok] = self.A * (omgsq[ok] * omgsq[ok]) * np.exp(-0.5 * gsq[ok] * self.sig2inv)

    def fill_lnprob_array2d(self, g1arr, g2arr, output):
        gsq = g1arr * g1arr + g2arr * g2arr
        omgsq = 1.0 - gsq
        ok = omgsq > 0.0
        output[ok] = 2 * np.log(omgsq[ok]) - 0.5 * gsq[ok] * self.sig2inv

    def get_prob_scalar1d(self, g):
        gsq = g * g
        omgsq = 1.0 - gsq
        p = 0.0
        if omgsq > 0.0:
            p = (omgsq * omgsq) * np.exp(-0.5 * gsq * self.sig2inv)
            p *= 2 * np.pi * g
        return self.A * p

    def fill_prob_array1d(self, g, output):
        gsq = g * g
        omgsq = 1.0 - gsq
        ok = omgsq > 0.0
        vals = (omgsq[ok] * omgsq[ok]) * np.exp(-0.5 * gsq[ok] * self.sig2inv)
        output[ok] = vals * (self.A * 2 * np.pi * g[ok])

    def _get_guess(self, num, n=None):
        centre = [1.3 * num * (self.xdata[1] - self.xdata[0]), 0.16]
        count, scalar = _count(n)
        guess = np.zeros((count, 2))
        guess[:, 0] = centre[0] * (1.0 + 0.2 * srandu(count, rng=self.rng))
        guess[:, 1] = centre[1] * (1.0 + 0.2 * srandu(count, rng=self.rng))
        return guess[0, :] if scalar else guess


class ZDisk2D(PriorBase):
    """uniform inside a disk of the given radius, zero (or a GMixRangeError
    for ln p) outside (priors/shape.py:665-803)"""

    def __init__(self, radius, rng):
        super().__init__(rng=rng)
        self.radius = radius
        self.radius_sq = radius ** 2

    def get_lnprob_scalar1d(self, r):
        if r >= self.radius:
            raise GMixRangeError("position out of bounds")
        return 0.0

    def get_prob_scalar1d(self, r):
        return 0.0 if r >= self.radius else 1.0

    def get_lnprob_scalar2d(self, x, y):
        if x ** 2 + y ** 2 >= self.radius_sq:
            raise GMixRangeError("position out of bounds")
        return 0.0

    def get_prob_scalar2d(self, x, y):
        return 0.0 if x ** 2 + y ** 2 >= self.radius_sq else 1.0

    def get_prob_array2d(self, x, y):
        x = np.asarray(x, dtype="f8").reshape(-1)
        y = np.asarray(y, dtype="f8").reshape(-1)
        return np.where(x ** 2 + y ** 2 < self.radius_sq, 1.0, 0.0)

    def sample1d(self, nrand=None):
        n, scalar = _count(nrand)
        r = np.sqrt(self.radius_sq * self.rng.uniform(size=n))
        return r[0] if scalar else r

    def sample2d(self, nrand=None):
        n, scalar = _count(nrand)
        radius = self.sample1d(nrand=n)
        theta = 2.0 * np.pi * self.rng.uniform(size=n)
        x = radius * np.cos(theta)
        y = radius * np.sin(theta)
        return (x[0], y[0]) if scalar else (x, y)


# ---------------------------------------------------------------------------
# centres

class CenPrior(PriorBase):
    """independent gaussians on the two centre offsets
    (priors/multivariate.py:8-110)"""

    def __init__(self, cen1, cen2, sigma1, sigma2, rng):
        super().__init__(rng=rng)
        self.cen1 = float(cen1)
        self.cen2 = float(cen2)
        self.sigma1 = float(sigma1)
        self.sigma2 = float(sigma2)
        self.sinv1 = 1.0 / self.sigma1
        self.sinv2 = 1.0 / self.sigma2
        self.s2inv1 = 1.0 / self.sigma1 ** 2
        self.s2inv2 = 1.0 / self.sigma2 ** 2

    def get_fdiff(self, x1, x2):
        return (x1 - self.cen1) * self.sinv1, (x2 - self.cen2) * self.sinv2

    def get_lnprob_scalar_sep(self, x1, x2):
        d1 = self.cen1 - x1
        d2 = self.cen2 - x2
        return -0.5 * d1 * d1 * self.s2inv1, -0.5 * d2 * d2 * self.s2inv2

    def get_lnprob_scalar(self, x1, x2):
        d1 = self.cen1 - x1
        d2 = self.cen2 - x2
        return -0.5 * d1 * d1 * self.s2inv1 - 0.5 * d2 * d2 * self.s2inv2

    def get_prob_scalar(self, x1, x2):
        # math.exp: scalars only, as in the reference
        return math.exp(self.get_lnprob_scalar(x1, x2))

    get_prob_array = get_prob_scalar
    get_lnprob_array = get_lnprob_scalar

    def sample(self, nrand=None):
        rng = self.rng
        first = rng.normal(loc=self.cen1, scale=self.sigma1, size=nrand)
        second = rng.normal(loc=self.cen2, scale=self.sigma2, size=nrand)
        return first, second

    sample2d = sample


SimpleGauss2D = CenPrior


class KDE(object):
    """samples from a gaussian kernel density estimate of `data`
    (scipy.stats.gaussian_kde; priors/kde.py:4-67)"""

    def __init__(self, data, kde_factor, rng):
        import scipy.stats
        self.rng = rng
        self.is_1d = len(data.shape) == 1
        self.kde = scipy.stats.gaussian_kde(data.transpose(), bw_method=kde_factor)

    def sample(self, nrand=None):
        n, scalar = _count(nrand)
        r = self.kde.resample(size=n, seed=self.rng).transpose()
        if self.is_1d:
            r = r[:, 0]
        return r[0] if scalar else r
