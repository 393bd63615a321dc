"""
Separable joint priors over the parameter vector of an LM fit -- what
Fitter(prior=...) and the prior-drawing guessers take -- with the reference's
names, arguments and numbers (ngmix/joint_prior.py).

A joint prior here is a row of TERMS over the parameter vector

    [cen1, cen2 | g1, g2 | T, (1-d terms of the model) ..., F_band ...]

a centre prior (two parameters, two residual rows), a shape prior (two
parameters, one row) and one 1-d prior per remaining parameter.  The classes
differ only in which 1-d terms stand between T and the fluxes and in how a
residual row is made: PriorSimpleSep and PriorCoellipSame take sqrt(-2 ln p)
of every term, PriorBDSep / PriorBDFSep ask each term for its own get_fdiff.
Everything else -- ln p, sampling order, bounds for leastsqbound, widths --
is one engine over that row.

    fill_fdiff(pars, fdiff) -> number of rows written   (FitModel.calc_fdiff)
    get_lnprob_scalar(pars), get_lnprob_array(pars), get_prob_*
    sample(nrand=None)       draws in term order from each term's own rng
    bounds                   None, or one (lo, hi) per parameter

prior_batch.as_batch_prior turns a PriorSimpleSep of CenPrior / GPriorBA /
FlatPrior, TwoSidedErf or Normal terms into the form the lock-step driver
evaluates on the device for all fits at once.
"""
import numpy as np

from . import gmix as _gmix

__all__ = ["PriorSimpleSep", "PriorGalsimSimpleSep", "PriorBDSep", "PriorBDFSep",
           "PriorCoellipSame"]


class PriorSimpleSep(object):
    """
    [cen1, cen2, g1, g2, T, F_band...] (joint_prior.py:5-236)

    cen_prior: get_lnprob_scalar(x1, x2), get_lnprob_scalar_sep, sample
    g_prior: get_lnprob_scalar2d(g1, g2), sample2d
    T_prior, F_prior: 1-d priors; F_prior a LIST for several bands
    """

    # the 1-d terms between T and the fluxes, by attribute name
    _middle = ()
    # residual rows from ln p (True) or from each term's get_fdiff (False)
    _rows_from_lnprob = True
    # container types of F_prior that mean "one prior per band"
    _band_containers = (list,)

    def __init__(self, cen_prior, g_prior, T_prior, F_prior):
        self.cen_prior = cen_prior
        self.g_prior = g_prior
        self.T_prior = T_prior
        self._set_flux_priors(F_prior)
        self.set_bounds()

    def _set_flux_priors(self, F_prior):
        if isinstance(F_prior, self._band_containers):
            self.nband = len(F_prior)
        else:
            self.nband = 1
            F_prior = [F_prior]
        self.F_priors = F_prior

    def _scalar_terms(self):
        """the 1-d priors in parameter order, the first at parameter 4"""
        return [self.T_prior] + [getattr(self, n) for n in self._middle] + self.F_priors

    def set_bounds(self):
        """None when no 1-d term has bounds, else (None, None) for the centre
        and shape and each 1-d term's own (or (None, None))"""
        bounds = [(None, None)] * 4
        some = False
        for p in self._scalar_terms():
            if p.has_bounds():
                some = True
                bounds.append((p.bounds[0], p.bounds[1]))
            else:
                bounds.append((None, None))
        self.bounds = bounds if some else None

    def get_widths(self, nrand=10000):
        """rough one-sigma widths from samples (the shears set to 2), made
        once"""
        if not hasattr(self, "_sigma_estimates"):
            sigmas = self.sample(nrand).std(axis=0)
            sigmas[2] = 2.0
            sigmas[3] = 2.0
            self._sigma_estimates = sigmas
        return self._sigma_estimates

    def fill_fdiff(self, pars, fdiff):
        """the prior's residual rows at the head of fdiff; returns how many"""
        scalars = self._scalar_terms()
        if self._rows_from_lnprob:
            fdiff[0], fdiff[1] = self.cen_prior.get_lnprob_scalar_sep(pars[0], pars[1])
            fdiff[2] = self.g_prior.get_lnprob_scalar2d(pars[2], pars[3])
            for k, p in enumerate(scalars):
                fdiff[3 + k] = p.get_lnprob_scalar(pars[4 + k])
            nrows = 3 + len(scalars)
            chi2 = -2 * fdiff[0:nrows]
            chi2.clip(min=0.0, max=None, out=chi2)
            fdiff[0:nrows] = np.sqrt(chi2)
            return nrows
        fdiff[0], fdiff[1] = self.cen_prior.get_fdiff(pars[0], pars[1])
        fdiff[2] = self.g_prior.get_fdiff(pars[2], pars[3])
        for k, p in enumerate(scalars):
            fdiff[3 + k] = p.get_fdiff(pars[4 + k])
        return 3 + len(scalars)

    def get_lnprob_scalar(self, pars):
        lnp = self.cen_prior.get_lnprob_scalar(pars[0], pars[1])
        lnp += self.g_prior.get_lnprob_scalar2d(pars[2], pars[3])
        for k, p in enumerate(self._scalar_terms()):
            lnp += p.get_lnprob_scalar(pars[4 + k])
        return lnp

    def get_prob_scalar(self, pars):
        return np.exp(self.get_lnprob_scalar(pars))

    def get_lnprob_array(self, pars):
        lnp = self.cen_prior.get_lnprob_array(pars[:, 0], pars[:, 1])
        lnp += self.g_prior.get_lnprob_array2d(pars[:, 2], pars[:, 3])
        for k, p in enumerate(self._scalar_terms()):
            lnp += p.get_lnprob_array(pars[:, 4 + k])
        return lnp

    def get_prob_array(self, pars):
        return np.exp(self.get_lnprob_array(pars))

    def sample(self, nrand=None):
        """(nrand, npars) draws, or one vector for nrand=None: centre, shape,
        then the 1-d terms in parameter order"""
        scalar = nrand is None
        n = 1 if scalar else nrand
        scalars = self._scalar_terms()
        samples = np.zeros((n, 4 + len(scalars)))
        samples[:, 0], samples[:, 1] = self.cen_prior.sample(n)
        samples[:, 2], samples[:, 3] = self.g_prior.sample2d(n)
        for k, p in enumerate(scalars):
            samples[:, 4 + k] = p.sample(n)
        return samples[0, :] if scalar else samples

    def __repr__(self):
        terms = [self.cen_prior, self.g_prior] + self._scalar_terms()
        return "\n".join(str(t) for t in terms)


class PriorGalsimSimpleSep(PriorSimpleSep):
    """PriorSimpleSep with the size term named r50 (joint_prior.py:239-264)"""

    def __init__(self, cen_prior, g_prior, r50_prior, F_prior):
        super().__init__(cen_prior=cen_prior, g_prior=g_prior, T_prior=r50_prior,
                         F_prior=F_prior)


class PriorBDFSep(PriorSimpleSep):
    """[cen1, cen2, g1, g2, T, fracdev, F_band...]: bulge + disk with the size
    ratio fixed (joint_prior.py:484-674); rows are the terms' get_fdiff"""

    _middle = ("fracdev_prior",)
    _rows_from_lnprob = False
    _band_containers = (list, tuple)

    def __init__(self, cen_prior, g_prior, T_prior, fracdev_prior, F_prior):
        self.cen_prior = cen_prior
        self.g_prior = g_prior
        self.T_prior = T_prior
        self.fracdev_prior = fracdev_prior
        self._set_flux_priors(F_prior)
        self.set_bounds()


class PriorBDSep(PriorSimpleSep):
    """[cen1, cen2, g1, g2, T, logTratio, fracdev, F_band...]
    (joint_prior.py:267-481); rows are the terms' get_fdiff"""

    _middle = ("logTratio_prior", "fracdev_prior")
    _rows_from_lnprob = False
    _band_containers = (list, tuple)

    def __init__(self, cen_prior, g_prior, T_prior, logTratio_prior, fracdev_prior, F_prior):
        self.cen_prior = cen_prior
        self.g_prior = g_prior
        self.T_prior = T_prior
        self.logTratio_prior = logTratio_prior
        self.fracdev_prior = fracdev_prior
        self._set_flux_priors(F_prior)
        self.set_bounds()


class PriorCoellipSame(PriorSimpleSep):
    """
    [cen1, cen2, g1, g2, T_1..T_ngauss, F_1..F_ngauss]: the SAME T prior on
    every component's size and the same flux prior on every component's flux
    (joint_prior.py:874-1031); one band.
    """

    def __init__(self, ngauss, cen_prior, g_prior, T_prior, F_prior):
        self.ngauss = ngauss
        self.npars = _gmix.get_coellip_npars(ngauss)
        super().__init__(cen_prior, g_prior, T_prior, F_prior)
        if self.nband != 1:
            raise ValueError("coellip only supports one band")

    def _scalar_terms(self):
        return [self.T_prior] * self.ngauss + [self.F_priors[0]] * self.ngauss

    def set_bounds(self):
        # the reference appends, for each of the ngauss T terms and the flux
        # term, ngauss copies of a ONE-ELEMENT LIST holding the pair; kept as
        # is (it is what a caller of the reference sees in .bounds)
        bounds = [(None, None)] * 4
        some = False
        for p in [self.T_prior] * self.ngauss + self.F_priors:
            if p.has_bounds():
                some = True
                pair = [(p.bounds[0], p.bounds[1])]
            else:
                pair = [(None, None)]
            bounds += [pair] * self.ngauss
        self.bounds = bounds if some else None

    def __repr__(self):
        return "\n".join(str(t) for t in [self.cen_prior, self.g_prior, self.T_prior]
                         + self.F_priors)

    def _check_size(self, pars):
        if len(pars) != self.npars:
            raise ValueError('pars size %d expected %d' % (len(pars), self.npars))

    def get_lnprob_scalar(self, pars):
        self._check_size(pars)
        return super().get_lnprob_scalar(pars)

    def fill_fdiff(self, pars, fdiff):
        self._check_size(pars)
        return super().fill_fdiff(pars, fdiff)

    def get_lnprob_array(self, pars):
        # not specialised in the reference: the simple layout's columns
        lnp = self.cen_prior.get_lnprob_array(pars[:, 0], pars[:, 1])
        lnp += self.g_prior.get_lnprob_array2d(pars[:, 2], pars[:, 3])
        lnp += self.T_prior.get_lnprob_array(pars[:, 4])
        lnp += self.F_priors[0].get_lnprob_array(pars[:, 5])
        return lnp

    def sample(self, nrand=None):
        scalar = nrand is None
        n = 1 if scalar else nrand
        ng = self.ngauss
        samples = np.zeros((n, self.npars))
        samples[:, 0], samples[:, 1] = self.cen_prior.sample(n)
        samples[:, 2], samples[:, 3] = self.g_prior.sample2d(n)
        # the first size gets one draw more than the others, added in
        samples[:, 4] = self.T_prior.sample(n)
        for i in range(ng):
            samples[:, 4 + i] += self.T_prior.sample(n)
        for i in range(ng):
            samples[:, 4 + ng + i] = self.F_priors[0].sample(n)
        return samples[0, :] if scalar else samples
