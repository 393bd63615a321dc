"""
A joint prior with closed-form rows, for pinning the HOST path of
FitModel.calc_fdiff / calc_jacobian / calc_lnprob with prior rows (SURVEY.md
8a h3) without restating the reference's out-of-scope priors package.  The
same file defines the prior for the fixture generator (oracle/gen_golden_r2.py,
which hands it to the REFERENCE's FitModel) and for the test (which hands it to
ngmix_amd's), so both sides see bit-identical prior rows and any difference is
the FitModel's: residual layout [prior rows | pixels], the reserved-but-unused
row, the one-sided difference jacobian of the prior rows, lnprob = ln p +
loglike.

Interface = what FitModel calls on a prior (ngmix/fitting/results.py:389-396,
480-483, 586-625): fill_fdiff(pars, fdiff) -> number of rows,
get_lnprob_scalar(pars), optional .bounds.
"""
import numpy as np


class RowsPrior(object):
    """rows: cen1, cen2 (gaussian), |g| (gaussian in the shear magnitude,
    GMixRangeError for |g| >= 1), T (log-normal), one flux row per band
    (gaussian): 4 + nband rows, like PriorSimpleSep"""

    def __init__(self, nband, range_error, cen_sigma=0.05, g_sigma=0.2,
                 T_mean=0.5, T_sigma=0.4, F_mean=90.0, F_sigma=35.0):
        self.nband = nband
        self.range_error = range_error
        self.c, self.g, self.Tm, self.Ts = cen_sigma, g_sigma, T_mean, T_sigma
        self.Fm, self.Fs = F_mean, F_sigma

    def fill_fdiff(self, pars, fdiff):
        g = np.sqrt(pars[2] * pars[2] + pars[3] * pars[3])
        if g >= 1.0:
            raise self.range_error("g too big")
        if pars[4] <= 0.0:
            raise self.range_error("T <= 0")
        fdiff[0] = pars[0] / self.c
        fdiff[1] = pars[1] / self.c
        fdiff[2] = g / self.g
        fdiff[3] = (np.log(pars[4]) - np.log(self.Tm)) / self.Ts
        for b in range(self.nband):
            fdiff[4 + b] = (pars[5 + b] - self.Fm) / self.Fs
        return 4 + self.nband

    def get_lnprob_scalar(self, pars):
        f = np.zeros(4 + self.nband)
        self.fill_fdiff(pars, f)
        return -0.5 * float((f * f).sum())
