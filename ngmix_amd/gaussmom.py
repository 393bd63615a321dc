"""
Gaussian weighted moments (reference: ngmix/gaussmom.py:7-94): a fixed round
gaussian weight at the jacobian centre, moments = GMix.get_weighted_moments
(the weighted-sums HIP kernel, gmix_nb.py:681-821) with the pixel area taken
out so fluxes are in image units.  GaussMomBatch measures N stamps in one
kernel launch.
"""
import logging

import numpy as np

from . import moments
from .gmix import GMixModel, get_weighted_moments_stats
from .observation import Observation

logger = logging.getLogger(__name__)

__all__ = ["GaussMom", "GaussMomBatch"]


def _make_weight(fwhm):
    T = moments.fwhm_to_T(fwhm)
    # centred on the coordinate origin defined by the jacobian; peak 1.0 so
    # that the weighted flux is well scaled (gaussmom.py:76-94)
    weight = GMixModel([0.0, 0.0, 0.0, 0.0, T, 1.0], "gauss")
    weight.set_norms()
    norm = weight.get_data()["norm"][0]
    weight.set_flux(1.0 / norm)
    return weight


def _remove_area(res, area):
    fac = 1.0 / area
    res["flux"] *= fac
    res["flux_err"] *= fac
    res["pars"][5] *= fac
    res["sums"] *= fac
    res["sums_cov"] *= fac ** 2
    res["sums_norm"] *= fac
    res["wsum"] *= fac
    res["sums_err"] *= fac
    return res


class GaussMom(object):
    """
    measure gaussian weighted moments

    fwhm: FWHM of the gaussian weight function
    with_higher_order: also return the 17-moment sums
    """
    kind = "wmom"

    def __init__(self, fwhm, with_higher_order=False):
        self.fwhm = fwhm
        self.with_higher_order = with_higher_order
        self.weight = _make_weight(fwhm)

    def go(self, obs):
        if not isinstance(obs, Observation):
            raise ValueError("input obs must be an Observation")
        res = self.weight.get_weighted_moments(
            obs=obs, with_higher_order=self.with_higher_order)
        if res["flags"] != 0:
            logger.debug("        moments failed: %s" % res["flagstr"])
            return res
        return _remove_area(res, obs.jacobian.area)


class GaussMomBatch(object):
    """
    GaussMom over a StampBatch: one weighted-sums launch for all stamps, the
    per-stamp statistics (make_mom_result, O(1) scalar work) on the host.
    go() returns a list of the same result dicts GaussMom.go returns.
    """

    def __init__(self, fwhm, with_higher_order=False):
        self.fwhm = fwhm
        self.with_higher_order = with_higher_order
        self.weight = _make_weight(fwhm)

    def go(self, stamps):
        from .batch import GMixBatch, records_to_numpy
        from . import _lib
        n = stamps.n
        nmom = 17 if self.with_higher_order else 6
        wt = np.tile(self.weight.get_data(), (n, 1))
        wtb = GMixBatch.from_numpy(wt, device=stamps.device)
        wtb.set_norms()
        maxrad = np.full(n, 100.0 * np.sqrt(self.weight.get_T() / 2.0))
        res, status = stamps.weighted_sums(wtb, maxrad, nmom=nmom)
        rec = records_to_numpy(res, _lib.moments_result_dtype(nmom))
        area = stamps.jac[:, 6].abs().cpu().numpy()
        out = []
        for i in range(n):
            r = get_weighted_moments_stats(rec[i])
            if r["flags"] == 0:
                _remove_area(r, area[i])
            out.append(r)
        return out
