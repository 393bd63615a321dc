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

__all__ = ["GaussMom", "GaussMomBatch", "MomBatchResult"]


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

    def go_many(self, obs):
        """the moments of a sequence of Observations by ONE launch (the loop
        over a catalogue around go()): a MomBatchResult -- arrays by key, go()'s
        dict for observation i by position"""
        from .batch import StampBatch
        for o in obs:
            if not isinstance(o, Observation):
                raise ValueError("input obs must be an Observation")
        return GaussMomBatch(self.fwhm, with_higher_order=self.with_higher_order).go(
            StampBatch.from_observations(list(obs)))


class MomBatchResult(object):
    """
    The weighted moments of N stamps.  By key, arrays over the stamps --
    res["T"] is (N,), res["e"] (N, 2), res["sums_cov"] (N, nm, nm): every key of
    GaussMom.go's dict but the flag strings (moments.make_mom_result_batch, one
    vectorised pass); by position, res[i] is exactly the dict GaussMom.go returns
    for stamp i (built on request from the kernel's records, which come down
    from the device when the first one is asked for).  len() and iteration are
    over the stamps.
    """

    def __init__(self, arrays, records, area, n=None):
        self._arrays = arrays
        self._records = records         # a record array, or a callable that fetches it
        self._area = area
        self._n = n if n is not None else records.size

    def __len__(self):
        return self._n

    def keys(self):
        return self._arrays.keys()

    def __contains__(self, key):
        return key in self._arrays

    def __getitem__(self, key):
        if isinstance(key, str):
            return self._arrays[key]
        i = int(key)
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(key)
        if callable(self._records):
            self._records = self._records()
        if self._records is None:
            raise RuntimeError("the per-stamp records were released (MomBatchResult.release)")
        r = get_weighted_moments_stats(self._records[i])
        if r["flags"] == 0:
            _remove_area(r, self._area[i])
        return r

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def release(self, fetch=False):
        """Let go of the kernel's records on the DEVICE (2.7 kB per stamp at 17
        moments: a result kept around pins them until the first per-stamp dict
        is asked for).  fetch=True downloads them first, so that res[i] keeps
        working; fetch=False drops them: the arrays by key stay, res[i] raises."""
        if callable(self._records):
            self._records = self._records() if fetch else None

    @property
    def holds_device_memory(self):
        return callable(self._records)


class GaussMomBatch(object):
    """
    GaussMom over a StampBatch: one weighted-sums launch for all stamps, the
    statistics of make_mom_result for all of them ON THE DEVICE (moments.
    make_mom_result_batch on the records where the kernel left them) and ONE
    download of every array.  go() returns a MomBatchResult: arrays by key,
    GaussMom.go's dict by stamp index.
    """

    def __init__(self, fwhm, with_higher_order=False):
        self.fwhm = fwhm
        self.with_higher_order = with_higher_order
        self.weight = _make_weight(fwhm)

    def go(self, stamps):
        from .batch import GMixBatch
        from . import _lib
        import torch
        n = stamps.n
        nmom = 17 if self.with_higher_order else 6
        wt = np.tile(self.weight.get_data(), (n, 1))
        wtb = GMixBatch.from_numpy(wt, device=stamps.device)
        wtb.set_norms()
        maxrad = np.full(n, 100.0 * np.sqrt(self.weight.get_T() / 2.0))
        res, status = stamps.weighted_sums(wtb, maxrad, nmom=nmom)
        dtype = _lib.moments_result_dtype(nmom)
        off = {k: v[1] // 8 for k, v in dtype.fields.items()}
        d_sums = res[:, off["sums"]:off["sums"] + nmom]
        d_cov = res[:, off["sums_cov"]:off["sums_cov"] + nmom * nmom].reshape(n, nmom, nmom)
        d_wsum = res[:, off["wsum"]]
        d_npix = res.view(torch.int32)[:, 1]
        d_area = stamps.jac[:, 6].abs()
        arrays = moments.make_mom_result_batch(d_sums, d_cov, d_wsum)
        arrays["npix"] = d_npix
        arrays["wsum"] = d_wsum
        # the record's own flags (the kernel's) are 0 in the reference's record too;
        # the pixel area out of the flux-like quantities of the successful ones
        # (gaussmom.py:62-74)
        good = arrays["flags"] == 0
        fac = torch.where(good, 1.0 / d_area, torch.ones_like(d_area))
        for key in ("flux", "flux_err", "sums_norm", "wsum"):
            arrays[key] = arrays[key] * fac
        pars = arrays["pars"].clone()
        pars[:, 5] *= fac
        arrays["pars"] = pars
        arrays["sums"] = arrays["sums"] * fac[:, None]
        arrays["sums_err"] = arrays["sums_err"] * fac[:, None]
        arrays["sums_cov"] = arrays["sums_cov"] * (fac * fac)[:, None, None]
        arrays["_area"] = d_area
        # every array a contiguous segment of ONE flat float64 buffer, in its own
        # (N, ...) layout: one download through pinned memory (a pageable copy
        # of this size takes ten times the kernel), the host arrays views of it
        ints = ("flags", "flux_flags", "T_flags", "npix")
        layout, total = [], 0
        for key, t in arrays.items():
            layout.append((key, tuple(t.shape[1:]), total, t.numel()))
            total += t.numel()
        packed = torch.empty(total, dtype=torch.float64, device=res.device)
        for (key, shape, o, m), t in zip(layout, arrays.values()):
            packed[o:o + m] = t.reshape(-1)
        staged = torch.empty(total, dtype=torch.float64, pin_memory=True)
        staged.copy_(packed)
        host = staged.numpy()
        out = {}
        for key, shape, o, m in layout:
            a = host[o:o + m].reshape((n,) + shape)
            # (npix keeps the record's int32, the flag columns are int64 as the
            # per-object results make them)
            out[key] = (a.astype(np.int32) if key == "npix" else
                        a.astype(np.int64) if key in ints else a)
        area = out.pop("_area")

        def records():
            staged_r = torch.empty(res.shape, dtype=res.dtype, pin_memory=True)
            staged_r.copy_(res)
            return staged_r.numpy().reshape(-1).view(dtype)
        return MomBatchResult(out, records, area, n=n)
