"""
Gaussian-mixture model bookkeeping: model ids, parameter and gaussian counts
(reference: ngmix/gmix/gmix.py:1100-1193, 1245-1282).  The GMix host classes
live in ngmix_amd/gmix_classes.py.
"""

GMIX_FULL = 0
GMIX_GAUSS = 1
GMIX_TURB = 2
GMIX_EXP = 3
GMIX_DEV = 4
GMIX_BDC = 5
GMIX_BDF = 6
GMIX_COELLIP = 7
GMIX_CM = 9
GMIX_BD = 10

_NAMES = {
    GMIX_FULL: "full", GMIX_GAUSS: "gauss", GMIX_TURB: "turb", GMIX_EXP: "exp",
    GMIX_DEV: "dev", GMIX_BDC: "bdc", GMIX_BDF: "bdf", GMIX_COELLIP: "coellip",
    GMIX_CM: "cm", GMIX_BD: "bd",
}
_NUMS = {name: num for num, name in _NAMES.items()}

_NPARS = {
    GMIX_GAUSS: 6, GMIX_TURB: 6, GMIX_EXP: 6, GMIX_DEV: 6, GMIX_CM: 6,
    GMIX_BD: 8, GMIX_BDF: 7, GMIX_BDC: 8,
}

_NGAUSS = {
    GMIX_GAUSS: 1, "gauss": 1, GMIX_TURB: 3, "turb": 3, GMIX_EXP: 6, "exp": 6,
    GMIX_DEV: 10, "dev": 10, GMIX_CM: 16, GMIX_BD: 16, GMIX_BDF: 16,
    GMIX_BDC: 16,
    "em1": 1, "em2": 2, "em3": 3, "em4": 4, "em5": 5,
    "coellip1": 1, "coellip2": 2, "coellip3": 3, "coellip4": 4, "coellip5": 5,
}


def get_model_num(model):
    """numerical id for a model given by name or id"""
    if model in _NUMS:
        return _NUMS[model]
    if model in _NAMES:
        return model
    raise ValueError("unknown model: '%s'" % model)


def get_model_name(model):
    """string name for a model given by name or id"""
    if model in _NAMES:
        return _NAMES[model]
    if model in _NUMS:
        return model
    raise ValueError("unknown model: '%s'" % model)


def get_model_ngauss(model):
    """number of gaussians of a model"""
    if model not in _NGAUSS:
        raise ValueError("unknown model: '%s'" % model)
    return _NGAUSS[model]


def get_model_npars(model):
    """number of parameters of a model"""
    if model not in _NUMS and model not in _NAMES:
        raise ValueError("bad model: '%s'" % model)
    return _NPARS[get_model_num(model)]


def get_coellip_npars(ngauss):
    return 4 + 2 * ngauss


def get_coellip_ngauss(npars):
    return (npars - 4) // 2


# ---------------------------------------------------------------------------
# host classes (reference API: ngmix/gmix/gmix.py).  They own the gauss2d
# record array; O(ngauss) prep goes through the C ABI's host entry points and
# every pixel loop through its GPU kernels.
# ---------------------------------------------------------------------------
import ctypes as _ctypes  # noqa: E402

import numpy as _np  # noqa: E402

from . import _lib  # noqa: E402
from . import moments as _moments  # noqa: E402
from . import shape as _shape  # noqa: E402

_gauss2d_dtype = _lib.GAUSS2D_DTYPE
GMIX_LOW_DETVAL = 1.0e-200

# kernel flavour used by the host classes: False = fused (default),
# True = exact (per-pixel values bit-identical to the reference)
_EXACT = False


def set_exact_kernels(flag):
    """choose the exact (bit-identical, slower) or fused (default) pixel
    kernels for GMix / Observation level calls"""
    global _EXACT
    _EXACT = bool(flag)


def get_exact_kernels():
    return _EXACT


def make_gmix_model(pars, model):
    """a GMix (or subclass) for the named model"""
    num = get_model_num(model)
    if num == GMIX_COELLIP:
        return GMixCoellip(pars)
    if num == GMIX_FULL:
        return GMix(pars=pars)
    return GMixModel(pars, num)


def get_moments_result_dtype(with_higher_order=False):
    nmom = 17 if with_higher_order else 6
    return [
        ('flags', 'i4'), ('npix', 'i4'), ('wsum', 'f8'),
        ('sums', 'f8', nmom), ('sums_cov', 'f8', (nmom, nmom)),
        ('pars', 'f8', nmom), ('F', 'f8', nmom),
    ]


def pack_to_dict(res):
    loglike, s2n_numer, s2n_denom, npix = res
    return {"loglike": loglike, "s2n_numer": s2n_numer,
            "s2n_denom": s2n_denom, "npix": npix}


def get_weighted_moments_stats(ares):
    """sums record -> dict with e1, e2, T, s2n, ... added"""
    res = {}
    for n in ares.dtype.names:
        res[n] = ares[n].copy() if n in ("sums", "sums_cov") else ares[n]
    res.update(_moments.make_mom_result(res["sums"].copy(),
                                        res["sums_cov"].copy(), res["wsum"]))
    return res


def gmix_concat(gmixes):
    """one GMix holding the gaussians of all the inputs"""
    if len(gmixes) == 0:
        raise ValueError("send at least one gmix")
    pars = []
    for gm in gmixes:
        pars += list(gm.get_full_pars())
    return GMix(pars=pars)


class GMix(object):
    """
    A two-dimensional gaussian mixture: send ngauss= (zeroed) or pars=
    [p1,row1,col1,irr1,irc1,icc1, p2,...].
    """

    def __init__(self, ngauss=None, pars=None):
        self._model, self._model_name = GMIX_FULL, "full"
        if pars is None:
            if ngauss is None:
                raise ValueError("send ngauss= or pars=")
            self._allocate(ngauss, 6 * ngauss)
            return
        count, leftover = divmod(len(pars), 6)
        if leftover:
            raise ValueError("len(pars) must be mutiple of 6 got %s" % len(pars))
        self._allocate(count, len(pars))
        self._fill(pars)

    def _allocate(self, ngauss, npars):
        """size the parameter vector and the gauss2d record array (zeroed)"""
        self._ngauss, self._npars = ngauss, npars
        self.reset()

    # ---- storage
    def reset(self):
        self._pars = _np.zeros(self._npars)
        self._data = _np.zeros(self._ngauss, dtype=_gauss2d_dtype)

    def get_data(self):
        return self._data

    def __len__(self):
        return self._ngauss

    def get_full_pars(self):
        gm = self._data
        pars = _np.zeros(6 * self._ngauss)
        for k, name in enumerate(("p", "row", "col", "irr", "irc", "icc")):
            pars[k::6] = gm[name]
        return pars

    # ---- fills
    def fill(self, pars):
        if len(pars) != self._npars:
            raise ValueError("model '%s' requires %s pars, got %s" % (
                self._model_name, self._npars, len(pars)))
        self._fill(pars)

    def _fill(self, pars):
        self._pars[:] = pars
        st = _lib.lib().ngmix_fill_model(
            _lib.ptr(self._data), self._ngauss, int(self._model),
            _lib.ptr(self._pars), self._npars)
        _lib.check(st, "ngmix_fill_model")

    # ---- summary quantities (host numpy, as in the reference)
    def get_cen(self):
        gm = self._data
        psum = gm["p"].sum()
        return (gm["row"] * gm["p"]).sum() / psum, (gm["col"] * gm["p"]).sum() / psum

    def set_cen(self, row, col):
        row0, col0 = self.get_cen()
        self._data["row"] += row - row0
        self._data["col"] += col - col0

    def _second_moments(self):
        """flux-weighted covariance of the whole mixture about its centroid: the
        mean of the components' own covariances plus the scatter of their
        centres.  Each moment is a numpy sum of (second moment) * p times
        1 / sum(p), as gmix.py:181-266 writes it, so the getters agree with the
        reference to the bit."""
        gm = self._data
        row0, col0 = self.get_cen()
        dr = gm["row"] - row0
        dc = gm["col"] - col0
        p = gm["p"]
        ipsum = 1.0 / p.sum()
        return (((gm["irr"] + dr ** 2) * p).sum() * ipsum,
                ((gm["irc"] + dr * dc) * p).sum() * ipsum,
                ((gm["icc"] + dc ** 2) * p).sum() * ipsum)

    def get_T(self):
        irr, _, icc = self._second_moments()
        return irr + icc

    def get_sigma(self):
        return _np.sqrt(self.get_T() / 2.0)

    def get_e1e2T(self):
        irr, irc, icc = self._second_moments()
        T = irr + icc
        return (icc - irr) / T, 2.0 * irc / T, T

    def get_g1g2T(self):
        e1, e2, T = self.get_e1e2T()
        g1, g2 = _shape.e1e2_to_g1g2(e1, e2)
        return g1, g2, T

    def get_e1e2sigma(self):
        e1, e2, T = self.get_e1e2T()
        return e1, e2, _np.sqrt(T / 2)

    def get_g1g2sigma(self):
        g1, g2, T = self.get_g1g2T()
        return g1, g2, _np.sqrt(T / 2)

    def get_flux(self):
        return self._data["p"].sum()

    get_psum = get_flux

    def set_flux(self, psum):
        gm = self._data
        gm["p"] *= psum / gm["p"].sum()
        gm["norm_set"] = 0

    set_psum = set_flux

    def scale_T(self, scale):
        if scale < 0.0:
            raise ValueError(f"Requested scale {scale} < 0")
        gm = self._data.copy()
        row0, col0 = self.get_cen()
        root = _np.sqrt(scale)
        gm["row"] = (gm["row"] - row0) * root + row0
        gm["col"] = (gm["col"] - col0) * root + col0
        for n in ("irr", "irc", "icc"):
            gm[n] *= scale
        gm["norm_set"] = 0
        self._data = gm

    def get_gaussap_flux(self, fwhm=None, sigma=None, T=None):
        """flux inside a round gaussian aperture"""
        if fwhm is not None:
            sigma = _moments.fwhm_to_sigma(fwhm)
        elif T is not None:
            sigma = _np.sqrt(T / 2.0)
        elif sigma is not None:
            sigma = float(sigma)
        else:
            raise ValueError("send weight function sigma, fwhm, or T")
        wt_inv = _np.eye(2) / sigma ** 2
        apflux = 0.0
        for g in self._data:
            fac = 1.0
            if g["det"] > GMIX_LOW_DETVAL:
                mat = _np.array([[g["irr"], g["irc"]], [g["irc"], g["icc"]]])
                try:
                    newmat = _np.linalg.inv(_np.linalg.inv(mat) + wt_inv)
                    fac = min(_np.sqrt(_np.linalg.det(newmat) / g["det"]), 1)
                except _np.linalg.LinAlgError:
                    pass
            apflux += g["p"] * fac
        return apflux

    # ---- norms
    def set_norms(self):
        st = _lib.lib().ngmix_set_norms(_lib.ptr(self._data), self._ngauss)
        _lib.check(st, "ngmix_set_norms")

    def set_norms_if_needed(self):
        if self._data["norm_set"][0] == 0:
            self.set_norms()

    # ---- copies and transforms
    def copy(self):
        gmix = GMix(ngauss=self._ngauss)
        gmix._data[:] = self._data[:]
        return gmix

    def __copy__(self):
        return self.copy()

    def __deepcopy__(self, memo):
        result = self.copy()
        memo[id(self)] = result
        return result

    def __eq__(self, gm):
        if not isinstance(gm, GMix):
            raise ValueError(f"expected GMix, got {type(gm)}")
        a, b = self._data, gm.get_data()
        return all(_np.all(a[n] == b[n])
                   for n in ("p", "row", "col", "irr", "irc", "icc", "det"))

    def get_sheared(self, s1, s2=None):
        if isinstance(s1, _shape.Shape):
            shear1, shear2 = s1.g1, s1.g2
        elif s2 is not None:
            shear1, shear2 = s1, s2
        else:
            raise ValueError("send a Shape or s1,s2")
        new_gmix = self.copy()
        nd = new_gmix.get_data()
        nd["norm_set"] = 0
        for i in range(len(self)):
            irr, irc, icc = _moments.get_sheared_moments(
                nd["irr"][i], nd["irc"][i], nd["icc"][i], shear1, shear2)
            nd["irr"][i] = irr
            nd["irc"][i] = irc
            nd["icc"][i] = icc
            nd["det"][i] = irr * icc - irc * irc
        return new_gmix

    def convolve(self, psf):
        if not isinstance(psf, GMix):
            raise TypeError("Can only convolve with another GMix  got type %s"
                            % type(psf))
        output = GMix(ngauss=len(self) * len(psf))
        st = _lib.lib().ngmix_convolve_fill(
            _lib.ptr(output._data), _lib.ptr(self._data), len(self),
            _lib.ptr(psf._data), len(psf))
        _lib.check(st, "ngmix_convolve_fill")
        return output

    def make_round(self, preserve_size=False):
        gm = self.copy()
        if preserve_size:
            e1, e2, T = gm.get_e1e2T()
            irr, irc, icc = _moments.e2mom(e1, e2, T)
            eigs = _np.linalg.eigvals(_np.array([[irr, irc], [irc, icc]]))
            factor = eigs.max() / (T / 2.0)
        else:
            g1, g2, T = gm.get_g1g2T()
            factor = _shape.get_round_factor(g1, g2)
        gd = gm.get_data()
        gd["norm_set"] = 0
        for i in range(len(gm)):
            Ti = gd["irr"][i] + gd["icc"][i]
            gd["irc"][i] = 0.0
            gd["irr"][i] = 0.5 * Ti * factor
            gd["icc"][i] = 0.5 * Ti * factor
        return gm

    # ---- pixel operations (GPU)
    def make_image(self, dims, jacobian=None, fast_exp=False):
        dims = _np.array(dims, ndmin=1, dtype="i8")
        if dims.size != 2:
            raise ValueError("images must have two dimensions, got %s" % str(dims))
        image = _np.zeros(dims, dtype="f8")
        # (a fresh image: the overwriting form of the render kernel -- no upload
        # of the zeros, no read of the buffer; bit-identical to zeros + add)
        self._fill_image(image, jacobian=jacobian, fast_exp=fast_exp, fresh=True)
        return image

    def _fill_image(self, image, jacobian=None, fast_exp=False, fresh=False):
        """ADD the rendered mixture into image (render_nb.py:9-36)"""
        from .jacobian import Jacobian, UnitJacobian
        from .batch import render_single
        if jacobian is None:
            cen = (_np.array(image.shape) - 1.0) / 2.0
            jacobian = UnitJacobian(row=cen[0], col=cen[1])
        else:
            assert isinstance(jacobian, Jacobian)
        self.set_norms_if_needed()
        render_single(self._data, image, jacobian._data, fast_exp, exact=_EXACT, fresh=fresh)

    def get_loglike(self, obs, more=False):
        self.set_norms_if_needed()
        res = obs._device_stamp().loglike_single(self._data, exact=_EXACT)
        return pack_to_dict(res) if more else res[0]

    def fill_fdiff(self, obs, fdiff, start=0):
        nuse = fdiff.size - start
        if nuse < obs.image.size:
            raise ValueError("fdiff from start must have len >= %d, got %d"
                             % (obs.image.size, nuse))
        self.set_norms_if_needed()
        obs._device_stamp().fdiff_single(self._data, fdiff, start, exact=_EXACT)

    def get_model_s2n_sum(self, obs):
        self.set_norms_if_needed()
        return obs._device_stamp().s2n_single(self._data, exact=_EXACT)

    def get_model_s2n(self, obs):
        return _np.sqrt(self.get_model_s2n_sum(obs))

    def get_weighted_sums(self, obs, maxrad=None, with_higher_order=False,
                          res=None):
        self.set_norms_if_needed()
        if maxrad is None:
            maxrad = 100 * _np.sqrt(self.get_T() / 2)
        if res is None:
            dt = _np.dtype(get_moments_result_dtype(with_higher_order), align=True)
            res = _np.zeros(1, dtype=dt)[0]
        nmom = 17 if with_higher_order else 6
        obs._device_stamp().wsums_single(self._data, res, nmom, float(maxrad))
        return res

    def get_weighted_moments(self, obs, maxrad=None, with_higher_order=False):
        res = self.get_weighted_sums(obs, maxrad=maxrad,
                                     with_higher_order=with_higher_order)
        return get_weighted_moments_stats(res)

    def __repr__(self):
        fmt = "p: %.4g row: %.4g col: %.4g irr: %.4g irc: %.4g icc: %.4g"
        return "\n".join(fmt % (t["p"], t["row"], t["col"], t["irr"], t["irc"],
                                t["icc"]) for t in self._data)


class GMixModel(GMix):
    """mixture built from model parameters, e.g. GMixModel(pars, 'exp')"""

    _KNOWN = frozenset(("gauss", "turb", "exp", "dev", "bd", "bdf", "cm", "coellip", "full"))

    def __init__(self, pars, model):
        num = get_model_num(model)
        name = get_model_name(num)
        if name not in self._KNOWN:
            raise ValueError("bad model: '%s'" % name)
        self._model, self._model_name = num, name
        self._allocate(get_model_ngauss(num), get_model_npars(num))
        self.fill(pars)

    def copy(self):
        return GMixModel(self._pars, self._model_name)

    def set_cen(self, row, col):
        super().set_cen(row, col)
        self._pars[0] = row
        self._pars[1] = col


class GMixCM(GMixModel):
    """composite exp+dev model with fixed fracdev and TdByTe"""

    def __init__(self, fracdev, TdByTe, pars):
        self._fracdev = fracdev
        self._TdByTe = TdByTe
        tf = _ctypes.c_double()
        st = _lib.lib().ngmix_get_cm_Tfactor(float(fracdev), float(TdByTe),
                                             _ctypes.byref(tf))
        _lib.check(st, "ngmix_get_cm_Tfactor")
        self._Tfactor = tf.value
        super().__init__(pars, "cm")

    def copy(self):
        return GMixCM(self._fracdev, self._TdByTe, self._pars)

    def _fill(self, pars):
        self._pars[:] = pars
        st = _lib.lib().ngmix_fill_cm(_lib.ptr(self._data), float(self._fracdev),
                                      float(self._TdByTe), float(self._Tfactor),
                                      _lib.ptr(self._pars))
        _lib.check(st, "ngmix_fill_cm")

    def __repr__(self):
        return "\n".join(["fracdev: %g" % self._fracdev,
                          "TdByTe:  %g" % self._TdByTe, super().__repr__()])


class GMixCoellip(GMixModel):
    """co-centric, co-elliptical gaussians: [cen1,cen2,g1,g2,T1..,F1..]"""

    def __init__(self, pars):
        # four shared parameters, then one (T, F) pair per gaussian
        pairs, odd = divmod(len(pars) - 4, 2)
        if odd:
            raise ValueError("coellip must have len(pars)==4+2*ngauss, got %s" % len(pars))
        self._model, self._model_name = GMIX_COELLIP, "coellip"
        self._allocate(pairs, len(pars))
        self._fill(pars)

    def copy(self):
        return GMixCoellip(self._pars)


class _TypedList(list):
    """a list that only takes items of one type (append and item assignment)"""
    _item_type = object
    _what = "item"

    def _check(self, item):
        assert isinstance(item, self._item_type), "%s should be of type %s" % (
            self._what, self._item_type.__name__)

    def append(self, item):
        self._check(item)
        super().append(item)

    def __setitem__(self, index, item):
        self._check(item)
        super().__setitem__(index, item)


class GMixList(_TypedList):
    """the mixtures of one band's observations, in order (reference:
    ngmix/gmix/gmix_lists.py:6-27)"""
    _item_type = GMix
    _what = "gmix"

    def append(self, gmix):
        super().append(gmix)


class MultiBandGMixList(_TypedList):
    """one GMixList per band (reference: ngmix/gmix/gmix_lists.py:30-57)"""
    _item_type = GMixList
    _what = "gmix_list"

    def append(self, gmix_list):
        super().append(gmix_list)
