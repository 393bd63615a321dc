"""
Observation containers (reference API: ngmix/observation.py:114-1139).

An Observation owns the host image / weight / jacobian exactly as the
reference does; what changes underneath is that the pixel loops read a compact
device-resident copy (ngmix_amd.batch.SingleStamp), created lazily and dropped
whenever image, weight or jacobian change -- the same events that rebuild the
reference's pixel array (observation.py:814-830, 858-860).  `obs.pixels`
still returns the reference's AoS record array, built on the GPU by the
fill_pixels kernel on first access.
"""
import copy

import numpy as np

from .gexceptions import GMixFatalError
from .gmix import GMix
from .jacobian import Jacobian, UnitJacobian
from .pixels import make_pixels

__all__ = ["Observation", "ObsList", "MultiBandObsList", "get_mb_obs"]


class MetadataMixin(object):
    @property
    def meta(self):
        return self._meta

    @meta.setter
    def meta(self, meta):
        self.set_meta(meta)

    def set_meta(self, meta):
        if meta is None:
            meta = {}
        if not isinstance(meta, dict):
            raise TypeError("meta data must be in dictionary form, got %s"
                            % type(meta))
        self._meta = meta

    def update_meta_data(self, meta):
        if not isinstance(meta, dict):
            raise TypeError("meta data must be in dictionary form, got %s"
                            % type(meta))
        self._meta.update(meta)


def _simple_s2n(Isum, Vsum):
    return Isum / np.sqrt(Vsum) if Vsum > 0.0 else -9999.0


class Observation(MetadataMixin):
    """
    image plus optional weight (default ones), bmask, ormask, noise, jacobian
    (default unit jacobian at the canonical centre), gmix, psf (an
    Observation), meta, mfrac.  Arrays are exposed read-only; modify them
    inside `with obs.writeable():`, which refreshes the pixel data on exit.
    """

    def __init__(self, image, weight=None, bmask=None, ormask=None, noise=None,
                 jacobian=None, gmix=None, psf=None, meta=None, mfrac=None,
                 store_pixels=True, ignore_zero_weight=True):
        self._writeable = False
        self._ignore_zero_weight = ignore_zero_weight
        self._store_pixels = store_pixels
        self._pixels = self._stamp = self._stamp_batch = None
        # image, weight and jacobian define the pixel list: set all three, then
        # derive it once; the rest are plain attributes
        for setter, value in ((self.set_image, image), (self.set_weight, weight),
                              (self.set_jacobian, jacobian)):
            setter(value, update_pixels=False)
        self.update_pixels()
        for setter, value in ((self.set_meta, meta), (self.set_bmask, bmask),
                              (self.set_ormask, ormask), (self.set_noise, noise),
                              (self.set_gmix, gmix), (self.set_psf, psf),
                              (self.set_mfrac, mfrac)):
            setter(value)

    # ---- views
    def _get_view(self, data):
        view = data.view()
        view.flags["WRITEABLE"] = self._writeable
        return view

    def writeable(self):
        return self

    def __enter__(self):
        self._writeable = True
        return self

    def __exit__(self, exception_type, exception_value, traceback):
        self._writeable = False
        self.update_pixels()

    # ---- pixel data
    def update_pixels(self):
        """invalidate the derived pixel data (AoS list and device copy)"""
        self._pixels = None
        self._stamp = None
        self._stamp_batch = None
        if not self._store_pixels:
            return
        if self._ignore_zero_weight and not np.any(self._weight > 0.0):
            raise GMixFatalError("no weights > 0")

    @property
    def pixels(self):
        """the reference's pixel record array (always read-only); None when
        store_pixels is False"""
        if not self._store_pixels:
            return None
        if self._pixels is None:
            pixels = make_pixels(self._image, self._weight, self._jacobian,
                                 ignore_zero_weight=self._ignore_zero_weight)
            pixels.flags["WRITEABLE"] = False
            self._pixels = pixels
        return self._pixels


    def _device_batch(self):
        """this observation as a one-stamp StampBatch, made once (what the
        fitters read: a second fit of the same observation -- another model,
        another guess -- uploads nothing); update_pixels() drops it"""
        if self._stamp_batch is None:
            from .batch import StampBatch
            self._stamp_batch = StampBatch.from_observations([self])
        return self._stamp_batch

    def __getstate__(self):
        """pickling (worker pools, checkpoints) carries the host arrays only:
        the device-resident copies are made again where they are needed"""
        state = self.__dict__.copy()
        state["_stamp"] = None
        state["_stamp_batch"] = None
        return state

    def _device_stamp(self):
        """the compact device-resident copy the kernels read"""
        if self._stamp is None:
            from .batch import SingleStamp
            self._stamp = SingleStamp(self._image, self._weight,
                                      self._jacobian._data,
                                      self._ignore_zero_weight)
        return self._stamp

    @property
    def store_pixels(self):
        return self._store_pixels

    @store_pixels.setter
    def store_pixels(self, store_pixels):
        changed = store_pixels != self._store_pixels
        self._store_pixels = store_pixels
        if changed:
            self.update_pixels()

    @property
    def ignore_zero_weight(self):
        return self._ignore_zero_weight

    @ignore_zero_weight.setter
    def ignore_zero_weight(self, ignore_zero_weight):
        changed = ignore_zero_weight != self._ignore_zero_weight
        self._ignore_zero_weight = ignore_zero_weight
        if changed:
            self.update_pixels()

    # ---- image / weight / jacobian
    @property
    def image(self):
        return self._get_view(self._image)

    @image.setter
    def image(self, image):
        self.set_image(image)

    def set_image(self, image, update_pixels=True):
        old = getattr(self, "_image", None)
        image = np.asarray(image, dtype="f8")
        assert len(image.shape) == 2, "image must be 2d"
        if old is not None:
            assert image.shape == old.shape, (
                "old and new image must have same shape, to maintain "
                "consistency, got %s vs %s" % (image.shape, old.shape))
        self._image = image
        if update_pixels:
            self.update_pixels()

    @property
    def weight(self):
        return self._get_view(self._weight)

    @weight.setter
    def weight(self, weight):
        self.set_weight(weight)

    def set_weight(self, weight, update_pixels=True):
        """the weight map, same shape as the image; None means unit weights"""
        shape = self._image.shape
        if weight is None:
            wt = np.ones(shape)
        else:
            wt = np.asarray(weight, dtype="f8")
            assert wt.ndim == 2, "weight must be 2d"
            assert wt.shape == shape, "image and weight must be same shape"
        self._weight = wt
        if update_pixels:
            self.update_pixels()

    @property
    def jacobian(self):
        return self.get_jacobian()

    @jacobian.setter
    def jacobian(self, jacobian):
        self.set_jacobian(jacobian)

    def set_jacobian(self, jacobian, update_pixels=True):
        if jacobian is None:
            cen = (np.array(self._image.shape) - 1.0) / 2.0
            jac = UnitJacobian(row=cen[0], col=cen[1])
        else:
            assert isinstance(jacobian, Jacobian), (
                "jacobian must be of type Jacobian, got %s" % type(jacobian))
            jac = jacobian.copy()
        self._jacobian = jac
        if update_pixels:
            self.update_pixels()

    def get_jacobian(self):
        """a copy whose data is a (read-only) view of ours"""
        j = self._jacobian.copy()
        j._data = self._get_view(self._jacobian._data)
        return j

    # ---- optional same-shape arrays: bmask, ormask, noise, mfrac
    def _set_optional(self, name, arr, dtype=None):
        attr = "_" + name
        if arr is None:
            if hasattr(self, attr):
                delattr(self, attr)
            return
        arr = np.asarray(arr) if dtype is None else np.asarray(arr, dtype=dtype)
        assert len(arr.shape) == 2, "%s must be 2d" % name
        assert arr.shape == self._image.shape, (
            "image and %s must be same shape" % name)
        setattr(self, attr, arr)

    def set_bmask(self, bmask):
        self._set_optional("bmask", bmask)

    def set_ormask(self, ormask):
        self._set_optional("ormask", ormask)

    def set_noise(self, noise):
        self._set_optional("noise", noise)

    def set_mfrac(self, mfrac):
        self._set_optional("mfrac", mfrac)

    def has_bmask(self):
        return hasattr(self, "_bmask")

    def has_ormask(self):
        return hasattr(self, "_ormask")

    def has_noise(self):
        return hasattr(self, "_noise")

    def has_mfrac(self):
        return hasattr(self, "_mfrac")

    @property
    def bmask(self):
        return self._get_view(self._bmask)

    @bmask.setter
    def bmask(self, bmask):
        self.set_bmask(bmask)

    @property
    def ormask(self):
        return self._get_view(self._ormask)

    @ormask.setter
    def ormask(self, ormask):
        self.set_ormask(ormask)

    @property
    def noise(self):
        return self._get_view(self._noise)

    @noise.setter
    def noise(self, noise):
        self.set_noise(noise)

    @property
    def mfrac(self):
        return self._get_view(self._mfrac)

    @mfrac.setter
    def mfrac(self, mfrac):
        self.set_mfrac(mfrac)

    # ---- gmix and psf
    @property
    def gmix(self):
        return self.get_gmix()

    @gmix.setter
    def gmix(self, gmix):
        self.set_gmix(gmix)

    def set_gmix(self, gmix):
        if self.has_gmix():
            del self._gmix
        if gmix is not None:
            assert isinstance(gmix, GMix), (
                "gmix must be of type GMix, got %s" % type(gmix))
            self._gmix = gmix.copy()

    def get_gmix(self):
        if not self.has_gmix():
            raise RuntimeError("this obs has not gmix set")
        return self._gmix.copy()

    def has_gmix(self):
        return hasattr(self, "_gmix")

    @property
    def psf(self):
        return self._psf

    @psf.setter
    def psf(self, psf):
        self.set_psf(psf)

    def set_psf(self, psf):
        if self.has_psf():
            del self._psf
        if psf is not None:
            assert isinstance(psf, Observation), (
                "psf must be of Observation, got %s" % type(psf))
            self._psf = psf

    def get_psf(self):
        if not self.has_psf():
            raise RuntimeError("this obs has no psf set")
        return self._psf

    def has_psf(self):
        return hasattr(self, "_psf")

    def get_psf_gmix(self):
        if not self.has_psf_gmix():
            raise RuntimeError("this obs has not psf set with a gmix")
        return self.psf.get_gmix()

    def has_psf_gmix(self):
        return self.has_psf() and self.psf.has_gmix()

    # ---- simple s/n
    def get_s2n(self):
        Isum, Vsum, _ = self.get_s2n_sums()
        return _simple_s2n(Isum, Vsum)

    def get_s2n_sums(self):
        w = np.where(self._weight > 0)
        if w[0].size > 0:
            return (self._image[w].sum(), (1.0 / self._weight[w]).sum(), w[0].size)
        return 0.0, 0.0, 0

    # ---- copy / compare
    def copy(self, memo=None):
        def opt(name):
            return getattr(self, name).copy() if getattr(self, "has_" + name)() else None
        return Observation(
            self.image.copy(), weight=self.weight.copy(), bmask=opt("bmask"),
            ormask=opt("ormask"), noise=opt("noise"),
            gmix=self.gmix if self.has_gmix() else None,
            jacobian=self.jacobian,
            meta=copy.deepcopy(self._meta, memo=memo),
            psf=self.psf.copy() if self.has_psf() else None,
            mfrac=opt("mfrac"), store_pixels=self._store_pixels,
            ignore_zero_weight=self._ignore_zero_weight)

    def __copy__(self):
        return self.copy()

    def __deepcopy__(self, memo):
        result = self.copy(memo=memo)
        memo[id(self)] = result
        return result

    def __eq__(self, obs):
        if not isinstance(obs, Observation):
            raise ValueError(f"expected Observation, got {type(obs)}")
        if self.meta != obs.meta:
            return False
        for attr in ("image", "weight", "bmask", "ormask", "mfrac", "noise",
                     "psf", "gmix", "jacobian", "meta"):
            has = "has_" + attr
            if hasattr(self, has):
                self_has = getattr(self, has)()
                obs_has = getattr(obs, has)()
            else:
                self_has = obs_has = True
            if self_has or obs_has:
                if not (self_has and obs_has):
                    return False
                if not np.all(getattr(self, attr) == getattr(obs, attr)):
                    return False
        return True


class ObsList(list, MetadataMixin):
    """a list of Observation (type checked), e.g. the epochs of one band"""

    def __init__(self, meta=None):
        super().__init__()
        self.set_meta(meta)

    def append(self, obs):
        assert isinstance(obs, Observation), (
            "obs should be of type Observation, got %s" % type(obs))
        super().append(obs)

    def __setitem__(self, index, obs):
        assert isinstance(obs, Observation), "obs should be of type Observation"
        super().__setitem__(index, obs)

    def get_s2n_sums(self):
        Isum = Vsum = 0.0
        Npix = 0
        for obs in self:
            a, b, c = obs.get_s2n_sums()
            Isum += a
            Vsum += b
            Npix += c
        return Isum, Vsum, Npix

    def get_s2n(self):
        Isum, Vsum, _ = self.get_s2n_sums()
        return _simple_s2n(Isum, Vsum)

    def copy(self, memo=None):
        new = ObsList(meta=copy.deepcopy(self._meta, memo))
        for obs in self:
            new.append(obs.copy(memo=memo))
        return new

    def __copy__(self):
        return self.copy()

    def __deepcopy__(self, memo):
        result = self.copy(memo=memo)
        memo[id(self)] = result
        return result

    def __eq__(self, other):
        if not isinstance(other, ObsList):
            raise ValueError(f"expected ObsList, got {type(other)}")
        return len(self) == len(other) and all(a == b for a, b in zip(self, other))


class MultiBandObsList(list, MetadataMixin):
    """a list of ObsList (type checked), one per band"""

    def __init__(self, meta=None):
        super().__init__()
        self.set_meta(meta)

    def append(self, obs_list):
        assert isinstance(obs_list, ObsList), "obs_list should be of type ObsList"
        super().append(obs_list)

    def __setitem__(self, index, obs_list):
        assert isinstance(obs_list, ObsList), "obs_list should be of type ObsList"
        super().__setitem__(index, obs_list)

    def get_s2n_sums(self):
        Isum = Vsum = 0.0
        Npix = 0
        for obslist in self:
            a, b, c = obslist.get_s2n_sums()
            Isum += a
            Vsum += b
            Npix += c
        return Isum, Vsum, Npix

    def get_s2n(self):
        Isum, Vsum, _ = self.get_s2n_sums()
        return _simple_s2n(Isum, Vsum)

    def copy(self, memo=None):
        new = MultiBandObsList(meta=copy.deepcopy(self._meta, memo=memo))
        for obslist in self:
            new.append(obslist.copy(memo=memo))
        return new

    def __copy__(self):
        return self.copy()

    def __deepcopy__(self, memo):
        result = self.copy(memo=memo)
        memo[id(self)] = result
        return result

    def __eq__(self, other):
        if not isinstance(other, MultiBandObsList):
            raise ValueError(f"expected MultiBandObsList, got {type(other)}")
        return len(self) == len(other) and all(a == b for a, b in zip(self, other))


def get_mb_obs(obs_in):
    """wrap an Observation / ObsList into a MultiBandObsList"""
    if isinstance(obs_in, Observation):
        obs_list = ObsList()
        obs_list.append(obs_in)
        mb = MultiBandObsList()
        mb.append(obs_list)
        return mb
    if isinstance(obs_in, ObsList):
        mb = MultiBandObsList()
        mb.append(obs_in)
        return mb
    if isinstance(obs_in, MultiBandObsList):
        return obs_in
    raise ValueError("obs should be Observation, ObsList, or MultiBandObsList")
