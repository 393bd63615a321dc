"""
Device-resident batches: the MI355X-native form of the reference's
per-Observation loops.

The reference has no batch API (every entry point takes one Observation and
loops in Python); here N independent stamps live in HBM
in a compact layout -- 8 B val + 8 B ierr per pixel, one 64-byte jacobian and
one gaussian mixture per stamp -- and each operation of the hot path is ONE
kernel launch over the whole batch (one work-group per stamp).

torch is used only as the owner of device memory and streams; every compute
call goes through the C ABI in include/ngmix_hip.h.
"""
import ctypes

import numpy as np

from . import _lib


def _torch():
    import torch
    return torch


def _stream():
    """the current stream of the current device as a hipStream_t (the raw
    handle straight from torch's C layer: torch.cuda.current_stream() builds a
    Stream object per call, ~9 us -- a tenth of a one-object fit's host time
    when every launch asks for it)"""
    torch = _torch()
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:
        return ctypes.c_void_p(raw(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _on_device(object):
    """torch.cuda.device(dev) only when dev is not already the current device
    (entering and leaving that context costs ~12 us; nearly every call here is
    made on the current device)"""

    def __init__(self, dev):
        torch = _torch()
        idx = dev.index if hasattr(dev, "index") else int(dev)
        self._ctx = None
        if idx is not None and idx != torch.cuda.current_device():
            self._ctx = torch.cuda.device(dev)

    def __enter__(self):
        if self._ctx is not None:
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self._ctx is not None:
            return self._ctx.__exit__(*exc)
        return False


def _dptr(t):
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def _as_device_f64(x, device):
    torch = _torch()
    if isinstance(x, np.ndarray):
        x = np.ascontiguousarray(x, dtype=np.float64)
        if not x.flags["WRITEABLE"]:
            x = x.copy()  # torch cannot wrap read-only arrays (obs.image views)
        x = torch.from_numpy(x)
    return x.to(device=device, dtype=torch.float64).contiguous()


def _require_cuda(device):
    torch = _torch()
    if not torch.cuda.is_available():
        raise RuntimeError(
            "ngmix_amd needs a GPU: the HIP kernels are the only compute path")
    dev = torch.device(device if device is not None else "cuda")
    if dev.type != "cuda":
        raise ValueError("device must be a cuda (ROCm) device")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


class GMixBatch(object):
    """
    n mixtures of `ngauss` gaussians each, on the device, in the reference's
    gauss2d record layout (ngmix/gmix/gmix.py:1196-1210): a float64 tensor of
    shape (n*ngauss, 13) whose column 7 carries the int64 norm_set bits.
    """

    def __init__(self, data, n, ngauss):
        self.data = data
        self.n = int(n)
        self.ngauss = int(ngauss)

    @property
    def device(self):
        return self.data.device

    @classmethod
    def empty(cls, n, ngauss, device=None):
        torch = _torch()
        dev = _require_cuda(device)
        data = torch.zeros((n * ngauss, 13), dtype=torch.float64, device=dev)
        return cls(data, n, ngauss)

    @classmethod
    def from_numpy(cls, arr, device=None):
        """arr: structured array of shape (n, ngauss) or (ngauss,) with
        _lib.GAUSS2D_DTYPE"""
        torch = _torch()
        dev = _require_cuda(device)
        arr = np.ascontiguousarray(arr, dtype=_lib.GAUSS2D_DTYPE)
        if arr.ndim == 1:
            arr = arr[None, :]
        n, ngauss = arr.shape
        flat = arr.reshape(-1).view(np.float64).reshape(-1, 13)
        return cls(torch.from_numpy(flat.copy()).to(dev), n, ngauss)

    @classmethod
    def from_pars(cls, pars, model, device=None, cm_extra=None, ngauss=None):
        """
        Fill n model mixtures on the device (gmix_nb.py:307-558).  pars is
        (n, npars).  Returns (GMixBatch, status) where status[i] != 0 marks
        a stamp whose fill the reference would have rejected (g >= 1 ...).
        """
        torch = _torch()
        from .gmix import get_model_num, get_model_ngauss
        dev = _require_cuda(device)
        modnum = get_model_num(model)
        pars = _as_device_f64(pars, dev)
        if pars.ndim == 1:
            pars = pars[None, :]
        n, npars = pars.shape
        if ngauss is None:
            ngauss = get_model_ngauss(modnum)
        out = cls.empty(n, ngauss, dev)
        status = torch.zeros(n, dtype=torch.int32, device=dev)
        extra = None
        if cm_extra is not None:
            extra = _as_device_f64(cm_extra, dev)
        with _on_device(dev):
            st = _lib.lib().ngmix_fill_model_batch(
                _dptr(out.data), n, ngauss, modnum, _dptr(pars), npars,
                _dptr(extra), _dptr(status), _stream())
        _lib.check(st, "ngmix_fill_model_batch")
        return out, status

    def convolve(self, psf):
        """per-stamp gmix_convolve_fill (gmix_nb.py:609-649);
        returns (GMixBatch, status)"""
        torch = _torch()
        assert psf.n == self.n
        out = GMixBatch.empty(self.n, self.ngauss * psf.ngauss, self.device)
        status = torch.zeros(self.n, dtype=torch.int32, device=self.device)
        with _on_device(self.device):
            st = _lib.lib().ngmix_convolve_fill_batch(
                _dptr(out.data), _dptr(self.data), self.ngauss, _dptr(psf.data),
                psf.ngauss, self.n, _dptr(status), _stream())
        _lib.check(st, "ngmix_convolve_fill_batch")
        return out, status

    def set_norms(self):
        """gmix_set_norms per stamp (gmix_nb.py:176-218); returns status"""
        torch = _torch()
        status = torch.zeros(self.n, dtype=torch.int32, device=self.device)
        with _on_device(self.device):
            st = _lib.lib().ngmix_set_norms_batch(
                _dptr(self.data), self.ngauss, self.n, _dptr(status), _stream())
        _lib.check(st, "ngmix_set_norms_batch")
        return status

    def to_numpy(self):
        """structured (n, ngauss) array with the reference dtype"""
        flat = self.data.detach().cpu().numpy()
        return flat.reshape(-1).view(_lib.GAUSS2D_DTYPE).reshape(
            self.n, self.ngauss).copy()

    def clone(self):
        return GMixBatch(self.data.clone(), self.n, self.ngauss)

    def select(self, index):
        """the mixtures `index` (any order, repeats allowed) as a new batch;
        the companion of StampBatch.select"""
        torch = _torch()
        idx = torch.as_tensor(np.ascontiguousarray(index, dtype=np.int64),
                              device=self.device)
        rows = self.data.reshape(self.n, self.ngauss * 13)[idx]
        return GMixBatch(rows.reshape(-1, 13).contiguous(), idx.numel(), self.ngauss)


class StampBatch(object):
    """
    N stamps resident in HBM.

    val / ierr: float64 tensors, full-frame row-major stamps back to back
    jac:        (N, 8) float64, the reference's jacobian record per stamp
    geometry:   host structured array (nrow, ncol, pix_off, flags, npix_kept)

    The reference's pixel list (Observation.pixels) is implicit: the k-th
    pixel with weight > 0 in row-major order (pixels_nb.py:33-38).
    """

    def __init__(self, val, ierr, jac, nrow, ncol, pix_off, ignore_zero_weight,
                 npix_kept=None):
        torch = _torch()
        self.val = val
        self.ierr = ierr
        self.jac = jac
        self.device = ierr.device if ierr is not None else jac.device
        self.n = int(jac.shape[0])
        self.nrow = np.ascontiguousarray(nrow, dtype=np.int32)
        self.ncol = np.ascontiguousarray(ncol, dtype=np.int32)
        self.pix_off = np.ascontiguousarray(pix_off, dtype=np.int64)
        izw = np.broadcast_to(np.asarray(ignore_zero_weight, dtype=bool), (self.n,))
        self.flags = np.where(izw, _lib.STAMP_IGNORE_ZERO_WEIGHT, 0).astype(np.int32)
        self.npix = self.nrow.astype(np.int64) * self.ncol
        # diagnostic: fused kernels through the compiler-tracked load path
        self.tracked_loads = False
        self.max_npix = int(self.npix.max()) if self.n else 0
        self.total_pix = int(self.npix.sum())
        self._stamp_tables = {}
        # count kept pixels once (device) so masked stamps are known
        self.npix_kept = self.npix.astype(np.int32).copy()
        if npix_kept is not None:
            # (counted by the caller from the host weights: no kernel, no read-back)
            self.npix_kept = np.ascontiguousarray(npix_kept, dtype=np.int32)
        elif ierr is not None and self.n:
            tab = self._make_table(self.npix_kept, 0, 0)
            with _on_device(self.device):
                st = _lib.lib().ngmix_count_kept_batch(
                    _dptr(tab), self.n, _dptr(self.ierr), _stream())
            _lib.check(st, "ngmix_count_kept_batch")
            host = tab.cpu().numpy().reshape(-1).view(_lib.STAMP_DTYPE)
            self.npix_kept = host["npix_kept"].copy()
        self.any_masked = bool(np.any(self.npix_kept != self.npix))

    # ------------------------------------------------------------ builders
    @classmethod
    def from_images(cls, images, weights=None, jacobians=None,
                    ignore_zero_weight=True, device=None):
        """
        images:    (N, nrow, ncol) array/tensor (same shape for every stamp)
        weights:   same shape, or None for all-ones (observation.py:376)
        jacobians: (N, 8) / (8,) reference jacobian records, Jacobian objects,
                   or None for a unit jacobian at the stamp centre
        """
        torch = _torch()
        dev = _require_cuda(device)
        val = _as_device_f64(images, dev)
        if val.ndim == 2:
            val = val[None]
        n, nrow, ncol = val.shape
        if weights is None:
            ierr = torch.ones_like(val)
        else:
            w = _as_device_f64(weights, dev)
            if w.ndim == 2:
                w = w[None]
            assert w.shape == val.shape, "image and weight must match"
            ierr = torch.empty_like(w)
            with _on_device(dev):
                st = _lib.lib().ngmix_weight_to_ierr_batch(
                    _dptr(w), _dptr(ierr), w.numel(), _stream())
            _lib.check(st, "ngmix_weight_to_ierr_batch")
        jac = cls._jacobian_tensor(jacobians, n, nrow, ncol, dev)
        npix = nrow * ncol
        return cls(val.reshape(-1), ierr.reshape(-1), jac,
                   np.full(n, nrow), np.full(n, ncol),
                   np.arange(n, dtype=np.int64) * npix, ignore_zero_weight)

    @classmethod
    def from_observations(cls, obs_list, device=None):
        """ragged batch from Observation objects (any mix of shapes)"""
        # the Observations' own arrays (their public properties hand out
        # read-only views and a COPY of the jacobian: 10 us per object, more than
        # the packing itself)
        n = len(obs_list)
        if n >= 32:
            shape = obs_list[0]._image.shape
            if all(o._image.shape == shape for o in obs_list):
                # one shape (a survey's stamps): the packing is three np.stack
                # calls and the per-stamp work numpy's, not a Python loop
                # (~10 us per object: 1e5 objects a second against the 1.6e7 the
                # device fits)
                return cls.from_stacked(
                    np.stack([o._image for o in obs_list]),
                    np.stack([o._weight for o in obs_list]),
                    np.stack([o._jacobian._data for o in obs_list]),
                    np.fromiter((o._ignore_zero_weight for o in obs_list), dtype=bool,
                                count=n), device=device)
        return cls.from_arrays([o._image for o in obs_list],
                               [o._weight for o in obs_list],
                               [o._jacobian._data for o in obs_list],
                               [o._ignore_zero_weight for o in obs_list], device=device)

    @classmethod
    def from_arrays(cls, imgs, weights, jac_records, ignore_zero_weight, device=None):
        """ragged batch from host arrays: 2-d images and weight maps, jacobian
        records (the reference dtype, or 8 doubles), one flag per stamp"""
        torch = _torch()
        dev = _require_cuda(device)
        nrow = np.array([im.shape[0] for im in imgs], dtype=np.int32)
        ncol = np.array([im.shape[1] for im in imgs], dtype=np.int32)
        npix = nrow.astype(np.int64) * ncol
        off = np.concatenate([[0], np.cumsum(npix)[:-1]]).astype(np.int64)
        izw = np.array(ignore_zero_weight, dtype=bool)
        # ONE upload: [val | weight | jacobian records] (a host-to-device copy
        # costs ~30 us whatever its size: three of them were a third of a
        # one-object fit's set-up), and the listed pixels are counted here
        # from the host weights instead of by a kernel and a read-back
        tot = int(npix.sum())
        n = len(imgs)
        host = np.empty(2 * tot + 8 * n + 4 * n)      # ... | stamp table (32 B records)
        kept = np.empty(n, dtype=np.int32)
        for i in range(n):
            a, b = int(off[i]), int(off[i] + npix[i])
            host[a:b] = np.asarray(imgs[i], dtype="f8").ravel()
            w = np.asarray(weights[i], dtype="f8").ravel()
            host[tot + a:tot + b] = w
            kept[i] = np.count_nonzero(w > 0.0) if izw[i] else npix[i]
            host[2 * tot + 8 * i:2 * tot + 8 * i + 8] = \
                np.ascontiguousarray(jac_records[i]).view(np.float64).reshape(8)
        # (the stamp table of one-gaussian-per-stamp mixtures -- what the lock-step
        # fits and the moments kernels ask for -- rides along: one upload less)
        tab = np.zeros(n, dtype=_lib.STAMP_DTYPE)
        tab["pix_off"], tab["nrow"], tab["ncol"] = off, nrow, ncol
        tab["gm_off"] = np.arange(n, dtype=np.int32)
        tab["ngauss"] = 1
        tab["flags"] = np.where(izw, _lib.STAMP_IGNORE_ZERO_WEIGHT, 0)
        tab["npix_kept"] = kept
        host[2 * tot + 8 * n:] = tab.view(np.float64)
        dev_all = torch.from_numpy(host).to(dev)
        dval = dev_all[:tot]
        dw = dev_all[tot:2 * tot]
        djac = dev_all[2 * tot:2 * tot + 8 * n].reshape(n, 8)
        dtab = dev_all[2 * tot + 8 * n:].view(torch.int32).reshape(n, 8)
        # in place (the kernel is element-wise, as in ngmix_batch_upload): the
        # weight slice of the one allocation BECOMES ierr, so a batch holds two
        # doubles per pixel, not three
        with _on_device(dev):
            st = _lib.lib().ngmix_weight_to_ierr_batch(
                _dptr(dw), _dptr(dw), dw.numel(), _stream())
        _lib.check(st, "ngmix_weight_to_ierr_batch")
        sb = cls(dval, dw, djac, nrow, ncol, off, izw, npix_kept=kept)
        sb._stamp_tables[1] = dtab
        return sb

    @classmethod
    def from_stacked(cls, images, weights, jac_records, ignore_zero_weight=True,
                     device=None):
        """
        N stamps of ONE shape from HOST arrays -- images, weights (N, nrow,
        ncol) float64 (torch tensors: or float32, widened on the device),
        jacobian records (N,) of the reference dtype or (N, 8) doubles.  What a
        host-resident catalogue costs is the PCIe transfer of 16 bytes per
        pixel (8 for float32 stamps):

          * numpy arrays: one host-to-device copy of [val | weight | jacobians
            | stamp table] assembled in a staging buffer, the listed pixels
            counted from the host weights;
          * torch CPU tensors (PINNED memory: a catalogue read into
            torch.empty(..., pin_memory=True)): copied from where they lie,
            asynchronously, with no staging pass over them on the host; the
            listed pixels are counted by a kernel.
        """
        torch = _torch()
        dev = _require_cuda(device)
        if isinstance(images, torch.Tensor):
            ok = (torch.float64, torch.float32)
            assert images.dtype in ok and weights.dtype in ok, "float64 or float32 stamps"
            n, nrow, ncol = images.shape
            assert weights.shape == images.shape, "image and weight must match"
            # float32 stamps (what survey postage stamps are stored as) cross
            # the link as they are, 4 bytes per value, and are widened on the
            # device: the conversion is exact, so the float64 arithmetic sees
            # the values np.array(image, dtype='f8') gives the reference
            # (observation.py:373-376)
            dval = images.to(dev, non_blocking=True).reshape(-1)
            dw = weights.to(dev, non_blocking=True).reshape(-1)
            if dval.dtype != torch.float64:
                dval = dval.to(torch.float64)
            if dw.dtype != torch.float64:
                dw = dw.to(torch.float64)
            jr = jac_records if isinstance(jac_records, torch.Tensor) else \
                torch.from_numpy(np.ascontiguousarray(jac_records).view(np.float64).reshape(n, 8))
            djac = jr.to(dev, non_blocking=True).reshape(n, 8)
            if dw.data_ptr() == weights.data_ptr():
                dw = dw.clone()     # (already on the device: the caller keeps its weights)
            with _on_device(dev):
                st = _lib.lib().ngmix_weight_to_ierr_batch(
                    _dptr(dw), _dptr(dw), dw.numel(), _stream())
            _lib.check(st, "ngmix_weight_to_ierr_batch")
            npix = nrow * ncol
            return cls(dval, dw, djac, np.full(n, nrow), np.full(n, ncol),
                       np.arange(n, dtype=np.int64) * npix, ignore_zero_weight)
        images = np.asarray(images, dtype="f8")
        weights = np.asarray(weights, dtype="f8")
        n, nrow, ncol = images.shape
        assert weights.shape == images.shape, "image and weight must match"
        npix = nrow * ncol
        tot = n * npix
        jr = np.ascontiguousarray(jac_records)
        if jr.dtype.names is not None:
            jr = jr.view(np.float64)
        jr = jr.reshape(n, 8)
        izw = np.broadcast_to(np.asarray(ignore_zero_weight, dtype=bool), (n,))
        kept = np.where(izw, np.count_nonzero(weights.reshape(n, -1) > 0.0, axis=1),
                        npix).astype(np.int32)
        off = np.arange(n, dtype=np.int64) * npix
        tab = np.zeros(n, dtype=_lib.STAMP_DTYPE)
        tab["pix_off"], tab["nrow"], tab["ncol"] = off, nrow, ncol
        tab["gm_off"] = np.arange(n, dtype=np.int32)
        tab["ngauss"] = 1
        tab["flags"] = np.where(izw, _lib.STAMP_IGNORE_ZERO_WEIGHT, 0)
        tab["npix_kept"] = kept
        host = np.empty(2 * tot + 8 * n + 4 * n)
        host[:tot] = images.reshape(-1)
        host[tot:2 * tot] = weights.reshape(-1)
        host[2 * tot:2 * tot + 8 * n] = jr.reshape(-1)
        host[2 * tot + 8 * n:] = tab.view(np.float64)
        dev_all = torch.from_numpy(host).to(dev)
        dval = dev_all[:tot]
        dw = dev_all[tot:2 * tot]
        djac = dev_all[2 * tot:2 * tot + 8 * n].reshape(n, 8)
        dtab = dev_all[2 * tot + 8 * n:].view(torch.int32).reshape(n, 8)
        with _on_device(dev):
            st = _lib.lib().ngmix_weight_to_ierr_batch(
                _dptr(dw), _dptr(dw), dw.numel(), _stream())
        _lib.check(st, "ngmix_weight_to_ierr_batch")
        sb = cls(dval, dw, djac, np.full(n, nrow), np.full(n, ncol), off, izw, npix_kept=kept)
        sb._stamp_tables[1] = dtab
        return sb

    @classmethod
    def from_observations_geometry(cls, obs_list, device=None):
        """shapes and jacobians only (no pixel data): every pixel of every
        stamp is listed -- for render / deriv_images over full frames"""
        dev = _require_cuda(device)
        nrow = np.array([o.image.shape[0] for o in obs_list], dtype=np.int32)
        ncol = np.array([o.image.shape[1] for o in obs_list], dtype=np.int32)
        npix = nrow.astype(np.int64) * ncol
        off = np.concatenate([[0], np.cumsum(npix)[:-1]]).astype(np.int64)
        jac = np.stack([o.jacobian.get_data().view(np.float64).reshape(8)
                        for o in obs_list])
        return cls(None, None, _as_device_f64(jac, dev), nrow, ncol, off, False)

    @staticmethod
    def _jacobian_tensor(jacobians, n, nrow, ncol, dev):
        if jacobians is None:
            rec = np.zeros(8)
            rec[0] = (nrow - 1.0) / 2.0
            rec[1] = (ncol - 1.0) / 2.0
            rec[2] = rec[5] = rec[6] = rec[7] = 1.0
            arr = np.tile(rec, (n, 1))
        elif hasattr(jacobians, "get_data"):
            arr = np.tile(jacobians.get_data().view(np.float64).reshape(8), (n, 1))
        elif isinstance(jacobians, (list, tuple)):
            arr = np.stack([j.get_data().view(np.float64).reshape(8)
                            for j in jacobians])
        else:
            torch = _torch()
            if isinstance(jacobians, torch.Tensor):
                t = jacobians.to(device=dev, dtype=torch.float64)
                if t.ndim == 1:
                    t = t[None].expand(n, 8)
                return t.contiguous()
            arr = np.asarray(jacobians)
            if arr.dtype.names is not None:
                arr = arr.view(np.float64)
            arr = arr.reshape(-1, 8)
            if arr.shape[0] == 1 and n > 1:
                arr = np.tile(arr, (n, 1))
        assert arr.shape == (n, 8)
        return _as_device_f64(arr, dev)

    # ----------------------------------------------------------- internals
    def _make_table(self, npix_kept, gm_off, ngauss):
        torch = _torch()
        tab = np.zeros(self.n, dtype=_lib.STAMP_DTYPE)
        tab["pix_off"] = self.pix_off
        tab["nrow"] = self.nrow
        tab["ncol"] = self.ncol
        tab["gm_off"] = gm_off
        tab["ngauss"] = ngauss
        tab["flags"] = self.flags
        tab["npix_kept"] = npix_kept
        flat = tab.view(np.int32).reshape(self.n, 8)
        return torch.from_numpy(flat.copy()).to(self.device)

    def stamp_table(self, ngauss):
        """device ngmix_stamp records for mixtures laid out stamp-major.
        ngauss: one int (gm_off = i*ngauss) or a per-stamp sequence (ragged,
        gm_off = running sum); cached per layout"""
        if np.ndim(ngauss) == 0:
            key = int(ngauss)
            ng = np.full(self.n, key, dtype=np.int64)
        else:
            ng = np.asarray(ngauss, dtype=np.int64)
            assert ng.shape == (self.n,)
            key = tuple(ng.tolist())
        if key not in self._stamp_tables:
            gm_off = np.concatenate([[0], np.cumsum(ng)[:-1]]) if self.n else ng
            assert (gm_off[-1] + ng[-1]) < 2 ** 31 if self.n else True
            self._stamp_tables[key] = self._make_table(
                self.npix_kept, gm_off.astype(np.int32), ng.astype(np.int32))
        return self._stamp_tables[key]

    def _batch(self, ngauss, no_skip=False, exact=False):
        b = _lib.Batch()
        b.nstamps = self.n
        b.stamps = self.stamp_table(ngauss).data_ptr()
        if np.ndim(ngauss) != 0:
            ngauss = int(np.max(ngauss)) if self.n else 0
        b.val = self.val.data_ptr() if self.val is not None else None
        b.ierr = self.ierr.data_ptr() if self.ierr is not None else None
        b.jac = self.jac.data_ptr()
        b.max_ngauss = ngauss
        b.max_npix = self.max_npix
        b.any_masked = int(self.any_masked)
        b.max_nrow = int(self.nrow.max()) if self.n else 0
        b.max_ncol = int(self.ncol.max()) if self.n else 0
        b.flags = (_lib.BATCH_NO_SKIP if no_skip else 0) | \
            (_lib.BATCH_EXACT if exact else 0) | \
            (_lib.BATCH_TRACKED_LOADS if self.tracked_loads else 0)
        return b

    def _packed(self):
        """the stamps tile [0, total_pix) without gaps or overlaps, in order"""
        if not hasattr(self, "_packed_layout"):
            start = np.concatenate([[0], np.cumsum(self.npix)[:-1]]) if self.n else \
                np.zeros(0, dtype=np.int64)
            self._packed_layout = bool(np.array_equal(self.pix_off, start))
        return self._packed_layout

    def kept_offsets(self):
        """start of each stamp's segment in a packed per-kept-pixel array
        (the layout of the LM residual vector, results.py:410-421)"""
        kept = self.npix_kept.astype(np.int64)
        return np.concatenate([[0], np.cumsum(kept)[:-1]]).astype(np.int64)

    # ----------------------------------------------------------- operations
    def loglike(self, gm, out=None, status=None, no_skip=False, exact=False):
        """
        get_loglike for every stamp (gmix_nb.py:824-874) in one launch.
        Returns (out, status): out is (N, 4) = loglike, s2n_numer, s2n_denom,
        npix.  Norms are set lazily in-kernel as in the reference.

        exact=False (default): fused kernels (FMA, shared-centre algebra),
        per-pixel model values within ~1e-13 relative of the reference.
        exact=True: no-FMA kernels in the reference's operation order,
        per-pixel values bit-identical to the reference (about 2x slower).
        """
        torch = _torch()
        assert gm.n == self.n
        if out is None:
            out = torch.empty((self.n, 4), dtype=torch.float64, device=self.device)
        if status is None:
            status = torch.empty(self.n, dtype=torch.int32, device=self.device)
        b = self._batch(gm.ngauss, no_skip, exact)
        with _on_device(self.device):
            st = _lib.lib().ngmix_loglike_batch(
                ctypes.byref(b), _dptr(gm.data), _dptr(out), _dptr(status),
                _stream())
        _lib.check(st, "ngmix_loglike_batch")
        return out, status

    def loglike_objects(self, gm, obj_start, out=None, status=None, exact=False):
        """
        get_loglike summed over the observations (epochs / bands) of each
        object, as FitModel.calc_lnprob does (results.py:142-210): stamps
        obj_start[i] .. obj_start[i+1] belong to object i.  One launch over all
        (object, epoch) stamps plus a fixed-order segmented sum on the device.
        Returns (per_object (nobj, 4), per_stamp (N, 4), status).
        """
        per_stamp, status = self.loglike(gm, out=out, status=status, exact=exact)
        per_obj = self.sum_over_epochs(per_stamp, obj_start)
        return per_obj, per_stamp, status

    def sum_over_epochs(self, per_stamp, obj_start):
        """fixed-order sum of per-stamp records over each object's stamps
        obj_start[i] .. obj_start[i+1] (on the device)"""
        torch = _torch()
        obj_start = np.asarray(obj_start, dtype=np.int64)
        lengths = np.diff(obj_start)
        assert obj_start[0] == 0 and obj_start[-1] == self.n and np.all(lengths > 0)
        width = per_stamp.shape[1]
        if np.all(lengths == lengths[0]):
            return per_stamp.reshape(-1, int(lengths[0]), width).sum(dim=1)
        return torch.segment_reduce(
            per_stamp, "sum", lengths=torch.from_numpy(lengths).to(self.device), axis=0)

    def fill_fdiff(self, gm, fdiff=None, fdiff_start=None, status=None,
                   no_skip=False, exact=False):
        """
        fill_fdiff for every stamp (gmix_nb.py:877-900): the k-th kept pixel
        of stamp i goes to fdiff[fdiff_start[i] + k].  Default layout packs the
        stamps back to back.  Returns (fdiff, status).
        """
        torch = _torch()
        assert gm.n == self.n
        if fdiff_start is None:
            fdiff_start = self.kept_offsets()
        if isinstance(fdiff_start, np.ndarray):
            fdiff_start = torch.from_numpy(
                np.ascontiguousarray(fdiff_start, dtype=np.int64)).to(self.device)
        if fdiff is None:
            fdiff = torch.zeros(int(self.npix_kept.sum()), dtype=torch.float64,
                                device=self.device)
        if status is None:
            status = torch.empty(self.n, dtype=torch.int32, device=self.device)
        b = self._batch(gm.ngauss, no_skip, exact)
        with _on_device(self.device):
            st = _lib.lib().ngmix_fill_fdiff_batch(
                ctypes.byref(b), _dptr(gm.data), _dptr(fdiff),
                _dptr(fdiff_start), _dptr(status), _stream())
        _lib.check(st, "ngmix_fill_fdiff_batch")
        return fdiff, status

    def render(self, gm, image=None, fast_exp=True, status=None, no_skip=False,
               exact=False):
        """
        render every stamp's mixture (render_nb.py:9-36), ADDING into `image`
        (flat, same layout as val).  With image=None a fresh image is made as
        GMix.make_image does (gmix.py:561-562: zeros, then the render) -- fused:
        the kernel writes the model without reading the buffer
        (NGMIX_BATCH_RENDER_OVERWRITE), 8 bytes per pixel instead of a memset
        plus a read-modify-write.  Returns (image, status).
        """
        torch = _torch()
        assert gm.n == self.n
        overwrite = False
        if image is None:
            if self._packed():
                image = torch.empty(self.total_pix, dtype=torch.float64,
                                    device=self.device)
                overwrite = True
            else:
                image = torch.zeros(self.total_pix, dtype=torch.float64,
                                    device=self.device)
        if status is None:
            status = torch.empty(self.n, dtype=torch.int32, device=self.device)
        b = self._batch(gm.ngauss, no_skip, exact)
        if overwrite:
            b.flags |= _lib.BATCH_RENDER_OVERWRITE
        with _on_device(self.device):
            st = _lib.lib().ngmix_render_batch(
                ctypes.byref(b), _dptr(gm.data), _dptr(image), int(fast_exp),
                _dptr(status), _stream())
        _lib.check(st, "ngmix_render_batch")
        return image, status

    def model_s2n_sum(self, gm, out=None, status=None, exact=False):
        """get_model_s2n_sum per stamp (gmix_nb.py:903-937)"""
        torch = _torch()
        assert gm.n == self.n
        if out is None:
            out = torch.empty(self.n, dtype=torch.float64, device=self.device)
        if status is None:
            status = torch.empty(self.n, dtype=torch.int32, device=self.device)
        b = self._batch(gm.ngauss, False, exact)
        with _on_device(self.device):
            st = _lib.lib().ngmix_model_s2n_sum_batch(
                ctypes.byref(b), _dptr(gm.data), _dptr(out), _dptr(status),
                _stream())
        _lib.check(st, "ngmix_model_s2n_sum_batch")
        return out, status

    # ------------------------------------------------ moments / iterative ops
    def weighted_sums(self, wt, maxrad, nmom=6, res=None, status=None, exact=False):
        """
        get_weighted_sums / get_higher_order_weighted_sums per stamp
        (gmix_nb.py:681-821); wt must have its norms set.  res: (N, nbytes/8)
        float64 tensor of result records, ADDED into (zeros when None).
        Returns (res, status); view res with records_to_numpy(res, dtype).

        exact=False (default): register accumulators and a fixed-order tree
        (sums agree with the reference to summation-order rounding);
        exact=True: every sum accumulated in the reference's sequential pixel
        order (bit-identical to the seam form, ~60x slower).
        """
        torch = _torch()
        assert wt.n == self.n and nmom in (6, 17)
        nd = _lib.moments_result_dtype(nmom).itemsize // 8
        if res is None:
            res = torch.zeros((self.n, nd), dtype=torch.float64, device=self.device)
        if status is None:
            status = torch.empty(self.n, dtype=torch.int32, device=self.device)
        maxrad = _as_device_f64(np.broadcast_to(np.asarray(maxrad, dtype="f8"),
                                                (self.n,)).copy(), self.device) \
            if not isinstance(maxrad, torch.Tensor) else maxrad
        b = self._batch(wt.ngauss, False, exact)
        with _on_device(self.device):
            st = _lib.lib().ngmix_weighted_sums_batch(
                ctypes.byref(b), _dptr(wt.data), _dptr(res), nmom, _dptr(maxrad),
                _dptr(status), _stream())
        _lib.check(st, "ngmix_weighted_sums_batch")
        return res, status

    def admom(self, wt, maxiter=200, shiftmax=5.0, etol=1.0e-5, Ttol=1.0e-3,
              cenonly=False, res=None, status=None, no_cov=False):
        """
        adaptive moments of every stamp (admom_nb.py:13-108) in one launch.
        wt: GMixBatch with ONE gaussian per stamp, the guess; updated in place
        (it is the weight, as in the reference).  Returns (res, status) with
        res an (N, 73) float64 tensor of 584-byte result records.
        no_cov: skip the 7 x 7 covariance sums (sums_cov stays zero) -- for runs
        that only want the converged weight gaussian, e.g. a fit's guess.
        """
        torch = _torch()
        assert wt.n == self.n and wt.ngauss == 1
        conf = np.zeros(1, dtype=_lib.ADMOM_CONF_DTYPE)
        conf["maxiter"] = maxiter
        conf["shiftmax"] = shiftmax
        conf["etol"] = etol
        conf["Ttol"] = Ttol
        conf["cenonly"] = cenonly
        conf["no_cov"] = no_cov
        if res is None:
            res = torch.zeros((self.n, 73), dtype=torch.float64, device=self.device)
        if status is None:
            status = torch.empty(self.n, dtype=torch.int32, device=self.device)
        b = self._batch(1)
        with _on_device(self.device):
            st = _lib.lib().ngmix_admom_batch(
                _lib.ptr(conf), ctypes.byref(b), _dptr(wt.data), _dptr(res),
                _dptr(status), _stream())
        _lib.check(st, "ngmix_admom_batch")
        return res, status

    def em(self, gm, psf, conv=None, sky=0.0, kind=0, miniter=40, maxiter=500,
           tol=1.0e-5, vary_sky=False, fill_zero_weight=False, out=None,
           status=None):
        """
        EM fit of every stamp (em_nb.py em_run / _fixcen / _fixcov / _fluxonly
        for kind 0..3).  gm: pre-psf guess (updated in place), psf: per-stamp
        psf mixture normalised as EMFitter.go does, conv: their convolution
        (made here when None; updated in place).  sky: scalar or (N,).
        Returns (out, status, conv) with out (N,3) = numiter, frac_diff, sky.
        """
        torch = _torch()
        assert gm.n == self.n and psf.n == self.n
        if conv is None:
            conv, _ = gm.convolve(psf)
        conf = np.zeros(1, dtype=_lib.EM_CONF_DTYPE)
        conf["tol"] = tol
        conf["maxiter"] = maxiter
        conf["miniter"] = miniter
        conf["vary_sky"] = vary_sky
        if not isinstance(sky, torch.Tensor):
            sky = _as_device_f64(np.broadcast_to(np.asarray(sky, dtype="f8"),
                                                 (self.n,)).copy(), self.device)
        if out is None:
            out = torch.empty((self.n, 3), dtype=torch.float64, device=self.device)
        if status is None:
            status = torch.empty(self.n, dtype=torch.int32, device=self.device)
        b = self._batch(conv.ngauss)
        with _on_device(self.device):
            st = _lib.lib().ngmix_em_batch(
                int(kind), _lib.ptr(conf), ctypes.byref(b), _dptr(gm.data),
                gm.ngauss, _dptr(psf.data), psf.ngauss, _dptr(conv.data),
                _dptr(sky), int(fill_zero_weight), _dptr(out), _dptr(status),
                _stream())
        _lib.check(st, "ngmix_em_batch")
        return out, status, conv

    def select(self, index):
        """
        a new batch holding stamps `index` (any order, repeats allowed): their
        pixels are gathered into a fresh contiguous buffer on the device --
        for refitting a subset (Runner's retries) or processing in chunks
        """
        torch = _torch()
        index = np.ascontiguousarray(index, dtype=np.int64)
        npix = self.npix[index]
        off = np.concatenate([[0], np.cumsum(npix)[:-1]]).astype(np.int64)
        d_idx = torch.from_numpy(index).to(self.device)
        if index.size and np.all(npix == npix[0]) and np.all(self.npix == npix[0]) and \
                np.all(self.pix_off == np.arange(self.n, dtype=np.int64) * npix[0]):
            take = lambda t: t.reshape(self.n, -1)[d_idx].reshape(-1)  # noqa: E731
        else:
            src = torch.from_numpy(np.repeat(self.pix_off[index] - off, npix) +
                                   np.arange(int(npix.sum()), dtype=np.int64)).to(self.device)
            take = lambda t: t[src]  # noqa: E731
        izw = (self.flags[index] & _lib.STAMP_IGNORE_ZERO_WEIGHT) != 0
        return StampBatch(take(self.val) if self.val is not None else None,
                          take(self.ierr) if self.ierr is not None else None,
                          self.jac[d_idx], self.nrow[index], self.ncol[index], off, izw)

    def prep_em(self):
        """
        the batched prep_obs of the EM fitters (em.py prep_image): every stamp
        shifted so that its minimum is 0.001*(max-min).  Returns (batch, sky)
        with batch sharing ierr / jacobians with self and sky an (N,) device
        tensor, the value em() wants as `sky`.
        """
        torch = _torch()
        if self.n and np.all(self.npix == self.npix[0]):
            v = self.val.reshape(self.n, -1)
            vmin, vmax = v.amin(dim=1), v.amax(dim=1)
            sky = 0.001 * (vmax - vmin) - vmin
            val = (v + sky[:, None]).reshape(-1)
        else:
            lengths = torch.from_numpy(self.npix).to(self.device)
            vmin = torch.segment_reduce(self.val, "min", lengths=lengths)
            vmax = torch.segment_reduce(self.val, "max", lengths=lengths)
            sky = 0.001 * (vmax - vmin) - vmin
            val = self.val + torch.repeat_interleave(sky, lengths)
        izw = (self.flags & _lib.STAMP_IGNORE_ZERO_WEIGHT) != 0
        return StampBatch(val, self.ierr, self.jac, self.nrow, self.ncol,
                          self.pix_off, izw), sky

    def deriv_images(self, gpars, dcov, ngauss, out=None, out_start=None):
        """
        deriv_images per stamp (derivs_nb.py:40-127).  gpars (N*ngauss, 6) and
        dcov (N*ngauss, 3, 3) device tensors; out is flat, stamp i occupying
        6*npix_kept[i] doubles from out_start[i], as (6, npix_kept) row-major.
        """
        torch = _torch()
        kept = self.npix_kept.astype(np.int64)
        if out_start is None:
            out_start = np.concatenate([[0], np.cumsum(6 * kept)[:-1]])
        if isinstance(out_start, np.ndarray):
            out_start = torch.from_numpy(
                np.ascontiguousarray(out_start, dtype=np.int64)).to(self.device)
        if out is None:
            out = torch.zeros(int(6 * kept.sum()), dtype=torch.float64,
                              device=self.device)
        gpars = _as_device_f64(gpars, self.device)
        dcov = _as_device_f64(dcov, self.device)
        b = self._batch(ngauss)
        with _on_device(self.device):
            st = _lib.lib().ngmix_deriv_images_batch(
                ctypes.byref(b), _dptr(gpars), _dptr(dcov), _dptr(out),
                _dptr(out_start), _stream())
        _lib.check(st, "ngmix_deriv_images_batch")
        return out


def flatten_observations(obs):
    """
    a sequence of per-object Observation / ObsList / MultiBandObsList as ONE
    StampBatch: returns (stamps, stamp_obj, stamp_band, nband, psf) with psf the
    (nstamps, npsf) host gauss2d records of the stamps' psf mixtures, or None
    when the observations carry none (FitModel._setup_fit's rule: the first
    observation decides, results.py:289-296)
    """
    from .observation import Observation, ObsList, MultiBandObsList
    flat, sobj, sband = [], [], []
    if all(type(o) is Observation for o in obs):
        flat = list(obs)
        nobj = len(flat)
        sobj = np.arange(nobj, dtype=np.int32)
        sband = np.zeros(nobj, dtype=np.int32)
        nband = 1
    else:
        nband = None
        for i, o in enumerate(obs):
            if isinstance(o, Observation):
                bands = [[o]]
            elif isinstance(o, ObsList):
                bands = [o]
            elif isinstance(o, MultiBandObsList):
                bands = o
            else:
                raise ValueError("obs should be Observation, ObsList, or MultiBandObsList")
            if nband is None:
                nband = len(bands)
            elif len(bands) != nband:
                raise ValueError("every object of a batch needs the same number of bands")
            for b, ol in enumerate(bands):
                if len(ol) == 0:
                    raise ValueError("object %d has no observation in band %d" % (i, b))
                for e in ol:
                    flat.append(e)
                    sobj.append(i)
                    sband.append(b)
        sobj = np.array(sobj, dtype=np.int32)
        sband = np.array(sband, dtype=np.int32)
    if not flat:
        raise ValueError("no observations")
    stamps = StampBatch.from_observations(flat)
    psf = None
    if flat[0].has_psf_gmix():
        recs = [o._psf._gmix._data for o in flat]
        npsf = recs[0].size
        if any(r.size != npsf for r in recs):
            raise ValueError("the psf mixtures of a batch need one size")
        psf = np.stack(recs)
    return stamps, sobj, sband, nband, psf


def records_to_numpy(t, dtype):
    """an (N, nbytes/8) float64 record tensor as a structured host array; large
    device tensors come back through pinned memory (PyTorch's caching host
    allocator: ten times the rate of a pageable .cpu())"""
    t = t.detach()
    if t.is_cuda and t.numel() * t.element_size() >= (1 << 20):
        torch = _torch()
        staged = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        staged.copy_(t)
        return staged.numpy().reshape(-1).view(dtype)
    a = t.cpu().numpy()
    return a.reshape(-1).view(dtype).copy()


# ---------------------------------------------------------------------------
# single-object conveniences used by the GMix / Observation host classes: a
# one-stamp batch kept resident on the device so repeated evaluations (an LM
# fit calls fill_fdiff hundreds of times on the same pixels) upload only the
# 104-byte-per-gaussian mixture.
# ---------------------------------------------------------------------------

def _gm_batch(gm_data, device):
    return GMixBatch.from_numpy(np.ascontiguousarray(gm_data), device=device)


_NP_OF = {}


def _fetch(*tensors):
    """several small device tensors in ONE download (a device-to-host copy is
    20-40 us of host time whatever its size, and the per-object calls made
    three to five): the tensors are concatenated as bytes on the device and come
    back as host arrays of their own shapes and types.  8-byte types first."""
    torch = _torch()
    if not _NP_OF:
        _NP_OF.update({torch.float64: np.float64, torch.int32: np.int32,
                       torch.int64: np.int64, torch.uint8: np.uint8})
    flat = [t.detach().contiguous().reshape(-1).view(torch.uint8) for t in tensors]
    host = torch.cat(flat).cpu().numpy()
    out, at = [], 0
    for t, f in zip(tensors, flat):
        n = f.numel()
        out.append(host[at:at + n].view(_NP_OF[t.dtype]).reshape(tuple(t.shape)))
        at += n
    return out


def _upload(device, *arrays):
    """several small host arrays of doubles (or gauss2d records) in ONE upload;
    returns float64 device views of their element counts, in order"""
    torch = _torch()
    flats = [np.ascontiguousarray(a).reshape(-1).view(np.float64) for a in arrays]
    dev_all = torch.from_numpy(np.concatenate(flats)).to(device)
    out, at = [], 0
    for f in flats:
        out.append(dev_all[at:at + f.size])
        at += f.size
    return out


def _gm_view(flat, gm_data):
    """a GMixBatch over an uploaded view of one mixture's records"""
    return GMixBatch(flat.reshape(-1, 13), 1, int(np.asarray(gm_data).size))


def _gm_records(flat, gmb):
    """the host copy of a GMixBatch's tensor as the reference's record array"""
    return flat.reshape(-1).view(_lib.GAUSS2D_DTYPE).reshape(gmb.n, gmb.ngauss)


def _finish(gm_data, gmb, status, context, *results):
    """the common tail of the per-object calls in one download: raise on the
    status, mirror the (possibly mutated) mixture onto the caller's records,
    return the host copies of `results`"""
    got = _fetch(*results, gmb.data, status)
    st = int(got[-1][0])
    if st != 0:
        _lib.check(st, context)
    gm_data[:] = _gm_records(got[-2], gmb)[0]
    return got[:-2]


def render_single(gm_data, image, jac_record, fast_exp, exact=False, fresh=False):
    """ADD the mixture into a host image (GMix._fill_image); fresh: the image
    is known to be zeros (GMix.make_image, gmix.py:561-562) -- nothing is
    uploaded and the kernel writes the model without reading the buffer
    (NGMIX_BATCH_RENDER_OVERWRITE: bit-identical to zeros + accumulate)"""
    torch = _torch()
    dev = _require_cuda(None)
    nrow, ncol = image.shape
    jac = np.ascontiguousarray(jac_record).view(np.float64).reshape(1, 8)
    sb = StampBatch(None, None, _as_device_f64(jac, dev), [nrow], [ncol], [0], True)
    gmb = _gm_batch(gm_data, dev)
    dimg = None if fresh else \
        torch.from_numpy(np.ascontiguousarray(image, dtype="f8").ravel()).to(dev)
    dimg, status = sb.render(gmb, image=dimg, fast_exp=fast_exp, exact=exact)
    (himg,) = _finish(gm_data, gmb, status, "render", dimg)
    image[:, :] = himg.reshape(nrow, ncol)


class SingleStamp(object):
    """an Observation's pixels resident on the device (1-stamp StampBatch)"""

    def __init__(self, image, weight, jac_record, ignore_zero_weight, device=None):
        # (one upload, the listed pixels counted on the host: from_arrays)
        self.sb = StampBatch.from_arrays([np.asarray(image)], [np.asarray(weight)],
                                         [jac_record], [bool(ignore_zero_weight)],
                                         device=device)
        self.device = self.sb.device

    @property
    def npix_kept(self):
        return int(self.sb.npix_kept[0])

    def loglike_single(self, gm_data, exact=False):
        gmb = _gm_batch(gm_data, self.device)
        out, status = self.sb.loglike(gmb, exact=exact)
        (o,) = _finish(gm_data, gmb, status, "get_loglike", out)
        o = o[0]
        return float(o[0]), float(o[1]), float(o[2]), int(o[3])

    def fdiff_single(self, gm_data, fdiff, start, exact=False):
        torch = _torch()
        gmb = _gm_batch(gm_data, self.device)
        nk = self.npix_kept
        d = torch.empty(nk, dtype=torch.float64, device=self.device)
        _, status = self.sb.fill_fdiff(gmb, fdiff=d, fdiff_start=np.zeros(1, "i8"),
                                       exact=exact)
        (hd,) = _finish(gm_data, gmb, status, "fill_fdiff", d)
        fdiff[start:start + nk] = hd

    def s2n_single(self, gm_data, exact=False):
        gmb = _gm_batch(gm_data, self.device)
        out, status = self.sb.model_s2n_sum(gmb, exact=exact)
        (o,) = _finish(gm_data, gmb, status, "get_model_s2n_sum", out)
        return float(o.reshape(-1)[0])

    def wsums_single(self, gm_data, res, nmom, maxrad):
        """res: numpy record (void scalar) accumulated into"""
        torch = _torch()
        dt = _lib.moments_result_dtype(nmom)
        host = np.zeros(1, dtype=dt)
        for n in dt.names:
            host[n] = res[n]
        dgm, dres = _upload(self.device, np.ascontiguousarray(gm_data, dtype=_lib.GAUSS2D_DTYPE),
                            host)
        gmb = _gm_view(dgm, gm_data)
        dres = dres.reshape(1, -1)
        _, status = self.sb.weighted_sums(gmb, maxrad, nmom=nmom, res=dres)
        hres, hst = _fetch(dres, status)
        if int(hst[0]) != 0:
            _lib.check(int(hst[0]), "get_weighted_sums")
        back = hres.reshape(-1).view(dt)
        for n in dt.names:
            res[n] = back[n][0]

    def admom_single(self, wt_data, conf, res):
        """conf / res: 1-element record arrays (reference dtypes); wt and res
        are updated in place.  Returns the C-ABI status."""
        torch = _torch()
        dwt, dres = _upload(self.device, np.ascontiguousarray(wt_data, dtype=_lib.GAUSS2D_DTYPE),
                            np.ascontiguousarray(res))
        wtb = _gm_view(dwt, wt_data)
        dres = dres.reshape(1, -1)
        c = conf[0] if conf.ndim else conf
        _, status = self.sb.admom(
            wtb, maxiter=int(c["maxiter"]), shiftmax=float(c["shiftmax"]),
            etol=float(c["etol"]), Ttol=float(c["Ttol"]), cenonly=bool(c["cenonly"]),
            res=dres)
        hres, hwt, hst = _fetch(dres, wtb.data, status)
        res[:] = hres.reshape(-1).view(_lib.ADMOM_RESULT_DTYPE)
        wt_data[:] = _gm_records(hwt, wtb)[0]
        return int(hst[0])

    def em_single(self, kind, conf, gm_data, psf_data, conv_data, fill_zero_weight):
        """returns (status, numiter, frac_diff, sky); mixtures updated in place"""
        G = _lib.GAUSS2D_DTYPE
        dgm, dpsf, dconv, dsky = _upload(
            self.device, np.ascontiguousarray(gm_data, dtype=G),
            np.ascontiguousarray(psf_data, dtype=G), np.ascontiguousarray(conv_data, dtype=G),
            np.array([float(conf["sky"])]))
        gmb, psfb, convb = _gm_view(dgm, gm_data), _gm_view(dpsf, psf_data), \
            _gm_view(dconv, conv_data)
        out, status, _ = self.sb.em(
            gmb, psfb, convb, sky=dsky, kind=kind,
            miniter=int(conf["miniter"]), maxiter=int(conf["maxiter"]),
            tol=float(conf["tol"]), vary_sky=bool(conf["vary_sky"]),
            fill_zero_weight=fill_zero_weight)
        ho, hgm, hconv, hst = _fetch(out, gmb.data, convb.data, status)
        o = ho[0]
        gm_data[:] = _gm_records(hgm, gmb)[0]
        conv_data[:] = _gm_records(hconv, convb)[0]
        return int(hst[0]), int(o[0]), float(o[1]), float(o[2])
