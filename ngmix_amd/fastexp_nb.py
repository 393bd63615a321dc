"""
The fast exponential and the apodisation window of the fast pixel evaluation,
callable on their own (reference: ngmix/fastexp_nb.py): in ngmix_amd these are
device functions inlined into every pixel kernel (csrc/common.hpp); this module
runs the same device code over an array (ngmix_fastexp_batch), so that
fexp(x) here is, to the bit, the number the kernels multiply by.

    fexp(x) = exp5_smooth(x): exp(int(x - 0.5)) from a 16-entry table times a
        fifth-order polynomial in the remainder; no range check, as in the
        reference -- x must lie in (-15.5, 1.5) (callers guard 0 <= chi2 < 25)
    apod_window(chi2), apod_window_deriv(chi2): the quintic smoothstep from 1
        at FASTEXP_APOD_CHI2 to 0 at FASTEXP_MAX_CHI2 and its derivative

Scalars in, a float out; arrays in, arrays out (fexp_arr is the reference's
name for the vectorised form).
"""
import numpy as np

from . import _lib

__all__ = ["fexp", "fexp_arr", "exp5_smooth", "apod_window", "apod_window_deriv",
           "FASTEXP_MAX_CHI2", "FASTEXP_APOD_CHI2"]

FASTEXP_MAX_CHI2 = 25.0
FASTEXP_APOD_CHI2 = 20.0
_APOD_IWIDTH = 1.0 / (FASTEXP_MAX_CHI2 - FASTEXP_APOD_CHI2)


def _on_device(x, which):
    import torch
    from .batch import _require_cuda, _dptr, _stream
    dev = _require_cuda(None)
    arr = np.asarray(x, dtype="f8")
    if which == 0 and arr.size and not (np.all(arr > -15.5) and np.all(arr < 1.5)):
        # the reference reads outside its table here (undefined); refuse
        raise ValueError("fexp: argument outside (-15.5, 1.5)")
    d_x = torch.from_numpy(np.ascontiguousarray(arr).reshape(-1).copy()).to(dev)
    d_out = torch.empty_like(d_x)
    with torch.cuda.device(dev):
        st = _lib.lib().ngmix_fastexp_batch(_dptr(d_x), _dptr(d_out), d_x.numel(), which,
                                            _stream())
    _lib.check(st, "ngmix_fastexp_batch")
    out = d_out.cpu().numpy().reshape(arr.shape)
    return float(out) if arr.ndim == 0 else out


def fexp(x):
    return _on_device(x, 0)


fexp_arr = fexp
exp5_smooth = fexp


def apod_window(chi2):
    return _on_device(chi2, 1)


def apod_window_deriv(chi2):
    return _on_device(chi2, 2)
