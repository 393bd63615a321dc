"""
The reference's AoS pixel / coord arrays (ngmix/pixels/pixels.py), built on
the GPU by the fill_pixels / fill_coords kernels through the C ABI.  The
kernels read the compact val/ierr layout instead; these arrays exist so that
`Observation.pixels` keeps its meaning and bit-exact contents.
"""
import numpy as np

from . import _lib
from .gexceptions import GMixFatalError

__all__ = ["make_pixels", "make_coords"]

_pixels_dtype = _lib.PIXEL_DTYPE
_coords_dtype = _lib.COORD_DTYPE


def make_pixels(image, weight, jacob, ignore_zero_weight=True):
    """
    1-d array of pixel records (u, v, area, val, ierr, fdiff), row-major,
    dropping weight <= 0 pixels when ignore_zero_weight (pixels.py:6-52,
    pixels_nb.py:6-58).
    """
    image = np.ascontiguousarray(image, dtype="f8")
    weight = np.ascontiguousarray(weight, dtype="f8")
    if ignore_zero_weight:
        npixels = int((weight > 0.0).sum())
        if npixels == 0:
            raise GMixFatalError("no weights > 0")
    else:
        npixels = image.size
    pixels = np.zeros(npixels, dtype=_pixels_dtype)
    rec = np.ascontiguousarray(jacob._data)
    st = _lib.lib().ngmix_fill_pixels(
        _lib.ptr(pixels), npixels, _lib.ptr(image), _lib.ptr(weight),
        image.shape[0], image.shape[1], _lib.ptr(rec), int(ignore_zero_weight))
    _lib.check(st, "ngmix_fill_pixels")
    return pixels


def make_coords(dims, jacob):
    """1-d array of (u, v, area) records for every pixel of a dims image"""
    nrow, ncol = dims
    coords = np.zeros(nrow * ncol, dtype=_coords_dtype)
    rec = np.ascontiguousarray(jacob._data)
    st = _lib.lib().ngmix_fill_coords(_lib.ptr(coords), int(nrow), int(ncol),
                                      _lib.ptr(rec))
    _lib.check(st, "ngmix_fill_coords")
    return coords
