"""
ctypes front end for the CPU parity oracle (oracle/libngmix_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the ngmix_amd product.

The function names and argument order mirror the reference's njit seam
 so parity tests read like the reference's own tests:
numpy structured arrays in, results written in place.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# NGMIX_ORACLE_LIB selects another build (the sanitizer build: make -C oracle asan)
_LIBPATH = os.environ.get("NGMIX_ORACLE_LIB", os.path.join(_HERE, "libngmix_oracle.so"))

# status codes (ngmix_oracle.h)
OK = 0
ERR_DET_TOO_LOW = 1
ERR_T_TOO_LOW = 2
ERR_G_RANGE = 3
ERR_GTOT_ZERO = 4
ERR_ELOGL_ZERO = 5
ERR_ZERO_DIV = 6
ERR_PIXELS_NOT_FILLED = 7

# the reference's struct layouts
GAUSS2D_DTYPE = np.dtype([
    ("p", "f8"), ("row", "f8"), ("col", "f8"),
    ("irr", "f8"), ("irc", "f8"), ("icc", "f8"), ("det", "f8"),
    ("norm_set", "i8"),
    ("drr", "f8"), ("drc", "f8"), ("dcc", "f8"), ("norm", "f8"), ("pnorm", "f8"),
])
PIXEL_DTYPE = np.dtype([
    ("u", "f8"), ("v", "f8"), ("area", "f8"),
    ("val", "f8"), ("ierr", "f8"), ("fdiff", "f8"),
])
COORD_DTYPE = np.dtype([("u", "f8"), ("v", "f8"), ("area", "f8")])
JACOBIAN_DTYPE = np.dtype([
    ("row0", "f8"), ("col0", "f8"), ("dvdrow", "f8"), ("dvdcol", "f8"),
    ("dudrow", "f8"), ("dudcol", "f8"), ("det", "f8"), ("scale", "f8"),
])
ADMOM_CONF_DTYPE = np.dtype([
    ("maxiter", "i4"), ("shiftmax", "f8"), ("etol", "f8"), ("Ttol", "f8"),
    ("cenonly", bool),
], align=True)
ADMOM_RESULT_DTYPE = np.dtype([
    ("flags", "i4"), ("numiter", "i4"), ("npix", "i4"), ("wsum", "f8"),
    ("sums", "f8", 7), ("sums_cov", "f8", (7, 7)), ("pars", "f8", 6),
    ("rho4", "f8"), ("F", "f8", 7),
], align=True)
EM_CONF_DTYPE = np.dtype([
    ("tol", "f8"), ("maxiter", "i4"), ("miniter", "i4"), ("sky", "f8"),
    ("vary_sky", "bool"),
], align=True)
EM_SUMS_NDOUBLE = {0: 14, 1: 10, 2: 8, 3: 2}

MODEL_NUMS = {"full": 0, "gauss": 1, "turb": 2, "exp": 3, "dev": 4,
              "bdf": 6, "coellip": 7, "cm": 9, "bd": 10}


def moments_result_dtype(nmom):
    return np.dtype([
        ("flags", "i4"), ("npix", "i4"), ("wsum", "f8"),
        ("sums", "f8", nmom), ("sums_cov", "f8", (nmom, nmom)),
        ("pars", "f8", nmom), ("F", "f8", nmom),
    ], align=True)


assert GAUSS2D_DTYPE.itemsize == 104
assert PIXEL_DTYPE.itemsize == 48
assert ADMOM_CONF_DTYPE.itemsize == 40
assert ADMOM_RESULT_DTYPE.itemsize == 584
assert EM_CONF_DTYPE.itemsize == 32
assert moments_result_dtype(6).itemsize == 448
assert moments_result_dtype(17).itemsize == 2736


def build():
    """compile the oracle (gcc); called by __graft_entry__.build()"""
    subprocess.run(["make", "-C", _HERE, "libngmix_oracle.so"], check=True,
                   capture_output=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIBPATH):
            build()
        L = ctypes.CDLL(_LIBPATH)
        L.ora_fexp.restype = ctypes.c_double
        L.ora_fexp.argtypes = [ctypes.c_double]
        L.ora_apod_window.restype = ctypes.c_double
        L.ora_apod_window.argtypes = [ctypes.c_double]
        L.ora_apod_window_deriv.restype = ctypes.c_double
        L.ora_apod_window_deriv.argtypes = [ctypes.c_double]
        L.ora_gmix_eval_pixel_fast.restype = ctypes.c_double
        L.ora_gmix_eval_pixel.restype = ctypes.c_double
        _lib = L
    return _lib


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


def _i64(x):
    return ctypes.c_int64(int(x))


def _d(x):
    return ctypes.c_double(float(x))


def _check_c(a, dtype=None):
    assert a.flags["C_CONTIGUOUS"]
    if dtype is not None:
        assert a.dtype == dtype, (a.dtype, dtype)


def fexp(x):
    x = np.ascontiguousarray(x, dtype="f8")
    out = np.empty_like(x)
    lib().ora_fexp_array(_p(x), _p(out), _i64(x.size))
    return out


def apod(chi2):
    chi2 = np.ascontiguousarray(chi2, dtype="f8")
    w = np.empty_like(chi2)
    dw = np.empty_like(chi2)
    lib().ora_apod_array(_p(chi2), _p(w), _p(dw), _i64(chi2.size))
    return w, dw


def gmix_set_norms(gmix):
    _check_c(gmix, GAUSS2D_DTYPE)
    return lib().ora_gmix_set_norms(_p(gmix), _i64(gmix.size))


def g1g2_to_e1e2(g1, g2):
    e1 = ctypes.c_double()
    e2 = ctypes.c_double()
    st = lib().ora_g1g2_to_e1e2(_d(g1), _d(g2), ctypes.byref(e1),
                                ctypes.byref(e2))
    return st, e1.value, e2.value


def get_cm_Tfactor(fracdev, TdByTe):
    out = ctypes.c_double()
    st = lib().ora_get_cm_Tfactor(_d(fracdev), _d(TdByTe), ctypes.byref(out))
    return st, out.value


def gmix_fill(gmix, pars, model, fracdev=0.0, TdByTe=0.0, Tfactor=0.0):
    _check_c(gmix, GAUSS2D_DTYPE)
    pars = np.ascontiguousarray(pars, dtype="f8")
    return lib().ora_gmix_fill(_p(gmix), _i64(gmix.size), _p(pars),
                               _i64(pars.size), ctypes.c_int(MODEL_NUMS[model]),
                               _d(fracdev), _d(TdByTe), _d(Tfactor))


def gmix_convolve_fill(out, gmix, psf):
    _check_c(out, GAUSS2D_DTYPE)
    _check_c(gmix, GAUSS2D_DTYPE)
    _check_c(psf, GAUSS2D_DTYPE)
    assert out.size == gmix.size * psf.size
    return lib().ora_gmix_convolve_fill(_p(out), _p(gmix), _i64(gmix.size),
                                        _p(psf), _i64(psf.size))


def jacobian_get_vu(jacob, row, col):
    v = ctypes.c_double()
    u = ctypes.c_double()
    lib().ora_jacobian_get_vu(_p(jacob), _d(row), _d(col), ctypes.byref(v),
                              ctypes.byref(u))
    return v.value, u.value


def jacobian_get_rowcol(jacob, v, u):
    row = ctypes.c_double()
    col = ctypes.c_double()
    st = lib().ora_jacobian_get_rowcol(_p(jacob), _d(v), _d(u),
                                       ctypes.byref(row), ctypes.byref(col))
    return st, row.value, col.value


def fill_pixels(pixels, image, weight, jacob, ignore_zero_weight=True):
    _check_c(pixels, PIXEL_DTYPE)
    image = np.ascontiguousarray(image, dtype="f8")
    weight = np.ascontiguousarray(weight, dtype="f8")
    nrow, ncol = image.shape
    return lib().ora_fill_pixels(_p(pixels), _i64(pixels.size), _p(image),
                                 _p(weight), _i64(nrow), _i64(ncol), _p(jacob),
                                 ctypes.c_int(int(ignore_zero_weight)))


def make_pixels(image, weight, jacob, ignore_zero_weight=True):
    """ngmix/pixels/pixels.py:6-52 restated around ora_fill_pixels"""
    if ignore_zero_weight:
        npixels = int((np.asarray(weight) > 0.0).sum())
    else:
        npixels = np.asarray(image).size
    pixels = np.zeros(npixels, dtype=PIXEL_DTYPE)
    st = fill_pixels(pixels, image, weight, jacob, ignore_zero_weight)
    assert st == OK, st
    return pixels


def fill_coords(coords, nrow, ncol, jacob):
    _check_c(coords, COORD_DTYPE)
    lib().ora_fill_coords(_p(coords), _i64(nrow), _i64(ncol), _p(jacob))


def make_coords(dims, jacob):
    nrow, ncol = dims
    coords = np.zeros(nrow * ncol, dtype=COORD_DTYPE)
    fill_coords(coords, nrow, ncol, jacob)
    return coords


def render(gmix, coords, image, fast_exp=0):
    _check_c(gmix, GAUSS2D_DTYPE)
    _check_c(coords, COORD_DTYPE)
    _check_c(image)
    assert image.dtype == np.float64 and image.size == coords.size
    return lib().ora_render(_p(gmix), _i64(gmix.size), _p(coords),
                            _i64(coords.size), _p(image),
                            ctypes.c_int(int(fast_exp)))


def get_loglike(gmix, pixels):
    _check_c(gmix, GAUSS2D_DTYPE)
    _check_c(pixels, PIXEL_DTYPE)
    ll = ctypes.c_double()
    sn = ctypes.c_double()
    sd = ctypes.c_double()
    npix = ctypes.c_int64()
    st = lib().ora_get_loglike(_p(gmix), _i64(gmix.size), _p(pixels),
                               _i64(pixels.size), ctypes.byref(ll),
                               ctypes.byref(sn), ctypes.byref(sd),
                               ctypes.byref(npix))
    return st, (ll.value, sn.value, sd.value, npix.value)


def fill_fdiff(gmix, pixels, fdiff, start=0):
    _check_c(gmix, GAUSS2D_DTYPE)
    _check_c(pixels, PIXEL_DTYPE)
    _check_c(fdiff)
    assert fdiff.dtype == np.float64 and fdiff.size >= start + pixels.size
    return lib().ora_fill_fdiff(_p(gmix), _i64(gmix.size), _p(pixels),
                                _i64(pixels.size), _p(fdiff), _i64(start))


def get_model_s2n_sum(gmix, pixels):
    out = ctypes.c_double()
    st = lib().ora_get_model_s2n_sum(_p(gmix), _i64(gmix.size), _p(pixels),
                                     _i64(pixels.size), ctypes.byref(out))
    return st, out.value


def get_weighted_sums(wt, pixels, res, maxrad):
    """res: 1-element array of moments_result_dtype(6 or 17); accumulates"""
    _check_c(wt, GAUSS2D_DTYPE)
    _check_c(pixels, PIXEL_DTYPE)
    nmom = res.dtype["sums"].shape[0]
    return lib().ora_get_weighted_sums(_p(wt), _i64(wt.size), _p(pixels),
                                       _i64(pixels.size), _p(res),
                                       ctypes.c_int(nmom), _d(maxrad))


def admom(conf, wt, pixels, res):
    _check_c(conf, ADMOM_CONF_DTYPE)
    _check_c(wt, GAUSS2D_DTYPE)
    _check_c(pixels, PIXEL_DTYPE)
    _check_c(res, ADMOM_RESULT_DTYPE)
    return lib().ora_admom(_p(conf), _p(wt), _p(pixels), _i64(pixels.size),
                           _p(res))


def em_run(kind, conf, pixels, sums, gmix, gmix_psf, gmix_conv,
           fill_zero_weight=False):
    """kind: 0 em_run, 1 fixcen, 2 fixcov, 3 fluxonly.
    Returns (status, numiter, frac_diff, sky)."""
    _check_c(conf, EM_CONF_DTYPE)
    _check_c(pixels, PIXEL_DTYPE)
    _check_c(gmix, GAUSS2D_DTYPE)
    _check_c(gmix_psf, GAUSS2D_DTYPE)
    _check_c(gmix_conv, GAUSS2D_DTYPE)
    assert sums.flags["C_CONTIGUOUS"]
    assert sums.nbytes == 8 * EM_SUMS_NDOUBLE[kind] * gmix.size
    numiter = ctypes.c_int32()
    frac = ctypes.c_double()
    sky = ctypes.c_double()
    st = lib().ora_em_run(ctypes.c_int(kind), _p(conf), _p(pixels),
                          _i64(pixels.size), _p(sums), _p(gmix),
                          _i64(gmix.size), _p(gmix_psf), _i64(gmix_psf.size),
                          _p(gmix_conv), ctypes.c_int(int(fill_zero_weight)),
                          ctypes.byref(numiter), ctypes.byref(frac),
                          ctypes.byref(sky))
    return st, numiter.value, frac.value, sky.value


def deriv_images(gpars, dcov, vv, uu, area, out):
    gpars = np.ascontiguousarray(gpars, dtype="f8")
    dcov = np.ascontiguousarray(dcov, dtype="f8")
    vv = np.ascontiguousarray(vv, dtype="f8")
    uu = np.ascontiguousarray(uu, dtype="f8")
    area = np.ascontiguousarray(area, dtype="f8")
    _check_c(out)
    assert out.shape == (6, vv.size) and out.dtype == np.float64
    lib().ora_deriv_images(_p(gpars), _p(dcov), _i64(gpars.shape[0]), _p(vv),
                           _p(uu), _p(area), _i64(vv.size), _p(out))


def num_threads():
    return lib().ora_num_threads()


def render_loglike_batch(gm_all, pixels_all, coords_all, images_all, nthreads):
    """cpu_baseline leg: gm_all (nstamps, ng) gauss2d with norms set;
    pixels_all / coords_all (nstamps, npix); images_all (nstamps, npix) f8,
    accumulated into.  Returns loglike per stamp."""
    nstamps, ng = gm_all.shape
    npix = pixels_all.shape[1]
    out = np.zeros(nstamps)
    lib().ora_render_loglike_batch(_p(gm_all), _i64(ng), _p(pixels_all),
                                   _p(coords_all), _i64(npix), _p(images_all),
                                   _i64(nstamps), _p(out),
                                   ctypes.c_int(int(nthreads)))
    return out


def loglike_batch(gm_all, pixels_all, nthreads):
    """cpu_baseline leg (config 5): gm_all (nstamps, ng) gauss2d with norms
    set; pixels_all (nstamps, npix).  Returns loglike per stamp."""
    _check_c(gm_all, GAUSS2D_DTYPE)
    _check_c(pixels_all, PIXEL_DTYPE)
    nstamps, ng = gm_all.shape
    out = np.zeros(nstamps)
    lib().ora_loglike_batch(_p(gm_all), _i64(ng), _p(pixels_all),
                            _i64(pixels_all.shape[1]), _i64(nstamps), _p(out),
                            ctypes.c_int(int(nthreads)))
    return out


def admom_batch(conf, wt_all, pixels_all, res_all, nthreads):
    """cpu_baseline leg: one admom per stamp; wt_all (n,) gauss2d guesses
    (updated), pixels_all (n, npix), res_all (n,) result records"""
    _check_c(conf, ADMOM_CONF_DTYPE)
    _check_c(wt_all, GAUSS2D_DTYPE)
    _check_c(pixels_all, PIXEL_DTYPE)
    _check_c(res_all, ADMOM_RESULT_DTYPE)
    n, npix = pixels_all.shape
    lib().ora_admom_batch(_p(conf), _p(wt_all), _p(pixels_all), _i64(npix), _i64(n),
                          _p(res_all), ctypes.c_int(int(nthreads)))


def em_batch(conf, pixels_all, gmix_all, psf_all, conv_all, nthreads):
    """cpu_baseline leg: one em_run (kind 0) per stamp; gmix_all (n, ng),
    psf_all (n, npsf), conv_all (n, ng*npsf); returns (numiter, status)"""
    _check_c(conf, EM_CONF_DTYPE)
    _check_c(pixels_all, PIXEL_DTYPE)
    _check_c(gmix_all, GAUSS2D_DTYPE)
    _check_c(psf_all, GAUSS2D_DTYPE)
    _check_c(conv_all, GAUSS2D_DTYPE)
    n, npix = pixels_all.shape
    numiter = np.zeros(n, dtype=np.int32)
    status = np.zeros(n, dtype=np.int32)
    lib().ora_em_batch(_p(conf), _p(pixels_all), _i64(npix), _i64(n), _p(gmix_all),
                       _i64(gmix_all.shape[1]), _p(psf_all), _i64(psf_all.shape[1]),
                       _p(conv_all), _p(numiter), _p(status),
                       ctypes.c_int(int(nthreads)))
    return numiter, status
