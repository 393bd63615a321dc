"""
ctypes binding of libngmix_hip.so (include/ngmix_hip.h).

The HIP library IS the product: there is no CPU fallback.  If the shared
object is missing we try to build it in-tree with hipcc (same image on the
GPU box); if that fails, importing any compute entry point raises.
"""
import ctypes
import os
import subprocess

import numpy as np

from . import gexceptions

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
# NGMIX_HIP_LIB selects another build of the same library (A/B kernel timing)
LIB_PATH = os.environ.get("NGMIX_HIP_LIB", os.path.join(_HERE, "libngmix_hip.so"))

# ---- status codes (include/ngmix_hip.h) ----
OK = 0
ERR_DET_TOO_LOW = 1
ERR_T_TOO_LOW = 2
ERR_G_RANGE = 3
ERR_GTOT_ZERO = 4
ERR_ELOGL_ZERO = 5
ERR_ZERO_DIV = 6
ERR_PIXELS_NOT_FILLED = 7
ERR_HIP = -100
ERR_BAD_ARG = -101

_RANGE_MESSAGES = {
    ERR_DET_TOO_LOW: "det too low",
    ERR_T_TOO_LOW: "T too low",
    ERR_G_RANGE: "g >= 1",
    ERR_GTOT_ZERO: "gtot == 0",
    ERR_ELOGL_ZERO: "elogL == 0",
}

STAMP_IGNORE_ZERO_WEIGHT = 1
BATCH_NO_SKIP = 1
BATCH_EXACT = 2
BATCH_TRACKED_LOADS = 4
BATCH_RENDER_OVERWRITE = 8

# ---- record layouts = the reference's numpy dtypes ----
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
    ("no_cov", bool),   # batch extension in the reference record's padding
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
STAMP_DTYPE = np.dtype([
    ("pix_off", "i8"), ("nrow", "i4"), ("ncol", "i4"), ("gm_off", "i4"),
    ("ngauss", "i4"), ("flags", "i4"), ("npix_kept", "i4"),
])
EM_SUMS_NDOUBLE = {0: 14, 1: 10, 2: 8, 3: 2}

assert GAUSS2D_DTYPE.itemsize == 104 and PIXEL_DTYPE.itemsize == 48
assert ADMOM_CONF_DTYPE.itemsize == 40 and ADMOM_RESULT_DTYPE.itemsize == 584
assert EM_CONF_DTYPE.itemsize == 32 and STAMP_DTYPE.itemsize == 32


def moments_result_dtype(nmom):
    """get_moments_result_dtype, ngmix/gmix/gmix.py:1314-1330 (align=True)"""
    return np.dtype([
        ("flags", "i4"), ("npix", "i4"), ("wsum", "f8"),
        ("sums", "f8", nmom), ("sums_cov", "f8", (nmom, nmom)),
        ("pars", "f8", nmom), ("F", "f8", nmom),
    ], align=True)


LM_NPMAX = 14
LM_PRECISE_MIN_NLOC = 10  # ngmix_lm_precise_cov_batch serves nloc >= this
LM_NPARS_GENERIC = 255   # ngmix_lm_advance_batch: asks for the generic step
LM_NSUM = 28
LM_PHASE_DONE = 2
LM_PHASE_JAC = 3
LM_MODE_ANALYTIC = 0
LM_MODE_FD = 1
LM_MODE_ANALYTIC_LAZY = 2
# ngmix_lm_state (include/ngmix_hip.h): one re-entrant lmder iteration
LM_STATE_DTYPE = np.dtype([
    ("x", "f8", LM_NPMAX), ("xt", "f8", LM_NPMAX), ("diag", "f8", LM_NPMAX),
    ("R", "f8", (LM_NPMAX, LM_NPMAX)), ("qtf", "f8", LM_NPMAX),
    ("step", "f8", LM_NPMAX),
    ("fnorm", "f8"), ("xnorm", "f8"), ("delta", "f8"), ("par", "f8"),
    ("gnorm", "f8"), ("pnorm", "f8"),
    ("ftol", "f8"), ("xtol", "f8"), ("gtol", "f8"), ("factor", "f8"),
    ("xi", "f8", LM_NPMAX), ("xti", "f8", LM_NPMAX),
    ("lo", "f8", LM_NPMAX), ("hi", "f8", LM_NPMAX),
    ("xstep", "f8", LM_NPMAX), ("hstep", "f8", LM_NPMAX),
    ("ipvt", "i4", LM_NPMAX),
    ("n", "i4"), ("iter", "i4"), ("nfev", "i4"), ("njev", "i4"), ("info", "i4"),
    ("phase", "i4"), ("maxfev", "i4"), ("mode", "i4"),
    ("bounded", "i4"), ("fonly", "i4"),
], align=True)


PRIOR_FLAT = 0
PRIOR_TWO_SIDED_ERF = 1
PRIOR_NORMAL = 2
PRIOR_LOGNORMAL = 3
PRIOR_TRUNCATED_GAUSSIAN = 4
PRIOR_MAXBAND = 3
PRIOR_MAXMID = 2
PRIOR_ROWS_LNPROB = 0
PRIOR_ROWS_FDIFF = 1
# ngmix_simple_sep_prior (include/ngmix_hip.h)
SIMPLE_SEP_PRIOR_DTYPE = np.dtype([
    ("cen1", "f8"), ("cen2", "f8"), ("cen_s2inv1", "f8"), ("cen_s2inv2", "f8"),
    ("g_sig2inv", "f8"), ("T_par", "f8", 4), ("F_par", "f8", (PRIOR_MAXBAND, 4)),
    ("T_kind", "i4"), ("nband", "i4"), ("F_kind", "i4", PRIOR_MAXBAND), ("nmid", "i4"),
    ("cen_sinv1", "f8"), ("cen_sinv2", "f8"), ("mid_par", "f8", (PRIOR_MAXMID, 4)),
    ("mid_kind", "i4", PRIOR_MAXMID), ("rows_mode", "i4"), ("pad_", "i4"),
], align=True)
assert SIMPLE_SEP_PRIOR_DTYPE.itemsize == 288


class Batch(ctypes.Structure):
    """ngmix_batch: host struct of device pointers"""
    _fields_ = [
        ("nstamps", ctypes.c_int64),
        ("stamps", ctypes.c_void_p),
        ("val", ctypes.c_void_p),
        ("ierr", ctypes.c_void_p),
        ("jac", ctypes.c_void_p),
        ("max_ngauss", ctypes.c_int32),
        ("max_npix", ctypes.c_int32),
        ("any_masked", ctypes.c_int32),
        ("flags", ctypes.c_int32),
        ("max_nrow", ctypes.c_int32),
        ("max_ncol", ctypes.c_int32),
    ]


class LMProblem(ctypes.Structure):
    """ngmix_lm_problem: the arguments of one lock-step round
    (ngmix_lm_rounds_batch)"""
    _fields_ = [
        ("batch", ctypes.POINTER(Batch)),
        ("states", ctypes.c_void_p),
        ("nobj", ctypes.c_int64),
        ("stamp_obj", ctypes.c_void_p),
        ("stamp_band", ctypes.c_void_p),
        ("obj_start", ctypes.c_void_p),
        ("psf", ctypes.c_void_p),
        ("sums", ctypes.c_void_p),
        ("status", ctypes.c_void_p),
        ("stamp_stats", ctypes.c_void_p),
        ("obj_stats", ctypes.c_void_p),
        ("prior", ctypes.c_void_p),
        ("obj_sums", ctypes.c_void_p),
        ("prior_step", ctypes.c_double),
        ("model", ctypes.c_int32),
        ("fd", ctypes.c_int32),
        ("npsf", ctypes.c_int32),
        ("nloc_npars", ctypes.c_int32),
        ("jac_point", ctypes.c_void_p),
    ]


_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_i32 = ctypes.c_int
_f64 = ctypes.c_double
_sz = ctypes.c_size_t
_pd = ctypes.POINTER(ctypes.c_double)
_pb = ctypes.POINTER(Batch)

# name -> (restype, argtypes); every symbol include/ngmix_hip.h declares
SIGNATURES = {
    "ngmix_version": (ctypes.c_char_p, []),
    "ngmix_last_error": (ctypes.c_char_p, []),
    "ngmix_device_count": (_i32, []),
    "ngmix_set_device": (_i32, [_i32]),
    "ngmix_device_malloc": (_i32, [ctypes.POINTER(_vp), _sz]),
    "ngmix_device_free": (_i32, [_vp]),
    "ngmix_memcpy_h2d": (_i32, [_vp, _vp, _sz, _vp]),
    "ngmix_memcpy_d2h": (_i32, [_vp, _vp, _sz, _vp]),
    "ngmix_memset_device": (_i32, [_vp, _i32, _sz, _vp]),
    "ngmix_stream_synchronize": (_i32, [_vp]),
    # seam forms
    "ngmix_set_norms": (_i32, [_vp, _i64]),
    "ngmix_fill_model": (_i32, [_vp, _i64, _i32, _vp, _i64]),
    "ngmix_fill_cm": (_i32, [_vp, _f64, _f64, _f64, _vp]),
    "ngmix_get_cm_Tfactor": (_i32, [_f64, _f64, _pd]),
    "ngmix_g1g2_to_e1e2": (_i32, [_f64, _f64, _pd, _pd]),
    "ngmix_convolve_fill": (_i32, [_vp, _vp, _i64, _vp, _i64]),
    "ngmix_jacobian_get_vu": (None, [_vp, _f64, _f64, _pd, _pd]),
    "ngmix_jacobian_get_rowcol": (_i32, [_vp, _f64, _f64, _pd, _pd]),
    "ngmix_fill_pixels": (_i32, [_vp, _i64, _vp, _vp, _i64, _i64, _vp, _i32]),
    "ngmix_fill_coords": (_i32, [_vp, _i64, _i64, _vp]),
    "ngmix_render": (_i32, [_vp, _i64, _vp, _i64, _vp, _i32]),
    "ngmix_get_loglike": (_i32, [_vp, _i64, _vp, _i64, _pd, _pd, _pd,
                                 ctypes.POINTER(_i64)]),
    "ngmix_fill_fdiff": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64]),
    "ngmix_get_model_s2n_sum": (_i32, [_vp, _i64, _vp, _i64, _pd]),
    "ngmix_get_weighted_sums": (_i32, [_vp, _i64, _vp, _i64, _vp, _i32, _f64]),
    "ngmix_admom": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "ngmix_em_run": (_i32, [_i32, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64,
                            _vp, _i32, ctypes.POINTER(ctypes.c_int32), _pd, _pd]),
    "ngmix_deriv_images": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    # batch forms
    "ngmix_weight_to_ierr_batch": (_i32, [_vp, _vp, _i64, _vp]),
    "ngmix_count_kept_batch": (_i32, [_vp, _i64, _vp, _vp]),
    "ngmix_template_sums_batch": (_i32, [_vp, _vp, _vp, _vp, _vp]),
    "ngmix_fill_model_batch": (_i32, [_vp, _i64, _i32, _i32, _vp, _i32, _vp,
                                      _vp, _vp]),
    "ngmix_convolve_fill_batch": (_i32, [_vp, _vp, _i32, _vp, _i32, _i64, _vp,
                                         _vp]),
    "ngmix_set_norms_batch": (_i32, [_vp, _i32, _i64, _vp, _vp]),
    "ngmix_loglike_batch": (_i32, [_pb, _vp, _vp, _vp, _vp]),
    "ngmix_fill_fdiff_batch": (_i32, [_pb, _vp, _vp, _vp, _vp, _vp]),
    "ngmix_render_batch": (_i32, [_pb, _vp, _vp, _i32, _vp, _vp]),
    "ngmix_model_s2n_sum_batch": (_i32, [_pb, _vp, _vp, _vp, _vp]),
    "ngmix_weighted_sums_batch": (_i32, [_pb, _vp, _vp, _i32, _vp, _vp, _vp]),
    "ngmix_admom_batch": (_i32, [_vp, _pb, _vp, _vp, _vp, _vp]),
    "ngmix_em_batch": (_i32, [_i32, _vp, _pb, _vp, _i32, _vp, _i32, _vp, _vp,
                              _i32, _vp, _vp, _vp]),
    "ngmix_deriv_images_batch": (_i32, [_pb, _vp, _vp, _vp, _vp, _vp]),
    # library-owned stamp store and the RCCL gather of result records
    "ngmix_batch_create": (_i32, [ctypes.POINTER(_pb), _i64, _vp, _vp, _i32, _i32]),
    "ngmix_batch_upload": (_i32, [_pb, _vp, _vp, _vp, _vp]),
    "ngmix_batch_npix_kept": (_i32, [_pb, _vp]),
    "ngmix_batch_free": (_i32, [_pb]),
    "ngmix_comm_unique_id": (_i32, [_vp]),
    "ngmix_comm_init_rank": (_i32, [ctypes.POINTER(_vp), _i32, _vp, _i32]),
    "ngmix_comm_destroy": (_i32, [_vp]),
    "ngmix_allgather_results": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp]),
    # batched Levenberg-Marquardt
    "ngmix_abi_sizeof": (_i64, [ctypes.c_char_p]),
    "ngmix_lm_init": (_i32, [_vp, _i64, _i32, _vp, _f64, _f64, _f64, _i32, _f64, _i32,
                              _vp, _vp]),
    "ngmix_lm_init_batch": (_i32, [_vp, _i64, _i32, _vp, _f64, _f64, _f64, _i32, _f64,
                                    _i32, _vp, _vp, _vp]),
    "ngmix_lm_prior_sums_batch": (_i32, [_vp, _i64, _vp, _f64, _vp, _vp]),
    "ngmix_simple_sep_prior_eval": (_i32, [_vp, _vp, _vp, _vp]),
    "ngmix_lm_advance_host": (_i64, [_vp, _i64, _vp, _vp, _vp]),
    "ngmix_lm_prior_sums_host": (_i32, [_vp, _i64, _vp, _f64, _vp]),
    "ngmix_fastexp_batch": (_i32, [_vp, _vp, _i64, _i32, _vp]),
    "ngmix_prepsf_sums_batch": (_i32, [_vp] * 8 + [_f64] + [_vp] * 7 + [_i64, _i32, _i64, _i64, _i32,
                                                               _i32, _f64, _f64, _vp, _vp]),
    "ngmix_lm_prior_finish_batch": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "ngmix_first_pixels_fdiff2_batch": (_i32, [_pb, _vp, _vp, _i32, _i64, _i32, _vp, _vp]),
    "ngmix_lm_eval_batch": (_i32, [_pb, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp,
                                   _vp, _vp, _vp]),
    "ngmix_lm_advance_batch": (_i32, [_vp, _i64, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp,
                                      _vp]),
    "ngmix_lm_finalize_batch": (_i32, [_vp, _i64, _vp, _vp, _f64, _f64, _vp, _vp]),
    "ngmix_launch_census": (_i64, [ctypes.c_char_p, _i64, _i32]),
    "ngmix_lm_pack_batch": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ngmix_lm_rounds_batch": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp]),
    "ngmix_lm_precise_cov_batch": (_i32, [_vp, _vp, _vp]),
    "ngmix_events_create": (_i32, [_i32, _vp]),
    "ngmix_events_destroy": (_i32, [_i32, _vp]),
    "ngmix_event_record": (_i32, [_vp, _vp]),
    "ngmix_event_synchronize": (_i32, [_vp]),
    "ngmix_event_elapsed_ms": (_i32, [_vp, _vp, _vp]),
}
LM_NCOLS = 12


def _makefile_list(name):
    """the file list `name = ...` of csrc/Makefile (SRCS, HDRS, UNIT_HDRS)"""
    for line in open(os.path.join(_CSRC, "Makefile")):
        if line.startswith(name + " ="):
            return line.split("=", 1)[1].split()
    raise RuntimeError("csrc/Makefile has no %s" % name)


def source_hash():
    """sha256 of the kernel sources in the Makefile's order: `make` leaves the
    same digest next to the library (libngmix_hip.so.srchash)"""
    import hashlib
    h = hashlib.sha256()
    for name in _makefile_list("SRCS") + _makefile_list("HDRS") + _makefile_list("UNIT_HDRS"):
        with open(os.path.join(_CSRC, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def library_is_stale():
    """True when the in-tree library was built from other sources than the
    ones in csrc/ now (an edit without a rebuild); a library selected with
    NGMIX_HIP_LIB is taken as it is"""
    if "NGMIX_HIP_LIB" in os.environ:
        return False
    try:
        with open(LIB_PATH + ".srchash") as f:
            return f.read().strip() != source_hash()
    except OSError:
        return True


def build(verbose=False, force=False):
    """compile libngmix_hip.so in-tree for gfx950 (hipcc)"""
    jobs = max(2, min(8, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity")
                      else (os.cpu_count() or 4)))
    res = subprocess.run(["make", "-C", _CSRC, "-j%d" % jobs] + (["-B"] if force else []),
                         capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0:
        raise RuntimeError("building libngmix_hip.so failed")


_lib = None


class _build_lock(object):
    """exclusive advisory lock (fcntl.flock) on csrc/.build.lock around the
    staleness check + make: concurrent forced rebuilds would write the same
    .o / .so.tmp files"""

    def __enter__(self):
        import fcntl
        self._f = open(os.path.join(_CSRC, ".build.lock"), "w")
        fcntl.flock(self._f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl
        fcntl.flock(self._f, fcntl.LOCK_UN)
        self._f.close()
        return False


def lib():
    """load the HIP library; raises (never falls back) if unavailable"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH) or library_is_stale():
            # never run kernels that do not correspond to the sources.  The
            # ranks of one job all get here together (launch_local_ranks):
            # one of them builds under the lock, the others find it done
            with _build_lock():
                if not os.path.exists(LIB_PATH):
                    build()
                elif library_is_stale():
                    build(force=True)
                if library_is_stale():
                    raise RuntimeError("libngmix_hip.so does not match ngmix_amd/csrc "
                                       "and could not be rebuilt")
        # One HIP runtime per process: PyTorch-ROCm ships its own
        # libamdhip64; importing torch first makes our library bind to that
        # copy (same SONAME) instead of initialising a second runtime that
        # cannot see the devices the first one opened.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if a symbol is missing
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = L
    return _lib


def launch_census(reset=False):
    """{kernel variant: launches so far in this process} (ngmix_launch_census)"""
    L = lib()
    n = L.ngmix_launch_census(None, 0, 0)
    buf = ctypes.create_string_buffer(int(n) + 64)
    L.ngmix_launch_census(buf, len(buf), 1 if reset else 0)
    out = {}
    for line in buf.value.decode().splitlines():
        name, _, count = line.rpartition("\t")
        out[name] = int(count)
    return out


def last_error():
    return lib().ngmix_last_error().decode()


def check(status, context=""):
    """map a C-ABI status onto the exception the reference would raise"""
    if status == OK:
        return
    if status in _RANGE_MESSAGES:
        raise gexceptions.GMixRangeError(_RANGE_MESSAGES[status])
    if status == ERR_ZERO_DIV:
        raise ZeroDivisionError("division by zero")
    if status == ERR_PIXELS_NOT_FILLED:
        raise RuntimeError("some pixels were not filled")
    if status == ERR_BAD_ARG:
        raise ValueError("%s: bad argument: %s" % (context, last_error()))
    raise RuntimeError("%s: HIP failure (%d): %s" % (context, status, last_error()))


def ptr(a):
    """address of a C-contiguous numpy array"""
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("array must be C contiguous")
    return ctypes.c_void_p(a.ctypes.data)
