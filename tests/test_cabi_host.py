"""
CPU-only checks of the C ABI: the library loads, exports every symbol
include/ngmix_hip.h declares, and its O(ngauss) HOST parameter-prep entry
points (norms, fills, convolve, jacobian transforms) reproduce the reference
goldens.  No kernel is launched here.
"""
import ctypes
import os
import re

import numpy as np
import pytest

from ngmix_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GFIELDS = ("p", "row", "col", "irr", "irc", "icc", "det")
NFIELDS = ("drr", "drc", "dcc", "norm", "pnorm")


def as_gauss(a):
    out = np.zeros(a.size, dtype=_lib.GAUSS2D_DTYPE)
    for n in _lib.GAUSS2D_DTYPE.names:
        out[n] = a[n]
    return out


def test_every_declared_symbol_is_exported_and_bound():
    header = open(os.path.join(ROOT, "include", "ngmix_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(ngmix_[A-Za-z0-9_]+)\s*\(", header))
    assert len(declared) >= 41
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), "library does not export " + name
    assert declared == set(_lib.SIGNATURES), (
        declared ^ set(_lib.SIGNATURES))
    L = _lib.lib()
    assert b"gfx950" in L.ngmix_version()


def test_struct_layouts_match_reference():
    """SURVEY.md 8b, measured from the reference's dtypes"""
    d = _lib.GAUSS2D_DTYPE
    assert [d.fields[n][1] for n in ("p", "det", "norm_set", "drr", "pnorm")] == \
        [0, 48, 56, 64, 96]
    d = _lib.PIXEL_DTYPE
    assert [d.fields[n][1] for n in ("u", "v", "area", "val", "ierr", "fdiff")] == \
        [0, 8, 16, 24, 32, 40]
    d = _lib.ADMOM_RESULT_DTYPE
    assert [d.fields[n][1] for n in ("flags", "numiter", "npix", "wsum", "sums",
                                     "sums_cov", "pars", "rho4", "F")] == \
        [0, 4, 8, 16, 24, 80, 472, 520, 528]
    d = _lib.ADMOM_CONF_DTYPE
    # the reference's five fields where the reference has them; 'no_cov' is the
    # batch extension living in its record's padding (itemsize unchanged: 40)
    assert [d.fields[n][1] for n in d.names] == [0, 8, 16, 24, 32, 33] and d.itemsize == 40
    d = _lib.EM_CONF_DTYPE
    assert [d.fields[n][1] for n in d.names] == [0, 8, 12, 16, 24]
    assert _lib.moments_result_dtype(6).itemsize == 448
    assert _lib.moments_result_dtype(17).itemsize == 2736


def test_binding_record_sizes_match_the_library():
    """the numpy / ctypes mirrors of every ABI record against sizeof() in the
    library itself (ngmix_abi_sizeof)"""
    import ctypes
    L = _lib.lib()
    sizes = {
        "ngmix_gauss2d": _lib.GAUSS2D_DTYPE.itemsize,
        "ngmix_pixel": _lib.PIXEL_DTYPE.itemsize,
        "ngmix_coord": _lib.COORD_DTYPE.itemsize,
        "ngmix_jacobian": _lib.JACOBIAN_DTYPE.itemsize,
        "ngmix_admom_conf": _lib.ADMOM_CONF_DTYPE.itemsize,
        "ngmix_admom_result": _lib.ADMOM_RESULT_DTYPE.itemsize,
        "ngmix_em_conf": _lib.EM_CONF_DTYPE.itemsize,
        "ngmix_stamp": _lib.STAMP_DTYPE.itemsize,
        "ngmix_batch": ctypes.sizeof(_lib.Batch),
        "ngmix_lm_state": _lib.LM_STATE_DTYPE.itemsize,
        "ngmix_simple_sep_prior": _lib.SIMPLE_SEP_PRIOR_DTYPE.itemsize,
        "ngmix_lm_problem": ctypes.sizeof(_lib.LMProblem),
    }
    for name, size in sizes.items():
        assert L.ngmix_abi_sizeof(name.encode()) == size, name
    assert L.ngmix_abi_sizeof(b"no_such_record") == -1


FILL_CASES = ["gauss", "exp", "dev", "turb", "bdf", "bd", "coellip", "full",
              "exp_round", "exp_highg"]
MODEL_IDS = {"full": 0, "gauss": 1, "turb": 2, "exp": 3, "dev": 4, "bdf": 6,
             "coellip": 7, "cm": 9, "bd": 10}


@pytest.mark.parametrize("name", FILL_CASES)
def test_host_fill_convolve_norms(golden, name):
    g = golden("fills")
    L = _lib.lib()
    model = name.split("_")[0]
    ref = g["gmix_" + name]
    pars = np.ascontiguousarray(g["pars_" + name])
    gm = np.zeros(ref.size, dtype=_lib.GAUSS2D_DTYPE)
    assert L.ngmix_fill_model(_lib.ptr(gm), gm.size, MODEL_IDS[model],
                              _lib.ptr(pars), pars.size) == 0
    for n in GFIELDS:
        np.testing.assert_allclose(gm[n], ref[n], rtol=1e-14, atol=0)
    assert np.all(gm["norm_set"] == 0) and np.all(np.isnan(gm["pnorm"]))
    gm = as_gauss(ref)
    for pname in ("psf1", "psf3", "psf_off"):
        psf = as_gauss(g[pname])
        out = np.zeros(gm.size * psf.size, dtype=_lib.GAUSS2D_DTYPE)
        assert L.ngmix_convolve_fill(_lib.ptr(out), _lib.ptr(gm), gm.size,
                                     _lib.ptr(psf), psf.size) == 0
        refc = g["conv_%s_%s" % (name, pname)]
        for n in GFIELDS:
            np.testing.assert_array_equal(out[n], refc[n])
        refn = g["convnorm_%s_%s" % (name, pname)]
        st = L.ngmix_set_norms(_lib.ptr(out), out.size)
        if np.all(refn["norm_set"] == 1):
            assert st == 0
            for n in GFIELDS + NFIELDS:
                np.testing.assert_array_equal(out[n], refn[n])


def test_host_cm_shape_errors(golden):
    g = golden("fills")
    L = _lib.lib()
    tf = ctypes.c_double()
    assert L.ngmix_get_cm_Tfactor(float(g["cm_fracdev"]), float(g["cm_TdByTe"]),
                                  ctypes.byref(tf)) == 0
    assert tf.value == float(g["cm_Tfactor"])
    gm = np.zeros(16, dtype=_lib.GAUSS2D_DTYPE)
    pars = np.ascontiguousarray(g["pars_exp"])
    assert L.ngmix_fill_cm(_lib.ptr(gm), float(g["cm_fracdev"]),
                           float(g["cm_TdByTe"]), tf.value, _lib.ptr(pars)) == 0
    for n in GFIELDS:
        np.testing.assert_allclose(gm[n], g["gmix_cm"][n], rtol=1e-14, atol=0)
    e1, e2 = ctypes.c_double(), ctypes.c_double()
    assert L.ngmix_g1g2_to_e1e2(0.8, 0.7, ctypes.byref(e1),
                                ctypes.byref(e2)) == _lib.ERR_G_RANGE
    bad = np.array([0.0, 0.0, 0.9, 0.9, 1.0, 1.0])
    gm6 = np.zeros(6, dtype=_lib.GAUSS2D_DTYPE)
    assert L.ngmix_fill_model(_lib.ptr(gm6), 6, 3, _lib.ptr(bad), 6) == \
        _lib.ERR_G_RANGE
    with pytest.raises(_lib.gexceptions.GMixRangeError):
        _lib.check(_lib.ERR_G_RANGE)
    with pytest.raises(ZeroDivisionError):
        _lib.check(_lib.ERR_ZERO_DIV)
    # psf with zero total flux: numba raises ZeroDivisionError
    psf = np.zeros(1, dtype=_lib.GAUSS2D_DTYPE)
    out = np.zeros(6, dtype=_lib.GAUSS2D_DTYPE)
    assert L.ngmix_convolve_fill(_lib.ptr(out), _lib.ptr(gm6), 6, _lib.ptr(psf),
                                 1) == _lib.ERR_ZERO_DIV


@pytest.mark.parametrize("jname", ["unit", "diag", "sheared"])
def test_host_jacobian_transforms(golden, jname):
    g = golden("pixels")
    L = _lib.lib()
    jac = np.ascontiguousarray(g["jac_" + jname]).astype(_lib.JACOBIAN_DTYPE)
    a, b = ctypes.c_double(), ctypes.c_double()
    for (r, c), (v, u), (r2, c2) in zip(g["pts_" + jname], g["vu_" + jname],
                                        g["rowcol_" + jname]):
        L.ngmix_jacobian_get_vu(_lib.ptr(jac), r, c, ctypes.byref(a), ctypes.byref(b))
        assert (a.value, b.value) == (v, u)
        assert L.ngmix_jacobian_get_rowcol(_lib.ptr(jac), v, u, ctypes.byref(a),
                                           ctypes.byref(b)) == 0
        assert (a.value, b.value) == (r2, c2)


def test_compute_fails_loudly_without_gpu():
    """no silent CPU path: a pixel loop without a device is an error"""
    L = _lib.lib()
    if L.ngmix_device_count() > 0:
        pytest.skip("a GPU is present")
    gm = np.zeros(1, dtype=_lib.GAUSS2D_DTYPE)
    gm["p"] = gm["irr"] = gm["icc"] = gm["det"] = 1.0
    pix = np.zeros(4, dtype=_lib.PIXEL_DTYPE)
    ll = ctypes.c_double()
    npix = ctypes.c_int64()
    st = L.ngmix_get_loglike(_lib.ptr(gm), 1, _lib.ptr(pix), 4, ctypes.byref(ll),
                             ctypes.byref(ll), ctypes.byref(ll), ctypes.byref(npix))
    assert st == _lib.ERR_HIP
    with pytest.raises(RuntimeError):
        _lib.check(st, "ngmix_get_loglike")
    from ngmix_amd.batch import StampBatch
    with pytest.raises(RuntimeError):
        StampBatch.from_images(np.zeros((1, 4, 4)))


def test_kernel_resource_guard():
    """the build-time guard of the hand-scheduled kernels (tools/
    kernel_resources.py, run by `make` after linking): no scratch / VGPR
    spills in the kernels whose s_waitcnt vmcnt(N) are counted by hand, and
    their register count inside the validated occupancy band"""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location(
        "kernel_resources", os.path.join(root, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    table, problems = kr.check(_lib.LIB_PATH)
    assert problems == []
    wave = {k: v for k, v in table.items() if "pixpass_wave_kernel" in k}
    assert len(wave) >= 8     # loglike / fdiff / render / s2n, masked and not
    for name, md in wave.items():
        assert md.get("private_segment_fixed_size", 0) == 0, name
        assert md.get("vgpr_spill_count", 0) == 0, name
        assert md["vgpr_count"] <= kr.GUARDS["pixpass_wave_kernel"], name
    # the guard does fire: a kernel over its band is reported
    old = kr.GUARDS["pixpass_wave_kernel"]
    kr.GUARDS["pixpass_wave_kernel"] = 8
    try:
        _, problems = kr.check(_lib.LIB_PATH)
    finally:
        kr.GUARDS["pixpass_wave_kernel"] = old
    assert any("outside the validated occupancy band" in p for p in problems)


def test_isa_hazard_guard(tmp_path):
    """tools/isa_hazards.py (run by `make` on every object): the dataflow that
    follows each load to its s_waitcnt and the wait-state rules fire on every
    deliberately broken kernel of tests/helpers/hazard_cases.hip and on none of
    the repaired ones; the shipped library's fused pixel-pass kernels -- the
    ones with hand-counted vmcnt waits -- and the LM evaluation kernel with
    its inline-asm readfirstlanes pass"""
    import importlib.util
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location(
        "isa_hazards", os.path.join(root, "tools", "isa_hazards.py"))
    ih = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ih)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc here")
    obj = str(tmp_path / "hazard_cases.o")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-c",
                    os.path.join(root, "tests", "helpers", "hazard_cases.hip"), "-o", obj],
                   check=True, capture_output=True)
    problems, stats = ih.check([obj])
    flagged = {p.split(":")[0] for p in problems}
    assert flagged == {"bad_wait_count", "bad_loop_exit", "bad_readfirstlane", "bad_dpp",
                       "bad_sgpr_base"}, problems
    assert stats["functions"] == 9
    assert any("may be in flight" in p for p in problems)
    assert any("v_readlane/v_readfirstlane (1)" in p for p in problems)
    assert any("DPP read (2)" in p for p in problems)
    assert any("VMEM read of it (5)" in p for p in problems)
    # the shipped kernels
    problems, stats = ih.check([_lib.LIB_PATH], only=["pixpass_wave_kernel", "lm_eval_kernel"])
    assert problems == []
    assert stats["functions"] >= 10 and stats["loads"] > 1000 and stats["readlanes"] > 50


def test_library_matches_sources():
    """the loaded library was built from the sources in csrc/ (the digest `make`
    leaves next to it); _lib.lib() rebuilds or refuses otherwise"""
    assert _lib.source_hash() == open(_lib.LIB_PATH + ".srchash").read().strip()
    assert not _lib.library_is_stale()
    assert "pixpass.hip" in _lib._makefile_list("SRCS")
    assert "common.hpp" in _lib._makefile_list("HDRS")


def test_no_fallback_when_the_library_is_missing(tmp_path):
    """the product path fails loudly: a process told to load a library that is
    not there (NGMIX_HIP_LIB) gets an exception from the first call that needs
    it -- there is no CPU path to fall back to, and nothing under ngmix_amd/
    imports the oracle"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NGMIX_HIP_LIB=str(tmp_path / "no_such_library.so"))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from ngmix_amd import _lib\n"
            "_lib.lib()\n" % root)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, timeout=300)
    assert r.returncode != 0
    assert b"no_such_library.so" in r.stderr and b"OSError" in r.stderr
    # and no module of the package reaches for the oracle
    pkg = os.path.join(root, "ngmix_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                with open(os.path.join(dirpath, f), errors="replace") as fh:
                    text = fh.read()
                assert "ngmix_oracle" not in text and "from oracle" not in text \
                    and "import oracle" not in text, os.path.join(dirpath, f)


def test_compute_calls_need_a_gpu():
    """without a GPU every compute entry point of the Python shell raises (the
    host classes build, their pixel loops do not run anywhere else)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import ngmix_amd as ngmix
    from ngmix_amd.batch import StampBatch
    gm = ngmix.GMixModel([0.0, 0.0, 0.0, 0.0, 0.5, 1.0], "gauss")
    with pytest.raises(RuntimeError, match="needs a GPU"):
        gm.make_image((16, 16), jacobian=ngmix.DiagonalJacobian(row=7.5, col=7.5, scale=0.263))
    with pytest.raises(RuntimeError, match="needs a GPU"):
        StampBatch.from_images(np.zeros((2, 8, 8)), np.ones((2, 8, 8)),
                               np.array([3.5, 3.5, 1.0, 0.0, 0.0, 1.0, 1.0, 1.0]))


def test_bad_arguments_are_refused_before_anything_runs():
    """the entry points that check their arguments return NGMIX_ERR_BAD_ARG
    before the first device call: callable here, without a GPU (anything that
    went on to a hip* call would come back with a runtime error instead)"""
    L = _lib.lib()
    bad = _lib.ERR_BAD_ARG
    assert L.ngmix_events_create(-1, None) == bad
    assert L.ngmix_events_create(3, None) == bad
    assert L.ngmix_events_destroy(-2, None) == bad
    assert L.ngmix_event_record(None, None) == bad
    assert L.ngmix_event_synchronize(None) == bad
    assert L.ngmix_event_elapsed_ms(None, None, None) == bad
    gm = np.zeros(1, dtype=_lib.GAUSS2D_DTYPE)
    pix = np.zeros(4, dtype=_lib.PIXEL_DTYPE)
    pars = np.zeros(8)
    # NGMIX_MODEL_CM = 9: the composite model has its own entry point (gmix_fill_cm)
    assert L.ngmix_fill_model(_lib.ptr(gm), 1, 9, _lib.ptr(pars), 8) == bad
    img = np.zeros((2, 2))
    jac = np.zeros(1, dtype=_lib.JACOBIAN_DTYPE)
    for nrow, ncol in ((0, 5), (-1, 4), (1 << 16, 1 << 16)):
        assert L.ngmix_fill_pixels(_lib.ptr(pix), 4, _lib.ptr(img), _lib.ptr(img), nrow, ncol,
                                   _lib.ptr(jac), 1) == bad
    res = np.zeros(1, dtype=_lib.moments_result_dtype(6))
    for nmom in (0, 7, 16, 18):
        assert L.ngmix_get_weighted_sums(_lib.ptr(gm), 1, _lib.ptr(pix), 4, _lib.ptr(res), nmom,
                                         10.0) == bad
    conf = np.zeros(1, dtype=_lib.EM_CONF_DTYPE)
    sums = np.zeros(14)
    numiter, frac, sky = ctypes.c_int32(), ctypes.c_double(), ctypes.c_double()
    for kind, ng, npsf in ((-1, 1, 1), (4, 1, 1), (0, 0, 1), (0, 1, 0)):
        assert L.ngmix_em_run(kind, _lib.ptr(conf), _lib.ptr(pix), 4, _lib.ptr(sums), _lib.ptr(gm),
                              ng, _lib.ptr(gm), npsf, _lib.ptr(gm), 0, ctypes.byref(numiter),
                              ctypes.byref(frac), ctypes.byref(sky)) == bad
    states = np.zeros(2, dtype=_lib.LM_STATE_DTYPE)
    x0 = np.zeros((2, 6))
    for npars in (0, -3, _lib.LM_NPMAX + 1):
        assert L.ngmix_lm_init(_lib.ptr(states), 2, npars, _lib.ptr(x0), 1e-5, 1e-5, 0.0, 100,
                               100.0, 0, None, None) == bad
        assert L.ngmix_lm_init_batch(_lib.ptr(states), 2, npars, _lib.ptr(x0), 1e-5, 1e-5, 0.0,
                                     100, 100.0, 0, None, None, None) == bad
    assert L.ngmix_lm_init(_lib.ptr(states), -1, 6, _lib.ptr(x0), 1e-5, 1e-5, 0.0, 100, 100.0, 0,
                           None, None) == bad
