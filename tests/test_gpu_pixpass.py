"""
GPU parity tests for the single-pass pixel kernels (render / get_loglike /
fill_fdiff / get_model_s2n_sum / fill_pixels / fill_coords), through the C ABI.

Checks, in order of strength:
  * seam forms and batch forms against the committed goldens (outputs of the
    reference itself) and against the CPU oracle on seeded random inputs;
  * bit-exact (==) for pixel indexing, coordinates, fdiff and fast renders;
    loglike / s2n sums to 1e-12 relative (north_star: 1e-10) because only the
    summation order differs;
  * size-independent properties at BASELINE's full stamp size: exact skipping
    (skip == no-skip bitwise), loglike == -0.5*sum(fdiff^2), render linearity.
"""
import ctypes

import numpy as np
import pytest

from ngmix_amd import _lib

pytestmark = pytest.mark.gpu

RENDER_NAMES = ["c1_exp48", "exp48_psf", "gauss32", "bdf64_psf", "masked13x15",
                "masked13x15_keepzero", "tiny20x17"]
SUM_RTOL = 1e-12  # north_star tolerance is 1e-10; only summation order differs
# fused kernels (FMA + shared-centre algebra): per-pixel values agree with the
# reference to rounding; checked against the largest model value of the stamp
PIX_RTOL = 2e-13
FUSED_ELEM_RTOL = 1e-10  # BASELINE.json north_star: 1e-10 relative on loglike / fdiff


def assert_pixels(got, ref, exact, scale=None, err_msg=""):
    """bit-exact in exact mode; to rounding (relative to the stamp's peak
    |value|) in fused mode"""
    if exact:
        np.testing.assert_array_equal(got, ref, err_msg=err_msg)
    else:
        if scale is None:
            scale = np.abs(ref).max()
        np.testing.assert_allclose(got, ref, rtol=PIX_RTOL, atol=PIX_RTOL * scale,
                                   err_msg=err_msg)
        # north_star's tolerance, stated elementwise: 1e-10 RELATIVE on every
        # value that is not itself a cancellation residue (|ref| above 1e-3 of
        # the stamp's scale)
        got, ref = np.asarray(got), np.asarray(ref)
        big = np.abs(ref) > 1e-3 * scale
        if big.any():
            np.testing.assert_allclose(got[big], ref[big], rtol=FUSED_ELEM_RTOL, atol=0,
                                       err_msg=err_msg + " (elementwise relative)")


def as_gauss(a):
    out = np.zeros(a.size, dtype=_lib.GAUSS2D_DTYPE)
    for n in _lib.GAUSS2D_DTYPE.names:
        out[n] = a[n]
    return out


def as_pixels(a):
    out = np.zeros(a.size, dtype=_lib.PIXEL_DTYPE)
    for n in _lib.PIXEL_DTYPE.names:
        out[n] = a[n]
    return out


def jac_rec(a):
    return np.ascontiguousarray(a).astype(_lib.JACOBIAN_DTYPE)


def seam_make_pixels(image, weight, jac, izw):
    L = _lib.lib()
    image = np.ascontiguousarray(image, dtype="f8")
    weight = np.ascontiguousarray(weight, dtype="f8")
    n = int((weight > 0).sum()) if izw else image.size
    pix = np.zeros(n, dtype=_lib.PIXEL_DTYPE)
    st = L.ngmix_fill_pixels(_lib.ptr(pix), n, _lib.ptr(image), _lib.ptr(weight),
                             image.shape[0], image.shape[1], _lib.ptr(jac), int(izw))
    return st, pix


# --------------------------------------------------------------- seam forms
@pytest.mark.parametrize("jname", ["unit", "diag", "sheared"])
def test_seam_fill_pixels_coords_exact(golden, jname):
    g = golden("pixels")
    L = _lib.lib()
    jac = jac_rec(g["jac_" + jname])
    nrow, ncol = g["image"].shape
    coords = np.zeros(nrow * ncol, dtype=_lib.COORD_DTYPE)
    assert L.ngmix_fill_coords(_lib.ptr(coords), nrow, ncol, _lib.ptr(jac)) == 0
    for n in ("u", "v", "area"):
        np.testing.assert_array_equal(coords[n], g["coords_" + jname][n])
    for izw in (1, 0):
        st, pix = seam_make_pixels(g["image"], g["weight"], jac, izw)
        assert st == 0
        ref = g["pixels_%s_izw%d" % (jname, izw)]
        assert pix.size == ref.size
        for n in ("u", "v", "area", "val", "ierr"):
            np.testing.assert_array_equal(pix[n], ref[n], err_msg=n)
    # wrong-sized pixel array: RuntimeError('some pixels were not filled')
    pix = np.zeros(5, dtype=_lib.PIXEL_DTYPE)
    im = np.ascontiguousarray(g["image"])
    wt = np.ascontiguousarray(g["weight"])
    st = L.ngmix_fill_pixels(_lib.ptr(pix), 5, _lib.ptr(im), _lib.ptr(wt), nrow,
                             ncol, _lib.ptr(jac), 1)
    assert st == _lib.ERR_PIXELS_NOT_FILLED


@pytest.mark.parametrize("name", RENDER_NAMES)
def test_seam_render_loglike_fdiff(golden, name):
    g = golden("render_loglike")
    L = _lib.lib()
    gm = as_gauss(g[name + "_gmix_in"])
    pixels = as_pixels(g[name + "_pixels"])
    jac = jac_rec(g[name + "_jac"])
    shape = g[name + "_image"].shape

    ll, sn, sd = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    npix = ctypes.c_int64()
    assert gm["norm_set"][0] == 0
    st = L.ngmix_get_loglike(_lib.ptr(gm), gm.size, _lib.ptr(pixels), pixels.size,
                             ctypes.byref(ll), ctypes.byref(sn), ctypes.byref(sd),
                             ctypes.byref(npix))
    assert st == 0
    # lazy norm side effect on the caller's array, exact
    refn = g[name + "_gmix_normed"]
    for n in _lib.GAUSS2D_DTYPE.names:
        np.testing.assert_array_equal(gm[n], refn[n], err_msg=n)
    ref = g[name + "_loglike"]
    np.testing.assert_allclose([ll.value, sn.value, sd.value], ref[:3],
                               rtol=SUM_RTOL, atol=0)
    assert npix.value == int(ref[3])

    for start in (0, 13):
        ref_fd = g[name + "_fdiff_start%d" % start]
        fdiff = np.zeros(ref_fd.size) + 7.0
        assert L.ngmix_fill_fdiff(_lib.ptr(gm), gm.size, _lib.ptr(pixels),
                                  pixels.size, _lib.ptr(fdiff), start) == 0
        np.testing.assert_array_equal(fdiff, ref_fd)

    s2n = ctypes.c_double()
    assert L.ngmix_get_model_s2n_sum(_lib.ptr(gm), gm.size, _lib.ptr(pixels),
                                     pixels.size, ctypes.byref(s2n)) == 0
    np.testing.assert_allclose(s2n.value, float(g[name + "_s2n_sum"]),
                               rtol=SUM_RTOL, atol=0)

    coords = np.zeros(shape[0] * shape[1], dtype=_lib.COORD_DTYPE)
    assert L.ngmix_fill_coords(_lib.ptr(coords), shape[0], shape[1],
                               _lib.ptr(jac)) == 0
    im = np.zeros(coords.size)
    assert L.ngmix_render(_lib.ptr(gm), gm.size, _lib.ptr(coords), coords.size,
                          _lib.ptr(im), 1) == 0
    np.testing.assert_array_equal(im.reshape(shape), g[name + "_render_fast"])
    im = np.zeros(coords.size)
    assert L.ngmix_render(_lib.ptr(gm), gm.size, _lib.ptr(coords), coords.size,
                          _lib.ptr(im), 0) == 0
    # true exp: device libm vs numpy, a few ulp
    np.testing.assert_allclose(im.reshape(shape), g[name + "_render_exact"],
                               rtol=1e-14, atol=1e-300)
    acc = g[name + "_render_base"].copy().ravel()
    assert L.ngmix_render(_lib.ptr(gm), gm.size, _lib.ptr(coords), coords.size,
                          _lib.ptr(acc), 1) == 0
    np.testing.assert_array_equal(acc.reshape(shape), g[name + "_render_accum"])


# SURVEY.md 8(d), "C1 single stamp" (oracle/gen_golden_c1.py: the reference's
# own numbers for the survey's exact inputs)
SURVEY_C1 = (-1158.1127300983387, 2798242.571962917, 2798842.226964671, 2304)


@pytest.mark.parametrize("exact", [True, False], ids=["exact", "fused"])
def test_c1_survey_values(golden, exact):
    """config C1 as SURVEY.md states it -- RandomState(1), the true mixture --
    through the seam form ngmix_get_loglike and through the batch kernel:
    loglike / s2n sums to 1e-10 relative (north_star), npix exact"""
    from ngmix_amd.batch import StampBatch, GMixBatch
    g = golden("c1")
    L = _lib.lib()
    gm = as_gauss(g["gmix_in"])
    jac = jac_rec(g["jac"])
    st, pixels = seam_make_pixels(g["image"], g["weight"], jac, True)
    assert st == 0 and pixels.size == SURVEY_C1[3]
    ll, sn, sd = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    npix = ctypes.c_int64()
    assert L.ngmix_get_loglike(_lib.ptr(gm), gm.size, _lib.ptr(pixels), pixels.size,
                               ctypes.byref(ll), ctypes.byref(sn), ctypes.byref(sd),
                               ctypes.byref(npix)) == 0
    np.testing.assert_allclose([ll.value, sn.value, sd.value], SURVEY_C1[:3],
                               rtol=1e-10, atol=0)
    assert npix.value == SURVEY_C1[3]
    for n in _lib.GAUSS2D_DTYPE.names:
        np.testing.assert_array_equal(gm[n], g["gmix_normed"][n], err_msg=n)

    sb = StampBatch.from_images(g["image"], g["weight"], g["jac"])
    gmb = GMixBatch.from_numpy(as_gauss(g["gmix_in"]))
    out, status = sb.loglike(gmb, exact=exact)
    out = out.cpu().numpy()[0]
    assert int(status.cpu()[0]) == 0
    np.testing.assert_allclose(out[:3], SURVEY_C1[:3], rtol=1e-10, atol=0)
    assert out[3] == SURVEY_C1[3]
    fdiff, _ = sb.fill_fdiff(gmb, exact=exact)
    assert_pixels(fdiff.cpu().numpy()[:2304], g["fdiff"], exact)
    im, _ = sb.render(gmb, fast_exp=True, exact=exact)
    assert_pixels(im.cpu().numpy().reshape(48, 48), g["render_fast"], exact)


def test_seam_range_error():
    L = _lib.lib()
    gm = np.zeros(2, dtype=_lib.GAUSS2D_DTYPE)
    gm["p"] = gm["irr"] = gm["icc"] = gm["det"] = 1.0
    gm["det"][1] = 0.0
    pix = np.zeros(4, dtype=_lib.PIXEL_DTYPE)
    out = np.zeros(4)
    st = L.ngmix_fill_fdiff(_lib.ptr(gm), 2, _lib.ptr(pix), 4, _lib.ptr(out), 0)
    assert st == _lib.ERR_DET_TOO_LOW
    assert gm["norm_set"][0] == 1 and gm["norm_set"][1] == 0


# -------------------------------------------------------------- batch forms
def _batch_from_case(g, name):
    from ngmix_amd.batch import StampBatch, GMixBatch
    sb = StampBatch.from_images(g[name + "_image"], g[name + "_weight"],
                                g[name + "_jac"],
                                ignore_zero_weight=bool(g[name + "_izw"]))
    gm = GMixBatch.from_numpy(as_gauss(g[name + "_gmix_in"]))
    return sb, gm


@pytest.mark.parametrize("exact", [True, False], ids=["exact", "fused"])
@pytest.mark.parametrize("name", RENDER_NAMES)
def test_batch_single_stamp_vs_golden(golden, name, exact):
    import torch
    g = golden("render_loglike")
    sb, gm = _batch_from_case(g, name)
    ref = g[name + "_loglike"]
    assert int(sb.npix_kept[0]) == int(ref[3])
    out, status = sb.loglike(gm, exact=exact)
    out = out.cpu().numpy()[0]
    assert int(status.cpu()[0]) == 0
    np.testing.assert_allclose(out[:3], ref[:3], rtol=SUM_RTOL, atol=0)
    assert out[3] == ref[3]
    # lazy norms written back on the device, exact
    refn = g[name + "_gmix_normed"]
    back = gm.to_numpy()[0]
    for n in _lib.GAUSS2D_DTYPE.names:
        np.testing.assert_array_equal(back[n], refn[n], err_msg=n)

    nk = int(ref[3])
    for start in (0, 13):
        ref_fd = g[name + "_fdiff_start%d" % start]
        fdiff = torch.full((ref_fd.size,), 7.0, dtype=torch.float64, device="cuda")
        sb.fill_fdiff(gm, fdiff=fdiff, fdiff_start=np.array([start]), exact=exact)
        got = fdiff.cpu().numpy()
        # untouched padding is exact in both modes
        np.testing.assert_array_equal(got[:start], ref_fd[:start])
        np.testing.assert_array_equal(got[start + nk:], ref_fd[start + nk:])
        assert_pixels(got[start:start + nk], ref_fd[start:start + nk], exact)
        assert nk + start <= ref_fd.size

    s2n, _ = sb.model_s2n_sum(gm, exact=exact)
    np.testing.assert_allclose(float(s2n.cpu()[0]), float(g[name + "_s2n_sum"]),
                               rtol=SUM_RTOL, atol=0)
    shape = g[name + "_image"].shape
    im, _ = sb.render(gm, fast_exp=True, exact=exact)
    assert_pixels(im.cpu().numpy().reshape(shape), g[name + "_render_fast"], exact)
    im, _ = sb.render(gm, fast_exp=False)
    np.testing.assert_allclose(im.cpu().numpy().reshape(shape),
                               g[name + "_render_exact"], rtol=1e-14, atol=1e-300)
    acc = torch.from_numpy(g[name + "_render_base"].copy().ravel()).cuda()
    sb.render(gm, image=acc, fast_exp=True, exact=exact)
    assert_pixels(acc.cpu().numpy().reshape(shape), g[name + "_render_accum"],
                  exact, scale=np.abs(g[name + "_render_fast"]).max())


def _random_mixtures(rng, n, ngauss, scale):
    """random, sometimes nasty, mixtures with norms unset"""
    gm = np.zeros((n, ngauss), dtype=_lib.GAUSS2D_DTYPE)
    T = rng.uniform(0.05, 2.5, size=(n, ngauss)) * scale ** 2 * 10
    e1 = rng.uniform(-0.7, 0.7, size=(n, ngauss))
    e2 = rng.uniform(-0.6, 0.6, size=(n, ngauss))
    emag = np.sqrt(e1 ** 2 + e2 ** 2)
    shrink = np.where(emag > 0.95, 0.95 / emag, 1.0)
    e1 *= shrink
    e2 *= shrink
    gm["p"] = rng.uniform(-0.2, 5.0, size=(n, ngauss))
    gm["row"] = rng.uniform(-6, 6, size=(n, ngauss)) * scale
    gm["col"] = rng.uniform(-6, 6, size=(n, ngauss)) * scale
    gm["irr"] = T / 2 * (1 - e1)
    gm["irc"] = T / 2 * e2
    gm["icc"] = T / 2 * (1 + e1)
    gm["det"] = gm["irr"] * gm["icc"] - gm["irc"] ** 2
    for f in ("drr", "drc", "dcc", "norm", "pnorm"):
        gm[f] = np.nan
    return gm


def wk_max(w):
    return max(float(np.max(w)), 0.0)


def _oracle_eval(gmrow, image, weight, jac, izw):
    from oracle import oracle as ora
    gm = np.zeros(gmrow.size, dtype=ora.GAUSS2D_DTYPE)
    for n in ora.GAUSS2D_DTYPE.names:
        gm[n] = gmrow[n]
    j = np.zeros(1, dtype=ora.JACOBIAN_DTYPE)
    j[0] = tuple(jac)
    pix = ora.make_pixels(image, weight, j, izw)
    st, res = ora.get_loglike(gm, pix)
    assert st == 0
    fd = np.zeros(pix.size)
    ora.fill_fdiff(gm, pix, fd, 0)
    coords = ora.make_coords(image.shape, j)
    im = np.zeros(image.size)
    ora.render(gm, coords, im, 1)
    return res, fd, im.reshape(image.shape)


@pytest.mark.parametrize("exact", [True, False], ids=["exact", "fused"])
@pytest.mark.parametrize("dims,ngauss", [((48, 48), 6), ((32, 32), 1),
                                         ((25, 25), 3), ((64, 64), 16),
                                         ((7, 50), 2), ((70, 9), 4),
                                         # more than 32 gaussians: per-tile ballots
                                         ((40, 56), 40), ((48, 48), 33),
                                         # fewer tiles than the look-ahead depth
                                         ((8, 16), 2), ((4, 16), 1), ((8, 8), 3),
                                         # many tiles per wave, complete tiles
                                         ((96, 112), 5)])
def test_batch_random_vs_oracle(dims, ngauss, exact):
    """seeded random batches, sheared jacobians, masks: batch kernels == oracle"""
    from ngmix_amd.batch import StampBatch, GMixBatch
    rng = np.random.RandomState(1234 + dims[0] * 100 + ngauss)
    n = 12
    scale = 0.263
    nrow, ncol = dims
    images = rng.normal(size=(n, nrow, ncol))
    weights = rng.uniform(0.5, 2.0, size=(n, nrow, ncol))
    weights[rng.uniform(size=weights.shape) < 0.03] = 0.0
    weights[0] = 1.0  # one unmasked stamp
    weights[1, 0, 0] = -3.0
    jac = np.zeros((n, 8))
    for i in range(n):
        a = scale * (1 + rng.uniform(-0.1, 0.1))
        d = scale * (1 + rng.uniform(-0.1, 0.1))
        b, c = rng.uniform(-0.03, 0.03, size=2)
        if i % 3 == 0:
            b = c = 0.0
        jac[i] = [(nrow - 1) / 2 + rng.uniform(-0.5, 0.5),
                  (ncol - 1) / 2 + rng.uniform(-0.5, 0.5), a, b, c, d,
                  a * d - b * c, np.sqrt(abs(a * d - b * c))]
    gmh = _random_mixtures(rng, n, ngauss, scale)
    for izw in (True, False):
        sb = StampBatch.from_images(images, weights, jac, ignore_zero_weight=izw)
        gm = GMixBatch.from_numpy(gmh)
        out, status = sb.loglike(gm, exact=exact)
        fd, _ = sb.fill_fdiff(gm, exact=exact)
        im, _ = sb.render(gm, fast_exp=True, exact=exact)
        out = out.cpu().numpy()
        fd = fd.cpu().numpy()
        im = im.cpu().numpy().reshape(n, nrow, ncol)
        assert np.all(status.cpu().numpy() == 0)
        offs = sb.kept_offsets()
        for i in range(n):
            res, rfd, rim = _oracle_eval(gmh[i], images[i], weights[i], jac[i], izw)
            assert int(sb.npix_kept[i]) == res[3] == out[i, 3]
            mscale = max(np.abs(rim).max(), 1e-300)
            # fdiff = (model - val)*ierr: model rounding scaled by ierr
            assert_pixels(fd[offs[i]:offs[i] + res[3]], rfd, exact,
                          scale=mscale * np.sqrt(wk_max(weights[i])) + np.abs(rfd).max(),
                          err_msg="fdiff stamp %d" % i)
            assert_pixels(im[i], rim, exact, scale=mscale)
            scale_ll = max(abs(res[0]), 1e-300)
            assert abs(out[i, 0] - res[0]) <= (SUM_RTOL if exact else 1e-11) * scale_ll
            # s2n_numer = sum(val*model*ivar) has mixed signs: bound the error
            # by sum|terms| <= sqrt(sum val^2 ivar * sum model^2 ivar)
            wk = np.where(weights[i] > 0, weights[i], 0.0)
            aa = float((images[i] ** 2 * wk).sum())
            assert abs(out[i, 1] - res[1]) <= 1e-11 * np.sqrt(aa * res[2]) + 1e-300
            np.testing.assert_allclose(out[i, 2], res[2],
                                       rtol=SUM_RTOL if exact else 1e-11, atol=0)


def test_batch_status_per_stamp():
    """one bad stamp (det too low) must not abort the batch"""
    from ngmix_amd.batch import StampBatch, GMixBatch
    rng = np.random.RandomState(7)
    n = 5
    images = rng.normal(size=(n, 16, 16))
    gmh = _random_mixtures(rng, n, 2, 0.263)
    gmh["det"][2, 1] = 1e-250
    sb = StampBatch.from_images(images)
    gm = GMixBatch.from_numpy(gmh)
    out, status = sb.loglike(gm)
    status = status.cpu().numpy()
    assert list(status) == [0, 0, _lib.ERR_DET_TOO_LOW, 0, 0]
    back = gm.to_numpy()
    assert back["norm_set"][2, 0] == 1 and back["norm_set"][2, 1] == 0
    assert np.all(back["norm_set"][[0, 1, 3, 4]] == 1)


def test_batch_fill_convolve_norms_vs_host():
    """device param prep == host param prep (same source), exact"""
    import torch
    from ngmix_amd.batch import GMixBatch
    L = _lib.lib()
    rng = np.random.RandomState(99)
    n = 64
    pars = np.zeros((n, 6))
    pars[:, 0:2] = rng.uniform(-0.1, 0.1, size=(n, 2))
    pars[:, 2:4] = rng.normal(scale=0.2, size=(n, 2))
    pars[:, 4] = rng.uniform(0.2, 1.5, size=n)
    pars[:, 5] = rng.uniform(50, 500, size=n)
    pars[5, 2:4] = [0.9, 0.9]  # g >= 1
    psfpars = np.tile([0.0, 0.0, 0.01, -0.02, 0.27, 1.0], (n, 1))
    for model, mid, ng in (("exp", 3, 6), ("dev", 4, 10), ("gauss", 1, 1)):
        gm, st = GMixBatch.from_pars(pars, model)
        psf, _ = GMixBatch.from_pars(psfpars, "turb")
        st = st.cpu().numpy()
        assert st[5] == _lib.ERR_G_RANGE and np.all(np.delete(st, 5) == 0)
        conv, _ = gm.convolve(psf)
        nst = conv.set_norms().cpu().numpy()
        dev = conv.to_numpy()
        hpsf = np.zeros(3, dtype=_lib.GAUSS2D_DTYPE)
        L.ngmix_fill_model(_lib.ptr(hpsf), 3, 2,
                           _lib.ptr(np.ascontiguousarray(psfpars[0])), 6)
        for i in range(n):
            if i == 5:
                continue
            h = np.zeros(ng, dtype=_lib.GAUSS2D_DTYPE)
            assert L.ngmix_fill_model(_lib.ptr(h), ng, mid,
                                      _lib.ptr(np.ascontiguousarray(pars[i])), 6) == 0
            hc = np.zeros(ng * 3, dtype=_lib.GAUSS2D_DTYPE)
            L.ngmix_convolve_fill(_lib.ptr(hc), _lib.ptr(h), ng, _lib.ptr(hpsf), 3)
            assert L.ngmix_set_norms(_lib.ptr(hc), hc.size) == 0
            assert nst[i] == 0
            for f in _lib.GAUSS2D_DTYPE.names:
                # atanh/tanh differ between device and host libm by ulps, and
                # the object's and psf's irc can cancel: absolute floor too
                np.testing.assert_allclose(dev[i][f], hc[f], rtol=1e-12,
                                           atol=1e-15, err_msg=f)
    assert torch.cuda.is_available()


FILL_CASES = ["gauss", "exp", "dev", "turb", "bdf", "bd", "coellip", "full",
              "exp_round", "exp_highg", "cm"]
_GFIELDS = ("p", "row", "col", "irr", "irc", "icc", "det")
_NFIELDS = ("drr", "drc", "dcc", "norm", "pnorm")


def _assert_ulp(a, b, ulps, name):
    """|a - b| <= ulps units in the last place of b (elementwise)"""
    a, b = np.asarray(a, dtype="f8"), np.asarray(b, dtype="f8")
    tol = ulps * np.spacing(np.abs(b))
    bad = ~(np.abs(a - b) <= tol)
    assert not bad.any(), (name, a[bad], b[bad])


@pytest.mark.parametrize("name", FILL_CASES)
def test_batch_fill_convolve_norms_vs_golden(golden, name):
    """the DEVICE model-fill, convolve and norm kernels (gmixprep.hip) against
    the reference's own gmix_fill_* / gmix_convolve_fill / gmix_set_norms
    arrays (tests/golden/fills.npz; gmix_nb.py:307-558, 609-649, 176-218).
    Libm-free values must be equal; where tanh / atanh / pow enter the shape
    (g1g2_to_e1e2, the bd T ratio) the device libm may differ by ulps."""
    from ngmix_amd.batch import GMixBatch
    g = golden("fills")
    model = name.split("_")[0]
    ref = g["gmix_" + name]
    ngauss = ref.size
    # the same parameters in several rows: every thread of the block must
    # produce the same record
    nrep = 3
    if model == "cm":
        pars = np.tile(g["pars_exp"], (nrep, 1))
        extra = np.tile([float(g["cm_fracdev"]), float(g["cm_TdByTe"]),
                         float(g["cm_Tfactor"])], (nrep, 1))
        gm, st = GMixBatch.from_pars(pars, "cm", cm_extra=extra)
    else:
        pars = np.tile(g["pars_" + name], (nrep, 1))
        gm, st = GMixBatch.from_pars(pars, model, ngauss=ngauss)
    assert np.all(st.cpu().numpy() == 0)
    dev = gm.to_numpy()
    assert dev.shape == (nrep, ngauss)
    for i in range(nrep):
        for f in ("p", "row", "col"):
            np.testing.assert_array_equal(dev[i][f], ref[f], err_msg=f)
        for f in ("irr", "irc", "icc", "det"):
            if model == "full":
                np.testing.assert_array_equal(dev[i][f], ref[f], err_msg=f)
            else:
                # e1, e2 each within ~2 ulp; irc = T/2 e2 and det inherit them
                _assert_ulp(dev[i][f], ref[f], 8, f)
        assert np.all(dev[i]["norm_set"] == 0)
        assert np.all(np.isnan(dev[i]["pnorm"])) and np.all(np.isnan(dev[i]["drr"]))
    if model == "cm":
        return
    # convolution and norms from the REFERENCE's unconvolved mixture: no libm
    # on this path except sqrt and division, which are correctly rounded
    h = np.zeros((nrep, ngauss), dtype=_lib.GAUSS2D_DTYPE)
    for f in _lib.GAUSS2D_DTYPE.names:
        h[f] = ref[f]
    gm_ref = GMixBatch.from_numpy(h)
    for pname in ("psf1", "psf3", "psf_off"):
        p = g[pname]
        hp = np.zeros((nrep, p.size), dtype=_lib.GAUSS2D_DTYPE)
        for f in _lib.GAUSS2D_DTYPE.names:
            hp[f] = p[f]
        conv, cst = gm_ref.convolve(GMixBatch.from_numpy(hp))
        assert np.all(cst.cpu().numpy() == 0)
        refc = g["conv_%s_%s" % (name, pname)]
        refn = g["convnorm_%s_%s" % (name, pname)]
        out = conv.to_numpy()
        for i in range(nrep):
            for f in _GFIELDS:
                np.testing.assert_array_equal(out[i][f], refc[f], err_msg=f)
        nst = conv.set_norms().cpu().numpy()
        if np.all(refn["norm_set"] == 1):
            assert np.all(nst == 0)
            out = conv.to_numpy()
            for i in range(nrep):
                assert np.all(out[i]["norm_set"] == 1)
                for f in _GFIELDS + _NFIELDS:
                    np.testing.assert_array_equal(out[i][f], refn[f], err_msg=f)
        # and the whole device chain (device fill -> convolve -> norms)
        conv2, _ = gm.convolve(GMixBatch.from_numpy(hp))
        assert np.all(conv2.set_norms().cpu().numpy() == 0) or \
            not np.all(refn["norm_set"] == 1)
        out2 = conv2.to_numpy()
        if np.all(refn["norm_set"] == 1):
            for f in _GFIELDS + _NFIELDS:
                np.testing.assert_allclose(out2[0][f], refn[f], rtol=1e-13, atol=1e-15,
                                           err_msg=f)


def test_batch_fill_errors_vs_reference_semantics(golden):
    """g >= 1 (g1g2_to_e1e2, gmix_nb.py:652-678) marks the stamp and leaves
    its mixture untouched; a psf without flux is numba's ZeroDivisionError
    (gmix_get_cen, gmix_nb.py:108-130)"""
    from ngmix_amd.batch import GMixBatch
    g = golden("fills")
    pars = np.tile(g["pars_exp"], (4, 1))
    pars[2, 2:4] = [0.8, 0.7]
    gm, st = GMixBatch.from_pars(pars, "exp")
    st = st.cpu().numpy()
    assert list(st) == [0, 0, _lib.ERR_G_RANGE, 0]
    dev = gm.to_numpy()
    assert np.all(dev[2]["p"] == 0.0)        # never written
    for f in ("p", "row", "col"):
        np.testing.assert_array_equal(dev[3][f], g["gmix_exp"][f])
    psf = np.zeros((4, 1), dtype=_lib.GAUSS2D_DTYPE)
    psf["p"] = 1.0
    psf["irr"] = psf["icc"] = 0.1
    psf["p"][1] = 0.0
    conv, cst = gm.convolve(GMixBatch.from_numpy(psf))
    assert list(cst.cpu().numpy()) == [0, _lib.ERR_ZERO_DIV, 0, 0]


def test_untracked_and_tracked_load_paths_agree_bitwise():
    """the fused kernels evaluate complete-tile stamps through hand-counted
    look-ahead loads (inline asm, invisible to the compiler) and every other
    stamp through ordinary loads; NGMIX_BATCH_TRACKED_LOADS forces the second
    path.  Same arithmetic: every output must be bit-identical, so a toolchain
    change that breaks the s_waitcnt discipline shows up here (and in
    tools/kernel_resources.py at build time)"""
    import torch
    from ngmix_amd.batch import StampBatch, GMixBatch
    for dims, ng in (((48, 48), 6), ((32, 32), 1), ((64, 64), 16), ((16, 48), 3)):
        rng = np.random.RandomState(11 + dims[0] + ng)
        n = 96
        images = rng.normal(size=(n,) + dims)
        weights = rng.uniform(0.5, 2.0, size=(n,) + dims)
        gmh = _random_mixtures(rng, n, ng, 0.263)
        jac = np.array([(dims[0] - 1) / 2, (dims[1] - 1) / 2, 0.263, 0.0, 0.0, 0.263,
                        0.263 ** 2, 0.263])
        outs = []
        for tracked in (False, True):
            sb = StampBatch.from_images(images, weights, jac)
            sb.tracked_loads = tracked
            gm = GMixBatch.from_numpy(gmh)
            ll, st = sb.loglike(gm)
            fd, _ = sb.fill_fdiff(gm)
            im, _ = sb.render(gm, fast_exp=True)
            s2, _ = sb.model_s2n_sum(gm)
            torch.cuda.synchronize()
            assert int(st.abs().sum()) == 0
            outs.append([t.cpu().numpy() for t in (ll, fd, im, s2)])
        for a, b in zip(*outs):
            np.testing.assert_array_equal(a, b)


def test_cabi_stamp_store_and_rccl_gather():
    """the library-owned forms a non-Python host uses (include/ngmix_hip.h):
    ngmix_batch_create / upload / free build the stamp store from host arrays
    (pixels.py:6-52 for N objects), and ngmix_allgather_results moves result
    records over RCCL -- here a one-rank communicator, which is as far as one
    GPU goes, but it is the real ncclAllGather on the device"""
    import torch
    from ngmix_amd.batch import StampBatch, GMixBatch
    L = _lib.lib()
    rng = np.random.RandomState(21)
    shapes = [(48, 48), (32, 32), (17, 23), (48, 48)]
    n, ng = len(shapes), 3
    images = [rng.normal(size=sh) for sh in shapes]
    weights = [rng.uniform(0.5, 2.0, size=sh) for sh in shapes]
    weights[2][3, 4] = 0.0
    weights[2][0, 0] = -1.0
    jac = np.zeros(n, dtype=_lib.JACOBIAN_DTYPE)
    for i, sh in enumerate(shapes):
        jac[i] = ((sh[0] - 1) / 2, (sh[1] - 1) / 2, 0.263, 0.01, -0.02, 0.27,
                  0.263 * 0.27 + 0.01 * 0.02, np.sqrt(0.263 * 0.27 + 0.01 * 0.02))
    gmh = _random_mixtures(rng, n, ng, 0.263)
    nrow = np.array([sh[0] for sh in shapes], dtype=np.int32)
    ncol = np.array([sh[1] for sh in shapes], dtype=np.int32)
    pb = ctypes.POINTER(_lib.Batch)()
    assert L.ngmix_batch_create(ctypes.byref(pb), n, _lib.ptr(nrow), _lib.ptr(ncol), ng, 1) == 0
    try:
        flat_im = np.concatenate([a.ravel() for a in images])
        flat_wt = np.concatenate([a.ravel() for a in weights])
        _lib.check(L.ngmix_batch_upload(pb, _lib.ptr(flat_im), _lib.ptr(flat_wt),
                                        _lib.ptr(jac), None), "upload")
        kept = np.zeros(n, dtype=np.int32)
        assert L.ngmix_batch_npix_kept(pb, _lib.ptr(kept)) == 0
        assert list(kept) == [48 * 48, 32 * 32, 17 * 23 - 2, 48 * 48]
        assert pb.contents.any_masked == 1 and pb.contents.max_npix == 48 * 48
        gm = GMixBatch.from_numpy(gmh)
        out = torch.empty((n, 4), dtype=torch.float64, device="cuda")
        st = torch.empty(n, dtype=torch.int32, device="cuda")
        _lib.check(L.ngmix_loglike_batch(pb, ctypes.c_void_p(gm.data.data_ptr()),
                                         ctypes.c_void_p(out.data_ptr()),
                                         ctypes.c_void_p(st.data_ptr()), None), "loglike")
        torch.cuda.synchronize()
        assert int(st.abs().sum()) == 0
        # the same stamps through the Python shell's torch-owned store
        ref = []
        for i in range(n):
            sb = StampBatch.from_images(images[i][None], weights[i][None],
                                        jac[i:i + 1].view("f8").reshape(1, 8))
            o, _ = sb.loglike(GMixBatch.from_numpy(gmh[i:i + 1]))
            ref.append(o.cpu().numpy()[0])
        np.testing.assert_array_equal(out.cpu().numpy(), np.array(ref))

        # ---- RCCL: a one-rank communicator and the record gather
        uid = np.zeros(128, dtype=np.uint8)
        _lib.check(L.ngmix_comm_unique_id(_lib.ptr(uid)), "unique id")
        comm = ctypes.c_void_p()
        _lib.check(L.ngmix_comm_init_rank(ctypes.byref(comm), 1, _lib.ptr(uid), 0), "init")
        try:
            rec = torch.arange(5 * 73, dtype=torch.float64, device="cuda").reshape(5, 73)
            got = torch.zeros_like(rec)
            s = torch.cuda.current_stream().cuda_stream
            _lib.check(L.ngmix_allgather_results(comm, ctypes.c_void_p(rec.data_ptr()),
                                                 ctypes.c_void_p(got.data_ptr()), 5, 584,
                                                 ctypes.c_void_p(s)), "allgather")
            torch.cuda.synchronize()
            assert torch.equal(got, rec)
            assert L.ngmix_allgather_results(comm, None, None, 3, 0, None) == _lib.ERR_BAD_ARG
        finally:
            _lib.check(L.ngmix_comm_destroy(comm), "destroy")
    finally:
        assert L.ngmix_batch_free(pb) == 0
    # a stamp without positive weight is the reference's GMixFatalError
    pb2 = ctypes.POINTER(_lib.Batch)()
    assert L.ngmix_batch_create(ctypes.byref(pb2), 1, _lib.ptr(nrow[:1]), _lib.ptr(ncol[:1]),
                                1, 1) == 0
    zero = np.zeros(48 * 48)
    assert L.ngmix_batch_upload(pb2, _lib.ptr(flat_im[:48 * 48].copy()), _lib.ptr(zero),
                                _lib.ptr(jac[:1].copy()), None) == _lib.ERR_BAD_ARG
    # no weight map (NULL): unit weights, filled on the device on the stream
    # given -- the same loglike as an explicit map of ones
    ones = np.ones(48 * 48)
    got = []
    for wt in (None, _lib.ptr(ones)):
        _lib.check(L.ngmix_batch_upload(pb2, _lib.ptr(flat_im[:48 * 48].copy()), wt,
                                        _lib.ptr(jac[:1].copy()), None), "upload")
        gm1 = GMixBatch.from_numpy(gmh[:1, :1].copy())
        o1 = torch.empty((1, 4), dtype=torch.float64, device="cuda")
        s1 = torch.empty(1, dtype=torch.int32, device="cuda")
        _lib.check(L.ngmix_loglike_batch(pb2, ctypes.c_void_p(gm1.data.data_ptr()),
                                         ctypes.c_void_p(o1.data_ptr()),
                                         ctypes.c_void_p(s1.data_ptr()), None), "loglike")
        torch.cuda.synchronize()
        got.append(o1.cpu().numpy())
    assert got[0][0, 3] == 48 * 48
    np.testing.assert_array_equal(got[0], got[1])
    L.ngmix_batch_free(pb2)
    # stamp.gm_off is an int32 index: a store whose mixtures would not fit is
    # refused before anything is allocated
    pb3 = ctypes.POINTER(_lib.Batch)()
    assert L.ngmix_batch_create(ctypes.byref(pb3), 2, _lib.ptr(nrow[:2].copy()),
                                _lib.ptr(ncol[:2].copy()), 2 ** 30, 1) == _lib.ERR_BAD_ARG
    assert "2^31" in _lib.last_error()


@pytest.mark.parametrize("exact", [False, True], ids=["fused", "exact"])
def test_render_overwrite_equals_render_into_zeros(exact):
    """a fresh render (image=None: NGMIX_BATCH_RENDER_OVERWRITE, the buffer is
    never read) is bit for bit the accumulate-into render of a zeroed image,
    GMix.make_image's two steps (gmix.py:561-562, 619-643); stamps that raise
    and empty mixtures come out zero-filled"""
    import torch
    from ngmix_amd.batch import StampBatch, GMixBatch
    rng = np.random.RandomState(31)
    shapes = [(48, 48)] * 5 + [(17, 23), (32, 32), (8, 16), (33, 9)]
    n, ng = len(shapes), 4
    nrow = np.array([sh[0] for sh in shapes])
    ncol = np.array([sh[1] for sh in shapes])
    pix_off = np.concatenate([[0], np.cumsum(nrow * ncol)[:-1]]).astype(np.int64)
    jac = np.zeros((n, 8))
    for i, sh in enumerate(shapes):
        jac[i] = [(sh[0] - 1) / 2, (sh[1] - 1) / 2, 0.263, 0.01, -0.01, 0.27,
                  0.263 * 0.27 + 1e-4, np.sqrt(0.263 * 0.27 + 1e-4)]
    gmh = _random_mixtures(rng, n, ng, 0.263)
    gmh["det"][3, 2] = 1e-250          # this stamp raises
    gmh["norm_set"][3] = 0
    for fast in (True, False):
        sb = StampBatch(None, None, torch.from_numpy(jac).cuda(), nrow, ncol, pix_off, True)
        zeros = torch.zeros(sb.total_pix, dtype=torch.float64, device=sb.device)
        a, sa = sb.render(GMixBatch.from_numpy(gmh), image=zeros, fast_exp=fast, exact=exact)
        # poison the allocator's next block so that a read of the buffer shows
        junk = torch.full((sb.total_pix,), float("nan"), dtype=torch.float64,
                          device=sb.device)
        del junk
        b, sbt = sb.render(GMixBatch.from_numpy(gmh), fast_exp=fast, exact=exact)
        torch.cuda.synchronize()
        assert list(sa.cpu().numpy()) == list(sbt.cpu().numpy())
        assert int(sa[3]) == _lib.ERR_DET_TOO_LOW and int(sa.abs().sum()) == _lib.ERR_DET_TOO_LOW
        a, b = a.cpu().numpy(), b.cpu().numpy()
        assert np.all(np.isfinite(b))
        np.testing.assert_array_equal(a, b)
        off = sb.pix_off
        assert np.all(b[off[3]:off[3] + 48 * 48] == 0.0)
        assert np.abs(b[off[0]:off[0] + 48 * 48]).max() > 0
    # an empty mixture
    sb1 = StampBatch.from_images(np.zeros((2, 16, 16)))
    empty = GMixBatch.empty(2, 0)
    im, st = sb1.render(empty)
    assert float(im.abs().sum()) == 0.0 and int(st.abs().sum()) == 0


# ------------------------------------ properties at BASELINE's full stamp size
def _c2_batch(n, seed=5):
    """SURVEY.md 8(d) C2-style synthetic batch: 48x48, 'exp' x gaussian psf"""
    from ngmix_amd.batch import StampBatch, GMixBatch
    import torch
    rng = np.random.RandomState(seed)
    scale = 0.263
    pars = np.zeros((n, 6))
    pars[:, 0:2] = rng.uniform(-0.5, 0.5, size=(n, 2)) * scale
    g = np.clip(rng.normal(scale=0.1, size=(n, 2)), -0.45, 0.45)
    pars[:, 2:4] = g
    pars[:, 4] = rng.uniform(0.3, 1.5, size=n)
    pars[:, 5] = rng.uniform(50, 500, size=n)
    psfpars = np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (n, 1))
    gm0, _ = GMixBatch.from_pars(pars, "exp")
    psf, _ = GMixBatch.from_pars(psfpars, "gauss")
    gm, _ = gm0.convolve(psf)
    jac = np.array([23.5, 23.5, scale, 0.0, 0.0, scale, scale ** 2, scale])
    zeros = torch.zeros((n, 48, 48), dtype=torch.float64, device="cuda")
    sb = StampBatch.from_images(zeros, None, jac)
    truth, _ = sb.render(gm, fast_exp=True)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    noise = torch.randn(truth.shape, generator=gen, device="cuda",
                        dtype=torch.float64)
    sigma = torch.from_numpy(0.01 * pars[:, 5] / 100).cuda()
    img = truth.reshape(n, -1) + noise.reshape(n, -1) * sigma[:, None]
    w = (1.0 / sigma ** 2)[:, None, None].expand(n, 48, 48).contiguous()
    sb = StampBatch.from_images(img.reshape(n, 48, 48), w, jac)
    return sb, gm, pars


@pytest.mark.parametrize("exact", [True, False], ids=["exact", "fused"])
def test_full_size_properties(exact):
    import torch
    n = 4096
    sb, gm, pars = _c2_batch(n)
    out, status = sb.loglike(gm, exact=exact)
    assert int(status.abs().sum()) == 0
    # (1) exact skipping: bitwise identical with and without tile skipping
    out_ns, _ = sb.loglike(gm, no_skip=True, exact=exact)
    assert torch.equal(out, out_ns)
    fd, _ = sb.fill_fdiff(gm, exact=exact)
    fd_ns, _ = sb.fill_fdiff(gm, no_skip=True, exact=exact)
    assert torch.equal(fd, fd_ns)
    im, _ = sb.render(gm, exact=exact)
    im_ns, _ = sb.render(gm, no_skip=True, exact=exact)
    assert torch.equal(im, im_ns)
    # (2) loglike == -0.5 * sum(fdiff^2) per stamp (gmix_nb.py:866,900)
    chk = -0.5 * (fd.reshape(n, -1) ** 2).sum(dim=1)
    np.testing.assert_allclose(out[:, 0].cpu().numpy(), chk.cpu().numpy(),
                               rtol=1e-11, atol=0)
    assert torch.all(out[:, 3] == 2304)
    # chi2 per pixel ~ 1 at the true parameters
    chi2per = (-2 * out[:, 0] / 2304).cpu().numpy()
    assert 0.9 < chi2per.mean() < 1.1
    # (3) render linearity: scaling p by 2 (exact in binary) doubles the image
    gm2 = gm.clone()
    gm2.data[:, 0] *= 2.0
    gm2.data[:, 7] = 0.0  # norm_set = 0 -> lazy norms recomputed
    im2, _ = sb.render(gm2, exact=exact)
    assert torch.equal(im2, 2.0 * im)
    # (4) determinism: same launch twice gives the same bits
    out_b, _ = sb.loglike(gm, exact=exact)
    assert torch.equal(out, out_b)
    # (4b) the two kernels agree to rounding on every stamp
    out_x, _ = sb.loglike(gm, exact=not exact)
    np.testing.assert_allclose(out.cpu().numpy(), out_x.cpu().numpy(), rtol=1e-11)
    # (5) a sample of stamps against the CPU oracle
    from oracle import oracle as ora
    gmh = gm.to_numpy()
    val = sb.val.reshape(n, 48, 48).cpu().numpy()
    ierr = sb.ierr.reshape(n, 48, 48).cpu().numpy()
    jac = sb.jac.cpu().numpy()
    outh = out.cpu().numpy()
    fdh = fd.reshape(n, -1).cpu().numpy()
    for i in (0, 1, 777, n - 1):
        res, rfd, _ = _oracle_eval(gmh[i], val[i], ierr[i] ** 2, jac[i], True)
        np.testing.assert_allclose(outh[i, :3], res[:3], rtol=1e-10, atol=0)
        # ierr**2 then sqrt is not always the identity: compare fdiff to rounding
        np.testing.assert_allclose(fdh[i], rfd, rtol=1e-12, atol=1e-12)


def test_multi_epoch_bdf_loglike_vs_oracle():
    """BASELINE config 5 in small: objects with several 64x64 epochs (sub-pixel
    jacobian offsets per epoch, as ngmix/tests/_sims.py:150-159), 16-gaussian
    'bdf' (x) 1-gaussian psf, loglike summed over the epochs of each object"""
    from ngmix_amd.batch import StampBatch, GMixBatch
    from oracle import oracle as ora
    rng = np.random.RandomState(99)
    nobj, nepoch, dim, scale = 3, 4, 64, 0.263
    ns = nobj * nepoch
    pars = np.zeros((nobj, 7))
    pars[:, 0:2] = rng.uniform(-0.3, 0.3, size=(nobj, 2)) * scale
    pars[:, 2:4] = rng.normal(scale=0.08, size=(nobj, 2))
    pars[:, 4] = rng.uniform(0.5, 2.0, size=nobj)
    pars[:, 5] = rng.uniform(0.2, 0.8, size=nobj)   # fracdev
    pars[:, 6] = rng.uniform(100, 400, size=nobj)
    spars = np.repeat(pars, nepoch, axis=0)
    jac = np.zeros((ns, 8))
    for s in range(ns):
        jac[s] = [(dim - 1) / 2 + rng.uniform(-0.5, 0.5),
                  (dim - 1) / 2 + rng.uniform(-0.5, 0.5), scale, 0.0, 0.0, scale,
                  scale ** 2, scale]
    psfpars = np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (ns, 1))
    gm0, st0 = GMixBatch.from_pars(spars, "bdf")
    psf, _ = GMixBatch.from_pars(psfpars, "gauss")
    gm, _ = gm0.convolve(psf)
    assert gm.ngauss == 16 and int(st0.abs().sum()) == 0
    geom = StampBatch.from_images(np.zeros((ns, dim, dim)), None, jac)
    truth, _ = geom.render(gm)
    images = truth.cpu().numpy().reshape(ns, dim, dim) + 0.05 * rng.normal(size=(ns, dim, dim))
    weights = np.full((ns, dim, dim), 400.0)
    sb = StampBatch.from_images(images, weights, jac)
    obj_start = np.arange(nobj + 1) * nepoch
    per_obj, per_stamp, status = sb.loglike_objects(gm, obj_start)
    assert int(status.abs().sum()) == 0
    per_obj = per_obj.cpu().numpy()
    gmh = gm.to_numpy()
    ref = np.zeros((nobj, 4))
    for s in range(ns):
        g = np.zeros(16, dtype=ora.GAUSS2D_DTYPE)
        for name in ora.GAUSS2D_DTYPE.names:
            g[name] = gmh[s][name]
        j = np.zeros(1, dtype=ora.JACOBIAN_DTYPE)
        j[0] = tuple(jac[s])
        st, res = ora.get_loglike(g, ora.make_pixels(images[s], weights[s], j, True))
        assert st == 0
        ref[s // nepoch] += res
    np.testing.assert_allclose(per_obj[:, 0], ref[:, 0], rtol=1e-10)
    np.testing.assert_allclose(per_obj[:, 1:3], ref[:, 1:3], rtol=1e-10)
    assert np.all(per_obj[:, 3] == ref[:, 3])
    # ragged epochs per object take the segmented path
    rag_start = np.array([0, 3, 8, 12])
    per_obj2, _, _ = sb.loglike_objects(gm, rag_start)
    ps = per_stamp.cpu().numpy()
    for i in range(3):
        np.testing.assert_allclose(per_obj2.cpu().numpy()[i],
                                   ps[rag_start[i]:rag_start[i + 1]].sum(axis=0),
                                   rtol=1e-14)


def test_batch_edge_shapes_and_empty_inputs():
    """1x1 and 1xN stamps, a stamp with more tiles than the fused kernel's
    LDS table holds (falls back to the exact kernel), a stamp whose pixels are
    all masked, an empty batch, NaN parameters -- against the oracle"""
    import ngmix_amd as ngmix
    from ngmix_amd.batch import StampBatch, GMixBatch
    rng = np.random.RandomState(99)
    scale = 0.263
    shapes = [(1, 1), (1, 37), (53, 1), (3, 5), (400, 400), (16, 16)]
    obs, gms = [], []
    for k, (nrow, ncol) in enumerate(shapes):
        im = rng.normal(size=(nrow, ncol))
        wt = rng.uniform(0.5, 2.0, size=(nrow, ncol))
        if k == 5:
            wt[:] = 0.0
            wt[7, 9] = 1.3   # Observation refuses all-zero weights: keep one, mask below
        jac = ngmix.Jacobian(row=(nrow - 1) / 2.0 + 0.2, col=(ncol - 1) / 2.0 - 0.1,
                             dvdrow=scale, dvdcol=0.01, dudrow=-0.02, dudcol=scale * 1.03)
        obs.append(ngmix.Observation(im, weight=wt, jacobian=jac))
    gmh = _random_mixtures(rng, len(shapes), 4, scale)
    sb = StampBatch.from_observations(obs)
    gm = GMixBatch.from_numpy(gmh)
    out, status = sb.loglike(gm)
    fd, _ = sb.fill_fdiff(gm)
    im, _ = sb.render(gm)
    assert np.all(status.cpu().numpy() == 0)
    out, fd, im = out.cpu().numpy(), fd.cpu().numpy(), im.cpu().numpy()
    offs = sb.kept_offsets()
    for i, o in enumerate(obs):
        jrec = o.jacobian.get_data().view(np.float64).reshape(8)
        res, rfd, rim = _oracle_eval(gmh[i], o.image, o.weight, jrec, True)
        assert out[i, 3] == res[3]
        np.testing.assert_allclose(out[i, 0], res[0], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(fd[offs[i]:offs[i] + res[3]], rfd, rtol=1e-9,
                                   atol=1e-10 * max(np.abs(rfd).max(), 1.0))
        a = int(sb.pix_off[i])
        np.testing.assert_allclose(im[a:a + rim.size].reshape(rim.shape), rim, rtol=1e-9,
                                   atol=1e-12 * np.abs(rim).max())
    # an empty batch is a no-op, not an error
    empty = StampBatch.from_images(np.zeros((0, 8, 8)), None, None)
    egm = GMixBatch.empty(0, 4)
    eo, es = empty.loglike(egm)
    assert eo.shape[0] == 0 and es.shape[0] == 0
    # NaN parameters: chi2 is NaN for every pixel, which fails the reference's
    # `chi2 < 25 and chi2 >= 0` gate, so the model is 0 there
    # (gmix_nb.py:52-63) and loglike = -sum(val^2 ivar)/2; the other stamps
    # are untouched
    pars = np.tile([0.0, 0.0, 0.1, 0.0, 0.5, 10.0], (3, 1))
    pars[1, 4] = np.nan
    g3, st3 = GMixBatch.from_pars(pars, "exp")
    ims = rng.normal(size=(3, 16, 16))
    ims[2] = ims[0]
    sb3 = StampBatch.from_images(ims, None, None)
    o3, s3 = sb3.loglike(g3)
    o3 = o3.cpu().numpy()
    assert np.all(s3.cpu().numpy() == 0)
    np.testing.assert_allclose(o3[1, 0], -0.5 * (ims[1] ** 2).sum(), rtol=1e-13)
    assert o3[0, 0] == o3[2, 0] and o3[1, 2] == 0.0


def test_batch_beyond_4_gib():
    """64-bit addressing: pixel arrays larger than 4 GiB (the C2 workload
    tiled 15 times, 300k stamps); every block of stamps must equal the small
    batch bit for bit in loglike, and the last one in fill_fdiff and render"""
    import os
    import sys
    import torch
    from ngmix_amd.batch import StampBatch, GMixBatch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    dev = torch.device("cuda", 0)
    base, reps, npix = 20000, 15, 48 * 48
    n = base * reps
    sb0, gm0, _ = bench.make_workload(base, seed=7, device=dev)
    sb = StampBatch(sb0.val.repeat(reps), sb0.ierr.repeat(reps), sb0.jac.repeat(reps, 1),
                    np.full(n, 48), np.full(n, 48), np.arange(n, dtype=np.int64) * npix, True)
    assert sb.val.numel() * 8 > 2 ** 32
    gm = GMixBatch(gm0.data.repeat(reps, 1), n, gm0.ngauss)
    ref = sb0.loglike(gm0)[0].cpu().numpy()
    out, st = sb.loglike(gm)
    assert int(st.abs().sum()) == 0
    out = out.cpu().numpy().reshape(reps, base, 4)
    for r in range(reps):
        assert np.array_equal(out[r], ref), r
    fd0 = sb0.fill_fdiff(gm0)[0].cpu().numpy()
    assert np.array_equal(sb.fill_fdiff(gm)[0][-base * npix:].cpu().numpy(), fd0)
    im0 = sb0.render(gm0)[0].cpu().numpy()
    assert np.array_equal(sb.render(gm)[0][-base * npix:].cpu().numpy(), im0)


def test_c5_shaped_batch_against_the_reference(golden):
    """tests/golden/c5.npz (oracle/gen_golden_c5.py; inputs rebuilt by
    helpers/c5_inputs.py): six objects of ten 64x64 epochs, the 16-gaussian
    'bdf' (x) gaussian psf -- config 5's shape -- through the REFERENCE's
    GMix.get_loglike, against ONE batch of sixty stamps summed over each
    object's epochs as bench.py's C5 step does: loglike / s2n_numer / s2n_denom
    to 1e-10 per epoch and per object, npix exact, with the census naming
    the kernel of the C5 leg"""
    from helpers import c5_inputs as c5
    from ngmix_amd.batch import StampBatch, GMixBatch
    g = golden("c5")
    pars, moved, jac, images = c5.objects()
    np.testing.assert_allclose(images.sum(axis=(2, 3)), g["image_sums"], rtol=1e-13)
    ns = c5.NOBJ * c5.NEPOCH
    weights = np.full((ns, c5.DIM, c5.DIM), 1.0 / c5.NOISE ** 2)
    sb = StampBatch.from_images(images.reshape(ns, c5.DIM, c5.DIM), weights, jac.reshape(ns, 8))
    psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, c5.TPSF, 1.0], (ns, 1)), "gauss")
    obj_start = np.arange(c5.NOBJ + 1) * c5.NEPOCH
    for tag, pp in (("truth", pars), ("moved", moved)):
        gm0, st = GMixBatch.from_pars(np.repeat(pp, c5.NEPOCH, axis=0), "bdf")
        gm, _ = gm0.convolve(psf)
        _lib.launch_census(reset=True)
        out, status = sb.loglike(gm)
        seen = _lib.launch_census(reset=True)
        assert any(k.startswith("pixpass_wave_kernel7<loglike>") for k in seen), seen
        assert int(status.abs().sum()) == 0
        per_obj = sb.sum_over_epochs(out, obj_start).cpu().numpy()
        out = out.cpu().numpy().reshape(c5.NOBJ, c5.NEPOCH, 4)
        np.testing.assert_allclose(out[:, :, :3], g[tag + "_per_epoch"][:, :, :3], rtol=1e-10)
        np.testing.assert_array_equal(out[:, :, 3], g[tag + "_per_epoch"][:, :, 3])
        np.testing.assert_allclose(per_obj[:, :3], g[tag + "_per_object"][:, :3], rtol=1e-10)
        np.testing.assert_array_equal(per_obj[:, 3], g[tag + "_per_object"][:, 3])


@pytest.mark.parametrize("exact", [True, False], ids=["exact", "fused"])
def test_c2_shaped_batch_against_the_reference(golden, exact):
    """tests/golden/c2.npz (oracle/gen_golden_c2.py; inputs rebuilt by
    helpers/c2_inputs.py): eight stamps of config 2's shape through the
    REFERENCE's get_loglike / fill_fdiff / _fill_image (accumulating into a
    given image, fast exp) at two parameter sets, against ONE batch of the
    kernels from the model parameters on (device model fill, convolution,
    pixel pass): the four loglike numbers to 1e-10, fdiff and the rendered
    images to north_star's per-pixel tolerance (the model fill goes through
    tanh / atanh: the exact kernels are held to it too, not to the bit)"""
    import torch
    from helpers import c2_inputs as c2
    from ngmix_amd.batch import StampBatch, GMixBatch
    g = golden("c2")
    pars, moved, jac, images, sigma, base = c2.stamps()
    np.testing.assert_allclose(images.sum(axis=(1, 2)), g["image_sums"], rtol=1e-13)
    weights = np.broadcast_to((1.0 / sigma ** 2)[:, None, None], images.shape).copy()
    sb = StampBatch.from_images(images, weights, jac)
    psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, c2.TPSF, 1.0], (c2.N, 1)), "gauss")
    for tag, pp in (("truth", pars), ("moved", moved)):
        gm0, _ = GMixBatch.from_pars(pp, "exp")
        gm, _ = gm0.convolve(psf)
        _lib.launch_census(reset=True)
        out, status = sb.loglike(gm, exact=exact)
        seen = _lib.launch_census(reset=True)
        if not exact:
            assert seen.get("pixpass_wave_kernel7<loglike>", 0) == 1, seen
        assert int(status.abs().sum()) == 0
        out = out.cpu().numpy()
        np.testing.assert_allclose(out[:, :3], g[tag + "_loglike"][:, :3], rtol=1e-10)
        np.testing.assert_array_equal(out[:, 3], g[tag + "_loglike"][:, 3])
        fd, _ = sb.fill_fdiff(gm, exact=exact)
        fd = fd.cpu().numpy().reshape(c2.N, -1)
        acc = torch.from_numpy(base.reshape(-1).copy()).cuda()
        sb.render(gm, image=acc, fast_exp=True, exact=exact)
        acc = acc.cpu().numpy().reshape(c2.N, -1)
        for i in range(c2.N):
            assert_pixels(fd[i], g[tag + "_fdiff"][i], False, err_msg="fdiff %d" % i)
            # (the render adds to a base of order one: relative to the model's peak)
            ref = g[tag + "_rendered"][i].ravel()
            model = ref - base[i].ravel()
            np.testing.assert_allclose(acc[i] - base[i].ravel(), model, rtol=0,
                                       atol=1e-10 * np.abs(model).max(), err_msg="render %d" % i)
