"""
GPU parity tests for the moment / iterative kernels: weighted sums (6 and 17
moments), adaptive moments, EM (4 kinds) and deriv_images -- seam forms on the
reference's own AoS inputs and batch forms on the compact stamp store, both
against goldens produced by the reference itself.

Tolerances: these kernels reduce over pixels in a tree, not sequentially, and
admom / EM iterate on those sums, so results agree to rounding, not bitwise:
1e-10 relative (north_star) on converged quantities, exact on numiter / flags
for these fixtures.  deriv_images has no reduction and is compared exactly;
weighted sums keep the reference's summation order (exact up to device exp()).
"""
import ctypes

import numpy as np
import pytest

from ngmix_amd import _lib

pytestmark = pytest.mark.gpu

RTOL = 1e-10


def conv_rec(a, dtype):
    out = np.zeros(a.size, dtype=dtype)
    for n in dtype.names:
        if n in a.dtype.names:   # (a reference record has no batch-extension fields)
            out[n] = a[n]
    return out


def as_gauss(a):
    return conv_rec(a, _lib.GAUSS2D_DTYPE)


def as_pixels(a):
    return conv_rec(a, _lib.PIXEL_DTYPE)


def close(a, b, rtol=RTOL, scale=None, err_msg=""):
    a = np.asarray(a, dtype="f8")
    b = np.asarray(b, dtype="f8")
    if scale is None:
        scale = np.abs(b).max() if b.size else 0.0
    np.testing.assert_allclose(a, b, rtol=rtol, atol=rtol * scale, err_msg=err_msg)


# ------------------------------------------------------------ weighted sums
def test_weighted_sums_seam_and_batch(golden):
    from ngmix_amd.batch import StampBatch, GMixBatch, records_to_numpy
    g = golden("wsums")
    L = _lib.lib()
    for name in [str(n) for n in g["names"]]:
        ref = g[name + "_res"]
        nmom = ref["sums"].shape[-1]
        dt = _lib.moments_result_dtype(nmom)
        wt = as_gauss(g[name + "_wt"])
        pixels = as_pixels(g[name + "_pixels"])
        maxrad = float(g[name + "_maxrad"])
        res = np.zeros(1, dtype=dt)
        assert L.ngmix_get_weighted_sums(_lib.ptr(wt), wt.size, _lib.ptr(pixels),
                                         pixels.size, _lib.ptr(res), nmom,
                                         maxrad) == 0, name
        sb = StampBatch.from_images(g[name + "_image"], g[name + "_weight"],
                                    g[name + "_jac"],
                                    ignore_zero_weight=bool(g[name + "_izw"]))
        bres, status = sb.weighted_sums(GMixBatch.from_numpy(wt), maxrad, nmom=nmom,
                                        exact=True)
        assert int(status.cpu()[0]) == 0
        bres = records_to_numpy(bres, dt)
        for got in (res, bres):
            assert got["npix"][0] == ref["npix"][0], name
            # same summation order as the reference; only exp() differs (ulps)
            for f in ("wsum", "sums", "sums_cov"):
                close(got[f][0], ref[f][0], rtol=1e-13, err_msg="%s %s" % (name, f))
            np.testing.assert_allclose(got["F"][0], ref["F"][0], rtol=1e-15, atol=0)
        # seam and exact batch forms run the same arithmetic in the same order
        for f in ("wsum", "sums", "sums_cov", "F"):
            np.testing.assert_array_equal(res[f], bres[f])
        # default batch form: register accumulators + tree, FMA arithmetic:
        # every sum to 1e-12 of its largest term's scale (sum of |terms| is
        # bounded by sqrt(cov_ii cov_jj) for the covariance, Cauchy-Schwarz)
        fres, status = sb.weighted_sums(GMixBatch.from_numpy(wt), maxrad, nmom=nmom)
        assert int(status.cpu()[0]) == 0
        fres = records_to_numpy(fres, dt)
        assert fres["npix"][0] == ref["npix"][0], name
        np.testing.assert_allclose(fres["wsum"][0], ref["wsum"][0], rtol=1e-12)
        cov = ref["sums_cov"][0]
        dscale = np.sqrt(np.abs(np.diag(cov)))
        assert np.all(np.abs(fres["sums_cov"][0] - cov) <=
                      1e-12 * np.outer(dscale, dscale)), name
        # sums[i] = sum w val F_i: |terms| <= sqrt(sum (w val)^2 * sum F_i^2)
        assert np.all(np.abs(fres["sums"][0] - ref["sums"][0]) <=
                      1e-11 * np.maximum(np.abs(ref["sums"][0]).max(), 1e-300)), name
        np.testing.assert_allclose(fres["F"][0], ref["F"][0], rtol=1e-14, atol=0)
        # accumulate-into: a second call doubles
        assert L.ngmix_get_weighted_sums(_lib.ptr(wt), wt.size, _lib.ptr(pixels),
                                         pixels.size, _lib.ptr(res), nmom,
                                         maxrad) == 0
        close(res["sums"][0], 2 * ref["sums"][0], rtol=1e-13)
        assert res["npix"][0] == 2 * ref["npix"][0]


def test_weighted_sums_zero_ierr():
    L = _lib.lib()
    wt = np.zeros(1, dtype=_lib.GAUSS2D_DTYPE)
    wt["p"] = wt["irr"] = wt["icc"] = wt["det"] = 1.0
    assert L.ngmix_set_norms(_lib.ptr(wt), 1) == 0
    pix = np.zeros(3, dtype=_lib.PIXEL_DTYPE)
    pix["area"] = 1.0
    pix["ierr"] = [1.0, 0.0, 1.0]
    res = np.zeros(1, dtype=_lib.moments_result_dtype(17))
    assert L.ngmix_get_weighted_sums(_lib.ptr(wt), 1, _lib.ptr(pix), 3,
                                     _lib.ptr(res), 17, 100.0) == _lib.ERR_ZERO_DIV
    res = np.zeros(1, dtype=_lib.moments_result_dtype(6))
    assert L.ngmix_get_weighted_sums(_lib.ptr(wt), 1, _lib.ptr(pix), 3,
                                     _lib.ptr(res), 6, 100.0) == 0
    assert res["npix"][0] == 2


# -------------------------------------------------------------------- admom
def _check_admom(name, res, wt, ref, ref_wt):
    assert res["flags"][0] == ref["flags"][0], name
    assert res["numiter"][0] == ref["numiter"][0], name
    assert res["npix"][0] == ref["npix"][0], name
    close(res["wsum"][0], ref["wsum"][0], err_msg=name)
    close(res["sums"][0], ref["sums"][0], err_msg=name + " sums")
    close(res["sums_cov"][0], ref["sums_cov"][0], err_msg=name + " cov")
    np.testing.assert_array_equal(np.isnan(res["pars"][0]), np.isnan(ref["pars"][0]))
    if not np.isnan(ref["pars"][0]).any():
        close(res["pars"][0], ref["pars"][0], err_msg=name + " pars")
        close(res["rho4"][0], ref["rho4"][0], err_msg=name + " rho4")
    close(res["F"][0], ref["F"][0], rtol=1e-9, err_msg=name + " F")
    for f in ("row", "col", "irr", "irc", "icc", "det"):
        close(wt[f], ref_wt[f], scale=1.0, err_msg=name + " wt " + f)


def test_admom_seam_and_batch(golden):
    from ngmix_amd.batch import StampBatch, GMixBatch, records_to_numpy
    g = golden("admom")
    L = _lib.lib()
    for name in [str(n) for n in g["names"]]:
        conf = conv_rec(g[name + "_conf"], _lib.ADMOM_CONF_DTYPE)
        ref = g[name + "_res"]
        ref_wt = g[name + "_wt_out"]
        # seam form on the reference's pixel array
        wt = as_gauss(g[name + "_wt_in"])
        pixels = as_pixels(g[name + "_pixels"])
        res = np.zeros(1, dtype=_lib.ADMOM_RESULT_DTYPE)
        st = L.ngmix_admom(_lib.ptr(conf), _lib.ptr(wt), _lib.ptr(pixels),
                           pixels.size, _lib.ptr(res))
        assert st == 0, name
        _check_admom(name + " seam", res, wt, ref, ref_wt)
        # batch form on the compact store
        sb = StampBatch.from_images(g[name + "_image"], g[name + "_weight"],
                                    g[name + "_jac"],
                                    ignore_zero_weight=bool(g[name + "_izw"]))
        wtb = GMixBatch.from_numpy(as_gauss(g[name + "_wt_in"]))
        bres, status = sb.admom(wtb, maxiter=int(conf["maxiter"][0]),
                                shiftmax=float(conf["shiftmax"][0]),
                                etol=float(conf["etol"][0]),
                                Ttol=float(conf["Ttol"][0]),
                                cenonly=bool(conf["cenonly"][0]))
        assert int(status.cpu()[0]) == 0, name
        bres = records_to_numpy(bres, _lib.ADMOM_RESULT_DTYPE)
        _check_admom(name + " batch", bres, wtb.to_numpy()[0], ref, ref_wt)


def test_admom_batch_many_vs_oracle():
    """32x32 config-4-like stamps: every record against the CPU oracle"""
    from ngmix_amd.batch import StampBatch, GMixBatch, records_to_numpy
    from oracle import oracle as ora
    rng = np.random.RandomState(31)
    n, dim, scale = 48, 32, 0.263
    pars = np.zeros((n, 6))
    pars[:, 0:2] = rng.uniform(-0.5, 0.5, size=(n, 2)) * scale
    pars[:, 2:4] = rng.normal(scale=0.1, size=(n, 2))
    pars[:, 4] = rng.uniform(0.3, 0.9, size=n) + 0.27
    pars[:, 5] = rng.uniform(50, 200, size=n)
    gm, _ = GMixBatch.from_pars(pars, "gauss")
    jac = np.array([15.5, 15.5, scale, 0, 0, scale, scale ** 2, scale])
    sb0 = StampBatch.from_images(np.zeros((n, dim, dim)), None, jac)
    truth, _ = sb0.render(gm)
    images = truth.cpu().numpy().reshape(n, dim, dim) + \
        rng.normal(scale=0.01, size=(n, dim, dim))
    weights = np.full((n, dim, dim), 1e4)
    weights[3, 5, 5] = 0.0
    sb = StampBatch.from_images(images, weights, jac)
    guess = np.zeros((n, 6))
    guess[:, 4] = pars[:, 4] * rng.uniform(0.9, 1.1, size=n)
    guess[:, 5] = 1.0
    wt, _ = GMixBatch.from_pars(guess, "gauss")
    wt_in = wt.to_numpy()
    res, status = sb.admom(wt)
    assert int(status.abs().sum()) == 0
    res = records_to_numpy(res, _lib.ADMOM_RESULT_DTYPE)
    wt_out = wt.to_numpy()
    j = np.zeros(1, dtype=ora.JACOBIAN_DTYPE)
    j[0] = tuple(jac)
    conf = np.zeros(1, dtype=ora.ADMOM_CONF_DTYPE)
    conf["maxiter"], conf["shiftmax"], conf["etol"], conf["Ttol"] = 200, 5.0, 1e-5, 1e-3
    for i in range(n):
        pix = ora.make_pixels(images[i], weights[i], j, True)
        w = conv_rec(wt_in[i], ora.GAUSS2D_DTYPE)
        r = np.zeros(1, dtype=ora.ADMOM_RESULT_DTYPE)
        assert ora.admom(conf, w, pix, r) == 0
        _check_admom("stamp %d" % i, res[i:i + 1], wt_out[i], r, w)
        assert r["flags"][0] == 0 and 3 <= r["numiter"][0] <= 30


def test_admom_errors():
    """ierr == 0 in the list -> ZeroDivisionError; maxiter=0 -> MAXITER"""
    L = _lib.lib()
    conf = np.zeros(1, dtype=_lib.ADMOM_CONF_DTYPE)
    conf["maxiter"], conf["shiftmax"], conf["etol"], conf["Ttol"] = 10, 5.0, 1e-5, 1e-3
    wt = np.zeros(1, dtype=_lib.GAUSS2D_DTYPE)
    wt["p"] = 1.0
    wt["irr"] = wt["icc"] = 2.0
    wt["det"] = 4.0
    pix = np.zeros(25, dtype=_lib.PIXEL_DTYPE)
    pix["v"] = np.repeat(np.arange(5.0) - 2, 5)
    pix["u"] = np.tile(np.arange(5.0) - 2, 5)
    pix["area"] = 1.0
    pix["val"] = np.exp(-0.25 * (pix["v"] ** 2 + pix["u"] ** 2))
    pix["ierr"] = 1.0
    pix["ierr"][7] = 0.0
    res = np.zeros(1, dtype=_lib.ADMOM_RESULT_DTYPE)
    st = L.ngmix_admom(_lib.ptr(conf), _lib.ptr(wt.copy()), _lib.ptr(pix), 25,
                       _lib.ptr(res))
    assert st == _lib.ERR_ZERO_DIV
    conf["maxiter"] = 0
    res = np.zeros(1, dtype=_lib.ADMOM_RESULT_DTYPE)
    st = L.ngmix_admom(_lib.ptr(conf), _lib.ptr(wt.copy()), _lib.ptr(pix), 25,
                       _lib.ptr(res))
    assert st == 0 and res["numiter"][0] == 0 and res["flags"][0] == 32


# ----------------------------------------------------------------------- em
def _check_em(name, numiter, frac, sky, gm, conv, g, label=""):
    assert numiter == int(g[name + "_numiter"]), name
    assert abs(frac - float(g[name + "_frac_diff"])) <= \
        1e-6 * abs(float(g[name + "_frac_diff"])) + 1e-11, name
    close(sky, float(g[name + "_sky"]), scale=1.0, err_msg=name)
    for f in ("p", "row", "col", "irr", "irc", "icc", "det"):
        close(gm[f], g[name + "_gmix_out"][f], err_msg="%s gmix %s" % (name, f),
              scale=max(np.abs(g[name + "_gmix_out"][f]).max(), 1e-3))
        close(conv[f], g[name + "_conv_out"][f], err_msg="%s conv %s" % (name, f),
              scale=max(np.abs(g[name + "_conv_out"][f]).max(), 1e-3))
    assert np.all(gm["norm_set"] == 0)
    assert np.all(conv["norm_set"] == 1)


def test_em_seam_and_batch(golden):
    from ngmix_amd.batch import StampBatch, GMixBatch
    g = golden("em")
    L = _lib.lib()
    for name in [str(n) for n in g["names"]]:
        kind = int(g[name + "_kind"])
        conf = conv_rec(g[name + "_conf"], _lib.EM_CONF_DTYPE)
        fzw = bool(g[name + "_fzw"])
        # seam
        pixels = as_pixels(g[name + "_pixels"])
        gm = as_gauss(g[name + "_gmix_in"])
        psf = as_gauss(g[name + "_psf_in"])
        conv = as_gauss(g[name + "_conv_in"])
        sums = np.zeros((gm.size, _lib.EM_SUMS_NDOUBLE[kind]))
        numiter = ctypes.c_int32()
        frac, sky = ctypes.c_double(), ctypes.c_double()
        st = L.ngmix_em_run(kind, _lib.ptr(conf), _lib.ptr(pixels), pixels.size,
                            _lib.ptr(sums), _lib.ptr(gm), gm.size, _lib.ptr(psf),
                            psf.size, _lib.ptr(conv), int(fzw),
                            ctypes.byref(numiter), ctypes.byref(frac),
                            ctypes.byref(sky))
        assert st == 0, name
        _check_em(name, numiter.value, frac.value, sky.value, gm, conv, g)
        # batch
        sb = StampBatch.from_images(g[name + "_image"], g[name + "_weight"],
                                    g[name + "_jac"],
                                    ignore_zero_weight=bool(g[name + "_izw"]))
        gmb = GMixBatch.from_numpy(as_gauss(g[name + "_gmix_in"]))
        psfb = GMixBatch.from_numpy(as_gauss(g[name + "_psf_in"]))
        convb = GMixBatch.from_numpy(as_gauss(g[name + "_conv_in"]))
        out, status, _ = sb.em(gmb, psfb, convb, sky=float(conf["sky"][0]),
                               kind=kind, miniter=int(conf["miniter"][0]),
                               maxiter=int(conf["maxiter"][0]),
                               tol=float(conf["tol"][0]),
                               vary_sky=bool(conf["vary_sky"][0]),
                               fill_zero_weight=fzw)
        assert int(status.cpu()[0]) == 0, name
        out = out.cpu().numpy()[0]
        _check_em(name, int(out[0]), out[1], out[2], gmb.to_numpy()[0],
                  convb.to_numpy()[0], g, label="batch")


def test_em_errors():
    L = _lib.lib()
    conf = np.zeros(1, dtype=_lib.EM_CONF_DTYPE)
    conf["maxiter"], conf["miniter"], conf["tol"] = 10, 2, 1e-5
    pix = np.zeros(9, dtype=_lib.PIXEL_DTYPE)
    pix["area"] = pix["ierr"] = pix["val"] = 1.0
    pix["v"] = np.repeat(np.arange(3.0) + 100, 3)
    pix["u"] = np.tile(np.arange(3.0) + 100, 3)

    def setup():
        gm = np.zeros(1, dtype=_lib.GAUSS2D_DTYPE)
        L.ngmix_fill_model(_lib.ptr(gm), 1, 1,
                           _lib.ptr(np.array([0, 0, 0, 0, 1.0, 1.0])), 6)
        psf = np.zeros(1, dtype=_lib.GAUSS2D_DTYPE)
        L.ngmix_fill_model(_lib.ptr(psf), 1, 1,
                           _lib.ptr(np.array([0, 0, 0, 0, 0.0, 1.0])), 6)
        conv = np.zeros(1, dtype=_lib.GAUSS2D_DTYPE)
        L.ngmix_convolve_fill(_lib.ptr(conv), _lib.ptr(gm), 1, _lib.ptr(psf), 1)
        return gm, psf, conv

    def run():
        gm, psf, conv = setup()
        sums = np.zeros((1, 14))
        numiter = ctypes.c_int32()
        frac, sky = ctypes.c_double(), ctypes.c_double()
        st = L.ngmix_em_run(0, _lib.ptr(conf), _lib.ptr(pix), 9, _lib.ptr(sums),
                            _lib.ptr(gm), 1, _lib.ptr(psf), 1, _lib.ptr(conv), 0,
                            ctypes.byref(numiter), ctypes.byref(frac),
                            ctypes.byref(sky))
        return st, numiter.value

    assert run()[0] == _lib.ERR_GTOT_ZERO        # sky 0, all values 0
    conf["sky"] = 1.0
    assert run()[0] == _lib.ERR_ZERO_DIV         # pnew == 0 -> 1/p
    conf["maxiter"] = 0
    st, numiter = run()
    assert st == 0 and numiter == 0              # EM_MAXITER by numiter>=maxiter


# ------------------------------------------------------------- deriv_images
def test_deriv_images_seam_and_batch(golden):
    from ngmix_amd.batch import StampBatch
    g = golden("derivs")
    L = _lib.lib()
    nrow, ncol = [int(x) for x in g["dims"]]
    v = np.ascontiguousarray(g["v"])
    u = np.ascontiguousarray(g["u"])
    area = np.ascontiguousarray(g["area"])
    for name in [str(n) for n in g["names"]]:
        gpars = np.ascontiguousarray(g[name + "_gpars"])
        dcov = np.ascontiguousarray(g[name + "_dcov"])
        out = np.zeros((6, v.size))
        assert L.ngmix_deriv_images(_lib.ptr(gpars), _lib.ptr(dcov), gpars.shape[0],
                                    _lib.ptr(v), _lib.ptr(u), _lib.ptr(area),
                                    v.size, _lib.ptr(out)) == 0
        np.testing.assert_array_equal(out, g[name + "_out"], err_msg=name)
        sb = StampBatch.from_images(np.zeros((1, nrow, ncol)), None, g["jac"])
        bout = sb.deriv_images(gpars, dcov, gpars.shape[0])
        np.testing.assert_array_equal(bout.cpu().numpy().reshape(6, -1),
                                      g[name + "_out"], err_msg=name + " batch")


def test_deriv_images_masked_layout(golden):
    """masked stamps: the batch form writes only the reference's pixel list"""
    from ngmix_amd.batch import StampBatch
    g = golden("derivs")
    nrow, ncol = [int(x) for x in g["dims"]]
    name = "exp_psf1"
    gpars, dcov = g[name + "_gpars"], g[name + "_dcov"]
    w = np.ones((2, nrow, ncol))
    w[1, 3, 4] = 0.0
    w[1, 10, :5] = 0.0
    sb = StampBatch.from_images(np.zeros((2, nrow, ncol)), w,
                                np.tile(g["jac"].view("f8").reshape(1, 8), (2, 1)))
    gp2 = np.concatenate([gpars, gpars])
    dc2 = np.concatenate([dcov, dcov])
    out = sb.deriv_images(gp2, dc2, gpars.shape[0]).cpu().numpy()
    n0 = nrow * ncol
    full = g[name + "_out"]
    np.testing.assert_array_equal(out[:6 * n0].reshape(6, n0), full)
    keep = (w[1] > 0).ravel()
    nk = int(keep.sum())
    np.testing.assert_array_equal(out[6 * n0:].reshape(6, nk), full[:, keep])


@pytest.mark.parametrize("dim,kind", [(40, 0), (48, 0), (48, 2), (64, 1), (33, 3)])
def test_em_multi_wave_vs_reference_order_kernel(dim, kind):
    """stamps above 32x32 run the fused EM body on 2 or 4 waves; the 256-thread
    reference-order kernel (NGMIX_EM_NT=256) is the check: same iteration
    count, mixtures to 1e-9"""
    import os
    from ngmix_amd.batch import StampBatch, GMixBatch
    rng = np.random.RandomState(100 + dim + kind)
    n, scale = 6, 0.263
    pars = np.zeros((n, 6))
    pars[:, 0:2] = rng.uniform(-0.5, 0.5, size=(n, 2)) * scale
    pars[:, 2:4] = rng.normal(scale=0.05, size=(n, 2))
    pars[:, 4] = rng.uniform(0.5, 1.2, size=n)
    pars[:, 5] = rng.uniform(50, 200, size=n)
    cen = (dim - 1) / 2.0
    jac = np.array([cen, cen, scale, 0.0, 0.0, scale, scale ** 2, scale])
    gm_true, _ = GMixBatch.from_pars(pars, "gauss")
    geom = StampBatch.from_images(np.zeros((n, dim, dim)), None, jac)
    truth, _ = geom.render(gm_true)
    sky = 0.05
    images = truth.cpu().numpy().reshape(n, dim, dim) + sky + \
        0.002 * rng.normal(size=(n, dim, dim))
    weights = np.full((n, dim, dim), 1.0 / 0.002 ** 2)
    weights[1, 5, 7] = 0.0
    sb = StampBatch.from_images(images, weights, jac)
    guess = pars.copy()
    guess[:, 4] = (pars[:, 4] - 0.27) * rng.uniform(0.9, 1.1, size=n)
    guess[:, 5] = pars[:, 5] * scale ** 2 * rng.uniform(0.9, 1.1, size=n)
    psfpars = np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (n, 1))
    results = []
    for env in (None, "256"):
        if env is None:
            os.environ.pop("NGMIX_EM_NT", None)
        else:
            os.environ["NGMIX_EM_NT"] = env
        try:
            gm0, _ = GMixBatch.from_pars(guess, "gauss")
            psf, _ = GMixBatch.from_pars(psfpars, "gauss")
            out, status, conv = sb.em(gm0, psf, sky=sky, kind=kind, miniter=20,
                                      maxiter=200, tol=1e-6, fill_zero_weight=True)
            assert int(status.abs().sum()) == 0
            results.append((out.cpu().numpy(), gm0.to_numpy(), conv.to_numpy()))
        finally:
            os.environ.pop("NGMIX_EM_NT", None)
    (o1, g1, c1), (o2, g2, c2) = results
    np.testing.assert_array_equal(o1[:, 0], o2[:, 0])
    np.testing.assert_allclose(o1[:, 2], o2[:, 2], rtol=1e-9)
    for f in ("p", "row", "col", "irr", "irc", "icc"):
        np.testing.assert_allclose(g1[f], g2[f], rtol=1e-9, atol=1e-11, err_msg=f)
        np.testing.assert_allclose(c1[f], c2[f], rtol=1e-9, atol=1e-11, err_msg=f)


@pytest.mark.parametrize("shape", [(25, 25), (32, 32), (40, 44), (45, 47), (48, 48), (56, 60),
                                   (64, 64), (72, 70)])
def test_admom_kernel_variants_vs_oracle(shape):
    """the kernel is chosen by the batch's largest stamp: one batch per shape
    class runs each variant on its own -- one wave with 16 or 36 register
    slots per lane (<= 32x32, 48x48), two waves with 16 or 32 (<= 2048 px,
    64x64), four waves, the streaming kernel -- against the oracle: exact
    numiter and flags, the record to 1e-10"""
    import ngmix_amd as ngmix
    from ngmix_amd.batch import StampBatch, GMixBatch, records_to_numpy
    from oracle import oracle as ora
    nrow, ncol = shape
    rng = np.random.RandomState(nrow * 100 + ncol)
    scale = 0.263
    nst = 5
    obs, jrecs, Ts = [], [], []
    for k in range(nst):
        jac = ngmix.DiagonalJacobian(row=(nrow - 1) / 2.0 + rng.uniform(-0.4, 0.4),
                                     col=(ncol - 1) / 2.0 + rng.uniform(-0.4, 0.4), scale=scale)
        T = 0.3 + 0.012 * min(nrow, ncol) * rng.uniform(0.7, 1.2)
        gm = ngmix.GMixModel([rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05),
                              rng.uniform(-0.15, 0.15), rng.uniform(-0.15, 0.15), T, 40.0],
                             "gauss")
        im = gm.make_image((nrow, ncol), jacobian=jac)
        im += 0.003 * rng.normal(size=im.shape)
        wt = np.full(im.shape, 1.0 / 0.003 ** 2)
        if k % 2:
            wt[nrow // 3, ncol // 2] = 0.0      # a masked pixel in every other stamp
        obs.append(ngmix.Observation(im, weight=wt, jacobian=jac))
        jrecs.append(jac.get_data().view(np.float64).reshape(8))
        Ts.append(T)
    sb = StampBatch.from_observations(obs)
    guess = np.zeros((nst, 6))
    guess[:, 4] = np.array(Ts) * rng.uniform(0.85, 1.2, size=nst)
    guess[:, 5] = 1.0
    wt, _ = GMixBatch.from_pars(guess, "gauss")
    wt_in = wt.to_numpy()
    res, status = sb.admom(wt)
    assert int(status.abs().sum()) == 0
    res = records_to_numpy(res, _lib.ADMOM_RESULT_DTYPE)
    wt_out = wt.to_numpy()
    conf = np.zeros(1, dtype=ora.ADMOM_CONF_DTYPE)
    conf["maxiter"], conf["shiftmax"], conf["etol"], conf["Ttol"] = 200, 5.0, 1e-5, 1e-3
    for i, o in enumerate(obs):
        j = np.zeros(1, dtype=ora.JACOBIAN_DTYPE)
        j[0] = tuple(jrecs[i])
        pix = ora.make_pixels(o.image, o.weight, j, True)
        w = conv_rec(wt_in[i], ora.GAUSS2D_DTYPE)
        r = np.zeros(1, dtype=ora.ADMOM_RESULT_DTYPE)
        assert ora.admom(conf, w, pix, r) == 0
        assert r["flags"][0] == 0 and 2 <= r["numiter"][0] < 60
        _check_admom("shape %s stamp %d" % (shape, i), res[i:i + 1], wt_out[i], r, w)


@pytest.mark.parametrize("shape", [(25, 25), (32, 32), (40, 44), (45, 47), (48, 48), (56, 60)])
@pytest.mark.parametrize("ngauss", [1, 2])
def test_em_kernel_variants_vs_oracle(shape, ngauss):
    """one batch per shape class, so that each EM kernel variant runs on its own
    (one wave / two waves with 16 or 18 register slots / four waves): em_run with
    one and two gaussians against the oracle, exact numiter, mixtures to 1e-9"""
    import ngmix_amd as ngmix
    from ngmix_amd.batch import StampBatch, GMixBatch
    from oracle import oracle as ora
    nrow, ncol = shape
    rng = np.random.RandomState(nrow * 10 + ncol + ngauss)
    scale, sky, nst = 0.263, 0.01, 4
    obs, jrecs, Ts = [], [], []
    for k in range(nst):
        jac = ngmix.DiagonalJacobian(row=(nrow - 1) / 2.0 + rng.uniform(-0.4, 0.4),
                                     col=(ncol - 1) / 2.0 + rng.uniform(-0.4, 0.4), scale=scale)
        T = 0.3 + 0.01 * min(nrow, ncol) * rng.uniform(0.8, 1.2)
        gm = ngmix.GMixModel([rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05),
                              rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1), T, 30.0], "exp")
        im = gm.make_image((nrow, ncol), jacobian=jac) + sky
        im += 0.001 * rng.normal(size=im.shape)
        wt = np.full(im.shape, 1.0 / 0.001 ** 2)
        if k % 2:
            wt[nrow // 4, ncol // 3] = 0.0
        obs.append(ngmix.Observation(im, weight=wt, jacobian=jac))
        jrecs.append(jac.get_data().view(np.float64).reshape(8))
        Ts.append(T)
    sb = StampBatch.from_observations(obs)
    full = np.zeros((nst, ngauss, 6))
    for i in range(ngauss):
        full[:, i, 0] = 30.0 * scale ** 2 / ngauss * rng.uniform(0.9, 1.1, size=nst)
        full[:, i, 1:3] = rng.uniform(-0.03, 0.03, size=(nst, 2))
        full[:, i, 3] = 0.5 * np.array(Ts) * (0.6 + 0.9 * i)
        full[:, i, 5] = 0.5 * np.array(Ts) * (0.6 + 0.9 * i)
    gm0, _ = GMixBatch.from_pars(full.reshape(nst, -1), "full", ngauss=ngauss)
    delta = np.zeros((nst, 6))
    delta[:, 5] = 1.0
    psf, _ = GMixBatch.from_pars(delta, "gauss")
    gm_in = gm0.to_numpy().reshape(nst, ngauss)
    psf_in = psf.to_numpy()
    out, status, conv = sb.em(gm0, psf, sky=sky, miniter=20, maxiter=60, tol=1e-6)
    assert int(status.abs().sum()) == 0
    out = out.cpu().numpy()
    gm_out = gm0.to_numpy().reshape(nst, ngauss)
    econf = np.zeros(1, dtype=ora.EM_CONF_DTYPE)
    econf["tol"], econf["maxiter"], econf["miniter"], econf["sky"] = 1e-6, 60, 20, sky
    for i, o in enumerate(obs):
        j = np.zeros(1, dtype=ora.JACOBIAN_DTYPE)
        j[0] = tuple(jrecs[i])
        pix = ora.make_pixels(o.image, o.weight, j, True)
        g = np.zeros(ngauss, dtype=ora.GAUSS2D_DTYPE)
        for k in range(ngauss):
            g[k] = conv_rec(gm_in[i, k], ora.GAUSS2D_DTYPE)[0]
        p = conv_rec(psf_in[i], ora.GAUSS2D_DTYPE)
        c = np.zeros(ngauss, dtype=ora.GAUSS2D_DTYPE)
        ora.gmix_convolve_fill(c, g, p)
        sums = np.zeros((ngauss, ora.EM_SUMS_NDOUBLE[0]))
        st, numiter, frac, _ = ora.em_run(0, econf, pix, sums, g, p, c)
        assert st == 0
        assert int(out[i, 0]) == numiter, (shape, i)
        for f in ("p", "row", "col", "irr", "irc", "icc"):
            np.testing.assert_allclose(gm_out[i][f], g[f], rtol=1e-9, atol=1e-12,
                                       err_msg="%s stamp %d %s" % (shape, i, f))


@pytest.mark.parametrize("shape", [(25, 25), (32, 32), (45, 47), (48, 48), (56, 60), (64, 64),
                                   (70, 66)])
@pytest.mark.parametrize("ngauss,npsf,kind", [(3, 1, 0), (4, 1, 0), (5, 1, 0), (6, 1, 0),
                                              (4, 3, 0), (6, 3, 0), (4, 1, 1), (5, 1, 2),
                                              (6, 1, 3), (1, 3, 0), (2, 3, 1), (7, 1, 0),
                                              (8, 1, 0), (8, 3, 0), (7, 1, 2), (8, 1, 3),
                                              (9, 1, 0)])
def test_em_many_gaussians_vs_oracle(shape, ngauss, npsf, kind):
    """em_run is general in the gaussian counts (em_nb.py:160-246, 284-354): the
    fused kernels for up to six object gaussians (em_wave.hip / em_wave_hi.hip:
    one wave up to 32x32, two up to 48x48, four up to 64x64) and for seven and
    eight on one or two waves (em_wave_8.hip), with one- and three-gaussian psfs
    and every run kind, against the oracle -- exact numiter, mixtures to 1e-9;
    the census says which kernel served each case (the generic one only for
    nine gaussians, seven / eight beyond 48x48, and stamps beyond 64x64)"""
    import ngmix_amd as ngmix
    from ngmix_amd.batch import StampBatch, GMixBatch
    from oracle import oracle as ora
    nrow, ncol = shape
    rng = np.random.RandomState(nrow * 7 + ncol + 31 * ngauss + npsf + 5 * kind)
    scale, sky, nst = 0.263, 0.01, 3
    obs, jrecs, Ts = [], [], []
    for k in range(nst):
        jac = ngmix.DiagonalJacobian(row=(nrow - 1) / 2.0 + rng.uniform(-0.4, 0.4),
                                     col=(ncol - 1) / 2.0 + rng.uniform(-0.4, 0.4), scale=scale)
        T = 0.3 + 0.01 * min(nrow, ncol) * rng.uniform(0.8, 1.2)
        gm = ngmix.GMixModel([rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05),
                              rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1), T, 30.0], "dev")
        im = gm.make_image((nrow, ncol), jacobian=jac) + sky
        im += 0.001 * rng.normal(size=im.shape)
        wt = np.full(im.shape, 1.0 / 0.001 ** 2)
        if k % 2:
            wt[nrow // 4, ncol // 3] = 0.0
        obs.append(ngmix.Observation(im, weight=wt, jacobian=jac))
        jrecs.append(jac.get_data().view(np.float64).reshape(8))
        Ts.append(T)
    sb = StampBatch.from_observations(obs)
    full = np.zeros((nst, ngauss, 6))
    for i in range(ngauss):
        full[:, i, 0] = 30.0 * scale ** 2 / ngauss * rng.uniform(0.9, 1.1, size=nst)
        full[:, i, 1:3] = rng.uniform(-0.03, 0.03, size=(nst, 2))
        full[:, i, 3] = 0.5 * np.array(Ts) * (0.3 + 0.5 * i)
        full[:, i, 5] = 0.5 * np.array(Ts) * (0.3 + 0.5 * i)
    gm0, _ = GMixBatch.from_pars(full.reshape(nst, -1), "full", ngauss=ngauss)
    if npsf == 1:
        ppars = np.tile([0.0, 0.0, 0.0, 0.0, 0.05, 1.0], (nst, 1))
        psf, _ = GMixBatch.from_pars(ppars, "gauss")
    else:
        ppars = np.tile([0.0, 0.0, 0.01, -0.02, 0.06, 1.0], (nst, 1))
        psf, _ = GMixBatch.from_pars(ppars, "turb")
    gm_in = gm0.to_numpy().reshape(nst, ngauss)
    psf_in = psf.to_numpy().reshape(nst, npsf)
    miniter = 20 if kind != 3 else 5
    _lib.launch_census(reset=True)
    out, status, conv = sb.em(gm0, psf, sky=sky, kind=kind, miniter=miniter, maxiter=40,
                              tol=1e-6)
    seen = _lib.launch_census(reset=True)
    # which kernel served it: the fused one- / two-wave kernels up to 2048 pixels
    # (2304 in the full run, 4096 for <= 3 object gaussians) with the psf count
    # compile-time for 1 and (one wave, <= 3 gaussians) 3; the generic beyond
    npix = nrow * ncol
    fused = ngauss <= 8 and npix <= (4096 if ngauss <= 6 else (2304 if kind == 0 else 2048))
    assert len(seen) == 1, seen
    name = list(seen)[0]
    if fused:
        nt = 64 if npix <= 1024 else (128 if npix <= 2304 and (npix <= 2048 or kind == 0)
                                      else 256)
        cpsf = 1 if npsf == 1 else (3 if npsf == 3 and ngauss <= 3 and nt == 64 else 0)
        assert name.startswith("em_wave_kernel<%d, " % nt), seen
        assert name.endswith(", %d, %d, %d>" % (kind, ngauss, cpsf)), seen
    else:
        assert name.startswith("em_grid_kernel<"), seen
    assert int(status.abs().sum()) == 0
    out = out.cpu().numpy()
    gm_out = gm0.to_numpy().reshape(nst, ngauss)
    econf = np.zeros(1, dtype=ora.EM_CONF_DTYPE)
    econf["tol"], econf["maxiter"], econf["miniter"], econf["sky"] = 1e-6, 40, miniter, sky
    for i, o in enumerate(obs):
        j = np.zeros(1, dtype=ora.JACOBIAN_DTYPE)
        j[0] = tuple(jrecs[i])
        pix = ora.make_pixels(o.image, o.weight, j, True)
        g = np.zeros(ngauss, dtype=ora.GAUSS2D_DTYPE)
        for k in range(ngauss):
            g[k] = conv_rec(gm_in[i, k], ora.GAUSS2D_DTYPE)[0]
        p = np.zeros(npsf, dtype=ora.GAUSS2D_DTYPE)
        for k in range(npsf):
            p[k] = conv_rec(psf_in[i, k], ora.GAUSS2D_DTYPE)[0]
        c = np.zeros(ngauss * npsf, dtype=ora.GAUSS2D_DTYPE)
        ora.gmix_convolve_fill(c, g, p)
        sums = np.zeros((ngauss, ora.EM_SUMS_NDOUBLE[kind]))
        st, numiter, frac, _ = ora.em_run(kind, econf, pix, sums, g, p, c)
        assert st == 0
        assert int(out[i, 0]) == numiter, (shape, i)
        for f in ("p", "row", "col", "irr", "irc", "icc"):
            np.testing.assert_allclose(gm_out[i][f], g[f], rtol=1e-9, atol=1e-12,
                                       err_msg="%s stamp %d %s" % (shape, i, f))


@pytest.mark.parametrize("shape", [(32, 32), (48, 48), (64, 64), (90, 80)])
def test_admom_no_cov_changes_nothing_but_the_covariance(shape):
    """conf.no_cov (the batch extension in the reference record's padding): the
    iteration, the weight, the sums and the flags are the same bytes; only
    sums_cov (and what get_result derives from it) stays zero"""
    from ngmix_amd.batch import StampBatch, GMixBatch, records_to_numpy
    import ngmix_amd as ngmix
    nrow, ncol = shape
    rng = np.random.RandomState(nrow + ncol)
    obs = []
    for k in range(4):
        jac = ngmix.DiagonalJacobian(row=(nrow - 1) / 2.0 + rng.uniform(-0.3, 0.3),
                                     col=(ncol - 1) / 2.0 + rng.uniform(-0.3, 0.3), scale=0.263)
        T = 0.3 + 0.01 * min(nrow, ncol)
        gm = ngmix.GMixModel([0.0, 0.0, rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1), T, 50.0],
                             "gauss")
        im = gm.make_image((nrow, ncol), jacobian=jac) + 0.002 * rng.normal(size=(nrow, ncol))
        obs.append(ngmix.Observation(im, weight=np.full(im.shape, 2.5e5), jacobian=jac))
    sb = StampBatch.from_observations(obs)
    guess = np.zeros((4, 6))
    guess[:, 4] = 0.3 + 0.011 * min(nrow, ncol)
    guess[:, 5] = 1.0
    out = {}
    for no_cov in (False, True):
        wt, _ = GMixBatch.from_pars(guess, "gauss")
        res, status = sb.admom(wt, no_cov=no_cov)
        assert int(status.abs().sum()) == 0
        out[no_cov] = (records_to_numpy(res, _lib.ADMOM_RESULT_DTYPE), wt.to_numpy())
    (full, wfull), (lean, wlean) = out[False], out[True]
    assert np.all(full["flags"] == 0) and np.abs(full["sums_cov"]).max() > 0
    for name in ("flags", "numiter", "npix", "wsum", "sums", "pars", "rho4", "F"):
        np.testing.assert_array_equal(lean[name], full[name], name)
    assert np.all(lean["sums_cov"] == 0.0)
    for f in wfull.dtype.names:
        np.testing.assert_array_equal(wlean[f], wfull[f], f)


def test_admom_and_em_ragged_shapes_vs_oracle():
    """one batch holding tiny, odd-shaped and > 4096-pixel stamps (every
    kernel variant: one / two / four waves per stamp and the generic
    256-thread kernels): admom records and EM mixtures against the oracle"""
    import ngmix_amd as ngmix
    from ngmix_amd.batch import StampBatch, GMixBatch, records_to_numpy
    from oracle import oracle as ora
    rng = np.random.RandomState(77)
    scale = 0.263
    shapes = [(9, 9), (17, 23), (32, 32), (40, 44), (48, 48), (64, 64), (70, 66), (101, 97)]
    obs, gms, jrecs = [], [], []
    for nrow, ncol in shapes:
        jac = ngmix.DiagonalJacobian(row=(nrow - 1) / 2.0 + 0.3, col=(ncol - 1) / 2.0 - 0.2,
                                     scale=scale)
        T = 0.25 + 0.01 * min(nrow, ncol)
        gm = ngmix.GMixModel([0.02, -0.03, 0.08, -0.05, T, 30.0], "gauss")
        im = gm.make_image((nrow, ncol), jacobian=jac)
        im += 0.002 * rng.normal(size=im.shape)
        wt = np.full(im.shape, 1.0 / 0.002 ** 2)
        wt[nrow // 3, ncol // 2] = 0.0
        obs.append(ngmix.Observation(im, weight=wt, jacobian=jac))
        gms.append(T)
        jrecs.append(jac.get_data().view(np.float64).reshape(8))
    n = len(obs)
    sb = StampBatch.from_observations(obs)
    guess = np.zeros((n, 6))
    guess[:, 4] = np.array(gms) * 1.1
    guess[:, 5] = 1.0
    wt, _ = GMixBatch.from_pars(guess, "gauss")
    wt_in = wt.to_numpy()
    res, status = sb.admom(wt)
    assert int(status.abs().sum()) == 0
    res = records_to_numpy(res, _lib.ADMOM_RESULT_DTYPE)
    wt_out = wt.to_numpy()
    conf = np.zeros(1, dtype=ora.ADMOM_CONF_DTYPE)
    conf["maxiter"], conf["shiftmax"], conf["etol"], conf["Ttol"] = 200, 5.0, 1e-5, 1e-3
    pixlist = []
    for i, o in enumerate(obs):
        j = np.zeros(1, dtype=ora.JACOBIAN_DTYPE)
        j[0] = tuple(jrecs[i])
        pix = ora.make_pixels(o.image, o.weight, j, True)
        pixlist.append(pix)
        w = conv_rec(wt_in[i], ora.GAUSS2D_DTYPE)
        r = np.zeros(1, dtype=ora.ADMOM_RESULT_DTYPE)
        assert ora.admom(conf, w, pix, r) == 0
        _check_admom("shape %s" % (shapes[i],), res[i:i + 1], wt_out[i], r, w)
    # EM: one gaussian, unit psf, image + sky
    sky = 0.01
    skyobs = [ngmix.Observation(o.image + sky, weight=o.weight, jacobian=o.jacobian)
              for o in obs]
    sbe = StampBatch.from_observations(skyobs)
    eg = np.zeros((n, 6))
    eg[:, 4] = np.array(gms) * 0.9
    eg[:, 5] = 30.0 * scale ** 2
    gm0, _ = GMixBatch.from_pars(eg, "gauss")
    delta = np.zeros((n, 6))
    delta[:, 5] = 1.0
    psf, _ = GMixBatch.from_pars(delta, "gauss")
    gm_in, psf_in = gm0.to_numpy(), psf.to_numpy()
    out, status, conv = sbe.em(gm0, psf, sky=sky, miniter=20, maxiter=300, tol=1e-6)
    assert int(status.abs().sum()) == 0
    out = out.cpu().numpy()
    gm_out = gm0.to_numpy()
    econf = np.zeros(1, dtype=ora.EM_CONF_DTYPE)
    econf["tol"], econf["maxiter"], econf["miniter"], econf["sky"] = 1e-6, 300, 20, sky
    for i in range(n):
        pix = pixlist[i].copy()
        pix["val"] += sky
        g = conv_rec(gm_in[i], ora.GAUSS2D_DTYPE)
        p = conv_rec(psf_in[i], ora.GAUSS2D_DTYPE)
        c = np.zeros(1, dtype=ora.GAUSS2D_DTYPE)
        ora.gmix_convolve_fill(c, g, p)
        sums = np.zeros((1, ora.EM_SUMS_NDOUBLE[0]))
        st, numiter, frac, _ = ora.em_run(0, econf, pix, sums, g, p, c)
        assert st == 0
        assert int(out[i, 0]) == numiter, shapes[i]
        for f in ("p", "row", "col", "irr", "irc", "icc"):
            np.testing.assert_allclose(gm_out[i][f], g[f], rtol=1e-9, atol=1e-12,
                                       err_msg="%s %s" % (shapes[i], f))


@pytest.mark.parametrize("shape", [(32, 32), (48, 48)])
@pytest.mark.parametrize("bad", [np.nan, np.inf])
def test_admom_nonfinite_pixel_outside_the_weight(shape, bad):
    """a NaN / inf pixel far outside the weight's 5-sigma cut: the reference
    forms weight * val for EVERY pixel (admom_nb.py:111-175), so 0 * NaN = NaN
    poisons its sums whatever the pixel's distance.  The kernel skips slots no
    lane of which is inside the cut -- but never a slot that holds a non-finite
    value: the record (flags, numiter, the NaN pattern) is the oracle's"""
    import ngmix_amd as ngmix
    from ngmix_amd.batch import StampBatch, GMixBatch, records_to_numpy
    from oracle import oracle as ora
    nrow, ncol = shape
    rng = np.random.RandomState(5)
    jac = ngmix.DiagonalJacobian(row=(nrow - 1) / 2.0, col=(ncol - 1) / 2.0, scale=0.263)
    gm = ngmix.GMixModel([0.0, 0.0, 0.05, -0.03, 0.35, 40.0], "gauss")   # small object
    obs, pix_all = [], []
    for corner in ((0, 0), (nrow - 1, ncol - 1), (0, ncol // 2)):
        im = gm.make_image((nrow, ncol), jacobian=jac) + 0.003 * rng.normal(size=shape)
        im[corner] = bad
        obs.append(ngmix.Observation(im, weight=np.full(shape, 1.0 / 0.003 ** 2), jacobian=jac))
    sb = StampBatch.from_observations(obs)
    guess = np.tile([0.0, 0.0, 0.0, 0.0, 0.4, 1.0], (len(obs), 1))
    wt, _ = GMixBatch.from_pars(guess, "gauss")
    wt_in = wt.to_numpy()
    res, status = sb.admom(wt)
    res = records_to_numpy(res, _lib.ADMOM_RESULT_DTYPE)
    conf = np.zeros(1, dtype=ora.ADMOM_CONF_DTYPE)
    conf["maxiter"], conf["shiftmax"], conf["etol"], conf["Ttol"] = 200, 5.0, 1e-5, 1e-3
    j = np.zeros(1, dtype=ora.JACOBIAN_DTYPE)
    j[0] = tuple(jac.get_data().view(np.float64).reshape(8))
    for i, o in enumerate(obs):
        pix = ora.make_pixels(o.image, o.weight, j, True)
        w = conv_rec(wt_in[i], ora.GAUSS2D_DTYPE)
        r = np.zeros(1, dtype=ora.ADMOM_RESULT_DTYPE)
        st = ora.admom(conf, w, pix, r)
        assert int(status[i]) == st
        assert res["flags"][i] == r["flags"][0], (i, res["flags"][i], r["flags"][0])
        assert res["numiter"][i] == r["numiter"][0]
        for f in ("sums", "pars"):
            np.testing.assert_array_equal(np.isfinite(res[f][i]), np.isfinite(r[f][0]), err_msg=f)


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_em_full_wave_form_is_the_general_form_to_the_bit(kind, monkeypatch):
    """the one-gaussian one-wave kernels run a pixel pass without per-slot masks,
    the component and the sky in SGPRs, when every slot of every lane holds a
    listed pixel (32x32 stamps): the same expressions as the general form --
    numiter, frac_diff, sky and the mixtures are the same BYTES
    (NGMIX_EM_NO_FULL sends full waves through the general form; a second
    process, since the knob is read once)"""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
import ngmix_amd as ngmix
from ngmix_amd.batch import StampBatch, GMixBatch
rng = np.random.RandomState(17)
n, dim, scale, sky = 64, 32, 0.263, 0.01
obs = []
for k in range(n):
    jac = ngmix.DiagonalJacobian(row=15.5 + rng.uniform(-0.4, 0.4), col=15.5 + rng.uniform(-0.4, 0.4), scale=scale)
    gm = ngmix.GMixModel([rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), rng.uniform(-0.1, 0.1),
                          rng.uniform(-0.1, 0.1), rng.uniform(0.4, 0.9), 30.0], "gauss")
    im = gm.make_image((dim, dim), jacobian=jac) + sky + 0.001 * rng.normal(size=(dim, dim))
    obs.append(ngmix.Observation(im, weight=np.full((dim, dim), 1.0e6), jacobian=jac))
sb = StampBatch.from_observations(obs)
g = np.zeros((n, 6)); g[:, 4] = rng.uniform(0.3, 0.6, size=n); g[:, 5] = 30.0 * scale ** 2
gm0, _ = GMixBatch.from_pars(g, "gauss")
psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.05, 1.0], (n, 1)), "gauss")
out, status, conv = sb.em(gm0, psf, sky=sky, kind=%d, miniter=20 if %d != 3 else 5, maxiter=60, tol=1e-6)
assert int(status.abs().sum()) == 0
sys.stdout.buffer.write(out.cpu().numpy().tobytes() + gm0.to_numpy().tobytes() + conv.to_numpy().tobytes())
""" % (root, kind, kind)
    env = {k: v for k, v in os.environ.items() if k != "NGMIX_EM_NO_FULL"}
    a = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, timeout=280)
    assert a.returncode == 0, a.stderr[-2000:]
    b = subprocess.run([sys.executable, "-c", code], env=dict(env, NGMIX_EM_NO_FULL="1"),
                       capture_output=True, timeout=280)
    assert b.returncode == 0, b.stderr[-2000:]
    assert len(a.stdout) > 64 * 3 * 8 and a.stdout == b.stdout


def test_c4_shaped_batch_against_the_reference(golden):
    """tests/golden/c4.npz (oracle/gen_golden_c4.py): thirty-two objects of
    config 4's shape -- 32x32, gaussian (x) gaussian psf, off-grid centres,
    bench.py's guesses -- through the REFERENCE's admom (AdmomFitter's default
    configuration) and em_run (40 iterations at config 4's settings; and with
    a poorer guess and the stopping rule deciding: 15-23 iterations), against
    ONE batch of the kernels each: flags / numiter exact, the moment sums and
    their covariance to 1e-10, the mixtures to 1e-9 -- the direct link of
    config 4 to the reference that lm_c3.npz is for config 3"""
    from ngmix_amd.batch import StampBatch, GMixBatch, records_to_numpy
    g = golden("c4")
    n = g["images"].shape[0]
    weights = np.full(g["images"].shape, 1.0 / float(g["noise"]) ** 2)
    sb = StampBatch.from_images(g["images"], weights, g["jac"])
    wt, _ = GMixBatch.from_pars(g["admom_wt_in"].reshape(n, 6), "full", ngauss=1)
    maxiter, shiftmax, etol, Ttol = g["admom_conf"]
    _lib.launch_census(reset=True)
    res, status = sb.admom(wt, maxiter=int(maxiter), shiftmax=float(shiftmax), etol=float(etol),
                           Ttol=float(Ttol))
    seen = _lib.launch_census(reset=True)
    assert any(k.startswith("admom_grid_kernel<64") for k in seen), seen
    assert int(status.abs().sum()) == 0
    res = records_to_numpy(res, _lib.ADMOM_RESULT_DTYPE)
    wt_out = wt.to_numpy()
    np.testing.assert_array_equal(res["flags"], g["admom_flags"])
    np.testing.assert_array_equal(res["numiter"], g["admom_numiter"])
    np.testing.assert_array_equal(res["npix"], g["admom_npix"])
    for i in range(n):
        for f in ("wsum", "sums", "sums_cov", "pars"):
            close(res[f][i], g["admom_" + f][i], err_msg="%s object %d" % (f, i))
        for k, f in enumerate(("row", "col", "irr", "irc", "icc")):
            close(wt_out[f][i], g["admom_wt_out"][i, 0, k + 1], scale=1.0,
                  err_msg="weight %s object %d" % (f, i))
    sky = float(g["sky"])
    sb_em = StampBatch.from_images(g["images"] + sky, weights, g["jac"])
    psf, _ = GMixBatch.from_pars(np.tile(g["psf"].reshape(1, 6), (n, 1)), "full", ngauss=1)
    for tag in ("em", "em2"):
        tol, miniter, maxiter = g[tag + "_conf"]
        gm, _ = GMixBatch.from_pars(g[tag + "_gmix_in"].reshape(n, 6), "full", ngauss=1)
        _lib.launch_census(reset=True)
        out, status, conv = sb_em.em(gm, psf, sky=sky, tol=float(tol), miniter=int(miniter),
                                     maxiter=int(maxiter))
        seen = _lib.launch_census(reset=True)
        assert any(k.startswith("em_wave_kernel<64") for k in seen), seen
        assert int(status.abs().sum()) == 0
        out = out.cpu().numpy()
        np.testing.assert_array_equal(out[:, 0].astype(int), g[tag + "_numiter"])
        np.testing.assert_array_equal(out[:, 2], g[tag + "_sky"])
        gm_out = gm.to_numpy().reshape(n)
        for k, f in enumerate(("p", "row", "col", "irr", "irc", "icc")):
            np.testing.assert_allclose(gm_out[f], g[tag + "_gmix_out"][:, 0, k], rtol=1e-9,
                                       atol=1e-12, err_msg="%s %s" % (tag, f))
