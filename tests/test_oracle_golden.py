"""
Pins the CPU oracle (oracle/ngmix_oracle.c) against golden vectors produced by
running the reference itself (oracle/gen_golden.py).  CPU only.

Tolerances: pixel indexing / coordinates / fdiff / render / per-pixel values
are compared EXACTLY (==) wherever the computation involves only + - * / and
sqrt; libm-dependent quantities (exp, log, atanh, tanh, pow) get a few ulp.
"""
import numpy as np
import pytest

from oracle import oracle as ora

GFIELDS = ("p", "row", "col", "irr", "irc", "icc", "det")
NFIELDS = ("drr", "drc", "dcc", "norm", "pnorm")


def as_gauss(a):
    out = np.zeros(a.size, dtype=ora.GAUSS2D_DTYPE)
    for n in ora.GAUSS2D_DTYPE.names:
        out[n] = a[n]
    return out


def as_pixels(a):
    out = np.zeros(a.size, dtype=ora.PIXEL_DTYPE)
    for n in ora.PIXEL_DTYPE.names:
        out[n] = a[n]
    return out


def assert_gauss_equal(a, b, fields=GFIELDS, rtol=0.0):
    for n in fields:
        if rtol == 0.0:
            np.testing.assert_array_equal(a[n], b[n], err_msg=n)
        else:
            np.testing.assert_allclose(a[n], b[n], rtol=rtol, atol=0, err_msg=n)


# ---------------------------------------------------------------- fastexp
def test_fexp_exact(golden):
    g = golden("fastexp")
    np.testing.assert_array_equal(ora.fexp(g["x"]), g["fexp"])
    w, dw = ora.apod(g["chi2"])
    np.testing.assert_array_equal(w, g["apod"])
    np.testing.assert_array_equal(dw, g["apod_deriv"])
    # the embedded table is numpy.exp(arange(-15,1)) bit for bit
    x = -np.arange(0.0, 13.0)
    np.testing.assert_array_equal(
        ora.fexp(x), g["lookup"][15 - np.arange(13)] * g["coeffs"][0])


def test_fexp_accuracy():
    """the reference's own bound, ngmix/tests/test_fastexp.py:20-27"""
    x = np.linspace(-15.0, 0.0, 100000)
    rel = ora.fexp(x) / np.exp(x) - 1
    assert np.abs(rel).max() < 2.5e-6
    assert abs(rel.mean()) < 1e-7


def test_apod_ends():
    """ngmix/tests/test_fastexp.py:62-97"""
    w, dw = ora.apod(np.array([20.0, 25.0]))
    assert w[0] == 1.0 and w[1] == 0.0
    assert dw[0] == 0.0 and dw[1] == 0.0


# ----------------------------------------------------------------- pixels
@pytest.mark.parametrize("jname", ["unit", "diag", "sheared"])
def test_pixels_exact(golden, jname):
    g = golden("pixels")
    jac = g["jac_" + jname].astype(ora.JACOBIAN_DTYPE)
    coords = ora.make_coords(g["image"].shape, jac)
    for n in ("u", "v", "area"):
        np.testing.assert_array_equal(coords[n], g["coords_" + jname][n])
    for izw in (1, 0):
        pix = ora.make_pixels(g["image"], g["weight"], jac, bool(izw))
        ref = g["pixels_%s_izw%d" % (jname, izw)]
        assert pix.size == ref.size
        for n in ("u", "v", "area", "val", "ierr"):
            np.testing.assert_array_equal(pix[n], ref[n], err_msg=n)
    for (r, c), (v, u), (r2, c2) in zip(g["pts_" + jname], g["vu_" + jname],
                                        g["rowcol_" + jname]):
        assert ora.jacobian_get_vu(jac, r, c) == (v, u)
        st, rr, cc = ora.jacobian_get_rowcol(jac, v, u)
        assert st == 0 and (rr, cc) == (r2, c2)


def test_pixels_not_filled():
    jac = np.zeros(1, dtype=ora.JACOBIAN_DTYPE)
    jac["dvdrow"] = jac["dudcol"] = jac["det"] = jac["scale"] = 1.0
    im = np.ones((3, 3))
    pix = np.zeros(5, dtype=ora.PIXEL_DTYPE)
    assert ora.fill_pixels(pix, im, im, jac, True) == ora.ERR_PIXELS_NOT_FILLED


# ------------------------------------------------------------------ fills
FILL_CASES = ["gauss", "exp", "dev", "turb", "bdf", "bd", "coellip", "full",
              "exp_round", "exp_highg"]


@pytest.mark.parametrize("name", FILL_CASES)
def test_fill_and_convolve(golden, name):
    g = golden("fills")
    model = name.split("_")[0]
    ref = g["gmix_" + name]
    gm = np.zeros(ref.size, dtype=ora.GAUSS2D_DTYPE)
    assert ora.gmix_fill(gm, g["pars_" + name], model) == 0
    # e1,e2 go through atanh/tanh (numpy SIMD vs glibc): few-ulp agreement
    assert_gauss_equal(gm, ref, rtol=1e-14)
    assert np.all(gm["norm_set"] == 0) and np.all(np.isnan(gm["pnorm"]))
    # convolution and norms from the reference's own pre-psf gaussians: exact
    gm = as_gauss(ref)
    for pname in ("psf1", "psf3", "psf_off"):
        psf = as_gauss(g[pname])
        out = np.zeros(gm.size * psf.size, dtype=ora.GAUSS2D_DTYPE)
        assert ora.gmix_convolve_fill(out, gm, psf) == 0
        assert_gauss_equal(out, g["conv_%s_%s" % (name, pname)])
        st = ora.gmix_set_norms(out)
        refn = g["convnorm_%s_%s" % (name, pname)]
        if np.all(refn["norm_set"] == 1):
            assert st == 0
            assert_gauss_equal(out, refn, fields=GFIELDS + NFIELDS)
            assert np.all(out["norm_set"] == 1)


def test_fill_cm_and_shape(golden):
    g = golden("fills")
    st, tf = ora.get_cm_Tfactor(float(g["cm_fracdev"]), float(g["cm_TdByTe"]))
    assert st == 0 and tf == float(g["cm_Tfactor"])
    gm = np.zeros(16, dtype=ora.GAUSS2D_DTYPE)
    assert ora.gmix_fill(gm, g["pars_exp"], "cm", float(g["cm_fracdev"]),
                         float(g["cm_TdByTe"]), tf) == 0
    assert_gauss_equal(gm, g["gmix_cm"], rtol=1e-14)
    for (g1, g2), (e1, e2) in zip(g["g"], g["e"]):
        st, a, b = ora.g1g2_to_e1e2(g1, g2)
        assert st == 0
        np.testing.assert_allclose([a, b], [e1, e2], rtol=1e-14, atol=0)
    assert ora.g1g2_to_e1e2(0.8, 0.7)[0] == ora.ERR_G_RANGE


def test_norm_errors():
    gm = np.zeros(2, dtype=ora.GAUSS2D_DTYPE)
    gm["p"] = 1.0
    gm["irr"] = gm["icc"] = 1.0
    gm["det"] = 1.0
    gm["det"][1] = 1e-201
    assert ora.gmix_set_norms(gm) == ora.ERR_DET_TOO_LOW
    assert gm["norm_set"][0] == 1 and gm["norm_set"][1] == 0
    gm["det"][1] = 1.0
    gm["irr"][1] = gm["icc"][1] = 0.0
    assert ora.gmix_set_norms(gm) == ora.ERR_T_TOO_LOW


# ------------------------------------------------------- render / loglike
def _render_names(golden):
    return [str(n) for n in golden("render_loglike")["names"]]


RENDER_NAMES = ["c1_exp48", "exp48_psf", "gauss32", "bdf64_psf", "masked13x15",
                "masked13x15_keepzero", "tiny20x17"]


def test_render_names_complete(golden):
    assert sorted(_render_names(golden)) == sorted(RENDER_NAMES)


@pytest.mark.parametrize("name", RENDER_NAMES)
def test_render_loglike_fdiff(golden, name):
    g = golden("render_loglike")
    gm_in = as_gauss(g[name + "_gmix_in"])
    jac = g[name + "_jac"].astype(ora.JACOBIAN_DTYPE)
    image, weight = g[name + "_image"], g[name + "_weight"]
    izw = bool(g[name + "_izw"])
    pixels = ora.make_pixels(image, weight, jac, izw)
    ref_pix = g[name + "_pixels"]
    for n in ("u", "v", "area", "val", "ierr"):
        np.testing.assert_array_equal(pixels[n], ref_pix[n])

    gm = gm_in.copy()
    st, (ll, sn, sd, npix) = ora.get_loglike(gm, pixels)
    assert st == 0
    # lazy norms: sqrt and division only -> exact
    assert_gauss_equal(gm, g[name + "_gmix_normed"], fields=GFIELDS + NFIELDS)
    assert np.all(gm["norm_set"] == 1)
    ref = g[name + "_loglike"]
    # same sequential summation order as the reference -> exact
    assert (ll, sn, sd, npix) == (ref[0], ref[1], ref[2], int(ref[3]))

    for start in (0, 13):
        ref_fd = g[name + "_fdiff_start%d" % start]
        fdiff = np.zeros(ref_fd.size) + 7.0
        assert ora.fill_fdiff(gm, pixels, fdiff, start) == 0
        np.testing.assert_array_equal(fdiff, ref_fd)

    st, s2n = ora.get_model_s2n_sum(gm, pixels)
    assert st == 0 and s2n == float(g[name + "_s2n_sum"])

    coords = ora.make_coords(image.shape, jac)
    im = np.zeros(image.size)
    assert ora.render(gm, coords, im, 1) == 0
    np.testing.assert_array_equal(im.reshape(image.shape), g[name + "_render_fast"])
    im = np.zeros(image.size)
    assert ora.render(gm, coords, im, 0) == 0
    # true exp: glibc vs numpy's exp
    np.testing.assert_allclose(im.reshape(image.shape), g[name + "_render_exact"],
                               rtol=2e-15, atol=1e-300)
    acc = g[name + "_render_base"].copy().ravel()
    assert ora.render(gm, coords, acc, 1) == 0
    np.testing.assert_array_equal(acc.reshape(image.shape), g[name + "_render_accum"])


# SURVEY.md 8(d), "C1 single stamp": the numbers the survey measured from the
# reference (oracle/gen_golden_c1.py asserts them before writing c1.npz)
SURVEY_C1 = (-1158.1127300983387, 2798242.571962917, 2798842.226964671, 2304)


def test_c1_survey_values(golden):
    """SURVEY.md 8(d) C1 with the survey's exact inputs (RandomState(1), the
    true mixture): the oracle returns the survey's numbers to the bit"""
    g = golden("c1")
    assert tuple(g["loglike"][:3]) + (int(g["loglike"][3]),) == SURVEY_C1
    jac = g["jac"].astype(ora.JACOBIAN_DTYPE)
    pixels = ora.make_pixels(g["image"], g["weight"], jac, True)
    gm = np.zeros(6, dtype=ora.GAUSS2D_DTYPE)
    assert ora.gmix_fill(gm, g["pars"], "exp") == 0
    assert_gauss_equal(gm, g["gmix_in"], rtol=1e-14)
    gm = as_gauss(g["gmix_in"])
    st, (ll, sn, sd, npix) = ora.get_loglike(gm, pixels)
    assert st == 0
    assert (ll, sn, sd, npix) == SURVEY_C1
    assert_gauss_equal(gm, g["gmix_normed"], fields=GFIELDS + NFIELDS)
    fdiff = np.zeros(pixels.size)
    assert ora.fill_fdiff(gm, pixels, fdiff, 0) == 0
    np.testing.assert_array_equal(fdiff, g["fdiff"])
    # loglike = -chi2 / 2 over the same residuals
    assert abs(-0.5 * np.sum(fdiff ** 2) / SURVEY_C1[0] - 1.0) < 1e-13
    coords = ora.make_coords(g["image"].shape, jac)
    im = np.zeros(g["image"].size)
    assert ora.render(gm, coords, im, 1) == 0
    np.testing.assert_array_equal(im.reshape(g["image"].shape), g["render_fast"])


# ----------------------------------------------------------- weighted sums
def test_weighted_sums(golden):
    g = golden("wsums")
    for name in [str(n) for n in g["names"]]:
        ref = g[name + "_res"]
        nmom = ref["sums"].shape[-1]
        wt = as_gauss(g[name + "_wt"])
        pixels = as_pixels(g[name + "_pixels"])
        res = np.zeros(1, dtype=ora.moments_result_dtype(nmom))
        st = ora.get_weighted_sums(wt, pixels, res, float(g[name + "_maxrad"]))
        assert st == 0, name
        assert res["npix"][0] == ref["npix"][0], name
        # true exp in the weight: compare to a few ulp, scaled by sum |terms|
        for f in ("wsum", "sums", "sums_cov"):
            a, b = res[f][0], ref[f][0]
            scale = np.abs(b).max() if np.ndim(b) else abs(b)
            np.testing.assert_allclose(a, b, rtol=1e-13, atol=1e-13 * scale,
                                       err_msg="%s %s" % (name, f))
        # accumulate-into semantics: a second call doubles the sums
        st = ora.get_weighted_sums(wt, pixels, res, float(g[name + "_maxrad"]))
        np.testing.assert_allclose(res["sums"][0], 2 * ref["sums"][0],
                                   rtol=1e-13, atol=1e-13 * np.abs(ref["sums"]).max())


def test_higher_order_zero_ierr_divides():
    wt = np.zeros(1, dtype=ora.GAUSS2D_DTYPE)
    wt["p"] = 1
    wt["irr"] = wt["icc"] = 1.0
    wt["det"] = 1.0
    assert ora.gmix_set_norms(wt) == 0
    pix = np.zeros(3, dtype=ora.PIXEL_DTYPE)
    pix["area"] = 1.0
    pix["ierr"] = [1.0, 0.0, 1.0]
    res = np.zeros(1, dtype=ora.moments_result_dtype(17))
    assert ora.get_weighted_sums(wt, pix, res, 100.0) == ora.ERR_ZERO_DIV
    res = np.zeros(1, dtype=ora.moments_result_dtype(6))
    assert ora.get_weighted_sums(wt, pix, res, 100.0) == 0
    assert res["npix"][0] == 2


# ------------------------------------------------------------------ admom
def test_admom(golden):
    g = golden("admom")
    for name in [str(n) for n in g["names"]]:
        conf = np.zeros(1, dtype=ora.ADMOM_CONF_DTYPE)
        for n in conf.dtype.names:
            conf[n] = g[name + "_conf"][n]
        wt = as_gauss(g[name + "_wt_in"])
        pixels = as_pixels(g[name + "_pixels"])
        res = np.zeros(1, dtype=ora.ADMOM_RESULT_DTYPE)
        assert ora.admom(conf, wt, pixels, res) == 0
        ref = g[name + "_res"]
        # same operation order, only + - * / sqrt and fexp: bitwise identical
        for f in ref.dtype.names:
            np.testing.assert_array_equal(res[f], ref[f], err_msg="%s %s" % (name, f))
        assert_gauss_equal(wt, g[name + "_wt_out"], fields=GFIELDS + NFIELDS)


def test_admom_maxiter_zero():
    conf = np.zeros(1, dtype=ora.ADMOM_CONF_DTYPE)
    wt = np.zeros(1, dtype=ora.GAUSS2D_DTYPE)
    pix = np.zeros(4, dtype=ora.PIXEL_DTYPE)
    res = np.zeros(1, dtype=ora.ADMOM_RESULT_DTYPE)
    assert ora.admom(conf, wt, pix, res) == 0
    assert res["numiter"][0] == 0 and res["flags"][0] == 32


# --------------------------------------------------------------------- em
def test_em(golden):
    g = golden("em")
    for name in [str(n) for n in g["names"]]:
        kind = int(g[name + "_kind"])
        conf = np.zeros(1, dtype=ora.EM_CONF_DTYPE)
        for n in conf.dtype.names:
            conf[n] = g[name + "_conf"][n]
        pixels = as_pixels(g[name + "_pixels"])
        gm = as_gauss(g[name + "_gmix_in"])
        psf = as_gauss(g[name + "_psf_in"])
        conv = as_gauss(g[name + "_conv_in"])
        sums = np.zeros((gm.size, ora.EM_SUMS_NDOUBLE[kind]))
        st, numiter, frac, sky = ora.em_run(kind, conf, pixels, sums, gm, psf,
                                            conv, bool(g[name + "_fzw"]))
        assert st == 0, name
        assert numiter == int(g[name + "_numiter"]), name
        # logtau/logdet use libm log (numpy vs glibc): the convergence
        # statistic agrees to rounding; the mixture itself does not depend on it
        np.testing.assert_allclose(frac, float(g[name + "_frac_diff"]),
                                   rtol=1e-6, atol=1e-13, err_msg=name)
        assert sky == float(g[name + "_sky"]), name
        assert_gauss_equal(gm, g[name + "_gmix_out"])
        assert np.all(gm["norm_set"] == 0)
        assert_gauss_equal(conv, g[name + "_conv_out"], fields=GFIELDS + NFIELDS)
        np.testing.assert_array_equal(pixels["val"], g[name + "_pixels_out"]["val"])


def test_em_errors():
    conf = np.zeros(1, dtype=ora.EM_CONF_DTYPE)
    conf["maxiter"] = 10
    conf["miniter"] = 2
    conf["tol"] = 1e-5
    pix = np.zeros(9, dtype=ora.PIXEL_DTYPE)
    pix["area"] = 1.0
    pix["ierr"] = 1.0
    pix["v"] = np.repeat(np.arange(3.0) + 100, 3)   # far from the gaussian
    pix["u"] = np.tile(np.arange(3.0) + 100, 3)
    pix["val"] = 1.0
    gm = np.zeros(1, dtype=ora.GAUSS2D_DTYPE)
    ora.gmix_fill(gm, [0, 0, 0, 0, 1.0, 1.0], "gauss")
    psf = np.zeros(1, dtype=ora.GAUSS2D_DTYPE)
    ora.gmix_fill(psf, [0, 0, 0, 0, 0.0, 1.0], "gauss")
    conv = np.zeros(1, dtype=ora.GAUSS2D_DTYPE)
    ora.gmix_convolve_fill(conv, gm, psf)
    sums = np.zeros((1, 14))
    # sky=0 and every gaussian value 0 -> gtot == 0
    st = ora.em_run(0, conf, pix, sums, gm, psf, conv)[0]
    assert st == ora.ERR_GTOT_ZERO
    # with a sky, pnew == 0 -> 1/p raises ZeroDivisionError under numba
    conf["sky"] = 1.0
    ora.gmix_convolve_fill(conv, gm, psf)
    st = ora.em_run(0, conf, pix, sums, gm, psf, conv)[0]
    assert st == ora.ERR_ZERO_DIV
    # maxiter = 0: numiter defined as 0 (EM_MAXITER by numiter >= maxiter)
    conf["maxiter"] = 0
    ora.gmix_fill(gm, [0, 0, 0, 0, 1.0, 1.0], "gauss")
    ora.gmix_convolve_fill(conv, gm, psf)
    st, numiter, _, _ = ora.em_run(0, conf, pix, sums, gm, psf, conv)
    assert st == 0 and numiter == 0


# ----------------------------------------------------------------- derivs
def test_deriv_images(golden):
    g = golden("derivs")
    for name in [str(n) for n in g["names"]]:
        out = np.zeros((6, g["v"].size))
        ora.deriv_images(g[name + "_gpars"], g[name + "_dcov"], g["v"], g["u"],
                         g["area"], out)
        np.testing.assert_array_equal(out, g[name + "_out"], err_msg=name)


def test_cpu_lm_leg_is_the_reference_fit(golden):
    """bench.py's config-3 cpu_baseline leg (bench.c3_lmder_fit: scipy's MINPACK
    lmder around the C port of fill_fdiff / deriv_images at DEFAULT_LM_PARS)
    against the REFERENCE's own Fitter on the forty objects of
    tests/golden/lm_c3.npz: the same nfev and ier fit by fit, the same
    solution -- the CPU figure printed beside the GPU's is the reference's
    algorithm, not a look-alike"""
    import bench
    g = golden("lm_c3")
    psf = np.zeros(1, dtype=ora.GAUSS2D_DTYPE)
    ora.gmix_fill(psf, np.array([0.0, 0.0, 0.0, 0.0, 0.27, 1.0]), "gauss")
    work = {}
    for i in range(g["images"].shape[0]):
        jac = np.zeros(1, dtype=ora.JACOBIAN_DTYPE)
        for n in ora.JACOBIAN_DTYPE.names:
            jac[n] = g["jac"][i][n]
        wt = np.full(g["images"][i].shape, 1.0 / g["sigma"][i] ** 2)
        pix = ora.make_pixels(g["images"][i], wt, jac, True)
        out = bench.c3_lmder_fit(ora, pix, psf, g["guess"][i].copy(), work)
        assert out[2]["nfev"] == int(g["nfev"][i]), i
        assert out[4] == int(g["ier"][i]), i
        np.testing.assert_allclose(out[0], g["pars"][i], rtol=1e-8, atol=1e-10)
        # leastsq's cov_x is the reference's pars_cov0
        np.testing.assert_allclose(out[1], g["pars_cov0"][i], rtol=1e-6)


# ------------------------------------------------- config 4's shape, at scale
def _c4_records(rows):
    g = np.zeros(len(rows), dtype=ora.GAUSS2D_DTYPE)
    for k, f in enumerate(("p", "row", "col", "irr", "irc", "icc")):
        g[f] = np.asarray(rows)[:, k]
    g["det"] = g["irr"] * g["icc"] - g["irc"] ** 2
    return g


def test_c4_shaped_objects_against_the_reference(golden):
    """tests/golden/c4.npz (oracle/gen_golden_c4.py): thirty-two objects of
    config 4's shape through the REFERENCE's admom and em_run; the oracle
    reproduces the adaptive moments bit for bit and EM's numiter and mixtures"""
    g = golden("c4")
    n = g["images"].shape[0]
    weight = np.full(g["images"].shape[1:], 1.0 / float(g["noise"]) ** 2)
    conf = np.zeros(1, dtype=ora.ADMOM_CONF_DTYPE)
    conf["maxiter"], conf["shiftmax"], conf["etol"], conf["Ttol"] = g["admom_conf"]
    psf = _c4_records(g["psf"])
    for i in range(n):
        j = np.zeros(1, dtype=ora.JACOBIAN_DTYPE)
        j[0] = tuple(g["jac"][i])
        pix = ora.make_pixels(g["images"][i], weight, j, True)
        wt = _c4_records(g["admom_wt_in"][i])
        res = np.zeros(1, dtype=ora.ADMOM_RESULT_DTYPE)
        assert ora.admom(conf, wt, pix, res) == 0
        for f in ("flags", "numiter", "npix", "wsum", "sums", "sums_cov", "pars"):
            np.testing.assert_array_equal(res[f][0], g["admom_" + f][i], err_msg="%d %s" % (i, f))
        for k, f in enumerate(("p", "row", "col", "irr", "irc", "icc")):
            assert wt[f][0] == g["admom_wt_out"][i, 0, k], (i, f)
        pix_e = ora.make_pixels(g["images"][i] + float(g["sky"]), weight, j, True)
        for tag in ("em", "em2"):
            econf = np.zeros(1, dtype=ora.EM_CONF_DTYPE)
            econf["tol"], econf["miniter"], econf["maxiter"] = g[tag + "_conf"]
            econf["sky"] = float(g["sky"])
            gm = _c4_records(g[tag + "_gmix_in"][i])
            conv = np.zeros(1, dtype=ora.GAUSS2D_DTYPE)
            ora.gmix_convolve_fill(conv, gm, psf)
            sums = np.zeros((1, ora.EM_SUMS_NDOUBLE[0]))
            st, numiter, frac, sky = ora.em_run(0, econf, pix_e, sums, gm, psf, conv)
            assert st == 0 and numiter == int(g[tag + "_numiter"][i]), (tag, i, numiter)
            assert sky == float(g[tag + "_sky"][i])
            for k, f in enumerate(("p", "row", "col", "irr", "irc", "icc")):
                assert gm[f][0] == g[tag + "_gmix_out"][i, 0, k], (tag, i, f)


def test_c5_shaped_objects_against_the_reference(golden):
    """tests/golden/c5.npz (oracle/gen_golden_c5.py; inputs rebuilt by
    helpers/c5_inputs.py): the REFERENCE's GMix.get_loglike on six objects of
    ten 64x64 epochs, 'bdf' (x) gaussian psf -- config 5's shape -- at two
    parameter sets; the oracle's model fill, convolution and loglike give the
    same four numbers per epoch"""
    from helpers import c5_inputs as c5
    g = golden("c5")
    pars, moved, jac, images = c5.objects()
    np.testing.assert_allclose(images.sum(axis=(2, 3)), g["image_sums"], rtol=1e-13)
    weight = np.full((c5.DIM, c5.DIM), 1.0 / c5.NOISE ** 2)
    psf = np.zeros(1, dtype=ora.GAUSS2D_DTYPE)
    assert ora.gmix_fill(psf, [0.0, 0.0, 0.0, 0.0, c5.TPSF, 1.0], "gauss") == 0
    for tag, pp in (("truth", pars), ("moved", moved)):
        for o in range(c5.NOBJ):
            gm = np.zeros(16, dtype=ora.GAUSS2D_DTYPE)
            assert ora.gmix_fill(gm, pp[o], "bdf") == 0
            conv = np.zeros(16, dtype=ora.GAUSS2D_DTYPE)
            ora.gmix_convolve_fill(conv, gm, psf)
            for e in range(c5.NEPOCH):
                j = np.zeros(1, dtype=ora.JACOBIAN_DTYPE)
                j[0] = tuple(jac[o, e])
                pix = ora.make_pixels(images[o, e], weight, j, True)
                st, res = ora.get_loglike(conv, pix)
                assert st == 0
                np.testing.assert_allclose(res[:3], g[tag + "_per_epoch"][o, e, :3], rtol=1e-12,
                                           err_msg="%s object %d epoch %d" % (tag, o, e))
                assert res[3] == g[tag + "_per_epoch"][o, e, 3]


def test_c2_shaped_stamps_against_the_reference(golden):
    """tests/golden/c2.npz (oracle/gen_golden_c2.py; inputs rebuilt by
    helpers/c2_inputs.py): the REFERENCE's get_loglike / fill_fdiff /
    _fill_image (accumulating, fast exp) on eight stamps of config 2's shape at
    two parameter sets; the oracle's model fill, convolution and pixel loops
    give the same numbers (the model fill goes through libm's tanh / atanh:
    to rounding, not to the bit)"""
    from helpers import c2_inputs as c2
    g = golden("c2")
    pars, moved, jac, images, sigma, base = c2.stamps()
    np.testing.assert_allclose(images.sum(axis=(1, 2)), g["image_sums"], rtol=1e-13)
    psf = np.zeros(1, dtype=ora.GAUSS2D_DTYPE)
    assert ora.gmix_fill(psf, [0.0, 0.0, 0.0, 0.0, c2.TPSF, 1.0], "gauss") == 0
    for tag, pp in (("truth", pars), ("moved", moved)):
        for i in range(c2.N):
            gm = np.zeros(6, dtype=ora.GAUSS2D_DTYPE)
            assert ora.gmix_fill(gm, pp[i], "exp") == 0
            conv = np.zeros(6, dtype=ora.GAUSS2D_DTYPE)
            ora.gmix_convolve_fill(conv, gm, psf)
            j = np.zeros(1, dtype=ora.JACOBIAN_DTYPE)
            j[0] = tuple(jac[i])
            weight = np.full(images[i].shape, 1.0 / sigma[i] ** 2)
            pix = ora.make_pixels(images[i], weight, j, True)
            st, res = ora.get_loglike(conv, pix)
            assert st == 0 and res[3] == g[tag + "_loglike"][i, 3]
            np.testing.assert_allclose(res[:3], g[tag + "_loglike"][i, :3], rtol=1e-11)
            fd = np.zeros(pix.size)
            ora.fill_fdiff(conv, pix, fd, 0)
            ref = g[tag + "_fdiff"][i]
            np.testing.assert_allclose(fd, ref, rtol=0, atol=1e-12 * np.abs(ref).max())
            im = base[i].copy().ravel()
            ora.render(conv, ora.make_coords((c2.DIM, c2.DIM), j), im, 1)
            ref = g[tag + "_rendered"][i].ravel()
            np.testing.assert_allclose(im, ref, rtol=0, atol=1e-13 * np.abs(ref).max())
