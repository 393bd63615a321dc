#!/usr/bin/env python
"""
Generate golden input/output vectors for the ngmix pixel hot path by running
the REFERENCE ITSELF (/root/reference/ngmix) as interpreted Python under the
no-op numba shim in oracle/shim.

Runs only in the build container (the reference never travels); the fixtures
it writes under tests/golden/ are committed.  TEST INFRASTRUCTURE ONLY.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [--only NAME]

Known divergences of the shim from real numba (cases below avoid them):
  * x**2 on numpy scalars goes through libm pow and differs from x*x by 1 ulp
    in ~0.1% of arguments; numba lowers **2 to a multiplication.  The pixel
    scales used here were checked to have pow(s,2) == s*s.
  * float division by zero returns inf/nan instead of raising.
"""
import argparse
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "shim"), "/root/reference"]

import numpy as np  # noqa: E402
import ngmix  # noqa: E402
from ngmix import fastexp_nb  # noqa: E402
from ngmix.gmix import gmix_nb  # noqa: E402
from ngmix.gmix.gmix import get_moments_result_dtype  # noqa: E402
from ngmix.pixels import make_pixels, make_coords  # noqa: E402
from ngmix.admom import admom as admom_mod  # noqa: E402
from ngmix.admom.admom_nb import admom as admom_nb  # noqa: E402
from ngmix.em import em as em_mod  # noqa: E402
from ngmix.em import em_nb  # noqa: E402
from ngmix.fitting.derivs_nb import deriv_images  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
SCALE = 0.263
assert SCALE ** 2 == SCALE * SCALE and 0.25 ** 2 == 0.25 * 0.25


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %s (%.1f kB)" % (path, os.path.getsize(path) / 1e3))


def jac_data(j):
    return j.get_data().copy()


def jacobians():
    return {
        "unit": ngmix.UnitJacobian(row=6.0, col=7.0),
        "diag": ngmix.DiagonalJacobian(row=5.7, col=7.2, scale=SCALE),
        "sheared": ngmix.Jacobian(
            row=6.3, col=6.9,
            dvdrow=-0.15179598030886227, dvdcol=0.13007044200963258,
            dudrow=-0.13014613410130665, dudcol=-0.15185634442578344),
    }


# --------------------------------------------------------------------------
def gen_fastexp():
    rng = np.random.RandomState(1001)
    x = np.concatenate([
        rng.uniform(-15.0, 0.0, size=8000),
        -np.arange(0, 31) * 0.5,               # integers and half integers
        np.nextafter(-np.arange(1, 30) * 0.5, 0.0),
        np.nextafter(-np.arange(1, 30) * 0.5, -20.0),
        np.array([-12.5, -12.499999999, -10.0, -1e-300, -0.0, 0.0]),
    ])
    x = x[(x >= -15.0) & (x <= 0.0)]
    y = np.array([fastexp_nb.fexp(v) for v in x])
    chi2 = np.concatenate([
        rng.uniform(20.0, 25.0, size=2000),
        np.array([20.0, 25.0, np.nextafter(20.0, 30), np.nextafter(25.0, 0)]),
    ])
    w = np.array([fastexp_nb.apod_window(c) for c in chi2])
    dw = np.array([fastexp_nb.apod_window_deriv(c) for c in chi2])
    save("fastexp", x=x, fexp=y, chi2=chi2, apod=w, apod_deriv=dw,
         lookup=fastexp_nb._EXP_LOOKUP.copy(),
         coeffs=fastexp_nb._EXP5_SMOOTH_COEFFS.copy())


# --------------------------------------------------------------------------
def gen_pixels():
    rng = np.random.RandomState(1002)
    out = {}
    nrow, ncol = 13, 15
    image = rng.normal(size=(nrow, ncol))
    weight = rng.uniform(0.5, 2.0, size=(nrow, ncol))
    weight[2, 3] = 0.0
    weight[0, 0] = 0.0
    weight[12, 14] = -1.0
    weight[7, :4] = 0.0
    out["image"] = image
    out["weight"] = weight
    for name, jac in jacobians().items():
        out["jac_" + name] = jac_data(jac)
        out["coords_" + name] = make_coords((nrow, ncol), jac)
        for izw in (True, False):
            pix = make_pixels(image, weight, jac, ignore_zero_weight=izw)
            out["pixels_%s_izw%d" % (name, int(izw))] = pix
        # scalar transforms
        pts = rng.uniform(-3, 16, size=(20, 2))
        vu = np.array([jac.get_vu(r, c) for r, c in pts])
        rc = np.array([jac.get_rowcol(v, u) for v, u in vu])
        out["pts_" + name] = pts
        out["vu_" + name] = vu
        out["rowcol_" + name] = rc
    save("pixels", **out)


# --------------------------------------------------------------------------
def gm_data(gm):
    return gm.get_data().copy()


def gen_fills():
    out = {}
    cases = {
        "gauss": [0.1, -0.2, 0.11, -0.07, 0.8, 12.0],
        "exp": [0.1, -0.05, 0.1, 0.05, 0.6, 100.0],
        "dev": [-0.3, 0.2, -0.2, 0.3, 1.7, 55.0],
        "turb": [0.02, 0.01, 0.03, -0.04, 0.27, 1.0],
        "bdf": [0.1, 0.2, 0.2, -0.1, 1.2, 0.35, 80.0],
        "bd": [0.1, 0.2, -0.15, 0.25, 0.9, 0.2, 0.6, 40.0],
        "coellip": [0.05, -0.02, 0.1, 0.2, 0.3, 0.9, 2.5, 0.2, 0.5, 0.3],
        "full": [1.0, 0.1, 0.2, 0.5, 0.05, 0.6,
                 2.0, -0.1, 0.3, 0.9, -0.1, 0.7],
        # round, and nearly maximal ellipticity (e clamp branch)
        "exp_round": [0.0, 0.0, 0.0, 0.0, 0.5, 1.0],
        "exp_highg": [0.0, 0.0, 0.9999999999, 0.0, 0.5, 1.0],
    }
    psf1 = ngmix.GMixModel([0.01, -0.02, 0.02, 0.01, 0.27, 1.0], "gauss")
    psf3 = ngmix.GMixModel([0.0, 0.0, -0.01, 0.03, 0.27, 0.9], "turb")
    # a psf with offset components, to exercise the centroid handling
    psf_off = ngmix.GMix(pars=[0.6, 0.05, -0.03, 0.14, 0.01, 0.13,
                               0.4, -0.08, 0.04, 0.3, -0.02, 0.33])
    out["psf1"] = gm_data(psf1)
    out["psf3"] = gm_data(psf3)
    out["psf_off"] = gm_data(psf_off)
    for name, pars in cases.items():
        model = name.split("_")[0]
        gm = ngmix.gmix.make_gmix_model(pars, model)
        out["pars_" + name] = np.array(pars)
        out["gmix_" + name] = gm_data(gm)
        for pname, psf in (("psf1", psf1), ("psf3", psf3), ("psf_off", psf_off)):
            gmc = gm.convolve(psf)
            out["conv_%s_%s" % (name, pname)] = gm_data(gmc)
            gmc.set_norms()
            out["convnorm_%s_%s" % (name, pname)] = gm_data(gmc)
    gm = ngmix.gmix.GMixCM(0.3, 1.7, cases["exp"])
    out["cm_fracdev"] = np.array(0.3)
    out["cm_TdByTe"] = np.array(1.7)
    out["cm_Tfactor"] = np.array(gmix_nb.get_cm_Tfactor(0.3, 1.7))
    out["gmix_cm"] = gm_data(gm)
    g = np.array([[0.0, 0.0], [0.1, 0.2], [-0.5, 0.3], [0.7, -0.7], [0.0, 0.95]])
    out["g"] = g
    out["e"] = np.array([gmix_nb.g1g2_to_e1e2(a, b) for a, b in g])
    save("fills", **out)


# --------------------------------------------------------------------------
def make_case_obs(rng, dims, jac, gm_true, noise, mask=False,
                  ignore_zero_weight=True, var_weight=False):
    im = gm_true.make_image(dims, jacobian=jac, fast_exp=True)
    im = im + rng.normal(scale=noise, size=im.shape)
    if var_weight:
        weight = rng.uniform(0.5, 1.5, size=im.shape) / noise ** 2
    else:
        weight = np.full(im.shape, 1.0 / noise ** 2)
    if mask:
        weight[1, 2] = 0.0
        weight[dims[0] // 2, dims[1] // 2 + 1] = 0.0
        weight[dims[0] - 1, :3] = 0.0
        weight[3, 4] = -2.0
    return ngmix.Observation(im, weight=weight, jacobian=jac,
                             ignore_zero_weight=ignore_zero_weight)


def render_cases():
    rng = np.random.RandomState(1003)
    psf = ngmix.GMixModel([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], "gauss")
    cases = {}
    # C1 of SURVEY 8(d): the single-stamp plumbing case
    jac = ngmix.DiagonalJacobian(row=23.5, col=23.5, scale=SCALE)
    gm = ngmix.GMixModel([0.1, -0.05, 0.1, 0.05, 0.6, 100.0], "exp")
    cases["c1_exp48"] = dict(dims=(48, 48), jac=jac, gm=gm, noise=0.01,
                             rng=np.random.RandomState(1))
    # convolved exp, off-centre
    gmc = ngmix.GMixModel([0.07, -0.11, -0.15, 0.2, 0.9, 230.0], "exp").convolve(psf)
    cases["exp48_psf"] = dict(dims=(48, 48), jac=jac, gm=gmc, noise=0.02, rng=rng,
                              var_weight=True)
    jac32 = ngmix.DiagonalJacobian(row=15.2, col=16.1, scale=SCALE)
    gm1 = ngmix.GMixModel([0.02, 0.03, 0.05, -0.1, 0.55, 20.0], "gauss")
    cases["gauss32"] = dict(dims=(32, 32), jac=jac32, gm=gm1, noise=0.005, rng=rng)
    jac64 = ngmix.Jacobian(row=31.3, col=32.4, dvdrow=0.26, dvdcol=0.01,
                           dudrow=-0.02, dudcol=0.265)
    gmb = ngmix.GMixModel([0.1, 0.2, 0.2, -0.1, 1.2, 0.35, 80.0], "bdf").convolve(psf)
    cases["bdf64_psf"] = dict(dims=(64, 64), jac=jac64, gm=gmb, noise=0.01, rng=rng)
    jsh = jacobians()["sheared"]
    gms = ngmix.GMixModel([0.05, -0.1, 0.2, 0.1, 0.4, 5.0], "gauss").convolve(psf)
    cases["masked13x15"] = dict(dims=(13, 15), jac=jsh, gm=gms, noise=0.05,
                                rng=rng, mask=True)
    cases["masked13x15_keepzero"] = dict(dims=(13, 15), jac=jsh, gm=gms,
                                         noise=0.05, rng=rng, mask=True,
                                         ignore_zero_weight=False)
    # tiny gaussian: most of the stamp is beyond chi2=25
    gmt = ngmix.GMixModel([0.3, -0.2, 0.0, 0.0, 0.02, 3.0], "gauss")
    cases["tiny20x17"] = dict(dims=(20, 17), jac=jac32, gm=gmt, noise=0.1, rng=rng)
    return cases


def gen_render_loglike():
    out = {}
    names = []
    for name, c in render_cases().items():
        names.append(name)
        dims, jac, gm_true = c["dims"], c["jac"], c["gm"]
        obs = make_case_obs(c["rng"], dims, jac, gm_true, c["noise"],
                            mask=c.get("mask", False),
                            ignore_zero_weight=c.get("ignore_zero_weight", True),
                            var_weight=c.get("var_weight", False))
        # evaluate at a perturbed model so residuals are not pure noise
        pars = gm_true.get_full_pars()
        pars[0::6] *= 1.03
        pars[1::6] += 0.01
        gm = ngmix.GMix(pars=pars)
        out[name + "_gmix_in"] = gm_data(gm)
        out[name + "_image"] = obs.image.copy()
        out[name + "_weight"] = obs.weight.copy()
        out[name + "_jac"] = jac_data(jac)
        out[name + "_izw"] = np.array(c.get("ignore_zero_weight", True))
        out[name + "_pixels"] = obs.pixels.copy()

        res = gm.get_loglike(obs, more=True)
        out[name + "_gmix_normed"] = gm_data(gm)   # lazy norm side effect
        out[name + "_loglike"] = np.array(
            [res["loglike"], res["s2n_numer"], res["s2n_denom"], res["npix"]])
        for start in (0, 13):
            fdiff = np.zeros(obs.image.size + 13 + 5) + 7.0
            gm.fill_fdiff(obs, fdiff, start=start)
            out[name + "_fdiff_start%d" % start] = fdiff
        out[name + "_s2n_sum"] = np.array(gm.get_model_s2n_sum(obs))
        out[name + "_render_fast"] = gm.make_image(dims, jacobian=jac, fast_exp=True)
        out[name + "_render_exact"] = gm.make_image(dims, jacobian=jac, fast_exp=False)
        # render accumulates into what is there
        base = np.arange(dims[0] * dims[1], dtype="f8").reshape(dims) * 1e-3
        acc = base.copy()
        gm._fill_image(acc, jacobian=jac, fast_exp=True)
        out[name + "_render_base"] = base
        out[name + "_render_accum"] = acc
    out["names"] = np.array(names)
    save("render_loglike", **out)


# --------------------------------------------------------------------------
def gen_wsums():
    rng = np.random.RandomState(1004)
    out = {}
    names = []
    jac = ngmix.DiagonalJacobian(row=15.2, col=16.1, scale=SCALE)
    jsh = jacobians()["sheared"]
    psf = ngmix.GMixModel([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], "gauss")
    gm_true = ngmix.GMixModel([0.02, 0.03, 0.05, -0.1, 0.55, 20.0], "exp").convolve(psf)
    specs = {
        "diag32": dict(dims=(32, 32), jac=jac, mask=False, izw=True, maxrad=None),
        "diag32_maxrad": dict(dims=(32, 32), jac=jac, mask=False, izw=True, maxrad=1.7),
        "sheared_masked": dict(dims=(13, 15), jac=jsh, mask=True, izw=True, maxrad=None),
        "sheared_keepzero": dict(dims=(13, 15), jac=jsh, mask=True, izw=False, maxrad=None),
    }
    for name, s in specs.items():
        obs = make_case_obs(rng, s["dims"], s["jac"], gm_true, 0.01,
                            mask=s["mask"], ignore_zero_weight=s["izw"],
                            var_weight=True)
        wt = ngmix.GMixModel([0.03, -0.02, 0.1, 0.05, 0.7, 1.0], "gauss")
        wt2 = ngmix.GMix(pars=[0.7, 0.03, -0.02, 0.4, 0.02, 0.35,
                               0.3, 0.0, 0.01, 0.9, -0.05, 1.1])
        for wname, w in (("w1", wt), ("w2", wt2)):
            for ho in (False, True):
                if ho and not s["izw"]:
                    # ierr == 0 pixels divide by zero in the higher order sums
                    continue
                key = "%s_%s_n%d" % (name, wname, 17 if ho else 6)
                names.append(key)
                w = w.copy()
                res = w.get_weighted_sums(obs, maxrad=s["maxrad"],
                                          with_higher_order=ho)
                out[key + "_res"] = np.array([res])  # 1-element record array
                out[key + "_wt"] = gm_data(w)
                T = w.get_T()
                out[key + "_maxrad"] = np.array(
                    s["maxrad"] if s["maxrad"] is not None
                    else 100 * np.sqrt(T / 2))
                out[key + "_pixels"] = obs.pixels.copy()
                out[key + "_image"] = obs.image.copy()
                out[key + "_weight"] = obs.weight.copy()
                out[key + "_jac"] = jac_data(s["jac"])
                out[key + "_izw"] = np.array(s["izw"])
    out["names"] = np.array(names)
    save("wsums", **out)


# --------------------------------------------------------------------------
def gen_admom():
    rng = np.random.RandomState(1005)
    out = {}
    names = []
    jac = ngmix.DiagonalJacobian(row=15.3, col=15.9, scale=SCALE)
    jsh = ngmix.Jacobian(row=15.6, col=15.2, dvdrow=0.26, dvdcol=0.012,
                         dudrow=-0.015, dudcol=0.262)
    psf = ngmix.GMixModel([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], "gauss")
    gm_true = ngmix.GMixModel([0.05, -0.04, 0.1, -0.06, 0.5, 50.0], "gauss").convolve(psf)
    gm_exp = ngmix.GMixModel([-0.05, 0.03, -0.2, 0.1, 0.8, 80.0], "exp").convolve(psf)

    def runcase(name, obs, guess_pars, **kw):
        names.append(name)
        conf = dict(maxiter=200, shiftmax=5.0, etol=1.0e-5, Ttol=1.0e-3,
                    cenonly=False)
        conf.update(kw)
        fitter = admom_mod.AdmomFitter(**conf)
        guess = ngmix.GMixModel(guess_pars, "gauss")
        out[name + "_wt_in"] = gm_data(guess)
        out[name + "_conf"] = fitter.conf.copy()
        out[name + "_pixels"] = obs.pixels.copy()
        out[name + "_image"] = obs.image.copy()
        out[name + "_weight"] = obs.weight.copy()
        out[name + "_jac"] = jac_data(obs.jacobian)
        out[name + "_izw"] = np.array(obs.ignore_zero_weight)
        ares = fitter._get_am_result()
        wt = guess._data
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            admom_nb(fitter.conf, wt, obs.pixels, ares)
        out[name + "_res"] = ares.copy()
        out[name + "_wt_out"] = wt.copy()
        print("  admom %-22s flags=%d numiter=%d" % (
            name, ares["flags"][0], ares["numiter"][0]))

    obs = make_case_obs(rng, (32, 32), jac, gm_true, 0.002)
    runcase("gauss32", obs, [0.01, 0.02, 0.0, 0.0, 0.7, 1.0])
    runcase("gauss32_cenonly", obs, [0.01, 0.02, 0.0, 0.0, 0.7, 1.0], cenonly=True)
    runcase("gauss32_maxiter3", obs, [0.01, 0.02, 0.0, 0.0, 0.7, 1.0], maxiter=3)
    runcase("gauss32_censhift", obs, [1.5, -1.0, 0.0, 0.0, 0.7, 1.0], shiftmax=0.05)
    obs2 = make_case_obs(rng, (32, 32), jsh, gm_exp, 0.01, var_weight=True)
    runcase("exp32_sheared", obs2, [-0.02, 0.05, 0.05, -0.05, 1.2, 1.0])
    obs3 = make_case_obs(rng, (32, 32), jac, gm_true, 0.002, mask=True)
    runcase("gauss32_masked", obs3, [0.0, 0.0, 0.1, 0.1, 0.6, 1.0])
    # negative image -> NONPOS_FLUX
    with obs.writeable():
        pass
    obsneg = ngmix.Observation(-obs.image, weight=obs.weight.copy(), jacobian=jac)
    runcase("negative_flux", obsneg, [0.0, 0.0, 0.0, 0.0, 0.7, 1.0])
    # noise only -> usually some failure flag or a long run
    imn = rng.normal(scale=1.0, size=(24, 24))
    jn = ngmix.DiagonalJacobian(row=11.5, col=11.5, scale=SCALE)
    obsn = ngmix.Observation(imn, weight=np.ones_like(imn), jacobian=jn)
    runcase("noise24", obsn, [0.0, 0.0, 0.0, 0.0, 0.5, 1.0], maxiter=30)
    out["names"] = np.array(names)
    save("admom", **out)


# --------------------------------------------------------------------------
def gen_em():
    out = {}
    names = []
    sys.path.insert(0, "/root/reference/ngmix/tests")
    import _sims  # the reference's own test sims (ngmix-only, no galsim)

    def randomize(rng, gmix, pixel_scale):
        # same spirit as ngmix/tests/test_em.py randomize_gmix
        gm = gmix.get_data()
        for g in gm:
            g["p"] *= 1.0 + rng.uniform(-0.05, 0.05)
            g["row"] += rng.uniform(-0.2, 0.2) * pixel_scale
            g["col"] += rng.uniform(-0.2, 0.2) * pixel_scale
            g["irr"] *= 1.0 + rng.uniform(-0.05, 0.05)
            g["irc"] *= 1.0 + rng.uniform(-0.05, 0.05)
            g["icc"] *= 1.0 + rng.uniform(-0.05, 0.05)
            g["det"] = g["irr"] * g["icc"] - g["irc"] ** 2
        gm["norm_set"] = 0

    runners = {0: em_nb.em_run, 1: em_nb.em_run_fixcen,
               2: em_nb.em_run_fixcov, 3: em_nb.em_run_fluxonly}
    fitters = {0: em_mod.EMFitter, 1: em_mod.EMFitterFixCen,
               2: em_mod.EMFitterFixCov, 3: em_mod.EMFitterFluxOnly}

    def runcase(name, obs, guess, kind, sky=None, zero_weight=False, **kw):
        names.append(name)
        fitter = fitters[kind](**kw)
        if sky is None:
            obs_sky, sky = em_mod.prep_obs(obs)
        else:
            obs_sky = obs
        if not obs_sky.has_psf() or not obs_sky.psf.has_gmix():
            gmix_psf = ngmix.GMixModel([0., 0., 0., 0., 0., 1.0], "gauss")
        else:
            gmix_psf = obs_sky.psf.gmix
            gmix_psf.set_flux(1.0)
        conf = fitter._make_conf(obs_sky)
        conf["sky"] = sky
        gm = guess.copy()
        gmc = gm.convolve(gmix_psf)
        sums = fitter._make_sums(len(gm))
        pixels = obs_sky.pixels.copy()
        fzw = bool(np.any(pixels["ierr"] <= 0.0))
        out[name + "_kind"] = np.array(kind)
        out[name + "_conf"] = np.array([conf])
        out[name + "_pixels"] = pixels.copy()
        out[name + "_image"] = obs_sky.image.copy()
        out[name + "_weight"] = obs_sky.weight.copy()
        out[name + "_jac"] = jac_data(obs_sky.jacobian)
        out[name + "_izw"] = np.array(obs_sky.ignore_zero_weight)
        out[name + "_gmix_in"] = gm_data(gm)
        out[name + "_psf_in"] = gm_data(gmix_psf)
        out[name + "_conv_in"] = gm_data(gmc)
        out[name + "_fzw"] = np.array(fzw)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            numiter, fdiff, skyout = runners[kind](
                conf, pixels, sums, gm.get_data(), gmix_psf.get_data(),
                gmc.get_data(), fill_zero_weight=fzw)
        out[name + "_numiter"] = np.array(numiter)
        out[name + "_frac_diff"] = np.array(fdiff)
        out[name + "_sky"] = np.array(skyout)
        out[name + "_gmix_out"] = gm_data(gm)
        out[name + "_conv_out"] = gm_data(gmc)
        out[name + "_pixels_out"] = pixels.copy()
        print("  em %-26s kind=%d numiter=%d frac_diff=%.3g" % (
            name, kind, numiter, fdiff))

    # 1 gaussian, no psf, seed as in test_em.py
    rng = np.random.RandomState(42587)
    data = _sims.get_ngauss_obs(rng=rng, ngauss=1, noise=0.0)
    guess = data["gmix"].copy()
    randomize(rng, guess, data["obs"].jacobian.scale)
    runcase("g1_nopsf", data["obs"], guess, 0)
    runcase("g1_nopsf_fixcen", data["obs"], guess, 1)
    runcase("g1_nopsf_fixcov", data["obs"], guess, 2)
    runcase("g1_nopsf_fluxonly", data["obs"], guess, 3)
    runcase("g1_nopsf_varysky", data["obs"], guess, 0, vary_sky=True)

    # 2 gaussians, noisy
    rng = np.random.RandomState(587)
    data = _sims.get_ngauss_obs(rng=rng, ngauss=2, noise=0.05)
    guess = data["gmix"].copy()
    randomize(rng, guess, data["obs"].jacobian.scale)
    runcase("g2_noisy", data["obs"], guess, 0, maxiter=60, miniter=10)
    runcase("g2_noisy_fixcen", data["obs"], guess, 1, maxiter=60, miniter=10)
    runcase("g2_noisy_fluxonly", data["obs"], guess, 3, maxiter=60, miniter=10)

    # with a 3-gaussian psf
    rng = np.random.RandomState(4587)
    data = _sims.get_ngauss_obs(rng=rng, ngauss=2, noise=0.0, with_psf=True)
    obs = data["obs"]
    obs.psf.set_gmix(data["psf_gmix"])
    guess = data["gmix"].copy()
    randomize(rng, guess, obs.jacobian.scale)
    runcase("g2_turbpsf", obs, guess, 0, maxiter=50, miniter=20)
    runcase("g2_turbpsf_fixcov", obs, guess, 2, maxiter=50, miniter=20)

    # zero-weight pixels kept: fill_zero_weight path
    rng = np.random.RandomState(77)
    data = _sims.get_ngauss_obs(rng=rng, ngauss=1, noise=0.01)
    obs0 = data["obs"]
    weight = obs0.weight.copy()
    weight[10:13, 11:14] = 0.0
    obsz = ngmix.Observation(obs0.image.copy(), weight=weight,
                             jacobian=obs0.jacobian, ignore_zero_weight=False)
    guess = data["gmix"].copy()
    randomize(rng, guess, obsz.jacobian.scale)
    runcase("g1_zeroweight", obsz, guess, 0, maxiter=45, miniter=40)
    # maxiter reached
    runcase("g1_maxiter5", obs0, guess, 0, maxiter=5, miniter=2, tol=1e-12)
    out["names"] = np.array(names)
    save("em", **out)


# --------------------------------------------------------------------------
def gen_derivs():
    from ngmix.fitting.results import get_model_deriv_data
    rng = np.random.RandomState(1007)
    out = {}
    names = []
    jac = ngmix.Jacobian(row=11.7, col=12.4, dvdrow=0.26, dvdcol=0.01,
                         dudrow=-0.02, dudcol=0.265)
    coords = make_coords((24, 25), jac)
    psf1 = ngmix.GMixModel([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], "gauss")
    psf3 = ngmix.GMixModel([0.0, 0.0, 0.02, -0.01, 0.27, 1.0], "turb")
    for model, pars in (("gauss", [0.1, -0.2, 0.11, -0.07, 0.8, 12.0]),
                        ("exp", [0.1, -0.05, 0.1, 0.05, 0.6, 100.0]),
                        ("dev", [-0.03, 0.02, -0.2, 0.3, 1.7, 55.0])):
        for pname, psf in (("nopsf", None), ("psf1", psf1), ("psf3", psf3)):
            name = "%s_%s" % (model, pname)
            names.append(name)
            gm0 = ngmix.GMixModel(pars, model)
            gmc = gm0 if psf is None else gm0.convolve(psf)
            gpars, dcov = get_model_deriv_data(
                gm0, gmc, pars[2], pars[3], pars[4])
            outim = np.zeros((6, coords.size))
            deriv_images(gpars, dcov, coords["v"].copy(), coords["u"].copy(),
                         coords["area"].copy(), outim)
            out[name + "_pars"] = np.array(pars)
            out[name + "_gpars"] = gpars
            out[name + "_dcov"] = dcov
            out[name + "_out"] = outim
            if psf is not None:
                out[name + "_psf"] = gm_data(psf)
    out["v"] = coords["v"].copy()
    out["u"] = coords["u"].copy()
    out["area"] = coords["area"].copy()
    out["jac"] = jac_data(jac)
    out["dims"] = np.array([24, 25])
    out["names"] = np.array(names)
    _ = rng
    save("derivs", **out)


GENERATORS = {
    "fastexp": gen_fastexp,
    "pixels": gen_pixels,
    "fills": gen_fills,
    "render_loglike": gen_render_loglike,
    "wsums": gen_wsums,
    "admom": gen_admom,
    "em": gen_em,
    "derivs": gen_derivs,
}



# --------------------------------------------------------------------------
# host-level callers: outputs of the reference's
# own classes, for the API shell to reproduce
def _obs_arrays(prefix, obs, out):
    out[prefix + "_image"] = obs.image.copy()
    out[prefix + "_weight"] = obs.weight.copy()
    out[prefix + "_jac"] = jac_data(obs.jacobian)
    if obs.has_psf():
        out[prefix + "_psf_image"] = obs.psf.image.copy()
        out[prefix + "_psf_weight"] = obs.psf.weight.copy()
        out[prefix + "_psf_jac"] = jac_data(obs.psf.jacobian)
        if obs.psf.has_gmix():
            out[prefix + "_psf_gmix_pars"] = obs.psf.gmix.get_full_pars()


def gen_api():
    sys.path.insert(0, "/root/reference/ngmix/tests")
    import _sims
    from ngmix.fitting import Fitter
    from ngmix.fitting.results import FitModel
    out = {}

    # ---- LM: 2 bands x 2 epochs, 'exp' with a turb psf gmix set
    rng = np.random.RandomState(8821)
    data = _sims.get_model_obs(rng=rng, model="exp", noise=0.005,
                               set_psf_gmix=True, nepoch=2, nband=2)
    mbobs = data["obs"]
    truth = np.array(data["pars"])
    out["lm_truth"] = truth
    out["lm_nband"] = np.array(len(mbobs))
    out["lm_nepoch"] = np.array(len(mbobs[0]))
    for b, obslist in enumerate(mbobs):
        for e, obs in enumerate(obslist):
            _obs_arrays("lm_b%d_e%d" % (b, e), obs, out)
    guess = truth.copy()
    guess[0:2] += rng.uniform(-0.02, 0.02, size=2)
    guess[2:4] += rng.uniform(-0.02, 0.02, size=2)
    guess[4] *= 1.0 + rng.uniform(-0.1, 0.1)
    guess[5:] *= 1.0 + rng.uniform(-0.1, 0.1, size=guess.size - 5)
    out["lm_guess"] = guess
    fm = FitModel(obs=mbobs, model="exp", guess=guess)
    for tag, p in (("guess", guess), ("truth", truth)):
        out["lm_fdiff_" + tag] = fm.calc_fdiff(p)
        out["lm_jac_" + tag] = fm.calc_jacobian(p)
        ln = fm.calc_lnprob(p, more=True)
        out["lm_lnprob_" + tag] = np.array(
            [ln["lnprob"], ln["s2n_numer"], ln["s2n_denom"], ln["npix"]])
    bad = guess.copy()
    bad[2:4] = [0.8, 0.7]
    out["lm_bad_pars"] = bad
    out["lm_fdiff_bad"] = fm.calc_fdiff(bad)
    out["lm_jac_bad"] = fm.calc_jacobian(bad)
    for analytic in (True, False):
        tag = "lm_fit_analytic%d" % int(analytic)
        res = Fitter(model="exp", analytic_jacobian=analytic).go(obs=mbobs, guess=guess)
        for k in ("flags", "nfev", "ier", "pars", "pars_err", "pars_cov0",
                  "pars_cov", "lnprob", "s2n_numer", "s2n_denom", "npix",
                  "chi2per", "dof", "s2n", "g", "g_cov", "g_err", "T", "T_err",
                  "flux", "flux_cov", "flux_err"):
            out[tag + "_" + k] = np.array(res[k])
        print("  LM analytic=%s flags=%d nfev=%d ier=%d" % (
            analytic, res["flags"], res["nfev"], res["ier"]))

    # ---- single obs, gauss model, no psf: simplest Fitter path
    rng = np.random.RandomState(311)
    jac = ngmix.DiagonalJacobian(row=15.4, col=15.7, scale=SCALE)
    gm_true = ngmix.GMixModel([0.05, -0.03, 0.1, -0.05, 0.7, 60.0], "gauss")
    im = gm_true.make_image((32, 32), jacobian=jac, fast_exp=True)
    im = im + rng.normal(scale=0.01, size=im.shape)
    obs1 = ngmix.Observation(im, weight=np.full(im.shape, 1e4), jacobian=jac)
    _obs_arrays("g1", obs1, out)
    g1_guess = np.array([0.0, 0.0, 0.05, 0.0, 0.8, 55.0])
    out["g1_guess"] = g1_guess
    res = Fitter(model="gauss").go(obs=obs1, guess=g1_guess)
    for k in ("flags", "nfev", "ier", "pars", "pars_err", "pars_cov", "lnprob",
              "chi2per", "dof", "s2n", "flux", "flux_err", "T", "T_err"):
        out["g1_fit_" + k] = np.array(res[k])
    print("  LM gauss flags=%d nfev=%d" % (res["flags"], res["nfev"]))

    # ---- admom through the public API: every derived key
    rng = np.random.RandomState(911)
    guess_gm = ngmix.GMixModel([0.01, 0.02, 0.0, 0.0, 0.8, 1.0], "gauss")
    ares = ngmix.admom.run_admom(obs1, guess_gm)
    for k, v in ares.items():
        if isinstance(v, str):
            continue
        out["am_" + k] = np.array(v)
    out["am_gmix_pars"] = ares.get_gmix().get_full_pars()
    ares2 = ngmix.admom.run_admom(obs1, 0.8, rng=np.random.RandomState(12))
    out["am_T_guess_pars"] = np.array(ares2["pars"])
    out["am_T_guess_numiter"] = np.array(ares2["numiter"])
    cen = ngmix.admom.find_cen_admom(obs1, fwhm=1.2)
    out["am_cen"] = np.array(cen["cen"])
    out["am_cen_flags"] = np.array(cen["flags"])

    # ---- em through the public API
    rng = np.random.RandomState(42587)
    d = _sims.get_ngauss_obs(rng=rng, ngauss=2, noise=0.01)
    obs_em = d["obs"]
    _obs_arrays("em", obs_em, out)
    guess_em = d["gmix"].copy()
    gd = guess_em.get_data()
    gd["p"] *= 1.04
    gd["row"] += 0.02
    gd["irr"] *= 0.97
    gd["det"] = gd["irr"] * gd["icc"] - gd["irc"] ** 2
    out["em_guess_pars"] = guess_em.get_full_pars()
    for tag, kw in (("full", {}), ("fixcen", {"fixcen": True}),
                    ("fluxonly", {"fluxonly": True})):
        r = ngmix.em.run_em(obs_em, guess_em, miniter=10, maxiter=80, **kw)
        out["em_%s_numiter" % tag] = np.array(r["numiter"])
        out["em_%s_fdiff" % tag] = np.array(r["fdiff"])
        out["em_%s_sky" % tag] = np.array(r["sky"])
        out["em_%s_flags" % tag] = np.array(r["flags"])
        out["em_%s_pars" % tag] = r.get_gmix().get_full_pars()
        out["em_%s_image" % tag] = r.make_image()
        print("  em %s numiter=%d flags=%d" % (tag, r["numiter"], r["flags"]))

    # ---- GMix API odds and ends
    gm = ngmix.GMixModel([0.1, -0.05, 0.1, 0.05, 0.6, 100.0], "exp")
    out["gm_default_jac_image"] = gm.make_image((21, 24), fast_exp=True)
    out["gm_e1e2T"] = np.array(gm.get_e1e2T())
    out["gm_g1g2T"] = np.array(gm.get_g1g2T())
    out["gm_cen"] = np.array(gm.get_cen())
    out["gm_sheared_pars"] = gm.get_sheared(0.02, -0.01).get_full_pars()
    out["gm_round_pars"] = gm.make_round().get_full_pars()
    wm = ngmix.GMixModel([0.0, 0.0, 0.0, 0.0, 0.8, 1.0], "gauss")
    wres = wm.get_weighted_moments(obs1)
    for k in ("flags", "flux", "flux_err", "T", "T_err", "e1", "e2", "s2n",
              "e_err", "wsum", "npix", "sums", "sums_cov"):
        out["wm_" + k] = np.array(wres[k])
    out["gm_loglike_more"] = np.array(list(gm_true.get_loglike(obs1, more=True).values()),
                                      dtype="f8")
    out["gm_model_s2n"] = np.array(gm_true.get_model_s2n(obs1))
    out["obs_s2n"] = np.array(obs1.get_s2n())
    save("api", **out)


GENERATORS["api"] = gen_api

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    for name, func in GENERATORS.items():
        if args.only is None or args.only == name:
            print("generating", name)
            func()
