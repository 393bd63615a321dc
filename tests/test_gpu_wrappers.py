"""
GPU tests of GaussMom / GaussMomBatch / PSFFluxFitter against outputs of the
reference's own classes (tests/golden/extra.npz, oracle/gen_golden_extra.py).
"""
import numpy as np
import pytest

import ngmix_amd as ngmix

pytestmark = pytest.mark.gpu


def _jac(rec):
    r = rec[0] if getattr(rec, "ndim", 0) else rec
    return ngmix.Jacobian(row=float(r["row0"]), col=float(r["col0"]),
                          dvdrow=float(r["dvdrow"]), dvdcol=float(r["dvdcol"]),
                          dudrow=float(r["dudrow"]), dudcol=float(r["dudcol"]))


def _obs(g, with_gmix=None):
    psf = ngmix.Observation(g["psf_image"], jacobian=_jac(g["psf_jac"]),
                            gmix=ngmix.GMix(pars=g["psf_pars"]))
    return ngmix.Observation(g["image"], weight=g["weight"], jacobian=_jac(g["jac"]),
                             psf=psf, gmix=with_gmix)


def _check_mom(res, g, tag):
    assert res["flags"] == int(g[tag + "_flags"]) == 0
    assert res["npix"] == int(g[tag + "_npix"])
    for k in ("flux", "flux_err", "T", "T_err", "s2n", "e1", "e2", "wsum",
              "sums_norm"):
        np.testing.assert_allclose(res[k], float(g[tag + "_" + k]), rtol=1e-11,
                                   err_msg=k)
    for k in ("pars", "sums", "sums_err", "e_err"):
        np.testing.assert_allclose(res[k], g[tag + "_" + k], rtol=1e-11,
                                   atol=1e-13 * np.abs(g[tag + "_" + k]).max(),
                                   err_msg=k)
    ref = g[tag + "_sums_cov"]
    np.testing.assert_allclose(res["sums_cov"], ref, rtol=1e-11,
                               atol=1e-13 * np.abs(ref).max())


@pytest.mark.parametrize("tag,hi", [("gm6", False), ("gm17", True)])
def test_gaussmom(golden, tag, hi):
    g = golden("extra")
    res = ngmix.GaussMom(fwhm=1.2, with_higher_order=hi).go(_obs(g))
    _check_mom(res, g, tag)


def test_gaussmom_batch(golden):
    from ngmix_amd.batch import StampBatch
    g = golden("extra")
    rng = np.random.RandomState(2)
    images = np.stack([g["image"], g["image"] + 0.01 * rng.normal(size=g["image"].shape)])
    weights = np.stack([g["weight"], g["weight"]])
    jac = np.tile(g["jac"].view("f8").reshape(1, 8), (2, 1))
    sb = StampBatch.from_images(images, weights, jac)
    out = ngmix.GaussMomBatch(fwhm=1.2).go(sb)
    _check_mom(out[0], g, "gm6")
    one = ngmix.GaussMom(fwhm=1.2).go(
        ngmix.Observation(images[1], weight=weights[1], jacobian=_jac(g["jac"])))
    for k in ("flux", "T", "e1", "e2", "s2n"):
        np.testing.assert_allclose(out[1][k], one[k], rtol=1e-12)
    # by key: arrays over the stamps, equal to the per-stamp dicts
    assert len(out) == 2 and [r["flags"] for r in out] == [0, 0]
    for k in ("flags", "flux", "flux_err", "T", "T_err", "s2n", "e1", "e2", "e", "e_err",
              "e_cov", "pars", "sums", "sums_cov", "sums_err", "sums_norm", "wsum", "npix",
              "MT", "MT_err", "M1", "M2_err"):
        for i in range(2):
            np.testing.assert_allclose(np.asarray(out[k][i], dtype="f8"),
                                       np.asarray(out[i][k], dtype="f8"), rtol=1e-14,
                                       err_msg=k)


def test_gaussmom_batch_many_is_fast_and_flags_by_array():
    """20k stamps: the statistics are one vectorised pass (seconds of per-object
    Python before), and the failures are visible in the flags array"""
    import time
    from ngmix_amd.batch import StampBatch
    rng = np.random.RandomState(5)
    n, dim = 20000, 32
    jac = np.array([15.5, 15.5, 0.263, 0.0, 0.0, 0.263, 0.263 ** 2, 0.263])
    gm = ngmix.GMixModel([0.0, 0.0, 0.05, -0.03, 0.5, 100.0], "gauss")
    im0 = gm.make_image((dim, dim), jacobian=ngmix.DiagonalJacobian(row=15.5, col=15.5,
                                                                  scale=0.263))
    images = im0[None] + 0.05 * rng.normal(size=(n, dim, dim))
    images[::50] = -np.abs(images[::50])          # negative stamps: NONPOS_FLUX
    weights = np.full((n, dim, dim), 400.0)
    sb = StampBatch.from_images(images, weights, jac)
    fitter = ngmix.GaussMomBatch(fwhm=1.2)
    fitter.go(sb)
    t0 = time.perf_counter()
    res = fitter.go(sb)
    dt = time.perf_counter() - t0
    assert dt < 1.0, dt
    assert np.all(res["flags"][::50] != 0) and np.mean(res["flags"] == 0) > 0.95
    good = np.nonzero(res["flags"] == 0)[0][:5]
    for i in list(good) + [0]:
        one = res[int(i)]
        assert one["flags"] == res["flags"][i]
        np.testing.assert_allclose(res["flux"][i], one["flux"], rtol=1e-14)
        np.testing.assert_allclose(res["T"][i], one["T"], rtol=1e-14, equal_nan=True)


def test_psf_flux(golden):
    g = golden("extra")
    for tag, kw in (("pf", {}), ("pf_nonorm", {"normalize_psf": False})):
        res = ngmix.PSFFluxFitter(**kw).go(_obs(g))
        assert res["flags"] == int(g[tag + "_flags"])
        for k in ("chi2per", "dof", "flux", "flux_err"):
            np.testing.assert_allclose(res[k], float(g[tag + "_" + k]), rtol=1e-9,
                                       err_msg=tag + k)
    obs2 = _obs(g, with_gmix=ngmix.GMix(pars=g["tf_gmix_pars"]))
    res = ngmix.PSFFluxFitter(do_psf=False).go(obs2)
    assert res["flags"] == int(g["tf_flags"])
    for k in ("chi2per", "dof", "flux", "flux_err"):
        np.testing.assert_allclose(res[k], float(g["tf_" + k]), rtol=1e-9)


def test_psf_flux_templates_and_obslist(golden):
    """PSFFluxFitter on template images (psf obs without a mixture) and on a
    two-epoch ObsList, against the reference's own fits
    (results.py:677-914; tests/golden/api2.npz)"""
    g, g2 = golden("extra"), golden("api2")
    for tag, kw in (("pft", {}), ("pft_nonorm", {"normalize_psf": False})):
        psf_obs = ngmix.Observation(g["psf_image"], jacobian=_jac(g["psf_jac"]))
        psf_obs.template = g2["pft_template"]
        obs = ngmix.Observation(g["image"], weight=g["weight"], jacobian=_jac(g["jac"]),
                                psf=psf_obs)
        res = ngmix.PSFFluxFitter(**kw).go(obs)
        assert res["flags"] == int(g2[tag + "_flags"]) and res["model"] == "template"
        for k in ("chi2per", "dof", "flux", "flux_err"):
            np.testing.assert_allclose(res[k], float(g2[tag + "_" + k]), rtol=1e-10,
                                       err_msg=tag + k)
    ol = _nc_obslist(g)
    for tag, kw in (("pfol", {}), ("pfol_nonorm", {"normalize_psf": False})):
        res = ngmix.PSFFluxFitter(**kw).go(ol)
        assert res["flags"] == int(g2[tag + "_flags"])
        for k in ("chi2per", "dof", "flux", "flux_err"):
            np.testing.assert_allclose(res[k], float(g2[tag + "_" + k]), rtol=1e-9,
                                       err_msg=tag + k)
    with pytest.raises(ValueError):
        ngmix.PSFFluxFitter().go([obs])
    bare = ngmix.Observation(g["image"], weight=g["weight"], jacobian=_jac(g["jac"]),
                             psf=ngmix.Observation(g["psf_image"],
                                                   jacobian=_jac(g["psf_jac"])))
    with pytest.raises(ValueError):
        ngmix.PSFFluxFitter().go(bare)


def _nc_obslist(g):
    psf = ngmix.Observation(g["psf_image"], jacobian=_jac(g["psf_jac"]),
                            gmix=ngmix.GMix(pars=g["psf_pars"]))
    ol = ngmix.ObsList()
    for e in range(2):
        pre = "nc_e%d_" % e
        ol.append(ngmix.Observation(g[pre + "image"], weight=g[pre + "weight"],
                                    jacobian=_jac(g[pre + "jac"]), psf=psf,
                                    noise=g[pre + "noise"]))
    return ol


def test_noise_cov_sandwich(golden):
    """Fitter(use_noise_image=True): the noise-power sandwich covariance of the
    reference (noise_cov.py) on two epochs with stationary correlated noise"""
    g = golden("extra")
    ol = _nc_obslist(g)
    for tag, uni in (("nc_plain", False), ("nc_sandwich", True)):
        res = ngmix.fitting.Fitter(model="exp", use_noise_image=uni).go(
            obs=ol, guess=g["nc_guess"])
        assert res["flags"] == int(g[tag + "_flags"]) == 0
        assert res["nfev"] == int(g[tag + "_nfev"])
        np.testing.assert_allclose(res["pars"], g[tag + "_pars"], rtol=1e-7, atol=1e-9)
        refcov = g[tag + "_pars_cov"]
        sig = np.sqrt(np.diag(refcov))
        assert np.all(np.abs(res["pars_cov"] - refcov) <=
                      1e-5 * np.abs(refcov) + 1e-8 * np.outer(sig, sig)), tag
        np.testing.assert_allclose(res["pars_err"], g[tag + "_pars_err"], rtol=1e-5)
    # correlated noise: the sandwich errors are well above the chi2-scaled ones
    assert np.all(g["nc_sandwich_pars_err"] > 2 * g["nc_plain_pars_err"])


def test_noise_cov_central_difference_images(golden):
    from ngmix_amd import noise_cov
    from ngmix_amd.fitting import FitModel
    g = golden("extra")
    ol = _nc_obslist(g)
    fm = FitModel(obs=ol, model="exp", guess=g["nc_guess"])
    ims = noise_cov._dmodel_images_all(fm, g["nc_sandwich_pars"], force_fd=True)[0]
    ref = g["nc_fd_images_e0"]
    for a in range(6):
        np.testing.assert_allclose(ims[a], ref[a], rtol=1e-6,
                                   atol=1e-7 * np.abs(ref[a]).max())
    # and the analytic images agree with them to the accuracy of the differences
    ana = noise_cov._dmodel_images_all(fm, g["nc_sandwich_pars"])[0]
    for a in range(6):
        assert np.abs(ana[a] - ref[a]).max() <= 2e-4 * np.abs(ref[a]).max()


def test_psf_flux_batch_matches_per_object():
    """PSFFluxBatch against PSFFluxFitter object by object: single- and
    multi-epoch objects, masked pixels, normalised and raw psf mixtures"""
    from ngmix_amd.batch import StampBatch, GMixBatch
    rng = np.random.RandomState(3)
    scale = 0.263
    objs, stamps_obs, sobj, psf_recs = [], [], [], []
    for o in range(7):
        nep = 1 + o % 3
        ol = ngmix.ObsList()
        for e in range(nep):
            dim = 17 + 2 * ((o + e) % 3)
            cen = (dim - 1) / 2.0 + rng.uniform(-0.3, 0.3, size=2)
            jac = ngmix.DiagonalJacobian(row=cen[0], col=cen[1], scale=scale)
            pgm = ngmix.GMix(pars=[0.7, 0.0, 0.0, 0.06, 0.002, 0.07,
                                   0.5, 0.01, -0.01, 0.15, 0.0, 0.16])
            flux = rng.uniform(5.0, 50.0)
            im = pgm.make_image((dim, dim), jacobian=jac) * (flux / pgm.get_flux())
            im += 0.01 * rng.normal(size=im.shape)
            wt = np.full(im.shape, 1.0e4)
            if o == 4:
                wt[3:5, 2:9] = 0.0
            pobs = ngmix.Observation(np.zeros((5, 5)), jacobian=jac, gmix=pgm)
            ob = ngmix.Observation(im, weight=wt, jacobian=jac, psf=pobs)
            ol.append(ob)
            stamps_obs.append(ob)
            sobj.append(o)
            psf_recs.append(pgm.get_data().copy())
        objs.append(ol)
    sb = StampBatch.from_observations(stamps_obs)
    psf = GMixBatch.from_numpy(np.stack(psf_recs))
    for normalize in (True, False):
        res = ngmix.PSFFluxBatch(normalize_psf=normalize).go(sb, psf, stamp_obj=sobj)
        for o, ol in enumerate(objs):
            one = ngmix.PSFFluxFitter(normalize_psf=normalize).go(ol)
            assert res["flags"][o] == one["flags"] == 0
            np.testing.assert_allclose(res["flux"][o], one["flux"], rtol=1e-11)
            np.testing.assert_allclose(res["flux_err"][o], one["flux_err"], rtol=1e-9)
            np.testing.assert_allclose(res["chi2per"][o], one["chi2per"], rtol=1e-9)
            np.testing.assert_allclose(res["dof"][o], one["dof"])


def test_noise_cov_batch_matches_per_object(golden):
    """calc_noise_cov_batch / apply_noise_cov_batch on LMBatchFitter results
    against Fitter(use_noise_image=True) object by object (which the golden
    pins to the reference): the golden two-epoch object, plus single-epoch
    objects of another shape with correlated noise"""
    from ngmix_amd.batch import StampBatch, GMixBatch
    from ngmix_amd.lm_batch import LMBatchFitter
    from ngmix_amd.noise_cov import apply_noise_cov_batch
    g = golden("extra")
    objs = [(_nc_obslist(g), g["nc_guess"])]
    rng = np.random.RandomState(12)
    scale = 0.263
    for o in range(3):
        dim = 32
        jac = ngmix.DiagonalJacobian(row=15.5 + 0.2 * o, col=15.5 - 0.1 * o, scale=scale)
        pgm = ngmix.GMixModel([0.0, 0.0, 0.01, 0.02, 0.25, 1.0], "gauss")
        truth = np.array([0.02, -0.03, 0.1, -0.05, 0.5 + 0.1 * o, 80.0])
        im = ngmix.GMixModel(truth, "exp").convolve(pgm).make_image(
            (dim, dim), jacobian=jac, fast_exp=True)
        white = rng.normal(size=(2, dim, dim))
        # correlate along rows: stationary, non-white
        noise = [0.02 * (w + np.roll(w, 1, axis=0) + np.roll(w, 1, axis=1)) for w in white]
        pobs = ngmix.Observation(np.zeros((5, 5)), jacobian=jac, gmix=pgm)
        ob = ngmix.Observation(im + noise[0], weight=np.full(im.shape, 1.0 / (3 * 0.02 ** 2)),
                               jacobian=jac, psf=pobs, noise=noise[1])
        ol = ngmix.ObsList()
        ol.append(ob)
        objs.append((ol, truth * (1.0 + 0.03 * rng.uniform(-1, 1, size=6))))
    flat, sobj, psfs, guesses = [], [], [], []
    for i, (ol, guess) in enumerate(objs):
        for ob in ol:
            flat.append(ob)
            sobj.append(i)
            psfs.append(ob.psf.gmix.get_data().copy())
        guesses.append(guess)
    sb = StampBatch.from_observations(flat)
    noise = np.concatenate([np.asarray(ob.noise, dtype="f8").ravel() for ob in flat])
    import torch
    d_noise = torch.from_numpy(noise).to(sb.device)
    # one gaussian count per batch: pad the smaller psf mixtures with
    # zero-flux components (they add nothing to any convolved model)
    ngmax = max(p.size for p in psfs)
    padded = []
    for p in psfs:
        q = np.concatenate([p] + [p[:1]] * (ngmax - p.size))
        q["p"][p.size:] = 0.0
        padded.append(q)
    psf = GMixBatch.from_numpy(np.stack(padded))
    psf.set_norms()
    sobj = np.array(sobj, dtype=np.int32)
    res = LMBatchFitter("exp").go(sb, np.array(guesses), psf=psf, stamp_obj=sobj)
    plain_err = res["pars_err"].copy()
    apply_noise_cov_batch(res, sb, d_noise, "exp", psf=psf, stamp_obj=sobj)
    assert np.all(res["flags"] == 0)
    for i, (ol, guess) in enumerate(objs):
        one = ngmix.fitting.Fitter(model="exp", use_noise_image=True).go(obs=ol, guess=guess)
        assert one["flags"] == 0
        np.testing.assert_allclose(res["pars"][i], one["pars"], rtol=1e-6, atol=1e-8)
        sig = np.sqrt(np.diag(one["pars_cov"]))
        assert np.all(np.abs(res["pars_cov"][i] - one["pars_cov"]) <=
                      1e-4 * np.abs(one["pars_cov"]) + 1e-6 * np.outer(sig, sig)), i
        np.testing.assert_allclose(res["pars_err"][i], one["pars_err"], rtol=1e-4)
    # correlated noise: well above the chi2-scaled errors
    assert np.all(res["pars_err"][0] > 2 * plain_err[0])


def _nc_batch(g):
    """the golden two-epoch object as a device batch: stamps, noise, psf"""
    import torch
    from ngmix_amd.batch import StampBatch, GMixBatch
    ol = _nc_obslist(g)
    sb = StampBatch.from_observations(list(ol))
    noise = torch.from_numpy(np.concatenate(
        [np.asarray(ob.noise, dtype="f8").ravel() for ob in ol])).to(sb.device)
    psf = GMixBatch.from_numpy(np.stack([ob.psf.gmix.get_data().copy() for ob in ol]))
    return ol, sb, noise, psf


@pytest.mark.parametrize("model", ["turb", "bdf"])
def test_noise_cov_batch_central_differences_vs_reference(golden, model):
    """the sandwich covariance for the models without analytic derivative
    images (noise_cov.py:140-224: central differences of two fast renders per
    parameter), batched, against the REFERENCE's own Fitter(model,
    use_noise_image=True) on the two-epoch object (tests/golden/api2.npz)"""
    from ngmix_amd.noise_cov import calc_noise_cov_batch
    g, g2 = golden("extra"), golden("api2")
    ol, sb, noise, psf = _nc_batch(g)
    pre = "ncfd_%s_" % model
    assert int(g2[pre + "flags"]) == 0
    pars, cov0, ref = g2[pre + "pars"], g2[pre + "pars_cov0"], g2[pre + "pars_cov"]
    sobj = np.zeros(2, dtype=np.int64)
    cov = calc_noise_cov_batch(sb, noise, model, pars[None], cov0[None], psf=psf,
                               stamp_obj=sobj)[0]
    sig = np.sqrt(np.diag(ref))
    np.testing.assert_allclose(cov, ref, rtol=1e-6, atol=1e-8 * np.outer(sig, sig).max())
    # the same object twice in one batch, one copy shifted to another chunk
    sb2 = type(sb).from_observations(list(ol) + list(ol))
    import torch
    cov2 = calc_noise_cov_batch(sb2, torch.cat([noise, noise]), model,
                                np.stack([pars, pars]), np.stack([cov0, cov0]),
                                psf=type(psf)(torch.cat([psf.data, psf.data]), 4, psf.ngauss),
                                stamp_obj=np.array([0, 0, 1, 1]), chunk_stamps=3)
    np.testing.assert_allclose(cov2[0], cov, rtol=1e-12)
    np.testing.assert_allclose(cov2[1], cov, rtol=1e-12)
    # the per-object Fitter of the shell goes through _dmodel for these models
    one = ngmix.fitting.Fitter(model=model, use_noise_image=True).go(
        obs=ol, guess=g2[pre + "guess"])
    assert one["flags"] == 0
    np.testing.assert_allclose(one["pars"], pars, rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(one["pars_cov"], ref, rtol=5e-3,
                               atol=1e-5 * np.outer(sig, sig).max())


def test_noise_cov_batch_forced_central_differences(golden):
    """force_fd on an analytic model: the central-difference derivative images
    equal the reference's (extra.npz nc_fd_images_e0) and the covariance the
    analytic one to the differencing error"""
    from ngmix_amd.batch import StampBatch, GMixBatch
    from ngmix_amd.noise_cov import calc_noise_cov_batch, _central_difference_images
    import torch
    g = golden("extra")
    ol, sb, noise, psf = _nc_batch(g)
    pars, cov0 = g["nc_sandwich_pars"], g["nc_sandwich_pars_cov0"]
    sobj = np.zeros(2, dtype=np.int64)
    a = calc_noise_cov_batch(sb, noise, "exp", pars[None], cov0[None], psf=psf,
                             stamp_obj=sobj)[0]
    b = calc_noise_cov_batch(sb, noise, "exp", pars[None], cov0[None], psf=psf,
                             stamp_obj=sobj, force_fd=True)[0]
    sig = np.sqrt(np.diag(a))
    np.testing.assert_allclose(b, a, rtol=1e-5, atol=1e-7 * np.outer(sig, sig).max())
    geom = StampBatch(None, None, sb.jac[:1], np.array([32]), np.array([32]),
                      np.zeros(1, dtype=np.int64), True)
    bp = torch.from_numpy(pars[None].copy()).to(sb.device)
    D, bad = _central_difference_images(
        geom, "exp", bp, GMixBatch(psf.data[:psf.ngauss].contiguous(), 1, psf.ngauss), 5)
    assert not bool(bad.any())
    ref = g["nc_fd_images_e0"]
    got = D.cpu().numpy().reshape(ref.shape)
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-11 * np.abs(ref).max())


def test_template_sums_kernel_vs_numpy():
    """ngmix_template_sums_batch (csrc/template.hip) on a ragged batch with
    zero and negative weights, with and without multipliers, against the sums
    written out in numpy (results.py:700-770's expressions); run to run the
    same to the bit"""
    import ctypes
    import torch
    from ngmix_amd import _lib
    from ngmix_amd.batch import StampBatch, _dptr, _stream
    rng = np.random.RandomState(12)
    shapes = [(25, 25), (33, 31), (48, 48), (7, 64), (1, 1), (40, 17)]
    imgs = [rng.normal(size=s) for s in shapes]
    wts = [rng.uniform(0.5, 2.0, size=s) for s in shapes]
    wts[1][3:6, 4:9] = 0.0
    wts[3][2, :] = -1.0
    jacs = [np.array([(s[0] - 1) / 2, (s[1] - 1) / 2, 0.263, 0.0, 0.0, 0.263, 0.263 ** 2, 0.263])
            for s in shapes]
    sb = StampBatch.from_arrays(imgs, wts, jacs, [True] * len(shapes))
    models = [rng.uniform(0.0, 1.0, size=s) for s in shapes]
    d_model = torch.from_numpy(np.concatenate([m.ravel() for m in models])).cuda()
    mult = rng.uniform(0.5, 3.0, size=len(shapes))
    d_mult = torch.from_numpy(mult).cuda()
    L = _lib.lib()
    b = sb._batch(1)

    def run(dm):
        out = torch.empty((len(shapes), 4), dtype=torch.float64, device="cuda")
        _lib.check(L.ngmix_template_sums_batch(ctypes.byref(b), _dptr(d_model), _dptr(dm)
                                               if dm is not None else None, _dptr(out),
                                               _stream()), "template_sums")
        torch.cuda.synchronize()
        return out.cpu().numpy()
    for dm, a in ((None, np.ones(len(shapes))), (d_mult, mult)):
        got = run(dm)
        assert np.array_equal(got, run(dm))
        for i, s in enumerate(shapes):
            ie = np.sqrt(np.clip(wts[i], 0.0, None))
            w = ie * ie
            mm = models[i] * a[i]
            ref = [np.sum(mm * imgs[i] * w), np.sum(mm * mm * w),
                   np.sum((mm - imgs[i]) ** 2 * w), float(np.sum(ie > 0))]
            np.testing.assert_allclose(got[i], ref, rtol=1e-12, atol=1e-13)
            assert got[i, 3] == ref[3]
