"""
GPU tests of the batched Levenberg-Marquardt driver (ngmix_amd/lm_batch.py,
csrc/lmfit.hip) against (i) the reference's own Fitter.go on a 2-band x
2-epoch object (tests/golden/api.npz, generated from the reference by
oracle/gen_golden.py) and (ii) this package's per-object Fitter, which runs
scipy's MINPACK over the same residual / jacobian kernels.

The batched iteration follows lmder from the normal equations J^T J, J^T f
instead of the QR of J, so iterates agree to the rounding of that
factorisation; parameters are compared at 1e-6 relative and the iteration
counts exactly.
"""
import numpy as np
import pytest

import ngmix_amd as ngmix
from ngmix_amd import _lib
from ngmix_amd.batch import StampBatch, GMixBatch
from ngmix_amd.lm_batch import LMBatchFitter

pytestmark = pytest.mark.gpu


def _jac(rec):
    r = rec[0] if getattr(rec, "ndim", 0) else rec
    return ngmix.Jacobian(row=float(r["row0"]), col=float(r["col0"]),
                          dvdrow=float(r["dvdrow"]), dvdcol=float(r["dvdcol"]),
                          dudrow=float(r["dudrow"]), dudcol=float(r["dudcol"]))


def _psf_records(pars):
    gm = ngmix.GMix(pars=pars)
    return gm.get_data().copy()


def test_golden_multiband_object(golden):
    g = golden("api")
    nband, nepoch = int(g["lm_nband"]), int(g["lm_nepoch"])
    obs, psfs, band = [], [], []
    for b in range(nband):
        for e in range(nepoch):
            pre = "lm_b%d_e%d" % (b, e)
            obs.append(ngmix.Observation(g[pre + "_image"], weight=g[pre + "_weight"],
                                         jacobian=_jac(g[pre + "_jac"])))
            psfs.append(_psf_records(g[pre + "_psf_gmix_pars"]))
            band.append(b)
    sb = StampBatch.from_observations(obs)
    psf = GMixBatch.from_numpy(np.stack(psfs))
    fitter = LMBatchFitter("exp")
    res = fitter.go(sb, g["lm_guess"][None, :], psf=psf,
                    stamp_obj=np.zeros(len(obs), dtype=np.int32),
                    stamp_band=np.array(band, dtype=np.int32))
    tag = "lm_fit_analytic1"
    assert res["flags"][0] == int(g[tag + "_flags"]) == 0
    assert res["ier"][0] == int(g[tag + "_ier"])
    assert res["nfev"][0] == int(g[tag + "_nfev"])
    np.testing.assert_allclose(res["pars"][0], g[tag + "_pars"], rtol=1e-6, atol=1e-8)
    refcov = g[tag + "_pars_cov"]
    sig = np.sqrt(np.diag(refcov))
    tol = 1e-4 * np.abs(refcov) + 1e-7 * np.outer(sig, sig)
    assert np.all(np.abs(res["pars_cov"][0] - refcov) <= tol)
    np.testing.assert_allclose(res["pars_err"][0], g[tag + "_pars_err"], rtol=1e-4)
    for k in ("lnprob", "chi2per", "s2n", "T", "T_err"):
        np.testing.assert_allclose(res[k][0], float(g[tag + "_" + k]), rtol=1e-5)
    assert res["dof"][0] == int(g[tag + "_dof"]) and res["npix"][0] == int(g[tag + "_npix"])
    np.testing.assert_allclose(res["g"][0], g[tag + "_g"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(res["flux"][0], g[tag + "_flux"], rtol=1e-6)
    np.testing.assert_allclose(res["flux_err"][0], g[tag + "_flux_err"], rtol=1e-4)
    assert res["flux_cov"].shape == (1, 2, 2)


def _make_objects(n, model, rng, dim=32, scale=0.263, noise=0.01):
    pars = np.zeros((n, 6))
    pars[:, 0:2] = rng.uniform(-0.5, 0.5, size=(n, 2)) * scale
    pars[:, 2:4] = rng.normal(scale=0.1, size=(n, 2))
    pars[:, 4] = rng.uniform(0.3, 1.0, size=n)
    pars[:, 5] = rng.uniform(50.0, 200.0, size=n)
    psf_pars = np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (n, 1))
    cen = (dim - 1) / 2.0
    jac = np.array([cen, cen, scale, 0.0, 0.0, scale, scale ** 2, scale])
    gm0, _ = GMixBatch.from_pars(pars, model)
    psf, _ = GMixBatch.from_pars(psf_pars, "gauss")
    gm, _ = gm0.convolve(psf)
    geom = StampBatch.from_images(np.zeros((n, dim, dim)), None, jac)
    truth, _ = geom.render(gm)
    images = truth.cpu().numpy().reshape(n, dim, dim) + noise * rng.normal(size=(n, dim, dim))
    weights = np.full((n, dim, dim), 1.0 / noise ** 2)
    sb = StampBatch.from_images(images, weights, jac)
    guess = pars * rng.uniform(0.9, 1.1, size=pars.shape)
    guess[:, 0:2] = pars[:, 0:2] + rng.uniform(-0.05, 0.05, size=(n, 2))
    guess[:, 2:4] = pars[:, 2:4] + rng.uniform(-0.03, 0.03, size=(n, 2))
    return pars, guess, images, weights, jac, sb, psf


@pytest.mark.parametrize("model", ["exp", "gauss", "dev"])
def test_batch_matches_per_object_fitter(model):
    rng = np.random.RandomState({"exp": 11, "gauss": 12, "dev": 13}[model])
    n = 24
    pars, guess, images, weights, jac, sb, psf = _make_objects(n, model, rng)
    res = LMBatchFitter(model).go(sb, guess, psf=psf)
    assert np.all(res["flags"] == 0)
    psf_rec = psf.to_numpy()
    jobj = ngmix.Jacobian(row=jac[0], col=jac[1], dvdrow=jac[2], dvdcol=jac[3],
                          dudrow=jac[4], dudcol=jac[5])
    same_nfev = 0
    for i in range(n):
        pgm = ngmix.GMix(ngauss=1)
        pgm.get_data()[:] = psf_rec[i]
        pobs = ngmix.Observation(np.zeros((5, 5)), jacobian=jobj, gmix=pgm)
        obs = ngmix.Observation(images[i], weight=weights[i], jacobian=jobj, psf=pobs)
        one = ngmix.fitting.Fitter(model=model).go(obs=obs, guess=guess[i])
        assert one["flags"] == 0
        assert res["ier"][i] == one["ier"]
        same_nfev += int(res["nfev"][i] == one["nfev"])
        err = one["pars_err"]
        # the same minimum: far inside the statistical error
        assert np.all(np.abs(res["pars"][i] - one["pars"]) <= 1e-4 * err), i
        np.testing.assert_allclose(res["pars_err"][i], err, rtol=1e-3)
        np.testing.assert_allclose(res["lnprob"][i], one["lnprob"], rtol=1e-6)
        np.testing.assert_allclose(res["s2n"][i], one["s2n"], rtol=1e-6)
        assert res["dof"][i] == one["dof"]
    # the same iteration path for (nearly) all of them
    assert same_nfev >= n - 2
    # and the fits recover the truth within the errors
    pull = (res["pars"] - pars) / res["pars_err"]
    assert np.all(np.abs(pull) < 6.0)


@pytest.mark.parametrize("nsplit", [2, 3])
def test_batch_in_pieces_on_streams_is_the_same_fit(nsplit, monkeypatch):
    """LMBatchFitter.nsplit: the batch advanced in pieces on separate streams,
    and the generic lm_advance (NGMIX_LM_GENERIC is read once per process, so
    here only the pieces are compared): every result to the bit"""
    rng = np.random.RandomState(77)
    n = 50
    pars, guess, images, weights, jac, sb, psf = _make_objects(n, "exp", rng)
    whole = LMBatchFitter("exp").go(sb, guess, psf=psf)
    f = LMBatchFitter("exp")
    f.nsplit = nsplit
    pieces = f.go(sb, guess, psf=psf)
    assert f.nsplit_used == nsplit
    for key in ("flags", "nfev", "njev", "ier", "pars", "pars_err", "pars_cov", "lnprob",
                "s2n", "dof"):
        np.testing.assert_array_equal(np.asarray(whole[key]), np.asarray(pieces[key]), key)


@pytest.mark.parametrize("npars,mode,bounds", [
    (6, 0, False), (6, 0, True), (7, 1, False), (8, 1, True), (14, 1, False)])
def test_device_init_equals_host_init(npars, mode, bounds):
    """ngmix_lm_init_batch against ngmix_lm_init (lmcore::lm_init): the same
    records over everything a fit of npars parameters reads -- every scalar, the
    first npars entries of each per-parameter array (lo / hi / ipvt: all of
    them), the leading npars x npars block of R.  The dead part of the 14-
    parameter record is left as allocated (the device init writes ~1 kB per
    six-parameter fit instead of zero-filling 2.9 kB)"""
    import torch
    from ngmix_amd.batch import _dptr, _stream
    L = _lib.lib()
    rng = np.random.RandomState(npars * 10 + mode)
    n = 300
    x0 = rng.uniform(0.2, 2.0, size=(n, npars))
    x0[5, 0] = 0.0           # fdjac2's step at zero
    lo = hi = None
    if bounds:
        lo = np.full(npars, -np.inf)
        hi = np.full(npars, np.inf)
        lo[0], hi[0] = -1.0, 4.0
        lo[2] = 0.01
        hi[3] = 9.0
    host = np.zeros(n, dtype=_lib.LM_STATE_DTYPE)
    host.view(np.uint8)[:] = 0xAB        # lm_init must define every byte it owns
    args = (1e-7, 1e-9, 0.0, 123, 50.0, mode,
            None if lo is None else _lib.ptr(lo), None if hi is None else _lib.ptr(hi))
    assert L.ngmix_lm_init(_lib.ptr(host), n, npars, _lib.ptr(x0), *args) == 0
    dev = torch.full((n, _lib.LM_STATE_DTYPE.itemsize), 0xCD, dtype=torch.uint8,
                     device="cuda")
    d_x0 = torch.from_numpy(x0).cuda()
    assert L.ngmix_lm_init_batch(_dptr(dev), n, npars, _dptr(d_x0), *args,
                                 _stream()) == 0
    torch.cuda.synchronize()
    got = dev.cpu().numpy().reshape(-1).view(_lib.LM_STATE_DTYPE)
    # with bounds the transforms go through asin / sin / sqrt: device and host
    # libm differ by an ulp or two there, everything else is the same bytes
    libm = ("x", "xt", "xi", "xti", "xstep", "hstep") if bounds else ()
    for name in _lib.LM_STATE_DTYPE.names:
        a, b = got[name], host[name]
        if name == "R":
            a, b = a[:, :npars, :npars], b[:, :npars, :npars]
        elif a.ndim == 2 and name not in ("lo", "hi", "ipvt"):
            a, b = a[:, :npars], b[:, :npars]
        if name in libm:
            np.testing.assert_allclose(a, b, rtol=1e-14, atol=0, err_msg=name)
        else:
            np.testing.assert_array_equal(a, b, name)
    # the dead part was not touched
    if npars < _lib.LM_NPMAX:
        assert np.all(got["R"][:, npars:, :].view(np.uint8) == 0xCD)
        assert np.all(got["diag"][:, npars:].view(np.uint8) == 0xCD)


def test_batch_out_of_range_start_and_masked():
    rng = np.random.RandomState(5)
    n = 6
    pars, guess, images, weights, jac, _, psf = _make_objects(n, "exp", rng)
    guess[2, 2] = 1.2          # |g| >= 1 at the starting point: no finite residual
    weights[4, 10:14, 8:20] = 0.0  # a masked strip
    sb = StampBatch.from_images(images, weights, jac)
    res = LMBatchFitter("exp").go(sb, guess, psf=psf)
    # the per-object Fitter (as the reference's) refuses such a guess when it
    # builds the model; in a batch it is what MINPACK makes of -inf residuals
    # and a zero jacobian: the gradient test ends the fit at once (ier 4) with
    # a singular factor
    jobj = ngmix.Jacobian(row=jac[0], col=jac[1], dvdrow=jac[2], dvdcol=jac[3],
                          dudrow=jac[4], dudcol=jac[5])
    with pytest.raises(ngmix.GMixRangeError):
        _fit_one("exp", images[2], weights[2], jobj, psf.to_numpy()[2], guess[2], True)
    assert res["flags"][2] == ngmix.flags.LM_SINGULAR_MATRIX
    assert res["nfev"][2] == 1 and res["ier"][2] == 4
    np.testing.assert_array_equal(res["pars"][2], guess[2])
    okmask = np.arange(n) != 2
    assert np.all(res["flags"][okmask] == 0)
    assert res["npix"][4] == 32 * 32 - 4 * 12
    pull = (res["pars"][okmask] - pars[okmask]) / res["pars_err"][okmask]
    assert np.all(np.abs(pull) < 6.0)


def test_bootstrap_batch_recovers_truth():
    """psf admom -> guess admom -> batched LM, everything on the device"""
    from ngmix_amd.pipeline import bootstrap_batch
    rng = np.random.RandomState(77)
    n, dim, scale, noise = 200, 40, 0.263, 0.01
    pars = np.zeros((n, 6))
    pars[:, 0:2] = rng.uniform(-0.5, 0.5, size=(n, 2)) * scale
    pars[:, 2:4] = rng.normal(scale=0.1, size=(n, 2))
    pars[:, 4] = rng.uniform(0.3, 1.0, size=n)
    pars[:, 5] = rng.uniform(50.0, 200.0, size=n)
    psf_T = rng.uniform(0.22, 0.32, size=n)
    psf_pars = np.zeros((n, 6))
    psf_pars[:, 2:4] = rng.normal(scale=0.02, size=(n, 2))
    psf_pars[:, 4] = psf_T
    psf_pars[:, 5] = 1.0
    cen = (dim - 1) / 2.0
    jac = np.array([cen, cen, scale, 0.0, 0.0, scale, scale ** 2, scale])
    gm0, _ = GMixBatch.from_pars(pars, "exp")
    psf, _ = GMixBatch.from_pars(psf_pars, "gauss")
    gm, _ = gm0.convolve(psf)
    geom = StampBatch.from_images(np.zeros((n, dim, dim)), None, jac)
    truth, _ = geom.render(gm)
    images = truth.cpu().numpy().reshape(n, dim, dim) + noise * rng.normal(size=(n, dim, dim))
    sb = StampBatch.from_images(images, np.full((n, dim, dim), 1.0 / noise ** 2), jac)
    pdim = 25
    pjac = np.array([12.0, 12.0, scale, 0.0, 0.0, scale, scale ** 2, scale])
    pgeom = StampBatch.from_images(np.zeros((n, pdim, pdim)), None, pjac)
    pim, _ = pgeom.render(psf)
    pimages = pim.cpu().numpy().reshape(n, pdim, pdim) + 1e-5 * rng.normal(size=(n, pdim, pdim))
    psb = StampBatch.from_images(pimages, np.full((n, pdim, pdim), 1e10), pjac)
    res = bootstrap_batch(sb, psb, model="exp", psf_Tguess=0.3)
    assert np.all(res["psf_flags"] == 0)
    np.testing.assert_allclose(res["psf_T"], psf_T, rtol=2e-3)
    assert np.all(res["flags"] == 0)
    pull = (res["pars"] - pars) / res["pars_err"]
    assert np.all(np.abs(pull) < 6.0)
    assert 0.7 < np.sqrt((pull ** 2).mean()) < 1.3


def test_prep_em_matches_prep_image():
    """the batched sky shift of the EM fitters, uniform and ragged batches"""
    from ngmix_amd.em import prep_image
    rng = np.random.RandomState(5)
    ims = rng.normal(size=(7, 12, 9))
    sb = StampBatch.from_images(ims, None, None)
    b2, sky = sb.prep_em()
    for i in range(7):
        im, s = prep_image(ims[i])
        np.testing.assert_allclose(sky[i].item(), s, rtol=1e-14)
        np.testing.assert_allclose(b2.val.reshape(7, 12, 9)[i].cpu().numpy(), im,
                                   rtol=0, atol=1e-15)
    obs = [ngmix.Observation(rng.normal(size=(5 + i, 8 - i))) for i in range(4)]
    rb = StampBatch.from_observations(obs)
    b3, sky3 = rb.prep_em()
    v = b3.val.cpu().numpy()
    for i, o in enumerate(obs):
        im, s = prep_image(o.image)
        np.testing.assert_allclose(sky3[i].item(), s, rtol=1e-14)
        a = rb.pix_off[i]
        np.testing.assert_allclose(v[a:a + im.size], im.ravel(), rtol=0, atol=1e-15)


def test_bootstrap_batch_with_em_psf():
    """a two-gaussian psf: the EM psf fit (psf_ngauss=2) removes the size bias
    the single adaptive-moments gaussian leaves"""
    from ngmix_amd.pipeline import bootstrap_batch
    rng = np.random.RandomState(78)
    n, dim, scale, noise = 150, 40, 0.263, 0.005
    pars = np.zeros((n, 6))
    pars[:, 0:2] = rng.uniform(-0.5, 0.5, size=(n, 2)) * scale
    pars[:, 2:4] = rng.normal(scale=0.1, size=(n, 2))
    pars[:, 4] = rng.uniform(0.3, 0.8, size=n)
    pars[:, 5] = rng.uniform(100.0, 200.0, size=n)
    full = np.zeros((n, 2, 6))
    Tc = rng.uniform(0.18, 0.24, size=n)
    full[:, 0, 0], full[:, 1, 0] = 0.7, 0.3
    full[:, 0, 3] = full[:, 0, 5] = 0.5 * Tc
    full[:, 1, 3] = full[:, 1, 5] = 0.5 * Tc * 3.0
    full[:, 1, 4] = 0.02 * Tc
    psf, _ = GMixBatch.from_pars(full.reshape(n, -1), "full", ngauss=2)
    psf_T = 0.7 * Tc + 0.3 * 3.0 * Tc
    cen = (dim - 1) / 2.0
    jac = np.array([cen, cen, scale, 0.0, 0.0, scale, scale ** 2, scale])
    gm0, _ = GMixBatch.from_pars(pars, "exp")
    gm, _ = gm0.convolve(psf)
    geom = StampBatch.from_images(np.zeros((n, dim, dim)), None, jac)
    truth, _ = geom.render(gm)
    images = truth.cpu().numpy().reshape(n, dim, dim) + noise * rng.normal(size=(n, dim, dim))
    sb = StampBatch.from_images(images, np.full((n, dim, dim), 1.0 / noise ** 2), jac)
    pdim = 33
    pjac = np.array([16.0, 16.0, scale, 0.0, 0.0, scale, scale ** 2, scale])
    pgeom = StampBatch.from_images(np.zeros((n, pdim, pdim)), None, pjac)
    pim, _ = pgeom.render(psf)
    pimages = pim.cpu().numpy().reshape(n, pdim, pdim) + 1e-6 * rng.normal(size=(n, pdim, pdim))
    psb = StampBatch.from_images(pimages, np.full((n, pdim, pdim), 1e12), pjac)

    # the same psf fit with co-elliptical gaussians in the lock-step LM
    # (psf images with noise 1e-6 of the flux: the forward-difference jacobian
    # in the nearly-zero centre parameters is at the edge of double precision,
    # for lmdif as for this driver; PSFRunner's retry covers the stray failure)
    resc = bootstrap_batch(sb, psb, model="exp", psf_Tguess=0.3, psf_ngauss=2,
                           psf_fitter="coellip", psf_ntry=3)
    assert np.all(resc["psf_em_flags"] == 0) and np.all(resc["flags"] == 0)
    pullc = (resc["pars"] - pars) / resc["pars_err"]
    assert np.all(np.abs(pullc) < 6.0) and np.sqrt((pullc ** 2).mean()) < 1.4
    res2 = bootstrap_batch(sb, psb, model="exp", psf_Tguess=0.3, psf_ngauss=2)
    assert np.all(res2["psf_em_flags"] == 0)
    assert np.all(res2["flags"] == 0)
    fitted = res2["psf_gmix"].to_numpy()
    np.testing.assert_allclose(fitted["p"].sum(axis=1), 1.0, rtol=1e-12)
    Tfit = (fitted["p"] * (fitted["irr"] + fitted["icc"])).sum(axis=1)
    np.testing.assert_allclose(Tfit, psf_T, rtol=0.02)
    pull2 = (res2["pars"] - pars) / res2["pars_err"]
    assert np.all(np.abs(pull2) < 6.0)
    assert np.sqrt((pull2 ** 2).mean()) < 1.4
    # the one-gaussian psf misses the wings: its T is biased
    res1 = bootstrap_batch(sb, psb, model="exp", psf_Tguess=0.3, psf_ngauss=1)
    pull1 = (res1["pars"] - pars) / res1["pars_err"]
    assert np.abs(pull1[:, 4].mean()) > 2.0 * np.abs(pull2[:, 4].mean()) + 0.5


def _fit_one(model, image, weight, jobj, psf_rec, guess, analytic):
    pgm = ngmix.GMix(ngauss=1)
    pgm.get_data()[:] = psf_rec
    pobs = ngmix.Observation(np.zeros((5, 5)), jacobian=jobj, gmix=pgm)
    obs = ngmix.Observation(image, weight=weight, jacobian=jobj, psf=pobs)
    return ngmix.fitting.Fitter(model=model, analytic_jacobian=analytic).go(
        obs=obs, guess=guess)


@pytest.mark.parametrize("model", ["exp", "bdf", "turb"])
def test_batch_forward_difference_mode(model):
    """lmdif in lock step (the jacobian by MINPACK's forward differences inside
    the pixel pass) against the per-object Fitter's scipy lmdif"""
    rng = np.random.RandomState({"exp": 21, "bdf": 22, "turb": 23}[model])
    n, dim, scale, noise = 12, 32, 0.263, 0.01
    base = "exp" if model == "bdf" else model
    pars, guess, images, weights, jac, sb, psf = _make_objects(n, base, rng, dim=dim)
    if model == "bdf":
        # re-simulate as bdf: [cen1, cen2, g1, g2, T, fracdev, flux]
        fracdev = rng.uniform(0.2, 0.8, size=n)
        pars = np.column_stack([pars[:, :5], fracdev, pars[:, 5]])
        gm0, _ = GMixBatch.from_pars(pars, "bdf")
        gm, _ = gm0.convolve(psf)
        geom = StampBatch.from_images(np.zeros((n, dim, dim)), None, jac)
        truth, _ = geom.render(gm)
        images = truth.cpu().numpy().reshape(n, dim, dim) + noise * rng.normal(size=(n, dim, dim))
        sb = StampBatch.from_images(images, weights, jac)
        guess = np.column_stack([guess[:, :5], fracdev * rng.uniform(0.9, 1.1, size=n),
                                 guess[:, 5]])
    fitter = LMBatchFitter(model, analytic_jacobian=False)
    assert fitter.fd
    res = fitter.go(sb, guess, psf=psf)
    psf_rec = psf.to_numpy()
    jobj = ngmix.Jacobian(row=jac[0], col=jac[1], dvdrow=jac[2], dvdcol=jac[3],
                          dudrow=jac[4], dudcol=jac[5])
    same_nfev = 0
    checked = 0
    for i in range(n):
        one = _fit_one(model, images[i], weights[i], jobj, psf_rec[i], guess[i], False)
        if one["flags"] != 0 or res["flags"][i] != 0:
            # a noisy fracdev can walk out of range; both paths must agree on it
            assert (one["flags"] != 0) == (res["flags"][i] != 0), i
            continue
        checked += 1
        same_nfev += int(res["nfev"][i] == one["nfev"])
        err = one["pars_err"]
        # forward differences: the two paths agree far inside the errors
        assert np.all(np.abs(res["pars"][i] - one["pars"]) <= 2e-2 * err), (i, model)
        np.testing.assert_allclose(res["pars_err"][i], err, rtol=2e-2)
        np.testing.assert_allclose(res["lnprob"][i], one["lnprob"], rtol=1e-4)
    assert checked >= n - 2
    assert same_nfev >= checked - 3
    ok = res["flags"] == 0
    pull = (res["pars"][ok] - pars[ok]) / res["pars_err"][ok]
    if model != "bdf":   # fracdev is nearly unconstrained at this S/N
        assert np.all(np.abs(pull) < 6.0)


def _multi_gauss_psf(n, ngauss, offset, rng):
    """n psf mixtures of ngauss gaussians (flux 1), co-centred or with offset
    components, as 'full' parameter rows and as a GMixBatch"""
    full = np.zeros((n, ngauss, 6))
    frac = np.array([0.55, 0.25, 0.12, 0.05, 0.03])[:ngauss]
    frac = frac / frac.sum()
    for i in range(ngauss):
        full[:, i, 0] = frac[i]
        if offset:
            full[:, i, 1:3] = rng.uniform(-0.04, 0.04, size=(n, 2)) * (i > 0)
        sig2 = 0.135 * (1.0 + 0.9 * i)
        full[:, i, 3] = sig2 * (1.0 + 0.05 * i)
        full[:, i, 4] = 0.01 * sig2 * (-1) ** i
        full[:, i, 5] = sig2
    gm, st = GMixBatch.from_pars(full.reshape(n, -1), "full", ngauss=ngauss)
    assert int(st.abs().sum()) == 0
    return full.reshape(n, -1), gm


def _objects_with_psf(n, model, psf, rng, dim=40, scale=0.263, noise=0.01, extra=None):
    pars = np.zeros((n, 6))
    pars[:, 0:2] = rng.uniform(-0.5, 0.5, size=(n, 2)) * scale
    pars[:, 2:4] = rng.normal(scale=0.1, size=(n, 2))
    pars[:, 4] = rng.uniform(0.3, 0.8, size=n)
    pars[:, 5] = rng.uniform(80.0, 200.0, size=n)
    if extra is not None:
        pars = np.column_stack([pars[:, :5], extra, pars[:, 5]])
    cen = (dim - 1) / 2.0
    jac = np.array([cen, cen, scale, 0.0, 0.0, scale, scale ** 2, scale])
    gm0, _ = GMixBatch.from_pars(pars, model)
    gm, _ = gm0.convolve(psf)
    geom = StampBatch.from_images(np.zeros((n, dim, dim)), None, jac)
    truth, _ = geom.render(gm)
    images = truth.cpu().numpy().reshape(n, dim, dim) + noise * rng.normal(size=(n, dim, dim))
    weights = np.full((n, dim, dim), 1.0 / noise ** 2)
    sb = StampBatch.from_images(images, weights, jac)
    guess = pars * rng.uniform(0.93, 1.07, size=pars.shape)
    guess[:, 0:2] = pars[:, 0:2] + rng.uniform(-0.04, 0.04, size=(n, 2))
    guess[:, 2:4] = pars[:, 2:4] + rng.uniform(-0.02, 0.02, size=(n, 2))
    jobj = ngmix.Jacobian(row=jac[0], col=jac[1], dvdrow=jac[2], dvdcol=jac[3],
                          dudrow=jac[4], dudcol=jac[5])
    return pars, guess, images, weights, jobj, sb


def _fit_one_psf(model, image, weight, jobj, psf_full_row, guess, analytic):
    pobs = ngmix.Observation(np.zeros((5, 5)), jacobian=jobj, gmix=ngmix.GMix(pars=psf_full_row))
    obs = ngmix.Observation(image, weight=weight, jacobian=jobj, psf=pobs)
    return ngmix.fitting.Fitter(model=model, analytic_jacobian=analytic).go(obs=obs, guess=guess)


@pytest.mark.parametrize("model,npsf,offset", [("dev", 4, False), ("dev", 5, True),
                                               ("exp", 3, True)])
def test_analytic_fits_with_many_psf_gaussians(model, npsf, offset):
    """lm_eval_kernel with more than 32 composed gaussians (dev (x) 4 or 5: the
    per-tile box tests instead of the ten-tiles-per-ballot ones) and with psf
    components off the psf centre, against the per-object Fitter"""
    rng = np.random.RandomState(100 + npsf)
    n = 6
    psf_rows, psf = _multi_gauss_psf(n, npsf, offset, rng)
    pars, guess, images, weights, jobj, sb = _objects_with_psf(n, model, psf, rng)
    res = LMBatchFitter(model).go(sb, guess, psf=psf)
    assert np.all(res["flags"] == 0)
    for i in range(n):
        one = _fit_one_psf(model, images[i], weights[i], jobj, psf_rows[i], guess[i], True)
        assert one["flags"] == 0 and res["ier"][i] == one["ier"]
        assert abs(int(res["nfev"][i]) - int(one["nfev"])) <= 1
        assert np.all(np.abs(res["pars"][i] - one["pars"]) <= 1e-4 * one["pars_err"]), i
        np.testing.assert_allclose(res["pars_err"][i], one["pars_err"], rtol=1e-3)
        np.testing.assert_allclose(res["lnprob"][i], one["lnprob"], rtol=1e-6)


@pytest.mark.parametrize("model,npsf,offset", [("bdf", 2, False), ("bdf", 2, True),
                                               ("turb", 3, True), ("exp", 1, False)])
def test_forward_difference_fits_psf_centres_and_zero_flux_guess(model, npsf, offset):
    """lm_eval_fd_kernel: the one-centre-per-set path (co-centred psf) and the
    per-gaussian-centre path (psf components off centre), with the flux column
    taken from the base model -- and, for one object, a guess with flux exactly
    0 (that shortcut is then off) -- against the per-object Fitter's lmdif"""
    rng = np.random.RandomState(300 + npsf + int(offset))
    n = 6
    psf_rows, psf = _multi_gauss_psf(n, npsf, offset, rng)
    extra = rng.uniform(0.3, 0.7, size=n) if model == "bdf" else None
    pars, guess, images, weights, jobj, sb = _objects_with_psf(n, model, psf, rng, extra=extra)
    guess[0, -1] = 0.0
    res = LMBatchFitter(model, analytic_jacobian=False).go(sb, guess, psf=psf)
    checked = 0
    for i in range(n):
        one = _fit_one_psf(model, images[i], weights[i], jobj, psf_rows[i], guess[i], False)
        if one["flags"] != 0 or res["flags"][i] != 0:
            assert (one["flags"] != 0) == (res["flags"][i] != 0), i
            continue
        checked += 1
        assert np.all(np.abs(res["pars"][i] - one["pars"]) <= 2e-2 * one["pars_err"]), (i, model)
        np.testing.assert_allclose(res["pars_err"][i], one["pars_err"], rtol=2e-2)
        np.testing.assert_allclose(res["lnprob"][i], one["lnprob"], rtol=1e-4)
    assert checked >= n - 1


def test_analytic_fit_of_a_stamp_with_more_tiles_than_lds_records():
    """a 264 x 264 stamp has 1089 8x8 tiles, more than the tile records the
    analytic LM kernel keeps in LDS: the kernel makes them on the fly"""
    rng = np.random.RandomState(9)
    dim, scale = 264, 0.263
    pars = np.array([[0.05, -0.03, 0.1, -0.05, 20.0, 5000.0],
                     [-0.1, 0.08, -0.08, 0.12, 35.0, 9000.0]])
    psf_rows, psf = _multi_gauss_psf(2, 1, False, rng)
    cen = (dim - 1) / 2.0
    jac = np.array([cen, cen, scale, 0.0, 0.0, scale, scale ** 2, scale])
    gm0, _ = GMixBatch.from_pars(pars, "gauss")
    gm, _ = gm0.convolve(psf)
    geom = StampBatch.from_images(np.zeros((2, dim, dim)), None, jac)
    truth, _ = geom.render(gm)
    images = truth.cpu().numpy().reshape(2, dim, dim) + 0.001 * rng.normal(size=(2, dim, dim))
    weights = np.full((2, dim, dim), 1.0e6)
    sb = StampBatch.from_images(images, weights, jac)
    guess = pars * np.array([1.0, 1.0, 0.9, 1.1, 1.05, 0.95])
    res = LMBatchFitter("gauss").go(sb, guess, psf=psf)
    assert np.all(res["flags"] == 0)
    jobj = ngmix.Jacobian(row=jac[0], col=jac[1], dvdrow=jac[2], dvdcol=jac[3],
                          dudrow=jac[4], dudcol=jac[5])
    for i in range(2):
        one = _fit_one_psf("gauss", images[i], weights[i], jobj, psf_rows[i], guess[i], True)
        assert one["flags"] == 0 and res["ier"][i] == one["ier"]
        assert np.all(np.abs(res["pars"][i] - one["pars"]) <= 1e-4 * one["pars_err"]), i
        np.testing.assert_allclose(res["lnprob"][i], one["lnprob"], rtol=1e-6)


@pytest.mark.parametrize("model,fd,npsf,offset", [("exp", 0, 1, False), ("dev", 0, 3, True),
                                                  ("bdf", 1, 2, True), ("turb", 1, 1, False)])
def test_lm_eval_skipping_is_exact(model, fd, npsf, offset):
    """the (tile, gaussian) pairs the LM kernels skip -- the bounding box of the
    chi2 < 25 ellipse -- contribute exactly nothing: the normal-equation
    sums with NGMIX_BATCH_NO_SKIP are the same, bit for bit (sheared mixtures,
    off-centre objects and psf components, both kernels)"""
    import ctypes
    import torch
    from ngmix_amd.batch import _dptr, _stream
    from ngmix_amd.gmix import get_model_num
    L = _lib.lib()
    rng = np.random.RandomState(500 + npsf + fd)
    n = 64
    psf_rows, psf = _multi_gauss_psf(n, npsf, offset, rng)
    extra = rng.uniform(0.3, 0.7, size=n) if model == "bdf" else None
    pars, guess, images, weights, jobj, sb = _objects_with_psf(n, model, psf, rng, dim=48,
                                                              extra=extra)
    # strongly sheared, off-centre trial points: thin ellipses across the tiles
    guess[:, 2] = rng.uniform(-0.6, 0.6, size=n)
    guess[:, 3] = rng.uniform(-0.6, 0.6, size=n)
    guess[:, 0:2] += rng.uniform(-1.5, 1.5, size=(n, 2))
    npars = guess.shape[1]
    nsum = npars * (npars + 1) // 2 + npars + 1
    out = []
    for no_skip in (False, True):
        st = torch.empty((n, _lib.LM_STATE_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
        dg = torch.from_numpy(np.ascontiguousarray(guess)).cuda()
        _lib.check(L.ngmix_lm_init_batch(_dptr(st), n, npars, _dptr(dg), 1e-8, 1e-8, 0.0, 100,
                                         100.0, _lib.LM_MODE_FD if fd else _lib.LM_MODE_ANALYTIC,
                                         None, None, _stream()), "init")
        sobj = torch.arange(n, dtype=torch.int32, device="cuda")
        sband = torch.zeros(n, dtype=torch.int32, device="cuda")
        sums = torch.zeros((n, nsum), dtype=torch.float64, device="cuda")
        status = torch.zeros(n, dtype=torch.int32, device="cuda")
        b = sb._batch(1, no_skip=no_skip)
        _lib.check(L.ngmix_lm_eval_batch(ctypes.byref(b), get_model_num(model), fd, _dptr(st),
                                         _dptr(sobj), _dptr(sband), _dptr(psf.data), npsf,
                                         _dptr(sums), _dptr(status), None, _stream()), "eval")
        torch.cuda.synchronize()
        assert int(status.abs().sum()) == 0
        out.append(sums.cpu().numpy())
    assert np.all(np.isfinite(out[0])) and np.abs(out[0]).max() > 0
    np.testing.assert_array_equal(out[0], out[1])


@pytest.mark.parametrize("model,dim", [("bdf", 25), ("turb", 33), ("exp", 27)])
def test_lm_eval_fd_row_major_tiles(model, dim, monkeypatch):
    """the forward-difference pixel pass takes the pixels of stamps that fill
    8 x 8 tiles badly (25 x 25: 16 tiles for 625 pixels) 64 at a time in row-major
    order (10 tiles): the normal-equation sums are those of the 8 x 8 form to
    the rounding of a different order of addition, skipping is still exact (the
    row-range box test), masked pixels and the jacobian phase included; the
    launch census names the form, and NGMIX_LM_FD_TILES forces either"""
    import ctypes
    import torch
    from ngmix_amd.batch import _dptr, _stream
    from ngmix_amd.gmix import get_model_num
    L = _lib.lib()
    rng = np.random.RandomState(600 + dim)
    n = 48
    psf_rows, psf = _multi_gauss_psf(n, 2, True, rng)
    extra = rng.uniform(0.3, 0.7, size=n) if model == "bdf" else None
    pars, guess, images, weights, jobj, sb = _objects_with_psf(n, model, psf, rng, dim=dim,
                                                              extra=extra)
    weights[::5, 3:7, 2:11] = 0.0
    jac = np.array([(dim - 1) / 2.0, (dim - 1) / 2.0, 0.263, 0.0, 0.0, 0.263, 0.263 ** 2, 0.263])
    sb = StampBatch.from_images(images, weights, jac)
    guess[:, 2:4] = rng.uniform(-0.5, 0.5, size=(n, 2))
    npars = guess.shape[1]
    nsum = npars * (npars + 1) // 2 + npars + 1

    def sums_of(tiles, no_skip):
        if tiles is None:
            monkeypatch.delenv("NGMIX_LM_FD_TILES", raising=False)
        else:
            monkeypatch.setenv("NGMIX_LM_FD_TILES", tiles)
        st = torch.empty((n, _lib.LM_STATE_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
        dg = torch.from_numpy(np.ascontiguousarray(guess)).cuda()
        _lib.check(L.ngmix_lm_init_batch(_dptr(st), n, npars, _dptr(dg), 1e-8, 1e-8, 0.0, 100,
                                         100.0, _lib.LM_MODE_FD, None, None, _stream()), "init")
        sobj = torch.arange(n, dtype=torch.int32, device="cuda")
        sband = torch.zeros(n, dtype=torch.int32, device="cuda")
        sums = torch.zeros((n, nsum), dtype=torch.float64, device="cuda")
        status = torch.zeros(n, dtype=torch.int32, device="cuda")
        b = sb._batch(1, no_skip=no_skip)
        _lib.launch_census(reset=True)
        _lib.check(L.ngmix_lm_eval_batch(ctypes.byref(b), get_model_num(model), 1, _dptr(st),
                                         _dptr(sobj), _dptr(sband), _dptr(psf.data), 2,
                                         _dptr(sums), _dptr(status), None, _stream()), "eval")
        torch.cuda.synchronize()
        seen = _lib.launch_census(reset=True)
        assert int(status.abs().sum()) == 0
        return sums.cpu().numpy(), seen
    auto, seen = sums_of(None, False)
    assert any("lm_eval_fd_kernel<%d, linear>" % npars in k for k in seen), seen
    lin, _ = sums_of("linear", False)
    lin_ns, _ = sums_of("linear", True)
    two, seen2 = sums_of("2d", False)
    assert any(k == "lm_eval_fd_kernel<%d>" % npars for k in seen2), seen2
    np.testing.assert_array_equal(auto, lin)
    np.testing.assert_array_equal(lin, lin_ns)          # skipping is exact
    scale = np.abs(two).max(axis=1, keepdims=True)
    assert np.all(np.isfinite(two)) and scale.min() > 0
    np.testing.assert_allclose(lin, two, rtol=0, atol=1e-12 * scale.max())
    assert np.all(np.abs(lin - two) <= 1e-11 * scale)
    # a 32 x 32 batch keeps the 8 x 8 tiles
    monkeypatch.delenv("NGMIX_LM_FD_TILES", raising=False)
    rng2 = np.random.RandomState(9)
    p2, g2, im2, w2, j2, sb2, psf2 = _make_objects(8, "exp", rng2)
    _lib.launch_census(reset=True)
    LMBatchFitter("exp", analytic_jacobian=False).go(sb2, g2, psf=psf2)
    seen3 = _lib.launch_census(reset=True)
    assert "lm_eval_fd_kernel<6>" in seen3 and not any("linear" in k for k in seen3), seen3


def test_bootstrap_batch_on_ragged_and_selected_stamps():
    """the flux guess of bootstrap_batch on stamps of different shapes (a
    segmented sum on the device) and on a selection that shares its parent's
    pixel arrays: the pixel sums of the stamps themselves"""
    from ngmix_amd.pipeline import bootstrap_batch
    rng = np.random.RandomState(12)
    scale = 0.263
    shapes = [(32, 32), (40, 36), (32, 32), (36, 40), (48, 48), (32, 32)]
    obs, pobs = [], []
    for nrow, ncol in shapes:
        jac = ngmix.DiagonalJacobian(row=(nrow - 1) / 2.0, col=(ncol - 1) / 2.0, scale=scale)
        pgm = ngmix.GMixModel([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], "gauss")
        gm = ngmix.GMixModel([0.02, -0.03, 0.08, -0.05, 0.6, rng.uniform(80, 160)], "exp")
        im = gm.convolve(pgm).make_image((nrow, ncol), jacobian=jac, fast_exp=True)
        im += 0.01 * rng.normal(size=im.shape)
        obs.append(ngmix.Observation(im, weight=np.full(im.shape, 1e4), jacobian=jac))
        pj = ngmix.DiagonalJacobian(row=12.0, col=12.0, scale=scale)
        pim = pgm.make_image((25, 25), jacobian=pj) + 1e-5 * rng.normal(size=(25, 25))
        pobs.append(ngmix.Observation(pim, weight=np.full((25, 25), 1e8), jacobian=pj))
    sb = StampBatch.from_observations(obs)
    psb = StampBatch.from_observations(pobs)
    res = bootstrap_batch(sb, psb, model="exp")
    assert np.all(res["flags"] == 0)
    sums = np.array([o.image.sum() for o in obs])
    np.testing.assert_allclose(res["guess"][:, 5], sums, rtol=1e-12)
    idx = np.array([4, 1, 3])
    sub = bootstrap_batch(sb.select(idx), psb.select(idx), model="exp")
    np.testing.assert_allclose(sub["guess"][:, 5], sums[idx], rtol=1e-12)
    assert np.all(sub["flags"] == 0)


def test_bootstrap_batch_multiband_multiepoch():
    """objects with 2 bands x 2 epochs, different psf per epoch, sub-pixel
    offsets per epoch: one bootstrap_batch call recovers shape, size and both
    fluxes"""
    from ngmix_amd.pipeline import bootstrap_batch
    rng = np.random.RandomState(101)
    nobj, nband, nepoch, dim, pdim, scale, noise = 60, 2, 2, 36, 25, 0.263, 0.01
    truth = np.zeros((nobj, 7))
    truth[:, 0:2] = rng.uniform(-0.3, 0.3, size=(nobj, 2)) * scale
    truth[:, 2:4] = rng.normal(scale=0.1, size=(nobj, 2))
    truth[:, 4] = rng.uniform(0.3, 0.8, size=nobj)
    truth[:, 5] = rng.uniform(60.0, 120.0, size=nobj)
    truth[:, 6] = rng.uniform(100.0, 200.0, size=nobj)
    ns = nobj * nband * nepoch
    sobj = np.repeat(np.arange(nobj), nband * nepoch)
    sband = np.tile(np.repeat(np.arange(nband), nepoch), nobj)
    bp = np.zeros((ns, 6))
    bp[:, :5] = truth[sobj, :5]
    bp[:, 5] = truth[sobj, 5 + sband]
    psf_pars = np.zeros((ns, 6))
    psf_pars[:, 2:4] = rng.normal(scale=0.02, size=(ns, 2))
    psf_pars[:, 4] = rng.uniform(0.22, 0.32, size=ns)
    psf_pars[:, 5] = 1.0
    jac = np.zeros((ns, 8))
    jac[:, 0] = (dim - 1) / 2.0 + rng.uniform(-0.5, 0.5, size=ns)
    jac[:, 1] = (dim - 1) / 2.0 + rng.uniform(-0.5, 0.5, size=ns)
    jac[:, 2] = jac[:, 5] = jac[:, 7] = scale
    jac[:, 6] = scale ** 2
    gm0, _ = GMixBatch.from_pars(bp, "exp")
    psf, _ = GMixBatch.from_pars(psf_pars, "gauss")
    gm, _ = gm0.convolve(psf)
    geom = StampBatch.from_images(np.zeros((ns, dim, dim)), None, jac)
    images = geom.render(gm)[0].cpu().numpy().reshape(ns, dim, dim)
    images = images + noise * rng.normal(size=images.shape)
    sb = StampBatch.from_images(images, np.full(images.shape, 1.0 / noise ** 2), jac)
    pjac = np.array([12.0, 12.0, scale, 0.0, 0.0, scale, scale ** 2, scale])
    pgeom = StampBatch.from_images(np.zeros((ns, pdim, pdim)), None, pjac)
    pim = pgeom.render(psf)[0].cpu().numpy().reshape(ns, pdim, pdim)
    pim = pim + 1e-5 * rng.normal(size=pim.shape)
    psb = StampBatch.from_images(pim, np.full(pim.shape, 1e10), pjac)
    res = bootstrap_batch(sb, psb, model="exp", psf_Tguess=0.3, stamp_obj=sobj,
                          stamp_band=sband)
    assert res["pars"].shape == (nobj, 7)
    assert np.all(res["flags"] == 0)
    assert np.all(res["npix"] == nband * nepoch * dim * dim)
    pull = (res["pars"] - truth) / res["pars_err"]
    assert np.all(np.abs(pull) < 6.0)
    assert 0.7 < np.sqrt((pull ** 2).mean()) < 1.3
    assert res["flux"].shape == (nobj, 2) and res["flux_cov"].shape == (nobj, 2, 2)


def test_select_and_bootstrap_retries():
    """StampBatch.select gathers stamps (uniform and ragged batches);
    bootstrap_batch(ntry=...) refits only the failed objects"""
    from ngmix_amd.pipeline import bootstrap_batch
    rng = np.random.RandomState(55)
    n, dim, scale = 40, 32, 0.263
    pars, guess, images, weights, jac, _, psf = _make_objects(n, "exp", rng)
    sb = StampBatch.from_images(images, weights, jac)
    gm0, _ = GMixBatch.from_pars(pars, "exp")
    gm, _ = gm0.convolve(psf)
    full = sb.loglike(gm)[0].cpu().numpy()
    idx = np.array([5, 0, 17, 17, 39, 2])
    sub = sb.select(idx)
    gsub = GMixBatch(gm.data.reshape(n, gm.ngauss, 13)[idx].reshape(-1, 13).contiguous(),
                     idx.size, gm.ngauss)
    np.testing.assert_array_equal(sub.loglike(gsub)[0].cpu().numpy(), full[idx])
    obs = [ngmix.Observation(rng.normal(size=(6 + i, 9 - i)),
                             weight=rng.uniform(0.5, 2, size=(6 + i, 9 - i)))
           for i in range(5)]
    rb = StampBatch.from_observations(obs)
    pick = np.array([3, 1, 4])
    rs = rb.select(pick)
    assert list(rs.nrow) == [9, 7, 10] and rs.total_pix == sum(obs[i].image.size for i in pick)
    v = rs.val.cpu().numpy()
    a = 0
    for i in pick:
        np.testing.assert_array_equal(v[a:a + obs[i].image.size], obs[i].image.ravel())
        a += obs[i].image.size
    # retries: with a tiny maxfev some fits end with flags != 0; those, and
    # only those, are tried again
    pdim = 25
    pjac = np.array([12.0, 12.0, scale, 0.0, 0.0, scale, scale ** 2, scale])
    pgeom = StampBatch.from_images(np.zeros((n, pdim, pdim)), None, pjac)
    pim = pgeom.render(psf)[0].cpu().numpy().reshape(n, pdim, pdim)
    psb = StampBatch.from_images(pim + 1e-6 * rng.normal(size=pim.shape),
                                 np.full(pim.shape, 1e10), pjac)
    ref = bootstrap_batch(sb, psb, model="exp", fit_pars={"ftol": 1e-10, "xtol": 1e-10},
                          rng=np.random.RandomState(1))
    assert ref["nfev"].min() < ref["nfev"].max()
    fp = {"maxfev": int(np.sort(ref["nfev"])[n // 2]), "ftol": 1e-10, "xtol": 1e-10}
    one = bootstrap_batch(sb, psb, model="exp", fit_pars=fp, rng=np.random.RandomState(1))
    three = bootstrap_batch(sb, psb, model="exp", fit_pars=fp, rng=np.random.RandomState(1),
                            ntry=3)
    failed = one["flags"] != 0
    assert failed.any() and not failed.all()
    assert np.all(three["ntry"][~failed] == 1) and np.all(three["ntry"][failed] >= 2)
    np.testing.assert_array_equal(three["pars"][~failed], one["pars"][~failed])
    assert (three["flags"] != 0).sum() <= failed.sum()
    # the merge of a retry covers the keys that stay on the device until read
    # (pars_cov0): a retried object that ends well carries ITS fit's matrix,
    # pars_cov = pars_cov0 * chi2 / dof (run_leastsq, fitters.py:312-330)
    fixed = failed & (three["flags"] == 0)
    assert fixed.any()
    assert "pars_cov0" in dict(three) and len(three) == len(three.keys())
    ratio = three["pars_cov"][fixed] / three["pars_cov0"][fixed]
    np.testing.assert_allclose(ratio, np.broadcast_to(
        three["chi2per"][fixed][:, None, None], ratio.shape), rtol=1e-6)
    ok = (~failed) & (three["flags"] == 0)
    np.testing.assert_array_equal(three["pars_cov0"][ok], one["pars_cov0"][ok])


@pytest.mark.parametrize("model", ["turb", "bdf", "dev", "bd"])
def test_reference_lmdif_fits(golden, model):
    """tests/golden/lmfd.npz: the REFERENCE's Fitter on two-band objects for
    the models it fits with MINPACK lmdif (no analytic derivatives).  The
    per-object Fitter reproduces nfev / ier / pars; the batched forward-
    difference mode (fdjac2 inside the pixel pass) reaches the same solution
    ('bd' with two bands has 9 of the batch state's 10 parameters)"""
    g = golden("lmfd")
    nband = 2
    obs, band = [], []
    mb = ngmix.MultiBandObsList()
    for b in range(nband):
        jac = _jac(g["%s_jac%d" % (model, b)])
        pobs = ngmix.Observation(np.zeros((5, 5)), jacobian=jac,
                                 gmix=ngmix.GMix(pars=g["psf_pars"]))
        o = ngmix.Observation(g["%s_image%d" % (model, b)],
                              weight=g["%s_weight%d" % (model, b)], jacobian=jac, psf=pobs)
        ol = ngmix.ObsList()
        ol.append(o)
        mb.append(ol)
        obs.append(o)
        band.append(b)
    guess = g[model + "_guess"]
    ref = {k: g["%s_%s" % (model, k)] for k in ("flags", "nfev", "ier", "pars", "pars_err",
                                                 "pars_cov", "lnprob", "chi2per")}
    assert int(ref["flags"]) == 0
    one = ngmix.fitting.Fitter(model=model, analytic_jacobian=False).go(obs=mb, guess=guess)
    assert one["flags"] == 0 and one["ier"] == int(ref["ier"])
    assert abs(one["nfev"] - int(ref["nfev"])) <= 2 * guess.size + 2
    # forward differences of a fast-exp model: the two implementations' 1e-13
    # pixel differences are amplified by 1/h ~ 1e8 in the jacobian
    assert np.all(np.abs(one["pars"] - ref["pars"]) <= 2e-3 * ref["pars_err"])
    np.testing.assert_allclose(one["pars_err"], ref["pars_err"], rtol=2e-3)
    np.testing.assert_allclose(one["lnprob"], float(ref["lnprob"]), rtol=1e-8)
    sb = StampBatch.from_observations(obs)
    psf = GMixBatch.from_numpy(np.stack([o.psf.gmix.get_data().copy() for o in obs]))
    res = LMBatchFitter(model, analytic_jacobian=False).go(
        sb, guess[None, :], psf=psf, stamp_obj=np.zeros(nband, dtype=np.int32),
        stamp_band=np.array(band, dtype=np.int32))
    assert res["flags"][0] == 0
    assert abs(res["nfev"][0] - int(ref["nfev"])) <= 2 * guess.size + 2
    assert np.all(np.abs(res["pars"][0] - ref["pars"]) <= 2e-3 * ref["pars_err"])
    np.testing.assert_allclose(res["pars_err"][0], ref["pars_err"], rtol=2e-3)
    np.testing.assert_allclose(res["lnprob"][0], float(ref["lnprob"]), rtol=1e-8)
    np.testing.assert_allclose(res["chi2per"][0], float(ref["chi2per"]), rtol=1e-7)


@pytest.mark.parametrize("ngauss", [2, 3])
def test_reference_coellip_fits_batched(golden, ngauss):
    """LMBatchFitter('coellip', ngauss=...) -- the batched CoellipFitter, the
    psf fitter of the LM psf runners -- against the reference's own fits of a
    psf image (tests/golden/lmfd.npz), several copies of it in one batch"""
    g = golden("lmfd")
    jac = _jac(g["coellip_jac"])
    im = g["coellip_image"]
    wt = np.full(im.shape, 1.0 / 2.0e-4 ** 2)
    pre = "coellip%d_" % ngauss
    guess = g[pre + "guess"]
    nrep = 3
    obs = [ngmix.Observation(im, weight=wt, jacobian=jac) for _ in range(nrep)]
    sb = StampBatch.from_observations(obs)
    fitter = LMBatchFitter("coellip", ngauss=ngauss)
    res = fitter.go(sb, np.tile(guess, (nrep, 1)))
    assert res["pars"].shape == (nrep, 4 + 2 * ngauss)
    err = g[pre + "pars_err"]
    for i in range(nrep):
        assert res["flags"][i] == int(g[pre + "flags"]) == 0
        assert res["ier"][i] in (1, 2, 3)
        # (round 6: the reference's evaluation count to the unit, the solution to
        # 1e-3 sigma -- measured: equal nfev, 1e-6 / 7e-6 sigma, errors to 1e-5)
        assert res["nfev"][i] == int(g[pre + "nfev"])
        assert np.all(np.abs(res["pars"][i] - g[pre + "pars"]) <= 1e-3 * err), i
        np.testing.assert_allclose(res["pars_err"][i], err, rtol=1e-3)
        np.testing.assert_allclose(res["lnprob"][i], float(g[pre + "lnprob"]), rtol=1e-6)
    np.testing.assert_array_equal(res["pars"][0], res["pars"][1])
    fitted = fitter.gmix.to_numpy()[0]
    ref = g[pre + "gmix_pars"].reshape(ngauss, 6)
    np.testing.assert_allclose(np.sort(fitted["p"]), np.sort(ref[:, 0]), rtol=2e-3)
    with pytest.raises(ValueError):
        LMBatchFitter("coellip", ngauss=6)


@pytest.mark.parametrize("ngauss", [4, 5])
def test_reference_coellip_four_and_five_gaussians(golden, ngauss):
    """CoellipFitter with four and five gaussians (12 and 14 parameters; the
    reference's psf guessers go up to five, guessers.py:795-797), batched, against
    the reference's own fits (tests/golden/api2.npz) -- the normal equations of
    these run on v_mfma_f64_16x16x4_f64 -- and the per-object CoellipFitter of
    the shell (scipy lmdif over the fdiff kernel)"""
    g = golden("api2")
    pre = "coellip%d_" % ngauss
    jac = _jac(g[pre + "jac"])
    im = g[pre + "image"]
    wt = np.full(im.shape, 1.0 / 2.0e-5 ** 2)
    guess = g[pre + "guess"]
    nrep = 3
    obs = [ngmix.Observation(im, weight=wt, jacobian=jac) for _ in range(nrep)]
    sb = StampBatch.from_observations(obs)
    fitter = LMBatchFitter("coellip", ngauss=ngauss)
    res = fitter.go(sb, np.tile(guess, (nrep, 1)))
    assert res["pars"].shape == (nrep, 4 + 2 * ngauss)
    err = g[pre + "pars_err"]
    assert int(g[pre + "flags"]) == 0
    for i in range(nrep):
        assert res["flags"][i] == 0
        assert res["ier"][i] in (1, 2, 3)
        # round 6: on these fits the driver follows lmdif's path to the unit of
        # nfev (measured: 53 / 53 and 121 / 121, parameters to 4e-5 sigma, errors
        # to 2e-5 -- the covariance from the double-double factor of
        # lm_precise.hip); over random fits of this kind the count drifts in the
        # degenerate valleys (profiles/r06_fuzz_lm_fd_vs_minpack.log), so one
        # jacobian of slack is left
        assert abs(res["nfev"][i] - int(g[pre + "nfev"])) <= 4 + 2 * ngauss + 1
        assert np.all(np.abs(res["pars"][i] - g[pre + "pars"]) <= 1e-3 * err), i
        np.testing.assert_allclose(res["pars_err"][i], err, rtol=1e-3)
        np.testing.assert_allclose(res["lnprob"][i], float(g[pre + "lnprob"]), rtol=1e-6)
    np.testing.assert_array_equal(res["pars"][0], res["pars"][2])
    one = ngmix.fitting.CoellipFitter(ngauss=ngauss).go(obs=obs[0], guess=guess)
    assert one["flags"] == 0
    assert np.all(np.abs(one["pars"] - g[pre + "pars"]) <= 3e-2 * err)
    # the truth is recovered
    assert np.all(np.abs(res["pars"][0] - g[pre + "truth"]) <= 5.0 * err)


def test_go_stream_pipeline_equals_go():
    """LMBatchFitter.go_stream (finalise / download of batch i under the first
    rounds of batch i + 1) returns, for every batch of the sequence, the
    arrays go() returns for it -- bit for bit, in order; an empty sequence
    and a sequence of one work too"""
    from ngmix_amd.lm_batch import LMBatchFitter
    rng = np.random.RandomState(91)
    items = []
    for k, (n, model) in enumerate([(300, "exp"), (120, "exp"), (257, "exp"), (64, "exp")]):
        pars, guess, images, weights, jac, _, psf = _make_objects(n, model, rng)
        sb = StampBatch.from_images(images, weights, jac)
        items.append((sb, guess, {"psf": psf}))
    fitter = LMBatchFitter("exp")
    ref = [fitter.go(sb, g, **kw) for sb, g, kw in items]
    got = list(fitter.go_stream(items))
    assert len(got) == len(ref)
    for a, b in zip(ref, got):
        assert set(a.keys()) == set(b.keys())
        for key in ("flags", "nfev", "njev", "ier", "pars", "pars_err", "pars_cov", "pars_cov0",
                    "lnprob", "chi2per", "s2n", "npix", "dof", "g_cov", "T_err", "flux"):
            np.testing.assert_array_equal(a[key], b[key], err_msg=key)
    assert list(fitter.go_stream([])) == []
    one = list(fitter.go_stream(items[:1]))
    np.testing.assert_array_equal(one[0]["pars"], ref[0]["pars"])
    # a forward-difference fitter and a prior go through the same stages
    fd = LMBatchFitter("exp", analytic_jacobian=False)
    r1 = fd.go(items[1][0], items[1][1], psf=items[1][2]["psf"])
    r2 = list(fd.go_stream(items[1:2]))[0]
    np.testing.assert_array_equal(r1["pars"], r2["pars"])
    np.testing.assert_array_equal(r1["lnprob"], r2["lnprob"])


def _same_fit(a, b, keys=("flags", "nfev", "njev", "ier", "pars", "pars_err", "pars_cov",
                          "pars_cov0", "lnprob", "chi2per", "s2n", "s2n_numer", "s2n_denom",
                          "npix", "dof")):
    for key in keys:
        np.testing.assert_array_equal(a[key], b[key], err_msg=key)


@pytest.mark.parametrize("model", ["exp", "gauss", "dev"])
def test_lazy_jacobian_is_the_eager_fit_to_the_bit(model):
    """mode ANALYTIC_LAZY (the default): the trials predicted to end a fit are
    evaluated for |f|^2 alone by the lean pixel pass of lm_eval_kernel; a failed
    prediction gets its jacobian one round later.  The result -- nfev, njev,
    ier, parameters, covariance, statistics -- is the eager fit's, bit for bit,
    on single stamps, on multi-epoch two-band objects, masked pixels, poor
    guesses (rejected steps) and with the kernel prior."""
    from ngmix_amd import prior_batch as pb
    rng = np.random.RandomState({"exp": 41, "gauss": 42, "dev": 43}[model])
    n = 200
    pars, guess, images, weights, jac, sb, psf = _make_objects(n, model, rng)
    weights[::7, 3:9, 4:11] = 0.0            # masked stamps
    sb = StampBatch.from_images(images, weights, jac)
    guess[::5] = pars[::5] * rng.uniform(0.5, 1.8, size=(pars[::5].shape))   # poor guesses
    guess[::5, 2:4] = rng.uniform(-0.3, 0.3, size=guess[::5, 2:4].shape)
    lazy = LMBatchFitter(model)
    eager = LMBatchFitter(model)
    eager.lazy_jacobian = False
    r_lazy = lazy.go(sb, guess, psf=psf)
    st = lazy.states()
    assert np.all(st["mode"] == _lib.LM_MODE_ANALYTIC_LAZY)
    r_eager = eager.go(sb, guess, psf=psf)
    assert np.all(eager.states()["mode"] == _lib.LM_MODE_ANALYTIC)
    _same_fit(r_lazy, r_eager)
    assert np.all(r_lazy["flags"][np.arange(n) % 5 != 0] == 0)

    # two bands x two epochs per object
    nobj = n // 4
    sobj = np.repeat(np.arange(nobj), 4).astype(np.int32)
    sband = np.tile([0, 0, 1, 1], nobj).astype(np.int32)
    g2 = np.concatenate([guess[::4][:, :5], guess[::4][:, 5:6], guess[2::4][:, 5:6]], axis=1)
    _same_fit(lazy.go(sb, g2, psf=psf, stamp_obj=sobj, stamp_band=sband),
              eager.go(sb, g2, psf=psf, stamp_obj=sobj, stamp_band=sband))

    # the kernel prior (rows added by ngmix_lm_prior_sums_batch every round)
    prior = pb.PriorSimpleSepBatch(pb.GaussianCen(0.0, 0.0, 0.3, 0.3), pb.GPriorBA(0.3),
                                   pb.TwoSidedErf(-1.0, 0.1, 1.0e3, 1.0),
                                   [pb.TwoSidedErf(-1.0e3, 1.0, 1.0e5, 10.0)])
    lp = LMBatchFitter(model, prior=prior)
    ep = LMBatchFitter(model, prior=prior)
    ep.lazy_jacobian = False
    keys = ("flags", "nfev", "njev", "ier", "pars", "pars_cov", "lnprob", "chi2per")
    _same_fit(lp.go(sb, guess, psf=psf), ep.go(sb, guess, psf=psf), keys)
    assert lp.prior_path == "kernel"


def test_host_free_rounds_equal_the_host_driven_loop():
    """ngmix_lm_rounds_batch (rounds queued blind by one call, results queued
    behind them, the count checked afterwards) against the loop that reads the
    count of running fits every round: the same fit, and the launch census
    shows the rounds went through the one-call path"""
    rng = np.random.RandomState(77)
    pars, guess, images, weights, jac, sb, psf = _make_objects(300, "exp", rng)
    guess[::3] = pars[::3] * rng.uniform(0.4, 2.0, size=pars[::3].shape)  # a long tail
    fast = LMBatchFitter("exp")
    slow = LMBatchFitter("exp")
    slow.host_loop = True
    a = fast.go(sb, guess, psf=psf)
    b = slow.go(sb, guess, psf=psf)
    _same_fit(a, b)
    assert fast.rounds == slow.rounds
    assert fast.rounds_launched >= fast.rounds
    # the second call sizes its blind chunk by the first (one spare round)
    _lib.launch_census(reset=True)
    c = fast.go(sb, guess, psf=psf)
    seen = _lib.launch_census(reset=True)
    _same_fit(a, c)
    assert seen.get("lm_eval_kernel<true, true>") == fast.rounds_launched
    from ngmix_amd.lm_batch import ROUNDS_HINT_CAP
    assert fast.rounds_launched == min(fast.rounds + 1, ROUNDS_HINT_CAP) or \
        fast.rounds >= ROUNDS_HINT_CAP
    # timing events through the C ABI
    fast.time_kernels = True
    d = fast.go(sb, guess, psf=psf)
    _same_fit(a, d)
    assert set(fast.kernel_ms) == {"lm_eval", "lm_advance", "lm_init", "lm_finalize", "lm_pack"}
    assert all(v > 0.0 for v in fast.kernel_ms.values())
    assert len(fast.eval_launches) == fast.rounds_launched


def test_blind_rounds_too_few_and_a_consumer_on_another_stream():
    """the miss path of _collect(): a batch whose blind chunk was too short gets
    more rounds and its results re-made ON THE STREAM IT WAS QUEUED ON, waited
    for through that chunk's own event -- also when the consumer of go_stream()
    iterates under a different torch stream than the one active at enqueue.
    Every batch is the fit go() returns, bit for bit."""
    import torch
    rng = np.random.RandomState(78)
    items = []
    for n in (200, 130, 257):
        pars, guess, images, weights, jac, sb, psf = _make_objects(n, "exp", rng)
        guess[::3] = pars[::3] * rng.uniform(0.4, 2.0, size=pars[::3].shape)  # a long tail
        items.append((sb, guess, {"psf": psf}))
    fitter = LMBatchFitter("exp")
    ref = [fitter.go(sb, g, **kw) for sb, g, kw in items]
    needed = fitter.rounds
    assert needed > 2

    class Short(LMBatchFitter):
        # every batch is queued with ONE blind round: every collect misses
        def _enqueue(self, *a, **k):
            self._rounds_hint = 1
            return LMBatchFitter._enqueue(self, *a, **k)
    short = Short("exp")
    torch.cuda.synchronize()
    other = torch.cuda.Stream()
    got = []
    it = short.go_stream(items)
    # the generator body (enqueue AND collect) runs inside next(): the first
    # batches are queued under `other`, later collects under the default
    # stream and vice versa
    with torch.cuda.stream(other):
        got.append(next(it))
    got.append(next(it))
    with torch.cuda.stream(other):
        got.append(next(it))
    assert len(got) == len(ref)
    for a, b in zip(ref, got):
        _same_fit(a, b)
    assert short.rounds == needed
    assert short.rounds_launched > needed        # 1 + 2 + 4 + ... rounds
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_small_batch_arena_equals_separate_buffers(monkeypatch):
    """a small batch's buffers out of one allocation and one download
    (lm_batch.SMALL_BATCH) against the separate-buffer path of the large
    batches: every key of the result to the bit, several stamps per object"""
    import torch
    import bench
    from ngmix_amd.batch import GMixBatch
    from ngmix_amd.lm_batch import LMBatchFitter, SMALL_BATCH
    nobj, nband = 5, 2
    ns = nobj * nband
    assert nobj <= SMALL_BATCH
    sb, _, pars = bench.make_workload(ns, 1000, "cuda")
    rng = np.random.RandomState(3)
    guess = np.concatenate([pars[::nband, :5], pars[:, 5].reshape(nobj, nband)], axis=1)
    guess = guess * rng.uniform(0.97, 1.03, size=guess.shape)
    psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (ns, 1)), "gauss")
    sobj = np.repeat(np.arange(nobj), nband)
    sband = np.tile(np.arange(nband), nobj)

    def run():
        f = LMBatchFitter("exp")
        r = f.go(sb, guess, psf=psf, stamp_obj=sobj, stamp_band=sband)
        return {k: np.array(r[k]) for k in r.keys() if k != "model"}
    a = run()
    monkeypatch.setenv("NGMIX_LM_NO_ARENA", "1")
    b = run()
    assert set(a) == set(b)
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    assert np.all(a["flags"] == 0)


def test_c3_shaped_batch_against_the_reference_fit_by_fit(golden):
    """forty independent 48x48 'exp' (x) gaussian-psf objects (config 3's shape
    and guess; off-grid centres, some sheared jacobians, s/n 85-400) fitted by
    the REFERENCE's Fitter (MINPACK lmder, DEFAULT_LM_PARS: oracle/gen_golden_c3.py)
    against ONE lock-step batch of the driver: the same nfev and ier for every
    object, pars / pars_cov / statistics to the tolerances the single golden
    object is held to (test_gpu_api.py) -- the direct link the round-3 review
    asked for, not through this package's own per-object Fitter"""
    g = golden("lm_c3")
    n = g["images"].shape[0]
    weights = np.broadcast_to((1.0 / g["sigma"] ** 2)[:, None, None], g["images"].shape).copy()
    sb = StampBatch.from_images(g["images"], weights, g["jac"])
    psf = GMixBatch.from_numpy(np.tile(
        ngmix.GMix(pars=g["psf_pars"])._data[None, :], (n, 1)).reshape(n, -1))
    for lazy in (True, False):
        f = LMBatchFitter("exp")
        f.lazy_jacobian = lazy
        res = f.go(sb, g["guess"], psf=psf)
        assert np.all(res["flags"] == g["flags"]) and np.all(g["flags"] == 0)
        assert np.array_equal(res["nfev"], g["nfev"])
        assert np.array_equal(res["ier"], g["ier"])
        np.testing.assert_allclose(res["pars"], g["pars"], rtol=1e-6, atol=1e-8)
        sig = np.sqrt(np.einsum("ijj->ij", g["pars_cov"]))
        tol = 1e-4 * np.abs(g["pars_cov"]) + 1e-7 * sig[:, :, None] * sig[:, None, :]
        assert np.all(np.abs(res["pars_cov"] - g["pars_cov"]) <= tol)
        np.testing.assert_allclose(res["pars_err"], g["pars_err"], rtol=1e-4)
        for k in ("lnprob", "chi2per", "s2n"):
            np.testing.assert_allclose(res[k], g[k], rtol=1e-5)
        assert np.array_equal(res["dof"], g["dof"]) and np.array_equal(res["npix"], g["npix"])
