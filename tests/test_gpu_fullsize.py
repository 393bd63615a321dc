"""
Parity at BASELINE.json's FULL sizes, on the very workloads bench.py times
(its own builders are imported): the oracle cannot cover 100,000 stamps in
seconds, so each configuration is checked through size-independent properties
of the domain plus an oracle comparison of a seeded sample drawn from the full
batch.

  C2  100,000 48x48 stamps x 6 gaussians: exact skipping (skip == no-skip
      bitwise), order independence (a permuted batch gives the permuted results
      bitwise), loglike == -0.5 sum(fdiff^2), the render -> loglike round trip
      (a stamp holding its own model has loglike == 0 exactly and
      s2n_numer == s2n_denom), exact linearity of the render, the checksum of
      the per-stamp sums, a sample against the oracle to 1e-10
  C2+ 250,000 stamps: pixel planes beyond 2^32 bytes (64-bit offsets);
      960,000 stamps: pixel indices beyond 2^31, 17.7 GB planes
  C3  100,000 LM fits: all converge, pulls, order independence, a sample
      against the per-object Fitter (scipy MINPACK driving the exact kernels)
  C3+ 30,000 objects x 6 bands (11 parameters, the team form of the lmder
      step): convergence, pulls, order independence, the one-thread form, a
      sample against the per-object Fitter on MultiBandObsLists
  C4  125,000 32x32 stamps (the per-GPU share of 1M / 8): admom and em_run,
      order independence, a sample against the oracle
  C5  20,000 objects x 10 epochs of 64x64 x 16 gaussians: per-object sums ==
      the sums of the per-epoch values, a sample against the oracle
"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-10   # BASELINE.json north_star: 1e-10 relative on float64 loglike / fdiff


@pytest.fixture(scope="module")
def bench():
    import bench as b
    return b


def _oracle_loglike(gmrow, val, ierr, jac):
    from oracle import oracle as ora
    gm = np.zeros(gmrow.size, dtype=ora.GAUSS2D_DTYPE)
    for name in ora.GAUSS2D_DTYPE.names:
        gm[name] = gmrow[name]
    j = np.zeros(1, dtype=ora.JACOBIAN_DTYPE)
    j[0] = tuple(jac)
    pix = np.zeros(val.size, dtype=ora.PIXEL_DTYPE)
    ora.fill_pixels(pix, val, np.ones_like(val), j, True)
    pix["ierr"] = ierr.reshape(-1)   # the batch's own ierr, not sqrt(ierr^2)
    st, res = ora.get_loglike(gm, pix)
    assert st == 0
    fd = np.zeros(pix.size)
    ora.fill_fdiff(gm, pix, fd, 0)
    return res, fd


def _census(run):
    """the batch kernel variants `run()` dispatched (ngmix_launch_census)"""
    from ngmix_amd import _lib
    _lib.launch_census(reset=True)
    run()
    return _lib.launch_census(reset=True)


def _c2_checks(bench, n, nsample):
    import torch
    sb, gm, pars = bench.make_workload(n, 11, "cuda")
    npix = 48 * 48
    # which kernels serve config 2: the fused one-wave-per-stamp kernels with
    # the hand-counted look-ahead (7-wave loglike build), not the 256-thread
    # reference-order ones
    seen = _census(lambda: (sb.loglike(gm), sb.fill_fdiff(gm), sb.render(gm),
                            sb.model_s2n_sum(gm)))
    assert seen.get("pixpass_wave_kernel7<loglike>", 0) == 1
    assert seen.get("pixpass_wave_kernel<fdiff>", 0) == 1
    assert seen.get("pixpass_wave_kernel<render>", 0) == 1
    assert seen.get("pixpass_wave_kernel<s2n>", 0) == 1
    assert not any(k.startswith("pixpass_grid_kernel") for k in seen), seen
    out, status = sb.loglike(gm)
    assert int(status.abs().sum()) == 0
    assert torch.all(out[:, 3] == npix)
    assert bool(torch.isfinite(out).all())

    # exact skipping, on every stamp
    out_ns, _ = sb.loglike(gm, no_skip=True)
    assert torch.equal(out, out_ns)
    fd, _ = sb.fill_fdiff(gm)
    fd_ns, _ = sb.fill_fdiff(gm, no_skip=True)
    assert torch.equal(fd, fd_ns)
    del fd_ns

    # loglike == -0.5 sum(fdiff^2) (gmix_nb.py:866,900), on every stamp
    chk = -0.5 * (fd.reshape(n, npix) ** 2).sum(dim=1)
    np.testing.assert_allclose(out[:, 0].cpu().numpy(), chk.cpu().numpy(), rtol=1e-11, atol=0)

    # the checksum of the per-stamp sums: device sum against an exact host sum
    host = out.cpu().numpy()
    for k in range(3):
        np.testing.assert_allclose(float(out[:, k].sum()), math.fsum(host[:, k]), rtol=1e-12)

    # order independence: stamp i's result does not depend on where it sits
    perm = np.random.RandomState(5).permutation(n)
    out_p, st_p = sb.select(perm).loglike(gm.select(perm))
    assert int(st_p.abs().sum()) == 0
    assert torch.equal(out_p, out[torch.from_numpy(perm).cuda()])
    del out_p

    # render -> loglike round trip: a stamp holding exactly its own model
    model, _ = sb.render(gm)
    from ngmix_amd.batch import StampBatch
    own = StampBatch(model, sb.ierr, sb.jac, sb.nrow, sb.ncol, sb.pix_off, True)
    rt, _ = own.loglike(gm)
    # the fused kernels place a pixel relative to its tile (4x16 tiles in the
    # render, 8x8 in loglike): the two models agree to rounding, 2e-13 of the
    # stamp's peak per pixel ...
    peak = model.reshape(n, npix).abs().amax(dim=1)
    bound = 0.5 * npix * (2e-13 * peak * sb.ierr.reshape(n, npix)[:, 0]) ** 2
    assert bool((rt[:, 0].abs() <= bound).all())
    np.testing.assert_allclose(rt[:, 1].cpu().numpy(), rt[:, 2].cpu().numpy(), rtol=1e-13)
    # ... and exactly in the reference-order kernels: loglike == 0 to the bit
    model_x, _ = sb.render(gm, exact=True)
    own = StampBatch(model_x, sb.ierr, sb.jac, sb.nrow, sb.ncol, sb.pix_off, True)
    rt, _ = own.loglike(gm, exact=True)
    assert float(rt[:, 0].abs().max()) == 0.0
    assert torch.equal(rt[:, 1], rt[:, 2])
    np.testing.assert_allclose(model.cpu().numpy(), model_x.cpu().numpy(), rtol=0,
                               atol=2e-13 * float(peak.max()))
    del model_x
    # ... and the fresh render equals the accumulate-into-zeros render
    acc = torch.zeros_like(model)
    sb.render(gm, image=acc)
    assert torch.equal(acc, model)
    # exact linearity: doubling p (exact in binary) doubles every pixel
    gm2 = gm.clone()
    gm2.data[:, 0] *= 2.0
    gm2.data[:, 7] = 0.0
    im2, _ = sb.render(gm2)
    assert torch.equal(im2, 2.0 * model)
    del im2, acc, own

    # a seeded sample of the full batch against the oracle
    idx = np.random.RandomState(6).choice(n, size=nsample, replace=False)
    idx = np.concatenate([[0, n - 1], idx])
    d_idx = torch.from_numpy(idx).cuda()
    gmh = gm.select(idx).to_numpy()
    val = sb.val.reshape(n, 48, 48)[d_idx].cpu().numpy()
    ierr = sb.ierr.reshape(n, 48, 48)[d_idx].cpu().numpy()
    jac = sb.jac[d_idx].cpu().numpy()
    fdh = fd.reshape(n, npix)[d_idx].cpu().numpy()
    for k, i in enumerate(idx):
        res, rfd = _oracle_loglike(gmh[k], val[k], ierr[k], jac[k])
        np.testing.assert_allclose(host[i, :3], res[:3], rtol=RTOL, atol=0)
        assert host[i, 3] == res[3]
        big = np.abs(rfd) > 1e-3 * np.abs(rfd).max()
        np.testing.assert_allclose(fdh[k][big], rfd[big], rtol=RTOL, atol=0)
        np.testing.assert_allclose(fdh[k], rfd, rtol=0, atol=2e-13 * np.abs(rfd).max() + 1e-300)
    return out


def test_c2_full_size(bench):
    """BASELINE configs[1]: 100,000 stamps of 48x48 x 6 gaussians"""
    _c2_checks(bench, 100000, 24)


def test_c2_planes_beyond_four_gigabytes(bench):
    """250,000 stamps: val / ierr / image planes of 4.6 GB each, pixel offsets
    and byte offsets beyond 2^32"""
    n = 250000
    assert n * 2304 * 8 > 2 ** 32
    _c2_checks(bench, n, 6)


def test_c2_pixel_indices_beyond_2_31(bench):
    """960,000 48x48 stamps resident at once: 2.2e9 pixels per plane (pixel
    INDICES beyond 2^31, planes of 17.7 GB -- the sizes a 288 GB device is
    for), built on the device by tiling a 20,000-stamp batch.  Every block of
    the big batch gives the small batch's results bit for bit: loglike,
    fill_fdiff, both renders, model_s2n_sum; a ragged tail and a permuted
    selection address across the 2^31 boundary too."""
    import torch
    from ngmix_amd.batch import StampBatch, GMixBatch
    base, reps = 20000, 48
    n = base * reps
    npix = 48 * 48
    assert n * npix > 2 ** 31
    free, _ = torch.cuda.mem_get_info()
    if free < 120 * 2 ** 30:
        pytest.skip("needs 120 GB of free device memory")
    sb0, gm0, _ = bench.make_workload(base, 7, "cuda")
    sb = StampBatch(sb0.val.repeat(reps), sb0.ierr.repeat(reps), sb0.jac.repeat(reps, 1),
                    np.full(n, 48), np.full(n, 48), np.arange(n, dtype=np.int64) * npix, True)
    gm = GMixBatch(gm0.data.repeat(reps, 1), n, gm0.ngauss)
    ref, st0 = sb0.loglike(gm0)
    assert int(st0.abs().sum()) == 0
    out, st = sb.loglike(gm)
    assert int(st.abs().sum()) == 0
    assert torch.equal(out.reshape(reps, base, 4), ref[None].expand(reps, base, 4))
    s2n0, _ = sb0.model_s2n_sum(gm0)
    s2n, _ = sb.model_s2n_sum(gm)
    assert torch.equal(s2n.reshape(reps, base), s2n0.reshape(1, base).expand(reps, base))
    del out, s2n
    fd0, _ = sb0.fill_fdiff(gm0)
    fd, _ = sb.fill_fdiff(gm)
    for r in (0, reps // 2, reps - 2, reps - 1):        # (the last two lie beyond 2^31)
        assert torch.equal(fd[r * base * npix:(r + 1) * base * npix], fd0), r
    del fd
    im0, _ = sb0.render(gm0)
    im, _ = sb.render(gm)
    for r in (0, reps // 2, reps - 2, reps - 1):
        assert torch.equal(im[r * base * npix:(r + 1) * base * npix], im0), r
    # accumulate-into form on the same planes: twice the model, exactly
    sb.render(gm, image=im)
    assert torch.equal(im[-base * npix:], 2.0 * im0)
    del im
    # a selection reaching across the boundary, in a scrambled order
    pick = np.random.RandomState(3).choice(n, size=5000, replace=False)
    sel = sb.select(pick)
    o2, _ = sel.loglike(gm.select(pick))
    assert torch.equal(o2, ref[torch.from_numpy(pick % base).cuda()])


def test_lm_and_moments_with_pixel_indices_beyond_2_31(bench):
    """the iterative kernels on planes beyond 2^31 pixels: 960,000 lock-step LM
    fits (48x48) and 2,100,000 adaptive-moment / EM fits (32x32) in one batch
    each, tiled from small batches -- every block reproduces the small batch's
    results to the bit (the fits are independent of where a stamp sits)"""
    import torch
    from ngmix_amd.batch import StampBatch, GMixBatch
    from ngmix_amd.lm_batch import LMBatchFitter
    free, _ = torch.cuda.mem_get_info()
    if free < 120 * 2 ** 30:
        pytest.skip("needs 120 GB of free device memory")
    # ---- config 3 shaped: LM
    base, reps = 20000, 48
    n, npix = base * reps, 48 * 48
    sb0, _, pars = bench.make_workload(base, 7, "cuda")
    rng = np.random.RandomState(17)
    guess0 = pars * rng.uniform(0.9, 1.1, size=pars.shape)
    guess0[:, 0:2] = pars[:, 0:2] + rng.uniform(-0.05, 0.05, size=(base, 2))
    guess0[:, 2:4] = pars[:, 2:4] + rng.uniform(-0.03, 0.03, size=(base, 2))
    psf0, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (base, 1)), "gauss")
    ref = LMBatchFitter("exp").go(sb0, guess0, psf=psf0)
    sb = StampBatch(sb0.val.repeat(reps), sb0.ierr.repeat(reps), sb0.jac.repeat(reps, 1),
                    np.full(n, 48), np.full(n, 48), np.arange(n, dtype=np.int64) * npix, True)
    assert n * npix > 2 ** 31
    psf = GMixBatch(psf0.data.repeat(reps, 1), n, 1)
    res = LMBatchFitter("exp").go(sb, np.tile(guess0, (reps, 1)), psf=psf)
    assert np.all(res["flags"] == np.tile(ref["flags"], reps))
    for key in ("nfev", "ier", "pars", "pars_err", "lnprob", "s2n"):
        got = res[key].reshape((reps, base) + res[key].shape[1:])
        for r in (0, reps // 2, reps - 1):
            np.testing.assert_array_equal(got[r], ref[key], err_msg="%s block %d" % (key, r))
    del sb, psf, res
    torch.cuda.empty_cache()
    # ---- config 4 shaped: adaptive moments and EM
    base, reps = 30000, 70
    n = base * reps
    w = bench.make_c4(base, 5, "cuda")
    dim = w["dim"]
    npix = dim * dim
    assert n * npix > 2 ** 31
    a0 = w["sb"].admom(w["wt0"].clone())
    conv0, _ = w["gm0"].convolve(w["psf"])
    e0 = w["sb_em"].em(w["gm0"].clone(), w["psf"], conv=conv0, sky=w["sky"])

    def tile(s):
        return StampBatch(s.val.repeat(reps), s.ierr.repeat(reps), s.jac.repeat(reps, 1),
                          np.full(n, dim), np.full(n, dim), np.arange(n, dtype=np.int64) * npix,
                          True)
    big = tile(w["sb"])
    wt = GMixBatch(w["wt0"].data.repeat(reps, 1), n, w["wt0"].ngauss)
    a = big.admom(wt)
    for x, x0 in zip(a, a0):
        if torch.is_tensor(x):
            xr = x.reshape((reps, base) + tuple(x.shape[1:]))
            assert torch.equal(xr[0], x0) and torch.equal(xr[reps - 1], x0)
    del big, a
    torch.cuda.empty_cache()
    big = tile(w["sb_em"])
    gm = GMixBatch(w["gm0"].data.repeat(reps, 1), n, w["gm0"].ngauss)
    psf = GMixBatch(w["psf"].data.repeat(reps, 1), n, w["psf"].ngauss)
    conv, _ = gm.convolve(psf)
    sky = w["sky"].repeat(reps) if torch.is_tensor(w["sky"]) else w["sky"]
    e = big.em(gm, psf, conv=conv, sky=sky)
    for x, x0 in zip(e, e0):
        if torch.is_tensor(x):
            xr = x.reshape((reps, base) + tuple(x.shape[1:]))
            assert torch.equal(xr[0], x0) and torch.equal(xr[reps - 1], x0)
    # (the fitted mixtures, NaN-filled derived fields included: compared as bits)
    bits = gm.data.reshape(reps, -1).view(torch.int64)
    assert torch.equal(bits[reps - 1], bits[0]) and torch.equal(bits[reps // 2], bits[0])


def test_c3_full_size(bench):
    """BASELINE configs[2]: 100,000 psf-convolved 'exp' fits in lock step"""
    import ngmix_amd as ngmix
    from ngmix_amd.batch import GMixBatch
    from ngmix_amd.lm_batch import LMBatchFitter
    n = 100000
    sb, _, pars = bench.make_workload(n, 11, "cuda")
    rng = np.random.RandomState(77)
    guess = pars * rng.uniform(0.9, 1.1, size=pars.shape)
    guess[:, 0:2] = pars[:, 0:2] + rng.uniform(-0.05, 0.05, size=(n, 2))
    guess[:, 2:4] = pars[:, 2:4] + rng.uniform(-0.03, 0.03, size=(n, 2))
    psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (n, 1)), "gauss")
    box = {}
    seen = _census(lambda: box.update(res=LMBatchFitter("exp").go(sb, guess, psf=psf)))
    res = box["res"]
    # config 3 runs the raw-basis pixel pass with LDS tile records and the
    # six-parameter register form of the lmder step, one launch each per round
    # (+ the round in flight when the count reaches zero); no stats pixel pass
    rounds = seen.get("lm_eval_kernel<true, true>", 0)
    assert 4 <= rounds <= 8 and seen.get("lm_advance_kernel<6, true>", 0) == rounds, seen
    assert not any(k.startswith(("pixpass", "lm_eval_fd", "lm_advance_kernel<14"))
                   for k in seen), seen
    assert np.all(res["flags"] == 0)
    assert np.all((res["ier"] >= 1) & (res["ier"] <= 4))
    assert np.all(res["npix"] == 2304) and np.all(res["dof"] == 2304 - 6)
    # the truth is recovered within the errors, with unit-width pulls
    pull = (res["pars"] - pars) / res["pars_err"]
    assert np.all(np.abs(pull) < 7.0)
    assert np.all(np.abs(pull.std(axis=0) - 1.0) < 0.05)
    assert np.all(np.abs(pull.mean(axis=0)) < 0.05)
    assert abs(res["chi2per"].mean() - 1.0) < 0.01

    # the same batch twice through the software pipeline (go_stream: batch 2's
    # first rounds run under batch 1's finalise / download): both results are
    # go()'s, to the bit
    fitter = LMBatchFitter("exp")
    piped = list(fitter.go_stream([(sb, guess, {"psf": psf})] * 2))
    assert len(piped) == 2
    for r in piped:
        for key in ("nfev", "flags", "pars", "pars_err", "lnprob", "s2n"):
            assert np.array_equal(r[key], res[key], equal_nan=True), key
    assert np.array_equal(piped[1]["pars_cov"], res["pars_cov"])
    del piped

    # order independence of the lock-step driver: a permuted subset
    sub = np.random.RandomState(8).choice(n, size=5000, replace=False)
    r2 = LMBatchFitter("exp").go(sb.select(sub), guess[sub], psf=psf.select(sub))
    assert np.array_equal(r2["nfev"], res["nfev"][sub])
    assert np.array_equal(r2["pars"], res["pars"][sub])
    assert np.array_equal(r2["pars_cov"], res["pars_cov"][sub])
    # the default mode leaves the jacobian out of the trials predicted to end a
    # fit (here: the fourth pixel pass of nearly every fit); with the jacobian
    # at every evaluation the fits are the same, to the bit
    eager = LMBatchFitter("exp")
    eager.lazy_jacobian = False
    r3 = eager.go(sb.select(sub), guess[sub], psf=psf.select(sub))
    for key in ("nfev", "njev", "ier", "flags", "pars", "pars_cov", "lnprob", "s2n"):
        assert np.array_equal(r3[key], r2[key], equal_nan=True), key

    # a sample against the per-object Fitter: scipy's MINPACK calling the exact
    # (reference-order) fdiff / jacobian kernels once per function evaluation
    import torch
    idx = np.concatenate([[0, n - 1], sub[:10]])
    d_idx = torch.from_numpy(idx).cuda()
    val = sb.val.reshape(n, 48, 48)[d_idx].cpu().numpy()
    ierr = sb.ierr.reshape(n, 48, 48)[d_idx].cpu().numpy()
    jac = sb.jac[0].cpu().numpy()
    jobj = ngmix.Jacobian(row=jac[0], col=jac[1], dvdrow=jac[2], dvdcol=jac[3],
                          dudrow=jac[4], dudcol=jac[5])
    pgm = ngmix.GMixModel([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], "gauss")
    same_nfev = 0
    for k, i in enumerate(idx):
        pobs = ngmix.Observation(np.zeros((5, 5)), jacobian=jobj, gmix=pgm)
        obs = ngmix.Observation(val[k], weight=ierr[k] ** 2, jacobian=jobj, psf=pobs)
        one = ngmix.fitting.Fitter(model="exp").go(obs=obs, guess=guess[i])
        assert one["flags"] == 0 and one["ier"] == res["ier"][i]
        same_nfev += int(one["nfev"] == res["nfev"][i])
        assert np.all(np.abs(res["pars"][i] - one["pars"]) <= 1e-4 * one["pars_err"])
        np.testing.assert_allclose(res["pars_err"][i], one["pars_err"], rtol=1e-3)
        np.testing.assert_allclose(res["lnprob"][i], one["lnprob"], rtol=1e-6)
    assert same_nfev >= len(idx) - 1


def test_multiband_fits_full_size(bench):
    """joint fits over bands at scale -- 30,000 objects x 6 bands of 48x48
    stamps (11 parameters: the lmder step runs as the team form, lm_team.hip):
    every fit converges, the truth is recovered with unit-width pulls in all
    eleven parameters, a permuted subset reproduces the fits to the bit, the
    one-thread form of the step gives the same fits to the bit, and a sample
    equals the per-object Fitter on the MultiBandObsList (scipy's MINPACK over
    the exact kernels)"""
    import torch
    import ngmix_amd as ngmix
    from ngmix_amd.batch import StampBatch, GMixBatch
    from ngmix_amd.lm_batch import LMBatchFitter
    nobj, nband, dim = 30000, 6, 48
    ns = nobj * nband
    rng = np.random.RandomState(61)
    S = bench.SCALE
    truth = np.zeros((nobj, 5 + nband))
    truth[:, 0:2] = rng.uniform(-0.5, 0.5, size=(nobj, 2)) * S
    truth[:, 2:4] = np.clip(rng.normal(scale=0.1, size=(nobj, 2)), -0.5, 0.5)
    truth[:, 4] = rng.uniform(0.3, 1.5, size=nobj)
    truth[:, 5:] = rng.uniform(50.0, 500.0, size=(nobj, 1)) * rng.uniform(0.5, 1.5,
                                                                         size=(nobj, nband))
    sobj = np.repeat(np.arange(nobj), nband).astype(np.int32)
    sband = np.tile(np.arange(nband), nobj).astype(np.int32)
    spars = np.concatenate([truth[sobj, :5], truth[sobj, 5 + sband][:, None]], axis=1)
    gm0, _ = GMixBatch.from_pars(spars, "exp")
    psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (ns, 1)), "gauss")
    gm, _ = gm0.convolve(psf)
    cen = (dim - 1) / 2.0
    jac = np.array([cen, cen, S, 0.0, 0.0, S, S * S, S])
    d_jac = torch.from_numpy(np.tile(jac, (ns, 1))).cuda()
    shape = np.full(ns, dim)
    off = np.arange(ns, dtype=np.int64) * dim * dim
    model, _ = StampBatch(None, None, d_jac, shape, shape, off, True).render(gm, fast_exp=True)
    sigma = 0.5
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    val = model + sigma * torch.randn(model.shape, generator=gen, device="cuda",
                                      dtype=torch.float64)
    sb = StampBatch(val, torch.full_like(val, 1.0 / sigma), d_jac, shape, shape, off, True)
    guess = truth * rng.uniform(0.9, 1.1, size=truth.shape)
    guess[:, 0:2] = truth[:, 0:2] + rng.uniform(-0.05, 0.05, size=(nobj, 2))
    guess[:, 2:4] = truth[:, 2:4] + rng.uniform(-0.03, 0.03, size=(nobj, 2))
    kw = dict(psf=psf, stamp_obj=sobj, stamp_band=sband)
    box = {}
    seen = _census(lambda: box.update(res=LMBatchFitter("exp").go(sb, guess, **kw)))
    res = box["res"]
    rounds = seen.get("lm_eval_kernel<true, true>", 0)
    assert 4 <= rounds <= 12 and seen.get("lm_advance_team_kernel<4, 12>", 0) == rounds, seen
    assert not any(k.startswith("lm_advance_kernel<") for k in seen), seen
    assert np.all(res["flags"] == 0)
    assert np.all(res["npix"] == nband * dim * dim)
    pull = (res["pars"] - truth) / res["pars_err"]
    assert np.all(np.abs(pull) < 7.0)
    assert np.all(np.abs(pull.std(axis=0) - 1.0) < 0.06), pull.std(axis=0)
    assert np.all(np.abs(pull.mean(axis=0)) < 0.06), pull.mean(axis=0)
    assert abs(res["chi2per"].mean() - 1.0) < 0.01
    # order independence, and the one-thread form of the step
    sub = np.random.RandomState(8).choice(nobj, size=3000, replace=False)
    sidx = (sub[:, None] * nband + np.arange(nband)[None, :]).reshape(-1)
    so = np.repeat(np.arange(sub.size), nband).astype(np.int32)
    sbd = np.tile(np.arange(nband), sub.size).astype(np.int32)
    r2 = LMBatchFitter("exp").go(sb.select(sidx), guess[sub], psf=psf.select(sidx),
                                 stamp_obj=so, stamp_band=sbd)
    generic = LMBatchFitter("exp")
    generic.advance_hint = False
    r3 = generic.go(sb.select(sidx), guess[sub], psf=psf.select(sidx), stamp_obj=so,
                    stamp_band=sbd)
    for key in ("nfev", "njev", "ier", "flags", "pars", "pars_cov", "lnprob", "s2n"):
        assert np.array_equal(r2[key], res[key][sub], equal_nan=True), key
        assert np.array_equal(r3[key], r2[key], equal_nan=True), key
    # a sample against the per-object Fitter on the MultiBandObsList
    jobj = ngmix.Jacobian(row=jac[0], col=jac[1], dvdrow=jac[2], dvdcol=jac[3],
                          dudrow=jac[4], dudcol=jac[5])
    pgm = ngmix.GMixModel([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], "gauss")
    h_val = sb.val.reshape(ns, dim, dim)
    same_nfev = 0
    sample = [0, nobj - 1] + list(sub[:4])
    for i in sample:
        mb = ngmix.MultiBandObsList()
        for b in range(nband):
            ol = ngmix.ObsList()
            pobs = ngmix.Observation(np.zeros((5, 5)), jacobian=jobj, gmix=pgm)
            ol.append(ngmix.Observation(h_val[i * nband + b].cpu().numpy(),
                                        weight=np.full((dim, dim), 1.0 / sigma ** 2),
                                        jacobian=jobj, psf=pobs))
            mb.append(ol)
        one = ngmix.fitting.Fitter(model="exp", batched=False).go(obs=mb, guess=guess[i])
        assert one["flags"] == 0 and one["ier"] == res["ier"][i]
        same_nfev += int(one["nfev"] == res["nfev"][i])
        assert np.all(np.abs(res["pars"][i] - one["pars"]) <= 1e-4 * one["pars_err"])
        np.testing.assert_allclose(res["pars_err"][i], one["pars_err"], rtol=1e-3)
        np.testing.assert_allclose(res["lnprob"][i], one["lnprob"], rtol=1e-6)
    assert same_nfev >= len(sample) - 1


def _conv_rec(row, dtype):
    out = np.zeros(row.size, dtype=dtype)
    for name in dtype.names:
        out[name] = row[name]
    return out


def test_c4_full_size(bench):
    """BASELINE configs[3], one GPU's share (1M / 8): 125,000 32x32 stamps
    through admom and em_run"""
    import torch
    from ngmix_amd import _lib
    from ngmix_amd.batch import records_to_numpy
    from oracle import oracle as ora
    n = 125000
    w = bench.make_c4(n, 5, "cuda")
    sb, sb_em = w["sb"], w["sb_em"]

    wt = w["wt0"].clone()
    box = {}
    seen = _census(lambda: box.update(r=sb.admom(wt)))
    res, st = box["r"]
    # one wave per 32 x 32 stamp, sixteen register slots per lane
    assert seen == {"admom_grid_kernel<64, 16>": 1}, seen
    assert int(st.abs().sum()) == 0
    rec = records_to_numpy(res, _lib.ADMOM_RESULT_DTYPE)
    assert np.all(rec["flags"] == 0) and np.all(rec["npix"] == 1024)
    assert np.all(rec["numiter"] < 200)

    gm = w["gm0"].clone()
    conv, _ = gm.convolve(w["psf"])
    seen = _census(lambda: box.update(e=sb_em.em(gm, w["psf"], conv=conv, sky=w["sky"])))
    out_e, st_e, _ = box["e"]
    # the fused one-wave kernel: full run, one object gaussian, one psf gaussian
    assert seen == {"em_wave_kernel<64, 16, 0, 1, 1>": 1}, seen
    assert int(st_e.abs().sum()) == 0
    numiter = out_e[:, 0].cpu().numpy()
    assert np.all(numiter >= 40) and np.all(numiter < 500)

    # order independence, both kernels, a permuted subset
    sub = np.random.RandomState(9).choice(n, size=20000, replace=False)
    d_sub = torch.from_numpy(sub).cuda()
    wt2 = w["wt0"].select(sub)
    res2, _ = sb.select(sub).admom(wt2)
    assert torch.equal(res2, res[d_sub])
    assert torch.equal(wt2.data[:, :7], wt.select(sub).data[:, :7])
    gm2 = w["gm0"].select(sub)
    psf2 = w["psf"].select(sub)
    conv2, _ = gm2.convolve(psf2)
    out2, _, _ = sb_em.select(sub).em(gm2, psf2, conv=conv2, sky=w["sky"])
    assert torch.equal(out2, out_e[d_sub])
    # (the derived fields of a record are NaN after gauss2d_set: compare p .. det)
    assert torch.equal(gm2.data[:, :7], gm.select(sub).data[:, :7])

    # a sample against the oracle
    idx = np.concatenate([[0, n - 1], sub[:14]])
    d_idx = torch.from_numpy(idx).cuda()
    val = sb.val.reshape(n, 32, 32)[d_idx].cpu().numpy()
    ierr = sb.ierr.reshape(n, 32, 32)[d_idx].cpu().numpy()
    jac = sb.jac[d_idx].cpu().numpy()
    wt_in = w["wt0"].select(idx).to_numpy()
    gm_in = w["gm0"].select(idx).to_numpy()
    psf_in = w["psf"].select(idx).to_numpy()
    gm_out = gm.select(idx).to_numpy()
    conf = np.zeros(1, dtype=ora.ADMOM_CONF_DTYPE)
    conf["maxiter"], conf["shiftmax"], conf["etol"], conf["Ttol"] = 200, 5.0, 1e-5, 1e-3
    econf = np.zeros(1, dtype=ora.EM_CONF_DTYPE)
    econf["tol"], econf["maxiter"], econf["miniter"], econf["sky"] = 1e-5, 500, 40, w["sky"]
    out_h = out_e.cpu().numpy()
    for k, i in enumerate(idx):
        j = np.zeros(1, dtype=ora.JACOBIAN_DTYPE)
        j[0] = tuple(jac[k])
        pix = ora.make_pixels(val[k], ierr[k] ** 2, j, True)
        r = np.zeros(1, dtype=ora.ADMOM_RESULT_DTYPE)
        assert ora.admom(conf, _conv_rec(wt_in[k], ora.GAUSS2D_DTYPE), pix, r) == 0
        assert r["flags"][0] == rec["flags"][i] and r["numiter"][0] == rec["numiter"][i]
        assert r["npix"][0] == rec["npix"][i]
        for name in ("wsum", "sums", "sums_cov", "pars"):
            np.testing.assert_allclose(rec[name][i], r[name][0], rtol=1e-9, atol=1e-12,
                                       err_msg=name)
        pix_em = ora.make_pixels(val[k] + w["sky"], ierr[k] ** 2, j, True)
        g = _conv_rec(gm_in[k], ora.GAUSS2D_DTYPE)
        p = _conv_rec(psf_in[k], ora.GAUSS2D_DTYPE)
        c = np.zeros(1, dtype=ora.GAUSS2D_DTYPE)
        ora.gmix_convolve_fill(c, g, p)
        sums = np.zeros((1, ora.EM_SUMS_NDOUBLE[0]))
        st1, nit, frac, _ = ora.em_run(0, econf, pix_em, sums, g, p, c)
        assert st1 == 0 and nit == int(out_h[i, 0])
        for name in ("p", "row", "col", "irr", "irc", "icc"):
            np.testing.assert_allclose(gm_out[k][name], g[name], rtol=1e-9, atol=1e-12,
                                       err_msg=name)


def test_c5_full_size(bench):
    """BASELINE configs[4], one GPU's share: 20,000 objects x 10 epochs of
    64x64 pixels, 16-gaussian 'bdf' (x) psf"""
    import torch
    nobj, nepoch = 20000, 10
    sb, gm, obj_start = bench.make_c5(nobj, 3, "cuda")
    ns = nobj * nepoch
    box = {}
    seen = _census(lambda: box.update(r=sb.loglike_objects(gm, obj_start)))
    per_obj, per_stamp, status = box["r"]
    # the same fused loglike kernel as config 2, on 64-tile stamps with 16 gaussians
    assert seen.get("pixpass_wave_kernel7<loglike>", 0) == 1, seen
    assert int(status.abs().sum()) == 0
    assert torch.all(per_stamp[:, 3] == 4096) and torch.all(per_obj[:, 3] == 40960)
    # the per-object records are the sums of the epochs' (exact host sums)
    ps = per_stamp.cpu().numpy().reshape(nobj, nepoch, 4)
    po = per_obj.cpu().numpy()
    ref = np.array([[math.fsum(ps[o, :, k]) for k in range(3)] for o in range(0, nobj, 97)])
    np.testing.assert_allclose(po[::97, :3], ref, rtol=1e-13)
    # exact skipping with 16 gaussians (ballot path) on every epoch
    out_ns, _ = sb.loglike(gm, no_skip=True)
    assert torch.equal(out_ns, per_stamp)
    # chi2 per pixel ~ 1 at the truth
    assert abs(float(-2 * per_obj[:, 0].sum()) / (ns * 4096) - 1.0) < 1e-3
    # ragged objects (1..19 epochs each) through the segmented sum
    rng = np.random.RandomState(4)
    lengths = []
    while sum(lengths) < ns:
        lengths.append(int(rng.randint(1, 20)))
    lengths[-1] -= sum(lengths) - ns
    if lengths[-1] == 0:
        lengths.pop()
    ragged = np.concatenate([[0], np.cumsum(lengths)])
    po2 = sb.sum_over_epochs(per_stamp, ragged).cpu().numpy()
    flat = per_stamp.cpu().numpy()
    for o in range(0, len(lengths), 501):
        seg = flat[ragged[o]:ragged[o + 1]]
        np.testing.assert_allclose(po2[o, :3], [math.fsum(seg[:, k]) for k in range(3)],
                                   rtol=1e-13)
    # a sample of epochs against the oracle
    idx = np.concatenate([[0, ns - 1], rng.choice(ns, size=6, replace=False)])
    d_idx = torch.from_numpy(idx).cuda()
    gmh = gm.select(idx).to_numpy()
    val = sb.val.reshape(ns, 64, 64)[d_idx].cpu().numpy()
    ierr = sb.ierr.reshape(ns, 64, 64)[d_idx].cpu().numpy()
    jac = sb.jac[d_idx].cpu().numpy()
    for k, i in enumerate(idx):
        res, _ = _oracle_loglike(gmh[k], val[k], ierr[k], jac[k])
        np.testing.assert_allclose(flat[i, :3], res[:3], rtol=RTOL, atol=0)
        assert flat[i, 3] == res[3]
