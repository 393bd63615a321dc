"""A fuzz campaign of the GPU kernels against the CPU oracle: the bodies of the
suite's own oracle comparisons (tests/test_gpu_pixpass.py::
test_batch_random_vs_oracle -- loglike / fill_fdiff / render, exact and fused,
sheared jacobians, masks, both pixel-list modes --, tests/test_gpu_iter.py::
test_admom_kernel_variants_vs_oracle, ::test_em_kernel_variants_vs_oracle and
::test_em_many_gaussians_vs_oracle -- exact numiter / flags, every EM kind) over RANDOM stamp shapes and mixture sizes instead of
the suite's dozen fixed ones.  Every case derives its own data seed from its
shape, so a failure is reproduced by its printed (shape, ngauss).

usage: python tools/fuzz_vs_oracle.py [seconds] [seed]   (default 120 s)
A log of a run is kept under profiles/."""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import test_gpu_pixpass as tp  # noqa: E402
import test_gpu_iter as ti  # noqa: E402

budget = 120.0
rng = np.random.RandomState(20261003)


def dim():
    """stamp sides: mostly what surveys cut (multiples of 8 / 16), some odd,
    some tiny, a few beyond one wave's register slots"""
    u = rng.uniform()
    if u < 0.45:
        return int(rng.choice([8, 16, 24, 32, 40, 48, 56, 64, 72, 96]))
    if u < 0.85:
        # (>= 3: the suite's comparison wants a listed pixel in every stamp, and
        # with 3 % of the weights zeroed a stamp of one or two pixels may have none)
        return int(rng.randint(3, 100))
    return int(rng.randint(100, 160))


def admom_noisy(seed):
    """adaptive moments at every signal-to-noise, with small maxiter / shiftmax:
    the flag paths (MAXITER, CEN_SHIFT, NONPOS_FLUX, NONPOS_SIZE, LOW_DET) --
    status, flags and numiter equal to the oracle's, the records to 1e-10"""
    from ngmix_amd import _lib
    from ngmix_amd.batch import StampBatch, GMixBatch, records_to_numpy
    from oracle import oracle as ora
    r = np.random.RandomState(seed)
    n, scale = 16, 0.263
    nrow, ncol = int(r.randint(16, 65)), int(r.randint(16, 65))
    pars = np.zeros((n, 6))
    pars[:, 0:2] = r.uniform(-1.5, 1.5, size=(n, 2)) * scale
    pars[:, 2:4] = r.normal(scale=0.15, size=(n, 2)).clip(-0.6, 0.6)
    pars[:, 4] = r.uniform(0.2, 1.5, size=n)
    pars[:, 5] = 10.0 ** r.uniform(-0.5, 2.3, size=n)
    gm, _ = GMixBatch.from_pars(pars, "gauss")
    jac = np.array([(nrow - 1) / 2.0, (ncol - 1) / 2.0, scale, 0, 0, scale, scale ** 2, scale])
    truth, _ = StampBatch.from_images(np.zeros((n, nrow, ncol)), None, jac).render(gm)
    sigma = 10.0 ** r.uniform(-2.0, 0.0)
    images = truth.cpu().numpy().reshape(n, nrow, ncol) + r.normal(scale=sigma,
                                                                     size=(n, nrow, ncol))
    weights = np.full((n, nrow, ncol), 1.0 / sigma ** 2)
    weights[r.uniform(size=weights.shape) < 0.01] = 0.0
    sb = StampBatch.from_images(images, weights, jac)
    guess = np.zeros((n, 6))
    guess[:, 4] = pars[:, 4] * r.uniform(0.5, 2.0, size=n)
    guess[:, 5] = 1.0
    wt, _ = GMixBatch.from_pars(guess, "gauss")
    wt_in = wt.to_numpy()
    maxiter, shiftmax = int(r.choice([5, 20, 200])), float(r.choice([0.5, 5.0]))
    res, status = sb.admom(wt, maxiter=maxiter, shiftmax=shiftmax)
    status = status.cpu().numpy()
    res = records_to_numpy(res, _lib.ADMOM_RESULT_DTYPE)
    wt_out = wt.to_numpy()
    j = np.zeros(1, dtype=ora.JACOBIAN_DTYPE)
    j[0] = tuple(jac)
    conf = np.zeros(1, dtype=ora.ADMOM_CONF_DTYPE)
    conf["maxiter"], conf["shiftmax"], conf["etol"], conf["Ttol"] = maxiter, shiftmax, 1e-5, 1e-3
    # The iteration amplifies rounding on the stamps it does not converge on --
    # above all the ones that end in LOW_DET, whose weight collapses: the ORACLE's
    # own record moves by 1e-7 ... O(1) (another flag, another numiter) when its
    # input image moves by one ulp.  So: a stamp is compared where the oracle
    # agrees with itself (status, flags, numiter) under four one-ulp
    # perturbations of the image; its flags always, its whole record
    # at 1e-10 when the fit succeeded (flags == 0) and the oracle's sums move by
    # < 1e-13 under those perturbations.
    nflag = nill = 0
    for i in range(n):
        refs = []
        for k in range(5):
            wig = np.random.RandomState((seed + 7919 * k) % (1 << 31)).uniform(
                -1, 1, size=images[i].shape) if k else 0.0
            pix = ora.make_pixels(images[i] * (1.0 + 2.0e-16 * wig), weights[i], j, True)
            w = ti.conv_rec(wt_in[i], ora.GAUSS2D_DTYPE)
            ref = np.zeros(1, dtype=ora.ADMOM_RESULT_DTYPE)
            refs.append((ora.admom(conf, w, pix, ref), ref, w))
        st, ref, w = refs[0]
        keys = {(r[0], int(r[1]["flags"][0]), int(r[1]["numiter"][0])) for r in refs}
        if len(keys) != 1:
            nill += 1
            continue
        diverged = 2 | 4 | 8 | 16
        if st != status[i]:
            # a range error raised by the norms of a collapsed weight (status 1 /
            # 2: det / T too low) is one more exit of a diverging iteration: seen
            # once in 22,000 noisy stamps, against the oracle's LOW_DET two
            # iterations earlier; anything else is a failure
            g_div = int(status[i]) in (1, 2) or (status[i] == 0 and res["flags"][i] & diverged)
            o_div = st in (1, 2) or (st == 0 and ref["flags"][0] & diverged)
            assert g_div and o_div, ("status", i, st, int(status[i]), "oracle flags / numiter",
                                     int(ref["flags"][0]), int(ref["numiter"][0]),
                                     "gpu flags / numiter", int(res["flags"][i]),
                                     int(res["numiter"][i]))
            nill += 1
            continue
        if st != 0:
            continue
        # (CEN_SHIFT | NONPOS_FLUX | NONPOS_SIZE | LOW_DET: the exits of an
        # iteration that diverges; which of them it trips first -- or whether it
        # runs out of iterations first, MAXITER -- hangs on the last bit: seen, six
        # stamps in 160,000 took LOW_DET in the oracle and NONPOS_SIZE, CEN_SHIFT
        # or MAXITER here.  A fit that SUCCEEDS on one side and not on the other
        # would be a failure.)
        gf, rf = int(res["flags"][i]), int(ref["flags"][0])
        assert gf == rf or (gf and rf and ((gf | rf) & diverged)), ("flags", i, gf, rf)
        # (a fit that diverges -- LOW_DET, NONPOS_FLUX / SIZE, CEN_SHIFT -- trips
        # its test an iteration earlier or later with the last bit of its
        # iterates: seen, four stamps in 33,700, all LOW_DET, the oracle steady
        # under the four perturbations above; numiter is held where the fit
        # converged or ran into maxiter)
        if rf in (0, 32) and gf == rf:
            assert res["numiter"][i] == ref["numiter"][0], (
                "numiter", i, "flags", int(ref["flags"][0]), "gpu", int(res["numiter"][i]),
                "oracle", int(ref["numiter"][0]), "maxiter", maxiter, "shiftmax", shiftmax)
        nflag += int(ref["flags"][0] != 0)
        big = max(np.abs(ref["sums"][0]).max(), 1e-300)
        spread = max(np.abs(ref["sums"][0] - r[1]["sums"][0]).max() for r in refs[1:])
        if ref["flags"][0] == 0 and spread < 1e-13 * big:
            ti._check_admom("stamp %d" % i, res[i:i + 1], wt_out[i], ref, w)
        elif ref["flags"][0] == 0:
            nill += 1
    return nflag, nill


def _em_case(seed):
    """the inputs of one em_noisy case, numpy only (a failure can be studied
    against the oracle on the CPU): 8 stamps of one shape, each one or two
    elliptical gaussians (fluxes 0.3-300 x the pixel noise) + noise; 'prepped'
    cases (two in three) shift the image as the reference's prep_image does
    (em.py:95-120: no pixel below 0.001 of the range, sky = the shift), the
    others keep a sky inside the noise -- negative pixels, which em_run's
    val / gtot weights take as they come"""
    r = np.random.RandomState(seed)
    n, scale = 8, 0.263
    nrow, ncol = int(r.randint(16, 65)), int(r.randint(16, 65))
    ngauss, kind = int(r.randint(1, 4)), int(r.choice([0, 0, 1, 2, 3]))
    cen = r.uniform(-1.0, 1.0, size=(n, 2)) * scale
    e = r.normal(scale=0.2, size=(n, 2)).clip(-0.6, 0.6)
    T = r.uniform(0.2, 1.5, size=n)
    flux = 10.0 ** r.uniform(-0.5, 2.5, size=n)
    row0, col0 = (nrow - 1) / 2.0 + r.uniform(-0.5, 0.5), (ncol - 1) / 2.0 + r.uniform(-0.5, 0.5)
    jac = np.array([row0, col0, scale, 0, 0, scale, scale ** 2, scale])
    v = ((np.arange(nrow) - row0) * scale)[None, :, None] - cen[:, 0, None, None]
    u = ((np.arange(ncol) - col0) * scale)[None, None, :] - cen[:, 1, None, None]
    truth = np.zeros((n, nrow, ncol))
    for frac, grow in ((0.6, 1.0), (0.4, 2.5)):
        irr = (0.5 * T * grow * (1 - e[:, 0]))[:, None, None]
        icc = (0.5 * T * grow * (1 + e[:, 0]))[:, None, None]
        irc = (0.5 * T * grow * e[:, 1])[:, None, None]
        det = irr * icc - irc ** 2
        chi2 = (icc * v * v + irr * u * u - 2 * irc * u * v) / det
        truth += frac * flux[:, None, None] * np.exp(-0.5 * chi2) / (2 * np.pi * np.sqrt(det)) \
            * scale ** 2
    sigma = 10.0 ** r.uniform(-2.5, 0.0)
    images = truth + r.normal(scale=sigma, size=truth.shape)
    prepped = r.uniform() < 0.67
    if prepped:
        lo, hi = images.min(), images.max()
        sky = 0.001 * (hi - lo) - lo
    else:
        sky = sigma * 10.0 ** r.uniform(-0.5, 2.0)
    images = images + sky
    weights = np.full(images.shape, 1.0 / sigma ** 2)
    weights[r.uniform(size=weights.shape) < 0.01] = 0.0
    full = np.zeros((n, ngauss, 6))
    for i in range(ngauss):
        full[:, i, 0] = flux * scale ** 2 / ngauss * r.uniform(0.5, 2.0, size=n)
        full[:, i, 1:3] = cen + r.uniform(-0.3, 0.3, size=(n, 2)) * scale
        full[:, i, 3] = 0.5 * T * (0.5 + 1.2 * i) * r.uniform(0.5, 2.0, size=n)
        full[:, i, 5] = 0.5 * T * (0.5 + 1.2 * i) * r.uniform(0.5, 2.0, size=n)
    npsf = int(r.choice([1, 3]))
    if npsf == 1:
        psf = np.array([[1.0, 0.0, 0.0, 0.0, 0.0, 0.0]])
    else:
        psf = np.array([[0.55, 0.0, 0.0, 0.06, 0.002, 0.07], [0.3, 0.0, 0.0, 0.15, -0.004, 0.14],
                        [0.15, 0.0, 0.0, 0.4, 0.01, 0.42]])
    return dict(n=n, nrow=nrow, ncol=ncol, ngauss=ngauss, kind=kind, jac=jac, images=images,
                weights=weights, sky=float(sky), full=full, npsf=npsf, psf=psf, prepped=prepped,
                tol=float(r.choice([1e-3, 1e-5, 1e-6, 1e-7])),
                maxiter=int(r.choice([10, 100, 2000])), miniter=int(r.choice([0, 20, 40])))


def _em_oracle(c, i, wiggle):
    """the oracle's em_run on stamp i of a case, its image moved by `wiggle`
    (0: as is; k: a relative perturbation of 1e-13 drawn from k -- the size of
    a change of summation order over a stamp's pixels)"""
    from oracle import oracle as ora
    j = np.zeros(1, dtype=ora.JACOBIAN_DTYPE)
    j[0] = tuple(c["jac"])
    im = c["images"][i]
    if wiggle:
        im = im * (1.0 + 1.0e-13 * np.random.RandomState(7919 * wiggle).uniform(-1, 1,
                                                                              size=im.shape))
    pix = ora.make_pixels(im, c["weights"][i], j, True)

    def recs(rows):
        g = np.zeros(len(rows), dtype=ora.GAUSS2D_DTYPE)
        for k, f in enumerate(("p", "row", "col", "irr", "irc", "icc")):
            g[f] = np.asarray(rows)[:, k]
        return g
    g, p = recs(c["full"][i]), recs(c["psf"])
    cv = np.zeros(g.size * p.size, dtype=ora.GAUSS2D_DTYPE)
    ora.gmix_convolve_fill(cv, g, p)
    econf = np.zeros(1, dtype=ora.EM_CONF_DTYPE)
    econf["tol"], econf["maxiter"], econf["miniter"], econf["sky"] = (
        c["tol"], c["maxiter"], c["miniter"], c["sky"])
    sums = np.zeros((g.size, ora.EM_SUMS_NDOUBLE[c["kind"]]))
    st, numiter, frac, _ = ora.em_run(c["kind"], econf, pix, sums, g, p, cv)
    return st, numiter, frac, g


EM_FIELDS = ("p", "row", "col", "irr", "irc", "icc")


def _em_same(c, i, status, out, gm_out, ref, rtol, atol):
    """None when the kernel's result on stamp i is the oracle run `ref`'s, else
    what differs"""
    st, numiter, frac, g = ref
    if st != status[i]:
        return ("status", st, int(status[i]), "numiter", numiter, int(out[i, 0]))
    if st != 0:
        return None
    if int(out[i, 0]) != numiter:
        return ("numiter", int(out[i, 0]), numiter, "tol", c["tol"], "frac", frac,
                float(out[i, 1]))
    for f in EM_FIELDS:
        if not np.allclose(gm_out[i][f], g[f], rtol=rtol, atol=atol):
            return (f, gm_out[i][f].tolist(), g[f].tolist())
    return None


def em_noisy(seed):
    """em_run at every signal-to-noise, 1-3 gaussians from guesses good to poor,
    one- and three-gaussian psfs, tol 1e-3 ... 1e-7, maxiter 10 ... 2000, every
    run kind, prepared images and images with negative pixels (_em_case):
    status and numiter equal to the oracle's, the mixture to 1e-9.

    Where that does not hold the stamp is tried against the oracle's OWN scatter:
    em_run's stopping rule compares sums of a per-pixel logL that jumps when a
    pixel crosses chi2 = 25, a run at low signal-to-noise takes 2000 iterations,
    a gaussian that has lost its pixels collapses, and with val / gtot weights of
    both signs (no prep_image) the iteration is not a contraction -- the oracle
    run on the image moved by 1e-13 (the size of a change of summation order)
    then stops elsewhere, or ends somewhere else altogether (neutral directions
    of a broad gaussian on noise: seen, five runs agreeing to 1e-8 and the sixth
    with p = 12.06 for 8.78 -- the kernel's 12.06).  The kernel's result has to
    be one the oracle gives for one of 40 such images (same status and numiter,
    mixture to 1e-6), or the oracle's own runs have to scatter (another status
    or numiter, or a mixture moved by > 1e-7: the 1e-13 amplified a million
    times); a kernel that differs where the oracle is steady is a failure.
    Returns (stamps the oracle raises on, stamps settled by the scatter)."""
    from ngmix_amd.batch import StampBatch, GMixBatch
    c = _em_case(seed)
    n, ngauss, npsf = c["n"], c["ngauss"], c["npsf"]
    sb = StampBatch.from_images(c["images"], c["weights"], c["jac"])
    gm0, _ = GMixBatch.from_pars(c["full"].reshape(n, -1), "full", ngauss=ngauss)
    psf, _ = GMixBatch.from_pars(np.tile(c["psf"].reshape(1, -1), (n, 1)), "full", ngauss=npsf)
    out, status, conv = sb.em(gm0, psf, sky=c["sky"], miniter=c["miniter"], maxiter=c["maxiter"],
                              tol=c["tol"], kind=c["kind"])
    status = status.cpu().numpy()
    out = out.cpu().numpy()
    gm_out = gm0.to_numpy().reshape(n, ngauss)
    nraise = nill = 0
    for i in range(n):
        ref = _em_oracle(c, i, 0)
        nraise += int(ref[0] != 0)
        why = _em_same(c, i, status, out, gm_out, ref, 1e-9, 1e-11)
        if why is None:
            continue
        unsteady = False
        for k in range(1, 41):
            other = _em_oracle(c, i, k)
            if _em_same(c, i, status, out, gm_out, other, 1e-6, 1e-8) is None:
                break
            unsteady = unsteady or other[:2] != ref[:2] or (ref[0] == 0 and any(
                not np.allclose(other[3][f], ref[3][f], rtol=1e-7, atol=1e-9) for f in EM_FIELDS))
        else:
            assert unsteady, (i, "prepped" if c["prepped"] else "raw", "ngauss", ngauss,
                              "kind", c["kind"], "the oracle is steady") + why
        nill += 1
    return nraise, nill


def wsums_and_derivs(seed):
    """get_weighted_sums / get_higher_order_weighted_sums (true exp, no cut,
    maxrad, the ierr > 0 test of the 6-moment form) and deriv_images (the LM
    jacobian's images, masked layout) on a ragged batch with sheared jacobians
    against the oracle: sums to 1e-11 of their scale, images to 2e-13 of peak"""
    import torch
    from ngmix_amd import _lib
    from ngmix_amd.batch import StampBatch, GMixBatch, records_to_numpy
    from oracle import oracle as ora
    r = np.random.RandomState(seed)
    n, scale = 6, 0.263
    shapes = [(int(r.randint(3, 72)), int(r.randint(3, 72))) for _ in range(n)]
    imgs = [r.normal(size=sh) + 5.0 for sh in shapes]
    wts = [r.uniform(0.5, 2.0, size=sh) for sh in shapes]
    for w in wts:
        w[r.uniform(size=w.shape) < 0.03] = 0.0
    jacs = []
    for sh in shapes:
        a, d = scale * (1 + r.uniform(-0.1, 0.1, size=2))
        b, c = r.uniform(-0.03, 0.03, size=2)
        jacs.append(np.array([(sh[0] - 1) / 2 + r.uniform(-0.5, 0.5),
                              (sh[1] - 1) / 2 + r.uniform(-0.5, 0.5), a, b, c, d, a * d - b * c,
                              np.sqrt(abs(a * d - b * c))]))
    # the 17-moment form divides by ierr^2 with no test (gmix_nb.py:772): no zeros there
    for nmom, izw in ((6, True), (6, False), (17, True)):
        ww = wts if nmom == 6 else [np.where(w > 0, w, 1.0) for w in wts]
        sb = StampBatch.from_arrays(imgs, ww, jacs, [izw] * n)
        wpars = np.zeros((n, 6))
        wpars[:, 0:2] = r.uniform(-1, 1, size=(n, 2)) * scale
        wpars[:, 2:4] = r.uniform(-0.2, 0.2, size=(n, 2))
        wpars[:, 4] = r.uniform(0.2, 2.0, size=n)
        wpars[:, 5] = 1.0
        wt, _ = GMixBatch.from_pars(wpars, "gauss")
        wt.set_norms()
        wt_h = wt.to_numpy()
        maxrad = r.uniform(1.0, 12.0, size=n)
        res, status = sb.weighted_sums(wt, maxrad, nmom=nmom)
        assert int(status.abs().sum()) == 0
        rec = records_to_numpy(res, _lib.moments_result_dtype(nmom))
        for i in range(n):
            j = np.zeros(1, dtype=ora.JACOBIAN_DTYPE)
            j[0] = tuple(jacs[i])
            pix = ora.make_pixels(imgs[i], ww[i], j, izw)
            ref = np.zeros(1, dtype=ora.moments_result_dtype(nmom))
            w = ti.conv_rec(wt_h[i], ora.GAUSS2D_DTYPE)
            assert ora.get_weighted_sums(w, pix, ref, float(maxrad[i])) == 0
            assert rec["npix"][i] == ref["npix"][0], ("npix", nmom, i)
            for f in ("wsum", "sums", "sums_cov"):
                a_, b_ = np.asarray(rec[f][i], dtype="f8"), np.asarray(ref[f][0], dtype="f8")
                np.testing.assert_allclose(a_, b_, rtol=0, atol=1e-11 * max(np.abs(b_).max(), 1e-300),
                                           err_msg="%s nmom %d stamp %d" % (f, nmom, i))
    # deriv_images: gauss / exp / dev mixtures (x) a psf, the masked layout
    sb = StampBatch.from_arrays(imgs, wts, jacs, [True] * n)
    model = str(r.choice(["gauss", "exp", "dev"]))
    pars = np.zeros((n, 6))
    pars[:, 0:2] = r.uniform(-1, 1, size=(n, 2)) * scale
    pars[:, 2:4] = r.uniform(-0.4, 0.4, size=(n, 2))
    pars[:, 4] = r.uniform(0.1, 1.5, size=n)
    pars[:, 5] = r.uniform(1.0, 100.0, size=n)
    gm0, _ = GMixBatch.from_pars(pars, model)
    psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.02, -0.01, 0.27, 1.0], (n, 1)), "gauss")
    gmc, _ = gm0.convolve(psf)
    G = gmc.ngauss
    GC = gmc.data.reshape(n, G, 13)
    gpars = GC[:, :, 0:6].contiguous()
    dcov = torch.from_numpy(r.normal(size=(n, G, 3, 3))).cuda()
    out = sb.deriv_images(gpars.reshape(-1, 6), dcov.reshape(-1, 3, 3), G).cpu().numpy()
    gp, dc = gpars.cpu().numpy(), dcov.cpu().numpy()
    at = 0
    for i in range(n):
        j = np.zeros(1, dtype=ora.JACOBIAN_DTYPE)
        j[0] = tuple(jacs[i])
        pix = ora.make_pixels(imgs[i], wts[i], j, True)
        ref = np.zeros((6, pix.size))
        ora.deriv_images(gp[i], dc[i], pix["v"], pix["u"], pix["area"], ref)
        got = out[at:at + 6 * pix.size].reshape(6, pix.size)
        at += 6 * pix.size
        for k in range(6):
            np.testing.assert_allclose(got[k], ref[k], rtol=0,
                                       atol=2e-13 * max(np.abs(ref[k]).max(), 1e-300),
                                       err_msg="deriv image %d stamp %d" % (k, i))


def seam_forms(seed):
    """the per-object entry points the reference's njit seam binds to
    (ngmix_fill_pixels / fill_coords / get_loglike / fill_fdiff / render /
    get_model_s2n_sum on host arrays): pixel and coordinate arrays, fdiff, the
    fast render and the norms written into the caller's mixture BIT-IDENTICAL to
    the oracle's; the sums to 1e-12"""
    import ctypes
    from ngmix_amd import _lib
    from oracle import oracle as ora
    L = _lib.lib()
    r = np.random.RandomState(seed)
    nrow, ncol = int(r.randint(1, 70)), int(r.randint(1, 70))
    scale = 0.263
    a, d = scale * (1 + r.uniform(-0.1, 0.1, size=2))
    b, c = r.uniform(-0.03, 0.03, size=2)
    jv = (float((nrow - 1) / 2 + r.uniform(-0.5, 0.5)), float((ncol - 1) / 2 + r.uniform(-0.5, 0.5)),
          float(a), float(b), float(c), float(d), float(a * d - b * c),
          float(np.sqrt(abs(a * d - b * c))))
    jac = np.zeros(1, dtype=_lib.JACOBIAN_DTYPE)
    jac[0] = jv
    joc = np.zeros(1, dtype=ora.JACOBIAN_DTYPE)
    joc[0] = jv
    image = r.normal(size=(nrow, ncol))
    weight = r.uniform(0.5, 2.0, size=(nrow, ncol))
    weight[r.uniform(size=weight.shape) < 0.05] = 0.0
    weight[0, 0] = 1.0
    izw = bool(r.randint(2))
    ref_pix = ora.make_pixels(image, weight, joc, izw)
    pix = np.zeros(ref_pix.size, dtype=_lib.PIXEL_DTYPE)
    assert L.ngmix_fill_pixels(_lib.ptr(pix), pix.size, _lib.ptr(image), _lib.ptr(weight), nrow,
                               ncol, _lib.ptr(jac), int(izw)) == 0
    for f in ("u", "v", "area", "val", "ierr"):
        assert np.array_equal(pix[f], ref_pix[f]), f
    ng = int(r.randint(1, 20))
    gmh = tp._random_mixtures(r, 1, ng, scale)[0]
    gm = gmh.copy()
    gmo = np.zeros(ng, dtype=ora.GAUSS2D_DTYPE)
    for nm in ora.GAUSS2D_DTYPE.names:
        gmo[nm] = gmh[nm]
    ll, sn, sd = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    npix = ctypes.c_int64()
    st = L.ngmix_get_loglike(_lib.ptr(gm), gm.size, _lib.ptr(pix), pix.size, ctypes.byref(ll),
                             ctypes.byref(sn), ctypes.byref(sd), ctypes.byref(npix))
    sto, res = ora.get_loglike(gmo, ref_pix)
    assert st == sto == 0
    for nm in ora.GAUSS2D_DTYPE.names:        # the lazy norms, written in place
        assert np.array_equal(gm[nm], gmo[nm], equal_nan=True), nm
    assert npix.value == res[3]
    np.testing.assert_allclose([ll.value, sn.value, sd.value], res[:3], rtol=1e-12, atol=1e-300)
    start = int(r.randint(0, 20))
    fd = np.full(start + pix.size, 7.0)
    rfd = np.full(start + pix.size, 7.0)
    assert L.ngmix_fill_fdiff(_lib.ptr(gm), gm.size, _lib.ptr(pix), pix.size, _lib.ptr(fd),
                              start) == 0
    ora.fill_fdiff(gmo, ref_pix, rfd, start)
    assert np.array_equal(fd, rfd)
    s2n = ctypes.c_double()
    assert L.ngmix_get_model_s2n_sum(_lib.ptr(gm), gm.size, _lib.ptr(pix), pix.size,
                                     ctypes.byref(s2n)) == 0
    np.testing.assert_allclose(s2n.value, ora.get_model_s2n_sum(gmo, ref_pix)[1], rtol=1e-12)
    coords = np.zeros(nrow * ncol, dtype=_lib.COORD_DTYPE)
    assert L.ngmix_fill_coords(_lib.ptr(coords), nrow, ncol, _lib.ptr(jac)) == 0
    rco = ora.make_coords((nrow, ncol), joc)
    for f in ("u", "v", "area"):
        assert np.array_equal(coords[f], rco[f]), f
    base = r.normal(size=coords.size)
    im, rim = base.copy(), base.copy()
    assert L.ngmix_render(_lib.ptr(gm), gm.size, _lib.ptr(coords), coords.size, _lib.ptr(im),
                          1) == 0
    ora.render(gmo, rco, rim, 1)
    assert np.array_equal(im, rim)              # accumulate-into, fast exp: to the bit
    im, rim = np.zeros(coords.size), np.zeros(coords.size)
    assert L.ngmix_render(_lib.ptr(gm), gm.size, _lib.ptr(coords), coords.size, _lib.ptr(im),
                          0) == 0
    ora.render(gmo, rco, rim, 0)
    # (true exp: the device's libm and the host's, a few ulp per term; the
    # random mixtures have amplitudes of both signs, so relative to the image)
    np.testing.assert_allclose(im, rim, rtol=1e-14, atol=1e-14 * np.abs(rim).max() + 1e-300)


def main():
    global budget, rng
    if len(sys.argv) > 1:
        budget = float(sys.argv[1])
    if len(sys.argv) > 2:
        rng = np.random.RandomState(int(sys.argv[2]))
    if os.environ.get("FUZZ_EM_SEEDS"):
        for sd in os.environ["FUZZ_EM_SEEDS"].split(","):
            try:
                print(sd, em_noisy(int(sd)))
            except AssertionError as e:
                print(sd, "AssertionError", e)
        sys.exit(0)
    if os.environ.get("FUZZ_ONLY"):
        # one family only: FUZZ_ONLY=em_noisy python tools/fuzz_vs_oracle.py 120
        fn = {"em_noisy": em_noisy, "admom_noisy": admom_noisy, "seam": seam_forms,
              "wsums_derivs": wsums_and_derivs}[os.environ["FUZZ_ONLY"]]
        t0, nrun, bad, tally = time.time(), 0, [], np.zeros(2, dtype=int)
        while time.time() - t0 < budget and len(bad) < 10:
            sd = int(rng.randint(1 << 30))
            try:
                got = fn(sd)
                if isinstance(got, tuple):
                    tally += np.array(got)
            except Exception:
                bad.append(sd)
                print("FAIL", sd)
                print(traceback.format_exc(limit=3))
            nrun += 1
        print("fuzz_vs_oracle %s: %.0f s, %d cases, (flagged / raising, ill-conditioned) stamps %s, "
              "failures %s" % (os.environ["FUZZ_ONLY"], time.time() - t0, nrun, tally, bad))
        sys.exit(1 if bad else 0)
    if os.environ.get("FUZZ_ADMOM_SEEDS"):
        for sd in os.environ["FUZZ_ADMOM_SEEDS"].split(","):
            try:
                print(sd, admom_noisy(int(sd)))
            except AssertionError as e:
                print(sd, "AssertionError", e)
        sys.exit(0)
    t0 = time.time()
    counts = {"pixpass": 0, "admom": 0, "em": 0, "em_many": 0, "wsums_derivs": 0, "seam": 0, "admom_noisy": 0, "em_noisy": 0,
              "em_noisy_range_errors": 0, "em_noisy_unsteady_stamps": 0,
              "admom_noisy_flagged_stamps": 0, "admom_noisy_ill_conditioned_stamps": 0}
    failures = []
    while time.time() - t0 < budget:
        u = rng.uniform()
        try:
            if u < 0.6:
                dims = (dim(), dim())
                ng = int(rng.choice([1, 2, 3, 4, 6, 8, 10, 16, 24, 33, 40, 48]))
                exact = bool(rng.randint(2))
                case = ("pixpass", dims, ng, exact)
                tp.test_batch_random_vs_oracle(dims, ng, exact)
                counts["pixpass"] += 1
            elif u < 0.62:
                seed = int(rng.randint(1 << 30))
                case = ("seam", seed)
                seam_forms(seed)
                counts["seam"] += 1
            elif u < 0.65:
                seed = int(rng.randint(1 << 30))
                case = ("wsums_derivs", seed)
                wsums_and_derivs(seed)
                counts["wsums_derivs"] += 1
            elif u < 0.7:
                seed = int(rng.randint(1 << 30))
                case = ("admom_noisy", seed)
                nflag, nill = admom_noisy(seed)
                counts["admom_noisy_flagged_stamps"] += nflag
                counts["admom_noisy_ill_conditioned_stamps"] += nill
                counts["admom_noisy"] += 1
            elif u < 0.75:
                seed = int(rng.randint(1 << 30))
                case = ("em_noisy", seed)
                nfail, nill = em_noisy(seed)
                counts["em_noisy_range_errors"] += nfail
                counts["em_noisy_unsteady_stamps"] += nill
                counts["em_noisy"] += 1
            elif u < 0.8:
                # (the iterative cases draw objects that fit the stamp: sides >= 16)
                shape = (max(dim(), 16), max(dim(), 16))
                shape = (min(shape[0], 96), min(shape[1], 96))
                case = ("admom", shape)
                ti.test_admom_kernel_variants_vs_oracle(shape)
                counts["admom"] += 1
            elif u < 0.9:
                shape = (min(max(dim(), 20), 64), min(max(dim(), 20), 64))
                ng = int(rng.randint(1, 3))
                case = ("em", shape, ng)
                ti.test_em_kernel_variants_vs_oracle(shape, ng)
                counts["em"] += 1
            else:
                # every run kind, up to nine object gaussians, 1- and 3-gaussian psfs
                # (with the comparison's own assertion of which kernel served it)
                shape = (min(max(dim(), 20), 80), min(max(dim(), 20), 80))
                ng, npsf, kind = int(rng.randint(1, 10)), int(rng.choice([1, 3])), int(rng.randint(4))
                case = ("em_many", shape, ng, npsf, kind)
                ti.test_em_many_gaussians_vs_oracle(shape, ng, npsf, kind)
                counts["em_many"] += 1
        except Exception:
            failures.append((case, traceback.format_exc(limit=3)))
            print("FAIL", case)
            print(failures[-1][1])
            sys.stdout.flush()
            if len(failures) >= 10:
                break
    print("fuzz_vs_oracle: %.0f s, cases run %s, failures %d" % (time.time() - t0, counts,
                                                                 len(failures)))
    for case, tb in failures:
        print("  ", case)
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
