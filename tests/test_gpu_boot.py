"""
bootstrap_batch against the REFERENCE'S OWN Bootstrapper (tests/golden/boot.npz,
oracle/gen_golden_boot.py: Bootstrapper(Runner(Fitter), PSFRunner(...)) run
object by object with stored guesses): one batch per set started from the same
guesses reproduces, stamp by stamp and object by object,

  * which psf fits pass, on which attempt, with how many evaluations;
  * which epochs are dropped (remove_failed_psf_obs) and which objects are
    lost to BootPSFFailure (a band with no epoch left);
  * the object fits on the epochs that are left: flags, attempts, nfev (exact
    for the lmder models, to a few evaluations for the lmdif ones) and the
    parameters;
  * the guessers' psf fluxes (PSFFluxFitter per band with the fitted psfs).

Tolerances: psf and object parameters of lmder fits to 1e-6 of their error bar
or 1e-7 relative (two LM implementations stopping at ftol = xtol = 1e-5 from
the same start: the iterates agree to rounding, see test_gpu_lm_batch), lmdif
fits to 1e-3 sigma; psf fluxes 1e-7 relative.
"""
import numpy as np
import pytest

import ngmix_amd as ngmix
from ngmix_amd import prior_batch as pb
from ngmix_amd.batch import StampBatch
from ngmix_amd.pipeline import bootstrap_batch, BOOT_PSF_FAILURE

pytestmark = pytest.mark.gpu


def _set(g, tag):
    ns = g[tag + "_images"].shape[0]
    sig = g[tag + "_sigma"]
    w = np.ones_like(g[tag + "_images"]) / sig[:, None, None] ** 2
    sb = StampBatch.from_images(g[tag + "_images"], w, g[tag + "_jac"])
    psig = g[tag + "_psf_sigma"]
    pw = np.ones_like(g[tag + "_psf_images"]) / psig[:, None, None] ** 2
    psb = StampBatch.from_images(g[tag + "_psf_images"], pw, g[tag + "_psf_jac"])
    assert sb.n == psb.n == ns
    kind = str(g[tag + "_psf_kind"])
    kw = dict(model=str(g[tag + "_model"]), psf_fitter=kind,
              psf_ngauss=int(g[tag + "_psf_ngauss"]),
              psf_ntry=int(g[tag + "_psf_ntry"]), ntry=int(g[tag + "_ntry"]),
              psf_guess=g[tag + "_psf_guess"],
              stamp_obj=g[tag + "_stamp_obj"], stamp_band=g[tag + "_stamp_band"])
    if kind == "em":
        kw["em_pars"] = {"maxiter": int(g[tag + "_psf_fit_maxiter"]),
                         "tol": float(g[tag + "_psf_fit_tol"])}
    else:
        kw["psf_fit_pars"] = {k: (int if k == "maxfev" else float)(g[tag + "_psf_fit_" + k])
                              for k in ("maxfev", "ftol", "xtol")}
    if tag + "_fit_maxfev" in g:
        kw["fit_pars"] = {k: (int if k == "maxfev" else float)(g[tag + "_fit_" + k])
                          for k in ("maxfev", "ftol", "xtol")}
    if tag + "_prior_cen_sigma" in g:
        nband = int(g[tag + "_nband"].max())
        cs, gs = float(g[tag + "_prior_cen_sigma"]), float(g[tag + "_prior_g_sigma"])
        # the joint prior as the generator gave it to the reference's
        # Bootstrapper: PriorSimpleSep of CenPrior / GPriorBA / TwoSidedErf,
        # here from ngmix_amd's own priors / joint_prior modules
        from ngmix_amd import priors, joint_prior
        rng = np.random.RandomState(0)
        kw["prior"] = joint_prior.PriorSimpleSep(
            priors.CenPrior(0.0, 0.0, cs, cs, rng=rng), priors.GPriorBA(gs, rng=rng),
            priors.TwoSidedErf(*g[tag + "_prior_T_erf"], rng=rng),
            [priors.TwoSidedErf(*g[tag + "_prior_F_erf"], rng=rng) for _ in range(nband)])
    return sb, psb, kw


def test_host_and_batch_priors_are_one_fit(golden):
    """set A with the host joint prior and with the batch prior built by hand:
    the same records"""
    g = golden("boot")
    sb, psb, kw = _set(g, "A")
    res = bootstrap_batch(sb, psb, guess=g["A_guess"], **kw)
    hp = kw["prior"]
    kw["prior"] = pb.PriorSimpleSepBatch(
        pb.GaussianCen(0.0, 0.0, hp.cen_prior.sigma1, hp.cen_prior.sigma2),
        pb.GPriorBA(hp.g_prior.sigma),
        pb.TwoSidedErf(*g["A_prior_T_erf"]),
        [pb.TwoSidedErf(*g["A_prior_F_erf"]) for _ in range(hp.nband)])
    res2 = bootstrap_batch(sb, psb, guess=g["A_guess"], **kw)
    for k in ("flags", "nfev", "pars", "pars_cov", "lnprob"):
        np.testing.assert_array_equal(res[k], res2[k], err_msg=k)


LMDER_PSF = {"A": True, "B": False, "C": None, "D": True}
LMDER_OBJ = {"A": True, "B": False, "C": True, "D": False}


@pytest.mark.parametrize("tag", ["A", "B", "C", "D"])
def test_bootstrap_batch_vs_reference_bootstrapper(golden, tag):
    g = golden("boot")
    sb, psb, kw = _set(g, tag)
    guess = g[tag + "_guess"]
    res = bootstrap_batch(sb, psb, guess=guess, **kw)

    # ---- the psf stage, stamp by stamp
    ref_flags = g[tag + "_ref_psf_flags"]
    kept = g[tag + "_ref_kept"]
    np.testing.assert_array_equal(res["psf_flags"] == 0, ref_flags == 0)
    np.testing.assert_array_equal(res["psf_ntry"], g[tag + "_ref_psf_ntry"])
    if LMDER_PSF[tag] is not None:
        # LM psf fits: the failures are maxfev stops (ier 5 -> flags 1)
        np.testing.assert_array_equal(res["psf_flags"], ref_flags)
    else:
        # EM: a maxiter stop is the same flag; a degenerate mixture is a range
        # error here (numba raises ZeroDivisionError there; under the shim the
        # reference's division returns inf and the failure surfaces as either)
        ok = ref_flags == 32
        np.testing.assert_array_equal(res["psf_flags"][ok], ref_flags[ok])
    nfev, rnfev = res["psf_nfev"], g[tag + "_ref_psf_nfev"]
    if LMDER_PSF[tag] or LMDER_PSF[tag] is None:
        # lmder evaluations / EM iterations: exact (of the fits that passed; a
        # failed EM fit's count is not in the reference's result)
        np.testing.assert_array_equal(nfev[kept | (rnfev >= 0)], rnfev[kept | (rnfev >= 0)])
    else:
        assert np.all(np.abs(nfev - rnfev) <= 2 * (4 + 2 * kw["psf_ngauss"]) + 2)
    # the fitted psf mixtures of the stamps that stay (the reference keeps
    # res['pars'] / the EM mixture; psf_gmix is flux-normalised as convolve
    # uses it)
    ref_pars = g[tag + "_ref_psf_pars"]
    gm = res["psf_gmix"].to_numpy()
    kind = kw["psf_fitter"]
    for i in np.nonzero(kept)[0]:
        pp = ref_pars[i]
        if kind == "em":
            full = pp.reshape(-1, 6)
            p = full[:, 0] / full[:, 0].sum()
            order_ref, order = np.argsort(full[:, 3] + full[:, 5]), \
                np.argsort(gm[i]["irr"] + gm[i]["icc"])
            np.testing.assert_allclose(gm[i]["p"][order], p[order_ref], rtol=1e-6)
            np.testing.assert_allclose((gm[i]["irr"] + gm[i]["icc"])[order],
                                       (full[:, 3] + full[:, 5])[order_ref], rtol=1e-6)
            np.testing.assert_allclose(gm[i]["row"][order], full[order_ref, 1],
                                       rtol=1e-5, atol=1e-8)
        elif kind == "coellip":
            ng = kw["psf_ngauss"]
            np.testing.assert_allclose(gm[i]["irr"] + gm[i]["icc"], pp[4:4 + ng], rtol=2e-4)
            np.testing.assert_allclose(gm[i]["p"], pp[4 + ng:] / pp[4 + ng:].sum(), rtol=2e-4)
            np.testing.assert_allclose(gm[i]["row"], pp[0], rtol=1e-3, atol=1e-6)
        else:
            np.testing.assert_allclose((gm[i]["irr"] + gm[i]["icc"])[0], pp[4], rtol=1e-7)
            np.testing.assert_allclose(gm[i]["row"][0], pp[0], rtol=1e-6, atol=1e-9)
            np.testing.assert_allclose(gm[i]["col"][0], pp[1], rtol=1e-6, atol=1e-9)

    # ---- the epochs that leave, the objects that are lost
    boot_failed = g[tag + "_ref_obj_boot_failed"]
    np.testing.assert_array_equal(res["boot_failed"], boot_failed)
    sobj = g[tag + "_stamp_obj"]
    np.testing.assert_array_equal(res["kept"], kept & ~boot_failed[sobj])
    assert kept.sum() < kept.size        # (every set drops something)
    np.testing.assert_array_equal(res["flags"][boot_failed], BOOT_PSF_FAILURE)
    assert np.all(res["nfev"][boot_failed] == 0)
    assert np.all(np.isnan(res["pars"][boot_failed]))

    # ---- the object fits on what is left
    ok = ~boot_failed
    np.testing.assert_array_equal(res["flags"][ok], g[tag + "_ref_obj_flags"][ok])
    np.testing.assert_array_equal(res["ntry"][ok], g[tag + "_ref_obj_ntry"][ok])
    assert np.all(res["ntry"][boot_failed] == 0)
    npars = res["pars"].shape[1]
    rpars = g[tag + "_ref_obj_pars"][:, :npars]
    rerr = g[tag + "_ref_obj_pars_err"][:, :npars]
    if LMDER_OBJ[tag]:
        np.testing.assert_array_equal(res["nfev"][ok], g[tag + "_ref_obj_nfev"][ok])
        tol = 1e-6 if LMDER_PSF[tag] else 2e-3   # (an lmdif / EM psf underneath)
    else:
        assert np.all(np.abs(res["nfev"][ok] - g[tag + "_ref_obj_nfev"][ok]) <= 2 * npars + 2)
        tol = 2e-3
    assert np.all(np.abs(res["pars"][ok] - rpars[ok]) <= tol * rerr[ok] +
                  1e-7 * np.abs(rpars[ok]))
    np.testing.assert_allclose(res["pars_err"][ok], rerr[ok], rtol=1e-3 if tol > 1e-5 else 1e-5)
    np.testing.assert_allclose(res["lnprob"][ok], g[tag + "_ref_obj_lnprob"][ok],
                               rtol=1e-6 if tol > 1e-5 else 1e-9)
    # the attempts: set A holds objects that needed the second guess
    if tag == "A":
        assert (res["ntry"] == 2).sum() >= 2 and (res["psf_ntry"] == 2).sum() >= 4
        assert np.any((res["psf_ntry"] == 2) & (res["psf_flags"] == 0))
    np.testing.assert_array_equal(res["guess"][ok],
                                  np.where((res["ntry"] == 2)[:, None], guess[-1],
                                           guess[0])[ok][:, :npars])


@pytest.mark.parametrize("tag", ["A", "B", "D"])
def test_psf_flux_guess_vs_reference(golden, tag):
    """guesser='psfflux': the per-band template fluxes of _get_psf_fluxes
    (guessers.py:205-262) over the epochs the bootstrap kept, with the psf
    mixtures the psf stage fitted; the guess scatters around them as
    TPSFFluxGuesser's does and the fits recover the reference's answer"""
    g = golden("boot")
    sb, psb, kw = _set(g, tag)
    kw["ntry"] = 3
    res = bootstrap_batch(sb, psb, guesser="psfflux", Tguess=0.5, rng=np.random.RandomState(3),
                          **kw)
    boot_failed = g[tag + "_ref_obj_boot_failed"]
    ok = ~boot_failed
    nband = int(g[tag + "_nband"].max())
    ref_flux = g[tag + "_ref_obj_psf_flux"][:, :nband]
    assert np.all(res["psf_flux_flags"][ok] == 0)
    np.testing.assert_allclose(res["psf_flux"][ok], ref_flux[ok],
                               rtol=1e-7 if LMDER_PSF[tag] else 2e-5)
    npars = res["pars"].shape[1]
    nshape = npars - nband
    gs = res["guess"][ok]
    if "prior" in kw:
        # (set A: a host joint prior -- centre and shape are draws from it, as
        # TPSFFluxAndPriorGuesser's are)
        assert all(np.isfinite(kw["prior"].get_lnprob_scalar(p)) for p in gs)
        assert np.abs(gs[:, 0:2]).max() > 0.01
    else:
        assert np.all(np.abs(gs[:, 0:2]) <= 0.01) and np.all(np.abs(gs[:, 2:4]) <= 0.02)
    assert np.all(np.abs(gs[:, 4] / 0.5 - 1.0) <= 0.1)
    assert np.all(np.abs(gs[:, nshape:] / res["psf_flux"][ok] - 1.0) <= 0.1)
    # the same minimum as the reference reached from its own guess
    conv = ok & (res["flags"] == 0)
    assert conv.sum() >= ok.sum() - 1
    rpars = g[tag + "_ref_obj_pars"][:, :npars]
    rerr = g[tag + "_ref_obj_pars_err"][:, :npars]
    assert np.all(np.abs(res["pars"][conv] - rpars[conv]) <= 0.05 * rerr[conv])


def test_keep_failed_psf_epochs_flags_the_object(golden):
    """drop_failed_psf=False (the reference's ignore_failed_psf=False runs the
    fit on everything): every object is fitted and the ones with a failed psf
    carry BOOT_PSF_FAILURE next to their fit's flags"""
    g = golden("boot")
    sb, psb, kw = _set(g, "D")
    res = bootstrap_batch(sb, psb, guess=g["D_guess"], drop_failed_psf=False, **kw)
    sobj = g["D_stamp_obj"]
    bad_obj = np.bincount(sobj, weights=(g["D_ref_psf_flags"] != 0).astype("f8")) > 0
    assert np.all(res["kept"]) and not res["boot_failed"].any()
    assert np.all((res["flags"] & BOOT_PSF_FAILURE != 0) == bad_obj)
    assert np.all(res["nfev"] > 0) and np.all(res["ntry"] == 1)
    clean = ~bad_obj
    assert np.all(res["flags"][clean] == 0) and np.all(np.isfinite(res["pars"][clean]))


def test_bootstrap_batch_edge_cases(golden):
    """every object lost to failed psf fits; the psf model fitters without a
    caller's guess (their start comes from the adaptive moments); the models
    with extra shape parameters from either guesser"""
    g = golden("boot")
    # ---- set D with every psf guess far off and one attempt: all psf fits fail
    sb, psb, kw = _set(g, "D")
    far = g["D_psf_guess"].copy()
    far[..., 0:2] = [2.0, -2.0]
    far[..., 4] = 20.0
    kw.update(psf_guess=far, psf_ntry=1)
    res = bootstrap_batch(sb, psb, guess=g["D_guess"], **kw)
    assert res["boot_failed"].all() and not res["kept"].any()
    assert np.all(res["flags"] == BOOT_PSF_FAILURE) and np.all(res["nfev"] == 0)
    assert res["pars"].shape == g["D_guess"][0].shape and np.all(np.isnan(res["pars"]))
    assert np.all(res["ntry"] == 0) and res["rounds"] == 0

    # ---- psf model fits started from the adaptive moments, on the good stamps of set A
    sb, psb, kw = _set(g, "A")
    good = np.nonzero(g["A_ref_kept"] & ~g["A_ref_obj_boot_failed"][g["A_stamp_obj"]])[0]
    sobj = g["A_stamp_obj"][good]
    _, sobj = np.unique(sobj, return_inverse=True)
    sband = g["A_stamp_band"][good]
    gs, gp = sb.select(good), psb.select(good)
    truth = g["A_truth"][~g["A_ref_obj_boot_failed"]]
    for fitter in ("gauss", "turb", "admom"):
        res = bootstrap_batch(gs, gp, model="exp", psf_fitter=fitter, stamp_obj=sobj,
                              stamp_band=sband, rng=np.random.RandomState(4))
        assert np.all(res["psf_flags"] == 0) and res["kept"].all()
        assert np.all(res["flags"] == 0)
        # (a one-gaussian psf model for a turbulent psf: the size is biased, the
        # centre and the fluxes are not)
        assert np.all(np.abs(res["pars"][:, 0:2] - truth[:, 0:2]) < 5 * res["pars_err"][:, 0:2])
        np.testing.assert_allclose(res["pars"][:, 5:], truth[:, 5:], rtol=0.15)
    np.testing.assert_allclose(res["psf_T"], 0.3, rtol=0.35)

    # ---- 'bdf' and 'bd' from either guesser (set B's stamps: two bands)
    sb, psb, kw = _set(g, "B")
    keep = np.nonzero(g["B_ref_kept"])[0]
    sobj, sband = g["B_stamp_obj"][keep], g["B_stamp_band"][keep]
    gs, gp = sb.select(keep), psb.select(keep)
    for model, npars in (("bdf", 8), ("bd", 9)):
        for guesser in ("admom", "psfflux"):
            res = bootstrap_batch(gs, gp, model=model, psf_fitter="coellip", psf_ngauss=2,
                                  guesser=guesser, Tguess=0.5, ntry=3, stamp_obj=sobj,
                                  stamp_band=sband, rng=np.random.RandomState(8))
            assert res["pars"].shape == (3, npars) and res["guess"].shape == (3, npars)
            assert np.all(res["psf_flags"] == 0)
            ok = res["flags"] == 0
            assert ok.sum() >= 2
            # (data drawn from a 'bdf' profile: 'bd' has a size ratio to spend)
            np.testing.assert_allclose(res["pars"][ok][:, npars - 2:],
                                       g["B_truth"][ok][:, 6:8], rtol=0.2)


def test_bootstrap_many_from_reference_style_objects(golden):
    """bootstrap_many: set A of boot.npz as a list of MultiBandObsLists whose
    observations carry their psf observations -- what the reference's
    Bootstrapper.go takes one at a time -- gives bootstrap_batch's result for the
    same stamps, per object, and leaves the psf results on the observations"""
    import ngmix_amd as ngmix
    from ngmix_amd.pipeline import bootstrap_many
    g = golden("boot")
    sb, psb, kw = _set(g, "A")
    sobj, sband = g["A_stamp_obj"], g["A_stamp_band"]
    objs = []
    for o in range(int(sobj.max()) + 1):
        mb = ngmix.MultiBandObsList()
        for b in range(int(sband.max()) + 1):
            ol = ngmix.ObsList()
            for s in np.nonzero((sobj == o) & (sband == b))[0]:
                def jac_of(rec):
                    return ngmix.Jacobian(row=float(rec["row0"]), col=float(rec["col0"]),
                                          dvdrow=float(rec["dvdrow"]), dvdcol=float(rec["dvdcol"]),
                                          dudrow=float(rec["dudrow"]), dudcol=float(rec["dudcol"]))
                pim = g["A_psf_images"][s]
                pobs = ngmix.Observation(pim, weight=np.full(pim.shape, 1.0 / g["A_psf_sigma"][s] ** 2),
                                         jacobian=jac_of(g["A_psf_jac"][s]))
                im = g["A_images"][s]
                ol.append(ngmix.Observation(im, weight=np.full(im.shape, 1.0 / g["A_sigma"][s] ** 2),
                                            jacobian=jac_of(g["A_jac"][s]), psf=pobs))
            mb.append(ol)
        objs.append(mb)
    kw.pop("stamp_obj")
    kw.pop("stamp_band")
    many = bootstrap_many(objs, guess=g["A_guess"], set_psf_results=True, **kw)
    ref = bootstrap_batch(sb, psb, guess=g["A_guess"], stamp_obj=sobj, stamp_band=sband, **kw)
    assert len(many) == len(objs)
    for i in range(len(objs)):
        r = many[i]
        assert r["flags"] == ref["flags"][i] and r["nfev"] == ref["nfev"][i]
        np.testing.assert_array_equal(r["pars"], ref["pars"][i])
        if r["flags"] == 0:
            assert r["flux"].shape == (3,) and r["s2n"] == ref["s2n"][i]
        elif r["flags"] == BOOT_PSF_FAILURE:
            assert np.all(np.isnan(r["pars"])) and "lnprob" not in r
    # the psf stage's results on the psf observations (PSFRunner's side effect)
    flat = [e for mb in objs for ol in mb for e in ol]
    for s, e in enumerate(flat):
        assert e.psf.meta["result"]["flags"] == ref["psf_flags"][s]
        assert e.psf.has_gmix() == (ref["psf_flags"][s] == 0)
    k = int(np.nonzero(ref["psf_flags"] == 0)[0][0])
    np.testing.assert_allclose(flat[k].psf.gmix.get_T(), ref["psf_T"][k], rtol=1e-12)


@pytest.mark.parametrize("model", ["exp", "bdf", "bd"])
def test_bootstrap_batch_with_host_joint_priors_and_prior_guesses(golden, model):
    """the whole pipeline as a caller of the reference sets it up for bulge +
    disk fits: a host joint prior (PriorSimpleSep / PriorBDFSep / PriorBDSep),
    psf-flux guesses drawn through it (TPSFFluxAndPriorGuesser /
    BDFPSFFluxGuesser), the fits with the prior's rows evaluated on the device
    (set B's two-band stamps, data drawn from a 'bdf' profile)"""
    from ngmix_amd import priors, joint_prior
    g = golden("boot")
    sb, psb, kw = _set(g, "B")
    keep = np.nonzero(g["B_ref_kept"])[0]
    sobj, sband = g["B_stamp_obj"][keep], g["B_stamp_band"][keep]
    gs, gp = sb.select(keep), psb.select(keep)
    rng = np.random.RandomState(12)
    scale = 0.263
    cen = priors.CenPrior(0.0, 0.0, scale, scale, rng=rng)
    gp_ = priors.GPriorBA(0.3, rng=rng)
    Tp = priors.TwoSidedErf(-0.1, 0.03, 100.0, 1.0, rng=rng)
    Fp = [priors.TwoSidedErf(-10.0, 1.0, 1.0e5, 100.0, rng=rng) for _ in range(2)]
    fd = priors.Normal(0.5, 0.1, rng=rng, bounds=(0.0, 1.0))
    if model == "exp":
        prior, npars = joint_prior.PriorSimpleSep(cen, gp_, Tp, Fp), 7
    elif model == "bdf":
        prior, npars = joint_prior.PriorBDFSep(cen, gp_, Tp, fd, Fp), 8
    else:
        prior, npars = joint_prior.PriorBDSep(cen, gp_, Tp, priors.Normal(0.0, 0.3, rng=rng),
                                              fd, Fp), 9
    res = bootstrap_batch(gs, gp, model=model, psf_fitter="coellip", psf_ngauss=2,
                          guesser="psfflux", Tguess=0.5, ntry=3, stamp_obj=sobj,
                          stamp_band=sband, rng=np.random.RandomState(8), prior=prior)
    assert res["pars"].shape == (3, npars) and res["guess"].shape == (3, npars)
    assert np.all(res["psf_flags"] == 0)
    ok = res["flags"] == 0
    assert ok.sum() >= 2
    np.testing.assert_allclose(res["pars"][ok][:, npars - 2:], g["B_truth"][ok][:, 6:8], rtol=0.2)
    if model != "exp":
        # the bounded fracdev stays inside its bounds
        fcol = 5 if model == "bdf" else 6
        assert np.all((res["pars"][ok][:, fcol] >= 0.0) & (res["pars"][ok][:, fcol] <= 1.0))
    # every first guess is one the prior accepts, its centre drawn from the prior
    assert all(np.isfinite(prior.get_lnprob_scalar(p)) for p in res["guess"])
    # and the per-object Fitter from the same guesses and mixtures agrees
    # (MINPACK, the host prior)
    psf_gm = res["psf_gmix"].to_numpy()
    for o in np.nonzero(ok)[0][:2]:
        mb = ngmix.MultiBandObsList()
        for b in range(2):
            ol = ngmix.ObsList()
            for i in np.nonzero((sobj == o) & (sband == b))[0]:
                k = keep[i]
                jr = g["B_jac"][k]
                jac = ngmix.Jacobian(row=float(jr["row0"]), col=float(jr["col0"]),
                                     dvdrow=float(jr["dvdrow"]), dvdcol=float(jr["dvdcol"]),
                                     dudrow=float(jr["dudrow"]), dudcol=float(jr["dudcol"]))
                pm = ngmix.GMix(ngauss=2)
                pm.get_data()[:] = psf_gm[i]
                pobs = ngmix.Observation(np.zeros((5, 5)), jacobian=jac, gmix=pm)
                w = np.full(g["B_images"][k].shape, 1.0 / g["B_sigma"][k] ** 2)
                ol.append(ngmix.Observation(g["B_images"][k], weight=w, jacobian=jac, psf=pobs))
            mb.append(ol)
        tries = int(res["ntry"][o])
        if tries != 1:
            continue
        one = ngmix.fitting.Fitter(model=model, prior=prior).go(obs=mb, guess=res["guess"][o])
        assert one["flags"] == 0
        err = one["pars_err"]
        assert np.all(np.abs(res["pars"][o] - one["pars"]) <= 5e-3 * err)
