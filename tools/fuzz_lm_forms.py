"""A differential fuzz of the lock-step LM driver: random fits -- model (gauss /
exp / dev with the analytic jacobian or forward differences; turb, bdf by
forward differences), 1-9 bands, 1-3 epochs per band in any order, 1-3 psf
gaussians (co-centred or not), stamp sizes 24-48 with masked pixels, guesses
from good to bad, for up to three bands half of the time the separable prior
with random terms and bounds -- run through the form of the lmder step the launcher picks
(registers for 6-8 parameters, the team form for 9-14) and through the generic
one-thread form: flags, nfev, njev, ier, parameters, covariance and lnprob must
be the same TO THE BIT.  A failure prints its case seed.

usage: python tools/fuzz_lm_forms.py [seconds] [seed]   (default 120 s)"""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import test_gpu_lm_batch as tb  # noqa: E402
from ngmix_amd import _lib  # noqa: E402
from ngmix_amd.lm_batch import LMBatchFitter  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
master = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
KEYS = ("flags", "nfev", "njev", "ier", "pars", "pars_cov", "lnprob")


def one_case(seed):
    rng = np.random.RandomState(seed)
    model = str(rng.choice(["exp", "gauss", "dev", "turb", "bdf"]))
    analytic = model in ("exp", "gauss", "dev") and rng.uniform() < 0.7
    nshape = 6 if model == "bdf" else 5
    nband = int(rng.randint(1, 14 - nshape + 1))
    nobj = int(rng.randint(3, 40))
    npsf = int(rng.randint(1, 4))
    dim = int(rng.choice([24, 25, 32, 33, 40, 48]))
    # stamps: every object has 1..3 epochs per band, in a random order
    sobj, sband = [], []
    for o in range(nobj):
        bands = np.concatenate([np.full(rng.randint(1, 4), b) for b in range(nband)])
        if rng.uniform() < 0.5:
            bands = rng.permutation(bands)
        sobj += [o] * bands.size
        sband += list(bands)
    sobj = np.array(sobj, dtype=np.int32)
    sband = np.array(sband, dtype=np.int32)
    ns = sobj.size
    psf_rows, psf = tb._multi_gauss_psf(ns, npsf, bool(rng.randint(2)), rng)
    extra = rng.uniform(0.3, 0.7, size=ns) if model == "bdf" else None
    pars, guess, images, weights, jobj, sb = tb._objects_with_psf(ns, model, psf, rng, dim=dim,
                                                                 extra=extra)
    if rng.uniform() < 0.5:
        from ngmix_amd.batch import StampBatch
        weights[rng.uniform(size=weights.shape) < 0.02] = 0.0
        cen = (dim - 1) / 2.0
        jac = np.array([cen, cen, 0.263, 0.0, 0.0, 0.263, 0.263 ** 2, 0.263])
        sb = StampBatch.from_images(images, weights, jac)
    first = np.searchsorted(sobj, np.arange(nobj))
    flux = np.stack([guess[first, -1] * rng.uniform(0.8, 1.2, size=nobj) for _ in range(nband)],
                    axis=1)
    g2 = np.concatenate([guess[first, :nshape], flux], axis=1)
    bad = rng.uniform(size=nobj) < 0.3
    g2[bad, 4] *= rng.uniform(0.4, 2.5, size=int(bad.sum()))
    g2[rng.uniform(size=nobj) < 0.1, 2:4] = rng.uniform(-0.7, 0.7, size=2)
    pars_lm = {"maxfev": int(rng.choice([40, 200, 4000])), "ftol": 1e-5, "xtol": 1e-5}
    kw = dict(psf=psf, stamp_obj=sobj, stamp_band=sband)
    prior = None
    if nband <= 3 and model != "bdf" and rng.uniform() < 0.5:
        # the separable prior of the reference (its rows join the normal
        # equations; the kernel form holds three bands), with random terms and,
        # half of the time, leastsqbound's bounds on T and the fluxes
        from ngmix_amd import prior_batch as pb
        bounded = rng.uniform() < 0.5

        def term(lo, hi, centre, width):
            kind = int(rng.randint(3))
            b = None
            if bounded:
                b = [(lo, hi), (lo, None), (None, hi)][int(rng.randint(3))]
            if kind == 0:
                return pb.Flat(lo, hi, bounds=b)
            if kind == 1:
                return pb.TwoSidedErf(lo, 0.1 * width, hi, width, bounds=b)
            return pb.Normal(centre, width, bounds=b)
        prior = pb.PriorSimpleSepBatch(
            pb.GaussianCen(0.0, 0.0, float(rng.uniform(0.05, 0.5)), float(rng.uniform(0.05, 0.5))),
            pb.GPriorBA(float(rng.uniform(0.1, 0.5))),
            term(-1.0, 50.0, 0.6, 1.0),
            [term(-100.0, 1.0e5, 150.0, 300.0) for _ in range(nband)])
        kw_fit = dict(prior=prior)
    else:
        kw_fit = {}
    picked = LMBatchFitter(model, analytic_jacobian=analytic, fit_pars=pars_lm, **kw_fit)
    # (half of the small fits through the team form too: priors and bounds only
    # exist for up to three bands, which the launcher gives to the register form)
    os.environ.pop("NGMIX_LM_TEAM_MIN", None)
    if rng.uniform() < 0.5:
        os.environ["NGMIX_LM_TEAM_MIN"] = "6"
    _lib.launch_census(reset=True)
    a = picked.go(sb, g2, **kw)
    seen = _lib.launch_census(reset=True)
    os.environ.pop("NGMIX_LM_TEAM_MIN", None)
    form = [k for k in seen if k.startswith("lm_advance")]
    generic = LMBatchFitter(model, analytic_jacobian=analytic, fit_pars=pars_lm, **kw_fit)
    generic.advance_hint = False
    b = generic.go(sb, g2, **kw)
    for k in KEYS:
        if not np.array_equal(a[k], b[k], equal_nan=True):
            raise AssertionError("%s differs (model %s, n %d, %s)" % (k, model, g2.shape[1], form))
    return (g2.shape[1], (form[0] if form else "?") + (" +prior" if prior is not None else ""),
            float(np.mean(a["flags"] == 0)), picked.rounds)


t0 = time.time()
ncase, by_n, failures, conv, rounds, forms = 0, {}, [], [], [], {}
while time.time() - t0 < budget:
    seed = int(master.randint(1 << 30))
    try:
        n, form, ok, nr = one_case(seed)
        by_n[n] = by_n.get(n, 0) + 1
        forms[form.split('(')[0]] = forms.get(form.split('(')[0], 0) + 1
        conv.append(ok)
        rounds.append(nr)
        ncase += 1
    except Exception:
        failures.append((seed, traceback.format_exc(limit=2)))
        print("FAIL seed", seed)
        print(failures[-1][1])
        sys.stdout.flush()
        if len(failures) >= 10:
            break
print("fuzz_lm_forms: %.0f s, %d cases (by parameter count: %s), failures %d"
      % (time.time() - t0, ncase, dict(sorted(by_n.items())), len(failures)))
print("   forms of the step picked: %s" % dict(sorted(forms.items())))
print("   fits with flags == 0: mean %.2f of a case's fits; lock-step rounds per case: median %d, "
      "max %d" % (np.mean(conv), int(np.median(rounds)), int(np.max(rounds))))
sys.exit(1 if failures else 0)
