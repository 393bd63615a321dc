"""The lock-step LM driver against scipy's MINPACK on random fits: each case's
objects (gauss / exp / dev with the analytic jacobian: the reference's lmder
path; the same models and turb by forward differences: its lmdif path; 1-4
bands, 1-2 epochs per band, 1-3 psf gaussians, masked pixels, stamps of 24-48
pixels) are fitted as one batch by LMBatchFitter and one by one by
Fitter(batched=False) -- MINPACK's own lmder (its QR of the jacobian; the
driver iterates on the normal equations) calling the exact seam kernels once
per evaluation.  Counted: fits whose ier / flags agree, whose nfev agree
exactly, and the largest parameter difference in units of the quoted error.

usage: python tools/fuzz_lm_vs_minpack.py [seconds] [seed]   (default 120 s)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("NGMIX_FITTER_BATCHED", "0")
import test_gpu_lm_batch as tb  # noqa: E402
import ngmix_amd as ngmix  # noqa: E402
from ngmix_amd.lm_batch import LMBatchFitter  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
master = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 9)

t0 = time.time()
nfit = nfd = same_ier = same_nfev = both_ok = 0
worst = 0.0
worst_case = None
odd = []
odd_nfev = []
while time.time() - t0 < budget:
    seed = int(master.randint(1 << 30))
    rng = np.random.RandomState(seed)
    model = str(rng.choice(["exp", "gauss", "dev", "turb"]))
    analytic = model != "turb" and rng.uniform() < 0.6
    nband = int(rng.randint(1, 5))
    nobj = int(rng.randint(2, 6))
    npsf = int(rng.randint(1, 4))
    dim = int(rng.choice([24, 32, 33, 40, 48]))
    sobj, sband = [], []
    for o in range(nobj):
        bands = np.concatenate([np.full(rng.randint(1, 3), b) for b in range(nband)])
        sobj += [o] * bands.size
        sband += list(bands)
    sobj = np.array(sobj, dtype=np.int32)
    sband = np.array(sband, dtype=np.int32)
    ns = sobj.size
    psf_rows, psf = tb._multi_gauss_psf(ns, npsf, bool(rng.randint(2)), rng)
    pars, guess, images, weights, jobj, sb = tb._objects_with_psf(ns, model, psf, rng, dim=dim)
    first = np.searchsorted(sobj, np.arange(nobj))
    # the stamps of an object were drawn independently: refit them to one shape
    # (the guess), one flux per band -- both routes see the same problem
    flux = np.stack([guess[first, 5] * rng.uniform(0.9, 1.1, size=nobj) for _ in range(nband)],
                    axis=1)
    g2 = np.concatenate([guess[first, :5], flux], axis=1)
    res = LMBatchFitter(model, analytic_jacobian=analytic).go(sb, g2, psf=psf, stamp_obj=sobj, stamp_band=sband)
    for o in range(nobj):
        mb = ngmix.MultiBandObsList()
        for b in range(nband):
            ol = ngmix.ObsList()
            for s in np.nonzero((sobj == o) & (sband == b))[0]:
                pobs = ngmix.Observation(np.zeros((5, 5)), jacobian=jobj,
                                         gmix=ngmix.GMix(pars=psf_rows[s]))
                ol.append(ngmix.Observation(images[s], weight=weights[s], jacobian=jobj,
                                            psf=pobs))
            mb.append(ol)
        one = ngmix.fitting.Fitter(model=model, analytic_jacobian=analytic,
                                  batched=False).go(obs=mb, guess=g2[o])
        nfit += 1
        nfd += int(not analytic)
        same = int(one["ier"] == res["ier"][o] and one["flags"] == res["flags"][o])
        same_ier += same
        same_nfev += int(one["nfev"] == res["nfev"][o])
        if same and one["nfev"] != res["nfev"][o]:
            odd_nfev.append((seed, o, model, analytic, int(one["nfev"]), int(res["nfev"][o])))
        if not same:
            odd.append((seed, o, model, analytic, int(one["ier"]), int(res["ier"][o]), int(one["flags"]),
                        int(res["flags"][o])))
        if one["flags"] == 0 and res["flags"][o] == 0:
            both_ok += 1
            d = float(np.max(np.abs(res["pars"][o] - one["pars"]) / one["pars_err"]))
            if d > worst:
                worst, worst_case = d, (seed, o, model, nband, analytic)
print("fuzz_lm_vs_minpack: %.0f s, %d fits (%d of them by forward differences); ier and flags agree: %d; nfev agree exactly: %d; "
      "both converged: %d, largest |dpars| / pars_err among them: %.2e %s"
      % (time.time() - t0, nfit, nfd, same_ier, same_nfev, both_ok, worst, worst_case))
for rec in odd[:20]:
    print("   differs: seed %d object %d (%s, analytic %s): ier %d / %d, flags %d / %d (MINPACK / driver)" % rec)
for rec in odd_nfev[:20]:
    print("   nfev differs: seed %d object %d (%s, analytic %s): %d / %d (MINPACK / driver)" % rec)
