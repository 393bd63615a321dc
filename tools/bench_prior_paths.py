#!/usr/bin/env python
"""
What a joint prior costs a batch of LM fits, by the path that evaluates its
rows: none / the prior kernel inside the device loop / torch ops with the
rounds driven from the host / the per-object host adapter.  'exp' (lmder) and
'bdf' (lmdif) on N 32x32 stamps.

    python tools/bench_prior_paths.py [N]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import ngmix_amd as ngmix  # noqa: E402
from ngmix_amd import priors, joint_prior, prior_batch as pb  # noqa: E402
from ngmix_amd.batch import StampBatch, GMixBatch  # noqa: E402
from ngmix_amd.lm_batch import LMBatchFitter  # noqa: E402


def scene(model, N, dim=32, scale=0.263, seed=3):
    rng = np.random.RandomState(seed)
    npars = 6 if model == "exp" else 7
    pars = np.zeros((N, npars))
    pars[:, 0:2] = rng.uniform(-0.05, 0.05, size=(N, 2))
    pars[:, 2:4] = rng.uniform(-0.2, 0.2, size=(N, 2))
    pars[:, 4] = rng.uniform(0.4, 0.9, size=N)
    if model == "bdf":
        pars[:, 5] = rng.uniform(0.2, 0.8, size=N)
    pars[:, -1] = rng.uniform(80, 200, size=N)
    psf = ngmix.GMixModel([0.0, 0.0, 0.01, -0.02, 0.27, 1.0], "gauss")
    psfb = GMixBatch.from_numpy(np.tile(psf.get_data(), (N, 1)))
    gmb, _ = GMixBatch.from_pars(pars, model)
    jac = ngmix.DiagonalJacobian(row=15.5, col=15.5, scale=scale)
    sb0 = StampBatch.from_images(np.zeros((N, dim, dim)), None, [jac] * N)
    conv = gmb.convolve(psfb)
    img, _ = sb0.render(conv[0] if isinstance(conv, tuple) else conv)
    sigma = pars[:, -1] / 300.0
    img = img.reshape(N, dim, dim).cpu().numpy() + sigma[:, None, None] * rng.normal(size=(N, dim, dim))
    w = np.ones((N, dim, dim)) / sigma[:, None, None] ** 2
    sb = StampBatch.from_images(img, w, [jac] * N)
    guess = pars.copy()
    guess[:, 4:] *= rng.uniform(0.95, 1.05, size=(N, npars - 4))
    return sb, psfb, guess


def host_prior(model, rng):
    cen = priors.CenPrior(0.0, 0.0, 0.263, 0.263, rng=rng)
    g = priors.GPriorBA(0.3, rng=rng)
    T = priors.TwoSidedErf(-0.1, 0.03, 100.0, 1.0, rng=rng)
    F = priors.TwoSidedErf(-10.0, 1.0, 1.0e5, 100.0, rng=rng)
    if model == "exp":
        return joint_prior.PriorSimpleSep(cen, g, T, F)
    return joint_prior.PriorBDFSep(cen, g, T, priors.Normal(0.5, 0.1, rng=rng, bounds=(0.0, 1.0)), F)


def timed(fitter, sb, guess, psf, reps=3):
    fitter.go(sb, guess, psf=psf)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        res = fitter.go(sb, guess, psf=psf)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best, res


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    for model in ("exp", "bdf"):
        sb, psf, guess = scene(model, N)
        hp = host_prior(model, np.random.RandomState(1))
        paths = [("no prior", dict(prior=None))]
        bp = pb.as_batch_prior(hp)
        if getattr(bp, "descriptor", lambda: None)() is not None:
            paths.append(("prior kernel (device loop)", dict(prior=hp)))
        paths.append(("torch rows (host-driven rounds)", dict(prior=hp, device_prior=False)))
        if N <= 20000:
            paths.append(("host adapter (per object)", dict(prior=pb.PriorBatchAdapter(hp))))
        base = None
        for name, kw in paths:
            t, res = timed(LMBatchFitter(model, **kw), sb, guess, psf)
            ok = int((res["flags"] == 0).sum())
            base = base or t
            print("%-4s %-34s %9.2f ms  %10.3e fits/s  x%.2f  converged %d/%d  mean nfev %.1f" % (
                model, name, t * 1e3, N / t, t / base, ok, N, res["nfev"].mean()), flush=True)


if __name__ == "__main__":
    main()
