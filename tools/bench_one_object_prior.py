#!/usr/bin/env python
"""the per-object interface with a joint prior: Fitter.go on one Observation, the lock-step route (prior kernel)
against the MINPACK route (prior.fill_fdiff on the host per evaluation), 'exp' and 'bdf'"""
import os
import sys
import time

sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import ngmix_amd as ngmix  # noqa: E402
from ngmix_amd import priors, joint_prior  # noqa: E402
from test_gpu_fitter_routes import _object, SCALE  # noqa: E402

for model in ("exp", "bdf"):
    rng = np.random.RandomState(3)
    obs, truth = _object(model, rng, noise=0.02)
    guess = truth * 1.02
    prng = np.random.RandomState(1)
    cen = priors.CenPrior(0.0, 0.0, SCALE, SCALE, rng=prng)
    g = priors.GPriorBA(0.3, rng=prng)
    T = priors.TwoSidedErf(-0.1, 0.03, 100.0, 1.0, rng=prng)
    F = priors.TwoSidedErf(-10.0, 1.0, 1.0e5, 100.0, rng=prng)
    if model == "bdf":
        prior = joint_prior.PriorBDFSep(cen, g, T, priors.Normal(0.5, 0.1, rng=prng, bounds=(0.0, 1.0)), F)
    else:
        prior = joint_prior.PriorSimpleSep(cen, g, T, F)
    for batched in (True, False):
        f = ngmix.fitting.Fitter(model=model, prior=prior, batched=batched)
        for _ in range(5):
            r = f.go(obs=obs, guess=guess)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            r = f.go(obs=obs, guess=guess)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n * 1e3
        print("%-4s %-28s %7.3f ms per fit  nfev %d flags %d" % (
            model, "lock-step route (prior kernel)" if batched else "MINPACK route (host prior)", dt,
            r["nfev"], r["flags"]), flush=True)
