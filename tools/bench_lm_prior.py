"""config 3 with a joint prior: the cost of the prior rows in the lock-step
driver (torch ops between the two launches of every round).
python tools/bench_lm_prior.py [nstamps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from ngmix_amd.batch import GMixBatch  # noqa: E402
from ngmix_amd.lm_batch import LMBatchFitter  # noqa: E402
from ngmix_amd import prior_batch as pb  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
sb, _, pars = bench.make_workload(n, seed=1000, device=dev)
rng = np.random.RandomState(7)
guess = pars * rng.uniform(0.9, 1.1, size=pars.shape)
guess[:, 0:2] = pars[:, 0:2] + rng.uniform(-0.05, 0.05, size=(n, 2))
guess[:, 2:4] = pars[:, 2:4] + rng.uniform(-0.03, 0.03, size=(n, 2))
psfpars = np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (n, 1))
psf, _ = GMixBatch.from_pars(psfpars, "gauss", device=dev)


def prior(bounded):
    bT = (-0.1, 1.0e3) if bounded else None
    bF = (-10.0, None) if bounded else None
    return pb.PriorSimpleSepBatch(
        pb.GaussianCen(0.0, 0.0, 0.263, 0.263), pb.GPriorBA(0.3),
        pb.TwoSidedErf(-0.1, 0.03, 1.0e3, 1.0, bounds=bT),
        pb.TwoSidedErf(-10.0, 1.0, 1.0e6, 1.0e3, bounds=bF))


for label, pr in (("no prior", None), ("PriorSimpleSepBatch", prior(False)),
                  ("PriorSimpleSepBatch + bounds", prior(True))):
    for analytic in (True, False):
        fitter = LMBatchFitter("exp", prior=pr, analytic_jacobian=analytic)
        fitter.go(sb, guess, psf=psf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = fitter.go(sb, guess, psf=psf)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ok = res["flags"] == 0
        print("%-30s %-6s %d fits: %.3f s end to end, loop %.4f s (%.3g fits/s), rounds %d, "
              "ok %d, nfev median %d" % (
                  label, "lmder" if analytic else "lmdif", n, dt, fitter.loop_seconds,
                  n / fitter.loop_seconds, fitter.rounds, int(ok.sum()),
                  np.median(res["nfev"][ok])))
