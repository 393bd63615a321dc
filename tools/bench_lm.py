"""config 3 of BASELINE.json: LM fits ('exp' model, psf-convolved) of N 48x48
stamps on one GPU -- the batched lock-step driver, and (on a small sample) the
per-object scipy/MINPACK path over the same kernels for comparison.
python tools/bench_lm.py [nstamps] [nsample_per_object_path]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
import ngmix_amd as ngmix  # noqa: E402
from ngmix_amd.batch import GMixBatch  # noqa: E402
from ngmix_amd.lm_batch import LMBatchFitter  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
nsample = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
sb, _, pars = bench.make_workload(n, seed=1000, device=dev)
rng = np.random.RandomState(7)
guess = pars * rng.uniform(0.9, 1.1, size=pars.shape)
guess[:, 0:2] = pars[:, 0:2] + rng.uniform(-0.05, 0.05, size=(n, 2))
guess[:, 2:4] = pars[:, 2:4] + rng.uniform(-0.03, 0.03, size=(n, 2))
psfpars = np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (n, 1))
psf, _ = GMixBatch.from_pars(psfpars, "gauss", device=dev)

fitter = LMBatchFitter("exp")
fitter.go(sb, guess, psf=psf)  # warm up
torch.cuda.synchronize()
t0 = time.perf_counter()
res = fitter.go(sb, guess, psf=psf)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ok = res["flags"] == 0
pull = (res["pars"][ok] - pars[ok]) / res["pars_err"][ok]
print("batched LM: %d fits in %.3f s -> %.3g fits/s (lock-step loop alone %.4f s "
      "-> %.3g fits/s); rounds %d; flags==0: %d; "
      "nfev median %d max %d; |pull| rms %s" % (
          n, dt, n / dt, fitter.loop_seconds, n / fitter.loop_seconds,
          fitter.rounds, int(ok.sum()), np.median(res["nfev"][ok]),
          res["nfev"][ok].max(), np.round(np.sqrt((pull ** 2).mean(axis=0)), 2)))

if nsample > 0:
    val = sb.val.cpu().numpy().reshape(n, 48, 48)
    ierr = sb.ierr.cpu().numpy().reshape(n, 48, 48)
    j = sb.jac[0].cpu().numpy()
    jobj = ngmix.Jacobian(row=j[0], col=j[1], dvdrow=j[2], dvdcol=j[3], dudrow=j[4],
                          dudcol=j[5])
    pgm = ngmix.GMixModel([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], "gauss")
    t0 = time.perf_counter()
    nf = []
    for i in range(nsample):
        pobs = ngmix.Observation(np.zeros((5, 5)), jacobian=jobj, gmix=pgm)
        obs = ngmix.Observation(val[i], weight=ierr[i] ** 2, jacobian=jobj, psf=pobs)
        one = ngmix.fitting.Fitter(model="exp").go(obs=obs, guess=guess[i])
        nf.append(one["nfev"])
        assert abs(one["pars"][4] - res["pars"][i][4]) < 1e-3 * one["pars_err"][4]
    dt1 = (time.perf_counter() - t0) / nsample
    print("per-object Fitter (scipy MINPACK, one kernel launch per evaluation): "
          "%.2f ms per fit -> %.3g fits/s; nfev median %d" % (
              dt1 * 1e3, 1.0 / dt1, np.median(nf)))

# where the time goes
import cProfile  # noqa: E402
import pstats  # noqa: E402
pr = cProfile.Profile()
pr.enable()
fitter.go(sb, guess, psf=psf)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
