"""the per-object reference API on the GPU: where one Fitter.go spends its time
(python tools/bench_fitter.py [nfits])"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngmix_amd as ngmix  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.RandomState(3)
scale = 0.263
jac = ngmix.DiagonalJacobian(row=23.5, col=23.5, scale=scale)
pgm = ngmix.GMixModel([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], "gauss")
truth = np.array([0.02, -0.03, 0.1, -0.05, 0.6, 100.0])
gm = ngmix.GMixModel(truth, "exp").convolve(pgm)
obs_list = []
for i in range(n):
    im = gm.make_image((48, 48), jacobian=jac, fast_exp=True) + 0.01 * rng.normal(size=(48, 48))
    pobs = ngmix.Observation(np.zeros((5, 5)), jacobian=jac, gmix=pgm)
    obs_list.append(ngmix.Observation(im, weight=np.full((48, 48), 1e4), jacobian=jac, psf=pobs))
guess = truth * (1.0 + 0.05 * rng.uniform(-1, 1, size=6))
fitter = ngmix.fitting.Fitter(model="exp")
fitter.go(obs=obs_list[0], guess=guess)
t0 = time.perf_counter()
nf = []
for obs in obs_list:
    res = fitter.go(obs=obs, guess=guess)
    nf.append(res["nfev"])
dt = (time.perf_counter() - t0) / n
print("Fitter.go: %.2f ms per fit (nfev median %d) -> %.0f fits/s" % (dt * 1e3, np.median(nf), 1 / dt))
pr = cProfile.Profile()
pr.enable()
for obs in obs_list[:20]:
    fitter.go(obs=obs, guess=guess)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
