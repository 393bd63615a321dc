import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, "/root/repo")
import ngmix_amd as ngmix
rng = np.random.RandomState(1)
jac = ngmix.DiagonalJacobian(row=23.5, col=23.5, scale=0.263)
pgm = ngmix.GMixModel([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], "gauss")
gm = ngmix.GMixModel([0.1, -0.05, 0.1, 0.05, 0.6, 100.0], "exp").convolve(pgm)
im = gm.make_image((48, 48), jacobian=jac, fast_exp=True) + 0.01 * rng.normal(size=(48, 48))
wt = np.full((48, 48), 1e4)
pj = ngmix.DiagonalJacobian(row=12, col=12, scale=0.263)
pobs = ngmix.Observation(pgm.make_image((25, 25), jacobian=pj), jacobian=pj, gmix=pgm)
obs = ngmix.Observation(im, weight=wt, jacobian=jac, psf=pobs)
guess = np.array([0.1, -0.05, 0.1, 0.05, 0.6, 100.0]) * 1.03
fb = ngmix.fitting.Fitter(model="exp", batched=True)
for _ in range(20): fb.go(obs=obs, guess=guess)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): fb.go(obs=obs, guess=guess)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
bf = fb._batch_fitter
print("host_ms", bf.host_ms, "rounds_launched", bf.rounds_launched, "useful", getattr(bf, "useful_rounds", None))
bf.time_kernels = True
for _ in range(5): fb.go(obs=obs, guess=guess)
print("kernel_ms", bf.kernel_ms, "loop_seconds", getattr(bf, "loop_seconds", None))
bf.time_kernels = False
t0 = time.perf_counter()
for _ in range(500): fb.go(obs=obs, guess=guess)
print("Fitter.go: %.1f us" % ((time.perf_counter() - t0) / 500 * 1e6))
