#!/usr/bin/env python
"""PGaussMom over a catalogue: N 33x33 stamps + psf stamps as ONE batch of the device (measure_arrays: the padded
FFTs, the deconvolution and the kernel sums), and through go_many (Observation objects in, result dicts out)
    python tools/bench_prepsfmom.py [N]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import ngmix_amd as ngmix  # noqa: E402
from ngmix_amd.prepsfmom import PGaussMom, KSigmaMom  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dim = 33
rng = np.random.RandomState(1)
jac = ngmix.DiagonalJacobian(row=16.0, col=16.0, scale=0.2)
psf_gm = ngmix.GMixModel([0.0, 0.0, 0.02, -0.01, 0.3, 1.0], "turb")
gm = ngmix.GMixModel([0.0, 0.0, 0.15, -0.1, 0.5, 50.0], "exp").convolve(psf_gm)
base = gm.make_image((dim, dim), jacobian=jac)
pbase = psf_gm.make_image((dim, dim), jacobian=jac)
images = base[None] + 0.02 * rng.normal(size=(n, dim, dim))
pimages = pbase[None] + 1e-4 * rng.normal(size=(n, dim, dim))
weights = np.full((n, dim, dim), 1 / 0.02 ** 2)
cen = np.tile([16.0, 16.0], (n, 1)) + rng.uniform(-0.3, 0.3, size=(n, 2))
deriv = (0.2, 0.0, 0.0, 0.2)
d_images, d_weights, d_pimages = (torch.from_numpy(a).cuda() for a in (images, weights, pimages))
for cls, fwhm in ((PGaussMom, 1.2), (KSigmaMom, 2.0)):
    f = cls(fwhm)
    for label, args in (("host arrays", (images, weights, cen, deriv, pimages, cen)),
                        ("stamps in HBM", (d_images, d_weights, cen, deriv, d_pimages, cen))):
        f.measure_arrays(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            mom, cov, _, D = f.measure_arrays(*args)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print("%-10s measure_arrays, %-13s: %d stamps %dx%d padded to %d: %6.1f ms  %.3e stamps/s  "
              "(flux %.3f +- %.3f)" % (cls.__name__, label, n, dim, dim, D, dt * 1e3, n / dt,
                                        mom[:, 5].mean(), mom[:, 5].std()), flush=True)
m = min(n, 2000)
obs = [ngmix.Observation(images[i], weight=weights[i],
                         jacobian=ngmix.DiagonalJacobian(row=cen[i, 0], col=cen[i, 1], scale=0.2),
                         psf=ngmix.Observation(pimages[i], jacobian=ngmix.DiagonalJacobian(
                             row=cen[i, 0], col=cen[i, 1], scale=0.2))) for i in range(m)]
f = PGaussMom(1.2)
t0 = time.perf_counter()
res = f.go_many(obs)
dt = time.perf_counter() - t0
print("PGaussMom  go_many on %d Observations -> result dicts: %.1f ms  %.3e stamps/s" % (m, dt * 1e3, m / dt))
t0 = time.perf_counter()
for o in obs[:200]:
    f.go(o)
dt = time.perf_counter() - t0
print("PGaussMom  go, one Observation at a time: %.2f ms per stamp" % (dt / 200 * 1e3))
