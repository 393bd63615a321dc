"""batched forward-difference LM ('bdf', 7 parameters, 16 gaussians (x) psf)
on N 48x48 stamps.  python tools/bench_lm_fd.py [n] [model]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ngmix_amd.batch import StampBatch, GMixBatch  # noqa: E402
from ngmix_amd.lm_batch import LMBatchFitter  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
model = sys.argv[2] if len(sys.argv) > 2 else "bdf"
dim, scale, noise = 48, 0.263, 0.01
rng = np.random.RandomState(4)
pars = np.zeros((n, 6))
pars[:, 0:2] = rng.uniform(-0.5, 0.5, size=(n, 2)) * scale
pars[:, 2:4] = rng.normal(scale=0.1, size=(n, 2))
pars[:, 4] = rng.uniform(0.3, 1.0, size=n)
pars[:, 5] = rng.uniform(200.0, 800.0, size=n)
if model == "bdf":
    pars = np.column_stack([pars[:, :5], rng.uniform(0.3, 0.7, size=n), pars[:, 5]])
psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (n, 1)), "gauss")
gm0, _ = GMixBatch.from_pars(pars, model)
gm, _ = gm0.convolve(psf)
cen = (dim - 1) / 2.0
jac = np.array([cen, cen, scale, 0.0, 0.0, scale, scale ** 2, scale])
geom = StampBatch.from_images(np.zeros((1, dim, dim)), None, jac)
dev = gm.device
jt = torch.from_numpy(np.tile(jac, (n, 1))).to(dev)
off = np.arange(n, dtype=np.int64) * dim * dim
geom = StampBatch(None, None, jt, np.full(n, dim), np.full(n, dim), off, True)
truth, _ = geom.render(gm)
gen = torch.Generator(device=dev)
gen.manual_seed(2)
val = truth + noise * torch.randn(truth.shape, generator=gen, device=dev, dtype=torch.float64)
sb = StampBatch(val, torch.full_like(val, 1.0 / noise), jt, np.full(n, dim), np.full(n, dim),
                off, True)
guess = pars * rng.uniform(0.95, 1.05, size=pars.shape)
guess[:, 0:2] = pars[:, 0:2] + rng.uniform(-0.03, 0.03, size=(n, 2))
guess[:, 2:4] = pars[:, 2:4] + rng.uniform(-0.02, 0.02, size=(n, 2))
fitter = LMBatchFitter(model, analytic_jacobian=False)
fitter.go(sb, guess, psf=psf)
torch.cuda.synchronize()
t0 = time.perf_counter()
res = fitter.go(sb, guess, psf=psf)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ok = res["flags"] == 0
print("batched lmdif '%s': %d fits in %.3f s -> %.3g fits/s (loop %.4f s); rounds %d; "
      "flags==0: %d; nfev median %d max %d" % (
          model, n, dt, n / dt, fitter.loop_seconds, fitter.rounds, int(ok.sum()),
          np.median(res["nfev"][ok]), res["nfev"][ok].max()))
