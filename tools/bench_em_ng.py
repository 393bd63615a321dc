"""em_run time against the number of gaussians on 25x25 psf stamps, fixed 100
iterations: the per-iteration cost that is not per-gaussian pixel work (the
M-step on lane 0, reductions).  python tools/bench_em_ng.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from ngmix_amd.batch import StampBatch, GMixBatch
n, pdim, scale = 50000, 25, bench.SCALE
dev = torch.device("cuda", 0)
psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.02, -0.01, 0.4, 1.0], (n, 1)), "turb", device=dev)
pjac = np.array([12.0, 12.0, scale, 0.0, 0.0, scale, scale ** 2, scale])
pj = torch.from_numpy(np.tile(pjac, (n, 1))).to(dev)
off = np.arange(n, dtype=np.int64) * pdim * pdim
geom = StampBatch(None, None, pj, np.full(n, pdim), np.full(n, pdim), off, True)
pim, _ = geom.render(psf)
pim = pim + 0.001
psb = StampBatch(pim, torch.full_like(pim, 1e5), pj, np.full(n, pdim), np.full(n, pdim), off, True)
rng = np.random.RandomState(1)
delta = np.zeros((n, 6)); delta[:, 5] = 1.0
nopsf, _ = GMixBatch.from_pars(delta, "gauss", device=dev)
for ng in (1, 2, 3):
    full = np.zeros((n, ng, 6))
    for i in range(ng):
        full[:, i, 0] = 1.0 / ng
        full[:, i, 3] = full[:, i, 5] = 0.2 * (1 + i) * rng.uniform(0.9, 1.1, size=n)
    ts = []
    for rep in range(3):
        gm, _ = GMixBatch.from_pars(full.reshape(n, -1), "full", device=dev, ngauss=ng)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out, st, _ = psb.em(gm, nopsf, sky=0.001, miniter=100, maxiter=100, tol=1e-30)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("ngauss %d: %.2f ms for %d stamps x 100 iterations (status!=0: %d)" % (ng, min(ts) * 1e3, n, int((st != 0).sum())))
