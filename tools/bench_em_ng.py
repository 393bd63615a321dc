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

# ---- config-4-like stamps (32 x 32, a round-ish gaussian (x) psf, 40
# iterations): objects/s by (object gaussians, psf gaussians) -- the psf with
# one gaussian (compile-time in the kernel) or a 3-gaussian 'turb' mixture
n = 100000
w = bench.make_c4(n, 5, dev)
sb_em = w["sb_em"]
turb, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (n, 1)), "turb",
                              device=dev)
base = None
for ng, psfname, psf_gm in ((1, "1", w["psf"]), (1, "3 (turb)", turb), (2, "1", w["psf"]),
                            (2, "3 (turb)", turb), (3, "1", w["psf"]), (3, "3 (turb)", turb),
                            (4, "1", w["psf"]), (6, "1", w["psf"])):
    full = np.zeros((n, ng, 6))
    for i in range(ng):
        full[:, i, 0] = 100.0 * bench.SCALE ** 2 / ng
        full[:, i, 1] = full[:, i, 2] = 0.01 * (i - 0.5 * (ng - 1))
        full[:, i, 3] = full[:, i, 5] = 0.25 * (1 + 0.6 * i) * rng.uniform(0.9, 1.1, size=n)
    ts = []
    for rep in range(3):
        gm, _ = GMixBatch.from_pars(full.reshape(n, -1), "full", device=dev, ngauss=ng)
        conv, _ = gm.convolve(psf_gm)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out, st, _ = sb_em.em(gm, psf_gm, conv=conv, sky=w["sky"], miniter=40, maxiter=40,
                              tol=1e-30)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    rate = n / min(ts)
    if base is None:
        base = rate
    print("ngauss %d, psf gaussians %s: %.2f ms per %d stamps x 40 iterations = %.3g objects/s "
          "(%.2f of the 1 x 1 rate; status != 0: %d)" % (
              ng, psfname, min(ts) * 1e3, n, rate, rate / base, int((st != 0).sum())))
