"""bootstrap_batch with the different psf fitters (adaptive moments only, EM with
2 / 3 gaussians, co-elliptical LM with 2 / 3) on a turbulent-psf workload.
python tools/bench_boot_psf.py [only]   (only: index of the one variant to run)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from ngmix_amd.batch import StampBatch, GMixBatch
from ngmix_amd.pipeline import bootstrap_batch
n = 50000
dev = torch.device("cuda", 0)
sb, _, pars = bench.make_workload(n, seed=1000, device=dev)
scale, pdim = bench.SCALE, 25
psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.02, -0.01, 0.4, 1.0], (n, 1)), "turb", device=dev)  # a realistic (Kolmogorov-like) psf: three gaussians
pjac = np.array([12.0, 12.0, scale, 0.0, 0.0, scale, scale ** 2, scale])
pj = torch.from_numpy(np.tile(pjac, (n, 1))).to(dev)
off = np.arange(n, dtype=np.int64) * pdim * pdim
geom = StampBatch(None, None, pj, np.full(n, pdim), np.full(n, pdim), off, True)
pim, _ = geom.render(psf)
gen = torch.Generator(device=dev); gen.manual_seed(3)
pim += 1e-5 * torch.randn(pim.shape, generator=gen, device=dev, dtype=torch.float64)
psb = StampBatch(pim, torch.full_like(pim, 1e5), pj, np.full(n, pdim), np.full(n, pdim), off, True)
variants = ({}, {"psf_ngauss": 2}, {"psf_ngauss": 3}, {"psf_ngauss": 2, "psf_fitter": "coellip"},
            {"psf_ngauss": 3, "psf_fitter": "coellip"})
if len(sys.argv) > 1:
    variants = (variants[int(sys.argv[1])],)
for kw in variants:
    bootstrap_batch(sb, psb, model="exp", **kw)
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = bootstrap_batch(sb, psb, model="exp", **kw)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(kw, "%.1f ms per %d objects; flags==0 %d" % (min(ts) * 1e3, n, int((res["flags"] == 0).sum())))
