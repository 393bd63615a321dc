"""end-to-end batched bootstrap (psf admom -> guess admom -> LM) on N objects:
48x48 'exp' (x) gaussian psf stamps + 25x25 psf stamps, everything in HBM.
python tools/bench_boot.py [n]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from ngmix_amd.batch import StampBatch, GMixBatch  # noqa: E402
from ngmix_amd.pipeline import bootstrap_batch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
sb, _, pars = bench.make_workload(n, seed=1000, device=dev)
scale, pdim = bench.SCALE, 25
psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (n, 1)), "gauss",
                             device=dev)
pjac = np.array([12.0, 12.0, scale, 0.0, 0.0, scale, scale ** 2, scale])
pj = torch.from_numpy(np.tile(pjac, (n, 1))).to(dev)
off = np.arange(n, dtype=np.int64) * pdim * pdim
geom = StampBatch(None, None, pj, np.full(n, pdim), np.full(n, pdim), off, True)
pim, _ = geom.render(psf)
gen = torch.Generator(device=dev)
gen.manual_seed(3)
pim += 1e-5 * torch.randn(pim.shape, generator=gen, device=dev, dtype=torch.float64)
psb = StampBatch(pim, torch.full_like(pim, 1e5), pj, np.full(n, pdim), np.full(n, pdim),
                 off, True)
bootstrap_batch(sb, psb, model="exp")
ts = []
for _ in range(7):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = bootstrap_batch(sb, psb, model="exp")
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
dt = min(ts)
print("seconds per call:", " ".join("%.3f" % t for t in ts))
ok = res["flags"] == 0
pull = (res["pars"][ok] - pars[ok]) / res["pars_err"][ok]
print("bootstrap_batch: %d objects in %.3f s -> %.3g objects/s; LM rounds %d; flags==0 %d; "
      "psf failures %d; guess failures %d; pull rms %s" % (
          n, dt, n / dt, res["rounds"], int(ok.sum()), int((res["psf_flags"] != 0).sum()),
          int((res["guess_flags"] != 0).sum()),
          np.round(np.sqrt((pull ** 2).mean(axis=0)), 2)))

if len(sys.argv) > 2:
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    bootstrap_batch(sb, psb, model="exp")
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
