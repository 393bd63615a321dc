import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from ngmix_amd.batch import GMixBatch
from ngmix_amd.lm_batch import LMBatchFitter
n = 100000
dev = torch.device("cuda", 0)
sb, _, pars = bench.make_workload(n, seed=1000, device=dev)
rng = np.random.RandomState(7)
guess = pars * rng.uniform(0.5, 1.6, size=pars.shape)
guess[:, 0:2] = pars[:, 0:2] + rng.uniform(-0.3, 0.3, size=(n, 2))
guess[:, 2:4] = np.clip(pars[:, 2:4] + rng.uniform(-0.25, 0.25, size=(n, 2)), -0.7, 0.7)
psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (n, 1)), "gauss", device=dev)
f = LMBatchFitter("exp")
f.go(sb, guess, psf=psf)
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = f.go(sb, guess, psf=psf)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
nf = res["nfev"]
print("poor guesses: %.1f ms end to end, loop %.1f ms, rounds %d; nfev percentiles 50/90/99/max: %d %d %d %d; flags==0 %d" % (
    min(ts) * 1e3, f.loop_seconds * 1e3, f.rounds, *np.percentile(nf, [50, 90, 99]), nf.max(), int((res["flags"] == 0).sum())))
h = np.bincount(nf)
alive = n - np.cumsum(h)
print("fits still running after round k:", [int(alive[k]) for k in range(0, len(alive), max(1, len(alive) // 12))])
