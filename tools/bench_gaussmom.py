"""GaussMomBatch end to end (weighted sums kernel + the statistics of
make_mom_result) on the C2 stamps.  python tools/bench_gaussmom.py [n]"""
import os
import sys
import time
import cProfile
import pstats
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
import ngmix_amd as ngmix  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
sb, _, pars = bench.make_workload(n, seed=1000, device=torch.device("cuda", 0))
fitter = ngmix.GaussMomBatch(fwhm=1.2)
fitter.go(sb)
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = fitter.go(sb)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print("GaussMomBatch.go: %.2f ms per %d stamps -> %.3g stamps/s; flags==0: %d" % (
    min(ts) * 1e3, n, n / min(ts), int((res["flags"] == 0).sum())))
pr = cProfile.Profile()
pr.enable()
fitter.go(sb)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
