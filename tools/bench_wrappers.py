"""end-to-end times of the batch wrappers on the C2 stamps (host packaging
included): PSFFluxBatch, GaussMomBatch (6 / 17 moments), StampBatch.from_images,
StampBatch.select.  python tools/bench_wrappers.py [n]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
import ngmix_amd as ngmix  # noqa: E402
from ngmix_amd.batch import StampBatch, GMixBatch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device("cuda", 0)
sb, gm, pars = bench.make_workload(n, seed=1000, device=dev)
psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (n, 1)), "gauss", device=dev)


def best(fn, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


print("PSFFluxBatch.go          %8.1f ms per %d stamps" % (best(lambda: ngmix.PSFFluxBatch().go(sb, psf)), n))
print("GaussMomBatch.go (6)     %8.1f ms" % best(lambda: ngmix.GaussMomBatch(fwhm=1.2).go(sb)))
print("GaussMomBatch.go (17)    %8.1f ms" % best(lambda: ngmix.GaussMomBatch(fwhm=1.2, with_higher_order=True).go(sb)))
m = min(n, 20000)
images = np.random.RandomState(0).normal(size=(m, 48, 48))
weights = np.ones((m, 48, 48))
jac = np.array([23.5, 23.5, 0.263, 0.0, 0.0, 0.263, 0.263 ** 2, 0.263])
print("StampBatch.from_images   %8.1f ms per %d stamps (host arrays -> HBM)" % (best(lambda: StampBatch.from_images(images, weights, jac)), m))
idx = np.random.RandomState(1).choice(n, size=n // 10, replace=False)
print("StampBatch.select        %8.1f ms per %d of %d stamps" % (best(lambda: sb.select(idx)), idx.size, n))
