"""config 3 as a pipeline: the batch in P pieces, each fitted by its own
LMBatchFitter on its own stream from its own host thread, so that one piece's
packaging / download runs under the other's device loop and the GPU never
idles between calls.  python tools/lm_pipeline.py [nstamps] [pieces]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from ngmix_amd.batch import GMixBatch  # noqa: E402
from ngmix_amd.lm_batch import LMBatchFitter  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
rng = np.random.RandomState(7)
pieces = []
for k in range(P):
    m = n // P
    sb, _, pars = bench.make_workload(m, seed=1000 + k, device=dev)
    guess = pars * rng.uniform(0.9, 1.1, size=pars.shape)
    guess[:, 0:2] = pars[:, 0:2] + rng.uniform(-0.05, 0.05, size=(m, 2))
    guess[:, 2:4] = pars[:, 2:4] + rng.uniform(-0.03, 0.03, size=(m, 2))
    psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (m, 1)), "gauss",
                                 device=dev)
    pieces.append((sb, guess, psf, LMBatchFitter("exp"), torch.cuda.Stream(device=dev)))

steps = 20


def worker(k, out):
    sb, guess, psf, fitter, stream = pieces[k]
    torch.cuda.set_device(dev)
    with torch.cuda.stream(stream):
        for i in range(steps):
            res = fitter.go(sb, guess, psf=psf)
            out[k] = int((res["flags"] != 0).sum())


for rep in range(3):
    out = [None] * P
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(k, out)) for k in range(P)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%d pieces x %d stamps, %d steps: %.3f ms per step of %d fits -> %.3g fits/s (bad %s)" % (
        P, n // P, steps, dt / steps * 1e3, n, n * steps / dt, out))
