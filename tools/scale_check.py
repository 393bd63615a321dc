"""how long the GPU must be busy before the pixel-pass kernels run at their
sustained rate (the clock governor's ramp), and the sustained per-kernel times
on the C2 workload.  python tools/scale_test.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, reps):
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for n in (25000, 100000):
    sb, gm, _ = bench.make_workload(n, seed=1000, device=dev)
    out = torch.empty((n, 4), dtype=torch.float64, device=dev)
    st = torch.empty(n, dtype=torch.int32, device=dev)
    for reps in (5, 40, 200):
        torch.cuda.synchronize()
        time.sleep(0.5)  # let the clocks fall back
        t = timed(lambda: sb.loglike(gm, out=out, status=st), reps)
        print("loglike %6d stamps x %3d launches from idle: %.4f ms  %.2f TB/s" % (
            n, reps, t, 37008 * n / t / 1e9))

n = 100000
image = torch.zeros(sb.total_pix, dtype=torch.float64, device=dev)
fdiff = torch.zeros(sb.total_pix, dtype=torch.float64, device=dev)
ops = [("loglike", lambda: sb.loglike(gm, out=out, status=st), 37008),
       ("render", lambda: sb.render(gm, image=image, status=st), 36864),
       ("fill_fdiff", lambda: sb.fill_fdiff(gm, fdiff=fdiff, status=st), 55440),
       ("model_s2n_sum", lambda: sb.model_s2n_sum(gm, status=st), 18432 + 144)]
for name, fn, nbytes in ops:
    timed(fn, 150)
    t = timed(fn, 200)
    print("sustained %-14s %.4f ms per 100k stamps  %.3g stamps/s  %.2f TB/s algorithmic" % (
        name, t, n / (t * 1e-3), nbytes * n / t / 1e9))
