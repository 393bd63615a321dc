import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bench
from ngmix_amd.batch import StampBatch, GMixBatch
dev = torch.device("cuda", 0)
for n in (25000, 50000, 100000, 200000):
    sb, gm, _ = bench.make_workload(n, seed=1000, device=dev)
    out = torch.empty((n, 4), dtype=torch.float64, device=dev)
    st = torch.empty(n, dtype=torch.int32, device=dev)
    for reps in (5, 40, 200):
        sb.loglike(gm, out=out, status=st); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            sb.loglike(gm, out=out, status=st)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps
        print(n, reps, "%.4f ms  %.1f ns/stamp  %.2f TB/s" % (t, t * 1e6 / n, 37008 * n / t / 1e9))
    del sb, gm
