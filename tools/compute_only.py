"""how much of the fused loglike / render time is instruction issue: the C2
workload with every stamp aliased onto the pixels of stamp 0 (all loads hit
L2 / L1) against the real layout.  python tools/compute_only.py [nstamps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from ngmix_amd.batch import StampBatch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device("cuda", 0)
sb, gm, _ = bench.make_workload(n, seed=1000, device=dev)
alias = StampBatch(sb.val, sb.ierr, sb.jac, np.full(n, 48), np.full(n, 48),
                   np.zeros(n, dtype=np.int64), True)
out = torch.empty((n, 4), dtype=torch.float64, device=dev)
status = torch.empty(n, dtype=torch.int32, device=dev)


def timeit(fn, reps=200):
    for _ in range(150):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


if len(sys.argv) > 2:
    # counter pass: a few launches of each, real first
    for b in (sb, alias):
        for _ in range(4):
            b.loglike(gm, out=out, status=status)
        torch.cuda.synchronize()
else:
    for name, b in (("real", sb), ("aliased", alias), ("real", sb), ("aliased", alias)):
        t = timeit(lambda: b.loglike(gm, out=out, status=status))
        print("loglike %-8s %.4f ms" % (name, t))
