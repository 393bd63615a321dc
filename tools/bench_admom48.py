"""adaptive moments on 48x48 stamps (the north-star stamp: the guess stage of
the bootstrap): threads per stamp A/B through NGMIX_ADMOM_NT.
python tools/bench_admom48.py [n] [dim]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 48
w = bench.make_c4(n, 5, "cuda", dim=dim)
ts = []
for r in range(5):
    wt = w["wt0"].clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res, st = w["sb"].admom(wt)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
rec = res.cpu().numpy() if hasattr(res, "cpu") else res
print("NGMIX_ADMOM_NT=%s: admom %dx%d, %d stamps: %.3f ms (best of 5); status!=0: %d" % (
    os.environ.get("NGMIX_ADMOM_NT", "default"), dim, dim, n, min(ts) * 1e3, int((st != 0).sum())))
