"""profiling driver: a few render+loglike steps on the C2 workload
(python tools/prof_pixpass.py [nstamps] [steps])"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
exact = len(sys.argv) > 3 and sys.argv[3] == "exact"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
sb, gm, _ = bench.make_workload(n, seed=1000, device=dev)
image = torch.zeros(sb.total_pix, dtype=torch.float64, device=dev)
out = torch.empty((n, 4), dtype=torch.float64, device=dev)
status = torch.empty(n, dtype=torch.int32, device=dev)
for _ in range(steps):
    sb.render(gm, image=image, fast_exp=True, status=status, exact=exact)
    sb.loglike(gm, out=out, status=status, exact=exact)
torch.cuda.synchronize()
print("done", float(out[:, 0].sum()))
