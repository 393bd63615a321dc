"""time the four batched pixel-pass operations on the C2 workload
python tools/time_ops.py [nstamps]   (NGMIX_HIP_LIB selects the build)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
sb, gm, _ = bench.make_workload(n, seed=1000, device=dev)
image = torch.zeros(sb.total_pix, dtype=torch.float64, device=dev)
fdiff = torch.zeros(sb.total_pix, dtype=torch.float64, device=dev)
out = torch.empty((n, 4), dtype=torch.float64, device=dev)
s2n = torch.empty(n, dtype=torch.float64, device=dev)
status = torch.empty(n, dtype=torch.int32, device=dev)


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ops = [
    ("render", 36864, lambda: sb.render(gm, image=image, fast_exp=True, status=status)),
    ("loglike", 37008, lambda: sb.loglike(gm, out=out, status=status)),
    ("fdiff", 55440, lambda: sb.fill_fdiff(gm, fdiff=fdiff, status=status)),
    ("s2n", 18576, lambda: sb.model_s2n_sum(gm, out=s2n, status=status)),
]
res = []
for name, nbytes, fn in ops:
    t = timeit(fn)
    res.append("%s %.4f ms %.2f TB/s" % (name, t, nbytes * n / t / 1e9))
print(os.environ.get("NGMIX_HIP_LIB", "default")[-24:], " | ".join(res))
