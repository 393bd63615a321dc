"""memory floor of the pixel-pass kernels: the C2 workload with every gaussian
moved off the stamp, so each tile is loaded / stored but no pair is evaluated.
python tools/floor_test.py [nstamps]"""
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
out = torch.empty((n, 4), dtype=torch.float64, device=dev)
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


for label in ("normal", "off-stamp"):
    if label == "off-stamp":
        gm.data[:, 1] += 1000.0  # row
    tr = timeit(lambda: sb.render(gm, image=image, fast_exp=True, status=status))
    tl = timeit(lambda: sb.loglike(gm, out=out, status=status))
    print("%-10s render %.4f ms  loglike %.4f ms  (%.2f / %.2f TB/s algorithmic)" % (
        label, tr, tl, 36864 * n / tr / 1e9, 37008 * n / tl / 1e9))
