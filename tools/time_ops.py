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

# weighted sums (GaussMom's kernel) and deriv_images on the same stamps
import numpy as np  # noqa: E402
from ngmix_amd.batch import GMixBatch  # noqa: E402
from ngmix_amd import _lib  # noqa: E402

wt, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.6, 1.0], (n, 1)), "gauss",
                            device=dev)
wt.set_norms()
maxrad = np.full(n, 100.0 * np.sqrt(0.3))
res = []
for nmom in (6, 17):
    nd = _lib.moments_result_dtype(nmom).itemsize // 8
    r = torch.zeros((n, nd), dtype=torch.float64, device=dev)
    t = timeit(lambda: sb.weighted_sums(wt, maxrad, nmom=nmom, res=r, status=status))
    nbytes = 16 * 2304 + nd * 8 * 2
    res.append("wsums%d %.4f ms %.2f TB/s" % (nmom, t, nbytes * n / t / 1e9))
gmh = gm.to_numpy()
gpars = np.stack([gmh[k] for k in ("p", "row", "col", "irr", "irc", "icc")], axis=-1)
gpars = np.ascontiguousarray(gpars.reshape(-1, 6))
dcov = np.tile(np.eye(3)[None] * 0.1, (gpars.shape[0], 1, 1))
dgp = torch.from_numpy(gpars).to(dev)
ddc = torch.from_numpy(dcov).to(dev)
dout = torch.zeros(6 * sb.total_pix, dtype=torch.float64, device=dev)
ostart = torch.from_numpy(np.arange(n, dtype=np.int64) * 6 * 2304).to(dev)
t = timeit(lambda: sb.deriv_images(dgp, ddc, 6, out=dout, out_start=ostart), reps=5)
res.append("deriv_images %.4f ms %.2f TB/s" % (t, (6 * 2 * 8 * 2304) * n / t / 1e9))
print(" | ".join(res))
