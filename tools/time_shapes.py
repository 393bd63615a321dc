"""loglike / render / fdiff time per stamp for stamp shapes and masks off the
tuned path (48x48 unmasked): odd shapes, masked pixels, ragged batches.
python tools/time_shapes.py [nstamps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ngmix_amd.batch import StampBatch, GMixBatch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
dev = torch.device("cuda", 0)
rng = np.random.RandomState(4)
scale = 0.263


def batch(shapes, mask_frac):
    nrow = np.array([s[0] for s in shapes], dtype=np.int32)
    ncol = np.array([s[1] for s in shapes], dtype=np.int32)
    npix = nrow.astype(np.int64) * ncol
    off = np.concatenate([[0], np.cumsum(npix)[:-1]])
    tot = int(npix.sum())
    jac = np.zeros((len(shapes), 8))
    jac[:, 0] = (nrow - 1) / 2.0
    jac[:, 1] = (ncol - 1) / 2.0
    jac[:, 2] = jac[:, 5] = jac[:, 7] = scale
    jac[:, 6] = scale * scale
    val = torch.randn(tot, dtype=torch.float64, device=dev) * 0.01
    ierr = torch.full((tot,), 100.0, dtype=torch.float64, device=dev)
    if mask_frac > 0:
        m = torch.rand(tot, device=dev) < mask_frac
        ierr[m] = 0.0
    return StampBatch(val, ierr, torch.from_numpy(jac).to(dev), nrow, ncol, off, True)


def pars(m):
    p = np.zeros((m, 6))
    p[:, 0:2] = rng.uniform(-0.5, 0.5, size=(m, 2)) * scale
    p[:, 2:4] = rng.normal(scale=0.1, size=(m, 2)).clip(-0.5, 0.5)
    p[:, 4] = rng.uniform(0.3, 1.5, size=m) + 0.27
    p[:, 5] = rng.uniform(50, 500, size=m)
    return p


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


cases = [
    ("48x48", [(48, 48)] * n, 0.0),
    ("48x48, 1% masked", [(48, 48)] * n, 0.01),
    ("50x50", [(50, 50)] * n, 0.0),
    ("33x33", [(33, 33)] * n, 0.0),
    ("32x32", [(32, 32)] * n, 0.0),
    ("64x64", [(64, 64)] * n, 0.0),
    ("ragged 24..64 (multiples of 8)", [(int(d), int(d)) for d in rng.choice([24, 32, 48, 64], n)], 0.0),
    ("ragged 21..67 (any)", [(int(a), int(b)) for a, b in rng.randint(21, 68, size=(n, 2))], 0.005),
]
for name, shapes, mf in cases:
    sb = batch(shapes, mf)
    gm, _ = GMixBatch.from_pars(pars(len(shapes)), "exp", device=dev)
    image = torch.zeros(sb.total_pix, dtype=torch.float64, device=dev)
    t_l = timeit(lambda: sb.loglike(gm))
    t_r = timeit(lambda: sb.render(gm, image=image))
    t_f = timeit(lambda: sb.fill_fdiff(gm))
    px = sb.total_pix
    print("%-34s %9d px: loglike %.3f ms (%.2f TB/s alg)  render %.3f ms (%.2f)  fdiff %.3f ms (%.2f)" % (
        name, px, t_l * 1e3, 16 * px / t_l / 1e12, t_r * 1e3, 16 * px / t_r / 1e12,
        t_f * 1e3, 24 * px / t_f / 1e12))
