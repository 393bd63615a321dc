"""64-bit addressing: a batch whose pixel arrays are larger than 4 GiB (the
C2 workload tiled), last stamps against the same stamps in a small batch.
python tools/big_batch_check.py [nstamps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from ngmix_amd.batch import StampBatch, GMixBatch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
dev = torch.device("cuda", 0)
base = 20000
sb0, gm0, _ = bench.make_workload(base, seed=7, device=dev)
reps = n // base
val = sb0.val.repeat(reps)
ierr = sb0.ierr.repeat(reps)
jac = sb0.jac.repeat(reps, 1)
npix = 48 * 48
sb = StampBatch(val, ierr, jac, np.full(n, 48), np.full(n, 48),
                np.arange(n, dtype=np.int64) * npix, True)
gm = GMixBatch(gm0.data.repeat(reps, 1), n, gm0.ngauss)
print("bytes per pixel array: %.2f GiB" % (val.numel() * 8 / 2 ** 30))
ref = sb0.loglike(gm0)[0].cpu().numpy()
out, st = sb.loglike(gm)
out = out.cpu().numpy()
assert int(st.abs().sum()) == 0
for r in (0, reps // 2, reps - 1):
    assert np.array_equal(out[r * base:(r + 1) * base], ref), r
fd0 = sb0.fill_fdiff(gm0)[0].cpu().numpy()
fd = sb.fill_fdiff(gm)[0]
assert np.array_equal(fd[-base * npix:].cpu().numpy(), fd0)
im0 = sb0.render(gm0)[0].cpu().numpy()
im = sb.render(gm)[0]
assert np.array_equal(im[-base * npix:].cpu().numpy(), im0)
print("loglike / fill_fdiff / render of %d stamps: the last block equals the small batch bit for bit" % n)
