"""the fused loglike / render kernels with every (tile, gaussian) pair skipped
(mixtures centred far outside the stamps): what the kernel's own load / store
structure sustains when instruction issue is out of the way.
python tools/memory_only.py [nstamps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from ngmix_amd.batch import GMixBatch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device("cuda", 0)
sb, gm, pars = bench.make_workload(n, seed=1000, device=dev)
far = pars.copy()
far[:, 0] += 500.0     # v of the centre: hundreds of stamp widths away
g0, _ = GMixBatch.from_pars(far, "exp", device=dev)
psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (n, 1)), "gauss", device=dev)
gfar, _ = g0.convolve(psf)
gfar.set_norms()
out = torch.empty((n, 4), dtype=torch.float64, device=dev)
img = torch.zeros(sb.total_pix, dtype=torch.float64, device=dev)
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


for name, g in (("real mixtures", gm), ("all pairs skipped", gfar)):
    tl = timeit(lambda: sb.loglike(g, out=out, status=status))
    tr = timeit(lambda: sb.render(g, image=img, status=status))
    print("%-18s loglike %.4f ms (%.2f TB/s)   render %.4f ms (%.2f TB/s)" % (
        name, tl, bench.LOGLIKE_BYTES * n / tl / 1e9, tr, bench.RENDER_BYTES * n / tr / 1e9))
