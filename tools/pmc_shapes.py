"""instruction counts of the fused loglike kernel as a function of the stamp
shape (prologue vs per-tile vs per-pair cost): loglike on C2-like stamps of
several shapes, `reps` launches each, in a fixed order.
  python tools/pmc_shapes.py [nstamps] [reps]        (under rocprofv3 --pmc)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ngmix_amd.batch import StampBatch, GMixBatch  # noqa: E402

SHAPES = [(8, 8), (8, 16), (16, 16), (16, 32), (32, 32), (48, 48)]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = torch.device("cuda", 0)
    rng = np.random.RandomState(1)
    scale = 0.263
    pars = np.zeros((n, 6))
    pars[:, 0:2] = rng.uniform(-0.5, 0.5, size=(n, 2)) * scale
    pars[:, 2:4] = rng.normal(scale=0.1, size=(n, 2))
    pars[:, 4] = rng.uniform(0.3, 1.5, size=n)
    pars[:, 5] = rng.uniform(50.0, 500.0, size=n)
    gm0, _ = GMixBatch.from_pars(pars, "exp", device=dev)
    psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (n, 1)), "gauss",
                                 device=dev)
    gm, _ = gm0.convolve(psf)
    gm.set_norms()
    for (nr, nc) in SHAPES:
        jac = np.array([(nr - 1) / 2, (nc - 1) / 2, scale, 0.0, 0.0, scale, scale ** 2, scale])
        val = torch.randn(n * nr * nc, dtype=torch.float64, device=dev)
        ierr = torch.ones_like(val)
        sb = StampBatch(val, ierr, torch.from_numpy(np.tile(jac, (n, 1))).to(dev),
                        np.full(n, nr), np.full(n, nc),
                        np.arange(n, dtype=np.int64) * nr * nc, True)
        for _ in range(reps):
            out, st = sb.loglike(gm, no_skip=bool(int(os.environ.get('NO_SKIP', '0'))))
        torch.cuda.synchronize()
        print("shape %dx%d ok %d" % (nr, nc, int(st.abs().sum())))


if __name__ == "__main__":
    main()
