#!/usr/bin/env python
"""PGaussMom.measure_arrays on stamps in HBM, for rocprofv3 --kernel-trace --stats"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from ngmix_amd.prepsfmom import PGaussMom  # noqa: E402

n, dim = (int(sys.argv[1]) if len(sys.argv) > 1 else 10000), 33
rng = np.random.RandomState(1)
images = torch.from_numpy(rng.normal(size=(n, dim, dim))).cuda()
weights = torch.full((n, dim, dim), 2500.0, dtype=torch.float64).cuda()
pim = np.exp(-0.5 * ((np.arange(dim) - 16.0)[:, None] ** 2 + (np.arange(dim) - 16.0)[None, :] ** 2) / 4.0)
pimages = torch.from_numpy(np.tile(pim / pim.sum(), (n, 1, 1))).cuda()
cen = np.tile([16.0, 16.0], (n, 1)) + rng.uniform(-0.3, 0.3, size=(n, 2))
f = PGaussMom(1.2)
for _ in range(6):
    f.measure_arrays(images, weights, cen, (0.2, 0.0, 0.0, 0.2), pimages, cen)
torch.cuda.synchronize()
