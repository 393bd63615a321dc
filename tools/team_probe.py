"""one multi-band 'exp' batch through the lock-step fits (for rocprofv3 counter
passes of lm_advance_team_kernel: tools/pmc_team.sh).
usage: python tools/team_probe.py [nobj] [nband]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from ngmix_amd.batch import GMixBatch  # noqa: E402
from ngmix_amd.lm_batch import LMBatchFitter  # noqa: E402

nobj = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
nband = int(sys.argv[2]) if len(sys.argv) > 2 else 9
ns = nobj * nband
sb, _, pars = bench.make_workload(ns, 1000, "cuda")
rng = np.random.RandomState(7)
guess = np.concatenate([pars[::nband, :5], pars[:, 5].reshape(nobj, nband)], axis=1)
guess = guess * rng.uniform(0.95, 1.05, size=guess.shape)
psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (ns, 1)), "gauss")
sobj = np.repeat(np.arange(nobj), nband)
sband = np.tile(np.arange(nband), nobj)
f = LMBatchFitter("exp")
for _ in range(2):
    res = f.go(sb, guess, psf=psf, stamp_obj=sobj, stamp_band=sband)
torch.cuda.synchronize()
print("n=%d nobj=%d rounds %d flags==0 %.3f" % (5 + nband, nobj, f.rounds_launched,
                                                 float(np.mean(res["flags"] == 0))))
