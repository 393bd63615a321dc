#!/usr/bin/env python
"""one configuration of tools/bench_prior_paths.py, for rocprofv3:
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_prior -- python3 tools/prof_prior.py exp 100000"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from bench_prior_paths import scene, host_prior, timed  # noqa: E402
from ngmix_amd.lm_batch import LMBatchFitter  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "exp"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
sb, psf, guess = scene(model, N)
f = LMBatchFitter(model, prior=host_prior(model, np.random.RandomState(1)))
t, res = timed(f, sb, guess, psf, reps=5)
print(model, N, "%.2f ms" % (t * 1e3), f.prior_path)
