"""lm_advance per launch for n = 6..14 parameters (an "exp" fit over 1..9
bands), each form of the step A/B'd in one process: the register form (6..10),
the team form with 4 / 2 / 1 fits per wave (NGMIX_LM_TEAM_MIN / NGMIX_LM_TEAMS
are read at every launch), the generic one-thread form.  HIP-event time of
the advance launches and of the whole fit.
usage: python tools/lm_advance_sweep.py [nobj] [bands ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from ngmix_amd.batch import GMixBatch  # noqa: E402
from ngmix_amd.lm_batch import LMBatchFitter  # noqa: E402

nobj = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
bands = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4, 5, 6, 8, 9]
FORMS = [("default", {}, True), ("team x4", {"NGMIX_LM_TEAM_MIN": "6", "NGMIX_LM_TEAMS": "4"}, True),
         ("team x2", {"NGMIX_LM_TEAM_MIN": "6", "NGMIX_LM_TEAMS": "2"}, True),
         ("team x1", {"NGMIX_LM_TEAM_MIN": "6", "NGMIX_LM_TEAMS": "1"}, True),
         ("generic", {}, False)]
for nband in bands:
    ns = nobj * nband
    sb, _, pars = bench.make_workload(ns, 1000, "cuda")
    rng = np.random.RandomState(7)
    # every band of an object shows the object's first stamp's galaxy: keep
    # the shape of stamp i*nband, one flux per band
    shape = pars[::nband, :5]
    guess = np.concatenate([shape, pars[:, 5].reshape(nobj, nband)], axis=1)
    guess = guess * rng.uniform(0.95, 1.05, size=guess.shape)
    psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (ns, 1)), "gauss")
    sobj = np.repeat(np.arange(nobj), nband)
    sband = np.tile(np.arange(nband), nobj)
    ref = None
    for tag, env, hint in FORMS:
        for k in ("NGMIX_LM_TEAM_MIN", "NGMIX_LM_TEAMS"):
            os.environ.pop(k, None)
        os.environ.update(env)
        f = LMBatchFitter("exp")
        f.advance_hint = hint
        f.time_kernels = True
        for _ in range(3):
            res = f.go(sb, guess, psf=psf, stamp_obj=sobj, stamp_band=sband)
        torch.cuda.synchronize()
        k = f.kernel_ms
        if ref is None:
            ref = res["pars"].copy()
        same = bool(np.array_equal(ref, res["pars"]))
        print("n=%2d %-8s rounds %2d  lm_advance %.3f ms/launch  (total %.2f ms; lm_eval %.2f ms)  "
              "flags==0: %.3f  same pars: %s"
              % (5 + nband, tag, f.rounds_launched, k["lm_advance"] / f.rounds_launched,
                 k["lm_advance"], k["lm_eval"], float(np.mean(res["flags"] == 0)), same))
        sys.stdout.flush()
