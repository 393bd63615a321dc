"""config 3, mode ANALYTIC_LAZY: per lock-step round, how many fits ask for
|f|^2 alone, how many for the jacobian (and how many of those through phase
JAC, i.e. after a failed prediction), and the HIP-event time of the round's
lm_eval launch.  usage: python tools/lm_lazy_probe.py [nstamps]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from ngmix_amd import _lib  # noqa: E402
from ngmix_amd.batch import GMixBatch  # noqa: E402
from ngmix_amd.lm_batch import LMBatchFitter  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
sb, _, pars = bench.make_workload(n, 1000, "cuda")
rng = np.random.RandomState(7)
guess = pars * rng.uniform(0.9, 1.1, size=pars.shape)
guess[:, 0:2] = pars[:, 0:2] + rng.uniform(-0.05, 0.05, size=(n, 2))
guess[:, 2:4] = pars[:, 2:4] + rng.uniform(-0.03, 0.03, size=(n, 2))
psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (n, 1)), "gauss")
f = LMBatchFitter("exp")
f.time_kernels = True
for _ in range(3):
    f.go(sb, guess, psf=psf)
# one round at a time, reading the states between rounds
job = f._enqueue.__func__  # noqa: F841  (documentation: the pieces used below)
f._rounds_hint = 1
j = f._enqueue(sb, guess, psf, None, None, 1, False)
torch.cuda.synchronize()
L = _lib.lib()
r = 1
while True:
    st = j.d_states.cpu().numpy().reshape(-1).view(_lib.LM_STATE_DTYPE)
    live = st["phase"] != _lib.LM_PHASE_DONE
    fonly = live & (st["fonly"] != 0)
    jacph = live & (st["phase"] == _lib.LM_PHASE_JAC)
    if not live.any():
        break
    f._queue_rounds(j, 1)
    torch.cuda.synchronize()
    ev = j.chunks[-1][3]
    ms = ctypes.c_float()
    L.ngmix_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
    r += 1
    print("round %d: live %6d  |f|^2-only %6d  jacobian %6d (of them phase JAC %5d)  lm_eval %.3f ms"
          % (r, live.sum(), fonly.sum(), (live & ~fonly).sum(), jacph.sum(), ms.value))
print("nfev histogram:", np.bincount(st["nfev"]), " njev:", np.bincount(st["njev"]))
