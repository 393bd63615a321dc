"""why does lm_eval take 1.45 ms inside LMBatchFitter.go() and 1.30 ms alone?
Replays the first round's launch on go()'s own buffers (kept by the fitter) and
on fresh ones.  python tools/lm_eval_probe.py"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from ngmix_amd import _lib  # noqa: E402
from ngmix_amd.batch import GMixBatch, _dptr, _stream  # noqa: E402
from ngmix_amd.gmix import get_model_num  # noqa: E402
from ngmix_amd.lm_batch import LMBatchFitter  # noqa: E402

n = 100000
dev = torch.device("cuda", 0)
sb, _, pars = bench.make_workload(n, seed=1000, device=dev)
rng = np.random.RandomState(7)
guess = pars * rng.uniform(0.9, 1.1, size=pars.shape)
psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (n, 1)), "gauss",
                             device=dev)
L = _lib.lib()
fitter = LMBatchFitter("exp")
fitter.time_kernels = True
for _ in range(4):
    fitter.go(sb, guess, psf=psf)
print("inside go():", [round(t, 3) for t, w in fitter.eval_launches])


def timed(states, sums, status, stats, label, reps=10):
    dg = torch.from_numpy(np.ascontiguousarray(guess)).to(dev)
    b = sb._batch(1)
    ts = []
    for _ in range(reps):
        _lib.check(L.ngmix_lm_init_batch(_dptr(states), n, 6, _dptr(dg), 1.49012e-8, 1.49012e-8,
                                         0.0, 700, 100.0, _lib.LM_MODE_ANALYTIC, None, None,
                                         _stream()), "init")
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.check(L.ngmix_lm_eval_batch(ctypes.byref(b), get_model_num("exp"), 0,
                                         _dptr(states), None, None, _dptr(psf.data), 1,
                                         _dptr(sums), _dptr(status), _dptr(stats), _stream()),
                   "eval")
        e.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(e))
    print("%s: median %.4f min %.4f" % (label, float(np.median(ts)), min(ts)))


st = fitter._d_states
timed(st, torch.empty((n, 28), dtype=torch.float64, device=dev),
      torch.empty(n, dtype=torch.int32, device=dev),
      torch.empty((n, 2), dtype=torch.float64, device=dev), "go()'s state buffer, fresh outputs")
st2 = torch.empty_like(st)
timed(st2, torch.empty((n, 28), dtype=torch.float64, device=dev),
      torch.empty(n, dtype=torch.int32, device=dev),
      torch.empty((n, 2), dtype=torch.float64, device=dev), "fresh state buffer")
for _ in range(2):
    fitter.go(sb, guess, psf=psf)
print("inside go() again:", [round(t, 3) for t, w in fitter.eval_launches])
