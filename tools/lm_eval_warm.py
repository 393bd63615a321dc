"""does lm_eval speed depend on what the GPU did in the last second?  The same
launch, one at a time with a sync after each: cold (first thing the process
does), after 0.4 s of back-to-back launches, then every 50 ms while idling"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from ngmix_amd import _lib  # noqa: E402
from ngmix_amd.batch import GMixBatch, _dptr, _stream  # noqa: E402
from ngmix_amd.gmix import get_model_num  # noqa: E402

n = 100000
dev = torch.device("cuda", 0)
sb, _, pars = bench.make_workload(n, seed=1000, device=dev)
rng = np.random.RandomState(7)
guess = pars * rng.uniform(0.9, 1.1, size=pars.shape)
psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (n, 1)), "gauss",
                             device=dev)
L = _lib.lib()
st = torch.empty((n, _lib.LM_STATE_DTYPE.itemsize), dtype=torch.uint8, device=dev)
dg = torch.from_numpy(np.ascontiguousarray(guess)).to(dev)
_lib.check(L.ngmix_lm_init_batch(_dptr(st), n, 6, _dptr(dg), 1e-8, 1e-8, 0.0, 700, 100.0,
                                 _lib.LM_MODE_ANALYTIC, None, None, _stream()), "init")
sums = torch.zeros((n, 28), dtype=torch.float64, device=dev)
status = torch.zeros(n, dtype=torch.int32, device=dev)
stats = torch.zeros((n, 2), dtype=torch.float64, device=dev)
b = sb._batch(1)


def launch():
    _lib.check(L.ngmix_lm_eval_batch(ctypes.byref(b), get_model_num("exp"), 0, _dptr(st),
                                     None, None, _dptr(psf.data), 1, _dptr(sums),
                                     _dptr(status), _dptr(stats), _stream()), "eval")


def one():
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    launch()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e)


torch.cuda.synchronize()
time.sleep(0.5)
print("cold, one at a time:", [round(one(), 3) for _ in range(12)])
t0 = time.perf_counter()
k = 0
while time.perf_counter() - t0 < 0.4:
    for _ in range(10):
        launch()
    torch.cuda.synchronize()
    k += 10
print("after %d back-to-back launches (0.4 s):" % k, [round(one(), 3) for _ in range(12)])
out = []
for i in range(12):
    time.sleep(0.05)
    out.append(round(one(), 3))
print("one every 50 ms of idling:", out)
out = []
for i in range(30):
    time.sleep(0.0015)
    out.append(round(one(), 3))
print("one every 1.5 ms of idling (the duty cycle of the LM loop):", out)
