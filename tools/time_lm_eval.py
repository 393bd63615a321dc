"""lm_eval_kernel alone on the C3 workload: the launch at every fit's starting
point, HIP events over 30 launches (NGMIX_LM_JBASIS=1 selects the per-pixel
map for A/B)
python tools/time_lm_eval.py [nstamps]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
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
stats = None if os.environ.get("NGMIX_LM_JBASIS") else \
    torch.zeros((n, 2), dtype=torch.float64, device=dev)
b = sb._batch(1)


def launch():
    _lib.check(L.ngmix_lm_eval_batch(ctypes.byref(b), get_model_num("exp"), 0, _dptr(st),
                                     None, None, _dptr(psf.data), 1, _dptr(sums),
                                     _dptr(status), _dptr(stats), _stream()), "eval")


for _ in range(60):
    launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = []
for rep in range(5):
    e0.record()
    for _ in range(30):
        launch()
    e1.record()
    torch.cuda.synchronize()
    best.append(e0.elapsed_time(e1) / 30)
print("lm_eval %s: %.4f ms per launch of %d stamps (min of 5 x 30; all %s)" % (
    "J basis" if stats is None else "raw basis", min(best), n,
    [round(x, 4) for x in best]))
print("checksum", float(sums.sum()), int(status.abs().sum()))

# the same launch timed one by one, alone and with an lm_advance-sized kernel
# between the launches (a copy of the states is advanced, the evaluated one
# stays at the starting point)
st2 = st.clone()
nact = torch.zeros(1, dtype=torch.int32, device=dev)


def advance():
    _lib.check(L.ngmix_lm_advance_batch(_dptr(st2), n, None, None, _dptr(sums), 6 + 256 * 6,
                                        None, _dptr(nact), None, None, _stream()), "advance")


for label, between in (("alone", None), ("advance between", advance)):
    ts = []
    for _ in range(20):
        if between is not None:
            st2.copy_(st)
            between()
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        launch()
        b_.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b_))
    print("one launch per event pair, %s: median %.4f min %.4f ms" % (
        label, float(np.median(ts)), min(ts)))


def reinit():
    _lib.check(L.ngmix_lm_init_batch(_dptr(st), n, 6, _dptr(dg), 1e-8, 1e-8, 0.0, 700, 100.0,
                                     _lib.LM_MODE_ANALYTIC, None, None, _stream()), "init")


big = torch.empty(64 * 1024 * 1024, dtype=torch.float64, device=dev)   # 512 MB
for label, between in (("states re-initialised before each launch (as go() does)", reinit),
                       ("512 MB of unrelated memory written before each launch",
                        lambda: big.fill_(1.0))):
    ts = []
    for _ in range(20):
        between()
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        launch()
        b_.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b_))
    print("one launch per event pair, %s: median %.4f min %.4f ms" % (
        label, float(np.median(ts)), min(ts)))

# what precedes the launch: nothing, a busy kernel of lm_advance's length, an idle gap
import time as _time
gmx, _ = GMixBatch.from_pars(pars[:20000], "exp", device=dev)
sub = sb.select(np.arange(20000))
psub, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (20000, 1)), "gauss", device=dev)
gsub, _ = gmx.convolve(psub)


def busy():
    sub.loglike(gsub)


def idle():
    torch.cuda.synchronize()
    _time.sleep(0.0005)


for label, between in (("a 0.14 ms loglike launch before", busy), ("0.5 ms of idle GPU before", idle),
                       ("lm_advance before", advance)):
    ts = []
    for _ in range(30):
        if between is advance:
            st2.copy_(st)
        between()
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        launch()
        b_.record()
        between() if between is not idle else None
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b_))
    print("one launch per event pair, %s: median %.4f min %.4f ms" % (
        label, float(np.median(ts)), min(ts)))

# a 7 MB device-to-host copy beside lm_advance / beside lm_eval (the copies of
# a batch's results are blit kernels on this platform: tools/d2h_probe.py)
dsrc = torch.randn(7 * 1024 * 1024 // 8, dtype=torch.float64, device=dev)
hdst = torch.empty(dsrc.shape, dtype=torch.float64, pin_memory=True)
side = torch.cuda.Stream()


def timed_pair(kernel, with_copy, reps=20):
    ts = []
    for _ in range(reps):
        st2.copy_(st)
        torch.cuda.synchronize()
        t0 = _time.perf_counter()
        if with_copy:
            e = torch.cuda.Event()
            e.record()
            with torch.cuda.stream(side):
                side.wait_event(e)
                hdst.copy_(dsrc, non_blocking=True)
        kernel()
        torch.cuda.synchronize()
        ts.append((_time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


def copy_only():
    with torch.cuda.stream(side):
        hdst.copy_(dsrc, non_blocking=True)


print("wall ms (median of 20): 7 MB copy alone %.3f; lm_advance alone %.3f, with the copy beside it %.3f; "
      "lm_eval alone %.3f, with the copy beside it %.3f" % (
          timed_pair(copy_only, False), timed_pair(advance, False), timed_pair(advance, True),
          timed_pair(launch, False), timed_pair(launch, True)))

for mb, chunks in ((7, 1), (19, 1), (29, 1), (48, 1), (48, 7), (48, 24)):
    dsrc = torch.randn(mb * 1024 * 1024 // 8, dtype=torch.float64, device=dev)
    hdst = torch.empty(dsrc.shape, dtype=torch.float64, pin_memory=True)
    step_ = dsrc.numel() // chunks

    def copies():
        for c in range(chunks):
            hdst[c * step_:(c + 1) * step_].copy_(dsrc[c * step_:(c + 1) * step_],
                                                  non_blocking=True)

    def both():
        e = torch.cuda.Event()
        e.record()
        with torch.cuda.stream(side):
            side.wait_event(e)
            copies()
        launch()

    def only():
        with torch.cuda.stream(side):
            copies()

    def wall(fn):
        ts = []
        for _ in range(15):
            torch.cuda.synchronize()
            t0 = _time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((_time.perf_counter() - t0) * 1e3)
        return float(np.median(ts))
    print("%2d MB in %2d copies: alone %.3f ms; beside lm_eval (1.39 alone): %.3f ms" % (
        mb, chunks, wall(only), wall(both)))
