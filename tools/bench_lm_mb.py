"""multi-band lock-step LM: nband bands x 1 epoch per object, 'exp' (x) psf,
48x48 stamps: 5 + nband parameters.  python tools/bench_lm_mb.py [nobj] [nband]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from ngmix_amd.batch import GMixBatch  # noqa: E402
from ngmix_amd.lm_batch import LMBatchFitter  # noqa: E402

nobj = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
nband = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ns = nobj * nband
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
# stamps of one object: the same shape parameters, band fluxes in a ratio
sb, _, pars = bench.make_workload(ns, seed=1000, device=dev)
# (make_workload draws independent objects per stamp; for a timing run the bands
# only need consistent shape parameters: refit each object to its first stamp's
# shape by using that as the guess -- the fits still run their full course)
sobj = np.repeat(np.arange(nobj, dtype=np.int32), nband)
sband = np.tile(np.arange(nband, dtype=np.int32), nobj)
first = pars.reshape(nobj, nband, 6)[:, 0, :]
guess = np.zeros((nobj, 5 + nband))
guess[:, :5] = first[:, :5]
guess[:, 5:] = pars.reshape(nobj, nband, 6)[:, :, 5]
psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (ns, 1)), "gauss", device=dev)
fitter = LMBatchFitter("exp", fit_pars={"maxfev": 30, "ftol": 1e-5, "xtol": 1e-5})
fitter.go(sb, guess, psf=psf, stamp_obj=sobj, stamp_band=sband)
ts, loops = [], []
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = fitter.go(sb, guess, psf=psf, stamp_obj=sobj, stamp_band=sband)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
    loops.append(fitter.loop_seconds)
print("generic=%s: %d objects x %d bands (%d parameters): %.2f ms end to end, loop %.2f ms, "
      "rounds %d, median nfev %d" % (os.environ.get("NGMIX_LM_GENERIC", "0"), nobj, nband,
                                     5 + nband, min(ts) * 1e3, min(loops) * 1e3, fitter.rounds,
                                     np.median(res["nfev"])))
