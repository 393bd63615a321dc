"""where a LMBatchFitter.go() call spends its time (config 3): wall-clock per
phase with a device sync after each, next to the un-instrumented call
python tools/lm_phases.py [nstamps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from ngmix_amd.batch import GMixBatch  # noqa: E402
from ngmix_amd.lm_batch import LMBatchFitter  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device("cuda", 0)
sb, _, pars = bench.make_workload(n, seed=1000, device=dev)
rng = np.random.RandomState(7)
guess = pars * rng.uniform(0.9, 1.1, size=pars.shape)
psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (n, 1)), "gauss",
                             device=dev)
fitter = LMBatchFitter("exp")
for _ in range(3):
    fitter.go(sb, guess, psf=psf)
torch.cuda.synchronize()
best = 1e9
allt = []
for _ in range(12):
    t0 = time.perf_counter()
    res = fitter.go(sb, guess, psf=psf)
    res["pars"]
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    allt.append((round((t1 - t0) * 1e3, 2), round((t2 - t1) * 1e3, 2)))
    best = min(best, t2 - t0)
print("plain go(): best of 12 %.3f ms (loop %.3f ms, %d rounds); (go, trailing sync) ms: %s" % (
    best * 1e3, fitter.loop_seconds * 1e3, fitter.rounds, allt))
fitter.time_phases = True
acc = {}
for _ in range(5):
    fitter.go(sb, guess, psf=psf)
    for k, v in fitter.phase_ms.items():
        acc.setdefault(k, []).append(v)
print("phases (ms, min of 5, each closed by a sync):",
      {k: round(min(v), 3) for k, v in acc.items()},
      "sum %.3f" % sum(min(v) for v in acc.values()))

fitter.time_phases = "nosync"
acc = {}
for _ in range(5):
    t0 = time.perf_counter()
    res = fitter.go(sb, guess, psf=psf)
    dt = (time.perf_counter() - t0) * 1e3
    for k, v in fitter.phase_ms.items():
        acc.setdefault(k, []).append(v)
    acc.setdefault("total", []).append(dt)
print("host time per phase without syncs (ms, all 5):",
      {k: [round(x, 2) for x in v] for k, v in acc.items()})

# per-launch times of the pixel pass in one fit (HIP events) and the stamps
# each launch still had to evaluate
fitter.time_phases = False
fitter.time_kernels = True
guess2 = guess.copy()
guess2[:, 0:2] = pars[:, 0:2] + rng.uniform(-0.05, 0.05, size=(n, 2))
guess2[:, 2:4] = pars[:, 2:4] + rng.uniform(-0.03, 0.03, size=(n, 2))
for g, name in ((guess, "guess = truth x U(0.9, 1.1)"), (guess2, "+ centre / shear offsets (bench C3)")):
    fitter.go(sb, g, psf=psf)
    import ngmix_amd.lm_batch as lb
    print(name, "rounds", fitter.rounds, "loop %.3f ms" % (fitter.loop_seconds * 1e3),
          "eval total %.3f ms over %.0f stamp evaluations" % (
              fitter.eval_ms_total, fitter.eval_stamps_total), [(round(t, 3), int(w)) for t, w in fitter.eval_launches])
