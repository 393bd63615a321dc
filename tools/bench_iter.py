"""config-4-like timing of the iterative kernels: N 32x32 stamps,
round-ish gaussian (x) gaussian psf objects; admom and 1-gaussian em_run.
python tools/bench_iter.py [nstamps] [reps] [dim]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ngmix_amd import _lib  # noqa: E402
from ngmix_amd.batch import StampBatch, GMixBatch, records_to_numpy  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 32
scale = 0.263
rng = np.random.RandomState(5)
pars = np.zeros((n, 6))
pars[:, 0:2] = rng.uniform(-0.5, 0.5, size=(n, 2)) * scale
pars[:, 2:4] = rng.normal(scale=0.05, size=(n, 2))
pars[:, 4] = rng.uniform(0.3, 0.9, size=n) + 0.27
pars[:, 5] = rng.uniform(50, 200, size=n)
gm_true, _ = GMixBatch.from_pars(pars, "gauss")
jac = np.array([(dim - 1) / 2, (dim - 1) / 2, scale, 0, 0, scale, scale ** 2, scale])
geom = StampBatch(None, None, torch.from_numpy(np.tile(jac, (n, 1))).cuda(),
                  np.full(n, dim), np.full(n, dim),
                  np.arange(n, dtype=np.int64) * dim * dim, True)
truth, _ = geom.render(gm_true)
gen = torch.Generator(device="cuda")
gen.manual_seed(1)
val = truth.reshape(n, -1) + 0.01 * torch.randn((n, dim * dim), generator=gen,
                                                device="cuda", dtype=torch.float64)
ierr = torch.full((n * dim * dim,), 100.0, dtype=torch.float64, device="cuda")
sb = StampBatch(val.reshape(-1), ierr, geom.jac, np.full(n, dim), np.full(n, dim),
                np.arange(n, dtype=np.int64) * dim * dim, True)

guess = np.zeros((n, 6))
guess[:, 4] = pars[:, 4] * rng.uniform(0.9, 1.1, size=n)
guess[:, 5] = 1.0


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts)


wt0, _ = GMixBatch.from_pars(guess, "gauss")
state = {}


def run_admom():
    wt = wt0.clone()
    state["res"], state["st"] = sb.admom(wt)


t = timeit(run_admom)
res = records_to_numpy(state["res"], _lib.ADMOM_RESULT_DTYPE)
print("admom: %.2f ms for %d stamps -> %.3g obj/s; numiter median %d, flags!=0: %d, status!=0: %d" % (
    t * 1e3, n, n / t, np.median(res["numiter"]), (res["flags"] != 0).sum(),
    int((state["st"] != 0).sum())))
# flops: ~ (35 + 45) per pixel per iteration (passes without cov) + 300 for cov pass
iters = res["numiter"].mean()
print("   mean numiter %.2f" % iters)

psfpars = np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (n, 1))
emguess = pars.copy()
emguess[:, 4] = (pars[:, 4] - 0.27) * rng.uniform(0.9, 1.1, size=n)
emguess[:, 5] = pars[:, 5] * scale ** 2 * rng.uniform(0.9, 1.1, size=n)
gm0, _ = GMixBatch.from_pars(emguess, "gauss")
psf, _ = GMixBatch.from_pars(psfpars, "gauss")
sky = 0.05


def run_em():
    g = gm0.clone()
    state["em"], state["emst"], _ = sb.em(g, psf, sky=sky)


# EM wants positive images: add a sky
sb.val += sky
t = timeit(run_em)
o = state["em"].cpu().numpy()
print("em_run: %.2f ms for %d stamps -> %.3g obj/s; numiter median %d, status!=0: %d" % (
    t * 1e3, n, n / t, np.median(o[:, 0]), int((state["emst"] != 0).sum())))
