"""order independence / run-to-run determinism of the batched LM on the C3
workload: python tools/lm_order_check.py [n]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from ngmix_amd.batch import GMixBatch  # noqa: E402
from ngmix_amd.lm_batch import LMBatchFitter  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
sb, _, pars = bench.make_workload(n, 11, "cuda")
rng = np.random.RandomState(77)
guess = pars * rng.uniform(0.9, 1.1, size=pars.shape)
guess[:, 0:2] = pars[:, 0:2] + rng.uniform(-0.05, 0.05, size=(n, 2))
guess[:, 2:4] = pars[:, 2:4] + rng.uniform(-0.03, 0.03, size=(n, 2))
psf, _ = GMixBatch.from_pars(np.tile([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], (n, 1)), "gauss")
a = LMBatchFitter("exp").go(sb, guess, psf=psf)
b = LMBatchFitter("exp").go(sb, guess, psf=psf)
print("run to run: pars equal", np.array_equal(a["pars"], b["pars"]),
      "differing objects", int(np.any(a["pars"] != b["pars"], axis=1).sum()))
sub = np.random.RandomState(8).choice(n, size=min(5000, n), replace=False)
c = LMBatchFitter("exp").go(sb.select(sub), guess[sub], psf=psf.select(sub))
bad = np.any(c["pars"] != a["pars"][sub], axis=1)
print("subset: differing objects", int(bad.sum()), "nfev equal",
      np.array_equal(c["nfev"], a["nfev"][sub]))
if bad.any():
    i = np.nonzero(bad)[0][:5]
    print("positions in subset", i, "objects", sub[i])
    print((c["pars"][i] - a["pars"][sub][i]) / a["pars_err"][sub][i])
# one evaluation of the sums at the guess: whole batch against the subset
import ctypes
from ngmix_amd import _lib
from ngmix_amd.batch import _dptr, _stream
from ngmix_amd.gmix import get_model_num
L = _lib.lib()


def sums_at_guess(sbx, gx, psfx):
    m = gx.shape[0]
    dev = sbx.device
    st = torch.empty((m, _lib.LM_STATE_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    dg = torch.from_numpy(np.ascontiguousarray(gx)).to(dev)
    _lib.check(L.ngmix_lm_init_batch(_dptr(st), m, 6, _dptr(dg), 1e-8, 1e-8, 0.0, 100, 100.0,
                                     _lib.LM_MODE_ANALYTIC, None, None, _stream()), "init")
    sobj = torch.arange(m, dtype=torch.int32, device=dev)
    sband = torch.zeros(m, dtype=torch.int32, device=dev)
    sums = torch.zeros((m, 28), dtype=torch.float64, device=dev)
    status = torch.zeros(m, dtype=torch.int32, device=dev)
    b = sbx._batch(1)
    _lib.check(L.ngmix_lm_eval_batch(ctypes.byref(b), get_model_num("exp"), 0, _dptr(st),
                                     _dptr(sobj), _dptr(sband), _dptr(psfx.data), 1,
                                     _dptr(sums), _dptr(status), None, _stream()), "eval")
    torch.cuda.synchronize()
    return sums.cpu().numpy(), st


s_all, st_all = sums_at_guess(sb, guess, psf)
s_all2, _ = sums_at_guess(sb, guess, psf)
s_sub, st_sub = sums_at_guess(sb.select(sub), guess[sub], psf.select(sub))
print("eval run to run equal:", np.array_equal(s_all, s_all2))
d = np.any(s_sub != s_all[sub], axis=1)
print("eval subset: differing stamps", int(d.sum()), "of", d.size)
if d.any():
    i = np.nonzero(d)[0][0]
    print(i, s_sub[i] - s_all[sub][i])
print("run to run fits:", np.array_equal(a["pars"], b["pars"]))
