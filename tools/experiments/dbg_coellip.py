import sys, numpy as np
sys.path.insert(0, "/root/repo")
import torch
import ngmix_amd as ngmix
from ngmix_amd.batch import StampBatch, GMixBatch
from ngmix_amd import lm_batch, pipeline
rng = np.random.RandomState(78)
n, dim, scale, noise = 150, 40, 0.263, 0.005
pars = np.zeros((n, 6))
pars[:, 0:2] = rng.uniform(-0.5, 0.5, size=(n, 2)) * scale
pars[:, 2:4] = rng.normal(scale=0.1, size=(n, 2))
pars[:, 4] = rng.uniform(0.3, 0.8, size=n)
pars[:, 5] = rng.uniform(100.0, 200.0, size=n)
full = np.zeros((n, 2, 6))
Tc = rng.uniform(0.18, 0.24, size=n)
full[:, 0, 0], full[:, 1, 0] = 0.7, 0.3
full[:, 0, 3] = full[:, 0, 5] = 0.5 * Tc
full[:, 1, 3] = full[:, 1, 5] = 0.5 * Tc * 3.0
full[:, 1, 4] = 0.02 * Tc
psf, _ = GMixBatch.from_pars(full.reshape(n, -1), "full", ngauss=2)
cen = (dim - 1) / 2.0
jac = np.array([cen, cen, scale, 0.0, 0.0, scale, scale ** 2, scale])
gm0, _ = GMixBatch.from_pars(pars, "exp")
gm, _ = gm0.convolve(psf)
geom = StampBatch.from_images(np.zeros((n, dim, dim)), None, jac)
truth, _ = geom.render(gm)
images = truth.cpu().numpy().reshape(n, dim, dim) + noise * rng.normal(size=(n, dim, dim))
sb = StampBatch.from_images(images, np.full((n, dim, dim), 1.0 / noise ** 2), jac)
pdim = 33
pjac = np.array([16.0, 16.0, scale, 0.0, 0.0, scale, scale ** 2, scale])
pgeom = StampBatch.from_images(np.zeros((n, pdim, pdim)), None, pjac)
pim, _ = pgeom.render(psf)
pimages = pim.cpu().numpy().reshape(n, pdim, pdim) + 1e-6 * rng.normal(size=(n, pdim, pdim))
psb = StampBatch.from_images(pimages, np.full((n, pdim, pdim), 1e12), pjac)
orig = lm_batch.LMBatchFitter.go
cap = {}
def go(self, stamps, guess, *a, **k):
    r = orig(self, stamps, guess, *a, **k)
    cap.setdefault("runs", []).append((self.model, np.array(guess), r))
    return r
lm_batch.LMBatchFitter.go = go
resc = pipeline.bootstrap_batch(sb, psb, model="exp", psf_Tguess=0.3, psf_ngauss=2,
                                psf_fitter="coellip")
print("psf flags", np.nonzero(resc["psf_em_flags"])[0], "fit flags", np.nonzero(resc["flags"])[0], resc["flags"][resc["flags"] != 0])
for model, guess, r in cap["runs"]:
    bad = np.nonzero(r["flags"])[0]
    print(model, "bad", bad, r["flags"][bad], "nfev", r["nfev"][bad], "ier", r["ier"][bad], "median nfev", np.median(r["nfev"]))
    for b in bad[:3]:
        print(" guess", guess[b]); print(" pars ", r["pars"][b])
        if model == "coellip":
            obs = ngmix.Observation(pimages[b], weight=np.full((pdim, pdim), 1e12),
                                    jacobian=ngmix.DiagonalJacobian(row=16.0, col=16.0, scale=scale))
            one = ngmix.fitting.CoellipFitter(ngauss=2).go(obs=obs, guess=guess[b])
            print(" per-object:", one["flags"], one["nfev"], one["pars"])
