"""Fits WITH a joint prior through the lock-step driver -- the host joint prior
of ngmix_amd.joint_prior put in its kernel form by as_batch_prior, its rows
evaluated inside the device loop -- against the same fits object by object
through MINPACK with the host prior's fill_fdiff (Fitter(batched=False)): random
models (gauss / exp / dev / turb by lmder, bdf / bd by lmdif), one or two bands,
random prior terms (two-sided erf, normal with and without leastsqbound bounds,
log-normal, truncated gaussian, flat) of random widths, tight enough to pull
the solution.

Per class: pass / fail agreement, ier, the nfev histogram in jacobians, the
largest parameter and ln p differences among the fits both routes converged.

usage: python tools/fuzz_lm_prior_vs_minpack.py [seconds] [seed] [classes]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("NGMIX_FITTER_BATCHED", "0")
import test_gpu_lm_batch as tb  # noqa: E402
import ngmix_amd as ngmix  # noqa: E402
from ngmix_amd import priors, joint_prior, prior_batch as pb  # noqa: E402
from ngmix_amd.lm_batch import LMBatchFitter  # noqa: E402

CLASSES = ["gauss", "exp", "dev", "turb", "bdf", "bd"]


def size_prior(rng, prng):
    kind = rng.randint(5)
    if kind == 0:
        return priors.TwoSidedErf(-0.05, 0.03, rng.uniform(1.0, 4.0), rng.uniform(0.1, 0.5), rng=prng)
    if kind == 1:
        return priors.Normal(rng.uniform(0.4, 0.7), rng.uniform(0.05, 0.4), rng=prng)
    if kind == 2:
        return priors.Normal(rng.uniform(0.4, 0.7), rng.uniform(0.1, 0.4), rng=prng,
                             bounds=(0.02, rng.uniform(1.5, 5.0)))
    if kind == 3:
        return priors.LogNormal(rng.uniform(0.4, 0.8), rng.uniform(0.1, 0.5), rng=prng)
    return priors.TruncatedGaussian(rng.uniform(0.4, 0.7), rng.uniform(0.2, 0.5), 0.01, 5.0, rng=prng)


def flux_prior(rng, prng):
    kind = rng.randint(4)
    if kind == 0:
        return priors.TwoSidedErf(-10.0, 1.0, rng.uniform(400.0, 5000.0), rng.uniform(10.0, 100.0),
                                  rng=prng)
    if kind == 1:
        return priors.FlatPrior(-50.0, 1.0e5, rng=prng)
    if kind == 2:
        return priors.Normal(rng.uniform(100.0, 180.0), rng.uniform(20.0, 100.0), rng=prng,
                             bounds=(1.0, None) if rng.randint(2) else None)
    return priors.LogNormal(rng.uniform(100.0, 180.0), rng.uniform(40.0, 120.0), rng=prng)


def make_prior(rng, model, nband):
    prng = np.random.RandomState(int(rng.randint(1 << 30)))
    scale = 0.263
    cen = priors.CenPrior(0.0, 0.0, scale * rng.uniform(0.2, 1.0), scale * rng.uniform(0.2, 1.0),
                          rng=prng)
    g = priors.GPriorBA(rng.uniform(0.1, 0.4), rng=prng)
    T = size_prior(rng, prng)
    F = [flux_prior(rng, prng) for _ in range(nband)]
    Farg = F if nband > 1 else F[0]
    if model in ("bdf", "bd"):
        if rng.randint(2):
            fd = priors.Normal(0.5, rng.uniform(0.05, 0.3), rng=prng, bounds=(0.0, 1.0))
        else:
            fd = priors.TruncatedGaussian(0.5, rng.uniform(0.1, 0.3), -0.5, 1.5, rng=prng)
        if model == "bdf":
            return joint_prior.PriorBDFSep(cen, g, T, fd, Farg)
        return joint_prior.PriorBDSep(cen, g, T, priors.Normal(0.0, rng.uniform(0.1, 0.5), rng=prng),
                                      fd, Farg)
    return joint_prior.PriorSimpleSep(cen, g, T, Farg)


def one_case(seed, classes=CLASSES):
    rng = np.random.RandomState(seed)
    model = classes[int(rng.randint(len(classes)))]
    nband = int(rng.randint(1, 3))
    nobj = int(rng.randint(2, 6))
    npsf = int(rng.randint(1, 3))
    dim = int(rng.choice([32, 40]))
    sobj = np.repeat(np.arange(nobj), nband).astype(np.int32)
    sband = np.tile(np.arange(nband), nobj).astype(np.int32)
    ns = sobj.size
    psf_rows, psf = tb._multi_gauss_psf(ns, npsf, False, rng)
    extra = None
    if model == "bdf":
        extra = np.repeat(rng.uniform(0.1, 0.9, size=nobj), nband)[:, None]
    elif model == "bd":
        extra = np.stack([np.repeat(rng.uniform(-0.3, 0.3, size=nobj), nband),
                          np.repeat(rng.uniform(0.1, 0.9, size=nobj), nband)], axis=1)
    pars, guess, images, weights, jobj, sb = tb._objects_with_psf(
        ns, model, psf, rng, dim=dim, extra=extra, noise=float(rng.choice([0.01, 0.03, 0.1])))
    nshape = {"bdf": 6, "bd": 7}.get(model, 5)
    first = np.arange(nobj) * nband
    flux = guess[:, nshape].reshape(nobj, nband)
    g2 = np.concatenate([guess[first, :nshape], flux], axis=1)
    prior = make_prior(rng, model, nband)
    bp = pb.as_batch_prior(prior)
    assert bp.descriptor() is not None
    fitter = LMBatchFitter(model, prior=prior)
    res = fitter.go(sb, g2, psf=psf, stamp_obj=sobj, stamp_band=sband)
    assert fitter.prior_path == "kernel"
    ones = []
    for o in range(nobj):
        mb = ngmix.MultiBandObsList()
        for b in range(nband):
            ol = ngmix.ObsList()
            s = o * nband + b
            pobs = ngmix.Observation(np.zeros((5, 5)), jacobian=jobj,
                                     gmix=ngmix.GMix(pars=psf_rows[s]))
            ol.append(ngmix.Observation(images[s], weight=weights[s], jacobian=jobj, psf=pobs))
            mb.append(ol)
        ones.append(ngmix.fitting.Fitter(model=model, prior=prior, batched=False).go(
            obs=mb, guess=g2[o]))
    return model, g2.shape[1], res, ones


def new_stats():
    return dict(n=0, flags=0, ier=0, conv_drv=0, conv_mp=0, both=0, worst=0.0, worst_seed=None,
                worst_lnp=0.0, hist=np.zeros(6, dtype=np.int64), maxd=0, odd=[])


def tally(st, seed, n, res, ones):
    for o, one in enumerate(ones):
        st["n"] += 1
        fl, ie, nf = int(res["flags"][o]), int(res["ier"][o]), int(res["nfev"][o])
        st["flags"] += int((one["flags"] == 0) == (fl == 0))
        st["ier"] += int(one["ier"] == ie)
        st["conv_drv"] += int(fl == 0)
        st["conv_mp"] += int(one["flags"] == 0)
        d = abs(nf - int(one["nfev"]))
        st["maxd"] = max(st["maxd"], d)
        jac = n + 1
        k = 0 if d == 0 else 1 if d <= jac else 2 if d <= 2 * jac else 3 if d <= 4 * jac \
            else 4 if d <= 8 * jac else 5
        st["hist"][k] += 1
        if (one["flags"] == 0) != (fl == 0) and len(st["odd"]) < 8:
            st["odd"].append((seed, o, int(one["flags"]), fl, int(one["ier"]), ie,
                              int(one["nfev"]), nf))
        if one["flags"] == 0 and fl == 0:
            st["both"] += 1
            w = float(np.max(np.abs(res["pars"][o] - one["pars"]) / one["pars_err"]))
            if w > st["worst"]:
                st["worst"], st["worst_seed"] = w, (seed, o)
            st["worst_lnp"] = max(st["worst_lnp"], abs(float(res["lnprob"][o]) - one["lnprob"]))


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    master = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 23)
    classes = sys.argv[3].split(",") if len(sys.argv) > 3 else CLASSES
    stats = {c: new_stats() for c in classes}
    EDGES = ["0", "<=1 jac", "<=2 jac", "<=4 jac", "<=8 jac", ">8 jac"]
    t0 = time.time()
    ncase = 0
    while time.time() - t0 < budget:
        seed = int(master.randint(1 << 30))
        cls, n, res, ones = one_case(seed, classes)
        tally(stats[cls], seed, n, res, ones)
        ncase += 1
    print("fuzz_lm_prior_vs_minpack: %.0f s, %d cases" % (time.time() - t0, ncase))
    for c in classes:
        st = stats[c]
        if not st["n"]:
            continue
        print("%-6s %6d fits | converged: driver %.4f MINPACK %.4f | pass/fail agrees %d (%.4f), ier equal "
              "%d | |dnfev| %s max %d | both converged %d: worst |dpars|/err %.2e %s, worst |dlnprob| %.2e"
              % (c, st["n"], st["conv_drv"] / st["n"], st["conv_mp"] / st["n"], st["flags"],
                 st["flags"] / st["n"], st["ier"],
                 " ".join("%s:%d" % (e, h) for e, h in zip(EDGES, st["hist"])), st["maxd"],
                 st["both"], st["worst"], st["worst_seed"], st["worst_lnp"]))
        for rec in st["odd"]:
            print("      pass/fail differs: seed %d fit %d: flags %d / %d, ier %d / %d, nfev %d / %d "
                  "(MINPACK / driver)" % rec)


if __name__ == "__main__":
    main()
