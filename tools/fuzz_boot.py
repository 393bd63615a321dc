"""The batched bootstrap against the reference-shaped per-object one, on random
scenarios: every case builds a handful of objects -- 1-3 bands, 1-3 epochs per
band, a turbulent psf; psf stamps with no star on them, psf and object guesses
far off on the first attempt, tight evaluation caps (what oracle/
gen_golden_boot.py does for eighteen objects under the reference itself) --
and runs them

  (a) one object at a time through ngmix_amd.bootstrap.Bootstrapper(
      Runner(Fitter), PSFRunner(psf fitter)) -- the reference's control flow
      (bootstrap.py:24-154, runners.py:116-223) over MINPACK calling the seam
      kernels (batched=False) -- with guessers that hand out stored arrays;
  (b) as ONE bootstrap_batch call from the same stored guesses.

Compared per stamp: whether the psf fit passed, its attempts, its nfev (lmder
psf fitter); per object: BootPSFFailure, the epochs kept, flags, attempts, nfev
(exact for the lmder models) and the parameters.

usage: python tools/fuzz_boot.py [seconds] [seed] [prior]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
os.environ.setdefault("NGMIX_FITTER_BATCHED", "0")
import ngmix_amd as ngmix  # noqa: E402
from ngmix_amd.batch import StampBatch  # noqa: E402
from ngmix_amd.bootstrap import Bootstrapper  # noqa: E402
from ngmix_amd.gexceptions import BootPSFFailure  # noqa: E402
from ngmix_amd.pipeline import bootstrap_batch, BOOT_PSF_FAILURE  # noqa: E402
from ngmix_amd.runners import Runner, PSFRunner  # noqa: E402

SCALE, DIM, PDIM = 0.263, 32, 25
FP = {"ftol": 1.0e-5, "xtol": 1.0e-5}


class Stored(object):
    """guesses[k] on the k-th call for the same observation"""

    def __init__(self, table=None, seq=None):
        self.table, self.seq, self.count, self.k = table, seq, {}, 0

    def __call__(self, obs, **kw):
        if self.seq is not None:
            g = self.seq[self.k]
            self.k += 1
            return np.array(g)
        k = self.count.get(id(obs), 0)
        self.count[id(obs)] = k + 1
        return np.array(self.table[id(obs)][k])


def _joint_prior(rng, nband):
    """a PriorSimpleSep of ngmix_amd.priors terms, wide enough not to fight the data"""
    from ngmix_amd import priors, joint_prior
    prng = np.random.RandomState(int(rng.randint(1 << 30)))
    T = priors.TwoSidedErf(-0.1, 0.03, 50.0, 1.0, rng=prng) if rng.randint(2) else \
        priors.LogNormal(0.6, 0.5, rng=prng)
    F = [priors.TwoSidedErf(-10.0, 1.0, 1.0e5, 100.0, rng=prng) if rng.randint(2) else
         priors.Normal(180.0, 200.0, rng=prng, bounds=(0.5, None)) for _ in range(nband)]
    return joint_prior.PriorSimpleSep(
        priors.CenPrior(0.0, 0.0, 0.263 * rng.uniform(0.5, 2.0), 0.263 * rng.uniform(0.5, 2.0),
                        rng=prng), priors.GPriorBA(rng.uniform(0.2, 0.4), rng=prng), T,
        F if nband > 1 else F[0])


def one_case(seed, models=("exp", "gauss", "dev", "turb"), psf_kinds=("gauss", "coellip"),
             with_prior=False):
    rng = np.random.RandomState(seed)
    model = str(rng.choice(list(models)))
    psf_kind = str(rng.choice(list(psf_kinds)))
    psf_ng = 1 if psf_kind == "gauss" else 2
    nband = int(rng.randint(1, 4))
    nobj = int(rng.randint(3, 7))
    psf_ntry, ntry = int(rng.randint(1, 3)), int(rng.randint(1, 3))
    psf_fp = dict(FP, maxfev=10 if psf_kind == "gauss" else 150)
    obj_fp = dict(FP, maxfev=10 if model in ("exp", "gauss", "dev") else 400)
    npars = 5 + nband
    objs, tables, obj_guess = [], [], []
    images, sigmas, jacs, pimages, pjacs, sobj, sband, psf_guess = [], [], [], [], [], [], [], []
    for i in range(nobj):
        truth = np.concatenate([rng.uniform(-0.08, 0.08, 2), rng.uniform(-0.2, 0.2, 2),
                                [rng.uniform(0.3, 0.7)], rng.uniform(80.0, 300.0, nband)])
        psf_true = ngmix.GMixModel([0.0, 0.0, rng.uniform(-0.03, 0.03), rng.uniform(-0.03, 0.03),
                                    rng.uniform(0.26, 0.34), 1.0], "turb")
        mb = ngmix.MultiBandObsList()
        table = {}
        for b in range(nband):
            ol = ngmix.ObsList()
            for e in range(int(rng.randint(1, 4))):
                jac = ngmix.DiagonalJacobian(row=(DIM - 1) / 2 + rng.uniform(-0.5, 0.5),
                                             col=(DIM - 1) / 2 + rng.uniform(-0.5, 0.5), scale=SCALE)
                pjac = ngmix.DiagonalJacobian(row=(PDIM - 1) / 2 + rng.uniform(-0.3, 0.3),
                                              col=(PDIM - 1) / 2 + rng.uniform(-0.3, 0.3),
                                              scale=SCALE)
                pb = np.concatenate([truth[:5], [truth[5 + b]]])
                im = ngmix.GMixModel(pb, model).convolve(psf_true).make_image((DIM, DIM),
                                                                              jacobian=jac)
                sigma = truth[5 + b] / rng.uniform(150.0, 500.0)
                im = im + sigma * rng.normal(size=im.shape)
                u = rng.uniform()
                if u < 0.12:
                    pim = np.full((PDIM, PDIM), 0.01)        # no star: a plateau
                elif u < 0.2:
                    pim = np.zeros((PDIM, PDIM))             # no star: noise
                else:
                    pim = psf_true.make_image((PDIM, PDIM), jacobian=pjac)
                pim = pim + 2.0e-4 * rng.normal(size=pim.shape)
                pobs = ngmix.Observation(pim, weight=np.full(pim.shape, 1.0 / 2.0e-4 ** 2),
                                         jacobian=pjac)
                ol.append(ngmix.Observation(im, weight=np.full(im.shape, 1.0 / sigma ** 2),
                                            jacobian=jac, psf=pobs))
                tries = []
                for t in range(psf_ntry):
                    far = t == 0 and rng.uniform() < 0.15
                    T = 20.0 if far else 0.3 * rng.uniform(0.9, 1.1)
                    cen = np.array([2.0, -2.0]) if far else rng.uniform(-0.02, 0.02, 2)
                    if psf_kind == "gauss":
                        g = np.array([cen[0], cen[1], rng.uniform(-0.02, 0.02),
                                      rng.uniform(-0.02, 0.02), T, rng.uniform(0.9, 1.1)])
                    else:
                        g = np.array([cen[0], cen[1], rng.uniform(-0.02, 0.02),
                                      rng.uniform(-0.02, 0.02), T * 0.58 * rng.uniform(0.95, 1.05),
                                      T * 1.62 * rng.uniform(0.95, 1.05), 0.6 * rng.uniform(0.9, 1.1),
                                      0.4 * rng.uniform(0.9, 1.1)])
                    tries.append(g)
                table[id(pobs)] = tries
                psf_guess.append(tries)
                images.append(im)
                sigmas.append(sigma)
                jacs.append(jac.get_data().copy())
                pimages.append(pim)
                pjacs.append(pjac.get_data().copy())
                sobj.append(i)
                sband.append(b)
            mb.append(ol)
        og = []
        for t in range(ntry):
            g = truth * rng.uniform(0.85, 1.15, size=npars)
            g[0:2] = truth[0:2] + rng.uniform(-0.03, 0.03, 2)
            g[2:4] = truth[2:4] + rng.uniform(-0.05, 0.05, 2)
            if t == 0 and rng.uniform() < 0.25:
                g[4] = truth[4] * 30.0
                g[0:2] = truth[0:2] + np.array([1.4, -1.3])
                g[5:] = truth[5:] * 0.02
            og.append(g)
        objs.append(mb)
        tables.append(table)
        obj_guess.append(og)

    prior = _joint_prior(np.random.RandomState(seed ^ 0x5bd1), nband) if with_prior else None
    # ---- (a) object by object
    ref = []
    for i, mb in enumerate(objs):
        if psf_kind == "gauss":
            pf = ngmix.fitting.Fitter(model="gauss", fit_pars=psf_fp, batched=False)
        else:
            pf = ngmix.fitting.CoellipFitter(ngauss=2, fit_pars=psf_fp, batched=False)
        psf_runner = PSFRunner(fitter=pf, guesser=Stored(table=tables[i]), ntry=psf_ntry)
        guesser = Stored(seq=obj_guess[i])
        runner = Runner(fitter=ngmix.fitting.Fitter(model=model, fit_pars=obj_fp, batched=False,
                                                    prior=prior),
                        guesser=guesser, ntry=ntry)
        try:
            res = Bootstrapper(runner=runner, psf_runner=psf_runner).go(mb)
            failed = False
        except BootPSFFailure:
            res, failed = None, True
        flat = [o for ol in mb for o in ol]
        ref.append(dict(failed=failed, res=res, ntry=guesser.k,
                        psf_flags=[o.psf.meta["result"]["flags"] for o in flat],
                        psf_nfev=[o.psf.meta["result"]["nfev"] for o in flat],
                        psf_ntry=[psf_runner.guesser.count[id(o.psf)] for o in flat]))

    # ---- (b) one batch
    images, pimages = np.array(images), np.array(pimages)
    sig = np.array(sigmas)
    sb = StampBatch.from_images(images, np.ones_like(images) / sig[:, None, None] ** 2,
                                np.concatenate(jacs))
    psb = StampBatch.from_images(pimages, np.full(pimages.shape, 1.0 / 2.0e-4 ** 2),
                                 np.concatenate(pjacs))
    guess = np.array([[og[t] for og in obj_guess] for t in range(ntry)])
    res = bootstrap_batch(sb, psb, model=model, psf_fitter=psf_kind, psf_ngauss=psf_ng,
                          psf_ntry=psf_ntry, ntry=ntry, psf_guess=np.array(psf_guess).transpose(1, 0, 2),
                          psf_fit_pars=psf_fp, fit_pars=obj_fp, guess=guess, prior=prior,
                          stamp_obj=np.array(sobj), stamp_band=np.array(sband))
    return dict(model=model, psf_kind=psf_kind, nobj=nobj, ref=ref, res=res,
                sobj=np.array(sobj), lmder=model in ("exp", "gauss", "dev"))


def compare(case, stats, seed):
    res, sobj = case["res"], case["sobj"]
    for i, r in enumerate(case["ref"]):
        st = np.nonzero(sobj == i)[0]
        stats["objects"] += 1
        stats["stamps"] += st.size
        stats["by_kind"][case["psf_kind"]][0] += 1
        bad = []
        pf = np.array(r["psf_flags"])
        if np.any((res["psf_flags"][st] == 0) != (pf == 0)):
            bad.append("psf pass/fail")
        if np.any(res["psf_ntry"][st] != np.array(r["psf_ntry"])):
            bad.append("psf attempts")
        if case["psf_kind"] == "gauss" and np.any(res["psf_nfev"][st] != np.array(r["psf_nfev"])):
            bad.append("psf nfev")
        if bool(res["boot_failed"][i]) != r["failed"]:
            bad.append("BootPSFFailure")
        stats["dropped"] += int((pf != 0).sum())
        stats["boot_failed"] += int(r["failed"])
        if not r["failed"] and not res["boot_failed"][i]:
            one = r["res"]
            if int(res["flags"][i]) != int(one["flags"]):
                bad.append("flags %d / %d" % (one["flags"], res["flags"][i]))
            if int(res["ntry"][i]) != r["ntry"]:
                bad.append("attempts")
            stats["retried"] += int(r["ntry"] > 1)
            if one["flags"] == 0 and res["flags"][i] == 0:
                dn = abs(int(res["nfev"][i]) - int(one["nfev"]))
                if case["lmder"] and case["psf_kind"] == "gauss":
                    if dn:
                        bad.append("nfev %d / %d" % (one["nfev"], res["nfev"][i]))
                elif dn > 2 * (res["pars"].shape[1] + 1):
                    bad.append("nfev %d / %d" % (one["nfev"], res["nfev"][i]))
                d = float(np.max(np.abs(res["pars"][i] - one["pars"]) / one["pars_err"]))
                if not bad:      # (same epochs kept, same attempts: the same fit)
                    stats["worst"] = max(stats["worst"], d)
                    k = case["psf_kind"]
                    stats["worst_kind"][k] = max(stats["worst_kind"][k], d)
                if d > (1e-4 if case["lmder"] and case["psf_kind"] == "gauss" else 5e-2):
                    bad.append("pars %.2e sigma" % d)
        elif res["boot_failed"][i] and res["flags"][i] != BOOT_PSF_FAILURE:
            bad.append("flag of a lost object")
        if bad:
            stats["by_kind"][case["psf_kind"]][1] += 1
            stats["odd"].append((seed, i, case["model"], case["psf_kind"], bad))


def new_stats():
    return dict(objects=0, stamps=0, dropped=0, boot_failed=0, retried=0, worst=0.0, odd=[],
                by_kind={"gauss": [0, 0], "coellip": [0, 0]},
                worst_kind={"gauss": 0.0, "coellip": 0.0})


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    master = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 17)
    with_prior = len(sys.argv) > 3 and sys.argv[3] == "prior"
    stats = new_stats()
    t0 = time.time()
    ncase = 0
    while time.time() - t0 < budget:
        seed = int(master.randint(1 << 30))
        compare(one_case(seed, with_prior=with_prior), stats, seed)
        ncase += 1
    if with_prior:
        print("(every object fit carries a random PriorSimpleSep of ngmix_amd.priors terms: the prior kernel "
              "in the batch, prior.fill_fdiff on the host object by object)")
    print("fuzz_boot: %.0f s, %d cases, %d objects, %d stamps; psf fits failed (epochs dropped): %d, "
          "objects lost to BootPSFFailure: %d, objects that needed a second attempt: %d; objects "
          "that differ from the per-object Bootstrapper: %d (psf fitter 'gauss', lmder: %d of %d; "
          "'coellip' with 2 gaussians, lmdif under a 150-evaluation cap: %d of %d); largest "
          "|dpars| / pars_err among the objects that agree in everything else: %.2e with the "
          "lmder psf fitter, %.2e with the lmdif one"
          % (time.time() - t0, ncase, stats["objects"], stats["stamps"], stats["dropped"],
             stats["boot_failed"], stats["retried"], len(stats["odd"]),
             stats["by_kind"]["gauss"][1], stats["by_kind"]["gauss"][0],
             stats["by_kind"]["coellip"][1], stats["by_kind"]["coellip"][0],
             stats["worst_kind"]["gauss"], stats["worst_kind"]["coellip"]))
    for kind in ("gauss", "coellip"):
        for rec in [r for r in stats["odd"] if r[3] == kind][:10]:
            print("   differs: seed %d object %d (%s, psf %s): %s" % rec)


if __name__ == "__main__":
    main()
