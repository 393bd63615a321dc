#!/usr/bin/env python
"""
Golden vectors for ngmix_amd/gaussap.py and ngmix_amd/simobs.py from the
REFERENCE's ngmix.gaussap.get_gaussap_flux and ngmix.simobs (under the numba
shim): aperture fluxes and flags for every model incl. 'cm' and 'bdf', several
bands, masked objects and |g| >= 1 rows; noise images for weight maps with
holes / all zero / a noise factor; simulated observations of a mixture through
a psf (Observation, ObsList, MultiBandObsList; weight_raw; gmix None).
-> tests/golden/utils.npz.  Build container only.  TEST INFRASTRUCTURE ONLY.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_utils.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "shim"), "/root/reference"]

import numpy as np  # noqa: E402
import ngmix  # noqa: E402
from ngmix import gaussap, simobs  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "utils.npz")


def gap_cases(out):
    rng = np.random.RandomState(8811)
    for model, nband in (("gauss", 1), ("exp", 3), ("dev", 1), ("turb", 2), ("bdf", 1),
                         ("bdf", 3), ("cm", 2)):
        n = 30
        nloc = 7 if model == "bdf" else 6
        pars = np.zeros((n, nloc - 1 + nband))
        pars[:, 0:2] = rng.normal(scale=0.1, size=(n, 2))
        pars[:, 2:4] = rng.normal(scale=0.25, size=(n, 2))
        pars[:, 4] = rng.uniform(-0.05, 2.0, size=n)      # some below the 1e-4 floor
        if model == "bdf":
            pars[:, 5] = rng.uniform(-0.2, 1.2, size=n)
        pars[:, nloc - 1:] = rng.uniform(-5.0, 300.0, size=(n, nband))
        pars[3, 2:4] = (0.8, 0.7)                           # |g| >= 1
        pars[11, 2:4] = (1.0, 0.0)
        mask = np.ones(n, dtype=bool)
        mask[[5, 17]] = False
        kw = {}
        if model == "cm":
            kw = dict(fracdev=rng.uniform(0.0, 1.0, size=n), TdByTe=rng.uniform(0.3, 3.0, size=n))
            out["gap_%s%d_fracdev" % (model, nband)] = kw["fracdev"]
            out["gap_%s%d_TdByTe" % (model, nband)] = kw["TdByTe"]
        tag = "gap_%s%d" % (model, nband)
        out[tag + "_pars"] = pars
        out[tag + "_mask"] = mask
        for fwhm in (0.9, 2.5):
            flux, flags = gaussap.get_gaussap_flux(pars, model, fwhm, mask=mask, verbose=False, **kw)
            out["%s_flux_%s" % (tag, fwhm)] = flux
            out["%s_flags_%s" % (tag, fwhm)] = flags
        flux, flags = gaussap.get_gaussap_flux(pars, model, 1.2, verbose=False, **kw)
        out[tag + "_flux_nomask"] = flux
        out[tag + "_flags_nomask"] = flags
    # a single parameter vector
    flux, flags = gaussap.get_gaussap_flux([0.0, 0.0, 0.1, 0.2, 0.5, 10.0], "exp", 1.5)
    out["gap_single_flux"], out["gap_single_flags"] = flux, flags


def noise_cases(out):
    rng = np.random.RandomState(77)
    w = np.abs(rng.normal(size=(8, 9))) + 0.2
    holes = w.copy()
    holes[2, 3] = 0.0
    holes[5, :2] = -1.0
    out["noise_w"], out["noise_holes"] = w, holes
    cases = {"plain": (w, {}), "holes_all": (holes, {}), "holes_notall": (holes, {"add_all": False}),
             "factor": (holes, {"noise_factor": 1.7}), "zero": (np.zeros((4, 5)), {})}
    for name, (wt, kw) in cases.items():
        out["noise_" + name] = simobs.get_noise_image(wt, np.random.RandomState(123), **kw)


def sim_cases(out):
    dim = 15
    jac = ngmix.Jacobian(row=7.2, col=6.9, dvdrow=0.26, dvdcol=0.01, dudrow=-0.015, dudcol=0.27)
    psf_gm = ngmix.GMixModel([0.0, 0.0, 0.02, -0.01, 0.3, 1.0], "turb")
    gm = ngmix.GMixModel([0.05, -0.08, 0.15, -0.1, 0.5, 40.0], "exp")
    gm2 = ngmix.GMixModel([0.05, -0.08, 0.15, -0.1, 0.5, 70.0], "dev")
    rng = np.random.RandomState(5)
    w = np.abs(rng.normal(size=(dim, dim))) * 50 + 20
    w[4, 4] = 0.0
    out["sim_jac"] = jac.get_data().copy()
    out["sim_weight"] = w
    out["sim_psf_pars"] = psf_gm.get_full_pars()
    out["sim_gm_pars"] = gm.get_full_pars()
    out["sim_gm2_pars"] = gm2.get_full_pars()

    def obs():
        pobs = ngmix.Observation(np.zeros((dim, dim)) + 1.0, jacobian=jac, gmix=psf_gm.copy())
        return ngmix.Observation(np.zeros((dim, dim)), weight=w.copy(), jacobian=jac, psf=pobs)
    o = simobs.simulate_obs(gm, obs(), add_noise=False)
    out["sim_model"] = o.image
    out["sim_model_weight"] = o.weight
    o = simobs.simulate_obs(gm, obs(), add_noise=False, convolve_psf=False)
    out["sim_model_nopsf"] = o.image
    o = simobs.simulate_obs(gm, obs(), rng=np.random.RandomState(9))
    out["sim_noisy"], out["sim_noise_image"] = o.image, o.noise_image
    o = simobs.simulate_obs(gm, obs(), rng=np.random.RandomState(9), noise_factor=2.0, add_all=False)
    out["sim_noisy_f2"], out["sim_weight_f2"] = o.image, o.weight
    raw = obs()
    raw.weight_raw = w * 4.0
    o = simobs.simulate_obs(gm, raw, rng=np.random.RandomState(9))
    out["sim_noisy_raw"] = o.image
    o = simobs.simulate_obs(gm, raw, rng=np.random.RandomState(9), use_raw_weight=False)
    out["sim_noisy_raw_unused"] = o.image
    o = simobs.simulate_obs(None, obs(), rng=np.random.RandomState(9))
    out["sim_pure_noise"] = o.image
    ol = ngmix.ObsList()
    ol.append(obs())
    ol.append(obs())
    r = simobs.simulate_obs(gm, ol, rng=np.random.RandomState(10))
    out["sim_obslist"] = np.array([x.image for x in r])
    mb = ngmix.MultiBandObsList()
    mb.append(ol)
    mb.append(ol)
    r = simobs.simulate_obs([gm, gm2], mb, rng=np.random.RandomState(11))
    out["sim_mb"] = np.array([[x.image for x in band] for band in r])


def main():
    out = {}
    gap_cases(out)
    noise_cases(out)
    sim_cases(out)
    np.savez_compressed(OUT, **out)
    print("wrote %s (%d arrays, %.1f kB)" % (OUT, len(out), os.path.getsize(OUT) / 1e3))


if __name__ == "__main__":
    main()
