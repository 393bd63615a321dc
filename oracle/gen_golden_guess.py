#!/usr/bin/env python
"""
Golden vectors for ngmix_amd/guessers.py: the draws of the REFERENCE's own
guessers (ngmix/guessers.py) from seeded RandomStates -- every guesser that
needs no joint prior: TFluxGuesser, TPSFFluxGuesser, ParsGuesser,
R50FluxGuesser, GMixPSFGuesser (1-5 gaussians, from the image sum and from
weighted moments), SimplePSFGuesser, CoellipPSFGuesser (1-5).  Each case stores
the seed, the inputs and the sequence of guesses successive calls return.
Build container only; tests/golden/guess.npz is committed.  TEST
INFRASTRUCTURE ONLY.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_guess.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "shim"), "/root/reference"]

import numpy as np  # noqa: E402
import ngmix  # noqa: E402
from ngmix import guessers as G  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "guess.npz")
SCALE = 0.263


def psf_obs(rng, dim=25):
    jac = ngmix.DiagonalJacobian(row=12.1, col=11.7, scale=SCALE)
    gm = ngmix.GMixModel([0.01, -0.02, 0.03, -0.02, 0.31, 1.3], "turb")
    im = gm.make_image((dim, dim), jacobian=jac) + 2.0e-4 * rng.normal(size=(dim, dim))
    return ngmix.Observation(im, weight=np.full(im.shape, 1.0 / 2.0e-4 ** 2), jacobian=jac), jac


def main():
    out = {}
    rng0 = np.random.RandomState(4242)
    pobs, pjac = psf_obs(rng0)
    out["psf_image"] = pobs.image.copy()
    out["psf_jac"] = pjac.get_data().copy()
    out["psf_sigma"] = np.array(2.0e-4)

    # ---- no observation needed
    g = G.TFluxGuesser(np.random.RandomState(11), 0.45, 120.0)
    out["tflux1"] = np.array([g() for _ in range(3)])
    g = G.TFluxGuesser(np.random.RandomState(12), 0.45, [120.0, 80.0, 33.0])
    out["tflux3"] = np.array([g() for _ in range(3)])
    out["tflux3_n4"] = g(nrand=4)
    g = G.R50FluxGuesser(np.random.RandomState(13), 0.7, [50.0, 60.0])
    out["r50"] = np.array([g() for _ in range(2)])
    out["r50_n3"] = g(nrand=3)
    pars = np.array([0.1, -0.2, 0.55, -0.62, 0.8, 210.0, 90.0])
    g = G.ParsGuesser(np.random.RandomState(14), pars)
    out["pars_in"] = pars
    out["pars_scalar"] = np.array([g() for _ in range(3)])
    out["pars_n5"] = g(nrand=5)
    widths = np.array([0.01, 0.01, 0.3, 0.3, 0.2, 0.05, 0.05])
    g = G.ParsGuesser(np.random.RandomState(15), pars, widths=widths)
    out["pars_widths"] = widths
    out["pars_w_n4"] = g(nrand=4)

    # ---- psf guessers on the psf observation
    for moms in (0, 1):
        for ng in range(1, 6):
            g = G.GMixPSFGuesser(np.random.RandomState(100 + ng), ng, guess_from_moms=bool(moms))
            out["gmixpsf_m%d_ng%d" % (moms, ng)] = np.array(
                [g(pobs).get_full_pars() for _ in range(2)])
            g = G.CoellipPSFGuesser(np.random.RandomState(200 + ng), ng,
                                    guess_from_moms=bool(moms))
            out["coellip_m%d_ng%d" % (moms, ng)] = np.array([g(pobs) for _ in range(2)])
        g = G.SimplePSFGuesser(np.random.RandomState(300), guess_from_moms=bool(moms))
        out["simplepsf_m%d" % moms] = np.array([g(pobs) for _ in range(3)])

    # ---- psf fluxes: a 2-band object, 2 + 1 epochs, psf mixtures set
    rng = np.random.RandomState(77)
    psf_gm = ngmix.GMixModel([0.0, 0.0, 0.02, -0.01, 0.28, 1.0], "gauss")
    truth = np.array([0.05, -0.03, 0.1, 0.2, 0.5, 150.0, 60.0])
    mb = ngmix.MultiBandObsList()
    images, sigmas, jacs, bands = [], [], [], []
    for b, nep in enumerate((2, 1)):
        ol = ngmix.ObsList()
        for e in range(nep):
            jac = ngmix.DiagonalJacobian(row=15.5 + rng.uniform(-0.4, 0.4),
                                         col=15.5 + rng.uniform(-0.4, 0.4), scale=SCALE)
            pb = np.concatenate([truth[:5], [truth[5 + b]]])
            im = ngmix.GMixModel(pb, "exp").convolve(psf_gm).make_image((32, 32), jacobian=jac)
            sigma = truth[5 + b] / 300.0
            im = im + sigma * rng.normal(size=im.shape)
            p = ngmix.Observation(np.zeros((5, 5)), jacobian=jac, gmix=psf_gm.copy())
            ol.append(ngmix.Observation(im, weight=np.full(im.shape, 1.0 / sigma ** 2),
                                        jacobian=jac, psf=p))
            images.append(im)
            sigmas.append(sigma)
            jacs.append(jac.get_data().copy())
            bands.append(b)
        mb.append(ol)
    out["mb_images"], out["mb_sigma"] = np.array(images), np.array(sigmas)
    out["mb_jac"], out["mb_band"] = np.concatenate(jacs), np.array(bands)
    out["mb_psf_pars"] = psf_gm.get_full_pars()
    g = G.TPSFFluxGuesser(np.random.RandomState(21), 0.5)
    out["tpsfflux"] = np.array([g(obs=mb) for _ in range(3)])
    out["tpsfflux_fluxes"] = np.array(g._psf_fluxes)
    np.savez_compressed(OUT, **out)
    print("wrote %s (%.1f kB)" % (OUT, os.path.getsize(OUT) / 1e3))


if __name__ == "__main__":
    main()
