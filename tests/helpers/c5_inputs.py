"""
The inputs of tests/golden/c5.npz, numpy only: oracle/gen_golden_c5.py feeds
them to the reference, the tests rebuild them and compare with the outputs it
stored (an image that differs in a last bit of numpy's exp between two
machines moves a loglike by 1e-16 of itself; the tests hold 1e-10).

Config 5's shape: objects of ten 64x64 epochs, a 7-parameter 'bdf' model
(16 gaussians) (x) a gaussian psf, every epoch with its own sub-pixel jacobian
offset (ngmix/tests/_sims.py:150-159), noise 0.05, weight 400.
"""
import numpy as np

NOBJ, NEPOCH, DIM, SCALE, NOISE, TPSF = 6, 10, 64, 0.263, 0.05, 0.27


def objects():
    """(pars (NOBJ, 7), perturbed pars (NOBJ, 7), jacobians (NOBJ, NEPOCH, 8),
    images (NOBJ, NEPOCH, DIM, DIM))"""
    rng = np.random.RandomState(5005)
    pars = np.zeros((NOBJ, 7))
    pars[:, 0:2] = rng.uniform(-0.3, 0.3, size=(NOBJ, 2)) * SCALE
    pars[:, 2:4] = rng.normal(scale=0.08, size=(NOBJ, 2))
    pars[:, 4] = rng.uniform(0.5, 2.0, size=NOBJ)
    pars[:, 5] = rng.uniform(0.2, 0.8, size=NOBJ)
    pars[:, 6] = rng.uniform(100, 400, size=NOBJ)
    moved = pars * rng.uniform(0.9, 1.1, size=pars.shape)
    moved[:, 0:4] = pars[:, 0:4] + rng.uniform(-0.03, 0.03, size=(NOBJ, 4))
    jac = np.zeros((NOBJ, NEPOCH, 8))
    jac[:, :, 0] = (DIM - 1) / 2 + rng.uniform(-0.5, 0.5, size=(NOBJ, NEPOCH))
    jac[:, :, 1] = (DIM - 1) / 2 + rng.uniform(-0.5, 0.5, size=(NOBJ, NEPOCH))
    jac[:, :, 2] = jac[:, :, 5] = jac[:, :, 7] = SCALE
    jac[:, :, 6] = SCALE ** 2
    images = np.zeros((NOBJ, NEPOCH, DIM, DIM))
    for o in range(NOBJ):
        T, flux = pars[o, 4] + TPSF, pars[o, 6]
        for e in range(NEPOCH):
            v = ((np.arange(DIM) - jac[o, e, 0]) * SCALE)[:, None] - pars[o, 0]
            u = ((np.arange(DIM) - jac[o, e, 1]) * SCALE)[None, :] - pars[o, 1]
            im = np.zeros((DIM, DIM))
            # (a two-gaussian stand-in of the profile: the data need not be the
            # model's own image, the loglike of any image is as good a check)
            for frac, grow in ((0.6, 0.7), (0.4, 3.0)):
                s2 = 0.5 * T * grow
                im += frac * flux * np.exp(-0.5 * (u * u + v * v) / s2) / (2 * np.pi * s2) \
                    * SCALE ** 2
            images[o, e] = im + NOISE * rng.normal(size=im.shape)
    return pars, moved, jac, images
