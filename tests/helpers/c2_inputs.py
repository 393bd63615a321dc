"""
The inputs of tests/golden/c2.npz, numpy only (as helpers/c5_inputs.py for
config 5): eight stamps of config 2's shape -- 48x48, an 'exp' model (6
gaussians after the gaussian psf), centres within half a pixel, g ~ N(0, 0.1),
T ~ U(0.3, 1.5), flux ~ U(50, 500), noise 0.01 flux / 100, uniform weight --
and the image each render accumulates into.
"""
import numpy as np

N, DIM, SCALE, TPSF = 8, 48, 0.263, 0.27


def stamps():
    """(pars (N, 6), moved pars, jacobians (N, 8), images (N, DIM, DIM), sigma (N,),
    base images the renders add to (N, DIM, DIM))"""
    rng = np.random.RandomState(2002)
    pars = np.zeros((N, 6))
    pars[:, 0:2] = rng.uniform(-0.5, 0.5, size=(N, 2)) * SCALE
    pars[:, 2:4] = rng.normal(scale=0.1, size=(N, 2)).clip(-0.5, 0.5)
    pars[:, 4] = rng.uniform(0.3, 1.5, size=N)
    pars[:, 5] = rng.uniform(50, 500, size=N)
    moved = pars * rng.uniform(0.9, 1.1, size=pars.shape)
    moved[:, 0:4] = pars[:, 0:4] + rng.uniform(-0.03, 0.03, size=(N, 4))
    jac = np.zeros((N, 8))
    jac[:, 0] = (DIM - 1) / 2 + rng.uniform(-0.5, 0.5, size=N)
    jac[:, 1] = (DIM - 1) / 2 + rng.uniform(-0.5, 0.5, size=N)
    jac[:, 2] = jac[:, 5] = jac[:, 7] = SCALE
    jac[:, 6] = SCALE ** 2
    sigma = 0.01 * pars[:, 5] / 100.0
    images = np.zeros((N, DIM, DIM))
    for i in range(N):
        v = ((np.arange(DIM) - jac[i, 0]) * SCALE)[:, None] - pars[i, 0]
        u = ((np.arange(DIM) - jac[i, 1]) * SCALE)[None, :] - pars[i, 1]
        for frac, grow in ((0.6, 0.6), (0.4, 2.5)):
            s2 = 0.5 * (pars[i, 4] + TPSF) * grow
            images[i] += frac * pars[i, 5] * np.exp(-0.5 * (u * u + v * v) / s2) \
                / (2 * np.pi * s2) * SCALE ** 2
        images[i] += sigma[i] * rng.normal(size=(DIM, DIM))
    base = rng.normal(size=(N, DIM, DIM))
    return pars, moved, jac, images, sigma, base
