"""host profile of the per-object calls (cProfile, 300 calls each).
python tools/prof_calls.py [admom|gaussmom|em|loglike|fdiff|image]"""
import cProfile
import os
import pstats
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngmix_amd as ngmix  # noqa: E402

rng = np.random.RandomState(1)
jac = ngmix.DiagonalJacobian(row=23.5, col=23.5, scale=0.263)
pgm = ngmix.GMixModel([0.0, 0.0, 0.0, 0.0, 0.27, 1.0], "gauss")
gm = ngmix.GMixModel([0.1, -0.05, 0.1, 0.05, 0.6, 100.0], "exp").convolve(pgm)
im = gm.make_image((48, 48), jacobian=jac, fast_exp=True) + 0.01 * rng.normal(size=(48, 48))
wt = np.full((48, 48), 1e4)
pj = ngmix.DiagonalJacobian(row=12, col=12, scale=0.263)
pobs = ngmix.Observation(pgm.make_image((25, 25), jacobian=pj), jacobian=pj, gmix=pgm)
obs = ngmix.Observation(im, weight=wt, jacobian=jac, psf=pobs)
calls = {
    "admom": lambda: ngmix.admom.run_admom(obs, 0.6, rng=rng),
    "gaussmom": lambda: ngmix.GaussMom(fwhm=1.2).go(obs),
    "em": lambda: ngmix.em.run_em(pobs, ngmix.GMixModel([0.0, 0.0, 0.0, 0.0, 0.3, 1.0], "gauss")),
    "loglike": lambda: gm.get_loglike(obs),
    "fdiff": lambda: gm.fill_fdiff(obs, np.zeros(2304)),
    "image": lambda: gm.make_image((48, 48), jacobian=jac, fast_exp=True),
}
for name in (sys.argv[1:] or list(calls)):
    fn = calls[name]
    for _ in range(20):
        fn()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        fn()
    pr.disable()
    print("=" * 20, name)
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
