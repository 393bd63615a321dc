#!/usr/bin/env python
"""
SURVEY.md 8(d) config C1, exactly as the survey states it, run through the
REFERENCE ITSELF (under the numba shim): one 48x48 Observation,
DiagonalJacobian(row=23.5, col=23.5, scale=0.263), GMixModel([0.1, -0.05, 0.1,
0.05, 0.6, 100], 'exp'), image = fast render + N(0, 0.01^2) from
RandomState(1), weight 1e4, and get_loglike / render AT THE TRUE MIXTURE (the
'c1_exp48' case of render_loglike.npz uses the same image but evaluates a
perturbed mixture).  The survey's numbers for this case are asserted here
before the fixture is written.  Build container only; tests/golden/c1.npz is
committed.  TEST INFRASTRUCTURE ONLY.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_c1.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "shim"), "/root/reference"]

import numpy as np  # noqa: E402
import ngmix  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "c1.npz")

# SURVEY.md 8(d), "C1 single stamp"
SURVEY_C1 = (-1158.1127300983387, 2798242.571962917, 2798842.226964671, 2304)


def main():
    jac = ngmix.DiagonalJacobian(row=23.5, col=23.5, scale=0.263)
    pars = np.array([0.1, -0.05, 0.1, 0.05, 0.6, 100.0])
    gm = ngmix.GMixModel(pars, "exp")
    rng = np.random.RandomState(1)
    render = gm.make_image((48, 48), jacobian=jac, fast_exp=True)
    image = render + rng.normal(scale=0.01, size=render.shape)
    weight = np.full(image.shape, 1.0e4)
    obs = ngmix.Observation(image, weight=weight, jacobian=jac)
    gm_in = ngmix.GMixModel(pars, "exp")
    out = {"pars": pars, "gmix_in": gm_in.get_data().copy(),
           "jac": jac.get_data().copy(), "image": image, "weight": weight,
           "render_fast": render}
    res = gm_in.get_loglike(obs, more=True)
    ll = (float(res["loglike"]), float(res["s2n_numer"]), float(res["s2n_denom"]),
          int(res["npix"]))
    assert ll == SURVEY_C1, ll
    out["loglike"] = np.array(ll)
    out["gmix_normed"] = gm_in.get_data().copy()
    fdiff = np.zeros(image.size)
    gm_in.fill_fdiff(obs, fdiff)
    out["fdiff"] = fdiff
    out["s2n_sum"] = np.array(gm_in.get_model_s2n_sum(obs))
    np.savez_compressed(OUT, **out)
    print("wrote %s (%.1f kB)" % (OUT, os.path.getsize(OUT) / 1e3), ll)


if __name__ == "__main__":
    main()
