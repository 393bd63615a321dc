#!/usr/bin/env python
"""
Host-side behaviours a differential audit of ngmix_amd against the REFERENCE
(both imported side by side in the build container, round 6) found differing,
pinned as data: format_pars strings, the exception a Jacobian raises for a
missing keyword, and the mixture summary getters (get_cen / get_T / get_sigma /
get_e1e2T / get_g1g2T / get_e1e2sigma / get_g1g2sigma) to the bit on mixtures
given by their full parameters.  The reference itself produces every expected
value (under the numba shim) -> tests/golden/host6.json.  Build container
only.  TEST INFRASTRUCTURE ONLY.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_host6.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "shim"), "/root/reference"]

import numpy as np  # noqa: E402
import ngmix  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "host6.json")


def exc_name(f):
    try:
        f()
    except Exception as e:          # noqa: BLE001
        return type(e).__name__
    return None


def main():
    out = {"format_pars": [], "jacobian_errors": [], "getters": []}
    for pars, fmt in (([1.0, 2.5e-8, 3e10], None), ([1.0, 2.5e-8, 3e10], "%.3f"), ([1.5], None),
                      ([], None), ([-0.1, 0.2, 100.0, 1e-3], "%10.4e")):
        kw = {} if fmt is None else {"fmt": fmt}
        out["format_pars"].append({"pars": pars, "fmt": fmt,
                                   "expected": ngmix.util.format_pars(np.array(pars), **kw)})
    for kw in ({"row": 1.0}, {"x": 1.0}, {"row": 1.0, "col": 2.0, "dvdrow": 1.0},
               {"x": 1.0, "y": 2.0, "dudx": 1.0, "dudy": 0.0, "dvdx": 0.0}, {"dvdrow": 1.0},
               {"row": 1.0, "col": 2.0, "dvdrow": 1.0, "dvdcol": 0.0, "dudrow": 0.0, "dudcol": 1.0}):
        out["jacobian_errors"].append({"kw": kw, "expected": exc_name(lambda: ngmix.Jacobian(**kw))})
    rng = np.random.RandomState(606)
    for trial in range(12):
        ngauss = 1 + trial % 5
        pars = np.zeros(6 * ngauss)
        for i in range(ngauss):
            irr, icc = rng.uniform(0.1, 2.0, size=2)
            irc = rng.uniform(-0.9, 0.9) * np.sqrt(irr * icc)
            pars[6 * i:6 * i + 6] = [rng.uniform(0.1, 3.0), rng.normal(scale=0.3),
                                     rng.normal(scale=0.3), irr, irc, icc]
        gm = ngmix.GMix(pars=pars)
        exp = {}
        for name in ("get_cen", "get_T", "get_sigma", "get_e1e2T", "get_g1g2T", "get_e1e2sigma",
                     "get_g1g2sigma", "get_flux"):
            exp[name] = [float(v).hex() for v in np.atleast_1d(getattr(gm, name)())]
        out["getters"].append({"pars": [float(p).hex() for p in pars], "expected": exp})
    # found by running the reference's own test files against the package
    # (oracle/audit/run_reference_tests.sh)
    import io
    import logging
    sh = ngmix.Shape(0.3, -0.4)
    out["shape_g"] = float(sh.g).hex()
    sh.set_g1g2(0.1, 0.2)
    out["shape_g_after_set"] = float(sh.g).hex()
    out["shape_get_sheared_one_arg"] = exc_name(lambda: ngmix.Shape(0.1, 0.2).get_sheared(0.1))
    buf = io.StringIO()
    ngmix.print_pars(None, stream=buf)
    ngmix.print_pars([1.0, 2.0], front="x:", stream=buf)
    out["print_pars_stream"] = buf.getvalue()
    rec = []
    handler = logging.Handler()
    handler.emit = lambda r: rec.append((r.levelname, r.getMessage()))
    lg = logging.getLogger("host6")
    lg.setLevel(logging.DEBUG)
    lg.addHandler(handler)
    ngmix.print_pars([1.0, 2.0], logger=lg)
    out["print_pars_logger"] = rec
    cases = []
    rng2 = np.random.RandomState(12)
    for _ in range(40):
        M1, M2, T = rng2.normal(), rng2.normal(), rng2.uniform(-0.2, 2.0)
        try:
            e = [float(v).hex() for v in ngmix.moments.moms_to_e1e2(M1, M2, T)]
        except Exception as err:      # noqa: BLE001
            e = type(err).__name__
        cases.append({"args": [float(M1).hex(), float(M2).hex(), float(T).hex()], "expected": e})
    out["moms_to_e1e2"] = cases
    out["moms_to_e1e2_array_bad"] = exc_name(lambda: ngmix.moments.moms_to_e1e2(
        np.array([0.1]), np.array([0.2]), np.array([-0.1])))
    with open(OUT, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
