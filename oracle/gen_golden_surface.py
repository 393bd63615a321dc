#!/usr/bin/env python
"""
The public names of the reference's modules on (and either side of) the pixel
hot path -- module-level functions and classes, and each class's public
attributes -- read off the REFERENCE ITSELF (imported under the numba shim)
into tests/golden/api_surface.json, and the parameter NAMES of every such
function, constructor and method into tests/golden/api_signatures.json.  tests/test_host_logic.py checks that
ngmix_amd offers every one of them except a short, explicit out-of-scope list
(galsim-backed and k-space objects).  Names only: no reference source travels.
Build container only.  TEST INFRASTRUCTURE ONLY.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_surface.py
"""
import inspect
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "shim"), "/root/reference"]

import ngmix  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "api_surface.json")
OUT_SIG = os.path.join(os.path.dirname(HERE), "tests", "golden", "api_signatures.json")
MODULES = ["gmix", "observation", "jacobian", "shape", "moments", "admom", "em", "fitting",
           "gaussmom", "guessers", "runners", "bootstrap", "pixels", "flags", "gexceptions",
           "util", "priors", "joint_prior", "gaussap", "simobs", "fastexp_nb", "gmix_ndim", "prepsfmom", "ksigmamom"]


def public(obj):
    return sorted(n for n in dir(obj) if not n.startswith("_"))


def params(f):
    try:
        return [p.name for p in inspect.signature(f).parameters.values() if p.name != "self"]
    except (ValueError, TypeError):
        return None


def class_params(cls):
    out = {"__init__": params(cls.__init__)}
    for a in public(cls):
        f = inspect.getattr_static(cls, a)
        if isinstance(f, (staticmethod, classmethod)):
            f = f.__func__
        if inspect.isfunction(f):
            out[a] = params(f)
    return out


def main():
    out = {}
    sigs = {}
    for m in MODULES:
        mod = getattr(ngmix, m)
        names = {}
        for n in public(mod):
            o = getattr(mod, n)
            # defined in this module (or, for a package, under it): names a
            # module merely imports are listed where they are defined
            if not (getattr(o, "__module__", "") + ".").startswith("ngmix.%s." % m):
                continue
            if inspect.isclass(o):
                names[n] = public(o)
                sigs["%s.%s" % (m, n)] = class_params(o)
            elif inspect.isfunction(o):
                names[n] = "function"
                sigs["%s.%s" % (m, n)] = params(o)
        out[m] = names
    out["__top__"] = public(ngmix)
    with open(OUT, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    with open(OUT_SIG, "w") as f:
        json.dump(sigs, f, indent=0, sort_keys=True)
    print("wrote", OUT_SIG, len(sigs))
    print("wrote", OUT, {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
