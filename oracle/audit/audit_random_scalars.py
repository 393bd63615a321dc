from common import *
"""every scalar conversion of moments.py / shape.py on many random inputs (scalars and arrays), bit for bit,
exceptions included"""
import inspect
rng = np.random.RandomState(123)
N = 400


def draw(kind, size=None):
    if kind == "g":      # a shear component, sometimes out of range
        return rng.uniform(-0.75, 0.75, size=size)
    if kind == "pos":    # a size, sometimes <= 0
        return rng.uniform(-0.1, 3.0, size=size)
    if kind == "any":
        return rng.normal(scale=2.0, size=size)
    if kind == "small":
        return rng.uniform(-0.05, 0.05, size=size)
    if kind == "angle":
        return rng.uniform(-4.0, 4.0, size=size)
    raise ValueError(kind)


SPEC = {
    "moments.fwhm_to_sigma": ["pos"], "moments.fwhm_to_T": ["pos"], "moments.sigma_to_fwhm": ["pos"],
    "moments.T_to_fwhm": ["pos"], "moments.r50_to_sigma": ["pos"], "moments.sigma_to_r50": ["pos"],
    "moments.r50_to_T": ["pos"], "moments.T_to_r50": ["pos"],
    "moments.moms_to_e1e2": ["any", "any", "pos"], "moments.e2mom": ["g", "g", "pos"],
    "moments.g2mom": ["g", "g", "pos"], "moments.mom2e": ["pos", "any", "pos"],
    "moments.mom2g": ["pos", "small", "pos"], "moments.get_Tround": ["pos", "g", "g"],
    "moments.get_T": ["pos", "g", "g"],
    "moments.get_sheared_M1M2T": ["small", "small", "pos", "small", "small"],
    "moments.get_sheared_g1g2T": ["g", "g", "pos", "small", "small"],
    "moments.get_sheared_moments": ["pos", "small", "pos", "small", "small"],
    "shape.g1g2_to_e1e2": ["g", "g"], "shape.e1e2_to_g1g2": ["g", "g"],
    "shape.e1e2_to_eta1eta2": ["g", "g"], "shape.eta1eta2_to_g1g2": ["any", "any"],
    "shape.g1g2_to_eta1eta2": ["g", "g"], "shape.shear_reduced": ["g", "g", "small", "small"],
    "shape.dgs_by_dgo_jacob": ["g", "g", "small", "small"], "shape.get_round_factor": ["g", "g"],
    "shape.rotate_shape": ["g", "g", "angle"],
}
for name, kinds in SPEC.items():
    m, f = name.split(".")
    fr, fo = getattr(getattr(ref, m), f), getattr(getattr(ours, m), f)
    before = ndiff[0]
    for _ in range(N):
        args = [float(draw(k)) for k in kinds]
        run(name, fr, fo, *args)
        if ndiff[0] - before >= 3:
            break
    for _ in range(20):
        args = [draw(k, size=5) for k in kinds]
        run(name + "[array]", fr, fo, *args)
        if ndiff[0] - before >= 5:
            break
    print("%-32s %s" % (name, "same" if ndiff[0] == before else "DIFFERS (%d)" % (ndiff[0] - before)))
print("ndiff", ndiff[0])
