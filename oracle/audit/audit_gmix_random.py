from common import *
"""GMix host methods on random mixtures given by their full parameters (no model fill: the shim's numpy tanh
differs from libm's by an ulp), bit for bit: getters, setters, scale_T, get_sheared, make_round, convolve,
gaussian-aperture flux (1e-12: two 2x2 inversions)"""
rng = np.random.RandomState(77)


def mix(ng):
    pars = np.zeros(6 * ng)
    for i in range(ng):
        irr, icc = rng.uniform(0.1, 2.0, size=2)
        irc = rng.uniform(-0.9, 0.9) * np.sqrt(irr * icc)
        pars[6 * i:6 * i + 6] = [rng.uniform(0.1, 3.0), rng.normal(scale=0.3), rng.normal(scale=0.3),
                                 irr, irc, icc]
    return pars


for trial in range(150):
    ng = 1 + trial % 6
    pars = mix(ng)
    gr, go = ref.GMix(pars=pars), ours.GMix(pars=pars)
    for meth in ("get_cen", "get_T", "get_sigma", "get_e1e2T", "get_g1g2T", "get_e1e2sigma",
                 "get_g1g2sigma", "get_flux", "get_psum", "get_full_pars", "copy"):
        run(meth, getattr(gr, meth), getattr(go, meth))
    s1, s2 = rng.uniform(-0.1, 0.1, size=2)
    run("get_sheared", gr.get_sheared, go.get_sheared, s1, s2)
    run("make_round", gr.make_round, go.make_round)
    run("make_round(True)", gr.make_round, go.make_round, preserve_size=True)
    ppars = mix(1 + trial % 3)
    run("convolve", lambda: gr.convolve(ref.GMix(pars=ppars)), lambda: go.convolve(ours.GMix(pars=ppars)))
    for meth, args in (("set_cen", tuple(rng.normal(size=2))), ("set_flux", (rng.uniform(0.5, 9.0),)),
                       ("set_psum", (rng.uniform(0.5, 9.0),)), ("scale_T", (rng.uniform(0.3, 2.5),))):
        try:
            getattr(gr, meth)(*args); er = None
        except Exception as e:      # noqa: BLE001
            er = type(e).__name__
        try:
            getattr(go, meth)(*args); eo = None
        except Exception as e:      # noqa: BLE001
            eo = type(e).__name__
        if er != eo:
            print("DIFF exc", meth, er, eo); ndiff[0] += 1
        run("after " + meth, gr.get_data, go.get_data)
    a, b = gr.get_gaussap_flux(fwhm=1.3), go.get_gaussap_flux(fwhm=1.3)
    if not np.isclose(a, b, rtol=1e-12):
        print("DIFF gaussap", a, b); ndiff[0] += 1
    if ndiff[0] > 12:
        break
run("scale_T(<0)", lambda: ref.GMix(pars=mix(2)).scale_T(-1.0), lambda: ours.GMix(pars=mix(2)).scale_T(-1.0))
print("ndiff", ndiff[0])
