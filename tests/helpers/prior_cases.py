"""
The calls that pin ngmix_amd.priors / ngmix_amd.joint_prior to the reference:
ONE script of constructor arguments and method calls, run by the fixture
generator (oracle/gen_golden_priors.py) on the REFERENCE's ngmix.priors /
ngmix.joint_prior and by tests/test_priors.py on ngmix_amd's, so that both
sides see the same seeds, the same arguments and the same order of calls on
every RandomState.  run(priors, joint_prior) returns {key: array}; a raised
exception is recorded by its class name.
"""
import numpy as np

GRID = np.array([-3.0, -1.0, -0.2, 0.0, 0.05, 0.3, 0.7, 0.999, 1.0, 1.5, 4.0, 30.0, 650.0])


def _rec(out, key, f):
    try:
        v = f()
    except Exception as e:      # noqa: BLE001
        out[key] = np.array("EXC:" + type(e).__name__)
        return
    if isinstance(v, tuple):
        for i, x in enumerate(v):
            out["%s/%d" % (key, i)] = np.asarray(x, dtype="f8")
    elif isinstance(v, list):
        out[key] = np.array(repr(v))
    elif v is None:
        out[key] = np.array("None")
    else:
        out[key] = np.asarray(v, dtype="f8")


def _scalar_methods(out, tag, p, names, values):
    for name in names:
        if not hasattr(p, name):
            out["%s/%s" % (tag, name)] = np.array("absent")
            continue
        for i, v in enumerate(values):
            _rec(out, "%s/%s/%d" % (tag, name, i), lambda: getattr(p, name)(float(v)))


def _array_methods(out, tag, p, names, values):
    for name in names:
        if not hasattr(p, name):
            out["%s/%s" % (tag, name)] = np.array("absent")
            continue
        _rec(out, "%s/%s" % (tag, name), lambda: getattr(p, name)(values.copy()))


def _samples(out, tag, p, rng, method="sample"):
    f = getattr(p, method)
    _rec(out, "%s/%s/scalar" % (tag, method), lambda: f())
    _rec(out, "%s/%s/5" % (tag, method), lambda: f(5))
    _rec(out, "%s/%s/scalar_again" % (tag, method), lambda: f(None))
    _rec(out, "%s/%s/200" % (tag, method), lambda: f(200))
    # where the generator stands afterwards: the number of deviates consumed
    out["%s/%s/rng_after" % (tag, method)] = np.asarray(rng.uniform(size=2))


ONE_D = ("get_prob_scalar", "get_lnprob_scalar", "get_fdiff")
ONE_D_ARRAYS = ("get_prob_array", "get_lnprob_array")


def one_d_priors(priors, seed):
    """(tag, prior, rng, inside-grid) for every 1-d prior"""
    def rs(k):
        return np.random.RandomState(seed + k)
    made = []
    r = rs(1)
    made.append(("flat", priors.FlatPrior(-0.5, 1.5, rng=r), r, GRID[(GRID >= -0.5) & (GRID <= 1.5)]))
    r = rs(2)
    made.append(("erf", priors.TwoSidedErf(-0.05, 0.03, 3.0, 0.3, rng=r), r, GRID))
    r = rs(3)
    made.append(("erf_flux", priors.TwoSidedErf(-1.0, 0.5, 500.0, 20.0, rng=r), r, GRID))
    r = rs(4)
    made.append(("normal", priors.Normal(0.5, 0.3, rng=r), r, GRID))
    r = rs(5)
    made.append(("normal_bounds", priors.Normal(70.0, 40.0, rng=r, bounds=(1.0, None)), r, GRID))
    r = rs(6)
    made.append(("lmbounds", priors.LMBounds(-1.0, 2.0, rng=r), r, GRID))
    r = rs(7)
    made.append(("lognormal", priors.LogNormal(0.7, 0.4, rng=r), r, GRID[GRID > 0]))
    r = rs(8)
    made.append(("lognormal_shift", priors.LogNormal(0.7, 0.4, rng=r, shift=-0.25), r,
                 GRID[GRID > -0.25]))
    r = rs(9)
    made.append(("sinh", priors.Sinh(0.2, 1.5, rng=r), r, GRID))
    r = rs(10)
    made.append(("truncgauss", priors.TruncatedGaussian(0.3, 0.5, -0.2, 1.0, rng=r), r,
                 GRID[(GRID > -0.2) & (GRID < 1.0)]))
    return made


def run(priors, joint_prior, guessers=None, seed=31415):
    out = {}
    if guessers is not None:
        run_guessers(out, guessers, priors, joint_prior, seed + 500)
    for tag, p, rng, inside in one_d_priors(priors, seed):
        out[tag + "/has_bounds"] = np.asarray(float(p.has_bounds()))
        out[tag + "/bounds"] = np.array(repr(p.bounds))
        _scalar_methods(out, tag, p, ONE_D, GRID)
        _array_methods(out, tag, p, ONE_D_ARRAYS, inside)
        _array_methods(out, tag + "/outside", p, ONE_D_ARRAYS, GRID)
        if tag in ("erf", "erf_flux"):
            _rec(out, tag + "/get_fdiff_array", lambda: p.get_fdiff(GRID.copy()))
        _samples(out, tag, p, rng)
        if tag.startswith("lognormal"):
            _samples(out, tag, p, rng, "sample_brute")
            for name in ("logmean", "logvar", "logsigma", "logivar", "mode", "log_mode",
                         "lnprob_max"):
                out["%s/%s" % (tag, name)] = np.asarray(getattr(p, name))
        if tag == "lmbounds":
            out[tag + "/mean_sigma"] = np.array([p.mean, p.sigma])
    _rec(out, "lognormal/bad_mean", lambda: priors.LogNormal(-1.0, 0.3, rng=np.random.RandomState(1)))
    _rec(out, "prior/no_rng", lambda: priors.FlatPrior(0.0, 1.0, rng=None))

    # Bounded1D over a normal
    r = np.random.RandomState(seed + 20)
    b = priors.Bounded1D(priors.Normal(0.5, 0.6, rng=r), (0.0, 1.0))
    out["bounded/bounds"] = np.array(repr(b.bounds))
    out["bounded/has_bounds"] = np.asarray(float(b.has_bounds()))
    _samples(out, "bounded", b, r)
    _rec(out, "bounded/size", lambda: b.sample(size=7))
    _rec(out, "bounded/bad1", lambda: priors.Bounded1D(None, (1.0,)))
    _rec(out, "bounded/bad2", lambda: priors.Bounded1D(None, 3.0))
    _rec(out, "bounded/bad3", lambda: priors.Bounded1D(None, (2.0, 1.0)))
    out["bounded/alias"] = np.asarray(float(priors.LimitPDF is priors.Bounded1D))

    # shapes
    gvals = [(0.0, 0.0), (0.1, -0.2), (0.6, 0.6), (0.8, 0.6), (0.9, 0.9), (-0.3, 0.05)]
    g1 = np.array([g[0] for g in gvals])
    g2 = np.array([g[1] for g in gvals])
    for tag, sigma, A in (("ba", 0.2, 1.0), ("ba_wide", 0.45, 2.5)):
        r = np.random.RandomState(seed + 30 + int(sigma * 100))
        p = priors.GPriorBA(sigma, rng=r, A=A)
        for i, (a, b_) in enumerate(gvals):
            for name in ("get_lnprob_scalar2d", "get_prob_scalar2d", "get_fdiff"):
                _rec(out, "%s/%s/%d" % (tag, name, i), lambda: getattr(p, name)(a, b_))
            _rec(out, "%s/get_prob_scalar1d/%d" % (tag, i), lambda: p.get_prob_scalar1d(abs(a)))
        _rec(out, tag + "/get_lnprob_array2d", lambda: p.get_lnprob_array2d(g1, g2))
        _rec(out, tag + "/get_prob_array2d", lambda: p.get_prob_array2d(g1, g2))
        _rec(out, tag + "/get_prob_array1d", lambda: p.get_prob_array1d(np.abs(g1)))
        _rec(out, tag + "/get_fdiff_array", lambda: p.get_fdiff(g1, g2))
        _rec(out, tag + "/sample1d", lambda: p.sample1d(50))
        out[tag + "/maxval1d"] = np.asarray(p.maxval1d)
        out[tag + "/maxval1d_loc"] = np.asarray(p.maxval1d_loc)
        _samples(out, tag, p, r, "sample2d")
        _rec(out, tag + "/sample2d_brute", lambda: p.sample2d_brute(40))
        out[tag + "/rng_end"] = np.asarray(r.uniform(size=2))
        for name in ("A", "sigma", "sig2", "sig4", "sig2inv", "sig4inv", "gmax"):
            out["%s/%s" % (tag, name)] = np.asarray(getattr(p, name))
    r = np.random.RandomState(seed + 40)
    p = priors.GPriorGauss(0.3, rng=r)
    _samples(out, "ggauss", p, r, "sample2d")
    _rec(out, "ggauss/sample1d", lambda: p.sample1d(3))
    r = np.random.RandomState(seed + 41)
    p = priors.ZDisk2D(0.8, rng=r)
    for i, (a, b_) in enumerate(gvals):
        for name in ("get_lnprob_scalar2d", "get_prob_scalar2d"):
            _rec(out, "zdisk/%s/%d" % (name, i), lambda: getattr(p, name)(a, b_))
        for name in ("get_lnprob_scalar1d", "get_prob_scalar1d"):
            _rec(out, "zdisk/%s/%d" % (name, i), lambda: getattr(p, name)(abs(a) * 1.2))
    _rec(out, "zdisk/get_prob_array2d", lambda: p.get_prob_array2d(g1, g2))
    _samples(out, "zdisk", p, r, "sample1d")
    _samples(out, "zdisk", p, r, "sample2d")
    r = np.random.RandomState(seed + 42)
    base = priors.GPriorBase([1.0, 2.0], rng=r)
    for name in ("get_lnprob_scalar2d", "get_prob_scalar2d"):
        _rec(out, "gbase/" + name, lambda: getattr(base, name)(0.1, 0.1))
    _rec(out, "gbase/get_prob_scalar1d", lambda: base.get_prob_scalar1d(0.1))
    _rec(out, "gbase/get_prob_array2d", lambda: base.get_prob_array2d(g1, g2))

    # centres
    r = np.random.RandomState(seed + 50)
    c = priors.CenPrior(0.1, -0.2, 0.05, 0.3, rng=r)
    for i, (a, b_) in enumerate(gvals):
        for name in ("get_fdiff", "get_lnprob_scalar", "get_lnprob_scalar_sep", "get_prob_scalar"):
            _rec(out, "cen/%s/%d" % (name, i), lambda: getattr(c, name)(a, b_))
    _rec(out, "cen/get_lnprob_array", lambda: c.get_lnprob_array(g1, g2))
    _samples(out, "cen", c, r)
    _samples(out, "cen", c, r, "sample2d")
    out["cen/alias"] = np.asarray(float(priors.SimpleGauss2D is priors.CenPrior))

    # random
    r = np.random.RandomState(seed + 60)
    _rec(out, "srandu/scalar", lambda: priors.srandu(rng=r))
    _rec(out, "srandu/4", lambda: priors.srandu(4, rng=r))
    out["make_rng/same"] = np.asarray(float(priors.make_rng(r) is r))
    out["make_rng/new"] = np.array(type(priors.make_rng()).__name__)

    # kde
    r = np.random.RandomState(seed + 61)
    data = np.random.RandomState(seed + 62).normal(size=(300, 2)) * [1.0, 0.2]
    k2 = priors.KDE(data, 0.3, rng=r)
    _samples(out, "kde2", k2, r)
    k1 = priors.KDE(data[:, 0], 0.2, rng=r)
    _samples(out, "kde1", k1, r)

    run_joint(out, priors, joint_prior, seed + 100)
    return out


def _pars_for(npars, rng, bad=None):
    pars = np.zeros(npars)
    pars[0:2] = rng.normal(scale=0.03, size=2)
    pars[2:4] = rng.normal(scale=0.15, size=2)
    pars[4:] = rng.uniform(0.2, 0.9, size=npars - 4)
    if bad == "g":
        pars[2:4] = (0.9, 0.9)
    elif bad == "last":
        pars[-1] = -50.0
    return pars


def _joint(out, tag, jp, npars, rngs, seed):
    out[tag + "/bounds"] = np.array(repr(jp.bounds))
    out[tag + "/nband"] = np.asarray(float(jp.nband))
    prng = np.random.RandomState(seed)
    many = []
    for i, bad in enumerate((None, None, None, "g", "last")):
        pars = _pars_for(npars, prng, bad)
        many.append(pars)
        fdiff = np.full(npars + 3, 7.0)

        def fill():
            n = jp.fill_fdiff(pars, fdiff)
            return np.concatenate([[n], fdiff])
        _rec(out, "%s/fill_fdiff/%d" % (tag, i), fill)
        _rec(out, "%s/get_lnprob_scalar/%d" % (tag, i), lambda: jp.get_lnprob_scalar(pars))
        _rec(out, "%s/get_prob_scalar/%d" % (tag, i), lambda: jp.get_prob_scalar(pars))
    good = np.array(many[:3])
    _rec(out, tag + "/get_lnprob_array", lambda: jp.get_lnprob_array(good.copy()))
    _rec(out, tag + "/get_prob_array", lambda: jp.get_prob_array(good.copy()))
    _rec(out, tag + "/sample/scalar", lambda: jp.sample())
    _rec(out, tag + "/sample/6", lambda: jp.sample(6))
    _rec(out, tag + "/get_widths", lambda: jp.get_widths(nrand=500))
    _rec(out, tag + "/get_widths_cached", lambda: jp.get_widths(nrand=5))
    out[tag + "/rng_after"] = np.array([r.uniform() for r in rngs])
    _rec(out, tag + "/short_pars", lambda: jp.get_lnprob_scalar(good[0][:-1]))


def run_joint(out, priors, joint_prior, seed):
    def terms(k, bounded=False, nflux=1):
        rs = [np.random.RandomState(seed + 10 * k + j) for j in range(6 + nflux)]
        cen = priors.CenPrior(0.0, 0.0, 0.05, 0.05, rng=rs[0])
        g = priors.GPriorBA(0.2, rng=rs[1])
        if bounded:
            T = priors.Normal(0.5, 0.3, rng=rs[2], bounds=(0.05, 2.0))
            F = [priors.Normal(0.6, 0.4, rng=rs[6 + j], bounds=(0.01, None)) for j in range(nflux)]
        else:
            T = priors.TwoSidedErf(-0.05, 0.03, 3.0, 0.3, rng=rs[2])
            F = [priors.TwoSidedErf(-1.0, 0.5, 500.0, 20.0, rng=rs[6 + j]) for j in range(nflux)]
        fracdev = priors.Normal(0.5, 0.1, rng=rs[3], bounds=(0.0, 1.0) if bounded else None)
        logTratio = priors.Normal(0.0, 0.5, rng=rs[4])
        flat = priors.FlatPrior(0.0, 1.0, rng=rs[5])
        return dict(cen=cen, g=g, T=T, F=F, fracdev=fracdev, logTratio=logTratio, flat=flat, rs=rs)

    t = terms(1)
    _joint(out, "simple1", joint_prior.PriorSimpleSep(t["cen"], t["g"], t["T"], t["F"][0]), 6,
           t["rs"], seed + 1)
    t = terms(2, nflux=3)
    _joint(out, "simple3", joint_prior.PriorSimpleSep(t["cen"], t["g"], t["T"], t["F"]), 8,
           t["rs"], seed + 2)
    t = terms(3, bounded=True, nflux=2)
    _joint(out, "simple2b", joint_prior.PriorSimpleSep(t["cen"], t["g"], t["T"], t["F"]), 7,
           t["rs"], seed + 3)
    t = terms(4)
    jp = joint_prior.PriorSimpleSep(t["cen"], t["g"], t["flat"], t["F"][0])
    _joint(out, "simple_flatT", jp, 6, t["rs"], seed + 4)
    t = terms(5, nflux=2)
    _rec(out, "simple_tuple/nband",
         lambda: joint_prior.PriorSimpleSep(t["cen"], t["g"], t["T"], tuple(t["F"])).nband)
    t = terms(6)
    _joint(out, "galsim1", joint_prior.PriorGalsimSimpleSep(t["cen"], t["g"], t["T"], t["F"][0]),
           6, t["rs"], seed + 6)
    t = terms(7, nflux=2)
    _rec(out, "bdf_tuple", lambda: joint_prior.PriorBDFSep(t["cen"], t["g"], t["T"], t["fracdev"],
                                                           tuple(t["F"])).nband)
    _joint(out, "bdf2", joint_prior.PriorBDFSep(t["cen"], t["g"], t["T"], t["fracdev"],
                                                t["F"]), 8, t["rs"], seed + 7)
    t = terms(8, bounded=True)
    _joint(out, "bdf1b", joint_prior.PriorBDFSep(t["cen"], t["g"], t["T"], t["fracdev"],
                                                 t["F"][0]), 7, t["rs"], seed + 8)
    t = terms(9, nflux=2)
    _joint(out, "bd2", joint_prior.PriorBDSep(t["cen"], t["g"], t["T"], t["logTratio"],
                                              t["fracdev"], t["F"]), 9, t["rs"], seed + 9)
    t = terms(10, bounded=True)
    _joint(out, "bd1b", joint_prior.PriorBDSep(t["cen"], t["g"], t["T"], t["logTratio"],
                                               t["fracdev"], t["F"][0]), 8, t["rs"], seed + 10)
    for ng, bounded in ((1, False), (3, False), (2, True)):
        t = terms(11 + ng + 5 * bounded, bounded=bounded)
        tag = "coellip%d%s" % (ng, "b" if bounded else "")
        jp = joint_prior.PriorCoellipSame(ng, t["cen"], t["g"], t["T"], t["F"][0])
        out[tag + "/npars"] = np.asarray(float(jp.npars))
        _joint(out, tag, jp, 4 + 2 * ng, t["rs"], seed + 20 + ng)
    t = terms(30, nflux=2)
    _rec(out, "coellip/two_bands",
         lambda: joint_prior.PriorCoellipSame(2, t["cen"], t["g"], t["T"], t["F"]))


def run_guessers(out, G, priors, joint_prior, seed):
    """the guessers that draw from (or are checked against) a joint prior and
    need no observation: successive guesses from seeded generators.  The T
    term of `tight` rejects many raw guesses, so the replace-by-a-prior-sample
    branch runs."""
    def joint(k, kind, nband, tight=False):
        rs = [np.random.RandomState(seed + 20 * k + j) for j in range(12)]
        cen = priors.CenPrior(0.0, 0.0, 0.05, 0.05, rng=rs[0])
        g = priors.GPriorBA(0.2, rng=rs[1])
        if tight:
            T = priors.FlatPrior(0.44, 0.50, rng=rs[2])
        else:
            T = priors.TwoSidedErf(-0.05, 0.03, 3.0, 0.3, rng=rs[2])
        F = [priors.TwoSidedErf(-1.0, 0.5, 500.0, 20.0, rng=rs[6 + j]) for j in range(nband)]
        Farg = F if nband > 1 else F[0]
        fracdev = priors.Normal(0.5, 0.1, rng=rs[3], bounds=(0.0, 1.0))
        logTratio = priors.Normal(0.0, 0.5, rng=rs[4])
        if kind == "simple":
            return joint_prior.PriorSimpleSep(cen, g, T, Farg)
        if kind == "bdf":
            return joint_prior.PriorBDFSep(cen, g, T, fracdev, Farg)
        return joint_prior.PriorBDSep(cen, g, T, logTratio, fracdev, Farg)

    def seq(tag, guesser, obs=None):
        _rec(out, "guess/%s/calls" % tag, lambda: np.array([guesser(obs=obs) for _ in range(4)]))
        _rec(out, "guess/%s/n5" % tag, lambda: guesser(nrand=5, obs=obs))

    k = 0
    for nband in (1, 3):
        flux = 120.0 if nband == 1 else [120.0, 80.0, 33.0]
        for tight in (False, True):
            tag = "%d%s" % (nband, "t" if tight else "")
            k += 1
            rng = np.random.RandomState(seed + 7 * k)
            seq("tflux_prior" + tag, G.TFluxGuesser(rng, 0.45, flux,
                                                    prior=joint(k, "simple", nband, tight)))
            k += 1
            rng = np.random.RandomState(seed + 7 * k)
            seq("tflux_and_prior" + tag, G.TFluxAndPriorGuesser(rng, 0.45, flux,
                                                                joint(k, "simple", nband, tight)))
            k += 1
            rng = np.random.RandomState(seed + 7 * k)
            seq("r50" + tag, G.R50FluxGuesser(rng, 0.46, flux,
                                              prior=joint(k, "simple", nband, tight)))
            k += 1
            seq("bdf" + tag, G.BDFGuesser(0.45, flux, joint(k, "bdf", nband, tight)))
            k += 1
            seq("bd" + tag, G.BDGuesser(0.45, flux, joint(k, "bd", nband, tight)))
            k += 1
            seq("prior" + tag, G.PriorGuesser(joint(k, "simple", nband, tight)))
            k += 1
            rng = np.random.RandomState(seed + 7 * k)
            pars = np.array([0.01, -0.02, 0.1, -0.05, 0.47] + [100.0] * nband)
            seq("pars" + tag, G.ParsGuesser(rng, pars, prior=joint(k, "simple", nband, tight)))
