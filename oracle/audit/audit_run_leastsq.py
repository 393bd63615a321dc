from common import *
"""run_leastsq (the host wrapper of MINPACK: flags, covariance scaling, defaults on failure, bounds) on toy
problems, both packages calling the same scipy"""
import ngmix.fitting.leastsqbound as rl
import ngmix_amd.fitting as of
rng = np.random.RandomState(9)
t = np.linspace(0.0, 1.0, 60)


def problem(kind):
    truth = np.array([1.5, 0.8, 0.3, 2.0])
    data = truth[0] * np.exp(-truth[1] * t) + truth[2] * np.sin(truth[3] * t) + 0.01 * rng.normal(size=t.size)

    def f(p):
        if kind == "nan" and p[0] > 1.55:
            return np.full(t.size + npri, np.nan)
        r = np.zeros(t.size + npri)
        r[npri:] = (p[0] * np.exp(-p[1] * t) + p[2] * np.sin(p[3] * t) - data) / 0.01
        if npri:
            r[:npri] = (p[:npri] - truth[:npri]) / 0.5
        if kind == "singular":
            r[npri:] = (p[0] + p[1]) * np.ones(t.size) - data    # p0 and p1 degenerate
        return r

    def J(p):
        out = np.zeros((t.size + npri, 4))
        out[npri:, 0] = np.exp(-p[1] * t) / 0.01
        out[npri:, 1] = -p[0] * t * np.exp(-p[1] * t) / 0.01
        out[npri:, 2] = np.sin(p[3] * t) / 0.01
        out[npri:, 3] = p[2] * t * np.cos(p[3] * t) / 0.01
        for i in range(npri):
            out[i, i] = 1 / 0.5
        return out
    npri = 2 if kind == "prior" else 0
    return f, J, truth, npri


for kind in ("plain", "prior", "singular", "nan", "maxfev", "bounds", "bounds_one_sided", "dfun", "zero_dof"):
    for trial in range(6):
        f, J, truth, npri = problem(kind)
        guess = truth * rng.uniform(0.85, 1.15, size=4)
        kw = {"maxfev": 4000, "ftol": 1e-5, "xtol": 1e-5}
        if kind == "maxfev":
            kw["maxfev"] = 7
        if kind == "bounds":
            kw["bounds"] = [(0.5, 3.0), (0.1, 2.0), (None, None), (1.0, 2.1)]
        if kind == "bounds_one_sided":
            kw["bounds"] = [(1.0, None), (None, 1.0), (None, None), (None, None)]
        if kind == "dfun":
            kw["Dfun"] = J
        if kind == "zero_dof":
            tt = t
            f0 = f
            f = lambda p, f0=f0: f0(p)[:4]      # noqa: E731  (as many residuals as parameters)
        run("run_leastsq %s" % kind, lambda: rl.run_leastsq(f, guess.copy(), npri, **kw),
            lambda: of.run_leastsq(f, guess.copy(), npri, **kw))
    print(kind, "ndiff so far", ndiff[0])
print("ndiff", ndiff[0])
