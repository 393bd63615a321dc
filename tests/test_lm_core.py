"""
The re-entrant Levenberg-Marquardt iteration (ngmix_amd/csrc/lm_core.hpp)
against MINPACK itself: scipy.optimize.leastsq with Dfun is lmder, the routine
the reference's Fitter runs for gauss/exp/dev (ngmix/fitting/leastsqbound.py:
445-447, fitters.py:93-97).  The core is driven from the normal equations
(|f|^2, J^T f, J^T J) through the host entry points of the C ABI, so this runs
without a GPU; the device kernel runs the same code.
"""
import numpy as np
import pytest
from scipy.optimize import leastsq

from ngmix_amd import _lib

NP = _lib.LM_NPMAX


def fd_jacobian(func, x, f0):
    """MINPACK fdjac2 with scipy's default epsfcn (machine epsilon)"""
    eps = np.sqrt(np.finfo("f8").eps)
    J = np.zeros((f0.size, x.size))
    for j in range(x.size):
        h = eps * abs(x[j])
        if h == 0.0:
            h = eps
        xp = x.copy()
        xp[j] = x[j] + h
        J[:, j] = (func(xp) - f0) / h
    return J


def run_lm(func, jac, x0, ftol=1e-5, xtol=1e-5, gtol=0.0, maxfev=4000,
           factor=100.0, maxrounds=10000, mode=0, bounds=None):
    """drive one fit through ngmix_lm_init / ngmix_lm_advance_host;
    mode 1: forward differences (jac is ignored)"""
    L = _lib.lib()
    x0 = np.ascontiguousarray(x0, dtype="f8").reshape(1, -1)
    n = x0.shape[1]
    st = np.zeros(1, dtype=_lib.LM_STATE_DTYPE)
    lo = hi = None
    if bounds is not None:
        lo = np.array([-np.inf if b[0] is None else b[0] for b in bounds])
        hi = np.array([np.inf if b[1] is None else b[1] for b in bounds])
    assert L.ngmix_lm_init(_lib.ptr(st), 1, n, _lib.ptr(x0), ftol, xtol, gtol,
                           maxfev, factor, mode,
                           None if lo is None else _lib.ptr(lo),
                           None if hi is None else _lib.ptr(hi)) == 0
    rounds = 0
    while st["phase"][0] != _lib.LM_PHASE_DONE:
        xt = st["xt"][0, :n].copy()
        f = func(xt)
        ff = np.array([np.dot(f, f)])
        g = np.zeros((1, NP))
        A = np.zeros((1, NP, NP))
        if mode == 0 or st["phase"][0] != 1:   # not a plain trial of FD mode
            if mode == 0:
                J = jac(xt)
            else:
                # the state's own fdjac2 points (internal steps under bounds)
                J = np.zeros((f.size, n))
                for j in range(n):
                    xp = xt.copy()
                    xp[j] = st["xstep"][0, j]
                    J[:, j] = (func(xp) - f) / st["hstep"][0, j]
            g[0, :n] = J.T @ f
            A[0, :n, :n] = J.T @ J
        L.ngmix_lm_advance_host(_lib.ptr(st), 1, _lib.ptr(ff), _lib.ptr(g),
                                _lib.ptr(A))
        rounds += 1
        assert rounds < maxrounds
    return st[0]


def cov_from_state(st):
    """what scipy.optimize.leastsq derives from fjac / ipvt"""
    n = int(st["n"])
    R = st["R"][:n, :n]
    perm = np.eye(n)[st["ipvt"][:n]]
    Rp = np.triu(R) @ perm
    return np.linalg.inv(Rp.T @ Rp)


def problems():
    rng = np.random.RandomState(31)
    t = np.linspace(0.0, 4.0, 40)

    # 1. exponential decay + offset
    ytrue = 3.0 * np.exp(-1.3 * t) + 0.5
    y = ytrue + 0.01 * rng.normal(size=t.size)

    def f1(p):
        return p[0] * np.exp(-p[1] * t) + p[2] - y

    def j1(p):
        e = np.exp(-p[1] * t)
        return np.stack([e, -p[0] * t * e, np.ones_like(t)], axis=1)

    yield "expdecay", f1, j1, np.array([1.0, 0.5, 0.0])
    yield "expdecay_far", f1, j1, np.array([10.0, 3.0, -2.0])

    # 2. a gaussian profile with very different parameter scales
    xx = np.linspace(-5, 5, 101)
    yg = 250.0 * np.exp(-0.5 * (xx - 0.3) ** 2 / 1.7 ** 2) + 0.5 * rng.normal(size=xx.size)

    def f2(p):
        return p[0] * np.exp(-0.5 * (xx - p[1]) ** 2 / p[2] ** 2) - yg

    def j2(p):
        e = np.exp(-0.5 * (xx - p[1]) ** 2 / p[2] ** 2)
        return np.stack([e, p[0] * e * (xx - p[1]) / p[2] ** 2,
                         p[0] * e * (xx - p[1]) ** 2 / p[2] ** 3], axis=1)

    yield "gaussprof", f2, j2, np.array([100.0, -0.5, 1.0])

    # 3. Rosenbrock as least squares (a curved valley: many rejected steps)
    def f3(p):
        return np.array([10.0 * (p[1] - p[0] ** 2), 1.0 - p[0]])

    def j3(p):
        return np.array([[-20.0 * p[0], 10.0], [-1.0, 0.0]])

    yield "rosenbrock", f3, j3, np.array([-1.2, 1.0])

    # 4. six parameters: sum of two exponentials + line
    yy = (2.0 * np.exp(-0.7 * t) + 1.0 * np.exp(-3.0 * t) + 0.2 + 0.05 * t +
          0.002 * rng.normal(size=t.size))

    def f4(p):
        return (p[0] * np.exp(-p[1] * t) + p[2] * np.exp(-p[3] * t) + p[4] +
                p[5] * t - yy)

    def j4(p):
        e1, e2 = np.exp(-p[1] * t), np.exp(-p[3] * t)
        return np.stack([e1, -p[0] * t * e1, e2, -p[2] * t * e2, np.ones_like(t), t],
                        axis=1)

    yield "twoexp", f4, j4, np.array([1.5, 0.5, 1.5, 2.0, 0.0, 0.0])


@pytest.mark.parametrize("case", list(problems()), ids=lambda c: c[0])
@pytest.mark.parametrize("tol", [1e-5, 1.49012e-8])
def test_lm_core_follows_minpack(case, tol):
    name, func, jac, x0 = case
    xs, cov, info, mesg, ier = leastsq(func, x0, Dfun=jac, full_output=1, ftol=tol,
                                       xtol=tol, maxfev=4000)
    st = run_lm(func, jac, x0, ftol=tol, xtol=tol)
    n = x0.size
    x = st["x"][:n]
    # the same iteration path: same termination code and evaluation counts
    assert int(st["info"]) == ier, (name, st["info"], ier, mesg)
    assert int(st["nfev"]) == info["nfev"], name
    assert int(st["njev"]) == info["njev"], name
    scale = np.maximum(np.abs(xs), 1e-3)
    assert np.all(np.abs(x - xs) <= 1e-7 * scale), (name, x, xs)
    f = func(x)
    np.testing.assert_allclose(st["fnorm"], np.sqrt(np.dot(f, f)), rtol=1e-12)
    if cov is not None:
        np.testing.assert_allclose(cov_from_state(st), cov, rtol=1e-5, atol=0)


@pytest.mark.parametrize("case", list(problems()), ids=lambda c: c[0])
@pytest.mark.parametrize("tol", [1e-5, 1.49012e-8])
def test_lm_core_fd_mode_follows_lmdif(case, tol):
    """forward-difference mode against scipy leastsq without Dfun (lmdif)"""
    name, func, jac, x0 = case
    xs, cov, info, mesg, ier = leastsq(func, x0, full_output=1, ftol=tol, xtol=tol,
                                       maxfev=4000)
    st = run_lm(func, None, x0, ftol=tol, xtol=tol, mode=1)
    n = x0.size
    assert int(st["info"]) == ier, (name, st["info"], ier, mesg)
    assert int(st["nfev"]) == info["nfev"], name
    # forward-difference noise times the conditioning of the normal equations
    # (twoexp is nearly degenerate): iterates agree to 1e-5, counts exactly
    scale = np.maximum(np.abs(xs), 1e-3)
    assert np.all(np.abs(st["x"][:n] - xs) <= 1e-5 * scale), (name, st["x"][:n], xs)
    if cov is not None:
        np.testing.assert_allclose(cov_from_state(st), cov, rtol=1e-3, atol=0)


def test_lm_core_maxfev_and_batch():
    """maxfev -> info 5; several fits in one call advance independently"""
    cases = list(problems())
    name, func, jac, x0 = cases[3]  # rosenbrock needs > 5 evaluations
    st = run_lm(func, jac, x0, maxfev=5)
    assert int(st["info"]) == 5 and int(st["nfev"]) == 5
    _, _, _, _, ier = leastsq(func, x0, Dfun=jac, full_output=1, ftol=1e-5,
                              xtol=1e-5, maxfev=5)
    assert ier == 5

    # a batch of three-parameter fits from different starts
    L = _lib.lib()
    _, f1, j1, _ = cases[0]
    starts = np.array([[1.0, 0.5, 0.0], [2.0, 1.0, 0.3], [5.0, 2.0, 1.0]])
    st = np.zeros(3, dtype=_lib.LM_STATE_DTYPE)
    L.ngmix_lm_init(_lib.ptr(st), 3, 3, _lib.ptr(starts), 1e-5, 1e-5, 0.0, 4000, 100.0, 0,
                    None, None)
    running = 3
    while running:
        ff = np.zeros(3)
        g = np.zeros((3, NP))
        A = np.zeros((3, NP, NP))
        for i in range(3):
            xt = st["xt"][i, :3]
            f, J = f1(xt), j1(xt)
            ff[i] = f @ f
            g[i, :3] = J.T @ f
            A[i, :3, :3] = J.T @ J
        running = L.ngmix_lm_advance_host(_lib.ptr(st), 3, _lib.ptr(ff), _lib.ptr(g),
                                          _lib.ptr(A))
    for i in range(3):
        xs, ier = leastsq(f1, starts[i], Dfun=j1, ftol=1e-5, xtol=1e-5, maxfev=4000)
        assert int(st["info"][i]) == ier
        np.testing.assert_allclose(st["x"][i, :3], xs, rtol=1e-7)


def test_lm_core_out_of_range_trial():
    """ff = +inf at a trial point is a rejected step, as MINPACK treats the
    reference's -inf residual vector (results.py:463-464)"""
    cases = list(problems())
    _, func, jac, x0 = cases[0]

    def fwall(p):
        f = func(p)
        if p[1] > 1.2:  # forbid part of the path: the true b is 1.3
            return np.full_like(f, np.inf)
        return f

    L = _lib.lib()
    x0 = x0.reshape(1, -1)
    st = np.zeros(1, dtype=_lib.LM_STATE_DTYPE)
    L.ngmix_lm_init(_lib.ptr(st), 1, 3, _lib.ptr(x0), 1e-5, 1e-5, 0.0, 200, 100.0, 0,
                    None, None)
    while st["phase"][0] != _lib.LM_PHASE_DONE:
        xt = st["xt"][0, :3].copy()
        f, J = fwall(xt), jac(xt)
        ff = np.array([np.inf if not np.all(np.isfinite(f)) else f @ f])
        g = np.zeros((1, NP))
        A = np.zeros((1, NP, NP))
        if np.isfinite(ff[0]):
            g[0, :3] = J.T @ f
            A[0, :3, :3] = J.T @ J
        L.ngmix_lm_advance_host(_lib.ptr(st), 1, _lib.ptr(ff), _lib.ptr(g), _lib.ptr(A))
    assert st["x"][0, 1] <= 1.2
    assert np.isfinite(st["fnorm"][0])
    assert int(st["info"][0]) in (1, 2, 3, 5)


# ---- bounds: leastsqbound's transform (ngmix/fitting/leastsqbound.py:183-262,
# 440-552) = MINPACK on the unconstrained internal parameters
def _i2e(v, b):
    lo, hi = b
    if lo is None and hi is None:
        return v
    if hi is None:
        return lo - 1.0 + np.sqrt(v * v + 1.0)
    if lo is None:
        return hi + 1.0 - np.sqrt(v * v + 1.0)
    return lo + ((hi - lo) / 2.0) * (np.sin(v) + 1.0)


def _e2i(x, b):
    lo, hi = b
    if lo is None and hi is None:
        return x
    if hi is None:
        return np.sqrt((x - lo + 1.0) ** 2 - 1.0)
    if lo is None:
        return np.sqrt((hi - x + 1.0) ** 2 - 1.0)
    return np.arcsin((2.0 * (x - lo) / (hi - lo)) - 1.0)


def _grad(v, b):
    lo, hi = b
    if lo is None and hi is None:
        return 1.0
    if hi is None:
        return v / np.sqrt(v * v + 1.0)
    if lo is None:
        return -v / np.sqrt(v * v + 1.0)
    return (hi - lo) * np.cos(v) / 2.0


def leastsqbound_oracle(func, x0, bounds, jac=None, **kw):
    """the reference's leastsqbound restated on scipy.optimize.leastsq"""
    i2e = lambda xi: np.array([_i2e(v, b) for v, b in zip(xi, bounds)])
    i0 = np.array([_e2i(v, b) for v, b in zip(x0, bounds)])
    wfunc = lambda xi: func(i2e(xi))
    wjac = None
    if jac is not None:
        wjac = lambda xi: jac(i2e(xi)) * np.array(
            [_grad(v, b) for v, b in zip(xi, bounds)])
    xi, cov, info, mesg, ier = leastsq(wfunc, i0, Dfun=wjac, full_output=1, **kw)
    g = np.array([_grad(v, b) for v, b in zip(xi, bounds)])
    if cov is not None:
        cov = cov * g[:, None] * g[None, :]
    return i2e(xi), cov, info, ier


BOUND_CASES = [
    # expdecay with the rate boxed (the truth 1.3 inside; then outside: the fit
    # ends against the wall), the amplitude bounded below, the offset above
    ("inside", 0, [(0.0, None), (0.2, 3.0), (None, 2.0)]),
    ("wall", 0, [(None, None), (0.1, 1.0), (None, None)]),
    ("gauss_box", 2, [(10.0, 1000.0), (-2.0, 2.0), (0.2, None)]),
]


@pytest.mark.parametrize("case", BOUND_CASES, ids=lambda c: c[0])
@pytest.mark.parametrize("mode", [0, 1], ids=["lmder", "lmdif"])
def test_lm_core_bounds_follow_leastsqbound(case, mode):
    name, iprob, bounds = case
    _, func, jac, x0 = list(problems())[iprob]
    xs, cov, info, ier = leastsqbound_oracle(
        func, x0, bounds, jac=jac if mode == 0 else None, ftol=1e-8, xtol=1e-8,
        maxfev=4000)
    st = run_lm(func, jac, x0, ftol=1e-8, xtol=1e-8, mode=mode, bounds=bounds)
    n = x0.size
    assert int(st["bounded"]) == 1
    assert int(st["info"]) == ier, (name, st["info"], ier)
    assert int(st["nfev"]) == info["nfev"], name
    tol = 1e-7 if mode == 0 else 1e-5
    scale = np.maximum(np.abs(xs), 1e-3)
    assert np.all(np.abs(st["x"][:n] - xs) <= tol * scale), (name, st["x"][:n], xs)
    for (lo, hi), v in zip(bounds, st["x"][:n]):
        assert (lo is None or v >= lo) and (hi is None or v <= hi)
    if cov is not None and name != "wall":
        g = np.array([_grad(v, b) for v, b in zip(st["xi"][:n], bounds)])
        got = cov_from_state(st) * g[:, None] * g[None, :]
        np.testing.assert_allclose(got, cov, rtol=1e-4 if mode == 0 else 1e-3)


def test_simple_sep_prior_host_eval_matches_the_batch_prior():
    """ngmix_simple_sep_prior_eval (the code the prior kernel runs) against
    PriorSimpleSepBatch's torch arithmetic, in and out of range"""
    import torch
    from ngmix_amd import prior_batch as pb
    prior = pb.PriorSimpleSepBatch(
        pb.GaussianCen(0.01, -0.02, 0.1, 0.2), pb.GPriorBA(0.3),
        pb.Flat(-0.1, 3.0), [pb.TwoSidedErf(-1.0, 0.5, 100.0, 10.0),
                             pb.TwoSidedErf(0.0, 0.1, 50.0, 1.0)])
    desc = prior.descriptor()
    assert desc is not None and int(desc["nband"][0]) == 2
    rng = np.random.RandomState(2)
    pars = np.zeros((40, 7))
    pars[:, 0:2] = rng.normal(scale=0.1, size=(40, 2))
    pars[:, 2:4] = rng.uniform(-0.75, 0.75, size=(40, 2))   # some with g >= 1
    pars[:, 4] = rng.uniform(-0.3, 3.3, size=40)             # some outside the flat T
    pars[:, 5] = rng.uniform(-3.0, 120.0, size=40)
    pars[:, 6] = rng.uniform(-0.3, 52.0, size=40)            # some with p == 0
    rows, bad = prior.fill_fdiff_batch(torch.from_numpy(pars))
    lnp = prior.get_lnprob_batch(torch.from_numpy(pars)).numpy()
    rows, bad = rows.numpy(), bad.numpy()
    assert bad.any() and not bad.all()
    L = _lib.lib()
    for i in range(40):
        r = np.zeros(7)
        l = np.zeros(1)
        k = L.ngmix_simple_sep_prior_eval(_lib.ptr(desc), _lib.ptr(pars[i].copy()),
                                          _lib.ptr(r), _lib.ptr(l))
        if bad[i]:
            assert k == -1
            continue
        assert k == 6
        np.testing.assert_allclose(r[:6], rows[i], rtol=1e-12, atol=1e-14)
        if np.isfinite(lnp[i]):
            np.testing.assert_allclose(l[0], lnp[i], rtol=1e-12)
        else:
            assert l[0] == -np.inf


@pytest.mark.parametrize("tag", ["b1", "b2", "bb"])
def test_batch_prior_equals_the_reference_prior(golden, tag):
    """rows and ln p of the reference's own PriorSimpleSep (tests/golden/
    prior.npz, generated by oracle/gen_golden_prior.py) from the torch prior on
    the CPU and from the C code the prior kernel runs"""
    import torch
    from ngmix_amd import prior_batch as pb
    g = golden("prior")
    nband = 2 if tag == "b2" else 1
    cs, gs = float(g["cen_sigma"]), float(g["g_sigma"])
    if tag == "bb":
        Tp = pb.Normal(*g["T_normal"], bounds=tuple(g["T_bounds"]))
        Fp = [pb.Normal(*g["F_normal"], bounds=(float(g["F_lower_bound"]), None))]
    else:
        Tp = pb.TwoSidedErf(*g["T_erf"])
        Fp = [pb.TwoSidedErf(*g["F_erf"]) for _ in range(nband)]
    prior = pb.PriorSimpleSepBatch(pb.GaussianCen(0.0, 0.0, cs, cs), pb.GPriorBA(gs), Tp, Fp)
    pts = g[tag + "_prior_pts"]
    rows, bad = prior.fill_fdiff_batch(torch.from_numpy(pts))
    assert not bool(bad.any())
    np.testing.assert_allclose(rows.numpy(), g[tag + "_prior_rows"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(prior.get_lnprob_batch(torch.from_numpy(pts)).numpy(),
                               g[tag + "_prior_lnp"], rtol=1e-12)
    desc = prior.descriptor()
    L = _lib.lib()
    for i in range(pts.shape[0]):
        r = np.zeros(7)
        l = np.zeros(1)
        k = L.ngmix_simple_sep_prior_eval(_lib.ptr(desc), _lib.ptr(pts[i].copy()),
                                          _lib.ptr(r), _lib.ptr(l))
        assert k == 4 + nband
        np.testing.assert_allclose(r[:k], g[tag + "_prior_rows"][i], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(l[0], g[tag + "_prior_lnp"][i], rtol=1e-12)


@pytest.mark.parametrize("n", [6, 7, 8, 9, 10])
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("bounded", [False, True])
def test_register_form_equals_generic_form_bit_for_bit(n, mode, bounded, monkeypatch):
    """lm_core_reg.hpp (compile-time parameter count, arrays in registers: what
    the device runs for 6-10 parameters) against lm_core.hpp: the same state
    record, byte for byte, after every step of every fit -- including a
    rank-deficient jacobian and rejected steps"""
    L = _lib.lib()
    rng = np.random.RandomState(100 * n + 10 * mode + int(bounded))
    nfit, m = 12, 40
    t = np.linspace(0.0, 1.0, m)

    def model(p):
        # a sum of smooth bumps whose widths / heights are the parameters
        out = np.zeros(m)
        for k in range(n // 2):
            out += p[2 * k] * np.exp(-0.5 * (t - (k + 0.5) / (n // 2)) ** 2 /
                                     (0.05 + p[2 * k + 1] ** 2))
        if n % 2:
            out += p[n - 1] * t
        return out
    truth = rng.uniform(0.5, 1.5, size=(nfit, n))
    data = np.array([model(p) for p in truth]) + 0.01 * rng.normal(size=(nfit, m))
    x0 = truth * rng.uniform(0.7, 1.3, size=truth.shape)
    lo = hi = None
    if bounded:
        lo = np.full(n, 0.05)
        hi = np.full(n, np.inf)
        hi[0] = 3.0
        lo[1] = -np.inf
    states = {}
    for form in ("generic", "register"):
        if form == "generic":
            monkeypatch.setenv("NGMIX_LM_GENERIC", "1")
        else:
            monkeypatch.delenv("NGMIX_LM_GENERIC", raising=False)
        st = np.zeros(nfit, dtype=_lib.LM_STATE_DTYPE)
        assert L.ngmix_lm_init(_lib.ptr(st), nfit, n, _lib.ptr(x0), 1e-8, 1e-8, 0.0,
                               60, 100.0, mode,
                               None if lo is None else _lib.ptr(lo),
                               None if hi is None else _lib.ptr(hi)) == 0
        history = []
        for rounds in range(400):
            if np.all(st["phase"] == _lib.LM_PHASE_DONE):
                break
            ff = np.zeros(nfit)
            g = np.zeros((nfit, NP))
            A = np.zeros((nfit, NP, NP))
            for i in range(nfit):
                xt = st["xt"][i, :n]
                f = model(xt) - data[i]
                ff[i] = f @ f
                J = np.zeros((m, n))
                for j in range(n):
                    xp = xt.copy()
                    if mode == 1:
                        xp[j] = st["xstep"][i, j] if st["hstep"][i, j] != 0 else xt[j] + 1e-7
                        h = st["hstep"][i, j] if st["hstep"][i, j] != 0 else 1e-7
                    else:
                        h = 1e-7
                        xp[j] = xt[j] + h
                    J[:, j] = (model(xp) - f - data[i]) / h
                if i == 3:
                    J[:, n - 1] = J[:, 0]          # a rank-deficient jacobian
                if i == 5 and rounds == 2:
                    ff[i] = np.inf                 # an out-of-range trial
                g[i, :n] = J.T @ f
                A[i, :n, :n] = J.T @ J
                if mode == 2 and st["fonly"][i] and st["phase"][i] == 1:
                    # an |f|^2-only trial: the step must not read the jacobian
                    g[i] = np.nan
                    A[i] = np.nan
            L.ngmix_lm_advance_host(_lib.ptr(st), nfit, _lib.ptr(ff), _lib.ptr(g),
                                    _lib.ptr(A))
            history.append(st.copy().tobytes())
        assert np.all(st["phase"] == _lib.LM_PHASE_DONE)
        states[form] = history
    assert len(states["generic"]) == len(states["register"])
    for r, (a, b) in enumerate(zip(states["generic"], states["register"])):
        assert a == b, "state records differ after step %d" % r
    # the fits did something: several steps, several outcomes
    assert len(states["generic"]) > 4


def run_lm_lazy(func, jac, x0, mode, **kw):
    """run_lm in mode 0 (the jacobian with every evaluation) or 2 (left out of
    the trials predicted to end the fit): returns the final state and the
    number of jacobians / of |f|^2-only evaluations the driver was asked for"""
    L = _lib.lib()
    x0 = np.ascontiguousarray(x0, dtype="f8").reshape(1, -1)
    n = x0.shape[1]
    st = np.zeros(1, dtype=_lib.LM_STATE_DTYPE)
    assert L.ngmix_lm_init(_lib.ptr(st), 1, n, _lib.ptr(x0), kw.get("ftol", 1e-5),
                           kw.get("xtol", 1e-5), 0.0, 4000, 100.0, mode, None, None) == 0
    njac = nfonly = rounds = 0
    while st["phase"][0] != _lib.LM_PHASE_DONE:
        xt = st["xt"][0, :n].copy()
        f = func(xt)
        ff = np.array([np.dot(f, f)])
        g = np.full((1, NP), np.nan)
        A = np.full((1, NP, NP), np.nan)
        if st["fonly"][0] and st["phase"][0] == 1:
            nfonly += 1
        else:
            J = jac(xt)
            njac += 1
            g[0] = 0.0
            A[0] = 0.0
            g[0, :n] = J.T @ f
            A[0, :n, :n] = J.T @ J
        L.ngmix_lm_advance_host(_lib.ptr(st), 1, _lib.ptr(ff), _lib.ptr(g), _lib.ptr(A))
        rounds += 1
        assert rounds < 10000
    return st[0], njac, nfonly


@pytest.mark.parametrize("tol", [1e-5, 1e-8])
def test_lazy_jacobian_mode_is_lmder_to_the_bit(tol):
    """NGMIX_LM_MODE_ANALYTIC_LAZY: a trial predicted to end the fit is
    evaluated for |f|^2 alone -- lmder itself never forms the jacobian at its
    last point; a failed prediction asks for the jacobian one round later
    (phase JAC).  Either way the fit is the one mode ANALYTIC runs: the same
    x, nfev, njev, info, factor R -- every byte of the state but the phase
    bookkeeping -- with fewer jacobians asked of the evaluator."""
    saved = 0
    for name, func, jac, x0 in problems():
        eager, njac0, nf0 = run_lm_lazy(func, jac, x0, 0, ftol=tol, xtol=tol)
        lazy, njac2, nf2 = run_lm_lazy(func, jac, x0, 2, ftol=tol, xtol=tol)
        assert nf0 == 0
        for key in _lib.LM_STATE_DTYPE.names:
            if key in ("mode", "fonly"):
                continue
            assert np.array_equal(eager[key], lazy[key], equal_nan=True), (name, key)
        # what lmder reports is the same; the evaluator was asked for no more
        # jacobians than lmder itself forms (njev) where every prediction held
        assert lazy["nfev"] == eager["nfev"] and lazy["njev"] == eager["njev"]
        assert njac2 <= njac0
        assert njac2 >= lazy["njev"]
        saved += njac0 - njac2
    assert saved > 0
