from common import *
"""admom.get_result (the host half of adaptive moments: every derived key and flag branch) on random and hand-built
result structs, side by side; and make_mom_result-style outputs"""
import ngmix.admom.admom as radm
import ngmix_amd.admom as oadm
rng = np.random.RandomState(31)
rdt = np.dtype(radm._admom_result_dtype, align=True)
odt = np.dtype(ours._lib.ADMOM_RESULT_DTYPE) if hasattr(ours._lib, "ADMOM_RESULT_DTYPE") else None
print("dtype equal:", odt == rdt if odt is not None else "n/a", rdt.itemsize)


def struct(kind):
    a = np.zeros(1, dtype=rdt)
    r = a[0]
    r["npix"] = 900
    r["numiter"] = rng.randint(3, 30)
    r["wsum"] = rng.uniform(0.5, 50.0)
    s = rng.normal(size=7)
    s[5] = abs(s[5]) * 10 + 1.0          # flux sum
    s[4] = abs(s[4]) * 3 + 0.5           # T sum
    s[6] = abs(s[6]) * 5 + 1.0           # rho4 sum
    m = rng.normal(size=(7, 7))
    cov = m @ m.T * 0.01
    pars = np.array([rng.normal(scale=0.1), rng.normal(scale=0.1), rng.normal(scale=0.1),
                     rng.normal(scale=0.05), rng.uniform(0.3, 1.2), 1.0])
    flags = 0
    if kind == 1:
        s[5] = -1.0
    elif kind == 2:
        s[4] = -0.3
    elif kind == 3:
        cov[:] = np.nan
    elif kind == 4:
        cov[5, 5] = -1.0
    elif kind == 5:
        flags = 2 ** rng.randint(0, 8)
    elif kind == 6:
        s[5] = 0.0
    elif kind == 7:
        pars[4] = -0.2
    elif kind == 8:
        cov[4, 4] = 0.0
    elif kind == 9:
        s[2] = 40.0                       # |e| > 1
    r["sums"], r["sums_cov"], r["pars"], r["flags"] = s, cov, pars, flags
    return a


for trial in range(300):
    kind = trial % 10
    a = struct(kind)
    area, wnorm = rng.uniform(0.01, 0.2), rng.uniform(0.5, 20.0)
    run("get_result kind %d" % kind, lambda: radm.get_result(a.copy(), area, wnorm),
        lambda: oadm.get_result(a.copy(), area, wnorm))
    if ndiff[0] > 6:
        break
print("ndiff", ndiff[0])
