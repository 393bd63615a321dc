"""
Moment / size conversions and the summary statistics of a set of weighted
moment sums (reference: ngmix/moments.py).  Host scalar math only: the sums
themselves come from the weighted-sums kernel.
"""
import numpy as np

from . import flags as ngflags
from .gexceptions import GMixRangeError
from . import shape
from .util import get_ratio_error

FWHM_FAC = 2.3548200450309493  # 2 sqrt(2 ln 2)

# index of each named moment in the sums vector (gmix_nb.py:790-813)
MOMENTS_NAME_MAP = {
    "Mv": 0, "Mu": 1, "M1": 2, "M2": 3, "MT": 4, "MF": 5,
    "M00": 5, "M10": 1, "M01": 0, "M11": 4, "M20": 2, "M02": 3,
    "M21": 6, "M12": 7, "M30": 8, "M03": 9,
    "M22": 10, "M31": 11, "M13": 12, "M40": 13, "M14": 14,
    "M33": 15, "M44": 16,
}


def sigma_to_fwhm(sigma):
    return sigma * FWHM_FAC


def T_to_fwhm(T):
    return sigma_to_fwhm(np.sqrt(T / 2.0))


def fwhm_to_sigma(fwhm):
    return fwhm / FWHM_FAC


def fwhm_to_T(fwhm):
    return 2 * fwhm_to_sigma(fwhm) ** 2


def r50_to_sigma(r50):
    return fwhm_to_sigma(2.0 * r50)


def sigma_to_r50(sigma):
    return sigma_to_fwhm(sigma) / 2.0


def r50_to_T(r50):
    return 2 * r50_to_sigma(r50) ** 2


def T_to_r50(T):
    return sigma_to_r50(np.sqrt(T / 2.0))


def moms_to_e1e2(M1, M2, T):
    """e1 = M1 / T, e2 = M2 / T as M * (1 / T); a size T <= 0 (any, for an
    array) is a GMixRangeError (moments.py:135-163)"""
    if isinstance(T, np.ndarray):
        nbad = int(np.count_nonzero(T <= 0.0))
        if nbad > 0:
            raise GMixRangeError("%d T were <= 0.0" % nbad)
    elif T <= 0.0:
        raise GMixRangeError("T <= 0.0: %g" % T)
    Tinv = 1.0 / T
    return M1 * Tinv, M2 * Tinv


def get_Tround(T, g1, g2):
    gsq = g1 ** 2 + g2 ** 2
    return T * (1 - gsq) / (1 + gsq)


def get_T(Tround, g1, g2):
    gsq = g1 ** 2 + g2 ** 2
    return Tround * (1 + gsq) / (1 - gsq)


def mom2e(Irr, Irc, Icc):
    T = Irr + Icc
    return (Icc - Irr) / T, 2.0 * Irc / T, T


def mom2g(Irr, Irc, Icc):
    e1, e2, T = mom2e(Irr, Irc, Icc)
    g1, g2 = shape.e1e2_to_g1g2(e1, e2)
    return g1, g2, T


def e2mom(e1, e2, T):
    Irc = e2 * T / 2.0
    Icc = (1 + e1) * T / 2.0
    Irr = (1 - e1) * T / 2.0
    return Irr, Irc, Icc


def g2mom(g1, g2, T):
    e1, e2 = shape.g1g2_to_e1e2(g1, g2)
    return e2mom(e1, e2, T)


def get_sheared_g1g2T(g1, g2, T, s1, s2):
    g1s, g2s = shape.shear_reduced(g1, g2, s1, s2)
    Ts = get_T(get_Tround(T, g1, g2), g1s, g2s)
    return g1s, g2s, Ts


def get_sheared_M1M2T(M1, M2, T, s1, s2):
    e1, e2 = moms_to_e1e2(M1, M2, T)
    g1, g2 = shape.e1e2_to_g1g2(e1, e2)
    g1s, g2s, Ts = get_sheared_g1g2T(g1, g2, T, s1, s2)
    e1s, e2s = shape.g1g2_to_e1e2(g1s, g2s)
    return Ts * e1s, Ts * e2s, Ts


def get_sheared_moments(irr, irc, icc, s1, s2):
    g1, g2, T = mom2g(irr, irc, icc)
    g1s, g2s, Ts = get_sheared_g1g2T(g1, g2, T, s1, s2)
    return g2mom(g1s, g2s, Ts)


def make_mom_result(sums, sums_cov, sums_norm=None):
    """
    Summary statistics (flux, T, e, their errors and flags) of unnormalised
    weighted moment sums ordered [Mv, Mu, M1, M2, MT, MF, ...]
    (reference: ngmix/moments.py:398-539).
    """
    if len(sums) not in (6, 17):
        raise ValueError(
            "You must pass exactly 6 or 17 unnormalized moments in the order "
            "[Mv, Mu, M1, M2, MT, MF, ...] for ngmix.moments.make_mom_result.")
    if sums_cov.shape not in ((6, 6), (17, 17)):
        raise ValueError(
            "You must pass a 6x6 or 17x17 matrix for ngmix.moments.make_mom_result.")

    iv, iu, i1, i2, it, iflux = 0, 1, 2, 3, 4, 5
    nan2 = np.array([np.nan, np.nan])
    res = {
        "flags": 0, "flagstr": "",
        "flux": sums[iflux],
        "sums": sums, "sums_cov": sums_cov,
        "sums_norm": sums_norm if sums_norm is not None else np.nan,
        "flux_flags": 0, "flux_flagstr": "",
        "T_flags": 0, "T_flagstr": "",
        "flux_err": np.nan, "T": np.nan, "T_err": np.nan, "s2n": np.nan,
        "e1": np.nan, "e2": np.nan,
        "e": nan2.copy(), "e_err": nan2.copy(), "e_cov": np.diag(nan2),
        "sums_err": np.full(6, np.nan),
    }
    var_f = sums_cov[iflux, iflux]
    var_t = sums_cov[it, it]

    if var_f > 0:
        res["flux_err"] = np.sqrt(var_f)
        res["s2n"] = res["flux"] / res["flux_err"]
    else:
        res["flux_flags"] |= ngflags.NONPOS_VAR

    if var_f > 0 and var_t > 0:
        if sums[iflux] > 0:
            res["T"] = sums[it] / sums[iflux]
            res["T_err"] = get_ratio_error(sums[it], sums[iflux], var_t, var_f,
                                           sums_cov[it, iflux])
        else:
            res["T_flags"] |= ngflags.NONPOS_FLUX
    else:
        res["T_flags"] |= ngflags.NONPOS_VAR

    diag = np.diagonal(sums_cov)
    if np.all(diag > 0):
        res["sums_err"] = np.sqrt(diag)
    else:
        res["flags"] |= ngflags.NONPOS_VAR

    if res["flags"] == 0:
        if res["flux"] > 0:
            if res["T"] > 0:
                res["e1"] = sums[i1] / sums[it]
                res["e2"] = sums[i2] / sums[it]
                res["e"] = np.array([res["e1"], res["e2"]])
                res["pars"] = np.array([sums[iv], sums[iu], res["e1"], res["e2"],
                                        res["T"], res["flux"]])
                e_err = np.array([
                    get_ratio_error(sums[i1], sums[it], sums_cov[i1, i1], var_t,
                                    sums_cov[i1, it]),
                    get_ratio_error(sums[i2], sums[it], sums_cov[i2, i2], var_t,
                                    sums_cov[i2, it]),
                ])
                if np.all(np.isfinite(e_err)):
                    res["e_err"] = e_err
                    res["e_cov"] = np.diag(e_err ** 2)
                else:
                    res["flags"] |= ngflags.NONPOS_SHAPE_VAR
            else:
                res["flags"] |= ngflags.NONPOS_SIZE
        else:
            res["flags"] |= ngflags.NONPOS_FLUX

    res["flagstr"] = ngflags.get_flags_str(res["flags"])
    res["flux_flagstr"] = ngflags.get_flags_str(res["flux_flags"])
    res["T_flagstr"] = ngflags.get_flags_str(res["T_flags"])

    # moments by name, normalised by the flux sum (moments.py:542-575)
    fsum = sums[iflux]
    with np.errstate(invalid="ignore"):
        fsum_err = np.sqrt(var_f)
    for name, ind in MOMENTS_NAME_MAP.items():
        if ind > sums.size - 1:
            continue
        err_name = "%s_err" % name
        if name in ("MF", "M00"):
            res[name] = fsum
            res[err_name] = fsum_err
        elif fsum > 0:
            res[name] = sums[ind] / fsum
            res[err_name] = get_ratio_error(sums[ind], fsum, sums_cov[ind, ind],
                                            var_f, sums_cov[ind, iflux])
        else:
            res[name] = np.nan
            res[err_name] = np.nan
    return res


def regularize_mom_shapes(res, fwhm_reg):
    """
    Shapes from moment sums with the size regularised: e_{1,2} = M_{1,2} /
    (T + T_reg), T_reg the T of a gaussian of FWHM fwhm_reg -- for gaussians,
    the shape after smoothing with that round kernel.  res: a make_mom_result
    dict; the flux, T and their flags are kept, the shapes and their errors
    recomputed from the transformed sums (reference: ngmix/moments.py:578-640).
    fwhm_reg <= 0 returns res itself.
    """
    if not fwhm_reg > 0:
        return res
    sums = np.array(res["sums"], dtype="f8")
    cov = np.asarray(res["sums_cov"], dtype="f8")
    # MT -> MT + T_reg * MF: a linear map of the six sums
    amat = np.eye(6)
    amat[4, 5] = fwhm_to_T(fwhm_reg)
    # (pre-psf fitters leave the centroid sums NaN: they pass through untouched)
    nan_cen = np.isnan(sums[:2])
    sums[:2] = np.where(nan_cen, 0.0, sums[:2])
    reg = amat @ sums
    reg[:2] = np.where(nan_cen, np.nan, reg[:2])
    out = make_mom_result(reg, amat @ (cov @ amat.T))
    for key in ("T", "T_err", "T_flags", "T_flagstr"):
        out[key] = res[key]
    out["flags"] |= res["flags"]
    out["flagstr"] = ngflags.get_flags_str(out["flags"])
    return out


class _NumpyOps(object):
    """the array operations make_mom_result_batch is written in, for numpy"""
    nan = np.nan

    @staticmethod
    def f64(x):
        return np.asarray(x, dtype="f8")

    where = staticmethod(np.where)
    sqrt = staticmethod(np.sqrt)
    isfinite = staticmethod(np.isfinite)

    @staticmethod
    def clip_lo(x):
        return np.clip(x, 0.0, np.inf)

    @staticmethod
    def full(shape, val, like):
        return np.full(shape, val)

    @staticmethod
    def zeros_int(n, like):
        return np.zeros(n, dtype=np.int64)

    @staticmethod
    def stack(cols, axis):
        return np.stack(cols, axis=axis)

    @staticmethod
    def rows(x):
        """x (N, k) -> (k, N) contiguous: each column a contiguous vector"""
        return np.ascontiguousarray(x.T)

    @staticmethod
    def diagonal(x):
        return np.diagonal(x, axis1=1, axis2=2)

    @staticmethod
    def all0(x):
        return np.all(x, axis=0)

    @staticmethod
    def quiet():
        return np.errstate(divide="ignore", invalid="ignore")


class _TorchOps(object):
    """the same operations on torch tensors (the device of the inputs)"""
    nan = float("nan")

    def __init__(self):
        import contextlib
        import torch
        self.t = torch
        self.where = torch.where
        self.sqrt = torch.sqrt
        self.isfinite = torch.isfinite
        self._null = contextlib.nullcontext

    def f64(self, x):
        return x.to(self.t.float64)

    def clip_lo(self, x):
        # (numpy's clip passes nan through; so does clamp)
        return self.t.clamp(x, min=0.0)

    def full(self, shape, val, like):
        return self.t.full(shape if isinstance(shape, tuple) else (shape,), val,
                           dtype=self.t.float64, device=like.device)

    def zeros_int(self, n, like):
        return self.t.zeros(n, dtype=self.t.int64, device=like.device)

    def stack(self, cols, axis):
        return self.t.stack(cols, dim=axis)

    def rows(self, x):
        return x.T.contiguous()

    def diagonal(self, x):
        return self.t.diagonal(x, dim1=1, dim2=2)

    def all0(self, x):
        return x.all(dim=0)

    def quiet(self):
        return self._null()


def _ratio_error_arrays(a, b, var_a, var_b, cov_ab, where, B=_NumpyOps):
    """get_ratio_error over arrays, evaluated only where `where` (elsewhere nan);
    util.get_ratio_var's expression (equal to the scalar routine to the last
    bit or two: numpy squares arrays by multiplying and scalars through pow)"""
    # evaluated everywhere (masked gathers cost more than the arithmetic), kept
    # where asked: the elementwise values do not depend on their neighbours
    with B.quiet():
        ratio = a / b
        var = (ratio * ratio) * (var_a / (a * a) + var_b / (b * b) - 2 * cov_ab / (a * b))
        err = B.sqrt(B.clip_lo(var))
    return B.where(where, err, B.full(err.shape[0], B.nan, err))


def make_mom_result_batch(sums, sums_cov, sums_norm=None):
    """
    make_mom_result for N objects at once: sums (N, 6 | 17), sums_cov (N, nm, nm),
    sums_norm (N,).  Returns a dict of arrays with make_mom_result's keys --
    flags / flux_flags / T_flags (N,) ints, flux, flux_err, s2n, T, T_err, e1,
    e2 (N,), e, e_err (N, 2), e_cov (N, 2, 2), pars (N, 6; nan where the scalar
    routine sets none), sums_err (N, 6), the named moments M.. and M.._err --
    the values make_mom_result(sums[i], sums_cov[i], sums_norm[i]) gives (flags and
    ratios identical, the propagated errors to the last bit or two; the flag
    strings are left to the per-object view).

    numpy arrays in, numpy arrays out; torch tensors in (GaussMomBatch: the
    records still on the device), tensors on the same device out -- one code,
    IEEE operations only: flags and ratios agree to the bit, what holds a square
    root to an ulp.
    """
    if isinstance(sums, np.ndarray) or not hasattr(sums, "device"):
        B = _NumpyOps
        sums = np.asarray(sums, dtype="f8")
        sums_cov = np.asarray(sums_cov, dtype="f8")
        if sums_norm is not None:
            sums_norm = np.asarray(sums_norm, dtype="f8")
    else:
        B = _TorchOps()
    n, nm = sums.shape
    if nm not in (6, 17) or tuple(sums_cov.shape) != (n, nm, nm):
        raise ValueError("sums must be (N, 6 | 17) and sums_cov (N, nm, nm)")
    iv, iu, i1, i2, it, iflux = 0, 1, 2, 3, 4, 5
    where, sqrt = B.where, B.sqrt
    nan = B.full(n, B.nan, sums)
    one = B.full(n, 1.0, sums)
    # The columns the statistics read, each as a contiguous (N,) vector (the
    # inputs are usually strided views of a record array: arithmetic on 448-byte
    # strides costs ten times the arithmetic): sums by moment, and of the
    # covariance its diagonal and the columns of MF and MT.
    sT = B.rows(sums)
    diagT = B.rows(B.diagonal(sums_cov))
    cov_fT = B.rows(sums_cov[:, :, iflux])
    cov_tT = B.rows(sums_cov[:, :, it])
    flux = sT[iflux] + 0.0
    var_f = diagT[iflux]
    var_t = diagT[it]
    flags = B.zeros_int(n, sums)
    flux_flags = B.zeros_int(n, sums)
    T_flags = B.zeros_int(n, sums)

    fpos = var_f > 0
    with B.quiet():
        flux_err = where(fpos, sqrt(where(fpos, var_f, one)), nan)
        s2n = where(fpos, flux / flux_err, nan)
    flux_flags[~fpos] |= ngflags.NONPOS_VAR

    both = fpos & (var_t > 0)
    t_ok = both & (flux > 0)
    with B.quiet():
        T = where(t_ok, sT[it] / where(t_ok, flux, one), nan)
    T_err = _ratio_error_arrays(sT[it], flux, var_t, var_f, cov_fT[it], t_ok, B)
    T_flags[both & ~(flux > 0)] |= ngflags.NONPOS_FLUX
    T_flags[~both] |= ngflags.NONPOS_VAR

    dpos = B.all0(diagT > 0)
    # (N, nm): sqrt(diag) where every diagonal entry is positive, else nan
    with B.quiet():
        sq = sqrt(where(diagT > 0, diagT, B.full(tuple(diagT.shape), 1.0, sums))).T
    sums_err = where(dpos[:, None], sq, B.full(tuple(sq.shape), B.nan, sums))
    flags[~dpos] |= ngflags.NONPOS_VAR

    ok = flags == 0
    with B.quiet():
        shape_ok = ok & (flux > 0) & (T > 0)
    flags[ok & ~(flux > 0)] |= ngflags.NONPOS_FLUX
    with B.quiet():
        flags[ok & (flux > 0) & ~(T > 0)] |= ngflags.NONPOS_SIZE
    with B.quiet():
        mt = where(shape_ok, sT[it], one)
        e1 = where(shape_ok, sT[i1] / mt, nan)
        e2 = where(shape_ok, sT[i2] / mt, nan)
    pars = where(shape_ok[:, None], B.stack([sT[iv], sT[iu], e1, e2, T, flux], 1),
                 B.full((n, 6), B.nan, sums))
    e1_err = _ratio_error_arrays(sT[i1], sT[it], diagT[i1], var_t, cov_tT[i1], shape_ok, B)
    e2_err = _ratio_error_arrays(sT[i2], sT[it], diagT[i2], var_t, cov_tT[i2], shape_ok, B)
    err_ok = shape_ok & B.isfinite(e1_err) & B.isfinite(e2_err)
    flags[shape_ok & ~err_ok] |= ngflags.NONPOS_SHAPE_VAR
    e1e = where(err_ok, e1_err, nan)
    e2e = where(err_ok, e2_err, nan)
    e_err = B.stack([e1e, e2e], 1)
    zero = B.full(n, 0.0, sums)
    # (np.diag(nan2) of the scalar routine: nan on the diagonal, zeros off it)
    e_cov = B.stack([B.stack([e1e * e1e, zero], 1), B.stack([zero, e2e * e2e], 1)], 1)

    res = {
        "flags": flags, "flux_flags": flux_flags, "T_flags": T_flags,
        "flux": flux, "flux_err": flux_err, "s2n": s2n, "T": T, "T_err": T_err,
        "e1": e1, "e2": e2, "e": B.stack([e1, e2], 1), "e_err": e_err, "e_cov": e_cov,
        "pars": pars, "sums": sums, "sums_cov": sums_cov,
        "sums_norm": B.f64(sums_norm) if sums_norm is not None else nan + 0.0,
        "sums_err": sums_err,
    }
    with B.quiet():
        fsum_err = sqrt(var_f)
    fgood = flux > 0
    for name, ind in MOMENTS_NAME_MAP.items():
        if ind > nm - 1:
            continue
        if name in ("MF", "M00"):
            res[name], res[name + "_err"] = flux + 0.0, fsum_err
            continue
        with B.quiet():
            res[name] = where(fgood, sT[ind] / where(fgood, flux, one), nan)
        res[name + "_err"] = _ratio_error_arrays(sT[ind], flux, diagT[ind], var_f,
                                                 cov_fT[ind], fgood, B)
    return res
