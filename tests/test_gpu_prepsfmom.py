"""
ngmix_amd.prepsfmom against the reference's (tests/golden/prepsf.npz,
oracle/gen_golden_prepsf.py): the pre-psf Fourier-space moments of
ngmix-rendered stamps for both kernels, with and without a psf, smoothing, no
apodisation, a noise image, a sheared jacobian, even / odd stamps, a psf stamp
of another size, a non-integer padding factor; the k-space kernels themselves;
the error cases; and a catalogue as one batch (go_many) against the per-object
calls.  Tolerance 1e-10 of each quantity's scale: the two sides differ by the
rounding of two FFT libraries and of the device's sin / cos.
"""
import numpy as np
import pytest

import ngmix_amd as ngmix
from ngmix_amd import prepsfmom

pytestmark = pytest.mark.gpu

CASES = {
    "pgauss": (dict(kernel="pgauss", fwhm=1.2), {}),
    "ksigma": (dict(kernel="ksigma", fwhm=2.0), {}),
    "gauss_alias": (dict(kernel="gauss", fwhm=1.2), {}),
    "pgauss_smooth": (dict(kernel="pgauss", fwhm=1.2, fwhm_smooth=0.8), {}),
    "ksigma_smooth": (dict(kernel="ksigma", fwhm=2.0, fwhm_smooth=0.8), {}),
    "pgauss_noap": (dict(kernel="pgauss", fwhm=1.2, ap_rad=0), {}),
    "pgauss_pad": (dict(kernel="pgauss", fwhm=1.2, pad_factor=3.5), {}),
    "pgauss_shear": (dict(kernel="pgauss", fwhm=1.2), {}),
    "ksigma_shear": (dict(kernel="ksigma", fwhm=2.2), {}),
    "pgauss_nopsf": (dict(kernel="pgauss", fwhm=1.2), {"no_psf": True}),
    "pgauss_noiseim": (dict(kernel="pgauss", fwhm=1.2, use_noise_image=True), {}),
}
KEYS = ("flux", "flux_err", "T", "T_err", "s2n", "e1", "e2", "e", "e_err", "e_cov", "sums",
        "sums_cov", "sums_norm", "pars", "wsum")


def _jac(rec):
    j = rec[0] if rec.ndim else rec
    return ngmix.Jacobian(row=float(j["row0"]), col=float(j["col0"]), dvdrow=float(j["dvdrow"]),
                          dvdcol=float(j["dvdcol"]), dudrow=float(j["dudrow"]),
                          dudcol=float(j["dudcol"]))


def _obs(g, tag):
    psf = ngmix.Observation(g[tag + "_in_pim"], jacobian=_jac(g[tag + "_in_pjac"]))
    return ngmix.Observation(g[tag + "_in_im"], weight=g[tag + "_in_wt"],
                             jacobian=_jac(g[tag + "_in_jac"]), psf=psf, noise=g[tag + "_in_nim"])


def _close(got, want, name):
    want = np.asarray(want, dtype="f8")
    got = np.asarray(got, dtype="f8")
    assert got.shape == want.shape, name
    fin = np.isfinite(want)
    np.testing.assert_array_equal(np.isfinite(got), fin, err_msg=name)
    scale = np.abs(want[fin]).max() if fin.any() else 1.0
    np.testing.assert_allclose(got[fin], want[fin], rtol=1e-9, atol=1e-10 * scale, err_msg=name)


@pytest.mark.parametrize("tag", sorted(CASES))
def test_moments_vs_reference(golden, tag):
    g = golden("prepsf")
    kw, gokw = CASES[tag]
    res = prepsfmom.PrePSFMom(**kw).go(_obs(g, tag), **gokw)
    assert sorted(res.keys()) == list(g[tag + "_keys"])
    for k in ("flags", "flux_flags", "T_flags", "npix"):
        if tag + "_" + k in g:
            assert res[k] == int(g[tag + "_" + k]), k
    for k in KEYS:
        if tag + "_" + k in g:
            _close(res[k], g[tag + "_" + k], "%s %s" % (tag, k))


def test_kernels_vs_reference(golden):
    g = golden("prepsf")
    psf = ngmix.Observation(g["kern_in_pim"], jacobian=_jac(g["kern_in_pjac"]))
    obs = ngmix.Observation(g["kern_in_im"], weight=g["kern_in_wt"], jacobian=_jac(g["kern_in_jac"]),
                            psf=psf)
    for name, cls, fwhm in (("kern_pgauss", prepsfmom.PGaussMom, 1.2),
                            ("kern_ksigma", prepsfmom.KSigmaMom, 2.0)):
        res = cls(fwhm).go(obs, return_kernels=True)
        kern = res["kernels"]
        assert sorted(kern) == ["fk00", "fkc", "fkf", "fkp", "fkr", "nrm"]
        for k, v in kern.items():
            want = g["%s_%s" % (name, k)]
            assert np.shape(v) == want.shape and np.asarray(v).dtype == want.dtype, k
            scale = np.abs(want).max()
            np.testing.assert_allclose(v, want, rtol=1e-12, atol=1e-13 * scale, err_msg=k)
    assert prepsfmom.PrePSFGaussMom is prepsfmom.PGaussMom
    assert ngmix.ksigmamom.KSigmaMom is prepsfmom.KSigmaMom


def test_errors_are_the_references(golden):
    g = golden("prepsf")
    obs = _obs(g, "pgauss")
    im, wt, jac = g["pgauss_in_im"], g["pgauss_in_wt"], _jac(g["pgauss_in_jac"])
    pim = g["pgauss_in_pim"]
    obs_nonoise = ngmix.Observation(im, weight=wt, jacobian=jac, psf=obs.psf)
    calls = {
        "too_big": lambda: prepsfmom.PGaussMom(30.0).go(obs),
        "not_square": lambda: prepsfmom.PGaussMom(1.2).go(ngmix.Observation(np.zeros((10, 12))),
                                                         no_psf=True),
        "no_psf_set": lambda: prepsfmom.PGaussMom(1.2).go(
            ngmix.Observation(im, weight=wt, jacobian=jac)),
        "wcs_differs": lambda: prepsfmom.PGaussMom(1.2).go(ngmix.Observation(
            im, weight=wt, jacobian=jac,
            psf=ngmix.Observation(pim, jacobian=ngmix.DiagonalJacobian(row=16, col=16, scale=0.3)))),
        "bad_kernel": lambda: prepsfmom.PrePSFMom(1.2, "blah"),
        "not_obs": lambda: prepsfmom.PGaussMom(1.2).go(3),
        "noise_missing": lambda: prepsfmom.PGaussMom(1.2, use_noise_image=True).go(obs_nonoise),
    }
    want = dict(zip(g["error_names"], g["error_types"]))
    assert set(want) == set(calls)
    for name, f in calls.items():
        try:
            f()
            got = "None"
        except Exception as e:      # noqa: BLE001
            got = type(e).__name__
        assert got == str(want[name]), (name, got, want[name])


def test_go_many_is_go(golden):
    """a catalogue of stamps of two shapes in one call: every element is the
    per-object result"""
    g = golden("prepsf")
    rng = np.random.RandomState(4)
    obs = []
    for tag in ("pgauss", "pgauss_noap", "pgauss_smooth", "pgauss_shear", "ksigma"):
        o = _obs(g, tag)
        for _ in range(3):
            im = o.image + 0.01 * rng.normal(size=o.image.shape)
            obs.append(ngmix.Observation(im, weight=o.weight, jacobian=o.jacobian, psf=o.psf))
    fitter = prepsfmom.PGaussMom(1.2)
    many = fitter.go_many(obs)
    assert len(many) == len(obs)
    for o, r in zip(obs, many):
        one = fitter.go(o)
        assert r["flags"] == one["flags"] == 0
        for k in ("flux", "T", "e1", "e2", "s2n", "flux_err", "T_err"):
            np.testing.assert_allclose(r[k], one[k], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(r["sums_cov"], one["sums_cov"], rtol=1e-11, atol=1e-20)
    nopsf = fitter.go_many(obs[:4], no_psf=True)
    assert all(r["flags"] == 0 for r in nopsf) and nopsf[0]["T"] > many[0]["T"]
    # by key: the arrays over the catalogue
    assert many["T"].shape == (len(obs),) and many["e"].shape == (len(obs), 2)
    np.testing.assert_allclose(many["T"], [r["T"] for r in many], rtol=1e-12)
    np.testing.assert_array_equal(many["flags"], [r["flags"] for r in many])
    assert len(many[2:5]) == 3 and many[-1]["flux"] == many[len(obs) - 1]["flux"]
    with pytest.raises(IndexError):
        many[len(obs)]


@pytest.mark.parametrize("tag", ["pgauss", "ksigma_shear", "pgauss_nopsf", "pgauss_noiseim"])
def test_the_fused_kernel_is_the_torch_stage(golden, tag, monkeypatch):
    """ngmix_prepsf_sums_batch (deconvolution, phases, the fourteen sums in one
    pass) against the same stage as torch operations, and the transform at the
    kernel's modes against a full zero-padded FFT"""
    g = golden("prepsf")
    kw, gokw = CASES[tag]
    obs = _obs(g, tag)
    base = prepsfmom.PrePSFMom(**kw).go(obs, **gokw)
    for env in ("NGMIX_PREPSF_TORCH_SUMS", "NGMIX_PREPSF_FULL_FFT"):
        monkeypatch.setenv(env, "1")
        other = prepsfmom.PrePSFMom(**kw).go(obs, **gokw)
        monkeypatch.delenv(env)
        _close(other["sums"], base["sums"], env)
        _close(other["sums_cov"], base["sums_cov"], env)


def test_go_batch_arrays_are_the_per_object_dicts(golden):
    """go_batch (arrays in, arrays out: make_mom_result_batch over the sums)
    against go() stamp by stamp, including stamps whose moments fail (a flux
    sum driven negative)"""
    g = golden("prepsf")
    o = _obs(g, "pgauss")
    rng = np.random.RandomState(8)
    n = 12
    images = np.stack([o.image + 0.02 * rng.normal(size=o.image.shape) for _ in range(n)])
    images[3] = -images[3]                       # negative flux: flags set
    weights = np.stack([o.weight] * n)
    pimages = np.stack([o.psf.image] * n)
    j, pj = o.jacobian, o.psf.jacobian
    cen = np.tile([j.row0, j.col0], (n, 1))
    pcen = np.tile([pj.row0, pj.col0], (n, 1))
    fitter = prepsfmom.PGaussMom(1.2)
    res = fitter.go_batch(images, weights, cen, (j.dvdrow, j.dvdcol, j.dudrow, j.dudcol), pimages,
                          pcen)
    assert (res["flags"] != 0).sum() >= 1
    for i in range(n):
        one = fitter.go(ngmix.Observation(images[i], weight=weights[i], jacobian=j, psf=o.psf))
        for k in ("flags", "flux_flags", "T_flags"):
            assert res[k][i] == one[k], (k, i)
        for k in ("flux", "flux_err", "T", "T_err", "s2n", "e1", "e2", "e", "e_err", "e_cov",
                  "sums", "sums_cov"):
            a, b = np.asarray(res[k][i], dtype="f8"), np.asarray(one[k], dtype="f8")
            np.testing.assert_array_equal(np.isfinite(a), np.isfinite(b), err_msg=k)
            fin = np.isfinite(b)
            np.testing.assert_allclose(a[fin], b[fin], rtol=1e-10, atol=1e-14, err_msg=k)


@pytest.mark.parametrize("case", range(36))
def test_random_sweep_vs_reference(golden, case):
    """36 configurations drawn over kernels, kernel sizes, stamp sizes 21-52,
    psf stamps of other sizes, padding factors, apodisation widths, smoothing,
    sheared jacobians, centre offsets, with / without a psf
    (tests/golden/prepsf_sweep.npz, oracle/gen_golden_prepsf_sweep.py)"""
    g = golden("prepsf_sweep")
    tag = "c%02d" % case
    conf = g[tag + "_conf"]
    kw = dict(kernel=("pgauss", "ksigma")[int(conf[0])], fwhm=float(conf[1]),
              pad_factor=float(conf[2]), ap_rad=float(conf[3]), fwhm_smooth=float(conf[4]))
    im, pim = g[tag + "_im"].astype("f8"), g[tag + "_pim"].astype("f8")
    wt = np.full(im.shape, float(g[tag + "_wt0"]))
    obs = ngmix.Observation(im, weight=wt, jacobian=_jac(g[tag + "_jac"]),
                            psf=ngmix.Observation(pim, jacobian=_jac(g[tag + "_pjac"])))
    res = prepsfmom.PrePSFMom(**kw).go(obs, no_psf=bool(conf[5]))
    assert res["flags"] == int(g[tag + "_flags"])
    for k in ("flux", "flux_err", "T", "T_err", "s2n", "e1", "e2", "e_err", "sums", "sums_cov"):
        _close(res[k], g["%s_%s" % (tag, k)], "%s %s" % (tag, k))
