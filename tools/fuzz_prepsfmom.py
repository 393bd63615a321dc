"""A differential fuzz of ngmix_amd.prepsfmom: random catalogues (kernel, kernel size, stamp and psf sizes, padding,
apodisation, smoothing, sheared jacobians, centre offsets, with / without a psf, noise images) through the shipped
path -- half-plane transform by real matrix products + ngmix_prepsf_sums_batch -- and through (a) a full zero-padded
FFT of every stamp and (b) the (stamp, mode) stage as torch operations: the sums and their covariance must agree to
1e-9 of their scale.  usage: python tools/fuzz_prepsfmom.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngmix_amd.prepsfmom import PrePSFMom  # noqa: E402
from ngmix_amd.gexceptions import FFTRangeError  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
master = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 3)


def one_case(seed):
    rng = np.random.RandomState(seed)
    kernel = str(rng.choice(["pgauss", "ksigma", "gauss"]))
    n = int(rng.randint(1, 40))
    dim = int(rng.randint(15, 64))
    pdim = int(rng.choice([dim, int(rng.randint(13, 56))]))
    scale = rng.uniform(0.1, 0.4)
    deriv = (scale, 0.0, 0.0, scale) if rng.uniform() < 0.5 else (
        scale * rng.uniform(0.9, 1.1), scale * rng.uniform(-0.1, 0.1),
        scale * rng.uniform(-0.1, 0.1), scale * rng.uniform(0.9, 1.1))
    fwhm = scale * rng.uniform(4.0, 12.0)
    kw = dict(kernel=kernel, fwhm=float(fwhm), pad_factor=float(rng.choice([4, 3.5, 5, 4.25, 3])),
              ap_rad=float(rng.choice([1.5, 0.0, 1.0, 2.5])),
              fwhm_smooth=float(rng.choice([0.0, 0.0, scale * 3, scale * 5])),
              use_noise_image=bool(rng.uniform() < 0.3))
    no_psf = rng.uniform() < 0.2
    ax = np.arange(dim) - (dim - 1) / 2
    sig = rng.uniform(1.5, 3.5)
    images = np.exp(-0.5 * (ax[:, None] ** 2 + ax[None, :] ** 2) / sig ** 2)[None] * \
        rng.uniform(10, 100, size=(n, 1, 1)) + 0.05 * rng.normal(size=(n, dim, dim))
    pax = np.arange(pdim) - (pdim - 1) / 2
    pimages = np.exp(-0.5 * (pax[:, None] ** 2 + pax[None, :] ** 2) / rng.uniform(1.0, 2.0) ** 2)[None] \
        + 1e-4 * rng.normal(size=(n, pdim, pdim))
    pimages /= pimages.sum(axis=(1, 2), keepdims=True)
    weights = np.full((n, dim, dim), 400.0)
    weights[:, 2, 3] = 0.0
    cen = np.tile([(dim - 1) / 2] * 2, (n, 1)) + rng.uniform(-0.7, 0.7, size=(n, 2))
    pcen = np.tile([(pdim - 1) / 2] * 2, (n, 1)) + rng.uniform(-0.5, 0.5, size=(n, 2))
    noise = 0.05 * rng.normal(size=(n, dim, dim)) if kw["use_noise_image"] else None
    args = (images, weights, cen, deriv, None if no_psf else pimages, None if no_psf else pcen, noise)
    f = PrePSFMom(**kw)
    out = {}
    for label, env in (("shipped", None), ("fft", "NGMIX_PREPSF_FULL_FFT"), ("torch", "NGMIX_PREPSF_TORCH_SUMS")):
        if env:
            os.environ[env] = "1"
        try:
            out[label] = f.measure_arrays(*args)[:2]
        finally:
            if env:
                del os.environ[env]
    return out


ncase = nskip = 0
worst = 0.0
t0 = time.time()
fails = []
while time.time() - t0 < budget:
    seed = int(master.randint(1 << 30))
    try:
        out = one_case(seed)
    except FFTRangeError:
        nskip += 1
        continue
    ncase += 1
    for other in ("fft", "torch"):
        for a, b in zip(out["shipped"], out[other]):
            fin = np.isfinite(b)
            if not np.array_equal(np.isfinite(a), fin):
                fails.append((seed, other, "finite"))
                continue
            scale = np.abs(b[fin]).max()
            d = np.abs(a[fin] - b[fin]).max() / scale
            worst = max(worst, d)
            if d > 1e-9:
                fails.append((seed, other, d))
print("fuzz_prepsfmom: %.0f s, %d cases (%d skipped: kernel too large for the stamp), largest difference / scale "
      "%.2e, failures %d %s" % (time.time() - t0, ncase, nskip, worst, len(fails), fails[:5]))
