"""
Pre-PSF moments in Fourier space (reference: ngmix/prepsfmom.py; the `pgauss`
and `ksigma` measurements of metadetection): the image is apodised at its
edge, zero-padded and transformed; divided by the transform of the psf image;
and summed against the Fourier transforms of a weight kernel and of its second
moments, all at the object's centre -- flux, size and the two shear-like
moments of the object as it was before the psf, with their covariance from
the noise power per mode.

MI355X form: N stamps of one shape are ONE batch on the device -- one batched
2-d FFT of the padded images, one of the psf images (rocFFT through torch.fft),
and the deconvolution, centre phases and kernel-weighted sums as a handful of
vector operations over the (stamp, mode) array restricted to the modes where
the kernel is not zero.  The k-space kernels depend only on the padded size,
the kernel width and the jacobian's derivatives: they are built once (host
float64, the fast exponential through the kernels' own device function) and
kept.  `go(obs)` is the reference's per-object call (a batch of one);
`go_many(list of obs)` runs a catalogue, grouped by shape.

    PGaussMom(fwhm).go(obs)          gaussian kernel of the given fwhm
    KSigmaMom(fwhm).go(obs)          Bernstein et al. (2016) k-sigma kernel, n = 4
    PrePSFMom(fwhm, kernel, pad_factor=4, ap_rad=1.5, fwhm_smooth=0,
              use_noise_image=False).go(obs, return_kernels=False, no_psf=False)

Results are moments.make_mom_result dicts (sums [nan, nan, M+, Mx, Mr, Mf] and
their covariance; flux, T, e, errors, flags), equal to the reference's to the
rounding of the transforms (tests/golden/prepsf.npz: 1e-10).
"""
import logging
import os

import numpy as np

from .observation import Observation
from .moments import fwhm_to_sigma, make_mom_result
from .gexceptions import FFTRangeError

__all__ = ["PrePSFMom", "KSigmaMom", "PGaussMom", "PrePSFGaussMom", "turn_on_fft_caching",
           "turn_off_fft_caching", "turn_on_kernel_caching", "turn_off_kernel_caching"]

logger = logging.getLogger(__name__)
FASTEXP_MAX_CHI2 = 25.0
_KERNELS = {}


# the reference can memoise its transforms and kernels; here the kernels are
# always kept (they are pure functions of their key) and a transform is never
# repeated within a batch, so the switches only clear what is held
def turn_on_fft_caching():
    pass


def turn_off_fft_caching():
    pass


def turn_on_kernel_caching():
    pass


def turn_off_kernel_caching():
    _KERNELS.clear()


def _torch():
    import torch
    return torch


# ---------------------------------------------------------------------------
# k-space kernels (host, float64)

def _modes(dim, deriv):
    """(fv, fu, |det| of the mode transform): angular frequencies of the dim x dim
    transform carried from pixel axes to the (v, u) plane by the inverse
    transpose of the jacobian's derivative matrix"""
    dvdrow, dvdcol, dudrow, dudcol = deriv
    f = np.fft.fftfreq(dim) * (2.0 * np.pi)
    fx, fy = f.reshape(1, -1), f.reshape(-1, 1)
    atinv = np.linalg.inv([[dvdrow, dvdcol], [dudrow, dudcol]]).T
    fv = atinv[0, 0] * fy + atinv[0, 1] * fx
    fu = atinv[1, 0] * fy + atinv[1, 1] * fx
    return fv, fu, np.abs(np.linalg.det(atinv))


def _fexp(x):
    from .fastexp_nb import fexp
    return fexp(x) if x.size else np.zeros(0)


def _smooth_profile(fwhm_smooth, fmag2):
    sigma = fwhm_to_sigma(fwhm_smooth)
    half_chi2 = sigma * sigma / 2 * fmag2
    out = np.zeros_like(fmag2)
    ok = (half_chi2 < FASTEXP_MAX_CHI2 / 2) & (half_chi2 >= 0)
    out[ok] = _fexp(-half_chi2[ok])
    return out


def _build_kernels(kind, dim, kernel_size, deriv, fwhm_smooth):
    """
    The weight kernel W(k^2) of unit peak in real space, times the k-space
    area element, on the modes where it is not zero, and its second moments:
    multiplying by u^2 in real space is -d^2/dku^2 in Fourier space, so with
    W' = dW/dk^2 and W'' = d^2 W / d(k^2)^2

        f   = c W                            flux
        r   = -4 c W' - 4 c k^2 W''          u^2 + v^2
        +   = -4 c (ku^2 - kv^2) W''         u^2 - v^2
        x   = -8 c ku kv W''                 2 u v

    gaussian: W = exp(-sigma^2 k^2 / 2) (cut where the fast exponential is);
    k-sigma: W = (1 - k^2 / kmax^2)^4 inside kmax^2 = 8 / sigma^2.
    """
    fv, fu, det = _modes(dim, deriv)
    sigma = fwhm_to_sigma(kernel_size)
    fmag2 = fu ** 2 + fv ** 2
    if kind == "pgauss":
        half = sigma * sigma / 2
        chi2_2 = half * fmag2
        msk = (chi2_2 < FASTEXP_MAX_CHI2 / 2) & (chi2_2 >= 0)
        fmag2, fu, fv = fmag2[msk], fu[msk], fv[msk]
        knrm = det * np.pi * 2 * sigma * sigma
        fkf = _fexp(-chi2_2[msk]) * knrm
        name = "gauss"
    else:
        n = 4
        kmax2 = 2 * n / sigma ** 2
        msk = fmag2 < kmax2
        fmag2, fu, fv = fmag2[msk], fu[msk], fv[msk]
        karg = 1.0 - fmag2 / kmax2
        karg2 = karg * karg
        karg3 = karg2 * karg
        knrm = det / (n / (sigma ** 2 * 10 * np.pi))
        fkf = karg2 * karg2 * knrm
        name = "ksigma"
    nrm = np.sum(fkf) / dim / dim
    if not np.allclose(nrm, 1.0, atol=1e-5, rtol=0):
        raise FFTRangeError("FFT size appears to be too small for %s kernel size %f: "
                            "norm = %f (should be 1)!" % (name, kernel_size, nrm))
    if kind == "pgauss":
        if fwhm_smooth > 0:
            fkf = fkf * _smooth_profile(fwhm_smooth, fmag2)
        s2, s4 = 2 * half, 4 * half ** 2
        fkr = (2 * s2 - s4 * fmag2) * fkf
        fkp = s4 * (fv ** 2 - fu ** 2) * fkf
        fkc = -2 * s4 * fu * fv * fkf
    else:
        two_d1 = (-knrm * 8.0 / kmax2) * karg3          # 2 c W'
        four_d2 = (knrm * 48 / kmax2 ** 2) * karg2       # 4 c W''
        if fwhm_smooth > 0:
            sm = _smooth_profile(fwhm_smooth, fmag2)
            fkf, two_d1, four_d2 = fkf * sm, two_d1 * sm, four_d2 * sm
        fkr = -2 * two_d1 - fmag2 * four_d2
        fkp = -(fu ** 2 - fv ** 2) * four_d2
        fkc = -2 * fu * fv * four_d2
    return dict(fkf=fkf, fkr=fkr, fkp=fkp, fkc=fkc, msk=msk, nrm=nrm, fk00=knrm)


def _kernels(kind, dim, kernel_size, deriv, fwhm_smooth):
    key = (kind, int(dim), float(kernel_size), tuple(float(d) for d in deriv), float(fwhm_smooth))
    if key not in _KERNELS:
        if len(_KERNELS) > 128:
            _KERNELS.clear()
        _KERNELS[key] = _build_kernels(kind, key[1], key[2], key[3], key[4])
    return _KERNELS[key]


# ---------------------------------------------------------------------------
# the pieces of the transform

def _apodization_edge(n, ap_rad):
    """the 1-d taper of one axis: a cumulative triweight kernel rising from 0
    to 1 over six ap_rad pixels at either end"""
    taper = np.ones(n)
    ap_range = int(6 * ap_rad + 0.5)
    for y in range(min(ap_range + 1, n)):
        t = (y - ap_range) / ap_rad + 3
        if t < -3:
            val = 0.0
        elif t > 3:
            val = 1.0
        else:
            val = -5 * t ** 7 / 69984 + 7 * t ** 5 / 2592 - 35 * t ** 3 / 864 + 35 * t / 96 + 1 / 2
        taper[y] *= val
        taper[n - 1 - y] *= val
    return taper


def _to_device(a, dev):
    """float64 on the device: numpy arrays are uploaded, tensors used where
    they lie"""
    torch = _torch()
    if isinstance(a, torch.Tensor):
        return a.to(device=dev, dtype=torch.float64)
    return torch.from_numpy(np.ascontiguousarray(a, dtype="f8")).to(dev)


def _pad_widths(dim, target_dim):
    extra = target_dim - dim
    return extra // 2, extra - extra // 2


def _padded_fft(images, target_dim, ap_rad, dev):
    """(N, D, D) complex transforms of the apodised, zero-padded stamps and the
    number of rows / columns of padding in front (the plain form: every mode)"""
    torch = _torch()
    n, dim, _ = images.shape
    d_im = _to_device(images, dev)
    if ap_rad > 0:
        edge = torch.from_numpy(_apodization_edge(dim, ap_rad)).to(dev)
        d_im = d_im * (edge[:, None] * edge[None, :])[None]
    before, _ = _pad_widths(dim, target_dim)
    padded = torch.zeros((n, target_dim, target_dim), dtype=torch.float64, device=dev)
    padded[:, before:before + dim, before:before + dim] = d_im
    return torch.fft.fft2(padded), before


def _mode_plan(kernels, dev):
    """
    Where the kernel lives in the D x D transform -- made once per kernel.

    The stamps are real, so the transform at -k is the conjugate of the one at
    k, and every term of the sums (real part of image / psf x phase, kernels
    even in k, |psf|^2, noise power) takes the same value at both: only the
    half plane of non-negative column frequencies is transformed and a mode
    with a partner in the other half counts twice.  (Not when the kernel
    reaches the Nyquist row or column, whose aliased frequencies break the
    symmetry of the kernels under a sheared jacobian: then every mode is kept.)

    Returns the rows and columns that hold kept modes, and per kept mode its
    place in the (rows x columns) block, its row and column there, its weight
    (1 or 2) and its position among the kernel's own arrays.
    """
    if "plan" not in kernels:
        msk = kernels["msk"]
        dim = msk.shape[0]
        rows, cols = np.nonzero(msk)
        nyq = dim // 2
        half = dim % 2 == 1 or not (msk[nyq, :].any() or msk[:, nyq].any())
        if half:
            keep = np.flatnonzero(cols <= nyq)
            wgt = np.where(cols[keep] == 0, 1.0, 2.0)
        else:
            keep = np.arange(rows.size)
            wgt = np.ones(rows.size)
        urows, irow = np.unique(rows[keep], return_inverse=True)
        ucols, icol = np.unique(cols[keep], return_inverse=True)
        kernels["plan"] = dict(urows=urows, ucols=ucols,
                               pick=(irow * ucols.size + icol).astype(np.int32),
                               irow=irow.astype(np.int32), icol=icol.astype(np.int32), wgt=wgt,
                               keep=keep, rows=rows[keep], cols=cols[keep])
    return kernels["plan"]


def _transform_at_modes(images, target_dim, ap_rad, kernels, dev):
    """
    The transform of the apodised, zero-padded stamps ON THE ROWS AND COLUMNS
    OF THE KERNEL'S MODES ONLY: real parts and imaginary parts, each (R, N C) --
    element (stamp n, row a, column b) at a N C + n C + b --, and the rows /
    columns of padding in front.

    A zero-padded stamp is dim x dim numbers in a D x D frame and the kernel
    keeps a disc of modes about k = 0.  Their values are the separable sums
    K[a, b] = sum_rc e^{-2 pi i a (r + pad) / D} im[r, c] e^{-2 pi i b (c + pad) / D}
    over the R rows a and C columns b the disc spans: small dense matrix
    products (on the fp64 matrix cores) instead of a D x D FFT of which all but
    a few per cent is thrown away -- nothing of size N D^2 is ever written, and
    the stamps being real, real products on the real and imaginary parts of
    the factors.
    """
    torch = _torch()
    n, dim, _ = images.shape
    plan = _mode_plan(kernels, dev)
    d_im = _to_device(images, dev)
    if ap_rad > 0:
        edge = torch.from_numpy(_apodization_edge(dim, ap_rad)).to(dev)
        d_im = d_im * (edge[:, None] * edge[None, :])[None]
    before, _ = _pad_widths(dim, target_dim)
    pos = np.arange(dim) + before

    def factors(modes):
        # cos and -sin of 2 pi a p / D with the product a p reduced modulo D
        # exactly (integers): as accurate at the far modes as at k = 0
        k = (np.outer(modes, pos) % target_dim) * (2.0 * np.pi / target_dim)
        return torch.from_numpy(np.cos(k)).to(dev), torch.from_numpy(-np.sin(k)).to(dev)
    er_re, er_im = factors(plan["urows"])                # (R, dim)
    ec_re, ec_im = factors(plan["ucols"])                # (C, dim)
    # along the columns, then along the rows; the stamps laid out (row, stamp,
    # column) so that each step is ONE product over all stamps: (dim N, dim) x
    # (dim, C), then (R, dim) x (dim, N C)
    by_row = d_im.permute(1, 0, 2).reshape(dim * n, dim)
    nc = ec_re.shape[0]
    t = torch.empty((2 * dim, n * nc), dtype=torch.float64, device=dev)   # [t_re ; t_im]
    torch.matmul(by_row, ec_re.T, out=t[:dim].view(dim * n, nc))
    torch.matmul(by_row, ec_im.T, out=t[dim:].view(dim * n, nc))
    # re = er_re t_re - er_im t_im, im = er_im t_re + er_re t_im: one product
    # each over the stacked halves (inner dimension 2 dim)
    re = torch.matmul(torch.cat([er_re, -er_im], dim=1), t)         # (R, N C)
    im = torch.matmul(torch.cat([er_im, er_re], dim=1), t)
    return re, im, before


def _same_wcs(a, b):
    return all(getattr(a, k) == getattr(b, k) for k in ("dvdrow", "dvdcol", "dudrow", "dudcol"))


def _check_obs_and_get_psf_obs(obs, no_psf):
    if not isinstance(obs, Observation):
        raise ValueError("input obs must be an Observation")
    shape = obs.image.shape
    if shape[0] != shape[1]:
        raise ValueError(f'pre-psf moments require a square image, got {shape}')
    if not obs.has_psf() and not no_psf:
        raise RuntimeError("The PSF must be set to measure a pre-PSF moment!")
    if no_psf:
        return None
    psf_obs = obs.get_psf()
    if not _same_wcs(psf_obs.jacobian, obs.jacobian):
        raise RuntimeError("The PSF and observation must have the same WCS Jacobian for "
                           "measuring pre-PSF moments.")
    return psf_obs


class PrePSFMom(object):
    """
    fwhm: size of the weight kernel (units of the jacobian)
    kernel: 'ksigma', or 'pgauss' / 'gauss'
    pad_factor: the stamps are zero-padded to int(pad_factor * the larger of
        the image and psf sizes)
    ap_rad: width of the edge taper of the image in pixels (0: none; psf stamps
        are never tapered)
    fwhm_smooth: an extra gaussian smoothing of the kernels
    use_noise_image: take the noise power per mode from obs.noise instead of
        the weight map
    """

    def __init__(self, fwhm, kernel, pad_factor=4, ap_rad=1.5, fwhm_smooth=0,
                 use_noise_image=False):
        self.fwhm = fwhm
        self.pad_factor = pad_factor
        self.kernel = kernel
        self.ap_rad = ap_rad
        self.fwhm_smooth = fwhm_smooth
        self.use_noise_image = use_noise_image
        if kernel == "ksigma":
            self.kind = "ksigma"
        elif kernel in ("gauss", "pgauss"):
            self.kind = "pgauss"
        else:
            raise ValueError("The kernel '%s' for PrePSFMom is not recognized!" % kernel)

    # ---- the reference's per-object call
    def go(self, obs, return_kernels=False, no_psf=False):
        psf_obs = _check_obs_and_get_psf_obs(obs, no_psf)
        res, kernels, fft_dim = self._measure([obs], [psf_obs])
        res = res[0]
        if res['flags'] != 0:
            logger.debug("pre-psf moments failed: %s" % res['flagstr'])
        if return_kernels:
            full = {}
            for k, v in kernels.items():
                if k in ("msk", "plan"):
                    continue
                if k == "nrm":
                    full[k] = v
                else:
                    full[k] = np.zeros((fft_dim, fft_dim), dtype=np.complex128)
                    full[k][kernels["msk"]] = v
            res["kernels"] = full
        return res

    # ---- a catalogue: one batch per (image size, psf size, jacobian derivatives)
    def go_many(self, obs_list, no_psf=False):
        """
        The measurements of go(obs) for every obs: stamps of one shape and
        pixel scale are measured as one batch on the device.  Returns a
        PrePSFManyResults: res[i] is go(obs_list[i])'s dict (made when asked
        for), res["T"], res["e"], res["flags"] ... the arrays over the
        catalogue (moments.make_mom_result_batch).
        """
        psfs = [_check_obs_and_get_psf_obs(o, no_psf) for o in obs_list]
        groups = {}
        for i, (o, p) in enumerate(zip(obs_list, psfs)):
            j = o.jacobian
            key = (o.image.shape[0], None if p is None else p.image.shape[0],
                   j.dvdrow, j.dvdcol, j.dudrow, j.dudcol)
            groups.setdefault(key, []).append(i)
        n = len(obs_list)
        sums, cov, norm = np.zeros((n, 6)), np.zeros((n, 6, 6)), np.zeros(n)
        for idx in groups.values():
            m, c, kernels, _ = self._measure_sums([obs_list[i] for i in idx], [psfs[i] for i in idx])
            sums[idx], cov[idx], norm[idx] = m, c, kernels["fk00"]
        return PrePSFManyResults(sums, cov, norm)

    def _measure(self, obs_list, psf_list):
        mom, cov, kernels, target_dim = self._measure_sums(obs_list, psf_list)
        res = [make_mom_result(mom[i], cov[i], sums_norm=kernels["fk00"])
               for i in range(len(obs_list))]
        return res, kernels, target_dim

    def _measure_sums(self, obs_list, psf_list):
        first, pfirst = obs_list[0], psf_list[0]
        jac = first.jacobian
        if self.use_noise_image and not all(o.has_noise() for o in obs_list):
            raise ValueError('obs.noise must be set when use_noise_image=True')
        mom, cov, kernels, target_dim = self.measure_arrays(
            np.stack([o.image for o in obs_list]),
            np.stack([o.weight for o in obs_list]),
            np.array([[o.jacobian.row0, o.jacobian.col0] for o in obs_list]),
            (jac.dvdrow, jac.dvdcol, jac.dudrow, jac.dudcol),
            None if pfirst is None else np.stack([p.image for p in psf_list]),
            None if pfirst is None else np.array([[p.jacobian.row0, p.jacobian.col0]
                                                  for p in psf_list]),
            np.stack([o.noise for o in obs_list]) if self.use_noise_image else None)
        return mom, cov, kernels, target_dim

    def go_batch(self, images, weights, cen, deriv, psf_images=None, psf_cen=None,
                 noise_images=None):
        """
        A catalogue as arrays in, arrays out: the arguments of measure_arrays;
        returns the dict of per-stamp arrays of moments.make_mom_result_batch
        (flags, flux, flux_err, T, T_err, e, e_err, e_cov, s2n, sums, sums_cov,
        ... every key of go()'s dict but the flag strings), one vectorised pass
        over the sums.
        """
        from .moments import make_mom_result_batch
        mom, cov, kernels, _ = self.measure_arrays(images, weights, cen, deriv, psf_images,
                                                   psf_cen, noise_images)
        return make_mom_result_batch(mom, cov, sums_norm=kernels["fk00"])

    def measure_arrays(self, images, weights, cen, deriv, psf_images=None, psf_cen=None,
                       noise_images=None):
        """
        The measurement on arrays: N stamps of one shape and one pixel scale.

        images, weights: (N, dim, dim) numpy arrays or torch tensors (device
        tensors are used where they lie); cen: (N, 2) the (row0, col0) of the
        jacobians; deriv: (dvdrow, dvdcol, dudrow, dudcol), the same for all;
        psf_images (N, pdim, pdim) and psf_cen (N, 2), or None for no psf (a
        pixel is deconvolved); noise_images (N, dim, dim) with use_noise_image.

        Returns (sums (N, 6) = [nan, nan, M+, Mx, Mr, Mf], their covariance
        (N, 6, 6), the kernels, the padded size): what make_mom_result takes.
        """
        torch = _torch()
        from .batch import _require_cuda
        dev = _require_cuda(None)
        n, dim = images.shape[0], images.shape[1]
        pdim = dim if psf_images is None else psf_images.shape[1]
        target_dim = int(max(dim, pdim) * self.pad_factor)
        eff_pad_factor = target_dim / dim
        kernels = _kernels(self.kind, target_dim, float(self.fwhm), deriv,
                           float(self.fwhm_smooth))
        plan = _mode_plan(kernels, dev)
        nmodes, nr, nc = plan["pick"].size, plan["urows"].size, plan["ucols"].size
        full = bool(os.environ.get("NGMIX_PREPSF_FULL_FFT"))      # A/B switch: every mode by FFT
        d_urows = torch.from_numpy(plan["urows"]).to(dev)
        d_ucols = torch.from_numpy(plan["ucols"]).to(dev)

        # where element (stamp n, row a, column b) of a transform lies
        stride_n, stride_r = (nr * nc, nc) if full else (nc, n * nc)

        def transform(stamps, ap_rad):
            if full:
                k, pad = _padded_fft(stamps, target_dim, ap_rad, dev)
                k = k[:, d_urows][:, :, d_ucols].reshape(n, -1)
                return k.real.contiguous(), k.imag.contiguous(), pad
            return _transform_at_modes(stamps, target_dim, ap_rad, kernels, dev)
        kim_re, kim_im, before = transform(images, self.ap_rad)
        im_row, im_col = cen[:, 0] + before, cen[:, 1] + before
        kpsf_re = kpsf_im = pix = None
        if psf_images is not None:
            kpsf_re, kpsf_im, pbefore = transform(psf_images, 0)
            psf_row, psf_col = psf_cen[:, 0] + pbefore, psf_cen[:, 1] + pbefore
            # the psf's flux: its transform at k = 0, the sum of its pixels
            max_amp = _to_device(psf_images, dev).reshape(n, -1).sum(dim=1).abs().contiguous()
        else:
            # a pixel in real space
            f = np.sinc(np.fft.fftfreq(target_dim))
            max_amp = torch.full((n,), float(abs(f[0] * f[0])), dtype=torch.float64, device=dev)
            pix = torch.from_numpy(f[plan["rows"]] * f[plan["cols"]]).to(dev).contiguous()
            psf_row = psf_col = np.zeros(n)

        # ---- the centres: exp(i k (centre of the image - centre of the psf)), separable: one
        # cosine / sine per (stamp, row of modes) and per (stamp, column of modes)
        drow, dcol = im_row - psf_row, im_col - psf_col
        py = px = None
        if np.any(drow != 0) or np.any(dcol != 0):
            f = np.fft.fftfreq(target_dim)
            ky = torch.from_numpy(f[plan["urows"]]).to(dev)[None, :] * \
                torch.from_numpy(2.0 * np.pi * drow).to(dev)[:, None]
            kx = torch.from_numpy(f[plan["ucols"]]).to(dev)[None, :] * \
                torch.from_numpy(2.0 * np.pi * dcol).to(dev)[:, None]
            py = torch.complex(torch.cos(ky), torch.sin(ky)).contiguous()
            px = torch.complex(torch.cos(kx), torch.sin(kx)).contiguous()
        d_irow = torch.from_numpy(plan["irow"]).to(dev)
        d_icol = torch.from_numpy(plan["icol"]).to(dev)
        d_wgt = torch.from_numpy(plan["wgt"]).to(dev)

        # ---- the noise power per mode
        kn_re = kn_im = pnoise_stamp = None
        if self.use_noise_image:
            if noise_images is None:
                raise ValueError('obs.noise must be set when use_noise_image=True')
            kn_re, kn_im, _ = transform(noise_images, 0)
        else:
            w = _to_device(weights, dev).reshape(n, -1)
            pos_w = w > 0
            tot_var = torch.where(pos_w, 1.0 / torch.where(pos_w, w, torch.ones_like(w)),
                                  torch.zeros_like(w)).sum(dim=1)
            pnoise_stamp = (tot_var * eff_pad_factor ** 2).contiguous()

        # ---- deconvolution, phases and the fourteen sums: one pass over (stamp, mode)
        df2 = (1 / target_dim) ** 2
        df4 = df2 * df2
        fk = torch.from_numpy(np.stack([kernels[k][plan["keep"]]
                                        for k in ("fkp", "fkc", "fkr", "fkf")])).to(dev)
        if os.environ.get("NGMIX_PREPSF_TORCH_SUMS"):        # A/B switch: the same stage in torch ops
            def modes(t):      # (N, M) from wherever the transform lies
                if t is None:
                    return None
                at = (torch.arange(n, device=dev) * stride_n)[:, None] + \
                    (d_irow.long() * stride_r + d_icol.long())[None, :]
                return t.reshape(-1)[at]
            sums = _sums_torch(modes(kim_re), modes(kim_im), modes(kpsf_re), modes(kpsf_im), pix,
                               modes(kn_re), modes(kn_im), pnoise_stamp, eff_pad_factor ** 2,
                               max_amp, py, px, d_irow, d_icol, fk, d_wgt, df2, df4)
        else:
            from . import _lib
            from .batch import _dptr, _stream
            sums = torch.empty((n, 14), dtype=torch.float64, device=dev)

            def ptr(t):
                return None if t is None else _dptr(t)
            with torch.cuda.device(dev):
                st = _lib.lib().ngmix_prepsf_sums_batch(
                    ptr(kim_re), ptr(kim_im), ptr(kpsf_re), ptr(kpsf_im), ptr(pix), ptr(kn_re),
                    ptr(kn_im), ptr(pnoise_stamp), float(eff_pad_factor ** 2), ptr(max_amp),
                    None if py is None else _dptr(torch.view_as_real(py)),
                    None if px is None else _dptr(torch.view_as_real(px)),
                    ptr(d_irow), ptr(d_icol), ptr(fk), ptr(d_wgt), n, int(nmodes),
                    int(stride_n), int(stride_r), int(nr), int(nc), df2, df4, ptr(sums), _stream())
            _lib.check(st, "ngmix_prepsf_sums_batch")
        sums = sums.cpu().numpy()
        mom = np.full((n, 6), np.nan)
        mom[:, 2:6] = sums[:, :4]
        cov = np.zeros((n, 6, 6))
        cov[:, 0, 0] = cov[:, 1, 1] = 1.0        # (as the reference leaves them)
        t = 4
        for a in range(4):
            for b in range(a, 4):
                cov[:, 2 + a, 2 + b] = cov[:, 2 + b, 2 + a] = sums[:, t]
                t += 1
        return mom, cov, kernels, target_dim


def _sums_torch(kim_re, kim_im, kpsf_re, kpsf_im, pix, kn_re, kn_im, pnoise_stamp, noise_scale,
                max_amp, py, px, d_irow, d_icol, fk, wgt, df2, df4):
    """the (stamp, mode) stage written as torch operations: what
    ngmix_prepsf_sums_batch does in one pass (kept as the check of the kernel)"""
    torch = _torch()
    n = kim_re.shape[0]
    kim = torch.complex(kim_re, kim_im)
    if kpsf_re is None:
        kpsf = pix.to(torch.complex128)[None].repeat(n, 1)
    else:
        kpsf = torch.complex(kpsf_re, kpsf_im)
    min_amp = (1e-5 * max_amp)[:, None]
    amp = kpsf.abs()
    low = amp <= min_amp
    safe = torch.where(amp == 0, torch.ones_like(amp), amp)
    kpsf = torch.where(low & (amp != 0), kpsf / safe * min_amp, kpsf)
    kpsf = torch.where(low & (amp == 0), min_amp.to(torch.complex128).expand_as(kpsf), kpsf)
    kim = kim / kpsf
    if py is not None:
        kim = kim * (px[:, d_icol.long()] * py[:, d_irow.long()])
    if kn_re is not None:
        pnoise = (kn_re ** 2 + kn_im ** 2) * noise_scale
    else:
        pnoise = pnoise_stamp[:, None]
    w = wgt[None, :] * pnoise / (kpsf.real ** 2 + kpsf.imag ** 2)
    cols = [(kim.real * (fk[c] * wgt)[None, :]).sum(dim=1) * df2 for c in range(4)]
    for a in range(4):
        for b in range(a, 4):
            cols.append((fk[a] * fk[b] * w).sum(dim=1) * df4)
    return torch.stack(cols, dim=1)


class PrePSFManyResults(object):
    """the measurements of a catalogue: by position the per-object result dict
    of go() (moments.make_mom_result on that stamp's sums, made when asked for),
    by key the arrays over the catalogue (moments.make_mom_result_batch, made
    once when first asked for); len() and iteration are over the stamps"""

    def __init__(self, sums, sums_cov, sums_norm):
        self.sums, self.sums_cov, self.sums_norm = sums, sums_cov, sums_norm
        self._arrays = None

    def __len__(self):
        return self.sums.shape[0]

    def arrays(self):
        if self._arrays is None:
            from .moments import make_mom_result_batch
            self._arrays = make_mom_result_batch(self.sums, self.sums_cov,
                                                 sums_norm=self.sums_norm)
        return self._arrays

    def __getitem__(self, key):
        if isinstance(key, str):
            return self.arrays()[key]
        if isinstance(key, slice):
            return [self[i] for i in range(*key.indices(len(self)))]
        i = int(key)
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(key)
        return make_mom_result(self.sums[i], self.sums_cov[i], sums_norm=self.sums_norm[i])

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def keys(self):
        return self.arrays().keys()


class KSigmaMom(PrePSFMom):
    """the k-sigma kernel of Bernstein et al. (2016), n = 4"""

    def __init__(self, fwhm, pad_factor=4, ap_rad=1.5, fwhm_smooth=0, use_noise_image=False):
        super().__init__(fwhm, 'ksigma', pad_factor=pad_factor, ap_rad=ap_rad,
                         fwhm_smooth=fwhm_smooth, use_noise_image=use_noise_image)


class PGaussMom(PrePSFMom):
    """a gaussian kernel: the pre-psf gaussian moments of metadetection"""

    def __init__(self, fwhm, pad_factor=4, ap_rad=1.5, fwhm_smooth=0, use_noise_image=False):
        super().__init__(fwhm, 'pgauss', pad_factor=pad_factor, ap_rad=ap_rad,
                         fwhm_smooth=fwhm_smooth, use_noise_image=use_noise_image)


PrePSFGaussMom = PGaussMom
