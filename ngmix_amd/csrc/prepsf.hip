// Pre-psf Fourier moments (reference: ngmix/prepsfmom.py:337-422, 584-603): the
// (stamp, mode) stage as ONE pass.  Inputs are the transforms of the image and
// of the psf at the M modes the weight kernel keeps; per stamp the kernel
// deconvolves (psf amplitudes below 1e-5 of the psf's flux are held there),
// applies the phase that moves the object's centre to the origin, and
// accumulates the four kernel-weighted sums and the ten entries of their
// covariance (noise power per mode / |psf|^2).  One block per stamp; every
// (stamp, mode) pair is read once: 32 bytes of transforms + the shared kernels.
#include "common.hpp"
#include "device_utils.hpp"
#include "launch.hpp"

namespace ngmix {

struct cplx { double x, y; };

__device__ __forceinline__ cplx cmul(cplx a, cplx b)
{
    return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}

// numpy's complex division (Smith's algorithm, npymath)
__device__ __forceinline__ cplx cdiv(cplx a, cplx b)
{
    const double abs_br = fabs(b.x), abs_bi = fabs(b.y);
    if (abs_br >= abs_bi) {
        if (abs_br == 0 && abs_bi == 0) return {a.x / abs_br, a.y / abs_bi};
        const double rat = b.y / b.x, scl = 1.0 / (b.x + b.y * rat);
        return {(a.x + a.y * rat) * scl, (a.y - a.x * rat) * scl};
    }
    const double rat = b.x / b.y, scl = 1.0 / (b.y + b.x * rat);
    return {(a.x * rat + a.y) * scl, (a.y * rat - a.x) * scl};
}

// The transforms arrive as they leave the matrix products: real and imaginary
// parts in two arrays over (stamp, row of modes, column of modes), element
// (n, a, b) at n * stride_n + a * stride_r + b -- (R C, C) for stamp-major
// blocks, (C, N C) for the row-major output of one product over all stamps;
// irow / icol (M,) are the row and column of each kept mode.  kim_*:
// the image, kpsf_* the psf (NULL: a pixel in real space is deconvolved, pix
// (M,) real), knoise_* a noise image (NULL: pnoise_stamp (N,) per stamp);
// max_amp (N,).  py (N, R), px (N, C) complex or NULL (no centre offset
// anywhere); irow, icol (M,).  fk: (4, M) = fkp, fkc, fkr, fkf; wgt (M,): 1, or
// 2 for a mode that stands for its conjugate partner as well (real stamps:
// only half of the plane is transformed).  out: (N, 14) = M+, Mx, Mr, Mf, then
// the upper triangle of their covariance in that order (pp, pc, pr, pf, cc,
// cr, cf, rr, rf, ff).
__global__ __launch_bounds__(BLOCK) void prepsf_sums_kernel(
    const double *__restrict__ kim_re, const double *__restrict__ kim_im,
    const double *__restrict__ kpsf_re, const double *__restrict__ kpsf_im,
    const double *__restrict__ pix, const double *__restrict__ knoise_re,
    const double *__restrict__ knoise_im, const double *__restrict__ pnoise_stamp,
    double noise_scale, const double *__restrict__ max_amp, const cplx *__restrict__ py,
    const cplx *__restrict__ px, const int32_t *__restrict__ irow,
    const int32_t *__restrict__ icol, const double *__restrict__ fk,
    const double *__restrict__ wgt, int M, int64_t stride_n, int64_t stride_r, int R, int C,
    double df2, double df4, double *__restrict__ out)
{
    __shared__ double red[NWAVES * 14];
    const int64_t n = blockIdx.x;
    const int64_t base = n * stride_n;
    const double min_amp = 1.0e-5 * max_amp[n];
    const double pn = pnoise_stamp ? pnoise_stamp[n] : 0.0;
    double acc[14];
#pragma unroll
    for (int i = 0; i < 14; i++) acc[i] = 0.0;
    for (int m = threadIdx.x; m < M; m += BLOCK) {
        const int64_t at = base + irow[m] * stride_r + icol[m];
        cplx a = kpsf_re ? cplx{kpsf_re[at], kpsf_im[at]} : cplx{pix[m], 0.0};
        const double amp = hypot(a.x, a.y);
        if (amp <= min_amp) {
            if (amp != 0.0) {
                a = {a.x / amp * min_amp, a.y / amp * min_amp};
            } else {
                a = {min_amp, 0.0};
            }
        }
        cplx k = cdiv(cplx{kim_re[at], kim_im[at]}, a);
        if (py) k = cmul(k, cmul(px[n * C + icol[m]], py[n * R + irow[m]]));
        double noise = pn;
        if (knoise_re) {
            const double nr = knoise_re[at], ni = knoise_im[at];
            noise = (nr * nr + ni * ni) * noise_scale;
        }
        const double g = wgt[m];
        const double w = g * noise / (a.x * a.x + a.y * a.y);
        const double fp = fk[m], fc = fk[M + m], fr = fk[2 * M + m], ff = fk[3 * M + m];
        const double kr = g * k.x;
        acc[0] += kr * fp;
        acc[1] += kr * fc;
        acc[2] += kr * fr;
        acc[3] += kr * ff;
        acc[4] += fp * fp * w;
        acc[5] += fp * fc * w;
        acc[6] += fp * fr * w;
        acc[7] += fp * ff * w;
        acc[8] += fc * fc * w;
        acc[9] += fc * fr * w;
        acc[10] += fc * ff * w;
        acc[11] += fr * fr * w;
        acc[12] += fr * ff * w;
        acc[13] += ff * ff * w;
    }
    block_sum<14>(acc, red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 14; i++) out[n * 14 + i] = acc[i] * (i < 4 ? df2 : df4);
    }
}

int launch_prepsf_sums(const double *kim_re, const double *kim_im, const double *kpsf_re,
                       const double *kpsf_im, const double *pix, const double *knoise_re,
                       const double *knoise_im, const double *pnoise_stamp, double noise_scale,
                       const double *max_amp, const double *py, const double *px,
                       const int32_t *irow, const int32_t *icol, const double *fk,
                       const double *wgt, int64_t nstamps, int M, int64_t stride_n,
                       int64_t stride_r, int R, int C, double df2, double df4, double *out,
                       hipStream_t s)
{
    if (nstamps <= 0) return NGMIX_OK;
    if (!kim_re || !kim_im || ((kpsf_re != nullptr) != (kpsf_im != nullptr)) ||
        (!kpsf_re && !pix) || ((knoise_re != nullptr) != (knoise_im != nullptr)) ||
        (!knoise_re && !pnoise_stamp) || !max_amp || !irow || !icol || !fk || !wgt || !out ||
        M <= 0 || ((py != nullptr) != (px != nullptr)))
        return NGMIX_ERR_BAD_ARG;
    hipLaunchKernelGGL(prepsf_sums_kernel, dim3((unsigned)nstamps), dim3(BLOCK), 0, s, kim_re,
                       kim_im, kpsf_re, kpsf_im, pix, knoise_re, knoise_im, pnoise_stamp,
                       noise_scale, max_amp, (const cplx *)py, (const cplx *)px, irow, icol, fk, wgt,
                       M, stride_n, stride_r, R, C, df2, df4, out);
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

}  // namespace ngmix
