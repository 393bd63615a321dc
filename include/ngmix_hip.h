/*
 * ngmix_hip.h -- C ABI of libngmix_hip.so, the MI355X (gfx950) implementation
 * of ngmix's numba pixel hot path.
 *
 * The reference has no FFI: its seam is the import of the njit functions in
 * ngmix/gmix/gmix_nb.py, ngmix/gmix/render_nb.py, ngmix/admom/admom_nb.py,
 * ngmix/em/em_nb.py, ngmix/fitting/derivs_nb.py, ngmix/pixels/pixels_nb.py and
 * ngmix/jacobian/jacobian_nb.py into the host classes (call sites listed at
 * each entry; paths relative to the reference checkout).  This header declares
 *
 *   (1) SEAM FORMS  -- one entry per njit function, same arguments (numpy
 *       structured arrays by pointer, HOST memory, caller-owned, results in
 *       place), so a maintainer can swap the import for a ctypes stub
 *       (INTEGRATION.md).  Pixel loops run on the GPU; O(ngauss) parameter
 *       prep (norms, model fills, convolution) is host arithmetic compiled
 *       from the same source as the device kernels.
 *   (2) BATCH FORMS -- the same operations over N independent stamps resident
 *       in HBM in a compact layout (8-byte val + 8-byte ierr per pixel,
 *       one 64-byte jacobian and one mixture per stamp); DEVICE pointers,
 *       asynchronous on the given hipStream_t.  This is the form the
 *       throughput numbers are quoted on.
 *
 * Numerics (all IEEE float64; no reduced precision anywhere):
 *   - SEAM forms, and BATCH forms called with NGMIX_BATCH_EXACT: no FMA
 *     contraction, the reference's operation order.  Per-pixel values (render,
 *     fill_fdiff, deriv_images, fill_pixels / fill_coords) are BIT-IDENTICAL
 *     to the reference's; sums over pixels (loglike, s2n, moments) differ only
 *     by summation order (<= 1e-12 relative in the tests).
 *   - BATCH forms by DEFAULT run the fused kernels: explicit fma(), a shared-
 *     centre form of chi^2 and the fexp cell index taken as round(chi^2/2)
 *     (differs from the reference's int(x - 0.5) only on exact ties, where its
 *     C2 polynomial changes by 1 ulp).  Same gates, same order of the sum over
 *     gaussians, results equal TO ROUNDING, not bit for bit: per-pixel values
 *     within 2e-13 of the stamp's peak and 1e-10 relative (BASELINE
 *     north_star's tolerance) wherever the value is not a cancellation
 *     residue; loglike / s2n sums within 1e-11 relative.  A binding that
 *     asserts equality must pass NGMIX_BATCH_EXACT.
 *
 * Return value: 0 = NGMIX_OK, >0 = the reference would have raised (code
 * below), <0 = runtime failure (ngmix_last_error() has the message).  Batch
 * forms additionally write one int32 status per stamp (same positive codes);
 * one bad stamp never aborts the batch.
 */
#ifndef NGMIX_HIP_H
#define NGMIX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------ */
#define NGMIX_OK 0
#define NGMIX_ERR_DET_TOO_LOW 1       /* GMixRangeError("det too low")  gmix_nb.py:203 */
#define NGMIX_ERR_T_TOO_LOW 2         /* GMixRangeError("T too low")    gmix_nb.py:207 */
#define NGMIX_ERR_G_RANGE 3           /* GMixRangeError("g >= 1")       gmix_nb.py:660 */
#define NGMIX_ERR_GTOT_ZERO 4         /* GMixRangeError("gtot == 0")    em_nb.py:91 */
#define NGMIX_ERR_ELOGL_ZERO 5        /* GMixRangeError("elogL == 0")   em_nb.py:113 */
#define NGMIX_ERR_ZERO_DIV 6          /* ZeroDivisionError (numba python error model) */
#define NGMIX_ERR_PIXELS_NOT_FILLED 7 /* RuntimeError               pixels_nb.py:57 */
#define NGMIX_ERR_HIP (-100)          /* HIP runtime / no device */
#define NGMIX_ERR_BAD_ARG (-101)

/* ---- flag bits written into result records (ngmix/flags.py:3-11) ------- */
#define NGMIX_FLAG_CEN_SHIFT 2
#define NGMIX_FLAG_NONPOS_FLUX 4
#define NGMIX_FLAG_NONPOS_SIZE 8
#define NGMIX_FLAG_LOW_DET 16
#define NGMIX_FLAG_MAXITER 32
/* LM result flags (ngmix/flags.py:14-20) */
#define NGMIX_FLAG_LM_SINGULAR_MATRIX (1 << 9)
#define NGMIX_FLAG_LM_NEG_COV_EIG (1 << 10)
#define NGMIX_FLAG_LM_NEG_COV_DIAG (1 << 11)
#define NGMIX_FLAG_LM_FUNC_NOTFINITE (1 << 12)
#define NGMIX_FLAG_EIG_NOTFINITE (1 << 13)
#define NGMIX_FLAG_ZERO_DOF (1 << 15)

/* ---- model ids (ngmix/gmix/gmix.py:1100-1110) -------------------------- */
#define NGMIX_MODEL_FULL 0
#define NGMIX_MODEL_GAUSS 1
#define NGMIX_MODEL_TURB 2
#define NGMIX_MODEL_EXP 3
#define NGMIX_MODEL_DEV 4
#define NGMIX_MODEL_BDF 6
#define NGMIX_MODEL_COELLIP 7
#define NGMIX_MODEL_CM 9
#define NGMIX_MODEL_BD 10

/* ---- record layouts = the reference's numpy dtypes --------------------- */

/* _gauss2d_dtype, ngmix/gmix/gmix.py:1196-1210 (104 bytes) */
typedef struct {
    double p, row, col, irr, irc, icc, det;
    int64_t norm_set;
    double drr, drc, dcc, norm, pnorm;
} ngmix_gauss2d;

/* _pixels_dtype / _coords_dtype, ngmix/pixels/pixels.py:72-86 */
typedef struct { double u, v, area, val, ierr, fdiff; } ngmix_pixel; /* 48 B */
typedef struct { double u, v, area; } ngmix_coord;                  /* 24 B */

/* _jacobian_dtype, ngmix/jacobian/jacobian.py:406-414 (64 bytes) */
typedef struct {
    double row0, col0, dvdrow, dvdcol, dudrow, dudcol, det, scale;
} ngmix_jacobian;

/* _admom_conf_dtype / _admom_result_dtype (align=True),
   ngmix/admom/admom.py:571-591 */
typedef struct {
    int32_t maxiter;
    double shiftmax, etol, Ttol;
    uint8_t cenonly;
    uint8_t no_cov; /* batch extension, in the reference record's padding (0 there):
                       skip the 7 x 7 covariance sums (sums_cov stays 0) -- for runs
                       that only want the converged weight, e.g. a fit's guess */
} ngmix_admom_conf; /* 40 B */

typedef struct {
    int32_t flags, numiter, npix;
    double wsum;
    double sums[7];
    double sums_cov[49];
    double pars[6];
    double rho4;
    double F[7];
} ngmix_admom_result; /* 584 B */

/* _em_conf_dtype (align=True), ngmix/em/em.py:440-449 */
typedef struct {
    double tol;
    int32_t maxiter, miniter;
    double sky;
    uint8_t vary_sky;
} ngmix_em_conf; /* 32 B */

/* EM run kinds: em_run / em_run_fixcen / em_run_fixcov / em_run_fluxonly.
   The per-gaussian sums record is the reference dtype of that kind
   (ngmix/em/em.py:451-521): 14 / 10 / 8 / 2 doubles. */
#define NGMIX_EM_FULL 0
#define NGMIX_EM_FIXCEN 1
#define NGMIX_EM_FIXCOV 2
#define NGMIX_EM_FLUXONLY 3

/* Weighted-moments result record for nmom in {6, 17}
   (get_moments_result_dtype, ngmix/gmix/gmix.py:1314-1330, align=True):
     int32 flags; int32 npix; double wsum; double sums[nmom];
     double sums_cov[nmom*nmom]; double pars[nmom]; double F[nmom];
   448 bytes for nmom=6, 2736 for nmom=17.  Passed as void*. */
#define NGMIX_MOMENTS_RESULT_BYTES(nmom) (16 + 8 * ((nmom) * (nmom) + 3 * (nmom)))

/* ======================================================================
 * runtime
 * ====================================================================== */
const char *ngmix_version(void);
const char *ngmix_last_error(void);
/* sizeof() of an ABI record by its type name ("ngmix_gauss2d", "ngmix_pixel",
   "ngmix_coord", "ngmix_jacobian", "ngmix_admom_conf", "ngmix_admom_result",
   "ngmix_em_conf", "ngmix_stamp", "ngmix_batch", "ngmix_lm_state",
   "ngmix_simple_sep_prior", "ngmix_lm_problem"), -1 for an
   unknown name: lets a binding check its own record layouts at load time */
int64_t ngmix_abi_sizeof(const char *type_name);
int ngmix_device_count(void);
int ngmix_set_device(int device);
/* thin wrappers so a numpy-only host can own device buffers */
int ngmix_device_malloc(void **ptr, size_t nbytes);
int ngmix_device_free(void *ptr);
int ngmix_memcpy_h2d(void *dst_dev, const void *src_host, size_t nbytes, void *stream);
int ngmix_memcpy_d2h(void *dst_host, const void *src_dev, size_t nbytes, void *stream);
int ngmix_memset_device(void *dst_dev, int value, size_t nbytes, void *stream);
int ngmix_stream_synchronize(void *stream);

/* ======================================================================
 * (1) SEAM FORMS -- host pointers, synchronous
 * ====================================================================== */

/* -- O(ngauss) parameter prep (host arithmetic) -- */

/* gmix_set_norms(gmix), gmix_nb.py:176-218; called at gmix.py:400,409 */
int ngmix_set_norms(ngmix_gauss2d *gmix, int64_t ngauss);
/* _gmix_fill_functions[name](gmix, pars), gmix_nb.py:307-427,469-558;
   called at gmix.py:443-445.  model = NGMIX_MODEL_* except CM */
int ngmix_fill_model(ngmix_gauss2d *gmix, int64_t ngauss, int model,
                     const double *pars, int64_t npars);
/* gmix_fill_cm(gmix, fracdev, TdByTe, Tfactor, pars), gmix_nb.py:430-466;
   called at gmix.py:1028-1030 */
int ngmix_fill_cm(ngmix_gauss2d *gmix, double fracdev, double TdByTe,
                  double Tfactor, const double *pars);
/* get_cm_Tfactor(fracdev, TdByTe), gmix_nb.py:561-593; gmix.py:1005 */
int ngmix_get_cm_Tfactor(double fracdev, double TdByTe, double *Tfactor);
/* g1g2_to_e1e2(g1, g2), gmix_nb.py:652-678 */
int ngmix_g1g2_to_e1e2(double g1, double g2, double *e1, double *e2);
/* gmix_convolve_fill(self, gmix, psf), gmix_nb.py:609-649;
   called at gmix.py:539, results.py:304 */
int ngmix_convolve_fill(ngmix_gauss2d *out, const ngmix_gauss2d *gmix,
                        int64_t ngauss, const ngmix_gauss2d *psf, int64_t npsf);
/* jacobian_get_vu / jacobian_get_rowcol, jacobian_nb.py:4-30;
   called at jacobian.py:160-182 */
void ngmix_jacobian_get_vu(const ngmix_jacobian *jacob, double row, double col,
                           double *v, double *u);
int ngmix_jacobian_get_rowcol(const ngmix_jacobian *jacob, double v, double u,
                              double *row, double *col);

/* -- pixel loops (GPU) -- */

/* fill_pixels(pixels, image, weight, jacob, ignore_zero_weight),
   pixels_nb.py:6-58; called at pixels.py:44-50 */
int ngmix_fill_pixels(ngmix_pixel *pixels, int64_t npixels, const double *image,
                      const double *weight, int64_t nrow, int64_t ncol,
                      const ngmix_jacobian *jacob, int ignore_zero_weight);
/* fill_coords(coords, nrow, ncol, jacob), pixels_nb.py:61-94; pixels.py:65-67 */
int ngmix_fill_coords(ngmix_coord *coords, int64_t nrow, int64_t ncol,
                      const ngmix_jacobian *jacob);
/* render(gmix, coords, image, fast_exp), render_nb.py:9-36; gmix.py:641-643.
   Adds into image; sets norms lazily (written back to gmix). */
int ngmix_render(ngmix_gauss2d *gmix, int64_t ngauss, const ngmix_coord *coords,
                 int64_t ncoords, double *image, int fast_exp);
/* get_loglike(gmix, pixels) -> (loglike, s2n_numer, s2n_denom, npix),
   gmix_nb.py:824-874; called at gmix.py:812 */
int ngmix_get_loglike(ngmix_gauss2d *gmix, int64_t ngauss,
                      const ngmix_pixel *pixels, int64_t npix, double *loglike,
                      double *s2n_numer, double *s2n_denom, int64_t *npix_out);
/* fill_fdiff(gmix, pixels, fdiff, start), gmix_nb.py:877-900;
   called at gmix.py:670-672, results.py:457-459 */
int ngmix_fill_fdiff(ngmix_gauss2d *gmix, int64_t ngauss,
                     const ngmix_pixel *pixels, int64_t npix, double *fdiff,
                     int64_t start);
/* get_model_s2n_sum(gmix, pixels), gmix_nb.py:903-937; gmix.py:775 */
int ngmix_get_model_s2n_sum(ngmix_gauss2d *gmix, int64_t ngauss,
                            const ngmix_pixel *pixels, int64_t npix,
                            double *s2n_sum);
/* get_weighted_sums / get_higher_order_weighted_sums(wt, pixels, res, maxrad),
   gmix_nb.py:681-821; called at gmix.py:750-752.  nmom = 6 or 17; adds into
   res.  wt must have its norms set (gmix.py:733). */
int ngmix_get_weighted_sums(const ngmix_gauss2d *wt, int64_t ngauss,
                            const ngmix_pixel *pixels, int64_t npix, void *res,
                            int nmom, double maxrad);
/* admom(confarray, wt, pixels, resarray), admom_nb.py:13-108; admom.py:349-354.
   wt (one gaussian) is updated in place, as in the reference. */
int ngmix_admom(const ngmix_admom_conf *conf, ngmix_gauss2d *wt,
                const ngmix_pixel *pixels, int64_t npix,
                ngmix_admom_result *res);
/* em_run{,_fixcen,_fixcov,_fluxonly}(conf, pixels, sums, gmix, gmix_psf,
   gmix_conv, fill_zero_weight) -> (numiter, frac_diff, sky),
   em_nb.py:15-127,357-469,702-816,1005-1106; called at em.py:278-286.
   pixels may be modified (zero-weight fill); gmix, gmix_conv are updated. */
int ngmix_em_run(int kind, const ngmix_em_conf *conf, ngmix_pixel *pixels,
                 int64_t npix, double *sums, ngmix_gauss2d *gmix, int64_t ngauss,
                 ngmix_gauss2d *gmix_psf, int64_t npsf, ngmix_gauss2d *gmix_conv,
                 int fill_zero_weight, int32_t *numiter, double *frac_diff,
                 double *sky);
/* deriv_images(gpars, dcov, vv, uu, area, out), derivs_nb.py:40-127;
   called at results.py:551-554, noise_cov.py:181-184.  out is (6, npix),
   accumulated into. */
int ngmix_deriv_images(const double *gpars, const double *dcov, int64_t ngauss,
                       const double *vv, const double *uu, const double *area,
                       int64_t npix, double *out);

/* ======================================================================
 * (2) BATCH FORMS -- device pointers, asynchronous on `stream`
 *
 * Compact stamp store in HBM:
 *   val[], ierr[]   float64, full-frame row-major stamps back to back;
 *                   ierr = sqrt(max(weight,0)) (pixels_nb.py:49-52)
 *   jac[]           one ngmix_jacobian per stamp
 *   stamps[]        one ngmix_stamp per stamp
 *   gmix[]          ngmix_gauss2d records, stamps[i].gm_off .. +ngauss
 * (v,u,area) are affine in (row,col) and are recomputed in-kernel with the
 * exact expression of jacobian_get_vu; the reference's pixel list (row-major,
 * weight<=0 dropped when ignore_zero_weight) is implicit: the k-th kept pixel
 * of a stamp is the reference's pixels[k].
 * ====================================================================== */

#define NGMIX_STAMP_IGNORE_ZERO_WEIGHT 1

typedef struct {
    int64_t pix_off; /* first pixel of the stamp in val/ierr/image arrays */
    int32_t nrow, ncol;
    int32_t gm_off;  /* first gaussian of the stamp's mixture in gmix[] */
    int32_t ngauss;
    int32_t flags;   /* NGMIX_STAMP_* */
    int32_t npix_kept; /* size of the reference's pixel list for this stamp
                          (== nrow*ncol when nothing is masked) */
} ngmix_stamp; /* 32 B */

/* diagnostic: evaluate every (gaussian, pixel) pair, no exact skipping */
#define NGMIX_BATCH_NO_SKIP 1
/* render / loglike / fdiff / s2n: use the EXACT kernels (no FMA contraction,
   the reference's operation order: per-pixel values bit-identical to the
   reference) instead of the default FUSED kernels (FMA + shared-centre
   algebra: the same values to <= ~1e-13 relative, about twice as fast) */
#define NGMIX_BATCH_EXACT 2
/* fused pixel-pass kernels: route complete-tile stamps through the compiler-
 * tracked load path instead of the hand-counted look-ahead loads (same
 * arithmetic, bit-identical results; a diagnostic for toolchain changes) */
#define NGMIX_BATCH_TRACKED_LOADS 4
/* ngmix_render_batch: image = model instead of image += model -- GMix.make_image's
 * zero-fill (ngmix/gmix/gmix.py:619-643) fused into the render, so that a fresh
 * image costs 8 written bytes per pixel instead of a memset plus a
 * read-modify-write; stamps that fail (and empty mixtures) are zero-filled */
#define NGMIX_BATCH_RENDER_OVERWRITE 8

/* A batch of stamps: HOST struct holding DEVICE pointers plus the few host
   facts a launch needs (LDS sizing, tile schedule). */
typedef struct {
    int64_t nstamps;
    const ngmix_stamp *stamps;  /* device, nstamps records */
    const double *val;          /* device; may be NULL for render / s2n */
    const double *ierr;         /* device; may be NULL for render */
    const ngmix_jacobian *jac;  /* device, nstamps records */
    int32_t max_ngauss;         /* max stamps[i].ngauss */
    int32_t max_npix;           /* max nrow*ncol */
    int32_t any_masked;         /* some stamp has npix_kept != nrow*ncol */
    int32_t flags;              /* NGMIX_BATCH_* */
    int32_t max_nrow;           /* max stamps[i].nrow, 0 = unknown */
    int32_t max_ncol;           /* max stamps[i].ncol, 0 = unknown (LDS is
                                   then sized from max_npix alone) */
} ngmix_batch;

/* Library-owned stamp store: builds the ngmix_batch above in HBM from HOST
   arrays -- for N objects at once what Observation.update_pixels -> make_pixels
   does per object (ngmix/observation.py:814-830, ngmix/pixels/pixels.py:6-52).
   create: stamp i is nrow[i] x ncol[i]; every stamp's mixture has `ngauss`
   gaussians (stamps[i].gm_off = i*ngauss).  upload: `images` (and `weights`,
   or NULL for unit weights) hold the stamps back to back, row-major; `jac` one
   record per stamp; ierr = sqrt(max(weight,0)) and the kept-pixel counts are
   computed on the device; a stamp without any positive weight is
   NGMIX_ERR_BAD_ARG (the reference's GMixFatalError).  The returned pointer
   is what the *_batch entry points take; release it with ngmix_batch_free. */
int ngmix_batch_create(ngmix_batch **out, int64_t nstamps, const int32_t *nrow,
                       const int32_t *ncol, int32_t ngauss, int ignore_zero_weight);
int ngmix_batch_upload(ngmix_batch *b, const double *images, const double *weights,
                       const ngmix_jacobian *jac, void *stream);
/* npix_kept[i] = size of the reference's pixel list of stamp i (after upload) */
int ngmix_batch_npix_kept(const ngmix_batch *b, int32_t *npix_kept);
int ngmix_batch_free(ngmix_batch *b);

/* DEVICE: the (stamp, mode) stage of the pre-psf Fourier moments
   (prepsfmom.py:337-422: _measure_moments_fft, _deconvolve_im_psf_inplace),
   one pass.  The transforms of the apodised, zero-padded image (kim_*), of the
   psf image (kpsf_*; NULL: a pixel in real space is deconvolved, pix (nmodes,)
   real) and of a noise image (knoise_*; NULL: the noise power is pnoise_stamp
   (nstamps,) per stamp; else |knoise|^2 * noise_scale per mode) arrive as
   arrays of real and of imaginary parts over (stamp, row of modes, column of
   modes), element (n, a, b) at n * stride_n + a * stride_r + b.  max_amp
   (nstamps,): |psf
   transform at k = 0|, amplitudes below 1e-5 of it are held there; py
   (nstamps, nrows), px (nstamps, ncols) complex128 or both NULL: exp(i k
   (image centre - psf centre)) per row / column of modes, irow / icol
   (nmodes,) the row / column of each mode; fk (4, nmodes) = the M+, Mx, Mr,
   Mf kernels; wgt (nmodes,) = 1, or 2 for a mode that also stands for its
   conjugate partner (real stamps: half of the plane is transformed).  out
   (nstamps, 14): the four sums x df2, then the upper triangle of their
   covariance x df4 */
int ngmix_prepsf_sums_batch(const double *kim_re, const double *kim_im, const double *kpsf_re,
                            const double *kpsf_im, const double *pix, const double *knoise_re,
                            const double *knoise_im, const double *pnoise_stamp,
                            double noise_scale, const double *max_amp, const double *py,
                            const double *px, const int32_t *irow, const int32_t *icol,
                            const double *fk, const double *wgt, int64_t nstamps, int nmodes,
                            int64_t stride_n, int64_t stride_r, int nrows, int ncols,
                            double df2, double df4, double *out, void *stream);
/* DEVICE: the innermost functions of the fast pixel evaluation over an array
   (fastexp_nb.py): which = 0 fexp = exp5_smooth(x) (:223-265; no range check:
   x in (-15.5, 1.5)), 1 apod_window(chi2) (:97-117), 2 apod_window_deriv(chi2)
   (:120-135).  x, out: n doubles on the device */
#define NGMIX_FASTEXP_FEXP 0
#define NGMIX_FASTEXP_APOD 1
#define NGMIX_FASTEXP_APOD_DERIV 2
int ngmix_fastexp_batch(const double *x, double *out, int64_t n, int which, void *stream);
/* ierr = sqrt(max(weight,0)) elementwise (pixels_nb.py:49-52); ierr == weight
 * (in place) is allowed; weight == NULL fills unit ierr */
int ngmix_weight_to_ierr_batch(const double *weight, double *ierr, int64_t n,
                               void *stream);
/* count kept pixels per stamp into stamps[i].npix_kept (pixels.py:33-37);
   a stamp with no positive weight gets npix_kept = 0 (the host raises
   GMixFatalError("no weights > 0") from that) */
int ngmix_count_kept_batch(ngmix_stamp *stamps, int64_t nstamps,
                           const double *ierr, void *stream);

/* The pixel sums of the linear template (psf) flux fit of
   PSFFluxFitModel.go (ngmix/fitting/results.py:700-770), one pass over the
   planes: model (total_pix doubles, the stamps' template images laid out as
   val), mult (nstamps,) or NULL (1): with mm = mult[s] * model,
   out[s] (nstamps, 4) = { sum(mm I w), sum(mm mm w), sum((mm - I)^2 w),
   #(ierr > 0) }, w = ierr^2.  First call with the templates' norms: xcorr and
   msq; second call with flux * norm: chi2. */
int ngmix_template_sums_batch(const ngmix_batch *batch, const double *model,
                              const double *mult, double *out, void *stream);

/* per-stamp model fill: pars is (nstamps, npars) row-major; gmix gets
   nstamps*ngauss records.  cm_extra is (nstamps,3) [fracdev,TdByTe,Tfactor]
   for NGMIX_MODEL_CM, else NULL. */
int ngmix_fill_model_batch(ngmix_gauss2d *gmix, int64_t nstamps, int ngauss,
                           int model, const double *pars, int npars,
                           const double *cm_extra, int32_t *status,
                           void *stream);
/* per-stamp convolution; psf holds nstamps*npsf records (one psf per stamp),
   out nstamps*ngauss*npsf */
int ngmix_convolve_fill_batch(ngmix_gauss2d *out, const ngmix_gauss2d *gmix,
                              int ngauss, const ngmix_gauss2d *psf, int npsf,
                              int64_t nstamps, int32_t *status, void *stream);
/* gmix_set_norms on every stamp's mixture of ngauss records */
int ngmix_set_norms_batch(ngmix_gauss2d *gmix, int ngauss, int64_t nstamps,
                          int32_t *status, void *stream);

/* get_loglike per stamp: out is (nstamps, 4) float64 =
   [loglike, s2n_numer, s2n_denom, npix].  Norms are set lazily in-kernel
   (written back) when gmix[gm_off].norm_set == 0, as the reference does. */
int ngmix_loglike_batch(const ngmix_batch *batch, ngmix_gauss2d *gmix,
                        double *out, int32_t *status, void *stream);
/* fill_fdiff per stamp: the k-th kept pixel writes fdiff[fdiff_start[i]+k] */
int ngmix_fill_fdiff_batch(const ngmix_batch *batch, ngmix_gauss2d *gmix,
                           double *fdiff, const int64_t *fdiff_start,
                           int32_t *status, void *stream);
/* render per stamp into image[pix_off + row*ncol + col] (adds) */
int ngmix_render_batch(const ngmix_batch *batch, ngmix_gauss2d *gmix,
                       double *image, int fast_exp, int32_t *status,
                       void *stream);
/* get_model_s2n_sum per stamp: out is (nstamps,) */
int ngmix_model_s2n_sum_batch(const ngmix_batch *batch, ngmix_gauss2d *gmix,
                              double *out, int32_t *status, void *stream);
/* get_weighted_sums per stamp: res is nstamps records of
   NGMIX_MOMENTS_RESULT_BYTES(nmom), added into; maxrad is (nstamps,) */
int ngmix_weighted_sums_batch(const ngmix_batch *batch,
                              const ngmix_gauss2d *gmix, void *res, int nmom,
                              const double *maxrad, int32_t *status,
                              void *stream);
/* admom per stamp: wt is one gaussian per stamp (stamps[i].gm_off, ngauss=1),
   updated in place; conf is a single record (host pointer) shared by all */
int ngmix_admom_batch(const ngmix_admom_conf *conf, const ngmix_batch *batch,
                      ngmix_gauss2d *wt, ngmix_admom_result *res,
                      int32_t *status, void *stream);
/* em_run per stamp.  gmix: nstamps*ngauss, gmix_psf: nstamps*npsf,
   gmix_conv: nstamps*ngauss*npsf, all updated as in the reference.
   sky_in (nstamps,) per-stamp sky (conf->sky is ignored); out (nstamps,3) =
   [numiter, frac_diff, sky].  val is read-only: the zero-weight fill of the
   reference acts on an on-chip copy. */
int ngmix_em_batch(int kind, const ngmix_em_conf *conf, const ngmix_batch *batch,
                   ngmix_gauss2d *gmix, int ngauss, ngmix_gauss2d *gmix_psf,
                   int npsf, ngmix_gauss2d *gmix_conv, const double *sky_in,
                   int fill_zero_weight, double *out, int32_t *status,
                   void *stream);
/* deriv_images per stamp over its kept pixels: gpars (sum ngauss,6) and dcov
   (sum ngauss,3,3) indexed by stamps[i].gm_off; out[out_start[i] + a*nk + k]
   for a in 0..5 and the k-th kept pixel, nk = npix_kept; accumulated into */
int ngmix_deriv_images_batch(const ngmix_batch *batch, const double *gpars,
                             const double *dcov, double *out,
                             const int64_t *out_start, void *stream);

/* ======================================================================
 * (3) BATCHED LEVENBERG-MARQUARDT (gauss / exp / dev, analytic jacobian)
 *
 * The reference runs one scipy.optimize.leastsq (MINPACK lmder) per object:
 * Fitter.go -> run_leastsq -> leastsqbound (ngmix/fitting/fitters.py:64-112,
 * leastsqbound.py:33-155,289-552) calling back into FitModel.calc_fdiff /
 * calc_jacobian (results.py:439-570) once per evaluation.  These entry
 * points advance N such fits in lock step, two launches per LM step:
 * ngmix_lm_eval_batch (objective + jacobian at every object's trial point,
 * reduced on chip to the 28 normal-equation sums per stamp) and
 * ngmix_lm_advance_batch (one step of the lmder logic per object).
 * ====================================================================== */
#define NGMIX_LM_NPMAX 14 /* parameters per object: bdf + 8 bands, bd + 7 bands, coellip with 5 gaussians */
/* per stamp sums over its nloc local parameters (the shared shape parameters
   followed by the flux of the stamp's band): J^T J upper triangle | J^T f | f.f */
#define NGMIX_LM_NPARS_GENERIC 255 /* ngmix_lm_advance_batch: the parameter-count hint that asks for the generic step */
#define NGMIX_LM_NSUMS(nloc) ((nloc) * ((nloc) + 1) / 2 + (nloc) + 1)
#define NGMIX_LM_NSUM NGMIX_LM_NSUMS(6) /* gauss / turb / exp / dev: 28 */

#define NGMIX_LM_PHASE_INIT 0   /* wants |f|^2, J^T f, J^T J at xt (= the guess) */
#define NGMIX_LM_PHASE_TRIAL 1  /* wants |f|^2 at xt (analytic mode: and J^T f, J^T J) */
#define NGMIX_LM_PHASE_DONE 2
#define NGMIX_LM_PHASE_JAC 3    /* wants J^T f, J^T J at xt (= x): forward-difference mode,
                                   and mode ANALYTIC_LAZY after an |f|^2-only trial */

/* ngmix_lm_state.mode */
#define NGMIX_LM_MODE_ANALYTIC 0 /* lmder; the jacobian comes with every evaluation */
#define NGMIX_LM_MODE_FD 1       /* lmdif: forward-difference jacobian only at accepted
                                    points, its n evaluations counted in nfev */
#define NGMIX_LM_MODE_ANALYTIC_LAZY 2 /* lmder, and a trial whose acceptance is predicted
                                    to END the fit (lmder's own predicted reduction <= ftol,
                                    or a step bound that will pass the xtol test) is
                                    evaluated for |f|^2 alone -- state.fonly = 1 -- as lmder
                                    itself does: it never forms the jacobian at the final
                                    point.  A prediction that fails asks for the jacobian
                                    at the accepted point in the next round (phase JAC),
                                    which is lmder's own order of evaluation: nfev, njev
                                    and every iterate are those of mode ANALYTIC, bit for
                                    bit */

/* one fit: lmder's loop variables, re-entrant (layout used by host and device) */
typedef struct {
    double x[NGMIX_LM_NPMAX];     /* last accepted point */
    double xt[NGMIX_LM_NPMAX];    /* trial point to evaluate next */
    double diag[NGMIX_LM_NPMAX];
    double R[NGMIX_LM_NPMAX * NGMIX_LM_NPMAX]; /* pivoted factor of the jacobian
                                      at x, upper triangle (MINPACK's fjac) */
    double qtf[NGMIX_LM_NPMAX];
    double step[NGMIX_LM_NPMAX];
    double fnorm, xnorm, delta, par, gnorm, pnorm;
    double ftol, xtol, gtol, factor;
    /* leastsqbound's bounds transform (leastsqbound.py:183-262): MINPACK
       iterates on the unconstrained internal parameters xi / xti; x / xt are
       their constrained images, the points the model is evaluated at.
       Without bounds the two coincide. */
    double xi[NGMIX_LM_NPMAX];    /* internal image of x */
    double xti[NGMIX_LM_NPMAX];   /* internal image of xt */
    double lo[NGMIX_LM_NPMAX];    /* lower bounds, -inf: none */
    double hi[NGMIX_LM_NPMAX];    /* upper bounds, +inf: none */
    /* forward-difference mode, fdjac2's points: column j of the jacobian is
       (f(xt with xt[j] := xstep[j]) - f(xt)) / hstep[j] */
    double xstep[NGMIX_LM_NPMAX];
    double hstep[NGMIX_LM_NPMAX];
    int32_t ipvt[NGMIX_LM_NPMAX]; /* 0-based */
    int32_t n, iter, nfev, njev, info, phase, maxfev, mode;
    int32_t bounded;
    int32_t fonly; /* 1: the evaluation at xt needs |f|^2 only (mode ANALYTIC_LAZY) */
} ngmix_lm_state;

/* HOST: initialise nobj states from the guesses x0 (nobj, npars);
   mode = NGMIX_LM_MODE_*; lo / hi: (npars,) bounds shared by all the fits
   (-inf / +inf: none) or NULL, the `bounds` of leastsqbound */
int ngmix_lm_init(ngmix_lm_state *states, int64_t nobj, int npars,
                  const double *x0, double ftol, double xtol, double gtol,
                  int maxfev, double factor, int mode, const double *lo,
                  const double *hi);
/* DEVICE: the same initialisation for states resident on the device; x0
   (nobj, npars) device array; lo / hi host arrays or NULL */
int ngmix_lm_init_batch(ngmix_lm_state *states, int64_t nobj, int npars,
                        const double *x0, double ftol, double xtol, double gtol,
                        int maxfev, double factor, int mode, const double *lo,
                        const double *hi, void *stream);
/* HOST: consume one evaluation per object -- ff (nobj,), g (nobj, NPMAX),
   A (nobj, NPMAX*NPMAX) at states[i].xt -- and advance; returns the number
   of fits still running.  The same code the device kernel runs; exists for
   small problems and for testing the iteration against MINPACK on the CPU. */
int64_t ngmix_lm_advance_host(ngmix_lm_state *states, int64_t nobj,
                              const double *ff, const double *g, const double *A);
/* DEVICE: evaluate every stamp of every running fit at its object's trial
   point.  fd = 0: residuals + analytic jacobian (gauss, exp, dev; states in
   NGMIX_LM_MODE_ANALYTIC); fd = 1: residuals, plus MINPACK's forward-difference
   jacobian when the object's phase asks for one (gauss, turb, exp, dev, bdf,
   bd, and co-elliptical gaussians as model = NGMIX_MODEL_COELLIP + 256 * ngauss
   with ngauss <= 5 and parameters cen1, cen2, g1, g2, T_1.., F_1..; states in
   NGMIX_LM_MODE_FD).  states: device array; stamp_obj (nstamps,)
   object of each stamp or NULL (stamp i = object i); stamp_band (nstamps,) or
   NULL (band 0); psf: nstamps*npsf gauss2d records or NULL with npsf = 0;
   sums: (nstamps, NGMIX_LM_NSUMS(nloc)), nloc = the model's npars (6, bdf 7,
   bd 8); status: per stamp (NGMIX_ERR_G_RANGE: model out of range at the
   trial point, sums then carry ff = +inf like the reference's LOWVAL
   residuals).  stamp_stats (may be NULL; fd = 0 only): (nstamps, 2), the
   s2n_numer and s2n_denom sums of get_loglike (gmix_nb.py:862-864) at the
   trial point -- they fall out of the normal-equation sums, so that the
   statistics of FitModel.set_fit_result (results.py:45-72) need no pixel
   pass of their own after the fit */
int ngmix_lm_eval_batch(const ngmix_batch *batch, int model, int fd,
                        const ngmix_lm_state *states, const int32_t *stamp_obj,
                        const int32_t *stamp_band, const ngmix_gauss2d *psf,
                        int npsf, double *sums, int32_t *status,
                        double *stamp_stats, void *stream);
/* DEVICE: fold stamps obj_start[i]..obj_start[i+1] (NULL: stamp i) into
   object i's normal equations and advance its state; *nactive (device int32,
   may be NULL) receives the number of fits still running.  obj_sums (may be
   NULL): (nobj, NGMIX_LM_NSUMS(n)) further rows of each object's residual
   vector already reduced over the object's n parameters -- the prior rows at
   the head of the reference's fdiff (results.py:454, joint_prior.py:86-120);
   in forward-difference mode their jacobian is by the state's xstep / hstep.
   nloc may carry the fits' parameter count as nloc + 256 * npars (npars =
   nloc - 1 + the number of bands; 0 = not said), which selects the form of
   the step: 6-8 parameters from registers, one fit per lane; 9-14 by a team
   of 16 lanes per fit with the fit's arrays in LDS (csrc/lm_team.hip); not
   said: the team form built for NGMIX_LM_NPMAX parameters (any fit; ahead of
   the one-thread code at every count); npars = NGMIX_LM_NPARS_GENERIC: the
   generic one-thread code with a private state of NGMIX_LM_NPMAX parameters
   (what the tests compare the other forms with).  The three forms leave
   byte-identical state records.
   stamp_stats / obj_stats (both or neither, may be NULL): whenever a fit
   moves to its trial point (lmder counts an iteration; the starting point on
   the first call) obj_stats[i] (nobj, 2) takes the sum of its stamps'
   stamp_stats rows from ngmix_lm_eval_batch: when the fit ends they are
   s2n_numer / s2n_denom at the solution, and lnprob = -fnorm^2 / 2 */
int ngmix_lm_advance_batch(ngmix_lm_state *states, int64_t nobj,
                           const int64_t *obj_start, const int32_t *stamp_band,
                           const double *sums, int nloc, const double *obj_sums,
                           int32_t *nactive, const double *stamp_stats,
                           double *obj_stats, void *stream);

/* The separable joint priors of the reference (joint_prior.py:
   PriorSimpleSep :10-120, PriorBDFSep :484-674, PriorBDSep :267-481) in a
   form a kernel can evaluate: gaussian centre terms (priors/multivariate.py
   CenPrior), the Bernstein-Armstrong shape prior (priors/shape.py GPriorBA),
   and one 1-d term of priors/priors.py per remaining parameter -- T, then
   nmid "middle" terms (fracdev for 'bdf'; logTratio, fracdev for 'bd'), then
   one flux term per band:
     FLAT                par = minval, maxval
     TWO_SIDED_ERF       par = minval, width_at_min, maxval, width_at_max
     NORMAL              par = mean, sigma
     LOGNORMAL           par = logmean, logivar, lnprob_max, shift (0: none)
     TRUNCATED_GAUSSIAN  par = mean, sigma, minval, maxval
   rows_mode ROWS_LNPROB: every row is sqrt(max(-2 ln p, 0)) of its term
   (PriorSimpleSep.fill_fdiff); ROWS_FDIFF: each term's own get_fdiff -- the
   signed (x - mean) / sigma of the gaussian terms, cen_sinv for the centre
   (PriorBDFSep / PriorBDSep.fill_fdiff).  The first 188 bytes are the round-3
   record; a record with nmid = 0, rows_mode = 0 is a PriorSimpleSep */
#define NGMIX_PRIOR_FLAT 0
#define NGMIX_PRIOR_TWO_SIDED_ERF 1
#define NGMIX_PRIOR_NORMAL 2
#define NGMIX_PRIOR_LOGNORMAL 3
#define NGMIX_PRIOR_TRUNCATED_GAUSSIAN 4
#define NGMIX_PRIOR_MAXBAND 3
#define NGMIX_PRIOR_MAXMID 2
#define NGMIX_PRIOR_ROWS_LNPROB 0
#define NGMIX_PRIOR_ROWS_FDIFF 1
typedef struct {
    double cen1, cen2, cen_s2inv1, cen_s2inv2;
    double g_sig2inv;
    double T_par[4];
    double F_par[NGMIX_PRIOR_MAXBAND][4];
    int32_t T_kind, nband;
    int32_t F_kind[NGMIX_PRIOR_MAXBAND];
    int32_t nmid;
    double cen_sinv1, cen_sinv2;
    double mid_par[NGMIX_PRIOR_MAXMID][4];
    int32_t mid_kind[NGMIX_PRIOR_MAXMID];
    int32_t rows_mode;
    int32_t pad_;
} ngmix_simple_sep_prior; /* 288 B */
/* DEVICE: the prior rows [cen1, cen2, g, T, mid..., F_band...] of every
   object at its trial point states[i].xt, their jacobian by
   differences (analytic mode: the reference's one-sided steps
   step_rel * max(1, |x_j|), backward where the forward point is out of range,
   results.py:572-625; forward-difference mode: the state's xstep / hstep),
   reduced to obj_sums (nobj, NGMIX_LM_NSUMS(5 + nmid + nband)) for
   ngmix_lm_advance_batch.  An out-of-range point (g >= 1, outside a flat
   prior) gives r.r = +inf: the reference's GMixRangeError -> -inf residuals */
int ngmix_lm_prior_sums_batch(const ngmix_lm_state *states, int64_t nobj,
                              const ngmix_simple_sep_prior *prior, double step_rel,
                              double *obj_sums, void *stream);
/* DEVICE: what the prior adds to the statistics of the finished fits, at the
   points they stand at (states[i].x): ffx (nobj,) = the sum of squares of the
   prior's finite rows -- the part of |f|^2 run_leastsq leaves out of chi2/dof
   (leastsqbound.py:97; ngmix_lm_finalize_batch's ff_extra) -- and lnp (nobj,)
   = ln p, which calc_lnprob adds to the loglike (results.py:410-437); outside
   the prior's range: 0 and -inf */
int ngmix_lm_prior_finish_batch(const ngmix_lm_state *states, int64_t nobj,
                                const ngmix_simple_sep_prior *prior, double *ffx,
                                double *lnp, void *stream);
/* DEVICE: out[o] = sum of fdiff^2 over the first nskip LISTED pixels of stamp
   stamp_of[o] under mixture o of gmix (nobj x ngauss normalised records): the
   reference reserves more residual rows for a joint prior than the prior
   fills (results.py:1050-1078 against joint_prior.py:86-120), the pixel rows
   start right after the filled ones (results.py:454-461), and chi2/dof leaves
   ALL reserved rows out (leastsqbound.py:97) -- so the first pixels go with
   the prior rows into ff_extra.  fill_fdiff's arithmetic (gmix_nb.py:877-900) */
int ngmix_first_pixels_fdiff2_batch(const ngmix_batch *batch, const int64_t *stamp_of,
                                    const ngmix_gauss2d *gmix, int ngauss, int64_t nobj,
                                    int nskip, double *out, void *stream);
/* HOST: the same sums for states in host memory (testing aid; the code the
   kernel runs) */
int ngmix_lm_prior_sums_host(const ngmix_lm_state *states, int64_t nobj,
                             const ngmix_simple_sep_prior *prior, double step_rel,
                             double *obj_sums);
/* HOST: rows (max 4 + nmid + nband) and ln p of one parameter vector: returns the
   number of rows, or -1 when the point is out of range (testing aid; the same
   code the kernel runs) */
int ngmix_simple_sep_prior_eval(const ngmix_simple_sep_prior *prior,
                                const double *pars, double *rows, double *lnprob);

/* DEVICE: package every fit as run_leastsq does (leastsqbound.py:33-155):
   rec is (nobj, 4 + 2n + 2n^2) doubles per object, n = states[i].n:
   [flags, nfev, ier, dof | pars (n) | pars_err (n) | pars_cov0 (n,n) |
   pars_cov (n,n)]; npix_obj (nobj,) = pixels in the object's residual vector;
   ff_extra (nobj,) or NULL: the part of |f|^2 that came through obj_sums (the
   prior rows are left out of chi2/dof, leastsqbound.py:97); pdef / cdef =
   the reference's PDEF / CDEF sentinels (defaults.py:10-11) */
int ngmix_lm_finalize_batch(const ngmix_lm_state *states, int64_t nobj,
                            const int64_t *npix_obj, const double *ff_extra,
                            double pdef, double cdef, double *rec, void *stream);

/* DEVICE: the statistics of FitModel.set_fit_result (results.py:45-72,
   398-408) and the integer columns of the records above, laid out for one
   contiguous download.  rec: the records of ngmix_lm_finalize_batch
   (npars = states[i].n for every i).  The loglike sums at the solutions come
   either from obj_stats (nobj, 2) kept by ngmix_lm_advance_batch (lnprob is
   then -fnorm^2 / 2 and npix = npix_obj) or from tot (nobj, 4) = lnprob,
   s2n_numer, s2n_denom, npix of a get_loglike pass (exactly one of the two is
   not NULL).  head: (nobj, 2 npars) = pars | pars_err; cols:
   (NGMIX_LM_NCOLS, nobj) column-major = flags, nfev, ier, dof (of the fit),
   njev, lnprob, s2n_numer, s2n_denom, npix, dof, chi2per, s2n -- the seven
   statistics are NaN where flags != 0, as set_fit_result leaves them out.
   cov_tri (may be NULL): (nobj, npars (npars + 1) / 2), the row-major upper
   triangle of pars_cov -- the matrix is symmetric to the bit, so the
   triangle is all a download needs */
#define NGMIX_LM_NCOLS 12
int ngmix_lm_pack_batch(const ngmix_lm_state *states, int64_t nobj, int npars,
                        const double *rec, const double *obj_stats,
                        const double *tot, const int64_t *npix_obj, double *head,
                        double *cols, double *cov_tri, void *stream);

/* DEVICE: `nrounds` lock-step rounds queued by ONE host call -- per round
   ngmix_lm_eval_batch, ngmix_lm_prior_sums_batch (when `prior` is set) and
   ngmix_lm_advance_batch, nothing between them on the host.  The reference's
   driver calls back into Python twice per LM step (leastsqbound.py:445-491);
   here the host is not in the loop at all: a fit that has finished is skipped
   by every later launch (its phase is read on the device), so rounds may be
   queued blind -- a round that finds every fit finished costs two empty
   launches.  The members of ngmix_lm_problem are the arguments of the three
   entry points above.  counts (device int32, nrounds slots, may be NULL):
   slot r receives the number of fits still running after round r;
   counts_host (pinned host memory or NULL) receives one copy of all the slots
   behind the last round.  events (may be NULL): 3 nrounds hipEvent_t recorded
   before the pixel pass, between the pixel pass and the step, and after the
   step of each round (ngmix_event_*: timing without a host in the loop). */
typedef struct {
    const ngmix_batch *batch;
    ngmix_lm_state *states;          /* device, nobj records */
    int64_t nobj;
    const int32_t *stamp_obj;        /* (nstamps,) or NULL */
    const int32_t *stamp_band;       /* (nstamps,) or NULL */
    const int64_t *obj_start;        /* (nobj + 1,) or NULL */
    const ngmix_gauss2d *psf;        /* nstamps * npsf records or NULL */
    double *sums;                    /* (nstamps, NGMIX_LM_NSUMS(nloc)) */
    int32_t *status;                 /* (nstamps,) */
    double *stamp_stats;             /* (nstamps, 2) or NULL */
    double *obj_stats;               /* (nobj, 2) or NULL */
    const ngmix_simple_sep_prior *prior; /* HOST record or NULL */
    double *obj_sums;                /* (nobj, NGMIX_LM_NSUMS(npars)) or NULL */
    double prior_step;               /* step_rel of ngmix_lm_prior_sums_batch */
    int32_t model, fd, npsf;
    int32_t nloc_npars;              /* nloc + 256 * npars, as ngmix_lm_advance_batch */
    double *jac_point;               /* (nobj, 3, NGMIX_LM_NPMAX) or NULL: forward-difference
                                        passes that evaluate a jacobian leave xt | xstep |
                                        hstep of the state there -- the point of the fit's
                                        LAST jacobian once it has ended
                                        (ngmix_lm_precise_cov_batch) */
} ngmix_lm_problem;
int ngmix_lm_rounds_batch(const ngmix_lm_problem *problem, int nrounds,
                          int32_t *counts, int32_t *counts_host, void **events,
                          void *stream);
/* DEVICE: the covariance factor of ill-conditioned forward-difference fits.
   scipy's leastsq takes cov_x from MINPACK's QR of the last jacobian
   (leastsqbound.py:76-118,535-552); the rounds above carry the Cholesky factor
   of J^T J in doubles, which stops existing numerically near cond(J) = 1e8 --
   where the co-elliptical psf fits with three and more gaussians live
   (CoellipFitter, fitters.py:120-141).  After the rounds of `problem` have
   ended and before ngmix_lm_finalize_batch, this call re-makes R and ipvt of
   every fit that ended with info 1-4: one more forward-difference pass at
   jac_point with X^T X accumulated in double-double (error-free
   transformations), then factor_normal's pivoted Cholesky in double-double
   arithmetic, one wave per fit.  Serves nloc >= NGMIX_LM_PRECISE_MIN_NLOC;
   psums: workspace (nstamps, 2, NGMIX_LM_NSUMS(nloc)).  The iterates, nfev
   and ier are untouched */
#define NGMIX_LM_PRECISE_MIN_NLOC 10
int ngmix_lm_precise_cov_batch(const ngmix_lm_problem *problem, double *psums,
                               void *stream);
/* HOST: n timing events (hipEvent_t) for ngmix_lm_rounds_batch / to record
   on a stream; elapsed milliseconds between two recorded events (both must
   have completed: synchronise the stream or the later event first) */
int ngmix_events_create(int n, void **events);
int ngmix_events_destroy(int n, void **events);
int ngmix_event_record(void *event, void *stream);
int ngmix_event_synchronize(void *event);
int ngmix_event_elapsed_ms(void *start, void *stop, float *ms);

/* HOST: the launch census -- how many times each batch kernel variant has been
   dispatched by this process, as "name<TAB>count" lines ("em_wave_kernel<64,
   16, 0, 1, 1>\t35\n...") written to buf (NUL-terminated, truncated to
   buflen); returns the size the full text needs.  reset != 0 clears the
   counts afterwards.  A test asserts with it WHICH kernel served a workload:
   a silent fall-back to a generic kernel is a performance bug no parity
   test sees. */
int64_t ngmix_launch_census(char *buf, int64_t buflen, int reset);

/* ======================================================================
 * (4) MULTI-GPU: one process per GPU; objects are sharded by contiguous
 * blocks and nothing crosses ranks except the per-object RESULT RECORDS
 * (32 B loglike, 584 B admom, ...), all-gathered over RCCL / xGMI.  The
 * reference loops objects serially (ngmix/runners.py:116-149,
 * ngmix/bootstrap.py:67-154); this is the exchange that replaces its loop's
 * result list.  RCCL is bound at first use (dlopen): without it these entry
 * points return NGMIX_ERR_HIP with a message, they never fall back.
 *   rank 0: ngmix_comm_unique_id(id) -> ship the 128 bytes to the other ranks
 *   every rank (after ngmix_set_device): ngmix_comm_init_rank(&comm, n, id, r)
 *   ngmix_allgather_results(comm, send, recv, nrecords, record_bytes, stream):
 *       recv[r*nrecords .. ] = rank r's `nrecords` records, device pointers,
 *       asynchronous on `stream` (every rank passes the same nrecords)
 * ====================================================================== */
int ngmix_comm_unique_id(void *id128);
int ngmix_comm_init_rank(void **comm, int nranks, const void *id128, int rank);
int ngmix_comm_destroy(void *comm);
int ngmix_allgather_results(void *comm, const void *send, void *recv,
                            int64_t nrecords, int64_t record_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NGMIX_HIP_H */
