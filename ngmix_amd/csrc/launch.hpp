// launch.hpp -- host-side launcher prototypes shared by the .hip files.
#pragma once

#include "common.hpp"

namespace ngmix {

// pixpass.hip
int launch_loglike_grid(const ngmix_batch *b, ngmix_gauss2d *gmix, double *out,
                        int32_t *status, void *stream);
int launch_fdiff_grid(const ngmix_batch *b, ngmix_gauss2d *gmix, double *fdiff,
                      const int64_t *fdiff_start, int32_t *status, void *stream);
int launch_render_grid(const ngmix_batch *b, ngmix_gauss2d *gmix, double *image,
                       int fast_exp, int32_t *status, void *stream);
int launch_s2n_grid(const ngmix_batch *b, ngmix_gauss2d *gmix, double *out,
                    int32_t *status, void *stream);
int list_partial_doubles(void);
int launch_render_list(const ngmix_gauss2d *gm, int ng, const ngmix_coord *coords,
                       int64_t n, double *image, int fast_exp, hipStream_t s);
int launch_pixpass_list(int op, const ngmix_gauss2d *gm, int ng,
                        const ngmix_pixel *pixels, int64_t n, double *fdiff,
                        int64_t start, double *partial, hipStream_t s);
int launch_fill_pixels(ngmix_pixel *pixels, int64_t npixels, const double *image,
                       const double *weight, int nrow, int ncol,
                       const ngmix_jacobian &jac, int izw, int *count_out,
                       hipStream_t s);
int launch_fill_coords(ngmix_coord *coords, int nrow, int ncol,
                       const ngmix_jacobian &jac, hipStream_t s);
int launch_weight_to_ierr(const double *w, double *ierr, int64_t n, hipStream_t s);
int launch_fastexp(const double *x, double *out, int64_t n, int which, hipStream_t s);
int launch_prepsf_sums(const double *kim_re, const double *kim_im, const double *kpsf_re,
                       const double *kpsf_im, const double *pix, const double *knoise_re,
                       const double *knoise_im, const double *pnoise_stamp, double noise_scale,
                       const double *max_amp, const double *py, const double *px,
                       const int32_t *irow, const int32_t *icol, const double *fk,
                       const double *wgt, int64_t nstamps, int M, int64_t stride_n,
                       int64_t stride_r, int R, int C, double df2, double df4, double *out,
                       hipStream_t s);
int launch_first_pixels_fdiff2(const ngmix_batch *b, const int64_t *stamp_of,
                               const ngmix_gauss2d *gm, int ngauss, int64_t nobj, int nskip,
                               double *out, hipStream_t s);
int launch_count_kept(ngmix_stamp *stamps, int64_t nstamps, const double *ierr,
                      hipStream_t s);

// template.hip
int launch_template_sums(const ngmix_batch *b, const double *model, const double *mult,
                         double *out, hipStream_t s);

// gmixprep.hip
int launch_fill_model(ngmix_gauss2d *gmix, int64_t nstamps, int ngauss, int model,
                      const double *pars, int npars, const double *cm_extra,
                      int32_t *status, hipStream_t s);
int launch_convolve_fill(ngmix_gauss2d *out, const ngmix_gauss2d *gmix, int ngauss,
                         const ngmix_gauss2d *psf, int npsf, int64_t nstamps,
                         int32_t *status, hipStream_t s);
int launch_set_norms(ngmix_gauss2d *gmix, int ngauss, int64_t nstamps,
                     int32_t *status, hipStream_t s);

// lmfit.hip
int launch_lm_eval(const ngmix_batch *b, int model, int fd, const ngmix_lm_state *states,
                   const int32_t *stamp_obj, const int32_t *stamp_band,
                   const ngmix_gauss2d *psf, int npsf, double *sums, int32_t *status,
                   double *stamp_stats, hipStream_t s, double *jac_point = nullptr,
                   bool precise = false);
// lm_precise.hip: the covariance factor of the ill-conditioned forward-difference
// fits from double-double normal equations
int launch_lm_precise_cov(const ngmix_lm_problem *p, double *psums, hipStream_t s);
int launch_lm_advance(ngmix_lm_state *states, int64_t nobj, const int64_t *obj_start,
                      const int32_t *stamp_band, const double *sums, int nloc,
                      const double *obj_sums, int32_t *nactive, const double *stamp_stats,
                      double *obj_stats, hipStream_t s, bool zero_count = true);
// lm_team.hip: the same step by 16 lanes per fit (teams = fits per wave: 1, 2, 4)
int launch_lm_advance_team(ngmix_lm_state *states, int64_t nobj, const int64_t *obj_start,
                           const int32_t *stamp_band, const double *sums, int nloc, int npars,
                           const double *obj_sums, int32_t *nactive,
                           const double *stamp_stats, double *obj_stats, int teams,
                           hipStream_t s);
int launch_lm_rounds(const ngmix_lm_problem *p, int nrounds, int32_t *counts,
                     int32_t *counts_host, void **events, hipStream_t s);

int launch_lm_prior_finish(const ngmix_lm_state *states, int64_t nobj,
                           const ngmix_simple_sep_prior *prior, double *ffx, double *lnp,
                           hipStream_t s);
int launch_lm_prior_sums(const ngmix_lm_state *states, int64_t nobj,
                         const ngmix_simple_sep_prior *prior, double step_rel,
                         double *obj_sums, hipStream_t s);
int launch_lm_init(ngmix_lm_state *states, int64_t nobj, int npars, const double *x0,
                   double ftol, double xtol, double gtol, int maxfev, double factor,
                   int mode, const double *lo, const double *hi, hipStream_t s);
int launch_lm_finalize(const ngmix_lm_state *states, int64_t nobj,
                       const int64_t *npix_obj, const double *ff_extra, double pdef,
                       double cdef, double *rec, hipStream_t s);

int launch_lm_pack(const ngmix_lm_state *states, int64_t nobj, int npars, const double *rec,
                   const double *obj_stats, const double *tot, const int64_t *npix_obj,
                   double *head, double *cols, double *cov_tri, hipStream_t s);

}  // namespace ngmix
