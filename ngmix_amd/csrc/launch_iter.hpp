// launch_iter.hpp -- launcher prototypes for the moment / iterative kernels
// (moments.hip, em.hip, derivs.hip).
#pragma once

#include "common.hpp"

namespace ngmix {

// moments.hip
int launch_weighted_sums_grid(const ngmix_batch *b, const ngmix_gauss2d *gmix,
                              void *res, int nmom, const double *maxrad,
                              int32_t *status, hipStream_t s);
int launch_weighted_sums_list(const ngmix_gauss2d *wt, int ng,
                              const ngmix_pixel *pixels, int64_t n, void *res,
                              int nmom, double maxrad, int32_t *status,
                              hipStream_t s);
int launch_admom_grid(const ngmix_admom_conf *conf, const ngmix_batch *b,
                      ngmix_gauss2d *wt, ngmix_admom_result *res, int32_t *status,
                      hipStream_t s);
int launch_admom_list(const ngmix_admom_conf *conf, const ngmix_pixel *pixels,
                      int64_t n, ngmix_gauss2d *wt, ngmix_admom_result *res,
                      int32_t *status, hipStream_t s);

// em.hip
int launch_em_grid(int kind, const ngmix_em_conf *conf, const ngmix_batch *b,
                   ngmix_gauss2d *gmix, int ngauss, ngmix_gauss2d *psf, int npsf,
                   ngmix_gauss2d *conv, const double *sky_in, int fzw, double *out,
                   int32_t *status, hipStream_t s);
int launch_em_list(int kind, const ngmix_em_conf *conf, ngmix_pixel *pixels,
                   int64_t n, double *sums, ngmix_gauss2d *gmix, int ngauss,
                   ngmix_gauss2d *psf, int npsf, ngmix_gauss2d *conv, int fzw,
                   double *out3, int32_t *status, hipStream_t s);

// em_wave.hip: stamps of <= 64x64 pixels, 1..3 object gaussians
int launch_em_wave(int kind, const ngmix_em_conf *conf, const ngmix_batch *b,
                   ngmix_gauss2d *gmix, int ngauss, ngmix_gauss2d *psf, int npsf,
                   ngmix_gauss2d *conv, const double *sky_in, int fzw, double *out,
                   int32_t *status, hipStream_t s);
int launch_em_wave_hi(int kind, const ngmix_em_conf *conf, const ngmix_batch *b,
                   ngmix_gauss2d *gmix, int ngauss, ngmix_gauss2d *psf, int npsf,
                   ngmix_gauss2d *conv, const double *sky_in, int fzw, double *out,
                   int32_t *status, hipStream_t s);
int launch_em_wave_8(int kind, const ngmix_em_conf *conf, const ngmix_batch *b,
                   ngmix_gauss2d *gmix, int ngauss, ngmix_gauss2d *psf, int npsf,
                   ngmix_gauss2d *conv, const double *sky_in, int fzw, double *out,
                   int32_t *status, hipStream_t s);

// derivs.hip
int launch_deriv_list(const double *gpars, const double *dcov, int ng,
                      const double *vv, const double *uu, const double *area,
                      int64_t npix, double *out, hipStream_t s);
int launch_deriv_grid(const ngmix_batch *b, const double *gpars, const double *dcov,
                      double *out, const int64_t *out_start, hipStream_t s);

}  // namespace ngmix
