/*
 * ngmix_oracle.h -- CPU restatement (plain C) of the ngmix pixel hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle for the HIP kernels in
 * ngmix_amd/csrc.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (ngmix_amd) never does.
 *
 * Every function follows the operation order of the reference's numba source
 * (file:line given at each definition in ngmix_oracle.c, relative to
 * /root/reference).  Build with -ffp-contract=off and no fast-math so each
 * IEEE-754 double operation matches the un-contracted LLVM code numba emits.
 *
 * Parity pinned: checked against golden vectors produced by importing the
 * reference itself (oracle/gen_golden*.py -> tests/golden/ *.npz; per-function
 * vectors and, at the shape of every BASELINE config, the reference's own
 * render / loglike / fdiff (c2, c5), admom and em_run (c4: bit for bit) and
 * complete fits (lm_c3 through bench.py's CPU leg)); see
 * tests/test_oracle_golden.py.
 *
 * Struct layouts are the reference's numpy dtypes.
 */
#ifndef NGMIX_ORACLE_H
#define NGMIX_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes shared with include/ngmix_hip.h */
#define ORA_OK 0
#define ORA_ERR_DET_TOO_LOW 1       /* GMixRangeError("det too low") */
#define ORA_ERR_T_TOO_LOW 2         /* GMixRangeError("T too low") */
#define ORA_ERR_G_RANGE 3           /* GMixRangeError("g >= 1") */
#define ORA_ERR_GTOT_ZERO 4         /* GMixRangeError("gtot == 0") */
#define ORA_ERR_ELOGL_ZERO 5        /* GMixRangeError("elogL == 0") */
#define ORA_ERR_ZERO_DIV 6          /* ZeroDivisionError (numba error model) */
#define ORA_ERR_PIXELS_NOT_FILLED 7 /* RuntimeError */

/* ngmix/gmix/gmix.py:1196-1210, 104 bytes */
typedef struct {
    double p, row, col, irr, irc, icc, det;
    int64_t norm_set;
    double drr, drc, dcc, norm, pnorm;
} ora_gauss2d;

/* ngmix/pixels/pixels.py:72-86 */
typedef struct { double u, v, area, val, ierr, fdiff; } ora_pixel; /* 48 B */
typedef struct { double u, v, area; } ora_coord;                  /* 24 B */

/* ngmix/jacobian/jacobian.py:406-414 */
typedef struct {
    double row0, col0, dvdrow, dvdcol, dudrow, dudcol, det, scale;
} ora_jacobian;

/* ngmix/admom/admom.py:571-591 (align=True) */
typedef struct {
    int32_t maxiter;
    double shiftmax, etol, Ttol;
    uint8_t cenonly;
} ora_admom_conf; /* 40 B */

typedef struct {
    int32_t flags, numiter, npix;
    double wsum;
    double sums[7];
    double sums_cov[49];
    double pars[6];
    double rho4;
    double F[7];
} ora_admom_result; /* 584 B */

/* ngmix/em/em.py:440-449 (align=True) */
typedef struct {
    double tol;
    int32_t maxiter, miniter;
    double sky;
    uint8_t vary_sky;
} ora_em_conf; /* 32 B */

/* scalar math: ngmix/fastexp_nb.py */
double ora_fexp(double x);
double ora_apod_window(double chi2);
double ora_apod_window_deriv(double chi2);
void ora_fexp_array(const double *x, double *out, int64_t n);
void ora_apod_array(const double *chi2, double *w, double *dw, int64_t n);

/* pixel-gaussian evaluation: ngmix/gmix/gmix_nb.py:14-105 */
double ora_gmix_eval_pixel_fast(const ora_gauss2d *gm, int64_t ng,
                                double v, double u, double area);
double ora_gmix_eval_pixel(const ora_gauss2d *gm, int64_t ng,
                           double v, double u, double area);

/* parameter prep: gmix_nb.py:176-240, 307-678 */
int ora_gmix_set_norms(ora_gauss2d *gm, int64_t ng);
void ora_gauss2d_set(ora_gauss2d *g, double p, double row, double col,
                     double irr, double irc, double icc);
int ora_g1g2_to_e1e2(double g1, double g2, double *e1, double *e2);
int ora_get_cm_Tfactor(double fracdev, double TdByTe, double *Tfactor);
/* model: 0 full,1 gauss,2 turb,3 exp,4 dev,6 bdf,7 coellip,9 cm,10 bd */
int ora_gmix_fill(ora_gauss2d *gm, int64_t ng, const double *pars,
                  int64_t npars, int model, double fracdev, double TdByTe,
                  double Tfactor);
int ora_gmix_convolve_fill(ora_gauss2d *out, const ora_gauss2d *gm, int64_t ng,
                           const ora_gauss2d *psf, int64_t npsf);

/* pixels / jacobian: ngmix/pixels/pixels_nb.py, ngmix/jacobian/jacobian_nb.py */
void ora_jacobian_get_vu(const ora_jacobian *j, double row, double col,
                         double *v, double *u);
int ora_jacobian_get_rowcol(const ora_jacobian *j, double v, double u,
                            double *row, double *col);
int ora_fill_pixels(ora_pixel *pixels, int64_t npixels, const double *image,
                    const double *weight, int64_t nrow, int64_t ncol,
                    const ora_jacobian *jacob, int ignore_zero_weight);
void ora_fill_coords(ora_coord *coords, int64_t nrow, int64_t ncol,
                     const ora_jacobian *jacob);

/* pixel loops: render_nb.py:9-36, gmix_nb.py:681-937 */
int ora_render(ora_gauss2d *gm, int64_t ng, const ora_coord *coords,
               int64_t ncoords, double *image, int fast_exp);
int ora_get_loglike(ora_gauss2d *gm, int64_t ng, const ora_pixel *pixels,
                    int64_t npix, double *loglike, double *s2n_numer,
                    double *s2n_denom, int64_t *npix_out);
int ora_fill_fdiff(ora_gauss2d *gm, int64_t ng, const ora_pixel *pixels,
                   int64_t npix, double *fdiff, int64_t start);
int ora_get_model_s2n_sum(ora_gauss2d *gm, int64_t ng, const ora_pixel *pixels,
                          int64_t npix, double *s2n_sum);
/* result record layout for nmom moments (gmix.py:1314-1330, align=True):
   i4 flags, i4 npix, f8 wsum, f8 sums[n], f8 sums_cov[n*n], f8 pars[n], f8 F[n] */
int ora_get_weighted_sums(const ora_gauss2d *wt, int64_t ng,
                          const ora_pixel *pixels, int64_t npix, void *res,
                          int nmom, double maxrad);

/* admom: ngmix/admom/admom_nb.py */
int ora_admom(const ora_admom_conf *conf, ora_gauss2d *wt,
              const ora_pixel *pixels, int64_t npix, ora_admom_result *res);

/* em: ngmix/em/em_nb.py.  kind: 0 full, 1 fixcen, 2 fixcov, 3 fluxonly.
   sums is the caller's record array with the reference dtype of that kind
   (14 / 10 / 8 / 2 doubles per gaussian). */
int ora_em_run(int kind, const ora_em_conf *conf, ora_pixel *pixels,
               int64_t npix, double *sums, ora_gauss2d *gmix, int64_t ngauss,
               ora_gauss2d *gmix_psf, int64_t npsf, ora_gauss2d *gmix_conv,
               int fill_zero_weight, int32_t *numiter, double *frac_diff,
               double *sky_out);

/* derivative images: ngmix/fitting/derivs_nb.py:40-127 */
void ora_deriv_images(const double *gpars, const double *dcov, int64_t ngauss,
                      const double *vv, const double *uu, const double *area,
                      int64_t npix, double *out);

/* ---- batch helpers for bench.py's cpu_baseline leg (OpenMP over stamps).
   Same AoS inputs the reference's numba path reads. ---- */
void ora_admom_batch(const ora_admom_conf *conf, ora_gauss2d *wt_all,
                     const ora_pixel *pixels_all, int64_t npix, int64_t nstamps,
                     ora_admom_result *res_all, int nthreads);
void ora_em_batch(const ora_em_conf *conf, ora_pixel *pixels_all, int64_t npix,
                  int64_t nstamps, ora_gauss2d *gmix_all, int64_t ngauss,
                  ora_gauss2d *psf_all, int64_t npsf, ora_gauss2d *conv_all,
                  int32_t *numiter_out, int32_t *status_out, int nthreads);
void ora_loglike_batch(const ora_gauss2d *gm_all, int64_t ng,
                       const ora_pixel *pixels_all, int64_t npix, int64_t nstamps,
                       double *loglike_out, int nthreads);
int ora_num_threads(void);
void ora_render_loglike_batch(const ora_gauss2d *gm_all, int64_t ng,
                              const ora_pixel *pixels_all,
                              const ora_coord *coords_all, int64_t npix,
                              double *images_all, int64_t nstamps,
                              double *loglike_out, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
