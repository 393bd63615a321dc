/*
 * ngmix_oracle.c -- CPU restatement of the ngmix numba pixel hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see ngmix_oracle.h).  Scalar, sequential, same
 * IEEE-754 double operation order as the reference's njit sources.  Citations
 * are file:line under /root/reference.
 *
 * Where the reference (under numba's default python error model) would raise
 * ZeroDivisionError on a float division by zero, the division is guarded and
 * ORA_ERR_ZERO_DIV is returned; GMixRangeError sites return their own code.
 * Integer powers (x**2) are multiplications, as numba lowers them.
 */
#include "ngmix_oracle.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define GMIX_LOW_DETVAL 1.0e-200 /* ngmix/gmix/gmix_nb.py:11 */
#define FASTEXP_MAX_CHI2 25.0    /* ngmix/fastexp_nb.py:80 */
#define FASTEXP_APOD_CHI2 20.0   /* ngmix/fastexp_nb.py:85 */
#define APOD_IWIDTH 0.2          /* 1.0/(25.0-20.0), fastexp_nb.py:86 */

/* ngmix/flags.py:3-11 */
#define FLAG_CEN_SHIFT 2
#define FLAG_NONPOS_FLUX 4
#define FLAG_NONPOS_SIZE 8
#define FLAG_LOW_DET 16
#define FLAG_MAXITER 32

/* numpy.exp(arange(-15, 1)), ngmix/fastexp_nb.py:5-16,90-94 (identical to
   glibc exp on this platform, verified in tests/test_oracle_golden.py) */
static const double EXP_LOOKUP[16] = {
    3.059023205018258e-07,  8.315287191035679e-07,  2.2603294069810542e-06,
    6.14421235332821e-06,   1.670170079024566e-05,  4.5399929762484854e-05,
    0.00012340980408667956, 0.00033546262790251185, 0.0009118819655545162,
    0.0024787521766663585,  0.006737946999085467,   0.01831563888873418,
    0.049787068367863944,   0.1353352832366127,     0.36787944117144233,
    1.0};

/* ---------------------------------------------------------------- a1, a2 */

/* exp5_smooth, ngmix/fastexp_nb.py:223-262 */
double ora_fexp(double x)
{
    int64_t ival = (int64_t)(x - 0.5); /* int() truncates toward zero */
    double f = x - (double)ival;
    double expval = EXP_LOOKUP[ival + 15];
    expval *= 1.0000011318561302 +
              f * (0.999993601071577 +
                   f * (0.49992478810274166 +
                        f * (0.16674612720799442 +
                             f * (0.042330947141114836 +
                                  f * 0.008197933236258961))));
    return expval;
}

/* ngmix/fastexp_nb.py:97-117 */
double ora_apod_window(double chi2)
{
    double u = (FASTEXP_MAX_CHI2 - chi2) * APOD_IWIDTH;
    return u * u * u * (10.0 + u * (-15.0 + 6.0 * u));
}

/* ngmix/fastexp_nb.py:120-135 */
double ora_apod_window_deriv(double chi2)
{
    double u = (FASTEXP_MAX_CHI2 - chi2) * APOD_IWIDTH;
    double umu = u * (1.0 - u);
    return -30.0 * umu * umu * APOD_IWIDTH;
}

void ora_fexp_array(const double *x, double *out, int64_t n)
{
    for (int64_t i = 0; i < n; i++) out[i] = ora_fexp(x[i]);
}

void ora_apod_array(const double *chi2, double *w, double *dw, int64_t n)
{
    for (int64_t i = 0; i < n; i++) {
        w[i] = ora_apod_window(chi2[i]);
        dw[i] = ora_apod_window_deriv(chi2[i]);
    }
}

/* ---------------------------------------------------------------- a3, a4 */

/* gauss2d_eval_pixel_fast, ngmix/gmix/gmix_nb.py:28-63 */
static inline double gauss_eval_fast(const ora_gauss2d *g, double v, double u,
                                     double area)
{
    double model_val = 0.0;
    double vdiff = v - g->row;
    double udiff = u - g->col;
    double chi2 = g->dcc * vdiff * vdiff + g->drr * udiff * udiff -
                  2.0 * g->drc * vdiff * udiff;
    if (chi2 < FASTEXP_MAX_CHI2 && chi2 >= 0.0) {
        model_val = g->pnorm * ora_fexp(-0.5 * chi2) * area;
        if (chi2 > FASTEXP_APOD_CHI2) model_val *= ora_apod_window(chi2);
    }
    return model_val;
}

/* gauss2d_eval_pixel, ngmix/gmix/gmix_nb.py:66-92: true exp, no cut */
static inline double gauss_eval_exact(const ora_gauss2d *g, double v, double u,
                                      double area)
{
    double vdiff = v - g->row;
    double udiff = u - g->col;
    double chi2 = g->dcc * vdiff * vdiff + g->drr * udiff * udiff -
                  2.0 * g->drc * vdiff * udiff;
    return g->pnorm * exp(-0.5 * chi2) * area;
}

/* gmix_eval_pixel_fast, gmix_nb.py:14-25 */
double ora_gmix_eval_pixel_fast(const ora_gauss2d *gm, int64_t ng, double v,
                                double u, double area)
{
    double model_val = 0.0;
    for (int64_t i = 0; i < ng; i++)
        model_val += gauss_eval_fast(&gm[i], v, u, area);
    return model_val;
}

/* gmix_eval_pixel, gmix_nb.py:95-105 */
double ora_gmix_eval_pixel(const ora_gauss2d *gm, int64_t ng, double v,
                           double u, double area)
{
    double model_val = 0.0;
    for (int64_t i = 0; i < ng; i++)
        model_val += gauss_eval_exact(&gm[i], v, u, area);
    return model_val;
}

/* ------------------------------------------------------------- a13 norms */

/* gauss2d_set_norm, gmix_nb.py:190-218 */
static int gauss2d_set_norm(ora_gauss2d *g)
{
    if (g->det < GMIX_LOW_DETVAL) return ORA_ERR_DET_TOO_LOW;
    double T = g->irr + g->icc;
    if (T <= GMIX_LOW_DETVAL) return ORA_ERR_T_TOO_LOW;

    double idet = 1.0 / g->det;
    g->drr = g->irr * idet;
    g->drc = g->irc * idet;
    g->dcc = g->icc * idet;
    g->norm = 1.0 / (2 * M_PI * sqrt(g->det));
    g->pnorm = g->p * g->norm;
    g->norm_set = 1;
    return ORA_OK;
}

/* gmix_set_norms, gmix_nb.py:176-187.  Gaussians before the failing one keep
   their freshly set norms, as in the reference (partial writes persist). */
int ora_gmix_set_norms(ora_gauss2d *gm, int64_t ng)
{
    for (int64_t i = 0; i < ng; i++) {
        int st = gauss2d_set_norm(&gm[i]);
        if (st) return st;
    }
    return ORA_OK;
}

/* gauss2d_set, gmix_nb.py:221-240 */
void ora_gauss2d_set(ora_gauss2d *g, double p, double row, double col,
                     double irr, double irc, double icc)
{
    g->norm_set = 0;
    g->drr = NAN;
    g->drc = NAN;
    g->dcc = NAN;
    g->norm = NAN;
    g->pnorm = NAN;
    g->p = p;
    g->row = row;
    g->col = col;
    g->irr = irr;
    g->irc = irc;
    g->icc = icc;
    g->det = irr * icc - irc * irc;
}

/* ------------------------------------------------------------- a14 fills */

/* gmix_nb.py:243-304 */
static const double PVALS_EXP[6] = {
    0.00061601229677880041, 0.0079461395724623237, 0.053280454055540001,
    0.21797364640726541,    0.45496740582554868,   0.26521634184240478};
static const double FVALS_EXP[6] = {
    0.002467115141477932, 0.018147435573256168, 0.07944063151366336,
    0.27137669897479122,  0.79782256866993773,  2.1623306025075739};
static const double PVALS_DEV[10] = {
    6.5288960012625658e-05, 0.00044199216814302695, 0.0020859587871659754,
    0.0075913681418996841,  0.02260266219257237,    0.056532254390212859,
    0.11939049233042602,    0.20969545753234975,    0.29254151133139222,
    0.28905301416582552};
static const double FVALS_DEV[10] = {
    2.9934935706271918e-07, 3.4651596338231207e-06, 2.4807910570562753e-05,
    1.4307404300535354e-04, 7.2753169298239500e-04, 3.4582464394427260e-03,
    1.6086645440719100e-02, 7.7006776775654429e-02, 4.1012562102501476e-01,
    2.9812509778548648e00};
static const double PVALS_TURB[3] = {0.596510042804182, 0.4034898268889178,
                                     1.303069003078001e-07};
static const double FVALS_TURB[3] = {0.5793612389470884, 1.621860687127999,
                                     7.019347162356363};
static const double PVALS_GAUSS[1] = {1.0};
static const double FVALS_GAUSS[1] = {1.0};

/* g1g2_to_e1e2, gmix_nb.py:652-678 */
int ora_g1g2_to_e1e2(double g1, double g2, double *e1, double *e2)
{
    double g = sqrt(g1 * g1 + g2 * g2);
    if (g >= 1) return ORA_ERR_G_RANGE;
    if (g == 0.0) {
        *e1 = 0.0;
        *e2 = 0.0;
    } else {
        double eta = 2 * atanh(g);
        double e = tanh(eta);
        if (e >= 1.0) e = 0.99999999;
        double fac = e / g;
        *e1 = fac * g1;
        *e2 = fac * g2;
    }
    return ORA_OK;
}

/* get_cm_Tfactor, gmix_nb.py:561-593 */
int ora_get_cm_Tfactor(double fracdev, double TdByTe, double *Tfactor_out)
{
    double ifracdev = 1.0 - fracdev;
    double Tfactor = 0.0;
    for (int i = 0; i < 6; i++) {
        double p = PVALS_EXP[i] * ifracdev;
        double f = FVALS_EXP[i];
        Tfactor += p * f;
    }
    for (int i = 0; i < 10; i++) {
        double p = PVALS_DEV[i] * fracdev;
        double f = FVALS_DEV[i] * TdByTe;
        Tfactor += p * f;
    }
    if (Tfactor == 0.0) return ORA_ERR_ZERO_DIV;
    *Tfactor_out = 1.0 / Tfactor;
    return ORA_OK;
}

/* gmix_fill_simple, gmix_nb.py:307-340 */
static int fill_simple(ora_gauss2d *gm, int64_t ng, const double *pars,
                       const double *fvals, const double *pvals)
{
    double row = pars[0], col = pars[1], g1 = pars[2], g2 = pars[3];
    double T = pars[4], flux = pars[5];
    double e1, e2;
    int st = ora_g1g2_to_e1e2(g1, g2, &e1, &e2);
    if (st) return st;
    for (int64_t i = 0; i < ng; i++) {
        double T_i_2 = 0.5 * T * fvals[i];
        double flux_i = flux * pvals[i];
        ora_gauss2d_set(&gm[i], flux_i, row, col, T_i_2 * (1 - e1), T_i_2 * e2,
                        T_i_2 * (1 + e1));
    }
    return ORA_OK;
}

/* the shared 16-gaussian body of gmix_fill_cm / _bd / _bdf,
   gmix_nb.py:430-558 */
static int fill_composite(ora_gauss2d *gm, double row, double col, double g1,
                          double g2, double T, double flux, double fracdev,
                          double TdByTe)
{
    double ifracdev = 1.0 - fracdev;
    double e1, e2;
    int st = ora_g1g2_to_e1e2(g1, g2, &e1, &e2);
    if (st) return st;
    for (int i = 0; i < 16; i++) {
        double p, f;
        if (i < 6) {
            p = PVALS_EXP[i] * ifracdev;
            f = FVALS_EXP[i];
        } else {
            p = PVALS_DEV[i - 6] * fracdev;
            f = FVALS_DEV[i - 6] * TdByTe;
        }
        double T_i_2 = 0.5 * T * f;
        double flux_i = flux * p;
        ora_gauss2d_set(&gm[i], flux_i, row, col, T_i_2 * (1 - e1), T_i_2 * e2,
                        T_i_2 * (1 + e1));
    }
    return ORA_OK;
}

int ora_gmix_fill(ora_gauss2d *gm, int64_t ng, const double *pars,
                  int64_t npars, int model, double fracdev, double TdByTe,
                  double Tfactor)
{
    (void)npars;
    switch (model) {
    case 0: /* gmix_fill_full, gmix_nb.py:408-427 */
        for (int64_t i = 0; i < ng; i++) {
            const double *q = pars + 6 * i;
            ora_gauss2d_set(&gm[i], q[0], q[1], q[2], q[3], q[4], q[5]);
        }
        return ORA_OK;
    case 1:
        return fill_simple(gm, ng, pars, FVALS_GAUSS, PVALS_GAUSS);
    case 2:
        return fill_simple(gm, ng, pars, FVALS_TURB, PVALS_TURB);
    case 3:
        return fill_simple(gm, ng, pars, FVALS_EXP, PVALS_EXP);
    case 4:
        return fill_simple(gm, ng, pars, FVALS_DEV, PVALS_DEV);
    case 7: { /* gmix_fill_coellip, gmix_nb.py:375-405 */
        double e1, e2;
        int st = ora_g1g2_to_e1e2(pars[2], pars[3], &e1, &e2);
        if (st) return st;
        for (int64_t i = 0; i < ng; i++) {
            double T = pars[4 + i];
            double Thalf = 0.5 * T;
            double flux = pars[4 + ng + i];
            ora_gauss2d_set(&gm[i], flux, pars[0], pars[1], Thalf * (1 - e1),
                            Thalf * e2, Thalf * (1 + e1));
        }
        return ORA_OK;
    }
    case 9: /* gmix_fill_cm, gmix_nb.py:430-466 */
        return fill_composite(gm, pars[0], pars[1], pars[2], pars[3],
                              pars[4] * Tfactor, pars[5], fracdev, TdByTe);
    case 10: { /* gmix_fill_bd, gmix_nb.py:469-512 */
        double lTrat = pars[5];
        double fd = pars[6];
        double tdte = pow(10.0, lTrat);
        double tf;
        int st = ora_get_cm_Tfactor(fd, tdte, &tf);
        if (st) return st;
        return fill_composite(gm, pars[0], pars[1], pars[2], pars[3],
                              pars[4] * tf, pars[7], fd, tdte);
    }
    case 6: { /* gmix_fill_bdf, gmix_nb.py:515-558 */
        double fd = pars[5];
        double tf;
        int st = ora_get_cm_Tfactor(fd, 1.0, &tf);
        if (st) return st;
        return fill_composite(gm, pars[0], pars[1], pars[2], pars[3],
                              pars[4] * tf, pars[6], fd, 1.0);
    }
    default:
        return -1;
    }
}

/* ---------------------------------------------------------- a15 convolve */

/* gmix_get_cen, gmix_nb.py:108-130 */
static int gmix_get_cen(const ora_gauss2d *gm, int64_t ng, double *row_out,
                        double *col_out, double *psum_out)
{
    double row = 0.0, col = 0.0, psum = 0.0;
    for (int64_t i = 0; i < ng; i++) {
        double p = gm[i].p;
        row += p * gm[i].row;
        col += p * gm[i].col;
        psum += p;
    }
    if (psum == 0.0) return ORA_ERR_ZERO_DIV;
    row /= psum;
    col /= psum;
    *row_out = row;
    *col_out = col;
    *psum_out = psum;
    return ORA_OK;
}

/* gmix_convolve_fill, gmix_nb.py:609-649 */
int ora_gmix_convolve_fill(ora_gauss2d *out, const ora_gauss2d *gm, int64_t ng,
                           const ora_gauss2d *psf, int64_t npsf)
{
    double psf_rowcen, psf_colcen, psf_psum;
    int st = gmix_get_cen(psf, npsf, &psf_rowcen, &psf_colcen, &psf_psum);
    if (st) return st;
    double psf_ipsum = 1.0 / psf_psum;
    int64_t itot = 0;
    for (int64_t iobj = 0; iobj < ng; iobj++) {
        const ora_gauss2d *o = &gm[iobj];
        for (int64_t ipsf = 0; ipsf < npsf; ipsf++) {
            const ora_gauss2d *q = &psf[ipsf];
            double p = o->p * q->p * psf_ipsum;
            double row = o->row + (q->row - psf_rowcen);
            double col = o->col + (q->col - psf_colcen);
            double irr = o->irr + q->irr;
            double irc = o->irc + q->irc;
            double icc = o->icc + q->icc;
            ora_gauss2d_set(&out[itot], p, row, col, irr, irc, icc);
            itot++;
        }
    }
    return ORA_OK;
}

/* ------------------------------------------------ a16 pixels / jacobian */

/* jacobian_get_vu, ngmix/jacobian/jacobian_nb.py:4-16 */
void ora_jacobian_get_vu(const ora_jacobian *j, double row, double col,
                         double *v, double *u)
{
    double rowdiff = row - j->row0;
    double coldiff = col - j->col0;
    *v = j->dvdrow * rowdiff + j->dvdcol * coldiff;
    *u = j->dudrow * rowdiff + j->dudcol * coldiff;
}

/* jacobian_get_rowcol, jacobian_nb.py:19-30 */
int ora_jacobian_get_rowcol(const ora_jacobian *j, double v, double u,
                            double *row, double *col)
{
    double rowdiff = j->dudcol * v - j->dvdcol * u;
    double coldiff = -j->dudrow * v + j->dvdrow * u;
    if (j->det == 0.0) return ORA_ERR_ZERO_DIV;
    *row = j->row0 + rowdiff / j->det;
    *col = j->col0 + coldiff / j->det;
    return ORA_OK;
}

/* fill_pixels, ngmix/pixels/pixels_nb.py:6-58 */
int ora_fill_pixels(ora_pixel *pixels, int64_t npixels, const double *image,
                    const double *weight, int64_t nrow, int64_t ncol,
                    const ora_jacobian *jacob, int ignore_zero_weight)
{
    double pixel_area = jacob->scale * jacob->scale; /* jacobian_nb.py:33-40 */
    int64_t ipixel = 0;
    for (int64_t row = 0; row < nrow; row++) {
        for (int64_t col = 0; col < ncol; col++) {
            double ivar = weight[row * ncol + col];
            if (ignore_zero_weight && ivar <= 0.0) continue;
            if (ipixel >= npixels) return ORA_ERR_PIXELS_NOT_FILLED;
            ora_pixel *pixel = &pixels[ipixel];
            double v, u;
            ora_jacobian_get_vu(jacob, (double)row, (double)col, &v, &u);
            pixel->v = v;
            pixel->u = u;
            pixel->area = pixel_area;
            pixel->val = image[row * ncol + col];
            if (ivar < 0.0) ivar = 0.0;
            pixel->ierr = sqrt(ivar);
            ipixel++;
        }
    }
    if (ipixel != npixels) return ORA_ERR_PIXELS_NOT_FILLED;
    return ORA_OK;
}

/* fill_coords, pixels_nb.py:61-94 */
void ora_fill_coords(ora_coord *coords, int64_t nrow, int64_t ncol,
                     const ora_jacobian *jacob)
{
    double pixel_area = jacob->scale * jacob->scale;
    int64_t icoord = 0;
    for (int64_t row = 0; row < nrow; row++) {
        for (int64_t col = 0; col < ncol; col++) {
            double v, u;
            ora_jacobian_get_vu(jacob, (double)row, (double)col, &v, &u);
            coords[icoord].v = v;
            coords[icoord].u = u;
            coords[icoord].area = pixel_area;
            icoord++;
        }
    }
}

/* ---------------------------------------------------- a5-a8 pixel loops */

static inline int norms_if_needed(ora_gauss2d *gm, int64_t ng)
{
    if (gm[0].norm_set == 0) return ora_gmix_set_norms(gm, ng);
    return ORA_OK;
}

/* render, ngmix/gmix/render_nb.py:9-36 (adds into image) */
int ora_render(ora_gauss2d *gm, int64_t ng, const ora_coord *coords,
               int64_t ncoords, double *image, int fast_exp)
{
    int st = norms_if_needed(gm, ng);
    if (st) return st;
    if (fast_exp) {
        for (int64_t i = 0; i < ncoords; i++)
            image[i] += ora_gmix_eval_pixel_fast(gm, ng, coords[i].v,
                                                 coords[i].u, coords[i].area);
    } else {
        for (int64_t i = 0; i < ncoords; i++)
            image[i] += ora_gmix_eval_pixel(gm, ng, coords[i].v, coords[i].u,
                                            coords[i].area);
    }
    return ORA_OK;
}

/* get_loglike, gmix_nb.py:824-874 */
int ora_get_loglike(ora_gauss2d *gm, int64_t ng, const ora_pixel *pixels,
                    int64_t n_pixels, double *loglike_out, double *s2n_numer_out,
                    double *s2n_denom_out, int64_t *npix_out)
{
    int st = norms_if_needed(gm, ng);
    if (st) return st;
    int64_t npix = 0;
    double loglike = 0.0, s2n_numer = 0.0, s2n_denom = 0.0;
    for (int64_t ip = 0; ip < n_pixels; ip++) {
        const ora_pixel *pixel = &pixels[ip];
        double model_val =
            ora_gmix_eval_pixel_fast(gm, ng, pixel->v, pixel->u, pixel->area);
        double ivar = pixel->ierr * pixel->ierr;
        double val = pixel->val;
        double diff = model_val - val;
        loglike += diff * diff * ivar;
        s2n_numer += val * model_val * ivar;
        s2n_denom += model_val * model_val * ivar;
        npix += 1;
    }
    loglike *= -0.5;
    *loglike_out = loglike;
    *s2n_numer_out = s2n_numer;
    *s2n_denom_out = s2n_denom;
    *npix_out = npix;
    return ORA_OK;
}

/* fill_fdiff, gmix_nb.py:877-900 */
int ora_fill_fdiff(ora_gauss2d *gm, int64_t ng, const ora_pixel *pixels,
                   int64_t n_pixels, double *fdiff, int64_t start)
{
    int st = norms_if_needed(gm, ng);
    if (st) return st;
    for (int64_t ip = 0; ip < n_pixels; ip++) {
        const ora_pixel *pixel = &pixels[ip];
        double model_val =
            ora_gmix_eval_pixel_fast(gm, ng, pixel->v, pixel->u, pixel->area);
        fdiff[start + ip] = (model_val - pixel->val) * pixel->ierr;
    }
    return ORA_OK;
}

/* get_model_s2n_sum, gmix_nb.py:903-937 */
int ora_get_model_s2n_sum(ora_gauss2d *gm, int64_t ng, const ora_pixel *pixels,
                          int64_t n_pixels, double *s2n_sum_out)
{
    int st = norms_if_needed(gm, ng);
    if (st) return st;
    double s2n_sum = 0.0;
    for (int64_t ip = 0; ip < n_pixels; ip++) {
        const ora_pixel *pixel = &pixels[ip];
        double model_val =
            ora_gmix_eval_pixel_fast(gm, ng, pixel->v, pixel->u, pixel->area);
        double ivar = pixel->ierr * pixel->ierr;
        s2n_sum += model_val * model_val * ivar;
    }
    *s2n_sum_out = s2n_sum;
    return ORA_OK;
}

/* -------------------------------------------------- a10, a10b moments */

/* get_weighted_sums (nmom=6), gmix_nb.py:681-734, and
   get_higher_order_weighted_sums (nmom=17), gmix_nb.py:737-821.
   The caller has set the norms (gmix.py:733). */
int ora_get_weighted_sums(const ora_gauss2d *wt, int64_t ng,
                          const ora_pixel *pixels, int64_t n_pixels, void *resv,
                          int nmom, double maxrad)
{
    char *base = (char *)resv;
    int32_t *npix = (int32_t *)(base + 4);
    double *wsum = (double *)(base + 8);
    double *sums = (double *)(base + 16);
    double *sums_cov = sums + nmom;
    double *F = sums_cov + nmom * nmom + nmom;

    double maxrad2 = maxrad * maxrad;
    double vcen = wt[0].row;
    double ucen = wt[0].col;

    for (int64_t ip = 0; ip < n_pixels; ip++) {
        const ora_pixel *pixel = &pixels[ip];
        if (nmom == 6) {
            double vmod = pixel->v - vcen;
            double umod = pixel->u - ucen;
            double rad2 = umod * umod + vmod * vmod;
            if (!(rad2 < maxrad2 && pixel->ierr > 0.0)) continue;
            double weight =
                ora_gmix_eval_pixel(wt, ng, pixel->v, pixel->u, pixel->area);
            double var = 1.0 / (pixel->ierr * pixel->ierr);
            double wdata = weight * pixel->val;
            double w2 = weight * weight;
            F[0] = pixel->v;
            F[1] = pixel->u;
            F[2] = umod * umod - vmod * vmod;
            F[3] = 2 * vmod * umod;
            F[4] = rad2;
            F[5] = 1.0;
            *wsum += weight;
            *npix += 1;
            for (int i = 0; i < 6; i++) {
                sums[i] += wdata * F[i];
                for (int j = 0; j < 6; j++)
                    sums_cov[i * 6 + j] += w2 * var * F[i] * F[j];
            }
        } else {
            double v = pixel->v - vcen;
            double u = pixel->u - ucen;
            double r2 = u * u + v * v;
            if (!(r2 < maxrad2)) continue;
            double weight =
                ora_gmix_eval_pixel(wt, ng, pixel->v, pixel->u, pixel->area);
            double ierr2 = pixel->ierr * pixel->ierr;
            if (ierr2 == 0.0) return ORA_ERR_ZERO_DIV;
            double var = 1.0 / ierr2;
            double wdata = weight * pixel->val;
            double w2 = weight * weight;
            double u2 = u * u, v2 = v * v, vu = v * u;
            double u4 = u2 * u2, v4 = v2 * v2;
            double r4 = r2 * r2, r6 = r4 * r2, r8 = r6 * r2;
            F[0] = pixel->v;
            F[1] = pixel->u;
            F[2] = u2 - v2;
            F[3] = 2 * vu;
            F[4] = r2;
            F[5] = 1.0;
            F[6] = u * r2;
            F[7] = v * r2;
            F[8] = u * (u2 - 3 * v2);
            F[9] = v * (3 * u2 - v2);
            F[10] = r4;
            F[11] = r2 * (u2 - v2);
            F[12] = r2 * 2 * u * v;
            F[13] = u4 - 6 * u2 * v2 + v4;
            F[14] = (u2 - v2) * 4 * u * v;
            F[15] = r6;
            F[16] = r8;
            *wsum += weight;
            *npix += 1;
            for (int i = 0; i < nmom; i++) {
                sums[i] += wdata * F[i];
                for (int j = 0; j < nmom; j++)
                    sums_cov[i * nmom + j] += w2 * var * F[i] * F[j];
            }
        }
    }
    return ORA_OK;
}

/* ------------------------------------------------------------- a9 admom */

/* clear_result, admom_nb.py:229-239 */
static void admom_clear_result(ora_admom_result *res)
{
    res->npix = 0;
    res->wsum = 0.0;
    for (int i = 0; i < 7; i++) res->sums[i] = 0.0;
    for (int i = 0; i < 49; i++) res->sums_cov[i] = 0.0;
    for (int i = 0; i < 6; i++) res->pars[i] = NAN;
    res->rho4 = NAN;
}

/* admom_censums, admom_nb.py:111-128 */
static void admom_censums(const ora_gauss2d *wt, const ora_pixel *pixels,
                          int64_t n_pixels, ora_admom_result *res)
{
    for (int64_t i = 0; i < n_pixels; i++) {
        const ora_pixel *pixel = &pixels[i];
        double weight =
            ora_gmix_eval_pixel_fast(wt, 1, pixel->v, pixel->u, pixel->area);
        double wdata = weight * pixel->val;
        res->npix += 1;
        res->sums[0] += wdata * pixel->v;
        res->sums[1] += wdata * pixel->u;
        res->sums[5] += wdata;
    }
}

/* admom_momsums, admom_nb.py:131-175 */
static int admom_momsums(const ora_gauss2d *wt, const ora_pixel *pixels,
                         int64_t n_pixels, ora_admom_result *res)
{
    double vcen = wt[0].row;
    double ucen = wt[0].col;
    double *F = res->F;
    for (int64_t ip = 0; ip < n_pixels; ip++) {
        const ora_pixel *pixel = &pixels[ip];
        double weight =
            ora_gmix_eval_pixel_fast(wt, 1, pixel->v, pixel->u, pixel->area);
        double ierr2 = pixel->ierr * pixel->ierr;
        if (ierr2 == 0.0) return ORA_ERR_ZERO_DIV;
        double var = 1.0 / ierr2;
        double vmod = pixel->v - vcen;
        double umod = pixel->u - ucen;
        double wdata = weight * pixel->val;
        double w2 = weight * weight;
        double chi2 = wt[0].dcc * vmod * vmod + wt[0].drr * umod * umod -
                      2.0 * wt[0].drc * vmod * umod;
        F[0] = pixel->v;
        F[1] = pixel->u;
        F[2] = umod * umod - vmod * vmod;
        F[3] = 2 * vmod * umod;
        F[4] = umod * umod + vmod * vmod;
        F[5] = 1.0;
        F[6] = chi2 * chi2;
        res->wsum += weight;
        res->npix += 1;
        for (int i = 0; i < 7; i++) {
            res->sums[i] += wdata * F[i];
            for (int j = 0; j < 7; j++)
                res->sums_cov[i * 7 + j] += w2 * var * F[i] * F[j];
        }
    }
    return ORA_OK;
}

/* deweight_moments, admom_nb.py:178-226 */
static void admom_deweight(ora_gauss2d *wt, double Irr, double Irc, double Icc,
                           ora_admom_result *res)
{
    double detm = Irr * Icc - Irc * Irc;
    if (detm <= GMIX_LOW_DETVAL) {
        res->flags = FLAG_LOW_DET;
        return;
    }
    double Wrr = wt[0].irr, Wrc = wt[0].irc, Wcc = wt[0].icc;
    double detw = Wrr * Wcc - Wrc * Wrc;
    if (detw <= GMIX_LOW_DETVAL) {
        res->flags = FLAG_LOW_DET;
        return;
    }
    double idetw = 1.0 / detw;
    double idetm = 1.0 / detm;
    double Nrr = Icc * idetm - Wcc * idetw;
    double Ncc = Irr * idetm - Wrr * idetw;
    double Nrc = -Irc * idetm + Wrc * idetw;
    double detn = Nrr * Ncc - Nrc * Nrc;
    if (detn <= GMIX_LOW_DETVAL) {
        res->flags = FLAG_LOW_DET;
        return;
    }
    double idetn = 1. / detn;
    wt[0].irr = Ncc * idetn;
    wt[0].icc = Nrr * idetn;
    wt[0].irc = -Nrc * idetn;
    wt[0].det = wt[0].irr * wt[0].icc - wt[0].irc * wt[0].irc;
}

/* admom, admom_nb.py:13-108 */
int ora_admom(const ora_admom_conf *conf, ora_gauss2d *wt,
              const ora_pixel *pixels, int64_t n_pixels, ora_admom_result *res)
{
    double roworig = wt[0].row;
    double colorig = wt[0].col;
    double e1old = NAN, e2old = NAN, Told = NAN;
    int32_t i = -1; /* maxiter<=0 leaves numba's loop variable undefined;
                       we define numiter=0 (=> MAXITER when maxiter==0) */
    for (int32_t it = 0; it < conf->maxiter; it++) {
        i = it;
        if (wt[0].det < GMIX_LOW_DETVAL) {
            res->flags = FLAG_LOW_DET;
            break;
        }
        int st = ora_gmix_set_norms(wt, 1);
        if (st) return st; /* GMixRangeError("T too low") can still fire */

        admom_clear_result(res);
        admom_censums(wt, pixels, n_pixels, res);

        if (res->sums[5] <= 0.0) {
            res->flags = FLAG_NONPOS_FLUX;
            break;
        }
        wt[0].row = res->sums[0] / res->sums[5];
        wt[0].col = res->sums[1] / res->sums[5];

        if (fabs(wt[0].row - roworig) > conf->shiftmax ||
            fabs(wt[0].col - colorig) > conf->shiftmax) {
            res->flags = FLAG_CEN_SHIFT;
            break;
        }

        admom_clear_result(res);
        st = admom_momsums(wt, pixels, n_pixels, res);
        if (st) return st;

        if (res->sums[5] <= 0.0) {
            res->flags = FLAG_NONPOS_FLUX;
            break;
        }

        double finv = 1.0 / res->sums[5];
        double M1 = res->sums[2] * finv;
        double M2 = res->sums[3] * finv;
        double T = res->sums[4] * finv;
        double Irr = 0.5 * (T - M1);
        double Icc = 0.5 * (T + M1);
        double Irc = 0.5 * M2;

        if (T <= 0.0) {
            res->flags = FLAG_NONPOS_SIZE;
            break;
        }
        double e1 = (Icc - Irr) / T;
        double e2 = 2 * Irc / T;

        /* Told is NaN on the first pass, and never 0 afterwards */
        if ((fabs(e1 - e1old) < conf->etol) && (fabs(e2 - e2old) < conf->etol) &&
            (fabs(T / Told - 1.) < conf->Ttol)) {
            res->pars[0] = wt[0].row;
            res->pars[1] = wt[0].col;
            res->pars[2] = wt[0].icc - wt[0].irr;
            res->pars[3] = 2.0 * wt[0].irc;
            res->pars[4] = wt[0].icc + wt[0].irr;
            res->pars[5] = 1.0;
            res->rho4 = res->sums[6] / res->sums[5];
            break;
        } else {
            if (!conf->cenonly) {
                admom_deweight(wt, Irr, Irc, Icc, res);
                if (res->flags != 0) break;
            }
            e1old = e1;
            e2old = e2;
            Told = T;
        }
    }
    res->numiter = i + 1;
    if (res->numiter == conf->maxiter) res->flags = FLAG_MAXITER;
    return ORA_OK;
}

/* --------------------------------------------------------------- a11 em */

/* offsets (in doubles) into the reference's per-gaussian sums record for
   each kind; -1 = field absent (ngmix/em/em.py:451-521) */
typedef struct {
    int stride;
    int gi, tvsum, tusum, tu2sum, tuvsum, tv2sum;
    int pnew, vsum, usum, u2sum, uvsum, v2sum;
} em_layout;

static const em_layout EM_LAYOUTS[4] = {
    /* full: gi,logtau,logdet,tvsum,tusum,tu2sum,tuvsum,tv2sum,pnew,vsum,usum,
       u2sum,uvsum,v2sum */
    {14, 0, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13},
    /* fixcen: gi,logtau,logdet,tu2sum,tuvsum,tv2sum,pnew,u2sum,uvsum,v2sum */
    {10, 0, -1, -1, 3, 4, 5, 6, -1, -1, 7, 8, 9},
    /* fixcov: gi,logtau,logdet,tvsum,tusum,pnew,vsum,usum */
    {8, 0, 3, 4, -1, -1, -1, 5, 6, 7, -1, -1, -1},
    /* fluxonly: gi,pnew */
    {2, 0, -1, -1, -1, -1, -1, 1, -1, -1, -1, -1, -1},
};

/* gmix_get_moms, em_nb.py:1260-1294 */
static int gmix_get_moms(const ora_gauss2d *gm, int64_t ng, double *irr_o,
                         double *irc_o, double *icc_o)
{
    double row, col, psum;
    int st = gmix_get_cen(gm, ng, &row, &col, &psum);
    if (st) return st;
    double irr = 0.0, irc = 0.0, icc = 0.0;
    for (int64_t i = 0; i < ng; i++) {
        double rowdiff = gm[i].row - row;
        double coldiff = gm[i].col - col;
        double p = gm[i].p;
        irr += p * (gm[i].irr + rowdiff * rowdiff);
        irc += p * (gm[i].irc + rowdiff * coldiff);
        icc += p * (gm[i].icc + coldiff * coldiff);
    }
    irr /= psum;
    irc /= psum;
    icc /= psum;
    *irr_o = irr;
    *irc_o = irc;
    *icc_o = icc;
    return ORA_OK;
}

/* fill_zero_weight_pixels, em_nb.py:1297-1315 */
static void em_fill_zero_weight(const ora_gauss2d *gm, int64_t ng,
                                ora_pixel *pixels, int64_t npix, double sky)
{
    for (int64_t i = 0; i < npix; i++) {
        if (pixels[i].ierr <= 0.0) {
            double val = ora_gmix_eval_pixel_fast(gm, ng, pixels[i].v,
                                                  pixels[i].u, pixels[i].area);
            pixels[i].val = sky + val;
        }
    }
}

/* M-step: gmix_set_from_sums{,_fixcen,_fixcov,_fluxonly},
   em_nb.py:284-354, 587-655, 954-1000, 1200-1241 */
static int em_set_from_sums(int kind, ora_gauss2d *gmix, int64_t ngauss,
                            const ora_gauss2d *gmix_psf, int64_t npsf,
                            ora_gauss2d *gmix_conv, const double *sums)
{
    const em_layout *L = &EM_LAYOUTS[kind];
    const double minval = 1.0e-4;
    double psf_irr = 0, psf_irc = 0, psf_icc = 0;
    if (kind == 0 || kind == 1) {
        int st = gmix_get_moms(gmix_psf, npsf, &psf_irr, &psf_irc, &psf_icc);
        if (st) return st;
    }
    for (int64_t i = 0; i < ngauss; i++) {
        const double *ts = sums + i * L->stride;
        ora_gauss2d *gauss = &gmix[i];
        double p = ts[L->pnew];
        if (kind == 3) {
            ora_gauss2d_set(gauss, p, gauss->row, gauss->col, gauss->irr,
                            gauss->irc, gauss->icc);
            continue;
        }
        if (p == 0.0) return ORA_ERR_ZERO_DIV;
        double pinv = 1.0 / p;
        if (kind == 2) {
            double v = ts[L->vsum] * pinv;
            double u = ts[L->usum] * pinv;
            ora_gauss2d_set(gauss, p, v, u, gauss->irr, gauss->irc, gauss->icc);
            continue;
        }
        double v = gauss->row, u = gauss->col;
        if (kind == 0) {
            v = ts[L->vsum] * pinv;
            u = ts[L->usum] * pinv;
        }
        double irr = ts[L->v2sum] * pinv;
        double irc = ts[L->uvsum] * pinv;
        double icc = ts[L->u2sum] * pinv;
        irr = irr - psf_irr;
        irc = irc - psf_irc;
        icc = icc - psf_icc;
        if (irr < 0.0 || icc < 0.0) {
            irr = minval;
            irc = 0.0;
            icc = minval;
        }
        double det = irr * icc - irc * irc;
        if (det < GMIX_LOW_DETVAL) {
            double T = irr + icc;
            irr = icc = T / 2;
            irc = 0.0;
        }
        ora_gauss2d_set(gauss, p, v, u, irr, irc, icc);
    }
    int st = ora_gmix_convolve_fill(gmix_conv, gmix, ngauss, gmix_psf, npsf);
    if (st) return st;
    return ora_gmix_set_norms(gmix_conv, ngauss * npsf);
}

/* em_run{,_fixcen,_fixcov,_fluxonly}: em_nb.py:15-127, 357-469, 702-816,
   1005-1106, with do_scratch_sums* (:160-246, 472-553, 841-920, 1109-1170)
   and do_sums* (:249-281, 556-584, 923-951, 1173-1197) inlined */
int ora_em_run(int kind, const ora_em_conf *conf, ora_pixel *pixels,
               int64_t npix, double *sums, ora_gauss2d *gmix, int64_t ngauss,
               ora_gauss2d *gmix_psf, int64_t npsf, ora_gauss2d *gmix_conv,
               int fill_zero_weight, int32_t *numiter_out,
               double *frac_diff_out, double *sky_out)
{
    const em_layout *L = &EM_LAYOUTS[kind];
    int64_t nconv = ngauss * npsf;
    int st = ora_gmix_set_norms(gmix_conv, nconv);
    if (st) return st;

    double logtau[nconv > 0 ? nconv : 1], logdet[nconv > 0 ? nconv : 1];
    double tol = conf->tol;
    double sky = conf->sky;
    double elogL_last = -9999.9e9;
    double p_last = 0.0;
    int32_t numiter = 0;
    double frac_diff = 0.0; /* unbound in the reference if never assigned */

    if (kind == 3)
        for (int64_t i = 0; i < ngauss; i++) p_last += gmix[i].p;

    for (int32_t it = 0; it < conf->maxiter; it++) {
        double elogL = 0.0;
        double skysum = 0.0;

        /* clear_sums*: every field except logtau/logdet */
        for (int64_t i = 0; i < ngauss; i++) {
            double *ts = sums + i * L->stride;
            for (int k = 0; k < L->stride; k++)
                if (kind == 3 || (k != 1 && k != 2)) ts[k] = 0.0;
        }
        if (kind != 3) {
            /* set_logtau_logdet, em_nb.py:658-675 */
            for (int64_t i = 0; i < nconv; i++) {
                logtau[i] = log(gmix_conv[i].p);
                logdet[i] = log(gmix_conv[i].det);
            }
        }
        if (fill_zero_weight)
            em_fill_zero_weight(gmix_conv, nconv, pixels, npix, sky);

        for (int64_t ip = 0; ip < npix; ip++) {
            const ora_pixel *pixel = &pixels[ip];
            double v = pixel->v, u = pixel->u;
            double gsum = 0.0, logL = 0.0;

            for (int64_t ii = 0; ii < ngauss; ii++) {
                double *ts = sums + ii * L->stride;
                ts[L->gi] = 0.0;
                if (L->tvsum >= 0) ts[L->tvsum] = 0.0, ts[L->tusum] = 0.0;
                if (L->tv2sum >= 0)
                    ts[L->tv2sum] = 0.0, ts[L->tuvsum] = 0.0,
                    ts[L->tu2sum] = 0.0;
                for (int64_t i = ii * npsf; i < (ii + 1) * npsf; i++) {
                    const ora_gauss2d *gauss = &gmix_conv[i];
                    double vdiff = v - gauss->row;
                    double udiff = u - gauss->col;
                    double u2 = udiff * udiff;
                    double v2 = vdiff * vdiff;
                    double uv = udiff * vdiff;
                    double chi2 = gauss->dcc * v2 + gauss->drr * u2 -
                                  2.0 * gauss->drc * uv;
                    double val;
                    if (chi2 < 25.0 && chi2 >= 0.0)
                        val = gauss->pnorm * ora_fexp(-0.5 * chi2) * pixel->area;
                    else
                        val = 0.0;
                    ts[L->gi] += val;
                    gsum += val;
                    if (L->tvsum >= 0) {
                        ts[L->tvsum] += v * val;
                        ts[L->tusum] += u * val;
                    }
                    if (L->tv2sum >= 0) {
                        ts[L->tv2sum] += v2 * val;
                        ts[L->tuvsum] += uv * val;
                        ts[L->tu2sum] += u2 * val;
                    }
                    if (kind != 3)
                        logL += val * (logtau[i] - 0.5 * logdet[i] - 0.5 * chi2);
                }
            }
            if (kind != 3) {
                if (gsum == 0.0)
                    logL = 0.0;
                else
                    logL *= 1.0 / gsum;
            }

            double gtot = gsum + sky;
            if (gtot == 0.0) return ORA_ERR_GTOT_ZERO;
            elogL += logL;
            skysum += sky * pixel->val / gtot;

            /* do_sums* */
            double factor = pixel->val / gtot;
            for (int64_t i = 0; i < ngauss; i++) {
                double *ts = sums + i * L->stride;
                double wtau = ts[L->gi] * factor;
                ts[L->pnew] += wtau;
                if (L->usum >= 0) {
                    ts[L->usum] += ts[L->tusum] * factor;
                    ts[L->vsum] += ts[L->tvsum] * factor;
                }
                if (L->u2sum >= 0) {
                    ts[L->u2sum] += ts[L->tu2sum] * factor;
                    ts[L->uvsum] += ts[L->tuvsum] * factor;
                    ts[L->v2sum] += ts[L->tv2sum] * factor;
                }
            }
        }

        st = em_set_from_sums(kind, gmix, ngauss, gmix_psf, npsf, gmix_conv,
                              sums);
        if (st) return st;

        if (conf->vary_sky) sky = skysum / (double)npix;

        numiter = it + 1;
        if (kind == 3) {
            double psum = 0.0;
            for (int64_t i = 0; i < ngauss; i++) psum += gmix[i].p;
            if (numiter >= conf->miniter) {
                if (p_last == 0.0) return ORA_ERR_ZERO_DIV;
                frac_diff = fabs(psum / p_last - 1);
                if (frac_diff < tol) break;
            }
            p_last = psum;
        } else {
            if (numiter >= conf->miniter) {
                if (elogL == 0.0) return ORA_ERR_ELOGL_ZERO;
                frac_diff = fabs((elogL - elogL_last) / elogL);
                if (frac_diff < tol) break;
            }
            elogL_last = elogL;
        }
    }

    for (int64_t i = 0; i < ngauss; i++) gmix[i].norm_set = 0;
    *numiter_out = numiter;
    *frac_diff_out = frac_diff;
    *sky_out = sky;
    return ORA_OK;
}

/* ------------------------------------------------------ a12 deriv_images */

/* deriv_images, ngmix/fitting/derivs_nb.py:40-127; out is (6, npix) */
void ora_deriv_images(const double *gpars, const double *dcov, int64_t ngauss,
                      const double *vv, const double *uu, const double *area,
                      int64_t npix, double *out)
{
    const double TWO_PI = 2.0 * M_PI;
    double trs[3];
    for (int64_t ig = 0; ig < ngauss; ig++) {
        const double *gp = gpars + 6 * ig;
        const double *dc = dcov + 9 * ig;
        double p = gp[0], vcen = gp[1], ucen = gp[2];
        double irr = gp[3], irc = gp[4], icc = gp[5];
        double det = irr * icc - irc * irc;
        if (det <= 0.0) continue;
        double norm = p / (TWO_PI * sqrt(det));
        double w11 = icc / det;
        double w12 = -irc / det;
        double w22 = irr / det;
        for (int a = 0; a < 3; a++)
            trs[a] = w11 * dc[a * 3 + 0] + 2.0 * w12 * dc[a * 3 + 1] +
                     w22 * dc[a * 3 + 2];
        for (int64_t ipix = 0; ipix < npix; ipix++) {
            double dv = vv[ipix] - vcen;
            double du = uu[ipix] - ucen;
            double qv = w11 * dv + w12 * du;
            double qu = w12 * dv + w22 * du;
            double chi2 = dv * qv + du * qu;
            if (chi2 >= FASTEXP_MAX_CHI2 || chi2 < 0.0) continue;
            double val = norm * ora_fexp(-0.5 * chi2) * area[ipix];
            double valc;
            if (chi2 > FASTEXP_APOD_CHI2) {
                double w = ora_apod_window(chi2);
                valc = val * (w - 2.0 * ora_apod_window_deriv(chi2));
                val *= w;
            } else {
                valc = val;
            }
            out[0 * npix + ipix] += val;
            out[1 * npix + ipix] += valc * qv;
            out[2 * npix + ipix] += valc * qu;
            for (int a = 0; a < 3; a++) {
                double quad = qv * qv * dc[a * 3 + 0] +
                              2.0 * qv * qu * dc[a * 3 + 1] +
                              qu * qu * dc[a * 3 + 2];
                out[(3 + a) * npix + ipix] += 0.5 * (valc * quad - val * trs[a]);
            }
        }
    }
}

/* ------------------------------------------- cpu_baseline batch drivers */

int ora_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* one render (render_nb.py) + one get_loglike (gmix_nb.py) per stamp, over
   the reference's own AoS coords / pixels arrays; stamps in parallel */
void ora_render_loglike_batch(const ora_gauss2d *gm_all, int64_t ng,
                              const ora_pixel *pixels_all,
                              const ora_coord *coords_all, int64_t npix,
                              double *images_all, int64_t nstamps,
                              double *loglike_out, int nthreads)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
    for (int64_t s = 0; s < nstamps; s++) {
        ora_gauss2d gm[64];
        int64_t n = ng < 64 ? ng : 64;
        memcpy(gm, gm_all + s * ng, n * sizeof(ora_gauss2d));
        double ll, sn, sd;
        int64_t np;
        ora_render(gm, n, coords_all + s * npix, npix, images_all + s * npix, 1);
        ora_get_loglike(gm, n, pixels_all + s * npix, npix, &ll, &sn, &sd, &np);
        loglike_out[s] = ll;
    }
    (void)nthreads;
}

/* config-5 cpu_baseline leg: get_loglike (gmix_nb.py:824-874) of every epoch
   stamp of the batch, stamps in parallel; the object sum is the caller's */
void ora_loglike_batch(const ora_gauss2d *gm_all, int64_t ng,
                       const ora_pixel *pixels_all, int64_t npix, int64_t nstamps,
                       double *loglike_out, int nthreads)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
    for (int64_t s = 0; s < nstamps; s++) {
        ora_gauss2d gm[64];
        int64_t n = ng < 64 ? ng : 64;
        memcpy(gm, gm_all + s * ng, n * sizeof(ora_gauss2d));
        double ll, sn, sd;
        int64_t np;
        ora_get_loglike(gm, n, pixels_all + s * npix, npix, &ll, &sn, &sd, &np);
        loglike_out[s] = ll;
    }
    (void)nthreads;
}

/* config-4 cpu_baseline legs: one admom / one 1-gaussian em_run per stamp over
   the reference's AoS pixel arrays, stamps in parallel (bench.py --cpu-baselines) */
void ora_admom_batch(const ora_admom_conf *conf, ora_gauss2d *wt_all,
                     const ora_pixel *pixels_all, int64_t npix, int64_t nstamps,
                     ora_admom_result *res_all, int nthreads)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
    for (int64_t s = 0; s < nstamps; s++)
        ora_admom(conf, wt_all + s, pixels_all + s * npix, npix, res_all + s);
    (void)nthreads;
}

void ora_em_batch(const ora_em_conf *conf, ora_pixel *pixels_all, int64_t npix,
                  int64_t nstamps, ora_gauss2d *gmix_all, int64_t ngauss,
                  ora_gauss2d *psf_all, int64_t npsf, ora_gauss2d *conv_all,
                  int32_t *numiter_out, int32_t *status_out, int nthreads)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
    for (int64_t s = 0; s < nstamps; s++) {
        double sums[14 * 8];
        double frac, sky;
        memset(sums, 0, sizeof(sums));
        status_out[s] = ora_em_run(0, conf, pixels_all + s * npix, npix, sums,
                                   gmix_all + s * ngauss, ngauss < 8 ? ngauss : 8,
                                   psf_all + s * npsf, npsf,
                                   conv_all + s * ngauss * npsf, 0,
                                   numiter_out + s, &frac, &sky);
    }
    (void)nthreads;
}
