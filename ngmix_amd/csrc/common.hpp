// common.hpp -- shared host/device arithmetic for libngmix_hip.so (gfx950).
//
// Everything here is float64, built with -ffp-contract=off: each operation
// rounds once, in the order the reference's numba source performs it, so
// per-pixel values are bit-identical to the reference.  Citations are
// file:line in the reference checkout (ngmix/...).
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/ngmix_hip.h"

#define NGMIX_HD __host__ __device__ __forceinline__

static_assert(sizeof(ngmix_gauss2d) == 104, "gauss2d layout");
static_assert(sizeof(ngmix_pixel) == 48, "pixel layout");
static_assert(sizeof(ngmix_coord) == 24, "coord layout");
static_assert(sizeof(ngmix_jacobian) == 64, "jacobian layout");
static_assert(sizeof(ngmix_admom_conf) == 40, "admom conf layout");
static_assert(sizeof(ngmix_admom_result) == 584, "admom result layout");
static_assert(sizeof(ngmix_em_conf) == 32, "em conf layout");
static_assert(sizeof(ngmix_stamp) == 32, "stamp layout");

namespace ngmix {

constexpr double LOW_DETVAL = 1.0e-200;  // gmix_nb.py:11
constexpr double MAX_CHI2 = 25.0;        // fastexp_nb.py:80
constexpr double APOD_CHI2 = 20.0;       // fastexp_nb.py:85
constexpr double APOD_IWIDTH = 0.2;      // 1/(25-20), fastexp_nb.py:86

// exp(i), i = -15..0 (fastexp_nb.py:5-16,90-94).  Device kernels stage this
// table in LDS; host code reads it directly.
#define NGMIX_EXP_TABLE                                                        \
    {3.059023205018258e-07,  8.315287191035679e-07,  2.2603294069810542e-06,  \
     6.14421235332821e-06,   1.670170079024566e-05,  4.5399929762484854e-05,  \
     0.00012340980408667956, 0.00033546262790251185, 0.0009118819655545162,   \
     0.0024787521766663585,  0.006737946999085467,   0.01831563888873418,     \
     0.049787068367863944,   0.1353352832366127,     0.36787944117144233, 1.0}

// exp5_smooth (fastexp_nb.py:223-262); tab points at the 16-entry table
NGMIX_HD double fexp(double x, const double *tab)
{
    int ival = (int)(x - 0.5);  // truncation toward zero, as python int()
    double f = x - (double)ival;
    double expval = tab[ival + 15];
    expval *= 1.0000011318561302 +
              f * (0.999993601071577 +
                   f * (0.49992478810274166 +
                        f * (0.16674612720799442 +
                             f * (0.042330947141114836 +
                                  f * 0.008197933236258961))));
    return expval;
}

// apod_window / apod_window_deriv (fastexp_nb.py:97-135)
NGMIX_HD double apod_window(double chi2)
{
    double u = (MAX_CHI2 - chi2) * APOD_IWIDTH;
    return u * u * u * (10.0 + u * (-15.0 + 6.0 * u));
}

NGMIX_HD double apod_window_deriv(double chi2)
{
    double u = (MAX_CHI2 - chi2) * APOD_IWIDTH;
    double umu = u * (1.0 - u);
    return -30.0 * umu * umu * APOD_IWIDTH;
}

// The six numbers a pixel evaluation needs from a gauss2d record.
// drc2 = 2*drc is exact, and (2.0*drc)*vdiff is the reference's own
// association (gmix_nb.py:55).
struct EvalGauss {
    double row, col, dcc, drr, drc2, pnorm;
};

NGMIX_HD EvalGauss make_eval(const ngmix_gauss2d &g)
{
    EvalGauss e;
    e.row = g.row;
    e.col = g.col;
    e.dcc = g.dcc;
    e.drr = g.drr;
    e.drc2 = 2.0 * g.drc;
    e.pnorm = g.pnorm;
    return e;
}

NGMIX_HD double gauss_chi2(const EvalGauss &g, double v, double u)
{
    double vdiff = v - g.row;
    double udiff = u - g.col;
    return g.dcc * vdiff * vdiff + g.drr * udiff * udiff - g.drc2 * vdiff * udiff;
}

// gauss2d_eval_pixel_fast (gmix_nb.py:28-63): apodized, NaN chi2 -> 0
NGMIX_HD double gauss_eval_fast(const EvalGauss &g, double v, double u,
                                double area, const double *tab)
{
    double model_val = 0.0;
    double chi2 = gauss_chi2(g, v, u);
    if (chi2 < MAX_CHI2 && chi2 >= 0.0) {
        model_val = g.pnorm * fexp(-0.5 * chi2, tab) * area;
        if (chi2 > APOD_CHI2) model_val *= apod_window(chi2);
    }
    return model_val;
}

// the hard-cut evaluation EM inlines (em_nb.py:222-227)
NGMIX_HD double gauss_eval_hardcut(const EvalGauss &g, double chi2, double area,
                                   const double *tab)
{
    if (chi2 < 25.0 && chi2 >= 0.0) return g.pnorm * fexp(-0.5 * chi2, tab) * area;
    return 0.0;
}

// gauss2d_eval_pixel (gmix_nb.py:66-92): true exp, no cut
NGMIX_HD double gauss_eval_exact(const EvalGauss &g, double v, double u,
                                 double area)
{
    double chi2 = gauss_chi2(g, v, u);
    return g.pnorm * exp(-0.5 * chi2) * area;
}

// jacobian_get_vu (jacobian_nb.py:4-16)
NGMIX_HD void jacobian_vu(const ngmix_jacobian &j, double row, double col,
                          double &v, double &u)
{
    double rowdiff = row - j.row0;
    double coldiff = col - j.col0;
    v = j.dvdrow * rowdiff + j.dvdcol * coldiff;
    u = j.dudrow * rowdiff + j.dudcol * coldiff;
}

// gauss2d_set_norm (gmix_nb.py:190-218)
NGMIX_HD int gauss_set_norm(ngmix_gauss2d &g)
{
    if (g.det < LOW_DETVAL) return NGMIX_ERR_DET_TOO_LOW;
    double T = g.irr + g.icc;
    if (T <= LOW_DETVAL) return NGMIX_ERR_T_TOO_LOW;
    double idet = 1.0 / g.det;
    g.drr = g.irr * idet;
    g.drc = g.irc * idet;
    g.dcc = g.icc * idet;
    g.norm = 1.0 / (2 * M_PI * sqrt(g.det));
    g.pnorm = g.p * g.norm;
    g.norm_set = 1;
    return NGMIX_OK;
}

// gauss2d_set (gmix_nb.py:221-240)
NGMIX_HD void gauss_set(ngmix_gauss2d &g, double p, double row, double col,
                        double irr, double irc, double icc)
{
    g.norm_set = 0;
    g.drr = NAN;
    g.drc = NAN;
    g.dcc = NAN;
    g.norm = NAN;
    g.pnorm = NAN;
    g.p = p;
    g.row = row;
    g.col = col;
    g.irr = irr;
    g.irc = irc;
    g.icc = icc;
    g.det = irr * icc - irc * irc;
}

// g1g2_to_e1e2 (gmix_nb.py:652-678)
NGMIX_HD int g1g2_to_e1e2(double g1, double g2, double &e1, double &e2)
{
    double g = sqrt(g1 * g1 + g2 * g2);
    if (g >= 1) return NGMIX_ERR_G_RANGE;
    if (g == 0.0) {
        e1 = 0.0;
        e2 = 0.0;
    } else {
        double eta = 2 * atanh(g);
        double e = tanh(eta);
        if (e >= 1.0) e = 0.99999999;
        double fac = e / g;
        e1 = fac * g1;
        e2 = fac * g2;
    }
    return NGMIX_OK;
}

// model tables (gmix_nb.py:243-304); indices 0-5 exp, 6-15 dev, 16-18 turb
struct ModelTables {
    double pvals[20];
    double fvals[20];
};

#define NGMIX_MODEL_TABLES                                                      \
    {{0.00061601229677880041, 0.0079461395724623237, 0.053280454055540001,     \
      0.21797364640726541, 0.45496740582554868, 0.26521634184240478,           \
      6.5288960012625658e-05, 0.00044199216814302695, 0.0020859587871659754,   \
      0.0075913681418996841, 0.02260266219257237, 0.056532254390212859,        \
      0.11939049233042602, 0.20969545753234975, 0.29254151133139222,           \
      0.28905301416582552, 0.596510042804182, 0.4034898268889178,              \
      1.303069003078001e-07, 1.0},                                             \
     {0.002467115141477932, 0.018147435573256168, 0.07944063151366336,         \
      0.27137669897479122, 0.79782256866993773, 2.1623306025075739,            \
      2.9934935706271918e-07, 3.4651596338231207e-06, 2.4807910570562753e-05,  \
      1.4307404300535354e-04, 7.2753169298239500e-04, 3.4582464394427260e-03,  \
      1.6086645440719100e-02, 7.7006776775654429e-02, 4.1012562102501476e-01,  \
      2.9812509778548648e00, 0.5793612389470884, 1.621860687127999,            \
      7.019347162356363, 1.0}}

// number of gaussians a model id produces (gmix.py:1170-1193); 0 = variable
NGMIX_HD int model_ngauss(int model)
{
    switch (model) {
    case NGMIX_MODEL_GAUSS: return 1;
    case NGMIX_MODEL_TURB: return 3;
    case NGMIX_MODEL_EXP: return 6;
    case NGMIX_MODEL_DEV: return 10;
    case NGMIX_MODEL_BDF:
    case NGMIX_MODEL_BD:
    case NGMIX_MODEL_CM: return 16;
    default: return 0;
    }
}

NGMIX_HD int model_npars(int model)
{
    switch (model) {
    case NGMIX_MODEL_GAUSS:
    case NGMIX_MODEL_TURB:
    case NGMIX_MODEL_EXP:
    case NGMIX_MODEL_DEV:
    case NGMIX_MODEL_CM: return 6;
    case NGMIX_MODEL_BDF: return 7;
    case NGMIX_MODEL_BD: return 8;
    default: return 0;
    }
}

// get_cm_Tfactor (gmix_nb.py:561-593)
NGMIX_HD int cm_Tfactor(const ModelTables &t, double fracdev, double TdByTe,
                        double &out)
{
    double ifracdev = 1.0 - fracdev;
    double Tfactor = 0.0;
    for (int i = 0; i < 6; i++) {
        double p = t.pvals[i] * ifracdev;
        double f = t.fvals[i];
        Tfactor += p * f;
    }
    for (int i = 0; i < 10; i++) {
        double p = t.pvals[6 + i] * fracdev;
        double f = t.fvals[6 + i] * TdByTe;
        Tfactor += p * f;
    }
    if (Tfactor == 0.0) return NGMIX_ERR_ZERO_DIV;
    out = 1.0 / Tfactor;
    return NGMIX_OK;
}

// One component of a model mixture.  Covers gmix_fill_simple (gmix_nb.py:
// 307-340), the composite body shared by cm/bd/bdf (:430-558), coellip
// (:375-405) and full (:408-427).  Per-model scalars are prepared once by
// fill_prepare, then component i is independent of the others.
struct FillCtx {
    int model, ngauss;
    double row, col, e1, e2, T, flux;
    double fracdev, ifracdev, TdByTe;
};

NGMIX_HD int fill_prepare(const ModelTables &t, int model, int ngauss,
                          const double *pars, const double *cm_extra,
                          FillCtx &c)
{
    c.model = model;
    c.ngauss = ngauss;
    c.fracdev = 0.0;
    c.ifracdev = 1.0;
    c.TdByTe = 1.0;
    if (model == NGMIX_MODEL_FULL) return NGMIX_OK;
    c.row = pars[0];
    c.col = pars[1];
    int st = g1g2_to_e1e2(pars[2], pars[3], c.e1, c.e2);
    c.T = pars[4];
    c.flux = pars[5];
    switch (model) {
    case NGMIX_MODEL_CM:
        c.fracdev = cm_extra[0];
        c.TdByTe = cm_extra[1];
        c.T = pars[4] * cm_extra[2];
        c.flux = pars[5];
        c.ifracdev = 1.0 - c.fracdev;
        break;
    case NGMIX_MODEL_BD: {
        double lTrat = pars[5];
        c.fracdev = pars[6];
        c.flux = pars[7];
        c.TdByTe = pow(10.0, lTrat);
        double tf;
        int st2 = cm_Tfactor(t, c.fracdev, c.TdByTe, tf);
        if (st2) return st2;
        c.T = pars[4] * tf;
        c.ifracdev = 1.0 - c.fracdev;
        break;
    }
    case NGMIX_MODEL_BDF: {
        c.fracdev = pars[5];
        c.flux = pars[6];
        c.TdByTe = 1.0;
        double tf;
        int st2 = cm_Tfactor(t, c.fracdev, c.TdByTe, tf);
        if (st2) return st2;
        c.T = pars[4] * tf;
        c.ifracdev = 1.0 - c.fracdev;
        break;
    }
    default:
        break;
    }
    // the reference evaluates g1g2_to_e1e2 after the Tfactor for bd/bdf; the
    // only observable difference is which error wins when both fail
    return st;
}

NGMIX_HD void fill_component(const ModelTables &t, const FillCtx &c,
                             const double *pars, int i, ngmix_gauss2d &g)
{
    if (c.model == NGMIX_MODEL_FULL) {
        const double *q = pars + 6 * i;
        gauss_set(g, q[0], q[1], q[2], q[3], q[4], q[5]);
        return;
    }
    double T_i_2, flux_i;
    if (c.model == NGMIX_MODEL_COELLIP) {
        double T = pars[4 + i];
        T_i_2 = 0.5 * T;
        flux_i = pars[4 + c.ngauss + i];
    } else if (c.model == NGMIX_MODEL_CM || c.model == NGMIX_MODEL_BD ||
               c.model == NGMIX_MODEL_BDF) {
        double p, f;
        if (i < 6) {
            p = t.pvals[i] * c.ifracdev;
            f = t.fvals[i];
        } else {
            p = t.pvals[i] * c.fracdev;
            f = t.fvals[i] * c.TdByTe;
        }
        T_i_2 = 0.5 * c.T * f;
        flux_i = c.flux * p;
    } else {
        int off = 0;
        if (c.model == NGMIX_MODEL_DEV) off = 6;
        if (c.model == NGMIX_MODEL_TURB) off = 16;
        if (c.model == NGMIX_MODEL_GAUSS) off = 19;
        T_i_2 = 0.5 * c.T * t.fvals[off + i];
        flux_i = c.flux * t.pvals[off + i];
    }
    gauss_set(g, flux_i, c.row, c.col, T_i_2 * (1 - c.e1), T_i_2 * c.e2,
              T_i_2 * (1 + c.e1));
}

// gmix_get_cen (gmix_nb.py:108-130)
NGMIX_HD int gmix_cen(const ngmix_gauss2d *gm, int n, double &row, double &col,
                      double &psum)
{
    row = 0.0;
    col = 0.0;
    psum = 0.0;
    for (int i = 0; i < n; i++) {
        double p = gm[i].p;
        row += p * gm[i].row;
        col += p * gm[i].col;
        psum += p;
    }
    if (psum == 0.0) return NGMIX_ERR_ZERO_DIV;
    row /= psum;
    col /= psum;
    return NGMIX_OK;
}

// one output component of gmix_convolve_fill (gmix_nb.py:632-649)
NGMIX_HD void convolve_component(const ngmix_gauss2d &o, const ngmix_gauss2d &q,
                                 double psf_rowcen, double psf_colcen,
                                 double psf_ipsum, ngmix_gauss2d &out)
{
    double p = o.p * q.p * psf_ipsum;
    double row = o.row + (q.row - psf_rowcen);
    double col = o.col + (q.col - psf_colcen);
    double irr = o.irr + q.irr;
    double irc = o.irc + q.irc;
    double icc = o.icc + q.icc;
    gauss_set(out, p, row, col, irr, irc, icc);
}

// gmix_get_moms (em_nb.py:1260-1294)
NGMIX_HD int gmix_moms(const ngmix_gauss2d *gm, int n, double &irr, double &irc,
                       double &icc)
{
    double row, col, psum;
    int st = gmix_cen(gm, n, row, col, psum);
    if (st) return st;
    irr = irc = icc = 0.0;
    for (int i = 0; i < n; i++) {
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
    return NGMIX_OK;
}

}  // namespace ngmix

// ---- host-side error plumbing (capi.hip owns the storage) ----------------
namespace ngmix {
void set_last_error(const char *what, hipError_t err);
void set_last_error_msg(const char *msg);
// launch census (capi.hip): every batch launcher names the kernel variant it
// dispatches, so that a test can assert WHICH kernel served a workload (a
// silent fall-back to a generic kernel is a performance bug no parity test
// sees).  A mutex-protected map; a few hundred nanoseconds per launch.
void census(const char *kernel);
}  // namespace ngmix

#define NGMIX_HIP_CHECK(expr)                                   \
    do {                                                        \
        hipError_t _e = (expr);                                 \
        if (_e != hipSuccess) {                                 \
            ngmix::set_last_error(#expr, _e);                   \
            return NGMIX_ERR_HIP;                               \
        }                                                       \
    } while (0)
