// lm_core.hpp -- a re-entrant Levenberg-Marquardt iteration with the decision
// logic of MINPACK's lmder (the routine scipy.optimize.leastsq runs when it is
// given Dfun, which is what the reference's Fitter does for gauss/exp/dev:
// ngmix/fitting/leastsqbound.py:445-447,485-493, fitters.py:93-97).
//
// MINPACK is callback driven: it owns the loop and calls the objective.  Here
// the loop is turned inside out so that N independent fits advance in lock
// step, one objective/jacobian kernel launch per step for all of them: the
// caller evaluates, at the trial point the state asks for,
//
//     ff = |f|^2,   g = J^T f,   A = J^T J          (n <= LM_NPMAX parameters)
//
// and lm_advance() consumes them and either finishes the fit or asks for the
// next trial point.  lmder works on the pivoted QR of J (R, ipvt, Q^T f);
// R is the pivoted Cholesky factor of A and the first n entries of Q^T f are
// R^-T P^T g, so the whole algorithm -- qrfac's pivot rule, the scaled
// gradient test, lmpar with qrsolv, the trust region update, the four
// convergence tests -- runs from (A, g, ff) and follows MINPACK's path up to
// the rounding of the factorisation (cond(J)^2 instead of cond(J)).
//
// Two modes.  ANALYTIC (lmder): the jacobian is evaluated at every trial
// point, not only at accepted ones -- one launch per step instead of two; nfev
// / njev still count what lmder counts.  FD (lmdif, what the reference runs for
// the models without analytic derivatives): a trial wants only |f|^2; after an
// accepted step the state asks (phase JAC) for the forward-difference jacobian
// at the new point and charges its n evaluations to nfev as fdjac2 does.
//
// Everything is host+device so the same code is tested on the CPU against
// scipy.optimize.leastsq (tests/test_lm_core.py).
#pragma once

#include <math.h>
#include <stdint.h>

#ifndef NGMIX_HD
#define NGMIX_HD __host__ __device__ __forceinline__
#endif

#include "../../include/ngmix_hip.h"

#define LM_NPMAX NGMIX_LM_NPMAX
#define LM_PHASE_INIT NGMIX_LM_PHASE_INIT    /* waiting for the evaluation at x0 */
#define LM_PHASE_TRIAL NGMIX_LM_PHASE_TRIAL  /* waiting for the evaluation at xt */
#define LM_PHASE_DONE NGMIX_LM_PHASE_DONE
#define LM_PHASE_JAC NGMIX_LM_PHASE_JAC     /* FD mode: waiting for the jacobian at x */

typedef ngmix_lm_state lm_state;

namespace lmcore {

constexpr double EPSMCH = 2.220446049250313e-16;
constexpr double DWARF = 2.2250738585072014e-308;

NGMIX_HD double enorm(int n, const double *x)
{
    // MINPACK's enorm guards against over/underflow with three accumulators;
    // the quantities here (parameter steps, scaled gradients) are far from
    // either limit, where it reduces to this
    double s = 0.0;
    for (int i = 0; i < n; i++) s += x[i] * x[i];
    return sqrt(s);
}

// The work arrays and the factor R are NP-strided, NP a template parameter:
// NP = LM_NPMAX on the ngmix_lm_state record itself (host entry points, tests),
// a smaller NP on a compact copy of the live part of the record (lm_state_n
// below) in the device kernel -- one thread per fit keeps everything in private
// memory, and a six-parameter fit in LM_NPMAX = 14 arrays moves five times the
// bytes it needs.
template <int NP>
struct lm_state_n {
    double x[NP], xt[NP], diag[NP], R[NP * NP], qtf[NP], step[NP];
    double fnorm, xnorm, delta, par, gnorm, pnorm;
    double ftol, xtol, gtol, factor;
    double xi[NP], xti[NP], lo[NP], hi[NP], xstep[NP], hstep[NP];
    int32_t ipvt[NP];
    int32_t n, iter, nfev, njev, info, phase, maxfev, mode, bounded, fonly;
};

// Pivoted Cholesky of A = J^T J with qrfac's pivot rule (largest remaining
// column norm first): A P = (QR)^T (QR) P  ->  R^T R = P^T A P.
// Returns R (upper, n x n in an NP-strided array), ipvt, and
// acnorm[j] = |J[:, j]|.
template <int NP = LM_NPMAX>
NGMIX_HD void factor_normal(int n, const double *A, double *R, int32_t *ipvt,
                            double *acnorm)
{
    double S[NP * NP];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) S[i * NP + j] = A[i * NP + j];
    for (int j = 0; j < n; j++) {
        ipvt[j] = j;
        const double d = A[j * NP + j];
        acnorm[j] = d > 0.0 ? sqrt(d) : 0.0;
    }
    // (only the leading n x n block of the NP-strided arrays is ever read)
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) R[i * NP + j] = 0.0;
    // S is kept in the permuted order: row/col k of S <-> parameter ipvt[k]
    for (int k = 0; k < n; k++) {
        int kmax = k;
        for (int j = k + 1; j < n; j++)
            if (S[j * NP + j] > S[kmax * NP + kmax]) kmax = j;
        if (kmax != k) {
            for (int i = 0; i < n; i++) {
                const double t = S[i * NP + k];
                S[i * NP + k] = S[i * NP + kmax];
                S[i * NP + kmax] = t;
            }
            for (int j = 0; j < n; j++) {
                const double t = S[k * NP + j];
                S[k * NP + j] = S[kmax * NP + j];
                S[kmax * NP + j] = t;
            }
            for (int i = 0; i < k; i++) {
                const double t = R[i * NP + k];
                R[i * NP + k] = R[i * NP + kmax];
                R[i * NP + kmax] = t;
            }
            const int32_t ti = ipvt[k];
            ipvt[k] = ipvt[kmax];
            ipvt[kmax] = ti;
        }
        const double d = S[k * NP + k];
        if (!(d > 0.0)) {
            // rank deficient: the remaining columns are (numerically) in the
            // span of the first k; qrfac leaves rdiag = 0 there
            for (int j = k; j < n; j++) R[k * NP + j] = 0.0;
            for (int kk = k + 1; kk < n; kk++)
                for (int j = kk; j < n; j++) R[kk * NP + j] = 0.0;
            return;
        }
        const double rkk = sqrt(d);
        R[k * NP + k] = rkk;
        for (int j = k + 1; j < n; j++) R[k * NP + j] = S[k * NP + j] / rkk;
        for (int i = k + 1; i < n; i++)
            for (int j = i; j < n; j++) {
                const double v = S[i * NP + j] -
                                 R[k * NP + i] * R[k * NP + j];
                S[i * NP + j] = v;
                S[j * NP + i] = v;
            }
    }
}

// first n components of Q^T f:  R^T qtf = P^T g
template <int NP = LM_NPMAX>
NGMIX_HD void qtf_from_gradient(int n, const double *R, const int32_t *ipvt,
                                const double *g, double *qtf)
{
    for (int j = 0; j < n; j++) {
        double s = g[ipvt[j]];
        for (int i = 0; i < j; i++) s -= R[i * NP + j] * qtf[i];
        const double rjj = R[j * NP + j];
        qtf[j] = rjj != 0.0 ? s / rjj : 0.0;
    }
}

// MINPACK qrsolv.  r: n x n with the upper triangle holding R; on output the
// strict lower triangle holds the strict upper triangle of S transposed and
// sdiag the diagonal of S.
template <int NP = LM_NPMAX>
NGMIX_HD void qrsolv(int n, double *r, const int32_t *ipvt, const double *diag,
                     const double *qtb, double *x, double *sdiag, double *wa)
{
    for (int j = 0; j < n; j++) {
        for (int i = j; i < n; i++) r[i * NP + j] = r[j * NP + i];
        x[j] = r[j * NP + j];
        wa[j] = qtb[j];
    }
    for (int j = 0; j < n; j++) {
        const int l = ipvt[j];
        if (diag[l] != 0.0) {
            for (int k = j; k < n; k++) sdiag[k] = 0.0;
            sdiag[j] = diag[l];
            double qtbpj = 0.0;
            for (int k = j; k < n; k++) {
                if (sdiag[k] == 0.0) continue;
                double cs, sn;
                const double rkk = r[k * NP + k];
                if (fabs(rkk) < fabs(sdiag[k])) {
                    const double cotan = rkk / sdiag[k];
                    sn = 0.5 / sqrt(0.25 + 0.25 * (cotan * cotan));
                    cs = sn * cotan;
                } else {
                    const double tn = sdiag[k] / rkk;
                    cs = 0.5 / sqrt(0.25 + 0.25 * (tn * tn));
                    sn = cs * tn;
                }
                r[k * NP + k] = cs * rkk + sn * sdiag[k];
                double temp = cs * wa[k] + sn * qtbpj;
                qtbpj = -sn * wa[k] + cs * qtbpj;
                wa[k] = temp;
                for (int i = k + 1; i < n; i++) {
                    temp = cs * r[i * NP + k] + sn * sdiag[i];
                    sdiag[i] = -sn * r[i * NP + k] + cs * sdiag[i];
                    r[i * NP + k] = temp;
                }
            }
        }
        sdiag[j] = r[j * NP + j];
        r[j * NP + j] = x[j];
    }
    int nsing = n;
    for (int j = 0; j < n; j++) {
        if (sdiag[j] == 0.0 && nsing == n) nsing = j;
        if (nsing < n) wa[j] = 0.0;
    }
    for (int k = 0; k < nsing; k++) {
        const int j = nsing - 1 - k;
        double sum = 0.0;
        for (int i = j + 1; i < nsing; i++) sum += r[i * NP + j] * wa[i];
        wa[j] = (wa[j] - sum) / sdiag[j];
    }
    for (int j = 0; j < n; j++) x[ipvt[j]] = wa[j];
}

// MINPACK lmpar.  r is modified as qrsolv leaves it (upper triangle intact).
template <int NP = LM_NPMAX>
NGMIX_HD void lmpar(int n, double *r, const int32_t *ipvt, const double *diag,
                    const double *qtb, double delta, double &par, double *x,
                    double *sdiag)
{
    double wa1[NP], wa2[NP];
    // gauss-newton direction
    int nsing = n;
    for (int j = 0; j < n; j++) {
        wa1[j] = qtb[j];
        if (r[j * NP + j] == 0.0 && nsing == n) nsing = j;
        if (nsing < n) wa1[j] = 0.0;
    }
    for (int k = 0; k < nsing; k++) {
        const int j = nsing - 1 - k;
        wa1[j] /= r[j * NP + j];
        const double temp = wa1[j];
        for (int i = 0; i < j; i++) wa1[i] -= r[i * NP + j] * temp;
    }
    for (int j = 0; j < n; j++) x[ipvt[j]] = wa1[j];

    int iter = 0;
    for (int j = 0; j < n; j++) wa2[j] = diag[j] * x[j];
    double dxnorm = enorm(n, wa2);
    double fp = dxnorm - delta;
    if (fp <= 0.1 * delta) {
        par = 0.0;
        return;
    }
    // lower bound
    double parl = 0.0;
    if (nsing >= n) {
        for (int j = 0; j < n; j++) {
            const int l = ipvt[j];
            wa1[j] = diag[l] * (wa2[l] / dxnorm);
        }
        for (int j = 0; j < n; j++) {
            double sum = 0.0;
            for (int i = 0; i < j; i++) sum += r[i * NP + j] * wa1[i];
            wa1[j] = (wa1[j] - sum) / r[j * NP + j];
        }
        const double temp = enorm(n, wa1);
        parl = ((fp / delta) / temp) / temp;
    }
    // upper bound
    for (int j = 0; j < n; j++) {
        double sum = 0.0;
        for (int i = 0; i <= j; i++) sum += r[i * NP + j] * qtb[i];
        wa1[j] = sum / diag[ipvt[j]];
    }
    const double gnorm = enorm(n, wa1);
    double paru = gnorm / delta;
    if (paru == 0.0) paru = DWARF / fmin(delta, 0.1);
    par = fmax(par, parl);
    par = fmin(par, paru);
    if (par == 0.0) par = gnorm / dxnorm;

    for (;;) {
        iter++;
        if (par == 0.0) par = fmax(DWARF, 0.001 * paru);
        double temp = sqrt(par);
        for (int j = 0; j < n; j++) wa1[j] = temp * diag[j];
        qrsolv<NP>(n, r, ipvt, wa1, qtb, x, sdiag, wa2);
        for (int j = 0; j < n; j++) wa2[j] = diag[j] * x[j];
        dxnorm = enorm(n, wa2);
        temp = fp;
        fp = dxnorm - delta;
        if (fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) ||
            iter == 10)
            break;
        // newton correction
        for (int j = 0; j < n; j++) {
            const int l = ipvt[j];
            wa1[j] = diag[l] * (wa2[l] / dxnorm);
        }
        for (int j = 0; j < n; j++) {
            wa1[j] /= sdiag[j];
            temp = wa1[j];
            for (int i = j + 1; i < n; i++) wa1[i] -= r[i * NP + j] * temp;
        }
        temp = enorm(n, wa1);
        const double parc = ((fp / delta) / temp) / temp;
        if (fp > 0.0) parl = fmax(parl, par);
        if (fp < 0.0) paru = fmin(paru, par);
        par = fmax(parl, par + parc);
    }
}

// ---- leastsqbound's parameter transforms (leastsqbound.py:183-262) ----------
NGMIX_HD double i2e(double v, double lo, double hi)
{
    const bool has_lo = lo > -INFINITY, has_hi = hi < INFINITY;
    if (!has_lo && !has_hi) return v;
    if (!has_hi) return lo - 1.0 + sqrt(v * v + 1.0);
    if (!has_lo) return hi + 1.0 - sqrt(v * v + 1.0);
    return lo + ((hi - lo) / 2.0) * (sin(v) + 1.0);
}

NGMIX_HD double e2i(double x, double lo, double hi)
{
    const bool has_lo = lo > -INFINITY, has_hi = hi < INFINITY;
    if (!has_lo && !has_hi) return x;
    if (!has_hi) return sqrt((x - lo + 1.0) * (x - lo + 1.0) - 1.0);
    if (!has_lo) return sqrt((hi - x + 1.0) * (hi - x + 1.0) - 1.0);
    return asin((2.0 * (x - lo) / (hi - lo)) - 1.0);
}

NGMIX_HD double i2e_grad(double v, double lo, double hi)
{
    const bool has_lo = lo > -INFINITY, has_hi = hi < INFINITY;
    if (!has_lo && !has_hi) return 1.0;
    if (!has_hi) return v / sqrt(v * v + 1.0);
    if (!has_lo) return -v / sqrt(v * v + 1.0);
    return (hi - lo) * cos(v) / 2.0;
}

// the external trial point from the internal one, and fdjac2's points
// (h = sqrt(eps) |x_j|, sqrt(eps) at 0, in the internal parameters)
template <class State>
NGMIX_HD void set_trial(State &s)
{
    constexpr double EPS = 1.4901161193847656e-08;  // sqrt(machine epsilon)
    for (int j = 0; j < s.n; j++) {
        s.xt[j] = s.bounded ? i2e(s.xti[j], s.lo[j], s.hi[j]) : s.xti[j];
        if (s.mode == NGMIX_LM_MODE_FD) {
            double h = EPS * fabs(s.xti[j]);
            if (h == 0.0) h = EPS;
            s.hstep[j] = h;
            s.xstep[j] = s.bounded ? i2e(s.xti[j] + h, s.lo[j], s.hi[j]) : s.xti[j] + h;
        }
    }
}

// lmpar on the stored factor, trial point, and the quantities the ratio test
// needs afterwards (lmder: the body of the inner loop up to the evaluation)
template <int NP = LM_NPMAX, class State = lm_state>
NGMIX_HD void propose(State &s)
{
    const int n = s.n;
    double r[NP * NP], sdiag[NP], p[NP], wa3[NP];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) r[i * NP + j] = s.R[i * NP + j];
    lmpar<NP>(n, r, s.ipvt, s.diag, s.qtf, s.delta, s.par, p, sdiag);
    for (int j = 0; j < n; j++) {
        s.step[j] = -p[j];
        s.xti[j] = s.xi[j] + s.step[j];
        wa3[j] = s.diag[j] * s.step[j];
    }
    set_trial(s);
    s.pnorm = enorm(n, wa3);
    if (s.iter == 1) s.delta = fmin(s.delta, s.pnorm);
    s.fonly = 0;
    if (s.mode == NGMIX_LM_MODE_ANALYTIC_LAZY) {
        // Will accepting this trial end the fit?  lmder's predicted reduction
        // is known before the evaluation (the same expression lm_advance forms
        // afterwards); near the solution the actual reduction follows it, and
        // the ftol test passes when both are <= ftol.  The xtol test compares
        // the step bound after the update, at most pnorm / 0.5, with xnorm.
        // If so the evaluation needs |f|^2 only -- lmder never forms the
        // jacobian at its last point; a miss costs this fit one more round
        // (phase JAC), never a different iterate.
        double w3[NP];
        for (int j = 0; j < n; j++) w3[j] = 0.0;
        for (int j = 0; j < n; j++) {
            const double temp = s.step[s.ipvt[j]];
            for (int i = 0; i <= j; i++) w3[i] += s.R[i * NP + j] * temp;
        }
        const double temp1 = enorm(n, w3) / s.fnorm;
        const double temp2 = (sqrt(s.par) * s.pnorm) / s.fnorm;
        const double prered = temp1 * temp1 + temp2 * temp2 / 0.5;
        if (prered <= s.ftol || s.pnorm / 0.5 <= s.xtol * s.xnorm) s.fonly = 1;
    }
}

// the outer-loop head of lmder at the point whose normal equations are
// (A, g): factor, scale, gradient test, then the first proposal.
// Returns true when the fit has terminated (info set).
template <int NP = LM_NPMAX, class State = lm_state>
NGMIX_HD bool new_jacobian(State &s, const double *A, const double *g)
{
    const int n = s.n;
    double acnorm[NP];
    s.njev++;
    factor_normal<NP>(n, A, s.R, s.ipvt, acnorm);
    if (s.iter == 1) {
        double wa3[NP];
        for (int j = 0; j < n; j++) {
            s.diag[j] = acnorm[j];
            if (acnorm[j] == 0.0) s.diag[j] = 1.0;
            wa3[j] = s.diag[j] * s.xi[j];
        }
        s.xnorm = enorm(n, wa3);
        s.delta = s.factor * s.xnorm;
        if (s.delta == 0.0) s.delta = s.factor;
    }
    qtf_from_gradient<NP>(n, s.R, s.ipvt, g, s.qtf);
    // norm of the scaled gradient
    s.gnorm = 0.0;
    if (s.fnorm != 0.0) {
        for (int j = 0; j < n; j++) {
            const int l = s.ipvt[j];
            if (acnorm[l] == 0.0) continue;
            double sum = 0.0;
            for (int i = 0; i <= j; i++) sum += s.R[i * NP + j] * (s.qtf[i] / s.fnorm);
            s.gnorm = fmax(s.gnorm, fabs(sum / acnorm[l]));
        }
    }
    if (s.gnorm <= s.gtol) {
        s.info = 4;
        s.phase = LM_PHASE_DONE;
        return true;
    }
    for (int j = 0; j < n; j++) s.diag[j] = fmax(s.diag[j], acnorm[j]);
    propose<NP>(s);
    return false;
}

NGMIX_HD void lm_init(lm_state &s, int n, const double *x0, double ftol, double xtol,
                      double gtol, int maxfev, double factor,
                      int mode = NGMIX_LM_MODE_ANALYTIC, const double *lo = nullptr,
                      const double *hi = nullptr)
{
    s.n = n;
    s.mode = mode;
    s.bounded = 0;
    s.fonly = 0;
    for (int j = 0; j < LM_NPMAX; j++) {
        s.lo[j] = (lo && j < n) ? lo[j] : -INFINITY;
        s.hi[j] = (hi && j < n) ? hi[j] : INFINITY;
        if (s.lo[j] > -INFINITY || s.hi[j] < INFINITY) s.bounded = 1;
    }
    for (int j = 0; j < LM_NPMAX; j++) {
        // i0 = e2i(x0); the first evaluation is at i2e(i0) (leastsqbound.py:454)
        s.xi[j] = j < n ? (s.bounded ? e2i(x0[j], s.lo[j], s.hi[j]) : x0[j]) : 0.0;
        s.xti[j] = s.xi[j];
        s.xt[j] = s.xstep[j] = s.hstep[j] = 0.0;
        s.diag[j] = 0.0;
        s.qtf[j] = 0.0;
        s.step[j] = 0.0;
        s.ipvt[j] = j;
    }
    set_trial(s);
    for (int j = 0; j < LM_NPMAX; j++) s.x[j] = s.xt[j];
    for (int i = 0; i < LM_NPMAX * LM_NPMAX; i++) s.R[i] = 0.0;
    s.fnorm = s.xnorm = s.delta = s.par = s.gnorm = s.pnorm = 0.0;
    s.ftol = ftol;
    s.xtol = xtol;
    s.gtol = gtol;
    s.factor = factor;
    s.maxfev = maxfev;
    s.iter = 1;
    s.nfev = s.njev = 0;
    s.info = 0;
    s.phase = LM_PHASE_INIT;
}

// Consume the evaluation at s.xt:  ff = |f|^2, g = J^T f, A = J^T J
// (A, g in NP-strided / NP-long arrays).  ff may be +inf (the
// model was out of range at xt: the reference's calc_fdiff returns -inf
// residuals there); A and g are then ignored.
template <int NP = LM_NPMAX, class State = lm_state>
NGMIX_HD void lm_advance(State &s, double ff, const double *g_in, const double *A_in)
{
    const int n = s.n;
    if (s.phase == LM_PHASE_DONE) return;
    // analytic jacobians are with respect to the external parameters: the
    // wrapped Dfun of leastsqbound.py:485-489 scales column j by d xt_j / d xti_j
    // (forward differences are taken in the internal parameters already)
    double gs[NP], As[NP * NP];
    const double *g = g_in, *A = A_in;
    if (s.bounded && s.mode != NGMIX_LM_MODE_FD) {
        double sc[NP];
        for (int j = 0; j < n; j++) sc[j] = i2e_grad(s.xti[j], s.lo[j], s.hi[j]);
        for (int j = 0; j < n; j++) {
            gs[j] = g_in[j] * sc[j];
            for (int k = 0; k < n; k++)
                As[j * NP + k] = A_in[j * NP + k] * sc[j] * sc[k];
        }
        g = gs;
        A = As;
    }
    if (s.phase == LM_PHASE_JAC) {
        // lmdif: the forward-difference jacobian at the accepted point cost
        // n evaluations (fdjac2); then the outer-loop head.  (Mode
        // ANALYTIC_LAZY: the analytic jacobian asked for after an |f|^2-only
        // trial -- lmder's own call with iflag = 2, counted by njev alone.)
        if (s.mode == NGMIX_LM_MODE_FD) s.nfev += n;
        s.phase = LM_PHASE_TRIAL;
        new_jacobian<NP>(s, A, g);
        return;
    }
    if (s.phase == LM_PHASE_INIT) {
        // lmder / lmdif: fvec at the starting point, then the outer loop
        s.nfev = s.mode == NGMIX_LM_MODE_FD ? 1 + n : 1;
        s.fnorm = sqrt(ff);
        s.par = 0.0;
        s.iter = 1;
        if (!(s.fnorm < INFINITY)) {
            // the guess itself is out of range: the reference hands MINPACK
            // -inf residuals and a zero jacobian (results.py:463-464,567-568),
            // for which lmder's gradient test ends the fit at once with
            // info 4 and a singular factor (-> LM_SINGULAR_MATRIX downstream)
            s.njev = 1;
            s.info = 4;
            s.phase = LM_PHASE_DONE;
            return;
        }
        s.phase = LM_PHASE_TRIAL;
        new_jacobian<NP>(s, A, g);
        return;
    }

    // ---- LM_PHASE_TRIAL: the rest of lmder's inner loop
    s.nfev++;
    // an out-of-range trial is a residual vector of -inf in the reference;
    // MINPACK's enorm of that is NaN (inf/inf in its scaled sums), and the
    // comparisons below then go the way they go for a NaN
    const double fnorm1 = ff < INFINITY ? sqrt(ff) : NAN;
    double actred = -1.0;
    if (0.1 * fnorm1 < s.fnorm) {
        const double t = fnorm1 / s.fnorm;
        actred = 1.0 - t * t;
    }
    // predicted reduction and directional derivative
    double wa3[NP];
    for (int j = 0; j < n; j++) wa3[j] = 0.0;
    for (int j = 0; j < n; j++) {
        const double temp = s.step[s.ipvt[j]];
        for (int i = 0; i <= j; i++) wa3[i] += s.R[i * NP + j] * temp;
    }
    const double temp1 = enorm(n, wa3) / s.fnorm;
    const double temp2 = (sqrt(s.par) * s.pnorm) / s.fnorm;
    const double prered = temp1 * temp1 + temp2 * temp2 / 0.5;
    const double dirder = -(temp1 * temp1 + temp2 * temp2);
    double ratio = 0.0;
    if (prered != 0.0) ratio = actred / prered;
    // update the step bound
    if (ratio <= 0.25) {
        double temp = 0.5;
        if (actred < 0.0) temp = 0.5 * dirder / (dirder + 0.5 * actred);
        if (0.1 * fnorm1 >= s.fnorm || temp < 0.1) temp = 0.1;
        s.delta = temp * fmin(s.delta, s.pnorm / 0.1);
        s.par = s.par / temp;
    } else if (s.par == 0.0 || ratio >= 0.75) {
        s.delta = s.pnorm / 0.5;
        s.par = 0.5 * s.par;
    }
    const bool accepted = ratio >= 1.0e-4;
    if (accepted) {
        double w[NP];
        for (int j = 0; j < n; j++) {
            s.x[j] = s.xt[j];
            s.xi[j] = s.xti[j];
            w[j] = s.diag[j] * s.xi[j];
        }
        s.xnorm = enorm(n, w);
        s.fnorm = fnorm1;
        s.iter++;
    }
    // convergence tests
    int info = 0;
    if (fabs(actred) <= s.ftol && prered <= s.ftol && 0.5 * ratio <= 1.0) info = 1;
    if (s.delta <= s.xtol * s.xnorm) info = 2;
    if (fabs(actred) <= s.ftol && prered <= s.ftol && 0.5 * ratio <= 1.0 && info == 2)
        info = 3;
    if (info == 0) {
        // termination and stringent tolerances
        if (s.nfev >= s.maxfev) info = 5;
        if (fabs(actred) <= EPSMCH && prered <= EPSMCH && 0.5 * ratio <= 1.0) info = 6;
        if (s.delta <= EPSMCH * s.xnorm) info = 7;
        if (s.gnorm <= EPSMCH) info = 8;
    }
    if (info != 0) {
        s.info = info;
        s.phase = LM_PHASE_DONE;
        return;
    }
    if (!accepted) {
        propose<NP>(s);  // same factor, smaller region
    } else if (s.mode == NGMIX_LM_MODE_FD || s.fonly) {
        // ask for the jacobian at the new point
        for (int j = 0; j < n; j++) s.xti[j] = s.xi[j];
        set_trial(s);
        s.phase = LM_PHASE_JAC;
        s.fonly = 0;
    } else {
        new_jacobian<NP>(s, A, g);  // the trial point's jacobian is the new one
    }
}

// ---- the separable joint prior (ngmix_simple_sep_prior) --------------------
// ln p of one 1-d term (T, a middle term or a flux); returns false where the
// reference raises GMixRangeError (FlatPrior / TruncatedGaussian outside their
// range, LogNormal at or below its shift)
NGMIX_HD bool prior_term_lnp(int kind, const double *par, double x, double &lnp)
{
    if (kind == NGMIX_PRIOR_TWO_SIDED_ERF) {
        // priors/priors.py:219-251
        const double p = 0.5 * erf((par[2] - x) / par[3]) + 0.5 * erf((x - par[0]) / par[1]);
        lnp = p > 0.0 ? log(p) : -INFINITY;
        return true;
    }
    if (kind == NGMIX_PRIOR_NORMAL) {
        // priors/priors.py:424-434
        const double diff = x - par[0];
        lnp = -0.5 * diff * diff * (1.0 / (par[1] * par[1]));
        return true;
    }
    if (kind == NGMIX_PRIOR_LOGNORMAL) {
        // priors/priors.py:732-756: ln p is 0 at the mode
        const double v = x - par[3];
        if (!(v > 0.0)) return false;
        const double logv = log(v), d = logv - par[0];
        lnp = -0.5 * (par[1] * (d * d)) - logv - par[2];
        return true;
    }
    if (kind == NGMIX_PRIOR_TRUNCATED_GAUSSIAN) {
        // priors/priors.py:1084-1096
        if (x < par[2] || x > par[3]) return false;
        const double diff = x - par[0];
        lnp = -0.5 * diff * diff * (1.0 / (par[1] * par[1]));
        return true;
    }
    lnp = 0.0;
    return !(x < par[0] || x > par[1]);
}

NGMIX_HD double prior_root(double lnp)
{
    double chi2 = -2.0 * lnp;
    if (chi2 < 0.0) chi2 = 0.0;
    return sqrt(chi2);
}

// one term's residual row and ln p: sqrt(max(-2 ln p, 0)), or -- rows from the
// terms' own get_fdiff -- the signed (x - mean) / sigma of the gaussian kinds
NGMIX_HD bool prior_term_row(int kind, const double *par, double x, bool own_fdiff,
                             double &row, double &lnp)
{
    if (!prior_term_lnp(kind, par, x, lnp)) return false;
    if (own_fdiff && (kind == NGMIX_PRIOR_NORMAL || kind == NGMIX_PRIOR_TRUNCATED_GAUSSIAN))
        row = (x - par[0]) * (1.0 / par[1]);
    else
        row = prior_root(lnp);
    return true;
}

constexpr int PRIOR_KMAX = 4 + NGMIX_PRIOR_MAXMID + NGMIX_PRIOR_MAXBAND;
constexpr int PRIOR_NMAX = 5 + NGMIX_PRIOR_MAXMID + NGMIX_PRIOR_MAXBAND;

// the prior is separable: parameter j belongs to exactly one row
NGMIX_HD int prior_row_of(int j) { return j < 2 ? j : (j < 4 ? 2 : j - 1); }

// one row of the prior at x and the ln p of its term: row 0 / 1 the centre
// terms, 2 the shape term, 3.. the 1-d terms of x[4..] (T, the middle terms, the
// fluxes); false where the reference raises GMixRangeError
// (joint_prior.py:86-120, 341-378, 556-590)
NGMIX_HD bool prior_row(const ngmix_simple_sep_prior &P, const double *x, int row, bool own,
                        double &val, double &lnp)
{
    if (row < 2) {
        const double cen = row == 0 ? P.cen1 : P.cen2;
        const double d = cen - x[row];
        lnp = -0.5 * d * d * (row == 0 ? P.cen_s2inv1 : P.cen_s2inv2);
        val = own ? (x[row] - cen) * (row == 0 ? P.cen_sinv1 : P.cen_sinv2) : prior_root(lnp);
        return true;
    }
    if (row == 2) {
        const double gsq = x[2] * x[2] + x[3] * x[3];
        const double omgsq = 1.0 - gsq;
        if (omgsq <= 0.0) return false;
        lnp = 2.0 * log(omgsq) - 0.5 * gsq * P.g_sig2inv;
        val = prior_root(lnp);
        return true;
    }
    const int m = row - 3;               // 0: T, 1..nmid: middle, then the fluxes
    const int kind = m == 0 ? P.T_kind : (m <= P.nmid ? P.mid_kind[m - 1] : P.F_kind[m - 1 - P.nmid]);
    const double *par = m == 0 ? P.T_par : (m <= P.nmid ? P.mid_par[m - 1] : P.F_par[m - 1 - P.nmid]);
    return prior_term_row(kind, par, x[row + 1], own, val, lnp);
}

// rows[k] for k = cen1, cen2, g, T, mid_0.., F_0.. ; false when out of range
NGMIX_HD bool simple_sep_rows(const ngmix_simple_sep_prior &P, const double *x,
                              double *rows, double *lnp_total)
{
    const bool own = P.rows_mode == NGMIX_PRIOR_ROWS_FDIFF;
    const int k = 4 + P.nmid + P.nband;
    double tot = 0.0;
    for (int i = 0; i < k; i++) {
        double lnp;
        if (!prior_row(P, x, i, own, rows[i], lnp)) return false;
        tot += lnp;
    }
    if (lnp_total) *lnp_total = tot;
    return true;
}

// the prior rows of one fit at its trial point as normal-equation sums
// [J^T J upper triangle | J^T r | r.r] over the fit's n = 5 + nmid + nband
// parameters.  The jacobian of the rows is by one-sided differences (analytic
// mode: step_rel * max(1, |x_j|), backward where the forward point is out of
// range, results.py:572-625; forward-difference mode: the state's own fdjac2
// points).  A step in x_j moves the one row x_j belongs to and no other, so
// only that row is evaluated again and column j of the jacobian is one number:
// the sums are those of the full (rows x parameters) difference jacobian, to
// the bit, at two evaluations of the prior instead of n + 1.
template <class State>
NGMIX_HD void simple_sep_normal_sums(const ngmix_simple_sep_prior &P, const State &s,
                                     double step_rel, double *out)
{
    const int n = s.n, k = 4 + P.nmid + P.nband, nt = n * (n + 1) / 2;
    const bool own = P.rows_mode == NGMIX_PRIOR_ROWS_FDIFF;
    double r0[PRIOR_KMAX], col[PRIOR_NMAX], x[PRIOR_NMAX];
    for (int i = 0; i < nt + n + 1; i++) out[i] = 0.0;
#pragma unroll
    for (int j = 0; j < PRIOR_NMAX; j++) x[j] = j < n ? s.xt[j] : 0.0;
    if (n > PRIOR_NMAX || !simple_sep_rows(P, x, r0, nullptr)) {
        out[nt + n] = INFINITY;
        return;
    }
    const bool fd = s.mode == NGMIX_LM_MODE_FD;
#pragma unroll
    for (int j = 0; j < PRIOR_NMAX; j++) {
        if (j >= n) {
            col[j] = 0.0;
            continue;
        }
        const int row = prior_row_of(j);
        double step, rj, lnp;
        bool ok;
        const double xj = x[j];
        if (fd) {
            step = s.hstep[j];
            x[j] = s.xstep[j];
            ok = prior_row(P, x, row, own, rj, lnp);
        } else {
            step = step_rel * fmax(1.0, fabs(xj));
            x[j] = xj + step;
            ok = prior_row(P, x, row, own, rj, lnp);
            if (!ok) {
                step = -step;
                x[j] = xj + step;
                ok = prior_row(P, x, row, own, rj, lnp);
            }
        }
        x[j] = xj;
        const bool good = ok && fabs(r0[row]) < INFINITY && fabs(rj) < INFINITY;
        col[j] = good ? (rj - r0[row]) / step : 0.0;
    }
    // J^T J: parameters of different rows do not meet (only g1 with g2 do)
    int t = 0;
#pragma unroll
    for (int a = 0; a < PRIOR_NMAX; a++) {
        if (a >= n) break;
        const int ra = prior_row_of(a);
        out[t] = 0.0 + col[a] * col[a];
        if (a == 2) out[t + 1] = 0.0 + col[2] * col[3];
        t += n - a;
        if (fabs(r0[ra]) < INFINITY) out[nt + a] = 0.0 + col[a] * r0[ra];
    }
    double ff = 0.0;
    for (int i = 0; i < k; i++) ff += r0[i] * r0[i];
    out[nt + n] = ff;
}

}  // namespace lmcore
