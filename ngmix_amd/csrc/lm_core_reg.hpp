// lm_core_reg.hpp -- lm_core.hpp's iteration for a parameter count N known at
// compile time, written so that every array lives in registers.
//
// Why: lm_advance_kernel runs one thread per fit, a wave or two per SIMD, and a
// single step of the generic code -- its work arrays in private memory, reached
// through run-time indices -- is a serial chain of ~2,000 scratch accesses:
// 0.5 ms per launch whatever the batch size (a 50k-fit launch takes as long as
// a 100k-fit one).  Here every loop has a compile-time trip count and unrolls,
// and the few genuinely run-time indices (MINPACK's pivot order ipvt, the
// pivot search, the rank nsing) are select chains over the N elements, so
// the compiler promotes all arrays to registers (512 per lane at one wave per
// SIMD) and the chain runs at ALU latency.
//
// The operations and their order are those of lm_core.hpp, statement for
// statement: the two produce bit-identical states (tests/test_lm_core.py runs
// both on the host), so MINPACK's path -- nfev, ier -- is unchanged.
#pragma once

#include "lm_core.hpp"

namespace lmreg {

using lmcore::DWARF;
using lmcore::EPSMCH;

// The live part of an ngmix_lm_state record in registers.  The arrays the step
// only writes (x on an accepted step; xstep / hstep of forward-difference mode)
// or reads under a bounds transform (lo, hi) are NOT copied: they are reached
// through pointers into the record itself -- fifty doubles per ten parameters
// that would otherwise sit in registers (or spill) for nothing.
template <int N>
struct lm_state_n {
    double xt[N], diag[N], R[N * N], qtf[N], step[N];
    double fnorm, xnorm, delta, par, gnorm, pnorm;
    double ftol, xtol, gtol, factor;
    double xi[N], xti[N];
    double *x, *xstep, *hstep;   // into the record
    const double *lo, *hi;       // into the record
    int32_t ipvt[N];
    int32_t n, iter, nfev, njev, info, phase, maxfev, mode, bounded, fonly;
};

#define LMREG_UNROLL _Pragma("unroll")

// On the device every element goes through an empty asm before the select
// chain: left as plain loads, LLVM folds "select of loads" into ONE load from a
// selected address -- a run-time index into the array after all, which pins
// the array (for a member: the whole state) in private memory.  (Found in
// round 4 in the IR of lm_advance_kernel<6, true>: 59 dynamically indexed
// accesses, 912 B of scratch holding the state the "register form" was meant
// to keep in registers.)  Only up to six parameters: from seven on the state
// no longer fits 512 registers and the allocator's spills cost more than the
// indexed private array did (tools/lm_advance_sweep.py, per launch at 50k
// fits: n=6 0.054 -> 0.047 ms, n=8 0.095 -> 0.109 ms).
#if defined(__HIP_DEVICE_COMPILE__)
#define LMREG_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define LMREG_OPAQUE(x) ((void)0)
#endif

// a[idx] / a[idx] = v for a run-time idx in [0, N) without indexing memory
template <int N, class T>
NGMIX_HD T dget(const T (&a)[N], int idx)
{
    T r = a[0];
    if constexpr (N <= 6) {
        LMREG_OPAQUE(r);
        LMREG_UNROLL
        for (int k = 1; k < N; k++) {
            T ak = a[k];
            LMREG_OPAQUE(ak);
            r = (idx == k) ? ak : r;
        }
    } else {
        LMREG_UNROLL
        for (int k = 1; k < N; k++) r = (idx == k) ? a[k] : r;
    }
    return r;
}

template <int N, class T>
NGMIX_HD void dset(T (&a)[N], int idx, T v)
{
    if constexpr (N <= 6) {
        LMREG_UNROLL
        for (int k = 0; k < N; k++) {
            T ak = a[k];
            LMREG_OPAQUE(ak);
            a[k] = (idx == k) ? v : ak;
        }
    } else {
        LMREG_UNROLL
        for (int k = 0; k < N; k++) a[k] = (idx == k) ? v : a[k];
    }
}

template <int N>
NGMIX_HD double enorm(const double (&x)[N])
{
    double s = 0.0;
    LMREG_UNROLL
    for (int i = 0; i < N; i++) s += x[i] * x[i];
    return sqrt(s);
}

// lmcore::factor_normal
template <int N>
NGMIX_HD void factor_normal(const double (&A)[N * N], double (&R)[N * N],
                            int32_t (&ipvt)[N], double (&acnorm)[N])
{
    double S[N * N];
    LMREG_UNROLL
    for (int i = 0; i < N * N; i++) S[i] = A[i];
    LMREG_UNROLL
    for (int j = 0; j < N; j++) {
        ipvt[j] = j;
        const double d = A[j * N + j];
        acnorm[j] = d > 0.0 ? sqrt(d) : 0.0;
    }
    LMREG_UNROLL
    for (int i = 0; i < N * N; i++) R[i] = 0.0;
    LMREG_UNROLL
    for (int k = 0; k < N; k++) {
        // the largest remaining diagonal, first one on ties
        int kmax = k;
        double dmax = S[k * N + k];
        LMREG_UNROLL
        for (int j = k + 1; j < N; j++)
            if (S[j * N + j] > dmax) {
                dmax = S[j * N + j];
                kmax = j;
            }
        LMREG_UNROLL
        for (int m = k + 1; m < N; m++) {
            const bool sw = kmax == m;
            LMREG_UNROLL
            for (int i = 0; i < N; i++) {
                const double a = S[i * N + k], b = S[i * N + m];
                S[i * N + k] = sw ? b : a;
                S[i * N + m] = sw ? a : b;
            }
            LMREG_UNROLL
            for (int j = 0; j < N; j++) {
                const double a = S[k * N + j], b = S[m * N + j];
                S[k * N + j] = sw ? b : a;
                S[m * N + j] = sw ? a : b;
            }
            LMREG_UNROLL
            for (int i = 0; i < k; i++) {
                const double a = R[i * N + k], b = R[i * N + m];
                R[i * N + k] = sw ? b : a;
                R[i * N + m] = sw ? a : b;
            }
            const int32_t a = ipvt[k], b = ipvt[m];
            ipvt[k] = sw ? b : a;
            ipvt[m] = sw ? a : b;
        }
        const double d = S[k * N + k];
        if (!(d > 0.0)) {
            // rank deficient (R beyond row k is still zero)
            LMREG_UNROLL
            for (int j = k; j < N; j++) R[k * N + j] = 0.0;
            return;
        }
        const double rkk = sqrt(d);
        R[k * N + k] = rkk;
        LMREG_UNROLL
        for (int j = k + 1; j < N; j++) R[k * N + j] = S[k * N + j] / rkk;
        LMREG_UNROLL
        for (int i = k + 1; i < N; i++)
            LMREG_UNROLL
            for (int j = i; j < N; j++) {
                const double v = S[i * N + j] - R[k * N + i] * R[k * N + j];
                S[i * N + j] = v;
                S[j * N + i] = v;
            }
    }
}

// lmcore::qtf_from_gradient
template <int N>
NGMIX_HD void qtf_from_gradient(const double (&R)[N * N], const int32_t (&ipvt)[N],
                                const double (&g)[N], double (&qtf)[N])
{
    LMREG_UNROLL
    for (int j = 0; j < N; j++) {
        double s = dget<N>(g, ipvt[j]);
        LMREG_UNROLL
        for (int i = 0; i < j; i++) s -= R[i * N + j] * qtf[i];
        const double rjj = R[j * N + j];
        qtf[j] = rjj != 0.0 ? s / rjj : 0.0;
    }
}

// lmcore::qrsolv
template <int N>
NGMIX_HD void qrsolv(double (&r)[N * N], const int32_t (&ipvt)[N], const double (&diag)[N],
                     const double (&qtb)[N], double (&x)[N], double (&sdiag)[N],
                     double (&wa)[N])
{
    LMREG_UNROLL
    for (int j = 0; j < N; j++) {
        LMREG_UNROLL
        for (int i = j; i < N; i++) r[i * N + j] = r[j * N + i];
        x[j] = r[j * N + j];
        wa[j] = qtb[j];
    }
    LMREG_UNROLL
    for (int j = 0; j < N; j++) {
        const double dl = dget<N>(diag, ipvt[j]);
        if (dl != 0.0) {
            LMREG_UNROLL
            for (int k = j; k < N; k++) sdiag[k] = 0.0;
            sdiag[j] = dl;
            double qtbpj = 0.0;
            LMREG_UNROLL
            for (int k = j; k < N; k++) {
                if (sdiag[k] == 0.0) continue;
                double cs, sn;
                const double rkk = r[k * N + k];
                if (fabs(rkk) < fabs(sdiag[k])) {
                    const double cotan = rkk / sdiag[k];
                    sn = 0.5 / sqrt(0.25 + 0.25 * (cotan * cotan));
                    cs = sn * cotan;
                } else {
                    const double tn = sdiag[k] / rkk;
                    cs = 0.5 / sqrt(0.25 + 0.25 * (tn * tn));
                    sn = cs * tn;
                }
                r[k * N + k] = cs * rkk + sn * sdiag[k];
                double temp = cs * wa[k] + sn * qtbpj;
                qtbpj = -sn * wa[k] + cs * qtbpj;
                wa[k] = temp;
                LMREG_UNROLL
                for (int i = k + 1; i < N; i++) {
                    temp = cs * r[i * N + k] + sn * sdiag[i];
                    sdiag[i] = -sn * r[i * N + k] + cs * sdiag[i];
                    r[i * N + k] = temp;
                }
            }
        }
        sdiag[j] = r[j * N + j];
        r[j * N + j] = x[j];
    }
    int nsing = N;
    LMREG_UNROLL
    for (int j = 0; j < N; j++) {
        if (sdiag[j] == 0.0 && nsing == N) nsing = j;
        if (nsing < N) wa[j] = 0.0;
    }
    // k = 0 .. nsing-1, j = nsing-1-k: j descending from nsing-1
    LMREG_UNROLL
    for (int j = N - 1; j >= 0; j--) {
        if (j < nsing) {
            double sum = 0.0;
            LMREG_UNROLL
            for (int i = j + 1; i < N; i++)
                if (i < nsing) sum += r[i * N + j] * wa[i];
            wa[j] = (wa[j] - sum) / sdiag[j];
        }
    }
    LMREG_UNROLL
    for (int j = 0; j < N; j++) dset<N>(x, ipvt[j], wa[j]);
}

// lmcore::lmpar
template <int N>
NGMIX_HD void lmpar(double (&r)[N * N], const int32_t (&ipvt)[N], const double (&diag)[N],
                    const double (&qtb)[N], double delta, double &par, double (&x)[N],
                    double (&sdiag)[N])
{
    double wa1[N], wa2[N];
    // gauss-newton direction
    int nsing = N;
    LMREG_UNROLL
    for (int j = 0; j < N; j++) {
        wa1[j] = qtb[j];
        if (r[j * N + j] == 0.0 && nsing == N) nsing = j;
        if (nsing < N) wa1[j] = 0.0;
    }
    LMREG_UNROLL
    for (int j = N - 1; j >= 0; j--) {
        if (j < nsing) {
            wa1[j] /= r[j * N + j];
            const double temp = wa1[j];
            LMREG_UNROLL
            for (int i = 0; i < j; i++) wa1[i] -= r[i * N + j] * temp;
        }
    }
    LMREG_UNROLL
    for (int j = 0; j < N; j++) dset<N>(x, ipvt[j], wa1[j]);

    int iter = 0;
    LMREG_UNROLL
    for (int j = 0; j < N; j++) wa2[j] = diag[j] * x[j];
    double dxnorm = enorm<N>(wa2);
    double fp = dxnorm - delta;
    if (fp <= 0.1 * delta) {
        par = 0.0;
        return;
    }
    // lower bound
    double parl = 0.0;
    if (nsing >= N) {
        LMREG_UNROLL
        for (int j = 0; j < N; j++) {
            const int l = ipvt[j];
            wa1[j] = dget<N>(diag, l) * (dget<N>(wa2, l) / dxnorm);
        }
        LMREG_UNROLL
        for (int j = 0; j < N; j++) {
            double sum = 0.0;
            LMREG_UNROLL
            for (int i = 0; i < j; i++) sum += r[i * N + j] * wa1[i];
            wa1[j] = (wa1[j] - sum) / r[j * N + j];
        }
        const double temp = enorm<N>(wa1);
        parl = ((fp / delta) / temp) / temp;
    }
    // upper bound
    LMREG_UNROLL
    for (int j = 0; j < N; j++) {
        double sum = 0.0;
        LMREG_UNROLL
        for (int i = 0; i <= j; i++) sum += r[i * N + j] * qtb[i];
        wa1[j] = sum / dget<N>(diag, ipvt[j]);
    }
    const double gnorm = enorm<N>(wa1);
    double paru = gnorm / delta;
    if (paru == 0.0) paru = DWARF / fmin(delta, 0.1);
    par = fmax(par, parl);
    par = fmin(par, paru);
    if (par == 0.0) par = gnorm / dxnorm;

    for (;;) {
        iter++;
        if (par == 0.0) par = fmax(DWARF, 0.001 * paru);
        double temp = sqrt(par);
        LMREG_UNROLL
        for (int j = 0; j < N; j++) wa1[j] = temp * diag[j];
        qrsolv<N>(r, ipvt, wa1, qtb, x, sdiag, wa2);
        LMREG_UNROLL
        for (int j = 0; j < N; j++) wa2[j] = diag[j] * x[j];
        dxnorm = enorm<N>(wa2);
        temp = fp;
        fp = dxnorm - delta;
        if (fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) ||
            iter == 10)
            break;
        // newton correction
        LMREG_UNROLL
        for (int j = 0; j < N; j++) {
            const int l = ipvt[j];
            wa1[j] = dget<N>(diag, l) * (dget<N>(wa2, l) / dxnorm);
        }
        LMREG_UNROLL
        for (int j = 0; j < N; j++) {
            wa1[j] /= sdiag[j];
            temp = wa1[j];
            LMREG_UNROLL
            for (int i = j + 1; i < N; i++) wa1[i] -= r[i * N + j] * temp;
        }
        temp = enorm<N>(wa1);
        const double parc = ((fp / delta) / temp) / temp;
        if (fp > 0.0) parl = fmax(parl, par);
        if (fp < 0.0) paru = fmin(paru, par);
        par = fmax(parl, par + parc);
    }
}

// lmcore::set_trial
template <int N>
NGMIX_HD void set_trial(lm_state_n<N> &s)
{
    constexpr double EPS = 1.4901161193847656e-08;  // sqrt(machine epsilon)
    LMREG_UNROLL
    for (int j = 0; j < N; j++) {
        s.xt[j] = s.bounded ? lmcore::i2e(s.xti[j], s.lo[j], s.hi[j]) : s.xti[j];
        if (s.mode == NGMIX_LM_MODE_FD) {
            double h = EPS * fabs(s.xti[j]);
            if (h == 0.0) h = EPS;
            s.hstep[j] = h;
            s.xstep[j] =
                s.bounded ? lmcore::i2e(s.xti[j] + h, s.lo[j], s.hi[j]) : s.xti[j] + h;
        }
    }
}

// lmcore::propose
template <int N>
NGMIX_HD void propose(lm_state_n<N> &s)
{
    double r[N * N], sdiag[N], p[N], wa3[N];
    LMREG_UNROLL
    for (int i = 0; i < N * N; i++) r[i] = s.R[i];
    // (lmpar reads p only after writing all of it through ipvt, a permutation)
    LMREG_UNROLL
    for (int j = 0; j < N; j++) {
        p[j] = 0.0;
        sdiag[j] = 0.0;
    }
    lmpar<N>(r, s.ipvt, s.diag, s.qtf, s.delta, s.par, p, sdiag);
    LMREG_UNROLL
    for (int j = 0; j < N; j++) {
        s.step[j] = -p[j];
        s.xti[j] = s.xi[j] + s.step[j];
        wa3[j] = s.diag[j] * s.step[j];
    }
    set_trial<N>(s);
    s.pnorm = enorm<N>(wa3);
    if (s.iter == 1) s.delta = fmin(s.delta, s.pnorm);
    s.fonly = 0;
    if (s.mode == NGMIX_LM_MODE_ANALYTIC_LAZY) {
        // (lmcore::propose: will accepting this trial end the fit?)
        double w3[N];
        LMREG_UNROLL
        for (int j = 0; j < N; j++) w3[j] = 0.0;
        LMREG_UNROLL
        for (int j = 0; j < N; j++) {
            const double temp = dget<N>(s.step, s.ipvt[j]);
            LMREG_UNROLL
            for (int i = 0; i <= j; i++) w3[i] += s.R[i * N + j] * temp;
        }
        const double temp1 = enorm<N>(w3) / s.fnorm;
        const double temp2 = (sqrt(s.par) * s.pnorm) / s.fnorm;
        const double prered = temp1 * temp1 + temp2 * temp2 / 0.5;
        if (prered <= s.ftol || s.pnorm / 0.5 <= s.xtol * s.xnorm) s.fonly = 1;
    }
}

// lmcore::new_jacobian
template <int N>
NGMIX_HD bool new_jacobian(lm_state_n<N> &s, const double (&A)[N * N], const double (&g)[N])
{
    double acnorm[N];
    s.njev++;
    factor_normal<N>(A, s.R, s.ipvt, acnorm);
    if (s.iter == 1) {
        double wa3[N];
        LMREG_UNROLL
        for (int j = 0; j < N; j++) {
            s.diag[j] = acnorm[j];
            if (acnorm[j] == 0.0) s.diag[j] = 1.0;
            wa3[j] = s.diag[j] * s.xi[j];
        }
        s.xnorm = enorm<N>(wa3);
        s.delta = s.factor * s.xnorm;
        if (s.delta == 0.0) s.delta = s.factor;
    }
    qtf_from_gradient<N>(s.R, s.ipvt, g, s.qtf);
    // norm of the scaled gradient
    s.gnorm = 0.0;
    if (s.fnorm != 0.0) {
        LMREG_UNROLL
        for (int j = 0; j < N; j++) {
            const double acl = dget<N>(acnorm, s.ipvt[j]);
            if (acl == 0.0) continue;
            double sum = 0.0;
            LMREG_UNROLL
            for (int i = 0; i <= j; i++) sum += s.R[i * N + j] * (s.qtf[i] / s.fnorm);
            s.gnorm = fmax(s.gnorm, fabs(sum / acl));
        }
    }
    if (s.gnorm <= s.gtol) {
        s.info = 4;
        s.phase = LM_PHASE_DONE;
        return true;
    }
    LMREG_UNROLL
    for (int j = 0; j < N; j++) s.diag[j] = fmax(s.diag[j], acnorm[j]);
    return false;   // the caller proposes (ONE copy of lmpar / qrsolv in the code)
}

// lmcore::lm_advance for a fit of exactly N parameters (s.n == N)
template <int N>
NGMIX_HD void lm_advance(lm_state_n<N> &s, double ff, const double (&g_in)[N],
                         const double (&A_in)[N * N])
{
    if (s.phase == LM_PHASE_DONE) return;
    double g[N], A[N * N];
    if (s.bounded && s.mode != NGMIX_LM_MODE_FD) {
        double sc[N];
        LMREG_UNROLL
        for (int j = 0; j < N; j++) sc[j] = lmcore::i2e_grad(s.xti[j], s.lo[j], s.hi[j]);
        LMREG_UNROLL
        for (int j = 0; j < N; j++) {
            g[j] = g_in[j] * sc[j];
            LMREG_UNROLL
            for (int k = 0; k < N; k++) A[j * N + k] = A_in[j * N + k] * sc[j] * sc[k];
        }
    } else {
        LMREG_UNROLL
        for (int j = 0; j < N; j++) g[j] = g_in[j];
        LMREG_UNROLL
        for (int i = 0; i < N * N; i++) A[i] = A_in[i];
    }
    // The three places lmder factors a new jacobian (and the two where it
    // proposes a step) set a flag and meet at ONE call site each at the end:
    // inlined at every site the step was 45,000 instructions for six parameters,
    // four copies of lmpar / qrsolv that no instruction cache holds.
    bool want_jacobian = false, want_proposal = false;
    if (s.phase == LM_PHASE_JAC) {
        if (s.mode == NGMIX_LM_MODE_FD) s.nfev += N;
        s.phase = LM_PHASE_TRIAL;
        want_jacobian = true;
    } else if (s.phase == LM_PHASE_INIT) {
        s.nfev = s.mode == NGMIX_LM_MODE_FD ? 1 + N : 1;
        s.fnorm = sqrt(ff);
        s.par = 0.0;
        s.iter = 1;
        if (!(s.fnorm < INFINITY)) {
            s.njev = 1;
            s.info = 4;
            s.phase = LM_PHASE_DONE;
            return;
        }
        s.phase = LM_PHASE_TRIAL;
        want_jacobian = true;
    } else {

    // ---- LM_PHASE_TRIAL: the rest of lmder's inner loop
    s.nfev++;
    const double fnorm1 = ff < INFINITY ? sqrt(ff) : NAN;
    double actred = -1.0;
    if (0.1 * fnorm1 < s.fnorm) {
        const double t = fnorm1 / s.fnorm;
        actred = 1.0 - t * t;
    }
    double wa3[N];
    LMREG_UNROLL
    for (int j = 0; j < N; j++) wa3[j] = 0.0;
    LMREG_UNROLL
    for (int j = 0; j < N; j++) {
        const double temp = dget<N>(s.step, s.ipvt[j]);
        LMREG_UNROLL
        for (int i = 0; i <= j; i++) wa3[i] += s.R[i * N + j] * temp;
    }
    const double temp1 = enorm<N>(wa3) / s.fnorm;
    const double temp2 = (sqrt(s.par) * s.pnorm) / s.fnorm;
    const double prered = temp1 * temp1 + temp2 * temp2 / 0.5;
    const double dirder = -(temp1 * temp1 + temp2 * temp2);
    double ratio = 0.0;
    if (prered != 0.0) ratio = actred / prered;
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
        double w[N];
        LMREG_UNROLL
        for (int j = 0; j < N; j++) {
            s.x[j] = s.xt[j];
            s.xi[j] = s.xti[j];
            w[j] = s.diag[j] * s.xi[j];
        }
        s.xnorm = enorm<N>(w);
        s.fnorm = fnorm1;
        s.iter++;
    }
    int info = 0;
    if (fabs(actred) <= s.ftol && prered <= s.ftol && 0.5 * ratio <= 1.0) info = 1;
    if (s.delta <= s.xtol * s.xnorm) info = 2;
    if (fabs(actred) <= s.ftol && prered <= s.ftol && 0.5 * ratio <= 1.0 && info == 2)
        info = 3;
    if (info == 0) {
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
        want_proposal = true;   // same factor, smaller region
    } else if (s.mode == NGMIX_LM_MODE_FD || s.fonly) {
        LMREG_UNROLL
        for (int j = 0; j < N; j++) s.xti[j] = s.xi[j];
        set_trial<N>(s);
        s.phase = LM_PHASE_JAC;
        s.fonly = 0;
    } else {
        want_jacobian = true;   // the trial point's jacobian is the new one
    }
    }
    if (want_jacobian) want_proposal = !new_jacobian<N>(s, A, g);
    if (want_proposal) propose<N>(s);
}

// between the ngmix_lm_state record (arrays of LM_NPMAX) and the compact state
template <int N>
NGMIX_HD void load_state(lm_state_n<N> &d, lm_state &g)
{
    d.n = g.n;
    d.iter = g.iter;
    d.nfev = g.nfev;
    d.njev = g.njev;
    d.info = g.info;
    d.phase = g.phase;
    d.maxfev = g.maxfev;
    d.mode = g.mode;
    d.bounded = g.bounded;
    d.fonly = g.fonly;
    d.fnorm = g.fnorm;
    d.xnorm = g.xnorm;
    d.delta = g.delta;
    d.par = g.par;
    d.gnorm = g.gnorm;
    d.pnorm = g.pnorm;
    d.ftol = g.ftol;
    d.xtol = g.xtol;
    d.gtol = g.gtol;
    d.factor = g.factor;
    d.x = g.x;
    d.xstep = g.xstep;
    d.hstep = g.hstep;
    d.lo = g.lo;
    d.hi = g.hi;
    LMREG_UNROLL
    for (int j = 0; j < N; j++) {
        d.xt[j] = g.xt[j];
        d.diag[j] = g.diag[j];
        d.qtf[j] = g.qtf[j];
        d.step[j] = g.step[j];
        d.xi[j] = g.xi[j];
        d.xti[j] = g.xti[j];
        d.ipvt[j] = g.ipvt[j];
        LMREG_UNROLL
        for (int k = 0; k < N; k++) d.R[j * N + k] = g.R[j * LM_NPMAX + k];
    }
}

// (x, xstep, hstep were written in place)
template <int N>
NGMIX_HD void store_state(lm_state &d, const lm_state_n<N> &g)
{
    d.iter = g.iter;
    d.nfev = g.nfev;
    d.njev = g.njev;
    d.info = g.info;
    d.phase = g.phase;
    d.fonly = g.fonly;
    d.fnorm = g.fnorm;
    d.xnorm = g.xnorm;
    d.delta = g.delta;
    d.par = g.par;
    d.gnorm = g.gnorm;
    d.pnorm = g.pnorm;
    LMREG_UNROLL
    for (int j = 0; j < N; j++) {
        d.xt[j] = g.xt[j];
        d.diag[j] = g.diag[j];
        d.qtf[j] = g.qtf[j];
        d.step[j] = g.step[j];
        d.xi[j] = g.xi[j];
        d.xti[j] = g.xti[j];
        d.ipvt[j] = g.ipvt[j];
        LMREG_UNROLL
        for (int k = 0; k < N; k++) d.R[j * LM_NPMAX + k] = g.R[j * N + k];
    }
}

}  // namespace lmreg
