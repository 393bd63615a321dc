// lm_precise.hip -- the covariance factor of the ill-conditioned forward-
// difference fits, from DOUBLE-DOUBLE normal equations.
//
// The reference's covariance is scipy.optimize.leastsq's: cov_x = inv(R^T R)
// from MINPACK's pivoted QR of the LAST jacobian (fjac / ipvt), flagged
// LM_SINGULAR_MATRIX when that inverse does not exist
// (ngmix/fitting/leastsqbound.py:76-118, 535-552).  The lock-step driver
// iterates on the normal equations: its R is the pivoted Cholesky factor of
// J^T J, equal to MINPACK's to the rounding of cond(J)^2 instead of cond(J).
// For the co-elliptical psf fits with three and more gaussians cond(J)
// reaches 1e8 at the solution (the gaussians of a psf wing are nearly
// exchangeable): J^T J formed and factored in doubles is then not numerically
// positive definite, the factorisation stops at a non-positive pivot and the
// fit is flagged singular where MINPACK returns a covariance -- 6 % / 15 % / 23 %
// of such fits with 3 / 4 / 5 gaussians (profiles/r06_fuzz_lm_fd_vs_minpack.log).
//
// After the rounds have ended, ngmix_lm_precise_cov_batch therefore re-makes
// R and ipvt of those fits:
//
//   1. lm_eval_fd_kernel<NLOC, LINEAR, PRECISE> (lmfit.hip) evaluates the
//      forward-difference jacobian once more at the point of the fit's last
//      jacobian (jac_point: recorded by the ordinary passes) and accumulates
//      J^T J in double-double -- exact to ~1e-32 of each sum;
//   2. lm_factor_dd_kernel below folds the stamps' sums into the object's matrix and runs factor_normal's pivoted Cholesky
//      (lm_core.hpp: qrfac's pivot rule) in double-double arithmetic, one wave
//      per fit, the matrices in LDS; R rounded to doubles and ipvt go into the
//      state record, where ngmix_lm_finalize_batch reads them.
//
// The factor is then as accurate as a QR of J in doubles (better: its error is
// cond(J)^2 1e-32), for cond(J) up to ~1e15.  The ITERATION is untouched: its
// steps come from the double factorisation as before (they agree with MINPACK's
// in nfev for most fits and end at the same minimum for the rest).
#include <stdio.h>
#include <string.h>

#include "device_utils.hpp"
#include "launch.hpp"
#include "lm_core.hpp"

namespace ngmix {

namespace dd {

struct dd_t {
    double hi, lo;
};

__device__ __forceinline__ dd_t two_sum(double a, double b)
{
    const double s = a + b;
    const double bb = s - a;
    return {s, (a - (s - bb)) + (b - bb)};
}

__device__ __forceinline__ dd_t fast_two_sum(double a, double b)   // |a| >= |b|
{
    const double s = a + b;
    return {s, b - (s - a)};
}

__device__ __forceinline__ dd_t two_prod(double a, double b)
{
    const double p = a * b;
    return {p, __builtin_fma(a, b, -p)};
}

__device__ __forceinline__ dd_t add(dd_t a, dd_t b)
{
    dd_t s = two_sum(a.hi, b.hi);
    const dd_t t = two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = fast_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return fast_two_sum(s.hi, s.lo);
}

__device__ __forceinline__ dd_t neg(dd_t a) { return {-a.hi, -a.lo}; }

__device__ __forceinline__ dd_t mul(dd_t a, dd_t b)
{
    dd_t p = two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return fast_two_sum(p.hi, p.lo);
}

__device__ __forceinline__ dd_t mul_d(dd_t a, double b)
{
    dd_t p = two_prod(a.hi, b);
    p.lo += a.lo * b;
    return fast_two_sum(p.hi, p.lo);
}

// a / b: three quotient digits (the QD library's accurate division)
__device__ __forceinline__ dd_t div(dd_t a, dd_t b)
{
    const double q1 = a.hi / b.hi;
    dd_t r = add(a, neg(mul_d(b, q1)));
    const double q2 = r.hi / b.hi;
    r = add(r, neg(mul_d(b, q2)));
    const double q3 = r.hi / b.hi;
    dd_t q = fast_two_sum(q1, q2);
    return add(q, {q3, 0.0});
}

// sqrt(a), a > 0: one Newton step from the double root (Karp & Markstein)
__device__ __forceinline__ dd_t sqrt_dd(dd_t a)
{
    const double x = 1.0 / sqrt(a.hi);
    const double ax = a.hi * x;
    const dd_t sq = two_prod(ax, ax);
    const double corr = add(a, neg(sq)).hi * (x * 0.5);
    return fast_two_sum(ax, corr);
}

}  // namespace dd

// One wave per fit.  LDS: S (the matrix, then its Schur complements) and R,
// each as a plane of high and a plane of low parts, LM_NPMAX-strided.
__global__ __launch_bounds__(WAVE) void lm_factor_dd_kernel(
    lm_state *states, int64_t nobj, const int64_t *__restrict__ obj_start,
    const int32_t *__restrict__ stamp_band, const double *__restrict__ psums, int nloc)
{
    using dd::dd_t;
    constexpr int NP = LM_NPMAX, NN = NP * NP;
    __shared__ double Sh[NN], Sl[NN], Rh[NN], Rl[NN];
    __shared__ int32_t piv[NP];
    __shared__ int bad_flag;
    const int64_t o = blockIdx.x;
    const int lane = threadIdx.x;
    lm_state &G = states[o];
    if (G.phase != LM_PHASE_DONE || G.info < 1 || G.info > 4 ||
        G.mode != NGMIX_LM_MODE_FD)
        return;
    const int n = G.n;
    if (n < nloc || n > NP) return;
    for (int i = lane; i < NN; i += WAVE) Sh[i] = Sl[i] = Rh[i] = Rl[i] = 0.0;
    if (lane < NP) piv[lane] = lane;
    if (lane == 0) bad_flag = 0;
    __syncthreads();

    // ---- A = sum over the object's stamps (stamp order, entry by entry)
    const int ntri = nloc * (nloc + 1) / 2, nsum = ntri + nloc + 1;
    const int64_t s0 = obj_start ? obj_start[o] : o;
    const int64_t s1 = obj_start ? obj_start[o + 1] : o + 1;
    for (int k = lane; k < ntri; k += WAVE) {
        int a = 0, row = 0;
        while (row + (nloc - a) <= k) {
            row += nloc - a;
            a++;
        }
        const int b = a + (k - row);
        // (entries that involve the flux land in a different element per band:
        // an element still receives its stamps in stamp order)
        for (int64_t st = s0; st < s1; st++) {
            const int band = stamp_band ? stamp_band[st] : 0;
            const int ga = a < nloc - 1 ? a : nloc - 1 + band;
            const int gb = b < nloc - 1 ? b : nloc - 1 + band;
            if (ga >= n || gb >= n) {
                bad_flag = 1;
                continue;
            }
            const double *v = psums + st * 2 * (int64_t)nsum;
            if (!(fabs(v[nsum - 1]) < INFINITY)) bad_flag = 1;   // an out-of-range pass
            const int e = ga * NP + gb;
            const dd_t t = dd::add({Sh[e], Sl[e]}, {v[k], v[nsum + k]});
            Sh[e] = t.hi;
            Sl[e] = t.lo;
        }
    }
    __syncthreads();
    if (bad_flag) return;   // (the state keeps the factor the iteration left)
    // the lower triangle
    for (int e = lane; e < NN; e += WAVE) {
        const int i = e / NP, j = e % NP;
        if (i > j && i < n) {
            Sh[e] = Sh[j * NP + i];
            Sl[e] = Sl[j * NP + i];
        }
    }
    __syncthreads();

    // ---- lmcore::factor_normal in double-double
    for (int k = 0; k < n; k++) {
        // qrfac's pivot rule: the largest remaining diagonal element, the first
        // one on ties (every lane scans the same LDS values)
        int kmax = k;
        double best = Sh[k * NP + k], bestl = Sl[k * NP + k];
        for (int j = k + 1; j < n; j++) {
            const double h = Sh[j * NP + j], l = Sl[j * NP + j];
            if (h > best || (h == best && l > bestl)) {
                best = h;
                bestl = l;
                kmax = j;
            }
        }
        __syncthreads();
        if (kmax != k) {
            if (lane < n) {   // columns k <-> kmax of S, of R above row k
                const int i = lane;
                double t = Sh[i * NP + k];
                Sh[i * NP + k] = Sh[i * NP + kmax];
                Sh[i * NP + kmax] = t;
                t = Sl[i * NP + k];
                Sl[i * NP + k] = Sl[i * NP + kmax];
                Sl[i * NP + kmax] = t;
                if (i < k) {
                    t = Rh[i * NP + k];
                    Rh[i * NP + k] = Rh[i * NP + kmax];
                    Rh[i * NP + kmax] = t;
                    t = Rl[i * NP + k];
                    Rl[i * NP + k] = Rl[i * NP + kmax];
                    Rl[i * NP + kmax] = t;
                }
            }
            __syncthreads();
            if (lane < n) {   // rows k <-> kmax of S
                const int j = lane;
                double t = Sh[k * NP + j];
                Sh[k * NP + j] = Sh[kmax * NP + j];
                Sh[kmax * NP + j] = t;
                t = Sl[k * NP + j];
                Sl[k * NP + j] = Sl[kmax * NP + j];
                Sl[kmax * NP + j] = t;
            }
            if (lane == 0) {
                const int32_t t = piv[k];
                piv[k] = piv[kmax];
                piv[kmax] = t;
            }
            __syncthreads();
        }
        const dd_t d = {Sh[k * NP + k], Sl[k * NP + k]};
        if (!(d.hi > 0.0)) {
            // rank deficient beyond double-double: the rows from k on stay zero,
            // as factor_normal (and qrfac's rdiag = 0) leave them
            break;
        }
        const dd_t rkk = dd::sqrt_dd(d);
        if (lane == 0) {
            Rh[k * NP + k] = rkk.hi;
            Rl[k * NP + k] = rkk.lo;
        }
        if (lane > k && lane < n) {
            const int j = lane;
            const dd_t r = dd::div({Sh[k * NP + j], Sl[k * NP + j]}, rkk);
            Rh[k * NP + j] = r.hi;
            Rl[k * NP + j] = r.lo;
        }
        __syncthreads();
        // S[i][j] -= R[k][i] R[k][j] for k < i <= j < n, mirrored
        for (int e = lane; e < NN; e += WAVE) {
            const int i = e / NP, j = e % NP;
            if (i > k && j >= i && j < n) {
                const dd_t pr = dd::mul({Rh[k * NP + i], Rl[k * NP + i]},
                                        {Rh[k * NP + j], Rl[k * NP + j]});
                const dd_t v = dd::add({Sh[e], Sl[e]}, dd::neg(pr));
                Sh[e] = v.hi;
                Sl[e] = v.lo;
                Sh[j * NP + i] = v.hi;
                Sl[j * NP + i] = v.lo;
            }
        }
        __syncthreads();
    }
    __syncthreads();
    // ---- into the record: R in doubles, the pivot order
    for (int e = lane; e < NN; e += WAVE) {
        const int i = e / NP, j = e % NP;
        if (i < n && j < n) G.R[e] = j >= i ? Rh[e] + Rl[e] : 0.0;
    }
    if (lane < n) G.ipvt[lane] = piv[lane];
}

int launch_lm_precise_cov(const ngmix_lm_problem *p, double *psums, hipStream_t s)
{
    if (!p || !p->batch || !p->states || !psums || !p->jac_point || !p->fd) {
        set_last_error_msg("lm_precise_cov: a forward-difference problem with jac_point "
                           "and a (nstamps, 2, NSUMS(nloc)) workspace is required");
        return NGMIX_ERR_BAD_ARG;
    }
    if (p->nobj <= 0 || p->batch->nstamps <= 0) return NGMIX_OK;
    const int nloc = p->nloc_npars & 0xff;
    if (nloc < NGMIX_LM_PRECISE_MIN_NLOC || nloc > LM_NPMAX) {
        set_last_error_msg("lm_precise_cov: serves fits of 10-14 local parameters");
        return NGMIX_ERR_BAD_ARG;
    }
    if (p->prior) {
        // (the kernel prior serves models with five shape parameters, nloc = 6:
        // no fit this pass serves carries one)
        set_last_error_msg("lm_precise_cov: fits with a kernel prior are not served");
        return NGMIX_ERR_BAD_ARG;
    }
    int rc = launch_lm_eval(p->batch, p->model, 1, p->states, p->stamp_obj, p->stamp_band,
                            p->psf, p->npsf, psums, nullptr, nullptr, s, p->jac_point, true);
    if (rc != NGMIX_OK) return rc;
    census("lm_factor_dd_kernel");
    hipLaunchKernelGGL(lm_factor_dd_kernel, dim3((unsigned)p->nobj), dim3(WAVE), 0, s,
                       p->states, p->nobj, p->obj_start, p->stamp_band, psums, nloc);
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

}  // namespace ngmix
