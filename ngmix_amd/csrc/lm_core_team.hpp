// lm_core_team.hpp -- lm_core.hpp's lmder step for a TEAM of 16 lanes per fit.
//
// One thread per fit (lm_core.hpp's generic form in the device kernel) keeps a
// fit's work arrays -- two n x n matrices and a dozen vectors, indexed at run
// time by MINPACK's pivot order -- in private memory: a step is a serial chain
// of ~2,000 dependent scratch accesses, 0.5-2.3 ms per launch for fits of
// 11-14 parameters however few fits there are (multi-band fits, co-elliptical
// psf fits with 4 / 5 gaussians).  The register form (lm_core_reg.hpp) removes
// the memory latency for 6-8 parameters, spills at 9 and 10 and does not fit
// beyond; this form serves 9-14.
//
// Here a fit belongs to 16 lanes and its arrays live in LDS:
//
//   * every lane runs lmder's SCALAR logic -- pivot searches, Givens
//     coefficients, the sums whose order of addition matters (enorm, the
//     triangular solves), every branch -- redundantly on the same LDS values:
//     all lanes of a team hold the same scalars to the bit and take the same
//     branches, with no broadcast step;
//   * the loops over independent elements (a rotation applied to the rows
//     below the pivot, the rank-1 update of the factorisation, row / column
//     swaps, the element-wise vector updates, the fold of the stamps' sums)
//     run one element per lane;
//   * the vectors that are only ever touched element-wise (x, xt, xi, xti, lo,
//     hi, xstep, hstep) are one register per lane.
//
// What a step costs is its number of DEPENDENT LDS round trips (~100 cycles
// each), not its arithmetic.  A sum over a row, a column or a vector therefore
// fetches all its operands with one batch of loads -- NP of them, NP the
// compile-time bound of the parameter count, the elements outside the sum's
// range replaced by +0.0 -- and adds in registers: a term (+0.0) * (+0.0) added
// to (or subtracted from) a sum that started at +0.0 or at a finite value
// changes no bit of it, so the sum is the serial loop's.
//
// The operations on every element and the order of every sum are lm_core.hpp's,
// statement for statement (and the build has no FMA contraction), so a state
// record after a step is BYTE-IDENTICAL to the generic form's:
// tests/test_gpu_lm_team.py compares the records of the two forms after every
// round; nfev / ier parity with MINPACK and the reference is therefore untouched.
//
// A team's lanes communicate through LDS only.  LDS instructions of one wave
// execute in order, so a phase boundary needs no hardware barrier: tsync() is a
// compiler fence (no memory operation may move across it) and a wave barrier.
// Within a phase no lane reads an element another lane writes.
#pragma once

#include "lm_core.hpp"

namespace lmteam {

constexpr int TEAM = 16;

__device__ __forceinline__ void tsync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

// the scalars of a fit: one copy per lane, identical across the team
struct Scal {
    double fnorm, xnorm, delta, par, gnorm, pnorm;
    double ftol, xtol, gtol, factor;
    int32_t n, iter, nfev, njev, info, phase, maxfev, mode, bounded, fonly;
};

// a team's view of its fit
struct Fit {
    int lane;       // 0 .. 15; lane j owns element j of every vector
    int ld;         // stride of the matrices (odd: 16 lanes walking a column hit 16 banks)
    // LDS
    double *R;      // the pivoted factor (state)
    double *M;      // A = J^T J on entry, S inside factor_normal, r inside lmpar
    double *diag, *qtf, *step, *g, *acnorm, *sdiag, *p, *wa1, *wa2, *wa3;
    int32_t *ipvt;
    // one element per lane (lane >= n: unused)
    double x, xt, xi, xti, lo, hi, xstep, hstep;
    Scal s;
};

// R(i, j), j >= i, of a team's factor: the upper triangle packed row-major
// (row i holds columns i .. np - 1) -- R[tri_row(np, i) + j].  Half the
// matrix's LDS: the fits in flight per CU are bound by LDS (4 teams per wave).
__host__ __device__ constexpr int tri_row(int np, int i)
{
    return i * (np - 1) - i * (i - 1) / 2;
}

// doubles of LDS per team for fits of up to np parameters: the work matrix M
// (np x ld), the packed factor R, nine vectors (g shares its storage with sdiag
// and acnorm with p: the first of each pair is dead before lmpar, the only user
// of the second, starts -- see Step::new_jacobian / propose)
__host__ __device__ inline int team_lds_doubles(int np)
{
    const int ld = np | 1;
    return np * ld + np * (np + 1) / 2 + 9 * np;
}

__device__ __forceinline__ void carve(Fit &f, double *base, int np)
{
    f.ld = np | 1;
    f.M = base;
    f.R = f.M + np * f.ld;
    f.diag = f.R + np * (np + 1) / 2;
    f.qtf = f.diag + np;
    f.step = f.qtf + np;
    f.g = f.step + np;
    f.acnorm = f.g + np;
    f.sdiag = f.g;        // (g is consumed by qtf_from_gradient before lmpar runs)
    f.p = f.acnorm;       // (acnorm's last reader is new_jacobian, before propose)
    f.wa1 = f.acnorm + np;
    f.wa2 = f.wa1 + np;
    f.wa3 = f.wa2 + np;
    f.ipvt = (int32_t *)(f.wa3 + np);
}

// "for (int i = lo; i < hi; i++)" with one element per lane (hi - lo <= 16)
#define TFOR(i, lo_, hi_) for (int i = (lo_) + f.lane, i##_once = 1; i##_once && i < (hi_); i##_once = 0)
#define LEAD if (f.lane == 0)

// The step for fits of up to NP parameters.  NP bounds the batches of loads
// (and the registers that hold them), not the arithmetic: the loops run to n.
template <int NP>
struct Step {
    // the stride of the matrices: a compile-time constant, so that every load of
    // a batch is one ds_read with an immediate offset
    static constexpr int LD = NP | 1;

    // v[i] = p[i * st] for lo <= i < hi, +0.0 elsewhere: one batch of loads.
    // The loads are unconditional (every index below NP is inside the team's
    // block, which is sized for NP) and the selection is two v_cndmask: no
    // branch per element.
    static __device__ __forceinline__ void gather(double (&v)[NP], const double *p, int st,
                                                  int lo, int hi)
    {
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const double t = p[i * st];
            v[i] = (i >= lo && i < hi) ? t : 0.0;
        }
    }

    // sum_{lo <= i < hi} a[i * sa] * b[i * sb], added in index order from +0.0
    static __device__ __forceinline__ double dot(const double *a, int sa, const double *b,
                                                 int sb, int lo, int hi)
    {
        double av[NP], bv[NP];
        gather(av, a, sa, lo, hi);
        gather(bv, b, sb, lo, hi);
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < NP; i++) s += av[i] * bv[i];
        return s;
    }

    // s - sum_{lo <= i < hi} a[i * sa] * b[i * sb], subtracted in index order
    static __device__ __forceinline__ double subdot(double s, const double *a, int sa,
                                                    const double *b, int sb, int lo, int hi)
    {
        double av[NP], bv[NP];
        gather(av, a, sa, lo, hi);
        gather(bv, b, sb, lo, hi);
#pragma unroll
        for (int i = 0; i < NP; i++) s -= av[i] * bv[i];
        return s;
    }

    // rows lo <= i < hi of COLUMN j of the packed factor (R(i, j), i <= j), +0.0
    // elsewhere: one batch of loads with immediate offsets (tri_row is a
    // compile-time constant per unrolled i; an address past the column's end
    // falls inside an earlier row of the triangle: read and discarded)
    static __device__ __forceinline__ void gather_col(double (&v)[NP], const double *R, int j,
                                                      int lo, int hi)
    {
        const double *c = R + j;
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const double t = c[tri_row(NP, i)];
            v[i] = (i >= lo && i < hi) ? t : 0.0;
        }
    }

    // dot / subdot with a column of the packed factor as the first operand
    static __device__ __forceinline__ double dot_col(const double *R, int j, const double *b,
                                                     int lo, int hi)
    {
        double av[NP], bv[NP];
        gather_col(av, R, j, lo, hi);
        gather(bv, b, 1, lo, hi);
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < NP; i++) s += av[i] * bv[i];
        return s;
    }

    static __device__ __forceinline__ double subdot_col(double s, const double *R, int j,
                                                        const double *b, int lo, int hi)
    {
        double av[NP], bv[NP];
        gather_col(av, R, j, lo, hi);
        gather(bv, b, 1, lo, hi);
#pragma unroll
        for (int i = 0; i < NP; i++) s -= av[i] * bv[i];
        return s;
    }

    // lmcore::enorm on an LDS vector: every lane adds in index order
    static __device__ __forceinline__ double enorm(int n, const double *x)
    {
        return sqrt(dot(x, 1, x, 1, 0, n));
    }

    // lmcore::factor_normal.  M holds A on entry and is destroyed (the serial
    // code's copy S); R, ipvt, acnorm are written.
    static __device__ __forceinline__ void factor_normal(Fit &f)
    {
        const int n = f.s.n;
        constexpr int ld = LD;
        double *S = f.M, *R = f.R;
        TFOR(j, 0, n) {
            f.ipvt[j] = j;
            const double d = S[j * ld + j];
            f.acnorm[j] = d > 0.0 ? sqrt(d) : 0.0;
#pragma unroll
            for (int k = 0; k < NP; k++)
                if (k >= j && k < n) R[tri_row(NP, j) + k] = 0.0;
        }
        tsync();
        for (int k = 0; k < n; k++) {
            // the largest remaining diagonal element, first one on ties (qrfac's
            // pivot rule): the diagonal in one batch, the scan in registers
            double dg[NP];
            gather(dg, S, ld + 1, k, n);
            int kmax = k;
            double d = 0.0;
#pragma unroll
            for (int j = 0; j < NP; j++) {
                if (j == k) {
                    d = dg[j];
                } else if (j > k && j < n && dg[j] > d) {
                    d = dg[j];
                    kmax = j;
                }
            }
            if (kmax != k) {
                tsync();
                TFOR(i, 0, n) {          // columns k <-> kmax of S; of R above row k
                    const double t = S[i * ld + k], u = S[i * ld + kmax];
                    S[i * ld + k] = u;
                    S[i * ld + kmax] = t;
                    if (i < k) {
                        const int ri = tri_row(NP, i);
                        const double a = R[ri + k], b = R[ri + kmax];
                        R[ri + k] = b;
                        R[ri + kmax] = a;
                    }
                }
                tsync();
                TFOR(j, 0, n) {          // rows k <-> kmax of S
                    const double t = S[k * ld + j], u = S[kmax * ld + j];
                    S[k * ld + j] = u;
                    S[kmax * ld + j] = t;
                }
                LEAD {
                    const int32_t ti = f.ipvt[k], tj = f.ipvt[kmax];
                    f.ipvt[k] = tj;
                    f.ipvt[kmax] = ti;
                }
                tsync();
            }
            // (d = S[k][k] after the swap: the element the search picked)
            if (!(d > 0.0)) {
                tsync();
                TFOR(kk, k, n) {
#pragma unroll
                    for (int j = 0; j < NP; j++)
                        if (j >= kk && j < n) R[tri_row(NP, kk) + j] = 0.0;
                }
                tsync();
                return;
            }
            const double rkk = sqrt(d);
            tsync();
            const int rk0 = tri_row(NP, k);
            LEAD R[rk0 + k] = rkk;
            TFOR(j, k + 1, n) R[rk0 + j] = S[k * ld + j] / rkk;
            tsync();
            TFOR(i, k + 1, n) {
                const double rki = R[rk0 + i];
                double sr[NP], rk[NP];
                gather(sr, S + i * ld, 1, i, n);
                gather(rk, R + rk0, 1, i, n);
#pragma unroll
                for (int j = 0; j < NP; j++) {
                    if (j >= i && j < n) {
                        const double v = sr[j] - rki * rk[j];
                        S[i * ld + j] = v;
                        S[j * ld + i] = v;
                    }
                }
            }
            tsync();
        }
    }

    // lmcore::qtf_from_gradient
    static __device__ __forceinline__ void qtf_from_gradient(Fit &f)
    {
        const int n = f.s.n;
        // P^T g, once
        TFOR(j, 0, n) f.wa1[j] = f.g[f.ipvt[j]];
        tsync();
        // the forward substitution in registers, replicated on every lane (see
        // qrsolv's back substitution): subdot()'s terms outside [0, j) are
        // s - (+0.0) * (+0.0), which leaves every s as it is
        double wv[NP], qv[NP];
        gather(wv, f.wa1, 1, 0, n);
#pragma unroll
        for (int jj = 0; jj < NP; jj++) {
            qv[jj] = 0.0;
            if (jj < n) {
                double s = wv[jj];
#pragma unroll
                for (int i = 0; i < jj; i++) s -= f.R[tri_row(NP, i) + jj] * qv[i];
                const double rjj = f.R[tri_row(NP, jj) + jj];
                qv[jj] = rjj != 0.0 ? s / rjj : 0.0;
            }
        }
        tsync();
        TFOR(j, 0, n) {
            double mine = 0.0;
#pragma unroll
            for (int i = 0; i < NP; i++) mine = i == j ? qv[i] : mine;
            f.qtf[j] = mine;
        }
        tsync();
    }

    // lmcore::qrsolv on r = f.M, with diag = dvec (an LDS vector), qtb = f.qtf,
    // x = f.p, sdiag = f.sdiag, wa = wa (an LDS vector)
    static __device__ __forceinline__ void qrsolv(Fit &f, const double *dvec, double *wa)
    {
        const int n = f.s.n;
        constexpr int ld = LD;
        double *r = f.M, *x = f.p, *sdiag = f.sdiag;
        // (x doubles as the store of the diagonal of R until the end, as in MINPACK)
        TFOR(j, 0, n) {
            double row[NP];
            gather(row, r + j * ld, 1, j, n);
#pragma unroll
            for (int i = 0; i < NP; i++)
                if (i >= j && i < n) r[i * ld + j] = row[i];
            x[j] = r[j * ld + j];
            wa[j] = f.qtf[j];
        }
        tsync();
        // The sweeps.  MINPACK eliminates the diagonal element of sweep j with
        // rotations (j, k), k = j .. n - 1, one sweep after the other: n^2 / 2
        // rotations, each a divide - square root - divide chain.  Rotation (j, k)
        // touches column k of r, wa[k] and sweep j's own work vector, and needs
        // only (j - 1, k) and (j, k - 1) to have happened: sweep j runs in LANE j
        // (its work vector and qtbpj in that lane's registers), rotation (j, k) at
        // step t = j + k, all sweeps at once -- 2 n - 1 steps instead of n^2 / 2,
        // the same operations on every element in an order its dependencies fix.
        {
            const int j = f.lane;
            double dl = 0.0;
            if (j < n) dl = dvec[f.ipvt[j]];
            double sd[NP];
#pragma unroll
            for (int i = 0; i < NP; i++) sd[i] = i == j ? dl : 0.0;
            double qtbpj = 0.0;
            for (int t = 0; t < 2 * n - 1; t++) {
                const int k = t - j;
                if (j < n && k >= j && k < n && dl != 0.0) {
                    double sk = 0.0;
#pragma unroll
                    for (int i = 0; i < NP; i++) sk = i == k ? sd[i] : sk;
                    // (the loads of the step in one batch: none depends on the chain)
                    const double rkk = r[k * ld + k];
                    const double wak = wa[k];
                    double col[NP];
#pragma unroll
                    for (int i = 0; i < NP; i++) col[i] = r[i * ld + k];
                    if (sk != 0.0) {
                        // cotan = rkk / sk or tan = sk / rkk: one chain for both cases
                        const bool steep = fabs(rkk) < fabs(sk);
                        const double ratio = steep ? rkk / sk : sk / rkk;
                        const double q = 0.5 / sqrt(0.25 + 0.25 * (ratio * ratio));
                        const double qr = q * ratio;
                        const double sn = steep ? q : qr, cs = steep ? qr : q;
                        const double temp = cs * wak + sn * qtbpj;
                        qtbpj = -sn * wak + cs * qtbpj;
                        r[k * ld + k] = cs * rkk + sn * sk;
                        wa[k] = temp;
#pragma unroll
                        for (int i = 0; i < NP; i++) {
                            if (i > k && i < n) {
                                const double rik = col[i], si = sd[i];
                                sd[i] = -sn * rik + cs * si;
                                r[i * ld + k] = cs * rik + sn * si;
                            }
                        }
                    }
                }
                tsync();
            }
            TFOR(jj, 0, n) {
                sdiag[jj] = r[jj * ld + jj];
                r[jj * ld + jj] = x[jj];
            }
            tsync();
        }
        // The back substitution with the solution in REGISTERS, replicated on every
        // lane (all lanes run the scalar chain anyway): w_j = (wa_j - sum_{i > j}
        // r[i][j] w_i) / sdiag_j needs no LDS write / barrier / read between two
        // elements -- the chain is the n divisions and their sums, the column loads
        // are independent of it.  Element by element the operations and their
        // order are the loop's above it replaces (dot()'s +0.0 terms outside the
        // range included: a sum that starts at +0.0 is never -0.0, so leaving out
        // the terms i <= j, which are all +0.0 * +0.0, changes no bit).
        double sd[NP], wv[NP];
        gather(sd, sdiag, 1, 0, n);
        gather(wv, wa, 1, 0, n);
        int nsing = n;
#pragma unroll
        for (int j = 0; j < NP; j++)
            if (j < n && sd[j] == 0.0 && nsing == n) nsing = j;
#pragma unroll
        for (int j = 0; j < NP; j++)
            if (j >= nsing) wv[j] = 0.0;
#pragma unroll
        for (int jj = NP - 1; jj >= 0; jj--) {
            if (jj < nsing) {
                double sum = 0.0;
#pragma unroll
                for (int i = jj + 1; i < NP; i++) {
                    const double a = r[i * ld + jj];
                    const bool in = i < nsing;
                    sum += (in ? a : 0.0) * (in ? wv[i] : 0.0);
                }
                wv[jj] = (wv[jj] - sum) / sd[jj];
            }
        }
        // (wa itself is dead: lmpar overwrites it next)
        tsync();
        TFOR(j, 0, n) {
            double mine = 0.0;
#pragma unroll
            for (int i = 0; i < NP; i++) mine = i == j ? wv[i] : mine;
            x[f.ipvt[j]] = mine;
        }
        tsync();
    }

    // lmcore::lmpar on r = f.M (a copy of R made by the caller), diag = f.diag,
    // qtb = f.qtf, delta / par in f.s; x = f.p, sdiag = f.sdiag
    static __device__ __forceinline__ void lmpar(Fit &f)
    {
        const int n = f.s.n;
        constexpr int ld = LD;
        double *r = f.M, *x = f.p, *wa1 = f.wa1, *wa2 = f.wa2;
        const double delta = f.s.delta;
        // gauss-newton direction
        int nsing = n;
        {
            double dg[NP];
            gather(dg, r, ld + 1, 0, n);
#pragma unroll
            for (int j = 0; j < NP; j++)
                if (j < n && dg[j] == 0.0 && nsing == n) nsing = j;
        }
        {
            // the gauss-newton direction by back substitution in registers
            double wv[NP];
            gather(wv, f.qtf, 1, 0, nsing);
#pragma unroll
            for (int jj = NP - 1; jj >= 0; jj--) {
                if (jj < nsing) {
                    const double temp = wv[jj] / r[jj * ld + jj];
                    wv[jj] = temp;
#pragma unroll
                    for (int i = 0; i < jj; i++) wv[i] -= r[i * ld + jj] * temp;
                }
            }
            tsync();
            TFOR(j, 0, n) {
                double mine = 0.0;
#pragma unroll
                for (int i = 0; i < NP; i++) mine = i == j ? wv[i] : mine;
                x[f.ipvt[j]] = mine;
            }
            tsync();
        }

        int iter = 0;
        TFOR(j, 0, n) wa2[j] = f.diag[j] * x[j];
        tsync();
        double dxnorm = enorm(n, wa2);
        double fp = dxnorm - delta;
        if (fp <= 0.1 * delta) {
            f.s.par = 0.0;
            return;
        }
        // lower bound
        double parl = 0.0;
        if (nsing >= n) {
            tsync();
            TFOR(j, 0, n) {
                const int l = f.ipvt[j];
                wa1[j] = f.diag[l] * (wa2[l] / dxnorm);
            }
            tsync();
            double wv[NP];
            gather(wv, wa1, 1, 0, n);
#pragma unroll
            for (int jj = 0; jj < NP; jj++) {
                if (jj < n) {
                    double sum = 0.0;
#pragma unroll
                    for (int i = 0; i < jj; i++) sum += r[i * ld + jj] * wv[i];
                    wv[jj] = (wv[jj] - sum) / r[jj * ld + jj];
                }
            }
            double ss = 0.0;
#pragma unroll
            for (int i = 0; i < NP; i++) ss += wv[i] * wv[i];
            const double temp = sqrt(ss);
            parl = ((fp / delta) / temp) / temp;
        }
        // upper bound
        tsync();
        TFOR(j, 0, n) {
            const double sum = dot(r + j, ld, f.qtf, 1, 0, j + 1);
            wa1[j] = sum / f.diag[f.ipvt[j]];
        }
        tsync();
        const double gnorm = enorm(n, wa1);
        double paru = gnorm / delta;
        if (paru == 0.0) paru = lmcore::DWARF / fmin(delta, 0.1);
        double par = f.s.par;
        par = fmax(par, parl);
        par = fmin(par, paru);
        if (par == 0.0) par = gnorm / dxnorm;

        for (;;) {
            iter++;
            if (par == 0.0) par = fmax(lmcore::DWARF, 0.001 * paru);
            double temp = sqrt(par);
            tsync();
            TFOR(j, 0, n) wa1[j] = temp * f.diag[j];
            tsync();
            qrsolv(f, wa1, wa2);
            TFOR(j, 0, n) wa2[j] = f.diag[j] * x[j];
            tsync();
            dxnorm = enorm(n, wa2);
            temp = fp;
            fp = dxnorm - delta;
            if (fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) ||
                iter == 10)
                break;
            // newton correction
            tsync();
            TFOR(j, 0, n) {
                const int l = f.ipvt[j];
                wa1[j] = f.diag[l] * (wa2[l] / dxnorm);
            }
            tsync();
            {
                // the forward substitution in registers (see qrsolv's back
                // substitution): wa1 is not read again before it is overwritten
                double sd[NP], wv[NP];
                gather(sd, f.sdiag, 1, 0, n);
                gather(wv, wa1, 1, 0, n);
#pragma unroll
                for (int jj = 0; jj < NP; jj++) {
                    if (jj < n) {
                        const double t = wv[jj] / sd[jj];
                        wv[jj] = t;
#pragma unroll
                        for (int i = jj + 1; i < NP; i++) {
                            const double a = r[i * ld + jj];
                            if (i < n) wv[i] -= a * t;
                        }
                    }
                }
                // enorm(n, wa1): the squares added in index order from +0.0
                double ss = 0.0;
#pragma unroll
                for (int i = 0; i < NP; i++) ss += wv[i] * wv[i];
                temp = sqrt(ss);
            }
            const double parc = ((fp / delta) / temp) / temp;
            if (fp > 0.0) parl = fmax(parl, par);
            if (fp < 0.0) paru = fmin(paru, par);
            par = fmax(parl, par + parc);
        }
        f.s.par = par;
    }

    // lmcore::set_trial: element j in lane j
    static __device__ __forceinline__ void set_trial(Fit &f)
    {
        constexpr double EPS = 1.4901161193847656e-08;  // sqrt(machine epsilon)
        TFOR(j, 0, f.s.n) {
            f.xt = f.s.bounded ? lmcore::i2e(f.xti, f.lo, f.hi) : f.xti;
            if (f.s.mode == NGMIX_LM_MODE_FD) {
                double h = EPS * fabs(f.xti);
                if (h == 0.0) h = EPS;
                f.hstep = h;
                f.xstep = f.s.bounded ? lmcore::i2e(f.xti + h, f.lo, f.hi) : f.xti + h;
            }
        }
    }

    // wa3 = R (P^T step) as lmder forms it, then |wa3| / fnorm  (used twice)
    static __device__ __forceinline__ double r_times_step_norm(Fit &f)
    {
        const int n = f.s.n;
        tsync();
        TFOR(j, 0, n) f.wa2[j] = f.step[f.ipvt[j]];
        tsync();
        TFOR(i, 0, n) f.wa3[i] = dot(f.R + tri_row(NP, i), 1, f.wa2, 1, i, n);
        tsync();
        return enorm(n, f.wa3) / f.s.fnorm;
    }

    // lmcore::propose
    static __device__ __forceinline__ void propose(Fit &f)
    {
        const int n = f.s.n;
        constexpr int ld = LD;
        tsync();
        TFOR(i, 0, n) {
            // (row i of the packed factor, +0.0 below the diagonal as the full
            // matrix holds it)
            double row[NP];
            gather(row, f.R + tri_row(NP, i), 1, i, n);
#pragma unroll
            for (int j = 0; j < NP; j++)
                if (j < n) f.M[i * ld + j] = row[j];
        }
        tsync();
        lmpar(f);
        tsync();
        TFOR(j, 0, n) {
            const double st = -f.p[j];
            f.step[j] = st;
            f.xti = f.xi + st;
            f.wa3[j] = f.diag[j] * st;
        }
        set_trial(f);
        tsync();
        f.s.pnorm = enorm(n, f.wa3);
        if (f.s.iter == 1) f.s.delta = fmin(f.s.delta, f.s.pnorm);
        f.s.fonly = 0;
        if (f.s.mode == NGMIX_LM_MODE_ANALYTIC_LAZY) {
            const double temp1 = r_times_step_norm(f);
            const double temp2 = (sqrt(f.s.par) * f.s.pnorm) / f.s.fnorm;
            const double prered = temp1 * temp1 + temp2 * temp2 / 0.5;
            if (prered <= f.s.ftol || f.s.pnorm / 0.5 <= f.s.xtol * f.s.xnorm) f.s.fonly = 1;
        }
    }

    // lmcore::new_jacobian with A = f.M, g = f.g, up to its call of propose:
    // returns true when the fit has terminated
    static __device__ __forceinline__ bool new_jacobian(Fit &f)
    {
        const int n = f.s.n;
        f.s.njev++;
        factor_normal(f);
        if (f.s.iter == 1) {
            TFOR(j, 0, n) {
                double d = f.acnorm[j];
                if (d == 0.0) d = 1.0;
                f.diag[j] = d;
                f.wa3[j] = d * f.xi;
            }
            tsync();
            f.s.xnorm = enorm(n, f.wa3);
            f.s.delta = f.s.factor * f.s.xnorm;
            if (f.s.delta == 0.0) f.s.delta = f.s.factor;
        }
        qtf_from_gradient(f);
        // norm of the scaled gradient: sum_j = sum_{i <= j} R[i][j] (qtf[i] / fnorm)
        // per column, one column per lane, then the maximum in column order
        f.s.gnorm = 0.0;
        if (f.s.fnorm != 0.0) {
            const double fnorm = f.s.fnorm;
            TFOR(i, 0, n) f.wa1[i] = f.qtf[i] / fnorm;
            tsync();
            TFOR(j, 0, n) {
                const double al = f.acnorm[f.ipvt[j]];
                const double sum = dot_col(f.R, j, f.wa1, 0, j + 1);
                // (a column lmder skips: any value fmax(gnorm, .) ignores)
                f.wa2[j] = al == 0.0 ? -1.0 : fabs(sum / al);
            }
            tsync();
            double c[NP];
            gather(c, f.wa2, 1, 0, n);
            double gn = 0.0;
#pragma unroll
            for (int j = 0; j < NP; j++)
                if (j < n) gn = fmax(gn, c[j]);
            f.s.gnorm = gn;
        }
        if (f.s.gnorm <= f.s.gtol) {
            f.s.info = 4;
            f.s.phase = LM_PHASE_DONE;
            return true;
        }
        tsync();
        TFOR(j, 0, n) f.diag[j] = fmax(f.diag[j], f.acnorm[j]);
        tsync();
        return false;
    }

    // lmcore::lm_advance with the evaluation folded into f.M (A), f.g (g), ff.
    // new_jacobian and propose have ONE call site each (the branches say what
    // they want): 9k instructions instead of 14k.
    static __device__ __forceinline__ void lm_advance(Fit &f, double ff)
    {
        const int n = f.s.n;
        constexpr int ld = LD;
        if (f.s.phase == LM_PHASE_DONE) return;
        if (f.s.bounded && f.s.mode != NGMIX_LM_MODE_FD) {
            // the wrapped Dfun of leastsqbound.py:485-489: column j scaled by
            // d xt_j / d xti_j
            double sc = 1.0;
            TFOR(j, 0, n) {
                sc = lmcore::i2e_grad(f.xti, f.lo, f.hi);
                f.wa1[j] = sc;
            }
            tsync();
            TFOR(j, 0, n) {
                f.g[j] = f.g[j] * sc;
                double row[NP], scv[NP];
                gather(row, f.M + j * ld, 1, 0, n);
                gather(scv, f.wa1, 1, 0, n);
#pragma unroll
                for (int k = 0; k < NP; k++)
                    if (k < n) f.M[j * ld + k] = row[k] * sc * scv[k];
            }
            tsync();
        }
        bool want_jacobian = false, want_proposal = false;
        if (f.s.phase == LM_PHASE_JAC) {
            if (f.s.mode == NGMIX_LM_MODE_FD) f.s.nfev += n;
            f.s.phase = LM_PHASE_TRIAL;
            want_jacobian = true;
        } else if (f.s.phase == LM_PHASE_INIT) {
            f.s.nfev = f.s.mode == NGMIX_LM_MODE_FD ? 1 + n : 1;
            f.s.fnorm = sqrt(ff);
            f.s.par = 0.0;
            f.s.iter = 1;
            if (!(f.s.fnorm < INFINITY)) {
                f.s.njev = 1;
                f.s.info = 4;
                f.s.phase = LM_PHASE_DONE;
                return;
            }
            f.s.phase = LM_PHASE_TRIAL;
            want_jacobian = true;
        } else {
            // ---- LM_PHASE_TRIAL: the rest of lmder's inner loop
            f.s.nfev++;
            const double fnorm1 = ff < INFINITY ? sqrt(ff) : NAN;
            double actred = -1.0;
            if (0.1 * fnorm1 < f.s.fnorm) {
                const double t = fnorm1 / f.s.fnorm;
                actred = 1.0 - t * t;
            }
            const double temp1 = r_times_step_norm(f);
            const double temp2 = (sqrt(f.s.par) * f.s.pnorm) / f.s.fnorm;
            const double prered = temp1 * temp1 + temp2 * temp2 / 0.5;
            const double dirder = -(temp1 * temp1 + temp2 * temp2);
            double ratio = 0.0;
            if (prered != 0.0) ratio = actred / prered;
            if (ratio <= 0.25) {
                double temp = 0.5;
                if (actred < 0.0) temp = 0.5 * dirder / (dirder + 0.5 * actred);
                if (0.1 * fnorm1 >= f.s.fnorm || temp < 0.1) temp = 0.1;
                f.s.delta = temp * fmin(f.s.delta, f.s.pnorm / 0.1);
                f.s.par = f.s.par / temp;
            } else if (f.s.par == 0.0 || ratio >= 0.75) {
                f.s.delta = f.s.pnorm / 0.5;
                f.s.par = 0.5 * f.s.par;
            }
            const bool accepted = ratio >= 1.0e-4;
            if (accepted) {
                tsync();
                TFOR(j, 0, n) {
                    f.x = f.xt;
                    f.xi = f.xti;
                    f.wa3[j] = f.diag[j] * f.xi;
                }
                tsync();
                f.s.xnorm = enorm(n, f.wa3);
                f.s.fnorm = fnorm1;
                f.s.iter++;
            }
            int info = 0;
            if (fabs(actred) <= f.s.ftol && prered <= f.s.ftol && 0.5 * ratio <= 1.0) info = 1;
            if (f.s.delta <= f.s.xtol * f.s.xnorm) info = 2;
            if (fabs(actred) <= f.s.ftol && prered <= f.s.ftol && 0.5 * ratio <= 1.0 &&
                info == 2)
                info = 3;
            if (info == 0) {
                if (f.s.nfev >= f.s.maxfev) info = 5;
                if (fabs(actred) <= lmcore::EPSMCH && prered <= lmcore::EPSMCH &&
                    0.5 * ratio <= 1.0)
                    info = 6;
                if (f.s.delta <= lmcore::EPSMCH * f.s.xnorm) info = 7;
                if (f.s.gnorm <= lmcore::EPSMCH) info = 8;
            }
            if (info != 0) {
                f.s.info = info;
                f.s.phase = LM_PHASE_DONE;
                return;
            }
            if (!accepted) {
                want_proposal = true;  // same factor, smaller region
            } else if (f.s.mode == NGMIX_LM_MODE_FD || f.s.fonly) {
                f.xti = f.xi;
                set_trial(f);
                f.s.phase = LM_PHASE_JAC;
                f.s.fonly = 0;
            } else {
                want_jacobian = true;  // the trial point's jacobian is the new one
            }
        }
        if (want_jacobian && !new_jacobian(f)) want_proposal = true;
        if (want_proposal) propose(f);
    }
};

}  // namespace lmteam
