// moments.hip -- fixed-weight moment sums and adaptive moments.
//
//   get_weighted_sums / get_higher_order_weighted_sums
//       (reference: ngmix/gmix/gmix_nb.py:681-821)
//   admom + admom_censums + admom_momsums + deweight_moments + clear_result
//       (reference: ngmix/admom/admom_nb.py:13-239)
//
// One work-group per stamp.  Both are compute-bound: the stamp
// is read from HBM once and the iteration runs on chip.
#include <stdio.h>

#include "iter_common.hpp"
#include "launch_iter.hpp"

#include <stdlib.h>

namespace ngmix {

__constant__ double c_exp_table_m[16] = NGMIX_EXP_TABLE;
__constant__ double c_fexp_coef_m[12] = NGMIX_FEXP_COEF;

// ===========================================================================
// weighted sums
// ===========================================================================
//
// Phase 1 (pixel-parallel): each thread evaluates one pixel of a 256-pixel
// chunk -- weight (true exp, no cut, gmix_nb.py:715), w2*var, wdata and the
// F vector -- into LDS.  Phase 2 (output-parallel): thread t owns output
// element t of the record (sums, sums_cov, wsum, npix, F) and adds the chunk's
// pixels to it in pixel order.  Every output is therefore accumulated in
// exactly the reference's sequential order, into the caller's existing value.

constexpr int WS_CHUNK = 256;

template <class Src>
__device__ __forceinline__ void weighted_sums_body(const Src &src,
                                                   const ngmix_gauss2d *wt, int ng,
                                                   char *resbase, int nmom,
                                                   double maxrad, int32_t *status,
                                                   char *smem)
{
    const int nrec = nmom + 3;  // F[nmom], w2var, wdata, weight
    double *rec = (double *)smem;                      // [WS_CHUNK][nrec]
    int *used = (int *)(rec + WS_CHUNK * nrec);        // [WS_CHUNK]
    EvalGauss *ge = (EvalGauss *)(used + WS_CHUNK + 2);  // [ng], 8-aligned
    int *err = used + WS_CHUNK;

    const int tid = threadIdx.x;
    for (int g = tid; g < ng; g += BLOCK) ge[g] = make_eval(wt[g]);
    if (tid == 0) *err = 0;
    __syncthreads();

    int32_t *r_npix = (int32_t *)(resbase + 4);
    double *r_wsum = (double *)(resbase + 8);
    double *r_sums = (double *)(resbase + 16);
    double *r_cov = r_sums + nmom;
    double *r_F = r_cov + nmom * nmom + nmom;

    // outputs owned by this thread: index o = tid + k*BLOCK
    //   [0,nmom) sums | [nmom, nmom+nmom^2) cov | wsum | npix | F[nmom]
    const int n_sums = nmom, n_cov = nmom * nmom;
    const int o_wsum = n_sums + n_cov, o_npix = o_wsum + 1, o_F = o_npix + 1;
    const int nout = o_F + nmom;
    constexpr int MAXOWN = 2;  // 17 moments: 325 outputs over 256 threads
    double acc[MAXOWN];
#pragma unroll
    for (int k = 0; k < MAXOWN; k++) {
        const int o = tid + k * BLOCK;
        acc[k] = 0.0;
        if (o < n_sums) acc[k] = r_sums[o];
        else if (o < o_wsum) acc[k] = r_cov[o - n_sums];
        else if (o == o_wsum) acc[k] = *r_wsum;
        else if (o == o_npix) acc[k] = (double)*r_npix;
        else if (o < nout) acc[k] = r_F[o - o_F];
    }

    const double maxrad2 = maxrad * maxrad;
    const double vcen = wt[0].row, ucen = wt[0].col;
    const int n = src.count();

    for (int c0 = 0; c0 < n; c0 += WS_CHUNK) {
        // ---- phase 1
        {
            const int p = c0 + tid;
            int use = 0;
            if (p < n) {
                double v, u, area, val, ierr;
                const bool kept = src.load(p, v, u, area, val, ierr);
                const double vmod = v - vcen, umod = u - ucen;
                const double rad2 = umod * umod + vmod * vmod;
                bool take = kept && rad2 < maxrad2;
                if (nmom == 6) take = take && ierr > 0.0;  // gmix_nb.py:713
                if (take) {
                    const double ierr2 = ierr * ierr;
                    if (ierr2 == 0.0) {
                        *err = NGMIX_ERR_ZERO_DIV;  // gmix_nb.py:775
                    } else {
                        double weight = 0.0;
                        for (int g = 0; g < ng; g++)
                            weight += gauss_eval_exact(ge[g], v, u, area);
                        const double var = 1.0 / ierr2;
                        double *r = rec + tid * nrec;
                        r[nmom + 0] = weight * weight * var;  // w2*var
                        r[nmom + 1] = weight * val;           // wdata
                        r[nmom + 2] = weight;
                        r[0] = v;
                        r[1] = u;
                        if (nmom == 6) {
                            r[2] = umod * umod - vmod * vmod;
                            r[3] = 2 * vmod * umod;
                            r[4] = rad2;
                            r[5] = 1.0;
                        } else {
                            // gmix_nb.py:780-813 (v,u here are vmod,umod)
                            const double uu = umod, vv = vmod, r2 = rad2;
                            const double u2 = uu * uu, v2 = vv * vv, vu = vv * uu;
                            const double u4 = u2 * u2, v4 = v2 * v2;
                            const double r4 = r2 * r2, r6 = r4 * r2, r8 = r6 * r2;
                            r[2] = u2 - v2;
                            r[3] = 2 * vu;
                            r[4] = r2;
                            r[5] = 1.0;
                            r[6] = uu * r2;
                            r[7] = vv * r2;
                            r[8] = uu * (u2 - 3 * v2);
                            r[9] = vv * (3 * u2 - v2);
                            r[10] = r4;
                            r[11] = r2 * (u2 - v2);
                            r[12] = r2 * 2 * uu * vv;
                            r[13] = u4 - 6 * u2 * v2 + v4;
                            r[14] = (u2 - v2) * 4 * uu * vv;
                            r[15] = r6;
                            r[16] = r8;
                        }
                        use = 1;
                    }
                }
            }
            used[tid] = use;
        }
        __syncthreads();
        // ---- phase 2
        const int cn = (n - c0) < WS_CHUNK ? (n - c0) : WS_CHUNK;
#pragma unroll
        for (int k = 0; k < MAXOWN; k++) {
            const int o = tid + k * BLOCK;
            if (o >= nout) continue;
            double a = acc[k];
            if (o < n_sums) {
                for (int q = 0; q < cn; q++)
                    if (used[q]) a += rec[q * nrec + nmom + 1] * rec[q * nrec + o];
            } else if (o < o_wsum) {
                const int i = (o - n_sums) / nmom, j = (o - n_sums) - i * nmom;
                for (int q = 0; q < cn; q++)
                    if (used[q])
                        a += rec[q * nrec + nmom] * rec[q * nrec + i] * rec[q * nrec + j];
            } else if (o == o_wsum) {
                for (int q = 0; q < cn; q++)
                    if (used[q]) a += rec[q * nrec + nmom + 2];
            } else if (o == o_npix) {
                for (int q = 0; q < cn; q++)
                    if (used[q]) a += 1.0;
            } else {
                for (int q = 0; q < cn; q++)
                    if (used[q]) a = rec[q * nrec + (o - o_F)];
            }
            acc[k] = a;
        }
        __syncthreads();
    }

    if (*err != 0) {
        // the reference raises mid-loop; the record is left untouched here
        if (tid == 0 && status) *status = *err;
        return;
    }
#pragma unroll
    for (int k = 0; k < MAXOWN; k++) {
        const int o = tid + k * BLOCK;
        if (o < n_sums) r_sums[o] = acc[k];
        else if (o < o_wsum) r_cov[o - n_sums] = acc[k];
        else if (o == o_wsum) *r_wsum = acc[k];
        else if (o == o_npix) *r_npix = (int32_t)acc[k];
        else if (o < nout) r_F[o - o_F] = acc[k];
    }
    if (tid == 0 && status) *status = NGMIX_OK;
}

// ---------------------------------------------------------------------------
// FUSED weighted sums (batch form default): one wave per stamp, every sum in a
// register accumulator per lane, FMA arithmetic, a fixed-order tree at the
// end.  The exact-order kernel above walks 256-pixel chunks with ONE thread
// per output element adding pixels sequentially (the reference's order, to
// the last bit) and is ~60x slower: 75 ms per 100k 48x48 stamps against
// ~1.3 ms here.  Results agree to summation-order rounding (<= 1e-12 of the
// sum of |terms|).
//
// NMOM = 6: one pass, 6 sums + 21 covariance entries (the reference fills all
// 36, symmetric by construction: mirrored on output) + wsum + npix.
// NMOM = 17: 17 + 153 + 2 accumulators do not fit the register file; four
// passes over the pixels each own a block of covariance rows (the stamp is
// re-read: 16 B/pixel per pass).
// ---------------------------------------------------------------------------

template <int NMOM>
__device__ __forceinline__ void wsums_F(double vmod, double umod, double v, double u,
                                        double rad2, double (&F)[NMOM])
{
    F[0] = v;
    F[1] = u;
    if (NMOM == 6) {
        F[2] = umod * umod - vmod * vmod;
        F[3] = 2 * vmod * umod;
        F[4] = rad2;
        F[5] = 1.0;
    } else {
        // gmix_nb.py:780-813
        const double uu = umod, vv = vmod, r2 = rad2;
        const double u2 = uu * uu, v2 = vv * vv, vu = vv * uu;
        const double u4 = u2 * u2, v4 = v2 * v2;
        const double r4 = r2 * r2, r6 = r4 * r2, r8 = r6 * r2;
        F[2] = u2 - v2;
        F[3] = 2 * vu;
        F[4] = r2;
        F[5] = 1.0;
        F[6 % NMOM] = uu * r2;
        F[7 % NMOM] = vv * r2;
        F[8 % NMOM] = uu * (u2 - 3 * v2);
        F[9 % NMOM] = vv * (3 * u2 - v2);
        F[10 % NMOM] = r4;
        F[11 % NMOM] = r2 * (u2 - v2);
        F[12 % NMOM] = r2 * 2 * uu * vv;
        F[13 % NMOM] = u4 - 6 * u2 * v2 + v4;
        F[14 % NMOM] = (u2 - v2) * 4 * uu * vv;
        F[15 % NMOM] = r6;
        F[16 % NMOM] = r8;
    }
}

struct WsumsWaveShared {
    double red[64 * 4];
    int err;
    int lastp;
};

// covariance rows [I0, I1) (entries j >= i), plus the sums / wsum / npix when
// WITH_SUMS; tot[] receives the wave totals in accumulator order
template <int NMOM, int I0, int I1, bool WITH_SUMS>
__device__ __forceinline__ void wsums_pass(const GridSrc &src, const EvalGauss *ge,
                                           int ng, double vcen, double ucen,
                                           double maxrad2, WsumsWaveShared &sh,
                                           double *tot, int &lastp_out)
{
    constexpr int NCOV = (I1 - I0) * NMOM - ((I1 - 1) * I1 - (I0 - 1) * I0) / 2;
    constexpr int NACC = NCOV + (WITH_SUMS ? NMOM + 2 : 0);
    static_assert(NACC <= 64, "too many accumulators for one pass");
    const int lane = threadIdx.x;
    const int lrow = lane / TILE_W, lcol = lane % TILE_W;
    const int nrow = src.nrow, ncol = src.ncol;
    double acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; k++) acc[k] = 0.0;
    int lastp = -1;
    bool zero_div = false;

    // Tiles in row-major order, two per loop trip, each one's val / ierr requested
    // a tile ahead: with the load at the top of its own tile every wave spent
    // 80 % of its life waiting for memory (a tile is only ~100 instructions).  The loads are unconditional -- a lane
    // outside the stamp reads pixel 0 and ignores it -- because a branch around
    // a load makes the compiler drain every outstanding load right there.
    const int ntx = (ncol + TILE_W - 1) / TILE_W, nty = (nrow + TILE_H - 1) / TILE_H;
    const int ntiles = ntx * nty;
    struct Tile {
        double val, ierr;
        int row, col;
        bool inb;
    };
    int f_ty = 0, f_tx = 0, f_T = 0;   // the next tile to request
    auto fetch = [&]() {
        Tile t;
        t.row = f_ty * TILE_H + lrow;
        t.col = f_tx * TILE_W + lcol;
        t.inb = f_T < ntiles && t.row < nrow && t.col < ncol;
        const int p = t.inb ? t.row * ncol + t.col : 0;
        t.val = src.val[p];
        t.ierr = src.ierr[p];
        f_T++;
        if (++f_tx == ntx) {
            f_tx = 0;
            f_ty++;
        }
        return t;
    };
    auto eval = [&](const Tile &t) {
        if (!t.inb) return;
        const int row = t.row, col = t.col, p = row * ncol + col;
        const double val = t.val, ierr = t.ierr;
        if (src.izw && !(ierr > 0.0)) return;  // not in the pixel list
            double v, u;
            {
                const double rd = (double)row - src.jac.row0, cd = (double)col - src.jac.col0;
                v = fma(src.jac.dvdrow, rd, src.jac.dvdcol * cd);
                u = fma(src.jac.dudrow, rd, src.jac.dudcol * cd);
            }
            const double vmod = v - vcen, umod = u - ucen;
            const double rad2 = fma(umod, umod, vmod * vmod);
            bool take = rad2 < maxrad2;
            if (NMOM == 6) take = take && ierr > 0.0;  // gmix_nb.py:713
            if (!take) return;
            const double ierr2 = ierr * ierr;
            if (ierr2 == 0.0) {
                zero_div = true;  // gmix_nb.py:775: 1/ierr^2
                return;
            }
            double weight = 0.0;
            for (int g = 0; g < ng; g++) {
                const EvalGauss e = ge[g];
                const double vd = v - e.row, ud = u - e.col;
                const double chi2 =
                    fma(e.dcc * vd, vd, fma(e.drr * ud, ud, -(e.drc2 * vd) * ud));
                weight = fma(e.pnorm * exp(-0.5 * chi2), src.area, weight);
            }
            // (fused kernel: v_rcp_f64 + two Newton steps, ~1 ulp, instead of the
            // ~25-instruction IEEE division)
            const double var = rcp_newton(ierr2);
            const double w2var = weight * weight * var;
            double F[NMOM];
            wsums_F<NMOM>(vmod, umod, v, u, rad2, F);
            int k = 0;
#pragma unroll
            for (int i = I0; i < I1; i++) {
                const double wf = w2var * F[i];
#pragma unroll
                for (int j = i; j < NMOM; j++) {
                    acc[k] = fma(wf, F[j], acc[k]);
                    k++;
                }
            }
            if (WITH_SUMS) {
                const double wdata = weight * val;
#pragma unroll
                for (int i = 0; i < NMOM; i++) acc[NCOV + i] = fma(wdata, F[i], acc[NCOV + i]);
                acc[NCOV + NMOM] += weight;
                acc[NCOV + NMOM + 1] += 1.0;
            }
            lastp = p > lastp ? p : lastp;
    };
    // two register sets, each overwritten right after it has been evaluated (no
    // copies): B's loads are in flight while A is evaluated and vice versa
    Tile ta = fetch(), tb = fetch();
    for (int T = 0; T < ntiles; T += 2) {
        eval(ta);
        ta = fetch();
        eval(tb);
        tb = fetch();
    }
    if (__ballot(zero_div) != 0ull && lane == 0) sh.err = NGMIX_ERR_ZERO_DIV;
    // wave totals: DPP inside rows of 16 lanes, the 4 row sums through LDS
#pragma unroll
    for (int k = 0; k < NACC; k++) {
        double x = acc[k];
        x += dpp_move_or_zero<0x111, 0xf>(x);
        x += dpp_move_or_zero<0x112, 0xf>(x);
        x += dpp_move_or_zero<0x114, 0xf>(x);
        x += dpp_move_or_zero<0x118, 0xf>(x);
        if ((lane & 15) == 15) sh.red[k * 4 + (lane >> 4)] = x;
    }
    __syncthreads();
    if (lane < NACC) {
        const double *r = sh.red + lane * 4;
        tot[lane] = ((r[0] + r[1]) + r[2]) + r[3];
    }
    // the position of the last pixel used (its F stays in the record)
    for (int off = WAVE / 2; off > 0; off >>= 1) {
        const int y = __shfl_down(lastp, off, WAVE);
        lastp = y > lastp ? y : lastp;
    }
    lastp_out = __builtin_amdgcn_readfirstlane(lastp);
    __syncthreads();
}

// add the totals of one pass into the record
template <int NMOM, int I0, int I1, bool WITH_SUMS>
__device__ __forceinline__ void wsums_store(const double *tot, char *resbase)
{
    constexpr int NCOV = (I1 - I0) * NMOM - ((I1 - 1) * I1 - (I0 - 1) * I0) / 2;
    const int lane = threadIdx.x;
    int32_t *r_npix = (int32_t *)(resbase + 4);
    double *r_wsum = (double *)(resbase + 8);
    double *r_sums = (double *)(resbase + 16);
    double *r_cov = r_sums + NMOM;
    if (lane == 0) {
        int k = 0;
        for (int i = I0; i < I1; i++)
            for (int j = i; j < NMOM; j++) {
                r_cov[i * NMOM + j] += tot[k];
                if (j != i) r_cov[j * NMOM + i] += tot[k];
                k++;
            }
        if (WITH_SUMS) {
            for (int i = 0; i < NMOM; i++) r_sums[i] += tot[NCOV + i];
            *r_wsum += tot[NCOV + NMOM];
            *r_npix += (int32_t)tot[NCOV + NMOM + 1];
        }
    }
}

// The 17-moment sums (get_higher_order_weighted_sums, gmix_nb.py:737-821) in
// ONE pass over the pixels.  sums_cov[i][j] = sum w^2 var F_i F_j is X^T X for
// X = (w / ierr) F, a sum of outer products over the pixels: its leading
// 16 x 16 block runs on v_mfma_f64_16x16x4_f64 (four pixels per instruction,
// the result in four accumulator registers per lane), row 16 and the 17 data
// sums sum w I F_i stay per-lane VALU accumulators.  With all 172 sums in VALU
// accumulators the kernel needed four passes over the pixels (and four true
// exp per pixel): 11.8 ms per 100k 48x48 stamps.
__device__ __forceinline__ void wsums17_single_pass(const GridSrc &src, const EvalGauss *ge,
                                                    int ng, double vcen, double ucen,
                                                    double maxrad2, WsumsWaveShared &sh,
                                                    char *resbase, int &lastp_out)
{
    constexpr int NMOM = 17;
    constexpr int XS = 17;   // doubles between the rows of the transposition tile
    __shared__ double xbuf[WAVE * XS];
    typedef double double4_t __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x;
    const int lrow = lane / TILE_W, lcol = lane % TILE_W;
    const int nrow = src.nrow, ncol = src.ncol;
    // two accumulators: consecutive MFMAs are independent of each other
    double4_t M = {0.0, 0.0, 0.0, 0.0}, M2 = {0.0, 0.0, 0.0, 0.0};
    double c16[NMOM], sm[NMOM];   // sums_cov[16][:] and sums[:]
#pragma unroll
    for (int k = 0; k < NMOM; k++) c16[k] = sm[k] = 0.0;
    double wsum = 0.0;
    int npix = 0, lastp = -1;
    bool zero_div = false;
    double *xrow = xbuf + lane * XS;
    const double *xsrc = xbuf + (lane >> 4) * XS + (lane & 15);

    for (int r0 = 0; r0 < nrow; r0 += TILE_H) {
        for (int c0 = 0; c0 < ncol; c0 += TILE_W) {
            const int row = r0 + lrow, col = c0 + lcol;
            bool take = row < nrow && col < ncol;
            const int p = row * ncol + col;
            double val = 0.0, ierr = 0.0;
            if (take) {
                val = src.val[p];
                ierr = src.ierr[p];
                if (src.izw && !(ierr > 0.0)) take = false;  // not in the pixel list
            }
            double v, u;
            {
                const double rd = (double)row - src.jac.row0, cd = (double)col - src.jac.col0;
                v = fma(src.jac.dvdrow, rd, src.jac.dvdcol * cd);
                u = fma(src.jac.dudrow, rd, src.jac.dudcol * cd);
            }
            const double vmod = v - vcen, umod = u - ucen;
            const double rad2 = fma(umod, umod, vmod * vmod);
            take = take && rad2 < maxrad2;
            if (take && ierr * ierr == 0.0) {
                zero_div = true;  // gmix_nb.py:775: 1/ierr^2
                take = false;
            }
            double F[NMOM];
            wsums_F<NMOM>(vmod, umod, v, u, rad2, F);
            double scale = 0.0, dcol = 0.0, weight = 0.0;
            if (take) {
                for (int g = 0; g < ng; g++) {
                    const EvalGauss e = ge[g];
                    const double vd = v - e.row, ud = u - e.col;
                    const double chi2 =
                        fma(e.dcc * vd, vd, fma(e.drr * ud, ud, -(e.drc2 * vd) * ud));
                    weight = fma(e.pnorm * exp(-0.5 * chi2), src.area, weight);
                }
                scale = weight * rcp_newton(ierr);   // scale^2 = w^2 var
                dcol = val * ierr;       // dcol * X_i = w I F_i
                lastp = p > lastp ? p : lastp;
            }
            npix += __popcll(__ballot(take));
            wsum += weight;
            // a pixel that is not used contributes a zero row (its F may be
            // anything finite: the coordinates are, the values are not read)
            const double x16 = take ? scale * F[16] : 0.0;
#pragma unroll
            for (int i = 0; i < NMOM; i++) {
                const double x = take ? scale * F[i] : 0.0;
                if (i < 16) xrow[i] = x;
                c16[i] = fma(x16, x, c16[i]);
                sm[i] = fma(dcol, x, sm[i]);
            }
            __syncthreads();   // one wave: orders the LDS traffic, no s_barrier
#pragma unroll
            for (int t = 0; t < 16; t += 2) {
                const double xa = xsrc[4 * t * XS], xb = xsrc[4 * (t + 1) * XS];
                M = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, xa, M, 0, 0, 0);
                M2 = __builtin_amdgcn_mfma_f64_16x16x4f64(xb, xb, M2, 0, 0, 0);
            }
            __syncthreads();
        }
    }
    if (__ballot(zero_div) != 0ull && lane == 0) sh.err = NGMIX_ERR_ZERO_DIV;
    // wave totals of the 35 VALU sums: DPP inside rows of 16 lanes, the 4 row
    // sums through LDS
    double *red = sh.red;   // 64 * 4 doubles
    auto fold = [&](double x, int k) {
        x += dpp_move_or_zero<0x111, 0xf>(x);
        x += dpp_move_or_zero<0x112, 0xf>(x);
        x += dpp_move_or_zero<0x114, 0xf>(x);
        x += dpp_move_or_zero<0x118, 0xf>(x);
        if ((lane & 15) == 15) red[k * 4 + (lane >> 4)] = x;
    };
#pragma unroll
    for (int k = 0; k < NMOM; k++) {
        fold(c16[k], k);
        fold(sm[k], NMOM + k);
    }
    fold(wsum, 2 * NMOM);
    for (int off = WAVE / 2; off > 0; off >>= 1) {
        const int y = __shfl_down(lastp, off, WAVE);
        lastp = y > lastp ? y : lastp;
    }
    lastp_out = __builtin_amdgcn_readfirstlane(lastp);
    __syncthreads();
    if (sh.err != 0) return;   // the reference raises: the record keeps its content

    // ---- add into the caller's record (it may come pre-filled, gmix.py:740-744)
    int32_t *r_npix = (int32_t *)(resbase + 4);
    double *r_wsum = (double *)(resbase + 8);
    double *r_sums = (double *)(resbase + 16);
    double *r_cov = r_sums + NMOM;
    {
        const int colm = lane & 15;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int rowm = (lane >> 4) + 4 * r;
            r_cov[rowm * NMOM + colm] += M[r] + M2[r];
        }
    }
    if (lane < 2 * NMOM + 1) {
        const double *q = red + lane * 4;
        const double t = ((q[0] + q[1]) + q[2]) + q[3];
        if (lane < NMOM) {
            r_cov[16 * NMOM + lane] += t;
            if (lane != 16) r_cov[lane * NMOM + 16] += t;
        } else if (lane < 2 * NMOM) {
            r_sums[lane - NMOM] += t;
        } else {
            *r_wsum += t;
            *r_npix += npix;
        }
    }
}

template <int NMOM>
__global__ __launch_bounds__(WAVE) void wsums_wave_kernel(
    const ngmix_stamp *stamps, const double *val, const double *ierr,
    const ngmix_jacobian *jacs, const ngmix_gauss2d *gmix, char *res,
    const double *maxrad, int32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ WsumsWaveShared sh;
    __shared__ double tot[64];
    EvalGauss *ge = (EvalGauss *)smem;
    const int s = blockIdx.x;
    const int lane = threadIdx.x;
    const ngmix_stamp st = stamps[s];
    GridSrc src;
    src.val = val + st.pix_off;
    src.ierr = ierr + st.pix_off;
    src.jac = jacs[s];
    src.area = src.jac.scale * src.jac.scale;
    src.nrow = st.nrow;
    src.ncol = st.ncol;
    src.izw = (st.flags & NGMIX_STAMP_IGNORE_ZERO_WEIGHT) != 0;
    const ngmix_gauss2d *wt = gmix + st.gm_off;
    const int ng = st.ngauss;
    for (int g = lane; g < ng; g += WAVE) ge[g] = make_eval(wt[g]);
    if (lane == 0) sh.err = 0;
    __syncthreads();
    const double vcen = wt[0].row, ucen = wt[0].col;
    const double mr = maxrad[s], maxrad2 = mr * mr;
    char *resbase = res + (size_t)s * NGMIX_MOMENTS_RESULT_BYTES(NMOM);
    int lastp = -1;

    // the reference raises mid-loop on 1/ierr^2 == inf and the caller's record
    // keeps its partial sums; here the record is left untouched, so the passes
    // first run into scratch totals and are stored only when none failed
    if constexpr (NMOM == 6) {
        wsums_pass<6, 0, 6, true>(src, ge, ng, vcen, ucen, maxrad2, sh, tot, lastp);
        if (sh.err == 0) wsums_store<6, 0, 6, true>(tot, resbase);
    } else {
        wsums17_single_pass(src, ge, ng, vcen, ucen, maxrad2, sh, resbase, lastp);
    }
    if (sh.err == 0 && lastp >= 0 && lane == 0) {
        // F of the last pixel used is left in the record (scratch field)
        const int row = lastp / src.ncol, col = lastp - row * src.ncol;
        double v, u;
        jacobian_vu(src.jac, (double)row, (double)col, v, u);
        const double vmod = v - vcen, umod = u - ucen;
        double F[NMOM];
        wsums_F<NMOM>(vmod, umod, v, u, umod * umod + vmod * vmod, F);
        double *r_F = (double *)(resbase + 16) + NMOM + NMOM * NMOM + NMOM;
        for (int i = 0; i < NMOM; i++) r_F[i] = F[i];
    }
    if (lane == 0 && status) status[s] = sh.err;
}

static size_t wsums_lds(int nmom, int ng)
{
    return (size_t)WS_CHUNK * (nmom + 3) * 8 + (WS_CHUNK + 2) * 4 +
           (size_t)ng * sizeof(EvalGauss) + 16;
}

__global__ __launch_bounds__(BLOCK) void weighted_sums_grid_kernel(
    const ngmix_stamp *stamps, const double *val, const double *ierr,
    const ngmix_jacobian *jacs, const ngmix_gauss2d *gmix, char *res, int nmom,
    const double *maxrad, int32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int s = blockIdx.x;
    const ngmix_stamp st = stamps[s];
    GridSrc src;
    src.val = val + st.pix_off;
    src.ierr = ierr + st.pix_off;
    src.jac = jacs[s];
    src.area = src.jac.scale * src.jac.scale;
    src.nrow = st.nrow;
    src.ncol = st.ncol;
    src.izw = (st.flags & NGMIX_STAMP_IGNORE_ZERO_WEIGHT) != 0;
    weighted_sums_body(src, gmix + st.gm_off, st.ngauss,
                       res + (size_t)s * NGMIX_MOMENTS_RESULT_BYTES(nmom), nmom,
                       maxrad[s], status ? status + s : nullptr, smem);
}

__global__ __launch_bounds__(BLOCK) void weighted_sums_list_kernel(
    const ngmix_gauss2d *wt, int ng, const ngmix_pixel *pixels, int n, char *res,
    int nmom, double maxrad, int32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ListSrc src;
    src.pix = pixels;
    src.n = n;
    weighted_sums_body(src, wt, ng, res, nmom, maxrad, status, smem);
}

int launch_weighted_sums_grid(const ngmix_batch *b, const ngmix_gauss2d *gmix,
                              void *res, int nmom, const double *maxrad,
                              int32_t *status, hipStream_t s)
{
    if (b->nstamps <= 0) return NGMIX_OK;
    if (nmom != 6 && nmom != 17) return NGMIX_ERR_BAD_ARG;
    const int ng = b->max_ngauss > 0 ? b->max_ngauss : 1;
    if (!(b->flags & NGMIX_BATCH_EXACT)) {
        const size_t lds = (size_t)ng * sizeof(EvalGauss) + 16;
        census(nmom == 6 ? "wsums_wave_kernel<6>" : "wsums_wave_kernel<17>");
        if (nmom == 6)
            hipLaunchKernelGGL(wsums_wave_kernel<6>, dim3((unsigned)b->nstamps),
                               dim3(WAVE), lds, s, b->stamps, b->val, b->ierr, b->jac,
                               gmix, (char *)res, maxrad, status);
        else
            hipLaunchKernelGGL(wsums_wave_kernel<17>, dim3((unsigned)b->nstamps),
                               dim3(WAVE), lds, s, b->stamps, b->val, b->ierr, b->jac,
                               gmix, (char *)res, maxrad, status);
        NGMIX_HIP_CHECK(hipGetLastError());
        return NGMIX_OK;
    }
    census("weighted_sums_grid_kernel");
    hipLaunchKernelGGL(weighted_sums_grid_kernel, dim3((unsigned)b->nstamps),
                       dim3(BLOCK), wsums_lds(nmom, ng), s, b->stamps, b->val,
                       b->ierr, b->jac, gmix, (char *)res, nmom, maxrad, status);
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

int launch_weighted_sums_list(const ngmix_gauss2d *wt, int ng,
                              const ngmix_pixel *pixels, int64_t n, void *res,
                              int nmom, double maxrad, int32_t *status,
                              hipStream_t s)
{
    if (nmom != 6 && nmom != 17) return NGMIX_ERR_BAD_ARG;
    hipLaunchKernelGGL(weighted_sums_list_kernel, dim3(1), dim3(BLOCK),
                       wsums_lds(nmom, ng), s, wt, ng, pixels, (int)n, (char *)res,
                       nmom, maxrad, status);
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

// ===========================================================================
// adaptive moments
// ===========================================================================
//
// The reference runs, per iteration, a 3-sum centroid pass and a moments pass
// that also accumulates a 7x7 covariance (49 sums) which only the LAST pass's
// values survive in the result.  Here the iteration passes accumulate just the
// 7 moment sums the iteration logic needs; the 49 covariance sums are computed
// once, after the loop, for the weight and centre of the last moments pass
// executed -- the same per-pixel terms the reference's last pass adds, so the
// result record is the same while the hot loop does ~1/3 of the flops.
// Scalar iteration logic (admom_nb.py:33-108) runs on thread 0 out of LDS.

struct AdmomShared {
    double tab[16];
    double red_scratch[NWAVES * 64];  // sized for the widest work-group
    double red_out[64];
    ngmix_gauss2d wt;       // current weight
    ngmix_gauss2d wt_used;  // weight of the last moments pass
    ngmix_admom_result res;
    int stop;       // loop exit requested
    int status;     // C-ABI status (range error / zero division)
    int mom_ran;    // a moments pass has been executed
    int res_is_mom; // res.sums currently hold a moments pass
    int iscratch[NWAVES + 4];
};

// clear_result, admom_nb.py:229-239
__device__ __forceinline__ void admom_clear(ngmix_admom_result &r)
{
    r.npix = 0;
    r.wsum = 0.0;
    for (int i = 0; i < 7; i++) r.sums[i] = 0.0;
    for (int i = 0; i < 49; i++) r.sums_cov[i] = 0.0;
    for (int i = 0; i < 6; i++) r.pars[i] = NAN;
    r.rho4 = NAN;
}

// deweight_moments, admom_nb.py:178-226
__device__ __forceinline__ void admom_deweight(ngmix_gauss2d &wt, double Irr,
                                               double Irc, double Icc,
                                               ngmix_admom_result &res)
{
    const double detm = Irr * Icc - Irc * Irc;
    if (detm <= LOW_DETVAL) {
        res.flags = NGMIX_FLAG_LOW_DET;
        return;
    }
    const double Wrr = wt.irr, Wrc = wt.irc, Wcc = wt.icc;
    const double detw = Wrr * Wcc - Wrc * Wrc;
    if (detw <= LOW_DETVAL) {
        res.flags = NGMIX_FLAG_LOW_DET;
        return;
    }
    const double idetw = 1.0 / detw;
    const double idetm = 1.0 / detm;
    const double Nrr = Icc * idetm - Wcc * idetw;
    const double Ncc = Irr * idetm - Wrr * idetw;
    const double Nrc = -Irc * idetm + Wrc * idetw;
    const double detn = Nrr * Ncc - Nrc * Nrc;
    if (detn <= LOW_DETVAL) {
        res.flags = NGMIX_FLAG_LOW_DET;
        return;
    }
    const double idetn = 1. / detn;
    wt.irr = Ncc * idetn;
    wt.icc = Nrr * idetn;
    wt.irc = -Nrc * idetn;
    wt.det = wt.irr * wt.icc - wt.irc * wt.irc;
}

// NT threads cooperate on one stamp (64, 128 or 256: fewer waves mean fewer
// reduction instructions and no cross-wave barriers, more waves mean fewer
// pixels per thread to keep in registers)
template <class Src, int NT, int PPT>
__device__ __forceinline__ void admom_body(const Src &src,
                                           const ngmix_admom_conf conf,
                                           ngmix_gauss2d *wt_io,
                                           ngmix_admom_result *res_io,
                                           int32_t *status, AdmomShared &sh)
{
    const int tid = threadIdx.x;
    PixCache<Src, NT, PPT> cache;
    cache.fill(src);

    // the moments pass divides by ierr^2 for every listed pixel
    // (admom_nb.py:146): any zero there is numba's ZeroDivisionError
    int my_last = -1, my_zero = 0;
    cache.for_each(src, [&](double, double, double, double, double ierr, int p) {
        if (ierr * ierr == 0.0) my_zero = 1;
        my_last = p > my_last ? p : my_last;
    });
    const int last_pos = group_max_int<NT>(my_last, sh.iscratch);
    const int has_zero = group_max_int<NT>(my_zero, sh.iscratch);

    if (tid < 16) sh.tab[tid] = c_exp_table_m[tid];
    if (tid == 0) {
        sh.wt = wt_io[0];
        sh.res = res_io[0];
        sh.stop = 0;
        sh.status = NGMIX_OK;
        sh.mom_ran = 0;
        sh.res_is_mom = 0;
    }
    __syncthreads();

    const double roworig = sh.wt.row, colorig = sh.wt.col;
    double e1old = NAN, e2old = NAN, Told = NAN;  // used by thread 0 only
    int iter_index = -1;

    for (int it = 0; it < conf.maxiter; it++) {
        iter_index = it;
        if (tid == 0) {
            if (sh.wt.det < LOW_DETVAL) {
                sh.res.flags = NGMIX_FLAG_LOW_DET;
                sh.stop = 1;
            } else {
                const int st = gauss_set_norm(sh.wt);
                if (st) {  // GMixRangeError("T too low") escapes admom()
                    sh.status = st;
                    sh.stop = 1;
                }
            }
        }
        __syncthreads();
        if (sh.stop) break;

        // ---- centroid pass (admom_censums, admom_nb.py:111-128)
        {
            const EvalGauss e = make_eval(sh.wt);
            double a[4] = {0.0, 0.0, 0.0, 0.0};
            cache.for_each(src, [&](double v, double u, double area, double val,
                                    double, int) {
                const double weight = gauss_eval_fast(e, v, u, area, sh.tab);
                const double wdata = weight * val;
                a[3] += 1.0;
                a[0] += wdata * v;
                a[1] += wdata * u;
                a[2] += wdata;
            });
            group_sum<NT, 4>(a, sh.red_scratch, sh.red_out);
        }
        if (tid == 0) {
            ngmix_admom_result &res = sh.res;
            admom_clear(res);
            sh.res_is_mom = 0;
            res.npix = (int32_t)sh.red_out[3];
            res.sums[0] = sh.red_out[0];
            res.sums[1] = sh.red_out[1];
            res.sums[5] = sh.red_out[2];
            if (res.sums[5] <= 0.0) {
                res.flags = NGMIX_FLAG_NONPOS_FLUX;
                sh.stop = 1;
            } else {
                sh.wt.row = res.sums[0] / res.sums[5];
                sh.wt.col = res.sums[1] / res.sums[5];
                if (fabs(sh.wt.row - roworig) > conf.shiftmax ||
                    fabs(sh.wt.col - colorig) > conf.shiftmax) {
                    res.flags = NGMIX_FLAG_CEN_SHIFT;
                    sh.stop = 1;
                } else if (has_zero) {
                    sh.status = NGMIX_ERR_ZERO_DIV;
                    sh.stop = 1;
                }
            }
        }
        __syncthreads();
        if (sh.stop) break;

        // ---- moments pass without the covariance (admom_momsums, :131-175)
        {
            const EvalGauss e = make_eval(sh.wt);
            const double vcen = sh.wt.row, ucen = sh.wt.col;
            double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            cache.for_each(src, [&](double v, double u, double area, double val,
                                    double, int) {
                const double weight = gauss_eval_fast(e, v, u, area, sh.tab);
                const double vmod = v - vcen, umod = u - ucen;
                const double wdata = weight * val;
                const double chi2 = e.dcc * vmod * vmod + e.drr * umod * umod -
                                    e.drc2 * vmod * umod;
                a[0] += wdata * v;
                a[1] += wdata * u;
                a[2] += wdata * (umod * umod - vmod * vmod);
                a[3] += wdata * (2 * vmod * umod);
                a[4] += wdata * (umod * umod + vmod * vmod);
                a[5] += wdata * 1.0;
                a[6] += wdata * (chi2 * chi2);
                a[7] += weight;
                a[8] += 1.0;
            });
            group_sum<NT, 9>(a, sh.red_scratch, sh.red_out);
        }
        if (tid == 0) {
            ngmix_admom_result &res = sh.res;
            admom_clear(res);
            for (int i = 0; i < 7; i++) res.sums[i] = sh.red_out[i];
            res.wsum = sh.red_out[7];
            res.npix = (int32_t)sh.red_out[8];
            sh.wt_used = sh.wt;
            sh.mom_ran = 1;
            sh.res_is_mom = 1;

            if (res.sums[5] <= 0.0) {
                res.flags = NGMIX_FLAG_NONPOS_FLUX;
                sh.stop = 1;
            } else {
                const double finv = 1.0 / res.sums[5];
                const double M1 = res.sums[2] * finv;
                const double M2 = res.sums[3] * finv;
                const double T = res.sums[4] * finv;
                const double Irr = 0.5 * (T - M1);
                const double Icc = 0.5 * (T + M1);
                const double Irc = 0.5 * M2;
                if (T <= 0.0) {
                    res.flags = NGMIX_FLAG_NONPOS_SIZE;
                    sh.stop = 1;
                } else {
                    const double e1 = (Icc - Irr) / T;
                    const double e2 = 2 * Irc / T;
                    if ((fabs(e1 - e1old) < conf.etol) &&
                        (fabs(e2 - e2old) < conf.etol) &&
                        (fabs(T / Told - 1.) < conf.Ttol)) {
                        res.pars[0] = sh.wt.row;
                        res.pars[1] = sh.wt.col;
                        res.pars[2] = sh.wt.icc - sh.wt.irr;
                        res.pars[3] = 2.0 * sh.wt.irc;
                        res.pars[4] = sh.wt.icc + sh.wt.irr;
                        res.pars[5] = 1.0;
                        res.rho4 = res.sums[6] / res.sums[5];
                        sh.stop = 1;
                    } else {
                        if (!conf.cenonly) {
                            admom_deweight(sh.wt, Irr, Irc, Icc, res);
                            if (res.flags != 0) sh.stop = 1;
                        }
                        e1old = e1;
                        e2old = e2;
                        Told = T;
                    }
                }
            }
        }
        __syncthreads();
        if (sh.stop) break;
    }

    // ---- the covariance of the last moments pass, and the F scratch
    if (sh.mom_ran) {
        const EvalGauss e = make_eval(sh.wt_used);
        const double vcen = sh.wt_used.row, ucen = sh.wt_used.col;
        const bool want_cov = sh.res_is_mom != 0 && sh.status == NGMIX_OK && !conf.no_cov;
        double c[49];
#pragma unroll
        for (int i = 0; i < 49; i++) c[i] = 0.0;
        cache.for_each(src, [&](double v, double u, double area, double val,
                                double ierr, int p) {
            const double vmod = v - vcen, umod = u - ucen;
            const double chi2 = e.dcc * vmod * vmod + e.drr * umod * umod -
                                e.drc2 * vmod * umod;
            double F[7];
            F[0] = v;
            F[1] = u;
            F[2] = umod * umod - vmod * vmod;
            F[3] = 2 * vmod * umod;
            F[4] = umod * umod + vmod * vmod;
            F[5] = 1.0;
            F[6] = chi2 * chi2;
            if (p == last_pos) {
#pragma unroll
                for (int i = 0; i < 7; i++) sh.res.F[i] = F[i];
            }
            if (want_cov) {
                const double weight = gauss_eval_fast(e, v, u, area, sh.tab);
                const double var = 1.0 / (ierr * ierr);
                const double w2var = weight * weight * var;
                (void)val;
#pragma unroll
                for (int i = 0; i < 7; i++) {
                    const double t = w2var * F[i];
#pragma unroll
                    for (int j = 0; j < 7; j++) c[i * 7 + j] += t * F[j];
                }
            }
        });
        if (want_cov) {
            // reduce in four slabs to bound the LDS scratch
            double part[13];
#pragma unroll
            for (int slab = 0; slab < 4; slab++) {
#pragma unroll
                for (int k = 0; k < 13; k++)
                    part[k] = (slab * 13 + k < 49) ? c[(slab * 13 + k) % 49] : 0.0;
                group_sum<NT, 13>(part, sh.red_scratch, sh.red_out);
                if (tid == 0) {
                    for (int k = 0; k < 13; k++)
                        if (slab * 13 + k < 49)
                            sh.res.sums_cov[slab * 13 + k] = sh.red_out[k];
                }
                __syncthreads();
            }
        }
    }
    __syncthreads();

    if (tid == 0) {
        // admom_nb.py:105-108.  With maxiter <= 0 the reference's loop
        // variable is undefined; numiter = 0 is used (=> MAXITER if 0 == maxiter)
        sh.res.numiter = iter_index + 1;
        if (sh.res.numiter == conf.maxiter) sh.res.flags = NGMIX_FLAG_MAXITER;
        res_io[0] = sh.res;
        wt_io[0] = sh.wt;
        if (status) *status = sh.status;
    }
}

// ---------------------------------------------------------------------------
// batch form: register-resident, barrier-light adaptive moments
// ---------------------------------------------------------------------------
// The stamp's (v, u, val) live in registers for the whole kernel (PPT pixels
// per thread), the exp table in LDS.  Every lane carries the full iteration
// state (weight, sums, flags) redundantly in registers: wave reductions are
// DPP adds returning uniform totals, so a one-wave work-group runs the whole
// fit without a single barrier or LDS round trip, and wider groups need one
// barrier per reduction (double-buffered partial slots).  Pixel arithmetic
// uses FMAs (the sums are tree-reduced, so bitwise agreement with the
// reference's sequential loop is not available anyway; 1e-10 is the bar);
// the scalar iteration logic keeps the reference's operations and order.
// Non-listed and out-of-range pixel slots hold val = 0, so every wdata-
// weighted sum ignores them without a branch; wsum uses the kept bit.

struct AdmomFusedShared {
    double tab[16];    // exp(-n), n = 0..15 (fused evaluator)
    double slots[2][NWAVES][32];
    int iscratch[NWAVES + 4];
};

#define NGMIX_UNIFORM(c) (__builtin_amdgcn_readfirstlane((int)(c)) != 0)

template <int NT, int NV>
__device__ __forceinline__ void group_total(double (&a)[NV],
                                            double (*slots)[NWAVES][32], int &phase,
                                            double *red = nullptr, double *tot = nullptr)
{
    constexpr int NW = NT / WAVE;
    // (a transposed LDS tile -- wave_reduce_lds, as the EM kernel uses -- was
    // 1.7x SLOWER here: this loop is a short serial chain per iteration and the
    // LDS round trips cost more latency than the DPP trees cost issue slots)
    (void)red;
    (void)tot;
    // (wave_total4 -- four sums per register through permlane swaps, 7
    // instructions per sum instead of 20 -- was measured here too: 3.85 against
    // 3.75 ms.  Its one long dependency chain loses to independent trees the
    // same way the LDS tile does.)
#pragma unroll
    for (int i = 0; i < NV; i++) a[i] = wave_total(a[i]);
    if (NW > 1) {
        const int w = wave_id();
        double(*buf)[32] = slots[phase];
        if (lane_id() == 0) {
#pragma unroll
            for (int i = 0; i < NV; i++) buf[w][i] = a[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NV; i++) {
            double t = buf[0][i];
#pragma unroll
            for (int k = 1; k < NW; k++) t += buf[k][i];
            a[i] = t;
        }
        phase ^= 1;
    }
}

template <int NT>
__device__ __forceinline__ int group_sum_int(int x, int *scratch)
{
    constexpr int NW = NT / WAVE;
    x = wave_sum_int(x);
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6;
    if (lane == 0) scratch[w] = x;
    __syncthreads();
    int m = scratch[0];
#pragma unroll
    for (int k = 1; k < NW; k++) m += scratch[k];
    __syncthreads();
    return m;
}

// gauss2d_eval_pixel_fast with FMAs on a precomputed chi2; pa = pnorm*area
// 0 <= chi2 < 25 (admom_nb.py:117-118, 141-142) as ONE unsigned compare on the
// high word: negative, nan and inf fail it as they fail the reference's two
// compares (chi2 is a positive-definite form here: never -0.0)
__device__ __forceinline__ bool chi2_in_cut(double chi2)
{
    return (unsigned)__double2hiint(chi2) < 0x40390000u;   // 25.0
}

__device__ __forceinline__ double weight_fused(double chi2, double pa,
                                               const double *tabr, const FexpCoef &K)
{
    double w = 0.0;
    if (chi2_in_cut(chi2)) {
        double e = fexp_neg_fused(0.5 * chi2, tabr, K);
        // (>= 20.0: the window is exactly 1 at chi2 == 20)
        if ((unsigned)__double2hiint(chi2) >= 0x40340000u) {
            const double au = (MAX_CHI2 - chi2) * APOD_IWIDTH;
            const double aq = fma(au, fma(au, 6.0, -15.0), 10.0);
            e *= (au * au) * (au * aq);
        }
        w = pa * e;
    }
    return w;
}

// ALLCHUNKS: all PPT register slots of every thread are evaluated (a 32x32
// stamp on one wave fills them; an empty slot holds val == 0 with kept bit 0
// and a finite weight, so it adds exactly nothing): the pixel loops then have
// a compile-time trip count and unroll fully.  With a run-time count the compiler keeps them rolled and
// reaches pv[k] / pu[k] / pval[k] through relative register addressing
// (s_set_gpr_idx + six v_mov_b32 per pixel per pass).
template <int NT, int PPT, bool ALLCHUNKS>
__device__ __forceinline__ void admom_fused_body(const GridSrc &src,
                                                 const ngmix_admom_conf conf,
                                                 ngmix_gauss2d *wt_io,
                                                 ngmix_admom_result *res_io,
                                                 int32_t *status,
                                                 AdmomFusedShared &sh, double *lds_ierr)
{
    const int tid = threadIdx.x;
    const int n = src.count();
    const int nchunk = (n + NT - 1) / NT;  // uniform

    // ---- the stamp, once from HBM
    double pv[PPT], pu[PPT], pval[PPT];
    // one kept bit per slot, slots 32 .. PPT-1 in a second word
    unsigned kept = 0u, kept_hi = 0u;
    int my_last = -1, my_zero = 0;
    // The loop below asks for one slot's val / ierr and tests ierr at once: left
    // alone that is sixteen to thirty-six SERIAL round trips to HBM per stamp.
    // This pass touches every slot first -- independent loads, issued back to
    // back, folded into a sum nobody needs -- so that the loop finds its lines in
    // cache.  (Holding the values themselves in registers until they are used
    // was tried: the run-time indexed pixel loops of the 16-slot kernels then
    // spill to private memory.)
    {
        double touch = 0.0;
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            const int p = tid + k * NT;
            const int pc = p < n ? p : 0;
            touch += src.val[pc] + src.ierr[pc];
        }
        if (touch == 1.2345678e300) my_zero = 1;   // (never: keeps the loads alive)
    }
#pragma unroll
    for (int k = 0; k < PPT; k++) {
        const int p = tid + k * NT;
        pv[k] = pu[k] = pval[k] = 0.0;
        if (p < n) {
            double a, val, ierr;
            if (src.load(p, pv[k], pu[k], a, val, ierr)) {
                // ierr waits in LDS for the covariance pass (read back from HBM
                // there, each load was waited for on the spot)
                lds_ierr[k * NT + tid] = ierr;
                if (k < 32) kept |= 1u << k;
                else kept_hi |= 1u << (k - 32);
                pval[k] = val;
                my_last = p;
                if (ierr * ierr == 0.0) my_zero = 1;  // admom_nb.py:146
            }
        }
    }
    // slots (of this wave) that hold a non-finite value somewhere: the reference
    // forms weight * val for every pixel, so a NaN / inf outside the weight's
    // cut is 0 * NaN = NaN in its sums (and flags the object) -- such a slot is
    // never skipped by the all-lanes-outside-the-cut test below
    unsigned long long nf_slots = 0ull;
#pragma unroll
    for (int k = 0; k < PPT; k++)
        if (__ballot(!(fabs(pval[k]) < INFINITY)) != 0ull) nf_slots |= 1ull << k;
    if (tid < 16) sh.tab[tid] = c_exp_table_m[15 - tid];  // exp(-n)
    const FexpCoef K = load_fexp_coef(c_fexp_coef_m);
    const int last_pos = group_max_int<NT>(my_last, sh.iscratch);
    const int has_zero = group_max_int<NT>(my_zero, sh.iscratch);
    const int npix = group_sum_int<NT>(__popc(kept) + __popc(kept_hi), sh.iscratch);
    const double area = src.area;
    int phase = 0;

    // ---- uniform iteration state, replicated in every lane
    ngmix_gauss2d wt = wt_io[0];
    const double roworig = wt.row, colorig = wt.col;
    double e1old = NAN, e2old = NAN, Told = NAN;
    int flags = res_io[0].flags, st = NGMIX_OK;
    int last_kind = 0;  // 0: no pass ran, 1: centroid pass, 2: moments pass
    bool mom_ran = false;
    int iter_index = -1;
    double sums[7] = {0, 0, 0, 0, 0, 0, 0}, wsum = 0.0;
    double pars[6] = {NAN, NAN, NAN, NAN, NAN, NAN}, rho4 = NAN;
    double used_row = 0, used_col = 0, used_dcc = 0, used_drr = 0, used_drc2 = 0,
           used_pa = 0;

    for (int it = 0; it < conf.maxiter; it++) {
        iter_index = it;
        if (NGMIX_UNIFORM(wt.det < LOW_DETVAL)) {  // admom_nb.py:40-42
            flags = NGMIX_FLAG_LOW_DET;
            break;
        }
        const int sn = gauss_set_norm(wt);
        if (NGMIX_UNIFORM(sn != 0)) {  // GMixRangeError escapes admom()
            st = sn;
            break;
        }
        const double dcc = wt.dcc, drr = wt.drr, mdrc2 = -2.0 * wt.drc;
        const double pa = wt.pnorm * area;

        // ---- centroid pass (admom_censums, admom_nb.py:111-128)
        {
            const double row = wt.row, col = wt.col;
            double a[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < PPT; k++) {
                if (!ALLCHUNKS && k >= nchunk) break;
                const double vd = pv[k] - row, ud = pu[k] - col;
                const double chi2 =
                    fma(dcc * vd, vd, fma(drr * ud, ud, (mdrc2 * vd) * ud));
                // a slot no lane of which is inside the weight's cut adds exact
                // zeros to every sum: skipped whole (the rows beyond 5 sigma)
                if (__ballot(chi2_in_cut(chi2)) == 0ull && !((nf_slots >> k) & 1ull))
                    continue;
                const double wdata = weight_fused(chi2, pa, sh.tab, K) * pval[k];
                a[0] = fma(wdata, pv[k], a[0]);
                a[1] = fma(wdata, pu[k], a[1]);
                a[2] += wdata;
                // one pixel at a time: left alone, the compiler evaluates all
                // sixteen weights first and sinks the accumulations below them
                // (409 registers -> one wave per SIMD); an opaque use of the
                // accumulators ends each pixel where it is written
                if (ALLCHUNKS) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]));
            }
            group_total<NT, 3>(a, sh.slots, phase);
            last_kind = 1;
#pragma unroll
            for (int i = 0; i < 6; i++) pars[i] = NAN;
            rho4 = NAN;
            sums[0] = a[0];
            sums[1] = a[1];
            sums[5] = a[2];
        }
        if (NGMIX_UNIFORM(sums[5] <= 0.0)) {
            flags = NGMIX_FLAG_NONPOS_FLUX;
            break;
        }
        wt.row = sums[0] / sums[5];
        wt.col = sums[1] / sums[5];
        if (NGMIX_UNIFORM(fabs(wt.row - roworig) > conf.shiftmax ||
                          fabs(wt.col - colorig) > conf.shiftmax)) {
            flags = NGMIX_FLAG_CEN_SHIFT;
            break;
        }
        if (has_zero) {
            st = NGMIX_ERR_ZERO_DIV;
            break;
        }

        // ---- moments pass without the covariance (admom_momsums, :131-175)
        {
            const double row = wt.row, col = wt.col;
            double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < PPT; k++) {
                if (!ALLCHUNKS && k >= nchunk) break;
                const double vd = pv[k] - row, ud = pu[k] - col;
                const double vv = vd * vd, uu = ud * ud, vu = vd * ud;
                const double chi2 = fma(dcc, vv, fma(drr, uu, mdrc2 * vu));
                if (__ballot(chi2_in_cut(chi2)) == 0ull && !((nf_slots >> k) & 1ull))
                    continue;
                const double weight = weight_fused(chi2, pa, sh.tab, K);
                const double wdata = weight * pval[k];
                a[0] = fma(wdata, pv[k], a[0]);
                a[1] = fma(wdata, pu[k], a[1]);
                a[2] = fma(wdata, uu - vv, a[2]);
                a[3] = fma(wdata, vu + vu, a[3]);
                a[4] = fma(wdata, uu + vv, a[4]);
                a[5] += wdata;
                a[6] = fma(wdata, chi2 * chi2, a[6]);
                // (the bit is extracted by a volatile asm: as plain C the
                // sixteen (double) conversions are hoisted out of the iteration
                // loop and pin 32 registers)
                const unsigned kword = k < 32 ? kept : kept_hi;
                const int kpos = k & 31;
                unsigned kbit = (kword >> kpos) & 1u;
                if (ALLCHUNKS)
                    asm volatile("v_bfe_u32 %0, %1, %2, 1" : "=v"(kbit) : "v"(kword), "v"(kpos));
                a[7] = fma(weight, (double)kbit, a[7]);
                if (ALLCHUNKS)
                    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]),
                                      "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
            }
            group_total<NT, 8>(a, sh.slots, phase);
            last_kind = 2;
            mom_ran = true;
#pragma unroll
            for (int i = 0; i < 7; i++) sums[i] = a[i];
            wsum = a[7];
            used_row = row;
            used_col = col;
            used_dcc = dcc;
            used_drr = drr;
            used_drc2 = mdrc2;
            used_pa = pa;
        }
        if (NGMIX_UNIFORM(sums[5] <= 0.0)) {
            flags = NGMIX_FLAG_NONPOS_FLUX;
            break;
        }
        {
            const double finv = 1.0 / sums[5];
            const double M1 = sums[2] * finv;
            const double M2 = sums[3] * finv;
            const double T = sums[4] * finv;
            const double Irr = 0.5 * (T - M1);
            const double Icc = 0.5 * (T + M1);
            const double Irc = 0.5 * M2;
            if (NGMIX_UNIFORM(T <= 0.0)) {
                flags = NGMIX_FLAG_NONPOS_SIZE;
                break;
            }
            const double e1 = (Icc - Irr) / T;
            const double e2 = 2 * Irc / T;
            if (NGMIX_UNIFORM((fabs(e1 - e1old) < conf.etol) &&
                              (fabs(e2 - e2old) < conf.etol) &&
                              (fabs(T / Told - 1.) < conf.Ttol))) {
                pars[0] = wt.row;
                pars[1] = wt.col;
                pars[2] = wt.icc - wt.irr;
                pars[3] = 2.0 * wt.irc;
                pars[4] = wt.icc + wt.irr;
                pars[5] = 1.0;
                rho4 = sums[6] / sums[5];
                break;
            }
            if (!conf.cenonly) {
                // deweight_moments, admom_nb.py:178-226
                const double detm = Irr * Icc - Irc * Irc;
                if (NGMIX_UNIFORM(detm <= LOW_DETVAL)) {
                    flags = NGMIX_FLAG_LOW_DET;
                    break;
                }
                const double Wrr = wt.irr, Wrc = wt.irc, Wcc = wt.icc;
                const double detw = Wrr * Wcc - Wrc * Wrc;
                if (NGMIX_UNIFORM(detw <= LOW_DETVAL)) {
                    flags = NGMIX_FLAG_LOW_DET;
                    break;
                }
                const double idetw = 1.0 / detw;
                const double idetm = 1.0 / detm;
                const double Nrr = Icc * idetm - Wcc * idetw;
                const double Ncc = Irr * idetm - Wrr * idetw;
                const double Nrc = -Irc * idetm + Wrc * idetw;
                const double detn = Nrr * Ncc - Nrc * Nrc;
                if (NGMIX_UNIFORM(detn <= LOW_DETVAL)) {
                    flags = NGMIX_FLAG_LOW_DET;
                    break;
                }
                const double idetn = 1. / detn;
                wt.irr = Ncc * idetn;
                wt.icc = Nrr * idetn;
                wt.irc = -Nrc * idetn;
                wt.det = wt.irr * wt.icc - wt.irc * wt.irc;
            }
            e1old = e1;
            e2old = e2;
            Told = T;
        }
    }

    // ---- covariance of the last moments pass (admom_nb.py:168-175) and the
    // F scratch of its last pixel.  sums_cov[i,j] and [j,i] differ in the
    // reference only by the rounding of (w2var*F[i])*F[j] vs (w2var*F[j])*F[i];
    // one triangle is accumulated and mirrored.
    const bool want_cov = last_kind == 2 && st == NGMIX_OK && !conf.no_cov;
    double c[28];
#pragma unroll
    for (int i = 0; i < 28; i++) c[i] = 0.0;
    if (mom_ran) {
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            if (!ALLCHUNKS && k >= nchunk) break;
            const int p = tid + k * NT;
            const double vd = pv[k] - used_row, ud = pu[k] - used_col;
            const double vv = vd * vd, uu = ud * ud, vu = vd * ud;
            const double chi2 = fma(used_dcc, vv, fma(used_drr, uu, used_drc2 * vu));
            double F[7];
            F[0] = pv[k];
            F[1] = pu[k];
            F[2] = uu - vv;
            F[3] = vu + vu;
            F[4] = uu + vv;
            F[5] = 1.0;
            F[6] = chi2 * chi2;
            if (p == last_pos) {
#pragma unroll
                for (int i = 0; i < 7; i++) res_io[0].F[i] = F[i];
            }
            if (want_cov) {
                const double weight = weight_fused(chi2, used_pa, sh.tab, K);
                double w2var = 0.0;
                if (((k < 32 ? kept : kept_hi) >> (k & 31)) & 1u) {
                    // (v_rcp + Newton, ~1 ulp: the fused kernels' contract; the
                    // IEEE division was a quarter of this pass)
                    const double ierr = lds_ierr[k * NT + tid];
                    w2var = weight * weight * rcp_newton(ierr * ierr);
                }
                int idx = 0;
#pragma unroll
                for (int i = 0; i < 7; i++) {
                    const double t = w2var * F[i];
#pragma unroll
                    for (int j = i; j < 7; j++) {
                        c[idx] = fma(t, F[j], c[idx]);
                        idx++;
                    }
                }
            }
            if (ALLCHUNKS) {
#pragma unroll
                for (int i = 0; i < 28; i++) asm volatile("" : "+v"(c[i]));
            }
        }
        if (want_cov) group_total<NT, 28>(c, sh.slots, phase);
    }

    if (tid == 0) {
        ngmix_admom_result *r = res_io;
        // admom_nb.py:105-108; maxiter <= 0 leaves the loop variable undefined
        // in the reference, numiter = 0 here
        const int numiter = iter_index + 1;
        if (numiter == conf.maxiter) flags = NGMIX_FLAG_MAXITER;
        r->flags = flags;
        r->numiter = numiter;
        if (last_kind != 0) {
            r->npix = npix;
            r->wsum = last_kind == 2 ? wsum : 0.0;
#pragma unroll
            for (int i = 0; i < 7; i++)
                r->sums[i] = (last_kind == 2 || i == 0 || i == 1 || i == 5) ? sums[i] : 0.0;
            int idx = 0;
#pragma unroll
            for (int i = 0; i < 7; i++) {
#pragma unroll
                for (int j = i; j < 7; j++) {
                    const double x = want_cov ? c[idx] : 0.0;
                    r->sums_cov[i * 7 + j] = x;
                    r->sums_cov[j * 7 + i] = x;
                    idx++;
                }
            }
#pragma unroll
            for (int i = 0; i < 6; i++) r->pars[i] = pars[i];
            r->rho4 = rho4;
        }
        wt_io[0] = wt;
        if (status) *status = st;
    }
}

template <int NT, int PPT>
__global__ __launch_bounds__(NT) void admom_grid_kernel(
    ngmix_admom_conf conf, const ngmix_stamp *stamps, const double *val,
    const double *ierr, const ngmix_jacobian *jacs, ngmix_gauss2d *wt,
    ngmix_admom_result *res, int32_t *status)
{
    const int s = blockIdx.x;
    const ngmix_stamp st = stamps[s];
    GridSrc src;
    src.val = val + st.pix_off;
    src.ierr = ierr + st.pix_off;
    src.jac = jacs[s];
    src.area = src.jac.scale * src.jac.scale;
    src.nrow = st.nrow;
    src.ncol = st.ncol;
    src.izw = (st.flags & NGMIX_STAMP_IGNORE_ZERO_WEIGHT) != 0;
    if constexpr (PPT > 0) {
        __shared__ AdmomFusedShared sh;
        __shared__ double lds_ierr[NT * PPT];
        // (nearly) every register slot holds a pixel: evaluate all PPT slots --
        // an empty slot has val == 0 and kept bit 0 and adds exactly nothing
        const int nchunk = (st.nrow * st.ncol + NT - 1) / NT;
        // More than sixteen slots per lane (36 x 64 and 18 x 128 = 48 x 48, the
        // north-star stamp): only the compile-time form -- the run-time indexed
        // one would put its arrays in private memory.
        constexpr bool kOnlyUnrolled = PPT > 16;
        if constexpr (kOnlyUnrolled) {
            admom_fused_body<NT, PPT, true>(src, conf, wt + st.gm_off, res + s,
                                            status ? status + s : nullptr, sh, lds_ierr);
        } else {
            if (NT == WAVE && nchunk * 8 >= PPT * 7)
                admom_fused_body<NT, PPT, NT == WAVE>(src, conf, wt + st.gm_off, res + s,
                                                status ? status + s : nullptr, sh, lds_ierr);
            else
                admom_fused_body<NT, PPT, false>(src, conf, wt + st.gm_off, res + s,
                                                 status ? status + s : nullptr, sh, lds_ierr);
        }
    } else {
        // stamps too large for registers: streaming passes, reference order
        __shared__ AdmomShared sh;
        admom_body<GridSrc, NT, 0>(src, conf, wt + st.gm_off, res + s,
                                   status ? status + s : nullptr, sh);
    }
}

__global__ __launch_bounds__(BLOCK) void admom_list_kernel(
    ngmix_admom_conf conf, const ngmix_pixel *pixels, int n, ngmix_gauss2d *wt,
    ngmix_admom_result *res, int32_t *status)
{
    __shared__ AdmomShared sh;
    ListSrc src;
    src.pix = pixels;
    src.n = n;
    admom_body<ListSrc, BLOCK, 0>(src, conf, wt, res, status, sh);
}

template <int NT, int PPT>
static void admom_launch(const ngmix_admom_conf *conf, const ngmix_batch *b,
                         ngmix_gauss2d *wt, ngmix_admom_result *res, int32_t *status,
                         hipStream_t s)
{
    {
        char name[64];
        snprintf(name, sizeof(name), "admom_grid_kernel<%d, %d>", NT, PPT);
        census(name);
    }
    hipLaunchKernelGGL((admom_grid_kernel<NT, PPT>), dim3((unsigned)b->nstamps),
                       dim3(NT), 0, s, *conf, b->stamps, b->val, b->ierr, b->jac, wt,
                       res, status);
}

int launch_admom_grid(const ngmix_admom_conf *conf, const ngmix_batch *b,
                      ngmix_gauss2d *wt, ngmix_admom_result *res, int32_t *status,
                      hipStream_t s)
{
    if (b->nstamps <= 0) return NGMIX_OK;
    // tuning hook: NGMIX_ADMOM_NT=64|128|256 forces the threads per stamp
    int nt = 0;
    if (const char *e = getenv("NGMIX_ADMOM_NT")) nt = atoi(e);
    const int np = b->max_npix;
    // (48 x 48 = 36 x 64 runs as ONE wave, every slot in registers: 9.6 ms per
    // 100k stamps against 11.6 with two waves and 18.7 with four; 64 x 64 = 32 x
    // 128 as two: 8.9 ms per 50k against 12.7 with four)
    if (nt == 0)
        nt = np <= 16 * 64 ? 64
             : (np <= 16 * 128 ? 128 : (np <= 36 * 64 ? 64 : (np <= 32 * 128 ? 128 : 256)));
    if (nt == 64 && np <= 8 * 64) admom_launch<64, 8>(conf, b, wt, res, status, s);
    else if (nt == 64 && np <= 16 * 64) admom_launch<64, 16>(conf, b, wt, res, status, s);
    else if (nt <= 128 && np <= 8 * 128) admom_launch<128, 8>(conf, b, wt, res, status, s);
    else if (nt <= 128 && np <= 16 * 128) admom_launch<128, 16>(conf, b, wt, res, status, s);
    else if (nt == 64 && np <= 36 * 64) admom_launch<64, 36>(conf, b, wt, res, status, s);
    else if (nt == 128 && np <= 32 * 128) admom_launch<128, 32>(conf, b, wt, res, status, s);
    else if (np <= 4 * BLOCK) admom_launch<BLOCK, 4>(conf, b, wt, res, status, s);
    else if (np <= 8 * BLOCK) admom_launch<BLOCK, 8>(conf, b, wt, res, status, s);
    else if (np <= 16 * BLOCK) admom_launch<BLOCK, 16>(conf, b, wt, res, status, s);
    else admom_launch<BLOCK, 0>(conf, b, wt, res, status, s);
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

int launch_admom_list(const ngmix_admom_conf *conf, const ngmix_pixel *pixels,
                      int64_t n, ngmix_gauss2d *wt, ngmix_admom_result *res,
                      int32_t *status, hipStream_t s)
{
    hipLaunchKernelGGL(admom_list_kernel, dim3(1), dim3(BLOCK), 0, s, *conf, pixels,
                       (int)n, wt, res, status);
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

}  // namespace ngmix
