// lmfit.hip -- batched Levenberg-Marquardt for the simple models
// (gauss / exp / dev, optionally psf-convolved): SURVEY.md 8(f)-1.
//
// Reference path being replaced, per object and per function evaluation:
// Fitter.go -> scipy leastsq (MINPACK lmder) -> FitModel.calc_fdiff /
// calc_jacobian (ngmix/fitting/results.py:439-570) -> fill_fdiff
// (gmix_nb.py:877-900) and deriv_images (derivs_nb.py:40-127) per observation.
//
// Here N fits advance in lock step, two launches per LM step for all of them:
//
//   lm_eval_kernel   one WAVE per stamp: fills the model mixture from the
//                    object's trial parameters, convolves it with the stamp's
//                    psf, evaluates value + 5 derivative images per pixel in
//                    registers (never written to HBM) and accumulates the
//                    stamp's normal equations  A = J^T J (21), g = J^T f (6),
//                    ff = |f|^2  -- 28 doubles out per stamp, 16 B/pixel in.
//   lm_advance_kernel  one THREAD per object: folds the object's stamps
//                    (epochs / bands) into its (5+nband)-parameter system and
//                    runs one step of the lmder logic (lm_core.hpp).
#include <type_traits>

#include <stdio.h>

#include "device_utils.hpp"
#include "launch.hpp"
#include "lm_core.hpp"
#include "lm_core_reg.hpp"


namespace ngmix {

__constant__ ModelTables c_tables_lm = NGMIX_MODEL_TABLES;
__constant__ double c_exp_table_lm[16] = NGMIX_EXP_TABLE;
__constant__ double c_fexp_coef_lm[12] = NGMIX_FEXP_COEF;

constexpr int LM_NSUM = NGMIX_LM_NSUM;  // 21 + 6 + 1

// One composed gaussian, staged in LDS.  For gauss / exp / dev every object
// gaussian is the same ellipse scaled by its own T_k (gmix_fill_simple,
// gmix_nb.py:307-351), so d(irr, irc, icc)_k / d(g1, g2, T) = T_k * u_a with
// three stamp-wide vectors u_a (results.py:955-1010): the record carries T_k
// and the pixel pass accumulates, per pixel,
//     N11 = sum_k T_k (ec_k qv_k^2    - e_k w11_k)
//     N12 = sum_k T_k (ec_k qv_k qu_k - e_k w12_k)
//     N22 = sum_k T_k (ec_k qu_k^2    - e_k w22_k)
// from which the three shape derivatives (derivs_nb.py:113-125) are the
// stamp-wide combinations h_a . (N11, N12, N22) -- ten instructions per pair
// instead of eighteen, seven doubles per gaussian instead of twenty.
struct DerivGauss {
    double row, col;
    double w11, w12, w22;  // Q = Sigma^-1
    double pa;             // norm * area
    double tk;             // irr + icc of the object's gaussian before the psf
    double pad_;
    TileBox box;           // device_utils.hpp: the chi2 < 25 box as the tile test reads it
};
static_assert(sizeof(DerivGauss) == 80, "DerivGauss");

// one record per 8x8 tile (as the fused pixel-pass kernels, pixpass.hip)
struct LmTile {
    double bv, bu;  // (v, u) of the tile's first pixel
    int off;        // byte offset of the tile's first pixel inside the stamp
    int r0, c0;     // its row / column
    int pad_;
};
static_assert(sizeof(LmTile) == 32, "LmTile");

constexpr int LM_TILE_CAP = 1024;  // tile records kept in LDS (<= 256 x 256 px)

// (dpp_row_shr_zero / row16_sum / swap_add16 / swap_add32: device_utils.hpp)

struct LmEvalShared {
    double tabr[16];
    double raw[LM_NSUM];   // the reduced sums in the raw basis (below)
    double hc[6][3];       // the rows of the raw -> parameter map, 3 terms each
};

// fexp(-chi2/2) for 0 <= chi2 < 25 (fastexp_nb.py:223-262) straight from chi2:
// 0.5 * chi2 is exact, so n and f are those of fexp_neg_fused(chi2 / 2)
__device__ __forceinline__ double fexp_neg_half(double chi2, const double *tabr,
                                                const FexpCoef &k)
{
    constexpr double MAGIC = 6755399441055744.0;  // 1.5 * 2^52
    const double t = fma(chi2, 0.5, MAGIC);
    const int n = __double2loint(t);
    const double nd = t - MAGIC;
    const double f = fma(chi2, -0.5, nd);
    const double tv = tabr[n];
    double p = fma(f, __hiloint2double(k.c5hi, k.c5lo), k.c4);
    p = fma(f, p, k.c3);
    p = fma(f, p, k.c2);
    p = fma(f, p, k.c1);
    p = fma(f, p, k.c0);
    return tv * p;
}

// LDS_TILES: the tile records of the stamp are staged in LDS (every batch whose
// largest stamp has at most LM_TILE_CAP tiles); otherwise made on the fly
// RAW: the normal equations are accumulated in the raw basis and mapped after
// the reduction (the default); false keeps the per-pixel map (NGMIX_LM_JBASIS,
// the A/B knob: no statistics then)
template <bool LDS_TILES, bool RAW = true>
__global__ __launch_bounds__(WAVE) void lm_eval_kernel(
    const ngmix_stamp *__restrict__ stamps, const double *__restrict__ val,
    const double *__restrict__ ierr, const ngmix_jacobian *__restrict__ jacs,
    int model, int ng0, const lm_state *__restrict__ states,
    const int32_t *__restrict__ stamp_obj, const int32_t *__restrict__ stamp_band,
    const ngmix_gauss2d *__restrict__ psf, int npsf, double *__restrict__ sums,
    int32_t *__restrict__ status, int no_skip, int tile_cap,
    double *__restrict__ stamp_stats)
{
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    __shared__ LmEvalShared sh;

    const int s = blockIdx.x;
    const int lane = threadIdx.x;
    const int obj = stamp_obj ? stamp_obj[s] : s;
    const lm_state &state = states[obj];
    if (state.phase == LM_PHASE_DONE) return;  // this object has finished
    const int band = stamp_band ? stamp_band[s] : 0;
    const ngmix_stamp st = stamps[s];
    const ngmix_jacobian jac = jacs[s];
    const int nrow = st.nrow, ncol = st.ncol;
    const double area = jac.scale * jac.scale;
    const bool izw = (st.flags & NGMIX_STAMP_IGNORE_ZERO_WEIGHT) != 0;
    const bool masked = izw && st.npix_kept != nrow * ncol;
    double *out = sums + (size_t)s * LM_NSUM;

    const int npsf1 = npsf > 0 ? npsf : 1;
    const int G = ng0 * npsf1;
    DerivGauss *dg = (DerivGauss *)dyn;
    LmTile *te = (LmTile *)(dyn + (size_t)G * sizeof(DerivGauss));

    // ---- tile records (need only the stamp's shape and jacobian)
    const int ntx = (ncol + TILE_W - 1) / TILE_W;
    const int nty = (nrow + TILE_H - 1) / TILE_H;
    const int ntiles = ntx * nty;
    // T / ntx by a multiply: exact for T * ntx < 2^32
    const unsigned inv_ntx = ntx > 1 ? 0xFFFFFFFFu / (unsigned)ntx + 1u : 0u;
    auto make_tile = [&](int T) {
        LmTile e;
        if (T < ntiles) {
            const int ty = ntx > 1 ? (int)__umulhi((unsigned)T, inv_ntx) : T;
            const int tx = T - ty * ntx;
            e.r0 = ty * TILE_H;
            e.c0 = tx * TILE_W;
            const double rd = (double)e.r0 - jac.row0, cd = (double)e.c0 - jac.col0;
            e.bv = fma(jac.dvdrow, rd, jac.dvdcol * cd);
            e.bu = fma(jac.dudrow, rd, jac.dudcol * cd);
            e.off = (e.r0 * ncol + e.c0) * 8;
        } else {
            // the sentinel past the last tile: no gaussian's box reaches it
            e.r0 = 1 << 30;
            e.c0 = 1 << 30;
            e.bv = 0.0;
            e.bu = 0.0;
            e.off = 0;
        }
        e.pad_ = 0;
        return e;
    };
    if (LDS_TILES)
        for (int T = lane; T <= ntiles; T += WAVE) te[T] = make_tile(T);

    // ---- the composed mixture and its derivative data at the trial point.
    // The parameter prep of the reference (g1g2_to_e1e2, gmix_fill_simple,
    // gmix_convolve_fill, gmix_set_norms: gmix_nb.py:176-218,307-351,609-678) is
    // O(ngauss) per stamp but paid by the whole wave, so it is written for few
    // instructions, equal to the reference's values to rounding:
    //   e = tanh(2 atanh g) = 2 g / (1 + g^2) without the hyperbolic functions,
    //   reciprocals / inverse square roots by v_rcp / v_rsq + Newton (~1 ulp).
    double p[6];
    for (int k = 0; k < 5; k++) p[k] = state.xt[k];
    p[5] = state.xt[5 + band];
    const double g1 = p[2], g2 = p[3], T = p[4], flux = p[5];
    int bad = 0;
    if (T == 0.0 || flux == 0.0) bad = 1;  // results.py:527-531
    const double gsq = g1 * g1 + g2 * g2;
    if (gsq >= 1.0) bad = 1;               // g >= 1: GMixRangeError
    const double opg = 1.0 + gsq;
    const double fgg = 2.0 * rcp_newton(opg);   // e / g
    double efac = fgg;
    if (4.0 * gsq >= opg * opg) efac = 0.99999999 / sqrt(gsq);  // e >= 1 is clamped
    FillCtx c;
    c.model = model;
    c.ngauss = ng0;
    c.row = p[0];
    c.col = p[1];
    c.e1 = efac * g1;
    c.e2 = efac * g2;
    c.T = T;
    c.flux = flux;
    c.fracdev = 0.0;
    c.ifracdev = 1.0;
    c.TdByTe = 1.0;
    double rowcen = 0.0, colcen = 0.0, ipsum = 1.0;
    const ngmix_gauss2d *q = psf ? psf + (size_t)s * npsf : nullptr;
    if (npsf > 0) {
        // gmix_get_cen (gmix_nb.py:108-130)
        double psum = 0.0;
        for (int i = 0; i < npsf; i++) {
            const double pp = q[i].p;
            rowcen = fma(pp, q[i].row, rowcen);
            colcen = fma(pp, q[i].col, colcen);
            psum += pp;
        }
        if (psum == 0.0) bad = 1;
        ipsum = rcp_newton(psum);
        rowcen *= ipsum;
        colcen *= ipsum;
    }
    if (lane < 16) sh.tabr[lane] = c_exp_table_lm[15 - lane];
    if (!bad) {
        for (int i = lane; i < G; i += WAVE) {
            const int io = i / npsf1, ip = i - io * npsf1;
            ngmix_gauss2d g0, gc;
            fill_component(c_tables_lm, c, p, io, g0);
            if (npsf > 0) convolve_component(g0, q[ip], rowcen, colcen, ipsum, gc);
            else gc = g0;
            // gauss2d_set_norm (gmix_nb.py:190-218)
            if (gc.det < LOW_DETVAL || gc.irr + gc.icc <= LOW_DETVAL) bad = 1;
            const double idet = rcp_newton(gc.det);
            double rs = __builtin_amdgcn_rsq(gc.det);   // 1 / sqrt(det)
            rs = fma(0.5 * rs, fma(-gc.det * rs, rs, 1.0), rs);
            rs = fma(0.5 * rs, fma(-gc.det * rs, rs, 1.0), rs);
            gc.drr = gc.irr * idet;
            gc.drc = gc.irc * idet;
            gc.dcc = gc.icc * idet;
            DerivGauss r;
            r.row = gc.row;
            r.col = gc.col;
            r.w11 = gc.dcc;   // icc / det
            r.w12 = -gc.drc;  // -irc / det
            r.w22 = gc.drr;   // irr / det
            r.pa = gc.p * (rs * 0.15915494309189535) * area;   // p / (2 pi sqrt(det))
            r.tk = g0.irr + g0.icc;
            r.pad_ = 0.0;
            r.box = tile_box(no_skip ? full_box() : gauss_pixel_box(gc, jac), TILE_H, TILE_W);
            dg[i] = r;
        }
    }
    if (__ballot(bad != 0) != 0ull) {
        // out of range at the trial point: calc_fdiff's LOWVAL vector
        if (lane == 0) {
            for (int k = 0; k < LM_NSUM - 1; k++) out[k] = 0.0;
            out[LM_NSUM - 1] = INFINITY;
            if (status) status[s] = NGMIX_ERR_G_RANGE;
            if (stamp_stats) {
                stamp_stats[2 * (size_t)s] = 0.0;
                stamp_stats[2 * (size_t)s + 1] = 0.0;
            }
        }
        return;
    }
    __syncthreads();

    // d(e1, e2)/d(g1, g2) for e = 2 g / (1 + g^2)  (results.py:985-992) and the
    // stamp-wide vectors h_a = {u_a0 / 2, u_a1, u_a2 / 2}, u_a = d(irr, irc,
    // icc)_k / d(g1 | g2 | T) / T_k
    const double dfac = -0.5 * fgg * fgg;       // -f / (1 + g^2)
    const double de1dg1 = fgg + 2.0 * g1 * g1 * dfac;
    const double de1dg2 = 2.0 * g1 * g2 * dfac;
    const double de2dg1 = de1dg2;
    const double de2dg2 = fgg + 2.0 * g2 * g2 * dfac;
    const double i2T = 0.5 * rcp_newton(T);
    const double h00 = uniform_f64(-0.25 * de1dg1), h01 = uniform_f64(0.5 * de2dg1);
    const double h10 = uniform_f64(-0.25 * de1dg2), h11 = uniform_f64(0.5 * de2dg2);
    const double h20 = uniform_f64(0.5 * ((1.0 - c.e1) * i2T)),
                 h21 = uniform_f64(c.e2 * i2T),
                 h22 = uniform_f64(0.5 * ((1.0 + c.e1) * i2T));
    const double iflux = uniform_f64(rcp_newton(flux));

    // ---- pixel pass: 8x8 tiles, value + 5 derivatives per pixel in registers
    const FexpCoef K = load_fexp_coef(c_fexp_coef_lm);
    const int lrow = lane / TILE_W, lcol = lane % TILE_W;
    const double olv = fma(jac.dvdrow, (double)lrow, jac.dvdcol * (double)lcol);
    const double olu = fma(jac.dudrow, (double)lrow, jac.dudcol * (double)lcol);
    const unsigned lane_off = (unsigned)(lrow * ncol + lcol) * 8u;
    const int rlim = nrow - lrow, clim = ncol - lcol;  // in bounds: r0 < rlim, c0 < clim
    const char *bval = (const char *)(val + st.pix_off);
    const char *bierr = (const char *)(ierr + st.pix_off);

    double acc[LM_NSUM];
#pragma unroll
    for (int k = 0; k < LM_NSUM; k++) acc[k] = 0.0;

    auto tile = [&](int T) { return LDS_TILES ? te[T] : make_tile(T); };
    auto load_tile = [&](int Tn, double &pv, double &pe) {
        pv = 0.0;
        pe = 0.0;
        if (Tn < ntiles) {
            const LmTile e = tile(Tn);
            if ((e.r0 < rlim) & (e.c0 < clim)) {
                const unsigned off = lane_off + (unsigned)e.off;
                pv = *(const double *)(bval + off);
                pe = *(const double *)(bierr + off);
            }
        }
    };

    // (tile, gaussian) box tests, CH tiles per ballot: lane = k * G + g holds
    // gaussian g's box and tests it against tile T + k
    const bool chunked = G <= 32;
    const int CH = chunked ? WAVE / G : 0;
    const unsigned ngmask = chunked ? (unsigned)((1ull << G) - 1ull) : 0u;
    int k_l = 0;
    bool lane_valid = false;
    TileBox mybox = dg[0].box;
    if (chunked) {
        k_l = lane / G;
        lane_valid = k_l < CH;
        mybox = dg[lane - k_l * G].box;
    }
    const unsigned long long valid_mask = __builtin_amdgcn_ballot_w64(lane_valid);
    // mode ANALYTIC_LAZY: a trial whose acceptance is predicted to end the fit
    // asks for |f|^2 alone (ngmix_hip.h): the same per-pixel model value -- the
    // same expressions in the same order, so f is what the full pass computes,
    // to the bit -- without the five derivative images and their 25 sums
    const bool fonly = __builtin_amdgcn_readfirstlane(state.fonly) != 0;

    auto pixel_pass = [&](auto jac_tag) {
    constexpr bool JAC = decltype(jac_tag)::value;
    unsigned long long allmask = 0ull;
    int kc = 0;

    // one tile: every surviving gaussian at its 64 pixels, then the pixels'
    // terms of the sums
    auto tile_body = [&](int Tc, double pval, double pierr) {
        const LmTile tc = tile(Tc);
        const double v = tc.bv + olv, u = tc.bu + olu;
        double o0 = 0.0, o1 = 0.0, o2 = 0.0, n11 = 0.0, n12 = 0.0, n22 = 0.0;

        for (int gb = 0; gb < G; gb += WAVE) {
            unsigned long long gmask;
            if (chunked) {
                if (kc == 0) {
                    int Tk = Tc + k_l;
                    if (Tk > ntiles) Tk = ntiles;  // the sentinel
                    const LmTile tk = tile(Tk);
                    allmask = tile_hits(mybox, tk.r0, tk.c0) & valid_mask;
                }
                gmask = (unsigned)allmask & ngmask;
                allmask >>= G;
                kc = (kc + 1 == CH) ? 0 : kc + 1;
            } else {
                // lane g tests gaussian gb+g's chi2<25 box against this tile
                const int gi = gb + lane < G ? gb + lane : gb;
                const TileBox box = dg[gi].box;
                gmask = tile_hits(box, tc.r0, tc.c0) &
                        __builtin_amdgcn_ballot_w64(gb + lane < G);
            }
            while (gmask) {
                const int g = gb + __builtin_ctzll(gmask);
                gmask &= gmask - 1ull;
                const DerivGauss &D = dg[g];
                const double dv = v - D.row, du = u - D.col;
                const double w11 = D.w11, w12 = D.w12, w22 = D.w22;
                const double qv = fma(w11, dv, w12 * du);
                const double qu = fma(w12, dv, w22 * du);
                const double chi2 = fma(dv, qv, du * qu);
                // derivs_nb.py:104-105: chi2 >= 25 or chi2 < 0 -> skip (one
                // unsigned compare on the high word: negative, nan, inf fail)
                if ((unsigned)__double2hiint(chi2) < 0x40390000u) {
                    double e = D.pa * fexp_neg_half(chi2, sh.tabr, K);
                    double ec = e;
                    if ((unsigned)__double2hiint(chi2) >= 0x40340000u) {
                        // W and W - 2 W' of the apodisation (fastexp_nb.py:97-135);
                        // both are exactly 1 at chi2 == 20
                        const double au = (25.0 - chi2) * 0.2;
                        const double aq = fma(au, fma(au, K.w6, K.wm15), K.w10);
                        const double w = (au * au) * (au * aq);
                        if (JAC) {
                            const double umu = au * (1.0 - au);
                            // -2 W' = 60 * 0.2 * umu^2
                            ec = e * fma(12.0 * umu, umu, w);
                        }
                        e *= w;
                    }
                    o0 += e;
                    if (JAC) {
                        const double tk = D.tk;
                        o1 = fma(ec, qv, o1);
                        o2 = fma(ec, qu, o2);
                        const double t = tk * ec, mte = -(tk * e);
                        const double tqv = t * qv, tqu = t * qu;
                        n11 = fma(tqv, qv, fma(mte, w11, n11));
                        n12 = fma(tqv, qu, fma(mte, w12, n12));
                        n22 = fma(tqu, qu, fma(mte, w22, n22));
                    }
                }
            }
        }

        // residual and jacobian row of this pixel (results.py:556-563); lanes
        // outside the stamp and zero-weight pixels have ierr == 0.  The row is
        // LINEAR in B = ierr * (o1, o2, N11, N12, N22, o0) with stamp-wide
        // coefficients (the h_a above, 1 / flux): J = H B.  So the normal
        // equations are accumulated in the raw basis -- B B^T, B f, f^2: the
        // same 28 fused accumulates, without the nine instructions per pixel
        // of the map -- and mapped once per stamp after the reduction:
        // J^T J = H (sum B B^T) H^T, J^T f = H (sum B f).
        if (!JAC) {
            // B5 = model * ierr and f: the three sums the step and the
            // statistics read, accumulated as the full pass accumulates them
            if (!masked || pierr > 0.0) {
                const double f = (o0 - pval) * pierr;
                const double b5 = o0 * pierr;
                acc[20] = fma(b5, b5, acc[20]);
                acc[26] = fma(b5, f, acc[26]);
                acc[27] = fma(f, f, acc[27]);
            }
        } else if (!masked || pierr > 0.0) {
            const double f = (o0 - pval) * pierr;
            double B[6];
            if (RAW) {
                B[0] = o1 * pierr;
                B[1] = o2 * pierr;
                B[2] = n11 * pierr;
                B[3] = n12 * pierr;
                B[4] = n22 * pierr;
                B[5] = o0 * pierr;
            } else {
                const double m11 = n11 * pierr, m12 = n12 * pierr, m22 = n22 * pierr;
                B[0] = o1 * pierr;
                B[1] = o2 * pierr;
                const double md = m11 - m22;   // u_a0 == -u_a2 for a = g1, g2
                B[2] = fma_sgpr(h00, md, mul_sgpr(h01, m12));
                B[3] = fma_sgpr(h10, md, mul_sgpr(h11, m12));
                B[4] = fma_sgpr(h20, m11, fma_sgpr(h21, m12, mul_sgpr(h22, m22)));
                B[5] = o0 * mul_sgpr(iflux, pierr);
            }
            int k = 0;
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int b = a; b < 6; b++) {
                    acc[k] = fma(B[a], B[b], acc[k]);
                    k++;
                }
#pragma unroll
            for (int a = 0; a < 6; a++) acc[21 + a] = fma(B[a], f, acc[21 + a]);
            acc[27] = fma(f, f, acc[27]);
        }
    };
    if (JAC) {
        // the full pass: a tile is ~165 instructions, four waves share the SIMD
        // -- the next tile's pixels asked for at the top of this one are there
        // when it ends
        double nval, nierr;
        load_tile(0, nval, nierr);
        for (int Tc = 0; Tc < ntiles; Tc++) {
            const double pval = nval, pierr = nierr;
            load_tile(Tc + 1, nval, nierr);
            tile_body(Tc, pval, pierr);
        }
    } else {
        // the |f|^2-only pass: a tile is ~75 instructions, too short to cover a
        // trip to HBM -- the pixels are asked for a GROUP of four tiles ahead,
        // by loads no branch goes around (a branch around a load makes the
        // compiler drain every outstanding load there): a lane outside the
        // stamp, or a tile past the last, reads the stamp's first pixel and
        // ignores it
        constexpr int PF = 4;
        double nv[PF], ne[PF];
        auto load_group = [&](int T0) {
#pragma unroll
            for (int k = 0; k < PF; k++) {
                int Tn = T0 + k;
                if (Tn > ntiles) Tn = ntiles;       // the sentinel: out of bounds
                const LmTile e = tile(Tn);
                const bool inb = (e.r0 < rlim) & (e.c0 < clim);
                const unsigned off = inb ? lane_off + (unsigned)e.off : 0u;
                const double a = *(const double *)(bval + off);
                const double b = *(const double *)(bierr + off);
                nv[k] = inb ? a : 0.0;
                ne[k] = inb ? b : 0.0;
            }
        };
        load_group(0);
        for (int Tg = 0; Tg < ntiles; Tg += PF) {
            double cv[PF], ce[PF];
#pragma unroll
            for (int k = 0; k < PF; k++) {
                cv[k] = nv[k];
                ce[k] = ne[k];
            }
            load_group(Tg + PF);
#pragma unroll
            for (int k = 0; k < PF; k++)
                if (Tg + k < ntiles) tile_body(Tg + k, cv[k], ce[k]);
        }
    }
    };
    if (RAW && fonly) pixel_pass(std::false_type{});
    else pixel_pass(std::true_type{});

    // ---- 28 sums over the wave.  v_permlane16_swap / v_permlane32_swap trade
    // rows of 16 / 32 lanes between two registers, so one add folds the rows of
    // TWO sums at once: four sums collapse into one register whose row w holds
    // the 16 lane-partials of sum 4 r + w (63 instructions), DPP finishes the
    // rows (84) -- 147 instead of 28 x 12 = 336, no LDS, a fixed order.
    static_assert(LM_NSUM % 4 == 0, "four sums per register");
#pragma unroll
    for (int r = 0; r < LM_NSUM / 4; r++) {
        // (an |f|^2-only pass has sums 20, 26 and 27: the last two registers,
        // folded exactly as the full pass folds them)
        if (RAW && fonly && r < 5) continue;
        const double ab = swap_add16(acc[4 * r + 0], acc[4 * r + 1]);
        const double cd = swap_add16(acc[4 * r + 2], acc[4 * r + 3]);
        const double t = row16_sum(swap_add32(ab, cd));
        if ((lane & 15) == 15) {
            if (RAW) sh.raw[4 * r + (lane >> 4)] = t;
            else out[4 * r + (lane >> 4)] = t;
        }
    }
    if (!RAW) {
        if (lane == 0 && status) status[s] = NGMIX_OK;
        return;
    }
    if (fonly) {
        // |f|^2 and the statistics; the other sums of this stamp are not read
        // (lm_advance: a trial that carries fonly uses ff alone)
        __syncthreads();
        if (lane == 0) {
            out[LM_NSUM - 1] = sh.raw[27];
            if (stamp_stats) {
                stamp_stats[2 * (size_t)s] = sh.raw[20] - sh.raw[26];
                stamp_stats[2 * (size_t)s + 1] = sh.raw[20];
            }
            if (status) status[s] = NGMIX_OK;
        }
        return;
    }
    // ---- raw basis -> parameters.  Row a of H has at most three terms,
    // (column, coefficient): cen1 = B0, cen2 = B1, g1 / g2 = h_a0 (B2 - B4) +
    // h_a1 B3 (u_a0 == -u_a2 for the shears), T = h20 B2 + h21 B3 + h22 B4,
    // flux = B5 / flux.  Lane k < 21 owns (J^T J)[a][b], lanes 21..26 (J^T f)[a].
    if (lane == 0) {
        const double z = 0.0;
        const double rows[6][3] = {{1.0, z, z},    {1.0, z, z},    {h00, h01, -h00},
                                   {h10, h11, -h10}, {h20, h21, h22}, {iflux, z, z}};
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int t = 0; t < 3; t++) sh.hc[a][t] = rows[a][t];
    }
    __syncthreads();
    if (lane < LM_NSUM) {
        // the columns of row a's terms: {0}, {1}, {2, 3, 4} x 3, {5}
        auto col0 = [](int a) { return a < 2 ? a : (a < 5 ? 2 : 5); };
        // index of raw (i, j), i <= j, in the packed upper triangle
        auto tri = [](int i, int j) { return i * 6 - i * (i - 1) / 2 + (j - i); };
        double r;
        if (lane < 21) {
            int a = 0, rem = lane;
            while (rem >= 6 - a) {
                rem -= 6 - a;
                a++;
            }
            const int b = a + rem;
            const int ca = col0(a), cb = col0(b);
            r = 0.0;
#pragma unroll
            for (int ta = 0; ta < 3; ta++) {
                double inner = 0.0;
#pragma unroll
                for (int tb = 0; tb < 3; tb++) {
                    // (padding terms carry a zero coefficient and re-read a
                    // valid column)
                    const int i = (a >= 2 && a < 5) ? ca + ta : ca;
                    const int j = (b >= 2 && b < 5) ? cb + tb : cb;
                    const double sij = sh.raw[i <= j ? tri(i, j) : tri(j, i)];
                    inner = fma(sh.hc[b][tb], sij, inner);
                }
                r = fma(sh.hc[a][ta], inner, r);
            }
        } else if (lane < 27) {
            const int a = lane - 21, ca = col0(a);
            r = 0.0;
#pragma unroll
            for (int ta = 0; ta < 3; ta++) {
                const int i = (a >= 2 && a < 5) ? ca + ta : ca;
                r = fma(sh.hc[a][ta], sh.raw[21 + i], r);
            }
        } else {
            r = sh.raw[27];
        }
        out[lane] = r;
    }
    // the sums of get_loglike's statistics at this trial point come with the
    // raw basis for free (gmix_nb.py:862-864): s2n_denom = sum (model ierr)^2
    // = (B B^T)[5][5], s2n_numer = sum val model ivar = that - sum (model ierr) f
    if (lane == 0) {
        if (stamp_stats) {
            stamp_stats[2 * (size_t)s] = sh.raw[20] - sh.raw[26];
            stamp_stats[2 * (size_t)s + 1] = sh.raw[20];
        }
        if (status) status[s] = NGMIX_OK;
    }
}

// One thread per object: gather its stamps' sums into the (nloc-1+nband)-
// parameter normal equations and advance the LM state.  A stamp's sums are
// [J^T J upper triangle | J^T f | f.f] over its nloc LOCAL parameters: the
// shared shape parameters followed by the flux of the stamp's band.
// Copy the live part of a state record: the scalars, the first n entries of
// each per-parameter array and the leading n x n block of R.  The record is
// sized for NGMIX_LM_NPMAX = 14 parameters (2.9 kB); a six-parameter fit uses a
// fifth of it, and lm_advance_kernel moves every record in and out once per
// round.  (Entries beyond n are never read: lm_init_kernel leaves them as allocated.)
template <int ND, int NS, class D, class S>
__device__ __forceinline__ void lm_state_copy_live(D &d, const S &g)
{
    // ND / NS: the strides of R in the destination / source
    const int n = g.n;
    d.n = n;
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
    for (int j = 0; j < n; j++) {
        d.x[j] = g.x[j];
        d.xt[j] = g.xt[j];
        d.diag[j] = g.diag[j];
        d.qtf[j] = g.qtf[j];
        d.step[j] = g.step[j];
        d.xi[j] = g.xi[j];
        d.xti[j] = g.xti[j];
        d.lo[j] = g.lo[j];
        d.hi[j] = g.hi[j];
        d.xstep[j] = g.xstep[j];
        d.hstep[j] = g.hstep[j];
        d.ipvt[j] = g.ipvt[j];
        for (int k = 0; k < n; k++) d.R[j * ND + k] = g.R[j * NS + k];
    }
}

// NP >= the fit's parameter count: the stride of the thread's private copy
template <int NP>
__device__ __forceinline__ void lm_advance_one(
    lm_state *states, int64_t o, const int64_t *__restrict__ obj_start,
    const int32_t *__restrict__ stamp_band, const double *__restrict__ sums, int nloc,
    const double *__restrict__ obj_sums, int32_t *nactive)
{
    lmcore::lm_state_n<NP> s;
    lm_state_copy_live<NP, LM_NPMAX>(s, states[o]);
    const int ntri = nloc * (nloc + 1) / 2, nsum = ntri + nloc + 1;
    double A[NP * NP], g[NP];
    for (int i = 0; i < s.n; i++) {
        for (int j = 0; j < s.n; j++) A[i * NP + j] = 0.0;
        g[i] = 0.0;
    }
    double ff = 0.0;
    const int64_t s0 = obj_start ? obj_start[o] : o;
    const int64_t s1 = obj_start ? obj_start[o + 1] : o + 1;
    for (int64_t st = s0; st < s1; st++) {
        const int band = stamp_band ? stamp_band[st] : 0;
        if (band < 0 || nloc - 1 + band >= s.n || nloc > s.n) {
            // a band the fit has no flux for (or nloc inconsistent with the
            // state): its sums would land outside A.  End the fit as MINPACK ends
            // a call with improper input (see lm_advance_dispatch)
            states[o].info = 0;
            states[o].phase = LM_PHASE_DONE;
            return;
        }
    }
    for (int64_t st = s0; st < s1; st++) {
        const double *v = sums + st * nsum;
        const int band = stamp_band ? stamp_band[st] : 0;
        int k = 0;
        for (int a = 0; a < nloc; a++) {
            const int ga = a < nloc - 1 ? a : nloc - 1 + band;
            for (int b = a; b < nloc; b++) {
                const int gb = b < nloc - 1 ? b : nloc - 1 + band;
                A[ga * NP + gb] += v[k];
                if (ga != gb) A[gb * NP + ga] += v[k];
                k++;
            }
            g[ga] += v[ntri + a];
        }
        ff += v[ntri + nloc];
    }
    if (obj_sums) {
        // rows over the object's own n parameters (the prior rows)
        const int n = s.n, nt = n * (n + 1) / 2;
        const double *v = obj_sums + o * (int64_t)(nt + n + 1);
        int k = 0;
        for (int a = 0; a < n; a++) {
            for (int b = a; b < n; b++) {
                A[a * NP + b] += v[k];
                if (a != b) A[b * NP + a] += v[k];
                k++;
            }
            g[a] += v[nt + a];
        }
        ff += v[nt + n];
    }
    lmcore::lm_advance<NP>(s, ff, g, A);
    lm_state_copy_live<LM_NPMAX, NP>(states[o], s);
    if (s.phase != LM_PHASE_DONE && nactive) atomicAdd(nactive, 1);
}

// The same step for fits of exactly N parameters with every array in registers
// (lm_core_reg.hpp).  A stamp's sums are over its local parameters; global
// parameter G is local G for the shared shape parameters and the stamp's band
// puts its flux at G = nloc - 1 + band.
template <int N>
__device__ __forceinline__ void lm_advance_one_reg(
    lm_state *states, int64_t o, const int64_t *__restrict__ obj_start,
    const int32_t *__restrict__ stamp_band, const double *__restrict__ sums, int nloc,
    const double *__restrict__ obj_sums, int32_t *nactive)
{
    lmreg::lm_state_n<N> s;
    lmreg::load_state<N>(s, states[o]);
    const int ntri = nloc * (nloc + 1) / 2, nsum = ntri + nloc + 1;
    double A[N * N], g[N];
#pragma unroll
    for (int i = 0; i < N * N; i++) A[i] = 0.0;
#pragma unroll
    for (int i = 0; i < N; i++) g[i] = 0.0;
    double ff = 0.0;
    const int64_t s0 = obj_start ? obj_start[o] : o;
    const int64_t s1 = obj_start ? obj_start[o + 1] : o + 1;
    for (int64_t st = s0; st < s1; st++) {
        const double *v = sums + st * nsum;
        const int band = stamp_band ? stamp_band[st] : 0;
        int la[N];
#pragma unroll
        for (int G = 0; G < N; G++)
            la[G] = G < nloc - 1 ? G : (G == nloc - 1 + band ? nloc - 1 : -1);
#pragma unroll
        for (int ga = 0; ga < N; ga++) {
            if (la[ga] < 0) continue;
            const int a = la[ga];
            const int row = a * nloc - a * (a - 1) / 2;   // index of (a, a)
#pragma unroll
            for (int gb = ga; gb < N; gb++) {
                if (la[gb] < 0) continue;
                const double t = v[row + (la[gb] - a)];
                A[ga * N + gb] += t;
                if (ga != gb) A[gb * N + ga] += t;
            }
            g[ga] += v[ntri + a];
        }
        ff += v[ntri + nloc];
    }
    if (obj_sums) {
        // rows over the object's own N parameters (the prior rows)
        constexpr int nt = N * (N + 1) / 2;
        const double *v = obj_sums + o * (int64_t)(nt + N + 1);
        int k = 0;
#pragma unroll
        for (int a = 0; a < N; a++) {
#pragma unroll
            for (int b = a; b < N; b++) {
                A[a * N + b] += v[k];
                if (a != b) A[b * N + a] += v[k];
                k++;
            }
            g[a] += v[nt + a];
        }
        ff += v[nt + N];
    }
    lmreg::lm_advance<N>(s, ff, g, A);
    lmreg::store_state<N>(states[o], s);
    if (s.phase != LM_PHASE_DONE && nactive) atomicAdd(nactive, 1);
}

template <int NP, bool REG>
__device__ __forceinline__ void lm_advance_dispatch(
    lm_state *states, int64_t o, const int64_t *__restrict__ obj_start,
    const int32_t *__restrict__ stamp_band, const double *__restrict__ sums, int nloc,
    const double *__restrict__ obj_sums, int32_t *nactive);

// NP = LM_NPMAX serves any fit (the generic code, private memory); when the
// launcher is told that every fit has 6 .. 8 parameters it runs the register
// form.  (9 and 10 parameters go to the team form of lm_team.hip by default;
// their register instantiations -- which spill part of their arrays to fixed
// private slots -- are reachable through NGMIX_LM_TEAM_MIN only: A/B.)
template <int NP, bool REG>
__global__ __launch_bounds__(WAVE) void lm_advance_kernel(
    lm_state *states, int64_t nobj, const int64_t *__restrict__ obj_start,
    const int32_t *__restrict__ stamp_band, const double *__restrict__ sums, int nloc,
    const double *__restrict__ obj_sums, int32_t *nactive,
    const double *__restrict__ stamp_stats, double *__restrict__ obj_stats)
{
    const int64_t o = blockIdx.x * (int64_t)WAVE + threadIdx.x;
    if (o >= nobj) return;
    if (states[o].phase == LM_PHASE_DONE) return;
    // lmder moves to the trial point exactly when it counts an iteration
    // (and the starting point is where it stands after the first call)
    const int iter0 = states[o].iter, phase0 = states[o].phase;
    lm_advance_dispatch<NP, REG>(states, o, obj_start, stamp_band, sums, nloc, obj_sums,
                                 nactive);
    if (stamp_stats && obj_stats &&
        (phase0 == LM_PHASE_INIT || states[o].iter != iter0)) {
        // the loglike statistics of the point the fit now stands at: the sums
        // lm_eval made at this trial point, over the object's stamps
        const int64_t s0 = obj_start ? obj_start[o] : o;
        const int64_t s1 = obj_start ? obj_start[o + 1] : o + 1;
        double a = 0.0, b = 0.0;
        for (int64_t st = s0; st < s1; st++) {
            a += stamp_stats[2 * st];
            b += stamp_stats[2 * st + 1];
        }
        obj_stats[2 * o] = a;
        obj_stats[2 * o + 1] = b;
    }
}

template <int NP, bool REG>
__device__ __forceinline__ void lm_advance_dispatch(
    lm_state *states, int64_t o, const int64_t *__restrict__ obj_start,
    const int32_t *__restrict__ stamp_band, const double *__restrict__ sums, int nloc,
    const double *__restrict__ obj_sums, int32_t *nactive)
{
    if (REG) {
        if (states[o].n != NP) {
            // the caller's parameter-count hint (nloc + 256 npars) was wrong for
            // this fit: end it as MINPACK ends a call with improper input
            // (info = 0 -> LM_FUNC_NOTFINITE and default pars in the record)
            // instead of leaving it un-advanced for finalize to package
            states[o].info = 0;
            states[o].phase = LM_PHASE_DONE;
            return;
        }
        lm_advance_one_reg<NP>(states, o, obj_start, stamp_band, sums, nloc, obj_sums,
                               nactive);
    } else {
        lm_advance_one<NP>(states, o, obj_start, stamp_band, sums, nloc, obj_sums, nactive);
    }
}

// ===========================================================================
// Forward-difference evaluation (the models without analytic derivatives:
// turb, bdf, bd -- and any model on request): MINPACK's fdjac2 inside the
// pixel pass.  A stamp evaluates, per pixel and in registers, the model at the
// trial point and (when its object asks for a jacobian: phases INIT / JAC) at
// the NLOC points x + h_j e_j, h_j = sqrt(eps) |x_j| (sqrt(eps) when x_j == 0),
// and accumulates J^T J, J^T f, |f|^2 with J_j = (f(x + h_j e_j) - f(x)) / h_j.
// One pass over the pixels does the work of fdjac2's NLOC + 1 residual vectors.
// ===========================================================================

struct FdGauss {
    double row, col, a, b, c, pa;  // y = chi2/2 = a dv^2 + b du^2 + c dv du
};
static_assert(sizeof(FdGauss) == 48, "FdGauss");

// LINEAR: the pixels are taken 64 at a time in ROW-MAJOR order instead of as
// 8 x 8 tiles -- for batches whose stamps fill 8 x 8 tiles badly (a 25 x 25 psf
// stamp is 16 such tiles for 625 pixels, and 10 row-major ones); a tile is then
// the rows it touches, whole, for the box test.  Chosen per launch from the
// batch's largest shape; correct for any shape.  The two forms add a stamp's
// pixels in different orders (sums equal to ~1e-11 relative): the low bits of a
// forward-difference fit's sums -- and, rarely, the nfev of an lmdif fit that
// sits on a decision threshold -- therefore depend on which form the BATCH
// chose, i.e. on the largest stamp it holds (NGMIX_LM_FD_TILES pins the form).
//
// PRECISE: the pass behind the covariance of the ill-conditioned fits
// (ngmix_lm_precise_cov_batch, lm_precise.hip).  It runs once, after the fits
// have ended, at the point of every fit's LAST jacobian -- jac_point, the
// record the ordinary passes leave there: (xt | xstep | hstep) of the state at
// the time -- and accumulates X^T X, X = [J | f], in DOUBLE-DOUBLE: every lane
// owns two entries of the triangle and adds the 64 pixels of a tile in pixel
// order, product and sum by error-free transformations (TwoProd by fma,
// TwoSum), so the sums are exact to ~1e-32 of their magnitude whatever the
// cancellation.  J^T J of a jacobian with cond(J) ~ 1e8 (co-elliptical psf fits
// with 3+ gaussians) has cond ~ 1e16: in plain doubles it is not numerically
// positive definite, and the Cholesky factor MINPACK's QR of J stands for does
// not exist.  sums: (nstamps, 2, NSUM) = high parts | low parts.
template <int NLOC, bool LINEAR, bool PRECISE = false>
// three waves per SIMD up to ten parameters (168 VGPRs, no spills; measured
// 9.0 -> 8.1 ms per 20k 'bdf' fits against two), two beyond -- and for ten
// parameters on row-major tiles, which needs two registers more than 168
__global__ __launch_bounds__(WAVE)
__attribute__((amdgpu_waves_per_eu(!PRECISE && NLOC <= 10 && !(LINEAR && NLOC == 10) ? 3 : 2,
                                   !PRECISE && NLOC <= 10 && !(LINEAR && NLOC == 10) ? 3 : 2)))
void lm_eval_fd_kernel(
    const ngmix_stamp *__restrict__ stamps, const double *__restrict__ val,
    const double *__restrict__ ierr, const ngmix_jacobian *__restrict__ jacs,
    int model, int ng0, const lm_state *__restrict__ states,
    const int32_t *__restrict__ stamp_obj, const int32_t *__restrict__ stamp_band,
    const ngmix_gauss2d *__restrict__ psf, int npsf, double *__restrict__ sums,
    int32_t *__restrict__ status, int no_skip, double *__restrict__ jac_point)
{
    constexpr int NTRI = NLOC * (NLOC + 1) / 2;
    constexpr int NSUM = NTRI + NLOC + 1;
    constexpr int NSETS = NLOC + 1;
    static_assert(NLOC + 1 <= 16, "the normal equations are one 16 x 16 MFMA tile");
    constexpr bool PACKED = NLOC + 1 <= 8;   // two pixel groups per MFMA (below)
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    __shared__ double tabr[16];
    // one tile's rows [J_0 .. J_{NLOC-1}, f, 0 ..] per pixel, 17 doubles apart
    constexpr int JSTRIDE = 17;
    __shared__ double jbuf[WAVE * JSTRIDE];
    // 1 / h_j and the three centres: wave-uniform, read back per tile with
    // broadcast ds_reads (as registers they pushed the kernel past three waves
    // per SIMD)
    __shared__ double unif[NLOC + 6];

    const int s = blockIdx.x;
    const int lane = threadIdx.x;
    const int obj = stamp_obj ? stamp_obj[s] : s;
    const lm_state &state = states[obj];
    if constexpr (PRECISE) {
        // the fits that ended with a covariance to make (leastsqbound.py:76-84)
        if (state.phase != LM_PHASE_DONE || state.info < 1 || state.info > 4) return;
    } else {
        if (state.phase == LM_PHASE_DONE) return;
    }
    const bool want_jac = PRECISE || state.phase != LM_PHASE_TRIAL;
    const int band = stamp_band ? stamp_band[s] : 0;
    const ngmix_stamp st = stamps[s];
    const ngmix_jacobian jac = jacs[s];
    const int nrow = st.nrow, ncol = st.ncol;
    const double area = jac.scale * jac.scale;
    const bool izw = (st.flags & NGMIX_STAMP_IGNORE_ZERO_WEIGHT) != 0;
    const bool masked = izw && st.npix_kept != nrow * ncol;
    double *out = sums + (size_t)s * (PRECISE ? 2 * NSUM : NSUM);
    // the point of this jacobian and its fdjac2 steps: kept for the PRECISE
    // pass (every stamp of an object writes the same record)
    double *jp = jac_point ? jac_point + (size_t)obj * 3 * LM_NPMAX : nullptr;
    if constexpr (!PRECISE) {
        if (jp && want_jac && lane < LM_NPMAX) {
            jp[lane] = state.xt[lane];
            jp[LM_NPMAX + lane] = state.xstep[lane];
            jp[2 * LM_NPMAX + lane] = state.hstep[lane];
        }
    }
    const int npsf1 = npsf > 0 ? npsf : 1;
    const int G = ng0 * npsf1;
    FdGauss *ev = (FdGauss *)dyn;                    // [NSETS][G]
    TileBox *boxes = (TileBox *)(ev + NSETS * G);    // [G], from the base set

    // local parameters and the fdjac2 points the state prepared (xstep /
    // hstep: the step is taken in leastsqbound's internal parameters)
    double p0[NLOC], ps[NLOC];
#pragma unroll
    for (int k = 0; k < NLOC; k++) {
        const int gk = k < NLOC - 1 ? k : NLOC - 1 + band;
        if constexpr (PRECISE) {
            p0[k] = jp[gk];
            ps[k] = jp[LM_NPMAX + gk];
            if (lane == 0) unif[k] = 1.0 / jp[2 * LM_NPMAX + gk];
        } else {
            p0[k] = state.xt[gk];
            ps[k] = state.xstep[gk];
            if (lane == 0) unif[k] = 1.0 / state.hstep[gk];
        }
    }
    // The model is linear in the flux (the last local parameter of every model
    // but coellip, gmix_nb.py:307-558: p_i = flux * pval_i, nothing else depends
    // on it), so the set shifted in flux is the base set times (flux + h) / flux:
    // its column of the jacobian is m_0 (h / flux) / h without a pixel pass.
    const bool lin_flux = model != NGMIX_MODEL_COELLIP && p0[NLOC - 1] != 0.0;
    const double flux_rel = lin_flux ? (ps[NLOC - 1] - p0[NLOC - 1]) / p0[NLOC - 1] : 0.0;
    const int nsets = want_jac ? (lin_flux ? NSETS - 1 : NSETS) : 1;
    double rowcen = 0.0, colcen = 0.0, ipsum = 1.0;
    const ngmix_gauss2d *q = psf ? psf + (size_t)s * npsf : nullptr;
    int bad = 0;
    // every gaussian of a set has one centre when the psf's components do (the
    // object's always do): dv^2, du^2, dv du are then formed once per pixel and
    // centre -- the base one, and the two of the sets shifted in cen1 / cen2
    bool cocen = true;
    if (npsf > 0) {
        double psum;
        if (gmix_cen(q, npsf, rowcen, colcen, psum) != NGMIX_OK) bad = 1;
        else ipsum = 1.0 / psum;
        for (int ip = 1; ip < npsf; ip++)
            if (q[ip].row != q[0].row || q[ip].col != q[0].col) cocen = false;
    }
    if (lane < 16) tabr[lane] = c_exp_table_lm[15 - lane];
    for (int w = lane; w < nsets * G && !bad; w += WAVE) {
        const int k = w / G, i = w - k * G;
        double p[NLOC];
#pragma unroll
        for (int j = 0; j < NLOC; j++) {
            p[j] = (j == k - 1) ? ps[j] : p0[j];
        }
        FillCtx c;
        if (fill_prepare(c_tables_lm, model, ng0, p, nullptr, c) != NGMIX_OK) {
            bad = 1;
            break;
        }
        const int io = i / npsf1, ip = i - io * npsf1;
        ngmix_gauss2d g0, gc;
        fill_component(c_tables_lm, c, p, io, g0);
        if (npsf > 0) convolve_component(g0, q[ip], rowcen, colcen, ipsum, gc);
        else gc = g0;
        if (gauss_set_norm(gc) != NGMIX_OK) {
            bad = 1;
            break;
        }
        FdGauss r;
        r.row = gc.row;
        r.col = gc.col;
        r.a = 0.5 * gc.dcc;
        r.b = 0.5 * gc.drr;
        r.c = -gc.drc;
        r.pa = gc.pnorm * area;
        ev[k * G + i] = r;
        if (k == 0) {
            if constexpr (LINEAR) {
                // rows a row-major tile of 64 pixels can touch; every column:
                // the test is on the rows alone (c0 = 0 passes)
                TileBox tb = tile_box(no_skip ? full_box() : gauss_pixel_box(gc, jac),
                                      (WAVE + ncol - 2) / ncol + 1, TILE_W);
                if (tb.r_lo != (1 << 30)) {
                    tb.c_lo = 0;
                    tb.c_span = 0xffffffffu;
                }
                boxes[i] = tb;
            } else {
                boxes[i] = tile_box(no_skip ? full_box() : gauss_pixel_box(gc, jac),
                                    TILE_H, TILE_W);
            }
        }
    }
    if (__ballot(bad != 0) != 0ull) {
        // out of range at (or one step from) the trial point: LOWVAL residuals
        if (lane == 0) {
            for (int k = 0; k < NSUM - 1; k++) out[k] = 0.0;
            out[NSUM - 1] = INFINITY;
            if constexpr (PRECISE)
                for (int k = 0; k < NSUM; k++) out[NSUM + k] = 0.0;
            if (status) status[s] = NGMIX_ERR_G_RANGE;
        }
        return;
    }
    __syncthreads();

    const FexpCoef K = load_fexp_coef(c_fexp_coef_lm);
    const int lrow = lane / TILE_W, lcol = lane % TILE_W;
    const int ntx = (ncol + TILE_W - 1) / TILE_W;
    const int nty = (nrow + TILE_H - 1) / TILE_H;
    const double *sval = val + st.pix_off;
    const double *sierr = ierr + st.pix_off;
    // (coellip of one gaussian has NLOC == 6 too, and no cen-shifted sets to
    // speak of beyond k = 1, 2: the centres are local parameters 0 and 1 of
    // every model)
    if (lane == 0) {
        unif[NLOC + 0] = ev[0].row;
        unif[NLOC + 1] = ev[0].col;
        unif[NLOC + 2] = want_jac ? ev[G].row : ev[0].row;
        unif[NLOC + 3] = want_jac ? ev[G].col : ev[0].col;
        unif[NLOC + 4] = want_jac ? ev[2 * G].row : ev[0].row;
        unif[NLOC + 5] = want_jac ? ev[2 * G].col : ev[0].col;
    }
    __syncthreads();

    // J^T J, J^T f and f.f are X^T X for X = [J | f] (pixels x 16): a sum of
    // outer products over the pixels -- v_mfma_f64_16x16x4_f64, four pixels
    // per instruction, the whole 16 x 16 result in four accumulator registers
    // per lane (row (lane >> 4) + 4 r, column lane & 15).  With the sums in
    // per-lane VALU accumulators (NSUM of them) the ten-parameter kernel held
    // 236 VGPRs = one wave per SIMD, and twelve parameters did not fit at all.
    typedef double double4_t __attribute__((ext_vector_type(4)));
    double4_t M = {0.0, 0.0, 0.0, 0.0};
    double ff = 0.0;   // trial evaluations want only |f|^2
    // PRECISE: this lane's entries (a <= b) of X^T X, X = [J | f], NX = NLOC + 1
    // columns -- entry e = lane + 64 q of the row-major upper triangle -- as
    // unevaluated sums hi + lo
    constexpr int NX = NLOC + 1, NE = NX * (NX + 1) / 2, NQ = (NE + WAVE - 1) / WAVE;
    int ea[NQ], eb[NQ];
    double dhi[NQ], dlo[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        ea[q] = eb[q] = -1;
        dhi[q] = dlo[q] = 0.0;
        if constexpr (PRECISE) {
            const int e = lane + WAVE * q;
            if (e < NE) {
                int a = 0, row = 0;
                while (row + (NX - a) <= e) {
                    row += NX - a;
                    a++;
                }
                ea[q] = a;
                eb[q] = a + (e - row);
            }
        }
    }
    {
        double *jb = jbuf + lane * JSTRIDE;
#pragma unroll
        for (int j = 0; j < JSTRIDE; j++) jb[j] = 0.0;
    }

    auto load_tile = [&](int ty, int tx, double &pv, double &pe) {
        if constexpr (LINEAR) {
            // (ty counts the row-major tiles, tx stays 0)
            const int p = ty * WAVE + lane;
            pv = 0.0;
            pe = 0.0;
            if (p < nrow * ncol) {
                pv = sval[p];
                pe = sierr[p];
            }
        } else {
            const int row = ty * TILE_H + lrow, col = tx * TILE_W + lcol;
            pv = 0.0;
            pe = 0.0;
            if (ty < nty && row < nrow && col < ncol) {
                pv = sval[row * ncol + col];
                pe = sierr[row * ncol + col];
            }
        }
    };
    // LINEAR: this lane's pixel in a row-major tile, (row, col) of pixel
    // T * 64 + lane, advanced by 64 pixels per tile without a division
    int prow = 0, pcol = 0;
    if constexpr (LINEAR) {
        prow = lane / ncol;
        pcol = lane - prow * ncol;
    }

    // the tile loop; COCEN: one centre per set (see above)
    auto tiles = [&](auto cocen_c) {
        constexpr bool COCEN = decltype(cocen_c)::value;
        int ty = 0, tx = 0;
        double nval, nierr;
        load_tile(ty, tx, nval, nierr);
        while (ty < (LINEAR ? (nrow * ncol + WAVE - 1) / WAVE : nty)) {
            const double pval = nval, pierr = nierr;
            int ty2 = ty, tx2 = tx + 1;
            if (LINEAR || tx2 == ntx) {
                tx2 = 0;
                ty2++;
            }
            load_tile(ty2, tx2, nval, nierr);

            int r0, c0;
            double rowd, cold;
            if constexpr (LINEAR) {
                // the first row the tile touches (lane 0's), column 0 (see the boxes)
                r0 = __builtin_amdgcn_readfirstlane(prow);
                c0 = 0;
                rowd = (double)prow - jac.row0;
                cold = (double)pcol - jac.col0;
                pcol += WAVE % ncol;
                prow += WAVE / ncol;
                if (pcol >= ncol) {
                    pcol -= ncol;
                    prow++;
                }
            } else {
                r0 = ty * TILE_H;
                c0 = tx * TILE_W;
                rowd = (double)(r0 + lrow) - jac.row0;
                cold = (double)(c0 + lcol) - jac.col0;
            }
            const double v = fma(jac.dvdrow, rowd, jac.dvdcol * cold);
            const double u = fma(jac.dudrow, rowd, jac.dudcol * cold);
            // dv^2, du^2, dv du about the three centres
            double q0[3] = {0.0, 0.0, 0.0}, q1[3] = {0.0, 0.0, 0.0}, q2[3] = {0.0, 0.0, 0.0};
            if (COCEN) {
                const double dv0 = v - unif[NLOC + 0], du0 = u - unif[NLOC + 1];
                q0[0] = dv0 * dv0;
                q0[1] = du0 * du0;
                q0[2] = dv0 * du0;
                if (want_jac) {
                    const double dv1 = v - unif[NLOC + 2], du1 = u - unif[NLOC + 3];
                    q1[0] = dv1 * dv1;
                    q1[1] = du1 * du1;
                    q1[2] = dv1 * du1;
                    const double dv2 = v - unif[NLOC + 4], du2 = u - unif[NLOC + 5];
                    q2[0] = dv2 * dv2;
                    q2[1] = du2 * du2;
                    q2[2] = dv2 * du2;
                }
            }
            double m[NSETS];
#pragma unroll
            for (int k = 0; k < NSETS; k++) m[k] = 0.0;

            for (int gb = 0; gb < G; gb += WAVE) {
                const int gi = gb + lane < G ? gb + lane : gb;
                const TileBox box = boxes[gi];
                unsigned long long gmask = tile_hits(box, r0, c0) &
                                           __builtin_amdgcn_ballot_w64(gb + lane < G);
                while (gmask) {
                    const int g = gb + __builtin_ctzll(gmask);
                    gmask &= gmask - 1ull;
#pragma unroll
                    for (int k = 0; k < NSETS; k++) {
                        if (k >= nsets) break;
                        const FdGauss &E = ev[k * G + g];
                        double y;
                        if (COCEN) {
                            const double *qq = k == 1 ? q1 : (k == 2 ? q2 : q0);
                            y = fma(E.a, qq[0], fma(E.b, qq[1], E.c * qq[2]));
                        } else {
                            const double dv = v - E.row, du = u - E.col;
                            y = fma(E.a, dv * dv, fma(E.b, du * du, E.c * (dv * du)));
                        }
                        // 0 <= chi2 < 25 <=> y in [+0, 12.5): one unsigned compare on
                        // the high word (the forms are positive definite after
                        // gauss_set_norm; negative, nan, inf fail as they fail the
                        // reference's test)
                        if ((unsigned)__double2hiint(y) < 0x40290000u) {
                            double e = fexp_neg_fused(y, tabr, K);
                            // the apodised band 20 <= chi2 < 25: under a wave-level
                            // branch (left to itself the compiler predicates the
                            // nine instructions into every evaluation)
                            const bool band = (unsigned)__double2hiint(y) >= 0x40240000u;
                            if (__ballot(band) != 0ull) {
                                if (band) {
                                    // the window in b = 0.8 u (pixpass.hip): no
                                    // step reads two SGPR constants
                                    const double bb = fma(y, K.wb, 4.0);
                                    const double bm = bb - 1.0;
                                    const double bq = fma(bm, bm, K.wq);
                                    e *= (bb * bb) * (bb * bq);
                                    e *= K.wk;
                                }
                            }
                            m[k] = fma(E.pa, e, m[k]);
                        }
                    }
                }
            }

            // a pixel outside the stamp or of zero weight has ierr == 0: its row
            // is zero, except that a masked pixel may hold a non-finite value
            const bool live = !masked || pierr > 0.0;
            const double f = live ? (m[0] - pval) * pierr : 0.0;
            if (!want_jac) {
                ff = fma(f, f, ff);
            } else {
                double *jb = jbuf + lane * JSTRIDE;
#pragma unroll
                for (int j = 0; j < NLOC; j++) {
                    double d = m[j + 1] - m[0];
                    if (j == NLOC - 1 && lin_flux) d = m[0] * flux_rel;
                    jb[j] = live ? d * (pierr * unif[j]) : 0.0;
                }
                jb[NLOC] = f;
                __syncthreads();   // one wave: orders the LDS traffic, no s_barrier
                if constexpr (PRECISE) {
                    // Dot2 (Ogita, Rump, Oishi): the product's rounding error by
                    // fma, the sum's by TwoSum, both gathered in lo
#pragma unroll 4
                    for (int pix = 0; pix < WAVE; pix++) {
                        const double *xr = jbuf + pix * JSTRIDE;
#pragma unroll
                        for (int q = 0; q < NQ; q++) {
                            if (ea[q] < 0) continue;
                            const double xa = xr[ea[q]], xb = xr[eb[q]];
                            const double pr = xa * xb;
                            const double pe = __builtin_fma(xa, xb, -pr);
                            const double sm = dhi[q] + pr;
                            const double bb = sm - dhi[q];
                            const double se = (dhi[q] - (sm - bb)) + (pr - bb);
                            dhi[q] = sm;
                            dlo[q] += se + pe;
                        }
                    }
                } else if (PACKED) {
                    // X has at most 8 columns: two groups of four pixels side by
                    // side, X' = [X_a | X_b] -- the diagonal 8 x 8 blocks of
                    // X'^T X' are X_a^T X_a and X_b^T X_b (added at the end), and
                    // eight instructions cover the tile instead of sixteen
                    const double *src = jbuf + ((lane >> 4) + ((lane >> 1) & 4)) * JSTRIDE +
                                        (lane & 7);
#pragma unroll
                    for (int t = 0; t < 8; t++) {
                        const double x = src[8 * t * JSTRIDE];
                        M = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, M, 0, 0, 0);
                    }
                } else {
                    // A[i][k] = B[k][i] = X[pixel 4 t + k][i]: lane (i, k) reads one double
                    const double *src = jbuf + (lane >> 4) * JSTRIDE + (lane & 15);
#pragma unroll
                    for (int t = 0; t < 16; t++) {
                        const double x = src[4 * t * JSTRIDE];
                        M = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, M, 0, 0, 0);
                    }
                }
                __syncthreads();
            }
            ty = ty2;
            tx = tx2;
        }
    };
    if (cocen) tiles(std::true_type{});
    else tiles(std::false_type{});

    if constexpr (PRECISE) {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            if (ea[q] < 0) continue;
            const int row = ea[q], col = eb[q];
            const int k = col < NLOC ? row * NLOC - row * (row - 1) / 2 + (col - row)
                                     : (row < NLOC ? NTRI + row : NSUM - 1);
            const double hi = dhi[q] + dlo[q];
            out[k] = hi;
            out[NSUM + k] = dlo[q] - (hi - dhi[q]);
        }
    } else if (!want_jac) {
        const double tot = wave_total(ff);
        // (the jacobian slots are not read for a trial evaluation)
        if (lane == 0) out[NSUM - 1] = tot;
    } else {
        const int col = lane & 15;
        if (PACKED) {
            // entry (row, col) += entry (row + 8, col + 8): eight lanes up, two
            // registers on
            M[0] += __shfl_down(M[2], 8, WAVE);
            M[1] += __shfl_down(M[3], 8, WAVE);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = (lane >> 4) + 4 * r;
            if (row > col || col > NLOC) continue;
            const double v = M[r];
            if (col < NLOC) {
                // upper triangle, row-major: (a, b) -> a NLOC - a (a - 1) / 2 + (b - a)
                out[row * NLOC - row * (row - 1) / 2 + (col - row)] = v;
            } else if (row < NLOC) {
                out[NTRI + row] = v;       // J^T f
            } else {
                out[NSUM - 1] = v;         // f . f
            }
        }
    }
    if (lane == 0 && status) status[s] = NGMIX_OK;
}

// run_leastsq's packaging (leastsqbound.py:33-155) for one fit per thread:
// cov_x as scipy.optimize.leastsq forms it from fjac / ipvt
// (inv((R P^T)^T (R P^T)) = P R^-1 R^-T P^T by back substitution), scaled by
// sum(fdiff^2)/dof; flags from ier and the covariance sanity tests.
// rec (nobj, 4 + n + 2 n^2 + n doubles):
//   [flags, nfev, ier, dof | pars n | pars_err n | cov0 n*n | cov n*n]
// NT > 0: the fit has exactly NT parameters -- compile-time trip counts, the
// work matrices in registers (as lm_core_reg.hpp does for the iteration)
template <int NT>
__device__ __forceinline__ void lm_finalize_one(
    const lm_state &s, int64_t o, const int64_t *__restrict__ npix_obj,
    const double *__restrict__ ff_extra, double pdef, double cdef, double *rec)
{
    constexpr int NS = NT > 0 ? NT : LM_NPMAX;   // stride of the work matrices
    const int n = NT > 0 ? NT : s.n;
    double *r = rec + o * (4 + 2 * (int64_t)n + 2 * (int64_t)n * n);
    double *pars = r + 4, *perr = pars + n, *cov0 = perr + n, *cov = cov0 + n * n;
    int flags = 0;
    const int ier = s.info;
    const long long dof = (long long)npix_obj[o] - n;
    _Pragma("unroll")
    for (int i = 0; i < n; i++) {
        pars[i] = s.x[i];
        perr[i] = cdef;
    }
    _Pragma("unroll")
    for (int i = 0; i < n * n; i++) cov0[i] = cov[i] = cdef;
    if (ier == 0) {
        flags |= NGMIX_FLAG_LM_FUNC_NOTFINITE;
        _Pragma("unroll")
        for (int i = 0; i < n; i++) pars[i] = pdef;
    } else if (ier > 4) {
        flags |= 1 << (ier - 5);
        _Pragma("unroll")
        for (int i = 0; i < n; i++) pars[i] = pdef;
    } else {
        // R^-1 (upper triangular), in the pivoted order.  The factor is read
        // once into a local copy: straight from the record every use was a load
        // waited for on the spot (a thousand of them in the unrolled form)
        double X[NS * NS];
        double Rl[NS * NS];
        _Pragma("unroll")
        for (int i = 0; i < n; i++)
            _Pragma("unroll")
            for (int k = i; k < n; k++) Rl[i * NS + k] = s.R[i * LM_NPMAX + k];
        bool singular = false;
        _Pragma("unroll")
        for (int j = 0; j < n; j++) {
            const double d = Rl[j * NS + j];
            if (d == 0.0 || !(fabs(d) < INFINITY)) singular = true;
        }
        if (!singular) {
            _Pragma("unroll")
            for (int i = 0; i < NS * NS; i++) X[i] = 0.0;
            _Pragma("unroll")
            for (int j = 0; j < n; j++) {
                X[j * NS + j] = 1.0 / Rl[j * NS + j];
                _Pragma("unroll")
                for (int i = j - 1; i >= 0; i--) {
                    double acc = 0.0;
                    _Pragma("unroll")
                    for (int k = i + 1; k <= j; k++)
                        acc += Rl[i * NS + k] * X[k * NS + j];
                    X[i * NS + j] = -acc / Rl[i * NS + i];
                }
            }
            _Pragma("unroll")
            for (int a = 0; a < n; a++)
                _Pragma("unroll")
                for (int b = 0; b < n; b++) {
                    double acc = 0.0;
                    _Pragma("unroll")
                    for (int k = (a > b ? a : b); k < n; k++)
                        acc += X[a * NS + k] * X[b * NS + k];
                    // internal -> external: fjac columns over the transform's
                    // gradient at the solution (leastsqbound.py:535-538)
                    const int pa = s.ipvt[a], pb = s.ipvt[b];
                    if (s.bounded)
                        acc *= lmcore::i2e_grad(s.xi[pa], s.lo[pa], s.hi[pa]) *
                               lmcore::i2e_grad(s.xi[pb], s.lo[pb], s.hi[pb]);
                    cov0[pa * n + pb] = acc;
                    if (!(fabs(acc) < INFINITY)) singular = true;
                }
        }
        if (singular) {
            flags |= NGMIX_FLAG_LM_SINGULAR_MATRIX;
            _Pragma("unroll")
            for (int i = 0; i < n * n; i++) cov0[i] = cdef;
        } else if (dof == 0) {
            flags |= NGMIX_FLAG_ZERO_DOF;
        } else {
            const double s_sq =
                (s.fnorm * s.fnorm - (ff_extra ? ff_extra[o] : 0.0)) / (double)dof;
            bool finite = true;
            _Pragma("unroll")
            for (int i = 0; i < n * n; i++) {
                cov[i] = cov0[i] * s_sq;
                if (!(fabs(cov[i]) < INFINITY)) finite = false;
            }
            int cflags = 0;
            if (!finite) {
                cflags |= NGMIX_FLAG_EIG_NOTFINITE;
            } else {
                // a negative eigenvalue <=> a negative pivot of LDL^T (inertia)
                double A[NS * NS];
                _Pragma("unroll")
                for (int a = 0; a < n; a++)
                    _Pragma("unroll")
                    for (int b = 0; b < n; b++) A[a * NS + b] = cov[a * n + b];
                bool neg = false, negdiag = false;
                _Pragma("unroll")
                for (int a = 0; a < n; a++)
                    if (cov[a * n + a] < 0.0) negdiag = true;
                _Pragma("unroll")
                for (int k = 0; k < n; k++) {
                    const double d = A[k * NS + k];
                    if (d < 0.0) neg = true;
                    const double safe = d != 0.0 ? d : 1.0;
                    _Pragma("unroll")
                    for (int a = k + 1; a < n; a++) {
                        const double c = A[a * NS + k] / safe;
                        _Pragma("unroll")
                        for (int b = k + 1; b < n; b++)
                            A[a * NS + b] -= c * A[k * NS + b];
                    }
                }
                if (neg) cflags |= NGMIX_FLAG_LM_NEG_COV_EIG;
                if (negdiag) cflags |= NGMIX_FLAG_LM_NEG_COV_DIAG;
            }
            flags |= cflags;
            if (cflags == 0)
                _Pragma("unroll")
                for (int a = 0; a < n; a++) perr[a] = sqrt(cov[a * n + a]);
        }
    }
    r[0] = (double)flags;
    r[1] = ier == 0 ? -1.0 : (double)s.nfev;
    r[2] = (double)ier;
    r[3] = (double)dof;
}

__global__ __launch_bounds__(WAVE) void lm_finalize_kernel(
    const lm_state *__restrict__ states, int64_t nobj,
    const int64_t *__restrict__ npix_obj, const double *__restrict__ ff_extra,
    double pdef, double cdef, double *rec)
{
    const int64_t o = blockIdx.x * (int64_t)WAVE + threadIdx.x;
    if (o >= nobj) return;
    const lm_state &s = states[o];
    // (the fits of a batch have one parameter count: a uniform branch)
    const int n = s.n;
    if (n == 6) lm_finalize_one<6>(s, o, npix_obj, ff_extra, pdef, cdef, rec);
    else if (n == 7) lm_finalize_one<7>(s, o, npix_obj, ff_extra, pdef, cdef, rec);
    else if (n == 8) lm_finalize_one<8>(s, o, npix_obj, ff_extra, pdef, cdef, rec);
    else if (n == 9) lm_finalize_one<9>(s, o, npix_obj, ff_extra, pdef, cdef, rec);
    else if (n == 10) lm_finalize_one<10>(s, o, npix_obj, ff_extra, pdef, cdef, rec);
    else lm_finalize_one<0>(s, o, npix_obj, ff_extra, pdef, cdef, rec);
}

__global__ __launch_bounds__(BLOCK) void lm_prior_sums_kernel(
    const lm_state *__restrict__ states, int64_t nobj, ngmix_simple_sep_prior P,
    double step_rel, double *__restrict__ obj_sums)
{
    const int64_t o = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
    if (o >= nobj) return;
    const lm_state &s = states[o];
    const int n = s.n;
    if (s.phase == LM_PHASE_DONE) return;
    // (straight into the fit's slot: a private copy of the sums is 960 bytes
    // of scratch per thread)
    lmcore::simple_sep_normal_sums(P, s, step_rel,
                                   obj_sums + o * (int64_t)(n * (n + 1) / 2 + n + 1));
}

// what the prior contributes to the statistics of a finished fit, at the
// point the fit stands at (states[i].x): ffx = the sum of squares of its
// finite rows -- the part of |f|^2 run_leastsq leaves out of chi2/dof
// (leastsqbound.py:97) -- and ln p (calc_lnprob adds it, results.py:410-437);
// a point outside the prior's range: ffx 0, ln p -inf
__global__ __launch_bounds__(BLOCK) void lm_prior_finish_kernel(
    const lm_state *__restrict__ states, int64_t nobj, ngmix_simple_sep_prior P,
    double *__restrict__ ffx, double *__restrict__ lnp)
{
    const int64_t o = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
    if (o >= nobj) return;
    const lm_state &s = states[o];
    double x[lmcore::PRIOR_NMAX], rows[lmcore::PRIOR_KMAX];
#pragma unroll
    for (int j = 0; j < lmcore::PRIOR_NMAX; j++) x[j] = j < s.n ? s.x[j] : 0.0;
    double tot = 0.0, ff = 0.0;
    if (s.n > lmcore::PRIOR_NMAX || !lmcore::simple_sep_rows(P, x, rows, &tot)) {
        tot = -INFINITY;
    } else {
        const int k = 4 + P.nmid + P.nband;
        for (int i = 0; i < k; i++) ff += rows[i] * rows[i];
        if (!(fabs(ff) < INFINITY)) ff = 0.0;
    }
    ffx[o] = ff;
    lnp[o] = tot;
}

int launch_lm_prior_finish(const lm_state *states, int64_t nobj,
                           const ngmix_simple_sep_prior *prior, double *ffx, double *lnp,
                           hipStream_t s)
{
    if (nobj <= 0) return NGMIX_OK;
    if (!prior || !ffx || !lnp || prior->nband < 1 || prior->nband > NGMIX_PRIOR_MAXBAND ||
        prior->nmid < 0 || prior->nmid > NGMIX_PRIOR_MAXMID)
        return NGMIX_ERR_BAD_ARG;
    hipLaunchKernelGGL(lm_prior_finish_kernel, dim3((unsigned)((nobj + BLOCK - 1) / BLOCK)),
                       dim3(BLOCK), 0, s, states, nobj, *prior, ffx, lnp);
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

int launch_lm_prior_sums(const lm_state *states, int64_t nobj,
                         const ngmix_simple_sep_prior *prior, double step_rel,
                         double *obj_sums, hipStream_t s)
{
    if (nobj <= 0) return NGMIX_OK;
    if (!prior || prior->nband < 1 || prior->nband > NGMIX_PRIOR_MAXBAND || prior->nmid < 0 ||
        prior->nmid > NGMIX_PRIOR_MAXMID)
        return NGMIX_ERR_BAD_ARG;
    hipLaunchKernelGGL(lm_prior_sums_kernel, dim3((unsigned)((nobj + BLOCK - 1) / BLOCK)),
                       dim3(BLOCK), 0, s, states, nobj, *prior, step_rel, obj_sums);
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

struct LmInitPars {
    double ftol, xtol, gtol, factor;
    double lo[LM_NPMAX], hi[LM_NPMAX];
    int n, maxfev, mode, has_bounds;
};

// lmcore::lm_init, the LIVE part of the record only: the scalars, the first n
// entries of every per-parameter array and the leading n x n block of R --
// what a fit of n parameters ever reads.  The record is sized for
// NGMIX_LM_NPMAX = 14 parameters (2.9 kB); zero-filling all of it first (round
// 3: a 290 MB hipMemsetAsync per 100k fits, 0.33 ms of HBM time per batch, 5 %
// of a config-3 step) wrote 2 kB per six-parameter fit that nothing reads.
// The dead part keeps whatever the allocation held.
//
// Sixteen lanes per fit (round 5; one thread per fit before: 132 stores of 8
// bytes, each 2.9 kB from its neighbour lane's -- 0.35 ms per 100k fits, a
// twentieth of a config-3 fit).  Lane j writes entry j of every per-parameter
// array, so a store instruction covers 48-112 contiguous bytes of each of the
// wave's four records; the leading n rows of R are zeroed as one contiguous
// run (their dead columns included), the ten double and ten int32 scalars by
// lanes 0-9.
constexpr int LM_INIT_LANES = 16;
static_assert(LM_NPMAX <= LM_INIT_LANES, "a lane per parameter");
static_assert(offsetof(lm_state, factor) - offsetof(lm_state, fnorm) == 9 * sizeof(double) &&
                  offsetof(lm_state, fonly) - offsetof(lm_state, n) == 9 * sizeof(int32_t),
              "lm_init_kernel writes the scalars as two runs of ten");

__global__ __launch_bounds__(BLOCK) void lm_init_kernel(lm_state *states, int64_t nobj,
                                                        const double *__restrict__ x0,
                                                        LmInitPars P)
{
    const int j = threadIdx.x & (LM_INIT_LANES - 1);
    const int64_t o = (blockIdx.x * (int64_t)BLOCK + threadIdx.x) / LM_INIT_LANES;
    if (o >= nobj) return;
    lm_state &s = states[o];
    const int n = P.n;
    const int bounded = P.has_bounds;
    if (j < LM_NPMAX) {
        s.lo[j] = P.lo[j];
        s.hi[j] = P.hi[j];
        s.ipvt[j] = j;
    }
    constexpr double EPS = 1.4901161193847656e-08;  // sqrt(machine epsilon)
    if (j < n) {
        // i0 = e2i(x0); the first evaluation is at i2e(i0) (leastsqbound.py:454)
        const double x = x0[o * n + j];
        const double xi = bounded ? lmcore::e2i(x, P.lo[j], P.hi[j]) : x;
        const double xt = bounded ? lmcore::i2e(xi, P.lo[j], P.hi[j]) : xi;
        s.xi[j] = xi;
        s.xti[j] = xi;
        s.xt[j] = xt;
        s.x[j] = xt;
        double h = 0.0, xs = 0.0;
        if (P.mode == NGMIX_LM_MODE_FD) {
            h = EPS * fabs(xi);
            if (h == 0.0) h = EPS;
            xs = bounded ? lmcore::i2e(xi + h, P.lo[j], P.hi[j]) : xi + h;
        }
        s.hstep[j] = h;
        s.xstep[j] = xs;
        s.diag[j] = 0.0;
        s.qtf[j] = 0.0;
        s.step[j] = 0.0;
    }
    for (int e = j; e < n * LM_NPMAX; e += LM_INIT_LANES) s.R[e] = 0.0;
    if (j < 10) {
        // fnorm, xnorm, delta, par, gnorm, pnorm, ftol, xtol, gtol, factor
        const double dv = j < 6 ? 0.0 : (j == 6 ? P.ftol : (j == 7 ? P.xtol : (j == 8 ? P.gtol
                                                                                  : P.factor)));
        (&s.fnorm)[j] = dv;
        // n, iter, nfev, njev, info, phase, maxfev, mode, bounded, fonly
        int iv = 0;
        if (j == 0) iv = n;
        else if (j == 1) iv = 1;
        else if (j == 5) iv = LM_PHASE_INIT;
        else if (j == 6) iv = P.maxfev;
        else if (j == 7) iv = P.mode;
        else if (j == 8) iv = bounded;
        (&s.n)[j] = iv;
    }
}

int launch_lm_init(lm_state *states, int64_t nobj, int npars, const double *x0,
                   double ftol, double xtol, double gtol, int maxfev, double factor,
                   int mode, const double *lo, const double *hi, hipStream_t s)
{
    if (nobj <= 0) return NGMIX_OK;
    if (npars < 1 || npars > LM_NPMAX) return NGMIX_ERR_BAD_ARG;
    LmInitPars P;
    P.ftol = ftol;
    P.xtol = xtol;
    P.gtol = gtol;
    P.factor = factor;
    P.n = npars;
    P.maxfev = maxfev;
    P.mode = mode;
    for (int j = 0; j < LM_NPMAX; j++) {
        P.lo[j] = (lo && j < npars) ? lo[j] : -INFINITY;
        P.hi[j] = (hi && j < npars) ? hi[j] : INFINITY;
    }
    // (the states are bounded when any bound is finite, as lmcore::lm_init has it)
    P.has_bounds = 0;
    for (int j = 0; j < LM_NPMAX; j++)
        if (P.lo[j] > -INFINITY || P.hi[j] < INFINITY) P.has_bounds = 1;
    constexpr int PER_BLOCK = BLOCK / LM_INIT_LANES;
    hipLaunchKernelGGL(lm_init_kernel, dim3((unsigned)((nobj + PER_BLOCK - 1) / PER_BLOCK)),
                       dim3(BLOCK), 0, s, states, nobj, x0, P);
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

int launch_lm_finalize(const lm_state *states, int64_t nobj, const int64_t *npix_obj,
                       const double *ff_extra, double pdef, double cdef, double *rec,
                       hipStream_t s)
{
    if (nobj <= 0) return NGMIX_OK;
    hipLaunchKernelGGL(lm_finalize_kernel, dim3((unsigned)((nobj + WAVE - 1) / WAVE)),
                       dim3(WAVE), 0, s, states, nobj, npix_obj, ff_extra, pdef, cdef, rec);
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

// FitModel.set_fit_result's statistics (results.py:45-72, 398-408) and the
// integer columns of run_leastsq's packaging, one thread per fit, laid out for
// ONE contiguous download: head (nobj, 2 n) = pars | pars_err row-major, then
// cols (NGMIX_LM_NCOLS, nobj) column-major -- every host array is then a
// contiguous view of the downloaded buffer
__global__ __launch_bounds__(256) void lm_pack_kernel(
    const lm_state *__restrict__ states, int64_t nobj, int n, const double *__restrict__ rec,
    const double *__restrict__ obj_stats, const double *__restrict__ tot,
    const int64_t *__restrict__ npix_obj, double *__restrict__ head,
    double *__restrict__ cols, double *__restrict__ cov_tri)
{
    const int64_t o = blockIdx.x * (int64_t)256 + threadIdx.x;
    if (o >= nobj) return;
    const double *r = rec + o * (4 + 2 * (int64_t)n + 2 * (int64_t)n * n);
    for (int k = 0; k < 2 * n; k++) head[o * 2 * n + k] = r[4 + k];
    if (cov_tri) {
        // pars_cov is symmetric to the bit (lm_finalize_one forms (a, b) and
        // (b, a) from the same products in the same order): its upper triangle
        // is all the host needs -- 21 instead of 36 doubles per six-parameter fit
        const double *cov = r + 4 + 2 * (int64_t)n + (int64_t)n * n;
        double *t = cov_tri + o * (int64_t)(n * (n + 1) / 2);
        int k = 0;
        for (int a = 0; a < n; a++)
            for (int b = a; b < n; b++) t[k++] = cov[a * n + b];
    }
    const bool ok = r[0] == 0.0;
    double lnprob, numer, denom, npix;
    if (obj_stats) {
        // the loop carried them: lnprob = -|f|^2 / 2 where the fit stands
        const double fn = states[o].fnorm;
        lnprob = -0.5 * fn * fn;
        numer = obj_stats[2 * o];
        denom = obj_stats[2 * o + 1];
        npix = (double)npix_obj[o];
    } else {
        lnprob = tot[4 * o];
        numer = tot[4 * o + 1];
        denom = tot[4 * o + 2];
        npix = rint(tot[4 * o + 3]);
    }
    const double dof = npix - (double)n;
    const double s2n = denom > 0.0 ? numer / sqrt(denom) : 0.0;
    double *c = cols + o;
    c[0 * nobj] = r[0];                       // flags
    c[1 * nobj] = r[1];                       // nfev
    c[2 * nobj] = r[2];                       // ier
    c[3 * nobj] = r[3];                       // dof of run_leastsq
    c[4 * nobj] = (double)states[o].njev;
    c[5 * nobj] = ok ? lnprob : NAN;
    c[6 * nobj] = ok ? numer : NAN;
    c[7 * nobj] = ok ? denom : NAN;
    c[8 * nobj] = npix;
    c[9 * nobj] = dof;
    c[10 * nobj] = ok ? lnprob / (-0.5) / dof : NAN;   // results.py:64
    c[11 * nobj] = ok ? s2n : NAN;
}

int launch_lm_pack(const lm_state *states, int64_t nobj, int npars, const double *rec,
                   const double *obj_stats, const double *tot, const int64_t *npix_obj,
                   double *head, double *cols, double *cov_tri, hipStream_t s)
{
    if (nobj <= 0) return NGMIX_OK;
    if (!states || !rec || !head || !cols || npars < 1 || npars > LM_NPMAX ||
        (obj_stats ? !npix_obj : !tot)) {
        set_last_error_msg("lm_pack: bad argument");
        return NGMIX_ERR_BAD_ARG;
    }
    hipLaunchKernelGGL(lm_pack_kernel, dim3((unsigned)((nobj + 255) / 256)), dim3(256), 0, s,
                       states, nobj, npars, rec, obj_stats, tot, npix_obj, head, cols, cov_tri);
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

static int model_ngauss_npars(int model, int &ng0, int &nloc)
{
    switch (model) {
    case NGMIX_MODEL_GAUSS: ng0 = 1; nloc = 6; return 0;
    case NGMIX_MODEL_TURB: ng0 = 3; nloc = 6; return 0;
    case NGMIX_MODEL_EXP: ng0 = 6; nloc = 6; return 0;
    case NGMIX_MODEL_DEV: ng0 = 10; nloc = 6; return 0;
    case NGMIX_MODEL_BDF: ng0 = 16; nloc = 7; return 0;
    case NGMIX_MODEL_BD: ng0 = 16; nloc = 8; return 0;
    default: break;
    }
    // co-elliptical gaussians: the count rides in the model argument,
    // NGMIX_MODEL_COELLIP + 256 * ngauss (parameters: cen1, cen2, g1, g2,
    // T_1..T_n, F_1..F_n; one band)
    if ((model & 0xff) == NGMIX_MODEL_COELLIP) {
        const int n = model >> 8;
        if (n < 1 || 4 + 2 * n > LM_NPMAX) return -1;
        ng0 = n;
        nloc = 4 + 2 * n;
        return 0;
    }
    return -1;
}

int launch_lm_eval(const ngmix_batch *b, int model, int fd, const lm_state *states,
                   const int32_t *stamp_obj, const int32_t *stamp_band,
                   const ngmix_gauss2d *psf, int npsf, double *sums, int32_t *status,
                   double *stamp_stats, hipStream_t s, double *jac_point, bool precise)
{
    if (b->nstamps <= 0) return NGMIX_OK;
    if (precise && (!fd || !jac_point)) {
        set_last_error_msg("lm_eval: the precise pass is a forward-difference pass at jac_point");
        return NGMIX_ERR_BAD_ARG;
    }
    int ng0, nloc;
    if (model_ngauss_npars(model, ng0, nloc) != 0 || npsf < 0) {
        set_last_error_msg("lm_eval: model must be gauss, turb, exp, dev, bdf, bd or "
                           "coellip + 256 * ngauss (ngauss <= 3)");
        return NGMIX_ERR_BAD_ARG;
    }
    model &= 0xff;
    const int no_skip = (b->flags & NGMIX_BATCH_NO_SKIP) ? 1 : 0;
    const int G = ng0 * (npsf > 0 ? npsf : 1);
    dim3 grid((unsigned)b->nstamps), block(WAVE);
    if (!fd) {
        if (!(model == NGMIX_MODEL_GAUSS || model == NGMIX_MODEL_EXP ||
              model == NGMIX_MODEL_DEV)) {
            set_last_error_msg("lm_eval: the analytic jacobian exists for gauss, exp, dev");
            return NGMIX_ERR_BAD_ARG;
        }
        // one 32-byte record per 8x8 tile next to the gaussians (exact when the
        // batch carries its largest stamp shape); beyond LM_TILE_CAP tiles the
        // kernel makes the records on the fly
        int tile_cap = b->max_npix / 8 + 1;
        if (b->max_nrow > 0 && b->max_ncol > 0)
            tile_cap = ((b->max_nrow + TILE_H - 1) / TILE_H) *
                       ((b->max_ncol + TILE_W - 1) / TILE_W);
        tile_cap += 1;
        const bool lds_tiles = tile_cap <= LM_TILE_CAP;
        if (!lds_tiles) tile_cap = 0;
        const size_t lds = (size_t)G * sizeof(DerivGauss) + (size_t)tile_cap * sizeof(LmTile);
        if (lds > 96 * 1024) {
            set_last_error_msg("lm_eval: too many composed gaussians for LDS");
            return NGMIX_ERR_BAD_ARG;
        }
        static const bool jbasis = getenv("NGMIX_LM_JBASIS") != nullptr;   // A/B knob
        if (jbasis && stamp_stats) {
            set_last_error_msg("lm_eval: NGMIX_LM_JBASIS carries no statistics");
            return NGMIX_ERR_BAD_ARG;
        }
        const void *kern =
            jbasis ? (lds_tiles ? (const void *)lm_eval_kernel<true, false>
                                : (const void *)lm_eval_kernel<false, false>)
                   : (lds_tiles ? (const void *)lm_eval_kernel<true, true>
                                : (const void *)lm_eval_kernel<false, true>);
        if (lds > 48 * 1024)
            NGMIX_HIP_CHECK(hipFuncSetAttribute(
                kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const ngmix_stamp *a_stamps = b->stamps;
        const double *a_val = b->val, *a_ierr = b->ierr;
        const ngmix_jacobian *a_jac = b->jac;
        int a_model = model, a_ng0 = ng0, a_npsf = npsf, a_ns = no_skip;
        void *args[] = {&a_stamps, &a_val, &a_ierr, &a_jac, &a_model, &a_ng0, &states,
                        &stamp_obj, &stamp_band, &psf, &a_npsf, &sums, &status, &a_ns,
                        &tile_cap, &stamp_stats};
        census(jbasis ? (lds_tiles ? "lm_eval_kernel<true, false>" : "lm_eval_kernel<false, false>")
                      : (lds_tiles ? "lm_eval_kernel<true, true>" : "lm_eval_kernel<false, true>"));
        NGMIX_HIP_CHECK(hipLaunchKernel(kern, grid, block, args, lds, s));
        return NGMIX_OK;
    }
    if (stamp_stats) {
        set_last_error_msg("lm_eval: the loglike statistics come with the analytic "
                           "kernel only (stamp_stats must be NULL in forward-difference mode)");
        return NGMIX_ERR_BAD_ARG;
    }
    const size_t lds = (size_t)(nloc + 1) * G * sizeof(FdGauss) + (size_t)G * sizeof(TileBox);
    if (lds > 128 * 1024) {
        set_last_error_msg("lm_eval: too many composed gaussians for LDS");
        return NGMIX_ERR_BAD_ARG;
    }
    // row-major tiles when the batch's largest shape needs a fifth fewer of them
    // than 8 x 8 tiles (NGMIX_LM_FD_TILES = 2d | linear forces one form: A/B)
    bool linear = false;
    {
        const int ntl = (b->max_nrow * b->max_ncol + WAVE - 1) / WAVE;
        const int nt2d = ((b->max_ncol + TILE_W - 1) / TILE_W) * ((b->max_nrow + TILE_H - 1) / TILE_H);
        linear = b->max_nrow > 0 && b->max_ncol > 0 && 5 * ntl <= 4 * nt2d;
        const char *e = getenv("NGMIX_LM_FD_TILES");
        if (e) linear = e[0] == 'l';
    }
#define NGMIX_FD_LAUNCH2(N, L, P, NAME)                                                 \
    do {                                                                                \
        census(NAME);                                                                   \
        if (lds > 48 * 1024)                                                            \
            NGMIX_HIP_CHECK(hipFuncSetAttribute(                                        \
                (const void *)lm_eval_fd_kernel<N, L, P>,                               \
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                 \
        hipLaunchKernelGGL((lm_eval_fd_kernel<N, L, P>), grid, block, lds, s, b->stamps, \
                           b->val, b->ierr, b->jac, model, ng0, states, stamp_obj,      \
                           stamp_band, psf, npsf, sums, status, no_skip, jac_point);    \
    } while (0)
#define NGMIX_FD_LAUNCH1(N, L)                                                          \
    NGMIX_FD_LAUNCH2(N, L, false,                                                       \
                     L ? "lm_eval_fd_kernel<" #N ", linear>" : "lm_eval_fd_kernel<" #N ">")
    if (precise) {
        // (built for the fits it serves: nine local parameters and up)
        if (nloc < NGMIX_LM_PRECISE_MIN_NLOC) {
            set_last_error_msg("lm_eval: the precise pass serves nloc >= 9");
            return NGMIX_ERR_BAD_ARG;
        }
#define NGMIX_FD_PRECISE(N)                                                                  \
    do {                                                                                     \
        if (linear) NGMIX_FD_LAUNCH2(N, true, true, "lm_eval_fd_kernel<" #N ", linear, precise>"); \
        else NGMIX_FD_LAUNCH2(N, false, true, "lm_eval_fd_kernel<" #N ", precise>");        \
    } while (0)
        if (nloc <= 10) NGMIX_FD_PRECISE(10);
        else if (nloc <= 12) NGMIX_FD_PRECISE(12);
        else NGMIX_FD_PRECISE(14);
#undef NGMIX_FD_PRECISE
        NGMIX_HIP_CHECK(hipGetLastError());
        return NGMIX_OK;
    }
#define NGMIX_FD_LAUNCH(N)                                                              \
    do {                                                                                \
        if (linear) NGMIX_FD_LAUNCH1(N, true);                                          \
        else NGMIX_FD_LAUNCH1(N, false);                                                \
    } while (0)
    if (nloc == 6) NGMIX_FD_LAUNCH(6);
    else if (nloc == 7) NGMIX_FD_LAUNCH(7);
    else if (nloc == 8) NGMIX_FD_LAUNCH(8);
    else if (nloc <= 10) NGMIX_FD_LAUNCH(10);
    else if (nloc <= 12) NGMIX_FD_LAUNCH(12);
    else NGMIX_FD_LAUNCH(14);
#undef NGMIX_FD_LAUNCH2
#undef NGMIX_FD_LAUNCH1
#undef NGMIX_FD_LAUNCH
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

int launch_lm_advance(lm_state *states, int64_t nobj, const int64_t *obj_start,
                      const int32_t *stamp_band, const double *sums, int nloc,
                      const double *obj_sums, int32_t *nactive, const double *stamp_stats,
                      double *obj_stats, hipStream_t s, bool zero_count)
{
    if (nobj <= 0) return NGMIX_OK;
    // nloc + 256 * npars: the fits' parameter count, if the caller says
    // (NGMIX_LM_NPARS_GENERIC: the generic one-thread code, whatever the count)
    int npars = (nloc >> 8) & 0xff;
    nloc &= 0xff;
    const bool ask_generic = npars == NGMIX_LM_NPARS_GENERIC;
    if (ask_generic) npars = 0;
    if (nloc < 2 || nloc > LM_NPMAX || npars > LM_NPMAX) return NGMIX_ERR_BAD_ARG;
    if (npars != 0 && npars < nloc) return NGMIX_ERR_BAD_ARG;
    if ((stamp_stats == nullptr) != (obj_stats == nullptr)) {
        // "both or neither" (ngmix_hip.h): with one of them the kernel would skip
        // the statistics silently and lm_pack would package an unwritten buffer
        set_last_error_msg("lm_advance: stamp_stats and obj_stats go together");
        return NGMIX_ERR_BAD_ARG;
    }
    if (nactive && zero_count)
        NGMIX_HIP_CHECK(hipMemsetAsync(nactive, 0, sizeof(int32_t), s));
    const dim3 grid((unsigned)((nobj + WAVE - 1) / WAVE)), block(WAVE);
    const bool generic = ask_generic || getenv("NGMIX_LM_GENERIC") != nullptr;   // A/B knob
    // The team form (16 lanes per fit, arrays in LDS: lm_team.hip) from 9
    // parameters up: the register form holds 6-8 without spilling and is ahead
    // there (one fit per lane: 16 times fewer wave instructions per fit); at 9 and
    // 10 it spills and the team form is 2x ahead (tools/lm_advance_sweep.py), and
    // beyond 10 the alternative is the private-memory code.  A/B knobs:
    // NGMIX_LM_TEAM_MIN = the smallest parameter count that goes to the team form
    // (6 = every count said), NGMIX_LM_TEAMS = fits per wave (1, 2, 4).
    // (read at every launch: the tests switch them between calls)
    // (getenv is not safe against a concurrent setenv: switch the knobs from one
    // thread, between calls; a value outside 6..LM_NPMAX + 1 is ignored)
    const char *e_min = getenv("NGMIX_LM_TEAM_MIN"), *e_teams = getenv("NGMIX_LM_TEAMS");
    int team_min = e_min ? atoi(e_min) : 9;
    if (team_min < 6 || team_min > LM_NPMAX + 1) team_min = 9;
    const int teams = e_teams ? atoi(e_teams) : 4;
    // (a count not said: the team form built for LM_NPMAX parameters serves any
    // fit and is ahead of the one-thread code at every count and batch size --
    // n = 6: 0.28 against 0.34 ms per launch of 100k fits, n = 8: 0.33 against 0.66)
    if (npars == 0 && !generic)
        return launch_lm_advance_team(states, nobj, obj_start, stamp_band, sums, nloc, LM_NPMAX,
                                      obj_sums, nactive, stamp_stats, obj_stats,
                                      teams == 1 || teams == 2 ? teams : 4, s);
    if (npars >= team_min && !generic)
        return launch_lm_advance_team(states, nobj, obj_start, stamp_band, sums, nloc, npars,
                                      obj_sums, nactive, stamp_stats, obj_stats,
                                      teams == 1 || teams == 2 ? teams : 4, s);
    {
        char name[64];
        if (npars >= 6 && npars <= 10 && !generic)
            snprintf(name, sizeof(name), "lm_advance_kernel<%d, true>", npars);
        else
            snprintf(name, sizeof(name), "lm_advance_kernel<%d, false>", LM_NPMAX);
        census(name);
    }
    if (npars == 6 && !generic)
        hipLaunchKernelGGL((lm_advance_kernel<6, true>), grid, block, 0, s, states, nobj,
                           obj_start, stamp_band, sums, nloc, obj_sums, nactive, stamp_stats,
                           obj_stats);
    else if (npars == 7 && !generic)
        hipLaunchKernelGGL((lm_advance_kernel<7, true>), grid, block, 0, s, states, nobj,
                           obj_start, stamp_band, sums, nloc, obj_sums, nactive, stamp_stats,
                           obj_stats);
    else if (npars == 8 && !generic)
        hipLaunchKernelGGL((lm_advance_kernel<8, true>), grid, block, 0, s, states, nobj,
                           obj_start, stamp_band, sums, nloc, obj_sums, nactive, stamp_stats,
                           obj_stats);
    else if (npars == 9 && !generic)
        hipLaunchKernelGGL((lm_advance_kernel<9, true>), grid, block, 0, s, states, nobj,
                           obj_start, stamp_band, sums, nloc, obj_sums, nactive, stamp_stats,
                           obj_stats);
    else if (npars == 10 && !generic)
        hipLaunchKernelGGL((lm_advance_kernel<10, true>), grid, block, 0, s, states, nobj,
                           obj_start, stamp_band, sums, nloc, obj_sums, nactive, stamp_stats,
                           obj_stats);
    else
        hipLaunchKernelGGL((lm_advance_kernel<LM_NPMAX, false>), grid, block, 0, s, states,
                           nobj, obj_start, stamp_band, sums, nloc, obj_sums, nactive,
                           stamp_stats, obj_stats);
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

// nrounds x {pixel pass, prior rows, lmder step} queued by one host call
// (ngmix_lm_rounds_batch): the host is not in the lock-step loop.  A finished
// fit is skipped by every later launch (its phase is read on the device), so
// the caller may queue rounds blind.
int launch_lm_rounds(const ngmix_lm_problem *p, int nrounds, int32_t *counts,
                     int32_t *counts_host, void **events, hipStream_t s)
{
    if (!p || !p->batch || !p->states || !p->sums || nrounds < 0) {
        set_last_error_msg("lm_rounds: problem, batch, states and sums are required");
        return NGMIX_ERR_BAD_ARG;
    }
    if (counts_host && !counts) {
        set_last_error_msg("lm_rounds: counts_host needs counts");
        return NGMIX_ERR_BAD_ARG;
    }
    if (p->prior && !p->obj_sums) {
        set_last_error_msg("lm_rounds: a prior needs obj_sums");
        return NGMIX_ERR_BAD_ARG;
    }
    if (nrounds == 0 || p->nobj <= 0) return NGMIX_OK;
    if (counts)
        NGMIX_HIP_CHECK(hipMemsetAsync(counts, 0, (size_t)nrounds * sizeof(int32_t), s));
    for (int r = 0; r < nrounds; r++) {
        if (events) NGMIX_HIP_CHECK(hipEventRecord((hipEvent_t)events[3 * r], s));
        int rc = launch_lm_eval(p->batch, p->model, p->fd, p->states, p->stamp_obj,
                                p->stamp_band, p->psf, p->npsf, p->sums, p->status,
                                p->stamp_stats, s, p->fd ? p->jac_point : nullptr);
        if (rc != NGMIX_OK) return rc;
        if (events) NGMIX_HIP_CHECK(hipEventRecord((hipEvent_t)events[3 * r + 1], s));
        if (p->prior) {
            rc = launch_lm_prior_sums(p->states, p->nobj, p->prior, p->prior_step,
                                      p->obj_sums, s);
            if (rc != NGMIX_OK) return rc;
        }
        rc = launch_lm_advance(p->states, p->nobj, p->obj_start, p->stamp_band, p->sums,
                               p->nloc_npars, p->obj_sums, counts ? counts + r : nullptr,
                               p->stamp_stats, p->obj_stats, s, false);
        if (rc != NGMIX_OK) return rc;
        if (events) NGMIX_HIP_CHECK(hipEventRecord((hipEvent_t)events[3 * r + 2], s));
    }
    if (counts_host)
        NGMIX_HIP_CHECK(hipMemcpyAsync(counts_host, counts, (size_t)nrounds * sizeof(int32_t),
                                       hipMemcpyDeviceToHost, s));
    return NGMIX_OK;
}

}  // namespace ngmix
