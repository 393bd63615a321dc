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
#include "device_utils.hpp"
#include "launch.hpp"
#include "lm_core.hpp"


namespace ngmix {

__constant__ ModelTables c_tables_lm = NGMIX_MODEL_TABLES;
__constant__ double c_exp_table_lm[16] = NGMIX_EXP_TABLE;
__constant__ double c_fexp_coef_lm[9] = NGMIX_FEXP_COEF;

constexpr int LM_NSUM = NGMIX_LM_NSUM;  // 21 + 6 + 1

// one composed gaussian, staged in LDS
struct DerivGauss {
    double row, col;
    double w11, w12, w22;  // Q = Sigma^-1
    double pa;             // norm * area
    double d[3][3];        // halved d(irr,irc,icc)/d(g1,g2,T), middle entry doubled
    double trh[3];         // tr(Q dSigma_a) / 2
    PixBox box;
};
static_assert(sizeof(DerivGauss) == 160, "DerivGauss");

template <int CTRL>
__device__ __forceinline__ double dpp_row_shr_zero(double x)
{
    // row_shr within each row of 16 lanes; lanes without a source read 0
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// sum of x over each row of 16 lanes, valid in the row's last lane
__device__ __forceinline__ double row16_sum(double x)
{
    x += dpp_row_shr_zero<0x111>(x);
    x += dpp_row_shr_zero<0x112>(x);
    x += dpp_row_shr_zero<0x114>(x);
    x += dpp_row_shr_zero<0x118>(x);
    return x;
}

struct LmEvalShared {
    double tabr[16];
    double red[LM_NSUM * 4];
    int ctl[4];
};

__global__ __launch_bounds__(WAVE) void lm_eval_kernel(
    const ngmix_stamp *__restrict__ stamps, const double *__restrict__ val,
    const double *__restrict__ ierr, const ngmix_jacobian *__restrict__ jacs,
    int model, int ng0, const lm_state *__restrict__ states,
    const int32_t *__restrict__ stamp_obj, const int32_t *__restrict__ stamp_band,
    const ngmix_gauss2d *__restrict__ psf, int npsf, double *__restrict__ sums,
    int32_t *__restrict__ status, int no_skip)
{
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    __shared__ LmEvalShared sh;
    DerivGauss *dg = (DerivGauss *)dyn;

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

    // ---- the composed mixture and its derivative data at the trial point
    double p[6];
    for (int k = 0; k < 5; k++) p[k] = state.xt[k];
    p[5] = state.xt[5 + band];
    const double g1 = p[2], g2 = p[3], T = p[4], flux = p[5];
    const int npsf1 = npsf > 0 ? npsf : 1;
    const int G = ng0 * npsf1;
    int bad = 0;
    if (T == 0.0 || flux == 0.0) bad = 1;  // results.py:527-531
    FillCtx c;
    if (fill_prepare(c_tables_lm, model, ng0, p, nullptr, c) != NGMIX_OK) bad = 1;
    double rowcen = 0.0, colcen = 0.0, ipsum = 1.0;
    const ngmix_gauss2d *q = psf ? psf + (size_t)s * npsf : nullptr;
    if (npsf > 0) {
        double psum;
        if (gmix_cen(q, npsf, rowcen, colcen, psum) != NGMIX_OK) bad = 1;
        else ipsum = 1.0 / psum;
    }
    if (lane < 16) sh.tabr[lane] = c_exp_table_lm[15 - lane];
    if (!bad) {
        // d(e1, e2)/d(g1, g2) for e = 2 g / (1 + g^2)  (results.py:985-992)
        const double gsq = g1 * g1 + g2 * g2;
        const double f = 2.0 / (1.0 + gsq);
        const double dfac = -f / (1.0 + gsq);
        const double de1dg1 = f + 2.0 * g1 * g1 * dfac;
        const double de1dg2 = 2.0 * g1 * g2 * dfac;
        const double de2dg1 = de1dg2;
        const double de2dg2 = f + 2.0 * g2 * g2 * dfac;
        for (int i = lane; i < G; i += WAVE) {
            const int io = i / npsf1, ip = i - io * npsf1;
            ngmix_gauss2d g0, gc;
            fill_component(c_tables_lm, c, p, io, g0);
            if (npsf > 0) convolve_component(g0, q[ip], rowcen, colcen, ipsum, gc);
            else gc = g0;
            if (gauss_set_norm(gc) != NGMIX_OK) bad = 1;
            DerivGauss r;
            r.row = gc.row;
            r.col = gc.col;
            r.w11 = gc.dcc;   // icc / det
            r.w12 = -gc.drc;  // -irc / det
            r.w22 = gc.drr;   // irr / det
            r.pa = gc.pnorm * area;
            const double Tk = g0.irr + g0.icc;
            const double dc[3][3] = {
                {-0.5 * Tk * de1dg1, 0.5 * Tk * de2dg1, 0.5 * Tk * de1dg1},
                {-0.5 * Tk * de1dg2, 0.5 * Tk * de2dg2, 0.5 * Tk * de1dg2},
                {g0.irr / T, g0.irc / T, g0.icc / T}};
            for (int a = 0; a < 3; a++) {
                r.d[a][0] = 0.5 * dc[a][0];
                r.d[a][1] = dc[a][1];
                r.d[a][2] = 0.5 * dc[a][2];
                r.trh[a] = 0.5 * (r.w11 * dc[a][0] + 2.0 * r.w12 * dc[a][1] +
                                  r.w22 * dc[a][2]);
            }
            r.box = no_skip ? full_box() : gauss_pixel_box(gc, jac);
            dg[i] = r;
        }
    }
    if (__ballot(bad != 0) != 0ull) {
        // out of range at the trial point: calc_fdiff's LOWVAL vector
        if (lane == 0) {
            for (int k = 0; k < LM_NSUM - 1; k++) out[k] = 0.0;
            out[LM_NSUM - 1] = INFINITY;
            if (status) status[s] = NGMIX_ERR_G_RANGE;
        }
        return;
    }
    __syncthreads();

    // ---- pixel pass: 8x8 tiles, value + 5 derivatives per pixel in registers
    const FexpCoef K = load_fexp_coef(c_fexp_coef_lm);
    const int lrow = lane / TILE_W, lcol = lane % TILE_W;
    const int ntx = (ncol + TILE_W - 1) / TILE_W;
    const int nty = (nrow + TILE_H - 1) / TILE_H;
    const double *sval = val + st.pix_off;
    const double *sierr = ierr + st.pix_off;
    const double iflux = 1.0 / flux;

    double acc[LM_NSUM];
#pragma unroll
    for (int k = 0; k < LM_NSUM; k++) acc[k] = 0.0;

    auto load_tile = [&](int ty, int tx, double &pv, double &pe) {
        const int row = ty * TILE_H + lrow, col = tx * TILE_W + lcol;
        pv = 0.0;
        pe = 0.0;
        if (ty < nty && row < nrow && col < ncol) {
            pv = sval[row * ncol + col];
            pe = sierr[row * ncol + col];
        }
    };

    int ty = 0, tx = 0;
    double nval, nierr;
    load_tile(ty, tx, nval, nierr);
    while (ty < nty) {
        const double pval = nval, pierr = nierr;
        int ty2 = ty, tx2 = tx + 1;
        if (tx2 == ntx) {
            tx2 = 0;
            ty2++;
        }
        load_tile(ty2, tx2, nval, nierr);

        const int r0 = ty * TILE_H, c0 = tx * TILE_W;
        const double rowd = (double)(r0 + lrow) - jac.row0;
        const double cold = (double)(c0 + lcol) - jac.col0;
        const double v = fma(jac.dvdrow, rowd, jac.dvdcol * cold);
        const double u = fma(jac.dudrow, rowd, jac.dudcol * cold);
        double o0 = 0.0, o1 = 0.0, o2 = 0.0, o3 = 0.0, o4 = 0.0, o5 = 0.0;

        for (int gb = 0; gb < G; gb += WAVE) {
            // lane g tests gaussian gb+g's chi2<25 box against this tile
            const int gi = gb + lane < G ? gb + lane : gb;
            const PixBox box = dg[gi].box;
            const bool hit = (gb + lane < G) & (r0 <= box.rmax) &
                             (r0 + TILE_H - 1 >= box.rmin) & (c0 <= box.cmax) &
                             (c0 + TILE_W - 1 >= box.cmin);
            unsigned long long gmask = __ballot(hit);
            while (gmask) {
                const int g = gb + __builtin_ctzll(gmask);
                gmask &= gmask - 1ull;
                const DerivGauss &D = dg[g];
                const double dv = v - D.row, du = u - D.col;
                const double qv = fma(D.w11, dv, D.w12 * du);
                const double qu = fma(D.w12, dv, D.w22 * du);
                const double y = 0.5 * fma(dv, qv, du * qu);  // chi2 / 2
                // derivs_nb.py:104-105: chi2 >= 25 or chi2 < 0 -> skip
                if (y < 12.5 && y >= 0.0) {
                    double e = D.pa * fexp_neg_fused(y, sh.tabr, K);
                    double ec = e;
                    if (y > 10.0) {
                        // W and W - 2 W' of the apodisation (fastexp_nb.py:97-135)
                        const double au = (12.5 - y) * 0.4;
                        const double aq = fma(au, fma(au, K.w6, K.wm15), K.w10);
                        const double w = (au * au) * (au * aq);
                        const double umu = au * (1.0 - au);
                        // -2 W' = 60 * 0.2 * umu^2
                        ec = e * fma(12.0 * umu, umu, w);
                        e *= w;
                    }
                    o0 += e;
                    o1 = fma(ec, qv, o1);
                    o2 = fma(ec, qu, o2);
                    const double qvv = qv * qv, qvu = qv * qu, quu = qu * qu;
                    const double q0 = fma(qvv, D.d[0][0], fma(qvu, D.d[0][1], quu * D.d[0][2]));
                    const double q1 = fma(qvv, D.d[1][0], fma(qvu, D.d[1][1], quu * D.d[1][2]));
                    const double q2 = fma(qvv, D.d[2][0], fma(qvu, D.d[2][1], quu * D.d[2][2]));
                    o3 = fma(ec, q0, fma(-e, D.trh[0], o3));
                    o4 = fma(ec, q1, fma(-e, D.trh[1], o4));
                    o5 = fma(ec, q2, fma(-e, D.trh[2], o5));
                }
            }
        }

        // residual and jacobian row of this pixel (results.py:556-563); lanes
        // outside the stamp and zero-weight pixels have ierr == 0
        if (!masked || pierr > 0.0) {
            const double f = (o0 - pval) * pierr;
            double J[6];
            J[0] = o1 * pierr;
            J[1] = o2 * pierr;
            J[2] = o3 * pierr;
            J[3] = o4 * pierr;
            J[4] = o5 * pierr;
            J[5] = o0 * (pierr * iflux);
            int k = 0;
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int b = a; b < 6; b++) {
                    acc[k] = fma(J[a], J[b], acc[k]);
                    k++;
                }
#pragma unroll
            for (int a = 0; a < 6; a++) acc[21 + a] = fma(J[a], f, acc[21 + a]);
            acc[27] = fma(f, f, acc[27]);
        }
        ty = ty2;
        tx = tx2;
    }

    // ---- 28 sums over the wave: DPP inside rows of 16, then 4 partials in LDS
#pragma unroll
    for (int k = 0; k < LM_NSUM; k++) {
        const double r = row16_sum(acc[k]);
        if ((lane & 15) == 15) sh.red[k * 4 + (lane >> 4)] = r;
    }
    __syncthreads();
    if (lane < LM_NSUM) {
        const double *r = sh.red + lane * 4;
        out[lane] = ((r[0] + r[1]) + r[2]) + r[3];
    }
    if (lane == 0 && status) status[s] = NGMIX_OK;
}

// One thread per object: gather its stamps' sums into the (nloc-1+nband)-
// parameter normal equations and advance the LM state.  A stamp's sums are
// [J^T J upper triangle | J^T f | f.f] over its nloc LOCAL parameters: the
// shared shape parameters followed by the flux of the stamp's band.
// Copy the live part of a state record: the scalars, the first n entries of
// each per-parameter array and the leading n x n block of R.  The record is
// sized for NGMIX_LM_NPMAX = 14 parameters (2.9 kB); a six-parameter fit uses a
// fifth of it, and lm_advance_kernel moves every record in and out once per
// round.  (Entries beyond n keep the zeros lm_init wrote.)
__device__ __forceinline__ void lm_state_copy_live(lm_state &d, const lm_state &g)
{
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
    d.pad_ = g.pad_;
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
        for (int k = 0; k < n; k++) d.R[j * LM_NPMAX + k] = g.R[j * LM_NPMAX + k];
    }
}

__global__ __launch_bounds__(BLOCK) void lm_advance_kernel(
    lm_state *states, int64_t nobj, const int64_t *__restrict__ obj_start,
    const int32_t *__restrict__ stamp_band, const double *__restrict__ sums, int nloc,
    const double *__restrict__ obj_sums, int32_t *nactive)
{
    const int64_t o = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
    if (o >= nobj) return;
    if (states[o].phase == LM_PHASE_DONE) return;
    lm_state s;
    lm_state_copy_live(s, states[o]);
    const int ntri = nloc * (nloc + 1) / 2, nsum = ntri + nloc + 1;
    double A[LM_NPMAX * LM_NPMAX], g[LM_NPMAX];
    for (int i = 0; i < s.n; i++) {
        for (int j = 0; j < s.n; j++) A[i * LM_NPMAX + j] = 0.0;
        g[i] = 0.0;
    }
    double ff = 0.0;
    const int64_t s0 = obj_start ? obj_start[o] : o;
    const int64_t s1 = obj_start ? obj_start[o + 1] : o + 1;
    for (int64_t st = s0; st < s1; st++) {
        const double *v = sums + st * nsum;
        const int band = stamp_band ? stamp_band[st] : 0;
        int k = 0;
        for (int a = 0; a < nloc; a++) {
            const int ga = a < nloc - 1 ? a : nloc - 1 + band;
            for (int b = a; b < nloc; b++) {
                const int gb = b < nloc - 1 ? b : nloc - 1 + band;
                A[ga * LM_NPMAX + gb] += v[k];
                if (ga != gb) A[gb * LM_NPMAX + ga] += v[k];
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
                A[a * LM_NPMAX + b] += v[k];
                if (a != b) A[b * LM_NPMAX + a] += v[k];
                k++;
            }
            g[a] += v[nt + a];
        }
        ff += v[nt + n];
    }
    lmcore::lm_advance(s, ff, g, A);
    lm_state_copy_live(states[o], s);
    if (s.phase != LM_PHASE_DONE && nactive) atomicAdd(nactive, 1);
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

template <int NLOC>
__global__ __launch_bounds__(WAVE) void lm_eval_fd_kernel(
    const ngmix_stamp *__restrict__ stamps, const double *__restrict__ val,
    const double *__restrict__ ierr, const ngmix_jacobian *__restrict__ jacs,
    int model, int ng0, const lm_state *__restrict__ states,
    const int32_t *__restrict__ stamp_obj, const int32_t *__restrict__ stamp_band,
    const ngmix_gauss2d *__restrict__ psf, int npsf, double *__restrict__ sums,
    int32_t *__restrict__ status, int no_skip)
{
    constexpr int NTRI = NLOC * (NLOC + 1) / 2;
    constexpr int NSUM = NTRI + NLOC + 1;
    constexpr int NSETS = NLOC + 1;
    static_assert(NLOC + 1 <= 16, "the normal equations are one 16 x 16 MFMA tile");
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    __shared__ double tabr[16];
    // one tile's rows [J_0 .. J_{NLOC-1}, f, 0 ..] per pixel, 17 doubles apart
    constexpr int JSTRIDE = 17;
    __shared__ double jbuf[WAVE * JSTRIDE];

    const int s = blockIdx.x;
    const int lane = threadIdx.x;
    const int obj = stamp_obj ? stamp_obj[s] : s;
    const lm_state &state = states[obj];
    if (state.phase == LM_PHASE_DONE) return;
    const bool want_jac = state.phase != LM_PHASE_TRIAL;
    const int nsets = want_jac ? NSETS : 1;
    const int band = stamp_band ? stamp_band[s] : 0;
    const ngmix_stamp st = stamps[s];
    const ngmix_jacobian jac = jacs[s];
    const int nrow = st.nrow, ncol = st.ncol;
    const double area = jac.scale * jac.scale;
    const bool izw = (st.flags & NGMIX_STAMP_IGNORE_ZERO_WEIGHT) != 0;
    const bool masked = izw && st.npix_kept != nrow * ncol;
    double *out = sums + (size_t)s * NSUM;
    const int npsf1 = npsf > 0 ? npsf : 1;
    const int G = ng0 * npsf1;
    FdGauss *ev = (FdGauss *)dyn;                    // [NSETS][G]
    PixBox *boxes = (PixBox *)(ev + NSETS * G);      // [G], from the base set

    // local parameters and the fdjac2 points the state prepared (xstep /
    // hstep: the step is taken in leastsqbound's internal parameters)
    double p0[NLOC], ps[NLOC], ih[NLOC];
#pragma unroll
    for (int k = 0; k < NLOC; k++) {
        const int gk = k < NLOC - 1 ? k : NLOC - 1 + band;
        p0[k] = state.xt[gk];
        ps[k] = state.xstep[gk];
        ih[k] = 1.0 / state.hstep[gk];
    }
    double rowcen = 0.0, colcen = 0.0, ipsum = 1.0;
    const ngmix_gauss2d *q = psf ? psf + (size_t)s * npsf : nullptr;
    int bad = 0;
    if (npsf > 0) {
        double psum;
        if (gmix_cen(q, npsf, rowcen, colcen, psum) != NGMIX_OK) bad = 1;
        else ipsum = 1.0 / psum;
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
        if (k == 0) boxes[i] = no_skip ? full_box() : gauss_pixel_box(gc, jac);
    }
    if (__ballot(bad != 0) != 0ull) {
        // out of range at (or one step from) the trial point: LOWVAL residuals
        if (lane == 0) {
            for (int k = 0; k < NSUM - 1; k++) out[k] = 0.0;
            out[NSUM - 1] = INFINITY;
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

    // J^T J, J^T f and f.f are X^T X for X = [J | f] (pixels x 16): a sum of
    // outer products over the pixels -- v_mfma_f64_16x16x4_f64, four pixels
    // per instruction, the whole 16 x 16 result in four accumulator registers
    // per lane (row (lane >> 4) + 4 r, column lane & 15).  With the sums in
    // per-lane VALU accumulators (NSUM of them) the ten-parameter kernel held
    // 236 VGPRs = one wave per SIMD, and twelve parameters did not fit at all.
    typedef double double4_t __attribute__((ext_vector_type(4)));
    double4_t M = {0.0, 0.0, 0.0, 0.0};
    double ff = 0.0;   // trial evaluations want only |f|^2
    {
        double *jb = jbuf + lane * JSTRIDE;
#pragma unroll
        for (int j = 0; j < JSTRIDE; j++) jb[j] = 0.0;
    }

    auto load_tile = [&](int ty, int tx, double &pv, double &pe) {
        const int row = ty * TILE_H + lrow, col = tx * TILE_W + lcol;
        pv = 0.0;
        pe = 0.0;
        if (ty < nty && row < nrow && col < ncol) {
            pv = sval[row * ncol + col];
            pe = sierr[row * ncol + col];
        }
    };

    int ty = 0, tx = 0;
    double nval, nierr;
    load_tile(ty, tx, nval, nierr);
    while (ty < nty) {
        const double pval = nval, pierr = nierr;
        int ty2 = ty, tx2 = tx + 1;
        if (tx2 == ntx) {
            tx2 = 0;
            ty2++;
        }
        load_tile(ty2, tx2, nval, nierr);

        const int r0 = ty * TILE_H, c0 = tx * TILE_W;
        const double rowd = (double)(r0 + lrow) - jac.row0;
        const double cold = (double)(c0 + lcol) - jac.col0;
        const double v = fma(jac.dvdrow, rowd, jac.dvdcol * cold);
        const double u = fma(jac.dudrow, rowd, jac.dudcol * cold);
        double m[NSETS];
#pragma unroll
        for (int k = 0; k < NSETS; k++) m[k] = 0.0;

        for (int gb = 0; gb < G; gb += WAVE) {
            const int gi = gb + lane < G ? gb + lane : gb;
            const PixBox box = boxes[gi];
            const bool hit = (gb + lane < G) & (r0 <= box.rmax) &
                             (r0 + TILE_H - 1 >= box.rmin) & (c0 <= box.cmax) &
                             (c0 + TILE_W - 1 >= box.cmin);
            unsigned long long gmask = __ballot(hit);
            while (gmask) {
                const int g = gb + __builtin_ctzll(gmask);
                gmask &= gmask - 1ull;
#pragma unroll
                for (int k = 0; k < NSETS; k++) {
                    if (k >= nsets) break;
                    const FdGauss &E = ev[k * G + g];
                    const double dv = v - E.row, du = u - E.col;
                    const double y = fma(E.a, dv * dv, fma(E.b, du * du, E.c * (dv * du)));
                    if (y < 12.5 && y >= 0.0) {
                        double e = fexp_neg_fused(y, tabr, K);
                        if (y > 10.0) {
                            const double au = (12.5 - y) * 0.4;
                            const double aq = fma(au, fma(au, K.w6, K.wm15), K.w10);
                            e *= (au * au) * (au * aq);
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
            for (int j = 0; j < NLOC; j++)
                jb[j] = live ? (m[j + 1] - m[0]) * (pierr * ih[j]) : 0.0;
            jb[NLOC] = f;
            __syncthreads();   // one wave: orders the LDS traffic, no s_barrier
            // A[i][k] = B[k][i] = X[pixel 4 t + k][i]: lane (i, k) reads one double
            const double *src = jbuf + (lane >> 4) * JSTRIDE + (lane & 15);
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const double x = src[4 * t * JSTRIDE];
                M = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, M, 0, 0, 0);
            }
            __syncthreads();
        }
        ty = ty2;
        tx = tx2;
    }

    if (!want_jac) {
        const double tot = wave_total(ff);
        // (the jacobian slots are not read for a trial evaluation)
        if (lane == 0) out[NSUM - 1] = tot;
    } else {
        const int col = lane & 15;
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
__global__ __launch_bounds__(BLOCK) void lm_finalize_kernel(
    const lm_state *__restrict__ states, int64_t nobj,
    const int64_t *__restrict__ npix_obj, const double *__restrict__ ff_extra,
    double pdef, double cdef, double *rec)
{
    const int64_t o = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
    if (o >= nobj) return;
    const lm_state &s = states[o];
    const int n = s.n;
    double *r = rec + o * (4 + 2 * (int64_t)n + 2 * (int64_t)n * n);
    double *pars = r + 4, *perr = pars + n, *cov0 = perr + n, *cov = cov0 + n * n;
    int flags = 0;
    const int ier = s.info;
    const long long dof = (long long)npix_obj[o] - n;
    for (int i = 0; i < n; i++) {
        pars[i] = s.x[i];
        perr[i] = cdef;
    }
    for (int i = 0; i < n * n; i++) cov0[i] = cov[i] = cdef;
    if (ier == 0) {
        flags |= NGMIX_FLAG_LM_FUNC_NOTFINITE;
        for (int i = 0; i < n; i++) pars[i] = pdef;
    } else if (ier > 4) {
        flags |= 1 << (ier - 5);
        for (int i = 0; i < n; i++) pars[i] = pdef;
    } else {
        // R^-1 (upper triangular), in the pivoted order
        double X[LM_NPMAX * LM_NPMAX];
        bool singular = false;
        for (int j = 0; j < n; j++) {
            const double d = s.R[j * LM_NPMAX + j];
            if (d == 0.0 || !(fabs(d) < INFINITY)) singular = true;
        }
        if (!singular) {
            for (int i = 0; i < LM_NPMAX * LM_NPMAX; i++) X[i] = 0.0;
            for (int j = 0; j < n; j++) {
                X[j * LM_NPMAX + j] = 1.0 / s.R[j * LM_NPMAX + j];
                for (int i = j - 1; i >= 0; i--) {
                    double acc = 0.0;
                    for (int k = i + 1; k <= j; k++)
                        acc += s.R[i * LM_NPMAX + k] * X[k * LM_NPMAX + j];
                    X[i * LM_NPMAX + j] = -acc / s.R[i * LM_NPMAX + i];
                }
            }
            for (int a = 0; a < n; a++)
                for (int b = 0; b < n; b++) {
                    double acc = 0.0;
                    for (int k = (a > b ? a : b); k < n; k++)
                        acc += X[a * LM_NPMAX + k] * X[b * LM_NPMAX + k];
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
            for (int i = 0; i < n * n; i++) cov0[i] = cdef;
        } else if (dof == 0) {
            flags |= NGMIX_FLAG_ZERO_DOF;
        } else {
            const double s_sq =
                (s.fnorm * s.fnorm - (ff_extra ? ff_extra[o] : 0.0)) / (double)dof;
            bool finite = true;
            for (int i = 0; i < n * n; i++) {
                cov[i] = cov0[i] * s_sq;
                if (!(fabs(cov[i]) < INFINITY)) finite = false;
            }
            int cflags = 0;
            if (!finite) {
                cflags |= NGMIX_FLAG_EIG_NOTFINITE;
            } else {
                // a negative eigenvalue <=> a negative pivot of LDL^T (inertia)
                double A[LM_NPMAX * LM_NPMAX];
                for (int a = 0; a < n; a++)
                    for (int b = 0; b < n; b++) A[a * LM_NPMAX + b] = cov[a * n + b];
                bool neg = false, negdiag = false;
                for (int a = 0; a < n; a++)
                    if (cov[a * n + a] < 0.0) negdiag = true;
                for (int k = 0; k < n; k++) {
                    const double d = A[k * LM_NPMAX + k];
                    if (d < 0.0) neg = true;
                    const double safe = d != 0.0 ? d : 1.0;
                    for (int a = k + 1; a < n; a++) {
                        const double c = A[a * LM_NPMAX + k] / safe;
                        for (int b = k + 1; b < n; b++)
                            A[a * LM_NPMAX + b] -= c * A[k * LM_NPMAX + b];
                    }
                }
                if (neg) cflags |= NGMIX_FLAG_LM_NEG_COV_EIG;
                if (negdiag) cflags |= NGMIX_FLAG_LM_NEG_COV_DIAG;
            }
            flags |= cflags;
            if (cflags == 0)
                for (int a = 0; a < n; a++) perr[a] = sqrt(cov[a * n + a]);
        }
    }
    r[0] = (double)flags;
    r[1] = ier == 0 ? -1.0 : (double)s.nfev;
    r[2] = (double)ier;
    r[3] = (double)dof;
}

__global__ __launch_bounds__(BLOCK) void lm_prior_sums_kernel(
    const lm_state *__restrict__ states, int64_t nobj, ngmix_simple_sep_prior P,
    double step_rel, double *__restrict__ obj_sums)
{
    const int64_t o = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
    if (o >= nobj) return;
    const lm_state &s = states[o];
    const int n = s.n;
    double out[NGMIX_LM_NSUMS(LM_NPMAX)];
    if (s.phase == LM_PHASE_DONE) return;
    lmcore::simple_sep_normal_sums(P, s, step_rel, out);
    double *dst = obj_sums + o * (int64_t)(n * (n + 1) / 2 + n + 1);
    for (int i = 0; i < n * (n + 1) / 2 + n + 1; i++) dst[i] = out[i];
}

int launch_lm_prior_sums(const lm_state *states, int64_t nobj,
                         const ngmix_simple_sep_prior *prior, double step_rel,
                         double *obj_sums, hipStream_t s)
{
    if (nobj <= 0) return NGMIX_OK;
    if (!prior || prior->nband < 1 || prior->nband > NGMIX_PRIOR_MAXBAND)
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

__global__ __launch_bounds__(BLOCK) void lm_init_kernel(lm_state *states, int64_t nobj,
                                                        const double *__restrict__ x0,
                                                        LmInitPars P)
{
    const int64_t o = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
    if (o >= nobj) return;
    lm_state s;
    lmcore::lm_init(s, P.n, x0 + o * P.n, P.ftol, P.xtol, P.gtol, P.maxfev, P.factor,
                    P.mode, P.has_bounds ? P.lo : nullptr, P.has_bounds ? P.hi : nullptr);
    states[o] = s;
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
    P.has_bounds = (lo != nullptr || hi != nullptr) ? 1 : 0;
    for (int j = 0; j < LM_NPMAX; j++) {
        P.lo[j] = (lo && j < npars) ? lo[j] : -INFINITY;
        P.hi[j] = (hi && j < npars) ? hi[j] : INFINITY;
    }
    hipLaunchKernelGGL(lm_init_kernel, dim3((unsigned)((nobj + BLOCK - 1) / BLOCK)),
                       dim3(BLOCK), 0, s, states, nobj, x0, P);
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

int launch_lm_finalize(const lm_state *states, int64_t nobj, const int64_t *npix_obj,
                       const double *ff_extra, double pdef, double cdef, double *rec,
                       hipStream_t s)
{
    if (nobj <= 0) return NGMIX_OK;
    hipLaunchKernelGGL(lm_finalize_kernel, dim3((unsigned)((nobj + BLOCK - 1) / BLOCK)),
                       dim3(BLOCK), 0, s, states, nobj, npix_obj, ff_extra, pdef, cdef, rec);
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
                   hipStream_t s)
{
    if (b->nstamps <= 0) return NGMIX_OK;
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
        const size_t lds = (size_t)G * sizeof(DerivGauss);
        if (lds > 96 * 1024) {
            set_last_error_msg("lm_eval: too many composed gaussians for LDS");
            return NGMIX_ERR_BAD_ARG;
        }
        if (lds > 48 * 1024)
            NGMIX_HIP_CHECK(hipFuncSetAttribute(
                (const void *)lm_eval_kernel,
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(lm_eval_kernel, grid, block, lds, s, b->stamps, b->val,
                           b->ierr, b->jac, model, ng0, states, stamp_obj, stamp_band,
                           psf, npsf, sums, status, no_skip);
        NGMIX_HIP_CHECK(hipGetLastError());
        return NGMIX_OK;
    }
    const size_t lds = (size_t)(nloc + 1) * G * sizeof(FdGauss) + (size_t)G * sizeof(PixBox);
    if (lds > 128 * 1024) {
        set_last_error_msg("lm_eval: too many composed gaussians for LDS");
        return NGMIX_ERR_BAD_ARG;
    }
#define NGMIX_FD_LAUNCH(N)                                                              \
    do {                                                                                \
        if (lds > 48 * 1024)                                                            \
            NGMIX_HIP_CHECK(hipFuncSetAttribute(                                        \
                (const void *)lm_eval_fd_kernel<N>,                                     \
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                 \
        hipLaunchKernelGGL(lm_eval_fd_kernel<N>, grid, block, lds, s, b->stamps, b->val, \
                           b->ierr, b->jac, model, ng0, states, stamp_obj, stamp_band,  \
                           psf, npsf, sums, status, no_skip);                           \
    } while (0)
    if (nloc == 6) NGMIX_FD_LAUNCH(6);
    else if (nloc == 7) NGMIX_FD_LAUNCH(7);
    else if (nloc == 8) NGMIX_FD_LAUNCH(8);
    else if (nloc <= 10) NGMIX_FD_LAUNCH(10);
    else if (nloc <= 12) NGMIX_FD_LAUNCH(12);
    else NGMIX_FD_LAUNCH(14);
#undef NGMIX_FD_LAUNCH
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

int launch_lm_advance(lm_state *states, int64_t nobj, const int64_t *obj_start,
                      const int32_t *stamp_band, const double *sums, int nloc,
                      const double *obj_sums, int32_t *nactive, hipStream_t s)
{
    if (nobj <= 0) return NGMIX_OK;
    if (nloc < 2 || nloc > LM_NPMAX) return NGMIX_ERR_BAD_ARG;
    if (nactive) NGMIX_HIP_CHECK(hipMemsetAsync(nactive, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL(lm_advance_kernel, dim3((unsigned)((nobj + BLOCK - 1) / BLOCK)),
                       dim3(BLOCK), 0, s, states, nobj, obj_start, stamp_band, sums, nloc,
                       obj_sums, nactive);
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

}  // namespace ngmix
