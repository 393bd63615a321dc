// em_wave_impl.hpp -- the fused EM kernels: 1, 2 or 4 waves per stamp with the
// run kind, the object-gaussian count and a one-gaussian psf as compile-time
// parameters.  Included by em_wave.hip (1 .. 3 object gaussians, 72 kernel
// instantiations) and em_wave_hi.hip (4 .. 6), two translation units so that
// the two halves compile side by side.
// Reference: ngmix/em/em_nb.py (em_run and its fixcen / fixcov / fluxonly
// variants); the generic reference-order kernel is in em.hip.
#pragma once
#include <stdio.h>

#include <type_traits>

#include "em_common.hpp"
#include "launch_iter.hpp"

namespace ngmix {

static __constant__ double c_exp_table_e[16] = NGMIX_EXP_TABLE;

// ===========================================================================
// One WAVE per stamp (stamps of <= 16*64 pixels, <= 3 object gaussians).
//
// The 40-500 iterations of a stamp are a serial chain of {pixel pass, 6*ng+2
// sums, O(ng) scalar M-step}; a 256-thread work-group idles three waves (and
// pays three barriers) during the scalar part of every iteration.  Here a
// stamp is one wave: v, u, val of its <= 16 pixels per lane stay in
// registers, there are no barriers, and the SIMD is shared by independent
// stamps.  The pixel pass is written for instruction count (every VALU
// instruction costs one issue slot on CDNA, fp64 included): FMA contraction,
// chi2/2 form with the magic-number fexp cell index, reciprocals by
// v_rcp_f64 + two Newton steps instead of IEEE division sequences (the
// reference divides per pixel three times: em_nb.py:240,255-256), and the
// sums are reduced through a transposed LDS tile instead of 6*ng+2 shuffle
// trees.  Results agree with the reference to rounding (the tests ask
// 1e-10 on the mixtures and the exact iteration count).
// ===========================================================================

struct EmConvF {
    double row, col;
    double a, b, c;   // y = chi2/2 = a v2 + b u2 + c uv
    double pa;        // pnorm * area
    double K;         // logtau - 0.5*logdet
    double pad;
};
static_assert(sizeof(EmConvF) == 64, "EmConvF");

// 1/x by v_rcp_f64 (4.6e-8 relative on gfx950, tools/microbench/rcp_accuracy.hip)
// and ONE Newton step: 2.1e-15, ten roundings' worth on a per-pixel factor of
// a self-correcting fixed-point iteration whose results are compared at 1e-10
// (x == 0 -> inf, as the reference's division)
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// values reduced per pass of em_group_reduce: the transposed tile of a several-
// wave group is NVC x (NT + 2) doubles and must fit the 64 kB of static LDS
// next to the rest (a 256-thread group with the 38 sums of six gaussians would
// need 78 kB: two passes of 19)
template <int NT, int NV>
constexpr int em_reduce_chunk()
{
    constexpr int cap = 40960 / (8 * (NT + 2));
    if (NT == WAVE || NV <= cap) return NV;
    // equal chunks
    constexpr int passes = (NV + cap - 1) / cap;
    return (NV + passes - 1) / passes;
}

template <int NT, int NV>
struct EmWaveShared {
    static constexpr int NVC = em_reduce_chunk<NT, NV>();
    double tab[16];    // exp(i), i = -15..0  (zero-weight fill, apodised evaluator)
    double tabr[16];   // exp(-n), n = 0..15  (fused evaluator)
    double red[NVC * (NT + 2)];
    double part[NVC * 16];
    double tot[NV];
    double sky, frac_diff, elogL_last, p_last, npix;
    double psf_irr, psf_irc, psf_icc, psf_row, psf_col, psf_ipsum;
    int numiter, stop, status, pad;
};

// sky + model at one zero-weight pixel (fill_zero_weight_pixels, em_nb.py:
// 1297-1315: the apodised evaluator).  Rare: kept out of line.
static __device__ __noinline__ double em_fill_value(const ngmix_gauss2d *conv, int nconv,
                                             double v, double u, double area,
                                             double sky, const double *tab)
{
    double m = 0.0;
    for (int i = 0; i < nconv; i++)
        m += gauss_eval_fast(make_eval(conv[i]), v, u, area, tab);
    return sky + m;
}

// The sum of each of NV per-thread values over an NT-thread work-group (NT = 64,
// 128, 256), left in tot[k]: the values go through a transposed LDS tile, NV*S
// threads each add one segment of one row, NV threads fold the S partials.
// Fixed order.  One wave: wave_reduce_lds (no barriers).  Several waves and
// many values: em_reduce_chunk() values per pass through the same tile.
template <int NT, int NV>
__device__ __forceinline__ void em_group_reduce_pass(const double (&acc)[NV], double *red,
                                                     double *part, double *tot)
{
    constexpr int STRIDE = NT + 2;
    // segments per value: the largest power of two with NV*S <= NT, at most 16
    constexpr int Q = NT / NV;
    constexpr int S = Q >= 16 ? 16 : Q >= 8 ? 8 : Q >= 4 ? 4 : 2;
    static_assert(NV * S <= NT && NT % S == 0, "em_group_reduce sizing");
    constexpr int SEGLEN = NT / S;
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < NV; k++) red[k * STRIDE + tid] = acc[k];
    __syncthreads();
    if (tid < NV * S) {
        const int k = tid / S, j = tid - k * S;
        const double *row = red + k * STRIDE + j * SEGLEN;
        double s = 0.0;
#pragma unroll 8
        for (int i = 0; i < SEGLEN; i++) s += row[i];
        part[k * 16 + j] = s;
    }
    __syncthreads();
    if (tid < NV) {
        const double *r = part + tid * 16;
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < S; j++) s += r[j];
        tot[tid] = s;
    }
    __syncthreads();
}

template <int NT, int NV>
__device__ __forceinline__ void em_group_reduce(const double (&acc)[NV], double *red,
                                                double *part, double *tot)
{
    if constexpr (NT == WAVE) {
        wave_reduce_lds<NV>(acc, red, tot);
    } else {
        constexpr int NVC = em_reduce_chunk<NT, NV>();
        if constexpr (NVC == NV) {
            em_group_reduce_pass<NT, NV>(acc, red, part, tot);
        } else {
#pragma unroll
            for (int c0 = 0; c0 < NV; c0 += NVC) {
                // (the last pass is padded with zeros: one instantiation)
                double sub[NVC];
#pragma unroll
                for (int k = 0; k < NVC; k++) sub[k] = c0 + k < NV ? acc[c0 + k] : 0.0;
                double *dst = tot + c0;
                // tot[] has NV entries: the padded tail of the last pass lands
                // in part[] only
                constexpr int STRIDE = NT + 2;
                constexpr int Q = NT / NVC;
                constexpr int S = Q >= 16 ? 16 : Q >= 8 ? 8 : Q >= 4 ? 4 : 2;
                static_assert(NVC * S <= NT && NT % S == 0, "em_group_reduce sizing");
                constexpr int SEGLEN = NT / S;
                const int tid = threadIdx.x;
#pragma unroll
                for (int k = 0; k < NVC; k++) red[k * STRIDE + tid] = sub[k];
                __syncthreads();
                if (tid < NVC * S) {
                    const int k = tid / S, j = tid - k * S;
                    const double *row = red + k * STRIDE + j * SEGLEN;
                    double s = 0.0;
#pragma unroll 8
                    for (int i = 0; i < SEGLEN; i++) s += row[i];
                    part[k * 16 + j] = s;
                }
                __syncthreads();
                if (tid < NVC && c0 + tid < NV) {
                    const double *r = part + tid * 16;
                    double s = 0.0;
#pragma unroll
                    for (int j = 0; j < S; j++) s += r[j];
                    dst[tid] = s;
                }
                __syncthreads();
            }
        }
    }
}

// KIND, the number of object gaussians NG and the psf's gaussian count NPSF (1:
// the shared-X form below; 3: the general form unrolled; 0: a run-time count) are
// compile-time: the pixel pass is straight-line code.  NT threads per stamp
// (1, 2 or 4 waves), PPT pixels per thread in registers.
template <int NT, int PPT, int KIND, int NG, int NPSF>
__device__ __forceinline__ void em_wave_body(
    const GridSrc &src, const ngmix_em_conf conf, double sky_in,
    ngmix_gauss2d *gmix_io, ngmix_gauss2d *psf_io, int npsf_rt,
    ngmix_gauss2d *conv_io, int fill_zero_weight, double *out3, int32_t *status,
    EmWaveShared<NT, 6 * NG + 2> &sh, char *dyn, const double *coef)
{
    constexpr int NV = 6 * NG + 2;
    constexpr int kind = KIND;
    constexpr int ngauss = NG;
    constexpr bool NPSF1 = NPSF == 1;
    const int npsf = NPSF > 0 ? NPSF : npsf_rt;
    const int lane = threadIdx.x;
    const int nconv = ngauss * npsf;
    ngmix_gauss2d *gmix = (ngmix_gauss2d *)dyn;
    ngmix_gauss2d *psf = gmix + ngauss;
    ngmix_gauss2d *conv = psf + npsf;
    EmConvF *ce = (EmConvF *)(conv + nconv);

    // ---- the stamp, once from HBM, into registers
    const int n = src.count();
    double pv[PPT], pu[PPT], pval[PPT];
    unsigned long long kept = 0ull, zw = 0ull;   // one bit per slot (PPT <= 64)
#pragma unroll
    for (int k = 0; k < PPT; k++) {
        const int p = lane + k * NT;
        pv[k] = pu[k] = pval[k] = 0.0;
        if (p < n) {
            double a, ierr;
            if (src.load(p, pv[k], pu[k], a, pval[k], ierr)) {
                kept |= 1ull << k;
                if (ierr <= 0.0) zw |= 1ull << k;
            }
        }
    }
    const double area = src.area;
    double cnt[1] = {(double)__popcll(kept)};
    // every slot of every lane holds a listed pixel (a 32 x 32 stamp fills a
    // wave's 16 slots exactly; config 4, most psf stamps): the one-gaussian
    // one-wave kernels then run a pixel pass without per-slot EXEC masks --
    // left in, the compiler keeps one 64-bit mask per slot in SGPRs across the
    // iteration loop, 32 of the wave's 102, and spills what no longer fits to
    // VGPR lanes (v_writelane / v_readlane in the VALU stream)
    constexpr bool kFullForm = NT == WAVE && NPSF == 1 && NG == 1 && PPT < 64;
    bool full = false;
    if constexpr (kFullForm)
        // (bit 1 of fill_zero_weight: the launcher's A/B and test knob
        // NGMIX_EM_NO_FULL -- the general form on full waves too)
        full = __ballot(kept == ((1ull << PPT) - 1ull)) == ~0ull && n == PPT * NT &&
               !(fill_zero_weight & 2);

    if (lane < 16) {
        sh.tab[lane] = c_exp_table_e[lane];
        sh.tabr[lane] = c_exp_table_e[15 - lane];
    }
    {
        double pad[6 * NG + 2];
#pragma unroll
        for (int k = 0; k < 6 * NG + 2; k++) pad[k] = k == 0 ? cnt[0] : 0.0;
        em_group_reduce<NT, 6 * NG + 2>(pad, sh.red, sh.part, sh.tot);
    }
    // (kept in LDS: lane 0 reads it once per iteration, a register pair held
    // across the iteration loop for that is one the pixel pass cannot have)
    if (lane == 0) sh.npix = sh.tot[0];
    __syncthreads();
    for (int i = lane; i < ngauss; i += NT) gmix[i] = gmix_io[i];
    for (int i = lane; i < npsf; i += NT) psf[i] = psf_io[i];
    for (int i = lane; i < nconv; i += NT) conv[i] = conv_io[i];
    __syncthreads();

    constexpr bool use_cen = (kind == NGMIX_EM_FULL || kind == NGMIX_EM_FIXCOV);
    constexpr bool use_cov = (kind == NGMIX_EM_FULL || kind == NGMIX_EM_FIXCEN);
    constexpr bool use_logl = (kind != NGMIX_EM_FLUXONLY);

    if (lane == 0) {
        sh.status = NGMIX_OK;
        sh.stop = 0;
        sh.sky = sky_in;
        sh.frac_diff = 0.0;  // unbound in the reference until first assigned
        sh.elogL_last = -9999.9e9;
        sh.numiter = 0;
        // gmix_set_norms(gmix_conv), em_nb.py:59
        for (int i = 0; i < nconv; i++) {
            const int st = gauss_set_norm(conv[i]);
            if (st) {
                sh.status = st;
                sh.stop = 1;
                break;
            }
        }
        double pl = 0.0;
        for (int i = 0; i < ngauss; i++) pl += gmix[i].p;
        sh.p_last = pl;  // em_nb.py:1059 (fluxonly)
        // the psf is constant over the iterations: its moments and centre
        // (gmix_get_moms / gmix_get_cen in every gmix_set_from_sums of the
        // reference) give the same values every time
        sh.psf_irr = sh.psf_irc = sh.psf_icc = 0.0;
        if (!sh.stop && use_cov) {
            const int st = gmix_moms(psf, npsf, sh.psf_irr, sh.psf_irc, sh.psf_icc);
            if (st) {
                // raised inside the first M-step in the reference: numiter 0
                sh.pad = st;
                sh.stop = 2;
            }
        }
        if (sh.stop == 0 || sh.stop == 2) {
            double psum;
            const int st = gmix_cen(psf, npsf, sh.psf_row, sh.psf_col, psum);
            if (st) {
                sh.pad = st;
                sh.stop = 2;
            } else {
                sh.psf_ipsum = 1.0 / psum;
            }
        }
    }
    __syncthreads();
    // stop == 2: the first M-step would raise; the reference gets there only
    // if maxiter > 0 and the first E-step does not raise first
    const FexpCoef K = load_fexp_coef(coef);

    for (int it = 0; it < conf.maxiter && sh.stop != 1; it++) {
        // elogL is read by the convergence test alone, and that test starts at
        // numiter = it + 1 >= miniter with the previous iteration's value
        // (em_nb.py:95-104): before it + 2 >= miniter no one can see it, so
        // set_logtau_logdet's two logs per gaussian are not computed (38 of
        // the 40 iterations of a fit that stops at the default miniter) -- the
        // pixel pass then sums a logL nobody reads (K = 0): branching around
        // its four instructions per pixel cost a wave of occupancy (172
        // registers; measured 13.4 against 10.7 ms).  Wave-uniform.
        // (Also measured and dropped: skipping the pixels outside every
        // component's cut, with the sky sum taken from sum val - sum w: on
        // 32 x 32 stamps nearly every row holds pixels inside the 5 sigma cut,
        // and the extra state cost more than the skipped rows saved.)
        const bool need_logl = use_logl && it + 2 >= conf.miniter;
        // set_logtau_logdet + the evaluation view of the convolved mixture
        for (int i = lane; i < nconv; i += NT) {
            const ngmix_gauss2d g = conv[i];
            EmConvF c;
            c.row = g.row;
            c.col = g.col;
            c.a = 0.5 * g.dcc;
            c.b = 0.5 * g.drr;
            c.c = -g.drc;
            c.pa = g.pnorm * area;
            c.K = need_logl ? log_fast(g.p) - 0.5 * log_fast(g.det) : 0.0;
            c.pad = 0.0;
            ce[i] = c;
        }
        __syncthreads();
        const double sky = sh.sky;

        // fill_zero_weight_pixels overwrites val of the zero-weight pixels
        // with sky + model, as the reference does in its pixel copy
        if ((fill_zero_weight & 1) && zw != 0ull) {
            // (opaque to the optimiser: sixteen loop-invariant EXEC masks would
            // otherwise live in SGPRs across the whole iteration loop)
            unsigned zlo = (unsigned)zw, zhi = (unsigned)(zw >> 32);
            asm volatile("" : "+v"(zlo), "+v"(zhi));
            const unsigned long long zwl = ((unsigned long long)zhi << 32) | zlo;
#pragma unroll
            for (int k = 0; k < PPT; k++)
                if (zwl & (1ull << k))
                    pval[k] = em_fill_value(conv, nconv, pv[k], pu[k], area, sky, sh.tab);
        }

        double acc[NV];
#pragma unroll
        for (int k = 0; k < NV; k++) acc[k] = 0.0;
        bool bad = false;

        // The pixel pass, with and without the logL term: the one-wave kernels
        // with a compile-time psf count and <= 3 object gaussians (config 4,
        // every psf fit, galaxy fits under a 1- or 3-gaussian psf) run the
        // second form while elogL cannot be seen (above) -- five instructions
        // per pixel of 43 for one gaussian.  Two copies of the loop, one
        // register budget.
        auto pixel_pass = [&](auto with_logl) {
            constexpr bool LOGL = decltype(with_logl)::value;
            // (kernels with the full-wave form: the slot masks of this general
            // form are made per pass, not kept in SGPRs across the iterations)
            unsigned klo = (unsigned)kept, khi = (unsigned)(kept >> 32);
            if constexpr (kFullForm) asm volatile("" : "+v"(klo), "+v"(khi));
            const unsigned long long keptl = ((unsigned long long)khi << 32) | klo;
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            if (!(keptl & (1ull << k))) continue;
            const double v = pv[k], u = pu[k];
            const double val_pix = pval[k];
            if constexpr (NPSF1) {
                // One psf gaussian: object gaussian ii has ONE convolved
                // component, so its scratch sums (em_nb.py:229-237) are all
                // products with the same value, tvsum = v gi, tv2sum = vdiff^2 gi
                // ..., and the sums (em_nb.py:262-279) are X (gi factor) for
                // X = 1, v, u, udiff^2, udiff vdiff, vdiff^2: one product
                // w = gi factor and six fused accumulates per gaussian instead
                // of five scratch products and six accumulates.  (The
                // reference rounds (X gi) factor; this rounds X (gi factor).)
                double gi[NG], v2s[NG], uvs[NG], u2s[NG];
                double gsum = 0.0, logL = 0.0;
#pragma unroll
                for (int ii = 0; ii < NG; ii++) {
                    const EmConvF c = ce[ii];
                    const double vdiff = v - c.row;
                    const double udiff = u - c.col;
                    u2s[ii] = udiff * udiff;
                    v2s[ii] = vdiff * vdiff;
                    uvs[ii] = udiff * vdiff;
                    const double y = fma(c.a, v2s[ii], fma(c.b, u2s[ii], c.c * uvs[ii]));
                    double val = 0.0;
                    // hard cut: chi2 < 25 and chi2 >= 0 (em_nb.py:222-227)
                    if (y < 12.5 && y >= 0.0) {
                        val = c.pa * fexp_neg_fused(y, sh.tabr, K);
                        // (one component in all: val (K - y) / gsum with
                        // gsum == val is K - y to three roundings -- no product,
                        // no reciprocal)
                        if (LOGL && NG == 1) logL = val != 0.0 ? c.K - y : 0.0;
                        if (LOGL && NG > 1) logL = fma(val, c.K - y, logL);
                    }
                    gi[ii] = val;
                    gsum = ii == 0 ? val : gsum + val;
                }
                const double gtot = gsum + sky;
                if (gtot == 0.0) {
                    bad = true;  // GMixRangeError('gtot == 0')
                    continue;
                }
                if (LOGL && NG > 1) logL = (gsum == 0.0) ? 0.0 : logL * fast_rcp(gsum);
                const double factor = val_pix * fast_rcp(gtot);
                if (LOGL) acc[6 * NG + 0] += logL;
                acc[6 * NG + 1] = fma(sky, factor, acc[6 * NG + 1]);
#pragma unroll
                for (int ii = 0; ii < NG; ii++) {
                    const double w = gi[ii] * factor;
                    acc[6 * ii + 0] += w;
                    if (use_cen) {
                        acc[6 * ii + 2] = fma(u, w, acc[6 * ii + 2]);
                        acc[6 * ii + 1] = fma(v, w, acc[6 * ii + 1]);
                    }
                    if (use_cov) {
                        acc[6 * ii + 3] = fma(u2s[ii], w, acc[6 * ii + 3]);
                        acc[6 * ii + 4] = fma(uvs[ii], w, acc[6 * ii + 4]);
                        acc[6 * ii + 5] = fma(v2s[ii], w, acc[6 * ii + 5]);
                    }
                }
                continue;
            }
            double gi[NG], tv[NG], tu[NG], tv2[NG], tuv[NG], tu2[NG];
            double gsum = 0.0, logL = 0.0;
#pragma unroll
            for (int ii = 0; ii < NG; ii++) {
                gi[ii] = tv[ii] = tu[ii] = tv2[ii] = tuv[ii] = tu2[ii] = 0.0;
                {
#pragma unroll
                    for (int ip = 0; ip < (NPSF > 0 ? NPSF : npsf); ip++) {
                        const int i = ii * npsf + ip;
                        const EmConvF c = ce[i];
                        const double vdiff = v - c.row;
                        const double udiff = u - c.col;
                        const double u2 = udiff * udiff;
                        const double v2 = vdiff * vdiff;
                        const double uv = udiff * vdiff;
                        const double y = fma(c.a, v2, fma(c.b, u2, c.c * uv));
                        // hard cut: chi2 < 25 and chi2 >= 0 (em_nb.py:222-227)
                        if (y < 12.5 && y >= 0.0) {
                            const double val = c.pa * fexp_neg_fused(y, sh.tabr, K);
                            gi[ii] += val;
                            gsum += val;
                            if (use_cen) {
                                tv[ii] = fma(v, val, tv[ii]);
                                tu[ii] = fma(u, val, tu[ii]);
                            }
                            if (use_cov) {
                                tv2[ii] = fma(v2, val, tv2[ii]);
                                tuv[ii] = fma(uv, val, tuv[ii]);
                                tu2[ii] = fma(u2, val, tu2[ii]);
                            }
                            if (LOGL) logL = fma(val, c.K - y, logL);
                        }
                    }
                }
            }
            if (LOGL) logL = (gsum == 0.0) ? 0.0 : logL * fast_rcp(gsum);
            const double gtot = gsum + sky;
            if (gtot == 0.0) {
                bad = true;  // GMixRangeError('gtot == 0')
                continue;
            }
            const double factor = val_pix * fast_rcp(gtot);
            if (LOGL) acc[6 * NG + 0] += logL;
            acc[6 * NG + 1] = fma(sky, factor, acc[6 * NG + 1]);
#pragma unroll
            for (int ii = 0; ii < NG; ii++) {
                {
                    acc[6 * ii + 0] = fma(gi[ii], factor, acc[6 * ii + 0]);
                    if (use_cen) {
                        acc[6 * ii + 2] = fma(tu[ii], factor, acc[6 * ii + 2]);
                        acc[6 * ii + 1] = fma(tv[ii], factor, acc[6 * ii + 1]);
                    }
                    if (use_cov) {
                        acc[6 * ii + 3] = fma(tu2[ii], factor, acc[6 * ii + 3]);
                        acc[6 * ii + 4] = fma(tuv[ii], factor, acc[6 * ii + 4]);
                        acc[6 * ii + 5] = fma(tv2[ii], factor, acc[6 * ii + 5]);
                    }
                }
            }
        }

        };
        // The same arithmetic for a full wave with ONE component in all: no
        // per-slot masks, and the component's seven numbers and the sky in
        // SGPRs (a VALU instruction reads one scalar operand for free) instead
        // of four LDS reads and their wait at the head of every pixel.
        auto pixel_pass_full1 = [&](auto with_logl) {
            constexpr bool LOGL = decltype(with_logl)::value;
            const EmConvF c0 = ce[0];
            const double crow = uniform_f64(c0.row), ccol = uniform_f64(c0.col);
            const double ca = uniform_f64(c0.a), cb = uniform_f64(c0.b);
            const double cc = uniform_f64(c0.c), cpa = uniform_f64(c0.pa);
            const double cK = LOGL ? uniform_f64(c0.K) : 0.0;
            const double ssky = uniform_f64(sky);
#pragma unroll
            for (int k = 0; k < PPT; k++) {
                const double v = pv[k], u = pu[k];
                const double vdiff = sub_sgpr_from(crow, v);
                const double udiff = sub_sgpr_from(ccol, u);
                const double u2 = udiff * udiff;
                const double v2 = vdiff * vdiff;
                const double uv = udiff * vdiff;
                const double y = fma_sgpr(ca, v2, fma_sgpr(cb, u2, mul_sgpr(cc, uv)));
                double val = 0.0, logL = 0.0;
                // hard cut 0 <= chi2 < 25 (em_nb.py:222-227) as one unsigned
                // compare on the high word of y = chi2 / 2: negative, nan and
                // inf fail it as they fail the reference's two compares
                if ((unsigned)__double2hiint(y) < 0x40290000u) {
                    val = mul_sgpr(cpa, fexp_neg_fused(y, sh.tabr, K));
                    if (LOGL) logL = val != 0.0 ? sgpr_minus(cK, y) : 0.0;
                }
                double gtot;
                asm("v_add_f64 %0, %1, %2" : "=v"(gtot) : "s"(ssky), "v"(val));
                if (gtot == 0.0) {
                    bad = true;  // GMixRangeError('gtot == 0')
                    continue;
                }
                const double factor = pval[k] * fast_rcp(gtot);
                if (LOGL) acc[6] += logL;
                acc[7] = fma_sgpr(ssky, factor, acc[7]);
                const double w = val * factor;
                acc[0] += w;
                if (use_cen) {
                    acc[2] = fma(u, w, acc[2]);
                    acc[1] = fma(v, w, acc[1]);
                }
                if (use_cov) {
                    acc[3] = fma(u2, w, acc[3]);
                    acc[4] = fma(uv, w, acc[4]);
                    acc[5] = fma(v2, w, acc[5]);
                }
                // (ends the pixel here: without the branches that used to sit
                // between the slots the scheduler runs several pixels side by
                // side and the kernel leaves its three-waves-per-SIMD band)
                asm volatile("" : "+v"(acc[0]), "+v"(acc[7]));
            }
        };
        constexpr bool kTwoForms = NT == WAVE && use_logl && NG <= 3 && (NPSF1 || NPSF == 3);
        if constexpr (kFullForm) {
            if (full) {
                if (use_logl && need_logl) pixel_pass_full1(std::true_type{});
                else pixel_pass_full1(std::false_type{});
            } else if (kTwoForms && !need_logl) {
                pixel_pass(std::integral_constant<bool, false>{});
            } else {
                pixel_pass(std::integral_constant<bool, use_logl>{});
            }
        } else {
            if (kTwoForms && !need_logl) pixel_pass(std::integral_constant<bool, false>{});
            else pixel_pass(std::integral_constant<bool, use_logl>{});
        }

        const bool anybad = __syncthreads_or(bad ? 1 : 0) != 0;
        em_group_reduce<NT, NV>(acc, sh.red, sh.part, sh.tot);

        if (lane == 0) {
            if (anybad) {
                sh.status = NGMIX_ERR_GTOT_ZERO;
                sh.stop = 1;
            } else if (sh.stop == 2) {
                sh.status = sh.pad;  // the psf has no flux: the M-step raises
                sh.stop = 1;
            } else {
                const double elogL = sh.tot[6 * NG + 0];
                const double skysum = sh.tot[6 * NG + 1];
                const int st = em_mstep_psf(kind, gmix, ngauss, psf, npsf, conv, sh.tot,
                                            sh.psf_irr, sh.psf_irc, sh.psf_icc,
                                            sh.psf_row, sh.psf_col, sh.psf_ipsum);
                if (st) {
                    sh.status = st;
                    sh.stop = 1;
                } else {
                    if (conf.vary_sky) sh.sky = skysum / sh.npix;
                    sh.numiter = it + 1;
                    if (kind == NGMIX_EM_FLUXONLY) {
                        double psum = 0.0;
                        for (int i = 0; i < ngauss; i++) psum += gmix[i].p;
                        if (sh.numiter >= conf.miniter) {
                            if (sh.p_last == 0.0) {
                                sh.status = NGMIX_ERR_ZERO_DIV;
                                sh.stop = 1;
                            } else {
                                sh.frac_diff = fabs(psum / sh.p_last - 1);
                                if (sh.frac_diff < conf.tol) sh.stop = 1;
                            }
                        }
                        sh.p_last = psum;
                    } else {
                        if (sh.numiter >= conf.miniter) {
                            if (elogL == 0.0) {
                                sh.status = NGMIX_ERR_ELOGL_ZERO;
                                sh.stop = 1;
                            } else {
                                sh.frac_diff = fabs((elogL - sh.elogL_last) / elogL);
                                if (sh.frac_diff < conf.tol) sh.stop = 1;
                            }
                        }
                        sh.elogL_last = elogL;
                    }
                }
            }
        }
        __syncthreads();
    }

    // write back.  The reference zeroes norm_set of the pre-psf mixture on a
    // normal exit (em_nb.py:125); on an exception it has no chance to.
    if (sh.status == NGMIX_OK)
        for (int i = lane; i < ngauss; i += NT) gmix[i].norm_set = 0;
    __syncthreads();
    for (int i = lane; i < ngauss; i += NT) gmix_io[i] = gmix[i];
    for (int i = lane; i < nconv; i += NT) conv_io[i] = conv[i];
    if (lane == 0) {
        out3[0] = (double)sh.numiter;
        out3[1] = sh.frac_diff;
        out3[2] = sh.sky;
        if (status) *status = sh.status;
    }
}

static __constant__ double c_fexp_coef_e[12] = NGMIX_FEXP_COEF;

// (the one-gaussian one-wave kernels are held to the three waves per SIMD
// they were tuned at: 168 registers; every other form takes what it needs)
template <int NT, int PPT, int KIND, int NG, int NPSF>
__global__ __launch_bounds__(NT)
__attribute__((amdgpu_waves_per_eu((NT == WAVE && NPSF == 1 && NG == 1) ? 3 : 1)))
void em_wave_kernel(
    ngmix_em_conf conf, const ngmix_stamp *stamps, const double *val,
    const double *ierr, const ngmix_jacobian *jacs, ngmix_gauss2d *gmix,
    ngmix_gauss2d *gmix_psf, int npsf, ngmix_gauss2d *gmix_conv,
    const double *sky_in, int fill_zero_weight, double *out, int32_t *status)
{
    __shared__ EmWaveShared<NT, 6 * NG + 2> sh;
    extern __shared__ __attribute__((aligned(16))) char dyn[];
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
    em_wave_body<NT, PPT, KIND, NG, NPSF>(
        src, conf, sky_in[s], gmix + (size_t)s * NG, gmix_psf + (size_t)s * npsf, npsf,
        gmix_conv + (size_t)s * NG * npsf, fill_zero_weight, out + 3 * (size_t)s,
        status ? status + s : nullptr, sh, dyn, c_fexp_coef_e);
}

template <int NT, int PPT, int KIND, int NG>
static void em_wave_launch_nt(const ngmix_em_conf *conf, const ngmix_batch *b,
                              ngmix_gauss2d *gmix, ngmix_gauss2d *psf, int npsf,
                              ngmix_gauss2d *conv, const double *sky_in, int fzw,
                              double *out, int32_t *status, hipStream_t s)
{
    const size_t nconv = (size_t)NG * npsf;
    const size_t lds = (NG + npsf + nconv) * sizeof(ngmix_gauss2d) +
                       nconv * sizeof(EmConvF) + 64;
    {
        char name[80];
        snprintf(name, sizeof(name), "em_wave_kernel<%d, %d, %d, %d, %d>", NT, PPT, KIND, NG,
                 npsf == 1 ? 1 : (npsf == 3 && NG <= 3 && NT == WAVE) ? 3 : 0);
        census(name);
    }
    static const bool no_full = getenv("NGMIX_EM_NO_FULL") != nullptr;
    if (no_full) fzw |= 2;
    if (npsf == 1)
        hipLaunchKernelGGL((em_wave_kernel<NT, PPT, KIND, NG, 1>),
                           dim3((unsigned)b->nstamps), dim3(NT), lds, s, *conf,
                           b->stamps, b->val, b->ierr, b->jac, gmix, psf, npsf, conv,
                           sky_in, fzw, out, status);
    else if (npsf == 3 && NG <= 3 && NT == WAVE)
        // (the 'turb' / coellip-3 psf of a galaxy fit on stamps of <= 32 x 32:
        // the component loop unrolled)
        hipLaunchKernelGGL((em_wave_kernel<NT, PPT, KIND, NG, (NG <= 3 && NT == WAVE) ? 3 : 0>),
                           dim3((unsigned)b->nstamps), dim3(NT), lds, s, *conf,
                           b->stamps, b->val, b->ierr, b->jac, gmix, psf, npsf, conv,
                           sky_in, fzw, out, status);
    else
        hipLaunchKernelGGL((em_wave_kernel<NT, PPT, KIND, NG, 0>),
                           dim3((unsigned)b->nstamps), dim3(NT), lds, s, *conf,
                           b->stamps, b->val, b->ierr, b->jac, gmix, psf, npsf, conv,
                           sky_in, fzw, out, status);
}

// one wave up to 32x32 pixels, two up to 45x45, four up to 64x64
template <int KIND, int NG>
static void em_wave_launch(const ngmix_em_conf *conf, const ngmix_batch *b,
                           ngmix_gauss2d *gmix, ngmix_gauss2d *psf, int npsf,
                           ngmix_gauss2d *conv, const double *sky_in, int fzw,
                           double *out, int32_t *status, hipStream_t s)
{
    const int np = b->max_npix;
    if (np <= 16 * WAVE)
        em_wave_launch_nt<WAVE, 16, KIND, NG>(conf, b, gmix, psf, npsf, conv, sky_in, fzw,
                                              out, status, s);
    else if (KIND == NGMIX_EM_FULL && np > 16 * 2 * WAVE && np <= 18 * 2 * WAVE)
        // 48 x 48 = 18 x 128: two waves, 18 register slots per lane, like the
        // two-wave kernel of <= 2048 pixels (9.6 ms per 50k 45 x 45 stamps) instead
        // of four waves with 16 (20 ms per 50k).  One wave with 36 slots was
        // measured too: 18 ms -- its 270-340 registers leave one wave per SIMD to
        // a serial per-pixel chain.  (The full run only: every (kind, ngauss,
        // psf) combination is a kernel of its own.)
        em_wave_launch_nt<2 * WAVE, (KIND == NGMIX_EM_FULL ? 18 : 16), KIND, NG>(
            conf, b, gmix, psf, npsf, conv, sky_in, fzw, out, status, s);
    else if (np <= 16 * 2 * WAVE)
        em_wave_launch_nt<2 * WAVE, 16, KIND, NG>(conf, b, gmix, psf, npsf, conv, sky_in,
                                                  fzw, out, status, s);
    else if constexpr (NG <= 6)
        // (four to six object gaussians: the 26 .. 38 sums go through the
        // reduction tile in two passes, em_reduce_chunk)
        em_wave_launch_nt<BLOCK, 16, KIND, NG>(conf, b, gmix, psf, npsf, conv, sky_in,
                                               fzw, out, status, s);
    // (seven and eight object gaussians: one or two waves, em_wave_8.hip; em.hip
    // sends larger stamps to the generic kernel)
}

}  // namespace ngmix
