// em.hip -- expectation-maximisation of a gaussian mixture on one stamp.
//
// Reference: ngmix/em/em_nb.py -- em_run (:15-127), em_run_fixcen (:357-469),
// em_run_fixcov (:702-816), em_run_fluxonly (:1005-1106) with their
// do_scratch_sums* / do_sums* / gmix_set_from_sums* / clear_sums* helpers,
// set_logtau_logdet (:658-675), gmix_get_moms (:1260-1294) and
// fill_zero_weight_pixels (:1297-1315).
//
// One work-group per stamp; the stamp is read from HBM once and stays in
// registers for the 40-500 iterations (compute-bound).  The
// E-step is pixel-parallel with a fixed-order reduction of the 6*ngauss+2
// sums; the M-step (O(ngauss) scalar work) runs on thread 0 out of LDS.
// EM evaluates gaussians with a HARD chi2<25 cut (em_nb.py:222-227), not the
// apodized evaluator; only the zero-weight fill uses the apodized one.
#include <stdio.h>

#include "em_common.hpp"
#include "launch_iter.hpp"

namespace ngmix {

__constant__ double c_exp_table_e[16] = NGMIX_EXP_TABLE;

// per-kind layout of the reference's per-gaussian sums record, in doubles
// (ngmix/em/em.py:451-521); -1 = field absent
struct EmLayout {
    int stride;
    int pnew, vsum, usum, u2sum, uvsum, v2sum;
};

__device__ __forceinline__ EmLayout em_layout(int kind)
{
    switch (kind) {
    case NGMIX_EM_FULL: return {14, 8, 9, 10, 11, 12, 13};
    case NGMIX_EM_FIXCEN: return {10, 6, -1, -1, 7, 8, 9};
    case NGMIX_EM_FIXCOV: return {8, 5, 6, 7, -1, -1, -1};
    default: return {2, 1, -1, -1, -1, -1, -1};
    }
}

struct EmConv {
    EvalGauss e;
    double logtau, logdet;
};

struct EmShared {
    double tab[16];
    double red_scratch[NWAVES * 64];
    double red_out[64];
    double sky;
    double frac_diff;
    double elogL_last;
    double p_last;
    int numiter;
    int stop;
    int status;
    int pad;
};

// NG: compile-time bound on the number of object gaussians (register arrays)
template <class Src, int NT, int PPT, int NG>
__device__ __forceinline__ void em_body(const Src &src, int kind,
                                        const ngmix_em_conf conf, double sky_in,
                                        ngmix_gauss2d *gmix_io, int ngauss,
                                        ngmix_gauss2d *psf_io, int npsf,
                                        ngmix_gauss2d *conv_io, double *sums_io,
                                        int fill_zero_weight, double *out3,
                                        int32_t *status, ngmix_pixel *pix_writeback,
                                        EmShared &sh, char *dyn)
{
    const int tid = threadIdx.x;
    const int nconv = ngauss * npsf;
    // dynamic LDS: gmix[ngauss] psf[npsf] conv[nconv] (records), ce[nconv],
    // tot[6*ngauss]
    ngmix_gauss2d *gmix = (ngmix_gauss2d *)dyn;
    ngmix_gauss2d *psf = gmix + ngauss;
    ngmix_gauss2d *conv = psf + npsf;
    EmConv *ce = (EmConv *)(conv + nconv);
    double *tot = (double *)(ce + nconv);

    PixCache<Src, NT, PPT> cache;
    cache.fill(src);
    int my_n = 0;
    cache.for_each(src, [&](double, double, double, double, double, int) { my_n++; });
    double cnt[1] = {(double)my_n};

    if (tid < 16) sh.tab[tid] = c_exp_table_e[tid];
    for (int i = tid; i < ngauss; i += NT) gmix[i] = gmix_io[i];
    for (int i = tid; i < npsf; i += NT) psf[i] = psf_io[i];
    for (int i = tid; i < nconv; i += NT) conv[i] = conv_io[i];
    __syncthreads();
    group_sum<NT, 1>(cnt, sh.red_scratch, sh.red_out);
    const double npix = sh.red_out[0];
    __syncthreads();

    if (tid == 0) {
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
    }
    __syncthreads();

    const bool use_cen = (kind == NGMIX_EM_FULL || kind == NGMIX_EM_FIXCOV);
    const bool use_cov = (kind == NGMIX_EM_FULL || kind == NGMIX_EM_FIXCEN);
    const bool use_logl = (kind != NGMIX_EM_FLUXONLY);

    for (int it = 0; it < conf.maxiter && !sh.stop; it++) {
        // set_logtau_logdet + the evaluation view of the convolved mixture
        for (int i = tid; i < nconv; i += NT) {
            EmConv c;
            c.e = make_eval(conv[i]);
            c.logtau = use_logl ? log(conv[i].p) : 0.0;
            c.logdet = use_logl ? log(conv[i].det) : 0.0;
            ce[i] = c;
        }
        __syncthreads();
        const double sky = sh.sky;

        constexpr int NV = 6 * NG + 2;
        double acc[NV];
#pragma unroll
        for (int k = 0; k < NV; k++) acc[k] = 0.0;
        int bad = 0;

        cache.for_each(src, [&](double v, double u, double area, double pval,
                                double pierr, int) {
            if (fill_zero_weight && pierr <= 0.0) {
                // fill_zero_weight_pixels: apodized evaluator (em_nb.py:1314)
                double m = 0.0;
                for (int i = 0; i < nconv; i++)
                    m += gauss_eval_fast(ce[i].e, v, u, area, sh.tab);
                pval = sky + m;
            }
            double gi[NG], tv[NG], tu[NG], tv2[NG], tuv[NG], tu2[NG];
            double gsum = 0.0, logL = 0.0;
#pragma unroll
            for (int ii = 0; ii < NG; ii++) {
                gi[ii] = tv[ii] = tu[ii] = tv2[ii] = tuv[ii] = tu2[ii] = 0.0;
                if (ii < ngauss) {
                    for (int i = ii * npsf; i < (ii + 1) * npsf; i++) {
                        const EmConv c = ce[i];
                        const double vdiff = v - c.e.row;
                        const double udiff = u - c.e.col;
                        const double u2 = udiff * udiff;
                        const double v2 = vdiff * vdiff;
                        const double uv = udiff * vdiff;
                        const double chi2 = c.e.dcc * v2 + c.e.drr * u2 - c.e.drc2 * uv;
                        const double val = gauss_eval_hardcut(c.e, chi2, area, sh.tab);
                        gi[ii] += val;
                        gsum += val;
                        if (use_cen) {
                            tv[ii] += v * val;
                            tu[ii] += u * val;
                        }
                        if (use_cov) {
                            tv2[ii] += v2 * val;
                            tuv[ii] += uv * val;
                            tu2[ii] += u2 * val;
                        }
                        if (use_logl)
                            logL += val * (c.logtau - 0.5 * c.logdet - 0.5 * chi2);
                    }
                }
            }
            if (use_logl) {
                if (gsum == 0.0) logL = 0.0;
                else logL *= 1.0 / gsum;
            }
            const double gtot = gsum + sky;
            if (gtot == 0.0) {
                bad = 1;  // GMixRangeError('gtot == 0')
                return;
            }
            acc[6 * NG + 0] += logL;
            acc[6 * NG + 1] += sky * pval / gtot;
            const double factor = pval / gtot;
#pragma unroll
            for (int ii = 0; ii < NG; ii++) {
                if (ii < ngauss) {
                    acc[6 * ii + 0] += gi[ii] * factor;
                    if (use_cen) {
                        acc[6 * ii + 2] += tu[ii] * factor;
                        acc[6 * ii + 1] += tv[ii] * factor;
                    }
                    if (use_cov) {
                        acc[6 * ii + 3] += tu2[ii] * factor;
                        acc[6 * ii + 4] += tuv[ii] * factor;
                        acc[6 * ii + 5] += tv2[ii] * factor;
                    }
                }
            }
        });

        const int anybad = __syncthreads_or(bad);
        group_sum<NT, NV>(acc, sh.red_scratch, sh.red_out);

        if (tid == 0) {
            if (anybad) {
                sh.status = NGMIX_ERR_GTOT_ZERO;
                sh.stop = 1;
            } else {
                for (int i = 0; i < 6 * ngauss; i++) tot[i] = sh.red_out[i];
                const double elogL = sh.red_out[6 * NG + 0];
                const double skysum = sh.red_out[6 * NG + 1];
                const int st = em_mstep(kind, gmix, ngauss, psf, npsf, conv, tot);
                if (st) {
                    sh.status = st;
                    sh.stop = 1;
                } else {
                    if (conf.vary_sky) sh.sky = skysum / npix;
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
        for (int i = tid; i < ngauss; i += NT) gmix[i].norm_set = 0;
    __syncthreads();
    for (int i = tid; i < ngauss; i += NT) gmix_io[i] = gmix[i];
    for (int i = tid; i < nconv; i += NT) conv_io[i] = conv[i];
    if (sums_io) {
        const EmLayout L = em_layout(kind);
        for (int i = tid; i < ngauss; i += NT) {
            double *ts = sums_io + (size_t)i * L.stride;
            const double *t = tot + 6 * i;
            ts[L.pnew] = t[0];
            if (L.vsum >= 0) ts[L.vsum] = t[1], ts[L.usum] = t[2];
            if (L.u2sum >= 0) ts[L.u2sum] = t[3], ts[L.uvsum] = t[4], ts[L.v2sum] = t[5];
        }
    }
    if (pix_writeback && fill_zero_weight && sh.status == NGMIX_OK && sh.numiter > 0) {
        // the reference leaves the last iteration's fill in the caller's
        // pixel array; that fill used the mixture and sky at the START of
        // the last iteration, which the M-step has since replaced.  Not
        // reproduced: the array is a private copy in EMFitter.go (em.py:269).
    }
    if (tid == 0) {
        out3[0] = (double)sh.numiter;
        out3[1] = sh.frac_diff;
        out3[2] = sh.sky;
        if (status) *status = sh.status;
    }
}

static size_t em_dyn_lds(int ngauss, int npsf)
{
    const size_t nconv = (size_t)ngauss * npsf;
    return (ngauss + npsf + nconv) * sizeof(ngmix_gauss2d) + nconv * sizeof(EmConv) +
           6 * (size_t)ngauss * 8 + 64;
}

template <int NT, int PPT, int NG>
__global__ __launch_bounds__(NT) void em_grid_kernel(
    int kind, ngmix_em_conf conf, const ngmix_stamp *stamps, const double *val,
    const double *ierr, const ngmix_jacobian *jacs, ngmix_gauss2d *gmix, int ngauss,
    ngmix_gauss2d *gmix_psf, int npsf, ngmix_gauss2d *gmix_conv,
    const double *sky_in, int fill_zero_weight, double *out, int32_t *status)
{
    __shared__ EmShared sh;
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
    em_body<GridSrc, NT, PPT, NG>(src, kind, conf, sky_in[s],
                              gmix + (size_t)s * ngauss, ngauss,
                              gmix_psf + (size_t)s * npsf, npsf,
                              gmix_conv + (size_t)s * ngauss * npsf, nullptr,
                              fill_zero_weight, out + 3 * (size_t)s,
                              status ? status + s : nullptr, nullptr, sh, dyn);
}

template <int NG>
__global__ __launch_bounds__(BLOCK) void em_list_kernel(
    int kind, ngmix_em_conf conf, ngmix_pixel *pixels, int n, double *sums,
    ngmix_gauss2d *gmix, int ngauss, ngmix_gauss2d *gmix_psf, int npsf,
    ngmix_gauss2d *gmix_conv, int fill_zero_weight, double *out3, int32_t *status)
{
    __shared__ EmShared sh;
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    ListSrc src;
    src.pix = pixels;
    src.n = n;
    em_body<ListSrc, BLOCK, 0, NG>(src, kind, conf, conf.sky, gmix, ngauss, gmix_psf, npsf,
                            gmix_conv, sums, fill_zero_weight, out3, status, pixels,
                            sh, dyn);
}

template <int NT, int PPT, int NG>
static void em_grid_launch(int kind, const ngmix_em_conf *conf, const ngmix_batch *b,
                           ngmix_gauss2d *gmix, int ngauss, ngmix_gauss2d *psf,
                           int npsf, ngmix_gauss2d *conv, const double *sky_in,
                           int fzw, double *out, int32_t *status, hipStream_t s)
{
    {
        char name[64];
        snprintf(name, sizeof(name), "em_grid_kernel<%d, %d, %d>", NT, PPT, NG);
        census(name);
    }
    hipLaunchKernelGGL((em_grid_kernel<NT, PPT, NG>), dim3((unsigned)b->nstamps),
                       dim3(NT), em_dyn_lds(ngauss, npsf), s, kind, *conf,
                       b->stamps, b->val, b->ierr, b->jac, gmix, ngauss, psf, npsf,
                       conv, sky_in, fzw, out, status);
}

template <int NT, int PPT>
static int em_grid_dispatch_ng(int kind, const ngmix_em_conf *conf,
                               const ngmix_batch *b, ngmix_gauss2d *gmix, int ngauss,
                               ngmix_gauss2d *psf, int npsf, ngmix_gauss2d *conv,
                               const double *sky_in, int fzw, double *out,
                               int32_t *status, hipStream_t s)
{
#define NGMIX_EM_CASE(N)                                                          \
    em_grid_launch<NT, PPT, N>(kind, conf, b, gmix, ngauss, psf, npsf, conv,    \
                               sky_in, fzw, out, status, s)
    if (ngauss <= 1) NGMIX_EM_CASE(1);
    else if (ngauss <= 2) NGMIX_EM_CASE(2);
    else if (ngauss <= 3) NGMIX_EM_CASE(3);
    else if (ngauss <= 4) NGMIX_EM_CASE(4);
    else if (ngauss <= 6) NGMIX_EM_CASE(6);
    else if (ngauss <= 10) NGMIX_EM_CASE(10);
    else {
        set_last_error_msg("em: more than 10 object gaussians not supported");
        return NGMIX_ERR_BAD_ARG;
    }
#undef NGMIX_EM_CASE
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

int launch_em_grid(int kind, const ngmix_em_conf *conf, const ngmix_batch *b,
                   ngmix_gauss2d *gmix, int ngauss, ngmix_gauss2d *psf, int npsf,
                   ngmix_gauss2d *conv, const double *sky_in, int fzw, double *out,
                   int32_t *status, hipStream_t s)
{
    if (b->nstamps <= 0) return NGMIX_OK;
    if (kind < 0 || kind > 3 || ngauss < 1 || npsf < 1) return NGMIX_ERR_BAD_ARG;
    // one wave per stamp while the stamp fits the registers of one wave (no
    // barriers, no idle waves during the scalar M-step); tuning hook
    // NGMIX_EM_NT=64|256 forces the threads per stamp
    int nt = 0;
    if (const char *e = getenv("NGMIX_EM_NT")) nt = atoi(e);
    const int np = b->max_npix;
    // the fused kernels: <= 6 object gaussians up to 64 x 64 pixels (one, two or
    // four waves per stamp), 7 and 8 up to 2304 pixels (one or two waves)
    // (two waves hold 16 pixels per lane, 18 in the full run only)
    const int np_hi = kind == NGMIX_EM_FULL ? 18 * 2 * WAVE : 16 * 2 * WAVE;
    const bool fused = (np <= 16 * BLOCK && ngauss <= 6) || (np <= np_hi && ngauss <= 8);
    if (nt == 0) nt = fused ? WAVE : BLOCK;
    if (nt == WAVE && fused)
        return launch_em_wave(kind, conf, b, gmix, ngauss, psf, npsf, conv, sky_in, fzw,
                              out, status, s);
    if (np <= 4 * BLOCK)
        return em_grid_dispatch_ng<BLOCK, 4>(kind, conf, b, gmix, ngauss, psf, npsf,
                                             conv, sky_in, fzw, out, status, s);
    if (np <= 9 * BLOCK && ngauss <= 3)
        return em_grid_dispatch_ng<BLOCK, 9>(kind, conf, b, gmix, ngauss, psf, npsf,
                                             conv, sky_in, fzw, out, status, s);
    return em_grid_dispatch_ng<BLOCK, 0>(kind, conf, b, gmix, ngauss, psf, npsf, conv,
                                         sky_in, fzw, out, status, s);
}

int launch_em_list(int kind, const ngmix_em_conf *conf, ngmix_pixel *pixels,
                   int64_t n, double *sums, ngmix_gauss2d *gmix, int ngauss,
                   ngmix_gauss2d *psf, int npsf, ngmix_gauss2d *conv, int fzw,
                   double *out3, int32_t *status, hipStream_t s)
{
    if (kind < 0 || kind > 3 || ngauss < 1 || npsf < 1) return NGMIX_ERR_BAD_ARG;
    const size_t lds = em_dyn_lds(ngauss, npsf);
#define NGMIX_EM_CASE(N)                                                         \
    hipLaunchKernelGGL((em_list_kernel<N>), dim3(1), dim3(BLOCK), lds, s, kind,  \
                       *conf, pixels, (int)n, sums, gmix, ngauss, psf, npsf,     \
                       conv, fzw, out3, status)
    if (ngauss <= 1) NGMIX_EM_CASE(1);
    else if (ngauss <= 2) NGMIX_EM_CASE(2);
    else if (ngauss <= 3) NGMIX_EM_CASE(3);
    else if (ngauss <= 4) NGMIX_EM_CASE(4);
    else if (ngauss <= 6) NGMIX_EM_CASE(6);
    else if (ngauss <= 10) NGMIX_EM_CASE(10);
    else {
        set_last_error_msg("em: more than 10 object gaussians not supported");
        return NGMIX_ERR_BAD_ARG;
    }
#undef NGMIX_EM_CASE
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

}  // namespace ngmix
