// moments.hip -- fixed-weight moment sums and adaptive moments.
//
//   get_weighted_sums / get_higher_order_weighted_sums
//       (reference: ngmix/gmix/gmix_nb.py:681-821)
//   admom + admom_censums + admom_momsums + deweight_moments + clear_result
//       (reference: ngmix/admom/admom_nb.py:13-239)
//
// One work-group per stamp.  Both are compute-bound: the stamp
// is read from HBM once and the iteration runs on chip.
#include "iter_common.hpp"
#include "launch_iter.hpp"

namespace ngmix {

__constant__ double c_exp_table_m[16] = NGMIX_EXP_TABLE;

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
    double red_scratch[NWAVES * 64];
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

template <class Src, int PPT>
__device__ __forceinline__ void admom_body(const Src &src,
                                           const ngmix_admom_conf conf,
                                           ngmix_gauss2d *wt_io,
                                           ngmix_admom_result *res_io,
                                           int32_t *status, AdmomShared &sh)
{
    const int tid = threadIdx.x;
    PixCache<Src, BLOCK, PPT> cache;
    cache.fill(src);

    // the moments pass divides by ierr^2 for every listed pixel
    // (admom_nb.py:146): any zero there is numba's ZeroDivisionError
    int my_last = -1, my_zero = 0;
    cache.for_each(src, [&](double, double, double, double, double ierr, int p) {
        if (ierr * ierr == 0.0) my_zero = 1;
        my_last = p > my_last ? p : my_last;
    });
    const int last_pos = group_max_int<BLOCK>(my_last, sh.iscratch);
    const int has_zero = group_max_int<BLOCK>(my_zero, sh.iscratch);

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
            group_sum<BLOCK, 4>(a, sh.red_scratch, sh.red_out);
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
            group_sum<BLOCK, 9>(a, sh.red_scratch, sh.red_out);
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
        const bool want_cov = sh.res_is_mom != 0 && sh.status == NGMIX_OK;
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
                group_sum<BLOCK, 13>(part, sh.red_scratch, sh.red_out);
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

template <int PPT>
__global__ __launch_bounds__(BLOCK) void admom_grid_kernel(
    ngmix_admom_conf conf, const ngmix_stamp *stamps, const double *val,
    const double *ierr, const ngmix_jacobian *jacs, ngmix_gauss2d *wt,
    ngmix_admom_result *res, int32_t *status)
{
    __shared__ AdmomShared sh;
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
    admom_body<GridSrc, PPT>(src, conf, wt + st.gm_off, res + s,
                             status ? status + s : nullptr, sh);
}

__global__ __launch_bounds__(BLOCK) void admom_list_kernel(
    ngmix_admom_conf conf, const ngmix_pixel *pixels, int n, ngmix_gauss2d *wt,
    ngmix_admom_result *res, int32_t *status)
{
    __shared__ AdmomShared sh;
    ListSrc src;
    src.pix = pixels;
    src.n = n;
    admom_body<ListSrc, 0>(src, conf, wt, res, status, sh);
}

int launch_admom_grid(const ngmix_admom_conf *conf, const ngmix_batch *b,
                      ngmix_gauss2d *wt, ngmix_admom_result *res, int32_t *status,
                      hipStream_t s)
{
    if (b->nstamps <= 0) return NGMIX_OK;
    dim3 grid((unsigned)b->nstamps), block(BLOCK);
    if (b->max_npix <= 4 * BLOCK) {
        hipLaunchKernelGGL(admom_grid_kernel<4>, grid, block, 0, s, *conf,
                           b->stamps, b->val, b->ierr, b->jac, wt, res, status);
    } else if (b->max_npix <= 9 * BLOCK) {
        hipLaunchKernelGGL(admom_grid_kernel<9>, grid, block, 0, s, *conf,
                           b->stamps, b->val, b->ierr, b->jac, wt, res, status);
    } else {
        hipLaunchKernelGGL(admom_grid_kernel<0>, grid, block, 0, s, *conf,
                           b->stamps, b->val, b->ierr, b->jac, wt, res, status);
    }
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
