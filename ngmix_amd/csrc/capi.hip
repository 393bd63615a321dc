// capi.hip -- extern "C" entry points of libngmix_hip.so (include/ngmix_hip.h).
//
// Seam forms take HOST pointers: they stage the arrays through a grow-only
// per-thread device workspace, run the kernel on the null stream and copy the
// results back.  Batch forms take DEVICE pointers and only enqueue work.
#include <stdio.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.hpp"
#include "launch.hpp"
#include "lm_core.hpp"
#include "lm_core_reg.hpp"
#include "launch_iter.hpp"

namespace ngmix {

static thread_local std::string g_last_error;

void set_last_error(const char *what, hipError_t err)
{
    g_last_error = std::string(what) + ": " + hipGetErrorString(err);
}

void set_last_error_msg(const char *msg) { g_last_error = msg; }

static std::mutex g_census_mutex;
static std::map<std::string, long long> g_census;

void census(const char *kernel)
{
    std::lock_guard<std::mutex> lock(g_census_mutex);
    g_census[kernel] += 1;
}

// grow-only device scratch, one set per host thread AND per device: after
// ngmix_set_device(other) on the same thread the seam forms must not reuse
// pointers that belong to the previous device
class Workspace {
public:
    static constexpr int NSLOT = 8;
    static constexpr int NDEV = 16;
    void *get(int slot, size_t nbytes)
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= NDEV) return nullptr;
        void *&ptr = ptr_[dev][slot];
        size_t &cap = cap_[dev][slot];
        if (nbytes == 0) nbytes = 8;
        if (cap < nbytes) {
            if (ptr) (void)hipFree(ptr);
            ptr = nullptr;
            cap = 0;
            size_t want = nbytes + nbytes / 4 + 256;
            if (hipMalloc(&ptr, want) != hipSuccess) {
                ptr = nullptr;
                return nullptr;
            }
            cap = want;
        }
        return ptr;
    }
    ~Workspace()
    {
        // the HIP runtime may already be gone at thread/process exit; leak
    }

private:
    void *ptr_[NDEV][NSLOT] = {{nullptr}};
    size_t cap_[NDEV][NSLOT] = {{0}};
};

static thread_local Workspace g_ws;

static const double h_exp_table[16] = NGMIX_EXP_TABLE;
static const ModelTables h_tables = NGMIX_MODEL_TABLES;

#define WS_GET(var, type, slot, nbytes)                                  \
    type *var = (type *)g_ws.get(slot, nbytes);                          \
    if (!var) {                                                          \
        set_last_error_msg("device workspace allocation failed");        \
        return NGMIX_ERR_HIP;                                            \
    }

static int host_norms_if_needed(ngmix_gauss2d *gm, int64_t ng)
{
    if (ng > 0 && gm[0].norm_set == 0) return ngmix_set_norms(gm, ng);
    return NGMIX_OK;
}

}  // namespace ngmix

using namespace ngmix;

// one host step through the register form of the iteration (lm_core_reg.hpp)
template <int N>
static void lm_advance_host_reg(ngmix_lm_state &st, double ff, const double *g,
                                const double *A)
{
    lmreg::lm_state_n<N> s;
    lmreg::load_state<N>(s, st);
    double gc[N], Ac[N * N];
    for (int i = 0; i < N; i++) {
        gc[i] = g[i];
        for (int j = 0; j < N; j++) Ac[i * N + j] = A[i * NGMIX_LM_NPMAX + j];
    }
    lmreg::lm_advance<N>(s, ff, gc, Ac);
    lmreg::store_state<N>(st, s);
}

extern "C" {

// ------------------------------------------------------------------ runtime

const char *ngmix_version(void) { return "ngmix_amd 0.1.0 (gfx950)"; }

const char *ngmix_last_error(void) { return g_last_error.c_str(); }

int64_t ngmix_abi_sizeof(const char *type_name)
{
    if (!type_name) return -1;
    const std::string t(type_name);
#define NGMIX_SIZEOF_CASE(T) \
    if (t == #T) return (int64_t)sizeof(T)
    NGMIX_SIZEOF_CASE(ngmix_gauss2d);
    NGMIX_SIZEOF_CASE(ngmix_pixel);
    NGMIX_SIZEOF_CASE(ngmix_coord);
    NGMIX_SIZEOF_CASE(ngmix_jacobian);
    NGMIX_SIZEOF_CASE(ngmix_admom_conf);
    NGMIX_SIZEOF_CASE(ngmix_admom_result);
    NGMIX_SIZEOF_CASE(ngmix_em_conf);
    NGMIX_SIZEOF_CASE(ngmix_stamp);
    NGMIX_SIZEOF_CASE(ngmix_batch);
    NGMIX_SIZEOF_CASE(ngmix_lm_state);
    NGMIX_SIZEOF_CASE(ngmix_simple_sep_prior);
    NGMIX_SIZEOF_CASE(ngmix_lm_problem);
#undef NGMIX_SIZEOF_CASE
    return -1;
}

int ngmix_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ngmix_set_device(int device)
{
    NGMIX_HIP_CHECK(hipSetDevice(device));
    return NGMIX_OK;
}

int ngmix_device_malloc(void **ptr, size_t nbytes)
{
    NGMIX_HIP_CHECK(hipMalloc(ptr, nbytes ? nbytes : 8));
    return NGMIX_OK;
}

int ngmix_device_free(void *ptr)
{
    NGMIX_HIP_CHECK(hipFree(ptr));
    return NGMIX_OK;
}

int ngmix_memcpy_h2d(void *dst, const void *src, size_t nbytes, void *stream)
{
    if (nbytes == 0) return NGMIX_OK;
    NGMIX_HIP_CHECK(hipMemcpyAsync(dst, src, nbytes, hipMemcpyHostToDevice,
                                   (hipStream_t)stream));
    return NGMIX_OK;
}

int ngmix_memcpy_d2h(void *dst, const void *src, size_t nbytes, void *stream)
{
    if (nbytes == 0) return NGMIX_OK;
    NGMIX_HIP_CHECK(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToHost,
                                   (hipStream_t)stream));
    NGMIX_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return NGMIX_OK;
}

int ngmix_memset_device(void *dst, int value, size_t nbytes, void *stream)
{
    if (nbytes == 0) return NGMIX_OK;
    NGMIX_HIP_CHECK(hipMemsetAsync(dst, value, nbytes, (hipStream_t)stream));
    return NGMIX_OK;
}

int ngmix_stream_synchronize(void *stream)
{
    NGMIX_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return NGMIX_OK;
}

// ------------------------------------------------ host parameter prep (O(G))

int ngmix_set_norms(ngmix_gauss2d *gmix, int64_t ngauss)
{
    for (int64_t i = 0; i < ngauss; i++) {
        int st = gauss_set_norm(gmix[i]);
        if (st) return st;
    }
    return NGMIX_OK;
}

int ngmix_fill_model(ngmix_gauss2d *gmix, int64_t ngauss, int model,
                     const double *pars, int64_t npars)
{
    (void)npars;
    if (model == NGMIX_MODEL_CM) return NGMIX_ERR_BAD_ARG;
    FillCtx c;
    int st = fill_prepare(h_tables, model, (int)ngauss, pars, nullptr, c);
    if (st) return st;
    for (int i = 0; i < (int)ngauss; i++)
        fill_component(h_tables, c, pars, i, gmix[i]);
    return NGMIX_OK;
}

int ngmix_fill_cm(ngmix_gauss2d *gmix, double fracdev, double TdByTe,
                  double Tfactor, const double *pars)
{
    const double extra[3] = {fracdev, TdByTe, Tfactor};
    FillCtx c;
    int st = fill_prepare(h_tables, NGMIX_MODEL_CM, 16, pars, extra, c);
    if (st) return st;
    for (int i = 0; i < 16; i++) fill_component(h_tables, c, pars, i, gmix[i]);
    return NGMIX_OK;
}

int ngmix_get_cm_Tfactor(double fracdev, double TdByTe, double *Tfactor)
{
    return cm_Tfactor(h_tables, fracdev, TdByTe, *Tfactor);
}

int ngmix_g1g2_to_e1e2(double g1, double g2, double *e1, double *e2)
{
    return g1g2_to_e1e2(g1, g2, *e1, *e2);
}

int ngmix_convolve_fill(ngmix_gauss2d *out, const ngmix_gauss2d *gmix,
                        int64_t ngauss, const ngmix_gauss2d *psf, int64_t npsf)
{
    double rowcen, colcen, psum;
    int st = gmix_cen(psf, (int)npsf, rowcen, colcen, psum);
    if (st) return st;
    const double ipsum = 1.0 / psum;
    int64_t itot = 0;
    for (int64_t io = 0; io < ngauss; io++)
        for (int64_t ip = 0; ip < npsf; ip++)
            convolve_component(gmix[io], psf[ip], rowcen, colcen, ipsum,
                               out[itot++]);
    return NGMIX_OK;
}

void ngmix_jacobian_get_vu(const ngmix_jacobian *jacob, double row, double col,
                           double *v, double *u)
{
    jacobian_vu(*jacob, row, col, *v, *u);
}

int ngmix_jacobian_get_rowcol(const ngmix_jacobian *j, double v, double u,
                              double *row, double *col)
{
    // jacobian_get_rowcol, jacobian_nb.py:19-30
    double rowdiff = j->dudcol * v - j->dvdcol * u;
    double coldiff = -j->dudrow * v + j->dvdrow * u;
    if (j->det == 0.0) return NGMIX_ERR_ZERO_DIV;
    *row = j->row0 + rowdiff / j->det;
    *col = j->col0 + coldiff / j->det;
    return NGMIX_OK;
}

// ------------------------------------------------------- seam pixel loops

int ngmix_fill_pixels(ngmix_pixel *pixels, int64_t npixels, const double *image,
                      const double *weight, int64_t nrow, int64_t ncol,
                      const ngmix_jacobian *jacob, int ignore_zero_weight)
{
    const int64_t npix = nrow * ncol;
    if (npix <= 0 || npix > (1ll << 30)) return NGMIX_ERR_BAD_ARG;
    WS_GET(d_img, double, 0, npix * 8);
    WS_GET(d_wt, double, 1, npix * 8);
    WS_GET(d_pix, ngmix_pixel, 2, (size_t)(npixels > 0 ? npixels : 1) * sizeof(ngmix_pixel));
    WS_GET(d_cnt, int, 3, 8);
    NGMIX_HIP_CHECK(hipMemcpy(d_img, image, npix * 8, hipMemcpyHostToDevice));
    NGMIX_HIP_CHECK(hipMemcpy(d_wt, weight, npix * 8, hipMemcpyHostToDevice));
    int st = launch_fill_pixels(d_pix, npixels, d_img, d_wt, (int)nrow, (int)ncol,
                                *jacob, ignore_zero_weight, d_cnt, nullptr);
    if (st) return st;
    int count = 0;
    NGMIX_HIP_CHECK(hipMemcpy(&count, d_cnt, sizeof(int), hipMemcpyDeviceToHost));
    const int64_t ncopy = count < npixels ? count : npixels;
    if (ncopy > 0)
        NGMIX_HIP_CHECK(hipMemcpy(pixels, d_pix, ncopy * sizeof(ngmix_pixel),
                                  hipMemcpyDeviceToHost));
    if (count != npixels) return NGMIX_ERR_PIXELS_NOT_FILLED;
    return NGMIX_OK;
}

int ngmix_fill_coords(ngmix_coord *coords, int64_t nrow, int64_t ncol,
                      const ngmix_jacobian *jacob)
{
    const int64_t npix = nrow * ncol;
    if (npix <= 0 || npix > (1ll << 30)) return NGMIX_ERR_BAD_ARG;
    WS_GET(d_c, ngmix_coord, 0, npix * sizeof(ngmix_coord));
    int st = launch_fill_coords(d_c, (int)nrow, (int)ncol, *jacob, nullptr);
    if (st) return st;
    NGMIX_HIP_CHECK(hipMemcpy(coords, d_c, npix * sizeof(ngmix_coord),
                              hipMemcpyDeviceToHost));
    return NGMIX_OK;
}

int ngmix_render(ngmix_gauss2d *gmix, int64_t ngauss, const ngmix_coord *coords,
                 int64_t ncoords, double *image, int fast_exp)
{
    int st = host_norms_if_needed(gmix, ngauss);
    if (st) return st;
    if (ncoords <= 0) return NGMIX_OK;
    WS_GET(d_gm, ngmix_gauss2d, 0, ngauss * sizeof(ngmix_gauss2d));
    WS_GET(d_c, ngmix_coord, 1, ncoords * sizeof(ngmix_coord));
    WS_GET(d_im, double, 2, ncoords * 8);
    NGMIX_HIP_CHECK(hipMemcpy(d_gm, gmix, ngauss * sizeof(ngmix_gauss2d), hipMemcpyHostToDevice));
    NGMIX_HIP_CHECK(hipMemcpy(d_c, coords, ncoords * sizeof(ngmix_coord), hipMemcpyHostToDevice));
    NGMIX_HIP_CHECK(hipMemcpy(d_im, image, ncoords * 8, hipMemcpyHostToDevice));
    st = launch_render_list(d_gm, (int)ngauss, d_c, ncoords, d_im, fast_exp, nullptr);
    if (st) return st;
    NGMIX_HIP_CHECK(hipMemcpy(image, d_im, ncoords * 8, hipMemcpyDeviceToHost));
    return NGMIX_OK;
}

static int seam_pixpass(int op, ngmix_gauss2d *gmix, int64_t ngauss,
                        const ngmix_pixel *pixels, int64_t npix, double *fdiff,
                        int64_t start, double out4[4])
{
    int st = host_norms_if_needed(gmix, ngauss);
    if (st) return st;
    out4[0] = out4[1] = out4[2] = out4[3] = 0.0;
    if (npix <= 0) return NGMIX_OK;
    WS_GET(d_gm, ngmix_gauss2d, 0, ngauss * sizeof(ngmix_gauss2d));
    WS_GET(d_px, ngmix_pixel, 1, npix * sizeof(ngmix_pixel));
    WS_GET(d_part, double, 2, list_partial_doubles() * 8);
    double *d_fd = nullptr;
    if (op == 1) {
        d_fd = (double *)g_ws.get(3, npix * 8);
        if (!d_fd) return NGMIX_ERR_HIP;
    }
    NGMIX_HIP_CHECK(hipMemcpy(d_gm, gmix, ngauss * sizeof(ngmix_gauss2d), hipMemcpyHostToDevice));
    NGMIX_HIP_CHECK(hipMemcpy(d_px, pixels, npix * sizeof(ngmix_pixel), hipMemcpyHostToDevice));
    st = launch_pixpass_list(op, d_gm, (int)ngauss, d_px, npix, d_fd, 0, d_part, nullptr);
    if (st) return st;
    if (op == 1) {
        NGMIX_HIP_CHECK(hipMemcpy(fdiff + start, d_fd, npix * 8, hipMemcpyDeviceToHost));
    } else {
        NGMIX_HIP_CHECK(hipMemcpy(out4, d_part + (list_partial_doubles() - 4), 32,
                                  hipMemcpyDeviceToHost));
    }
    return NGMIX_OK;
}

int ngmix_get_loglike(ngmix_gauss2d *gmix, int64_t ngauss,
                      const ngmix_pixel *pixels, int64_t npix, double *loglike,
                      double *s2n_numer, double *s2n_denom, int64_t *npix_out)
{
    double o[4];
    int st = seam_pixpass(0, gmix, ngauss, pixels, npix, nullptr, 0, o);
    if (st) return st;
    *loglike = o[0] * -0.5;
    *s2n_numer = o[1];
    *s2n_denom = o[2];
    *npix_out = (int64_t)o[3];
    return NGMIX_OK;
}

int ngmix_fill_fdiff(ngmix_gauss2d *gmix, int64_t ngauss,
                     const ngmix_pixel *pixels, int64_t npix, double *fdiff,
                     int64_t start)
{
    double o[4];
    return seam_pixpass(1, gmix, ngauss, pixels, npix, fdiff, start, o);
}

int ngmix_get_model_s2n_sum(ngmix_gauss2d *gmix, int64_t ngauss,
                            const ngmix_pixel *pixels, int64_t npix,
                            double *s2n_sum)
{
    double o[4];
    int st = seam_pixpass(4, gmix, ngauss, pixels, npix, nullptr, 0, o);
    if (st) return st;
    *s2n_sum = o[2];
    return NGMIX_OK;
}

// --------------------------------------------- seam moment / iterative forms

int ngmix_get_weighted_sums(const ngmix_gauss2d *wt, int64_t ngauss,
                            const ngmix_pixel *pixels, int64_t npix, void *res,
                            int nmom, double maxrad)
{
    if (nmom != 6 && nmom != 17) return NGMIX_ERR_BAD_ARG;
    if (npix <= 0 || ngauss <= 0) return NGMIX_OK;
    const size_t rbytes = NGMIX_MOMENTS_RESULT_BYTES(nmom);
    WS_GET(d_gm, ngmix_gauss2d, 0, ngauss * sizeof(ngmix_gauss2d));
    WS_GET(d_px, ngmix_pixel, 1, npix * sizeof(ngmix_pixel));
    WS_GET(d_res, char, 2, rbytes);
    WS_GET(d_st, int32_t, 3, 8);
    NGMIX_HIP_CHECK(hipMemcpy(d_gm, wt, ngauss * sizeof(ngmix_gauss2d), hipMemcpyHostToDevice));
    NGMIX_HIP_CHECK(hipMemcpy(d_px, pixels, npix * sizeof(ngmix_pixel), hipMemcpyHostToDevice));
    NGMIX_HIP_CHECK(hipMemcpy(d_res, res, rbytes, hipMemcpyHostToDevice));
    int st = launch_weighted_sums_list(d_gm, (int)ngauss, d_px, npix, d_res, nmom,
                                       maxrad, d_st, nullptr);
    if (st) return st;
    int32_t kst = 0;
    NGMIX_HIP_CHECK(hipMemcpy(&kst, d_st, 4, hipMemcpyDeviceToHost));
    if (kst) return kst;
    NGMIX_HIP_CHECK(hipMemcpy(res, d_res, rbytes, hipMemcpyDeviceToHost));
    return NGMIX_OK;
}

int ngmix_admom(const ngmix_admom_conf *conf, ngmix_gauss2d *wt,
                const ngmix_pixel *pixels, int64_t npix, ngmix_admom_result *res)
{
    WS_GET(d_wt, ngmix_gauss2d, 0, sizeof(ngmix_gauss2d));
    WS_GET(d_px, ngmix_pixel, 1, (npix > 0 ? npix : 1) * sizeof(ngmix_pixel));
    WS_GET(d_res, ngmix_admom_result, 2, sizeof(ngmix_admom_result));
    WS_GET(d_st, int32_t, 3, 8);
    NGMIX_HIP_CHECK(hipMemcpy(d_wt, wt, sizeof(ngmix_gauss2d), hipMemcpyHostToDevice));
    if (npix > 0)
        NGMIX_HIP_CHECK(hipMemcpy(d_px, pixels, npix * sizeof(ngmix_pixel), hipMemcpyHostToDevice));
    NGMIX_HIP_CHECK(hipMemcpy(d_res, res, sizeof(ngmix_admom_result), hipMemcpyHostToDevice));
    // the seam form takes the reference's record: whatever sits in its padding,
    // the batch extension is off
    ngmix_admom_conf seam_conf = *conf;
    seam_conf.no_cov = 0;
    int st = launch_admom_list(&seam_conf, d_px, npix, d_wt, d_res, d_st, nullptr);
    if (st) return st;
    int32_t kst = 0;
    NGMIX_HIP_CHECK(hipMemcpy(&kst, d_st, 4, hipMemcpyDeviceToHost));
    NGMIX_HIP_CHECK(hipMemcpy(res, d_res, sizeof(ngmix_admom_result), hipMemcpyDeviceToHost));
    NGMIX_HIP_CHECK(hipMemcpy(wt, d_wt, sizeof(ngmix_gauss2d), hipMemcpyDeviceToHost));
    return kst;
}

int ngmix_em_run(int kind, const ngmix_em_conf *conf, ngmix_pixel *pixels,
                 int64_t npix, double *sums, ngmix_gauss2d *gmix, int64_t ngauss,
                 ngmix_gauss2d *gmix_psf, int64_t npsf, ngmix_gauss2d *gmix_conv,
                 int fill_zero_weight, int32_t *numiter, double *frac_diff,
                 double *sky)
{
    if (kind < 0 || kind > 3 || ngauss < 1 || npsf < 1) return NGMIX_ERR_BAD_ARG;
    static const int stride[4] = {14, 10, 8, 2};
    const int64_t nconv = ngauss * npsf;
    const size_t sbytes = (size_t)ngauss * stride[kind] * 8;
    WS_GET(d_gm, ngmix_gauss2d, 0, ngauss * sizeof(ngmix_gauss2d));
    WS_GET(d_psf, ngmix_gauss2d, 1, npsf * sizeof(ngmix_gauss2d));
    WS_GET(d_conv, ngmix_gauss2d, 2, nconv * sizeof(ngmix_gauss2d));
    WS_GET(d_px, ngmix_pixel, 3, (npix > 0 ? npix : 1) * sizeof(ngmix_pixel));
    WS_GET(d_sums, double, 4, sbytes);
    WS_GET(d_out, double, 5, 64);
    NGMIX_HIP_CHECK(hipMemcpy(d_gm, gmix, ngauss * sizeof(ngmix_gauss2d), hipMemcpyHostToDevice));
    NGMIX_HIP_CHECK(hipMemcpy(d_psf, gmix_psf, npsf * sizeof(ngmix_gauss2d), hipMemcpyHostToDevice));
    NGMIX_HIP_CHECK(hipMemcpy(d_conv, gmix_conv, nconv * sizeof(ngmix_gauss2d), hipMemcpyHostToDevice));
    if (npix > 0)
        NGMIX_HIP_CHECK(hipMemcpy(d_px, pixels, npix * sizeof(ngmix_pixel), hipMemcpyHostToDevice));
    NGMIX_HIP_CHECK(hipMemcpy(d_sums, sums, sbytes, hipMemcpyHostToDevice));
    int32_t *d_st = (int32_t *)(d_out + 4);
    int st = launch_em_list(kind, conf, d_px, npix, d_sums, d_gm, (int)ngauss, d_psf,
                            (int)npsf, d_conv, fill_zero_weight, d_out, d_st, nullptr);
    if (st) return st;
    double out[5];
    NGMIX_HIP_CHECK(hipMemcpy(out, d_out, 40, hipMemcpyDeviceToHost));
    int32_t kst;
    memcpy(&kst, &out[4], 4);
    NGMIX_HIP_CHECK(hipMemcpy(gmix, d_gm, ngauss * sizeof(ngmix_gauss2d), hipMemcpyDeviceToHost));
    NGMIX_HIP_CHECK(hipMemcpy(gmix_conv, d_conv, nconv * sizeof(ngmix_gauss2d), hipMemcpyDeviceToHost));
    NGMIX_HIP_CHECK(hipMemcpy(sums, d_sums, sbytes, hipMemcpyDeviceToHost));
    *numiter = (int32_t)out[0];
    *frac_diff = out[1];
    *sky = out[2];
    return kst;
}

int ngmix_deriv_images(const double *gpars, const double *dcov, int64_t ngauss,
                       const double *vv, const double *uu, const double *area,
                       int64_t npix, double *out)
{
    if (npix <= 0 || ngauss <= 0) return NGMIX_OK;
    WS_GET(d_gp, double, 0, ngauss * 6 * 8);
    WS_GET(d_dc, double, 1, ngauss * 9 * 8);
    WS_GET(d_v, double, 2, npix * 8);
    WS_GET(d_u, double, 3, npix * 8);
    WS_GET(d_a, double, 4, npix * 8);
    WS_GET(d_o, double, 5, npix * 6 * 8);
    NGMIX_HIP_CHECK(hipMemcpy(d_gp, gpars, ngauss * 6 * 8, hipMemcpyHostToDevice));
    NGMIX_HIP_CHECK(hipMemcpy(d_dc, dcov, ngauss * 9 * 8, hipMemcpyHostToDevice));
    NGMIX_HIP_CHECK(hipMemcpy(d_v, vv, npix * 8, hipMemcpyHostToDevice));
    NGMIX_HIP_CHECK(hipMemcpy(d_u, uu, npix * 8, hipMemcpyHostToDevice));
    NGMIX_HIP_CHECK(hipMemcpy(d_a, area, npix * 8, hipMemcpyHostToDevice));
    NGMIX_HIP_CHECK(hipMemcpy(d_o, out, npix * 6 * 8, hipMemcpyHostToDevice));
    int st = launch_deriv_list(d_gp, d_dc, (int)ngauss, d_v, d_u, d_a, npix, d_o, nullptr);
    if (st) return st;
    NGMIX_HIP_CHECK(hipMemcpy(out, d_o, npix * 6 * 8, hipMemcpyDeviceToHost));
    return NGMIX_OK;
}

// ------------------------------------------------------------- batch forms

int ngmix_prepsf_sums_batch(const double *kim_re, const double *kim_im, const double *kpsf_re,
                            const double *kpsf_im, const double *pix, const double *knoise_re,
                            const double *knoise_im, const double *pnoise_stamp,
                            double noise_scale, const double *max_amp, const double *py,
                            const double *px, const int32_t *irow, const int32_t *icol,
                            const double *fk, const double *wgt, int64_t nstamps, int nmodes,
                            int64_t stride_n, int64_t stride_r, int nrows, int ncols,
                            double df2, double df4, double *out, void *stream)
{
    return launch_prepsf_sums(kim_re, kim_im, kpsf_re, kpsf_im, pix, knoise_re, knoise_im,
                              pnoise_stamp, noise_scale, max_amp, py, px, irow, icol, fk, wgt,
                              nstamps, nmodes, stride_n, stride_r, nrows, ncols, df2, df4, out,
                              (hipStream_t)stream);
}

int ngmix_fastexp_batch(const double *x, double *out, int64_t n, int which, void *stream)
{
    return launch_fastexp(x, out, n, which, (hipStream_t)stream);
}

int ngmix_weight_to_ierr_batch(const double *weight, double *ierr, int64_t n,
                               void *stream)
{
    return launch_weight_to_ierr(weight, ierr, n, (hipStream_t)stream);
}

int ngmix_template_sums_batch(const ngmix_batch *batch, const double *model,
                              const double *mult, double *out, void *stream)
{
    return launch_template_sums(batch, model, mult, out, (hipStream_t)stream);
}

int ngmix_count_kept_batch(ngmix_stamp *stamps, int64_t nstamps,
                           const double *ierr, void *stream)
{
    return launch_count_kept(stamps, nstamps, ierr, (hipStream_t)stream);
}

int ngmix_fill_model_batch(ngmix_gauss2d *gmix, int64_t nstamps, int ngauss,
                           int model, const double *pars, int npars,
                           const double *cm_extra, int32_t *status, void *stream)
{
    return launch_fill_model(gmix, nstamps, ngauss, model, pars, npars, cm_extra,
                             status, (hipStream_t)stream);
}

int ngmix_convolve_fill_batch(ngmix_gauss2d *out, const ngmix_gauss2d *gmix,
                              int ngauss, const ngmix_gauss2d *psf, int npsf,
                              int64_t nstamps, int32_t *status, void *stream)
{
    return launch_convolve_fill(out, gmix, ngauss, psf, npsf, nstamps, status,
                                (hipStream_t)stream);
}

int ngmix_set_norms_batch(ngmix_gauss2d *gmix, int ngauss, int64_t nstamps,
                          int32_t *status, void *stream)
{
    return launch_set_norms(gmix, ngauss, nstamps, status, (hipStream_t)stream);
}

int ngmix_loglike_batch(const ngmix_batch *batch, ngmix_gauss2d *gmix,
                        double *out, int32_t *status, void *stream)
{
    return launch_loglike_grid(batch, gmix, out, status, stream);
}

int ngmix_fill_fdiff_batch(const ngmix_batch *batch, ngmix_gauss2d *gmix,
                           double *fdiff, const int64_t *fdiff_start,
                           int32_t *status, void *stream)
{
    return launch_fdiff_grid(batch, gmix, fdiff, fdiff_start, status, stream);
}

int ngmix_render_batch(const ngmix_batch *batch, ngmix_gauss2d *gmix,
                       double *image, int fast_exp, int32_t *status,
                       void *stream)
{
    return launch_render_grid(batch, gmix, image, fast_exp, status, stream);
}

int ngmix_model_s2n_sum_batch(const ngmix_batch *batch, ngmix_gauss2d *gmix,
                              double *out, int32_t *status, void *stream)
{
    return launch_s2n_grid(batch, gmix, out, status, stream);
}

int ngmix_weighted_sums_batch(const ngmix_batch *batch, const ngmix_gauss2d *gmix,
                              void *res, int nmom, const double *maxrad,
                              int32_t *status, void *stream)
{
    return launch_weighted_sums_grid(batch, gmix, res, nmom, maxrad, status,
                                     (hipStream_t)stream);
}

int ngmix_admom_batch(const ngmix_admom_conf *conf, const ngmix_batch *batch,
                      ngmix_gauss2d *wt, ngmix_admom_result *res, int32_t *status,
                      void *stream)
{
    return launch_admom_grid(conf, batch, wt, res, status, (hipStream_t)stream);
}

int ngmix_em_batch(int kind, const ngmix_em_conf *conf, const ngmix_batch *batch,
                   ngmix_gauss2d *gmix, int ngauss, ngmix_gauss2d *gmix_psf,
                   int npsf, ngmix_gauss2d *gmix_conv, const double *sky_in,
                   int fill_zero_weight, double *out, int32_t *status, void *stream)
{
    return launch_em_grid(kind, conf, batch, gmix, ngauss, gmix_psf, npsf, gmix_conv,
                          sky_in, fill_zero_weight, out, status, (hipStream_t)stream);
}

int ngmix_deriv_images_batch(const ngmix_batch *batch, const double *gpars,
                             const double *dcov, double *out,
                             const int64_t *out_start, void *stream)
{
    return launch_deriv_grid(batch, gpars, dcov, out, out_start, (hipStream_t)stream);
}

int ngmix_lm_init(ngmix_lm_state *states, int64_t nobj, int npars, const double *x0,
                  double ftol, double xtol, double gtol, int maxfev, double factor,
                  int mode, const double *lo, const double *hi)
{
    if (npars < 1 || npars > NGMIX_LM_NPMAX || nobj < 0) {
        set_last_error_msg("ngmix_lm_init: npars must be 1..NGMIX_LM_NPMAX");
        return NGMIX_ERR_BAD_ARG;
    }
    for (int64_t i = 0; i < nobj; i++)
        lmcore::lm_init(states[i], npars, x0 + i * npars, ftol, xtol, gtol, maxfev,
                        factor, mode, lo, hi);
    return NGMIX_OK;
}

int ngmix_lm_init_batch(ngmix_lm_state *states, int64_t nobj, int npars, const double *x0,
                        double ftol, double xtol, double gtol, int maxfev, double factor,
                        int mode, const double *lo, const double *hi, void *stream)
{
    if (npars < 1 || npars > NGMIX_LM_NPMAX || nobj < 0) {
        set_last_error_msg("ngmix_lm_init_batch: npars must be 1..NGMIX_LM_NPMAX");
        return NGMIX_ERR_BAD_ARG;
    }
    return launch_lm_init(states, nobj, npars, x0, ftol, xtol, gtol, maxfev, factor, mode,
                          lo, hi, (hipStream_t)stream);
}

int ngmix_lm_prior_sums_batch(const ngmix_lm_state *states, int64_t nobj,
                              const ngmix_simple_sep_prior *prior, double step_rel,
                              double *obj_sums, void *stream)
{
    return launch_lm_prior_sums(states, nobj, prior, step_rel, obj_sums,
                                (hipStream_t)stream);
}

int ngmix_lm_prior_finish_batch(const ngmix_lm_state *states, int64_t nobj,
                                const ngmix_simple_sep_prior *prior, double *ffx, double *lnp,
                                void *stream)
{
    return launch_lm_prior_finish(states, nobj, prior, ffx, lnp, (hipStream_t)stream);
}

int ngmix_first_pixels_fdiff2_batch(const ngmix_batch *batch, const int64_t *stamp_of,
                                    const ngmix_gauss2d *gmix, int ngauss, int64_t nobj,
                                    int nskip, double *out, void *stream)
{
    return launch_first_pixels_fdiff2(batch, stamp_of, gmix, ngauss, nobj, nskip, out,
                                      (hipStream_t)stream);
}

int ngmix_lm_prior_sums_host(const ngmix_lm_state *states, int64_t nobj,
                             const ngmix_simple_sep_prior *prior, double step_rel,
                             double *obj_sums)
{
    if (!states || !prior || !obj_sums || prior->nband < 1 ||
        prior->nband > NGMIX_PRIOR_MAXBAND || prior->nmid < 0 || prior->nmid > NGMIX_PRIOR_MAXMID)
        return NGMIX_ERR_BAD_ARG;
    for (int64_t o = 0; o < nobj; o++) {
        const int n = states[o].n;
        if (states[o].phase == LM_PHASE_DONE) continue;
        lmcore::simple_sep_normal_sums(*prior, states[o], step_rel,
                                       obj_sums + o * (int64_t)(n * (n + 1) / 2 + n + 1));
    }
    return NGMIX_OK;
}

int ngmix_simple_sep_prior_eval(const ngmix_simple_sep_prior *prior, const double *pars,
                                double *rows, double *lnprob)
{
    if (!prior || prior->nband < 1 || prior->nband > NGMIX_PRIOR_MAXBAND || prior->nmid < 0 ||
        prior->nmid > NGMIX_PRIOR_MAXMID)
        return -2;
    double r[lmcore::PRIOR_KMAX], lnp = 0.0;
    if (!lmcore::simple_sep_rows(*prior, pars, r, &lnp)) return -1;
    const int k = 4 + prior->nmid + prior->nband;
    for (int i = 0; i < k; i++) rows[i] = r[i];
    if (lnprob) *lnprob = lnp;
    return k;
}

int64_t ngmix_lm_advance_host(ngmix_lm_state *states, int64_t nobj, const double *ff,
                              const double *g, const double *A)
{
    // fits of 6 .. 10 parameters run the code the device runs for them
    // (NGMIX_LM_GENERIC: the generic code, for comparing the two)
    const bool generic = getenv("NGMIX_LM_GENERIC") != nullptr;
    int64_t running = 0;
    for (int64_t i = 0; i < nobj; i++) {
        const double *gi = g + i * NGMIX_LM_NPMAX;
        const double *Ai = A + i * NGMIX_LM_NPMAX * NGMIX_LM_NPMAX;
        const int n = states[i].n;
        if (generic || n < 6 || n > 10) lmcore::lm_advance(states[i], ff[i], gi, Ai);
        else if (n == 6) lm_advance_host_reg<6>(states[i], ff[i], gi, Ai);
        else if (n == 7) lm_advance_host_reg<7>(states[i], ff[i], gi, Ai);
        else if (n == 8) lm_advance_host_reg<8>(states[i], ff[i], gi, Ai);
        else if (n == 9) lm_advance_host_reg<9>(states[i], ff[i], gi, Ai);
        else lm_advance_host_reg<10>(states[i], ff[i], gi, Ai);
        if (states[i].phase != NGMIX_LM_PHASE_DONE) running++;
    }
    return running;
}

int ngmix_lm_eval_batch(const ngmix_batch *batch, int model, int fd,
                        const ngmix_lm_state *states, const int32_t *stamp_obj,
                        const int32_t *stamp_band, const ngmix_gauss2d *psf, int npsf,
                        double *sums, int32_t *status, double *stamp_stats, void *stream)
{
    return launch_lm_eval(batch, model, fd, states, stamp_obj, stamp_band, psf, npsf,
                          sums, status, stamp_stats, (hipStream_t)stream);
}

int ngmix_lm_advance_batch(ngmix_lm_state *states, int64_t nobj,
                           const int64_t *obj_start, const int32_t *stamp_band,
                           const double *sums, int nloc, const double *obj_sums,
                           int32_t *nactive, const double *stamp_stats, double *obj_stats,
                           void *stream)
{
    return launch_lm_advance(states, nobj, obj_start, stamp_band, sums, nloc, obj_sums,
                             nactive, stamp_stats, obj_stats, (hipStream_t)stream);
}

int ngmix_lm_finalize_batch(const ngmix_lm_state *states, int64_t nobj,
                            const int64_t *npix_obj, const double *ff_extra,
                            double pdef, double cdef, double *rec, void *stream)
{
    return launch_lm_finalize(states, nobj, npix_obj, ff_extra, pdef, cdef, rec,
                              (hipStream_t)stream);
}

int ngmix_lm_pack_batch(const ngmix_lm_state *states, int64_t nobj, int npars,
                        const double *rec, const double *obj_stats, const double *tot,
                        const int64_t *npix_obj, double *head, double *cols,
                        double *cov_tri, void *stream)
{
    return launch_lm_pack(states, nobj, npars, rec, obj_stats, tot, npix_obj, head, cols,
                          cov_tri, (hipStream_t)stream);
}

int ngmix_lm_rounds_batch(const ngmix_lm_problem *problem, int nrounds, int32_t *counts,
                          int32_t *counts_host, void **events, void *stream)
{
    return launch_lm_rounds(problem, nrounds, counts, counts_host, events,
                            (hipStream_t)stream);
}

int ngmix_lm_precise_cov_batch(const ngmix_lm_problem *problem, double *psums, void *stream)
{
    return launch_lm_precise_cov(problem, psums, (hipStream_t)stream);
}

int ngmix_events_create(int n, void **events)
{
    if (n < 0 || (n > 0 && !events)) return NGMIX_ERR_BAD_ARG;
    for (int i = 0; i < n; i++) events[i] = nullptr;
    for (int i = 0; i < n; i++) {
        hipEvent_t e;
        const hipError_t err = hipEventCreate(&e);
        if (err != hipSuccess) {
            set_last_error("hipEventCreate", err);
            ngmix_events_destroy(i, events);
            return NGMIX_ERR_HIP;
        }
        events[i] = (void *)e;
    }
    return NGMIX_OK;
}

int ngmix_events_destroy(int n, void **events)
{
    if (n < 0 || (n > 0 && !events)) return NGMIX_ERR_BAD_ARG;
    for (int i = 0; i < n; i++) {
        if (events[i]) (void)hipEventDestroy((hipEvent_t)events[i]);
        events[i] = nullptr;
    }
    return NGMIX_OK;
}

int ngmix_event_record(void *event, void *stream)
{
    if (!event) return NGMIX_ERR_BAD_ARG;
    NGMIX_HIP_CHECK(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return NGMIX_OK;
}

int ngmix_event_synchronize(void *event)
{
    if (!event) return NGMIX_ERR_BAD_ARG;
    NGMIX_HIP_CHECK(hipEventSynchronize((hipEvent_t)event));
    return NGMIX_OK;
}

int ngmix_event_elapsed_ms(void *start, void *stop, float *ms)
{
    if (!start || !stop || !ms) return NGMIX_ERR_BAD_ARG;
    NGMIX_HIP_CHECK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return NGMIX_OK;
}

int64_t ngmix_launch_census(char *buf, int64_t buflen, int reset)
{
    std::lock_guard<std::mutex> lock(g_census_mutex);
    std::string text;
    for (const auto &kv : g_census)
        text += kv.first + "\t" + std::to_string(kv.second) + "\n";
    if (reset) g_census.clear();
    if (buf && buflen > 0) {
        const size_t n = text.size() < (size_t)buflen - 1 ? text.size() : (size_t)buflen - 1;
        memcpy(buf, text.data(), n);
        buf[n] = 0;
    }
    return (int64_t)text.size() + 1;
}

}  // extern "C"
