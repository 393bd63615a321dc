// gmixprep.hip -- batched O(ngauss) parameter prep on the device:
// model fills (ngmix/gmix/gmix_nb.py:307-558), gmix_convolve_fill (:609-649)
// and gmix_set_norms (:176-218), one thread per stamp.  These touch ~100 B
// per gaussian and exist so the LM / EM pipelines never leave the device
// between evaluations; they are not bandwidth- or compute-relevant.
#include "device_utils.hpp"
#include "launch.hpp"

namespace ngmix {

__constant__ ModelTables c_tables = NGMIX_MODEL_TABLES;

__global__ __launch_bounds__(BLOCK) void fill_model_kernel(
    ngmix_gauss2d *gmix, int64_t nstamps, int ngauss, int model,
    const double *pars, int npars, const double *cm_extra, int32_t *status)
{
    const int64_t s = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
    if (s >= nstamps) return;
    const double *p = pars + s * npars;
    FillCtx c;
    const int st = fill_prepare(c_tables, model, ngauss, p,
                                cm_extra ? cm_extra + 3 * s : nullptr, c);
    if (status) status[s] = st;
    if (st) return;  // the reference raises before touching the mixture
    ngmix_gauss2d *gm = gmix + s * ngauss;
    for (int i = 0; i < ngauss; i++) {
        ngmix_gauss2d g;
        fill_component(c_tables, c, p, i, g);
        gm[i] = g;
    }
}

__global__ __launch_bounds__(BLOCK) void convolve_fill_kernel(
    ngmix_gauss2d *out, const ngmix_gauss2d *gmix, int ngauss,
    const ngmix_gauss2d *psf, int npsf, int64_t nstamps, int32_t *status)
{
    const int64_t s = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
    if (s >= nstamps) return;
    const ngmix_gauss2d *q = psf + s * npsf;
    const ngmix_gauss2d *o = gmix + s * ngauss;
    ngmix_gauss2d *dst = out + s * (int64_t)ngauss * npsf;
    double rowcen, colcen, psum;
    const int st = gmix_cen(q, npsf, rowcen, colcen, psum);
    if (status) status[s] = st;
    if (st) return;
    const double ipsum = 1.0 / psum;
    int itot = 0;
    for (int io = 0; io < ngauss; io++) {
        for (int ip = 0; ip < npsf; ip++) {
            ngmix_gauss2d g;
            convolve_component(o[io], q[ip], rowcen, colcen, ipsum, g);
            dst[itot++] = g;
        }
    }
}

__global__ __launch_bounds__(BLOCK) void set_norms_kernel(ngmix_gauss2d *gmix,
                                                         int ngauss,
                                                         int64_t nstamps,
                                                         int32_t *status)
{
    const int64_t s = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
    if (s >= nstamps) return;
    ngmix_gauss2d *gm = gmix + s * ngauss;
    int st = NGMIX_OK;
    for (int i = 0; i < ngauss; i++) {
        ngmix_gauss2d g = gm[i];
        st = gauss_set_norm(g);
        if (st) break;  // earlier gaussians keep their fresh norms
        gm[i] = g;
    }
    if (status) status[s] = st;
}

static unsigned nblocks(int64_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }

int launch_fill_model(ngmix_gauss2d *gmix, int64_t nstamps, int ngauss, int model,
                      const double *pars, int npars, const double *cm_extra,
                      int32_t *status, hipStream_t s)
{
    if (nstamps <= 0) return NGMIX_OK;
    hipLaunchKernelGGL(fill_model_kernel, dim3(nblocks(nstamps)), dim3(BLOCK), 0, s,
                       gmix, nstamps, ngauss, model, pars, npars, cm_extra, status);
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

int launch_convolve_fill(ngmix_gauss2d *out, const ngmix_gauss2d *gmix, int ngauss,
                         const ngmix_gauss2d *psf, int npsf, int64_t nstamps,
                         int32_t *status, hipStream_t s)
{
    if (nstamps <= 0) return NGMIX_OK;
    hipLaunchKernelGGL(convolve_fill_kernel, dim3(nblocks(nstamps)), dim3(BLOCK), 0,
                       s, out, gmix, ngauss, psf, npsf, nstamps, status);
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

int launch_set_norms(ngmix_gauss2d *gmix, int ngauss, int64_t nstamps,
                     int32_t *status, hipStream_t s)
{
    if (nstamps <= 0) return NGMIX_OK;
    hipLaunchKernelGGL(set_norms_kernel, dim3(nblocks(nstamps)), dim3(BLOCK), 0, s,
                       gmix, ngauss, nstamps, status);
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

}  // namespace ngmix
