// derivs.hip -- value image + analytic derivative images of a composed
// gaussian mixture with respect to [cen1, cen2, g1, g2, T]
// (reference: ngmix/fitting/derivs_nb.py:40-127).
//
// The reference loops gaussian-outer / pixel-inner; every output element
// out[a, ipix] only ever receives terms of its own pixel, in gaussian index
// order, so a pixel-parallel kernel with the gaussian loop inside produces
// bit-identical images.  Per-gaussian constants (Q = Sigma^-1, norm, traces)
// are staged in LDS once.
#include "device_utils.hpp"
#include "launch_iter.hpp"

namespace ngmix {

__constant__ double c_exp_table_d[16] = NGMIX_EXP_TABLE;

struct DerivGauss {
    double vcen, ucen, norm, w11, w12, w22;
    double dc[9];
    double trs[3];
    double ok;  // 0 when det <= 0: the reference skips the gaussian
    double pad;
};

__device__ __forceinline__ void stage_deriv_gauss(double *tab, DerivGauss *dg,
                                                  const double *gpars,
                                                  const double *dcov, int ng)
{
    const double TWO_PI = 2.0 * M_PI;
    if (threadIdx.x < 16) tab[threadIdx.x] = c_exp_table_d[threadIdx.x];
    for (int ig = threadIdx.x; ig < ng; ig += BLOCK) {
        const double *gp = gpars + 6 * ig;
        const double *dc = dcov + 9 * ig;
        DerivGauss d;
        const double p = gp[0], irr = gp[3], irc = gp[4], icc = gp[5];
        d.vcen = gp[1];
        d.ucen = gp[2];
        const double det = irr * icc - irc * irc;
        d.ok = det <= 0.0 ? 0.0 : 1.0;
        d.norm = p / (TWO_PI * sqrt(det));
        d.w11 = icc / det;
        d.w12 = -irc / det;
        d.w22 = irr / det;
        for (int k = 0; k < 9; k++) d.dc[k] = dc[k];
        for (int a = 0; a < 3; a++)
            d.trs[a] = d.w11 * dc[a * 3 + 0] + 2.0 * d.w12 * dc[a * 3 + 1] +
                       d.w22 * dc[a * 3 + 2];
        d.pad = 0.0;
        dg[ig] = d;
    }
    __syncthreads();
}

// accumulate the six images' terms of one pixel into acc[6]
__device__ __forceinline__ void deriv_pixel(const DerivGauss *dg, int ng, double v,
                                            double u, double area,
                                            const double *tab, double (&acc)[6])
{
    for (int ig = 0; ig < ng; ig++) {
        const DerivGauss &d = dg[ig];
        if (d.ok == 0.0) continue;
        const double dv = v - d.vcen;
        const double du = u - d.ucen;
        const double qv = d.w11 * dv + d.w12 * du;
        const double qu = d.w12 * dv + d.w22 * du;
        const double chi2 = dv * qv + du * qu;
        // written as in derivs_nb.py:104 -- a NaN chi2 is NOT skipped here
        if (chi2 >= MAX_CHI2 || chi2 < 0.0) continue;
        double val = d.norm * fexp(-0.5 * chi2, tab) * area;
        double valc;
        if (chi2 > APOD_CHI2) {
            const double w = apod_window(chi2);
            valc = val * (w - 2.0 * apod_window_deriv(chi2));
            val *= w;
        } else {
            valc = val;
        }
        acc[0] += val;
        acc[1] += valc * qv;
        acc[2] += valc * qu;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const double quad = qv * qv * d.dc[a * 3 + 0] +
                                2.0 * qv * qu * d.dc[a * 3 + 1] +
                                qu * qu * d.dc[a * 3 + 2];
            acc[3 + a] += 0.5 * (valc * quad - val * d.trs[a]);
        }
    }
}

// NaN chi2 indexes outside the exp table in the reference (undefined there);
// fexp here clamps nothing either, so guard the table read against NaN only.

__global__ __launch_bounds__(BLOCK) void deriv_list_kernel(
    const double *gpars, const double *dcov, int ng, const double *vv,
    const double *uu, const double *area, int64_t npix, double *out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *tab = (double *)smem;
    DerivGauss *dg = (DerivGauss *)(tab + 16);
    stage_deriv_gauss(tab, dg, gpars, dcov, ng);
    for (int64_t i = blockIdx.x * (int64_t)BLOCK + threadIdx.x; i < npix;
         i += (int64_t)gridDim.x * BLOCK) {
        double acc[6];
#pragma unroll
        for (int a = 0; a < 6; a++) acc[a] = out[a * npix + i];
        deriv_pixel(dg, ng, vv[i], uu[i], area[i], tab, acc);
#pragma unroll
        for (int a = 0; a < 6; a++) out[a * npix + i] = acc[a];
    }
}

__global__ __launch_bounds__(BLOCK) void deriv_grid_kernel(
    const ngmix_stamp *stamps, const double *ierr, const ngmix_jacobian *jacs,
    const double *gpars, const double *dcov, double *out,
    const int64_t *out_start, int max_ngauss, int nchunks_cap)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *tab = (double *)smem;
    DerivGauss *dg = (DerivGauss *)(tab + 16);
    unsigned long long *cmask = (unsigned long long *)(dg + max_ngauss);
    int *cpre = (int *)(cmask + nchunks_cap);

    const int s = blockIdx.x;
    const ngmix_stamp st = stamps[s];
    const ngmix_jacobian jac = jacs[s];
    const int npix = st.nrow * st.ncol;
    const bool izw = (st.flags & NGMIX_STAMP_IGNORE_ZERO_WEIGHT) != 0;
    const bool masked = izw && st.npix_kept != npix;
    stage_deriv_gauss(tab, dg, gpars + 6 * (size_t)st.gm_off,
                      dcov + 9 * (size_t)st.gm_off, st.ngauss);
    if (masked) build_rank_tables(cmask, cpre, ierr + st.pix_off, npix);
    const double area = jac.scale * jac.scale;
    const int64_t nk = st.npix_kept;
    double *o = out + out_start[s];
    for (int p = threadIdx.x; p < npix; p += BLOCK) {
        if (masked && !(ierr[st.pix_off + p] > 0.0)) continue;
        const int row = p / st.ncol, col = p - row * st.ncol;
        double v, u;
        jacobian_vu(jac, (double)row, (double)col, v, u);
        const int k = masked ? kept_rank(cmask, cpre, p) : p;
        double acc[6];
#pragma unroll
        for (int a = 0; a < 6; a++) acc[a] = o[a * nk + k];
        deriv_pixel(dg, st.ngauss, v, u, area, tab, acc);
#pragma unroll
        for (int a = 0; a < 6; a++) o[a * nk + k] = acc[a];
    }
}

int launch_deriv_list(const double *gpars, const double *dcov, int ng,
                      const double *vv, const double *uu, const double *area,
                      int64_t npix, double *out, hipStream_t s)
{
    if (npix <= 0 || ng <= 0) return NGMIX_OK;
    int64_t nb = (npix + BLOCK - 1) / BLOCK;
    if (nb > 1024) nb = 1024;
    const size_t lds = 16 * 8 + (size_t)ng * sizeof(DerivGauss);
    hipLaunchKernelGGL(deriv_list_kernel, dim3((unsigned)nb), dim3(BLOCK), lds, s,
                       gpars, dcov, ng, vv, uu, area, npix, out);
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

int launch_deriv_grid(const ngmix_batch *b, const double *gpars, const double *dcov,
                      double *out, const int64_t *out_start, hipStream_t s)
{
    if (b->nstamps <= 0) return NGMIX_OK;
    const int max_ng = b->max_ngauss > 0 ? b->max_ngauss : 1;
    const int nchunks_cap = b->any_masked ? (b->max_npix + 63) / 64 : 0;
    const size_t lds = 16 * 8 + (size_t)max_ng * sizeof(DerivGauss) +
                       (size_t)nchunks_cap * 12 + 16;
    if (lds > 64 * 1024) {
        set_last_error_msg("deriv_images: LDS budget exceeded");
        return NGMIX_ERR_BAD_ARG;
    }
    hipLaunchKernelGGL(deriv_grid_kernel, dim3((unsigned)b->nstamps), dim3(BLOCK),
                       lds, s, b->stamps, b->ierr, b->jac, gpars, dcov, out,
                       out_start, max_ng, nchunks_cap);
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

}  // namespace ngmix
