// em_common.hpp -- the M-step shared by the EM kernels (em.hip, em_wave.hip).
#pragma once

#include "iter_common.hpp"

namespace ngmix {

// M-step: gmix_set_from_sums{,_fixcen,_fixcov,_fluxonly}
// (em_nb.py:284-354, 587-655, 954-1000, 1200-1241); tot[] holds, per object
// gaussian, [pnew, vsum, usum, u2sum, uvsum, v2sum].  The psf moments
// (gmix_get_moms) and centre / flux (gmix_get_cen) are passed in: the psf
// does not change during a run.
__device__ __forceinline__ int em_mstep_psf(int kind, ngmix_gauss2d *gmix, int ngauss,
                                            const ngmix_gauss2d *psf, int npsf,
                                            ngmix_gauss2d *conv, const double *tot,
                                            double psf_irr, double psf_irc,
                                            double psf_icc, double rowcen,
                                            double colcen, double ipsum)
{
    const double minval = 1.0e-4;
    for (int i = 0; i < ngauss; i++) {
        const double *ts = tot + 6 * i;
        ngmix_gauss2d &gauss = gmix[i];
        const double p = ts[0];
        if (kind == NGMIX_EM_FLUXONLY) {
            gauss_set(gauss, p, gauss.row, gauss.col, gauss.irr, gauss.irc, gauss.icc);
            continue;
        }
        if (p == 0.0) return NGMIX_ERR_ZERO_DIV;  // pinv = 1.0/p
        const double pinv = 1.0 / p;
        if (kind == NGMIX_EM_FIXCOV) {
            const double v = ts[1] * pinv;
            const double u = ts[2] * pinv;
            gauss_set(gauss, p, v, u, gauss.irr, gauss.irc, gauss.icc);
            continue;
        }
        double v = gauss.row, u = gauss.col;
        if (kind == NGMIX_EM_FULL) {
            v = ts[1] * pinv;
            u = ts[2] * pinv;
        }
        double irr = ts[5] * pinv;
        double irc = ts[4] * pinv;
        double icc = ts[3] * pinv;
        irr = irr - psf_irr;
        irc = irc - psf_irc;
        icc = icc - psf_icc;
        if (irr < 0.0 || icc < 0.0) {
            irr = minval;
            irc = 0.0;
            icc = minval;
        }
        const double det = irr * icc - irc * irc;
        if (det < LOW_DETVAL) {
            const double T = irr + icc;
            irr = icc = T / 2;
            irc = 0.0;
        }
        gauss_set(gauss, p, v, u, irr, irc, icc);
    }
    // gmix_convolve_fill + gmix_set_norms on the convolved mixture
    int itot = 0;
    for (int io = 0; io < ngauss; io++)
        for (int ip = 0; ip < npsf; ip++)
            convolve_component(gmix[io], psf[ip], rowcen, colcen, ipsum, conv[itot++]);
    for (int i = 0; i < ngauss * npsf; i++) {
        const int st = gauss_set_norm(conv[i]);
        if (st) return st;
    }
    return NGMIX_OK;
}

__device__ __forceinline__ int em_mstep(int kind, ngmix_gauss2d *gmix, int ngauss,
                                        const ngmix_gauss2d *psf, int npsf,
                                        ngmix_gauss2d *conv, const double *tot)
{
    double psf_irr = 0.0, psf_irc = 0.0, psf_icc = 0.0;
    if (kind == NGMIX_EM_FULL || kind == NGMIX_EM_FIXCEN) {
        const int st = gmix_moms(psf, npsf, psf_irr, psf_irc, psf_icc);
        if (st) return st;
    }
    // a zero-flux psf raises in gmix_get_cen only after the object gaussians
    // have been set; nothing of that partial state is observable
    double rowcen, colcen, psum;
    const int st = gmix_cen(psf, npsf, rowcen, colcen, psum);
    if (st) {
        // reproduce the reference's partial update before the raise
        em_mstep_psf(kind, gmix, ngauss, psf, 0, conv, tot, psf_irr, psf_irc, psf_icc,
                     0.0, 0.0, 0.0);
        return st;
    }
    return em_mstep_psf(kind, gmix, ngauss, psf, npsf, conv, tot, psf_irr, psf_irc,
                        psf_icc, rowcen, colcen, 1.0 / psum);
}

}  // namespace ngmix
