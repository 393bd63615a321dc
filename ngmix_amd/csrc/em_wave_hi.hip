// em_wave_hi.hip -- the fused EM kernels (em_wave_impl.hpp) for 4 .. 6 object
// gaussians: the reference's em_run is general in the gaussian count
// (em_nb.py:160-246,284-354), and psf / galaxy fits with four to six free
// gaussians used to fall to the generic 256-thread kernel of em.hip, six
// times slower per gaussian than the one-wave form (tools/bench_em_ng.py).
// One, two or four waves per stamp (<= 64 x 64 pixels); a translation unit of
// its own so that it compiles beside em_wave.hip.  Seven and eight gaussians:
// em_wave_8.hip.
#include "em_wave_impl.hpp"

namespace ngmix {

template <int KIND>
static void em_wave_launch_ng_hi(const ngmix_em_conf *conf, const ngmix_batch *b,
                                 ngmix_gauss2d *gmix, int ngauss, ngmix_gauss2d *psf,
                                 int npsf, ngmix_gauss2d *conv, const double *sky_in,
                                 int fzw, double *out, int32_t *status, hipStream_t s)
{
    if (ngauss == 4)
        em_wave_launch<KIND, 4>(conf, b, gmix, psf, npsf, conv, sky_in, fzw, out, status, s);
    else if (ngauss == 5)
        em_wave_launch<KIND, 5>(conf, b, gmix, psf, npsf, conv, sky_in, fzw, out, status, s);
    else
        em_wave_launch<KIND, 6>(conf, b, gmix, psf, npsf, conv, sky_in, fzw, out, status, s);
}

int launch_em_wave_hi(int kind, const ngmix_em_conf *conf, const ngmix_batch *b,
                      ngmix_gauss2d *gmix, int ngauss, ngmix_gauss2d *psf, int npsf,
                      ngmix_gauss2d *conv, const double *sky_in, int fzw, double *out,
                      int32_t *status, hipStream_t s)
{
    if (ngauss > 6)
        return launch_em_wave_8(kind, conf, b, gmix, ngauss, psf, npsf, conv, sky_in, fzw,
                                out, status, s);
    if (ngauss < 4 || b->max_npix > 16 * BLOCK) {
        set_last_error_msg("launch_em_wave_hi: 4..6 gaussians on stamps of <= 4096 pixels");
        return NGMIX_ERR_BAD_ARG;
    }
    switch (kind) {
    case NGMIX_EM_FULL:
        em_wave_launch_ng_hi<NGMIX_EM_FULL>(conf, b, gmix, ngauss, psf, npsf, conv, sky_in,
                                            fzw, out, status, s);
        break;
    case NGMIX_EM_FIXCEN:
        em_wave_launch_ng_hi<NGMIX_EM_FIXCEN>(conf, b, gmix, ngauss, psf, npsf, conv,
                                              sky_in, fzw, out, status, s);
        break;
    case NGMIX_EM_FIXCOV:
        em_wave_launch_ng_hi<NGMIX_EM_FIXCOV>(conf, b, gmix, ngauss, psf, npsf, conv,
                                              sky_in, fzw, out, status, s);
        break;
    default:
        em_wave_launch_ng_hi<NGMIX_EM_FLUXONLY>(conf, b, gmix, ngauss, psf, npsf, conv,
                                                sky_in, fzw, out, status, s);
        break;
    }
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

}  // namespace ngmix
