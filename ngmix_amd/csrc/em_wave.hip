// em_wave.hip -- the fused EM kernels (em_wave_impl.hpp) for 1 .. 3 object
// gaussians and the dispatch of launch_em_wave; 4 .. 6 are em_wave_hi.hip.
#include "em_wave_impl.hpp"

namespace ngmix {

template <int KIND>
static void em_wave_launch_ng(const ngmix_em_conf *conf, const ngmix_batch *b,
                              ngmix_gauss2d *gmix, int ngauss, ngmix_gauss2d *psf,
                              int npsf, ngmix_gauss2d *conv, const double *sky_in,
                              int fzw, double *out, int32_t *status, hipStream_t s)
{
    if (ngauss == 1)
        em_wave_launch<KIND, 1>(conf, b, gmix, psf, npsf, conv, sky_in, fzw, out, status, s);
    else if (ngauss == 2)
        em_wave_launch<KIND, 2>(conf, b, gmix, psf, npsf, conv, sky_in, fzw, out, status, s);
    else
        em_wave_launch<KIND, 3>(conf, b, gmix, psf, npsf, conv, sky_in, fzw, out, status, s);
}

// stamps of <= 16*256 pixels (64x64) with 1..3 object gaussians; 4..6 (stamps
// of <= 18*128 pixels) are em_wave_hi.hip's
int launch_em_wave(int kind, const ngmix_em_conf *conf, const ngmix_batch *b,
                            ngmix_gauss2d *gmix, int ngauss, ngmix_gauss2d *psf,
                            int npsf, ngmix_gauss2d *conv, const double *sky_in,
                            int fzw, double *out, int32_t *status, hipStream_t s)
{
    if (ngauss > 3)
        return launch_em_wave_hi(kind, conf, b, gmix, ngauss, psf, npsf, conv, sky_in, fzw,
                                 out, status, s);
    switch (kind) {
    case NGMIX_EM_FULL:
        em_wave_launch_ng<NGMIX_EM_FULL>(conf, b, gmix, ngauss, psf, npsf, conv, sky_in,
                                         fzw, out, status, s);
        break;
    case NGMIX_EM_FIXCEN:
        em_wave_launch_ng<NGMIX_EM_FIXCEN>(conf, b, gmix, ngauss, psf, npsf, conv,
                                           sky_in, fzw, out, status, s);
        break;
    case NGMIX_EM_FIXCOV:
        em_wave_launch_ng<NGMIX_EM_FIXCOV>(conf, b, gmix, ngauss, psf, npsf, conv,
                                           sky_in, fzw, out, status, s);
        break;
    default:
        em_wave_launch_ng<NGMIX_EM_FLUXONLY>(conf, b, gmix, ngauss, psf, npsf, conv,
                                             sky_in, fzw, out, status, s);
        break;
    }
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

}  // namespace ngmix
