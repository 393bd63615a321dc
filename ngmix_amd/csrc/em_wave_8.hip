// em_wave_8.hip -- the fused EM kernels (em_wave_impl.hpp) for 7 and 8 object
// gaussians on one or two waves per stamp (<= 2048 pixels, 2304 in the full
// run): the reference's em_run is general in the gaussian count
// (em_nb.py:160-246,284-354); larger problems run the generic kernel of em.hip.
// A translation unit of its own so that it compiles beside the others.
#include "em_wave_impl.hpp"

namespace ngmix {

template <int KIND, int NG>
static void em_wave_launch_8(const ngmix_em_conf *conf, const ngmix_batch *b,
                             ngmix_gauss2d *gmix, ngmix_gauss2d *psf, int npsf,
                             ngmix_gauss2d *conv, const double *sky_in, int fzw,
                             double *out, int32_t *status, hipStream_t s)
{
    const int np = b->max_npix;
    if (np <= 16 * WAVE)
        em_wave_launch_nt<WAVE, 16, KIND, NG>(conf, b, gmix, psf, npsf, conv, sky_in, fzw,
                                              out, status, s);
    else if (KIND == NGMIX_EM_FULL && np > 16 * 2 * WAVE)
        em_wave_launch_nt<2 * WAVE, (KIND == NGMIX_EM_FULL ? 18 : 16), KIND, NG>(
            conf, b, gmix, psf, npsf, conv, sky_in, fzw, out, status, s);
    else
        em_wave_launch_nt<2 * WAVE, 16, KIND, NG>(conf, b, gmix, psf, npsf, conv, sky_in,
                                                  fzw, out, status, s);
}

template <int KIND>
static void em_wave_launch_ng_8(const ngmix_em_conf *conf, const ngmix_batch *b,
                                ngmix_gauss2d *gmix, int ngauss, ngmix_gauss2d *psf,
                                int npsf, ngmix_gauss2d *conv, const double *sky_in,
                                int fzw, double *out, int32_t *status, hipStream_t s)
{
    if (ngauss == 7)
        em_wave_launch_8<KIND, 7>(conf, b, gmix, psf, npsf, conv, sky_in, fzw, out, status, s);
    else
        em_wave_launch_8<KIND, 8>(conf, b, gmix, psf, npsf, conv, sky_in, fzw, out, status, s);
}

int launch_em_wave_8(int kind, const ngmix_em_conf *conf, const ngmix_batch *b,
                     ngmix_gauss2d *gmix, int ngauss, ngmix_gauss2d *psf, int npsf,
                     ngmix_gauss2d *conv, const double *sky_in, int fzw, double *out,
                     int32_t *status, hipStream_t s)
{
    const int np_max = kind == NGMIX_EM_FULL ? 18 * 2 * WAVE : 16 * 2 * WAVE;
    if (ngauss < 7 || ngauss > 8 || b->max_npix > np_max) {
        set_last_error_msg("launch_em_wave_8: 7..8 gaussians on stamps of <= 2048 pixels "
                           "(2304 for the full run)");
        return NGMIX_ERR_BAD_ARG;
    }
    switch (kind) {
    case NGMIX_EM_FULL:
        em_wave_launch_ng_8<NGMIX_EM_FULL>(conf, b, gmix, ngauss, psf, npsf, conv, sky_in,
                                           fzw, out, status, s);
        break;
    case NGMIX_EM_FIXCEN:
        em_wave_launch_ng_8<NGMIX_EM_FIXCEN>(conf, b, gmix, ngauss, psf, npsf, conv,
                                             sky_in, fzw, out, status, s);
        break;
    case NGMIX_EM_FIXCOV:
        em_wave_launch_ng_8<NGMIX_EM_FIXCOV>(conf, b, gmix, ngauss, psf, npsf, conv,
                                             sky_in, fzw, out, status, s);
        break;
    default:
        em_wave_launch_ng_8<NGMIX_EM_FLUXONLY>(conf, b, gmix, ngauss, psf, npsf, conv,
                                               sky_in, fzw, out, status, s);
        break;
    }
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

}  // namespace ngmix
