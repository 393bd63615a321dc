// TEMPORARY: entry points whose kernels are still being written.
#include "common.hpp"
using namespace ngmix;
#define NOTYET { set_last_error_msg("not implemented yet"); return NGMIX_ERR_BAD_ARG; }
extern "C" {
int ngmix_get_weighted_sums(const ngmix_gauss2d *, int64_t, const ngmix_pixel *, int64_t, void *, int, double) NOTYET
int ngmix_admom(const ngmix_admom_conf *, ngmix_gauss2d *, const ngmix_pixel *, int64_t, ngmix_admom_result *) NOTYET
int ngmix_em_run(int, const ngmix_em_conf *, ngmix_pixel *, int64_t, double *, ngmix_gauss2d *, int64_t, ngmix_gauss2d *, int64_t, ngmix_gauss2d *, int, int32_t *, double *, double *) NOTYET
int ngmix_deriv_images(const double *, const double *, int64_t, const double *, const double *, const double *, int64_t, double *) NOTYET
int ngmix_weighted_sums_batch(const ngmix_batch *, const ngmix_gauss2d *, void *, int, const double *, int32_t *, void *) NOTYET
int ngmix_admom_batch(const ngmix_admom_conf *, const ngmix_batch *, ngmix_gauss2d *, ngmix_admom_result *, int32_t *, void *) NOTYET
int ngmix_em_batch(int, const ngmix_em_conf *, const ngmix_batch *, ngmix_gauss2d *, int, ngmix_gauss2d *, int, ngmix_gauss2d *, const double *, int, double *, int32_t *, void *) NOTYET
int ngmix_deriv_images_batch(const ngmix_batch *, const double *, const double *, double *, const int64_t *, void *) NOTYET
}
