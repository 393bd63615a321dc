// template.hip -- the pixel sums of the linear template (psf) flux fit:
// PSFFluxFitModel.go (ngmix/fitting/results.py:700-770) forms, per observation,
//   xcorr = sum(model * image * weight),  msq = sum(model * model * weight)
// and, with flux = xcorr / msq over the object's observations,
//   chi2 = sum((flux * model - image)^2 * weight).
// One wave per stamp, ONE pass over the three planes per call (the torch
// expressions of round 3 were ~20 passes over 1.8 GB planes): with a per-stamp
// multiplier a (the template's norm in the first call, flux * norm in the
// second), mm = a * model, the kernel returns
//   out[s] = { sum(mm I w), sum(mm mm w), sum((mm - I)^2 w), #(ierr > 0) },
// w = ierr^2 (the stamp store keeps ierr = sqrt(max(weight, 0)),
// pixels_nb.py:49-52).  Fixed-order reductions: deterministic.
#include "device_utils.hpp"
#include "launch.hpp"

namespace ngmix {

__global__ __launch_bounds__(WAVE) void template_sums_kernel(
    const ngmix_stamp *__restrict__ stamps, const double *__restrict__ model,
    const double *__restrict__ val, const double *__restrict__ ierr,
    const double *__restrict__ mult, double *__restrict__ out)
{
    const int64_t s = blockIdx.x;
    const ngmix_stamp st = stamps[s];
    const int npix = st.nrow * st.ncol;
    const double a = mult ? mult[s] : 1.0;
    const double *m = model + st.pix_off, *v = val + st.pix_off, *e = ierr + st.pix_off;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int p = threadIdx.x; p < npix; p += WAVE) {
        const double ie = e[p], I = v[p];
        const double w = ie * ie;
        const double mm = m[p] * a;
        const double d = mm - I;
        s0 += (mm * I) * w;
        s1 += (mm * mm) * w;
        s2 += (d * d) * w;
        s3 += ie > 0.0 ? 1.0 : 0.0;
    }
    s0 = wave_total(s0);
    s1 = wave_total(s1);
    s2 = wave_total(s2);
    s3 = wave_total(s3);
    if (threadIdx.x == 0) {
        double *o = out + 4 * s;
        o[0] = s0;
        o[1] = s1;
        o[2] = s2;
        o[3] = s3;
    }
}

int launch_template_sums(const ngmix_batch *b, const double *model, const double *mult,
                         double *out, hipStream_t s)
{
    if (!b || !model || !out || !b->val || !b->ierr) {
        set_last_error_msg("template_sums: batch with val / ierr, model and out are required");
        return NGMIX_ERR_BAD_ARG;
    }
    if (b->nstamps <= 0) return NGMIX_OK;
    census("template_sums_kernel");
    hipLaunchKernelGGL(template_sums_kernel, dim3((unsigned)b->nstamps), dim3(WAVE), 0, s,
                       b->stamps, model, b->val, b->ierr, mult, out);
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

}  // namespace ngmix
