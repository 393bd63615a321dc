// accuracy of device_utils.hpp's log_fast against the host's log()
// hipcc --offload-arch=gfx950 -O2 -I../../ngmix_amd/csrc -o log_accuracy log_accuracy.hip && ./log_accuracy
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "device_utils.hpp"
__global__ void k(const double *x, double *r, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) r[i] = ngmix::log_fast(x[i]);
}
static double ulp_of(double v)
{
    int e;
    frexp(v, &e);
    return ldexp(1.0, e - 53);
}
int main()
{
    const int n = 1 << 22;
    double *h = (double *)malloc(n * 8), *o = (double *)malloc(n * 8);
    srand(1);
    for (int i = 0; i < n; i++) {
        const double u = (double)rand() / RAND_MAX;
        if (i % 4 == 0) h[i] = exp((u - 0.5) * 1400.0);          // the whole normal range
        else if (i % 4 == 1) h[i] = 1.0 + (u - 0.5) * 0.8;       // around 1 (cancellation)
        else if (i % 4 == 2) h[i] = exp((u - 0.5) * 40.0);       // the kernels' range
        else if (i % 8 == 3) h[i] = ldexp(0.70710678118654752 + (u - 0.5) * 1e-6, (rand() % 200) - 100);
        else h[i] = ldexp(u + 1e-9, -1023 - rand() % 51);             // subnormals
    }
    const double special[] = {0.0, -1.0, -0.0, INFINITY, -INFINITY, NAN, 1.0};
    const int nsp = sizeof(special) / 8;
    memcpy(h, special, sizeof(special));
    double *dx, *dr;
    hipMalloc(&dx, n * 8);
    hipMalloc(&dr, n * 8);
    hipMemcpy(dx, h, n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dr, n);
    hipMemcpy(o, dr, n * 8, hipMemcpyDeviceToHost);
    double worst = 0, worst_x = 0;
    for (int i = nsp; i < n; i++) {
        const long double ref = logl((long double)h[i]);
        const double err = (double)fabsl((long double)o[i] - ref) /
                           (ref != 0 ? ulp_of((double)ref) : 4.9e-324);
        if (err > worst) {
            worst = err;
            worst_x = h[i];
        }
    }
    printf("log_fast: max error %.3f ulp (at x = %.17g) over %d points\n", worst, worst_x, n - nsp);
    for (int i = 0; i < nsp; i++)
        printf("  log_fast(%g) = %g   (host log: %g)\n", h[i], o[i], log(h[i]));
    return worst < 2.0 ? 0 : 1;
}
