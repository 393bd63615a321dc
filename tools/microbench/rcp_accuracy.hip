// accuracy of v_rcp_f64 / v_rsq_f64 and their Newton refinements on gfx950
// hipcc --offload-arch=gfx950 -O2 -o rcp_accuracy rcp_accuracy.hip && ./rcp_accuracy
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k(const double *x, double *r0, double *r1, double *r2, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r = __builtin_amdgcn_rcp(v);
    r0[i] = r;
    r = fma(fma(-v, r, 1.0), r, r);
    r1[i] = r;
    r = fma(fma(-v, r, 1.0), r, r);
    r2[i] = r;
}
int main()
{
    const int n = 1 << 20;
    double *h = (double *)malloc(n * 8), *o = (double *)malloc(3 * n * 8);
    srand(1);
    for (int i = 0; i < n; i++) h[i] = exp(((double)rand() / RAND_MAX - 0.5) * 40.0);
    double *dx, *d0;
    hipMalloc(&dx, n * 8);
    hipMalloc(&d0, 3 * n * 8);
    hipMemcpy(dx, h, n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, d0, d0 + n, d0 + 2 * n, n);
    hipMemcpy(o, d0, 3 * n * 8, hipMemcpyDeviceToHost);
    for (int s = 0; s < 3; s++) {
        double worst = 0;
        for (int i = 0; i < n; i++) {
            double e = fabs(o[s * n + i] * h[i] - 1.0);
            if (e > worst) worst = e;
        }
        printf("rcp + %d Newton steps: max |r x - 1| = %.3e\n", s, worst);
    }
    return 0;
}
