// HBM streaming micro-benchmark for the access patterns of the pixel-pass
// kernels: read-only and read-modify-write, 8 or 16 bytes per lane, linear or
// 8-row tile shaped (row stride = 48 doubles).  hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); exit(1);} } while (0)

constexpr int NSTAMP_PIX = 48 * 48;

// mode: 0 linear 8B, 1 linear 16B, 2 tile 8x8 8B (64 B rows), 3 tile 8x16 16B (128 B rows)
// each wave owns one stamp of 2304 doubles and walks it with PF loads in flight
template <int MODE, bool RMW, int PF>
__global__ __launch_bounds__(64) void k(const double *__restrict__ a, double *b, double *sink)
{
    const int lane = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * NSTAMP_PIX;
    const double *src = (RMW ? b : a) + base;
    double acc = 0.0;
    if (MODE == 0 || MODE == 2) {
        // 36 steps of 64 doubles
        auto off = [&](int t) -> int {
            if (MODE == 0) return t * 64 + lane;
            const int ty = t / 6, tx = t - ty * 6;
            return (ty * 8 + (lane >> 3)) * 48 + tx * 8 + (lane & 7);
        };
        double r[PF];
#pragma unroll
        for (int i = 0; i < PF; i++) r[i] = src[off(i)];
        for (int t = 0; t < 36; t += PF) {
#pragma unroll
            for (int i = 0; i < PF; i++) {
                const double v = r[i];
                if (t + i + PF < 36) r[i] = src[off(t + i + PF)];
                if (RMW) b[base + off(t + i)] = v + 1.0; else acc += v;
            }
        }
    } else {
        // 18 steps of 128 doubles (double2 per lane)
        auto off = [&](int t) -> int {
            if (MODE == 1) return t * 128 + lane * 2;
            const int ty = t / 3, tx = t - ty * 3;
            return (ty * 8 + (lane >> 3)) * 48 + tx * 16 + (lane & 7) * 2;
        };
        double2 r[PF];
#pragma unroll
        for (int i = 0; i < PF; i++) r[i] = *(const double2 *)(src + off(i));
        for (int t = 0; t < 18; t += PF) {
#pragma unroll
            for (int i = 0; i < PF; i++) {
                const double2 v = r[i];
                if (t + i + PF < 18) r[i] = *(const double2 *)(src + off(t + i + PF));
                if (RMW) { double2 w; w.x = v.x + 1.0; w.y = v.y + 1.0; *(double2 *)(b + base + off(t + i)) = w; }
                else acc += v.x + v.y;
            }
        }
    }
    if (!RMW && acc == 12345.678) sink[0] = acc;
}

static int g_reps = 10;

template <int MODE, bool RMW, int PF>
void run(const char *name, const double *a, double *b, double *sink, int n)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < (g_reps > 50 ? 100 : 3); i++) hipLaunchKernelGGL((k<MODE, RMW, PF>), dim3(n), dim3(64), 0, 0, a, b, sink);
    CHECK(hipEventRecord(e0));
    const int reps = g_reps;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k<MODE, RMW, PF>), dim3(n), dim3(64), 0, 0, a, b, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double bytes = (double)n * NSTAMP_PIX * 8 * (RMW ? 2 : 1);
    printf("%-28s PF=%d  %.4f ms  %.2f TB/s\n", name, PF, ms, bytes / ms / 1e9);
}

int main(int argc, char **argv)
{
    // reps per pattern: 10 measures a cold GPU, 300 the settled clocks
    if (argc > 1) g_reps = atoi(argv[1]);
    const int n = 100000;
    double *a, *b, *sink;
    CHECK(hipMalloc(&a, (size_t)n * NSTAMP_PIX * 8));
    CHECK(hipMalloc(&b, (size_t)n * NSTAMP_PIX * 8));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(a, 0, (size_t)n * NSTAMP_PIX * 8));
    CHECK(hipMemset(b, 0, (size_t)n * NSTAMP_PIX * 8));
    run<0, false, 3>("read linear 8B", a, b, sink, n);
    run<1, false, 3>("read linear 16B", a, b, sink, n);
    run<2, false, 3>("read tile8x8 8B", a, b, sink, n);
    run<3, false, 3>("read tile8x16 16B", a, b, sink, n);
    run<2, false, 6>("read tile8x8 8B", a, b, sink, n);
    run<3, false, 6>("read tile8x16 16B", a, b, sink, n);
    run<0, true, 3>("rmw linear 8B", a, b, sink, n);
    run<1, true, 3>("rmw linear 16B", a, b, sink, n);
    run<2, true, 3>("rmw tile8x8 8B", a, b, sink, n);
    run<3, true, 3>("rmw tile8x16 16B", a, b, sink, n);
    run<2, true, 6>("rmw tile8x8 8B", a, b, sink, n);
    run<3, true, 6>("rmw tile8x16 16B", a, b, sink, n);
    return 0;
}
