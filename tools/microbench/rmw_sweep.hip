// Read-modify-write streaming on MI355X: which launch shape / access form gets
// closest to the copy rate?  The render kernel is bound by exactly this.
// hipcc --offload-arch=gfx950 -O3 -o rmw_sweep rmw_sweep.hip ; ./rmw_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int PIX = 48 * 48;   // doubles per stamp

// A: one wave per stamp, in place, 8 B per lane, linear, PF loads in flight
template <int PF, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void wave_per_stamp(double *b, const double *a)
{
    const int lane = threadIdx.x & 63;
    const size_t base = ((size_t)blockIdx.x * WAVES + (threadIdx.x >> 6)) * PIX;
    const double *src = a ? a + base : b + base;
    double r[PF];
#pragma unroll
    for (int i = 0; i < PF; i++) r[i] = src[i * 64 + lane];
    for (int t = 0; t < 36; t += PF) {
#pragma unroll
        for (int i = 0; i < PF; i++) {
            const double v = r[i];
            if (t + i + PF < 36) r[i] = src[(t + i + PF) * 64 + lane];
            b[base + (t + i) * 64 + lane] = v + 1.0;
        }
    }
}

// B: one wave per stamp, 16 B per lane (18 steps), all loads first
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void wave_per_stamp_all(double *b, const double *a)
{
    const int lane = threadIdx.x & 63;
    const size_t base = ((size_t)blockIdx.x * WAVES + (threadIdx.x >> 6)) * PIX;
    const double2 *src = (const double2 *)(a ? a + base : b + base);
    double2 *dst = (double2 *)(b + base);
    double2 r[18];
#pragma unroll
    for (int i = 0; i < 18; i++) r[i] = src[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < 18; i++) {
        double2 w = r[i];
        w.x += 1.0;
        w.y += 1.0;
        dst[i * 64 + lane] = w;
    }
}

// C: classic grid-stride, 16 B per lane, UNR independent elements per trip
template <int UNR, bool NT>
__global__ __launch_bounds__(256) void grid_stride(double *b, const double *a, size_t n2)
{
    const double2 *src = (const double2 *)(a ? a : b);
    double2 *dst = (double2 *)b;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNR - 1) * stride < n2; i += UNR * stride) {
        double2 r[UNR];
#pragma unroll
        for (int k = 0; k < UNR; k++) {
            if (NT) {
                r[k].x = __builtin_nontemporal_load(&src[i + k * stride].x);
                r[k].y = __builtin_nontemporal_load(&src[i + k * stride].y);
            } else {
                r[k] = src[i + k * stride];
            }
        }
#pragma unroll
        for (int k = 0; k < UNR; k++) {
            double2 w = r[k];
            w.x += 1.0;
            w.y += 1.0;
            if (NT) {
                __builtin_nontemporal_store(w.x, &dst[i + k * stride].x);
                __builtin_nontemporal_store(w.y, &dst[i + k * stride].y);
            } else {
                dst[i + k * stride] = w;
            }
        }
    }
    for (; i < n2; i += stride) {
        double2 w = src[i];
        w.x += 1.0;
        w.y += 1.0;
        dst[i] = w;
    }
}

// D: contiguous chunk per block (a block of 256 threads owns 4 consecutive
// stamps and walks them front to back, 16 B per lane)
template <int UNR>
__global__ __launch_bounds__(256) void chunk_per_block(double *b, const double *a)
{
    const size_t base2 = (size_t)blockIdx.x * (4 * PIX / 2);
    const double2 *src = (const double2 *)(a ? a : b) + base2;
    double2 *dst = (double2 *)b + base2;
    constexpr int STEPS = 4 * PIX / 2 / 256;   // 18
    for (int t = 0; t < STEPS; t += UNR) {
        double2 r[UNR];
#pragma unroll
        for (int k = 0; k < UNR; k++)
            if (t + k < STEPS) r[k] = src[(t + k) * 256 + threadIdx.x];
#pragma unroll
        for (int k = 0; k < UNR; k++)
            if (t + k < STEPS) {
                double2 w = r[k];
                w.x += 1.0;
                w.y += 1.0;
                dst[(t + k) * 256 + threadIdx.x] = w;
            }
    }
}

template <class F>
static void timeit(const char *name, size_t bytes, F launch)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 150; i++) launch();   // settle the clocks
    CHECK(hipDeviceSynchronize());
    float best = 1e30f, tot = 0.f;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 50; i++) launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 50;
        tot += ms;
        if (ms < best) best = ms;
    }
    CHECK(hipGetLastError());
    printf("%-44s %.4f ms (mean %.4f)  %.2f TB/s\n", name, best, tot / 5, bytes / (best * 1e-3) / 1e12);
}

int main()
{
    const size_t n = 100000;
    const size_t nd = n * PIX;
    double *a, *b;
    CHECK(hipMalloc(&a, nd * 8));
    CHECK(hipMalloc(&b, nd * 8));
    CHECK(hipMemset(a, 0, nd * 8));
    CHECK(hipMemset(b, 0, nd * 8));
    const size_t bytes = 2 * nd * 8;
    const double *nul = nullptr;
    timeit("wave/stamp in place 8B PF4", bytes, [&] { hipLaunchKernelGGL((wave_per_stamp<4, 1>), dim3(n), dim3(64), 0, 0, b, nul); });
    timeit("wave/stamp in place 8B PF6", bytes, [&] { hipLaunchKernelGGL((wave_per_stamp<6, 1>), dim3(n), dim3(64), 0, 0, b, nul); });
    timeit("wave/stamp in place 8B PF4, 4 waves/WG", bytes, [&] { hipLaunchKernelGGL((wave_per_stamp<4, 4>), dim3(n / 4), dim3(256), 0, 0, b, nul); });
    timeit("wave/stamp copy 8B PF4", bytes, [&] { hipLaunchKernelGGL((wave_per_stamp<4, 1>), dim3(n), dim3(64), 0, 0, b, a); });
    timeit("wave/stamp in place 16B all loads first", bytes, [&] { hipLaunchKernelGGL((wave_per_stamp_all<1>), dim3(n), dim3(64), 0, 0, b, nul); });
    timeit("wave/stamp in place 16B all first, 4 w/WG", bytes, [&] { hipLaunchKernelGGL((wave_per_stamp_all<4>), dim3(n / 4), dim3(256), 0, 0, b, nul); });
    timeit("wave/stamp copy 16B all loads first", bytes, [&] { hipLaunchKernelGGL((wave_per_stamp_all<1>), dim3(n), dim3(64), 0, 0, b, a); });
    for (int g : {1024, 2048, 4096, 8192}) {
        char nm[96];
        snprintf(nm, sizeof nm, "grid-stride in place 16B x4, %d WGs", g);
        timeit(nm, bytes, [&] { hipLaunchKernelGGL((grid_stride<4, false>), dim3(g), dim3(256), 0, 0, b, nul, nd / 2); });
        snprintf(nm, sizeof nm, "grid-stride copy 16B x4, %d WGs", g);
        timeit(nm, bytes, [&] { hipLaunchKernelGGL((grid_stride<4, false>), dim3(g), dim3(256), 0, 0, b, a, nd / 2); });
    }
    timeit("grid-stride in place 16B x4 nt, 2048 WGs", bytes, [&] { hipLaunchKernelGGL((grid_stride<4, true>), dim3(2048), dim3(256), 0, 0, b, nul, nd / 2); });
    timeit("grid-stride copy 16B x4 nt, 2048 WGs", bytes, [&] { hipLaunchKernelGGL((grid_stride<4, true>), dim3(2048), dim3(256), 0, 0, b, a, nd / 2); });
    timeit("grid-stride in place 16B x8, 2048 WGs", bytes, [&] { hipLaunchKernelGGL((grid_stride<8, false>), dim3(2048), dim3(256), 0, 0, b, nul, nd / 2); });
    timeit("chunk/block (4 stamps) in place x6", bytes, [&] { hipLaunchKernelGGL((chunk_per_block<6>), dim3(n / 4), dim3(256), 0, 0, b, nul); });
    timeit("chunk/block (4 stamps) in place x18", bytes, [&] { hipLaunchKernelGGL((chunk_per_block<18>), dim3(n / 4), dim3(256), 0, 0, b, nul); });
    timeit("chunk/block (4 stamps) copy x6", bytes, [&] { hipLaunchKernelGGL((chunk_per_block<6>), dim3(n / 4), dim3(256), 0, 0, b, a); });
    return 0;
}
