// iter_common.hpp -- building blocks of the moment / iterative kernels
// (weighted sums, admom, em): where a stamp's pixels come from, how they are
// kept on chip across passes, and fixed-order work-group reductions.
#pragma once

#include "device_utils.hpp"

namespace ngmix {

// ---- pixel sources ---------------------------------------------------------
// A source enumerates the positions 0..count()-1 of a stamp in the reference's
// pixel order; load() returns false for positions the reference's pixel list
// does not contain (weight <= 0 with ignore_zero_weight).

// compact grid stamp: val/ierr arrays + jacobian (batch forms)
struct GridSrc {
    const double *val;
    const double *ierr;
    ngmix_jacobian jac;
    double area;
    int nrow, ncol;
    bool izw;

    __device__ __forceinline__ int count() const { return nrow * ncol; }
    __device__ __forceinline__ bool load(int p, double &v, double &u, double &a,
                                         double &pval, double &pierr) const
    {
        const int row = p / ncol, col = p - row * ncol;
        jacobian_vu(jac, (double)row, (double)col, v, u);
        a = area;
        pval = val[p];
        pierr = ierr[p];
        return !izw || pierr > 0.0;
    }
};

// the reference's AoS pixel list (seam forms)
struct ListSrc {
    const ngmix_pixel *pix;
    int n;

    __device__ __forceinline__ int count() const { return n; }
    __device__ __forceinline__ bool load(int p, double &v, double &u, double &a,
                                         double &pval, double &pierr) const
    {
        const ngmix_pixel q = pix[p];
        v = q.v;
        u = q.u;
        a = q.area;
        pval = q.val;
        pierr = q.ierr;
        return true;
    }
};

// ---- on-chip pixel cache ---------------------------------------------------
// PPT > 0: each thread keeps its PPT pixels (position tid + k*NT) in
// registers for the whole kernel, so an iterative algorithm reads the stamp
// from HBM exactly once.  PPT == 0: streaming, every pass re-loads through
// the source (list mode, or stamps too large to keep in registers).
template <class Src, int NT, int PPT>
struct PixCache {
    double v[PPT > 0 ? PPT : 1], u[PPT > 0 ? PPT : 1];
    double val[PPT > 0 ? PPT : 1], ierr[PPT > 0 ? PPT : 1];
    double area;
    unsigned kept;  // bit k: pixel k of this thread is in the pixel list

    __device__ __forceinline__ void fill(const Src &src)
    {
        kept = 0u;
        area = 0.0;
        if (PPT > 0) {
            const int n = src.count();
#pragma unroll
            for (int k = 0; k < PPT; k++) {
                const int p = threadIdx.x + k * NT;
                v[k] = u[k] = val[k] = ierr[k] = 0.0;
                if (p < n) {
                    double a;
                    if (src.load(p, v[k], u[k], a, val[k], ierr[k])) {
                        kept |= 1u << k;
                        area = a;
                    }
                }
            }
        }
    }

    // f(v, u, area, val, ierr, position)
    template <class F>
    __device__ __forceinline__ void for_each(const Src &src, F &&f) const
    {
        if (PPT > 0) {
#pragma unroll
            for (int k = 0; k < PPT; k++) {
                if (kept & (1u << k))
                    f(v[k], u[k], area, val[k], ierr[k], (int)(threadIdx.x + k * NT));
            }
        } else {
            const int n = src.count();
            for (int p = threadIdx.x; p < n; p += NT) {
                double pv, pu, pa, pval, pierr;
                if (src.load(p, pv, pu, pa, pval, pierr))
                    f(pv, pu, pa, pval, pierr, p);
            }
        }
    }
};

// ---- reductions ------------------------------------------------------------
// Sum NV per-thread doubles over the NT-thread work-group in a fixed order;
// the totals land in out[0..NV) (LDS), visible to every thread on return.
// scratch: (NT/64)*NV doubles of LDS.
template <int NT, int NV>
__device__ __forceinline__ void group_sum(double (&vals)[NV], double *scratch,
                                          double *out)
{
    constexpr int NW = NT / WAVE;
#pragma unroll
    for (int i = 0; i < NV; i++) vals[i] = wave_sum(vals[i]);
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6;
    if (NW == 1) {
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NV; i++) out[i] = vals[i];
        }
        __syncthreads();
    } else {
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NV; i++) scratch[w * NV + i] = vals[i];
        }
        __syncthreads();
        if (threadIdx.x < NV) {
            double s = scratch[threadIdx.x];
#pragma unroll
            for (int k = 1; k < NW; k++) s += scratch[k * NV + threadIdx.x];
            out[threadIdx.x] = s;
        }
        __syncthreads();
    }
}

template <int NT>
__device__ __forceinline__ int group_max_int(int x, int *scratch)
{
    constexpr int NW = NT / WAVE;
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) {
        const int y = __shfl_down(x, off, WAVE);
        x = y > x ? y : x;
    }
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6;
    if (lane == 0) scratch[w] = x;
    __syncthreads();
    int m = scratch[0];
#pragma unroll
    for (int k = 1; k < NW; k++) m = scratch[k] > m ? scratch[k] : m;
    __syncthreads();
    return m;
}

}  // namespace ngmix
