// device_utils.hpp -- wave64 / work-group helpers for the gfx950 kernels.
#pragma once

#include "common.hpp"

namespace ngmix {

constexpr int WAVE = 64;     // CDNA wavefront
constexpr int BLOCK = 256;   // 4 waves: one per SIMD of a CU
constexpr int NWAVES = BLOCK / WAVE;

// tile of pixels owned by one wave at a time: 8 columns x 8 rows.  A small
// gaussian's chi2<25 region misses most tiles entirely; square tiles give the
// fewest active (tile, gaussian) pairs (35% vs 38% for 16x4 on the C2
// workload) and the four waves of a work-group still sweep adjacent tiles, so
// every 128-byte line is consumed by two waves at the same time (L2 hit)
constexpr int TILE_W = 8;
constexpr int TILE_H = 8;

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }
// the wave index is uniform across a wave: tell the compiler so that everything
// derived from it (tile coordinates, loop control) stays in scalar registers
__device__ __forceinline__ int wave_id()
{
    return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}

// butterfly-free, fixed-order wave reduction: after the call lane 0 holds the
// sum of all 64 lanes (deterministic order; no atomics anywhere)
__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) x += __shfl_down(x, off, WAVE);
    return x;
}

// ---- the fused fexp evaluator (pixpass.hip, moments.hip, em.hip, lmfit.hip) ------------------------
// exp5_smooth coefficients c0..c5 (fastexp_nb.py:252-258) followed by the
// apodisation constants 10, -15, 6 and (the window in b = 0.8 u, pixpass.hip)
// -0.32, 1/15, 9.375/0.512: read with scalar loads so that they live
// in SGPRs and every Horner step is a single v_fma_f64 v, v, v, s
#define NGMIX_FEXP_COEF                                                      \
    {1.0000011318561302,  0.999993601071577,    0.49992478810274166,        \
     0.16674612720799442, 0.042330947141114836, 0.008197933236258961,       \
     10.0, -15.0, 6.0, -0.32, 1.0 / 15.0, 9.375 / 0.512}

struct FexpCoef {
    double c0, c1, c2, c3, c4, c5, w10, wm15, w6, wb, wq, wk;
    int c5lo, c5hi, w6lo, w6hi;
};

// coef: the translation unit's __constant__ copy of NGMIX_FEXP_COEF
__device__ __forceinline__ FexpCoef load_fexp_coef(const double *coef)
{
    // the index is opaque to the compiler (always 0), so the values stay
    // loaded SGPRs instead of being re-materialised as literals in VGPRs
    const double *c = coef + __builtin_amdgcn_readfirstlane(blockIdx.x >> 31);
    FexpCoef k;
    k.c0 = c[0]; k.c1 = c[1]; k.c2 = c[2]; k.c3 = c[3]; k.c4 = c[4]; k.c5 = c[5];
    k.w10 = c[6]; k.wm15 = c[7]; k.w6 = c[8];
    k.wb = c[9]; k.wq = c[10]; k.wk = c[11];
    // a VOP3 instruction reads at most one SGPR pair: the multiplicands of the
    // two-constant steps live in VGPRs (opaque moves, so they stay there)
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3"
                 : "=v"(k.c5lo), "=v"(k.c5hi)
                 : "s"(__double2loint(c[5])), "s"(__double2hiint(c[5])));
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3"
                 : "=v"(k.w6lo), "=v"(k.w6hi)
                 : "s"(__double2loint(c[8])), "s"(__double2hiint(c[8])));
    return k;
}

// fexp(-y) for 0 <= y < 12.5, y = chi2/2 (fastexp_nb.py:223-262).
// n = round-to-nearest(y) comes out of the low word of y + 1.5*2^52; the
// reference takes ival = trunc(-y - 0.5) = -n except on exact ties
// y = k + 0.5 with k even, where it uses the neighbouring cell of its
// C2-continuous piecewise polynomial (a 1-ulp difference).  tabr[n] = exp(-n).
__device__ __forceinline__ double fexp_neg_fused(double y, const double *tabr,
                                                 const FexpCoef &k)
{
    constexpr double MAGIC = 6755399441055744.0;  // 1.5 * 2^52
    const double t = y + MAGIC;
    const int n = __double2loint(t);
    const double nd = t - MAGIC;
    const double f = nd - y;  // = x - ival of the reference, x = -y
    const double tv = tabr[n];
    double p = fma(f, __hiloint2double(k.c5hi, k.c5lo), k.c4);
    p = fma(f, p, k.c3);
    p = fma(f, p, k.c2);
    p = fma(f, p, k.c1);
    p = fma(f, p, k.c0);
    return tv * p;
}

// 1/x by v_rcp_f64 and two Newton steps: within ~1 ulp of the IEEE quotient at
// a fifth of its instruction count (the fused kernels' results are "to
// rounding"; the exact kernels divide)
__device__ __forceinline__ double rcp_newton(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// ln(x) in ~45 straight-line VALU instructions (the library's is ~80): x = m 2^e
// with m in [sqrt(1/2), sqrt(2)), s = (m - 1) / (m + 1), ln m = 2 s (1 + z/3 +
// ... + z^10/21), z = s^2 <= 0.0295 (the first dropped term is 1e-18 relative),
// ln x = e ln2_hi + (e ln2_lo + ln m).  Within 2 ulp of the correctly rounded
// value over the whole positive range, subnormals included
// (tools/microbench/log_accuracy.hip); log(0) = -inf, log(x < 0) = log(nan) =
// nan, log(inf) = inf as the library's, by selects -- no branch, no call.  Used
// where a logarithm feeds a convergence test, not a per-pixel result.
__device__ __forceinline__ double log_fast(double x)
{
    const bool tiny = x < 2.2250738585072014e-308;           // subnormal (or <= 0)
    const double xs = tiny ? x * 18014398509481984.0 : x;    // 2^54
    double m = __builtin_amdgcn_frexp_mant(xs);              // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(xs) - (tiny ? 54 : 0);
    const bool low = m < 0.70710678118654752;
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    const double a = m - 1.0, b = m + 1.0;       // both exact
    const double r = rcp_newton(b);
    double sq = a * r;
    sq = fma(fma(-b, sq, a), r, sq);             // a / b to rounding
    const double z = sq * sq;
    double p = 1.0 / 21.0;
    p = fma(p, z, 1.0 / 19.0);
    p = fma(p, z, 1.0 / 17.0);
    p = fma(p, z, 1.0 / 15.0);
    p = fma(p, z, 1.0 / 13.0);
    p = fma(p, z, 1.0 / 11.0);
    p = fma(p, z, 1.0 / 9.0);
    p = fma(p, z, 1.0 / 7.0);
    p = fma(p, z, 1.0 / 5.0);
    p = fma(p, z, 1.0 / 3.0);
    // ln m = 2 s + 2 s z p
    const double s2 = sq + sq;
    const double lnm = fma(s2 * z, p, s2);
    const double ed = (double)e;
    // ln 2 split: the high part has 32 trailing zero bits, e * hi is exact
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    double res = fma(ed, LN2_HI, fma(ed, LN2_LO, lnm));
    res = x == INFINITY ? INFINITY : res;
    res = x == 0.0 ? -INFINITY : res;
    res = (x < 0.0 || x != x) ? NAN : res;
    return res;
}

// ---- DPP wave reductions ------------------------------------------------------
// The sum of x over the 64 lanes of a wave, returned uniform (in every lane):
// an inclusive scan inside each row of 16 lanes (row_shr 1,2,4,8), row
// broadcasts 15 and 31 to fold the four rows, then a readlane of lane 63.
// All VALU (v_mov_b32 dpp + v_add_f64): no LDS traffic, no waitcnt, fixed order.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move_or_zero(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double readlane_f64(double x, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

// sum over each row of 16 lanes, valid in the row's last lane (4 DPP steps
// with zero fill, no initialisation moves)
template <int CTRL>
__device__ __forceinline__ double dpp_row_shr_bc(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double row16_total(double x)
{
    x += dpp_row_shr_bc<0x111>(x);
    x += dpp_row_shr_bc<0x112>(x);
    x += dpp_row_shr_bc<0x114>(x);
    x += dpp_row_shr_bc<0x118>(x);
    return x;
}

__device__ __forceinline__ double wave_total(double x)
{
    x += dpp_move_or_zero<0x111, 0xf>(x);  // row_shr:1
    x += dpp_move_or_zero<0x112, 0xf>(x);  // row_shr:2
    x += dpp_move_or_zero<0x114, 0xf>(x);  // row_shr:4
    x += dpp_move_or_zero<0x118, 0xf>(x);  // row_shr:8
    x += dpp_move_or_zero<0x142, 0xa>(x);  // row_bcast:15 -> rows 1, 3
    x += dpp_move_or_zero<0x143, 0xc>(x);  // row_bcast:31 -> rows 2, 3
    return readlane_f64(x, 63);
}

// ---- four sums per register: permlane swaps ----------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_row_shr_zero(double x)
{
    // row_shr within each row of 16 lanes; lanes without a source read 0
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// sum of x over each row of 16 lanes, valid in the row's last lane
__device__ __forceinline__ double row16_sum(double x)
{
    x += dpp_row_shr_zero<0x111>(x);
    x += dpp_row_shr_zero<0x112>(x);
    x += dpp_row_shr_zero<0x114>(x);
    x += dpp_row_shr_zero<0x118>(x);
    return x;
}

// x, y hold one partial per lane of two sums; rows of 16 lanes r0..r3.
// permlane16_swap: x = [x_r0, y_r0, x_r2, y_r2], y = [x_r1, y_r1, x_r3, y_r3]
// -> x + y = [x_r0 + x_r1, y_r0 + y_r1, x_r2 + x_r3, y_r2 + y_r3]
__device__ __forceinline__ double swap_add16(double x, double y)
{
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(y),
                                                     false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(y),
                                                     false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}

// permlane32_swap: x = [x_lo, y_lo], y = [x_hi, y_hi] by halves of 32 lanes
// -> x + y = [x_lo + x_hi, y_lo + y_hi]
__device__ __forceinline__ double swap_add32(double x, double y)
{
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(y),
                                                     false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(y),
                                                     false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}

// The sums of four per-lane values over the wave, uniform in every lane:
// v_permlane16_swap / v_permlane32_swap trade rows of 16 / 32 lanes between
// two registers, so one add folds the rows of TWO sums -- the four collapse
// into one register whose row w holds the 16 lane-partials of value w, one
// DPP row sum finishes all four (21 instructions), two readlanes each bring
// them to SGPRs: 7 instructions per sum against 20 for four wave_total().
// A fixed order (not wave_total's: results differ by summation order only).
__device__ __forceinline__ void wave_total4(double &a, double &b, double &c, double &d)
{
    const double ab = swap_add16(a, b);
    const double cd = swap_add16(c, d);
    const double t = row16_sum(swap_add32(ab, cd));
    a = readlane_f64(t, 15);
    b = readlane_f64(t, 31);
    c = readlane_f64(t, 47);
    d = readlane_f64(t, 63);
}

// ---- many sums per wave through a transposed LDS tile ------------------------
// The sum of each of the NV per-lane values over the 64 lanes of a ONE-WAVE
// work-group, left in tot[k] (LDS): lane (k, j) adds the j-th segment of row k
// of the tile, the segments are then folded with log2 shuffles.  Fixed order.
// ~NVP + 3 log2(64/NVP) VALU instructions in all, against ~34 per value for
// a DPP tree; red: NV * WAVE_RED_STRIDE doubles, tot: NV doubles.
constexpr int WAVE_RED_STRIDE = 66;  // doubles per row (spreads the banks)

template <int NV>
__device__ __forceinline__ void wave_reduce_lds(const double (&acc)[NV], double *red,
                                                double *tot)
{
    constexpr int NVP = NV <= 2 ? 2 : NV <= 4 ? 4 : NV <= 8 ? 8 : NV <= 16 ? 16
                        : NV <= 32 ? 32 : 64;
    constexpr int SEGS = WAVE / NVP;   // lanes per value
    constexpr int SEGLEN = WAVE / SEGS;
    static_assert(NV <= 64, "wave_reduce_lds: too many sums");
    const int lane = threadIdx.x & (WAVE - 1);
#pragma unroll
    for (int k = 0; k < NV; k++) red[k * WAVE_RED_STRIDE + lane] = acc[k];
    __syncthreads();
    const int k = lane % NVP, j = lane / NVP;
    double s = 0.0;
    if (k < NV) {
        const double *row = red + k * WAVE_RED_STRIDE + j * SEGLEN;
#pragma unroll
        for (int i = 0; i < SEGLEN; i++) s += row[i];
    }
#pragma unroll
    for (int off = SEGS / 2; off > 0; off >>= 1) s += __shfl_down(s, off * NVP, WAVE);
    if (j == 0 && k < NV) tot[k] = s;
    __syncthreads();
}

__device__ __forceinline__ int wave_sum_int(int x)
{
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) x += __shfl_down(x, off, WAVE);
    return x;
}

// Sum NV doubles held per thread over the whole 256-thread work-group.
// scratch: NWAVES*NV doubles of LDS.  Result valid in thread 0 (vals[]).
template <int NV>
__device__ __forceinline__ void block_sum(double (&vals)[NV], double *scratch)
{
#pragma unroll
    for (int i = 0; i < NV; i++) vals[i] = wave_sum(vals[i]);
    const int lane = lane_id(), w = wave_id();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; i++) scratch[w * NV + i] = vals[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; i++) {
            double s = scratch[i];
#pragma unroll
            for (int k = 1; k < NWAVES; k++) s += scratch[k * NV + i];
            vals[i] = s;
        }
    }
}

// Conservative pixel-index bounding box of the region where a gaussian's
// chi2 can be < 25.  Outside it every evaluation of gauss2d_eval_pixel_fast
// returns exactly 0.0 (gmix_nb.py:46,58), and x + 0.0 == x, so skipping those
// pixel-gaussian pairs leaves results bit-identical.  Derived from the very
// coefficients the evaluation uses (dcc, drr, drc), inflated for rounding;
// any doubt (non positive-definite form, near-degenerate correlation or
// jacobian, non-finite input) returns the "everything" box.
struct PixBox {
    int rmin, rmax, cmin, cmax;
};

__device__ __forceinline__ PixBox full_box()
{
    PixBox b;
    b.rmin = -(1 << 30);
    b.cmin = -(1 << 30);
    b.rmax = (1 << 30);
    b.cmax = (1 << 30);
    return b;
}

__device__ __forceinline__ PixBox gauss_pixel_box(const ngmix_gauss2d &g,
                                                 const ngmix_jacobian &j)
{
    PixBox full = full_box();
    const double dcc = g.dcc, drr = g.drr, drc = g.drc;
    // chi2 = dcc dv^2 + drr du^2 - 2 drc dv du
    const double detq = dcc * drr - drc * drc;
    if (!(dcc > 0.0) || !(drr > 0.0) || !(detq > 0.0)) return full;
    // rho^2 < 1 - 1e-6, without the division
    if (!(drc * drc < (1.0 - 1.0e-6) * (dcc * drr))) return full;
    // The box only has to be CONSERVATIVE, so its arithmetic does not need
    // IEEE divisions / square roots (~25 instructions each, paid by the whole
    // wave for the few lanes that stage gaussians): v_rcp_f64 / v_rsq_f64 with
    // one Newton step are good to ~1e-10 relative, far inside the 1e-6
    // inflation below.
    auto rcp = [](double x) {
        double r = __builtin_amdgcn_rcp(x);
        return fma(fma(-x, r, 1.0), r, r);
    };
    auto sqrt_fast = [](double x) {
        if (!(x > 0.0)) return 0.0;
        double r = __builtin_amdgcn_rsq(x);          // ~1/sqrt(x)
        double s = x * r;                            // ~sqrt(x)
        return fma(fma(-s, s, x), 0.5 * r, s);       // one Newton step
    };
    // covariance of the form: [[var_v, cov],[cov, var_u]]
    const double idetq = rcp(detq);
    const double var_v = drr * idetq, var_u = dcc * idetq, cov = drc * idetq;
    // pixel = Jinv (v,u):  dr = ( d*v - b*u)/det ; dc = (-c*v + a*u)/det
    const double a = j.dvdrow, b = j.dvdcol, c = j.dudrow, d = j.dudcol;
    const double det = a * d - b * c;
    const double jn = a * a + b * b + c * c + d * d;
    if (!(fabs(det) > 1.0e-6 * jn) || !(jn > 0.0)) return full;
    const double idet = rcp(det);
    const double rr = d * idet, ru = -b * idet;   // dr = rr*v + ru*u
    const double cr = -c * idet, cu = a * idet;   // dc = cr*v + cu*u
    const double var_r = rr * rr * var_v + 2.0 * rr * ru * cov + ru * ru * var_u;
    const double var_c = cr * cr * var_v + 2.0 * cr * cu * cov + cu * cu * var_u;
    const double cen_r = j.row0 + (rr * g.row + ru * g.col);
    const double cen_c = j.col0 + (cr * g.row + cu * g.col);
    if (!(var_r >= 0.0) || !(var_c >= 0.0)) return full;
    // A pixel (integer row r) can have chi2 < 25 only if |r - cen_r| <=
    // 5 sigma_r.  Inflate by 1e-6 relative (chi2 at the edge is then
    // >= 25(1+2e-6), far above the <=1e-9 relative rounding of the evaluated
    // chi2 given rho2 < 1-1e-6) plus 1e-6 pixel absolute for the rounding of
    // the centre and pixel coordinates; the integer pixels inside
    // [lo, hi] are ceil(lo) .. floor(hi).
    const double hr = 5.0 * sqrt_fast(var_r) * (1.0 + 1.0e-6) + 1.0e-6;
    const double hc = 5.0 * sqrt_fast(var_c) * (1.0 + 1.0e-6) + 1.0e-6;
    const double lo_r = cen_r - hr, hi_r = cen_r + hr;
    const double lo_c = cen_c - hc, hi_c = cen_c + hc;
    const double big = 1.0e9;
    if (!(lo_r > -big) || !(hi_r < big) || !(lo_c > -big) || !(hi_c < big))
        return full;
    PixBox box;
    box.rmin = (int)ceil(lo_r);
    box.rmax = (int)floor(hi_r);
    box.cmin = (int)ceil(lo_c);
    box.cmax = (int)floor(hi_c);
    return box;
}

// The chi2 < 25 box of a gaussian as a tile test reads it: a TH x TW tile whose
// first pixel is (r0, c0) reaches the box when r0 is in [rmin - (TH-1), rmax],
// i.e. (unsigned)(r0 - r_lo) <= r_span -- one subtraction and one compare per
// axis (the four signed compares of the plain form came out of the compiler as
// 16-bit mask arithmetic: 20 instructions).
struct TileBox {
    int r_lo;
    unsigned r_span;
    int c_lo;
    unsigned c_span;
};

__device__ __forceinline__ TileBox tile_box(const PixBox &pb, int th, int tw)
{
    // (|rmin|, |rmax| <= 2^30: the spans fit an unsigned; an inverted box
    // would wrap to a huge span, so it is stored as one no tile reaches)
    TileBox tb;
    tb.r_lo = pb.rmin - (th - 1);
    tb.c_lo = pb.cmin - (tw - 1);
    const bool none = pb.rmax < tb.r_lo || pb.cmax < tb.c_lo;
    tb.r_span = none ? 0u : (unsigned)pb.rmax - (unsigned)tb.r_lo;
    tb.c_span = none ? 0u : (unsigned)pb.cmax - (unsigned)tb.c_lo;
    if (none) tb.r_lo = 1 << 30;
    return tb;
}

// lanes whose box is reached by the tile starting at (r0, c0): the two compares
// write their lane masks straight into scalar registers (a ballot of a combined
// bool costs a v_cndmask + v_cmp on top).  EXEC must be the whole wave.
__device__ __forceinline__ unsigned long long tile_hits(const TileBox &b, int r0, int c0)
{
    unsigned long long mr, mc;
    asm("v_cmp_le_u32 %0, %1, %2" : "=s"(mr) : "v"((unsigned)(r0 - b.r_lo)), "v"(b.r_span));
    asm("v_cmp_le_u32 %0, %1, %2" : "=s"(mc) : "v"((unsigned)(c0 - b.c_lo)), "v"(b.c_span));
    return mr & mc;
}

// Row-major rank of position p among the kept pixels of a masked stamp
// (the index into the reference's pixel list): per-64-pixel keep masks and
// their exclusive prefix counts live in LDS (cmask/cpre, one entry per chunk).
template <int NT = BLOCK>
__device__ __forceinline__ void build_rank_tables(unsigned long long *cmask,
                                                  int *cpre, const double *ierr,
                                                  int npix)
{
    const int nchunks = (npix + 63) >> 6;
    for (int base = wave_id() * WAVE; base < nchunks * WAVE; base += NT) {
        const int p = base + lane_id();
        const bool kept = p < npix && ierr[p] > 0.0;
        const unsigned long long m = __ballot(kept);
        if (lane_id() == 0) cmask[base >> 6] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int c = 0; c < nchunks; c++) {
            cpre[c] = run;
            run += __popcll(cmask[c]);
        }
    }
    __syncthreads();
}

__device__ __forceinline__ int kept_rank(const unsigned long long *cmask,
                                         const int *cpre, int p)
{
    const unsigned long long m = cmask[p >> 6];
    const unsigned long long below = m & ((1ull << (p & 63)) - 1ull);
    return cpre[p >> 6] + __popcll(below);
}

// a wave-uniform double moved to SGPRs (a VOP3 instruction reads one SGPR pair
// for free; as a VGPR pair it would cost two registers across the tile loop)
__device__ __forceinline__ double uniform_f64(double x)
{
    // asm with "=s" results: the builtin readfirstlane of a value the compiler
    // already knows to be uniform folds away and leaves it in VGPRs.  The
    // s_nop covers the wait states gfx950 wants between a VALU write of a VGPR
    // and a v_readfirstlane of it: the hazard recognizer does not look inside
    // inline asm (without it the constants were stale now and then).
    int lo, hi;
    asm volatile("s_nop 4\n\tv_readfirstlane_b32 %0, %2\n\tv_readfirstlane_b32 %1, %3"
                 : "=&s"(lo), "=s"(hi)
                 : "v"(__double2loint(x)), "v"(__double2hiint(x)));
    return __hiloint2double(hi, lo);
}

// a * b + c and a * b with the wave-uniform a read from its SGPR pair (left to
// itself the compiler copies such constants into VGPRs outside the tile loop)
__device__ __forceinline__ double fma_sgpr(double a, double b, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "s"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ double mul_sgpr(double a, double b)
{
    double r;
    asm("v_mul_f64 %0, %1, %2" : "=v"(r) : "s"(a), "v"(b));
    return r;
}

// b - a and a - b with the wave-uniform a in its SGPR pair
__device__ __forceinline__ double sub_sgpr_from(double a, double b)
{
    double r;
    asm("v_add_f64 %0, -%1, %2" : "=v"(r) : "s"(a), "v"(b));
    return r;
}

__device__ __forceinline__ double sgpr_minus(double a, double b)
{
    double r;
    asm("v_add_f64 %0, %1, -%2" : "=v"(r) : "s"(a), "v"(b));
    return r;
}

}  // namespace ngmix
