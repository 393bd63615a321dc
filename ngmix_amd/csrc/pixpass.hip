// pixpass.hip -- single-pass pixel kernels: render / get_loglike / fill_fdiff /
// get_model_s2n_sum (reference: ngmix/gmix/render_nb.py:9-36,
// ngmix/gmix/gmix_nb.py:824-937), plus fill_pixels / fill_coords
// (ngmix/pixels/pixels_nb.py:6-94).
//
// GRID kernels (batch forms): one 256-thread work-group per stamp, compact
// val/ierr arrays, (v,u) recomputed from (row,col) and the 64-byte jacobian.
// Each wave owns K tiles of 8x8 pixels per round; the stamp's gaussians are
// staged in LDS together with a conservative pixel-space box of their
// chi2<25 region, and a wave skips (gaussian, tile) pairs that cannot
// intersect -- exact, because those evaluations are exactly 0.0.
// HBM-bound by design: 16 B/pixel read (+8 written for fdiff / +8+8 for
// render), everything else stays on chip.
//
// LIST kernels (seam forms): the reference's AoS pixel / coord arrays,
// arbitrary coordinates, one thread per pixel.
#include <stdlib.h>

#include <type_traits>

#include "device_utils.hpp"
#include "launch.hpp"

namespace ngmix {

__constant__ double c_exp_table[16] = NGMIX_EXP_TABLE;
// exp5_smooth coefficients c0..c5 (fastexp_nb.py:252-258) followed by the
// apodisation constants 10, -15, 6: read with scalar loads so that they live
// in SGPRs and every Horner step is a single v_fma_f64 v, v, v, s
__constant__ double c_fexp_coef[12] = NGMIX_FEXP_COEF;

enum PassOp { OP_LOGLIKE = 0, OP_FDIFF = 1, OP_RENDER_FAST = 2,
              OP_RENDER_EXACT = 3, OP_S2N = 4 };

struct GaussLds {
    EvalGauss e;   // 48 B
    PixBox box;    // 16 B
};
static_assert(sizeof(GaussLds) == 64, "GaussLds");

// LDS layout (dynamic): [exp table 16 d][reduce scratch 16 d][int ctl 4]
//                       [GaussLds x max_ngauss][chunk prefix (masked stamps)]
struct LdsLayout {
    double *tab;
    double *red;
    int *ctl;
    GaussLds *gl;
    double *tb;   // fused kernels: TileEnt records (32 B each)
    unsigned long long *cmask;
    int *cpre;
};

__device__ __forceinline__ LdsLayout carve(char *base, int max_ngauss,
                                           int nchunks_cap, int tile_cap = 0)
{
    LdsLayout L;
    L.tab = (double *)base;
    L.red = L.tab + 16;
    L.ctl = (int *)(L.red + 16);
    L.gl = (GaussLds *)(base + 16 * 8 + 16 * 8 + 16);
    L.tb = (double *)(L.gl + max_ngauss);
    L.cmask = (unsigned long long *)(L.tb + 4 * tile_cap);
    L.cpre = (int *)(L.cmask + nchunks_cap);
    return L;
}

static size_t lds_bytes(int max_ngauss, int nchunks_cap, int tile_cap = 0)
{
    return 16 * 8 + 16 * 8 + 16 + (size_t)max_ngauss * sizeof(GaussLds) +
           (size_t)tile_cap * 32 + (size_t)nchunks_cap * 12 + 16;
}

// Set norms lazily exactly as the reference does (gmix_nb.py:850-851: all of
// them when gmix[0].norm_set == 0, stopping at the first failure; gaussians
// before the failing one keep their fresh norms).  Also stages the exp table.
// Returns the status for the stamp (uniform across the work-group).
template <int NT = BLOCK>
__device__ __forceinline__ int lazy_norms(const LdsLayout &L, ngmix_gauss2d *gm,
                                          int ng)
{
    const int tid = threadIdx.x;
    if (tid < 16) L.tab[tid] = c_exp_table[tid];
    if (tid == 0) {
        L.ctl[0] = 1 << 30;  // index of first failing gaussian
        L.ctl[1] = 0;        // its error code
        L.ctl[2] = 1;        // all gaussians share one centre (fused kernel)
        L.ctl[3] = 1;        // all chi2 forms positive definite (fused kernel)
    }
    __syncthreads();
    const bool need = ng > 0 && gm[0].norm_set == 0;
    if (need) {
        for (int g = tid; g < ng; g += NT) {
            ngmix_gauss2d t = gm[g];
            int st = gauss_set_norm(t);
            if (st) atomicMin(&L.ctl[0], g);
        }
        __syncthreads();
        const int first_fail = L.ctl[0];
        for (int g = tid; g < ng; g += NT) {
            if (g < first_fail) {
                ngmix_gauss2d t = gm[g];
                gauss_set_norm(t);
                gm[g] = t;  // the reference mutates the caller's array
            } else if (g == first_fail) {
                ngmix_gauss2d t = gm[g];
                L.ctl[1] = gauss_set_norm(t);
            }
        }
        __syncthreads();
        if (L.ctl[1] != 0) return L.ctl[1];
    }
    return NGMIX_OK;
}

__device__ __forceinline__ int stage_gaussians(const LdsLayout &L,
                                               ngmix_gauss2d *gm, int ng,
                                               const ngmix_jacobian &jac,
                                               bool want_box)
{
    const int st = lazy_norms(L, gm, ng);
    if (st != NGMIX_OK) return st;
    for (int g = threadIdx.x; g < ng; g += BLOCK) {
        ngmix_gauss2d t = gm[g];
        GaussLds r;
        r.e = make_eval(t);
        r.box = want_box ? gauss_pixel_box(t, jac) : full_box();
        L.gl[g] = r;
    }
    __syncthreads();
    return NGMIX_OK;
}

template <int OP, int K>
__global__ __launch_bounds__(BLOCK) void pixpass_grid_kernel(
    const ngmix_stamp *__restrict__ stamps, const double *__restrict__ val,
    const double *__restrict__ ierr, const ngmix_jacobian *__restrict__ jacs,
    ngmix_gauss2d *gmix, double *out, const int64_t *__restrict__ out_start,
    int32_t *status, int max_ngauss, int nchunks_cap, int no_skip)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LdsLayout L = carve(smem, max_ngauss, nchunks_cap);

    const int s = blockIdx.x;
    const ngmix_stamp st = stamps[s];
    const ngmix_jacobian jac = jacs[s];
    const int nrow = st.nrow, ncol = st.ncol, ng = st.ngauss;
    const int npix = nrow * ncol;
    ngmix_gauss2d *gm = gmix + st.gm_off;
    const double *sval = val + st.pix_off;
    const double *sierr = ierr + st.pix_off;
    const bool izw = (st.flags & NGMIX_STAMP_IGNORE_ZERO_WEIGHT) != 0;
    constexpr bool kFast = (OP != OP_RENDER_EXACT);
    constexpr bool kNeedsVal = (OP == OP_LOGLIKE || OP == OP_FDIFF);
    constexpr bool kNeedsIerr = (OP != OP_RENDER_FAST && OP != OP_RENDER_EXACT);

    const int stcode = stage_gaussians(L, gm, ng, jac, kFast && !(no_skip & 1));
    if (stcode != NGMIX_OK) {
        // (an overwriting render leaves a defined image for a stamp that raises)
        if ((OP == OP_RENDER_FAST || OP == OP_RENDER_EXACT) && (no_skip & 4))
            for (int p = threadIdx.x; p < npix; p += BLOCK) out[st.pix_off + p] = 0.0;
        if (threadIdx.x == 0) status[s] = stcode;
        return;
    }
    const bool masked = kNeedsIerr && izw && st.npix_kept != npix;
    if (OP == OP_FDIFF && masked) build_rank_tables(L.cmask, L.cpre, sierr, npix);

    const double area = jac.scale * jac.scale;  // jacobian_nb.py:33-40
    const int lane = lane_id(), w = wave_id();
    const int lrow = lane / TILE_W, lcol = lane % TILE_W;
    const int ntx = (ncol + TILE_W - 1) / TILE_W;
    const int nty = (nrow + TILE_H - 1) / TILE_H;
    const int ntiles = ntx * nty;
    const int nrounds = (ntiles + NWAVES * K - 1) / (NWAVES * K);

    double acc_ll = 0.0, acc_sn = 0.0, acc_sd = 0.0;
    int acc_np = 0;

    for (int round = 0; round < nrounds; round++) {
        const int tbase = (round * NWAVES + w) * K;
        double pv[K], pu[K], pval[K], pierr[K], model[K];
        int pidx[K];
        bool inb[K];
        // lane k (< K) also carries tile k's extent for the skip test
        int my_r0 = 0, my_c0 = 0;
        bool my_valid = false;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int T = tbase + k;
            const int ty = T / ntx, tx = T - ty * ntx;
            const int r0 = ty * TILE_H, c0 = tx * TILE_W;
            if (lane == k) {
                my_r0 = r0;
                my_c0 = c0;
                my_valid = T < ntiles;
            }
            const int row = r0 + lrow, col = c0 + lcol;
            inb[k] = (T < ntiles) && row < nrow && col < ncol;
            pidx[k] = row * ncol + col;
            pval[k] = 0.0;
            pierr[k] = 0.0;
            if (inb[k]) {
                if (kNeedsVal) pval[k] = sval[pidx[k]];
                if (kNeedsIerr) pierr[k] = sierr[pidx[k]];
                // (no_skip bit 2 = NGMIX_BATCH_RENDER_OVERWRITE: image = model)
                if ((OP == OP_RENDER_FAST || OP == OP_RENDER_EXACT) && !(no_skip & 4))
                    pval[k] = out[st.pix_off + pidx[k]];
            }
            jacobian_vu(jac, (double)row, (double)col, pv[k], pu[k]);
            model[k] = 0.0;
        }

        for (int g = 0; g < ng; g++) {
            const GaussLds gl = L.gl[g];
            unsigned long long tmask;
            if (kFast) {
                const bool hit = my_valid && my_r0 <= gl.box.rmax &&
                                 my_r0 + TILE_H - 1 >= gl.box.rmin &&
                                 my_c0 <= gl.box.cmax &&
                                 my_c0 + TILE_W - 1 >= gl.box.cmin;
                tmask = __ballot(hit);
            } else {
                tmask = ~0ull;
            }
#pragma unroll
            for (int k = 0; k < K; k++) {
                if ((tmask >> k) & 1ull) {
                    if (kFast)
                        model[k] += gauss_eval_fast(gl.e, pv[k], pu[k], area, L.tab);
                    else
                        model[k] += gauss_eval_exact(gl.e, pv[k], pu[k], area);
                }
            }
        }

#pragma unroll
        for (int k = 0; k < K; k++) {
            if (!inb[k]) continue;
            const bool kept = !izw || pierr[k] > 0.0 || !kNeedsIerr;
            if (OP == OP_LOGLIKE) {
                if (kept) {
                    const double ivar = pierr[k] * pierr[k];
                    const double diff = model[k] - pval[k];
                    acc_ll += diff * diff * ivar;
                    acc_sn += pval[k] * model[k] * ivar;
                    acc_sd += model[k] * model[k] * ivar;
                    acc_np += 1;
                }
            } else if (OP == OP_S2N) {
                if (kept) {
                    const double ivar = pierr[k] * pierr[k];
                    acc_sd += model[k] * model[k] * ivar;
                }
            } else if (OP == OP_FDIFF) {
                if (kept) {
                    const int rank = masked ? kept_rank(L.cmask, L.cpre, pidx[k]) : pidx[k];
                    out[out_start[s] + rank] = (model[k] - pval[k]) * pierr[k];
                }
            } else {
                out[st.pix_off + pidx[k]] = pval[k] + model[k];
            }
        }
    }

    if (OP == OP_LOGLIKE) {
        double v[4] = {acc_ll, acc_sn, acc_sd, (double)acc_np};
        block_sum<4>(v, L.red);
        if (threadIdx.x == 0) {
            out[4 * (int64_t)s + 0] = v[0] * -0.5;  // gmix_nb.py:872
            out[4 * (int64_t)s + 1] = v[1];
            out[4 * (int64_t)s + 2] = v[2];
            out[4 * (int64_t)s + 3] = v[3];
            status[s] = NGMIX_OK;
        }
    } else if (OP == OP_S2N) {
        double v[1] = {acc_sd};
        block_sum<1>(v, L.red);
        if (threadIdx.x == 0) {
            out[s] = v[0];
            status[s] = NGMIX_OK;
        }
    } else {
        if (threadIdx.x == 0) status[s] = NGMIX_OK;
    }
}

// ===========================================================================
// FUSED kernels (default): ONE WAVE PER STAMP.
//
// Same results as the exact kernels to rounding (<= ~1e-13 of the stamp's peak
// per pixel; north_star tolerance 1e-10) at about a third of the instructions:
//
//  * every VALU instruction costs the same issue slot on CDNA (fp64 FMA runs
//    at full rate), so the design minimises instructions per pixel-gaussian
//    pair and per stamp, not flops;
//  * a 64-thread work-group (one wave) owns a stamp: no barriers, one set of
//    per-stamp overheads instead of four, 32 stamps in flight per CU;
//  * chi2/2 with FMAs; when every gaussian of the stamp has the same centre
//    (object (x) centred psf) dv^2, du^2, dv*du are formed once per pixel and
//    each gaussian costs 3 instructions;
//  * the gate 0 <= chi2 < 25 is one unsigned compare on the high word of
//    chi2/2 -- exactly the reference's predicate when the form is positive
//    definite (chi2 is then never -0.0; negative, NaN and inf fail it as they
//    fail the reference's);
//  * the fexp cell index comes out of the low word of y + 1.5*2^52 (no
//    conversions), Horner with one v_fma_f64 per step, coefficients in SGPRs;
//  * per-tile records (coordinates, byte offset) are staged once in LDS, so
//    the tile loop has no index arithmetic; the masks of gaussians whose
//    chi2<25 box touches each tile come from one ballot per 64/ngauss tiles;
//  * FUSED_PF tiles of val/ierr are in flight per wave in rotating register
//    sets (no LDS staging of pixels, 8 waves per SIMD).
// The summation order over gaussians (index order) and the gate semantics
// are the reference's; only the rounding of individual operations differs.
// ===========================================================================

// In the 64 B per gaussian of the LDS layout: 48-byte records first, then the
// boxes, 16 B apart -- the box test reads lane g's box, and inside 64-byte
// records those reads fell on two banks (8-way conflicts on a 16-gaussian
// stamp: a quarter of the LDS cycles of config 5, rocprofv3
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, round 4).
struct GaussFused {
    double a, b, c, pa;  // y = chi2/2 = a dv^2 + b du^2 + c dv du ; pa = pnorm*area
    double row, col;
};
static_assert(sizeof(GaussFused) == 48, "GaussFused");
static_assert(sizeof(GaussFused) + sizeof(TileBox) == sizeof(GaussLds), "LDS budget");

constexpr int FUSED_PF = 4;         // register sets: tiles requested ahead
constexpr int FUSED_SENTINELS = 8;  // look-ahead past the last tile

// One record per 8x8 tile of the stamp, staged in LDS and read back with
// broadcast ds_reads: everything the tile loop needs without index arithmetic.
// Tile shape (template TW): 8x8 for the kernels that only read (fewest
// (tile, gaussian) pairs survive the box test: they are instruction bound);
// 4 rows x 16 columns for render, whose read-modify-write stream is HBM bound
// and wants whole 128-byte lines per wave access.
struct TileEnt {
    double bv, bu;   // (v, u) of the tile's first pixel
    int off;         // byte offset of the tile's first pixel inside the stamp
    int r0, c0;      // its row / column (r0 == nrow marks a sentinel)
    int pad;
};
static_assert(sizeof(TileEnt) == 32, "TileEnt");

// Untracked loads for the look-ahead of the FULL tile loop.  hipcc counts its
// own memory operations but is conservative across loop back-edges and
// branches (it drains to vmcnt(0) once per trip); these loads are invisible to
// it and are waited for by wait_vm<N>, N = number of such loads issued AFTER
// the one needed (loads complete in order).  Compiler-issued memory
// operations in between only make the wait conservative, never unsafe.
__device__ __forceinline__ void gload_f64(double &dst, const void *base, unsigned off)
{
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst) : "v"(off), "s"(base));
}

// The same load issued with EXEC = 0 when `on` (wave-uniform) is false: no
// memory request, dst untouched, but the instruction is still issued and
// counted -- so the number of loads behind any given one stays a constant and
// the register is never the target of a compiler-made copy (a branch around an
// asm load would make dst a phi of "loaded" and "old", and a copy of a
// register whose data has not landed yet reads garbage).
__device__ __forceinline__ void gload_f64_if(double &dst, const void *base,
                                             unsigned off, int on_mask)
{
    // on_mask: -1 (load) or 0 (EXEC = 0), wave-uniform, in an SGPR
    unsigned long long save;
    asm volatile(
        "s_mov_b64 %1, exec\n\t"
        "s_and_b32 exec_lo, exec_lo, %3\n\t"
        "s_and_b32 exec_hi, exec_hi, %3\n\t"
        "global_load_dwordx2 %0, %2, %4\n\t"
        "s_mov_b64 exec, %1"
        : "+v"(dst), "=&s"(save)
        : "v"(off), "s"(on_mask), "s"(base)
        : "scc");
}

// two loads (val and ierr of one tile) under one EXEC toggle
__device__ __forceinline__ void gload2_f64_if(double &d0, const void *base0, double &d1,
                                              const void *base1, unsigned off,
                                              int on_mask)
{
    unsigned long long save;
    asm volatile(
        "s_mov_b64 %2, exec\n\t"
        "s_and_b32 exec_lo, exec_lo, %4\n\t"
        "s_and_b32 exec_hi, exec_hi, %4\n\t"
        "global_load_dwordx2 %0, %3, %5\n\t"
        "global_load_dwordx2 %1, %3, %6\n\t"
        "s_mov_b64 exec, %2"
        : "+v"(d0), "+v"(d1), "=&s"(save)
        : "v"(off), "s"(on_mask), "s"(base0), "s"(base1)
        : "scc");
}

template <int N>
__device__ __forceinline__ void wait_vm(double &a, double &b)
{
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}

// vmcnt(0) with every look-ahead register tied to it: nothing that uses them
// can be scheduled above the wait
__device__ __forceinline__ void wait_vm_all(double &a, double &b, double &c, double &d,
                                            double &e, double &f, double &g, double &h)
{
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
}

// The tile loop of one stamp.  FAST = every gaussian of the stamp is positive
// definite and shares one centre.
template <int OP, bool MASKED, bool FAST, bool FULL, int TW>
__device__ __forceinline__ void wave_tiles(
    const LdsLayout &L, const GaussFused *gf, const TileBox *gbox, const TileEnt *te,
    int ng,
    const ngmix_stamp &st, const double *__restrict__ sval,
    const double *__restrict__ sierr, bool masked, double *out, int64_t out_base,
    const ngmix_jacobian &jac, double (&pv)[FUSED_PF], double (&pe)[FUSED_PF],
    double &acc_ll, double &acc_sn, double &acc_sd, const int keep = -1)
{
    constexpr bool kNeedsVal = (OP == OP_LOGLIKE || OP == OP_FDIFF);
    constexpr bool kNeedsIerr = (OP != OP_RENDER_FAST);
    constexpr int TH = WAVE / TW;  // tile = TH rows x TW columns = one wave
    // keep == 0: render with NGMIX_BATCH_RENDER_OVERWRITE -- the image is not
    // read (the look-ahead loads are issued with EXEC = 0, their registers stay
    // 0.0); keep == -1 otherwise.  A scalar mask, never a select.
    const bool overwrite = keep == 0;
    const int lane = threadIdx.x;
    const int lrow = lane / TW, lcol = lane % TW;
    const int nrow = st.nrow, ncol = st.ncol;
    const int ntx = (ncol + TW - 1) / TW;
    const int nty = (nrow + TH - 1) / TH;
    const int ntiles = __builtin_amdgcn_readfirstlane(ntx * nty);
    // FULL: every tile is complete -- no per-lane bounds tests, and every
    // load / store is unconditional so that the compiler can count them:
    // waiting for tile T's data is then s_waitcnt vmcnt(ops issued since),
    // which leaves the look-ahead loads in flight (a branch around a memory
    // instruction makes the count unknown and forces vmcnt(0))
    constexpr bool full = FULL;
    // lane-constant parts: offset inside a tile in (v,u), minus the shared
    // centre when there is one, and in bytes
    const double cen_row = FAST ? gf[0].row : 0.0, cen_col = FAST ? gf[0].col : 0.0;
    const double olv =
        fma(jac.dvdrow, (double)lrow, jac.dvdcol * (double)lcol) - cen_row;
    const double olu =
        fma(jac.dudrow, (double)lrow, jac.dudcol * (double)lcol) - cen_col;
    const unsigned lane_off = (unsigned)(lrow * ncol + lcol) * 8u;
    const int rlim = nrow - lrow, clim = ncol - lcol;  // in bounds: r0 < rlim, c0 < clim
    const char *bval = (const char *)sval;
    const char *bierr = (const char *)sierr;
    char *bimg = (char *)(out + st.pix_off);   // render: the stamp's image
    char *bfd = (char *)(out + out_base);      // fdiff: the stamp's residuals

    // (tile, gaussian) box tests, CH tiles per ballot: lane = k*ng + g holds
    // gaussian g's box and tests it against tile T + k
    const bool chunked = ng <= 32;
    const int CH = chunked ? WAVE / ng : 0;
    const unsigned ngmask = chunked ? (unsigned)((1ull << ng) - 1ull) : 0u;
    int k_l = 0;
    bool lane_valid = false;
    const TileBox *mybox_p = &gbox[0];   // re-read at every ballot: 4 registers less
    if (chunked) {
        k_l = lane / ng;
        lane_valid = k_l < CH;
        mybox_p = &gbox[lane - k_l * ng];
    }
    const unsigned long long valid_mask = __builtin_amdgcn_ballot_w64(lane_valid);
    unsigned long long allmask = 0ull;
    int kc = 0;
    const FexpCoef K = load_fexp_coef(c_fexp_coef);

    // issue the loads of tile Tn (a sentinel past the last tile loads
    // nothing); lanes outside the stamp carry val = ierr = 0
    auto prefetch = [&](int Tn, bool &inb_n, double &nval, double &nierr) {
        if (full) {
            inb_n = true;
            // past the last tile: issued with EXEC = 0 (te[] has sentinels)
            int on;
            asm("s_sub_i32 %0, %1, %2\n\ts_ashr_i32 %0, %0, 31"
                : "=s"(on) : "s"(Tn), "s"(ntiles) : "scc");
            const unsigned off2 = lane_off + (unsigned)te[Tn].off;
            if (kNeedsVal && kNeedsIerr) {
                gload2_f64_if(nval, bval, nierr, bierr, off2, on);
            } else {
                if (kNeedsVal) gload_f64_if(nval, bval, off2, on);
                if (kNeedsIerr) gload_f64_if(nierr, bierr, off2, on);
                if (OP == OP_RENDER_FAST) gload_f64_if(nval, bimg, off2, on & keep);
            }
        } else {
            const int r0n = te[Tn].r0, c0n = te[Tn].c0;
            inb_n = (r0n < rlim) & (c0n < clim);
            nval = 0.0;
            nierr = 0.0;
            if (inb_n) {
                const unsigned off2 = lane_off + (unsigned)te[Tn].off;
                if (kNeedsVal) nval = *(const double *)(bval + off2);
                if (kNeedsIerr) nierr = *(const double *)(bierr + off2);
                if (OP == OP_RENDER_FAST && !overwrite) nval = *(const double *)(bimg + off2);
            }
        }
    };

    // one tile: evaluate the gaussians that can reach it, accumulate / store
    constexpr int kLoadsPerTile = (kNeedsVal ? 1 : 0) + (kNeedsIerr ? 1 : 0) +
                                  (OP == OP_RENDER_FAST ? 1 : 0);
    auto compute = [&](auto nyounger, int Tc, bool inb, double &pval, double &pierr) {
        const double v = te[Tc].bv + olv, u = te[Tc].bu + olu;
        double dv = v, du = u;
        double v2 = dv * dv, u2 = du * du, vu = dv * du;
        double model = 0.0;

        for (int g0 = 0; g0 < ng; g0 += 32) {
            unsigned gmask;
            if (chunked) {
                if (kc == 0) {
                    int Tk;  // = Tc + k_l; asm so that it is not hoisted out
                    asm volatile("v_add_u32 %0, %1, %2" : "=v"(Tk) : "s"(Tc), "v"(k_l));
                    if (Tk > ntiles) Tk = ntiles;  // a sentinel
                    const int r0k = te[Tk].r0, c0k = te[Tk].c0;
                    const TileBox mybox = *mybox_p;
                    allmask = tile_hits(mybox, r0k, c0k) & valid_mask;
                }
                gmask = (unsigned)allmask & ngmask;
                allmask >>= ng;
                kc = (kc + 1 == CH) ? 0 : kc + 1;
            } else {
                // lane g tests gaussian g0+g's box against this tile
                const int r0 = te[Tc].r0, c0 = te[Tc].c0;
                const int gi = (lane < 32 && g0 + lane < ng) ? g0 + lane : g0;
                const TileBox box = gbox[gi];
                gmask = (unsigned)(tile_hits(box, r0, c0) &
                                   __builtin_amdgcn_ballot_w64((lane < 32) & (g0 + lane < ng)));
            }
            while (gmask) {
                const int g = g0 + __builtin_ctz(gmask);
                gmask &= gmask - 1u;
                const GaussFused &G = gf[g];
                const double ga = G.a, gb = G.b, gc = G.c, gpa = G.pa;
                if (!FAST) {
                    dv = v - G.row;
                    du = u - G.col;
                    v2 = dv * dv;
                    u2 = du * du;
                    vu = dv * du;
                }
                const double y = fma(ga, v2, fma(gb, u2, gc * vu));  // chi2/2
                // 0 <= chi2 < 25  <=>  y in [+0, 12.5)
                const bool pass = FAST ? ((unsigned)__double2hiint(y) < 0x40290000u)
                                       : (y < 12.5 && y >= 0.0);
                if (pass) {
                    double e = fexp_neg_fused(y, L.tab, K);
                    const bool band = FAST ? ((unsigned)__double2hiint(y) >= 0x40240000u)
                                           : (y > 10.0);
                    if (band) {
                        // apod_window (fastexp_nb.py:97-117): W = u^3 (10 - 15 u
                        // + 6 u^2), u = (12.5 - y) * 0.4, written in b = 0.8 u:
                        //   W = kappa b^3 ((b - 1)^2 + 1/15), kappa = 9.375 / 0.512
                        // -- b is ONE fma (4.0 and 1.0 are inline constants), and no
                        // step reads two SGPR constants, so none has to be moved to
                        // a VGPR first: 8 instructions against 10 for the textbook
                        // form.  W(chi2 == 20) = 1 to 1 ulp.
                        const double bb = fma(y, K.wb, 4.0);
                        const double bm = bb - 1.0;
                        const double bq = fma(bm, bm, K.wq);
                        e *= (bb * bb) * (bb * bq);
                        e *= K.wk;
                    }
                    model = fma(gpa, e, model);
                }
            }
        }

        // this tile's val / ierr: the younger tiles stay in flight (the
        // look-ahead past the last tile is issued with EXEC = 0 and completes
        // in order like any other load)
        if (full) wait_vm<decltype(nyounger)::value * kLoadsPerTile>(pval, pierr);
        if (OP == OP_LOGLIKE || OP == OP_S2N) {
            // lanes outside the stamp have ierr == 0 and add exactly 0; so do
            // zero-weight pixels, except that a masked pixel may hold a
            // non-finite val, hence the select in the MASKED kernels
            const double am = model * pierr;
            double t_ll = 0.0, t_sn = 0.0;
            if (OP == OP_LOGLIKE) {
                const double bv_ = pval * pierr;
                const double d = am - bv_;
                t_ll = fma(d, d, acc_ll);
                t_sn = fma(am, bv_, acc_sn);
            }
            const double t_sd = fma(am, am, acc_sd);
            if (!MASKED || !masked || pierr > 0.0) {
                acc_ll = t_ll;
                acc_sn = t_sn;
                acc_sd = t_sd;
            }
        } else if (full || inb) {
            const unsigned off = lane_off + (unsigned)te[Tc].off;
            if (OP == OP_RENDER_FAST) {
                *(double *)(bimg + off) = pval + model;
            } else if (!MASKED || !masked) {
                *(double *)(bfd + off) = (model - pval) * pierr;
            } else if (pierr > 0.0) {
                const int rank = kept_rank(L.cmask, L.cpre, (int)(off >> 3));
                out[out_base + rank] = (model - pval) * pierr;
            }
        }
    };

    // Four register sets hold tiles T .. T+3; tiles are requested in PAIRS of
    // neighbours (two 64-byte halves of the same 128-byte lines issued back
    // to back: with single requests a tile-time apart ~15 % of the second
    // halves had left L2 and were fetched from HBM again), so 2-3 tiles are
    // always in flight behind the one being evaluated.  The sets rotate by
    // unrolling, not by copying.  (FULL stamps: the first four tiles were
    // requested by the kernel before the gaussians were staged.)
    static_assert(FUSED_PF == 4, "the rotation below is written for four register sets");
    using Y3 = std::integral_constant<int, 3>;
    using Y2 = std::integral_constant<int, 2>;
    int T = 0;
    bool in0 = true, in1 = true, in2 = true, in3 = true;
    double va0 = pv[0], va1 = pv[1], va2 = pv[2], va3 = pv[3];
    double ie0 = pe[0], ie1 = pe[1], ie2 = pe[2], ie3 = pe[3];
    if (!full) {
        prefetch(0, in0, va0, ie0);
        prefetch(1, in1, va1, ie1);
        prefetch(2, in2, va2, ie2);
        prefetch(3, in3, va3, ie3);
    }
    // No exit from the middle of a group of four: the tiles past the last one
    // are SKIPPED (forward branches) while their look-ahead loads are still
    // issued (with EXEC = 0), so every path through the loop issues the same
    // loads in the same order and the hand-counted waits are provable on the
    // machine code (tools/isa_hazards.py follows every load to its s_waitcnt
    // over the control-flow graph; with `break`s inside the group the compiler
    // routed the exits back through the loop header, where no static count
    // holds).
    while (T < ntiles) {
        compute(Y3{}, T, in0, va0, ie0);
        if (T + 1 < ntiles) compute(Y2{}, T + 1, in1, va1, ie1);
        prefetch(T + 4, in0, va0, ie0);
        prefetch(T + 5, in1, va1, ie1);
        if (T + 2 < ntiles) compute(Y3{}, T + 2, in2, va2, ie2);
        if (T + 3 < ntiles) compute(Y2{}, T + 3, in3, va3, ie3);
        prefetch(T + 6, in2, va2, ie2);
        prefetch(T + 7, in3, va3, ie3);
        T += 4;
    }
    // every look-ahead load has landed from here on (those past the last tile
    // were issued with EXEC = 0 and return at once): the registers are free
    if (full) wait_vm_all(va0, ie0, va1, ie1, va2, ie2, va3, ie3);
}

template <int OP, bool MASKED, int TW>
__device__ __forceinline__ void pixpass_wave_body(
    const ngmix_stamp *__restrict__ stamps, const double *__restrict__ val,
    const double *__restrict__ ierr, const ngmix_jacobian *__restrict__ jacs,
    ngmix_gauss2d *gmix, double *out, const int64_t *__restrict__ out_start,
    int32_t *status, int max_ngauss, int nchunks_cap, int no_skip, int tile_cap)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LdsLayout L = carve(smem, max_ngauss, nchunks_cap, tile_cap);
    GaussFused *gf = (GaussFused *)L.gl;
    TileBox *gbox = (TileBox *)(gf + max_ngauss);
    TileEnt *te = (TileEnt *)L.tb;

    const int s = blockIdx.x;
    const ngmix_stamp st = stamps[s];
    const ngmix_jacobian jac = jacs[s];
    const int nrow = st.nrow, ncol = st.ncol, ng = st.ngauss;
    const int npix = nrow * ncol;
    ngmix_gauss2d *gm = gmix + st.gm_off;
    const double *sval = val + st.pix_off;
    const double *sierr = ierr + st.pix_off;
    const bool izw = (st.flags & NGMIX_STAMP_IGNORE_ZERO_WEIGHT) != 0;
    constexpr bool kNeedsVal = (OP == OP_LOGLIKE || OP == OP_FDIFF);
    const double area = jac.scale * jac.scale;
    const int lane = threadIdx.x;

    // ---- stage 1: tile records (need only the stamp shape and jacobian)
    constexpr int TH = WAVE / TW;
    const int ntx = (ncol + TW - 1) / TW;
    const int nty = (nrow + TH - 1) / TH;
    const int ntiles = ntx * nty;
    for (int T = lane; T < ntiles + FUSED_SENTINELS; T += WAVE) {
        TileEnt e;
        if (T < ntiles) {
            const int ty = T / ntx, tx = T - ty * ntx;
            e.r0 = ty * TH;
            e.c0 = tx * TW;
            const double rd = (double)e.r0 - jac.row0, cd = (double)e.c0 - jac.col0;
            e.bv = fma(jac.dvdrow, rd, jac.dvdcol * cd);
            e.bu = fma(jac.dudrow, rd, jac.dudcol * cd);
            e.off = (e.r0 * ncol + e.c0) * 8;
        } else {
            e.r0 = nrow;
            e.c0 = ncol;
            e.bv = 0.0;
            e.bu = 0.0;
            e.off = 0;
        }
        e.pad = 0;
        te[T] = e;
    }
    __syncthreads();  // one wave: orders the LDS writes above, no s_barrier

    // 0 when an overwriting render must not read the image, else -1 (integer
    // arithmetic on the kernel argument: stays on the scalar unit)
    const int keep = (OP == OP_RENDER_FAST) ? (((no_skip >> 2) & 1) - 1) : -1;
    // ---- request the first tiles now: they fly while the gaussians are staged
    // (no_skip bit 1 = NGMIX_BATCH_TRACKED_LOADS: the compiler-tracked path)
    const bool full = (nrow % TH) == 0 && (ncol % TW) == 0 && !(no_skip & 2);
    double pv[FUSED_PF], pe[FUSED_PF];
#pragma unroll
    for (int t = 0; t < FUSED_PF; t++) {
        pv[t] = 0.0;
        pe[t] = 0.0;
    }
    if (full && ng > 0 && ntiles > 0) {
        const unsigned lane_off = (unsigned)((lane / TW) * ncol + lane % TW) * 8u;
#pragma unroll
        for (int t = 0; t < FUSED_PF; t++) {
            const int on = __builtin_amdgcn_readfirstlane(-(int)(t < ntiles));
            const unsigned off = lane_off + (unsigned)te[t].off;
            if (kNeedsVal) gload_f64_if(pv[t], (const char *)sval, off, on);
            if (OP != OP_RENDER_FAST) gload_f64_if(pe[t], (const char *)sierr, off, on);
            if (OP == OP_RENDER_FAST)
                gload_f64_if(pv[t], (const char *)(out + st.pix_off), off, on & keep);
        }
    }

    // ---- stage 2: norms (lazily, as the reference) and gaussian records
    const bool overwrite = OP == OP_RENDER_FAST && (no_skip & 4);
    const int stcode = lazy_norms<WAVE>(L, gm, ng);
    if (stcode != NGMIX_OK || (overwrite && ng == 0)) {
        // the look-ahead registers are dead on this path and the compiler may
        // reuse them: not before their loads have landed
        // (no operands: tying the dead values to the asm made the compiler
        // copy them into fresh registers BEFORE the wait)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (an overwriting render leaves a defined image: zeros for a stamp the
        // reference would have raised on, and for an empty mixture)
        if (overwrite)
            for (int p = lane; p < npix; p += WAVE) out[st.pix_off + p] = 0.0;
        if (lane == 0) status[s] = stcode;
        return;
    }
    // the fused evaluator indexes the table by n = round(chi2/2): exp(-n)
    if (lane < 16) L.tab[lane] = c_exp_table[15 - lane];
    const double row0 = ng > 0 ? gm[0].row : 0.0, col0 = ng > 0 ? gm[0].col : 0.0;
    for (int g = lane; g < ng; g += WAVE) {
        const ngmix_gauss2d t = gm[g];
        GaussFused r;
        r.a = 0.5 * t.dcc;   // exact scalings: y == 0.5 * chi2 bit for bit
        r.b = 0.5 * t.drr;
        r.c = -t.drc;
        r.pa = t.pnorm * area;
        r.row = t.row;
        r.col = t.col;
        const PixBox pb = (no_skip & 1) ? full_box() : gauss_pixel_box(t, jac);
        const TileBox tb = tile_box(pb, TH, TW);
        gbox[g] = tb;
        gf[g] = r;
        if (!(t.row == row0 && t.col == col0)) L.ctl[2] = 0;
        const double detq = t.dcc * t.drr - t.drc * t.drc;
        if (!(t.dcc > 0.0 && t.drr > 0.0 && detq > 0.0)) L.ctl[3] = 0;
    }
    __syncthreads();
    // The first tiles have landed by now (the staging above waited for its own,
    // younger loads); saying so makes every later copy, select or reuse of
    // these registers provably safe on every static path -- the dispatch
    // below copies them into the loop's register sets (tools/isa_hazards.py)
    wait_vm_all(pv[0], pe[0], pv[1], pe[1], pv[2], pe[2], pv[3], pe[3]);
    const bool fast = L.ctl[2] != 0 && L.ctl[3] != 0;
    const bool masked = MASKED && izw && st.npix_kept != npix;
    if (OP == OP_FDIFF && masked) build_rank_tables<WAVE>(L.cmask, L.cpre, sierr, npix);
    const int64_t out_base = (OP == OP_FDIFF) ? out_start[s] : 0;

    double acc_ll = 0.0, acc_sn = 0.0, acc_sd = 0.0;
    if (ng > 0) {
        if (fast && full)
            wave_tiles<OP, MASKED, true, true, TW>(L, gf, gbox, te, ng, st, sval, sierr, masked,
                                               out, out_base, jac, pv, pe, acc_ll,
                                               acc_sn, acc_sd, keep);
        else if (fast)
            wave_tiles<OP, MASKED, true, false, TW>(L, gf, gbox, te, ng, st, sval, sierr, masked,
                                                out, out_base, jac, pv, pe, acc_ll,
                                                acc_sn, acc_sd, keep);
        else if (full)
            wave_tiles<OP, MASKED, false, true, TW>(L, gf, gbox, te, ng, st, sval, sierr, masked,
                                                out, out_base, jac, pv, pe, acc_ll,
                                                acc_sn, acc_sd, keep);
        else
            wave_tiles<OP, MASKED, false, false, TW>(L, gf, gbox, te, ng, st, sval, sierr,
                                                 masked, out, out_base, jac, pv, pe,
                                                 acc_ll, acc_sn, acc_sd, keep);
    } else if (OP != OP_RENDER_FAST) {
        // an empty mixture: model == 0 everywhere
        for (int p = lane; p < npix; p += WAVE) {
            const double pv = kNeedsVal ? sval[p] : 0.0, pe = sierr[p];
            if (masked && !(pe > 0.0)) continue;
            if (OP == OP_LOGLIKE) acc_ll = fma(pv * pv, pe * pe, acc_ll);
            if (OP == OP_FDIFF) {
                const int rank = masked ? kept_rank(L.cmask, L.cpre, p) : p;
                out[out_base + rank] = (0.0 - pv) * pe;
            }
        }
    }

    if (OP == OP_LOGLIKE) {
        // DPP inside rows of 16 lanes, the 4 row sums through LDS
        double *red = L.red;
        const double r0_ = row16_total(acc_ll), r1_ = row16_total(acc_sn),
                     r2_ = row16_total(acc_sd);
        if ((lane & 15) == 15) {
            red[0 + (lane >> 4)] = r0_;
            red[4 + (lane >> 4)] = r1_;
            red[8 + (lane >> 4)] = r2_;
        }
        __syncthreads();
        const double ll = ((red[0] + red[1]) + red[2]) + red[3];
        const double sn = ((red[4] + red[5]) + red[6]) + red[7];
        const double sd = ((red[8] + red[9]) + red[10]) + red[11];
        if (lane == 0) {
            out[4 * (int64_t)s + 0] = ll * -0.5;  // gmix_nb.py:872
            out[4 * (int64_t)s + 1] = sn;
            out[4 * (int64_t)s + 2] = sd;
            // the number of listed pixels is a property of the stamp
            out[4 * (int64_t)s + 3] = (double)(izw ? st.npix_kept : npix);
            status[s] = NGMIX_OK;
        }
    } else if (OP == OP_S2N) {
        const double sd = wave_total(acc_sd);
        if (lane == 0) {
            out[s] = sd;
            status[s] = NGMIX_OK;
        }
    } else {
        if (lane == 0) status[s] = NGMIX_OK;
    }
}

template <int OP, bool MASKED, int TW>
__global__ __launch_bounds__(WAVE) void pixpass_wave_kernel(
    const ngmix_stamp *__restrict__ stamps, const double *__restrict__ val,
    const double *__restrict__ ierr, const ngmix_jacobian *__restrict__ jacs,
    ngmix_gauss2d *gmix, double *out, const int64_t *__restrict__ out_start,
    int32_t *status, int max_ngauss, int nchunks_cap, int no_skip, int tile_cap)
{
    pixpass_wave_body<OP, MASKED, TW>(stamps, val, ierr, jacs, gmix, out, out_start, status,
                                      max_ngauss, nchunks_cap, no_skip, tile_cap);
}

// get_loglike at seven waves per SIMD: its body fits 72 VGPRs without spilling
// (the boxes re-read at each ballot, the window in the form that needs no VGPR constant), and
// with instruction issue and memory both ~80 % busy one more resident wave per
// SIMD is what overlaps them better
template <int OP, bool MASKED, int TW>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(7, 7)))
void pixpass_wave_kernel7(
    const ngmix_stamp *__restrict__ stamps, const double *__restrict__ val,
    const double *__restrict__ ierr, const ngmix_jacobian *__restrict__ jacs,
    ngmix_gauss2d *gmix, double *out, const int64_t *__restrict__ out_start,
    int32_t *status, int max_ngauss, int nchunks_cap, int no_skip, int tile_cap)
{
    pixpass_wave_body<OP, MASKED, TW>(stamps, val, ierr, jacs, gmix, out, out_start, status,
                                      max_ngauss, nchunks_cap, no_skip, tile_cap);
}

// ---------------------------------------------------------------- launchers

// fused kernels keep one 32-byte record per 8x8 tile in LDS; batches with a
// stamp of more tiles than this (> ~360x360 pixels) run the exact kernels
constexpr int FUSED_TILE_CAP = 2048;

static int pick_k(int max_npix)
{
    // tiles per wave per round: 9 covers 48x48 in one round, 4 covers 32x32
    if (max_npix <= 32 * 32) return 4;
    return 9;
}

template <int OP>
static int launch_grid(const ngmix_batch *b, ngmix_gauss2d *gmix, double *out,
                       const int64_t *out_start, int32_t *status, void *stream)
{
    if (b->nstamps <= 0) return NGMIX_OK;
    const bool need_rank = (OP == OP_FDIFF) && b->any_masked;
    const int nchunks_cap = need_rank ? (b->max_npix + 63) / 64 : 0;
    const int max_ng = b->max_ngauss > 0 ? b->max_ngauss : 1;
    const size_t lds = lds_bytes(max_ng, nchunks_cap);
    if (lds > 160 * 1024) {
        set_last_error_msg("stamp needs more than 160 KiB of LDS "
                           "(too many gaussians or masked stamp too large)");
        return NGMIX_ERR_BAD_ARG;
    }
    const int no_skip = ((b->flags & NGMIX_BATCH_NO_SKIP) ? 1 : 0) |
                        ((b->flags & NGMIX_BATCH_TRACKED_LOADS) ? 2 : 0) |
                        ((b->flags & NGMIX_BATCH_RENDER_OVERWRITE) ? 4 : 0);
    // per-tile records of the fused kernels: exact when the batch carries its
    // largest stamp shape, else ntiles <= npix/8 + 1 holds for any shape
    int a_tc = b->max_npix / 8 + 1;
    constexpr int TW = (OP == OP_RENDER_FAST || OP == OP_RENDER_EXACT || OP == OP_FDIFF) ? 16 : 8;
    constexpr int TH = WAVE / TW;
    if (b->max_nrow > 0 && b->max_ncol > 0)
        a_tc = ((b->max_nrow + TH - 1) / TH) * ((b->max_ncol + TW - 1) / TW);
    a_tc += FUSED_SENTINELS;
    // the true-exp render has no cut and no fused form
    const bool exact = (b->flags & NGMIX_BATCH_EXACT) || OP == OP_RENDER_EXACT ||
                       a_tc > FUSED_TILE_CAP;
    hipStream_t s = (hipStream_t)stream;
    const ngmix_stamp *a_stamps = b->stamps;
    const double *a_val = b->val, *a_ierr = b->ierr;
    const ngmix_jacobian *a_jac = b->jac;
    int a_ng = max_ng, a_nc = nchunks_cap, a_ns = no_skip;
    if (exact) {
        a_ns = no_skip & 5;
        dim3 grid((unsigned)b->nstamps), block(BLOCK);
        const bool k4 = pick_k(b->max_npix) == 4;
        const void *kern = k4 ? (const void *)pixpass_grid_kernel<OP, 4>
                              : (const void *)pixpass_grid_kernel<OP, 9>;
        if (lds > 64 * 1024)
            NGMIX_HIP_CHECK(hipFuncSetAttribute(
                kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        void *args[] = {&a_stamps, &a_val, &a_ierr, &a_jac, &gmix, &out, &out_start,
                        &status, &a_ng, &a_nc, &a_ns};
        census(OP == OP_LOGLIKE ? "pixpass_grid_kernel<loglike>"
               : OP == OP_FDIFF ? "pixpass_grid_kernel<fdiff>"
               : OP == OP_S2N   ? "pixpass_grid_kernel<s2n>"
                                : "pixpass_grid_kernel<render>");
        NGMIX_HIP_CHECK(hipLaunchKernel(kern, grid, block, args, lds, s));
        return NGMIX_OK;
    }
    constexpr int FOP = (OP == OP_RENDER_EXACT) ? OP_RENDER_FAST : OP;
    // zero-weight pixels only matter to the kernels that read ierr
    const bool mk = b->any_masked && FOP != OP_RENDER_FAST;
    const void *kern = mk ? (const void *)pixpass_wave_kernel<FOP, true, TW>
                          : (const void *)pixpass_wave_kernel<FOP, false, TW>;
    static const bool six_waves = getenv("NGMIX_LOGLIKE_6WAVES") != nullptr;   // A/B knob
    if (FOP == OP_LOGLIKE && !six_waves)
        kern = mk ? (const void *)pixpass_wave_kernel7<OP_LOGLIKE, true, 8>
                  : (const void *)pixpass_wave_kernel7<OP_LOGLIKE, false, 8>;
    const size_t flds = lds_bytes(max_ng, nchunks_cap, a_tc);
    if (flds > 160 * 1024) {
        set_last_error_msg("stamp needs more than 160 KiB of LDS");
        return NGMIX_ERR_BAD_ARG;
    }
    if (flds > 64 * 1024)
        NGMIX_HIP_CHECK(hipFuncSetAttribute(
            kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds));
    dim3 grid((unsigned)b->nstamps), block(WAVE);
    void *args[] = {&a_stamps, &a_val, &a_ierr, &a_jac, &gmix, &out, &out_start,
                    &status, &a_ng, &a_nc, &a_ns, &a_tc};
    census(FOP == OP_LOGLIKE ? (six_waves ? "pixpass_wave_kernel<loglike>"
                                          : "pixpass_wave_kernel7<loglike>")
           : FOP == OP_FDIFF ? "pixpass_wave_kernel<fdiff>"
           : FOP == OP_S2N   ? "pixpass_wave_kernel<s2n>"
                             : "pixpass_wave_kernel<render>");
    NGMIX_HIP_CHECK(hipLaunchKernel(kern, grid, block, args, flds, s));
    return NGMIX_OK;
}

int launch_loglike_grid(const ngmix_batch *b, ngmix_gauss2d *gmix, double *out,
                        int32_t *status, void *stream)
{
    return launch_grid<OP_LOGLIKE>(b, gmix, out, nullptr, status, stream);
}

int launch_fdiff_grid(const ngmix_batch *b, ngmix_gauss2d *gmix, double *fdiff,
                      const int64_t *fdiff_start, int32_t *status, void *stream)
{
    return launch_grid<OP_FDIFF>(b, gmix, fdiff, fdiff_start, status, stream);
}

int launch_render_grid(const ngmix_batch *b, ngmix_gauss2d *gmix, double *image,
                       int fast_exp, int32_t *status, void *stream)
{
    if (fast_exp)
        return launch_grid<OP_RENDER_FAST>(b, gmix, image, nullptr, status, stream);
    return launch_grid<OP_RENDER_EXACT>(b, gmix, image, nullptr, status, stream);
}

int launch_s2n_grid(const ngmix_batch *b, ngmix_gauss2d *gmix, double *out,
                    int32_t *status, void *stream)
{
    return launch_grid<OP_S2N>(b, gmix, out, nullptr, status, stream);
}

// =========================================================================
// LIST kernels: the reference's AoS arrays, one thread per pixel
// =========================================================================

constexpr int LIST_MAX_BLOCKS = 1024;

// stage up to `ng` gaussians in LDS; the seam form has already set the norms
// on the host (ngmix_set_norms) so no lazy path is needed here
__device__ __forceinline__ void stage_list(double *tab, EvalGauss *ge,
                                           const ngmix_gauss2d *gm, int ng)
{
    if (threadIdx.x < 16) tab[threadIdx.x] = c_exp_table[threadIdx.x];
    for (int g = threadIdx.x; g < ng; g += BLOCK) ge[g] = make_eval(gm[g]);
    __syncthreads();
}

__device__ __forceinline__ double eval_all_fast(const EvalGauss *ge, int ng,
                                                double v, double u, double area,
                                                const double *tab)
{
    double m = 0.0;
    for (int g = 0; g < ng; g++) m += gauss_eval_fast(ge[g], v, u, area, tab);
    return m;
}

__device__ __forceinline__ double eval_all_exact(const EvalGauss *ge, int ng,
                                                 double v, double u, double area)
{
    double m = 0.0;
    for (int g = 0; g < ng; g++) m += gauss_eval_exact(ge[g], v, u, area);
    return m;
}

// render over a coord list (render_nb.py:31-36)
__global__ __launch_bounds__(BLOCK) void render_list_kernel(
    const ngmix_gauss2d *gm, int ng, const ngmix_coord *coords, int64_t n,
    double *image, int fast_exp)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *tab = (double *)smem;
    EvalGauss *ge = (EvalGauss *)(tab + 16);
    stage_list(tab, ge, gm, ng);
    for (int64_t i = blockIdx.x * (int64_t)BLOCK + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * BLOCK) {
        const ngmix_coord c = coords[i];
        const double m = fast_exp ? eval_all_fast(ge, ng, c.v, c.u, c.area, tab)
                                  : eval_all_exact(ge, ng, c.v, c.u, c.area);
        image[i] += m;
    }
}

// get_loglike / get_model_s2n_sum / fill_fdiff over a pixel list.
// partial: gridDim.x * 4 doubles of per-block sums (OP_LOGLIKE / OP_S2N).
template <int OP>
__global__ __launch_bounds__(BLOCK) void pixpass_list_kernel(
    const ngmix_gauss2d *gm, int ng, const ngmix_pixel *pixels, int64_t n,
    double *fdiff, int64_t start, double *partial)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *tab = (double *)smem;
    double *red = tab + 16;
    EvalGauss *ge = (EvalGauss *)(red + 16);
    stage_list(tab, ge, gm, ng);
    double a_ll = 0.0, a_sn = 0.0, a_sd = 0.0, a_np = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)BLOCK + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * BLOCK) {
        const ngmix_pixel p = pixels[i];
        const double m = eval_all_fast(ge, ng, p.v, p.u, p.area, tab);
        if (OP == OP_FDIFF) {
            fdiff[start + i] = (m - p.val) * p.ierr;
        } else {
            const double ivar = p.ierr * p.ierr;
            if (OP == OP_LOGLIKE) {
                const double diff = m - p.val;
                a_ll += diff * diff * ivar;
                a_sn += p.val * m * ivar;
                a_np += 1.0;
            }
            a_sd += m * m * ivar;
        }
    }
    if (OP != OP_FDIFF) {
        double v[4] = {a_ll, a_sn, a_sd, a_np};
        block_sum<4>(v, red);
        if (threadIdx.x == 0) {
            for (int k = 0; k < 4; k++) partial[4 * blockIdx.x + k] = v[k];
        }
    }
}

// fixed-order final sum of the per-block partials
__global__ void sum_partials_kernel(const double *partial, int nblocks,
                                    double *out4)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        for (int b = 0; b < nblocks; b++)
            for (int k = 0; k < 4; k++) s[k] += partial[4 * b + k];
        for (int k = 0; k < 4; k++) out4[k] = s[k];
    }
}

static int list_blocks(int64_t n)
{
    int64_t nb = (n + BLOCK - 1) / BLOCK;
    if (nb < 1) nb = 1;
    if (nb > LIST_MAX_BLOCKS) nb = LIST_MAX_BLOCKS;
    return (int)nb;
}

int list_partial_doubles(void) { return LIST_MAX_BLOCKS * 4 + 4; }

int launch_render_list(const ngmix_gauss2d *gm, int ng, const ngmix_coord *coords,
                       int64_t n, double *image, int fast_exp, hipStream_t s)
{
    const size_t lds = 16 * 8 + (size_t)ng * sizeof(EvalGauss);
    hipLaunchKernelGGL(render_list_kernel, dim3(list_blocks(n)), dim3(BLOCK), lds,
                       s, gm, ng, coords, n, image, fast_exp);
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

// op: 0 loglike (out4 = sum diff^2 ivar, s2n_numer, s2n_denom, npix),
//     1 fdiff, 4 s2n (out4[2])
int launch_pixpass_list(int op, const ngmix_gauss2d *gm, int ng,
                        const ngmix_pixel *pixels, int64_t n, double *fdiff,
                        int64_t start, double *partial, hipStream_t s)
{
    const size_t lds = 32 * 8 + (size_t)ng * sizeof(EvalGauss);
    const int nb = list_blocks(n);
    if (op == OP_FDIFF) {
        hipLaunchKernelGGL(pixpass_list_kernel<OP_FDIFF>, dim3(nb), dim3(BLOCK),
                           lds, s, gm, ng, pixels, n, fdiff, start, partial);
    } else if (op == OP_LOGLIKE) {
        hipLaunchKernelGGL(pixpass_list_kernel<OP_LOGLIKE>, dim3(nb), dim3(BLOCK),
                           lds, s, gm, ng, pixels, n, fdiff, start, partial);
    } else {
        hipLaunchKernelGGL(pixpass_list_kernel<OP_S2N>, dim3(nb), dim3(BLOCK),
                           lds, s, gm, ng, pixels, n, fdiff, start, partial);
    }
    NGMIX_HIP_CHECK(hipGetLastError());
    if (op != OP_FDIFF) {
        hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, s, partial,
                           nb, partial + 4 * LIST_MAX_BLOCKS);
        NGMIX_HIP_CHECK(hipGetLastError());
    }
    return NGMIX_OK;
}

// =========================================================================
// fill_pixels / fill_coords (pixels_nb.py:6-94) and store preparation
// =========================================================================

// One work-group scans the image in row-major chunks of 256 pixels; kept
// pixels are compacted in order with a ballot prefix, so pixels[k] is the
// k-th kept pixel exactly as the reference's sequential loop produces it.
__global__ __launch_bounds__(BLOCK) void fill_pixels_kernel(
    ngmix_pixel *pixels, int64_t npixels, const double *image,
    const double *weight, int nrow, int ncol, ngmix_jacobian jac, int izw,
    int *count_out)
{
    __shared__ int wcount[NWAVES];
    __shared__ int base;
    const double area = jac.scale * jac.scale;
    const int64_t npix = (int64_t)nrow * ncol;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int64_t c0 = 0; c0 < npix; c0 += BLOCK) {
        const int64_t p = c0 + threadIdx.x;
        double ivar = 0.0;
        bool kept = false;
        if (p < npix) {
            ivar = weight[p];
            kept = !(izw && ivar <= 0.0);
        }
        const unsigned long long m = __ballot(kept);
        const int lane = lane_id(), w = wave_id();
        if (lane == 0) wcount[w] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int k = 0; k < w; k++) off += wcount[k];
        off += __popcll(m & ((1ull << lane) - 1ull));
        if (kept && off < npixels) {
            const int row = (int)(p / ncol), col = (int)(p - (int64_t)row * ncol);
            ngmix_pixel px;
            jacobian_vu(jac, (double)row, (double)col, px.v, px.u);
            px.area = area;
            px.val = image[p];
            if (ivar < 0.0) ivar = 0.0;
            px.ierr = sqrt(ivar);
            px.fdiff = 0.0;
            pixels[off] = px;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
            for (int k = 0; k < NWAVES; k++) t += wcount[k];
            base += t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *count_out = base;
}

__global__ __launch_bounds__(BLOCK) void fill_coords_kernel(ngmix_coord *coords,
                                                           int nrow, int ncol,
                                                           ngmix_jacobian jac)
{
    const double area = jac.scale * jac.scale;
    const int64_t npix = (int64_t)nrow * ncol;
    for (int64_t p = blockIdx.x * (int64_t)BLOCK + threadIdx.x; p < npix;
         p += (int64_t)gridDim.x * BLOCK) {
        const int row = (int)(p / ncol), col = (int)(p - (int64_t)row * ncol);
        ngmix_coord c;
        jacobian_vu(jac, (double)row, (double)col, c.v, c.u);
        c.area = area;
        coords[p] = c;
    }
}

int launch_fill_pixels(ngmix_pixel *pixels, int64_t npixels, const double *image,
                       const double *weight, int nrow, int ncol,
                       const ngmix_jacobian &jac, int izw, int *count_out,
                       hipStream_t s)
{
    hipLaunchKernelGGL(fill_pixels_kernel, dim3(1), dim3(BLOCK), 0, s, pixels,
                       npixels, image, weight, nrow, ncol, jac, izw, count_out);
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

int launch_fill_coords(ngmix_coord *coords, int nrow, int ncol,
                       const ngmix_jacobian &jac, hipStream_t s)
{
    const int64_t npix = (int64_t)nrow * ncol;
    hipLaunchKernelGGL(fill_coords_kernel, dim3(list_blocks(npix)), dim3(BLOCK), 0,
                       s, coords, nrow, ncol, jac);
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

__global__ __launch_bounds__(BLOCK) void weight_to_ierr_kernel(const double *w,
                                                              double *ierr,
                                                              int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)BLOCK + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * BLOCK) {
        // (no weight map: unit weights, Observation's default, observation.py:97-100)
        double ivar = w ? w[i] : 1.0;
        if (ivar < 0.0) ivar = 0.0;
        ierr[i] = sqrt(ivar);
    }
}

__global__ __launch_bounds__(BLOCK) void count_kept_kernel(ngmix_stamp *stamps,
                                                          const double *ierr)
{
    __shared__ double red[16];
    ngmix_stamp st = stamps[blockIdx.x];
    const int npix = st.nrow * st.ncol;
    const bool izw = (st.flags & NGMIX_STAMP_IGNORE_ZERO_WEIGHT) != 0;
    double cnt = 0.0;
    for (int p = threadIdx.x; p < npix; p += BLOCK)
        if (!izw || ierr[st.pix_off + p] > 0.0) cnt += 1.0;
    double v[1] = {cnt};
    block_sum<1>(v, red);
    if (threadIdx.x == 0) stamps[blockIdx.x].npix_kept = (int)v[0];
}

// fexp / apod_window / apod_window_deriv over an array: the innermost
// functions of every fast pixel evaluation, callable on their own
// (fastexp_nb.py:97-135, 223-265).  which: 0 fexp(x), 1 apod_window(x),
// 2 apod_window_deriv(x).  No range check, as in the reference: fexp reads its
// 16-entry table at int(x - 0.5) + 15, so x must lie in (-15.5, 1.5).
__global__ __launch_bounds__(BLOCK) void fastexp_kernel(const double *__restrict__ x,
                                                         double *__restrict__ out, int64_t n,
                                                         int which)
{
    __shared__ double tab[16];
    if (threadIdx.x < 16) tab[threadIdx.x] = c_exp_table[threadIdx.x];
    __syncthreads();
    for (int64_t i = blockIdx.x * (int64_t)BLOCK + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * BLOCK) {
        const double v = x[i];
        out[i] = which == 0 ? fexp(v, tab) : (which == 1 ? apod_window(v) : apod_window_deriv(v));
    }
}

int launch_fastexp(const double *x, double *out, int64_t n, int which, hipStream_t s)
{
    if (n <= 0) return NGMIX_OK;
    if (which < 0 || which > 2) return NGMIX_ERR_BAD_ARG;
    int64_t nb = (n + BLOCK - 1) / BLOCK;
    if (nb > 256 * 8) nb = 256 * 8;
    hipLaunchKernelGGL(fastexp_kernel, dim3((unsigned)nb), dim3(BLOCK), 0, s, x, out, n, which);
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

// Sum of fdiff^2 over the first nskip LISTED pixels (row-major; weight > 0
// where the stamp ignores zero weights) of one stamp per object, the model
// being mixture o of gm (ngauss normalised gaussians).  The reference reserves
// more residual rows for a joint prior than the prior fills (results.py:
// 1050-1078 against joint_prior.py:86-120) and leaves all reserved rows out of
// chi2/dof (leastsqbound.py:97): the first pixel rows go with them.  The
// arithmetic of fill_fdiff (gmix_nb.py:877-900), pixel by pixel.
__global__ __launch_bounds__(BLOCK) void first_pixels_fdiff2_kernel(
    const ngmix_stamp *__restrict__ stamps, const double *__restrict__ val,
    const double *__restrict__ ierr, const ngmix_jacobian *__restrict__ jac,
    const int64_t *__restrict__ stamp_of, const ngmix_gauss2d *__restrict__ gm, int ngauss,
    int64_t nobj, int nskip, double *__restrict__ out)
{
    const int64_t o = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
    if (o >= nobj) return;
    const int64_t si = stamp_of[o];
    const ngmix_stamp st = stamps[si];
    const ngmix_jacobian j = jac[si];
    const bool izw = (st.flags & NGMIX_STAMP_IGNORE_ZERO_WEIGHT) != 0;
    const double area = j.scale * j.scale;
    const ngmix_gauss2d *g = gm + o * (int64_t)ngauss;
    const int npix = st.nrow * st.ncol;
    double acc = 0.0;
    int found = 0;
    for (int p = 0; p < npix && found < nskip; p++) {
        const double e = ierr[st.pix_off + p];
        if (izw && !(e > 0.0)) continue;
        found++;
        double v, u;
        jacobian_vu(j, (double)(p / st.ncol), (double)(p % st.ncol), v, u);
        double model = 0.0;
        for (int k = 0; k < ngauss; k++)
            model += gauss_eval_fast(make_eval(g[k]), v, u, area, c_exp_table);
        const double fd = (model - val[st.pix_off + p]) * e;
        acc += fd * fd;
    }
    out[o] = acc;
}

int launch_first_pixels_fdiff2(const ngmix_batch *b, const int64_t *stamp_of,
                               const ngmix_gauss2d *gm, int ngauss, int64_t nobj, int nskip,
                               double *out, hipStream_t s)
{
    if (nobj <= 0) return NGMIX_OK;
    if (!b || !stamp_of || !gm || !out || ngauss <= 0 || nskip < 0) return NGMIX_ERR_BAD_ARG;
    hipLaunchKernelGGL(first_pixels_fdiff2_kernel, dim3((unsigned)((nobj + BLOCK - 1) / BLOCK)),
                       dim3(BLOCK), 0, s, (const ngmix_stamp *)b->stamps, (const double *)b->val,
                       (const double *)b->ierr, (const ngmix_jacobian *)b->jac, stamp_of, gm,
                       ngauss, nobj, nskip, out);
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

int launch_weight_to_ierr(const double *w, double *ierr, int64_t n, hipStream_t s)
{
    if (n <= 0) return NGMIX_OK;
    int64_t nb = (n + BLOCK - 1) / BLOCK;
    if (nb > 256 * 8) nb = 256 * 8;
    hipLaunchKernelGGL(weight_to_ierr_kernel, dim3((unsigned)nb), dim3(BLOCK), 0, s,
                       w, ierr, n);
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

int launch_count_kept(ngmix_stamp *stamps, int64_t nstamps, const double *ierr,
                      hipStream_t s)
{
    if (nstamps <= 0) return NGMIX_OK;
    hipLaunchKernelGGL(count_kept_kernel, dim3((unsigned)nstamps), dim3(BLOCK), 0,
                       s, stamps, ierr);
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

}  // namespace ngmix
