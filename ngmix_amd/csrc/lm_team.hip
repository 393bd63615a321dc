// lm_team.hip -- the lmder step of the lock-step fits by a TEAM of 16 lanes per
// fit with the fit's arrays in LDS (lm_core_team.hpp): what fits of 9-14
// parameters run, whose arrays fit no register file (9, 10: not without
// spilling) and whose generic one-thread form (lmfit.hip, lm_core.hpp) is a
// chain of private-memory round trips.  Reference: scipy's lmder / lmdif as ngmix/fitting/leastsqbound.py:
// 289-552 drives them; records byte-identical to the generic form's
// (tests/test_gpu_lm_team.py).
#include <stdio.h>

#include "launch.hpp"
#include "lm_core_team.hpp"

namespace ngmix {

// A, g, ff from the stamps' sums: entry k of a stamp's record always lands in
// the same element for a given band, so one lane per entry adds the stamps in
// stamp order -- the order of the one-thread loop, element by element.
__device__ __forceinline__ double lm_team_fold(
    lmteam::Fit &f, const int64_t s0, const int64_t s1,
    const int32_t *__restrict__ stamp_band, const double *__restrict__ sums, int nloc,
    const double *__restrict__ obj_row)
{
    using lmteam::TEAM;
    const int n = f.s.n, ld = f.ld;
    const int ntri = nloc * (nloc + 1) / 2, nsum = ntri + nloc + 1;
    // (M and g start as +0.0: the kernel zero-fills the team's block)
    // this lane's entries k = lane, lane + 16, ...: (a, b) of the triangle, or
    // a gradient entry (b = -1)
    constexpr int KMAX = (NGMIX_LM_NSUMS(LM_NPMAX) + TEAM - 1) / TEAM;
    int ka[KMAX], kb[KMAX];
#pragma unroll
    for (int q = 0; q < KMAX; q++) {
        const int k = f.lane + q * TEAM;
        ka[q] = kb[q] = -1;
        if (k < ntri) {
            int a = 0, row = 0;
            while (row + (nloc - a) <= k) {
                row += nloc - a;
                a++;
            }
            ka[q] = a;
            kb[q] = a + (k - row);
        } else if (k < ntri + nloc) {
            ka[q] = k - ntri;
        }
    }
    // The lane's entries accumulate in REGISTERS over the stamps (an LDS
    // read-modify-write per entry and stamp is a dependent round trip each).  An
    // entry that involves the flux lands in a different element for each band: when
    // the band changes the accumulators are written out and those of the new band
    // read in, so every element still adds its stamps in stamp order.
    double acc[KMAX];
#pragma unroll
    for (int q = 0; q < KMAX; q++) acc[q] = 0.0;
    auto target = [&](int q, int band, int &e0, int &e1) {
        // LDS slots of entry q for this band: e0 (and its mirror e1, or -1);
        // slots below n * ld are elements of M, the rest of g
        const int a = ka[q], b = kb[q];
        const int ga = a < nloc - 1 ? a : nloc - 1 + band;
        if (b < 0) {
            e0 = -1 - ga;          // g[ga]
            e1 = -1;
        } else {
            const int gb = b < nloc - 1 ? b : nloc - 1 + band;
            e0 = ga * ld + gb;
            e1 = ga != gb ? gb * ld + ga : -1;
        }
    };
    auto flush = [&](int band) {
#pragma unroll
        for (int q = 0; q < KMAX; q++) {
            if (ka[q] < 0) continue;
            int e0, e1;
            target(q, band, e0, e1);
            if (e0 < 0) {
                f.g[-1 - e0] = acc[q];
            } else {
                f.M[e0] = acc[q];
                if (e1 >= 0) f.M[e1] = acc[q];
            }
        }
    };
    double ff = 0.0;
    int cur = -1;
    for (int64_t st = s0; st < s1; st++) {
        const double *v = sums + st * nsum;
        const int band = stamp_band ? stamp_band[st] : 0;
        if (band != cur) {
            if (cur >= 0) flush(cur);
#pragma unroll
            for (int q = 0; q < KMAX; q++) {
                if (ka[q] < 0) continue;
                int e0, e1;
                target(q, band, e0, e1);
                acc[q] = e0 < 0 ? f.g[-1 - e0] : f.M[e0];
            }
            cur = band;
        }
#pragma unroll
        for (int q = 0; q < KMAX; q++)
            if (ka[q] >= 0) acc[q] += v[f.lane + q * TEAM];
        ff += v[ntri + nloc];
    }
    if (cur >= 0) flush(cur);
    if (obj_row) {
        // rows over the object's own n parameters (the prior rows)
        lmteam::tsync();
        const int nt = n * (n + 1) / 2;
        for (int k = f.lane; k < nt + n; k += TEAM) {
            const double t = obj_row[k];
            if (k < nt) {
                int a = 0, row = 0;
                while (row + (n - a) <= k) {
                    row += n - a;
                    a++;
                }
                const int b = a + (k - row);
                f.M[a * ld + b] += t;
                if (a != b) f.M[b * ld + a] += t;
            } else {
                f.g[k - nt] += t;
            }
        }
        ff += obj_row[nt + n];
    }
    lmteam::tsync();
    return ff;
}

// TEAMS fits per work-group (one wave of 16 * TEAMS lanes); NP: the compile-time
// bound of the fits' parameter count (the size of a batch of LDS loads and of a
// team's LDS block)
template <int TEAMS, int NP>
__global__ __launch_bounds__(TEAMS * lmteam::TEAM) void lm_advance_team_kernel(
    lm_state *states, int64_t nobj, const int64_t *__restrict__ obj_start,
    const int32_t *__restrict__ stamp_band, const double *__restrict__ sums, int nloc,
    const double *__restrict__ obj_sums, int32_t *nactive,
    const double *__restrict__ stamp_stats, double *__restrict__ obj_stats)
{
    constexpr int np = NP;
    using lmteam::TEAM;
    extern __shared__ double team_lds[];
    const int team = threadIdx.x / TEAM;
    const int64_t o = blockIdx.x * (int64_t)TEAMS + team;
    if (o >= nobj) return;
    lm_state &G = states[o];
    if (G.phase == LM_PHASE_DONE) return;
    lmteam::Fit f;
    f.lane = threadIdx.x % TEAM;
    double *block = team_lds + (size_t)team * lmteam::team_lds_doubles(np);
    lmteam::carve(f, block, np);
    // the block starts as +0.0 everywhere: the batched loads of lm_core_team.hpp
    // read whole rows / columns / vectors of NP elements and select afterwards
    for (int i = f.lane; i < lmteam::team_lds_doubles(np); i += TEAM) block[i] = 0.0;
    lmteam::tsync();
    const int n = G.n;
    if (n > np || n < 1) {
        // the caller's parameter-count hint was wrong for this fit: end it as
        // MINPACK ends a call with improper input (see lm_advance_dispatch)
        if (f.lane == 0) {
            G.info = 0;
            G.phase = LM_PHASE_DONE;
        }
        return;
    }
    // ---- the live part of the record: scalars to every lane, element j of the
    // vectors to lane j (registers or LDS), row j of R to LDS
    lmteam::Scal &s = f.s;
    s.n = n;
    s.iter = G.iter;
    s.nfev = G.nfev;
    s.njev = G.njev;
    s.info = G.info;
    s.phase = G.phase;
    s.maxfev = G.maxfev;
    s.mode = G.mode;
    s.bounded = G.bounded;
    s.fonly = G.fonly;
    s.fnorm = G.fnorm;
    s.xnorm = G.xnorm;
    s.delta = G.delta;
    s.par = G.par;
    s.gnorm = G.gnorm;
    s.pnorm = G.pnorm;
    s.ftol = G.ftol;
    s.xtol = G.xtol;
    s.gtol = G.gtol;
    s.factor = G.factor;
    const int iter0 = s.iter, phase0 = s.phase;
    f.x = f.xt = f.xi = f.xti = f.xstep = f.hstep = 0.0;
    f.lo = -INFINITY;
    f.hi = INFINITY;
    TFOR(j, 0, n) {
        f.x = G.x[j];
        f.xt = G.xt[j];
        f.xi = G.xi[j];
        f.xti = G.xti[j];
        f.lo = G.lo[j];
        f.hi = G.hi[j];
        f.xstep = G.xstep[j];
        f.hstep = G.hstep[j];
        f.diag[j] = G.diag[j];
        f.qtf[j] = G.qtf[j];
        f.step[j] = G.step[j];
        f.ipvt[j] = G.ipvt[j];
        // (the factor lives packed in LDS: its upper triangle, lm_core_team.hpp)
        for (int k = j; k < n; k++) f.R[lmteam::tri_row(np, j) + k] = G.R[j * LM_NPMAX + k];
    }
    const int64_t s0 = obj_start ? obj_start[o] : o;
    const int64_t s1 = obj_start ? obj_start[o + 1] : o + 1;
    {
        // a band the fit has no flux for (or nloc inconsistent with the state):
        // its sums would land outside the team's block, in a neighbouring fit's
        // arrays.  End the fit as for a wrong parameter-count hint.
        bool bad = nloc > n;
        for (int64_t st = s0 + f.lane; st < s1 && stamp_band; st += TEAM) {
            const int band = stamp_band[st];
            if (band < 0 || nloc - 1 + band >= n) bad = true;
        }
        const unsigned long long team_mask = 0xffffull << (TEAM * team);
        if (__ballot(bad) & team_mask) {
            if (f.lane == 0) {
                G.info = 0;
                G.phase = LM_PHASE_DONE;
            }
            return;
        }
    }
    const double ff = lm_team_fold(
        f, s0, s1, stamp_band, sums, nloc,
        obj_sums ? obj_sums + o * (int64_t)(n * (n + 1) / 2 + n + 1) : nullptr);
    lmteam::Step<NP>::lm_advance(f, ff);
    lmteam::tsync();
    TFOR(j, 0, n) {
        G.x[j] = f.x;
        G.xt[j] = f.xt;
        G.xi[j] = f.xi;
        G.xti[j] = f.xti;
        G.xstep[j] = f.xstep;
        G.hstep[j] = f.hstep;
        G.diag[j] = f.diag[j];
        G.qtf[j] = f.qtf[j];
        G.step[j] = f.step[j];
        G.ipvt[j] = f.ipvt[j];
        // (below the diagonal the record holds the zeros the one-thread form writes)
        for (int k = 0; k < n; k++)
            G.R[j * LM_NPMAX + k] = k >= j ? f.R[lmteam::tri_row(np, j) + k] : 0.0;
    }
    if (f.lane != 0) return;
    G.iter = s.iter;
    G.nfev = s.nfev;
    G.njev = s.njev;
    G.info = s.info;
    G.phase = s.phase;
    G.fonly = s.fonly;
    G.fnorm = s.fnorm;
    G.xnorm = s.xnorm;
    G.delta = s.delta;
    G.par = s.par;
    G.gnorm = s.gnorm;
    G.pnorm = s.pnorm;
    if (s.phase != LM_PHASE_DONE && nactive) atomicAdd(nactive, 1);
    if (stamp_stats && obj_stats && (phase0 == LM_PHASE_INIT || s.iter != iter0)) {
        double a = 0.0, b = 0.0;
        for (int64_t st = s0; st < s1; st++) {
            a += stamp_stats[2 * st];
            b += stamp_stats[2 * st + 1];
        }
        obj_stats[2 * o] = a;
        obj_stats[2 * o + 1] = b;
    }
}

int launch_lm_advance_team(lm_state *states, int64_t nobj, const int64_t *obj_start,
                           const int32_t *stamp_band, const double *sums, int nloc, int npars,
                           const double *obj_sums, int32_t *nactive,
                           const double *stamp_stats, double *obj_stats, int teams,
                           hipStream_t s)
{
    const int np = npars <= 8 ? 8 : npars <= 10 ? 10 : npars <= 12 ? 12 : LM_NPMAX;
    // (fits per wave: 1, 2 or 4 -- anything else an A/B knob says is the default)
    if (teams != 1 && teams != 2) teams = 4;
    char name[64];
    snprintf(name, sizeof(name), "lm_advance_team_kernel<%d, %d>", teams, np);
    census(name);
#define NGMIX_TEAM_LAUNCH(T, N)                                                                 \
    hipLaunchKernelGGL((lm_advance_team_kernel<T, N>), dim3((unsigned)((nobj + T - 1) / T)),      \
                       dim3(T * lmteam::TEAM),                                                    \
                       T * (size_t)lmteam::team_lds_doubles(N) * sizeof(double), s, states, nobj, \
                       obj_start, stamp_band, sums, nloc, obj_sums, nactive, stamp_stats,         \
                       obj_stats)
#define NGMIX_TEAM_NP(T)                                                                         \
    do {                                                                                          \
        if (np == 8) NGMIX_TEAM_LAUNCH(T, 8);                                                     \
        else if (np == 10) NGMIX_TEAM_LAUNCH(T, 10);                                              \
        else if (np == 12) NGMIX_TEAM_LAUNCH(T, 12);                                              \
        else NGMIX_TEAM_LAUNCH(T, LM_NPMAX);                                                      \
    } while (0)
    if (teams == 1) NGMIX_TEAM_NP(1);
    else if (teams == 2) NGMIX_TEAM_NP(2);
    else NGMIX_TEAM_NP(4);
#undef NGMIX_TEAM_NP
#undef NGMIX_TEAM_LAUNCH
    NGMIX_HIP_CHECK(hipGetLastError());
    return NGMIX_OK;
}

}  // namespace ngmix
