// gausslq.hip — least-squares Gaussian fit (picasso/gausslq.py:206-300 fit_spot / fit_spots).
//
// The reference calls scipy.optimize.leastsq(ftol=1e-2, xtol=1e-2) per spot, i.e.
// MINPACK lmdif with a forward-difference Jacobian (epsfcn = float32 eps because
// the residual vector is float32), gtol = 0, factor = 100, maxfev = 1400, on the
// residuals of a point-sampled elliptical Gaussian whose model is stored in
// float32 (gausslq.py:151-203).  This file restates that algorithm for one group of lanes
// per spot — a 16-lane DPP row for boxes up to 7x7 (four spots per wavefront), half a wavefront
// for boxes 9..15 (two spots), the whole wavefront for larger boxes:
//
//   - residual row r = i*size + j lives in lane r % GS, element r / GS, so the
//     m x 6 Jacobian is six register columns per lane; column norms and the
//     Householder dot products of qrfac are DPP wave reductions in float64;
//   - the 6 x 6 triangular factor, lmpar and qrsolv are wave-uniform register code
//     (all indices static; the pivot permutation goes through select chains);
//   - all float64 arithmetic is unfused (contract off), as compiled MINPACK is.
//
// Sums over the m rows are tree reductions here and sequential loops in MINPACK;
// the difference is in the last bits of float64 and disappears in the float32 theta
// except when it flips a float32 rounding of the stored model.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "lq_common.h"

#pragma clang fp contract(off)

namespace pmi {
namespace lq {

// ---- strict mode: the sums over the m rows in MINPACK's own (sequential) order ------------------------------------
// The group kernels hold residual row r in lane r % GS, element r / GS, and add over the rows with DPP trees; MINPACK
// adds them one after the other (enorm, qrfac's Householder products, Q^T fvec).  The two agree to the last bits of
// float64 — which is everything except where one of lmdif's accept / reject / terminate tests is decided within those
// bits.  Spots on which that happens (flagged by lq_step_kernel and qrfac_step below) are fitted again with these
// forms: the rows go through LDS and every lane of the group adds them in row order.
// doubles of chain scratch per lane group: six columns of box^2 (+ 1) rows.  (LQ_SBUF_PAD: the PMC counts bank conflicts in 30 %
// of the strict Jacobian kernel's LDS cycles, so the groups' column blocks were moved apart by 2 / 4 / 8 / 10 doubles: 7x7
// 9.0 -> 9.2 / 9.7 / 9.7 / 9.7 ms per 2^20 spots, 13x13 70.9 -> 68.8 / 69.3 / 69.3 / 69.0, 5x5 unchanged.  Left at 0.)
#ifndef LQ_SBUF_PAD
#define LQ_SBUF_PAD 0
#endif
__host__ __device__ constexpr size_t lq_sbuf_doubles(int m) { return (size_t)6 * (size_t)((m + 1) & ~1) + LQ_SBUF_PAD; }
__device__ __forceinline__ void grp_sync()
{
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
}
template <int GS, int E>
__device__ __forceinline__ void rows_to_lds(const double (&v)[E], int lane, int m, double *buf)
{
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int r = lane + GS * e;
        if (r < m) buf[r] = v[e];
    }
    grp_sync();
}
template <int GS, int E>
__device__ __forceinline__ double seq_sum(const double (&v)[E], int lane, int row_lo, int m, double *buf)
{
    rows_to_lds<GS, E>(v, lane, m, buf);
    double acc = 0;
#pragma unroll 8
    for (int r = row_lo; r < m; r++) acc += buf[r];       // (the reads do not depend on acc: eight in flight per step)
    grp_sync();
    return acc;
}
template <int GS, int E>
__device__ __forceinline__ double seq_enorm(const double (&v)[E], int lane, int row_lo, int m, double *buf)
{
    rows_to_lds<GS, E>(v, lane, m, buf);
    EnormAcc acc(m - row_lo);
#pragma unroll 4
    for (int r = row_lo; r < m; r++) acc.add(buf[r]);
    grp_sync();
    return acc.norm();
}

// enorm over rows [row_lo, m) of a register column.  Mid-range components (all of
// them, for finite photon data) take one reduction; the scaled accumulators of
// MINPACK are only formed when a component is tiny, huge or NaN.
template <int GS, int E>
__device__ __forceinline__ double enorm_rows(const double (&v)[E], int lane, int row_lo, int m)
{
    const double agiant = RGIANT / (double)(m - row_lo);
    double s2 = 0;
    bool odd = false;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int r = lane + GS * e;
        if (r >= row_lo && r < m) {
            const double xabs = fabs(v[e]);
            if (xabs > RDWARF && xabs < agiant) s2 += xabs * xabs;
            else if (xabs != 0) odd = true;
        }
    }
    s2 = Grp<GS>::sum_d(s2);
    if (!Grp<GS>::any(odd)) return sqrt(s2);
    double big = 0, small = 0;
    bool isnan_ = false;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int r = lane + GS * e;
        if (r >= row_lo && r < m) {
            const double xabs = fabs(v[e]);
            if (xabs != xabs) isnan_ = true;
            else if (xabs >= agiant) big = fmax(big, xabs);
            else if (xabs <= RDWARF) small = fmax(small, xabs);
        }
    }
    if (Grp<GS>::any(isnan_)) return __builtin_nan("");
    const double x1max = Grp<GS>::max_d(big), x3max = Grp<GS>::max_d(small);
    double s1 = 0, s3 = 0;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int r = lane + GS * e;
        if (r >= row_lo && r < m) {
            const double xabs = fabs(v[e]);
            if (xabs >= agiant) { double q = xabs / x1max; s1 += q * q; }
            else if (xabs <= RDWARF && xabs != 0) { double q = xabs / x3max; s3 += q * q; }
        }
    }
    s1 = Grp<GS>::sum_d(s1);
    s3 = Grp<GS>::sum_d(s3);
    if (s1 != 0) return x1max * sqrt(s1 + (s2 / x1max) / x1max);
    if (s2 != 0) {
        if (s2 >= x3max) return sqrt(s2 * (1 + (x3max / s2) * (x3max * s3)));
        return sqrt(x3max * ((s2 / x3max) + (x3max * s3)));
    }
    return x3max * sqrt(s3);
}

template <int GS, bool CR = false>
__device__ __forceinline__ void profiles(const double (&th)[6], int size, int lane, int which, float (&prof)[GS == 8 ? 2 : 1], unsigned &fragile)
{
    constexpr int NPROF = GS == 8 ? 2 : 1;
    const int hsz = size / 2;
    const double th0 = th[0], th1 = th[1], th4 = th[4], th5 = th[5];   // values, not an lvalue select
#pragma unroll
    for (int k = 0; k < NPROF; k++) {
        if (!((which >> k) & 1)) continue;
        const bool isy = NPROF == 2 ? k == 1 : lane >= size;
        const int idx = NPROF == 2 ? lane : (isy ? lane - size : lane);
        const double mu = isy ? th1 : th0, sg = isy ? th5 : th4;
        const double g = (double)(float)(idx - hsz);
        const double t = (g - mu) / sg;
        const double nrm = 0.3989422804014327 / sg;
        const double pv = nrm * lq_exp<CR>(-0.5 * (t * t));
        fragile |= fragile_f32_rounding(pv);
        prof[k] = (float)pv;
    }
}

template <int GS, int E>
__device__ __forceinline__ void residuals(const double (&th)[6], const float (&prof)[GS == 8 ? 2 : 1], const float (&sp)[E],
                                          const int (&ri)[E], const int (&rj)[E], const bool (&act)[E], int size, int lane,
                                          float (&out)[E])
{
    constexpr int NPROF = GS == 8 ? 2 : 1;
    const int gbase = GS == 64 ? 0 : (int)((threadIdx.x & 63u) & ~(unsigned)(GS - 1));     // first lane of this group
#pragma unroll
    for (int e = 0; e < E; e++) {
        float mxv, myv;
        if (NPROF > 1) {
            mxv = __shfl(prof[0], gbase + (rj[e] & (GS - 1)));
            myv = __shfl(prof[NPROF - 1], gbase + (ri[e] & (GS - 1)));
        } else {
            mxv = __shfl(prof[0], gbase + (rj[e] & (GS - 1)));
            myv = __shfl(prof[0], gbase + ((size + ri[e]) & (GS - 1)));
        }
        const float model = (float)(th[2] * (double)myv * (double)mxv + th[3]);
        const float res = sp[e] - model;
        out[e] = act[e] ? res : 0.0f;          // a float32 value: the callers widen it where MINPACK works in float64
    }
}

// One column of MINPACK qrfac (pivot, Householder vector, update of the trailing
// columns and of their running norms).  rdiag = wa1, wa = wa3.
template <int GS, int E, int j, bool STRICT>
__device__ __forceinline__ void qrfac_step(double (&a)[6][E], double (&wa1)[6], double (&wa3)[6], int (&ipvt)[6],
                                           int lane, int m, double *sbuf, bool &tie)
{
    int kmax = j;
    double best = wa1[j];
#pragma unroll
    for (int k = j + 1; k < 6; k++)
        if (wa1[k] > best) { best = wa1[k]; kmax = k; }
    if (!STRICT) {
        // the pivot is the largest of the running column norms: a second candidate within their rounding error of it may
        // be MINPACK's choice.  A running norm that was scaled down by q = wa1 / wa3 since it was last computed carries a
        // relative error of ~ eps / q^2; q < LQ_RANK never gets here unflagged (the rank test after the factorisation), so
        // LQ_PIVOT_TIE = 1e-9 covers eps / LQ_RANK^2 with room
#pragma unroll
        for (int k = j; k < 6; k++)
            if (k != kmax && !(fabs(wa1[k] - best) > LQ_PIVOT_TIE * best)) tie = tie || wa1[k] != 0 || best != 0;
    }
#pragma unroll
    for (int k = j + 1; k < 6; k++) {
        // selects, not a branch: a conditional swap sinks into stores through pointer phis
        // and the arrays then stay in scratch memory
        const bool sw = kmax == k;
#pragma unroll
        for (int e = 0; e < E; e++) {
            const double t = a[j][e], u = a[k][e];
            a[j][e] = sw ? u : t;
            a[k][e] = sw ? t : u;
        }
        // Left to itself the compiler turns these selects on kmax into an indexed access and keeps the six norms in
        // scratch memory (48 B per lane, 35 scratch loads / stores per Jacobian); opaque operands keep them in registers
        // — except in the three instantiations that sit at 256 VGPRs, where the twelve registers come back as spills
        // inside the factorisation (7x7: 8.40 ms per 1e6 spots with the norms in scratch, 8.51 in registers; boxes 5,
        // 9, 13: 1 % the other way; alternating runs on one box).
        constexpr bool NORMS_IN_REGS = !((GS == 8 && E == 7) || (GS == 32 && E == 8) || (GS == 64 && E == 7));
        if constexpr (NORMS_IN_REGS) {
            wa1[k] = sw ? opaque(wa1[j]) : opaque(wa1[k]);
            wa3[k] = sw ? opaque(wa3[j]) : opaque(wa3[k]);
        } else {
            wa1[k] = sw ? wa1[j] : wa1[k];
            wa3[k] = sw ? wa3[j] : wa3[k];
        }
        const int t = ipvt[j], u = ipvt[k];
        ipvt[j] = sw ? u : t;
        ipvt[k] = sw ? t : u;
    }
    double ajnorm = STRICT ? seq_enorm<GS, E>(a[j], lane, j, m, sbuf) : enorm_rows<GS, E>(a[j], lane, j, m);
    if (ajnorm != 0) {
        if (Grp<GS>::bcast_d(a[j][0], j) < 0) ajnorm = -ajnorm;
#pragma unroll
        for (int e = 0; e < E; e++)
            if (e > 0 || lane >= j) a[j][e] /= ajnorm;
        if (lane == j) a[j][0] += 1;
        const double ajj = Grp<GS>::bcast_d(a[j][0], j);
#pragma unroll
        for (int k = j + 1; k < 6; k++) {
            double sum = 0;
            if (STRICT) {
                double prod[E];
#pragma unroll
                for (int e = 0; e < E; e++) prod[e] = a[j][e] * a[k][e];
                sum = seq_sum<GS, E>(prod, lane, j, m, sbuf);
            } else {
#pragma unroll
                for (int e = 0; e < E; e++)
                    if (e > 0 || lane >= j) sum += a[j][e] * a[k][e];
                sum = Grp<GS>::sum_d(sum);
            }
            double temp = sum / ajj;
#pragma unroll
            for (int e = 0; e < E; e++)
                if (e > 0 || lane >= j) a[k][e] -= temp * a[j][e];
            if (wa1[k] != 0) {
                temp = Grp<GS>::bcast_d(a[k][0], j) / wa1[k];
                const double t2 = 1 - temp * temp;
                wa1[k] *= sqrt(t2 > 0 ? t2 : 0);
                const double q = wa1[k] / wa3[k];
                if (!STRICT && fabs(0.05 * (q * q) - EPSMCH) <= 1e-3 * EPSMCH) tie = true;      // recompute or not: decided in the noise
                if (0.05 * (q * q) <= EPSMCH) {
                    wa1[k] = STRICT ? seq_enorm<GS, E>(a[k], lane, j + 1, m, sbuf) : enorm_rows<GS, E>(a[k], lane, j + 1, m);
                    wa3[k] = wa1[k];
                }
            }
        }
    }
    wa1[j] = -ajnorm;
}

// ---- strict mode: MINPACK's m-long sums as chains, one column per lane ---------------------------------------------
// A column's rows go to LDS padded to MP = GS * E slots (zeros before row_lo and from row m on: adding +0 to a sum that
// started at +0 never changes it), and a lane adds ITS column's slots one after the other in row order — lmdif's own
// order (enorm, qrfac's Householder products, Q^T fvec) — while the other lanes of the group do the same for the other
// columns of the step.  The LDS reads of sixteen slots are issued together (a dependent add per read pays the LDS
// latency per element otherwise: measured 100+ cycles per element).
template <int GS, int E>
__device__ __forceinline__ void col_to_lds(const double (&v)[E], int lane, int row_lo, int m, double *col)
{
    const int mp = (m + 1) & ~1;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int r = lane + GS * e;
        if (r < mp) col[r] = (r >= row_lo && r < m) ? v[e] : 0.0;
    }
}
// does enorm have to leave its common branch for this lane's rows of the column? (a component that is tiny, huge or NaN)
template <int GS, int E>
__device__ __forceinline__ bool col_is_odd(const double (&v)[E], int lane, int row_lo, int m)
{
    const double agiant = RGIANT / (double)(m - row_lo);
    bool odd = false;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int r = lane + GS * e;
        const double xabs = fabs(v[e]);
        odd = odd || (r >= row_lo && r < m && !((xabs > RDWARF && xabs < agiant) || xabs == 0));
    }
    return odd;
}
// MINPACK enorm of an LDS column holding n = m - row_lo components (zeros elsewhere): the common branch as a chain of
// squares; `odd` (this lane's column has a component outside (RDWARF, agiant)) takes the published scaled accumulators
__device__ __forceinline__ double chain_enorm(const double *col, int mp, int n, bool odd)
{
    const double s2 = chain_sum<true>(col, mp);
    double res = sqrt(s2);
    if (odd) {
        EnormAcc acc(n);
#pragma unroll 1
        for (int r = 0; r < mp; r++) acc.add(col[r]);        // (a zero leaves the accumulators as they are)
        res = acc.norm();
    }
    return res;
}

// Strict mode, one column of qrfac.  The rows of every trailing column k > j — and of the residual vector, which the
// Householder vectors transform exactly as they transform a column (lmdif's Q^T fvec loop: sum / a_jj with the sign
// moved, the same bits) — go to LDS as products a_ij * a_ik, and lane k - j - 1 of the group adds ITS column: one chain
// per step instead of one per column.  a[6] is the residual column.  sbuf: 6 columns of m (rounded up to even) doubles per group.
template <int GS, int E, int j>
__device__ __forceinline__ void qrfac_strict_step(double (&a)[7][E], double (&wa1)[6], double (&wa3)[6], int (&ipvt)[6],
                                                  int lane, int m, double *sbuf)
{
    const int MP = (m + 1) & ~1;
    int kmax = j;
    double best = wa1[j];
#pragma unroll
    for (int k = j + 1; k < 6; k++)
        if (wa1[k] > best) { best = wa1[k]; kmax = k; }
#pragma unroll
    for (int k = j + 1; k < 6; k++) {
        const bool sw = kmax == k;
#pragma unroll
        for (int e = 0; e < E; e++) {
            const double t = a[j][e], u = a[k][e];
            a[j][e] = sw ? u : t;
            a[k][e] = sw ? t : u;
        }
        wa1[k] = sw ? opaque(wa1[j]) : opaque(wa1[k]);
        wa3[k] = sw ? opaque(wa3[j]) : opaque(wa3[k]);
        const int t = ipvt[j], u = ipvt[k];
        ipvt[j] = sw ? u : t;
        ipvt[k] = sw ? t : u;
    }
    col_to_lds<GS, E>(a[j], lane, j, m, sbuf);
    const bool odd_j = Grp<GS>::any(col_is_odd<GS, E>(a[j], lane, j, m));
    grp_sync();
    double ajnorm = chain_enorm(sbuf, MP, m - j, odd_j);        // every lane of the group: the reads are broadcasts
    grp_sync();
    if (ajnorm != 0) {
        if (Grp<GS>::bcast_d(a[j][0], j) < 0) ajnorm = -ajnorm;
#pragma unroll
        for (int e = 0; e < E; e++)
            if (e > 0 || lane >= j) a[j][e] /= ajnorm;
        if (lane == j) a[j][0] += 1;
        const double ajj = Grp<GS>::bcast_d(a[j][0], j);
        // the 6 - j sums of this step as parallel chains
#pragma unroll
        for (int k = j + 1; k < 7; k++) {
            double prod[E];
#pragma unroll
            for (int e = 0; e < E; e++) prod[e] = a[j][e] * a[k][e];
            col_to_lds<GS, E>(prod, lane, j, m, sbuf + (size_t)(k - j - 1) * MP);
        }
        grp_sync();
        const double acc = chain_sum<false>(sbuf + (size_t)(lane < 6 - j ? lane : 5 - j) * MP, MP);   // lanes beyond the columns repeat the last one
        grp_sync();
#pragma unroll
        for (int k = j + 1; k < 7; k++) {
            const double sum = Grp<GS>::bcast_d(acc, k - j - 1);
            double temp = sum / ajj;
#pragma unroll
            for (int e = 0; e < E; e++)
                if (e > 0 || lane >= j) a[k][e] -= temp * a[j][e];
            if (k < 6 && wa1[k] != 0) {
                temp = Grp<GS>::bcast_d(a[k][0], j) / wa1[k];
                const double t2 = 1 - temp * temp;
                wa1[k] *= sqrt(t2 > 0 ? t2 : 0);
                const double q = wa1[k] / wa3[k];
                if (0.05 * (q * q) <= EPSMCH) {
                    wa1[k] = seq_enorm<GS, E>(a[k], lane, j + 1, m, sbuf);
                    wa3[k] = wa1[k];
                }
            }
        }
    }
    wa1[j] = -ajnorm;
}

// ---- the fit as two kernels per outer iteration ----------------------------------------------------------------
// lmdif alternates two kinds of work: (a) residuals, forward-difference Jacobian and its pivoted QR factorisation —
// m x 6 data, done by a GROUP of lanes per spot (DPP reductions, the Jacobian columns in registers); (b) the
// Levenberg-Marquardt step on the 6 x 6 factor (lmpar / qrsolv: serial chains of float64 divisions and square roots),
// the trial evaluation and the accept / reject logic — scalar work per spot.  Run inside the group kernel, (b) is
// executed identically by all 16 lanes of a group (6 % of the lanes do distinct work) and its unrolled code (235 KB,
// 63 spilled registers) was most of the kernel.  Here (b) runs ONE SPOT PER LANE in its own kernel, the state of a
// fit travels through a scratch record (LqState), and the spots that need another Jacobian are compacted into the
// list of the next round.

// (a): residuals at x, Jacobian, QR.  list == nullptr: spots [first, min(first + count, n)).
// Waves per SIMD the Jacobian kernel leaves room for: three where a lane holds at most four rows of a small box (boxes
// 3 and 5: 4 / 35 spilled values, 5x5 5.67 -> 5.42 ms per 1e6 spots), two elsewhere (7x7 at three: 230 spills, 7.6 -> 11.6 ms).
// (strict mode, 5x5: the chains' buffers come on top — 84 spilled values at three waves, so two there)
constexpr int lq_jacobian_min_waves(int GS, int E, bool STRICT) { return GS * E <= (STRICT ? 16 : 32) ? 3 : LQ_MIN_WAVES; }
// CR (strict mode's second pass): the exp of the profiles rounded correctly (exp_cr.h)
template <int GS, int E, bool FROM_MOVIE, bool STRICT, bool CR = false>
__global__ __launch_bounds__(LQ_WAVES * 64, lq_jacobian_min_waves(GS, E, STRICT)) void lq_jacobian_kernel(Params p, LqState st, const int32_t *__restrict__ list,
                                                                      const unsigned *__restrict__ list_n, int64_t count)
{
    constexpr int NGRP = 64 / GS;                              // spots per wavefront
    extern __shared__ __attribute__((aligned(16))) char s_rows[];      // strict mode: 6 columns of box^2 (+ 1) doubles per group of the workgroup (the columns a step sums, side by side)
    double *sbuf = STRICT ? reinterpret_cast<double *>(s_rows) + (size_t)(((threadIdx.x >> 6) * NGRP) + (threadIdx.x & 63) / GS) * lq_sbuf_doubles(p.box * p.box)
                          : nullptr;
    const int lane = (threadIdx.x & 63) % GS;                  // lane inside the group
    const int grp = (threadIdx.x & 63) / GS;
    const int64_t wave0 = (int64_t)blockIdx.x * LQ_WAVES + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * LQ_WAVES;
    int64_t n = p.N;
    if (p.d_n) { const int64_t dn = *p.d_n; n = dn < n ? dn : n; }
    int64_t items = list ? (int64_t)*list_n : (st.first + count < n ? count : n - st.first);
    const int size = p.box, m = size * size, hsz = size / 2;
    int ri[E], rj[E];
    bool act[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int r = lane + GS * e;
        act[e] = r < m;
        const int rr = act[e] ? r : 0;
        ri[e] = rr / size;
        rj[e] = rr - ri[e] * size;
    }
    const double eps = sqrt(1.1920928955078125e-07);     // sqrt(max(epsfcn, epsmch)), epsfcn = float32 eps

    for (int64_t w0 = wave0 * NGRP; w0 < items; w0 += nwaves * NGRP) {
        // a group past the end repeats the last item and does not store (keeps the groups in lockstep)
        const bool store = w0 + grp < items;
        const int64_t w = store ? w0 + grp : items - 1;
        const int64_t s = list ? (int64_t)list[w] : st.first + w;
        const int64_t ls = s - st.first;
#include "lq_jacobian_body.inc"
    }
}

// The fit reads a spot once per round in each of its kernels.  Gathering 7 x 14 bytes from the movie every time costs
// more than the arithmetic, so the fused path cuts the ROIs of a batch ONCE (localize.py:917-931, :1101-1112) into a
// scratch (count, box, box) float32 array — what get_spots returns — and the rounds read that.
__global__ void lq_cut_kernel(Params p, int64_t first, int64_t count, float *__restrict__ out)
{
    int64_t n = p.N;
    if (p.d_n) { const int64_t dn = *p.d_n; n = dn < n ? dn : n; }
    const int size = p.box, m = size * size, hsz = size / 2;
    const int64_t rows = first + count < n ? count : n - first;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < rows * m; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t w = t / m;
        const int k = (int)(t - w * m);
        const int i = k / size, j = k - i * size;
        const int64_t s = first + w;
        const float raw = load_movie_px(p.movie, p.dtype, ((int64_t)p.frame[s] * p.Y + (p.y[s] - hsz + i)) * p.X + (p.x[s] - hsz + j));
        out[t] = div_const((raw - p.baseline) * p.sensitivity, p.gdiv);
    }
}

// start values (gausslq.py:95-112), one spot per lane, float64 sums in the reference's row-major order
// list == nullptr: spots [first, first + count); else the *list_n spots of the list (the strict re-fit starts them again)
template <bool FROM_MOVIE>
__global__ __launch_bounds__(256) void lq_init_kernel(Params p, LqState st, int64_t count, const int32_t *__restrict__ list,
                                                      const unsigned *__restrict__ list_n)
{
    int64_t n = p.N;
    if (p.d_n) { const int64_t dn = *p.d_n; n = dn < n ? dn : n; }
    if (list) { const int64_t ln = (int64_t)*list_n; count = ln < count ? ln : count; }
    if ((int64_t)blockIdx.x * blockDim.x >= count) return;             // (the grid is sized for the worst case of a device-side list)
    extern __shared__ __attribute__((aligned(16))) char s_dyn[];      // [256] spot indices, then the tile of 256 x m floats
    int64_t (&s_idx)[256] = *reinterpret_cast<int64_t (*)[256]>(s_dyn);
    float *s_tile = reinterpret_cast<float *>(s_dyn + 256 * sizeof(int64_t));
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t s = list ? (w < count ? (int64_t)list[w] : -1) : st.first + w;
    const bool mine = w < count && s >= 0 && s < n;
    const int size = p.box, m = size * size, hsz = size / 2;
    const bool staged = m <= LQ_TILE_MAXPIX;
    s_idx[threadIdx.x] = mine ? s : -1;
    __syncthreads();
    if (staged) { stage_spots<FROM_MOVIE, 256>(p, s_idx, s_tile, m, size, FROM_MOVIE ? 0 : st.first); __syncthreads(); }
    if (!mine) return;
    int64_t fr = 0, y0 = 0, x0 = 0;
    if (FROM_MOVIE) { fr = p.frame[s]; y0 = p.y[s] - hsz; x0 = p.x[s] - hsz; }
    const float *mytile = s_tile + (size_t)threadIdx.x * m;
    auto px = [&](int i, int j) -> float {
        if (staged) return mytile[i * size + j];
        if (FROM_MOVIE) {
            const float raw = load_movie_px(p.movie, p.dtype, (fr * p.Y + (y0 + i)) * p.X + (x0 + j));
            return div_const((raw - p.baseline) * p.sensitivity, p.gdiv);
        }
        return p.spots[s * m + i * size + j];
    };
    float mn = px(0, 0);
    for (int i = 0; i < size; i++)
        for (int j = 0; j < size; j++) { const float v = px(i, j); if (mn == mn && (v < mn || v != v)) mn = v; }   // np.min: NaN propagates
    double sy = 0, sx = 0, sum = 0;
    for (int i = 0; i < size; i++)
        for (int j = 0; j < size; j++) {
            const double v = (double)(float)(px(i, j) - mn);
            sy += v * (double)i; sx += v * (double)j; sum += v;
        }
    if (sum <= 0.0) { sum = 0.01; sy = (size - 1) / 2.0; sx = (size - 1) / 2.0; }
    else { sy /= sum; sx /= sum; }
    float t1 = (float)sy, t0 = (float)sx;
    const float t2 = (float)(1.0 > sum ? 1.0 : sum);
    double sdy = 0, sdx = 0;
    for (int i = 0; i < size; i++)
        for (int j = 0; j < size; j++) {
            const double v = (double)(float)(px(i, j) - mn);
            const double dy = (double)i - (double)t1, dx = (double)j - (double)t0;
            sdy += v * (dy * dy);
            sdx += v * (dx * dx);
        }
    const float t5 = (float)sqrt(sdy / sum), t4 = (float)sqrt(sdx / sum);
    t0 = t0 - (float)hsz;
    t1 = t1 - (float)hsz;
    const int64_t ls = s - st.first;
    LQD(st, 0, ls) = (double)t0; LQD(st, 1, ls) = (double)t1; LQD(st, 2, ls) = (double)t2; LQD(st, 3, ls) = (double)mn;
    LQD(st, 4, ls) = (double)t4; LQD(st, 5, ls) = (double)t5;
    LQI(st, 8, ls) = -1;                                        // fresh: lq_step_kernel starts its counters
    LQI(st, 9, ls) = 0;
}

// (lq_step_spot: lq_common.h; lq_step_kernel and its launcher: gausslq_step.hip)
int launch_step(bool strict, const Params &p, const LqState &st, const int32_t *list, const unsigned *list_n, int64_t count, int32_t *next_list,
                unsigned *next_n, int32_t *tie_list, unsigned *tie_n, size_t step_lds, int64_t max_blocks, hipStream_t s);

// ---- the fits the rounds have left, finished on the device ------------------------------------------------------
// MINPACK's loop has no bound a host could queue ahead of (maxfev = 1400: up to 200 rounds), and asking the device how many
// spots are left costs a synchronisation per look.  After the rounds that finish every fit of photon data, ONE launch of
// this kernel takes whatever is still running — and the whole second pass of the refit / strict modes, whose list lives on
// the device: every wavefront owns chunks of 64 spots of the list and alternates (a) and (b) for them, eight... one spot per
// Jacobian (the widest lane group serves every box; the sums in MINPACK's order whatever the mode: those bits do not
// depend on the lane layout), one spot per lane in the step, until none of its spots is running.  Nothing in here waits for
// another wavefront.
// The lane group of the Jacobian follows the box as in lq_jacobian_kernel, in three shapes: 8 lanes x 7 rows up to 7x7 (eight of
// the wavefront's running spots per pass), 32 x 8 up to 15x15 (two), 64 x 7 above (one).  With one 64-lane group for every box
// a 3x3 batch — nine residuals for six parameters, a third of the fits still running after the queued rounds — took 57 ms per
// 2^20 spots instead of 17.
template <int GS, int E, bool CR, bool FLAG, bool FRAG, bool STRICT = true>       // (STRICT a parameter so that the tree-sum branch of the body is discarded)
__global__ __launch_bounds__(LQ_STEP_NT, LQ_STEP_MIN_WAVES) void lq_finish_kernel(Params p, LqState st, const int32_t *__restrict__ list,
                                                                                  const unsigned *__restrict__ list_n,
                                                                                  int32_t *__restrict__ tie_list, unsigned *__restrict__ tie_n)
{
    constexpr int NGRP = LQ_STEP_NT / GS;
    constexpr bool FROM_MOVIE = false;
    static_assert(STRICT, "the finishing kernel adds in MINPACK's order");
    static_assert(LQ_STEP_NT == 64, "one wavefront per workgroup");
    // dynamic LDS: [64] spot indices, the x profile of the current evaluation (box x 64 floats), the tile of 64 x m floats
    // (boxes up to 9), the chains' 6 columns of box^2 (+ 1) doubles per lane group
    extern __shared__ __attribute__((aligned(16))) char s_dyn[];
    const int size = p.box, m = size * size, hsz = size / 2;
    const bool staged = m <= LQ_TILE_MAXPIX;
    int64_t (&s_idx)[LQ_STEP_NT] = *reinterpret_cast<int64_t (*)[LQ_STEP_NT]>(s_dyn);
    float (*s_px)[LQ_STEP_NT] = reinterpret_cast<float (*)[LQ_STEP_NT]>(s_dyn + LQ_STEP_NT * sizeof(int64_t));
    float *s_tile = reinterpret_cast<float *>(s_dyn + LQ_STEP_NT * sizeof(int64_t) + (size_t)size * LQ_STEP_NT * sizeof(float));
    const int tid = threadIdx.x, lane = tid % GS, grp = tid / GS;
    double *sbuf = reinterpret_cast<double *>(s_dyn + ((LQ_STEP_NT * sizeof(int64_t) + (size_t)size * LQ_STEP_NT * sizeof(float) +
                                                         (staged ? (size_t)LQ_STEP_NT * m * sizeof(float) : 0) + 15) & ~(size_t)15)) +
                   (size_t)grp * lq_sbuf_doubles(m);
    const int64_t items = (int64_t)*list_n;
    int ri[E], rj[E];
    bool act[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int r = lane + GS * e;
        act[e] = r < m;
        const int rr = act[e] ? r : 0;
        ri[e] = rr / size;
        rj[e] = rr - ri[e] * size;
    }
    const double eps = sqrt(1.1920928955078125e-07);
    // a wavefront works its spots' Jacobians off one after the other: a short list is spread over all the wavefronts (K spots
    // each, the other lanes idle in the step) rather than packed 64 to a wavefront
    const int K = (int)std::min<int64_t>(LQ_STEP_NT, std::max<int64_t>(1, (items + gridDim.x - 1) / gridDim.x));
    for (int64_t base = (int64_t)blockIdx.x * K; base < items; base += (int64_t)gridDim.x * K) {
        const int64_t w = base + tid;
        const int64_t s_mine = (tid < K && w < items) ? (int64_t)list[w] : -1;
        int info = s_mine >= 0 ? LQI(st, 8, s_mine - st.first) : 1;            // -1 fresh, 0 running, > 0 done
        __syncthreads();
        s_idx[tid] = info > 0 ? -1 : s_mine;
        __syncthreads();
        if (staged) { stage_spots<FROM_MOVIE, LQ_STEP_NT>(p, s_idx, s_tile, m, size, st.first); __syncthreads(); }
        const float *mytile = s_tile + (size_t)tid * m;
        bool running = info <= 0;
        for (;;) {
            const unsigned long long run = __ballot(running);
            if (run == 0ull) break;
            // (a) a Jacobian for every running spot, NGRP of them per pass: group g takes the g-th of those left
            const int any_src = (int)__builtin_ctzll(run);
            for (unsigned long long rest = run; rest != 0ull;) {
                int src = -1;
#pragma unroll
                for (int g = 0; g < NGRP; g++) {
                    if (rest != 0ull) {
                        if (g == grp) src = (int)__builtin_ctzll(rest);
                        rest &= rest - 1ull;
                    }
                }
                // a group without a spot repeats one and does not store (keeps the groups in lockstep)
                const bool store = src >= 0;
                const int64_t s = (int64_t)__shfl((int)s_mine, store ? src : any_src);       // spot indices are below 2^31
                const int64_t ls = s - st.first;
#include "lq_jacobian_body.inc"
            }
            __threadfence();                                   // the factors the lanes above wrote, read by the lane of each spot below
            // (b) the step of every running spot
            if (running) {
                unsigned tie = 0u;
                info = lq_step_spot<FROM_MOVIE, FLAG, FRAG, CR>(p, st, s_mine, info, tie_list != nullptr, mytile, s_px, tid, staged, tie);
                if (info != 0) {
                    running = false;
                    if ((FLAG || FRAG) && tie && tie_list) {
                        tie_list[atomicAdd(tie_n, 1u)] = (int32_t)s_mine;
                        for (int b = 0; b < 8; b++)
                            if (tie & (1u << b)) atomicAdd(tie_n + 1 + b, 1u);
                    }
                }
            }
            __threadfence();                                   // ... and the state a step left, read by the next Jacobian
        }
    }
}

// strict mode, boxes up to 7x7: one image column per lane (gausslq_w.hip)
int launch_jacobian_w(const Params &p, const LqState &st, const int32_t *list, const unsigned *list_n, int64_t count, int64_t max_blocks, hipStream_t s);

template <bool FROM_MOVIE, bool STRICT, bool CR = false>
static int launch_jacobian(const Params &p, const LqState &st, const int32_t *list, const unsigned *list_n, int64_t count,
                            int cus, hipStream_t s, bool light = false)
{
    const int m = p.box * p.box;
    dim3 block(LQ_WAVES * 64);
    auto grid_for = [&](int spots_per_wave) {
        const int64_t waves = (count + spots_per_wave - 1) / spots_per_wave;
        return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((waves + LQ_WAVES - 1) / LQ_WAVES, (int64_t)cus * 16)));
    };
    // strict mode: m doubles of LDS per group (the rows of one column at a time, summed in MINPACK's order)
    auto lds_for = [&](int spots_per_wave) { return STRICT ? (size_t)LQ_WAVES * spots_per_wave * lq_sbuf_doubles(m) * sizeof(double) : (size_t)0; };
    static const bool g16 = tuning_env("PMI_LQ_GROUP16") != nullptr;      // A/B: the 16-lane groups for boxes up to 7
    // strict mode, first pass: the image columns on the lanes (gausslq_w.hip), every box
    if constexpr (STRICT && !CR && !FROM_MOVIE) {
        (void)g16; (void)lds_for; (void)block;
        return launch_jacobian_w(p, st, list, list_n, count, (int64_t)cus * (light ? 4 : 16), s);
    } else {
    // (more than 64 KB of dynamic LDS per workgroup has to be asked for, once per kernel)
#define LQ_JAC(GS, E, SPW) do { \
        if (lds_for(SPW) > 65536) { \
            static bool asked[PMI_MAX_DEVICES] = {};     /* the opt-in is recorded per device and kernel */ \
            const int dv = current_device(); \
            if (!__atomic_load_n(&asked[dv], __ATOMIC_ACQUIRE)) { \
                PMI_HIP(hipFuncSetAttribute((const void *)lq_jacobian_kernel<GS, E, FROM_MOVIE, STRICT, CR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
                __atomic_store_n(&asked[dv], true, __ATOMIC_RELEASE); \
            } \
        } \
        hipLaunchKernelGGL((lq_jacobian_kernel<GS, E, FROM_MOVIE, STRICT, CR>), grid_for(SPW), block, lds_for(SPW), s, p, st, list, list_n, count); \
    } while (0)
    if (p.box <= 7 && !g16) {
        // eight spots per wavefront: the scalar chains of the factorisation (norm updates, Householder scalings:
        // float64 divisions and square roots every lane of a group repeats) are shared by twice as many fits
        if (m <= 16) LQ_JAC(8, 2, 8);
        else if (m <= 32) LQ_JAC(8, 4, 8);
        else LQ_JAC(8, 7, 8);
    } else if (p.box <= 7) {
        if (m <= 16) LQ_JAC(16, 1, 4);
        else if (m <= 32) LQ_JAC(16, 2, 4);
        else LQ_JAC(16, 4, 4);
    } else if (p.box <= 15) {
        if (m <= 96) LQ_JAC(32, 3, 2);
        else if (m <= 128) LQ_JAC(32, 4, 2);
        else if (m <= 192) LQ_JAC(32, 6, 2);
        else LQ_JAC(32, 8, 2);
    } else {
        const int e = (m + 63) / 64;
        if (e <= 5) LQ_JAC(64, 5, 1);
        else LQ_JAC(64, 7, 1);
    }
#undef LQ_JAC
    return PMI_OK;
    }
}

// Rounds of (Jacobian + QR, step) over the spots still running that are queued before the finishing kernel takes over (boxes
// up to 7x7; more for larger boxes, see launch)
#ifndef LQ_ROUNDS
#define LQ_ROUNDS 5
#endif
#ifndef LQ_BATCH_LOG2
#define LQ_BATCH_LOG2 21      // spots per batch (state: 552 B per spot); every batch ends with one host synchronisation
#endif
// How the sums over the residual rows are run (pmi_gausslq_set_mode): PMI_LQ_FAST tree sums only, PMI_LQ_REFIT tree sums
// and a second fit in MINPACK's order for the spots with a decision near its threshold, PMI_LQ_STRICT (default) every
// spot in MINPACK's order from the start: theta, info and nfev are lmdif's on every spot, at 1.2x the time of REFIT on 7x7
static int g_lq_mode = PMI_LQ_STRICT;
static int lq_mode_now()
{
    static const char *env = getenv("PMI_LQ_MODE");       // "fast" | "refit" | "strict" overrides pmi_gausslq_set_mode
    if (env) {
        if (!strcmp(env, "fast")) return PMI_LQ_FAST;
        if (!strcmp(env, "strict")) return PMI_LQ_STRICT;
        if (!strcmp(env, "refit")) return PMI_LQ_REFIT;
    }
    return g_lq_mode;
}
// statistics of the calling thread's last fit: a device buffer of its own ([0] spots fitted again, [1..7] why), read when asked for
static thread_local const unsigned *g_lq_stats[2] = {nullptr, nullptr};          // [1]: the second frame range of a fused call
static thread_local unsigned g_lq_stats_generation = 0;
static thread_local hipEvent_t g_lq_stats_done[2] = {nullptr, nullptr};       // recorded after the statistics kernel of the fit (the caller's stream may be gone when they are read); the events belong to the device's side lane (runtime.hip), not to this thread
static thread_local int g_lq_stats_device = 0;
static thread_local int g_lq_stats_bank = 0;      // and the scratch bank they were taken from
static thread_local bool g_lq_stats_second = false;                               // the fit being queued is that second range
static thread_local int g_lq_rounds[2] = {0, 0};

__global__ void lq_stats_add_kernel(const unsigned *__restrict__ tie_n, unsigned *__restrict__ stats, int refitted)
{
    if (threadIdx.x == 0 && refitted) stats[0] += tie_n[0];
    if (threadIdx.x >= 1 && threadIdx.x < 9) stats[threadIdx.x] += tie_n[threadIdx.x];
}

template <bool CR, bool FLAG, bool FRAG>
static void launch_finish(int shape, dim3 grid, size_t lds, hipStream_t s, const Params &p, const LqState &st, const int32_t *list, const unsigned *list_n,
                          int32_t *tie_list, unsigned *tie_n)
{
    if (shape == 0) hipLaunchKernelGGL((lq_finish_kernel<8, 7, CR, FLAG, FRAG>), grid, dim3(LQ_STEP_NT), lds, s, p, st, list, list_n, tie_list, tie_n);
    else if (shape == 1) hipLaunchKernelGGL((lq_finish_kernel<32, 8, CR, FLAG, FRAG>), grid, dim3(LQ_STEP_NT), lds, s, p, st, list, list_n, tie_list, tie_n);
    else hipLaunchKernelGGL((lq_finish_kernel<64, 7, CR, FLAG, FRAG>), grid, dim3(LQ_STEP_NT), lds, s, p, st, list, list_n, tie_list, tie_n);
}

// One call = batches of 2 Mi spots; a batch = start values, LQ_ROUNDS rounds of (Jacobian + QR, step) over the spots
// still running — on photon data every fit is done by then —, one launch of lq_finish_kernel for whatever is left, then
// the second pass of the mode (refit: the spots with a decision near its threshold; strict: those with a float32
// rounding that hangs on the last bit of an exp) as start values + one lq_finish_kernel over the tie list.  Every
// count lives on the device: nothing here waits for the stream.
template <bool FROM_MOVIE_IN>
static int launch(Params p, hipStream_t s)
{
    constexpr bool FROM_MOVIE = false;             // the rounds always read (N, box, box) float32 spots (lq_cut_kernel)
    const int cus = device_cu_count();
    const int64_t BATCH = (int64_t)1 << LQ_BATCH_LOG2;
    const int64_t Ntotal = p.N;                    // a capacity when the row count lives on the device: batches past the rows exit at once
    const int64_t cap = std::min<int64_t>(Ntotal, BATCH);
    void *ptr = nullptr, *sptr = nullptr;
    int rc;
    const size_t bytes = (size_t)cap * (LQ_NSD * sizeof(double) + LQ_NSI * sizeof(int32_t) + 3 * sizeof(int32_t)) + 2048;
    if ((rc = scratch(SCR_STAGE_D, bytes, &ptr)) != PMI_OK) return rc;
    if ((rc = scratch(SCR_LQ_STATS, 64, &sptr)) != PMI_OK) return rc;
    unsigned *stats = (unsigned *)sptr;
    PMI_HIP(hipMemsetAsync(stats, 0, 64, s));
    LqState st;
    st.d = (double *)ptr;
    st.i = (int32_t *)(st.d + (size_t)cap * LQ_NSD);
    st.stride = cap;
    int32_t *lists[2] = {st.i + (size_t)cap * LQ_NSI, st.i + (size_t)cap * (LQ_NSI + 1)};
    int32_t *tie_list = st.i + (size_t)cap * (LQ_NSI + 2);
    unsigned *counters = (unsigned *)(tie_list + cap);           // one per round + the tie counter, zeroed per batch
    constexpr int NCTR = 64;
    static_assert(LQ_ROUNDS >= 1 && LQ_ROUNDS + 7 + 36 <= NCTR, "one counter per round");
    unsigned *tie_n = counters + NCTR;
    const int mode = lq_mode_now();
    const bool no_strict = mode == PMI_LQ_FAST, all_strict = mode == PMI_LQ_STRICT;
    float *cut = nullptr;
    const int mpix = p.box * p.box;
    if (FROM_MOVIE_IN) {
        void *cptr = nullptr;
        if ((rc = scratch(SCR_STAGE_C, (size_t)cap * mpix * sizeof(float), &cptr)) != PMI_OK) return rc;
        cut = (float *)cptr;
    }
    const bool staged = mpix <= LQ_TILE_MAXPIX;
    const size_t init_lds = 256 * sizeof(int64_t) + (staged ? (size_t)256 * mpix * sizeof(float) : 0);
    const size_t step_lds = LQ_STEP_NT * sizeof(int64_t) + (size_t)p.box * LQ_STEP_NT * sizeof(float) +
                            (staged ? (size_t)LQ_STEP_NT * mpix * sizeof(float) : 0);
    const int fin_shape = p.box <= 7 ? 0 : (p.box <= 15 ? 1 : 2);          // lq_finish_kernel's lane group: 8 x 7, 32 x 8, 64 x 7
    const size_t fin_lds = ((step_lds + 15) & ~(size_t)15) + (size_t)(fin_shape == 0 ? 8 : (fin_shape == 1 ? 2 : 1)) * lq_sbuf_doubles(mpix) * sizeof(double);
    const dim3 fin_grid((unsigned)std::min<int64_t>((int64_t)cus * 8, std::max<int64_t>(1, cap)));
    // rounds queued before the finishing kernel: a fit of an n x n box takes (nfev - 1) / 7 of them — 2 to 4 at 7x7 (none left
    // after five), up to ten at 13x13, where the finishing kernel's one spot per Jacobian would be the slower way for many
    // (3x3: nine residuals for six parameters — a third of the fits are still running after five rounds, and a round costs little)
    const int full_rounds = p.box <= 3 ? LQ_ROUNDS + 7 : (p.box <= 7 ? LQ_ROUNDS : (p.box <= 9 ? LQ_ROUNDS + 1 : (p.box <= 13 ? LQ_ROUNDS + 4 : LQ_ROUNDS + 7)));
    // 3x3 (strict mode): a tenth of the fits is still running after the full rounds and the longest take another fifty —
    // light rounds (small grids, 10 ... 20 us each when their list is empty) before the finishing kernel, which works a
    // wavefront's 64 spots off eight Jacobians at a time
    const int rounds = full_rounds + (all_strict && p.box <= 3 ? 36 : 0);
    for (int64_t first = 0; first < Ntotal; first += BATCH) {
        const int64_t count = std::min<int64_t>(BATCH, Ntotal - first);
        st.first = first;
        if (FROM_MOVIE_IN) {
            const unsigned cb = (unsigned)std::min<int64_t>((count * mpix + 255) / 256, 65536);
            hipLaunchKernelGGL(lq_cut_kernel, dim3(cb), dim3(256), 0, s, p, first, count, cut);
            p.spots = cut - first * mpix;          // indexed by the absolute spot number
        }
        PMI_HIP(hipMemsetAsync(counters, 0, (NCTR + 16) * sizeof(unsigned), s));
        hipLaunchKernelGGL((lq_init_kernel<FROM_MOVIE>), dim3((unsigned)((count + 255) / 256)), dim3(256), init_lds, s, p, st, count,
                           (const int32_t *)nullptr, (const unsigned *)nullptr);
        // ---- pass 0: every spot of the batch (refit / fast: tree sums, decisions near a threshold collected in tie_list;
        // strict: MINPACK's order, fragile float32 roundings collected there)
        const int32_t *cur = nullptr;
        const unsigned *cur_n = nullptr;
        for (int round = 0; round < rounds; round++) {
            int32_t *nxt = lists[round & 1];
            unsigned *nxt_n = counters + round;
            // a round past the full ones runs on small grids: few spots are left, how many only the device knows
            const bool light = round >= full_rounds;
            if ((rc = all_strict ? launch_jacobian<FROM_MOVIE, true>(p, st, cur, cur_n, count, cus, s, light)
                                 : launch_jacobian<FROM_MOVIE, false>(p, st, cur, cur_n, count, cus, s, light)) != PMI_OK) return rc;
            if ((rc = launch_step(all_strict, p, st, cur, cur_n, count, nxt, nxt_n, tie_list, tie_n, step_lds,
                                  light ? (int64_t)cus * 8 : (int64_t)1 << 40, s)) != PMI_OK) return rc;
            cur = nxt; cur_n = nxt_n;
        }
        if (all_strict) launch_finish<false, false, true>(fin_shape, fin_grid, fin_lds, s, p, st, cur, cur_n, tie_list, tie_n);
        else launch_finish<false, true, false>(fin_shape, fin_grid, fin_lds, s, p, st, cur, cur_n, tie_list, tie_n);
        hipLaunchKernelGGL(lq_stats_add_kernel, dim3(1), dim3(64), 0, s, (const unsigned *)tie_n, stats, no_strict ? 0 : 1);
        // ---- pass 1: the spots of tie_list again from their start values (refit: sequential sums; strict: exp rounded correctly)
        if (!no_strict) {
            hipLaunchKernelGGL((lq_init_kernel<FROM_MOVIE>), dim3((unsigned)((count + 255) / 256)), dim3(256), init_lds, s, p, st, count,
                               (const int32_t *)tie_list, (const unsigned *)tie_n);
            if (all_strict) launch_finish<true, false, false>(fin_shape, fin_grid, fin_lds, s, p, st, tie_list, tie_n, nullptr, nullptr);
            else launch_finish<false, false, false>(fin_shape, fin_grid, fin_lds, s, p, st, tie_list, tie_n, nullptr, nullptr);
        }
        PMI_HIP(hipGetLastError());
    }
    g_lq_stats_device = current_device();
    g_lq_stats_bank = scratch_user_bank();
    SideLane *lane = nullptr;
    { const int rc_lane = side_lane(1, &lane); if (rc_lane != PMI_OK) return rc_lane; }
    hipEvent_t done = g_lq_stats_done[g_lq_stats_second ? 1 : 0] = lane->stats_done[g_lq_stats_second ? 1 : 0];
    PMI_HIP(hipEventRecord(done, s));
    g_lq_stats[g_lq_stats_second ? 1 : 0] = stats;
    if (!g_lq_stats_second) g_lq_stats[1] = nullptr;
    g_lq_stats_generation = scratch_generation_of(g_lq_stats_device, g_lq_stats_bank, SCR_LQ_STATS);
    g_lq_rounds[0] = rounds; g_lq_rounds[1] = no_strict ? 0 : 1;
    return PMI_OK;
}

static int read_lq_stats(unsigned (&h)[16])
{
    for (unsigned &v : h) v = 0;
    if (!g_lq_stats[0] || g_lq_stats_generation != scratch_generation_of(g_lq_stats_device, g_lq_stats_bank, SCR_LQ_STATS)) return PMI_OK;      // no fit yet, or its buffers are gone
    for (int k = 0; k < 2; k++) {
        if (!g_lq_stats[k]) continue;
        unsigned part[16];
        PMI_HIP(hipEventSynchronize(g_lq_stats_done[k]));
        PMI_HIP(hipMemcpy(part, g_lq_stats[k], 64, hipMemcpyDeviceToHost));
        for (int i = 0; i < 16; i++) h[i] += part[i];
    }
    return PMI_OK;
}

static int check_box(int box)
{
    // scipy refuses m < n ("func input vector length N=6 must not exceed func output vector length M")
    if (box < 3 || box > PMI_MAX_BOX || (box & 1) == 0) { set_error("gausslq needs an odd box in [3, %d], got %d", PMI_MAX_BOX, box); return PMI_ERR_ARG; }
    return PMI_OK;
}

// ---- locs_from_fits (gausslq.py:404-484, :547-589) -------------------------
// float32 array arithmetic of the reference, one rounding per operation.
__device__ __forceinline__ float lq_precision(float photons, float s, float s_orth, float bg, int em)
{
    const float s2 = s * s;
    const float sa2 = s2 + (float)(1.0 / 12.0);
    const float sa = sqrtf(sa2);
    const float sa_orth2 = s_orth * s_orth + (float)(1.0 / 12.0);
    const float sa_orth = sqrtf(sa_orth2);
    float v = (float)(8.0 * 3.141592653589793) * sa;
    v = v * sa_orth;
    v = v * bg;
    v = v / photons;
    v = (float)(16.0 / 9.0) + v;
    v = sa2 * v;
    v = v / photons;
    if (em) v = v * 2.0f;
    return sqrtf(v);
}

struct LqCols { void *c[PMI_LQ_COLUMNS]; };
__global__ void locs_from_fits_lq_kernel(const int32_t *__restrict__ frame, const int32_t *__restrict__ y,
                                         const int32_t *__restrict__ x, const float *__restrict__ ng,
                                         const float *__restrict__ th, int64_t N, const int64_t *__restrict__ d_n,
                                         int em, LqCols cols, const int64_t *__restrict__ d_row0)
{
    int64_t n = N;
    if (d_n) { const int64_t dn = *d_n; n = dn < n ? dn : n; }
    const int64_t src = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // row of the fit arrays
    if (src >= n) return;
    const int64_t i = src + (d_row0 ? *d_row0 : 0);                           // row of the table: a second frame range follows the first
    const float *t = th + src * 6;
    ((uint32_t *)cols.c[0])[i] = (uint32_t)frame[src];
    ((float *)cols.c[1])[i] = (float)((double)t[0] + (double)x[src]);
    ((float *)cols.c[2])[i] = (float)((double)t[1] + (double)y[src]);
    ((float *)cols.c[3])[i] = t[2];
    ((float *)cols.c[4])[i] = t[4];
    ((float *)cols.c[5])[i] = t[5];
    ((float *)cols.c[6])[i] = t[3];
    ((float *)cols.c[7])[i] = lq_precision(t[2], t[4], t[5], t[3], em);
    ((float *)cols.c[8])[i] = lq_precision(t[2], t[5], t[4], t[3], em);
    const float a = np_maxf(t[4], t[5]), b = np_minf(t[4], t[5]);
    ((float *)cols.c[9])[i] = (a - b) / a;
    ((float *)cols.c[10])[i] = ng[src];
}

}  // namespace lq

int identify_impl(const void *d_movie, int dtype, int64_t F, int64_t Y, int64_t X, int box, double min_ng,
                  const int64_t *roi4, int64_t f_lo, int64_t f_hi, int64_t label_offset,
                  int32_t *d_frame, int32_t *d_y, int32_t *d_x, float *d_ng, int64_t cap, int64_t *d_out_n,
                  hipStream_t s);

}  // namespace pmi

extern "C" {

int pmi_gausslq_set_mode(int mode)
{
    using namespace pmi;
    if (mode != PMI_LQ_FAST && mode != PMI_LQ_REFIT && mode != PMI_LQ_STRICT) { set_error("unknown gausslq mode %d", mode); return PMI_ERR_ARG; }
    lq::g_lq_mode = mode;
    return PMI_OK;
}

int pmi_gausslq_get_mode(int *mode)
{
    if (mode) *mode = pmi::lq::lq_mode_now();
    return PMI_OK;
}

int pmi_gausslq_last_refit_count(int64_t *n_refit)
{
    unsigned h[16];
    const int rc = pmi::lq::read_lq_stats(h);
    if (rc != PMI_OK) return rc;
    if (n_refit) *n_refit = h[0];
    return PMI_OK;
}

int pmi_gausslq_last_tie_reasons(int64_t *counts, int n)
{
    unsigned h[16];
    const int rc = pmi::lq::read_lq_stats(h);
    if (rc != PMI_OK) return rc;
    for (int i = 0; counts && i < n; i++) counts[i] = i < 8 ? h[1 + i] : (i < 10 ? pmi::lq::g_lq_rounds[i - 8] : 0);
    return PMI_OK;
}

int pmi_gausslq_dev(const float *d_spots, int64_t N, const int64_t *d_n, int box, float *d_thetas,
                    int32_t *d_info, int32_t *d_nfev, void *stream)
{
    using namespace pmi;
    int rc = lq::check_box(box);
    if (rc != PMI_OK) return rc;
    if (N <= 0) return PMI_OK;
    lq::Params p = {};
    p.spots = d_spots; p.N = N; p.d_n = d_n; p.box = box; p.thetas = d_thetas; p.info = d_info; p.nfev = d_nfev;
    ScopedKernelTimer tm((hipStream_t)stream, &g_last_times.fit_ms);
    rc = lq::launch<false>(p, (hipStream_t)stream);
    tm.stop();
    return rc;
}

int pmi_gausslq_movie_dev(const void *d_movie, int dtype, int64_t F, int64_t Y, int64_t X,
                          const int32_t *d_frame, const int32_t *d_y, const int32_t *d_x, int64_t N,
                          const int64_t *d_n, int box, double baseline, double sensitivity, double gain,
                          float *d_thetas, int32_t *d_info, int32_t *d_nfev, void *stream)
{
    (void)F;
    using namespace pmi;
    int rc = lq::check_box(box);
    if (rc != PMI_OK) return rc;
    if (dtype < 0 || dtype > PMI_F32) { set_error("unknown dtype code %d", dtype); return PMI_ERR_ARG; }
    if (N <= 0) return PMI_OK;
    lq::Params p = {};
    p.movie = d_movie; p.dtype = dtype; p.Y = Y; p.X = X; p.frame = d_frame; p.y = d_y; p.x = d_x;
    p.baseline = (float)baseline; p.sensitivity = (float)sensitivity; p.gain = (float)gain; p.gdiv = make_const_div((float)gain);
    p.N = N; p.d_n = d_n; p.box = box; p.thetas = d_thetas; p.info = d_info; p.nfev = d_nfev;
    ScopedKernelTimer tm((hipStream_t)stream, &g_last_times.fit_ms);
    rc = lq::launch<true>(p, (hipStream_t)stream);
    tm.stop();
    return rc;
}

int pmi_gausslq(const float *spots, int64_t N, int box, float *thetas, int32_t *info, int32_t *nfev)
{
    using namespace pmi;
    if (pmi_device_count() < 1) { set_error("no HIP device"); return PMI_ERR_NODEVICE; }
    int rc = lq::check_box(box);
    if (rc != PMI_OK) return rc;
    if (N == 0) return PMI_OK;
    if (!spots || !thetas) { set_error("null pointer"); return PMI_ERR_ARG; }
    void *d_in = nullptr, *d_out = nullptr;
    const size_t in_bytes = (size_t)N * box * box * sizeof(float);
    if ((rc = scratch(SCR_STAGE_A, in_bytes, &d_in)) != PMI_OK) return rc;
    if ((rc = scratch(SCR_STAGE_B, (size_t)N * 8 * 4, &d_out)) != PMI_OK) return rc;
    float *d_th = (float *)d_out;
    int32_t *d_info = (int32_t *)(d_th + N * 6), *d_nfev = d_info + N;
    PMI_HIP(hipMemcpy(d_in, spots, in_bytes, hipMemcpyHostToDevice));
    rc = pmi_gausslq_dev((const float *)d_in, N, nullptr, box, d_th, d_info, d_nfev, nullptr);
    if (rc != PMI_OK) return rc;
    PMI_HIP(hipMemcpy(thetas, d_th, (size_t)N * 24, hipMemcpyDeviceToHost));
    if (info) PMI_HIP(hipMemcpy(info, d_info, (size_t)N * 4, hipMemcpyDeviceToHost));
    if (nfev) PMI_HIP(hipMemcpy(nfev, d_nfev, (size_t)N * 4, hipMemcpyDeviceToHost));
    return PMI_OK;
}

int pmi_locs_from_fits_lq_dev(const int32_t *d_frame, const int32_t *d_y, const int32_t *d_x, const float *d_ng,
                              const float *d_thetas, int64_t N, const int64_t *d_n, int em, void *const *d_cols,
                              void *stream)
{
    using namespace pmi;
    if (N <= 0) return PMI_OK;
    lq::LqCols cols;
    for (int c = 0; c < PMI_LQ_COLUMNS; c++) cols.c[c] = d_cols[c];
    const unsigned blocks = (unsigned)((N + 255) / 256);
    hipLaunchKernelGGL(lq::locs_from_fits_lq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_frame, d_y,
                       d_x, d_ng, d_thetas, N, d_n, em, cols, (const int64_t *)nullptr);
    PMI_HIP(hipGetLastError());
    return PMI_OK;
}

namespace pmi {
namespace lq {
// rows: [0] rows of A to fit, [1] rows of B to fit, [2] rows of A for the table, [3] rows of B for the table, [4] row offset of B
__global__ void lq_rows_a_kernel(const int64_t *__restrict__ n_a, int64_t cap, int64_t *__restrict__ rows) { rows[0] = *n_a > cap ? 0 : *n_a; }
__global__ void lq_rows_b_kernel(const int64_t *__restrict__ n_a, const int64_t *__restrict__ n_b, int64_t cap, int64_t *__restrict__ rows,
                                 int64_t *__restrict__ d_out_n)
{
    const int64_t a = *n_a, b = *n_b, total = a + b;
    const bool fits = total <= cap;
    rows[1] = fits ? b : 0;
    rows[2] = fits ? a : 0;
    rows[3] = fits ? b : 0;
    rows[4] = a;
    *d_out_n = total;
}
}  // namespace lq
}  // namespace pmi

// identify -> fused cut + least-squares fit -> table.  Nothing in here waits for the stream (the fit's loops live on the
// device, lq::launch), so a large frame range is cut in two like pmi_localize_mle_dev's: the scan of the second half — bound
// by the memory side — runs on a side stream of the library beside the fit of the first, which is bound by instruction issue;
// the table is written once both counts are known, A's rows, then B's.
int pmi_localize_lq_dev(const void *d_movie, int dtype, int64_t F, int64_t Y, int64_t X, int box, double min_ng,
                        const int64_t *roi4, int64_t f_lo, int64_t f_hi, double baseline, double sensitivity,
                        double gain, int em, void *d_table, int64_t cap, int64_t *d_out_n, void *stream)
{
    using namespace pmi;
    if (cap <= 0) { set_error("capacity must be positive"); return PMI_ERR_ARG; }
    int rc = lq::check_box(box);
    if (rc != PMI_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    struct Ids { int32_t *f, *y, *x; float *ng, *th; };
    auto carve = [&](void *ptr) {
        Ids d;
        d.f = (int32_t *)ptr; d.y = d.f + cap; d.x = d.y + cap;
        d.ng = (float *)(d.x + cap);
        d.th = d.ng + cap;
        return d;
    };
    const size_t ids_bytes = (size_t)cap * (16 + 6 * 4);
    lq::LqCols cols;
    for (int c = 0; c < PMI_LQ_COLUMNS; c++) cols.c[c] = (char *)d_table + (size_t)c * cap * 4;
    const unsigned tblocks = (unsigned)((cap + 255) / 256);
    void *ptr = nullptr, *cptr = nullptr;
    if ((rc = scratch(SCR_ROWS, 8 * sizeof(int64_t), &cptr)) != PMI_OK) return rc;
    int64_t *d_na = (int64_t *)cptr, *d_nb = d_na + 1, *rows = d_na + 2;
    const int64_t lo = f_lo < 0 ? 0 : f_lo, hi = f_hi > F - 1 ? F - 1 : f_hi, nf = hi - lo + 1;
    const bool two = g_localize_ranges == 2 && !g_kernel_timing && nf >= 16 && (double)nf * (double)Y * (double)X >= 2.5e8;
    if (!two) {
        if ((rc = scratch(SCR_IDS, ids_bytes, &ptr)) != PMI_OK) return rc;
        const Ids d = carve(ptr);
        rc = identify_impl(d_movie, dtype, F, Y, X, box, min_ng, roi4, f_lo, f_hi, 0, d.f, d.y, d.x, d.ng, cap, d_out_n, s);
        if (rc != PMI_OK) return rc;
        hipLaunchKernelGGL(lq::lq_rows_a_kernel, dim3(1), dim3(1), 0, s, (const int64_t *)d_out_n, cap, rows);
        rc = pmi_gausslq_movie_dev(d_movie, dtype, F, Y, X, d.f, d.y, d.x, cap, rows + 0, box, baseline, sensitivity,
                                   gain, d.th, nullptr, nullptr, stream);
        if (rc != PMI_OK) return rc;
        hipLaunchKernelGGL(lq::locs_from_fits_lq_kernel, dim3(tblocks), dim3(256), 0, s, d.f, d.y, d.x, d.ng, d.th, cap,
                           (const int64_t *)(rows + 0), em, cols, (const int64_t *)nullptr);
        PMI_HIP(hipGetLastError());
        return PMI_OK;
    }
    SideLane *side_p = nullptr;
    if ((rc = side_lane(1, &side_p)) != PMI_OK) return rc;
    SideLane &side = *side_p;
    const int64_t mid = lo + nf / 2 - 1;                      // A = [lo, mid], B = [mid + 1, hi]
    if ((rc = scratch(SCR_IDS, ids_bytes, &ptr)) != PMI_OK) return rc;
    const Ids a = carve(ptr);
    PMI_HIP(hipEventRecord(side.ev_start, s));
    PMI_HIP(hipStreamWaitEvent(side.s2, side.ev_start, 0));
    struct Join {       // whatever happens, the caller's stream is ordered after the side stream before this call returns
        SideLane &sd; hipStream_t st; bool done = false;
        ~Join() { if (!done) { (void)hipEventRecord(sd.ev_b, sd.s2); (void)hipStreamWaitEvent(st, sd.ev_b, 0); } }
    } join{side, s};
    // ---- range A on the caller's stream
    rc = identify_impl(d_movie, dtype, F, Y, X, box, min_ng, roi4, lo, mid, 0, a.f, a.y, a.x, a.ng, cap, d_na, s);
    if (rc != PMI_OK) return rc;
    hipLaunchKernelGGL(lq::lq_rows_a_kernel, dim3(1), dim3(1), 0, s, (const int64_t *)d_na, cap, rows);
    PMI_HIP(hipEventRecord(side.ev_scan_a, s));
    rc = pmi_gausslq_movie_dev(d_movie, dtype, F, Y, X, a.f, a.y, a.x, cap, rows + 0, box, baseline, sensitivity, gain, a.th, nullptr,
                               nullptr, s);
    if (rc != PMI_OK) return rc;
    // ---- range B on the side stream, scratch from the inner bank; its scan starts when scan A is done
    PMI_HIP(hipStreamWaitEvent(side.s2, side.ev_scan_a, 0));
    const int outer = scratch_enter_inner();
    Ids b2 = {};
    rc = scratch(SCR_IDS, ids_bytes, &ptr);
    if (rc == PMI_OK) {
        b2 = carve(ptr);
        rc = identify_impl(d_movie, dtype, F, Y, X, box, min_ng, roi4, mid + 1, hi, 0, b2.f, b2.y, b2.x, b2.ng, cap, d_nb, side.s2);
    }
    if (rc == PMI_OK) {
        hipLaunchKernelGGL(lq::lq_rows_b_kernel, dim3(1), dim3(1), 0, side.s2, (const int64_t *)d_na, (const int64_t *)d_nb, cap, rows, d_out_n);
        lq::g_lq_stats_second = true;
        rc = pmi_gausslq_movie_dev(d_movie, dtype, F, Y, X, b2.f, b2.y, b2.x, cap, rows + 1, box, baseline, sensitivity, gain, b2.th,
                                   nullptr, nullptr, side.s2);
        lq::g_lq_stats_second = false;
    }
    scratch_leave_inner(outer);
    if (rc != PMI_OK) return rc;
    PMI_HIP(hipEventRecord(side.ev_b, side.s2));
    PMI_HIP(hipStreamWaitEvent(s, side.ev_b, 0));
    join.done = true;
    hipLaunchKernelGGL(lq::locs_from_fits_lq_kernel, dim3(tblocks), dim3(256), 0, s, a.f, a.y, a.x, a.ng, a.th, cap,
                       (const int64_t *)(rows + 2), em, cols, (const int64_t *)nullptr);
    hipLaunchKernelGGL(lq::locs_from_fits_lq_kernel, dim3(tblocks), dim3(256), 0, s, b2.f, b2.y, b2.x, b2.ng, b2.th, cap,
                       (const int64_t *)(rows + 3), em, cols, (const int64_t *)(rows + 4));
    PMI_HIP(hipGetLastError());
    return PMI_OK;
}

}  // extern "C"
