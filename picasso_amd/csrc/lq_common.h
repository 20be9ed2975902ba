// lq_common.h — what the least-squares kernels of gausslq.hip and gausslq_w.hip share: MINPACK constants, the parameter and
// state records, lane-group primitives, the 6-vector helpers, enorm, the float32-rounding test of a profile value, the chains.
#pragma once
#include "fit_common.h"
#include "exp_cr.h"

#pragma clang fp contract(off)

namespace pmi {
namespace lq {

constexpr double EPSMCH = 2.220446049250313e-16;
constexpr double DWARF = 2.2250738585072014e-308;
constexpr double RDWARF = 3.834e-20, RGIANT = 1.304e19;
constexpr int LQ_WAVES = 4;
#ifndef LQ_MIN_WAVES
#define LQ_MIN_WAVES 2
#endif

struct Params {
    const float *spots;
    const void *movie;
    const int32_t *frame, *y, *x;
    int dtype;
    int64_t Y, X;
    float baseline, sensitivity, gain;
    ConstDiv gdiv;
    int64_t N;
    const int64_t *d_n;
    int box;
    float *thetas;
    int32_t *info, *nfev;
};

__device__ __forceinline__ double readlane_d(double v, int lane)
{
    long long b = __builtin_bit_cast(long long, v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), lane);
    int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_max_d(double v)
{
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    return v;
}

// A spot is fitted by a group of GS lanes: the whole wavefront (GS = 64) or one 16-lane DPP row
// (GS = 16, four spots per wavefront — for boxes up to 7x7 the 6x6 stage, which every lane of the
// group executes identically, is most of the work).  All lanes of a group follow the same control
// flow, so row-wide DPP and bpermute never read an inactive lane.
template <int GS> struct Grp;
template <> struct Grp<64> {
    static __device__ __forceinline__ double sum_d(double v) { return wave_sum_d(v); }
    static __device__ __forceinline__ double max_d(double v) { return wave_max_d(v); }
    static __device__ __forceinline__ float min_f(float v) { return wave_min(v); }
    static __device__ __forceinline__ double bcast_d(double v, int k) { return readlane_d(v, k); }
    static __device__ __forceinline__ bool any(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0; }
};
template <> struct Grp<16> {
    static __device__ __forceinline__ double sum_d(double v)
    {
        v += dpp_d<0xB1>(v);          // quad_perm [1,0,3,2]
        v += dpp_d<0x4E>(v);          // quad_perm [2,3,0,1]
        v += dpp_d<0x141>(v);         // row_half_mirror
        v += dpp_d<0x140>(v);         // row_mirror: every lane of the row holds the row sum
        return v;
    }
    static __device__ __forceinline__ double max_d(double v)
    {
        for (int off = 8; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off));
        return v;
    }
    static __device__ __forceinline__ float min_f(float v)
    {
        for (int off = 8; off >= 1; off >>= 1) v = fminf(v, __shfl_xor(v, off));
        return v;
    }
    static __device__ __forceinline__ double bcast_d(double v, int k)
    {
        return __shfl(v, (int)((threadIdx.x & 63u) & ~15u) + k);
    }
    static __device__ __forceinline__ bool any(bool c)
    {
        const unsigned long long b = __builtin_amdgcn_ballot_w64(c);
        return ((b >> ((threadIdx.x & 63u) & ~15u)) & 0xffffull) != 0;
    }
};
template <> struct Grp<8> {           // half a DPP row: eight spots per wavefront (boxes up to 7x7: 49 residuals = 7 per lane)
    static __device__ __forceinline__ double sum_d(double v)
    {
        v += dpp_d<0xB1>(v);          // quad_perm [1,0,3,2]
        v += dpp_d<0x4E>(v);          // quad_perm [2,3,0,1]
        v += dpp_d<0x141>(v);         // row_half_mirror: every lane of the half row holds its sum
        return v;
    }
    static __device__ __forceinline__ double max_d(double v)
    {
        for (int off = 4; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off));
        return v;
    }
    static __device__ __forceinline__ float min_f(float v)
    {
        for (int off = 4; off >= 1; off >>= 1) v = fminf(v, __shfl_xor(v, off));
        return v;
    }
    static __device__ __forceinline__ double bcast_d(double v, int k)
    {
        return __shfl(v, (int)((threadIdx.x & 63u) & ~7u) + k);
    }
    static __device__ __forceinline__ bool any(bool c)
    {
        const unsigned long long b = __builtin_amdgcn_ballot_w64(c);
        return ((b >> ((threadIdx.x & 63u) & ~7u)) & 0xffull) != 0;
    }
};
template <> struct Grp<32> {          // half a wavefront: two DPP rows, the partner row through bpermute
    static __device__ __forceinline__ double sum_d(double v)
    {
        v = Grp<16>::sum_d(v);
        return v + __shfl_xor(v, 16);
    }
    static __device__ __forceinline__ double max_d(double v)
    {
        for (int off = 16; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off));
        return v;
    }
    static __device__ __forceinline__ float min_f(float v)
    {
        for (int off = 16; off >= 1; off >>= 1) v = fminf(v, __shfl_xor(v, off));
        return v;
    }
    static __device__ __forceinline__ double bcast_d(double v, int k)
    {
        return __shfl(v, (int)((threadIdx.x & 63u) & ~31u) + k);
    }
    static __device__ __forceinline__ bool any(bool c)
    {
        const unsigned long long b = __builtin_amdgcn_ballot_w64(c);
        return ((b >> ((threadIdx.x & 63u) & ~31u)) & 0xffffffffull) != 0;
    }
};
// Indexing a 6-vector by a run-time (wave-uniform) index without leaving registers.
// The empty asm hides the loads from InstCombine, which otherwise rewrites the select
// chain into one load through a computed address and pins the array in scratch memory.
__device__ __forceinline__ double opaque(double v) { asm("" : "+v"(v)); return v; }
__device__ __forceinline__ double get6(const double (&a)[6], int i)
{
    double v = opaque(a[0]);
#pragma unroll
    for (int k = 1; k < 6; k++) { const double ak = opaque(a[k]); v = (i == k) ? ak : v; }
    return v;
}
__device__ __forceinline__ void set6(double (&a)[6], int i, double v)
{
#pragma unroll
    for (int k = 0; k < 6; k++) { const double ak = opaque(a[k]); a[k] = (i == k) ? v : ak; }
}

// MINPACK enorm over six wave-uniform values, sequential as published.
__device__ __forceinline__ double enorm6(const double (&x)[6])
{
    double s1 = 0, s2 = 0, s3 = 0, x1max = 0, x3max = 0;
    const double agiant = RGIANT / 6.0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const double xabs = fabs(x[i]);
        if (xabs > RDWARF && xabs < agiant) { s2 += xabs * xabs; }
        else if (xabs <= RDWARF) {
            if (xabs > x3max) { double r = x3max / xabs; s3 = 1 + s3 * r * r; x3max = xabs; }
            else if (xabs != 0) { double r = xabs / x3max; s3 += r * r; }
        } else {
            if (xabs > x1max) { double r = x1max / xabs; s1 = 1 + s1 * r * r; x1max = xabs; }
            else { double r = xabs / x1max; s1 += r * r; }
        }
    }
    if (s1 != 0) return x1max * sqrt(s1 + (s2 / x1max) / x1max);
    if (s2 != 0) {
        if (s2 >= x3max) return sqrt(s2 * (1 + (x3max / s2) * (x3max * s3)));
        return sqrt(x3max * ((s2 / x3max) + (x3max * s3)));
    }
    return x3max * sqrt(s3);
}

// MINPACK enorm, one component at a time in the published order (the trial residuals are never stored)
struct EnormAcc {
    double s1 = 0, s2 = 0, s3 = 0, x1max = 0, x3max = 0, agiant;
    __device__ __forceinline__ explicit EnormAcc(int n) : agiant(RGIANT / (double)n) {}
    __device__ __forceinline__ void add(double xv)
    {
        const double xabs = fabs(xv);
        if (xabs > RDWARF && xabs < agiant) { s2 += xabs * xabs; }
        else if (xabs <= RDWARF) {
            if (xabs > x3max) { double r = x3max / xabs; s3 = 1 + s3 * r * r; x3max = xabs; }
            else if (xabs != 0) { double r = xabs / x3max; s3 += r * r; }
        } else {
            if (xabs > x1max) { double r = x1max / xabs; s1 = 1 + s1 * r * r; x1max = xabs; }
            else { double r = xabs / x1max; s1 += r * r; }
        }
    }
    __device__ __forceinline__ double norm() const
    {
        if (s1 != 0) return x1max * sqrt(s1 + (s2 / x1max) / x1max);
        if (s2 != 0) {
            if (s2 >= x3max) return sqrt(s2 * (1 + (x3max / s2) * (x3max * s3)));
            return sqrt(x3max * ((s2 / x3max) + (x3max * s3)));
        }
        return x3max * sqrt(s3);
    }
};

// Residuals of the float32-stored model (gausslq.py:151-203).  The 2 * size profile values of an evaluation are spread
// over the lanes of the group: one per lane (x profile at flat index f = lane < size, y profile at f - size), except for
// the 8-lane groups, where lane l holds x-profile value l in slot 0 and y-profile value l in slot 1.  `which` = the
// slots to evaluate (bit 0: x / the only slot, bit 1: y): a forward difference in x0 or sx leaves the y profile as it
// is, one in the photons or the background leaves both — the values are the same function of the same arguments, so
// what is reused is bit for bit what would have been recomputed (6 instead of 14 float64 exp per lane and Jacobian in
// the 8-lane groups, 5 instead of 7 in the others).
// A profile value is stored in float32 (gausslq.py:203): nrm * exp(..) is rounded to float64, then to float32.  The
// device's exp and a CPU libm's are different functions within an ulp of the true one; where the float64 product sits
// within a few of its ulps of a float32 rounding boundary, the last bit of exp decides the stored value.  `fragile`
// reports that (strict mode: such a spot is fitted again with exp rounded correctly, exp_cr.h — 2 of 29 million fuzz
// spots ended 8e-5 px from the oracle before this); in the float32 subnormal range fewer bits are kept.
__device__ __forceinline__ unsigned fragile_f32_rounding(double p)
{
    // Does the float32 value change when p moves by 8 of its ulps either way?  In the normal float32 range 29 bits are
    // dropped: fragile within 8 of their midpoint 2^28 — three 32-bit instructions on the low word (this sits in the hot
    // profile loops of kernels that have no register to spare; asking the conversion itself, (float)(p (1 +- 2^-50)) != (float)p,
    // costs three quarter-rate conversions per value: 7x7 +5 %, 13x13 +19 %).  Among the float32 subnormals fewer bits are
    // kept and the test is not the right one — but a profile value below 1e-38 enters the model as photons x value + background,
    // rounded to float32 again: whichever way it rounds, nothing a float64 sum of the fit can see.
    return (unsigned)((((unsigned)__double2loint(p) & 0x1fffffffu) - 0x0ffffff8u) <= 16u);
}
template <bool CR> __device__ __forceinline__ double lq_exp(double x) { if constexpr (CR) return exp_cr(x); else return exp(x); }

typedef double d2_t __attribute__((ext_vector_type(2)));
template <int NB, bool SQUARE>
__device__ __forceinline__ double chain_batch(const __attribute__((address_space(3))) d2_t *p, double acc)
{
    d2_t v[NB];
#pragma unroll
    for (int i = 0; i < NB; i++) v[i] = p[i];
    if constexpr (NB == 8)
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
    else
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
#pragma unroll
    for (int i = 0; i < NB; i++) {
        acc += SQUARE ? v[i].x * v[i].x : v[i].x;
        acc += SQUARE ? v[i].y * v[i].y : v[i].y;
    }
    return acc;
}
// sum (SQUARE: of the squares) of the mp slots (mp even) of an LDS column, in slot order.  (Reading the next sixteen
// slots while the current sixteen are added — 32 more registers in a kernel that sits at 256 — measured slower: 7x7 8.9 ->
// 10.8 ms per 1e6 spots.)
template <bool SQUARE>
__device__ __forceinline__ double chain_sum(const double *col, int mp)
{
    const __attribute__((address_space(3))) d2_t *p = (const __attribute__((address_space(3))) d2_t *)col;
    double acc = 0;
    int b = 0;
#pragma unroll 1
    for (; b + 16 <= mp; b += 16) acc = chain_batch<8, SQUARE>(p + b / 2, acc);
    if (b + 8 <= mp) { acc = chain_batch<4, SQUARE>(p + b / 2, acc); b += 8; }
#pragma unroll 1
    for (; b < mp; b += 2) {
        const d2_t v = p[b / 2];
        acc += SQUARE ? v.x * v.x : v.x;
        acc += SQUARE ? v.y * v.y : v.y;
    }
    return acc;
}
constexpr int LQ_NSD = 64, LQ_NSI = 10;         // doubles / ints of state per spot
struct LqState {                                 // structure of arrays, stride = spots of the batch
    double *d;                                   // [0..5] x  [6..11] diag  12 fnorm  13 delta  14 par  15 xnorm
                                                 // [16..51] R (row-major 6x6, upper)  [52..57] qtf  [58..63] acnorm
    int32_t *i;                                  // [0..5] ipvt  6 iter  7 nfev  8 info (-1 fresh, 0 running, > 0 done)
                                                 // 9 tie: some decision of this fit was taken within rounding distance of its threshold
    int64_t stride, first;                       // state index of spot s = s - first
};
#define LQD(st, f, ls) (st).d[(int64_t)(f) * (st).stride + (ls)]
#define LQI(st, f, ls) (st).i[(int64_t)(f) * (st).stride + (ls)]

constexpr double LQ_PIVOT_TIE = 1e-9;         // relative distance of two running column norms below which the pivot choice is a tie
constexpr double LQ_RANK = 1e-3;             // |R_jj| / |column| below which the factor counts as rank deficient (re-fit)
constexpr double LQ_TIE = 1e-12;              // relative distance of a decision from its threshold below which a spot is re-fitted

// MINPACK qrsolv on the wave-uniform 6x6 factor R (R[(row) * 6 + (col)]).
__device__ __forceinline__ void qrsolv(double *R, const int (&ipvt)[6], const double (&diag)[6],
                                       const double (&qtb)[6], double (&x)[6], double (&sdiag)[6])
{
    double wa[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
#pragma unroll
        for (int i = j; i < 6; i++) R[(i) * 6 + (j)] = R[(j) * 6 + (i)];
        x[j] = R[(j) * 6 + (j)];
        wa[j] = qtb[j];
    }
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const double dl = get6(diag, ipvt[j]);
        if (dl != 0) {
#pragma unroll
            for (int k = j; k < 6; k++) sdiag[k] = 0;
            sdiag[j] = dl;
            double qtbpj = 0;
#pragma unroll
            for (int k = j; k < 6; k++) {
                if (sdiag[k] != 0) {
                    double c, sn;
                    if (fabs(R[(k) * 6 + (k)]) < fabs(sdiag[k])) {
                        const double cotan = R[(k) * 6 + (k)] / sdiag[k];
                        sn = 0.5 / sqrt(0.25 + 0.25 * (cotan * cotan));
                        c = sn * cotan;
                    } else {
                        const double tn = sdiag[k] / R[(k) * 6 + (k)];
                        c = 0.5 / sqrt(0.25 + 0.25 * (tn * tn));
                        sn = c * tn;
                    }
                    R[(k) * 6 + (k)] = c * R[(k) * 6 + (k)] + sn * sdiag[k];
                    double temp = c * wa[k] + sn * qtbpj;
                    qtbpj = -sn * wa[k] + c * qtbpj;
                    wa[k] = temp;
#pragma unroll
                    for (int i = k + 1; i < 6; i++) {
                        temp = c * R[(i) * 6 + (k)] + sn * sdiag[i];
                        sdiag[i] = -sn * R[(i) * 6 + (k)] + c * sdiag[i];
                        R[(i) * 6 + (k)] = temp;
                    }
                }
            }
        }
        sdiag[j] = R[(j) * 6 + (j)];
        R[(j) * 6 + (j)] = x[j];
    }
    int nsing = 6;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        if (sdiag[j] == 0 && nsing == 6) nsing = j;
        if (nsing < 6) wa[j] = 0;
    }
#pragma unroll
    for (int j = 5; j >= 0; j--) {
        if (j < nsing) {
            double sum = 0;
#pragma unroll
            for (int i = j + 1; i < 6; i++)
                if (i < nsing) sum += R[(i) * 6 + (j)] * wa[i];
            wa[j] = (wa[j] - sum) / sdiag[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 6; j++) set6(x, ipvt[j], wa[j]);
}

// MINPACK lmpar.
// tie: set when one of its tests (the Gauss-Newton step inside the region? the 10 % band around it reached?) is decided
// within LQ_TIE of its threshold
template <bool FLAG>
__device__ __forceinline__ void lmpar(double *R, const int (&ipvt)[6], const double (&diag)[6],
                                      const double (&qtb)[6], double delta, double &par, double (&x)[6],
                                      double (&sdiag)[6], unsigned &tie)
{
    double wa1[6], wa2[6];
    int nsing = 6;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        wa1[j] = qtb[j];
        if (R[(j) * 6 + (j)] == 0 && nsing == 6) nsing = j;
        if (nsing < 6) wa1[j] = 0;
    }
#pragma unroll
    for (int j = 5; j >= 0; j--) {
        if (j < nsing) {
            wa1[j] /= R[(j) * 6 + (j)];
            const double temp = wa1[j];
#pragma unroll
            for (int i = 0; i < j; i++) wa1[i] -= R[(i) * 6 + (j)] * temp;
        }
    }
#pragma unroll
    for (int j = 0; j < 6; j++) set6(x, ipvt[j], wa1[j]);
    int iter = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) wa2[j] = diag[j] * x[j];
    double dxnorm = enorm6(wa2);
    double fp = dxnorm - delta;
    if constexpr (FLAG) tie |= !(fabs(fp - 0.1 * delta) > LQ_TIE * (dxnorm + delta)) ? 2u : 0u;
    if (fp <= 0.1 * delta) { par = 0; return; }
    double parl = 0;
    if (nsing >= 6) {
#pragma unroll
        for (int j = 0; j < 6; j++) { const int l = ipvt[j]; wa1[j] = get6(diag, l) * (get6(wa2, l) / dxnorm); }
#pragma unroll
        for (int j = 0; j < 6; j++) {
            double sum = 0;
#pragma unroll
            for (int i = 0; i < j; i++) sum += R[(i) * 6 + (j)] * wa1[i];
            wa1[j] = (wa1[j] - sum) / R[(j) * 6 + (j)];
        }
        const double temp = enorm6(wa1);
        parl = ((fp / delta) / temp) / temp;
    }
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double sum = 0;
#pragma unroll
        for (int i = 0; i <= j; i++) sum += R[(i) * 6 + (j)] * qtb[i];
        wa1[j] = sum / get6(diag, ipvt[j]);
    }
    const double gnorm = enorm6(wa1);
    double paru = gnorm / delta;
    if (paru == 0) paru = DWARF / (delta < 0.1 ? delta : 0.1);
    if (par < parl) par = parl;
    if (par > paru) par = paru;
    if (par == 0) par = gnorm / dxnorm;
    for (;;) {
        iter++;
        if (par == 0) { const double t = 0.001 * paru; par = DWARF > t ? DWARF : t; }
        double temp = sqrt(par);
#pragma unroll
        for (int j = 0; j < 6; j++) wa1[j] = temp * diag[j];
        qrsolv(R, ipvt, wa1, qtb, x, sdiag);
#pragma unroll
        for (int j = 0; j < 6; j++) wa2[j] = diag[j] * x[j];
        dxnorm = enorm6(wa2);
        temp = fp;
        fp = dxnorm - delta;
        if constexpr (FLAG)
            tie |= (!(fabs(fabs(fp) - 0.1 * delta) > LQ_TIE * (dxnorm + delta))
                    || (parl == 0 && temp < 0 && !(fabs(fp - temp) > LQ_TIE * (dxnorm + delta)))) ? 2u : 0u;
        if (fabs(fp) <= 0.1 * delta || (parl == 0 && fp <= temp && temp < 0) || iter == 10) break;
#pragma unroll
        for (int j = 0; j < 6; j++) { const int l = ipvt[j]; wa1[j] = get6(diag, l) * (get6(wa2, l) / dxnorm); }
#pragma unroll
        for (int j = 0; j < 6; j++) {
            wa1[j] /= sdiag[j];
            temp = wa1[j];
#pragma unroll
            for (int i = j + 1; i < 6; i++) wa1[i] -= R[(i) * 6 + (j)] * temp;
        }
        temp = enorm6(wa1);
        const double parc = ((fp / delta) / temp) / temp;
        if (fp > 0 && parl < par) parl = par;
        if (fp < 0 && paru > par) paru = par;
        const double np_ = par + parc;
        par = parl > np_ ? parl : np_;
    }
}

// The one-spot-per-lane kernels read a spot many times (start values: three passes; every trial evaluation: one).
// Lane-strided reads of (N, box, box) — or of the movie — touch a cache line per lane and instruction, so the
// spots of a workgroup are first copied, converted to photons (localize.py:1101-1112), into an LDS tile with a
// coalesced sweep; boxes above 9x9 (tile too large) read from memory.
constexpr int LQ_TILE_MAXPIX = 81;
template <bool FROM_MOVIE, int NT, int M = 0>
__device__ __forceinline__ void stage_spots(const Params &p, const int64_t (&sidx)[NT], float *tile, int m_rt, int size, int64_t first = 0)
{
    // The sweep runs seven elements at a time — their spot indices, then their loads, then their LDS stores, each kind in
    // flight together; one element per iteration waits for an LDS read, then a load from memory, then a store (49 round trips
    // to memory in a row at 7x7: a fifth of lq_step_kernel's time).  M > 0: the box is known when the kernel is built
    const int hsz = size / 2;
    const int m = M > 0 ? M : m_rt;
    auto fetch = [&](int q) -> float {
        const int t = q / m, k = q - t * m;
        const int64_t s = sidx[t];
        float v = 0.f;
        if (s >= 0) {
            if (FROM_MOVIE) {
                const int i = k / size, j = k - i * size;
                const float raw = load_movie_px(p.movie, p.dtype, ((int64_t)p.frame[s] * p.Y + (p.y[s] - hsz + i)) * p.X + (p.x[s] - hsz + j));
                v = div_const((raw - p.baseline) * p.sensitivity, p.gdiv);
            } else {
                // (32-bit offsets from the batch's first spot: a batch is at most 2^21 spots of at most 441 pixels)
                v = (p.spots + first * m)[(unsigned)((unsigned)(s - first) * (unsigned)m + (unsigned)k)];
            }
        }
        return v;
    };
    // seven loads in flight per round trip (with all of a 7x7 sweep laid out the addresses alone cost the step kernel its
    // registers: 0.70 -> 0.92 ms; seven at a time: 0.70 -> 0.58)
#pragma unroll 1
    for (int it0 = 0; it0 < m; it0 += 7) {
        float v[7];
#pragma unroll
        for (int u = 0; u < 7; u++) v[u] = (M > 0 && M % 7 == 0) || it0 + u < m ? fetch((int)threadIdx.x + NT * (it0 + u)) : 0.f;
#pragma unroll
        for (int u = 0; u < 7; u++)
            if ((M > 0 && M % 7 == 0) || it0 + u < m) tile[(int)threadIdx.x + NT * (it0 + u)] = v[u];
    }
}

#ifndef LQ_STEP_UNROLL_ROWS
#define LQ_STEP_UNROLL_ROWS 5      // boxes up to this size unroll the rows of the trial evaluation too
#endif
constexpr int LQ_STEP_NT = 64;        // one wave per workgroup: 7.58 -> 7.46 ms per 1e6 7x7 spots against 128 (256: 7.8), 13x13 -6 %; the tile of a wave's spots is its own
// One spot per lane, float64 throughout: chains of dependent divisions and square roots, i.e. latency, and left alone
// the kernel takes 256 VGPRs + 54 AGPRs = one wave per SIMD.  Asked to leave room for two, the compiler spills 59
// values (216 B of scratch per lane) and the kernel is still the faster for it: 7x7 8.40 -> 7.60 ms per 1e6 spots, 5x5
// and 13x13 -5 % / -4 % (alternating runs on one box).
#ifndef LQ_STEP_MIN_WAVES
#define LQ_STEP_MIN_WAVES 2
#endif

// (b): one spot per lane — the Levenberg-Marquardt step(s) on the factor lq_jacobian_kernel left, until the fit ends
// or needs a new Jacobian.  Spots that go on are appended to next_list.
// (b) for ONE spot per lane: the Levenberg-Marquardt step(s) on the factor lq_jacobian_spot left, until the fit ends (theta,
// info, nfev written; returns info > 0) or needs a new Jacobian (state written; returns 0).  `tie`: the decision flags.
// FLAG: every decision is also tested against a band around its threshold (the spots of the first pass of the refit mode)
// FRAG: a profile value whose float32 rounding hangs on the last bits of exp sets bit 7 (first pass of the strict mode);
// CR: exp rounded correctly (its second pass)
// BOX: the box size when the kernel is built for one (the pixel loops of the trial evaluation unroll and their LDS reads
// batch), 0: read from p.box
template <bool FROM_MOVIE, bool FLAG, bool FRAG, bool CR, int BOX = 0>
__device__ __forceinline__ int lq_step_spot(const Params &p, const LqState &st, int64_t s, int info, bool flagging, const float *mytile,
                                            float (*s_px)[LQ_STEP_NT], int tid, bool staged, unsigned &tie)
{
    const int size = BOX ? BOX : p.box, m = size * size, hsz = size / 2;
    const int64_t ls = s - st.first;
    const double ftol = 1e-2, xtol = 1e-2, gtol = 0.0, factor = 100.0;
    const int maxfev = 200 * (6 + 1);
    int64_t fr = 0, y0 = 0, x0 = 0;
    if (FROM_MOVIE) { fr = p.frame[s]; y0 = p.y[s] - hsz; x0 = p.x[s] - hsz; }

    // norm of the residuals of the float32-stored model (gausslq.py:151-203) at th
    unsigned fragile = 0u;
    float pxl[BOX > 0 ? BOX : 1];           // a kernel built for its box keeps the x profile of an evaluation in registers
    auto fnorm_at = [&](const double (&th)[6]) -> double {
        const double nx = 0.3989422804014327 / th[4], ny = 0.3989422804014327 / th[5];
#pragma clang loop unroll_count(BOX > 0 ? BOX : 1)
        for (int j = 0; j < size; j++) {
            const double t = ((double)(float)(j - hsz) - th[0]) / th[4];
            const double pv = nx * lq_exp<CR>(-0.5 * (t * t));
            if constexpr (FRAG) fragile |= fragile_f32_rounding(pv);
            if constexpr (BOX > 0) pxl[j] = (float)pv; else s_px[j][tid] = (float)pv;
        }
        EnormAcc acc(m);
#pragma clang loop unroll_count(BOX > 0 && BOX <= LQ_STEP_UNROLL_ROWS ? BOX : 1)
        for (int i = 0; i < size; i++) {
            const double t = ((double)(float)(i - hsz) - th[1]) / th[5];
            const double pvy = ny * lq_exp<CR>(-0.5 * (t * t));
            if constexpr (FRAG) fragile |= fragile_f32_rounding(pvy);
            const float myv = (float)pvy;
#pragma clang loop unroll_count(BOX > 0 ? BOX : 1)
            for (int j = 0; j < size; j++) {
                float spv;
                if (staged) {
                    spv = mytile[i * size + j];
                } else if (FROM_MOVIE) {
                    const float raw = load_movie_px(p.movie, p.dtype, (fr * p.Y + (y0 + i)) * p.X + (x0 + j));
                    spv = div_const((raw - p.baseline) * p.sensitivity, p.gdiv);
                } else {
                    spv = p.spots[s * m + i * size + j];
                }
                float mxv;
                if constexpr (BOX > 0) mxv = pxl[j]; else mxv = s_px[j][tid];
                const float model = (float)(th[2] * (double)myv * (double)mxv + th[3]);
                const float res = spv - model;
                acc.add((double)res);
            }
        }
        return acc.norm();
    };

    double x[6], diag[6], qtf[6], wa1[6], wa2[6], wa3[6], R[36];
    int ipvt[6];
#pragma unroll
    for (int j = 0; j < 6; j++) x[j] = LQD(st, j, ls);
    int nfev, iter;
    double par, delta, xnorm, fnorm;
    if (info < 0) {
        fnorm = fnorm_at(x);
        nfev = 1; iter = 1; par = 0; delta = 0; xnorm = 0;
#pragma unroll
        for (int j = 0; j < 6; j++) diag[j] = 0;
        info = 0;
    } else {
#pragma unroll
        for (int j = 0; j < 6; j++) diag[j] = LQD(st, 6 + j, ls);
        fnorm = LQD(st, 12, ls); delta = LQD(st, 13, ls); par = LQD(st, 14, ls); xnorm = LQD(st, 15, ls);
        iter = LQI(st, 6, ls); nfev = LQI(st, 7, ls);
    }
#pragma unroll
    for (int k = 0; k < 36; k++) R[k] = (k / 6 <= k % 6) ? LQD(st, 16 + k, ls) : 0.0;       // (below the diagonal: qrsolv's workspace, never read before it is written — not stored)
#pragma unroll
    for (int j = 0; j < 6; j++) { qtf[j] = LQD(st, 52 + j, ls); wa2[j] = LQD(st, 58 + j, ls); ipvt[j] = LQI(st, j, ls); }
    nfev += 6;                                                 // the forward differences of this round
    // tie: a decision of this fit fell within LQ_TIE of its threshold — here, in lmpar or in the pivoting of the Jacobian
    // kernel (slot 9).  The group kernel's tree sums differ from MINPACK's sequential ones in the last bits of float64,
    // so such a decision may be MINPACK's other branch: the spot is fitted again with sequential sums (tie_list).
    tie = (FLAG || FRAG) && flagging ? (unsigned)LQI(st, 9, ls) : 0u;       // bit 0: the pivot choice of the Jacobian kernel; 1: lmpar; 2..6 below
    if (!FRAG) tie &= ~128u;                                  // bit 7 (a float32 rounding that hangs on an exp, set by every strict Jacobian) is the strict first pass's alone
    double gnorm = 0, fnorm1, actred, prered, dirder, ratio, pnorm;
    if (iter == 1) {
#pragma unroll
        for (int j = 0; j < 6; j++) { diag[j] = wa2[j]; if (wa2[j] == 0) diag[j] = 1; }
#pragma unroll
        for (int j = 0; j < 6; j++) wa3[j] = diag[j] * x[j];
        xnorm = enorm6(wa3);
        delta = factor * xnorm;
        if (delta == 0) delta = factor;
    }
    if (fnorm != 0) {
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const double w2l = get6(wa2, ipvt[j]);
            if (w2l != 0) {
                double sum = 0;
#pragma unroll
                for (int i = 0; i <= j; i++) sum += R[(i) * 6 + (j)] * (qtf[i] / fnorm);
                const double g = fabs(sum / w2l);
                if (g > gnorm) gnorm = g;
            }
        }
    }
    if (gnorm <= gtol) info = 4;
    if (info == 0) {
#pragma unroll
        for (int j = 0; j < 6; j++)
            if (wa2[j] > diag[j]) diag[j] = wa2[j];
        for (;;) {
            lmpar<FLAG>(R, ipvt, diag, qtf, delta, par, wa1, wa2, tie);
#pragma unroll
            for (int j = 0; j < 6; j++) { wa1[j] = -wa1[j]; wa2[j] = x[j] + wa1[j]; wa3[j] = diag[j] * wa1[j]; }
            pnorm = enorm6(wa3);
            if (iter == 1 && pnorm < delta) delta = pnorm;
            fnorm1 = fnorm_at(wa2);
            nfev++;
            actred = -1;
            if (0.1 * fnorm1 < fnorm) { const double r = fnorm1 / fnorm; actred = 1 - r * r; }
#pragma unroll
            for (int j = 0; j < 6; j++) wa3[j] = 0;
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const double temp = get6(wa1, ipvt[j]);
#pragma unroll
                for (int i = 0; i <= j; i++) wa3[i] += R[(i) * 6 + (j)] * temp;
            }
            const double temp1 = enorm6(wa3) / fnorm, temp2 = (sqrt(par) * pnorm) / fnorm;
            prered = temp1 * temp1 + temp2 * temp2 / 0.5;
            dirder = -(temp1 * temp1 + temp2 * temp2);
            ratio = 0;
            if (prered != 0) ratio = actred / prered;
            if constexpr (FLAG) {
                // actred = 1 - (fnorm1 / fnorm)^2 carries an absolute error of a few eps (1 + r^2); ratio divides it by prered
                const double r1 = fnorm1 / fnorm;
                const double ea = LQ_TIE * (1.0 + r1 * r1);
                const double er = prered != 0 ? ea / prered + LQ_TIE * fabs(ratio) : 0.0;
                tie |= (!(fabs(0.1 * fnorm1 - fnorm) > LQ_TIE * fnorm) ? 4u : 0u)
                       | ((!(fabs(ratio - 0.25) > er) || (par != 0 && !(fabs(ratio - 0.75) > er)) || !(fabs(ratio - 1e-4) > er)
                           || !(fabs(ratio - 2.0) > er)) ? 8u : 0u)
                       | ((!(fabs(fabs(actred) - ftol) > ea) || !(fabs(prered - ftol) > LQ_TIE * prered)) ? 16u : 0u)
                       | ((fabs(actred) <= 64 * ea && prered <= 64 * ea) ? 32u : 0u);        // reduction at the noise level: the EPSMCH tests
            }
            if (ratio <= 0.25) {
                double temp = 0.5;
                if (actred < 0) temp = 0.5 * dirder / (dirder + 0.5 * actred);
                if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
                const double pd = pnorm / 0.1;
                delta = temp * (delta < pd ? delta : pd);
                par = par / temp;
            } else if (par == 0 || ratio >= 0.75) {
                delta = pnorm / 0.5;
                par = 0.5 * par;
            }
            if (ratio >= 1e-4) {
#pragma unroll
                for (int j = 0; j < 6; j++) { x[j] = wa2[j]; wa2[j] = diag[j] * x[j]; }
                xnorm = enorm6(wa2);
                fnorm = fnorm1;
                iter++;
            }
            if constexpr (FLAG) tie |= !(fabs(delta - xtol * xnorm) > LQ_TIE * delta) ? 64u : 0u;
            if (fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1) info = 1;
            if (delta <= xtol * xnorm) info = 2;
            if (fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1 && info == 2) info = 3;
            if (info != 0) break;
            if (nfev >= maxfev) info = 5;
            if (fabs(actred) <= EPSMCH && prered <= EPSMCH && 0.5 * ratio <= 1) info = 6;
            if (delta <= EPSMCH * xnorm) info = 7;
            if (gnorm <= EPSMCH) info = 8;
            if (info != 0) break;
            if (ratio >= 1e-4) break;
        }
    }
    if (FRAG && fragile) tie |= 128u;                       // bit 7: a float32 rounding of the model hangs on the last bit of an exp
    if (info != 0) {
#pragma unroll
        for (int j = 0; j < 6; j++) p.thetas[s * 6 + j] = (float)x[j];
        if (p.info) p.info[s] = info;
        if (p.nfev) p.nfev[s] = nfev;
        LQI(st, 8, ls) = info;
    } else {
#pragma unroll
        for (int j = 0; j < 6; j++) { LQD(st, j, ls) = x[j]; LQD(st, 6 + j, ls) = diag[j]; }
        LQD(st, 12, ls) = fnorm; LQD(st, 13, ls) = delta; LQD(st, 14, ls) = par; LQD(st, 15, ls) = xnorm;
        LQI(st, 6, ls) = iter; LQI(st, 7, ls) = nfev; LQI(st, 8, ls) = 0;
        if ((FLAG || FRAG) && tie && flagging) LQI(st, 9, ls) = (int32_t)tie;
    }
    return info;
}

}  // namespace lq
}  // namespace pmi
