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

}  // namespace lq
}  // namespace pmi
