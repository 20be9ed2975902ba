// fit_common.h — shared by the MLE fit kernels (gaussmle.hip, gaussmle_g8.hip).
#pragma once
#include <math.h>

#include "pmi_common.h"

namespace pmi {

#ifndef FIT_WAVES_N
#define FIT_WAVES_N 4
#endif
constexpr int FIT_WAVES = FIT_WAVES_N;       // waves per workgroup
constexpr int FIT_NT = FIT_WAVES * PMI_WAVE;
constexpr int FIT_MAXPIX = PMI_MAX_BOX * PMI_MAX_BOX;

// x / g for a wave-uniform divisor g, correctly rounded, in 3 instructions instead of the ~10 of an
// IEEE float division (seven of them per row of every spot, in every kernel that converts camera
// counts to photons: picasso/localize.py:1112 divides by Gain).  Markstein: with r = RN(1/g),
// q = RN(x r), e = x - q g (exact in an FMA), RN(q + e r) is the correctly rounded quotient, barring
// overflow / underflow — values outside a safe range take the real division.
struct ConstDiv { float g, r; int fast; };      // fast: 0 = always the real division, 1 = Markstein, 2 = g is 1 (x / 1 = x)
inline ConstDiv make_const_div(float g)
{
    ConstDiv c;
    c.g = g;
    c.r = (float)(1.0 / (double)g);                  // correctly rounded: double has more than 2*24+2 bits
    c.fast = (g == g) && fabsf(g) > 1e-15f && fabsf(g) < 1e15f;
    if (g == 1.0f) c.fast = 2;
    return c;
}
__device__ __forceinline__ float div_const(float x, const ConstDiv &c)
{
    if (c.fast == 2) return x;
    const float ax = fabsf(x);
    if (!c.fast || !((ax > 1e-15f && ax < 1e15f) || x == 0.0f)) return x / c.g;
    const float q = x * c.r;
    const float e = __builtin_fmaf(-q, c.g, x);
    return __builtin_fmaf(e, c.r, q);
}
// The same for the B values a lane holds, with ONE branch: the range test is accumulated over the row (largest
// magnitude, smallest non-zero magnitude, on the bit patterns) and the real divisions run for the whole row when any
// value fails it.  A zero is safe in the short form (q = e = 0 with the sign of x / g); a NaN gives NaN either way.
template <int B> __device__ __forceinline__ void div_const_row(float (&v)[B], const ConstDiv &c)
{
    if (c.fast == 2) return;
    unsigned hi = 0u, lo = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < B; i++) {
        const unsigned u = __float_as_uint(v[i]) & 0x7fffffffu;
        hi = u > hi ? u : hi;
        const unsigned w = u - 1u;                      // zero wraps to the top: not "small"
        lo = w < lo ? w : lo;
    }
    // 1e15f = 0x58635fa9, 1e-15f = 0x26901d7d
    if (c.fast && hi < 0x58635fa9u && lo >= 0x26901d7du) {
#pragma unroll
        for (int i = 0; i < B; i++) {
            const float q = v[i] * c.r;
            const float e = __builtin_fmaf(-q, c.g, v[i]);
            v[i] = __builtin_fmaf(e, c.r, q);
        }
    } else {
#pragma unroll
        for (int i = 0; i < B; i++) v[i] = v[i] / c.g;
    }
}

struct FitParams {
    // source: spots (float32) or movie + identifications
    const float *spots;
    float *spots_out;      // g8_init from a movie: the photon values of every spot of the batch are kept here (batch-relative)
    const void *movie;
    const int32_t *frame, *y, *x;
    // pixel hand-off (uint16 movies): where the scan's exact stage left the box rows of spot i — pix[slot[i]], box rows of
    // box / 2 + 1 packed uint16 pairs — or slot[i] < 0 / slot == nullptr: read the movie
    const uint32_t *pix;
    const int32_t *slot;
    int dtype;
    int64_t Y, X;
    float baseline, sensitivity, gain;
    ConstDiv gdiv;         // division by gain (make_const_div)
    // common
    int64_t N;             // capacity / number of rows
    const int64_t *d_n;    // optional device row count
    int box;
    int libm_glibc;        // mle_strict_kernel: erf / exp operation for operation as glibc's (libm_glibc.h) instead of the device library's
    double eps;
    int max_it;
    float *thetas, *crlbs, *loglik;
    int32_t *iterations;
    unsigned long long *queue;   // dynamic spot queue of this batch (counts from 0)
    unsigned *strict_queue;      // mle_strict_kernel over a list: the next entry to hand out (nullptr: entries are dealt round robin)
    int64_t first;               // first spot of this batch; the kernel handles [first, min(N, *d_n))
    double *fisher;              // upper triangle of the Fisher matrix, 21 doubles per spot of the batch
    // borderline-convergence flags (gaussmle_strict.hip): a spot whose largest tested step |delta| came within
    // [eps_lo, eps_hi) of eps in any iteration, or that ran into max_it, is appended to flag_list
    float eps_lo, eps_hi;
    int32_t *flag_list;          // spot indices, capacity = spots of the batch (nullptr: no flagging)
    unsigned *flag_count;
    unsigned *flag_reasons;      // FLAG_REASONS counters of the call: how many spots each criterion flagged (nullptr: not counted)
    // second pass (spots whose iteration turned out not to contract at the fitted theta, crlb_kernel): the Fisher pass over
    // the spots of this list only (nullptr: every spot of the batch)
    const int32_t *final_list;
    const unsigned *final_list_n;
    unsigned char *refit_mark;   // one byte per spot of the batch: set by the strict re-fit, so that the second list only takes spots the first did not
    // Deferred exact stage of identify (a fused call whose scan emitted CANDIDATES, pmi_common.h NG_DEFERRED_BITS): the
    // start-value kernel (g8_init) computes the float32 net gradient in the reference's (k, l) order and the first-argmax
    // test from the rows it reads for the fit anyway (picasso/localize.py:97-134, 202-244, 288), writes ng_io[i] and
    // accept[i]; the later stages skip the rejected, the Newton loop runs over the accepted (alist), the table compacts.
    float *ng_io;                // in: a net gradient, or NG_DEFERRED_BITS; out: the net gradient (nullptr: every row is an identification)
    unsigned char *accept;       // one byte per spot (absolute index)
    int crop_y0, crop_x0, crop_cy, crop_cx;      // the crop identify scanned: a stencil's row / column -1 wraps to the crop's last one
    double min_ng;
    const int32_t *alist;        // accepted spots of the batch (absolute indices), ascending; nullptr: all spots of the batch
    const unsigned *alist_n;
    // where the caller wants that list (N entries + the block counters of build_accept_list): a call of ONE batch builds it
    // there, and the caller's table needs no second pass over the flags (host-side fields: the kernels do not read them)
    int32_t *alist_out;
    unsigned *alist_blk_out;
};
// why a spot goes to the re-fit (a spot can carry several)
enum : unsigned { FLAG_MARGIN = 1u, FLAG_CURVATURE = 2u, FLAG_NARROW = 4u, FLAG_SWING = 8u, FLAG_WILD = 16u, FLAG_SLOW = 32u, FLAG_UNSTABLE = 64u };
constexpr int FLAG_REASONS = 7;
// The update treats every parameter on its own, theta_l -= num_l / den_l: near the fitted theta its Jacobian is
// I - D^-1 H with H the Hessian of the log-likelihood and D its diagonal, in expectation I - D^-1 M with the Fisher matrix
// M the final pass computes anyway.  The eigenvalues of D^-1 M are those of the symmetric C = D^-1/2 M D^-1/2 (unit
// diagonal: how strongly the parameters trade against each other); the iteration contracts iff lambda_max(C) < 2, and a
// rounding difference is multiplied by 1 - lambda_max per iteration.  Above FIT_UNSTABLE_LMAX the spot is re-fitted.
// (Simulated DNA-PAINT spots: lambda_max 1.2 ... 1.7; the fuzz residual of round 3 that no step-sequence rule caught:
// 2.356, its differences alternating with a factor of -1.35.)
constexpr double FIT_UNSTABLE_LMAX = 1.9;
constexpr int FISHER_STRIDE = 21;
enum { FIT_STAGE_NEWTON = 1, FIT_STAGE_FINAL = 2, FIT_STAGE_INIT_ONLY = 4, FIT_STAGE_ITERATE_ONLY = 8 };
// a fit that takes more iterations than this is re-fitted whatever its steps were.  Round 2 set it to 32 as a catch-all
// for fits that creep or wobble; with the wobble and contraction flags of round 3 (below) the count adds nothing — the
// CPU emulation (docs/history/tools/emul/slow_rule.py: 48 cases x 30 000 spots) and the fuzz run find the same escapes, none, at 32, 48,
// 64 and without it — and at 32 it re-fits healthy fits wherever they are naturally long (eps 1e-4: 6 % of config 2's
// spots, 5x5 boxes: 30 %).  64 keeps the fits that run into the default max_it among the re-fitted.
constexpr int FIT_SLOW_ITERATIONS = 64;
// the margin around eps inside which a tested step flags the spot grows by this per iteration beyond the sixteenth (see the
// fast kernels; -DPMI_MARGIN_GROWTH=... builds a library for A/B runs)
#ifndef PMI_MARGIN_GROWTH
#define PMI_MARGIN_GROWTH 0.0625f
#endif
constexpr float FIT_MARGIN_GROWTH = PMI_MARGIN_GROWTH;
constexpr float FIT_NARROW_SIGMA = 0.5f;      // a fitted width below this (px) sends the spot to the re-fit (0.3 until round 6: 5x5 and
                                              // 7x7 fits of 0.31 ... 0.495 px at eps 1e-4 ended an iteration off the reference, no other flag raised)
// the alternating component of a parameter's step sequence (second difference) that changes sign without shrinking below
// FIT_WOBBLE_RATIO of its previous size, above the rounding floor (FIT_WOBBLE_FLOOR x |value| = 16 float32 ulps), for
// FIT_WOBBLE_RUN iterations in a row — two in a row from iteration FIT_WOBBLE_LATE on, where a healthy fit's step clamps
// have long released: the iteration is not contracting, re-fit (see newton_step, gaussmle_g8.hip)
constexpr float FIT_WOBBLE_RATIO = 0.9f;
constexpr float FIT_WOBBLE_FLOOR = 1.9073486e-6f;
constexpr int FIT_WOBBLE_RUN = 3;
constexpr int FIT_WOBBLE_LATE = 9;
// max over the pixels of |data / model - 1| and |data / model^2| beyond which the float32 sums are not trusted: re-fit
constexpr float FIT_TOP_FLAG = 16.0f;

// ---- DPP wave reductions -------------------------------------------------
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_f(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
// sum over the 64 lanes, result uniform (broadcast from lane 63)
__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_f<0xB1>(v);          // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);          // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);         // row_half_mirror
    v += dpp_f<0x140>(v);         // row_mirror
    v += dpp_f<0x142, 0xA>(v);    // row_bcast15 -> rows 1,3
    v += dpp_f<0x143, 0xC>(v);    // row_bcast31 -> rows 2,3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_d(double v)
{
    long long b = __builtin_bit_cast(long long, v);
    int lo = (int)(b & 0xffffffffLL), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_sum_d(double v)
{
    v += dpp_d<0xB1>(v);
    v += dpp_d<0x4E>(v);
    v += dpp_d<0x141>(v);
    v += dpp_d<0x140>(v);
    v += dpp_d<0x142, 0xA>(v);
    v += dpp_d<0x143, 0xC>(v);
    long long b = __builtin_bit_cast(long long, v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), 63);
    int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ float wave_min(float v)
{
    for (int off = 32; off >= 1; off >>= 1) v = fminf(v, __shfl_xor(v, off));
    return v;
}

// numpy-style NaN-propagating max/min and sign
__device__ __forceinline__ float np_maxf(float a, float b) { return (a != a) ? a : ((b != b) ? b : (a > b ? a : b)); }
__device__ __forceinline__ float np_minf(float a, float b) { return (a != a) ? a : ((b != b) ? b : (a < b ? a : b)); }
__device__ __forceinline__ float np_signf(float a) { return (a != a) ? a : (a > 0.f ? 1.f : (a < 0.f ? -1.f : 0.f)); }

// One-dimensional pixel-integrated Gaussian terms for pixel index `i`:
//   E   = deltaE                      (gaussmle.py:268-280)
//   A   = dE/dmu,   A2 = d2E/dmu2     (gaussmle.py:283-303, without photons*PSF_other)
//   S   = dE/dsig,  S2 = d2E/dsig2    (gaussmle.py:306-336)
struct Terms { float E, A, A2, S, S2; };
__device__ __forceinline__ Terms gauss_terms(float i, float mu, float sigma)
{
    const float is = 1.0f / sigma;
    const float sn = 0.70710678118654757f * is;
    const float c1 = 0.3989422804014327f * is;      // 1/(sqrt(2 pi) sigma)
    const float is2 = is * is;
    const float dm = i - mu - 0.5f, dp = i - mu + 0.5f;
    Terms t;
    t.E = 0.5f * (erff(dp * sn) - erff(dm * sn));
    const float gm = __expf(-0.5f * dm * dm * is2), gp = __expf(-0.5f * dp * dp * is2);
    const float q1 = dm * gm - dp * gp;
    const float q3 = dm * dm * dm * gm - dp * dp * dp * gp;
    t.A = (gm - gp) * c1;
    t.A2 = q1 * c1 * is2;
    t.S = q1 * c1 * is;
    t.S2 = c1 * is2 * (q3 * is2 - 2.0f * q1);
    return t;
}

__device__ __forceinline__ void count_flag_reasons(unsigned *reasons, unsigned bits)
{
    if (!reasons) return;
    for (int b = 0; b < FLAG_REASONS; b++)
        if (bits & (1u << b)) atomicAdd(reasons + b, 1u);
}

__device__ __forceinline__ float load_movie_px(const void *movie, int dtype, int64_t idx)
{
    switch (dtype) {
    case PMI_U16: return (float)((const uint16_t *)movie)[idx];
    case PMI_U8:  return (float)((const uint8_t *)movie)[idx];
    case PMI_I16: return (float)((const int16_t *)movie)[idx];
    case PMI_U32: return (float)((const uint32_t *)movie)[idx];
    case PMI_I32: return (float)((const int32_t *)movie)[idx];
    default:      return ((const float *)movie)[idx];
    }
}


}  // namespace pmi
