// gaussmle_g8.hip — MLE fit for boxes up to 15x15, one ROW of the spot per lane: eight lanes
// per spot and eight spots per wavefront for boxes up to 7x7, a 16-lane DPP row per spot and
// four spots per wavefront for boxes 9..15 (GroupOf<B>::GS).  The text below says "eight" for
// the group size throughout.
//
// Same algorithm and quirks as mle_fit_kernel (gaussmle.hip; picasso/gaussmle.py
// :28-168, :268-383, :533-954); only the mapping to the machine differs:
//
//   * lane (g, j): group g = lane / 8 fits one spot, lane j holds row j (B pixels in
//     registers, j < B).  All eight lanes of a group are busy in the transcendental
//     stage: lane j evaluates the pixel BOUNDARY j (erf and exp at j - 1/2 - mu) for
//     x and for y — B+1 <= 8 boundaries serve all B pixels, a 7x cut over evaluating
//     four erf per pixel;
//   * boundary k+1 comes from lane j+1 by DPP row_shl:1; the per-column x terms of all
//     B columns are exchanged inside the group through LDS;
//   * the model is separable, so each lane accumulates ten row-local sums over its B
//     pixels (packed float32: two sums per v_pk_fma_f32) and multiplies by its row
//     constants once; the 12 group sums go through LDS so that lane l receives the
//     numerator and denominator of parameter l and updates that one parameter;
//   * three kernels: g8_init (initial theta and max_step per spot), g8_iterate (persistent
//     waves; a group that converges stores its theta and immediately REFILLS with the next
//     spot of its wave's chunk, so the eight groups never wait for the slowest spot — the
//     iteration count per spot ranges 3..100), g8_final (Fisher matrix, log-likelihood);
//     the 6x6 inverse runs one thread per spot in crlb_kernel (gaussmle.hip).
//
// Float32 Newton loop, float64 initial sums and Fisher matrix, like the wave-per-spot kernel.
#include <algorithm>

#include "fit_common.h"
#include "ng_common.h"

namespace pmi {

// 8-lane group reductions: every lane of the group ends with the result
__device__ __forceinline__ float sum8(float v)
{
    v += dpp_f<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);     // row_half_mirror
    return v;
}
__device__ __forceinline__ float min8(float v)
{
    v = fminf(v, dpp_f<0xB1>(v));
    v = fminf(v, dpp_f<0x4E>(v));
    v = fminf(v, dpp_f<0x141>(v));
    return v;
}
__device__ __forceinline__ double sum8_d(double v)
{
    v += dpp_d<0xB1>(v);
    v += dpp_d<0x4E>(v);
    v += dpp_d<0x141>(v);
    return v;
}
// group size: eight lanes per spot for boxes up to 7, a full 16-lane DPP row for boxes 9..15
template <int B> struct GroupOf { static constexpr int GS = B <= 7 ? 8 : 16; };
template <int GS> __device__ __forceinline__ float gsum(float v)
{
    v = sum8(v);
    if (GS == 16) v += dpp_f<0x140>(v);      // row_mirror
    return v;
}
template <int GS> __device__ __forceinline__ float gmin(float v)
{
    v = min8(v);
    if (GS == 16) v = fminf(v, dpp_f<0x140>(v));
    return v;
}
template <int GS> __device__ __forceinline__ double gsum_d(double v)
{
    v = sum8_d(v);
    if (GS == 16) v += dpp_d<0x140>(v);
    return v;
}
// group maximum of unsigned values (bit patterns of non-negative floats order like the floats, NaN on top)
template <int CTRL> __device__ __forceinline__ unsigned dpp_u(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
template <int GS> __device__ __forceinline__ unsigned gor_u(unsigned v)
{
    v |= dpp_u<0xB1>(v);
    v |= dpp_u<0x4E>(v);
    v |= dpp_u<0x141>(v);
    if (GS == 16) v |= dpp_u<0x140>(v);
    return v;
}
template <int GS> __device__ __forceinline__ unsigned gmax_u(unsigned v)
{
    v = max(v, dpp_u<0xB1>(v));
    v = max(v, dpp_u<0x4E>(v));
    v = max(v, dpp_u<0x141>(v));
    if (GS == 16) v = max(v, dpp_u<0x140>(v));
    return v;
}
// value of lane+1 / lane-1 (0 past the 16-lane DPP row; callers mask the group edges)
__device__ __forceinline__ float from_next(float v) { return dpp_f<0x101>(v); }   // row_shl:1
__device__ __forceinline__ double from_next_d(double v) { return dpp_d<0x101>(v); }
__device__ __forceinline__ double from_prev_d(double v) { return dpp_d<0x111>(v); } // row_shr:1

// erf(x), < 1 ulp, branch-free (two polynomial branches + select); one v_exp_f32
__device__ __forceinline__ float erf_f32(float a)
{
    const float t = fabsf(a), s = a * a;
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    const float big = copysignf(1.0f - __expf(r), a);
    float q = -5.96761703e-4f;
    q = fmaf(q, s, 4.99119423e-3f);
    q = fmaf(q, s, -2.67681349e-2f);
    q = fmaf(q, s, 1.12819925e-1f);
    q = fmaf(q, s, -3.76125336e-1f);
    q = fmaf(q, s, 1.28379166e-1f);
    const float small = fmaf(q, a, a);
    return t > 0.921875f ? big : small;
}

__device__ __forceinline__ float rcp_f32(float x) { return __builtin_amdgcn_rcpf(x); }       // v_rcp_f32, 1 ulp
// numpy clip / maximum / minimum semantics (NaN propagates) at 3 instructions instead of 8
// np.minimum(np.maximum(q, -lim), lim): NOT a median — max_step can be negative (background below zero
// after baseline subtraction), and then the reference's nested min/max always returns lim
__device__ __forceinline__ float clip_np(float q, float lim) { const float c = fminf(fmaxf(q, -lim), lim); return (q != q) ? q : c; }
__device__ __forceinline__ float max_np(float a, float b) { const float c = fmaxf(a, b); return (a != a) ? a : c; }   // b is a finite constant
__device__ __forceinline__ float min_np(float a, float b) { const float c = fminf(a, b); return (a != a) ? a : c; }

struct BTerms { float E, A, A2, S, S2; };
// per-pixel terms of index j from the boundary values of lane j (k = j) and lane j+1
__device__ __forceinline__ BTerms boundary_terms(float jf, float mu, float sigma)
{
    const float is = rcp_f32(sigma);
    const float sn = 0.70710678118654757f * is, c1 = 0.3989422804014327f * is, is2 = is * is;
    const float u0 = jf - 0.5f - mu;
    const float e0 = erf_f32(u0 * sn);
    const float g0 = __expf(-0.5f * u0 * u0 * is2);
    const float e1 = from_next(e0), g1 = from_next(g0);
    const float u1 = u0 + 1.0f;
    const float q1 = u0 * g0 - u1 * g1;
    const float q3 = u0 * u0 * u0 * g0 - u1 * u1 * u1 * g1;
    BTerms t;
    t.E = 0.5f * (e1 - e0);
    t.A = (g0 - g1) * c1;
    t.A2 = q1 * c1 * is2;
    t.S = q1 * c1 * is;
    t.S2 = c1 * is2 * (q3 * is2 - 2.0f * q1);
    return t;
}

// The same for BOTH axes of a lane at once: component .x = column j (mu = theta_0, sigma = theta_4), .y = row j (theta_1,
// theta_5 or theta_4).  The two evaluations are independent and made of the same operations, so every multiply, add and
// FMA of boundary_terms / erf_f32 becomes one packed instruction on a register pair (v_pk_mul_f32, v_pk_add_f32,
// v_pk_fma_f32) — component by component the same operation order as the scalar form.  Only v_rcp_f32, v_exp_f32, the
// DPP moves, the sign transfer and the final select stay per component.
#ifndef G8_PACKED_BOUNDARY
#define G8_PACKED_BOUNDARY 1      // A/B knob of the build: 0 = the scalar evaluation per axis
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct BTerms2 { f32x2 E, A, A2, S, S2; };
__device__ __forceinline__ f32x2 splat2(float v) { return (f32x2){v, v}; }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 erf_f32x2(f32x2 a)
{
    const f32x2 t = {fabsf(a.x), fabsf(a.y)}, s = a * a;
    f32x2 r = fma2(splat2(-1.72853470e-5f), t, splat2(3.83197126e-4f));
    const f32x2 u = fma2(splat2(-3.88396438e-3f), t, splat2(2.42546219e-2f));
    r = fma2(r, s, u);
    r = fma2(r, t, splat2(-1.06777877e-1f));
    r = fma2(r, t, splat2(-6.34846687e-1f));
    r = fma2(r, t, splat2(-1.28717512e-1f));
    r = fma2(r, t, -t);
    const f32x2 ex = {__expf(r.x), __expf(r.y)};
    const f32x2 one_m = splat2(1.0f) - ex;
    const f32x2 big = {copysignf(one_m.x, a.x), copysignf(one_m.y, a.y)};
    f32x2 q = splat2(-5.96761703e-4f);
    q = fma2(q, s, splat2(4.99119423e-3f));
    q = fma2(q, s, splat2(-2.67681349e-2f));
    q = fma2(q, s, splat2(1.12819925e-1f));
    q = fma2(q, s, splat2(-3.76125336e-1f));
    q = fma2(q, s, splat2(1.28379166e-1f));
    const f32x2 small = fma2(q, a, a);
    return (f32x2){t.x > 0.921875f ? big.x : small.x, t.y > 0.921875f ? big.y : small.y};
}
__device__ __forceinline__ BTerms2 boundary_terms2(float jf, f32x2 mu, f32x2 sigma)
{
    const f32x2 is = {rcp_f32(sigma.x), rcp_f32(sigma.y)};
    const f32x2 sn = splat2(0.70710678118654757f) * is, c1 = splat2(0.3989422804014327f) * is, is2 = is * is;
    const f32x2 u0 = splat2(jf - 0.5f) - mu;
    const f32x2 e0 = erf_f32x2(u0 * sn);
    const f32x2 ga = splat2(-0.5f) * u0 * u0 * is2;
    const f32x2 g0 = {__expf(ga.x), __expf(ga.y)};
    const f32x2 e1 = {from_next(e0.x), from_next(e0.y)}, g1 = {from_next(g0.x), from_next(g0.y)};
    const f32x2 u1 = u0 + splat2(1.0f);
    const f32x2 q1 = u0 * g0 - u1 * g1;
    const f32x2 q3 = u0 * u0 * u0 * g0 - u1 * u1 * u1 * g1;
    BTerms2 t;
    t.E = splat2(0.5f) * (e1 - e0);
    t.A = (g0 - g1) * c1;
    t.A2 = q1 * c1 * is2;
    t.S = q1 * c1 * is;
    t.S2 = c1 * is2 * (q3 * is2 - splat2(2.0f) * q1);
    return t;
}

// ---- pieces shared by the three kernels ------------------------------------
template <int B, bool FROM_MOVIE>
__device__ __forceinline__ void load_row(const FitParams &p, int64_t sidx, int j, bool valid, float (&d)[B])
{
    constexpr int H = B / 2;
#pragma unroll
    for (int i = 0; i < B; i++) d[i] = 0.f;
    if (valid) {
        if (FROM_MOVIE) {
            float raw[B];
            const int sl = p.slot ? p.slot[sidx] : -1;
            if (sl >= 0) {
                // the scan's exact stage left this spot's rows behind (identify_fast.hip): B rows of H + 1 packed pairs, one
                // or two cache lines per spot instead of B lines of the movie
                constexpr int NPR = H + 1;
                const uint32_t *q = p.pix + (size_t)sl * (size_t)(B * NPR) + j * NPR;
                uint32_t w[NPR];
#pragma unroll
                for (int t = 0; t < NPR; t++) w[t] = q[t];
#pragma unroll
                for (int i = 0; i < B; i++) raw[i] = (float)((i & 1) ? (w[i >> 1] >> 16) : (w[i >> 1] & 0xffffu));
            } else {
            const int64_t fr = p.frame[sidx], yy = p.y[sidx], xx = p.x[sidx];
            const int64_t o = (fr * p.Y + (yy - H + j)) * p.X + (xx - H);
            // one switch per row, not per pixel: the B loads of a row issue back to back
            switch (p.dtype) {
            case PMI_U16: { const uint16_t *q = (const uint16_t *)p.movie + o; _Pragma("unroll") for (int i = 0; i < B; i++) raw[i] = (float)q[i]; } break;
            case PMI_U8:  { const uint8_t *q = (const uint8_t *)p.movie + o;   _Pragma("unroll") for (int i = 0; i < B; i++) raw[i] = (float)q[i]; } break;
            case PMI_I16: { const int16_t *q = (const int16_t *)p.movie + o;   _Pragma("unroll") for (int i = 0; i < B; i++) raw[i] = (float)q[i]; } break;
            case PMI_U32: { const uint32_t *q = (const uint32_t *)p.movie + o; _Pragma("unroll") for (int i = 0; i < B; i++) raw[i] = (float)q[i]; } break;
            case PMI_I32: { const int32_t *q = (const int32_t *)p.movie + o;   _Pragma("unroll") for (int i = 0; i < B; i++) raw[i] = (float)q[i]; } break;
            default:      { const float *q = (const float *)p.movie + o;       _Pragma("unroll") for (int i = 0; i < B; i++) raw[i] = q[i]; } break;
            }
            }
#pragma unroll
            for (int i = 0; i < B; i++)      // localize.py:1112: float32 sub, mul, div in this order
                d[i] = (raw[i] - p.baseline) * p.sensitivity;
            div_const_row<B>(d, p.gdiv);
        } else {
            const float *sp = p.spots + sidx * (B * B) + j * B;
#pragma unroll
            for (int i = 0; i < B; i++) d[i] = sp[i];
        }
    }
}

// One Newton iteration for the eight groups of a wave (gaussmle.py:745-884 / :533-670).
//
// LDS of a group (G8_LDS floats):
//   cols[8][12]  per column i: (Ax, Ex, Sx, A2x | S2x, 1, Ax^2, Ex^2 | Sx^2, 1, Sx*Ex, -), written by lane i
//   red[8][12]   per row lane r: (num0, den0, ..., num5, den5) before the group sum
//   bc[8]        the updated parameters, written by the lane that owns each one
// The pixel loop feeds five packed-float32 accumulator pairs (v_pk_fma_f32: two sums per
// instruction); the group sum goes through LDS so that lane l ends up with num[l], den[l] only
// and updates ITS parameter (sixteen instructions in parallel instead of six parameters in
// sequence in every lane), then the six new values are broadcast back through LDS.
#ifndef G8_REFILL_K
#define G8_REFILL_K 2
#endif
// + 16 (8 for the 16-lane groups): the stride of a group is then 16 banks (mod 64), so the 12-float rows that the
// four groups of a half wave read in one ds_read_b64 of the reduction fall on disjoint banks (with a stride of
// 8 banks neighbouring groups overlapped: 16 % of the LDS cycles were bank conflicts, profiles/r02_g8_iterate_pmc.txt)
template <int GS> struct GLds { static constexpr int N = GS * 12 + GS * 12 + (GS == 8 ? 16 : 16); };

struct LaneRole {          // what lane j does in the update stage: parameter j (j < NP)
    float ms;              // max_step of its parameter
    float floor_, cap;     // lower / upper clamp of its parameter (-inf / +inf = none)
    float th;              // current value of its parameter
    float prev, prev2;     // the steps its parameter took in the two previous iterations
    float wprev;           // the second difference of its step sequence one iteration ago
    int run;               // consecutive iterations in which that second difference changed sign without shrinking
    bool conv_rel;         // its parameter takes part in the convergence test
};

template <int NP, int B>
__device__ __forceinline__ bool newton_step(const float (&d)[B], float (&th)[6], LaneRole &role, float *lds,
                                            int j, bool rowok, bool active, int &kk, double eps, int max_it,
                                            float eps_lo, float eps_hi, unsigned &flags)
{
    constexpr int GS = GroupOf<B>::GS;
    float *cols = lds, *red = lds + GS * 12, *bc = lds + 2 * GS * 12;
    const float jf = (float)j;
    const float sgy = NP == 6 ? th[5] : th[4];
    BTerms tx, ty;                                          // column j, row j
#if G8_PACKED_BOUNDARY
    {
        const BTerms2 b = boundary_terms2(jf, (f32x2){th[0], th[1]}, (f32x2){th[4], sgy});
        tx.E = b.E.x; tx.A = b.A.x; tx.A2 = b.A2.x; tx.S = b.S.x; tx.S2 = b.S2.x;
        ty.E = b.E.y; ty.A = b.A.y; ty.A2 = b.A2.y; ty.S = b.S.y; ty.S2 = b.S2.y;
    }
#else
    tx = boundary_terms(jf, th[0], th[4]);
    ty = boundary_terms(jf, th[1], sgy);
#endif
    __builtin_amdgcn_wave_barrier();
    {
        float4 *c = reinterpret_cast<float4 *>(cols + j * 12);
        c[0] = make_float4(tx.A, tx.E, tx.S, tx.A2);
        c[1] = make_float4(tx.S2, 1.0f, tx.A * tx.A, tx.E * tx.E);
        c[2] = make_float4(tx.S * tx.S, 1.0f, tx.S * tx.E, 0.0f);
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();

    const float N_ = th[2], bg = th[3];
    // (a lane beyond the last row holds a neighbour group's boundary in ty: its sums are never read, but its model
    // must not trip the guard test below)
    const float NEy = rowok ? N_ * ty.E : 0.f;
    f32x2 pA_E = {0.f, 0.f}, pS_A2 = {0.f, 0.f}, pS2_1 = {0.f, 0.f};      // cf * (Ax, Ex), (Sx, A2x), (S2x, 1)
    f32x2 qA_E = {0.f, 0.f}, qS_1 = {0.f, 0.f};                           // df * (Ax^2, Ex^2), (Sx^2, 1)
    float a_dSE = 0.f;
    // The reference guards every pixel (gaussmle.py:831-838: cf = df = 0 unless model > 10e-3, both capped at 10e4).
    // On real data no pixel of a wave ever needs either, so the sums are taken WITHOUT the guards (three instructions
    // less per pixel) while one compare and one v_max3 per pixel watch for the need; if any lane saw it, the whole
    // wave takes the sums again with the guards.  Where no guard acts the two forms are the same operations.
    bool guard = false;
    float top = 0.f;
#pragma unroll
    for (int i = 0; i < B; i++) {
        const float4 *c = reinterpret_cast<const float4 *>(cols + i * 12);
        const float4 c0 = c[0], c1 = c[1], c2 = c[2];
        const float model = __builtin_fmaf(NEy, c0.y, bg);
        const float r = rcp_f32(model);
        const float cf = __builtin_fmaf(d[i], r, -1.f);
        const float df = (d[i] * r) * r;
        guard = guard || !(model > 10e-3f);          // also true for a NaN model
        top = fmaxf(top, fmaxf(fabsf(cf), fabsf(df)));      // (source modifiers: still one v_max3_f32)
        const f32x2 cf2 = {cf, cf}, df2 = {df, df};
        pA_E = cf2 * (f32x2){c0.x, c0.y} + pA_E;
        pS_A2 = cf2 * (f32x2){c0.z, c0.w} + pS_A2;
        pS2_1 = cf2 * (f32x2){c1.x, c1.y} + pS2_1;
        qA_E = df2 * (f32x2){c1.z, c1.w} + qA_E;
        qS_1 = df2 * (f32x2){c2.x, c2.y} + qS_1;
        if (NP == 5) a_dSE += df * c2.z;
    }
    // (the sums are pinned here: left alone, the compiler sinks them behind the branch and keeps every cf, df and
    // column value alive across it — 126 registers instead of 72)
    asm volatile("" : "+v"(pA_E), "+v"(pS_A2), "+v"(pS2_1), "+v"(qA_E), "+v"(qS_1), "+v"(a_dSE));
    if (__any(guard || !(top <= 10e4f))) {
        asm volatile("" ::: "memory");      // read the columns again: keeping 21 registers of them alive for this path costs occupancy
        pA_E = (f32x2){0.f, 0.f}; pS_A2 = (f32x2){0.f, 0.f}; pS2_1 = (f32x2){0.f, 0.f};
        qA_E = (f32x2){0.f, 0.f}; qS_1 = (f32x2){0.f, 0.f};
        a_dSE = 0.f;
#pragma unroll
        for (int i = 0; i < B; i++) {
            const float4 *c = reinterpret_cast<const float4 *>(cols + i * 12);
            const float4 c0 = c[0], c1 = c[1], c2 = c[2];
            const float model = __builtin_fmaf(NEy, c0.y, bg);
            const float r = rcp_f32(model);
            const bool ok = model > 10e-3f;             // gaussmle.py:831: otherwise cf = df = 0
            // plain fminf: a NaN can only enter through a NaN pixel, and then theta is NaN from g8_init on
            const float cf = ok ? fminf(__builtin_fmaf(d[i], r, -1.f), 10e4f) : 0.f;
            const float df = ok ? fminf((d[i] * r) * r, 10e4f) : 0.f;
            const f32x2 cf2 = {cf, cf}, df2 = {df, df};
            pA_E = cf2 * (f32x2){c0.x, c0.y} + pA_E;
            pS_A2 = cf2 * (f32x2){c0.z, c0.w} + pS_A2;
            pS2_1 = cf2 * (f32x2){c1.x, c1.y} + pS2_1;
            qA_E = df2 * (f32x2){c1.z, c1.w} + qA_E;
            qS_1 = df2 * (f32x2){c2.x, c2.y} + qS_1;
            if (NP == 5) a_dSE += df * c2.z;
        }
    }
    const float a_cA = pA_E.x, a_cE = pA_E.y, a_cS = pS_A2.x, a_cA2 = pS_A2.y, a_cS2 = pS2_1.x, a_c = pS2_1.y;
    const float a_dA = qA_E.x, a_dE = qA_E.y, a_dS = qS_1.x, a_d = qS_1.y;
    float num[6], den[6];
    const float NAy = N_ * ty.A, NA2y = N_ * ty.A2, NSy = N_ * ty.S, NS2y = N_ * ty.S2;
    num[0] = NEy * a_cA;              den[0] = NEy * a_cA2 - NEy * NEy * a_dA;
    num[1] = NAy * a_cE;              den[1] = NA2y * a_cE - NAy * NAy * a_dE;
    num[2] = ty.E * a_cE;             den[2] = -ty.E * ty.E * a_dE;
    num[3] = a_c;                     den[3] = -a_d;
    if (NP == 6) {
        num[4] = NEy * a_cS;          den[4] = NEy * a_cS2 - NEy * NEy * a_dS;
        num[5] = NSy * a_cE;          den[5] = NS2y * a_cE - NSy * NSy * a_dE;
    } else {
        // isotropic sigma: du = N (Ey Sx + Ex Sy); d2u keeps the reference's precedence quirk
        num[4] = NEy * a_cS + NSy * a_cE;
        den[4] = (NEy * a_cS2 + 2.f * ty.S * a_cS + ty.S2 * a_cE)
                 - (NEy * NEy * a_dS + 2.f * NEy * NSy * a_dSE + NSy * NSy * a_dE);
        num[5] = 0.f; den[5] = 0.f;
    }
    {   // lanes beyond the last row write their slot too; the sum below reads the B rows only
        float4 *ro = reinterpret_cast<float4 *>(red + j * 12);
        ro[0] = make_float4(num[0], den[0], num[1], den[1]);
        ro[1] = make_float4(num[2], den[2], num[3], den[3]);
        ro[2] = make_float4(num[4], den[4], num[5], den[5]);
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    // lane l sums (num[l], den[l]) over the eight row lanes, in row order
    const int l = j < 6 ? j : 5;
    f32x2 nd = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < B; r++) {
        const float2 v = *reinterpret_cast<const float2 *>(red + r * 12 + 2 * l);
        nd = nd + (f32x2){v.x, v.y};
    }
    const float numl = nd.x, denl = nd.y;
    // update of this lane's parameter (gaussmle.py:860-884 / :647-670); zero denominator:
    // sigmaxy steps sign(num) * max_step (:873), sigma steps sign(num * max_step) = +-1 (:658)
    const float stepz = NP == 6 ? np_signf(numl) * role.ms : np_signf(numl * role.ms);
    const float stepn = clip_np(numl * rcp_f32(denl), role.ms);
    float nt = role.th - (denl == 0.0f ? stepz : stepn);
    nt = max_np(nt, role.floor_);
    nt = min_np(nt, role.cap);
    // D = the largest of the tested steps |delta| (x, y and for sigmaxy the two sigmas, gaussmle.py:844-852 /
    // :632-638); a NaN step stays on top of the unsigned maximum, so D is NaN and the test fails like the reference's
    const float D = __uint_as_float(gmax_u<GS>(role.conv_rel ? __float_as_uint(fabsf(role.th - nt)) : 0u));
    const bool conv = (double)D < eps;
    // a decision taken within the margin of eps may fall the other way in the reference's float64 arithmetic:
    // such spots are re-fitted by mle_strict_kernel (gaussmle_strict.hip)
    // (the margin widens with the iteration count: a slow fit takes many steps close to eps and the two arithmetics
    // drift apart by more than a few ulps), and so are spots whose curvature term is not negative for some
    // parameter — the update then runs uphill or into its clamp and the trajectory is chaotic in any arithmetic
    const float wide = fmaxf(1.0f, (float)kk * FIT_MARGIN_GROWTH);
    const float epsf = 0.5f * (eps_lo + eps_hi), epsm = 0.5f * (eps_hi - eps_lo) * wide;
    // ... and spots whose width collapses below a third of a pixel (a single hot pixel in a small box): the likelihood
    // is then flat in the position inside the pixel and the two arithmetics end up to 1e-2 px apart on equal counts
    const bool narrow = (j == 4 || (NP == 6 && j == 5)) && nt < FIT_NARROW_SIGMA;
    // ... and spots whose iteration does not contract.  The update treats every parameter on its own (a diagonal Newton
    // step), so photons, background and width — which trade against each other, the more the wider the spot is in its
    // box — over-correct jointly: the iteration map has an alternating mode, and where its factor reaches 1 a rounding
    // difference doubles from one iteration to the next instead of dying out (the two arithmetics end 1e-3 px and more
    // apart on equal iteration counts, every single decision taken far from eps).  The mode shows in a parameter's step
    // sequence d_k as a second difference w_k = d_k - 2 d_{k-1} + d_{k-2} that changes sign every iteration without
    // shrinking; FIT_WOBBLE_RUN such iterations in a row flag the spot (a damped wobble, factor < FIT_WOBBLE_RATIO, is
    // what a healthy fit shows while its step clamps release)
    const float step = role.th - nt;
    const float w = kk >= 2 ? (step - 2.0f * role.prev) + role.prev2 : 0.0f;
    const bool wob = j < NP && w * role.wprev < 0.0f && fabsf(w) > FIT_WOBBLE_RATIO * fabsf(role.wprev)
                     && fabsf(w) > FIT_WOBBLE_FLOOR * fabsf(nt);
    const int run = wob ? role.run + 1 : 0;
    const bool swing = run >= FIT_WOBBLE_RUN || (run >= 2 && kk + 1 >= FIT_WOBBLE_LATE);
    if (active) { role.run = run; role.wprev = w; role.prev2 = role.prev; role.prev = step; }
    // ... and spots with a pixel whose count is far off the model (|data / model - 1| or |data / model^2| beyond
    // FIT_TOP_FLAG: a model pinned at the 0.01 floors under negative or bright pixels): the sums then cancel from terms
    // orders of magnitude above the result and their float32 rounding moves the step by more than any margin
    const bool wild = top >= FIT_TOP_FLAG;
    const unsigned why = ((D >= epsf - epsm && D < epsf + epsm) ? FLAG_MARGIN : 0u) | ((j < NP && denl >= 0.0f) ? FLAG_CURVATURE : 0u)
                         | (narrow ? FLAG_NARROW : 0u) | (swing ? FLAG_SWING : 0u) | (wild ? FLAG_WILD : 0u)
                         | (kk >= FIT_SLOW_ITERATIONS ? FLAG_SLOW : 0u);
    flags |= active ? why : 0u;
    // the previous-iteration values the reference compares with ARE th (old_x = theta after every pass);
    // finished groups keep their state
    role.th = active ? nt : role.th;
    if (j < 8) bc[j] = role.th;
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    {
        const float4 t0 = *reinterpret_cast<const float4 *>(bc);
        const float2 t1 = *reinterpret_cast<const float2 *>(bc + 4);
        th[0] = t0.x; th[1] = t0.y; th[2] = t0.z; th[3] = t0.w; th[4] = t1.x;
        if (NP == 6) th[5] = t1.y;
    }
    kk += active ? 1 : 0;
    // (a fit that runs into max_it needs no flag of its own: with max_it above FIT_SLOW_ITERATIONS it carries the
    // slow-fit flag, below it the count says nothing about the fit — at max_it = 5 most healthy fits end that way)
    return active && !(conv || kk >= max_it);
}

// role of lane j for a freshly loaded spot
template <int NP, int B>
__device__ __forceinline__ LaneRole make_role(const float (&th)[6], const float (&ms)[6], int j)
{
    LaneRole r;
    r.ms = 0.f; r.th = 0.f; r.prev = 0.f; r.prev2 = 0.f; r.wprev = 0.f; r.run = 0;
#pragma unroll
    for (int l = 0; l < 6; l++) { r.ms = (j == l) ? ms[l] : r.ms; r.th = (j == l) ? th[l] : r.th; }
    r.floor_ = j == 2 ? 1.0f : ((j == 3 || j == 4 || (NP == 6 && j == 5)) ? 0.01f : -INFINITY);
    r.cap = (NP == 5 && j == 4) ? (float)B : INFINITY;
    r.conv_rel = NP == 6 ? (j == 0 || j == 1 || j == 4 || j == 5) : (j == 0 || j == 1);
    return r;
}

// ---- identify's exact stage, taken over from the packed scan on the fused path (FitParams::ng_io) ----------------
// unit vectors of picasso/localize.py:279-286 for a lane whose window row is only known at run time
template <int H> struct UnitTable {
    static constexpr int B = 2 * H + 1;
    float ux[B * B], uy[B * B];
    constexpr UnitTable() : ux(), uy()
    {
        for (int k = 0; k < B; k++)
            for (int l = 0; l < B; l++) { ux[k * B + l] = unit_x<H>(k, l); uy[k * B + l] = unit_y<H>(k, l); }
    }
};
template <int H> __device__ const UnitTable<H> g_unit_table = UnitTable<H>();

// one row of a candidate's neighbourhood as float32 (np.float32(frame), localize.py:332): W - 1 consecutive pixels from
// row[1] on, and row[c0] for the first column (c0 = 0, or the distance to the crop's last column when column -1 wraps).
// One switch per row: the loads of a row issue back to back; uint16 rows as one 2-byte load + packed pairs.
template <int W>
__device__ __forceinline__ void load_nb_row(const void *movie, int dtype, int64_t o, int c0, float (&nb)[W])
{
    switch (dtype) {
    case PMI_U16: {
        constexpr int NP = (W - 1) / 2;
        struct __attribute__((packed, aligned(2))) Pairs { uint32_t v[NP]; };
        const uint16_t *q = (const uint16_t *)movie + o;
        const Pairs t = *reinterpret_cast<const Pairs *>(q + 1);
        nb[0] = (float)q[c0];
#pragma unroll
        for (int k = 0; k < NP; k++) { nb[1 + 2 * k] = (float)(t.v[k] & 0xffffu); nb[2 + 2 * k] = (float)(t.v[k] >> 16); }
    } break;
    case PMI_U8:  { const uint8_t *q = (const uint8_t *)movie + o;   nb[0] = (float)q[c0]; _Pragma("unroll") for (int c = 1; c < W; c++) nb[c] = (float)q[c]; } break;
    case PMI_I16: { const int16_t *q = (const int16_t *)movie + o;   nb[0] = (float)q[c0]; _Pragma("unroll") for (int c = 1; c < W; c++) nb[c] = (float)q[c]; } break;
    case PMI_U32: { const uint32_t *q = (const uint32_t *)movie + o; nb[0] = (float)q[c0]; _Pragma("unroll") for (int c = 1; c < W; c++) nb[c] = (float)q[c]; } break;
    case PMI_I32: { const int32_t *q = (const int32_t *)movie + o;   nb[0] = (float)q[c0]; _Pragma("unroll") for (int c = 1; c < W; c++) nb[c] = (float)q[c]; } break;
    default:      { const float *q = (const float *)movie + o;       nb[0] = q[c0]; _Pragma("unroll") for (int c = 1; c < W; c++) nb[c] = q[c]; } break;
    }
}

// ---- kernel 1: initial parameters (gaussmle.py:28-168) ----------------------
template <int NP, int B, bool FROM_MOVIE>
__global__ __launch_bounds__(FIT_NT) void g8_init_kernel(FitParams p, float *__restrict__ state)
{
    constexpr int GS = GroupOf<B>::GS, NSPW = 64 / GS;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int g = lane / GS, j = lane & (GS - 1);
    const bool rowok = j < B;
    constexpr int H = B / 2;
    int64_t n = p.N;
    if (p.d_n) { int64_t dn = *p.d_n; n = dn < n ? dn : n; }
    const int64_t sidx = p.first + ((int64_t)blockIdx.x * FIT_WAVES + wid) * NSPW + g;
    const bool spot_ok = sidx < n;
    if (__builtin_amdgcn_readfirstlane((int)(p.first + ((int64_t)blockIdx.x * FIT_WAVES + wid) * NSPW >= n))) return;

    float d[B];
    bool keep = true;          // (deferred exact stage: the candidate is an identification)
    bool loaded = false;
    if constexpr (FROM_MOVIE) {
        if (p.ng_io) {
            // The scan left the exact stage of identify to this kernel (pmi_common.h NG_DEFERRED_BITS): the (B+2)^2
            // neighbourhood instead of the B^2 box — lane j < B reads neighbourhood row j + 1 (its box row with one more
            // pixel on either side), lane GS - 1 the first row (index -1 of the crop wraps to its last row, as under numba)
            // and the last row — then, per candidate: the float32 net gradient in the reference's (k, l) order
            // (picasso/localize.py:233-243: every term by the lane of its row, the sum as ONE chain of float32 additions), the
            // first-argmax rule of np.argmax (:128) and the threshold (:288).
            constexpr int W = B + 2;
            constexpr int TP = (B * B + 3) & ~3, GP = (TP + W * W + 3) & ~3;      // floats of a group: terms, then the neighbourhood
            __shared__ __attribute__((aligned(16))) float s_nb[FIT_WAVES][NSPW][GP];
            float *terms = &s_nb[wid][g][0], *nbt = terms + TP;
            const bool need = spot_ok && __float_as_uint(p.ng_io[sidx]) == NG_DEFERRED_BITS;
            const bool edge_lane = j == GS - 1;
            int64_t fr = 0;
            int yy = 0, xx = 0;
            if (spot_ok) { fr = p.frame[sidx]; yy = p.y[sidx]; xx = p.x[sidx]; }
            const int ci = yy - p.crop_y0, cj = xx - p.crop_x0;
            const int xc0 = cj - H - 1 < 0 ? p.crop_x0 + (cj - H - 1 + p.crop_cx) : xx - H - 1;
            const int r1 = edge_lane ? (ci - H - 1 < 0 ? p.crop_y0 + (ci - H - 1 + p.crop_cy) : yy - H - 1) : yy - H + j;
            // this lane's unit vectors (its window row is j): fetched before the pixels, used after them
            float ux[B], uy[B];
#pragma unroll
            for (int l = 0; l < B; l++) { ux[l] = g_unit_table<H>.ux[(rowok ? j : 0) * B + l]; uy[l] = g_unit_table<H>.uy[(rowok ? j : 0) * B + l]; }
            float nb[W], nb2[W];
#pragma unroll
            for (int c = 0; c < W; c++) { nb[c] = 0.f; nb2[c] = 0.f; }
            const int c0 = xc0 - (xx - H - 1);                     // 0 unless the first column wraps
            if (spot_ok && (rowok || edge_lane)) load_nb_row<W>(p.movie, p.dtype, (fr * p.Y + r1) * p.X + (xx - H - 1), c0, nb);
            if (spot_ok && edge_lane) load_nb_row<W>(p.movie, p.dtype, (fr * p.Y + (yy + H + 1)) * p.X + (xx - H - 1), c0, nb2);
#pragma unroll
            for (int i = 0; i < B; i++) d[i] = 0.f;
            if (spot_ok && rowok) {
#pragma unroll
                for (int i = 0; i < B; i++) d[i] = (nb[i + 1] - p.baseline) * p.sensitivity;      // localize.py:1112, as load_row
                div_const_row<B>(d, p.gdiv);
            }
            loaded = true;
            if (__any(need)) {
                if (rowok) {
#pragma unroll
                    for (int c = 0; c < W; c++) nbt[(j + 1) * W + c] = nb[c];
                }
                if (edge_lane) {
#pragma unroll
                    for (int c = 0; c < W; c++) { nbt[c] = nb[c]; nbt[(W - 1) * W + c] = nb2[c]; }
                }
                __builtin_amdgcn_wave_barrier();
                __threadfence_block();
                const float vc = nbt[(H + 1) * W + (H + 1)];
                bool fm = true;
                if (rowok) {
#pragma unroll
                    for (int l = 0; l < B; l++) {
                        // window pixel (j, l) sits at neighbourhood (j + 1, l + 1)
                        const float o = nb[l + 1];
                        const bool before = j < H || (j == H && l < H);
                        const bool centre = j == H && l == H;
                        fm = fm && (centre || (before ? vc > o : vc >= o));
                        const float gy = sub_rn(nbt[(j + 2) * W + l + 1], nbt[j * W + l + 1]);
                        const float gx = sub_rn(nb[l + 2], nb[l]);
                        terms[j * B + l] = add_rn(mul_rn(gy, uy[l]), mul_rn(gx, ux[l]));
                    }
                }
                __builtin_amdgcn_wave_barrier();
                __threadfence_block();
                float ng = 0.0f;
#pragma unroll
                for (int q = 0; q < B * B; q++)
                    if (q != H * B + H) ng = add_rn(ng, terms[q]);       // the centre is skipped (its unit vector is 0 / 0)
                const unsigned long long gm = (GS == 64 ? ~0ull : ((1ull << GS) - 1ull)) << (lane & ~(GS - 1));
                const bool first_max = (__ballot(fm) & gm) == gm;
                if (need) {
                    keep = first_max && (double)ng > p.min_ng;
                    if (j == 0) p.ng_io[sidx] = ng;
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (spot_ok && j == 0) p.accept[sidx] = keep ? 1 : 0;
        }
    }
    if (!loaded) load_row<B, FROM_MOVIE>(p, sidx, j, spot_ok && rowok, d);
    if (FROM_MOVIE && p.spots_out && spot_ok && rowok && keep) {
        // the Newton loop and the Fisher pass read the spot from here: 2 cache lines per spot instead of B
        float *o = p.spots_out + (sidx - p.first) * (B * B) + j * B;
#pragma unroll
        for (int i = 0; i < B; i++) o[i] = d[i];
    }

    double ps = 0.0, px = 0.0;
#pragma unroll
    for (int i = 0; i < B; i++) { ps += (double)d[i]; px += (double)d[i] * (double)i; }
    double sum = gsum_d<GS>(ps), sx_ = gsum_d<GS>(px), sy_ = gsum_d<GS>(ps * (double)j);
    // 3x3 edge-clipped mean filter: row-local 3-column sums, then the rows above / below
    float fmin_l = INFINITY;
    {
        double t3[B];
#pragma unroll
        for (int i = 0; i < B; i++) {
            double a = (double)d[i];
            if (i > 0) a = (double)d[i - 1] + a;
            if (i + 1 < B) a += (double)d[i + 1];
            t3[i] = a;
        }
        const bool up = j > 0, dn = j + 1 < B;
        const int nrow = 1 + (up ? 1 : 0) + (dn ? 1 : 0);
        // window sizes are nrow*2 (first / last column) or nrow*3: two reciprocals per lane, and
        // q = t*r corrected by one FMA residual step (= the correctly rounded quotient up to the
        // rare double-rounding case; the value is rounded to float32 right after)
        const double n2 = (double)(nrow * 2), n3 = (double)(nrow * 3);
        const double r2 = 1.0 / n2, r3 = 1.0 / n3;
#pragma unroll
        for (int i = 0; i < B; i++) {
            const double a = from_prev_d(t3[i]), c = from_next_d(t3[i]);
            double tot = t3[i];
            if (up) tot = a + tot;
            if (dn) tot += c;
            const bool edge = (i == 0) || (i + 1 == B);
            const double nn = edge ? n2 : n3, rr = edge ? r2 : r3;
            double q = tot * rr;
            q = fma(fma(-q, nn, tot), rr, q);
            const float filt = (float)q;
            if (rowok) fmin_l = fminf(fmin_l, filt);
        }
    }
    // np.min propagates NaN: a NaN pixel poisons the background (and through the sums everything else),
    // so the iteration kernel needs no per-pixel NaN guards
    const float bg0 = (sum != sum) ? (float)sum : gmin<GS>(fmin_l);
    double com_y, com_x;
    if (sum <= 0.0) { sum = 0.01; com_y = (B - 1) / 2.0; com_x = (B - 1) / 2.0; }
    else { com_y = sy_ / sum; com_x = sx_ / sum; }
    double photons = sum - (double)(B * B) * (double)bg0;
    photons = (photons != photons) ? photons : (photons > 1.0 ? photons : 1.0);
    // second moments of (spot - bg) along the centre column (over rows) and centre row (over columns)
    double a_sdy = 0.0, a_sy = 0.0, a_sdx = 0.0, a_sx = 0.0;
    if (rowok) {
        const float vm = d[H] - bg0;
        a_sdy = (double)vm * (double)((j - H) * (j - H));
        a_sy = (double)vm;
        if (j == H) {
#pragma unroll
            for (int i = 0; i < B; i++) {
                const float v2 = d[i] - bg0;
                a_sdx += (double)v2 * (double)((i - H) * (i - H));
                a_sx += (double)v2;
            }
        }
    }
    a_sdy = gsum_d<GS>(a_sdy); a_sy = gsum_d<GS>(a_sy); a_sdx = gsum_d<GS>(a_sdx); a_sx = gsum_d<GS>(a_sx);
    double isy = sqrt(a_sdy / a_sy), isx = sqrt(a_sdx / a_sx);
    if (!isfinite(isy)) isy = 0.01;
    if (!isfinite(isx)) isx = 0.01;
    if (isx == 0) isx = 0.01;
    if (isy == 0) isy = 0.01;

    float th[6], ms[6];
    th[0] = (float)com_x; th[1] = (float)com_y; th[2] = (float)photons; th[3] = bg0;
    if (NP == 6) { th[4] = (float)isx; th[5] = (float)isy; }
    else { th[4] = (float)((isx + isy) / 2); th[5] = 0.f; }
    ms[0] = th[4]; ms[1] = th[4];
    ms[2] = (float)(0.1 * (double)th[2]); ms[3] = (float)(0.1 * (double)th[3]);
    ms[4] = (float)(0.2 * (double)th[4]); ms[5] = (float)(0.2 * (double)th[5]);
    if (spot_ok && j == 0) {
        if (p.refit_mark) p.refit_mark[sidx - p.first] = 0;          // set by the strict re-fit of a flagged spot (gaussmle_strict.hip)
        float4 *so = reinterpret_cast<float4 *>(state + (sidx - p.first) * 12);
        so[0] = make_float4(th[0], th[1], th[2], th[3]);
        so[1] = make_float4(th[4], th[5], ms[0], ms[1]);
        so[2] = make_float4(ms[2], ms[3], ms[4], ms[5]);
    }
}

// ---- kernel 2: Newton iterations, persistent waves with per-group refill ----
template <int NP, int B, bool FROM_MOVIE>
__global__ __launch_bounds__(FIT_NT) void g8_iterate_kernel(FitParams p, const float *__restrict__ state)
{
    constexpr int GS = GroupOf<B>::GS, NSPW = 64 / GS;
    constexpr int REFILL_K = GS == 8 ? G8_REFILL_K : 1;
    __shared__ __attribute__((aligned(16))) float s_x[FIT_WAVES][NSPW][GLds<GS>::N];       // per group: columns, reduction, broadcast
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int g = lane / GS, j = lane & (GS - 1);
    const bool rowok = j < B;
    int64_t n = p.N;
    if (p.d_n) { int64_t dn = *p.d_n; n = dn < n ? dn : n; }
    // static partition: every wave owns a contiguous chunk of the batch (no queue atomics in the loop) — of the list of
    // accepted candidates when the start-value kernel took identify's exact stage over (FitParams::alist)
    const int64_t count = p.alist ? (int64_t)*p.alist_n : n - p.first;
    if (count <= 0) return;
    const int64_t total_waves = (int64_t)gridDim.x * FIT_WAVES;
    const int64_t chunk = (count + total_waves - 1) / total_waves;
    const int64_t wv = (int64_t)blockIdx.x * FIT_WAVES + __builtin_amdgcn_readfirstlane(wid);
    const int64_t base = p.alist ? 0 : p.first, lim = base + count;
    int64_t next = base + wv * chunk;
    const int64_t end = next + chunk < lim ? next + chunk : lim;
    if (next >= end) return;

    float *xs = &s_x[wid][g][0];
    float d[B], th[6], ms[6];
#pragma unroll
    for (int i = 0; i < B; i++) d[i] = 1.f;
#pragma unroll
    for (int l = 0; l < 6; l++) { th[l] = 1.f; ms[l] = 1.f; }
    LaneRole role = make_role<NP, B>(th, ms, j);
    int kk = 0;
    int64_t sidx = -1;
    bool active = false;
    unsigned flags = 0u;
    const unsigned long long below = (1ull << (lane & ~(GS - 1))) - 1ull;     // lanes of lower groups

    for (;;) {
        // Refills are batched: the refill block (publish theta, fetch state and pixels of the next spot)
        // costs about as much as a Newton iteration for the whole wave, and with eight groups that
        // converge after ~8 iterations each it would run almost every iteration.  A finished group
        // therefore waits until G8_REFILL_K groups are finished (or nothing else is running).
        const unsigned long long pend = __ballot(!active && sidx >= 0 && j == 0);      // finished, not yet published
        const unsigned long long empty = __ballot(!active && sidx < 0 && j == 0);      // no spot at all
        const bool any_active = __any(active);
        if (__popcll(pend) >= REFILL_K || (pend != 0 && !any_active) || (empty != 0 && next < end)) {
            // finished groups publish theta / iteration count, then take the next spots of the chunk
            // (the curvature flag is per parameter lane: any lane of the group flags the spot)
            const unsigned gflag = gor_u<GS>(flags);
            if (!active && sidx >= 0 && j == 0) {
                float *to = p.thetas + sidx * 6;
#pragma unroll
                for (int l = 0; l < 5; l++) to[l] = th[l];
                to[5] = NP == 6 ? th[5] : th[4];
                p.iterations[sidx] = kk;
                if (gflag && p.flag_list) {
                    p.flag_list[atomicAdd(p.flag_count, 1u)] = (int32_t)sidx;
                    count_flag_reasons(p.flag_reasons, gflag);
                }
            }
            const unsigned long long want = pend | empty;
            const int64_t cand = next + __popcll(want & below);
            if (!active) {
                sidx = -1;
                if (cand < end) {
                    sidx = p.alist ? (int64_t)p.alist[cand] : cand;
                    const float4 *si = reinterpret_cast<const float4 *>(state + (sidx - p.first) * 12);
                    const float4 s0 = si[0], s1 = si[1], s2 = si[2];
                    th[0] = s0.x; th[1] = s0.y; th[2] = s0.z; th[3] = s0.w; th[4] = s1.x; th[5] = s1.y;
                    ms[0] = s1.z; ms[1] = s1.w; ms[2] = s2.x; ms[3] = s2.y; ms[4] = s2.z; ms[5] = s2.w;
                    load_row<B, FROM_MOVIE>(p, sidx, j, rowok, d);
                    role = make_role<NP, B>(th, ms, j);
                    kk = 0;
                    flags = 0u;
                    active = p.max_it > 0;
                    if (!active && j == 0) {          // max_it == 0: the initial theta is the result
                        float *to = p.thetas + sidx * 6;
#pragma unroll
                        for (int l = 0; l < 5; l++) to[l] = th[l];
                        to[5] = NP == 6 ? th[5] : th[4];
                        p.iterations[sidx] = 0;
                        sidx = -1;
                    }
                }
            }
            next += __popcll(want);
            if (!__any(active)) {
                if (next >= end) break;
                continue;
            }
        } else if (!any_active) {
            break;                                     // nothing running, nothing pending, chunk exhausted
        }
        active = newton_step<NP, B>(d, th, role, xs, j, rowok, active, kk, p.eps, p.max_it, p.eps_lo, p.eps_hi, flags);
    }
}

// ---- kernel 3: Fisher matrix and log-likelihood (gaussmle.py:673-742, 887-954)
// M[k][l] = sum over pixels of du_k du_l / model.  Every derivative is (row factor) x (column
// function): du = (N Ey * Ax, N Ay * Ex, Ey * Ex, 1, N Ey * Sx, N Sy * Ex) — so a lane (one row)
// only accumulates the ten products of the four column functions {Ax, Ex, 1, Sx} weighted by
// 1/model (float64, like the reference's accumulation), forms its 21 entries from them with its
// row factors once, and the 8-lane sums go through LDS, lane j collecting entries j, j+8, j+16.
template <int NP, int B, bool FROM_MOVIE>
__global__ __launch_bounds__(FIT_NT) void g8_final_kernel(FitParams p)
{
    // per group: 8 columns x 12 floats (pair products), then reused as 8 lanes x 7 doubles per reduction round
    constexpr int GS = GroupOf<B>::GS, NSPW = 64 / GS;
    __shared__ __attribute__((aligned(16))) double s_x[FIT_WAVES][NSPW][GS * 7];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int g = lane / GS, j = lane & (GS - 1);
    const bool rowok = j < B;
    const float jf = (float)j;
    int64_t n = p.N;
    if (p.d_n) { int64_t dn = *p.d_n; n = dn < n ? dn : n; }
    int64_t sidx = p.first + ((int64_t)blockIdx.x * FIT_WAVES + wid) * NSPW + g;
    bool spot_ok = sidx < n;
    if (p.final_list) {      // the spots of the second re-fit only
        const int64_t w0 = ((int64_t)blockIdx.x * FIT_WAVES + wid) * NSPW, items = (int64_t)*p.final_list_n;
        if (__builtin_amdgcn_readfirstlane((int)(w0 >= items))) return;
        spot_ok = w0 + g < items;
        sidx = spot_ok ? (int64_t)p.final_list[w0 + g] : p.first;
    } else if (__builtin_amdgcn_readfirstlane((int)(p.first + ((int64_t)blockIdx.x * FIT_WAVES + wid) * NSPW >= n))) return;
    else if (p.accept) spot_ok = spot_ok && p.accept[sidx] != 0;        // a candidate identify's exact stage rejected (g8_init)

    float d[B], th[6];
    load_row<B, FROM_MOVIE>(p, sidx, j, spot_ok && rowok, d);
#pragma unroll
    for (int l = 0; l < 6; l++) th[l] = spot_ok ? p.thetas[sidx * 6 + l] : 1.f;
    double *red = &s_x[wid][g][0];
    float *cols = reinterpret_cast<float *>(red);
    const float sgy = NP == 6 ? th[5] : th[4];
    const BTerms tx = boundary_terms(jf, th[0], th[4]);
    const BTerms ty = boundary_terms(jf, th[1], sgy);
    {
        // column j: (AA, AE, A, AS | EE, E, ES, 1 | S, SS, -, -)
        float4 *c = reinterpret_cast<float4 *>(cols + j * 12);
        c[0] = make_float4(tx.A * tx.A, tx.A * tx.E, tx.A, tx.A * tx.S);
        c[1] = make_float4(tx.E * tx.E, tx.E, tx.E * tx.S, 1.0f);
        c[2] = make_float4(tx.S, tx.S * tx.S, 0.f, 0.f);
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    // T: AA AE A1 AS EE E1 ES 11 S1 SS
    double T[10];
#pragma unroll
    for (int e = 0; e < 10; e++) T[e] = 0.0;
    float ll_loc = 0.f;
    const float N_ = th[2];
    const float NEy = N_ * ty.E;
#pragma unroll
    for (int i = 0; i < B; i++) {
        const float4 *c = reinterpret_cast<const float4 *>(cols + i * 12);
        const float4 c0 = c[0], c1 = c[1];
        const float2 c2 = *reinterpret_cast<const float2 *>(cols + i * 12 + 8);
        const float model = NEy * c1.y + th[3];
        if (rowok) {
            const double md = (double)model;
            double inv = (double)rcp_f32(model);
            inv = inv * (2.0 - md * inv);                       // ~1e-14 relative: one Newton step on a 1-ulp float seed
            inv = inv * (2.0 - md * inv);
            T[0] += (double)c0.x * inv; T[1] += (double)c0.y * inv; T[2] += (double)c0.z * inv; T[3] += (double)c0.w * inv;
            T[4] += (double)c1.x * inv; T[5] += (double)c1.y * inv; T[6] += (double)c1.z * inv; T[7] += inv;
            T[8] += (double)c2.x * inv; T[9] += (double)c2.y * inv;
            if (model > 0.f) {
                if (d[i] > 0.f) ll_loc += d[i] * __logf(model / d[i]) - (model - d[i]);
                else ll_loc += -model;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // this row's contribution to the upper triangle, in the order (0,0) (0,1) ... (NP-1,NP-1)
    const double rNE = (double)NEy, rNA = (double)(N_ * ty.A), rE = (double)ty.E, rNS = (double)(N_ * ty.S);
    double Mloc[21];
#pragma unroll
    for (int e = 0; e < 21; e++) Mloc[e] = 0.0;
    if (NP == 6) {
        // column function of parameter k: A E E 1 S E; row factor: NE NA E 1 NE NS
        Mloc[0] = rNE * rNE * T[0];  Mloc[1] = rNE * rNA * T[1];  Mloc[2] = rNE * rE * T[1];   Mloc[3] = rNE * T[2];
        Mloc[4] = rNE * rNE * T[3];  Mloc[5] = rNE * rNS * T[1];
        Mloc[6] = rNA * rNA * T[4];  Mloc[7] = rNA * rE * T[4];   Mloc[8] = rNA * T[5];        Mloc[9] = rNA * rNE * T[6];
        Mloc[10] = rNA * rNS * T[4];
        Mloc[11] = rE * rE * T[4];   Mloc[12] = rE * T[5];        Mloc[13] = rE * rNE * T[6];  Mloc[14] = rE * rNS * T[4];
        Mloc[15] = T[7];             Mloc[16] = rNE * T[8];       Mloc[17] = rNS * T[5];
        Mloc[18] = rNE * rNE * T[9]; Mloc[19] = rNE * rNS * T[6];
        Mloc[20] = rNS * rNS * T[4];
    } else {
        // parameter 4 = isotropic sigma: du4 = NE * S + NS * E (two separable terms)
        Mloc[0] = rNE * rNE * T[0];  Mloc[1] = rNE * rNA * T[1];  Mloc[2] = rNE * rE * T[1];   Mloc[3] = rNE * T[2];
        Mloc[4] = rNE * (rNE * T[3] + rNS * T[1]);
        Mloc[5] = rNA * rNA * T[4];  Mloc[6] = rNA * rE * T[4];   Mloc[7] = rNA * T[5];
        Mloc[8] = rNA * (rNE * T[6] + rNS * T[4]);
        Mloc[9] = rE * rE * T[4];    Mloc[10] = rE * T[5];        Mloc[11] = rE * (rNE * T[6] + rNS * T[4]);
        Mloc[12] = T[7];             Mloc[13] = rNE * T[8] + rNS * T[5];
        Mloc[14] = rNE * rNE * T[9] + 2.0 * rNE * rNS * T[6] + rNS * rNS * T[4];
    }
    // group sums through LDS, seven entries per round; lane j keeps entries j, j+GS, j+2GS
    constexpr int NE_ = NP * (NP + 1) / 2;
    double mine[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int r0 = 0; r0 < 21; r0 += 7) {
        if (r0 < NE_) {
#pragma unroll
            for (int e = 0; e < 7; e++) red[j * 7 + e] = Mloc[r0 + e];
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const int e = j + GS * t - r0;                  // entry j + GS t lives in this round when 0 <= e < 7
                if (e >= 0 && e < 7) {
                    double acc = 0.0;
#pragma unroll
                    for (int r = 0; r < B; r++) acc += red[r * 7 + e];      // rows beyond the box hold zeros
                    mine[t] = acc;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    double *fo = p.fisher + (sidx - p.first) * FISHER_STRIDE;
    if (spot_ok) {
#pragma unroll
        for (int t = 0; t < 3; t++)
            if (j + GS * t < NE_) fo[j + GS * t] = mine[t];
    }
    const float ll = gsum<GS>(ll_loc);
    if (spot_ok && j == 0) p.loglik[sidx] = ll;
}

// stages: FIT_STAGE_NEWTON = initial parameters + Newton iterations (thetas, iterations, borderline flags),
// FIT_STAGE_FINAL = Fisher matrix and log-likelihood at the thetas in memory
template <int NP, int B, bool FROM_MOVIE>
static void launch_g8(const FitParams &p, float *state, int cu_count, int stages, hipStream_t s)
{
    constexpr int NSPW = 64 / GroupOf<B>::GS;
    const int64_t count = p.N - p.first;
    const int64_t waves = (count + NSPW - 1) / NSPW;
    const dim3 flat((unsigned)((waves + FIT_WAVES - 1) / FIT_WAVES));
    // persistent iterate grid: 6 workgroups (24 waves, 73 VGPRs each) per CU, each wave owning >= 64 spots when possible
    // (more, smaller chunks — 36 ... 144 waves' worth per CU, the surplus waiting for a slot — leave the fit where it is,
    // 1.59 ms on config 2, and cost the step with two ranges in flight 0.1 - 0.2 ms: the scan beside it gets its slots later)
    const int64_t pw = std::max<int64_t>(1, std::min<int64_t>((int64_t)cu_count * 24, (count + 63) / 64));
    const dim3 pers((unsigned)((pw + FIT_WAVES - 1) / FIT_WAVES));
    if (stages & (FIT_STAGE_NEWTON | FIT_STAGE_INIT_ONLY))
        hipLaunchKernelGGL((g8_init_kernel<NP, B, FROM_MOVIE>), flat, dim3(FIT_NT), 0, s, p, state);
    if (stages & (FIT_STAGE_NEWTON | FIT_STAGE_ITERATE_ONLY))
        hipLaunchKernelGGL((g8_iterate_kernel<NP, B, FROM_MOVIE>), pers, dim3(FIT_NT), 0, s, p, (const float *)state);
    if (stages & FIT_STAGE_FINAL)
        hipLaunchKernelGGL((g8_final_kernel<NP, B, FROM_MOVIE>), flat, dim3(FIT_NT), 0, s, p);
}

template <int NP, bool FROM_MOVIE>
static void launch_g8_box(const FitParams &p, float *state, int cu_count, int stages, hipStream_t s)
{
    switch (p.box) {
    case 3: launch_g8<NP, 3, FROM_MOVIE>(p, state, cu_count, stages, s); break;
    case 5: launch_g8<NP, 5, FROM_MOVIE>(p, state, cu_count, stages, s); break;
    case 7: launch_g8<NP, 7, FROM_MOVIE>(p, state, cu_count, stages, s); break;
    case 9: launch_g8<NP, 9, FROM_MOVIE>(p, state, cu_count, stages, s); break;
    case 11: launch_g8<NP, 11, FROM_MOVIE>(p, state, cu_count, stages, s); break;
    case 13: launch_g8<NP, 13, FROM_MOVIE>(p, state, cu_count, stages, s); break;
    default: launch_g8<NP, 15, FROM_MOVIE>(p, state, cu_count, stages, s); break;
    }
}

// boxes 3..15.  `state` = 12 floats per spot of the batch.  Returns false when the box is not handled.
bool launch_fit_g8(const FitParams &p, int method, bool from_movie, int cu_count, float *state, int stages, hipStream_t s)
{
    if (p.box > 15) return false;
    if (method == PMI_MLE_SIGMAXY) {
        if (from_movie) launch_g8_box<6, true>(p, state, cu_count, stages, s); else launch_g8_box<6, false>(p, state, cu_count, stages, s);
    } else {
        if (from_movie) launch_g8_box<5, true>(p, state, cu_count, stages, s); else launch_g8_box<5, false>(p, state, cu_count, stages, s);
    }
    return true;
}

}  // namespace pmi
