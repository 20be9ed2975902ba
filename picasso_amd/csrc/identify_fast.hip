// identify_fast.hip — the uint16 fast path of identify: a register-pipelined,
// packed-u16 first-argmax scan streaming rows straight from HBM (no LDS staging).
//
// Same semantics as identify_scan_kernel in identify.hip (picasso/localize.py:97-134
// _local_maxima, :202-244 _net_gradient, :288 threshold); only the schedule differs:
//
//   * one wavefront owns a band of RB rows x 512 columns of one frame (frames at most 256 / 128 pixels wide:
//     two / four consecutive bands side by side, template parameter P); lane l holds
//     8 consecutive pixels of the current row as four packed u16x2 registers (one
//     16-byte global load per row) plus the 4 pixels on either side, taken from the
//     neighbouring lanes' registers by DPP wave_shr/wave_shl (lanes 0 and 63 load
//     theirs with one masked 8-byte load);
//   * horizontal: L = max of the h pixels left of each pixel, R = max of the h to
//     the right, Hrow = max(L, v, R), all as v_pk_max_u16 on aligned/odd pixel pairs
//     (odd pairs by v_alignbit);
//   * vertical: U[r] = max(Hrow[r-h+1..r]) from a register ring; a pixel of row r'
//     is the FIRST maximum of its window iff v > max(U[r'-1], L) and
//     v >= max(R, U[r'+h]); the first half is folded into pre = max(sat(bef+1), R)
//     at row r', the second is tested h rows later: sat(max(pre, U) - v) == 0;
//   * candidates (~2 % of pixels on shot noise) go to a per-wave LDS list; a range
//     bound |ng| <= P_box * (max - min over statistics cells covering the stencil)
//     rejects the ones that cannot reach min_ng without touching memory again; the
//     survivors get the exact float32 net gradient in the reference's (k,l) order.
//
// HBM traffic: every pixel is fetched once per band plus 2(h+2) halo rows
// (RB = 64: +12.5 %, absorbed by L2/MALL because a frame's bands run on one XCD).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "pmi_common.h"

#pragma clang fp contract(off)

namespace pmi {

// float32 ops that must round exactly like the reference's unfused arithmetic.  They are
// defined HERE, under the pragma above, so the instructions carry no `contract` flag
// (the __f*_rn helpers of the HIP headers are compiled with contraction allowed).
static __device__ __forceinline__ float mul_rn(float a, float b) { return a * b; }
static __device__ __forceinline__ float add_rn(float a, float b) { return a + b; }
static __device__ __forceinline__ float sub_rn(float a, float b) { return a - b; }

// float32 sqrt of 0..128 (the squared lengths that occur for box <= 17), correctly rounded; the host
// checks them against sqrtf before the first launch (unit_vectors_match).
constexpr float SQRT_F32[129] = {
    0x0.0p+0f, 0x1.0p+0f, 0x1.6a09e6p+0f, 0x1.bb67aep+0f, 0x1.0p+1f, 0x1.1e377ap+1f, 0x1.3988e2p+1f,
    0x1.52a7fap+1f, 0x1.6a09e6p+1f, 0x1.8p+1f, 0x1.94c584p+1f, 0x1.a8872ap+1f, 0x1.bb67aep+1f, 0x1.cd82b4p+1f,
    0x1.deeea2p+1f, 0x1.efbdecp+1f, 0x1.0p+2f, 0x1.07e0f6p+2f, 0x1.0f876cp+2f, 0x1.16f834p+2f, 0x1.1e377ap+2f,
    0x1.2548ecp+2f, 0x1.2c2fc6p+2f, 0x1.32eee8p+2f, 0x1.3988e2p+2f, 0x1.4p+2f, 0x1.465656p+2f, 0x1.4c8dc2p+2f,
    0x1.52a7fap+2f, 0x1.58a68ap+2f, 0x1.5e8adep+2f, 0x1.64564p+2f, 0x1.6a09e6p+2f, 0x1.6fa6eap+2f, 0x1.752e5p+2f,
    0x1.7aa10ep+2f, 0x1.8p+2f, 0x1.854bfcp+2f, 0x1.8a85c2p+2f, 0x1.8fae0cp+2f, 0x1.94c584p+2f, 0x1.99cccap+2f,
    0x1.9ec474p+2f, 0x1.a3ad12p+2f, 0x1.a8872ap+2f, 0x1.ad5336p+2f, 0x1.b211b2p+2f, 0x1.b6c30cp+2f, 0x1.bb67aep+2f,
    0x1.cp+2f, 0x1.c48c6p+2f, 0x1.c90d2ap+2f, 0x1.cd82b4p+2f, 0x1.d1ed52p+2f, 0x1.d64d52p+2f, 0x1.daa2fep+2f,
    0x1.deeea2p+2f, 0x1.e3307cp+2f, 0x1.e768d4p+2f, 0x1.eb97e4p+2f, 0x1.efbdecp+2f, 0x1.f3db22p+2f, 0x1.f7efbep+2f,
    0x1.fbfbf8p+2f, 0x1.0p+3f, 0x1.01fe04p+3f, 0x1.03f82p+3f, 0x1.05ee68p+3f, 0x1.07e0f6p+3f, 0x1.09cfdcp+3f,
    0x1.0bbb3p+3f, 0x1.0da304p+3f, 0x1.0f876cp+3f, 0x1.11687ap+3f, 0x1.13464p+3f, 0x1.1520cep+3f, 0x1.16f834p+3f,
    0x1.18cc82p+3f, 0x1.1a9dc8p+3f, 0x1.1c6c16p+3f, 0x1.1e377ap+3f, 0x1.2p+3f, 0x1.21c5b8p+3f, 0x1.2388acp+3f,
    0x1.2548ecp+3f, 0x1.270682p+3f, 0x1.28c17cp+3f, 0x1.2a79e4p+3f, 0x1.2c2fc6p+3f, 0x1.2de32cp+3f, 0x1.2f9422p+3f,
    0x1.3142b4p+3f, 0x1.32eee8p+3f, 0x1.3498cap+3f, 0x1.364064p+3f, 0x1.37e5bep+3f, 0x1.3988e2p+3f, 0x1.3b29d8p+3f,
    0x1.3cc8aap+3f, 0x1.3e655ep+3f, 0x1.4p+3f, 0x1.419894p+3f, 0x1.432f24p+3f, 0x1.44c3b8p+3f, 0x1.465656p+3f,
    0x1.47e706p+3f, 0x1.4975cep+3f, 0x1.4b02b4p+3f, 0x1.4c8dc2p+3f, 0x1.4e16fep+3f, 0x1.4f9e6cp+3f, 0x1.512414p+3f,
    0x1.52a7fap+3f, 0x1.542a28p+3f, 0x1.55aaap+3f, 0x1.57296ap+3f, 0x1.58a68ap+3f, 0x1.5a2208p+3f, 0x1.5b9be6p+3f,
    0x1.5d142cp+3f, 0x1.5e8adep+3f, 0x1.6p+3f, 0x1.617398p+3f, 0x1.62e5acp+3f, 0x1.64564p+3f, 0x1.65c558p+3f,
    0x1.6732f8p+3f, 0x1.689f26p+3f, 0x1.6a09e6p+3f};
// unit vectors of picasso/localize.py:279-286 as compile-time float32 constants:
// ux[k][l] = (H - l) / |(H - l, H - k)|, uy[k][l] = (H - k) / |...|  (float32 sqrt and divide)
template <int H> constexpr float unit_x(int k, int l)
{
    const int vx = H - l, vy = H - k;
    return (vx == 0 && vy == 0) ? 0.0f : (float)vx / SQRT_F32[vx * vx + vy * vy];
}
template <int H> constexpr float unit_y(int k, int l) { return unit_x<H>(l, k); }

typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32;

__device__ __forceinline__ u32 pk_max(u32 a, u32 b)
{
    return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, b)));
}
__device__ __forceinline__ u32 pk_min(u32 a, u32 b)
{
    u32 d;     // asm: min(x, 1) would otherwise be rewritten into per-half compare/select (SDWA + nops)
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// v_pk_{add,sub}_u16 with the clamp bit (saturating); written as asm because the elementwise
// saturating builtins expand into compare/select sequences on this toolchain
__device__ __forceinline__ u32 pk_add_sat(u32 a, u32 b)
{
    u32 d;
    asm("v_pk_add_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ u32 pk_sub_sat(u32 a, u32 b)
{
    u32 d;
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

struct FastParams {
    const uint16_t *movie;
    int64_t Y, X;
    int y0, x0, cy, cx;
    int64_t f_lo, label_off;
    int nframes;
    int bands, segs, bpf;      // bands per frame, 512-col segments per row, blocks per frame
    int box;
    double min_ng;
    double bound_c;            // P_box: sum of the positive stencil weights, with margin
    int dbg;                   // PMI_IDENTIFY_DBG: 1 = skip the exact net gradient, 2 = skip the record append (timing only)
};

#ifndef FAST_WAVES_N
#define FAST_WAVES_N 1          // one wavefront per workgroup: frames with few bands leave no idle waves (128 x 128: 1.3 -> 2.2 TB/s), larger ones gain 2-3 %
#endif
constexpr int FAST_WAVES = FAST_WAVES_N;
#ifndef FAST_D_H3
#define FAST_D_H3 6          // rows prefetched ahead in the box-7 scan (divides its unroll period of 6): 6 instead of 3, 1.65 -> 1.60 ms
#endif
#ifndef FAST_D_H2
#define FAST_D_H2 2
#endif
#ifndef FAST_D_H4
#define FAST_D_H4 4
#endif
#ifndef FAST_D_H5
#define FAST_D_H5 5
#endif
#ifndef FAST_D_H6
#define FAST_D_H6 3
#endif
#ifndef FAST_MIN_WAVES
#define FAST_MIN_WAVES 4      // waves per SIMD the register allocator must leave room for
#endif

// pair `s` = pixels (s, s+1) relative to the lane's first pixel.  A holds NA packed pairs: the lane's
// own four in the middle, NB = NA - 4 neighbour pixels on either side (4 for boxes up to 9, 8 for
// boxes 11 and 13): A[k] = pixels (2k - NB, 2k - NB + 1), Bp[k] = (2k - NB - 1, 2k - NB).
template <int S, int NA>
__device__ __forceinline__ u32 pair_at(const u32 (&A)[NA], const u32 (&Bp)[NA])
{
    constexpr int NB = NA - 4;
    if constexpr (((S + NB) & 1) == 0) return A[(S + NB) / 2];
    else return Bp[(S + NB + 1) / 2];
}

// Q = own pair 0..3 (pixels 2Q, 2Q+1)
template <int H, int Q, int T, int NA>
struct LR {
    static __device__ __forceinline__ u32 left(const u32 (&A)[NA], const u32 (&Bp)[NA])
    {
        constexpr int c0 = 2 * Q;
        u32 v = pair_at<c0 - H + T, NA>(A, Bp);
        if constexpr (T + 1 < H) return pk_max(v, LR<H, Q, T + 1, NA>::left(A, Bp));
        else return v;
    }
    static __device__ __forceinline__ u32 right(const u32 (&A)[NA], const u32 (&Bp)[NA])
    {
        constexpr int c0 = 2 * Q;
        u32 v = pair_at<c0 + 1 + T, NA>(A, Bp);
        if constexpr (T + 1 < H) return pk_max(v, LR<H, Q, T + 1, NA>::right(A, Bp));
        else return v;
    }
};

struct RowRegs { uint4 m; uint4 e; };   // 8 own pixels + (lanes 0 / 63 only) the 4 or 8 pixels beyond the wave's edge (e.z, e.w: boxes 11, 13)

// value of lane-1 / lane+1 across the whole wavefront; `edge` is returned where no such lane exists
__device__ __forceinline__ u32 from_lane_below(u32 v, u32 edge) { return (u32)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x138, 0xf, 0xf, false); }   // wave_shr:1
__device__ __forceinline__ u32 from_lane_above(u32 v, u32 edge) { return (u32)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x130, 0xf, 0xf, false); }   // wave_shl:1

// Exact float32 net gradient of one candidate whose stencil does not wrap, in the reference's
// (k, l) order.  The (2H+3)^2 neighbourhood is fetched as packed pixel pairs (one wide load + one
// 2-byte load per row) in two batches of rows whose loads are all in flight together: the rows
// have usually left L2 by now and a row-by-row loop pays the memory latency 2H+1 times, while
// holding all 2H+3 rows at once does not fit the register budget of four waves per SIMD.
// The rows are only 2-byte aligned; gfx950 runs global loads in unaligned mode.
template <int H, int K0, int K1>
__device__ __forceinline__ float exact_ng_rows(const uint16_t *__restrict__ base, int64_t X, float ng)
{
    constexpr int BOX = 2 * H + 1, W = 2 * H + 3, NP = (W - 1) / 2;   // W is odd: NP pairs + one single pixel
    constexpr int R0 = K0, NR = K1 - K0 + 2;                           // window rows K0..K1-1 need neighbourhood rows K0..K1+1
    struct __attribute__((packed, aligned(2))) Pairs { u32 v[NP]; };
    u32 pk[NR][NP];
    u32 last[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const uint16_t *row = base + (int64_t)(R0 + r) * X;
        const Pairs t = *reinterpret_cast<const Pairs *>(row);
#pragma unroll
        for (int q = 0; q < NP; q++) pk[r][q] = t.v[q];
        last[r] = row[W - 1];
    }
    auto px = [&](int r, int b) -> float {
        r -= R0;
        if (b == W - 1) return (float)last[r];
        return (float)((b & 1) ? (pk[r][b >> 1] >> 16) : (pk[r][b >> 1] & 0xffffu));
    };
#pragma unroll
    for (int k = K0; k < K1; k++) {
#pragma unroll
        for (int l = 0; l < BOX; l++) {
            if (k == H && l == H) continue;                        // the centre is skipped (its unit vector is 0/0)
            // window pixel (k, l) sits at neighbourhood (k+1, l+1); the unit vectors are literals
            const float cy = unit_y<H>(k, l), cx = unit_x<H>(k, l);
            const float gy = sub_rn(px(k + 2, l + 1), px(k, l + 1));
            const float gx = sub_rn(px(k + 1, l + 2), px(k + 1, l));
            // A zero component contributes +-0, which never changes the float32 sum (ng starts at +0
            // and x + (-0) == x): the product with it is dropped instead of computed.
            const float sacc = cy == 0.0f ? mul_rn(gx, cx) : (cx == 0.0f ? mul_rn(gy, cy) : add_rn(mul_rn(gy, cy), mul_rn(gx, cx)));
            ng = add_rn(ng, sacc);
        }
    }
    return ng;
}

template <int H>
__device__ __forceinline__ float exact_ng_noWrap(const uint16_t *__restrict__ src, int64_t X, int i, int j)
{
    constexpr int BOX = 2 * H + 1, KM = (BOX + 1) / 2;
    const uint16_t *base = src + (int64_t)(i - H - 1) * X + (j - H - 1);
    float ng = exact_ng_rows<H, 0, KM>(base, X, 0.0f);
    // the second batch starts only when the first sum is done (keeps its loads from being hoisted
    // above the first batch, which would double the live registers)
    asm volatile("" : "+v"(ng), "+v"(base));
    return exact_ng_rows<H, KM, BOX>(base, X, ng);
}

// P > 1 (frames at most 512 / P pixels wide): the wave works on P consecutive bands of the frame at once, NL = 64 / P
// lanes each, in lock step (row step r handles row r of every sub-band).  Neighbour pixels that a lane at a
// sub-band's edge takes from the adjacent sub-band only reach positions outside the crop, which are masked.
template <int H, int RB, int D, int P = 1>
__global__ __launch_bounds__(FAST_WAVES * 64, (H <= 4 ? FAST_MIN_WAVES : (H <= 6 ? 3 : 2))) void identify_scan_u16_fast_kernel(
    FastParams p, const float *__restrict__ uxy, Record *__restrict__ recs, long long cap,
    unsigned long long *__restrict__ shard_cnt, int *__restrict__ frame_count)
{
    constexpr int BOX = 2 * H + 1;
    constexpr int NL = 64 / P;                             // lanes per sub-band
    static_assert(P == 1 || P == 2 || P == 4 || P == 8, "sub-bands split the wavefront evenly");
    // Boxes 11 to 17 (H = 5..8) need 8 neighbour pixels per side and keep an Hrow ring of H slots of which
    // the slot about to be overwritten is skipped: both rings then share the period H.
    constexpr bool WIDE = H >= 5;
    constexpr int NB = WIDE ? 8 : 4, NA = 4 + NB, OWN = NB / 2;
    constexpr int HR = WIDE ? H : (H > 1 ? H - 1 : 1);     // Hrow ring length (unused when H == 1)
    constexpr int U_ = WIDE ? (H == 5 ? 10 : 2 * H) : ((H <= 2) ? 4 : H * (H - 1));   // unroll period: a multiple of both ring periods
    static_assert(U_ % D == 0, "prefetch depth must divide the unroll period");
    constexpr int GS = U_ >= 6 ? U_ : 8;                   // rows per MIN statistics group (cells of 8 columns)
    constexpr int GM = H <= 2 ? 2 : H;                     // rows per MAX statistics group (cells of 4 columns); divides U_
    static_assert(U_ % GM == 0 && GS % GM == 0, "max groups close inside the unrolled body");
    static_assert(2 * H + 2 <= 3 * GM && 2 * H + 2 <= 2 * GS, "the stored windows (4 max groups, 3 min groups) must cover the stencil rows");
    constexpr int NR = RB + 2 * H + 2;                     // pipeline rows: band + H halo + 1 stats row each side
    constexpr int NRP = ((NR + GS - 1) / GS) * GS;
    constexpr int NG = NRP / GS, NGM = NRP / GM;
    // Local maxima are > H apart, so a band holds at most RB*512/(H+1)^2 of them; shot noise gives ~1/(2H+1)^2
    // per pixel.  The list is sized at 3/4 of the geometric bound (4 workgroups per CU fit in LDS);
    // a denser band takes the exact rescan path below.
    constexpr int LIST_GEO = (RB * 512 / ((H + 1) * (H + 1))) * 3 / 4;
    constexpr int LIST = LIST_GEO < 1024 ? LIST_GEO : 1024;
    static_assert(RB <= 128, "list entries keep the row in 7 bits");

    __shared__ unsigned short s_list[FAST_WAVES][LIST];   // (row - band_lo) << 9 | (col - 512 * seg)
    // Range statistics for the |ng| bound.  The minimum barely varies (background floor): one value
    // per lane (8 columns) per GS rows.  The maximum is what a nearby emitter inflates, so it is kept
    // on a finer grid: two 4-column cells per lane per GM rows (low half = left cell).  What is stored
    // for group g is already the extreme over the window of groups ending at g (4 max groups, 3 min
    // groups), so that a candidate's query is one entry per lane it touches instead of one per group.
    __shared__ unsigned short s_min[FAST_WAVES][NG][64];
    __shared__ unsigned s_max[FAST_WAVES][NGM][64];
    __shared__ float s_u[2 * BOX * BOX];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave id as a scalar: row addressing stays on the SALU
    for (int i = threadIdx.x; i < 2 * BOX * BOX; i += FAST_WAVES * 64) s_u[i] = uxy[i];
    __syncthreads();

    // XCD-aware mapping: all blocks of a frame share blockIdx % 8 (= the XCD they run on)
    const int m = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const int fi = (m / p.bpf) * 8 + xcd;
    const int unit = (m % p.bpf) * FAST_WAVES + w;
    if (fi >= p.nframes || unit >= p.bands * p.segs) return;
    const int band = unit / p.segs, seg = unit - band * p.segs;       // P > 1: `band` counts groups of P bands, seg = 0
    const int sub = P > 1 ? lane / NL : 0;                            // this lane's sub-band and its rows' offset
    const int sub_rows = sub * RB;

    // The crop may start at any column: lanes work on 8-pixel chunks aligned in the FRAME (16-byte
    // loads), xoff pixels of the first chunk lie left of the crop.  Positions outside the crop are
    // masked; every allowed position has its whole window inside the crop, so looking at frame pixels
    // beyond the crop edge never changes a decision.  Column indices in the candidate list and the
    // statistics cells are relative to the aligned origin, j = ja - xoff is relative to the crop.
    const int xoff = p.x0 & 7;
    const int nch = (xoff + p.cx + 7) >> 3;           // 8-pixel chunks per row
    const int c8 = P > 1 ? lane % NL : seg * 64 + lane;
    const bool lane_valid = c8 < nch;
    const int cm = min(c8, nch - 1);
    const uint16_t *src = p.movie + ((int64_t)(p.f_lo + fi) * p.Y + p.y0) * p.X + p.x0;
    const int col_m = cm * 8;
    const int col_l = cm > 0 ? col_m - NB : col_m;                // clamped copies feed invalid pixels only
    const int col_r = cm + 1 < nch ? col_m + 8 : col_m + 8 - NB;
    const int band_lo = band * (RB * P), band_hi = min(band_lo + RB, p.cy);  // P > 1: of sub-band 0
    const int row_lo = max(band_lo, H), row_hi = min(band_hi, p.cy - H - 1);   // rows that may hold a maximum
    const int rs0 = band_lo - H - 1;
    // P > 1: the same limits for this lane's sub-band, in sub-band 0's row numbering
    const int lo_rel = max(band_lo + sub_rows, H) - sub_rows;
    const int hi_rel = min(min(band_lo + sub_rows + RB, p.cy), p.cy - H - 1) - sub_rows;

    // which of the lane's 8 pixels may hold a maximum: even pixels -> bits 0..3, odd -> bits 16..19,
    // replicated for the four row slots of the candidate accumulator
    u32 colmask = 0;
    if (lane_valid) {
#pragma unroll
        for (int b = 0; b < 8; b++) {
            int j = c8 * 8 + b - xoff;
            if (j >= H && j < p.cx - H - 1) colmask |= 1u << ((b >> 1) + 16 * (b & 1));
        }
        colmask *= 0x1111u;
    }

    // per-lane byte offsets inside a row (32-bit) + a wave-uniform row base: the loads use
    // SGPR-base + VGPR-offset addressing, no 64-bit vector address arithmetic per row
    // Neighbour pixels come from the adjacent lanes' registers (DPP), not from memory: overlapping
    // 8-byte loads next to the 16-byte ones doubled the HBM-side traffic (requests to a line whose
    // fill is still in flight are not merged).  Only lanes 0 and 63 load the 4 pixels beyond the wave.
    const unsigned off_m = (unsigned)col_m * 2u;
    const unsigned off_e = (unsigned)(lane == 0 ? col_l : col_r) * 2u;
    const bool edge_lane = P == 1 && (lane == 0 || lane == 63);       // a sub-band's row lies wholly inside its lanes
    // Rows are fetched with buffer loads: the frame base sits in a scalar resource descriptor, the
    // row offset in a scalar register and the lane's column offset in one VGPR, so a row costs no
    // vector address arithmetic at all (a 64-bit global address per lane would take two VALU adds).
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    // num_records bounds the descriptor at the end of the frames of this call: when the width is not a
    // multiple of 8 the last chunk of a row reads into the next row (masked columns), and on the very last
    // row it would read past the movie — the buffer unit returns 0 there instead
    const long long remaining = ((long long)(p.nframes - fi) * p.Y * p.X - ((long long)p.y0 * p.X + p.x0 - xoff)) * 2;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t *>(src - xoff), 0, (int)(remaining < 0x7fffffffLL ? remaining : 0x7fffffffLL),
        0x00020000 /* raw 32-bit buffer, gfx94x/gfx950 */);
    const unsigned pitch = (unsigned)p.X * 2u;         // Y * pitch < 2^31 on this path
    // interior bands never touch a row outside the crop: no clamping in their row loop
    const bool interior = rs0 >= 0 && rs0 + NRP + D <= p.cy;
    auto load_row = [&](int r) -> RowRegs {
        RowRegs o;
        unsigned soff = 0;
        u32x4_t m;
        if constexpr (P > 1) {
            // every sub-band clamps its own row: the row offset joins the lane's column offset
            const int rl = min(max(r + sub_rows, 0), p.cy - 1);
            m = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(off_m + (unsigned)rl * pitch), 0, 0);
        } else {
            const int rc = interior ? r : min(max(r, 0), p.cy - 1);
            soff = (unsigned)rc * pitch;
            m = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off_m, (int)soff, 0);
        }
        o.m = make_uint4(m.x, m.y, m.z, m.w);
        // only lanes 0 and 63 ever read `e` (as the DPP fill value): the other lanes leave it undefined
        // instead of spending two or four v_mov per row on zeros
        asm("" : "=v"(o.e.x), "=v"(o.e.y), "=v"(o.e.z), "=v"(o.e.w));
        if (edge_lane) {
            if constexpr (WIDE) {
                const u32x4_t e = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off_e, (int)soff, 0);
                o.e = make_uint4(e.x, e.y, e.z, e.w);
            } else {
                const u32x2_t e = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)off_e, (int)soff, 0);
                o.e.x = e.x; o.e.y = e.y;
            }
        }
        return o;
    };

    u32 Hring[HR][4], Uprev[4], Dv[H][4], Dpre[H][4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        Uprev[q] = 0xffffffffu;
#pragma unroll
        for (int t = 0; t < HR; t++) Hring[t][q] = 0xffffffffu;
#pragma unroll
        for (int t = 0; t < H; t++) { Dv[t][q] = 0u; Dpre[t][q] = 0xffffffffu; }
    }
    u32 mn = 0xffffffffu, mxL = 0u, mxR = 0u;
    u32 mprev = 0u, mpair1 = 0u, mpair2 = 0u;         // previous max group, pair maxima of groups (g-1, g-2) and (g-2, g-3)
    u32 lo1 = 0xffffu, lo2 = 0xffffu;                 // minima of the previous two min groups
    u32 acc = 0;                                      // "failed the test" bits of up to four rows
    int cnt = 0;                                      // candidates found so far (wave-uniform)
    const u32 lane8 = (u32)lane << 3;

    RowRegs pf[D];
#pragma unroll
    for (int d = 0; d < D; d++) pf[d] = load_row(rs0 + d);

    for (int sb = 0; sb < NRP; sb += U_) {
#pragma unroll
        for (int u = 0; u < U_; u++) {
            const int st = sb + u;                    // pipeline step; row r = rs0 + st
            const RowRegs cur = pf[u % D];
            pf[u % D] = load_row(rs0 + st + D);
            u32 A[NA], Bp[NA];
            if constexpr (WIDE) {
                A[0] = from_lane_below(cur.m.x, cur.e.x); A[1] = from_lane_below(cur.m.y, cur.e.y);
                A[2] = from_lane_below(cur.m.z, cur.e.z); A[3] = from_lane_below(cur.m.w, cur.e.w);
                A[8] = from_lane_above(cur.m.x, cur.e.x); A[9] = from_lane_above(cur.m.y, cur.e.y);
                A[10] = from_lane_above(cur.m.z, cur.e.z); A[11] = from_lane_above(cur.m.w, cur.e.w);
            } else {
                A[0] = from_lane_below(cur.m.z, cur.e.x); A[1] = from_lane_below(cur.m.w, cur.e.y);
                A[6] = from_lane_above(cur.m.x, cur.e.x); A[7] = from_lane_above(cur.m.y, cur.e.y);
            }
            A[OWN] = cur.m.x; A[OWN + 1] = cur.m.y; A[OWN + 2] = cur.m.z; A[OWN + 3] = cur.m.w;
            Bp[0] = 0;
#pragma unroll
            for (int k = 1; k < NA; k++) Bp[k] = __builtin_amdgcn_alignbit(A[k], A[k - 1], 16);
            mn = pk_min(pk_min(mn, pk_min(A[OWN], A[OWN + 1])), pk_min(A[OWN + 2], A[OWN + 3]));
            mxL = pk_max(mxL, pk_max(A[OWN], A[OWN + 1]));
            mxR = pk_max(mxR, pk_max(A[OWN + 2], A[OWN + 3]));

            u32 L[4], R[4];
            L[0] = LR<H, 0, 0, NA>::left(A, Bp); R[0] = LR<H, 0, 0, NA>::right(A, Bp);
            L[1] = LR<H, 1, 0, NA>::left(A, Bp); R[1] = LR<H, 1, 0, NA>::right(A, Bp);
            L[2] = LR<H, 2, 0, NA>::left(A, Bp); R[2] = LR<H, 2, 0, NA>::right(A, Bp);
            L[3] = LR<H, 3, 0, NA>::left(A, Bp); R[3] = LR<H, 3, 0, NA>::right(A, Bp);
            u32 tq[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const u32 v = A[q + OWN];
                const u32 hrow = pk_max(pk_max(L[q], v), R[q]);
                u32 Ucur = hrow;
                if (H > 1) {
#pragma unroll
                    for (int t = 0; t < HR; t++)
                        if (!WIDE || t != u % HR) Ucur = pk_max(Ucur, Hring[t][q]);     // WIDE: slot u % H holds the row H steps back
                }
                const u32 bef = pk_max(Uprev[q], L[q]);
                const u32 pre = pk_max(pk_add_sat(bef, 0x00010001u), R[q]);
                // decision for the row H steps back (same ring slot): 0 in a half = passed
                const u32 thr = pk_max(Dpre[u % H][q], Ucur);
                tq[q] = pk_min(pk_sub_sat(thr, Dv[u % H][q]), 0x00010001u);
                Dv[u % H][q] = v;
                Dpre[u % H][q] = pre;
                if (H > 1) Hring[u % HR][q] = hrow;
                Uprev[q] = Ucur;
            }
            const int t4 = u % 4;                      // row slot inside the accumulator
            acc |= (tq[0] | (tq[1] << 1) | (tq[2] << 2) | (tq[3] << 3)) << (4 * t4);
            if (t4 == 3 || u == U_ - 1) {              // flush the candidates of the last t4+1 rows
                const int rd0 = rs0 + (st - t4) - H;   // decision row of slot 0
                u32 rowmask = 0;
#pragma unroll
                for (int tt = 0; tt <= t4; tt++) {
                    if constexpr (P > 1) { if (rd0 + tt >= lo_rel && rd0 + tt < hi_rel) rowmask |= 0x000f000fu << (4 * tt); }
                    else { if (rd0 + tt >= row_lo && rd0 + tt < row_hi) rowmask |= 0x000f000fu << (4 * tt); }
                }
                u32 pass = ~acc & rowmask & colmask;
                acc = 0;
                // Append to the wave's own list: every round each lane that still has a candidate emits
                // its lowest one, slots come from a ballot prefix count and the list length stays in a
                // scalar register.  (An LDS atomicAdd per lane is turned into a serial per-lane scan by
                // the compiler's atomic optimizer: ~9 SALU instructions per active lane, per flush.)
                const u32 ebase = (u32)((rd0 - band_lo) << 9) + lane8;
                for (;;) {
                    const bool has = pass != 0;
                    const unsigned long long bal = __ballot(has);
                    if (bal == 0) break;
                    if (has) {
                        const u32 b = (u32)__ffs(pass) - 1u;
                        pass &= pass - 1;
                        // bit b: row slot (b >> 2) & 3, pixel 2 * (b & 3) + (b >> 4)
                        const u32 e = ebase + (((b >> 2) & 3u) << 9) + ((b & 3u) << 1) + (b >> 4);
                        const int slot = cnt + (int)__builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
                        if (slot < LIST) s_list[w][slot] = (unsigned short)e;
                    }
                    cnt += __popcll(bal);
                }
            }
            if ((u + 1) % GM == 0) {                   // close a maximum group: (right cell << 16) | left cell
                const u32 lo2_ = __builtin_amdgcn_perm(mxR, mxL, 0x05040100u), hi2_ = __builtin_amdgcn_perm(mxR, mxL, 0x07060302u);
                const u32 m0 = pk_max(lo2_, hi2_);
                const u32 pair = pk_max(m0, mprev);                   // groups g, g-1
                s_max[w][sb / GM + (u + 1) / GM - 1][lane] = pk_max(pair, mpair2);   // + groups g-2, g-3
                mpair2 = mpair1; mpair1 = pair;
                mprev = m0;
                mxL = 0u; mxR = 0u;
            }
        }
        if (((sb + U_) % GS) == 0) {                   // close a minimum group (window of three groups)
            const u32 lo0 = min(mn & 0xffffu, mn >> 16);
            s_min[w][(sb + U_) / GS - 1][lane] = (unsigned short)min(lo0, min(lo1, lo2));
            lo2 = lo1; lo1 = lo0;
            mn = 0xffffffffu;
        }
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();

    const float *sux = s_u, *suy = s_u + BOX * BOX;
    // Results are buffered in registers and appended KBUF at a time: ONE slot-allocating atomic per
    // flush per wave, on the counter of this block's shard (= XCD).  A single hot counter costs
    // ~11 ns per atomic and was 40 % of the kernel.
    constexpr int KBUF = 4;
    int buf_i[KBUF], buf_j[KBUF], nbuf = 0;
    float buf_ng[KBUF];
    const int shard = blockIdx.x & 7;
    auto flush = [&]() {
        if (p.dbg & 2) { nbuf = 0; return; }
        unsigned long long bal[KBUF];
        int total = 0;
#pragma unroll
        for (int k = 0; k < KBUF; k++) { bal[k] = __ballot(k < nbuf); total += __popcll(bal[k]); }
        if (total) {
            unsigned long long basepos = 0;
            if (lane == 0) { basepos = atomicAdd(&shard_cnt[shard], (unsigned long long)total); atomicAdd(&frame_count[fi], total); }
            basepos = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(basepos >> 32)) << 32) |
                      (unsigned)__builtin_amdgcn_readfirstlane((int)(basepos & 0xffffffffu));
            int off = 0;
#pragma unroll
            for (int k = 0; k < KBUF; k++) {
                if (k < nbuf) {
                    const long long pos = (long long)basepos + off + __popcll(bal[k] & ((1ull << lane) - 1ull));
                    if (pos < cap) {
                        Record rec;
                        rec.frame = (int32_t)(p.f_lo + fi + p.label_off);
                        rec.y = buf_i[k] + p.y0;
                        rec.x = buf_j[k] + p.x0;
                        rec.ng = buf_ng[k];
                        recs[(long long)shard * cap + pos] = rec;
                    }
                }
                off += __popcll(bal[k]);
            }
        }
        nbuf = 0;
    };
    auto append = [&](int i, int j, float ng) {
        if ((double)ng > p.min_ng) {                   // localize.py:288
#pragma unroll
            for (int k = 0; k < KBUF; k++) if (k == nbuf) { buf_i[k] = i; buf_j[k] = j; buf_ng[k] = ng; }
            nbuf++;
        }
    };
    // slow exact path: wrapped stencils, saturated pixels, overflow rescans
    auto process_slow = [&](int i, int j, bool recheck) {
        const float v = (float)src[(int64_t)i * p.X + j];
        if (recheck) {
#pragma unroll 1
            for (int k = -H; k <= H; k++)
#pragma unroll 1
                for (int l = -H; l <= H; l++) {
                    float o = (float)src[(int64_t)(i + k) * p.X + (j + l)];
                    if ((k < 0 || (k == 0 && l < 0)) ? !(v > o) : !(v >= o)) return;
                }
        }
        float ng = 0.0f;
#pragma unroll 1
        for (int k = 0; k < BOX; k++) {
            const int rk = i - H + k;
            const int rm = rk - 1 < 0 ? rk - 1 + p.cy : rk - 1;          // numba negative-index wrap
            const uint16_t *rowm = src + (int64_t)rm * p.X;
            const uint16_t *row0 = src + (int64_t)rk * p.X;
            const uint16_t *rowp = src + (int64_t)(rk + 1) * p.X;
#pragma unroll 1
            for (int l = 0; l < BOX; l++) {
                if (k == H && l == H) continue;
                const int cl = j - H + l;
                const int clm = cl - 1 < 0 ? cl - 1 + p.cx : cl - 1;
                float gy = sub_rn((float)rowp[cl], (float)rowm[cl]);
                float gx = sub_rn((float)row0[cl + 1], (float)row0[clm]);
                float sacc = add_rn(mul_rn(gy, suy[k * BOX + l]), mul_rn(gx, sux[k * BOX + l]));
                ng = add_rn(ng, sacc);
            }
        }
        append(i, j, ng);
    };

    // list entry -> row, crop column, and the column inside the sub-band's (or the wave's) own lanes
    auto decode = [&](unsigned e, int &i, int &j, int &jown) {
        if constexpr (P > 1) {
            const int le = (int)(e & 511u) >> 3;
            i = band_lo + (le / NL) * RB + (int)(e >> 9);
            jown = (le % NL) * 8 + (int)(e & 7u);
            j = jown - xoff;
        } else {
            i = band_lo + (int)(e >> 9);
            jown = (int)(e & 511u);
            j = seg * 512 + jown - xoff;
        }
    };
    const int found = cnt;
    if (found <= LIST) {
        // pass 1: cheap level-1 bound on every candidate; survivors are compacted in place
        // (ballot + prefix count) so that the exact evaluation runs with full lanes
        int kept = 0;
        for (int q0 = 0; q0 < found; q0 += 64) {
            const int q = q0 + lane;
            bool keep = false;
            unsigned short e = 0;
            if (q < found) {
                e = s_list[w][q];
                int i, j, jown;
                decode(e, i, j, jown);
                keep = true;
                const int jl = (int)(e & 511u);          // column in the wave's lanes: indexes the statistics cells
                // |ng| <= P_box * (max - min) over statistics cells covering the (2H+3)^2 stencil; candidates whose
                // stencil wraps or leaves this wave's 512 columns (its sub-band's lanes) are always kept
                if (i != H && j != H && jown - H - 1 >= 0 && jown + H + 1 <= 8 * NL - 1) {
                    const int rr = (int)(e >> 9) + H + 1;       // = row - first pipeline row of its (sub-)band
                    const int jlo = jl - H - 1, jhi = jl + H + 1;
                    constexpr int NLANE = (2 * H + 2) / 8 + 2;               // lanes (8 columns each) a stencil row can touch
                    unsigned lo = 0xffffu, hi = 0u;
                    const unsigned short *mrow = s_min[w][(rr + H + 1) / GS];
                    const unsigned *xrow = s_max[w][(rr + H + 1) / GM];
                    const int la = jlo >> 3, lb = jhi >> 3, c0 = jlo >> 2, c1 = jhi >> 2;
#pragma unroll
                    for (int t = 0; t < NLANE; t++) {
                        const int l = min(la + t, lb);
                        lo = min(lo, (unsigned)mrow[l]);
                        const unsigned sv = xrow[l];
                        if (2 * l >= c0) hi = max(hi, sv & 0xffffu);         // left cell of lane l is cell 2l
                        if (2 * l + 1 <= c1) hi = max(hi, sv >> 16);
                    }
                    if ((double)(hi - lo) * p.bound_c < p.min_ng) keep = false;
                }
            }
            const unsigned long long bal = __ballot(keep);
            const int pos = kept + __popcll(bal & ((1ull << lane) - 1ull));
            __builtin_amdgcn_wave_barrier();
            if (keep) s_list[w][pos] = e;     // pos <= q: never overwrites an unread entry of a later chunk
            kept += __popcll(bal);
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        // pass 2: exact float32 net gradient
        int rounds = 0;
        for (int q0 = 0; q0 < kept; q0 += 64) {
            const int q = q0 + lane;
            if (q < kept) {
                const unsigned e = s_list[w][q];
                int i, j, jown;
                decode(e, i, j, jown);
                const bool wraps = (i == H) || (j == H);
                // the packed test cannot see ties at 65535 (saturating +1): those are rechecked exactly
                const bool saturated = src[(int64_t)i * p.X + j] == 0xffffu;
                if (p.dbg & 1) append(i, j, (e & 7) == 0 ? 1e9f : 0.0f);
                else if (!wraps && !saturated) append(i, j, exact_ng_noWrap<H>(src, p.X, i, j));
                else process_slow(i, j, saturated);
            }
            if (++rounds == KBUF) { flush(); rounds = 0; }
        }
        flush();
    } else {
        // More candidates than local maxima can exist: a saturated (65535) plateau flooded the
        // packed test.  Rescan this band pixel by pixel with the exact test (slow, rare).
        const int j0 = seg * 512 - xoff, j1 = min(j0 + 512, p.cx);
        for (int idx0 = 0; idx0 < RB * 512; idx0 += 64) {
            const int idx = idx0 + lane;
            if constexpr (P > 1) {
                int i, j, jown;
                decode((unsigned)idx, i, j, jown);       // same layout as a list entry: (row << 9) | column in the wave's lanes
                const int sb_lo = band_lo + (((idx & 511) >> 3) / NL) * RB;
                const int rlo = max(sb_lo, H), rhi = min(min(sb_lo + RB, p.cy), p.cy - H - 1);
                if (i >= rlo && i < rhi && jown < nch * 8 && j >= H && j < p.cx - H - 1) process_slow(i, j, true);
            } else {
                const int i = band_lo + idx / 512, j = j0 + (idx & 511);
                if (i >= row_lo && i < row_hi && j < j1 && j >= H && j < p.cx - H - 1) process_slow(i, j, true);
            }
            flush();
        }
    }
}

template <int H, int RB, int D, int P = 1>
static int launch_fast(const FastParams &p, const float *d_tab, Record *recs, long long cap,
                       unsigned long long *shard_cnt, int *frame_count, hipStream_t s)
{
    long long blocks = 8LL * p.bpf * ((p.nframes + 7) / 8);
    if (blocks > 0x7fffffffLL) { set_error("identify: too many blocks (%lld)", blocks); return PMI_ERR_ARG; }
    hipLaunchKernelGGL((identify_scan_u16_fast_kernel<H, RB, D, P>), dim3((unsigned)blocks), dim3(FAST_WAVES * 64), 0, s,
                       p, d_tab, recs, cap, shard_cnt, frame_count);
    PMI_HIP(hipGetLastError());
    return PMI_OK;
}

// The compile-time unit vectors against the runtime float32 sqrtf/divide the generic path uploads.
template <int H> static bool unit_vectors_match_h()
{
    for (int k = 0; k <= 2 * H; k++)
        for (int l = 0; l <= 2 * H; l++) {
            if (k == H && l == H) continue;
            volatile float vx = (float)(H - l), vy = (float)(H - k);
            volatile float n2 = vx * vx + vy * vy;
            volatile float n = sqrtf(n2);
            volatile float ux = vx / n, uy = vy / n;
            if (ux != unit_x<H>(k, l) || uy != unit_y<H>(k, l)) return false;
        }
    return true;
}
static bool unit_vectors_match()
{
    static const bool ok = unit_vectors_match_h<1>() && unit_vectors_match_h<2>() && unit_vectors_match_h<3>() && unit_vectors_match_h<4>() &&
                           unit_vectors_match_h<5>() && unit_vectors_match_h<6>() && unit_vectors_match_h<7>() &&
                           unit_vectors_match_h<8>();
    return ok;
}

// Returns PMI_OK and sets *handled when the fast path applies.
int launch_scan_u16_fast(const void *d_movie, int64_t Y, int64_t X, int y0, int x0, int cy, int cx, int64_t f_lo,
                         int64_t label_off, int nframes, int box, double min_ng, const float *d_tab, Record *recs,
                         long long cap, unsigned long long *n_total, int *frame_count, hipStream_t s, bool *handled)
{
    *handled = false;
    static const bool force_generic = getenv("PMI_IDENTIFY_GENERIC") != nullptr;
    if (force_generic) return PMI_OK;
    const int h = box / 2;
    if (h < 1 || h > 8) return PMI_OK;
    if (!unit_vectors_match()) return PMI_OK;          // never expected; the generic kernel uses the runtime table
    if ((X & 1) || cx < 16 || ((uintptr_t)d_movie & 3)) return PMI_OK;       // rows must start 4-byte aligned (buffer loads); the crop may not
    if (cy > 65535 || cx > 65535 || X > 65535 || Y * X * 2 >= (1LL << 31)) return PMI_OK;   // 32-bit row offsets
    int RB = h == 1 ? 16 : (h == 2 ? 32 : 64);           // keeps the per-wave candidate list <= 8.5 KB of LDS
    // narrow frames (box 7): several bands side by side in one wavefront instead of idle lanes
    const int nch = ((x0 & 7) + cx + 7) / 8;
    static const bool no_pack = getenv("PMI_IDENTIFY_NOPACK") != nullptr;
    int pack = 1;
    if (h >= 2 && h <= 6 && !no_pack) {
        if (nch <= 8 && h == 3) { pack = 8; RB = cy >= 256 ? 32 : (cy >= 128 ? 16 : 8); }   // <= 64 px wide: eight bands side by side
        else if (nch <= 16 && h <= 4) { pack = 4; RB = (h > 2 && cy >= 256) ? 64 : 32; }     // short frames: shorter bands, no idle sub-band
        else if (nch <= 32) pack = 2;
    }
    FastParams p;
    p.movie = (const uint16_t *)d_movie; p.Y = Y; p.X = X; p.y0 = y0; p.x0 = x0; p.cy = cy; p.cx = cx;
    p.f_lo = f_lo; p.label_off = label_off; p.nframes = nframes; p.box = box; p.min_ng = min_ng;
    p.bands = (cy + RB * pack - 1) / (RB * pack);
    p.segs = pack > 1 ? 1 : (nch + 63) / 64;
    p.bpf = (p.bands * p.segs + FAST_WAVES - 1) / FAST_WAVES;
    // ng is a linear functional sum_p w(p) f(p) of the (2H+3)^2 neighbourhood with sum_p w(p) = 0 (a constant
    // image has no gradient), hence |ng| <= P_box * (max - min), P_box = sum of the positive weights
    // (35.06 for box 7; the cruder sum of |ux| + |uy| is 60.9).  Double precision, +0.1 % margin for the
    // float32 rounding of the reference's own summation.
    double c = 0.0;
    {
        const int n = 2 * h + 3;
        std::vector<double> wgt((size_t)n * n, 0.0);
        for (int k = 0; k <= 2 * h; k++)
            for (int l = 0; l <= 2 * h; l++) {
                if (k == h && l == h) continue;
                const double vx = h - l, vy = h - k, r = std::sqrt(vx * vx + vy * vy);
                wgt[(size_t)(k + 2) * n + (l + 1)] += vy / r; wgt[(size_t)k * n + (l + 1)] -= vy / r;
                wgt[(size_t)(k + 1) * n + (l + 2)] += vx / r; wgt[(size_t)(k + 1) * n + l] -= vx / r;
            }
        for (double v : wgt) if (v > 0) c += v;
    }
    p.bound_c = c * 1.001;
    static const int dbg = getenv("PMI_IDENTIFY_DBG") ? atoi(getenv("PMI_IDENTIFY_DBG")) : 0;
    p.dbg = dbg;
    int rc;
    switch (h) {
    case 1: rc = launch_fast<1, 16, 2>(p, d_tab, recs, cap, n_total, frame_count, s); break;
    case 2:
        if (pack == 4) rc = launch_fast<2, 32, 2, 4>(p, d_tab, recs, cap, n_total, frame_count, s);
        else if (pack == 2) rc = launch_fast<2, 32, 2, 2>(p, d_tab, recs, cap, n_total, frame_count, s);
        else rc = launch_fast<2, 32, FAST_D_H2>(p, d_tab, recs, cap, n_total, frame_count, s);
        break;
    case 3:
        if (pack == 8 && RB == 32) rc = launch_fast<3, 32, FAST_D_H3, 8>(p, d_tab, recs, cap, n_total, frame_count, s);
        else if (pack == 8 && RB == 16) rc = launch_fast<3, 16, FAST_D_H3, 8>(p, d_tab, recs, cap, n_total, frame_count, s);
        else if (pack == 8) rc = launch_fast<3, 8, FAST_D_H3, 8>(p, d_tab, recs, cap, n_total, frame_count, s);
        else if (pack == 4 && RB == 64) rc = launch_fast<3, 64, FAST_D_H3, 4>(p, d_tab, recs, cap, n_total, frame_count, s);
        else if (pack == 4) rc = launch_fast<3, 32, FAST_D_H3, 4>(p, d_tab, recs, cap, n_total, frame_count, s);
        else if (pack == 2) rc = launch_fast<3, 64, FAST_D_H3, 2>(p, d_tab, recs, cap, n_total, frame_count, s);
        else rc = launch_fast<3, 64, FAST_D_H3>(p, d_tab, recs, cap, n_total, frame_count, s);
        break;
    case 4:
        if (pack == 4 && RB == 64) rc = launch_fast<4, 64, 4, 4>(p, d_tab, recs, cap, n_total, frame_count, s);
        else if (pack == 4) rc = launch_fast<4, 32, 4, 4>(p, d_tab, recs, cap, n_total, frame_count, s);
        else if (pack == 2) rc = launch_fast<4, 64, 4, 2>(p, d_tab, recs, cap, n_total, frame_count, s);
        else rc = launch_fast<4, 64, FAST_D_H4>(p, d_tab, recs, cap, n_total, frame_count, s);
        break;
    case 5:
        if (pack == 2) rc = launch_fast<5, 64, 2, 2>(p, d_tab, recs, cap, n_total, frame_count, s);
        else rc = launch_fast<5, 64, FAST_D_H5>(p, d_tab, recs, cap, n_total, frame_count, s);
        break;
    case 6:
        if (pack == 2) rc = launch_fast<6, 64, 3, 2>(p, d_tab, recs, cap, n_total, frame_count, s);
        else rc = launch_fast<6, 64, FAST_D_H6>(p, d_tab, recs, cap, n_total, frame_count, s);
        break;
    case 7: rc = launch_fast<7, 64, 2>(p, d_tab, recs, cap, n_total, frame_count, s); break;
    default: rc = launch_fast<8, 64, 2>(p, d_tab, recs, cap, n_total, frame_count, s); break;
    }
    if (rc == PMI_OK) *handled = true;
    return rc;
}

}  // namespace pmi
