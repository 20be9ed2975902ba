// identify_fast.hip — the 16-bit fast path of identify (uint16 / int16 / uint8 movies): a register-pipelined,
// packed-u16 first-argmax scan streaming rows straight from HBM (no LDS staging).
//
// Same semantics as identify_scan_kernel in identify.hip (picasso/localize.py:97-134
// _local_maxima, :202-244 _net_gradient, :288 threshold); only the schedule differs:
//
//   * persistent waves (one per workgroup, a few per CU): a wave takes UNITS of up to 256 rows x 512 columns of one
//     frame, `upw` of them in sequence (frames at most 256 / 128 / 64 pixels wide: 2 / 4 / 8 units side by side,
//     template parameter P); lane l holds 8 consecutive pixels of the current row as four packed u16x2 registers
//     (one 16-byte buffer load per row, D rows in flight ahead of the pipeline) plus the 4 pixels on either side,
//     taken from the neighbouring lanes' registers by DPP wave_shr/wave_shl (lanes 0 and 63 load theirs with one
//     masked 8-byte load);
//   * horizontal: Hrow = maximum of the 2h+1 pixels around each pixel, as v_pk_max_u16 on aligned / odd
//     pixel pairs (odd pairs by v_alignbit), window by doubling;
//   * vertical: U[r] = max(Hrow[r-h..r]) from a register ring; W[r'] = max(U[r'+h], U[r']) is the maximum of the
//     (2h+1)^2 window of row r', known h rows after it streamed in.  A pixel is a CANDIDATE iff it equals W and
//     reaches the floor (below): sat(max(W, floor) - v) == 0;
//   * the floor rejects maxima that cannot reach min_ng whatever their neighbourhood (the shot-noise maxima,
//     2 % of all pixels), see the kernel;
//   * candidates go to a per-wave ring in LDS; the exact float32 net gradient is evaluated for 32..64 of them
//     at a time, in the reference's (k,l) order, and the same neighbourhood decides whether the candidate is the
//     FIRST maximum of its window (np.argmax: strictly greater than the window pixels before it).
//
// HBM traffic: every pixel is fetched once; the 2h+2 halo rows of a unit were fetched by the same wave a few
// steps earlier (L2); the neighbourhoods of the candidates are fetched again (11 rows x 64 B each).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "ng_common.h"
#include "pmi_common.h"

#pragma clang fp contract(off)

namespace pmi {

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

// Pixel types of the packed scan: the window test and the floor work on unsigned 16-bit pairs.
//   PT_U16  as stored;
//   PT_U8   eight pixels = one 8-byte load per lane-row, widened to four u16 pairs with v_perm_b32;
//   PT_I16  as stored with the sign bit flipped (x ^ 0x8000 = x + 32768): order-preserving, and the net gradient is
//           a sum of DIFFERENCES of pixels (picasso/localize.py:233-243), which a common offset does not change —
//           float32(a + 32768) - float32(b + 32768) = float32(a) - float32(b) exactly for 16-bit values.
// (float32 / 32-bit integer movies compare as float32 in the reference, picasso/localize.py:332: generic kernel.)
//   PT_KEY  float32 movies with any content: the scan works on 16-bit KEYS — the upper half of the order-preserving integer
//           image of a float32 (key_of_float below: monotone, so every first maximum of the pixels is a maximum of the keys) —
//           built in registers from the float32 rows as they are loaded (round 5: two 16-byte loads per lane and row, three
//           integer operations per pixel, one v_perm per pair; round 4 scanned a 16-bit copy made by a kernel of its own:
//           8 bytes moved per pixel instead of 4); everything that is decided exactly — the first-argmax rule, the
//           net gradient, the threshold — reads the float32 pixels.  Equal keys do not mean equal pixels, so the neighbour
//           rule that drops the right-hand / lower one of two equal candidates does not apply; the floor is computed from
//           the lower edges of the keys' buckets and turned back into a key (conservative on both sides).
//   PT_KEY_I32 / PT_KEY_U32  32-bit integer movies (round 6): the reference casts every frame to float32 before it looks at it
//           (picasso/localize.py:332), so these are PT_KEY with one conversion per pixel in front — v_cvt_f32_i32 / _u32 as
//           the row is loaded and wherever a pixel is read for an exact decision.  (They used to go through a uint16 copy when
//           they held 16-bit counts — 2.4 TB/s — and through the generic LDS kernel otherwise — 0.4 TB/s.)
enum { PT_U16 = 0, PT_U8 = 1, PT_I16 = 2, PT_KEY = 3, PT_KEY_I32 = 4, PT_KEY_U32 = 5 };
constexpr bool pt_is_key(int PT) { return PT >= PT_KEY; }
// the float32 the reference sees for a 4-byte pixel of a key scan
template <int PT> __device__ __forceinline__ float wide_px(float raw)
{
    if constexpr (PT == PT_KEY_I32) return (float)(int32_t)__float_as_uint(raw);
    else if constexpr (PT == PT_KEY_U32) return (float)__float_as_uint(raw);
    else return raw;
}
template <int PT> __device__ __forceinline__ uint32_t wide_bits(uint32_t raw) { return __float_as_uint(wide_px<PT>(__uint_as_float(raw))); }
__host__ __device__ __forceinline__ uint32_t key_image(uint32_t bits) { return (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u); }
__device__ __forceinline__ unsigned key_of_float(float f) { return key_image(__float_as_uint(f)) >> 16; }
// a lower bound of every float32 whose key is k: the smallest of them — or, where that pattern is a NaN (the buckets that hold
// -inf / +inf also hold NaNs), -inf on the negative side and +inf on the positive one (only NaNs lie above +inf)
__device__ __forceinline__ float key_lower_edge(unsigned k)
{
    const unsigned u = k << 16;
    const float f = __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
    return f == f ? f : ((u & 0x80000000u) ? INFINITY : -INFINITY);
}
template <int PT> struct Px { typedef uint16_t T; };
template <> struct Px<PT_U8> { typedef uint8_t T; };
template <> struct Px<PT_KEY> { typedef float T; };
template <> struct Px<PT_KEY_I32> { typedef float T; };      // (4-byte pixels addressed as float32; wide_px makes the value)
template <> struct Px<PT_KEY_U32> { typedef float T; };
// the keys of two float32 pixels as one packed pair (low half: the first)
__device__ __forceinline__ uint32_t key_pair(uint32_t a, uint32_t b)
{
    const uint32_t ka = a ^ ((uint32_t)((int32_t)a >> 31) | 0x80000000u), kb = b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
    return __builtin_amdgcn_perm(kb, ka, 0x07060302u);
}
template <int PT> __device__ __forceinline__ float px_float(typename Px<PT>::T raw)
{
    if constexpr (PT == PT_I16) return (float)(int16_t)raw;
    else return (float)raw;
}

struct FastParams {
    const void *movie;
    int64_t Y, X;
    int y0, x0, cy, cx;
    int64_t f_lo, label_off;
    int nframes;
    int segs;                  // 512-column segments per row (1 when P > 1)
    int rbu;                   // rows per unit (per sub-band when P > 1)
    int upf;                   // units per frame: ceil(cy / (rbu * P)) * segs
    long long units;           // nframes * upf
    int pf;                    // P > 1: the P lane sets hold the same rows of P consecutive FRAMES (short frames: no halo inside a frame), not P row ranges of one
    int upw;                   // consecutive units per wavefront
    int lin_rows;              // > 0 (one row range per lane set, P == 1): a wave takes lin_rows consecutive rows of the
    long long lin_total;       //   sequence (segment, frame, row) of lin_total rows, cut into units only at frame boundaries
    int box;
    double min_ng;
    float filt_alpha, filt_t;  // floor filter: a maximum v can only reach min_ng if v > alpha * (minimum above it) + t; t < 0: off
    float filt_kr, filt_kc, filt_krc;   // the same for stencils that wrap through index -1 (row, column, both): v > k * cfloor + t
    int filt_slack;            // counts by which later pixels may undercut the minimum seen so far before a chunk is run again
    float filt_slackf;         // the same for the key scans, in the pixels' own unit (not clamped to 16 bits: a 32-bit integer movie's threshold may be 1e9)
    // pixel hand-off to the fit (uint16 movies): the exact stage has a candidate's (2H+3)^2 neighbourhood in registers and
    // leaves the box rows — BOX rows of H + 1 packed pairs (BOX + 1 pixels), 4 (H + 1) bytes per row — at pix[slot], so the
    // fit reads one or two cache lines per spot instead of BOX lines of the movie.  pix_cnt: one slot counter per shard,
    // pix_cap slots per shard; a round that finds its shard full hands nothing on (slot -1: the fit reads the movie).
    u32 *pix;
    unsigned *pix_cnt;
    unsigned pix_cap;
    int defer;                 // 1: a wave whose exact rounds accept most of its candidates may emit the rest undecided (net gradient NG_DEFERRED_BITS, pmi_common.h)
    const float *fmovie;       // PT_KEY: the float32 frames the keys in `movie` were made from (same frame indexing)
    int gate_want;             // the launch runs only while *gate equals this
    const int *gate;           // optional device flag: the launch does nothing unless it is 0 (32-bit movies narrowed to uint16, identify.hip)
    int dbg;                   // PMI_IDENTIFY_DBG: 1 = skip the exact net gradient, 2 = skip the record append, 4 = no floor filter (timing only)
};

#ifndef FAST_D_H3
#define FAST_D_H3 3          // rows prefetched ahead in the box-7 scan (divides its unroll period of 6).  6 / 3 / 2 -> 1.45 / 1.26 / 1.25 ms on one box: the twelve registers a deeper prefetch holds are worth more to the scheduler
#endif
#ifndef FAST_U_H3
#define FAST_U_H3 6         // unroll period of the box-7 row loop (a multiple of the ring period 3 and of FAST_D_H3)
#endif
#ifndef FAST_D_H2
#define FAST_D_H2 2
#endif
#ifndef FAST_D_H4
#define FAST_D_H4 4
#endif
#ifndef FAST_U_H4
#define FAST_U_H4 8
#endif
#ifndef FAST_U_H6
#define FAST_U_H6 12
#endif
#ifndef FAST_D_H5
#define FAST_D_H5 5
#endif
#ifndef FAST_D_H6
#define FAST_D_H6 6
#endif
// Candidates per in-loop round of exact net gradients: lanes busy (a round costs the same whatever it holds) against rows
// still near (a candidate's rows are re-read 100 ... 250 rows after they streamed with rounds of 64).  Round 3, on the
// kernel with the common row ring, alternating runs on one box: box 7 1.180 / 1.147 / 1.190 ms with rounds of 64 / 32 / 16
// (six runs each, 32 ahead in every one), boxes 5 and 9 alike at 64 and 32, box 13 8 % slower at 32 (its rounds cost
// three times box 7's).
#ifndef FAST_ROUND_SMALL
#define FAST_ROUND_SMALL 32
#endif
#ifndef FAST_ROUND_WIDE
#define FAST_ROUND_WIDE 64
#endif
constexpr int fast_round(int H) { return H <= 4 ? FAST_ROUND_SMALL : FAST_ROUND_WIDE; }
#ifndef FAST_MIN_WAVES
#define FAST_MIN_WAVES 4      // waves per SIMD the register allocator must leave room for
#endif
#ifndef FAST_WAVES_H4
#define FAST_WAVES_H4 4
#endif
#ifndef FAST_WAVES_H5
#define FAST_WAVES_H5 3
#endif
#ifndef FAST_WAVES_H6
#define FAST_WAVES_H6 3
#endif
// Waves per SIMD the register allocator must leave room for (and the number of persistent waves launched).  A value
// spilled INSIDE the row loop costs more than a wave: a scratch reload is a vector-memory load, and waiting for it
// drains the prefetched rows at every step (round 1, box 11: 1.45 -> 2.94 TB/s with one wave less and no spill).
// Round 3: with the rows in flight and the pixel history in one register ring (below) box 9 fits four waves (123 VGPRs),
// boxes 11 and 13 three (145 / 161) — except the variants for frames wider than a wave and for two row ranges side by
// side at box 13 (and the latter at box 11), which would spill at three and stay at two.
// PT_KEY (float32 / 32-bit integer pixels): a row in flight is 32 bytes per lane — eight registers until its keys are built —
// and the exact stage holds three float32 neighbourhood rows: at the register budgets of the 16-bit scans the compiler spilled
// 300 ... 650 registers from box 9 up (round 5: 9x9 325, 11x11 299, 13x13 387, 17x17 598).  One wave less per SIMD each.
#ifndef FAST_KEY_WAVES_H4
#define FAST_KEY_WAVES_H4 3
#endif
constexpr int fast_waves_per_simd(int H, int P, bool EDGE, int PT = 0)
{
    // (frames wider than the wave: the edge loads are 16 or 32 bytes more per row in flight — 246 spilled registers at box 7, 80 at box 9, 41-73 at box 5)
    if (PT >= 3 /* the key scans */ && EDGE && H >= 2) return H <= 3 ? 3 : 2;
    if (PT >= 3 && H >= 4) return H == 4 ? FAST_KEY_WAVES_H4 : 2;
    return H <= 3 ? FAST_MIN_WAVES : (H == 4 ? FAST_WAVES_H4 : (H == 5 ? (P == 1 ? FAST_WAVES_H5 : 2) : (H == 6 ? (P == 1 && !EDGE ? FAST_WAVES_H6 : 2) : 2)));
}
#ifndef FAST_RING_MARGIN
#define FAST_RING_MARGIN 128   // free entries of the candidate ring below which a chunk ends after the current flush group (= LIST - THRESH: where a ring that is drained only when nearly full is drained)
#endif
#ifndef FAST_UNI_MIN_H
#define FAST_UNI_MIN_H 3      // smallest half-width whose scan keeps rows in flight and pixel history in one ring (box 5: 4.6 -> 4.3 TB/s with it, its two rows in flight are issued too late in the step)
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

// maximum of the LEN packed pairs at offsets S .. S + LEN - 1: powers of two by doubling, other lengths as the largest
// power of two below them followed by the rest (13 = 8 + 4 + 1).  Every sub-window then starts an EVEN number of pixels
// after S, so a box touches only the aligned pairs (even H: the odd ones, and the v_alignbit that makes them, drop out)
// or only the odd ones; identical sub-windows of neighbouring pixel pairs are shared by common-subexpression
// elimination (box 13: 29 v_pk_max_u16 for the four windows of a row, 38 + 11 v_alignbit with overlapping halves).
template <int LEN, int S, int NA>
__device__ __forceinline__ u32 wmax(const u32 (&A)[NA], const u32 (&Bp)[NA])
{
    if constexpr (LEN == 1) return pair_at<S, NA>(A, Bp);
    else if constexpr ((LEN & (LEN - 1)) == 0) return pk_max(wmax<LEN / 2, S, NA>(A, Bp), wmax<LEN / 2, S + LEN / 2, NA>(A, Bp));
    else {
        constexpr int P2 = LEN > 16 ? 16 : (LEN > 8 ? 8 : (LEN > 4 ? 4 : 2));
        return pk_max(wmax<P2, S, NA>(A, Bp), wmax<LEN - P2, S + P2, NA>(A, Bp));
    }
}

// The four (2H+1)-pixel window maxima of a lane's row (pixel pairs q = 0..3, windows at pair offsets 2q-H .. 2q+H) for
// H >= 4: the offsets 6-H .. H are common to all four (the core), window q adds 6-2q offsets on the left and 2q on the
// right, and those are suffix / prefix maxima of pair maxima — 18 / 20 / 22 v_pk_max_u16 for boxes 9 / 11 / 13
// (21 / 26 / 29 by doubling each window).
template <int H, int NA>
__device__ __forceinline__ void window_maxima(const u32 (&A)[NA], const u32 (&Bp)[NA], u32 (&hrow)[4])
{
    const u32 l2 = pk_max(pair_at<4 - H, NA>(A, Bp), pair_at<5 - H, NA>(A, Bp));
    const u32 l1 = pk_max(pk_max(pair_at<2 - H, NA>(A, Bp), pair_at<3 - H, NA>(A, Bp)), l2);
    const u32 l0 = pk_max(pk_max(pair_at<0 - H, NA>(A, Bp), pair_at<1 - H, NA>(A, Bp)), l1);
    const u32 r1 = pk_max(pair_at<H + 1, NA>(A, Bp), pair_at<H + 2, NA>(A, Bp));
    const u32 r2 = pk_max(pk_max(pair_at<H + 3, NA>(A, Bp), pair_at<H + 4, NA>(A, Bp)), r1);
    const u32 r3 = pk_max(pk_max(pair_at<H + 5, NA>(A, Bp), pair_at<H + 6, NA>(A, Bp)), r2);
    const u32 core = wmax<2 * H - 5, 6 - H, NA>(A, Bp);
    hrow[0] = pk_max(core, l0);
    hrow[1] = pk_max(pk_max(core, l1), r1);
    hrow[2] = pk_max(pk_max(core, l2), r2);
    hrow[3] = pk_max(core, r3);
}

struct RowRegs { uint4 m; uint4 e; };   // 8 own pixels + (lanes 0 / 63 only) the 4 or 8 pixels beyond the wave's edge (e.z, e.w: boxes 11, 13)

// value of lane-1 / lane+1 across the whole wavefront; `edge` is returned where no such lane exists
__device__ __forceinline__ u32 from_lane_below(u32 v, u32 edge) { return (u32)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x138, 0xf, 0xf, false); }   // wave_shr:1
__device__ __forceinline__ u32 from_lane_above(u32 v, u32 edge) { return (u32)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x130, 0xf, 0xf, false); }   // wave_shl:1
// the same for the row pixels: ZF = nobody reads what lanes 0 / 63 receive (frames no wider than the wave), so the
// instruction may fill in zeros itself (bound_ctrl) — with a fill VALUE the destination has to be preset by a v_mov
// in front of every DPP move, eight per row for boxes 11 and 13
template <bool ZF> __device__ __forceinline__ u32 px_from_lane_below(u32 v, u32 edge)
{
    if constexpr (ZF) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);
    else return from_lane_below(v, edge);
}
template <bool ZF> __device__ __forceinline__ u32 px_from_lane_above(u32 v, u32 edge)
{
    if constexpr (ZF) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);
    else return from_lane_above(v, edge);
}
__device__ __forceinline__ u32 quad(const uint4 &v, int q) { return q == 0 ? v.x : (q == 1 ? v.y : (q == 2 ? v.z : v.w)); }

// Exact float32 net gradient of one candidate in the reference's (k, l) order (picasso/localize.py:202-244).
// The (2H+3)^2 neighbourhood is fetched as its first column (one 2-byte load per row) plus 2H + 2 pixels as packed
// pairs (one wide load per row), in two batches of rows whose loads are all in flight together: the rows have
// usually left L2 by now and a row-by-row loop pays the memory latency 2H+1 times, while holding all 2H+3 rows at
// once does not fit the register budget of four waves per SIMD.  The reference indexes row y - 1 and column x - 1
// without a bounds check: for a maximum in row H (column H) that is index -1, the LAST row (column) of the frame
// (numba wraps negative indices) — only the first row and the first column of the neighbourhood can wrap, and they
// are addressed separately here, so the same code serves every candidate.  The rows are only 2-byte aligned;
// gfx950 runs global loads in unaligned mode.
// `first` (in/out): the centre is still the FIRST maximum of its window after the rows seen so far — strictly greater
// than the window pixels before it in row-major order, not smaller than those after it (np.argmax, localize.py:128).
template <int H, int K0, int K1, int PT>
__device__ __forceinline__ float exact_ng_rows(const typename Px<PT>::T *__restrict__ base, const typename Px<PT>::T *__restrict__ row0w,
                                               int64_t X, int c0w, float ng, float &vc, bool &first_max, u32 *__restrict__ pixdst)
{
    typedef typename Px<PT>::T PX;
    // base = &src[i - H - 1][j - H]: neighbourhood row t, column 1;  row0w = &src[wrapped first row][j - H];
    // c0w = (wrapped first column) - (j - H): offset of the neighbourhood's column 0 from column 1 (-1 unless it wraps)
    constexpr int BOX = 2 * H + 1, W = 2 * H + 3, NP = (W - 1) / 2;   // W is odd: one single pixel + NP pairs
    constexpr int R0 = K0, NR = K1 - K0 + 2;                           // window rows K0..K1-1 need neighbourhood rows K0..K1+1
    struct __attribute__((packed, aligned(2))) Pairs { u32 v[NP]; };
    struct __attribute__((packed, aligned(1))) BytePairs { unsigned short v[NP]; };
    u32 pk[NR][NP];
    u32 first[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const PX *row = (R0 + r == 0) ? row0w : base + (int64_t)(R0 + r) * X;
        if constexpr (PT == PT_U8) {
            const BytePairs t = *reinterpret_cast<const BytePairs *>(row);
#pragma unroll
            for (int q = 0; q < NP; q++) pk[r][q] = ((u32)t.v[q] & 0xffu) | (((u32)t.v[q] & 0xff00u) << 8);
        } else {
            const Pairs t = *reinterpret_cast<const Pairs *>(row);
#pragma unroll
            for (int q = 0; q < NP; q++) pk[r][q] = PT == PT_I16 ? t.v[q] ^ 0x80008000u : t.v[q];
        }
        first[r] = PT == PT_I16 ? (u32)row[c0w] ^ 0x8000u : (u32)row[c0w];
    }
    if constexpr (PT == PT_U16) {
        // hand the box rows on: box row k = neighbourhood row k + 1, its pixels = the NP pairs (columns 1 .. BOX + 1).
        // The first batch holds box rows 0 .. K1, the second K0 + 1 .. BOX - 1.
        if (pixdst) {
            constexpr int KLO = K0 == 0 ? 0 : K0 + 1, KHI = K0 == 0 ? K1 : BOX - 1;
#pragma unroll
            for (int k = KLO; k <= KHI; k++) {
#pragma unroll
                for (int q = 0; q < NP; q++) pixdst[k * NP + q] = pk[k + 1 - R0][q];
            }
        }
    }
    auto px = [&](int r, int b) -> float {
        r -= R0;
        if (b == 0) return (float)first[r];
        return (float)(((b - 1) & 1) ? (pk[r][(b - 1) >> 1] >> 16) : (pk[r][(b - 1) >> 1] & 0xffffu));
    };
    if (K0 == 0) vc = px(H + 1, H + 1);                            // the centre row belongs to the first batch
#pragma unroll
    for (int k = K0; k < K1; k++) {
#pragma unroll
        for (int l = 0; l < BOX; l++) {
            if (k == H && l == H) continue;                        // the centre is skipped (its unit vector is 0/0)
            const float o = px(k + 1, l + 1);
            first_max = first_max && ((k < H || (k == H && l < H)) ? vc > o : vc >= o);
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

template <int H, int PT>
__device__ __forceinline__ float exact_ng(const typename Px<PT>::T *__restrict__ src, int64_t X, int cy, int cx, int i, int j, bool &first_max,
                                          u32 *__restrict__ pixdst)
{
    typedef typename Px<PT>::T PX;
    constexpr int BOX = 2 * H + 1, KM = (BOX + 1) / 2;
    const PX *base = src + (int64_t)(i - H - 1) * X + (j - H);
    const int r0 = i - H - 1 < 0 ? i - H - 1 + cy : i - H - 1;          // numba negative-index wrap
    const PX *row0w = src + (int64_t)r0 * X + (j - H);
    int c0w = j - H - 1 < 0 ? cx - 1 - (j - H) : -1;
    float vc = 0.0f;
    first_max = true;
    float ng = exact_ng_rows<H, 0, KM, PT>(base, row0w, X, c0w, 0.0f, vc, first_max, pixdst);
    // the second batch starts only when the first sum is done (keeps its loads from being hoisted
    // above the first batch, which would double the live registers)
    asm volatile("" : "+v"(ng), "+v"(base), "+v"(c0w));
    return exact_ng_rows<H, KM, BOX, PT>(base, row0w, X, c0w, ng, vc, first_max, pixdst);
}

// The same on float32 pixels (PT_KEY): three rows of the neighbourhood in registers, the next one fetched while a window
// row is summed; the reference's expression as it stands (a non-finite pixel makes the sum NaN there too), and the
// first-argmax rule on the float32 values.
template <int H, int PT = PT_KEY>
__device__ __forceinline__ float exact_ng_f32(const float *__restrict__ src, int64_t X, int cy, int cx, int i, int j, bool &first_max)
{
    constexpr int BOX = 2 * H + 1, W = 2 * H + 3;
    const float *base = src + (int64_t)(i - H - 1) * X + (j - H);           // neighbourhood row t, column 1
    const int r0 = i - H - 1 < 0 ? i - H - 1 + cy : i - H - 1;                // numba negative-index wrap
    const float *row0w = src + (int64_t)r0 * X + (j - H);
    const int c0w = j - H - 1 < 0 ? cx - 1 - (j - H) : -1;
    struct __attribute__((packed, aligned(4))) Row { float v[W - 1]; };
    auto load_row = [&](int t, float (&r)[W]) {
        const float *row = t == 0 ? row0w : base + (int64_t)t * X;
        const Row q = *reinterpret_cast<const Row *>(row);
        r[0] = wide_px<PT>(row[c0w]);
#pragma unroll
        for (int c = 1; c < W; c++) r[c] = wide_px<PT>(q.v[c - 1]);
    };
    float ra[W], rb[W], rc[W];
    load_row(0, ra);
    load_row(1, rb);
    const float vc = wide_px<PT>(base[(int64_t)(H + 1) * X + H]);
    float ng = 0.0f;
    first_max = true;
#pragma unroll
    for (int k = 0; k < BOX; k++) {
        load_row(k + 2, rc);
#pragma unroll
        for (int l = 0; l < BOX; l++) {
            if (k == H && l == H) continue;
            const float o = rb[l + 1];
            first_max = first_max && ((k < H || (k == H && l < H)) ? vc > o : vc >= o);
            const float gy = sub_rn(rc[l + 1], ra[l + 1]);
            const float gx = sub_rn(rb[l + 2], rb[l]);
            ng = add_rn(ng, add_rn(mul_rn(gy, unit_y<H>(k, l)), mul_rn(gx, unit_x<H>(k, l))));
        }
#pragma unroll
        for (int c = 0; c < W; c++) { ra[c] = rb[c]; rb[c] = rc[c]; }
    }
    return ng;
}

// The same with the window rows as a LOOP and the unit vectors from the table in LDS (boxes 9 and up): laid out flat the
// (2H+1)^2 terms carry a literal pair each, and at box 13 the exact stage alone asked for more scalar and vector registers
// than the scan around it (round 5: 387 spilled VGPRs, 531 SGPRs).  Same values (unit_vectors_match), same order.
// (The 16-bit scans keep their flat exact stage: as a loop it spills fewer scalars — 324 -> 123 at box 13 — but needs MORE
// vector registers than the two batches of rows, and pays a trip to memory per window row instead of two per candidate:
// measured, boxes 9 / 13 / 17 on 512 x 512 uint16: 4.04 / 3.41 / 2.06 -> 3.57 / 3.19 / 1.86 TB/s; not kept.)
template <int H, int PT = PT_KEY>
__device__ __forceinline__ float exact_ng_f32_rows(const float *__restrict__ src, int64_t X, int cy, int cx, int i, int j, bool &first_max,
                                                   const float *__restrict__ sux, const float *__restrict__ suy)
{
    constexpr int BOX = 2 * H + 1, W = 2 * H + 3;
    const float *base = src + (int64_t)(i - H - 1) * X + (j - H);           // neighbourhood row t, column 1
    const int r0 = i - H - 1 < 0 ? i - H - 1 + cy : i - H - 1;                // numba negative-index wrap
    const float *row0w = src + (int64_t)r0 * X + (j - H);
    const int c0w = j - H - 1 < 0 ? cx - 1 - (j - H) : -1;
    struct __attribute__((packed, aligned(4))) Row { float v[W - 1]; };
    auto load_row = [&](const float *row, float (&r)[W]) {
        const Row q = *reinterpret_cast<const Row *>(row);
        r[0] = wide_px<PT>(row[c0w]);
#pragma unroll
        for (int c = 1; c < W; c++) r[c] = wide_px<PT>(q.v[c - 1]);
    };
    float ra[W], rb[W], rc[W];
    load_row(row0w, ra);
    load_row(base + X, rb);
    const float vc = wide_px<PT>(base[(int64_t)(H + 1) * X + H]);
    float ng = 0.0f;
    first_max = true;
#pragma unroll 1
    for (int k = 0; k < BOX; k++) {
        load_row(base + (int64_t)(k + 2) * X, rc);
        const bool above = k < H, centre = k == H;
        const float *ux = sux + k * BOX, *uy = suy + k * BOX;
#pragma unroll
        for (int l = 0; l < BOX; l++) {
            const float o = rb[l + 1];
            const bool skip = l == H && centre;                             // the centre itself (its unit vector is 0 / 0)
            const bool ok = (above || (centre && l < H)) ? vc > o : vc >= o;
            first_max = first_max && (skip || ok);
            const float gy = sub_rn(rc[l + 1], ra[l + 1]);
            const float gx = sub_rn(rb[l + 2], rb[l]);
            const float t = add_rn(mul_rn(gy, uy[l]), mul_rn(gx, ux[l]));
            ng = skip ? ng : add_rn(ng, t);
        }
#pragma unroll
        for (int c = 0; c < W; c++) { ra[c] = rb[c]; rb[c] = rc[c]; }
    }
    return ng;
}

// One wavefront per workgroup, persistent: it owns p.upw consecutive UNITS.  A unit is rbu rows x 512 columns of one
// frame (P > 1, frames at most 512 / P pixels wide: P consecutive row ranges of rbu rows side by side, NL = 64 / P
// lanes each, in lock step).  Consecutive units of a wave are consecutive row ranges of the same frame, so the 2H + 2
// halo rows a unit re-reads were fetched by this very wave a few steps earlier (L2 hits, no HBM traffic).
//
// Candidates (first maxima that pass the floor filter) go to a ring list in LDS that survives from unit to unit; the
// exact net gradient is evaluated 64 candidates at a time — every lane busy — whenever the ring holds that many at the
// end of a unit, and for the rest when the wave runs out of units.
//
// Floor filter.  The net gradient is a linear functional ng = sum_p w(p) f(p) of the (2H+3)^2 neighbourhood whose
// positive weights (total P_box) all lie inside the box, where the candidate v is the maximum; the negative weights
// total -P_box, N_K of it in the rows the scan has already passed when it takes the decision.  Pixels are >= 0, so
//     ng <= P_box * v - N_K * min(those rows),
// and ng > min_ng needs v > (N_K / P_box) * min + min_ng / P_box.  The minimum is kept per lane over a window of
// the last rows (own columns and the neighbour lanes'), the right-hand side becomes a packed u16 floor that joins
// the first-maximum threshold with one v_pk_max_u16 per pixel pair, and the shot-noise maxima (2 % of all pixels,
// 20 x more than there are emitters) never reach the list.
// EDGE: the frame is wider than the wave's 512 columns (lanes 0 / 63 fetch the pixels beyond it).  A compile-time
// parameter, and the edge load is issued by EVERY lane (the others point it past the descriptor's bound: no access):
// the compiler then counts two loads per row and waits with the exact vmcnt — as a run-time branch it has to assume
// the smaller count on every path and the wide frames ran at half their prefetch depth.
template <int H, int D, int P = 1, int PT = PT_U16, bool EDGE = false>
__global__ __launch_bounds__(64, fast_waves_per_simd(H, P, EDGE, PT)) void identify_scan_u16_fast_kernel(
    FastParams p, const float *__restrict__ uxy, Record *__restrict__ recs, long long cap,
    unsigned long long *__restrict__ shard_cnt, int *__restrict__ frame_count)
{
    typedef typename Px<PT>::T PX;
    constexpr int PXB = (int)sizeof(PX);                   // bytes per pixel
    constexpr int BOX = 2 * H + 1;
    constexpr int NL = 64 / P;                             // lanes per sub-band
    static_assert(P == 1 || P == 2 || P == 4 || P == 8, "sub-bands split the wavefront evenly");
    // Boxes 11 to 17 (H = 5..8) need 8 neighbour pixels per side and keep an Hrow ring of H slots of which
    // the slot about to be overwritten is skipped: both rings then share the period H.
    constexpr bool WIDE = H >= 5;
    constexpr int NB = WIDE ? 8 : 4, NA = 4 + NB, OWN = NB / 2;
    constexpr int U_ = H <= 2 ? 4 : (H == 3 ? FAST_U_H3 : (H == 4 ? FAST_U_H4 : (H == 5 ? 10 : (H == 6 ? FAST_U_H6 : 2 * H))));   // unroll period: a multiple of the ring period H
    static_assert(U_ % D == 0, "prefetch depth must divide the unroll period");
    constexpr bool UNI = H >= FAST_UNI_MIN_H && U_ % (D + H) == 0;               // one register ring for the rows in flight and the last H rows' pixels
    constexpr int RP = UNI ? D + H : D;
    constexpr int NWL = (H + 1 + 7) / 8;                   // neighbour lanes (8 columns each) a stencil reaches on either side
    // Candidate ring: entries (row << 16 | column in the aligned row) + frame index.  A chunk of rows ends early when
    // the ring holds THRESH entries (only without the floor filter, i.e. min_ng <= 0: 2 % of the pixels are maxima).
    constexpr int LIST = 1024, THRESH = LIST - 128;
    constexpr bool RING_CHECK_PER_GROUP = !(H == 3 && PT == PT_U16 && P == 1 && !EDGE);      // see the row loop

    __shared__ unsigned s_pos[LIST];
    __shared__ unsigned s_fi[LIST];
    __shared__ float s_u[2 * BOX * BOX];

    const int lane = threadIdx.x;
    for (int i = lane; i < 2 * BOX * BOX; i += 64) s_u[i] = uxy[i];
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();

    // Work of this wave: `upw` consecutive units of rbu rows, or (lin) a run of lin_rows consecutive rows of the
    // sequence (segment, frame, row): the same number of rows for every wave, and halo rows are replayed only where
    // the run starts and where it crosses into the next frame — not every rbu rows.
    const bool lin = P == 1 && p.lin_rows > 0;
    long long unit0 = (long long)blockIdx.x * p.upw;
    long long unit1 = unit0 + p.upw < p.units ? unit0 + p.upw : p.units;
    if (lin) {
        unit0 = (long long)blockIdx.x * p.lin_rows;
        unit1 = unit0 + p.lin_rows < p.lin_total ? unit0 + p.lin_rows : p.lin_total;
    }
    if (unit0 >= unit1) return;
    if (p.gate && *p.gate != p.gate_want) return;
    const int sub = P > 1 ? lane / NL : 0;                            // this lane's sub-band and its rows' offset
    // Two ways to fill a wavefront with a narrow frame: P row ranges of ONE frame side by side (each pays its 2H + 2 halo rows:
    // 8 for every 8 rows of a 64 x 64 frame at box 7), or the same rows of P consecutive frames (p.pf; round 5)
    const bool fside = P > 1 && p.pf != 0;
    const int sub_rows = fside ? 0 : sub * p.rbu;

    // The crop may start at any column: lanes work on 8-pixel chunks aligned in the FRAME (16-byte
    // loads), xoff pixels of the first chunk lie left of the crop.  Positions outside the crop are
    // masked; every allowed position has its whole window inside the crop, so looking at frame pixels
    // beyond the crop edge never changes a decision.  Column indices in the candidate list are relative
    // to the aligned origin, j = ja - xoff is relative to the crop.
    const int xoff = p.x0 & 7;
    const int nch = (xoff + p.cx + 7) >> 3;           // 8-pixel chunks per row
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    const unsigned pitch = (unsigned)p.X * (unsigned)PXB;    // Y * pitch < 2^31 on this path
    const float *sux = s_u, *suy = s_u + BOX * BOX;

    int head = 0, tail = 0;                            // candidate ring (wave-uniform)
    // The exact net gradients are evaluated as soon as the ring holds a full round of 64: the chunk of rows ends
    // there, the round runs with every lane busy, and the scan resumes (2H + 2 halo rows, L2 hits).  The waves start
    // in lock step, so the FIRST round of a wave is taken at a staggered fill level: otherwise every wave of the
    // chip would send its 64 x (2H+3) scattered row fetches to HBM in the same few microseconds.  Without the floor
    // filter (min_ng <= 0) 2 % of the pixels are candidates and the ring is drained only when nearly full.
    const bool filter = p.filt_t >= 0.0f && !(p.dbg & 4);
    // The floor assumes that no pixel a candidate's neighbourhood can reach lies below `cfloor` (the minimum the lane's
    // columns showed in the chunk before, less half the margin t = min_ng / P_box): with p' = p - cfloor >= 0 the bound is
    //     v > alpha * min + (1 - alpha) * cfloor + t,
    // which is what makes it independent of the camera offset (without it a baseline of a few hundred counts eats the
    // whole margin).  The assumption is CHECKED when the chunk is done — the minimum of every row it streamed, per
    // lane window — and a chunk that saw a lower pixel is run again: with four times the slack first, then without.
    u32 cfloor = 0u;
    float cf = -INFINITY;                                  // PT_KEY: the same bound as a float32 (nothing is assumed before the first chunk)
    bool redo = false;
    constexpr int ROUND = fast_round(H);
    int trigger = filter ? 8 + (ROUND / 8) * (int)(blockIdx.x % 7u) : THRESH;
    // Deferred exact stage (p.defer): the wave starts with exact rounds like any other and counts how many of its candidates
    // they accept; once three in four of at least 32 were accepted it emits its candidates undecided (net gradient
    // NG_DEFERRED_BITS: the fit's start-value kernel decides them from the rows it reads anyway) — a round then only copies
    // ring entries to the records, nothing has to stay near, and the ring is drained when nearly full.  Where the floor
    // lets many shot-noise maxima through (a low threshold for the box) the waves keep deciding them here, one lane per
    // candidate, instead of sending the fit a group of lanes for every reject.
    int ex_seen = 0, ex_kept = 0;
    bool defer_now = false;
    int defer_rounds = 0;

    // ---- results: buffered in registers, appended KBUF rounds at a time with ONE slot-allocating atomic per flush
    // per wave on the counter of this block's shard (a single hot counter costs ~11 ns per atomic)
    constexpr int KBUF = 2;      // (2 / 4 / 8 rounds per flush measured alike; two keep the buffers at ten registers with the pixel slot)
    int buf_i[KBUF], buf_j[KBUF], buf_f[KBUF], buf_s[KBUF], nbuf = 0, rounds = 0;
    float buf_ng[KBUF];
    const int shard = blockIdx.x & 7;
    auto flush = [&]() {
        rounds = 0;
        if (p.dbg & 2) { nbuf = 0; return; }
        unsigned long long bal[KBUF];
        int total = 0;
#pragma unroll
        for (int k = 0; k < KBUF; k++) { bal[k] = __ballot(k < nbuf); total += __popcll(bal[k]); }
        if (total) {
            // per-frame counts: one atomic per distinct frame of a round (a round holds candidates of one frame, two
            // when it straddles a unit boundary) — one per lane means up to 64 atomics on one address, serialised in L2
#pragma unroll
            for (int k = 0; k < KBUF; k++) {
                unsigned long long rest = bal[k];
                while (rest) {
                    const int f0 = __builtin_amdgcn_readlane(buf_f[k], (int)__builtin_ctzll(rest));
                    const unsigned long long same = __ballot(k < nbuf && buf_f[k] == f0) & rest;
                    if (lane == (int)__builtin_ctzll(rest)) atomicAdd(&frame_count[f0], (int)__popcll(same));
                    rest &= ~same;
                }
            }
            unsigned long long basepos = 0;
            if (lane == 0) basepos = atomicAdd(&shard_cnt[shard], (unsigned long long)total);
            basepos = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(basepos >> 32)) << 32) |
                      (unsigned)__builtin_amdgcn_readfirstlane((int)(basepos & 0xffffffffu));
            int off = 0;
#pragma unroll
            for (int k = 0; k < KBUF; k++) {
                if (k < nbuf) {
                    const long long pos = (long long)basepos + off + __popcll(bal[k] & ((1ull << lane) - 1ull));
                    if (pos < cap) {
                        Record rec;
                        rec.frame = (int32_t)(p.f_lo + buf_f[k] + p.label_off);
                        rec.yx = pack_yx(buf_i[k] + p.y0, buf_j[k] + p.x0);
                        rec.slot = buf_s[k];
                        rec.ng = buf_ng[k];
                        recs[(long long)shard * cap + pos] = rec;
                    }
                }
                off += __popcll(bal[k]);
            }
        }
        nbuf = 0;
    };
    auto append = [&](int fi, int i, int j, float ng, int slot) {
        if ((double)ng > p.min_ng) {                   // localize.py:288
#pragma unroll
            for (int k = 0; k < KBUF; k++) if (k == nbuf) { buf_f[k] = fi; buf_i[k] = i; buf_j[k] = j; buf_ng[k] = ng; buf_s[k] = slot; }
            nbuf++;
        }
    };
    auto frame_src = [&](int fi) -> const PX * {
        return (const PX *)p.movie + ((int64_t)(p.f_lo + fi) * p.Y + p.y0) * p.X + p.x0;
    };
    auto frame_fsrc = [&](int fi) -> const float * {
        return p.fmovie + ((int64_t)(p.f_lo + fi) * p.Y + p.y0) * p.X + p.x0;
    };
    // slow exact path: overflow rescans (a plateau of equal pixels flooded the ring)
    auto process_slow = [&](const PX *ksrc, int fi, int i, int j, bool recheck) {
        // (PT_KEY: the pixels are the float32 ones; `pxs` indexes whichever frame holds them)
        const float *fsrc = pt_is_key(PT) ? frame_fsrc(fi) : nullptr;
        struct { const PX *k; const float *f; __device__ float operator[](int64_t idx) const {
            if constexpr (pt_is_key(PT)) return wide_px<PT>(f[idx]); else return px_float<PT>(k[idx]); } } pxs{ksrc, fsrc};
        const int64_t src = 0;
        const float v = pxs[(int64_t)i * p.X + j];
        if (recheck) {
#pragma unroll 1
            for (int k = -H; k <= H; k++)
#pragma unroll 1
                for (int l = -H; l <= H; l++) {
                    float o = pxs[(int64_t)(i + k) * p.X + (j + l)];
                    if ((k < 0 || (k == 0 && l < 0)) ? !(v > o) : !(v >= o)) return;
                }
        }
        float ng = 0.0f;
#pragma unroll 1
        for (int k = 0; k < BOX; k++) {
            const int rk = i - H + k;
            const int rm = rk - 1 < 0 ? rk - 1 + p.cy : rk - 1;          // numba negative-index wrap
            const int64_t rowm = src + (int64_t)rm * p.X, row0 = src + (int64_t)rk * p.X, rowp = src + (int64_t)(rk + 1) * p.X;
#pragma unroll 1
            for (int l = 0; l < BOX; l++) {
                if (k == H && l == H) continue;
                const int cl = j - H + l;
                const int clm = cl - 1 < 0 ? cl - 1 + p.cx : cl - 1;
                float gy = sub_rn(pxs[rowp + cl], pxs[rowm + cl]);
                float gx = sub_rn(pxs[row0 + cl + 1], pxs[row0 + clm]);
                float sacc = add_rn(mul_rn(gy, suy[k * BOX + l]), mul_rn(gx, sux[k * BOX + l]));
                ng = add_rn(ng, sacc);
            }
        }
        append(fi, i, j, ng, -1);
    };
    // exact float32 net gradient of the n (<= 64) oldest ring entries
    auto exact_round = [&](int n) {
        // pixel slots of the round's candidates: one atomic per round on the shard's counter
        unsigned pixbase = 0xffffffffu;
        if (PT == PT_U16 && p.pix) {
            if (lane == 0) pixbase = atomicAdd(&p.pix_cnt[shard], (unsigned)n);
            pixbase = (unsigned)__builtin_amdgcn_readfirstlane((int)pixbase);
        }
        bool kept = false;
        // a wave that defers still decides one round in sixteen itself: its accept rate follows the rows it is in (a wave that
        // starts on dense spots and then crosses shot-noise rows would otherwise hand every reject to the fit)
        const bool probe = defer_now && (++defer_rounds & 15) == 0;
        if (lane < n) {
            const int q = (head + lane) & (LIST - 1);
            const unsigned e = s_pos[q];
            const int fi = (int)s_fi[q];
            const int i = (int)(e >> 16), j = (int)(e & 0xffffu) - xoff;
            const PX *src = frame_src(fi);
            int slot = -1;
            u32 *pixdst = nullptr;
            if (PT == PT_U16 && p.pix && pixbase + (unsigned)lane < p.pix_cap) {       // (pixbase = ~0 without a buffer: never below the capacity)
                slot = (int)((unsigned)shard * p.pix_cap + pixbase + (unsigned)lane);
                pixdst = p.pix + (size_t)slot * (size_t)(BOX * (H + 1));
            }
            if (defer_now && !probe) {
#pragma unroll
                for (int k = 0; k < KBUF; k++) if (k == nbuf) { buf_f[k] = fi; buf_i[k] = i; buf_j[k] = j; buf_ng[k] = __uint_as_float(NG_DEFERRED_BITS); buf_s[k] = -1; }
                nbuf++;
            } else if (p.dbg & 1) append(fi, i, j, (e & 7) == 0 ? 1e9f : 0.0f, -1);
            else {
                bool first_max;
                float ng;
                if constexpr (pt_is_key(PT) && H >= 4) ng = exact_ng_f32_rows<H, PT>(frame_fsrc(fi), p.X, p.cy, p.cx, i, j, first_max, sux, suy);
                else if constexpr (pt_is_key(PT)) ng = exact_ng_f32<H, PT>(frame_fsrc(fi), p.X, p.cy, p.cx, i, j, first_max);
                else ng = exact_ng<H, PT>(src, p.X, p.cy, p.cx, i, j, first_max, pixdst);
                if (first_max) append(fi, i, j, ng, slot);
                kept = first_max && (double)ng > p.min_ng;
            }
        }
        if (p.defer && (!defer_now || probe)) {
            if (probe) { ex_seen >>= 1; ex_kept >>= 1; }       // the history fades: two probes that reject most outweigh it
            ex_seen += n;
            ex_kept += (int)__popcll(__ballot(kept));
            defer_now = ex_seen >= 32 && ex_kept * 4 >= ex_seen * 3;
        }
        head += n;
        if (++rounds == KBUF) flush();
    };

    for (long long unit = unit0; unit < unit1;) {
        int fi, seg, b0, unit_rows, bend;
        if (lin) {
            const long long per_seg = (long long)p.nframes * p.cy;
            seg = (int)(unit / per_seg);
            const long long rem = unit - (long long)seg * per_seg;
            fi = (int)(rem / p.cy);
            b0 = (int)(rem - (long long)fi * p.cy);
            unit_rows = (int)(unit1 - unit < (long long)(p.cy - b0) ? unit1 - unit : (long long)(p.cy - b0));
            bend = b0 + unit_rows;
            unit += unit_rows;
        } else if (fside) {
            const long long fg = unit / p.upf;             // a group of P consecutive frames, upf row ranges each
            fi = (int)fg * P;                              // the frame of sub-band 0
            seg = 0;
            const int band = (int)(unit - fg * p.upf);
            b0 = band * p.rbu;
            unit_rows = min(p.rbu, p.cy - b0);
            bend = b0 + p.rbu;
            unit++;
        } else {
            fi = (int)(unit / p.upf);
            const int rem = (int)(unit - (long long)fi * p.upf);
            // consecutive units = consecutive row ranges of one 512-column segment (P > 1: `band` counts groups of P row ranges, seg = 0)
            const int bands = p.upf / p.segs;
            seg = rem / bands;
            const int band = rem - seg * bands;
            b0 = band * (p.rbu * P);                       // first row of sub-band 0
            unit_rows = min(p.rbu, p.cy - b0);             // rows of sub-band 0 (the others may hold fewer: masked)
            bend = b0 + p.rbu;
            unit++;
        }
        const int c8 = P > 1 ? lane % NL : seg * 64 + lane;
        // (frames side by side: a lane set past the last frame repeats the last one and marks nothing)
        const int sub_f = fside ? min(sub, p.nframes - 1 - fi) : 0;
        const bool lane_valid = c8 < nch && (!fside || fi + sub < p.nframes);
        const int cm = min(c8, nch - 1);
        const PX *src = frame_src(fi);
        const int col_m = cm * 8;
        const int col_l = cm > 0 ? col_m - NB : col_m;                // clamped copies feed invalid pixels only
        const int col_r = cm + 1 < nch ? col_m + 8 : col_m + 8 - NB;

        // which of the lane's 8 pixels may hold a maximum: even pixels -> bits 0..3, odd -> bits 16..19,
        // replicated for the four row slots of the candidate accumulator
        u32 colmask = 0;
        bool wrapcol = false;                              // a column with j == H: its stencil wraps to the last column
        if (lane_valid) {
#pragma unroll
            for (int b = 0; b < 8; b++) {
                int j = c8 * 8 + b - xoff;
                if (j >= H && j < p.cx - H - 1) colmask |= 1u << ((b >> 1) + 16 * (b & 1));
                if (j == H) wrapcol = true;
            }
            colmask *= 0x1111u;
        }
        // lanes whose stencils reach columns no lane of this wave holds take no floor
        // (the 4 or 8 pixels the edge lanes hold from beyond the wave cover a stencil's reach of H + 1 for every box but 9 and 17)
        constexpr bool EDGE_COVERED = (H <= 4 ? 4 : 8) >= H + 1;
        const bool no_floor = P == 1 && !EDGE_COVERED && ((seg > 0 && lane < NWL) || (seg + 1 < p.segs && lane >= 64 - NWL));

        // per-lane byte offsets inside a row (32-bit) + a wave-uniform row base: the loads use
        // SGPR-base + VGPR-offset addressing, no 64-bit vector address arithmetic per row.
        // Neighbour pixels come from the adjacent lanes' registers (DPP), not from memory: overlapping
        // 8-byte loads next to the 16-byte ones doubled the HBM-side traffic (requests to a line whose
        // fill is still in flight are not merged).  Only lanes 0 and 63 load the 4 pixels beyond the wave.
        const unsigned off_m = (unsigned)col_m * (unsigned)PXB + (unsigned)sub_f * ((unsigned)p.Y * pitch);
        const unsigned off_e = (lane == 0 || lane == 63) ? (unsigned)(lane == 0 ? col_l : col_r) * (unsigned)PXB : 0x7ffffff0u;   // others: out of bounds
        // a sub-band's row lies wholly inside its lanes, and so does the row of a frame at most 512 pixels wide: the
        // pixels lanes 0 and 63 would take from beyond the wave then only feed masked positions
        constexpr bool any_edge = EDGE && P == 1;
        constexpr bool ZF = !any_edge;
        // num_records bounds the descriptor at the end of the frames of this call: when the width is not a
        // multiple of 8 the last chunk of a row reads into the next row (masked columns), and on the very last
        // row it would read past the movie — the buffer unit returns 0 there instead
        const long long remaining = ((long long)(p.nframes - fi) * p.Y * p.X - ((long long)p.y0 * p.X + p.x0 - xoff)) * PXB;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<PX *>(src - xoff), 0, (int)(remaining < 0x7fffffffLL ? remaining : 0x7fffffffLL),
            0x00020000 /* raw 32-bit buffer, gfx94x/gfx950 */);

        int o = 0;                                         // rows of the unit already decided
        while (o < unit_rows) {
            // ---- one chunk: rows [clo, clo + len) of every sub-band (len shrinks if the ring fills up) ----
            const int clo = b0 + o, len = unit_rows - o;
            const int rs0 = clo - H - 1;
            const int nr = len + 2 * H + 2;                // pipeline rows: the stencils of rows clo .. clo + len - 1 (H + 1 rows above and below)
            // rows that may hold a maximum, in sub-band 0's numbering
            const int lo_rel = max(clo + sub_rows, H) - sub_rows;
            const int hi_rel = min(min(clo + len, bend) + sub_rows, min(p.cy, p.cy - H - 1)) - sub_rows;
            // interior chunks never touch a row outside the crop: no clamping in their row loop
            const bool interior = P == 1 && rs0 >= 0 && rs0 + nr + U_ + D <= p.cy;
            auto load_row = [&](int r) -> RowRegs {
                RowRegs ro;
                unsigned soff = 0, voff = off_m;
                if constexpr (P > 1) {
                    // every sub-band clamps its own row: the row offset joins the lane's column offset
                    const int rl = min(max(r + sub_rows, 0), p.cy - 1);
                    voff = off_m + (unsigned)rl * pitch;
                } else {
                    const int rc = interior ? r : min(max(r, 0), p.cy - 1);
                    soff = (unsigned)rc * pitch;
                }
                if constexpr (PT == PT_U8) {
                    // 8 pixels = 8 bytes; bytes (b0, b1) -> the u16 pair b0 | b1 << 16 (selector 0x0c = constant zero)
                    const u32x2_t m = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)voff, (int)soff, 0);
                    ro.m = make_uint4(__builtin_amdgcn_perm(0u, m.x, 0x0c010c00u), __builtin_amdgcn_perm(0u, m.x, 0x0c030c02u),
                                      __builtin_amdgcn_perm(0u, m.y, 0x0c010c00u), __builtin_amdgcn_perm(0u, m.y, 0x0c030c02u));
                } else if constexpr (pt_is_key(PT)) {
                    // 8 float32 pixels = 32 bytes -> four packed pairs of keys (32-bit integers: converted first)
                    const u32x4_t a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0);
                    const u32x4_t b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(voff + 16u), (int)soff, 0);
                    ro.m = make_uint4(key_pair(wide_bits<PT>(a.x), wide_bits<PT>(a.y)), key_pair(wide_bits<PT>(a.z), wide_bits<PT>(a.w)),
                                      key_pair(wide_bits<PT>(b.x), wide_bits<PT>(b.y)), key_pair(wide_bits<PT>(b.z), wide_bits<PT>(b.w)));
                } else {
                    const u32x4_t m = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0);
                    ro.m = make_uint4(m.x, m.y, m.z, m.w);
                    if constexpr (PT == PT_I16) { ro.m.x ^= 0x80008000u; ro.m.y ^= 0x80008000u; ro.m.z ^= 0x80008000u; ro.m.w ^= 0x80008000u; }
                }
                // only lanes 0 and 63 ever read `e` (as the DPP fill value): the other lanes leave it undefined
                // instead of spending two or four v_mov per row on zeros
                asm("" : "=v"(ro.e.x), "=v"(ro.e.y), "=v"(ro.e.z), "=v"(ro.e.w));
                if constexpr (any_edge) {
                    if constexpr (PT == PT_U8) {
                        if constexpr (WIDE) {
                            const u32x2_t e = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)off_e, (int)soff, 0);
                            ro.e = make_uint4(__builtin_amdgcn_perm(0u, e.x, 0x0c010c00u), __builtin_amdgcn_perm(0u, e.x, 0x0c030c02u),
                                              __builtin_amdgcn_perm(0u, e.y, 0x0c010c00u), __builtin_amdgcn_perm(0u, e.y, 0x0c030c02u));
                        } else {
                            const unsigned e = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off_e, (int)soff, 0);
                            ro.e.x = __builtin_amdgcn_perm(0u, e, 0x0c010c00u); ro.e.y = __builtin_amdgcn_perm(0u, e, 0x0c030c02u);
                        }
                    } else if constexpr (pt_is_key(PT)) {
                        const u32x4_t e = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off_e, (int)soff, 0);
                        ro.e.x = key_pair(wide_bits<PT>(e.x), wide_bits<PT>(e.y)); ro.e.y = key_pair(wide_bits<PT>(e.z), wide_bits<PT>(e.w));
                        if constexpr (WIDE) {
                            const u32x4_t f = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(off_e + 16u), (int)soff, 0);
                            ro.e.z = key_pair(wide_bits<PT>(f.x), wide_bits<PT>(f.y)); ro.e.w = key_pair(wide_bits<PT>(f.z), wide_bits<PT>(f.w));
                        }
                    } else if constexpr (WIDE) {
                        const u32x4_t e = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off_e, (int)soff, 0);
                        ro.e = make_uint4(e.x, e.y, e.z, e.w);
                        if constexpr (PT == PT_I16) { ro.e.x ^= 0x80008000u; ro.e.y ^= 0x80008000u; ro.e.z ^= 0x80008000u; ro.e.w ^= 0x80008000u; }
                    } else {
                        const u32x2_t e = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)off_e, (int)soff, 0);
                        ro.e.x = e.x; ro.e.y = e.y;
                        if constexpr (PT == PT_I16) { ro.e.x ^= 0x80008000u; ro.e.y ^= 0x80008000u; }
                    }
                }
                return ro;
            };

            // rings of period H: the horizontal window maxima of the last H rows, the maxima over (H+1) rows ending
            // at each of the last H rows, and the pixels of the last H rows
            u32 Hring[H][4], Uring[H][4], Dv[H][4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
#pragma unroll
                for (int t = 0; t < H; t++) { Hring[t][q] = 0u; Uring[t][q] = 0xffffffffu; Dv[t][q] = 0u; }
            }
            u32 mn = 0xffffffffu;                             // packed minimum of the lane's pixels since the last flush
            u32 gring[H];                                     // the same for the H flush groups before (>= 2 rows each: 2H + 2 rows back)
#pragma unroll
            for (int t = 0; t < H; t++) gring[t] = 0xffffffffu;
            u32 F = 0u;                                       // floor, both halves
            u32 acc = 0;                                      // "failed the test" bits of up to four rows
            const int tail0 = tail;                           // ring state at the start of the chunk
            int added = 0;                                    // candidates of this chunk, counted even when the ring is full
            int tail_lf = tail, rd_lf = clo;                  // ring state before, and first row of, the latest flush group
            u32 cmin = 0xffffffffu;                           // minimum of every pixel this lane streamed in the chunk (both halves)
            const float beta = fmaf(1.0f - p.filt_alpha, pt_is_key(PT) ? cf : (float)cfloor, p.filt_t);
            // A maximum in row H (column H) has the first row (column) of its neighbourhood at index -1, i.e. in the LAST
            // row (column) of the frame (numba wraps; localize.py:233-243).  Those pixels are not among the ones the
            // running minimum or the cfloor check see — but they carry negative weights only (N_wrap in total) and are
            // >= 0, so they can be dropped from the bound: with every OTHER pixel >= cfloor,
            //     ng <= P_box (v - cfloor) + cfloor N_wrap,   i.e.   v > (1 - N_wrap / P_box) cfloor + t.
            const float beta_r = fmaf(p.filt_kr, (float)cfloor, p.filt_t), beta_c = fmaf(p.filt_kc, (float)cfloor, p.filt_t),
                        beta_rc = fmaf(p.filt_krc, (float)cfloor, p.filt_t);

            // Rows in flight and the pixel history are ONE ring of D + H slots where the unroll period allows it: the row
            // loaded now lands in the slot whose pixels (row r - H) were used for the last time in this very step, and no
            // row is ever copied from the registers it was loaded into (four v_mov per row otherwise).
            RowRegs pf[RP];
#pragma unroll
            for (int d = 0; d < RP; d++) {
                if (d < D) pf[d] = load_row(rs0 + d);
                else pf[d].m = make_uint4(0u, 0u, 0u, 0u);
            }

            int sb = 0;
            bool ring_full = false;
            for (; sb < nr; sb += U_) {
#pragma unroll
                for (int u = 0; u < U_; u++) {
                    const int st = sb + u;                    // pipeline step; row r = rs0 + st
                    const RowRegs cur = pf[u % RP];
                    if constexpr (!UNI) pf[u % RP] = load_row(rs0 + st + D);
                    u32 dvq[4];                               // the pixels of row r - H
#pragma unroll
                    for (int q = 0; q < 4; q++) dvq[q] = UNI ? quad(pf[(u + D) % RP].m, q) : Dv[u % H][q];
                    u32 A[NA], Bp[NA];
                    if constexpr (WIDE) {
                        A[0] = px_from_lane_below<ZF>(cur.m.x, cur.e.x); A[1] = px_from_lane_below<ZF>(cur.m.y, cur.e.y);
                        A[2] = px_from_lane_below<ZF>(cur.m.z, cur.e.z); A[3] = px_from_lane_below<ZF>(cur.m.w, cur.e.w);
                        A[8] = px_from_lane_above<ZF>(cur.m.x, cur.e.x); A[9] = px_from_lane_above<ZF>(cur.m.y, cur.e.y);
                        A[10] = px_from_lane_above<ZF>(cur.m.z, cur.e.z); A[11] = px_from_lane_above<ZF>(cur.m.w, cur.e.w);
                    } else {
                        A[0] = px_from_lane_below<ZF>(cur.m.z, cur.e.x); A[1] = px_from_lane_below<ZF>(cur.m.w, cur.e.y);
                        A[6] = px_from_lane_above<ZF>(cur.m.x, cur.e.x); A[7] = px_from_lane_above<ZF>(cur.m.y, cur.e.y);
                    }
                    A[OWN] = cur.m.x; A[OWN + 1] = cur.m.y; A[OWN + 2] = cur.m.z; A[OWN + 3] = cur.m.w;
                    Bp[0] = 0;
#pragma unroll
                    for (int k = 1; k < NA; k++) Bp[k] = __builtin_amdgcn_alignbit(A[k], A[k - 1], 16);
                    mn = pk_min(pk_min(mn, pk_min(A[OWN], A[OWN + 1])), pk_min(A[OWN + 2], A[OWN + 3]));
                    if (any_edge) {        // rows wider than the wave: the neighbour pixels count too (lanes 0 / 63: from beyond the wave)
#pragma unroll
                        for (int k = 0; k < OWN; k++) mn = pk_min(mn, pk_min(A[k], A[NA - 1 - k]));
                    }

                    // Row r - H holds a candidate where its pixel equals the maximum of its (2H+1)^2 window (rows
                    // r - 2H .. r) and reaches the floor.  np.argmax takes the FIRST maximum of the window
                    // (picasso/localize.py:128): equal pixels before the centre disqualify it — that is checked
                    // on the exact neighbourhood when the net gradient is evaluated, not here.
                    u32 hrow[4];
                    u32 tq[4];
                    if constexpr (H == 3) {
                        // box 7, hand-scheduled.  Horizontal: pair maxima E(s) = max(p(s), p(s+1)) at the six odd
                        // offsets, two of them doubled, 16 instructions for the four 7-pixel windows (22 by plain
                        // doubling).  Vertical: P2(r) = max(Hrow r, r-1), U(r) = max(P2(r), P2(r-2)) = rows r-3..r
                        // in two instructions (Hring[0] = Hrow(r-1), Hring[1..2] = P2 of the last two rows).
                        // The four pixel pairs advance stage by stage (the empty asm pins each stage): a packed
                        // instruction that consumes the result of the one just before it costs a wait state, and
                        // the scheduler, short of registers, otherwise walks one pair's chain after the other.
                        const u32 Em3 = pk_max(pair_at<-3, NA>(A, Bp), pair_at<-2, NA>(A, Bp));
                        const u32 Em1 = pk_max(pair_at<-1, NA>(A, Bp), pair_at<0, NA>(A, Bp));
                        const u32 E1 = pk_max(pair_at<1, NA>(A, Bp), pair_at<2, NA>(A, Bp));
                        const u32 E3 = pk_max(pair_at<3, NA>(A, Bp), pair_at<4, NA>(A, Bp));
                        const u32 E5 = pk_max(pair_at<5, NA>(A, Bp), pair_at<6, NA>(A, Bp));
                        const u32 E7 = pk_max(pair_at<7, NA>(A, Bp), pair_at<8, NA>(A, Bp));
                        const u32 Fm1 = pk_max(Em1, E1), F3 = pk_max(E3, E5);
                        u32 h0 = pk_max(Em3, Fm1), h1 = pk_max(Fm1, E3), h2 = pk_max(E1, F3), h3 = pk_max(F3, E7);
                        asm volatile("" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3));
                        hrow[0] = pk_max(h0, pair_at<3, NA>(A, Bp)); hrow[1] = pk_max(h1, pair_at<5, NA>(A, Bp));
                        hrow[2] = pk_max(h2, pair_at<7, NA>(A, Bp)); hrow[3] = pk_max(h3, pair_at<9, NA>(A, Bp));
                        asm volatile("" : "+v"(hrow[0]), "+v"(hrow[1]), "+v"(hrow[2]), "+v"(hrow[3]));
                        u32 p2[4], uc[4], t1[4], t2[4], d[4];
#pragma unroll
                        for (int q = 0; q < 4; q++) p2[q] = pk_max(hrow[q], Hring[0][q]);
                        asm volatile("" : "+v"(p2[0]), "+v"(p2[1]), "+v"(p2[2]), "+v"(p2[3]));
#pragma unroll
                        for (int q = 0; q < 4; q++) uc[q] = pk_max(p2[q], Hring[1 + u % 2][q]);     // P2(r-2)
                        asm volatile("" : "+v"(uc[0]), "+v"(uc[1]), "+v"(uc[2]), "+v"(uc[3]));
#pragma unroll
                        for (int q = 0; q < 4; q++) t1[q] = pk_max(uc[q], Uring[u % H][q]);
                        asm volatile("" : "+v"(t1[0]), "+v"(t1[1]), "+v"(t1[2]), "+v"(t1[3]));
#pragma unroll
                        for (int q = 0; q < 4; q++) t2[q] = pk_max(t1[q], F);
                        asm volatile("" : "+v"(t2[0]), "+v"(t2[1]), "+v"(t2[2]), "+v"(t2[3]));
#pragma unroll
                        for (int q = 0; q < 4; q++) d[q] = pk_sub_sat(t2[q], dvq[q]);
                        asm volatile("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            tq[q] = pk_min(d[q], 0x00010001u);
                            Uring[u % H][q] = uc[q];
                            Hring[0][q] = hrow[q];
                            Hring[1 + u % 2][q] = p2[q];
                        }
                    } else if constexpr (H >= 4 && H <= 6) {
                        // boxes 9 to 13.  Vertical by doubling, as box 7: P2(r) = max(Hrow r, r-1), P4(r) = max(P2(r), P2(r-2)),
                        // U(r) = max(P4(r), P4(r - (H-3))) = rows r-H .. r in three instructions per pixel pair instead of H.
                        // The H ring slots hold Hrow(r-1), P2 of the last two rows and P4 of the last H-3 rows.
                        constexpr int PQ = H - 3;
                        static_assert(U_ % 2 == 0 && U_ % PQ == 0, "the unroll period covers the periods of the P2 and P4 rings");
                        window_maxima<H, NA>(A, Bp, hrow);
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const u32 p2 = pk_max(hrow[q], Hring[0][q]);
                            const u32 p4 = pk_max(p2, Hring[1 + u % 2][q]);
                            const u32 Ucur = pk_max(p4, Hring[3 + u % PQ][q]);                 // rows r - H .. r
                            const u32 thr = pk_max(pk_max(Ucur, Uring[u % H][q]), F);       // slot u % H: the row H steps back
                            tq[q] = pk_min(pk_sub_sat(thr, dvq[q]), 0x00010001u);     // 0 in a half = candidate
                            Uring[u % H][q] = Ucur;
                            Hring[0][q] = hrow[q];
                            Hring[1 + u % 2][q] = p2;
                            Hring[3 + u % PQ][q] = p4;
                        }
                    } else {
                        hrow[0] = wmax<BOX, 0 - H, NA>(A, Bp); hrow[1] = wmax<BOX, 2 - H, NA>(A, Bp);
                        hrow[2] = wmax<BOX, 4 - H, NA>(A, Bp); hrow[3] = wmax<BOX, 6 - H, NA>(A, Bp);
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            u32 Ucur = hrow[q];                    // rows r - H .. r
#pragma unroll
                            for (int t = 0; t < H; t++) Ucur = pk_max(Ucur, Hring[t][q]);
                            const u32 thr = pk_max(pk_max(Ucur, Uring[u % H][q]), F);       // slot u % H: the row H steps back
                            tq[q] = pk_min(pk_sub_sat(thr, dvq[q]), 0x00010001u);     // 0 in a half = candidate
                            Uring[u % H][q] = Ucur;
                            Hring[u % H][q] = hrow[q];
                        }
                    }
                    if constexpr (UNI) {
                        asm volatile("" ::: "memory");         // the load below stays behind the last use of the slot it refills
                        pf[(u + D) % RP] = load_row(rs0 + st + D);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; q++) Dv[u % H][q] = A[q + OWN];
                    }
                    const int t4 = u % 4;                      // row slot inside the accumulator
                    acc |= (tq[0] | (tq[1] << 1) | (tq[2] << 2) | (tq[3] << 3)) << (4 * t4);
                    if (t4 == 3 || u == U_ - 1) {              // flush the candidates of the last t4+1 rows
                        const int rd0 = rs0 + (st - t4) - H;   // decision row of slot 0
                        u32 rowmask = 0;
#pragma unroll
                        for (int tt = 0; tt <= t4; tt++)
                            if (rd0 + tt >= lo_rel && rd0 + tt < hi_rel) rowmask |= 0x000f000fu << (4 * tt);
                        u32 pass = ~acc & rowmask & colmask;
                        {
                            // Two candidates side by side, or one above the other, hold the same value (each is the
                            // maximum of a window that contains the other), and np.argmax takes the first: the right-hand
                            // / lower one cannot be it.  Dropped here — within the lane's 8 pixels and the group's rows —
                            // a plateau of equal pixels (a saturated fiducial) sends its upper-left rim to the ring
                            // instead of every pixel.
                            // (keys: equal keys are not equal pixels — the right-hand one may be the larger)
                            const u32 c = pt_is_key(PT) ? 0u : pass;
                            pass &= ~((c & 0x0000ffffu) << 16);                    // odd pixel, left neighbour = the even pixel of its pair
                            pass &= ~(((c >> 16) << 1) & 0x0000eeeeu);             // even pixel 2q (q > 0), left neighbour = odd pixel of pair q - 1
                            pass &= ~((c & 0x0fff0fffu) << 4);                     // same pixel, one row up (previous row slot)
                        }
                        acc = 0;
                        if (RING_CHECK_PER_GROUP && ring_full) pass = 0u;   // the ring is nearly full: the rest of this period's rows are left to the next chunk
                        else { tail_lf = tail; rd_lf = rd0; }
                        // Append to the wave's ring: every round each lane that still has a candidate emits its
                        // lowest one, slots come from a ballot prefix count and the ring state stays in scalar
                        // registers.  (An LDS atomicAdd per lane is turned into a serial per-lane scan by the
                        // compiler's atomic optimizer: ~9 SALU instructions per active lane, per flush.)
                        const u32 ebase = ((u32)(rd0 + sub_rows) << 16) + (u32)(c8 << 3);
                        int grp_added = 0;                             // candidates of this flush group
                        for (;;) {
                            const bool has = pass != 0;
                            const unsigned long long bal = __ballot(has);
                            if (bal == 0) break;
                            if (has) {
                                const u32 b = (u32)__ffs(pass) - 1u;
                                pass &= pass - 1;
                                // bit b: row slot (b >> 2) & 3, pixel 2 * (b & 3) + (b >> 4)
                                const u32 e = ebase + (((b >> 2) & 3u) << 16) + ((b & 3u) << 1) + (b >> 4);
                                const int slot = tail + (int)__builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
                                if (slot - head < LIST) { s_pos[slot & (LIST - 1)] = e; s_fi[slot & (LIST - 1)] = (unsigned)(fi + sub_f); }
                            }
                            const int nb_ = __popcll(bal);
                            added += nb_;
                            if constexpr (RING_CHECK_PER_GROUP) grp_added += nb_;
                            tail = min(tail + nb_, head + LIST);
                        }
                        // ---- the floor for the decisions of the next flush group ----
                        if (filter) {
                            const u32 gm = pk_min(mn, __builtin_amdgcn_alignbit(mn, mn, 16));    // both halves = the lane's minimum
                            u32 wmin = gm;
#pragma unroll
                            for (int t = 0; t < H; t++) wmin = pk_min(wmin, gring[t]);
#pragma unroll
                            for (int t = H - 1; t > 0; t--) gring[t] = gring[t - 1];
                            gring[0] = gm;
                            cmin = pk_min(cmin, gm);
                            mn = 0xffffffffu;
                            u32 wl = wmin, wr = wmin;
#pragma unroll
                            for (int t = 0; t < NWL; t++) {
                                wl = from_lane_below(wl, 0xffffffffu); wr = from_lane_above(wr, 0xffffffffu);
                                wmin = pk_min(wmin, pk_min(wl, wr));
                            }
                            // the next group decides rows r - H + 1 .. r - H + 4 (r = this row); the row with index H
                            // reads its upper ring row from the LAST row (wrap): no floor then
                            const int rnext = rs0 + st + 1 - H + sub_rows;
                            const bool wraprow = rnext <= H && rnext + 3 >= H;
                            const float flw = wrapcol ? (wraprow ? beta_rc : beta_c) : beta_r;
                            u32 fi_;
                            if constexpr (pt_is_key(PT)) {
                                // the bound from the lower edge of the minimum's bucket, back to the key of the bucket it falls into:
                                // v > fl implies key(v) >= key(fl).  Wrapped stencils reach pixels the minimum has not seen and
                                // that may be negative here: no floor for them; none either when the bound is not a number.
                                const float fl = fmaf(p.filt_alpha, key_lower_edge(wmin & 0xffffu), beta);
                                fi_ = (wrapcol || wraprow || !(fl == fl)) ? 0u : key_of_float(fl);
                            } else {
                                const float fl = (wrapcol || wraprow) ? flw : fmaf(p.filt_alpha, (float)(wmin & 0xffffu), beta);
                                fi_ = (u32)fminf(fl, 65535.0f);                                // fl >= 0
                            }
                            fi_ = no_floor ? 0u : fi_;
                            F = fi_ | (fi_ << 16);
                        }
                        // The ring is looked at after every flush group, not only once per unroll period (6 ... 16 rows): on
                        // low-count 8-bit movies at boxes >= 15 — a quantised background where the floor cannot bite and two
                        // pixels in a hundred tie for their window's maximum — a period's rows overflowed it, the chunk counted
                        // as flooded and was rescanned pixel by pixel (20 GB/s instead of 1.3 Tpx/s, round 6).
                        // (no branch out of the unrolled period: it would cost the register rings their static indices.  From here
                        // to the period's end nothing is appended, and the chunk ends AFTER this group: one more row streams behind
                        // it — unless this is the period's last step —, so the check of the floor's assumption covers its rows.)
                        // (room for twice what this group added, at least FAST_RING_MARGIN: a tie-dense movie at a threshold the
                        // floor cannot bite on adds 150 and more per group)
                        // (the hand-scheduled box-7 scan of uint16 frames no wider than the wave — config 2's kernel — keeps the check
                        // per period: with this one the step with two frame ranges in flight was 1.5 % slower, same box, alternating:
                        // 2.71 - 2.73 against 2.67 - 2.68 ms; alone the two scans take the same 1.00 - 1.03 ms)
                        if constexpr (RING_CHECK_PER_GROUP)
                        if (!ring_full && u < U_ - 1 && tail - head > LIST - max(FAST_RING_MARGIN, 2 * grp_added) && rd0 + t4 + 1 > clo) { ring_full = true; tail_lf = tail; rd_lf = rd0 + t4 + 1; }
                    }
                }
                if (ring_full || (tail - head >= trigger && rd_lf > clo) || added > LIST - 64) { sb += U_; break; }
            }
            // Rows decided: every row whose decision step lies before sb.  A chunk that ends early (sb < nr) keeps
            // only the rows before its latest flush group: the neighbourhoods of those have streamed in completely,
            // so the check of the floor's assumption below covers them.
            bool flooded = added > LIST - (tail0 - head);
            int dn = min(len, sb - 2 * H - 1);
            if ((sb < nr || ring_full) && !flooded) { tail = tail_lf; dn = min(len, rd_lf - clo); }
            // a chunk that ended before it decided a single row (the ring filled within its first rows: whole rows of equal
            // keys, where the neighbour rule does not apply) makes no progress: one row the slow way
            if (dn < 1) { flooded = true; dn = 1; }
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            if (filter && !flooded) {
                u32 cw = cmin, wl = cmin, wr = cmin;
#pragma unroll
                for (int t = 0; t < NWL; t++) {
                    wl = from_lane_below(wl, 0xffffffffu); wr = from_lane_above(wr, 0xffffffffu);
                    cw = pk_min(cw, pk_min(wl, wr));
                }
                cw &= 0xffffu;
                bool broken;
                if constexpr (pt_is_key(PT)) {
                    const float cwf = key_lower_edge(cw), slackf = p.filt_slackf;
                    broken = cwf < cf;                         // (a NaN lower edge compares false, and makes the next floor NaN = none)
                    cf = cwf - slackf;
                    if (__any(broken)) {
                        tail = tail0;
                        cf = redo ? -INFINITY : cwf - 4.0f * slackf;
                        redo = true;
                        continue;
                    }
                } else {
                broken = cw < cfloor;                          // a pixel below the assumed floor: the decisions of this chunk are void
                const u32 slack = (u32)p.filt_slack;
                cfloor = cw > slack ? cw - slack : 0u;
                if (__any(broken)) {                           // run the chunk again; a second failure in a row drops the assumption
                    tail = tail0;
                    cfloor = (redo || cw <= 4u * slack) ? 0u : cw - 4u * slack;
                    redo = true;
                    continue;
                }
                }
                redo = false;
            }
            if (flooded) {
                // More candidates than the ring holds: a plateau of equal pixels above the floor (saturation) — every
                // one of them equals its window maximum.  Drop this chunk's entries and rescan its rows pixel by
                // pixel with the exact test (slow, rare).
                tail = tail0;
                const int wcols = P > 1 ? NL * 8 : 512;
                for (int s = 0; s < P; s++) {
                    const int rofs = fside ? 0 : s * p.rbu, fs = fside ? fi + s : fi;       // the lane set's rows / frame
                    if (fs >= p.nframes) break;
                    const PX *ssrc = fside ? frame_src(fs) : src;
                    const int r_lo = max(clo + rofs, H), r_hi = min(min(clo + dn, bend) + rofs, p.cy - H - 1);
                    for (int idx0 = 0; idx0 < dn * wcols; idx0 += 64) {
                        const int idx = idx0 + lane;
                        const int i = clo + rofs + idx / wcols;
                        const int ja = (P > 1 ? 0 : seg * 512) + idx % wcols, j = ja - xoff;
                        if (i >= r_lo && i < r_hi && ja < nch * 8 && j >= H && j < p.cx - H - 1) process_slow(ssrc, fs, i, j, true);
                        flush();
                    }
                }
            }
            // (evaluating in place, inside the row loop, was tried: the scan's 120 live registers and the round's 50
            // do not fit, and the spills land in the row loop — 5.1 ms instead of 1.5)
            if (!filter || tail - head >= trigger || ring_full) {
                while (tail - head >= 64) exact_round(64);
                if (filter) {
                    if (tail - head >= ROUND || (trigger < ROUND && tail > head)) exact_round(tail - head);
                    // A threshold so low that most maxima pass the floor (two and more candidates per row): ending the
                    // chunk every ROUND candidates would replay the 2H + 2 halo rows every dozen rows — the ring
                    // is then drained only when nearly full, as without a floor.
                    trigger = ((dn > 0 && added >= 2 * dn) || defer_now) ? THRESH : ROUND;
                }
            }
            o += dn;
        }
    }
    while (tail - head >= 64) exact_round(64);
    if (tail > head) exact_round(tail - head);
    flush();
}

// rows in flight of the key scans (4-byte pixels: a row in flight is eight registers until its keys are built): box 13 with
// the edge loads four instead of six (253 spilled registers -> 2), boxes 15 / 17 two (seven / eight: 250 spilled)
constexpr int key_depth(int H, int D, bool EDGE) { return H >= 7 ? 2 : (H == 6 && EDGE ? 4 : D); }

template <int H, int D, int P = 1>
static int launch_fast(const FastParams &p, int pt, const float *d_tab, Record *recs, long long cap,
                       unsigned long long *shard_cnt, int *frame_count, hipStream_t s)
{
    const long long blocks = (p.units + p.upw - 1) / p.upw;
    if (blocks > 0x7fffffffLL) { set_error("identify: too many blocks (%lld)", blocks); return PMI_ERR_ARG; }
    const dim3 g((unsigned)blocks), b(64);
    const bool edge = P == 1 && p.segs > 1;
    snprintf(g_last_scan_kernel, sizeof(g_last_scan_kernel), "identify_scan_u16_fast_kernel<%d, %d, %d, %d, %s>%s", H,
             pt_is_key(pt) ? key_depth(H, D, edge) : D, P, pt, edge ? "true" : "false", p.defer ? " defer" : "");
    if (P > 1 && p.pf) strncat(g_last_scan_kernel, " frames", sizeof(g_last_scan_kernel) - strlen(g_last_scan_kernel) - 1);
    if constexpr (P == 1) {
        if (p.segs > 1) {             // frames wider than a wave: the variant with the edge loads
            constexpr int DK = key_depth(H, D, true);
            if (pt == PT_KEY) hipLaunchKernelGGL((identify_scan_u16_fast_kernel<H, DK, 1, PT_KEY, true>), g, b, 0, s, p, d_tab, recs, cap, shard_cnt, frame_count);
            else if (pt == PT_KEY_I32) hipLaunchKernelGGL((identify_scan_u16_fast_kernel<H, DK, 1, PT_KEY_I32, true>), g, b, 0, s, p, d_tab, recs, cap, shard_cnt, frame_count);
            else if (pt == PT_KEY_U32) hipLaunchKernelGGL((identify_scan_u16_fast_kernel<H, DK, 1, PT_KEY_U32, true>), g, b, 0, s, p, d_tab, recs, cap, shard_cnt, frame_count);
            else if (pt == PT_U8) hipLaunchKernelGGL((identify_scan_u16_fast_kernel<H, D, 1, PT_U8, true>), g, b, 0, s, p, d_tab, recs, cap, shard_cnt, frame_count);
            else if (pt == PT_I16) hipLaunchKernelGGL((identify_scan_u16_fast_kernel<H, D, 1, PT_I16, true>), g, b, 0, s, p, d_tab, recs, cap, shard_cnt, frame_count);
            else hipLaunchKernelGGL((identify_scan_u16_fast_kernel<H, D, 1, PT_U16, true>), g, b, 0, s, p, d_tab, recs, cap, shard_cnt, frame_count);
            PMI_HIP(hipGetLastError());
            return PMI_OK;
        }
    }
    if constexpr (P == 1) {           // 32-bit integer pixels: one row range per lane set only (launch_scan_u16_fast)
        if (pt == PT_KEY_I32 || pt == PT_KEY_U32) {
            constexpr int DK = key_depth(H, D, false);
            if (pt == PT_KEY_I32) hipLaunchKernelGGL((identify_scan_u16_fast_kernel<H, DK, 1, PT_KEY_I32>), g, b, 0, s, p, d_tab, recs, cap, shard_cnt, frame_count);
            else hipLaunchKernelGGL((identify_scan_u16_fast_kernel<H, DK, 1, PT_KEY_U32>), g, b, 0, s, p, d_tab, recs, cap, shard_cnt, frame_count);
            PMI_HIP(hipGetLastError());
            return PMI_OK;
        }
    }
    if (pt == PT_KEY) hipLaunchKernelGGL((identify_scan_u16_fast_kernel<H, key_depth(H, D, false), P, PT_KEY>), g, b, 0, s, p, d_tab, recs, cap, shard_cnt, frame_count);
    else if (pt == PT_U8) hipLaunchKernelGGL((identify_scan_u16_fast_kernel<H, D, P, PT_U8>), g, b, 0, s, p, d_tab, recs, cap, shard_cnt, frame_count);
    else if (pt == PT_I16) hipLaunchKernelGGL((identify_scan_u16_fast_kernel<H, D, P, PT_I16>), g, b, 0, s, p, d_tab, recs, cap, shard_cnt, frame_count);
    else hipLaunchKernelGGL((identify_scan_u16_fast_kernel<H, D, P, PT_U16>), g, b, 0, s, p, d_tab, recs, cap, shard_cnt, frame_count);
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
int launch_scan_u16_fast(const void *d_movie, int dtype, int64_t Y, int64_t X, int y0, int x0, int cy, int cx, int64_t f_lo,
                         int64_t label_off, int nframes, int box, double min_ng, const float *d_tab, Record *recs,
                         long long cap, unsigned long long *n_total, int *frame_count, hipStream_t s, bool *handled,
                         const int *gate, uint32_t *pix, unsigned *pix_cnt, unsigned pix_cap, bool defer, const float *fmovie, int gate_want)
{
    *handled = false;
    static const bool force_generic = tuning_env("PMI_IDENTIFY_GENERIC") != nullptr;
    if (force_generic) return PMI_OK;
    const int h = box / 2;
    if (h < 1 || h > 8) return PMI_OK;
    if (!unit_vectors_match()) return PMI_OK;          // never expected; the generic kernel uses the runtime table
    // uint16, uint8 and int16 movies (see Px); rows need no alignment beyond the pixel's own: gfx950 runs buffer and
    // global loads in unaligned mode, so odd widths only cost the loads that straddle a 64-byte boundary
    // (fmovie: a float32 movie — d_movie and fmovie are the same frames — scanned through keys built in registers)
    const int pt = fmovie ? ((const void *)fmovie != d_movie ? -1 : (dtype == PMI_F32 ? PT_KEY : (dtype == PMI_I32 ? PT_KEY_I32 : (dtype == PMI_U32 ? PT_KEY_U32 : -1))))
                          : (dtype == PMI_U16 ? PT_U16 : (dtype == PMI_U8 ? PT_U8 : (dtype == PMI_I16 ? PT_I16 : -1)));
    if (pt < 0) return PMI_OK;
    const int pxb = pt == PT_U8 ? 1 : (pt_is_key(pt) ? 4 : 2);
    if (cx < 16 || ((uintptr_t)d_movie & (uintptr_t)(pxb - 1))) return PMI_OK;
    if (cy > 65535 || cx > 65520 || X > 65535 || Y * X * pxb >= (1LL << 31)) return PMI_OK;   // 32-bit row offsets, 16-bit list columns
    const int g_fast_cus = device_cu_count();
    // narrow frames: several row ranges side by side in one wavefront instead of idle lanes
    const int nch = ((x0 & 7) + cx + 7) / 8;
    static const bool no_pack = tuning_env("PMI_IDENTIFY_NOPACK") != nullptr;
    int pack = 1;
    if (h >= 2 && h <= 6 && !no_pack) {
        if (nch <= 8 && h == 3) pack = 8;              // <= 64 px wide: eight row ranges side by side
        else if (nch <= 16 && h <= 4) pack = 4;
        else if (nch <= 32) pack = 2;
    }
    // (the 32-bit integer scans are built for one row range per lane set: narrow frames keep the routes they had)
    if (pack > 1 && (pt == PT_KEY_I32 || pt == PT_KEY_U32)) return PMI_OK;
    FastParams p;
    p.movie = d_movie; p.Y = Y; p.X = X; p.y0 = y0; p.x0 = x0; p.cy = cy; p.cx = cx;
    p.f_lo = f_lo; p.label_off = label_off; p.nframes = nframes; p.box = box; p.min_ng = min_ng; p.gate = gate;
    p.pix = pix; p.pix_cnt = pix_cnt; p.pix_cap = pix_cap;
    p.defer = defer && !fmovie ? 1 : 0;
    p.fmovie = fmovie; p.gate_want = gate_want;
    p.segs = pack > 1 ? 1 : (nch + 63) / 64;
    // Rows per unit: long units amortise the 2H + 2 pipeline rows a unit spends on its halo, short ones balance the
    // persistent waves (every wave runs ceil(units / waves) units).  Pick the length with the least total work.
    const long long waves = (long long)g_fast_cus * 4 * fast_waves_per_simd(h, pack, pack == 1 && p.segs > 1, pt);
    static const int force_rbu = tuning_env("PMI_IDENTIFY_RBU") ? atoi(tuning_env("PMI_IDENTIFY_RBU")) : 0;
    int best_rbu = 0;
    double best_cost = 0.0;
    const int full = (cy + pack - 1) / pack;
    const int cand_rbu[] = {full, 1024, 512, 256, 128, 64, 32, 16, 8};
    for (int r : cand_rbu) {
        if (r > full || r < 1 || (r < 8 && r != full)) continue;
        if (force_rbu > 0) r = std::min(force_rbu, full);
        const long long upf = (long long)((cy + r * pack - 1) / (r * pack)) * p.segs;
        const long long units = upf * nframes;
        const long long per_wave = (units + waves - 1) / waves;
        const double cost = (double)per_wave * (r + 2 * h + 2);      // pipeline rows of the busiest wave
        if (!best_rbu || cost < best_cost) { best_rbu = r; best_cost = cost; }
    }
    p.rbu = best_rbu;
    p.upf = ((cy + p.rbu * pack - 1) / (p.rbu * pack)) * p.segs;
    p.units = (long long)p.upf * nframes;
    p.pf = 0;
    static const bool no_pf = tuning_env("PMI_IDENTIFY_NOPF") != nullptr;
    if (pack > 1 && !no_pf && force_rbu <= 0 && nframes >= pack && (long long)pack * Y * X * pxb < (1LL << 31)) {
        // the same rows of `pack` consecutive frames side by side instead of `pack` row ranges of one frame: no halo rows
        // inside a frame.  Same cost model; taken when it is the cheaper one (short frames: 64 x 64 at box 7 pays 72 pipeline
        // rows per frame group of eight instead of 16 per frame)
        const long long groups = (nframes + pack - 1) / pack;
        int pf_rbu = 0;
        double pf_cost = 0.0;
        const int cand_pf[] = {cy, 1024, 512, 256, 128, 64, 32, 16, 8};
        for (int r : cand_pf) {
            if (r > cy || r < 1 || (r < 8 && r != cy)) continue;
            const long long units = (long long)((cy + r - 1) / r) * groups;
            const double cost = (double)((units + waves - 1) / waves) * (r + 2 * h + 2);
            if (!pf_rbu || cost < pf_cost) { pf_rbu = r; pf_cost = cost; }
        }
        if (pf_cost < best_cost) {
            p.pf = 1;
            p.rbu = pf_rbu;
            p.upf = (cy + pf_rbu - 1) / pf_rbu;
            p.units = (long long)p.upf * groups;
        }
    }
    const long long blocks = std::min<long long>(p.units, waves);
    p.upw = (int)((p.units + blocks - 1) / blocks);
    p.lin_rows = 0; p.lin_total = 0;
    static const bool no_lin = tuning_env("PMI_IDENTIFY_NOLIN") != nullptr;
    if (pack == 1 && !no_lin && force_rbu <= 0) {
        // one row range per lane set: every wave the same number of consecutive rows (at least 64: a run pays 2h + 2
        // halo rows at its start)
        p.lin_total = (long long)p.segs * nframes * cy;
        const long long per = std::max<long long>(64, (p.lin_total + waves - 1) / waves);
        p.lin_rows = (int)std::min<long long>(per, 1 << 30);
        p.units = (p.lin_total + p.lin_rows - 1) / p.lin_rows;      // = workgroups (launch_fast)
        p.upw = 1;
    }
    // Floor filter (see the kernel): ng = sum_p w(p) f(p) over the (2H+3)^2 neighbourhood, sum_p w(p) = 0, positive
    // weights (total P_box) inside the box only.  N_K = negative weight in the rows a <= 2H - 3 of the neighbourhood,
    // the ones certainly behind the scan when it decides (a flush group is at most 4 rows).  Double precision;
    // 0.1 % margin for the float32 rounding of the reference's own summation.
    double P_box = 0.0, N_K = 0.0, N_row0 = 0.0, N_col0 = 0.0;
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
        for (int a = 0; a < n; a++)
            for (int b = 0; b < n; b++) {
                const double v = wgt[(size_t)a * n + b];
                if (v > 0) P_box += v;
                else if (a <= 2 * h - 3) N_K -= v;
                if (a == 0) N_row0 -= v;               // first row / first column of the neighbourhood: negative weights only
                if (b == 0) N_col0 -= v;
            }
    }
    if (min_ng > 0.0 && std::isfinite(min_ng)) {
        p.filt_alpha = (float)(N_K / P_box * 0.998);
        p.filt_t = (float)std::max(0.0, min_ng / (P_box * 1.001) - 1.0);     // one count of slack for the float32 evaluation
        p.filt_slack = (int)std::min(65535.0, std::max(2.0, std::ceil(0.5 * min_ng / P_box)));
        p.filt_slackf = (float)std::max(2.0, std::ceil(0.5 * min_ng / P_box));
        p.filt_kr = (float)std::max(0.0, (1.0 - N_row0 / P_box) * 0.998);
        p.filt_kc = (float)std::max(0.0, (1.0 - N_col0 / P_box) * 0.998);
        p.filt_krc = (float)std::max(0.0, (1.0 - (N_row0 + N_col0) / P_box) * 0.998);
    } else {
        p.filt_alpha = 0.0f; p.filt_t = -1.0f; p.filt_slack = 0; p.filt_slackf = 0.0f;
        p.filt_kr = p.filt_kc = p.filt_krc = 0.0f;
    }
    static const int dbg = tuning_env("PMI_IDENTIFY_DBG") ? atoi(tuning_env("PMI_IDENTIFY_DBG")) : 0;
    p.dbg = dbg;
    int rc;
    switch (h) {
    case 1: rc = launch_fast<1, 2>(p, pt, d_tab, recs, cap, n_total, frame_count, s); break;
    case 2:
        if (pack == 4) rc = launch_fast<2, 2, 4>(p, pt, d_tab, recs, cap, n_total, frame_count, s);
        else if (pack == 2) rc = launch_fast<2, 2, 2>(p, pt, d_tab, recs, cap, n_total, frame_count, s);
        else rc = launch_fast<2, FAST_D_H2>(p, pt, d_tab, recs, cap, n_total, frame_count, s);
        break;
    case 3:
        if (pack == 8) rc = launch_fast<3, FAST_D_H3, 8>(p, pt, d_tab, recs, cap, n_total, frame_count, s);
        else if (pack == 4) rc = launch_fast<3, FAST_D_H3, 4>(p, pt, d_tab, recs, cap, n_total, frame_count, s);
        else if (pack == 2) rc = launch_fast<3, FAST_D_H3, 2>(p, pt, d_tab, recs, cap, n_total, frame_count, s);
        else rc = launch_fast<3, FAST_D_H3>(p, pt, d_tab, recs, cap, n_total, frame_count, s);
        break;
    case 4:
        if (pack == 4) rc = launch_fast<4, 4, 4>(p, pt, d_tab, recs, cap, n_total, frame_count, s);
        else if (pack == 2) rc = launch_fast<4, 4, 2>(p, pt, d_tab, recs, cap, n_total, frame_count, s);
        else rc = launch_fast<4, FAST_D_H4>(p, pt, d_tab, recs, cap, n_total, frame_count, s);
        break;
    case 5:
        if (pack == 2) rc = launch_fast<5, 2, 2>(p, pt, d_tab, recs, cap, n_total, frame_count, s);
        else rc = launch_fast<5, FAST_D_H5>(p, pt, d_tab, recs, cap, n_total, frame_count, s);
        break;
    case 6:
        if (pack == 2) rc = launch_fast<6, 3, 2>(p, pt, d_tab, recs, cap, n_total, frame_count, s);
        else rc = launch_fast<6, FAST_D_H6>(p, pt, d_tab, recs, cap, n_total, frame_count, s);
        break;
    // boxes 15 and 17 (round 6): seven / eight rows in flight instead of two — with D = H the rows in flight and the pixel history
    // are one ring again and the kernel needs no more registers than before (215 / 240; its peak is the exact stage); at two
    // waves per SIMD and two rows each a CU had 16 KB of loads in flight, a third of what the memory system needs
    case 7: rc = launch_fast<7, 7>(p, pt, d_tab, recs, cap, n_total, frame_count, s); break;
    default: rc = launch_fast<8, 8>(p, pt, d_tab, recs, cap, n_total, frame_count, s); break;
    }
    if (rc == PMI_OK) *handled = true;
    return rc;
}

}  // namespace pmi
