// gaussmle_strict.hip — the MLE Newton loop in the reference's own arithmetic.
//
// picasso/gaussmle.py runs under numba: theta, the per-pixel derivative arrays and the
// numerator / denominator accumulators are float32 ARRAYS, every scalar intermediate is
// float64 (int64 pixel index - float32 theta -> float64), and the pixel loop is sequential
// (ii = column outer, jj = row inner, :798-839).  The fast kernels (gaussmle_g8.hip,
// gaussmle.hip) run that loop in float32; their result agrees to ~1e-5 px, but the
// convergence test |delta| < eps (:844-852 / :632-638) is a discrete decision, and when a
// step lands within rounding distance of eps the two arithmetics stop an iteration apart.
// The fast kernels therefore FLAG every spot whose decisive step came within a relative
// margin of eps (and every spot that ran into max_it), and this kernel re-fits the flagged
// spots from their initial parameters exactly as the reference does:
//
//   * float64 erf / exp / divisions, float32 rounding at every point where the reference
//     stores into a float32 array (dudt, d2udt2, numerator, denominator, theta);
//   * no contraction of a*b+c (numba without fastmath never fuses);
//   * the accumulators are advanced pixel by pixel in the reference's (ii, jj) order —
//     one lane per accumulator walks the per-pixel terms that the other lanes computed.
//
// The transcendental work is still separable: every function of (pixel index, mu, sigma)
// the reference evaluates per pixel is a deterministic function of the pixel BOUNDARY
// values (d - 1/2, d + 1/2), so it is evaluated once per boundary (phase A), combined per
// column / row (phase B) and per pixel (phase C) with the reference's operation order, which
// gives the same bits as evaluating it B^2 times.  (d + 1/2 of pixel k and d - 1/2 of pixel
// k + 1 are the same float64 whenever k - mu is exact, i.e. |mu| >= 2^-24 or mu = 0; a group
// with a coordinate in between evaluates the two boundaries of every pixel separately, see
// `split` in phase A.)  erf and exp themselves: with the bits of the reference's C library
// (libm_glibc.h) or the device library's, per launch (FitParams::libm_glibc; pmi_mle_set_libm:
// by default glibc's for every spot of the strict mode and in the lists of boxes up to 5x5).
//
// Mapping: a group of GS lanes per spot — GS = 16 / 32 / 64 (four / two / one spot per
// wavefront) for boxes <= 7 / <= 15 / larger, for the flagged-spot list and for PMI_MLE_STRICT
// alike: the kernel is bound by instruction issue, and the list's launch by its longest fits
// (launch_fit_strict: 64-lane groups for the list were measured slower, 0.45 against 0.30 ms).
// PMI_MLE_STRICT (the whole batch) runs mle_strict_start_kernel (start values, the groups side by side) and then the Newton
// loop with every group taking its next spot as soon as its fit has ended (REFILL); a list keeps its groups in lockstep.
#include <algorithm>
#include <cstdlib>

#include "fit_common.h"

#pragma clang fp contract(off)

#include "libm_glibc.h"

namespace pmi {

namespace {

// 2^(k/128) for pmi_glibc::exp
__constant__ uint64_t k_glibc_exp_tab[PMI_GLIBC_EXP_TABLE_WORDS] = {
#include "libm_glibc_exp_table.inc"
};

typedef double d2_t __attribute__((ext_vector_type(2)));
constexpr double K_SQRT_2PI = 2.5066282746310002;   // np.sqrt(2.0 * np.pi)
constexpr double K_SQRT_2 = 1.4142135623730951;
constexpr double K_SQRT_PI = 1.7724538509055159;

__device__ __forceinline__ double np_max_d(double a, double b) { return (a != a) ? a : ((b != b) ? b : (a > b ? a : b)); }
__device__ __forceinline__ double np_min_d(double a, double b) { return (a != a) ? a : ((b != b) ? b : (a < b ? a : b)); }
// float32 / float32, correctly rounded (float64 quotient rounds innocuously: 53 >= 2*24 + 2)
__device__ __forceinline__ float div_rn(float a, float b) { return (float)((double)a / (double)b); }

// n / d for a divisor all pixels of an iteration share, r = RN(1 / d): q = RN(n r) made faithful by one residual correction and
// correctly rounded by a second (Markstein: with r the correctly rounded reciprocal, q faithful and the residual n - q d exact,
// RN(q + (n - q d) r) is RN(n / d)) — five multiply-adds against a v_rcp_f64 (quarter rate), two v_div_scale, five fused
// multiply-adds, v_div_fmas and v_div_fixup.  Exactness of the residuals needs n, q and n - q d inside the exponent range: the
// caller takes this path only when d is in [2^-92, 2^93] and n is zero or in [2^-400, 2^530] (mid_or_zero on its factors).
// A zero numerator may come out as +0 where the division gives -0: every use of these quotients multiplies or adds them into
// float32 accumulators that never hold -0, so the sign of a zero is never seen.
__device__ __forceinline__ double shared_div(double n, double d, double r)
{
    double q = n * r;
    double e = __builtin_fma(-q, d, n);
    q = __builtin_fma(e, r, q);
    e = __builtin_fma(-q, d, n);
    return __builtin_fma(e, r, q);
}
// zero, or an exponent in [2^-200, 2^200) (NaN and infinities fail)
__device__ __forceinline__ bool mid_or_zero(double v)
{
    const unsigned e = ((unsigned)__double2hiint(v) >> 20) & 0x7ffu;
    return (e - 823u) < 400u || v == 0.0;
}

__device__ __forceinline__ void lds_sync()
{
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
}

// LDS of one group, in bytes: spot (float32), boundary values, column / row terms, per-pixel terms of one round,
// accumulators.  Round 6: every array is a structure of arrays — field q of record r at [q * stride + r] — so that the lanes
// of a group, which work on consecutive records, touch consecutive 8-byte slots whatever the field (rounds 2 - 5 kept the
// records together: 32-, 40- and 96-byte lane strides, SQ_LDS_BANK_CONFLICT 25 % of the LDS cycles of the all-strict launch).
// What is left to choose is where the NEXT group of the wavefront starts (two 16-lane groups share a 32-lane ds_read_b64,
// the chain lanes of two groups one ds_read_b128) and the row stride RS of the term array, whose rows the chain lanes read
// 16 bytes at a time: tools/emul/lds_strict_layout.py (the bank model that reproduced the least-squares kernel's counters)
// gives 1.51x the conflict-free cycles for the old layout at 7x7 and 1.06x for RS = 20 with the two groups of a pair 3408 B
// apart (80 bytes past a multiple of 256).
template <int GS> struct SLds {
    static constexpr int MAXB = GS == 16 ? 7 : (GS == 32 ? 15 : PMI_MAX_BOX);
    static constexpr int SPOT = ((MAXB * MAXB * 4 + 15) / 16) * 16;
    static constexpr int BS = 2 * (MAXB + 1);            // boundary records per field: (axis, k), k <= B
    static constexpr int CS = 2 * MAXB;                  // column / row records per field: (axis, i)
    static constexpr int RS = GS + (GS == 16 ? 4 : 2);   // slots per row of the term array (one row per accumulator)
    static constexpr int BND = 4 * BS * 8;
    static constexpr int COL = 5 * CS * 8;
    static constexpr int TERM = 12 * RS * 8;
    static constexpr int ACC = 24 * 4;      // 12 accumulators + 6 updated parameters
    static constexpr int BYTES = SPOT + BND + COL + TERM + ACC;
    // the second group of a 32-lane pair starts PAIR_PAD bytes later than BYTES would put it (16-lane groups only: with 32 or
    // 64 lanes a group has the LDS's lane groups to itself): group g of a wavefront at g * BYTES + ((g + 1) / 2) * PAIR_PAD
    static constexpr int PAIR_PAD = GS == 16 ? 112 : 0;
    static constexpr int NSPW = 64 / GS;
    static constexpr int WAVE_BYTES = NSPW * BYTES + (NSPW / 2) * PAIR_PAD;
    static_assert(BYTES % 16 == 0 && PAIR_PAD % 16 == 0 && (BS * 8) % 16 == 0 && (CS * 8) % 16 == 0 && (RS * 8) % 16 == 0, "LDS layout of a group");
    static_assert(GS != 16 || (BYTES + PAIR_PAD) % 256 == 80, "the chain lanes of a pair of groups share a ds_read_b128: rows 0..3 of one beside rows 4..11 of the other");
    // three workgroups of FIT_WAVES wavefronts per CU: 160 KB in granules of 1280 B (with groups a uniform 3408 B apart the
    // workgroup took 43 granules and only two fitted: the flag list of eps 1e-4 ran 13 % slower)
    static_assert(3 * ((FIT_WAVES * WAVE_BYTES + 4 + 1279) / 1280) * 1280 <= 160 * 1024, "three workgroups per CU");
};

}  // namespace

// ---- the spot, in photons (localize.py:917-931, 1101-1112), into the group's LDS ----
template <int GS, bool FROM_MOVIE>
__device__ __forceinline__ void strict_load_spot(const FitParams &p, int64_t sidx, bool have, int j, float *spot)
{
    const int B = p.box, npix = B * B, H = B / 2;
    int64_t base = 0;
    if (FROM_MOVIE && have) base = (p.frame[sidx] * p.Y + (p.y[sidx] - H)) * p.X + (p.x[sidx] - H);
    // (eight loads in flight per lane: one per trip to memory made the one-lane-per-spot start kernel wait 49 times in a row)
    for (int q0 = j; q0 < npix; q0 += 8 * GS) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int q = q0 + u * GS;
            v[u] = 0.f;
            if (have && q < npix) {
                if (FROM_MOVIE) {
                    const int a = q / B, c = q - a * B;
                    v[u] = load_movie_px(p.movie, p.dtype, base + (int64_t)a * p.X + c);
                } else {
                    v[u] = p.spots[sidx * npix + q];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int q = q0 + u * GS;
            if (q < npix) spot[q] = (FROM_MOVIE && have) ? div_const((v[u] - p.baseline) * p.sensitivity, p.gdiv) : v[u];
        }
    }
    lds_sync();
}

// ---- initial parameters (gaussmle.py:28-168) of the spot in the group's LDS; fscr: GS floats of the group ----
template <int NP, int GS>
__device__ __forceinline__ void strict_start_values(int B, int j, const float *spot, float *fscr, float (&th)[6])
{
    const int npix = B * B, H = B / 2;
    // 3x3 edge-clipped mean filter, float64 sum in (m, n) order, float32 store (:61-91); its minimum (:135)
    float best = INFINITY;
    for (int q = j; q < npix; q += GS) {
        const int k = q / B, l = q - k * B;
        const int m0 = k - 1 > 0 ? k - 1 : 0, m1 = k + 2 < B ? k + 2 : B;
        const int n0 = l - 1 > 0 ? l - 1 : 0, n1 = l + 2 < B ? l + 2 : B;
        double nsum = 0.0;
        for (int m = m0; m < m1; m++)
            for (int c = n0; c < n1; c++) nsum += (double)spot[m * B + c];
        const float f = (float)(nsum / (double)((m1 - m0) * (n1 - n0)));
        best = (best != best) ? best : ((f != f) ? f : (f < best ? f : best));        // np.min: NaN propagates
    }
    fscr[j] = best;
    lds_sync();
    float bg0 = INFINITY;
    for (int r = 0; r < GS; r++) {
        const float f = fscr[r];
        bg0 = (bg0 != bg0) ? bg0 : ((f != f) ? f : (f < bg0 ? f : bg0));
    }
    // sum and centre of mass, sequential float64 like the reference (:28-48); every lane does the same
    double s_ = 0.0, sy_ = 0.0, sx_ = 0.0;
    for (int a = 0; a < B; a++)
        for (int c = 0; c < B; c++) {
            const double v = (double)spot[a * B + c];
            sy_ += v * (double)a;
            sx_ += v * (double)c;
            s_ += v;
        }
    double com_y, com_x;
    if (s_ <= 0.0) { s_ = 0.01; com_y = (B - 1) / 2.0; com_x = (B - 1) / 2.0; }
    else { com_y = sy_ / s_; com_x = sx_ / s_; }
    const double photons0 = np_max_d(1.0, s_ - (double)(B * B) * (double)bg0);
    double sdy = 0.0, sdx = 0.0, sum_y = 0.0, sum_x = 0.0;
    for (int a = 0; a < B; a++) {
        const double d2 = (double)((a - H) * (a - H));
        const float vy = spot[a * B + H] - bg0;        // spot - bg is a float32 array (:105)
        const float vx = spot[H * B + a] - bg0;
        sdy += (double)vy * d2;
        sdx += (double)vx * d2;
        sum_y += (double)vy;
        sum_x += (double)vx;
    }
    double isy = sqrt(sdy / sum_y), isx = sqrt(sdx / sum_x);
    if (!isfinite(isy)) isy = 0.01;
    if (!isfinite(isx)) isx = 0.01;
    if (isx == 0) isx = 0.01;
    if (isy == 0) isy = 0.01;

    th[0] = (float)com_x; th[1] = (float)com_y; th[2] = (float)photons0; th[3] = bg0;
    if (NP == 6) { th[4] = (float)isx; th[5] = (float)isy; }
    else { th[4] = (float)((isx + isy) / 2); th[5] = 0.f; }                  // float64 mean, float32 store (:153)
    lds_sync();             // (fscr may be written again)
}

// The start values of every spot of the batch into p.thetas (slot 5 of the `sigma` method: a copy of slot 4, as in the fitted
// rows) — for mle_strict_kernel<.., REFILL = true>.  The sums of the start values are sequential (every lane of a group
// repeats them), so the groups are as small as the LDS allows: one LANE per spot for boxes up to 7x7 (64 spots per
// wavefront: 0.53 ms for config 2's million spots against 0.75 with 16-lane groups — what is left is the gather of seven
// movie rows per spot; building the kernel for the box bought 5 %), 4 lanes up to 15x15, 16 above.
template <int NP, int GS, bool FROM_MOVIE>
__global__ __launch_bounds__(FIT_NT) void mle_strict_start_kernel(FitParams p)
{
    constexpr int NSPW = 64 / GS;
    extern __shared__ __attribute__((aligned(16))) float s_start[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int g = lane / GS, j = lane & (GS - 1);
    const int npix = p.box * p.box, stride = (npix + GS) | 1;           // (odd: the lanes of a wavefront on different banks)
    float *spot = s_start + (size_t)(wid * NSPW + g) * stride;
    float *fscr = spot + npix;
    int64_t n = p.N;
    if (p.d_n) { int64_t dn = *p.d_n; n = dn < n ? dn : n; }
    const int64_t items = n - p.first;
    const int64_t total_groups = (int64_t)gridDim.x * FIT_WAVES * NSPW;
    for (int64_t w0 = ((int64_t)blockIdx.x * FIT_WAVES + wid) * NSPW; w0 < items; w0 += total_groups) {
        const int64_t w = w0 + g;
        const bool have = w < items;
        const int64_t sidx = p.first + (have ? w : 0);
        strict_load_spot<GS, FROM_MOVIE>(p, sidx, have, j, spot);
        float th[6];
        strict_start_values<NP, GS>(p.box, j, spot, fscr, th);
        if (have && j == 0) {
            float *to = p.thetas + sidx * 6;
#pragma unroll
            for (int l = 0; l < 5; l++) to[l] = th[l];
            to[5] = NP == 6 ? th[5] : th[4];
        }
        lds_sync();
    }
}

template <int NP, int GS>
static void launch_strict_start(const FitParams &p, bool from_movie, int64_t max_items, int cu_count, hipStream_t s)
{
    constexpr int NSPW = 64 / GS;
    const int64_t groups_per_block = (int64_t)FIT_WAVES * NSPW;
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((max_items + groups_per_block - 1) / groups_per_block, (int64_t)cu_count * 8));
    const size_t lds = (size_t)groups_per_block * (size_t)((p.box * p.box + GS) | 1) * sizeof(float);
    if (from_movie) hipLaunchKernelGGL((mle_strict_start_kernel<NP, GS, true>), dim3((unsigned)blocks), dim3(FIT_NT), lds, s, p);
    else hipLaunchKernelGGL((mle_strict_start_kernel<NP, GS, false>), dim3((unsigned)blocks), dim3(FIT_NT), lds, s, p);
}

// list: spot indices to fit (entries [0, *list_n)), or nullptr = every spot of [p.first, min(p.N, *p.d_n)).
template <int NP, int GS, bool FROM_MOVIE, bool REFILL>
__global__ __launch_bounds__(FIT_NT, 3) void mle_strict_kernel(FitParams p, const int32_t *__restrict__ list,
                                                            const unsigned *__restrict__ list_n)
{
    constexpr int NSPW = 64 / GS;
    __shared__ __attribute__((aligned(16))) char s_mem[FIT_WAVES][SLds<GS>::WAVE_BYTES];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int g = lane / GS, j = lane & (GS - 1);
    char *mem = &s_mem[wid][0] + g * SLds<GS>::BYTES + ((g + 1) >> 1) * SLds<GS>::PAIR_PAD;
    float *spot = reinterpret_cast<float *>(mem);
    double *bnd = reinterpret_cast<double *>(mem + SLds<GS>::SPOT);
    double *col = reinterpret_cast<double *>(mem + SLds<GS>::SPOT + SLds<GS>::BND);
    double *term = reinterpret_cast<double *>(mem + SLds<GS>::SPOT + SLds<GS>::BND + SLds<GS>::COL);
    float *accs = reinterpret_cast<float *>(mem + SLds<GS>::SPOT + SLds<GS>::BND + SLds<GS>::COL + SLds<GS>::TERM);

    constexpr int BS = SLds<GS>::BS, CS = SLds<GS>::CS, RS = SLds<GS>::RS;
    const int B = p.box, npix = B * B;
    int64_t n = p.N;
    if (p.d_n) { int64_t dn = *p.d_n; n = dn < n ? dn : n; }
    int64_t items = list ? (int64_t)*list_n : n - p.first;
    if (list && items > n - p.first) items = n - p.first;
    // REFILL (the whole batch, PMI_MLE_STRICT): every group of GS lanes works through spots of its own — when its fit has
    // converged it stores the result and takes the next spot while the other groups of the wavefront go on iterating (the
    // instruction stream of an iteration costs the same with one group active as with four: groups in lockstep ran
    // max-of-four iterations per spot, 1.5x the mean on config 2: 13.1 -> 10.9 ms).  A workgroup owns a contiguous share of
    // the batch and its groups take the spots of the share one at a time from a counter in LDS.
    // !REFILL (a list of flagged spots): the wavefront takes NSPW new entries when the LAST of its fits has ended.  The flagged
    // fits are long and alike, and a refill runs the start values for one group with the others masked — measured on one box
    // (tuning build, PMI_STRICT_LIST_REFILL): config 2 2.66 - 2.73 (lockstep) against 2.77 - 2.79 ms, eps 1e-4 and config 5 the same.
    // A list holds fits of 10 and of 100 iterations: dealt round robin, the groups with one entry more
    // than the others, or with two long fits, end the launch long after the rest (13x13, 26 000 entries over 6 144 groups:
    // 2.9 ms).  So only the first round is dealt; after it a wavefront takes its next NSPW entries from a queue word when
    // it is done with the last.  (Every wavefront asking the queue at its START was measured slower on short lists — 7 346
    // entries, config 2: 142 instead of 94 us — 3 072 atomics on one word from eight XCDs complete one after the other.)
    // ... and takes several rounds' worth of entries per visit when the list is long: the queue is ONE word, and the atomics
    // of three thousand wavefronts on it complete one after the other (~0.1 us each) — at eps 1e-4 config 2's list is
    // 113 000 entries = 28 000 visits of four, 2.8 of the launch's 3.4 ms; config 5's two lists of 26 000 13x13 spots paid
    // 1 ms each.  Up to eight rounds per visit, never more than half of what a wavefront's fair share would be.
    __shared__ unsigned s_next;
    if (threadIdx.x == 0) s_next = 0u;
    __syncthreads();
    const int64_t blk_lo = items * (int64_t)blockIdx.x / (int64_t)gridDim.x;
    const int64_t blk_hi = items * ((int64_t)blockIdx.x + 1) / (int64_t)gridDim.x;
    const int64_t total_groups = (int64_t)gridDim.x * FIT_WAVES * NSPW;
    unsigned *qw = list ? p.strict_queue : nullptr;
    const int64_t fair = items / (2 * total_groups);
    const unsigned chunk = (unsigned)NSPW * (unsigned)(fair < 1 ? 1 : (fair > 8 ? 8 : fair));
    int64_t w0 = ((int64_t)blockIdx.x * FIT_WAVES + wid) * NSPW - NSPW;  // wave-uniform: the entries before the first round's
    int64_t w_lim = w0 + 2 * NSPW;                                       // end of the entries this wavefront holds

    const int nb = B + 1;
    const bool glibc = p.libm_glibc != 0;          // erf / exp with glibc's bits (the reference's C library) or the device library's
    const int j0_ii = j / B, j0_jj = j - j0_ii * B, gs_ii = GS / B, gs_jj = GS - gs_ii * B;
    bool have = false, active = false, exhausted = false;
    int64_t sidx = 0;
    int kk = 0;
    float th[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ms[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto store_fit = [&]() {
        if (j == 0) {
            float *to = p.thetas + sidx * 6;
#pragma unroll
            for (int l = 0; l < 5; l++) to[l] = th[l];
            to[5] = NP == 6 ? th[5] : th[4];
            p.iterations[sidx] = kk;
            if (p.refit_mark) p.refit_mark[sidx - p.first] = 1;
        }
    };
    for (;;) {
        // ---- groups without a running fit take entries until one has iterations to do (max_it <= 0: none has) ----
        while ((REFILL ? !active : !__any(active)) && !exhausted) {
        int64_t w;
        if (REFILL) {
            unsigned tk = 0u;
            if (j == 0) tk = atomicAdd(&s_next, 1u);
            tk = (unsigned)__shfl((int)tk, lane & ~(GS - 1));
            w = blk_lo + (int64_t)tk;
            have = w < blk_hi;
            if (!have) { exhausted = true; break; }
        } else {
            w0 += NSPW;
            if (w0 >= w_lim) {
                if (qw) {
                    unsigned t = 0;
                    if (lane == 0) t = atomicAdd(qw, chunk);
                    w0 = total_groups + (int64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)t);
                    w_lim = w0 + chunk;
                } else {
                    w0 += total_groups - NSPW;
                    w_lim = w0 + NSPW;
                }
            }
            if (w0 >= items) { exhausted = true; break; }
            w = w0 + g;
            have = w < items;
        }
        if (have) {
        sidx = list ? (int64_t)list[w] : p.first + w;

        strict_load_spot<GS, FROM_MOVIE>(p, sidx, true, j, spot);
        if (REFILL && !list) {
            // (the start values of the batch were computed by mle_strict_start_kernel, four spots per wavefront side by side:
            // here they would run for this group alone with the rest of the wavefront masked)
            const float *from = p.thetas + sidx * 6;
#pragma unroll
            for (int l = 0; l < 5; l++) th[l] = from[l];
            th[5] = NP == 6 ? from[5] : 0.f;
        } else {
            strict_start_values<NP, GS>(p.box, j, spot, reinterpret_cast<float *>(term), th);
        }
        ms[0] = th[4]; ms[1] = th[4];
        ms[2] = (float)(0.1 * (double)th[2]); ms[3] = (float)(0.1 * (double)th[3]);
        ms[4] = (float)(0.2 * (double)th[4]); ms[5] = (float)(0.2 * (double)th[5]);

        kk = 0;
        active = p.max_it > 0;
        if (!active) store_fit();
        }
        }
        if (!__any(active)) break;
        // ---- one Newton iteration of every group with a running fit (:533-670 sigma, :745-884 sigmaxy) -----------------
        {
            const float sgy = NP == 6 ? th[5] : th[4];
            // phase A: lane -> (axis, boundary k)
            // A coordinate closer to zero than 2^-24 (and not zero) has bits below the last place of k - mu: (k - mu) + 1/2 and
            // ((k + 1) - mu) - 1/2 then round differently, and the plus boundary of pixel k is no longer the minus boundary of
            // pixel k + 1.  Such a group runs phases A and B twice, for the even and for the odd pixels: in pass p the records
            // of parity p are minus boundaries and each next one is the plus boundary of ITS pixel, so phase B still reads a
            // pixel's two boundaries from records i and i + 1.  (A centre that sits on the first pixel's middle to 1e-17 px is
            // a fuzzer's spot — but its bits are the reference's too.  The groups beside it in the wavefront repeat their pass.)
            const bool split = (th[0] != 0.f && fabsf(th[0]) < 0x1p-24f) || (th[1] != 0.f && fabsf(th[1]) < 0x1p-24f);
            const int npass = __any(split) ? 2 : 1;
            bool rec_ok = true;                                  // the factors of this lane's records are zero or of middling size (shared_div)
            for (int pass = 0; pass < npass; pass++) {
            for (int ja = j; ja < 2 * nb; ja += GS) {            // (one pass unless the group is narrower than 2 (B + 1) lanes)
                const int a = ja >= nb ? 1 : 0, k = ja - a * nb;
                const double dmu = (double)(a ? th[1] : th[0]);
                const float sgf = a ? sgy : th[4];
                const double ds = (double)sgf;
                const bool plus_rec = split ? ((k & 1) != pass) : (k == B);          // the plus boundary of pixel k - 1
                const double b = plus_rec ? ((double)(k - 1) - dmu) + 0.5 : ((double)k - dmu) - 0.5;
                const double sq_norm = 0.70710678118654757 / ds;                     // :276
                const double t = b / ds;
                double eA, eD;
                if (glibc) {
                    eA = pmi_glibc::erf(b * sq_norm, k_glibc_exp_tab);               // :279
                    eD = pmi_glibc::exp(-0.5 * (t * t), k_glibc_exp_tab);            // :294-295
                } else {
                    eA = erf(b * sq_norm);
                    eD = exp(-0.5 * (t * t));
                }
                double v2, v3;
                if (NP == 6) {
                    const float s2 = sgf * sgf;
                    const double aG = -(b * b) / (2.0 * (double)s2);
                    const double eG = glibc ? pmi_glibc::exp(aG, k_glibc_exp_tab) : exp(aG);      // :312-313
                    v2 = b * eG;                                                     // a ** 1 * exp
                    v3 = (b * (b * b)) * eG;                                         // a ** 3 * exp (square-and-multiply)
                } else {
                    const double ai = b / (K_SQRT_2 * ds);                           // :354-357
                    const double ex = glibc ? pmi_glibc::exp(-(ai * ai), k_glibc_exp_tab) : exp(-(ai * ai));
                    v2 = ai * ex;                                                    // term of Fx / Fy (:359)
                    v3 = (ai * ex) * (1.0 - 2.0 * (ai * ai));                        // :364-371
                }
                double *o = bnd + ja;                                                // record (a, k) = a * nb + k, field q at o[q * BS]
                o[0] = eA; o[BS] = eD; o[2 * BS] = v2; o[3 * BS] = v3;
            }
            lds_sync();
            // phase B: lane -> (axis, pixel index i): PSF, b - a, (d-.5) b - (d+.5) a, sigma terms
            for (int jb = j; jb < 2 * B; jb += GS) {
                const int a = jb >= B ? 1 : 0, i = jb - a * B;
                const double dmu = (double)(a ? th[1] : th[0]);
                const float sgf = a ? sgy : th[4];
                const double ds = (double)sgf;
                const double *mq = bnd + (a * nb + i);                               // minus boundary; the plus boundary is the next record
                const double m[4] = {mq[0], mq[BS], mq[2 * BS], mq[3 * BS]}, q[4] = {mq[1], mq[BS + 1], mq[2 * BS + 1], mq[3 * BS + 1]};
                const bool mine = !split || (i & 1) == pass;                         // (a split group: the pixels of this pass's parity)
                const double d = (double)i - dmu;
                const double PSF = 0.5 * (q[0] - m[0]);
                const double bma = m[1] - q[1];
                const double qq = (d - 0.5) * m[1] - (d + 0.5) * q[1];
                rec_ok = rec_ok && (!mine || (mid_or_zero(PSF) && mid_or_zero(bma) && mid_or_zero(qq)));
                double S1, S2;
                if (NP == 6) {
                    const float s2 = sgf * sgf, s3 = sgf * s2, s5 = sgf * (s2 * s2);   // sigma ** n, float32 (:315)
                    const double g21 = (m[2] - q[2]) / ((double)s2 * K_SQRT_2PI);
                    const double g31 = (m[2] - q[2]) / ((double)s3 * K_SQRT_2PI);
                    const double g53 = (m[3] - q[3]) / ((double)s5 * K_SQRT_2PI);
                    S1 = g21;                                                        // :330
                    S2 = g53 - 2.0 * g31;                                            // :334
                } else {
                    const double F = m[2] - q[2];                                    // :359
                    const double dPSF = F / (K_SQRT_PI * ds);                        // :361
                    const double dF = (q[3] - m[3]) / ds;                            // :364-367
                    const float s2 = sgf * sgf;
                    const float sinv = div_rn(1.0f, sgf);                            // sigma ** (-1), float32
                    S1 = dPSF;
                    S2 = (1.0 / K_SQRT_PI) * ((-F / (double)s2) + (double)sinv * dF);   // :372-374
                }
                if (mine) {
                    double *o = col + jb;                                            // record (a, i) = a * B + i
                    o[0] = PSF; o[CS] = bma; o[2 * CS] = qq; o[3 * CS] = S1; o[4 * CS] = S2;
                }
            }
            lds_sync();
            }
            // phase C + accumulation, GS pixels of the reference's (ii, jj) sequence per round
            float acc = 0.f;
            const double N_ = (double)th[2], bgd = (double)th[3];
            const float sgx = th[4];
            const double cx = K_SQRT_2PI * (double)sgx, cy = K_SQRT_2PI * (double)sgy;
            const double c3x = K_SQRT_2PI * (double)(sgx * (sgx * sgx)), c3y = K_SQRT_2PI * (double)(sgy * (sgy * sgy));
            // The four divisions by cx, cy, c3x, c3y of every pixel (gaussmle.py:296-302) share their divisors: reciprocal +
            // two corrections (shared_div) when every factor of the group's numerators and the divisors are inside the range
            // its proof needs — sigma in (2^-30, 2^30), photons finite, the column / row records zero or in [2^-200, 2^200) —,
            // the plain divisions otherwise (a collapsed or exploding fit).  The same bits either way.
            const unsigned long long okb = __ballot(rec_ok);
            const unsigned long long gmask = GS == 64 ? ~0ull : (((1ull << (GS & 63)) - 1ull) << (lane & ~(GS - 1)));
            const bool fast_div = (okb & gmask) == gmask && th[2] < 3.0e38f && sgx < 0x1p30f && sgy < 0x1p30f && sgx > 0x1p-30f && sgy > 0x1p-30f;
            const double rcx = 1.0 / cx, rcy = 1.0 / cy, rc3x = 1.0 / c3x, rc3y = 1.0 / c3y;
            int ii = j0_ii, jj = j0_jj;                    // (ii, jj) of pixel r0 + j of the sequence, stepped without a division
            for (int r0 = 0; r0 < npix; r0 += GS, ii += gs_ii, jj += gs_jj) {
                const int seq = r0 + j;
                if (jj >= B) { jj -= B; ii++; }
                if (seq < npix) {
                    const double *Xp = col + ii, *Yp = col + (B + jj);
                    const double X[5] = {Xp[0], Xp[CS], Xp[2 * CS], Xp[3 * CS], Xp[4 * CS]}, Yc[5] = {Yp[0], Yp[CS], Yp[2 * CS], Yp[3 * CS], Yp[4 * CS]};
                    const double PSFx = X[0], PSFy = Yc[0];
                    float du[6], d2[6];
                    if (fast_div) {
                        du[0] = (float)shared_div(N_ * PSFy * X[1], cx, rcx);        // :296
                        d2[0] = (float)shared_div(N_ * X[2] * PSFy, c3x, rc3x);      // :297-302
                        du[1] = (float)shared_div(N_ * PSFx * Yc[1], cy, rcy);
                        d2[1] = (float)shared_div(N_ * Yc[2] * PSFx, c3y, rc3y);
                    } else {
                        du[0] = (float)(N_ * PSFy * X[1] / cx);
                        d2[0] = (float)(N_ * X[2] * PSFy / c3x);
                        du[1] = (float)(N_ * PSFx * Yc[1] / cy);
                        d2[1] = (float)(N_ * Yc[2] * PSFx / c3y);
                    }
                    du[2] = (float)(PSFx * PSFy); d2[2] = 0.f;
                    du[3] = 1.f; d2[3] = 0.f;
                    if (NP == 6) {
                        du[4] = (float)(N_ * PSFy * X[3]);                           // :330-335
                        d2[4] = (float)(N_ * PSFy * X[4]);
                        du[5] = (float)(N_ * PSFx * Yc[3]);
                        d2[5] = (float)(N_ * PSFx * Yc[4]);
                    } else {
                        du[4] = (float)(N_ * (PSFy * X[3] + PSFx * Yc[3]));          // :379
                        d2[4] = (float)(N_ * PSFy * X[4] + 2.0 * X[3] * Yc[3] + PSFx * Yc[4]);   // :380-382
                        du[5] = 0.f; d2[5] = 0.f;
                    }
                    const double model = N_ * PSFx * PSFy + bgd;                     // :828
                    const double data = (double)spot[jj * B + ii];
                    double cf = 0.0, df = 0.0;
                    if (model > 10e-3) { cf = data / model - 1; df = data / (model * model); }
                    cf = np_min_d(cf, 10e4);
                    df = np_min_d(df, 10e4);
                    double *t = term + j;                                            // row l: the terms of accumulator l, slot j: this pixel
#pragma unroll
                    for (int l = 0; l < 6; l++) {
                        const float du2 = du[l] * du[l];                             // float32 ** 2
                        t[l * RS] = cf * (double)du[l];
                        t[(6 + l) * RS] = cf * (double)d2[l] - df * (double)du2;
                    }
                }
                lds_sync();
                if (j < 12) {
                    // :838-839, one accumulator per lane; eight terms are fetched ahead of the dependent chain of
                    // (float64 add, float32 round) steps
                    const int cnt = npix - r0 < GS ? npix - r0 : GS;
                    const double *row = term + j * RS;
                    if (cnt == GS) {
#pragma unroll
                        for (int s0 = 0; s0 < GS; s0 += 16) {
                            d2_t t[8];                                               // sixteen terms in eight 16-byte reads
#pragma unroll
                            for (int u = 0; u < 8; u++) t[u] = *reinterpret_cast<const d2_t *>(row + s0 + 2 * u);
#pragma unroll
                            for (int u = 0; u < 8; u++) { acc = (float)((double)acc + t[u].x); acc = (float)((double)acc + t[u].y); }
                        }
                    } else {
                        for (int s = 0; s < cnt; s++) acc = (float)((double)acc + row[s]);
                    }
                }
                lds_sync();
            }
            if (j < 12) accs[j] = acc;
            lds_sync();
            // update (:860-884 / :647-670): lane l updates parameter l, the six new values go round through LDS
            if (j < NP) {
                const float num = accs[j], den = accs[6 + j];
                float msl = ms[0], thl = th[0];
#pragma unroll
                for (int l = 1; l < 6; l++) { msl = j == l ? ms[l] : msl; thl = j == l ? th[l] : thl; }
                float upd;
                if (NP == 6) upd = den == 0.0f ? np_signf(num) * msl : np_minf(np_maxf(div_rn(num, den), -msl), msl);
                else upd = den == 0.0f ? np_signf(num * msl) : np_minf(np_maxf(div_rn(num, den), -msl), msl);
                float v = thl - upd;
                if (j == 2) v = np_maxf(v, 1.0f);
                if (j == 3 || j == 4 || j == 5) v = np_maxf(v, 0.01f);
                if (NP == 5 && j == 4) v = np_minf(v, (float)B);
                accs[12 + j] = v;
            }
            lds_sync();
            float nt[6];
#pragma unroll
            for (int l = 0; l < NP; l++) nt[l] = accs[12 + l];
            if (NP == 5) nt[5] = th[5];
            bool conv = ((double)fabsf(th[0] - nt[0]) < p.eps) && ((double)fabsf(th[1] - nt[1]) < p.eps);
            if (NP == 6) conv = conv && ((double)fabsf(th[4] - nt[4]) < p.eps) && ((double)fabsf(th[5] - nt[5]) < p.eps);
            if (active) {
#pragma unroll
                for (int l = 0; l < 6; l++) th[l] = nt[l];
                kk++;
                if (conv || kk >= p.max_it) active = false;
            }
            if (have && !active) { store_fit(); have = false; }
            lds_sync();
        }
    }
}

template <int NP, int GS>
static void launch_strict_gs(const FitParams &p, bool from_movie, const int32_t *list, const unsigned *list_n,
                             int64_t max_items, int cu_count, hipStream_t s)
{
    constexpr int NSPW = 64 / GS;
    const int64_t groups_per_block = (int64_t)FIT_WAVES * NSPW;
    int64_t blocks = (max_items + groups_per_block - 1) / groups_per_block;
    // A flag list: three workgroups per CU, all resident, the entries past the first round handed out through the queue word
    // (more workgroups change nothing: 6.19 - 6.27 ms at eps 1e-4 with 3 ... 24 per CU).  The whole batch (REFILL): every
    // workgroup owns a fixed share, so the launch ends with its slowest share — twelve workgroups per CU instead of three
    // (nine of them waiting for a slot) let the hardware deal the shares: config 2, all strict, one box: 10.78 ms per step
    // with 3 per CU, 10.12 with 4, 9.78 with 6, 9.52 with 12, 9.58 / 9.71 / 9.97 / 10.5 with 16 / 24 / 32 / 48.
    const int bl_mult = list ? 3 : 12;
    blocks = std::max<int64_t>(1, std::min<int64_t>(blocks, (int64_t)cu_count * bl_mult));
    static const char *renv = tuning_env("PMI_STRICT_LIST_REFILL");      // tuning: the list's groups refill one by one too
    if (list && renv && atoi(renv)) {
        if (from_movie) hipLaunchKernelGGL((mle_strict_kernel<NP, GS, true, true>), dim3((unsigned)blocks), dim3(FIT_NT), 0, s, p, list, list_n);
        else hipLaunchKernelGGL((mle_strict_kernel<NP, GS, false, true>), dim3((unsigned)blocks), dim3(FIT_NT), 0, s, p, list, list_n);
    } else if (list) {
        if (from_movie) hipLaunchKernelGGL((mle_strict_kernel<NP, GS, true, false>), dim3((unsigned)blocks), dim3(FIT_NT), 0, s, p, list, list_n);
        else hipLaunchKernelGGL((mle_strict_kernel<NP, GS, false, false>), dim3((unsigned)blocks), dim3(FIT_NT), 0, s, p, list, list_n);
    } else {
        if (p.box <= 7) launch_strict_start<NP, 1>(p, from_movie, max_items, cu_count, s);
        else if (p.box <= 15) launch_strict_start<NP, 4>(p, from_movie, max_items, cu_count, s);
        else launch_strict_start<NP, 16>(p, from_movie, max_items, cu_count, s);
        if (from_movie) hipLaunchKernelGGL((mle_strict_kernel<NP, GS, true, true>), dim3((unsigned)blocks), dim3(FIT_NT), 0, s, p, list, list_n);
        else hipLaunchKernelGGL((mle_strict_kernel<NP, GS, false, true>), dim3((unsigned)blocks), dim3(FIT_NT), 0, s, p, list, list_n);
    }
}

// Newton loop in the reference's arithmetic for the spots of `list` (device indices, *list_n of them, at most
// max_items) or, with list == nullptr, for every spot of the batch.  Writes thetas and iterations only.
void launch_fit_strict(const FitParams &p, int method, bool from_movie, const int32_t *list, const unsigned *list_n,
                       int64_t max_items, int cu_count, hipStream_t s)
{
    // four / two / one spot per wavefront: the kernel is bound by instruction issue, not by the latency of one fit
    // (measured on config 2's 6 500 flagged spots: 16-lane groups 0.30 ms, 64-lane groups 0.45 ms; 13x13 on 16-lane groups —
    // four spots per wavefront but two wavefronts per SIMD instead of three for the LDS: 1.23 against 0.95 ms per list of config 5)
    static const char *genv = tuning_env("PMI_STRICT_LIST_GS");      // tuning: group size for the flagged-spot list
    const int packed = p.box <= 7 ? 16 : (p.box <= 15 ? 32 : 64);
    const int gs = list ? std::max(packed, genv ? atoi(genv) : packed) : packed;
    if (method == PMI_MLE_SIGMAXY) {
        if (gs == 16) launch_strict_gs<6, 16>(p, from_movie, list, list_n, max_items, cu_count, s);
        else if (gs == 32) launch_strict_gs<6, 32>(p, from_movie, list, list_n, max_items, cu_count, s);
        else launch_strict_gs<6, 64>(p, from_movie, list, list_n, max_items, cu_count, s);
    } else {
        if (gs == 16) launch_strict_gs<5, 16>(p, from_movie, list, list_n, max_items, cu_count, s);
        else if (gs == 32) launch_strict_gs<5, 32>(p, from_movie, list, list_n, max_items, cu_count, s);
        else launch_strict_gs<5, 64>(p, from_movie, list, list_n, max_items, cu_count, s);
    }
}

__global__ void libm_eval_kernel(int fn, const double *__restrict__ x, int64_t n, double *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    out[i] = fn == 0 ? pmi_glibc::exp(v, k_glibc_exp_tab) : (fn == 1 ? pmi_glibc::erf(v, k_glibc_exp_tab) : (fn == 2 ? exp(v) : erf(v)));
}

}  // namespace pmi

extern "C" int pmi_libm_eval_dev(int fn, const double *d_x, int64_t n, double *d_out, void *stream)
{
    using namespace pmi;
    if (fn < 0 || fn > 3) { set_error("fn: 0 exp, 1 erf as in libm_glibc.h; 2 exp, 3 erf of the device library"); return PMI_ERR_ARG; }
    if (n < 0) { set_error("negative n"); return PMI_ERR_ARG; }
    if (n == 0) return PMI_OK;
    hipLaunchKernelGGL(libm_eval_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, fn, d_x, n, d_out);
    PMI_HIP(hipGetLastError());
    return PMI_OK;
}
