// gausslq_step.hip — the Levenberg-Marquardt step kernel of the least-squares fit (MINPACK lmdif's inner loop: lmpar, the trial
// evaluation, the accept / reject logic; picasso/gausslq.py:240-242 through scipy.optimize.leastsq), one spot per lane.
// A translation unit of its own, built per box size for the boxes of the fused pipelines (3, 5, 7).
#include <algorithm>

#include "lq_common.h"

#pragma clang fp contract(off)

namespace pmi {
namespace lq {

template <bool FROM_MOVIE, bool FLAG, bool FRAG, bool CR, int BOX>
__global__ __launch_bounds__(LQ_STEP_NT, LQ_STEP_MIN_WAVES) void lq_step_kernel(Params p, LqState st, const int32_t *__restrict__ list,
                                                             const unsigned *__restrict__ list_n, int64_t count,
                                                             int32_t *__restrict__ next_list, unsigned *__restrict__ next_n,
                                                             int32_t *__restrict__ tie_list, unsigned *__restrict__ tie_n)
{
    // dynamic LDS: [NT] spot indices, the x profile of the current evaluation (box x NT floats), the tile of NT x m floats
    extern __shared__ __attribute__((aligned(16))) char s_dyn[];
    const int size0 = BOX ? BOX : p.box;
    int64_t (&s_idx)[LQ_STEP_NT] = *reinterpret_cast<int64_t (*)[LQ_STEP_NT]>(s_dyn);
    float (*s_px)[LQ_STEP_NT] = reinterpret_cast<float (*)[LQ_STEP_NT]>(s_dyn + LQ_STEP_NT * sizeof(int64_t));
    float *s_tile = reinterpret_cast<float *>(s_dyn + LQ_STEP_NT * sizeof(int64_t) + (size_t)size0 * LQ_STEP_NT * sizeof(float));
    int64_t n = p.N;
    if (p.d_n) { const int64_t dn = *p.d_n; n = dn < n ? dn : n; }
    const int64_t items = list ? (int64_t)*list_n : (st.first + count < n ? count : n - st.first);
    const int size = BOX ? BOX : p.box, m = size * size;
    const int tid = threadIdx.x;
    const bool staged = m <= LQ_TILE_MAXPIX;
    const float *mytile = s_tile + (size_t)tid * m;
    // workgroups take the list in blocks of NT spots, round robin: a late round — few spots left, their number known to the
    // device only — is launched with a small grid
    for (int64_t blk = blockIdx.x; blk * LQ_STEP_NT < items; blk += gridDim.x) {
        const int64_t w = blk * LQ_STEP_NT + tid;
        const int64_t s = w < items ? (list ? (int64_t)list[w] : st.first + w) : -1;
        const int64_t ls = s - st.first;
        int info = s >= 0 ? LQI(st, 8, ls) : 1;
        __syncthreads();                                        // the previous block's tile is no longer read
        s_idx[tid] = info > 0 ? -1 : s;
        __syncthreads();
        if (staged) { stage_spots<FROM_MOVIE, LQ_STEP_NT, BOX * BOX>(p, s_idx, s_tile, m, size, st.first); __syncthreads(); }
        if (info > 0) continue;
        unsigned tie = 0u;
        info = lq_step_spot<FROM_MOVIE, FLAG, FRAG, CR, BOX>(p, st, s, info, tie_list != nullptr, mytile, s_px, tid, staged, tie);
        if (info != 0) {
            if ((FLAG || FRAG) && tie && tie_list) {
                tie_list[atomicAdd(tie_n, 1u)] = (int32_t)s;
                for (int b = 0; b < 8; b++)
                    if (tie & (1u << b)) atomicAdd(tie_n + 1 + b, 1u);      // why (diagnostics: pmi_gausslq_last_tie_reasons)
            }
        } else {
            next_list[atomicAdd(next_n, 1u)] = (int32_t)s;
        }
    }
}

template <bool FLAG, bool FRAG, int BOX>
static void launch_step_box(const Params &p, const LqState &st, const int32_t *list, const unsigned *list_n, int64_t count, int32_t *next_list,
                            unsigned *next_n, int32_t *tie_list, unsigned *tie_n, size_t step_lds, int64_t max_blocks, hipStream_t s)
{
    const unsigned sb = (unsigned)std::max<int64_t>(1, std::min<int64_t>((count + LQ_STEP_NT - 1) / LQ_STEP_NT, max_blocks));
    hipLaunchKernelGGL((lq_step_kernel<false, FLAG, FRAG, false, BOX>), dim3(sb), dim3(LQ_STEP_NT), step_lds, s, p, st, list, list_n, count,
                       next_list, next_n, tie_list, tie_n);
}

// max_blocks: at most that many workgroups (a late round's list is short)
// strict: the first pass of the strict mode (flags fragile float32 roundings); else that of the refit / fast modes (flags
// decisions near their thresholds)
int launch_step(bool strict, const Params &p, const LqState &st, const int32_t *list, const unsigned *list_n, int64_t count, int32_t *next_list,
                unsigned *next_n, int32_t *tie_list, unsigned *tie_n, size_t step_lds, int64_t max_blocks, hipStream_t s)
{
#define LQ_STEP(BOX) do { \
        if (strict) launch_step_box<false, true, BOX>(p, st, list, list_n, count, next_list, next_n, tie_list, tie_n, step_lds, max_blocks, s); \
        else launch_step_box<true, false, BOX>(p, st, list, list_n, count, next_list, next_n, tie_list, tie_n, step_lds, max_blocks, s); \
    } while (0)
    if (p.box == 7) LQ_STEP(7);
    else if (p.box == 5) LQ_STEP(5);
    else if (p.box == 3) LQ_STEP(3);
    else if (p.box == 9) LQ_STEP(9);
    else if (p.box == 13) LQ_STEP(13);
    else LQ_STEP(0);
#undef LQ_STEP
    PMI_HIP(hipGetLastError());
    return PMI_OK;
}

}  // namespace lq
}  // namespace pmi
