// gausslq_w.hip — the strict-mode Jacobian + QR kernel of the least-squares fit (picasso/gausslq.py:206-244
// through scipy.optimize.leastsq = MINPACK lmdif: fdjac2 + qrfac), one image column per lane.  A translation unit of its own:
// the kernel is the hot one of config 3 and is tuned on its own.
#include <algorithm>

#include "lq_common.h"

#pragma clang fp contract(off)

namespace pmi {
namespace lq {

#include "lq_jacobian_w.inc"

template <int W>
static int launch_w(const Params &p, const LqState &st, const int32_t *list, const unsigned *list_n, int64_t count, int64_t max_blocks, hipStream_t s)
{
    constexpr int NGRP = 64 / LqwBox<W>::GS;           // spots per wavefront
    const size_t lds = lqw_lds_bytes_of<W>();
    if (lds > 65536) {
        // more than 64 KB of dynamic LDS per workgroup has to be asked for, once per device and kernel
        static bool asked[PMI_MAX_DEVICES] = {};
        const int dv = current_device();
        if (!__atomic_load_n(&asked[dv], __ATOMIC_ACQUIRE)) {
            PMI_HIP(hipFuncSetAttribute((const void *)lq_jacobian_w_kernel<W>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            __atomic_store_n(&asked[dv], true, __ATOMIC_RELEASE);
        }
    }
    const int64_t waves = (count + NGRP - 1) / NGRP;
    const dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>((waves + LQ_WAVES - 1) / LQ_WAVES, max_blocks)));
    hipLaunchKernelGGL((lq_jacobian_w_kernel<W>), grid, dim3(LQ_WAVES * 64), lds, s, p, st, list, list_n, count);
    return PMI_OK;
}

// max_blocks: at most that many workgroups (the kernel walks its list with a stride; a late round is launched small)
// the Jacobian + QR of the spots of a round (list == nullptr: spots [st.first, st.first + count)), every odd box 3 .. 21
int launch_jacobian_w(const Params &p, const LqState &st, const int32_t *list, const unsigned *list_n, int64_t count, int64_t max_blocks, hipStream_t s)
{
    switch (p.box) {
    case 3: return launch_w<3>(p, st, list, list_n, count, max_blocks, s);
    case 5: return launch_w<5>(p, st, list, list_n, count, max_blocks, s);
    case 7: return launch_w<7>(p, st, list, list_n, count, max_blocks, s);
    case 9: return launch_w<9>(p, st, list, list_n, count, max_blocks, s);
    case 11: return launch_w<11>(p, st, list, list_n, count, max_blocks, s);
    case 13: return launch_w<13>(p, st, list, list_n, count, max_blocks, s);
    case 15: return launch_w<15>(p, st, list, list_n, count, max_blocks, s);
    case 17: return launch_w<17>(p, st, list, list_n, count, max_blocks, s);
    case 19: return launch_w<19>(p, st, list, list_n, count, max_blocks, s);
    case 21: return launch_w<21>(p, st, list, list_n, count, max_blocks, s);
    default: set_error("gausslq: no kernel for box %d", p.box); return PMI_ERR_ARG;
    }
}

}  // namespace lq
}  // namespace pmi
