// gausslq_w.hip — the strict-mode Jacobian + QR kernel of the least-squares fit for boxes up to 7x7 (picasso/gausslq.py:206-244
// through scipy.optimize.leastsq = MINPACK lmdif: fdjac2 + qrfac), one image column per lane.  A translation unit of its own:
// the kernel is the hot one of config 3 and is tuned on its own.
#include "lq_common.h"

#pragma clang fp contract(off)

namespace pmi {
namespace lq {

#include "lq_jacobian_w.inc"

int launch_jacobian_w(const Params &p, const LqState &st, const int32_t *list, const unsigned *list_n, int64_t count, dim3 grid, hipStream_t s)
{
    const dim3 block(LQ_WAVES * 64);
    const size_t lds = lqw_lds_bytes(p.box);
    if (lds > 65536) {
        // more than 64 KB of dynamic LDS per workgroup has to be asked for, once per device and kernel
        static bool asked[PMI_MAX_DEVICES] = {};
        const int dv = current_device();
        if (!__atomic_load_n(&asked[dv], __ATOMIC_ACQUIRE)) {
            PMI_HIP(hipFuncSetAttribute((const void *)lq_jacobian_w_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            __atomic_store_n(&asked[dv], true, __ATOMIC_RELEASE);
        }
    }
    if (p.box == 7) hipLaunchKernelGGL((lq_jacobian_w_kernel<7>), grid, block, lds, s, p, st, list, list_n, count);
    else if (p.box == 5) hipLaunchKernelGGL((lq_jacobian_w_kernel<5>), grid, block, lds, s, p, st, list, list_n, count);
    else hipLaunchKernelGGL((lq_jacobian_w_kernel<3>), grid, block, lds, s, p, st, list, list_n, count);
    return PMI_OK;
}

}  // namespace lq
}  // namespace pmi
