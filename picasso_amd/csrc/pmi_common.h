// pmi_common.h — shared host/device helpers for libpicasso_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/picasso_hip.h"

#define PMI_WAVE 64

namespace pmi {

// Launch-shape and debugging variables (PMI_IDENTIFY_*, PMI_FIT_*, PMI_LQ_*, ...) are honoured only by a tuning build
// (make TUNING=1 -> -DPMI_TUNING): the shipped library reads none of them, so nothing in the environment can make it
// skip work or take another kernel.  (PMI_MLE_MODE, documented in picasso_hip.h, is the one variable it reads.)
#ifdef PMI_TUNING
inline const char *tuning_env(const char *name) { return getenv(name); }
#else
inline const char *tuning_env(const char *) { return nullptr; }
#endif

void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define PMI_HIP(call)                                                          \
    do {                                                                       \
        hipError_t e_ = (call);                                                \
        if (e_ != hipSuccess) return pmi::hip_fail(e_, #call, __FILE__, __LINE__); \
    } while (0)

// Library state that lives in device memory or belongs to a device — scratch banks, unit-vector tables, the opt-in to more
// than 64 KB of dynamic LDS per kernel, FFT plans, side streams — is keyed by the device current in the calling thread
// (hipGetDevice), so that ONE process can drive several GPUs from one host thread each (localize_streamed(devices=...),
// INTEGRATION.md).  Settings (the modes, the number of frame ranges) stay process-wide.
constexpr int PMI_MAX_DEVICES = 16;
int current_device();                 // the calling thread's device, 0 if it cannot be asked; pmi_set_device refuses >= PMI_MAX_DEVICES
int checked_device(int *device);      // the same with a status: PMI_ERR_HIP / PMI_ERR_ARG (a device the library keeps no state for) instead of device 0
int device_cu_count();                // compute units of that device (cached per device)

// Grow-only scratch arena per device.  slot = purpose id.
enum ScratchSlot {
    SCR_RECORDS = 0,      // unordered identification records
    SCR_RECORDS2,         // frame-grouped records
    SCR_FRAME_COUNT,      // per-frame counts + cursors + bases
    SCR_COUNTERS,         // small counters
    SCR_IDS,              // frame/y/x/ng for the fused pipeline
    SCR_FIT,              // thetas/crlbs/ll/it for the fused pipeline
    SCR_STAGE_A,          // host-API staging (movie chunks)
    SCR_STAGE_B,
    SCR_STAGE_C,
    SCR_STAGE_D,
    SCR_ROWS,             // rows the fit stages of the fused pipelines may touch
    SCR_NARROW,           // uint16 copy of a chunk of a 32-bit movie (identify)
    SCR_GATES,            // per-chunk flags of that copy
    SCR_PIX,              // box pixels of the candidates, left by the scan's exact stage for the fit (identify_fast.hip)
    SCR_STATS,            // flag statistics of the last MLE fit (re-fit count, count per criterion)
    SCR_LQ_STATS,         // the same of the last least-squares fit (spots fitted again, count per reason)
    SCR_NUM
};
int scratch(int slot, size_t bytes, void **ptr);
int scratch_release_all();
int scratch_enter_inner();           // -> the bank to hand back to scratch_leave_inner
void scratch_leave_inner(int was);
unsigned scratch_generation(int slot);      // of the current device and the calling thread's bank; bumped whenever a buffer of that slot (the bank or its inner bank) is released: pointers into it taken before are stale
unsigned scratch_generation_of(int device, int user_bank, int slot);
int scratch_user_bank();                    // the bank pmi_scratch_bank selected for the calling thread (its inner bank counts as the same)
// The side stream a fused call runs its second frame range on, with the events that join it to the caller's stream, and the
// events recorded behind the statistics kernels of a fit.  Process-wide, one per device, user bank and pipeline
// (0: MLE, 1: least squares), created on first use with that device current and kept for the life of the process: host
// threads come and go (localize_streamed starts lane threads per call), what belongs to a device does not.  Two threads on
// one device and bank are serialised by the caller (picasso_amd/_lib.py lock()), as for the scratch buffers.
struct SideLane { hipStream_t s2 = nullptr; hipEvent_t ev_start = nullptr, ev_scan_a = nullptr, ev_b = nullptr, stats_done[2] = {nullptr, nullptr}; };
int side_lane(int pipeline, SideLane **lane);
// Fused pipelines: *d_rows = *d_total if it fits the caller's capacity, else 0 (the identification columns were not
// written; the caller sees *d_total > cap and resubmits), so that the fit stages never follow stale rows.
int rows_to_fit(const int64_t *d_total, int64_t cap, const int64_t **d_rows, hipStream_t s);

struct Record {   // one identification, 16 B
    int32_t frame;
    uint32_t yx;      // y << 16 | x (frames are at most 65535 x 65535 on this path)
    int32_t slot;     // where the exact stage of the scan left the spot's pixels (identify_fast.hip), -1: nowhere
    float ng;
};
__host__ __device__ __forceinline__ uint32_t pack_yx(int y, int x) { return ((uint32_t)y << 16) | (uint32_t)(x & 0xffff); }
// A fused pipeline may ask the packed scan to leave the exact stage (float32 net gradient in the reference's order,
// first-argmax rule: picasso/localize.py:97-134, 202-244, 288) to the fit's start-value kernel, which reads the very
// rows it needs: the scan then emits CANDIDATES whose net gradient is this NaN pattern (an accepted identification's
// net gradient is never NaN: it passed `ng > min_ng`), and the start-value kernel decides them (gaussmle_g8.hip).
constexpr uint32_t NG_DEFERRED_BITS = 0x7fc0d1feu;
// set (per thread) by a fused call around identify_impl: the packed scan may defer its exact stage
extern thread_local bool g_defer_exact;
// the scan kernel the calling thread launched last, as rocprofv3 names it (+ " defer" when it may leave its exact stage to
// the fit): pmi_last_scan_kernel, so that a benchmark can tell which kernel a committed counter file belongs to
extern thread_local char g_last_scan_kernel[128];
extern int g_localize_ranges;        // frame ranges a fused call keeps in flight (pmi_localize_set_ranges)

// pixel load as float32 (the reference's np.float32(frame), localize.py:332)
template <typename T>
__device__ __forceinline__ float px_f32(const T *p, int64_t i) { return (float)p[i]; }

// Pixel hand-off from identify to the fit: a fused call sets this (per thread) around identify_impl; the packed uint16
// scan then leaves every candidate's box rows at pix[slot] and the ordered identifications carry their slot.
struct PixHandoff {
    uint32_t *pix = nullptr;        // shards x cap_per_shard slots of box * (box / 2 + 1) uint32 (packed uint16 pairs)
    unsigned cap_per_shard = 0;
    int32_t *d_slot = nullptr;      // out: slot of every identification (-1: none), ordered like d_frame / d_y / d_x
    bool used = false;              // out: the scan that ran could fill it
};
extern thread_local PixHandoff g_handoff;

struct KernelTimes { float scan_ms, fit_ms; };
extern bool g_kernel_timing;
extern KernelTimes g_last_times;

struct ScopedKernelTimer {
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t s;
    float *dst;
    ScopedKernelTimer(hipStream_t stream, float *dst_) : s(stream), dst(dst_) {
        if (g_kernel_timing) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, s); }
    }
    void stop() {
        if (a) { (void)hipEventRecord(b, s); }
    }
    ~ScopedKernelTimer() {
        if (a) {
            (void)hipEventSynchronize(b);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, a, b);
            *dst = ms;
            (void)hipEventDestroy(a); (void)hipEventDestroy(b);
        }
    }
};

}  // namespace pmi
