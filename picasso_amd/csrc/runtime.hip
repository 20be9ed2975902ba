// runtime.hip — library state: errors, device selection, memory, scratch, events.
#include <stdarg.h>

#include <mutex>

#include "pmi_common.h"

namespace pmi {

static thread_local char g_err[512] = "";
bool g_kernel_timing = false;
KernelTimes g_last_times = {0.f, 0.f};
thread_local PixHandoff g_handoff;
thread_local bool g_defer_exact = false;
int g_localize_ranges = 2;          // pmi_localize_set_ranges (gaussmle.hip); read by both fused calls
thread_local char g_last_scan_kernel[128] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what, const char *file, int line)
{
    set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    return PMI_ERR_HIP;
}

struct ScratchBuf { void *p = nullptr; size_t bytes = 0; };
// Two banks: a caller that keeps two pipelines in flight on two streams (frame range A fitting while range B is
// scanned) gives each its own scratch (pmi_scratch_bank); everything else lives in bank 0.
constexpr int SCR_USER_BANKS = 2;                 // what pmi_scratch_bank selects
constexpr int SCR_BANKS = 2 * SCR_USER_BANKS;      // + one inner bank each: the second frame range a fused call keeps in flight
static ScratchBuf g_scratch_banks[PMI_MAX_DEVICES][SCR_BANKS][SCR_NUM];      // [device]: a buffer belongs to the device it was allocated on
// the bank is a property of the calling thread: two host threads that drive two streams select a bank each
// (pmi_scratch_bank) and never see each other's records or fit state
static thread_local int g_scratch_bank = 0;
static std::mutex g_scratch_mu[PMI_MAX_DEVICES];      // one per device: growth on one device (a device-wide synchronise) does not stall the others' lanes
// per device, USER bank (a bank and its inner bank count as one) and slot: a record in one slot outlives the growth of
// another, and the statistics of a lane on bank 0 outlive growth in the other lane of the same device
static unsigned g_scratch_generation[PMI_MAX_DEVICES][SCR_USER_BANKS][SCR_NUM] = {};
int scratch_user_bank() { return g_scratch_bank % SCR_USER_BANKS; }
unsigned scratch_generation_of(int device, int user_bank, int slot)
{
    std::lock_guard<std::mutex> lk(g_scratch_mu[device]);
    return g_scratch_generation[device][user_bank % SCR_USER_BANKS][slot];
}
unsigned scratch_generation(int slot) { return scratch_generation_of(current_device(), scratch_user_bank(), slot); }

int current_device()
{
    // pmi_set_device refuses devices >= PMI_MAX_DEVICES; a thread put on one behind the library's back (hipSetDevice, torch)
    // is refused by checked_device() at the head of every scratch() call instead of being aliased to device 0's state
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PMI_MAX_DEVICES) return 0;
    return dev;
}
int checked_device(int *device)
{
    int dev = 0;
    PMI_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= PMI_MAX_DEVICES) {
        set_error("the calling thread is on device %d: the library keeps state for devices 0 ... %d", dev, PMI_MAX_DEVICES - 1);
        return PMI_ERR_ARG;
    }
    *device = dev;
    return PMI_OK;
}
int device_cu_count()
{
    static int cus[PMI_MAX_DEVICES] = {};
    const int dev = current_device();
    int n = __atomic_load_n(&cus[dev], __ATOMIC_RELAXED);
    if (!n) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        __atomic_store_n(&cus[dev], n, __ATOMIC_RELAXED);
    }
    return n;
}

int scratch(int slot, size_t bytes, void **ptr)
{
    int dev = 0;
    const int rc = checked_device(&dev);
    if (rc != PMI_OK) return rc;
    std::lock_guard<std::mutex> lk(g_scratch_mu[dev]);
    ScratchBuf &b = g_scratch_banks[dev][g_scratch_bank][slot];
    if (b.bytes < bytes) {
        if (b.p) { PMI_HIP(hipDeviceSynchronize()); PMI_HIP(hipFree(b.p)); b.p = nullptr; b.bytes = 0; g_scratch_generation[dev][scratch_user_bank()][slot]++; }
        size_t want = bytes + bytes / 4 + 4096;   // headroom so repeated calls stop reallocating
        PMI_HIP(hipMalloc(&b.p, want));
        b.bytes = want;
    }
    *ptr = b.p;
    return PMI_OK;
}

int scratch_release_all()
{
    // every device's buffers: each is synchronised and freed with its own device current, the caller's device is restored
    const int was = current_device();
    int rc = PMI_OK;
    for (int dev = 0; dev < PMI_MAX_DEVICES; dev++) {
        std::lock_guard<std::mutex> lk(g_scratch_mu[dev]);
        bool any = false;
        for (auto &bank : g_scratch_banks[dev])
            for (auto &b : bank) any = any || b.p;
        for (auto &bank : g_scratch_generation[dev])
            for (unsigned &g : bank) g++;
        if (!any) continue;
        if (hipSetDevice(dev) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { rc = PMI_ERR_HIP; set_error("pmi_release_scratch: device %d cannot be synchronised", dev); continue; }
        for (auto &bank : g_scratch_banks[dev])
            for (auto &b : bank)
                if (b.p) { if (hipFree(b.p) != hipSuccess) rc = PMI_ERR_HIP; b.p = nullptr; b.bytes = 0; }
    }
    (void)hipSetDevice(was);
    return rc;
}
int scratch_select_bank(int bank)
{
    if (bank < 0 || bank >= SCR_USER_BANKS) { set_error("scratch bank %d out of range", bank); return PMI_ERR_ARG; }
    g_scratch_bank = bank;
    return PMI_OK;
}
// the inner bank of the calling thread's bank (and back): scratch of the second frame range of a fused call
int scratch_enter_inner() { const int was = g_scratch_bank; g_scratch_bank = was % SCR_USER_BANKS + SCR_USER_BANKS; return was; }
void scratch_leave_inner(int was) { g_scratch_bank = was; }

static SideLane g_side_lanes[PMI_MAX_DEVICES][SCR_USER_BANKS][2];
static std::mutex g_side_mu;
int side_lane(int pipeline, SideLane **lane)
{
    int dev = 0;
    const int rc = checked_device(&dev);
    if (rc != PMI_OK) return rc;
    std::lock_guard<std::mutex> lk(g_side_mu);
    SideLane &sd = g_side_lanes[dev][scratch_user_bank()][pipeline ? 1 : 0];
    if (!sd.s2) {
        hipStream_t s2 = nullptr;
        PMI_HIP(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        PMI_HIP(hipEventCreateWithFlags(&sd.ev_start, hipEventDisableTiming));
        PMI_HIP(hipEventCreateWithFlags(&sd.ev_scan_a, hipEventDisableTiming));
        PMI_HIP(hipEventCreateWithFlags(&sd.ev_b, hipEventDisableTiming));
        PMI_HIP(hipEventCreateWithFlags(&sd.stats_done[0], hipEventDisableTiming));
        PMI_HIP(hipEventCreateWithFlags(&sd.stats_done[1], hipEventDisableTiming));
        sd.s2 = s2;                        // last: a lane with a stream is complete
    }
    *lane = &sd;
    return PMI_OK;
}

__global__ void rows_to_fit_kernel(const int64_t *__restrict__ total, int64_t cap, int64_t *__restrict__ rows)
{
    const int64_t n = *total;
    *rows = n > cap ? 0 : n;
}

int rows_to_fit(const int64_t *d_total, int64_t cap, const int64_t **d_rows, hipStream_t s)
{
    void *ptr = nullptr;
    int rc = scratch(SCR_ROWS, sizeof(int64_t), &ptr);
    if (rc != PMI_OK) return rc;
    hipLaunchKernelGGL(rows_to_fit_kernel, dim3(1), dim3(1), 0, s, d_total, cap, (int64_t *)ptr);
    PMI_HIP(hipGetLastError());
    *d_rows = (const int64_t *)ptr;
    return PMI_OK;
}

void release_fft_plans();   // xcorr.hip

}  // namespace pmi

extern "C" {

int pmi_version(void) { return 106; }   // 0.1.6: round 6 (32-bit integer movies on the key scan, side lanes per (device, bank); + pmi_mle_set_libm / pmi_mle_get_libm / pmi_libm_eval_dev)

const char *pmi_last_error(void) { return pmi::g_err; }

int pmi_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int pmi_set_device(int device)
{
    // the calling THREAD's device (HIP keeps it per thread); everything the library keeps on a device is keyed by it
    if (device < 0 || device >= pmi::PMI_MAX_DEVICES || device >= pmi_device_count()) {
        pmi::set_error("pmi_set_device: no device %d (%d visible, at most %d supported)", device, pmi_device_count(), pmi::PMI_MAX_DEVICES);
        return PMI_ERR_ARG;
    }
    PMI_HIP(hipSetDevice(device));
    return PMI_OK;
}

int pmi_last_scan_kernel(char *name, size_t name_len)
{
    if (name && name_len) { strncpy(name, pmi::g_last_scan_kernel, name_len - 1); name[name_len - 1] = 0; }
    return PMI_OK;
}

int pmi_get_device(int *device)
{
    int dev = 0;
    PMI_HIP(hipGetDevice(&dev));
    if (device) *device = dev;
    return PMI_OK;
}

int pmi_device_info(char *name, size_t name_len, int *compute_units, size_t *total_mem_bytes)
{
    int dev = 0;
    PMI_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    PMI_HIP(hipGetDeviceProperties(&prop, dev));
    if (name && name_len) { strncpy(name, prop.gcnArchName, name_len - 1); name[name_len - 1] = 0; }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (total_mem_bytes) *total_mem_bytes = prop.totalGlobalMem;
    return PMI_OK;
}

int pmi_malloc(void **dptr, size_t bytes) { PMI_HIP(hipMalloc(dptr, bytes)); return PMI_OK; }
int pmi_free(void *dptr) { PMI_HIP(hipFree(dptr)); return PMI_OK; }
int pmi_memcpy_h2d(void *d, const void *h, size_t bytes) { PMI_HIP(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice)); return PMI_OK; }
int pmi_memcpy_d2h(void *h, const void *d, size_t bytes) { PMI_HIP(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost)); return PMI_OK; }
int pmi_stream_synchronize(void *stream) { PMI_HIP(hipStreamSynchronize((hipStream_t)stream)); return PMI_OK; }
int pmi_stream_create(void **stream)
{
    // non-blocking: work on it does not order against the default stream, so a host thread's blocking upload of the
    // next frame chunk runs beside this stream's kernels and row copies
    hipStream_t s;
    PMI_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return PMI_OK;
}
int pmi_stream_destroy(void *stream) { PMI_HIP(hipStreamDestroy((hipStream_t)stream)); return PMI_OK; }
int pmi_memcpy_d2h_async(void *h, const void *d, size_t bytes, void *stream)
{
    PMI_HIP(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return PMI_OK;
}
int pmi_release_scratch(void) { pmi::release_fft_plans(); return pmi::scratch_release_all(); }
int pmi_scratch_bank(int bank) { return pmi::scratch_select_bank(bank); }

int pmi_event_create(void **event)
{
    hipEvent_t e;
    PMI_HIP(hipEventCreate(&e));
    *event = (void *)e;
    return PMI_OK;
}
int pmi_event_record(void *event, void *stream) { PMI_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream)); return PMI_OK; }
int pmi_event_elapsed_ms(void *start, void *stop, float *ms)
{
    PMI_HIP(hipEventSynchronize((hipEvent_t)stop));
    PMI_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return PMI_OK;
}
int pmi_event_destroy(void *event) { PMI_HIP(hipEventDestroy((hipEvent_t)event)); return PMI_OK; }

int pmi_set_kernel_timing(int enabled) { pmi::g_kernel_timing = enabled != 0; return PMI_OK; }
int pmi_last_kernel_ms(float *scan_ms, float *fit_ms)
{
    if (scan_ms) *scan_ms = pmi::g_last_times.scan_ms;
    if (fit_ms) *fit_ms = pmi::g_last_times.fit_ms;
    return PMI_OK;
}

}  // extern "C"
