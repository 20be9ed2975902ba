// stream_read.hip — calibration of the L2 memory-side read counter (TCC_EA0_RDREQ) on the scan kernel's own access
// pattern, as MI355X_MICROARCH.md asks before an absolute byte count is read from it: persistent one-wave workgroups,
// every wave reads consecutive 1 KiB rows (16 B per lane and instruction) of its own contiguous range, every byte of
// the buffer exactly once.  usage: stream_read [GiB-ish bytes] [rows per unit]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(64) void stream_read_kernel(const uint4 *__restrict__ src, long long rows, int rows_per_unit,
                                                         long long units, int upw, unsigned *__restrict__ sink, int slow)
{
    const int lane = threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    const long long u0 = (long long)blockIdx.x * upw, u1 = u0 + upw < units ? u0 + upw : units;
    for (long long u = u0; u < u1; u++) {
        const long long r0 = u * rows_per_unit, r1 = r0 + rows_per_unit < rows ? r0 + rows_per_unit : rows;
        long long r = r0;
        if (slow == 2) {
            // the scan's addressing: a buffer descriptor per frame of 512 rows, scalar row offset, per-lane column
            // offset, 2H + 2 = 8 halo rows and three rows of prefetch beyond the unit (clamped to the frame)
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            const long long f0 = r0 / 512 * 512;                       // first row of the frame
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(src + f0 * 64), 0, 512 * 1024, 0x00020000);
            const int lo = (int)(r0 - f0) - 4, hi = (int)(r1 - f0) + 4 + 3;
            for (int q = lo; q < hi; q++) {
                const int rc = q < 0 ? 0 : (q > 511 ? 511 : q);
                const u32x4_t a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, rc * 1024, 0);
                unsigned t = a.x;
#pragma unroll
                for (int k = 0; k < 20; k++) { t = t * 1664525u + a.y; t ^= t >> 7; }
                acc.x ^= t; acc.y ^= a.z; acc.z ^= a.w;
            }
            r = r1;
        } else if (slow) {
            // the scan's cadence: one row per ~100 VALU instructions, three rows in flight
            uint4 p0 = src[r * 64 + lane], p1 = src[(r + 1 < r1 ? r + 1 : r1 - 1) * 64 + lane], p2 = src[(r + 2 < r1 ? r + 2 : r1 - 1) * 64 + lane];
            for (; r < r1; r++) {
                const uint4 a = p0; p0 = p1; p1 = p2;
                p2 = src[(r + 3 < r1 ? r + 3 : r1 - 1) * 64 + lane];
                unsigned t = a.x;
#pragma unroll
                for (int k = 0; k < 50; k++) { t = t * 1664525u + a.y; t ^= t >> 7; }
                acc.x ^= t; acc.y ^= a.z; acc.z ^= a.w;
            }
        }
        for (; r + 4 <= r1; r += 4) {
            const uint4 a = src[(r + 0) * 64 + lane], b = src[(r + 1) * 64 + lane], c = src[(r + 2) * 64 + lane], d = src[(r + 3) * 64 + lane];
            acc.x ^= a.x ^ b.y ^ c.z ^ d.w; acc.y ^= a.y ^ b.z ^ c.w ^ d.x; acc.z ^= a.z ^ b.w ^ c.x ^ d.y; acc.w ^= a.w ^ b.x ^ c.y ^ d.z;
        }
        for (; r < r1; r++) { const uint4 a = src[r * 64 + lane]; acc.x ^= a.x; acc.y ^= a.y; acc.z ^= a.z; acc.w ^= a.w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;      // never true for a zero buffer; keeps the loads
}

int main(int argc, char **argv)
{
    const long long bytes = argc > 1 ? atoll(argv[1]) : 5242880000LL;
    const int rpu = argc > 2 ? atoi(argv[2]) : 256;
    const int slow = argc > 3 ? atoi(argv[3]) : 0;
    const long long rows = bytes / 1024, units = (rows + rpu - 1) / rpu;
    uint4 *d; unsigned *sink;
    if (hipMalloc(&d, rows * 1024) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 0, rows * 1024); hipMemset(sink, 0, 4);
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const long long waves = (long long)cus * 16;
    const long long blocks = units < waves ? units : waves;
    const int upw = (int)((units + blocks - 1) / blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 4; it++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(stream_read_kernel, dim3((unsigned)blocks), dim3(64), 0, 0, d, rows, rpu, units, upw, sink, slow);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("read %lld bytes in %.3f ms = %.1f GB/s (%lld blocks x %d units of %d rows)\n", rows * 1024, ms, rows * 1024 / ms * 1e-6, blocks, upw, rpu);
    }
    return 0;
}
