// Micro-benchmark: does a DS read with few active lanes cost fewer LDS cycles?  ds_read_b128 / ds_read_b64 with the EXEC mask
// limited to given lanes, 8 waves per CU streaming reads; also the latency of a dependent v_add_f64 chain.
// hipcc --offload-arch=gfx950 -O3 lds_mask.hip -o lds_mask && ./lds_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double d2_t __attribute__((ext_vector_type(2)));
template <int WIDE>
__global__ __launch_bounds__(256) void k_read(double *out, int iters, unsigned long long mask)
{
    __shared__ __attribute__((aligned(16))) double s[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) s[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    double acc = 0;
    if ((mask >> lane) & 1ull) {
        const __attribute__((address_space(3))) d2_t *p = (const __attribute__((address_space(3))) d2_t *)(s + (threadIdx.x >> 6) * 1024 + 2 * lane);
        const __attribute__((address_space(3))) double *q = (const __attribute__((address_space(3))) double *)(s + (threadIdx.x >> 6) * 1024 + lane);
        for (int i = 0; i < iters; i++) {
            if (WIDE) {
                d2_t v[8];
#pragma unroll
                for (int k = 0; k < 8; k++) v[k] = p[64 * k % 448];
#pragma unroll
                for (int k = 0; k < 8; k++) asm volatile("" : "+v"(v[k]));
#pragma unroll
                for (int k = 0; k < 8; k++) acc += v[k].x;
            } else {
                double v[8];
#pragma unroll
                for (int k = 0; k < 8; k++) v[k] = q[64 * k];
#pragma unroll
                for (int k = 0; k < 8; k++) asm volatile("" : "+v"(v[k]));
#pragma unroll
                for (int k = 0; k < 8; k++) acc += v[k];
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ void k_addchain(double *out, int iters)
{
    double a = threadIdx.x, b = 1.000001;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
int main()
{
    double *out; hipMalloc(&out, 256 * 2048 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct { const char *name; unsigned long long m; } masks[] = {
        {"all 64", ~0ull}, {"lanes 0-31", 0xffffffffull}, {"lanes 0-15", 0xffffull}, {"lanes 0-5 + 32-37", 0x3f0000003full},
        {"lanes 0-11", 0xfffull}, {"lanes 0-3,12,13 + 32-35,44,45 (two b128 groups)", 0xfull | (0x3ull << 12) | (0xfull << 32) | (0x3ull << 44)},
        {"lanes 0-3 (one b128 group)", 0xfull}, {"lanes 0-3 + 4-7 (two b128 groups)", 0xffull}};
    const int iters = 4000, blocks = 256 * 2;          // 8 waves per CU
    for (int wide = 0; wide < 2; wide++)
        for (auto &mk : masks) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (wide) hipLaunchKernelGGL(k_read<1>, dim3(blocks), dim3(256), 0, 0, out, iters, mk.m);
                else hipLaunchKernelGGL(k_read<0>, dim3(blocks), dim3(256), 0, 0, out, iters, mk.m);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            // DS instructions per CU: 8 waves x iters x 8; cycles at 2.4 GHz
            const double cyc = ms * 1e-3 * 2.4e9 / (8.0 * iters * 8.0);
            printf("%s %-52s %.3f ms  %.2f cycles per wave-instruction per CU\n", wide ? "ds_read_b128" : "ds_read_b64 ", mk.name, ms, cyc);
        }
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_addchain, dim3(1024), dim3(64), 0, 0, out, 20000);      // one wave per SIMD
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    printf("dependent v_add_f64 chain, one wave per SIMD: %.2f cycles per add\n", ms * 1e-3 * 2.4e9 / (20000.0 * 16));
    return 0;
}
