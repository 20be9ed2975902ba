// Micro-benchmark: issue rate of the VALU ops the identify scan is built from.
// hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define OPS(NAME, ASMSTR)                                                                        \
    __global__ void NAME(unsigned *out, int iters)                                               \
    {                                                                                            \
        unsigned a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, \
                 a7 = a0 * 19, b = blockIdx.x * 0x10001u + 12345u;                               \
        for (int i = 0; i < iters; i++) {                                                        \
            asm volatile(ASMSTR "\n" : "+v"(a0) : "v"(b));                                        \
            asm volatile(ASMSTR "\n" : "+v"(a1) : "v"(b));                                        \
            asm volatile(ASMSTR "\n" : "+v"(a2) : "v"(b));                                        \
            asm volatile(ASMSTR "\n" : "+v"(a3) : "v"(b));                                        \
            asm volatile(ASMSTR "\n" : "+v"(a4) : "v"(b));                                        \
            asm volatile(ASMSTR "\n" : "+v"(a5) : "v"(b));                                        \
            asm volatile(ASMSTR "\n" : "+v"(a6) : "v"(b));                                        \
            asm volatile(ASMSTR "\n" : "+v"(a7) : "v"(b));                                        \
        }                                                                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;      \
    }
OPS(k_pk_max, "v_pk_max_u16 %0, %0, %1")
OPS(k_max_u32, "v_max_u32 %0, %0, %1")
OPS(k_max3_u32, "v_max3_u32 %0, %0, %1, %1")
OPS(k_perm, "v_perm_b32 %0, %0, %1, %1")
OPS(k_alignbit, "v_alignbit_b32 %0, %0, %1, 16")
OPS(k_pk_sub, "v_pk_sub_u16 %0, %0, %1 clamp")
OPS(k_add_f32, "v_add_f32 %0, %0, %1")
OPS(k_fma, "v_fma_f32 %0, %0, %1, %1")
OPS(k_exp, "v_exp_f32 %0, %0")
OPS(k_rcp, "v_rcp_f32 %0, %0")
OPS(k_mul_f32, "v_mul_f32 %0, %0, %1")
#define OPS2(NAME, ASMSTR)                                                                       \
    __global__ void NAME(unsigned *out, int iters)                                               \
    {                                                                                            \
        typedef float f2 __attribute__((ext_vector_type(2)));                                    \
        f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 * 3.f, a2 = a0 * 5.f, a3 = a0 * 7.f, a4 = a0 * 11.f, a5 = a0 * 13.f, a6 = a0 * 17.f, a7 = a0 * 19.f; \
        f2 b = {1.0001f, 0.9999f};                                                               \
        for (int i = 0; i < iters; i++) {                                                        \
            asm volatile(ASMSTR "\n" : "+v"(a0) : "v"(b));                                        \
            asm volatile(ASMSTR "\n" : "+v"(a1) : "v"(b));                                        \
            asm volatile(ASMSTR "\n" : "+v"(a2) : "v"(b));                                        \
            asm volatile(ASMSTR "\n" : "+v"(a3) : "v"(b));                                        \
            asm volatile(ASMSTR "\n" : "+v"(a4) : "v"(b));                                        \
            asm volatile(ASMSTR "\n" : "+v"(a5) : "v"(b));                                        \
            asm volatile(ASMSTR "\n" : "+v"(a6) : "v"(b));                                        \
            asm volatile(ASMSTR "\n" : "+v"(a7) : "v"(b));                                        \
        }                                                                                        \
        f2 r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = __float_as_uint(r.x + r.y);                  \
    }
OPS2(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %1")
OPS2(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
OPS2(k_pk_add_f32, "v_pk_add_f32 %0, %0, %1")
OPS2(k_fma_f64, "v_fma_f64 %0, %0, %1, %1")
OPS2(k_mul_f64, "v_mul_f64 %0, %0, %1")
OPS2(k_addd_f64, "v_add_f64 %0, %0, %1")
#define OPS3(NAME, ASMSTR, TD, TS)                                                               \
    __global__ void NAME(unsigned *out, int iters)                                               \
    {                                                                                            \
        TD a0 = (TD)threadIdx.x, a1 = a0 + 3, a2 = a0 + 5, a3 = a0 + 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19; \
        TS b = (TS)(threadIdx.x + 1.25);                                                         \
        for (int i = 0; i < iters; i++) {                                                        \
            asm volatile(ASMSTR "\n" : "+v"(a0) : "v"(b));                                        \
            asm volatile(ASMSTR "\n" : "+v"(a1) : "v"(b));                                        \
            asm volatile(ASMSTR "\n" : "+v"(a2) : "v"(b));                                        \
            asm volatile(ASMSTR "\n" : "+v"(a3) : "v"(b));                                        \
            asm volatile(ASMSTR "\n" : "+v"(a4) : "v"(b));                                        \
            asm volatile(ASMSTR "\n" : "+v"(a5) : "v"(b));                                        \
            asm volatile(ASMSTR "\n" : "+v"(a6) : "v"(b));                                        \
            asm volatile(ASMSTR "\n" : "+v"(a7) : "v"(b));                                        \
        }                                                                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7); \
    }
OPS3(k_cvt_f32_f64, "v_cvt_f32_f64 %0, %1", float, double)
OPS3(k_cvt_f64_f32, "v_cvt_f64_f32 %0, %1", double, float)
OPS3(k_rcp_f64, "v_rcp_f64 %0, %1", double, double)
OPS3(k_sqrt_f64, "v_sqrt_f64 %0, %1", double, double)
OPS3(k_cvt_f32_i32, "v_cvt_f32_i32 %0, %1", float, int)
OPS(k_lshl_or, "v_lshl_or_b32 %0, %0, 1, %1")
OPS(k_mov_dpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
OPS(k_add_dpp, "v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")

template <typename K> void run(const char *name, K k, int waves_per_simd)
{
    int dev; hipGetDevice(&dev); hipDeviceProp_t pr; hipGetDeviceProperties(&pr, dev);
    int cus = pr.multiProcessorCount;
    unsigned *out; hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
    int iters = 20000;
    dim3 grid(cus * waves_per_simd), block(256);   // 256 threads = 4 waves = 1 per SIMD
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, grid, block, 0, 0, out, 100);
    hipEventRecord(a);
    hipLaunchKernelGGL(k, grid, block, 0, 0, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double instr_per_simd = (double)iters * 8 * waves_per_simd;
    double clk = pr.clockRate * 1e3;   // kHz -> Hz
    printf("%-14s waves/SIMD %d: %.3f ms  -> %.2f cycles per wave-instruction (at %.0f MHz nominal)\n", name,
           waves_per_simd, ms, ms * 1e-3 * clk / instr_per_simd, clk / 1e6);
    hipFree(out);
}
int main()
{
    for (int w : {2, 8}) {
        run("v_pk_max_u16", k_pk_max, w); run("v_max_u32", k_max_u32, w); 
        
        run("v_exp_f32", k_exp, w);
        run("v_fma_f32", k_fma, w); run("v_mul_f32", k_mul_f32, w);
        run("v_pk_fma_f32", k_pk_fma_f32, w); run("v_pk_mul_f32", k_pk_mul_f32, w); run("v_pk_add_f32", k_pk_add_f32, w);
        run("v_fma_f64", k_fma_f64, w); run("v_mul_f64", k_mul_f64, w); run("v_add_f64", k_addd_f64, w); run("v_cvt_f32_f64", k_cvt_f32_f64, w); run("v_cvt_f64_f32", k_cvt_f64_f32, w); run("v_rcp_f64", k_rcp_f64, w); run("v_sqrt_f64", k_sqrt_f64, w); run("v_cvt_f32_i32", k_cvt_f32_i32, w);
        run("v_rcp_f32", k_rcp, w);
         run("v_add_f32_dpp", k_add_dpp, w);
    }
    return 0;
}
