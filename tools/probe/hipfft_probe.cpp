// Probe: does hipFFT (double, 2-D, odd sizes) work on this box without network?  hipcc tools/probe/hipfft_probe.cpp -lhipfft
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <chrono>
#include <cstdio>
#include <vector>
int main()
{
    for (int n : {512, 700, 2048}) {
        const int Y = n, X = n;
        std::vector<double> h((size_t)Y * X);
        for (size_t i = 0; i < h.size(); i++) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0;
        double *d_in; hipfftDoubleComplex *d_f;
        hipMalloc(&d_in, sizeof(double) * Y * X);
        hipMalloc(&d_f, sizeof(hipfftDoubleComplex) * Y * (X / 2 + 1));
        hipMemcpy(d_in, h.data(), sizeof(double) * Y * X, hipMemcpyHostToDevice);
        hipfftHandle plan, iplan;
        auto t0 = std::chrono::steady_clock::now();
        hipfftResult r1 = hipfftPlan2d(&plan, Y, X, HIPFFT_D2Z);
        hipfftResult r2 = hipfftPlan2d(&iplan, Y, X, HIPFFT_Z2D);
        hipfftResult r3 = hipfftExecD2Z(plan, d_in, d_f);
        hipDeviceSynchronize();
        auto t1 = std::chrono::steady_clock::now();
        for (int k = 0; k < 10; k++) { hipfftExecD2Z(plan, d_in, d_f); hipfftExecZ2D(iplan, d_f, d_in); }
        hipDeviceSynchronize();
        auto t2 = std::chrono::steady_clock::now();
        hipMemcpy(h.data(), d_in, sizeof(double) * 4, hipMemcpyDeviceToHost);
        printf("n=%d plan %d %d exec %d first %.1f ms, pair %.3f ms, h[1]=%g\n", n, (int)r1, (int)r2, (int)r3,
               std::chrono::duration<double, std::milli>(t1 - t0).count(),
               std::chrono::duration<double, std::milli>(t2 - t1).count() / 10, h[1]);
        hipfftDestroy(plan); hipfftDestroy(iplan); hipFree(d_in); hipFree(d_f);
    }
    return 0;
}
