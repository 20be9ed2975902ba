// zfit.hip — astigmatic z per localization and the "avg" ROI sum.
//
// zfit_kernel replaces the per-localization loop of picasso/zfit.py:327-382 (_fit_z):
// minimise (sqrt(sx) - sqrt(wx(z)))^2 + (sqrt(sy) - sqrt(wy(z)))^2 over z in [-1000, 1000]
// (picasso/zfit.py:254-291 _fit_z_target) with scipy.optimize.minimize_scalar(bounds=...) —
// third-party arithmetic: scipy/optimize/_optimize.py:_minimize_scalar_bounded (scipy 1.15.3;
// Brent's golden-section / parabolic-interpolation fminbound, xatol 1e-5, maxiter 500),
// restated here step for step in float64, one thread per localization.
//
// avgroi_kernel replaces picasso/avgroi.py:24-41 (_sum / fit_spot): float64 sum of the ROI.
#include <algorithm>
#include <cmath>

#include "pmi_common.h"

// the reference's target function is compiled by numba without FMA contraction; Brent's
// comparisons (fu <= fx) are sensitive to the last bit, so none here either
#pragma clang fp contract(off)

namespace pmi {

struct Calib { double cx[7], cy[7]; };

__device__ __forceinline__ double z_target(double z, double ssx, double ssy, const Calib &c)
{
    const double z2 = z * z, z3 = z * z2, z4 = z * z3, z5 = z * z4, z6 = z * z5;
    const double wx = c.cx[0] * z6 + c.cx[1] * z5 + c.cx[2] * z4 + c.cx[3] * z3 + c.cx[4] * z2 + c.cx[5] * z + c.cx[6];
    const double wy = c.cy[0] * z6 + c.cy[1] * z5 + c.cy[2] * z4 + c.cy[3] * z3 + c.cy[4] * z2 + c.cy[5] * z + c.cy[6];
    const double ax = ssx - sqrt(wx), ay = ssy - sqrt(wy);     // x ** 0.5 of a negative is NaN, like sqrt
    return ax * ax + ay * ay;
}

__device__ __forceinline__ double sgn(double v) { return (double)((v > 0) - (v < 0)); }

__global__ __launch_bounds__(256) void zfit_kernel(const float *__restrict__ sx, const float *__restrict__ sy,
                                                   int64_t N, const int64_t *__restrict__ d_n, Calib c,
                                                   double *__restrict__ z_out, double *__restrict__ sq_out)
{
    int64_t n = N;
    if (d_n) { int64_t dn = *d_n; n = dn < n ? dn : n; }
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double ssx = sqrt((double)sx[i]), ssy = sqrt((double)sy[i]);
    const double xatol = 1e-5;
    const int maxfun = 500;
    const double sqrt_eps = sqrt(2.2e-16), golden_mean = 0.5 * (3.0 - sqrt(5.0));
    double a = -1000.0, b = 1000.0;
    double fulc = a + golden_mean * (b - a), nfc = fulc, xf = fulc;
    double rat = 0.0, e = 0.0;
    double x = xf, fx = z_target(x, ssx, ssy, c);
    int num = 1;
    double fu, ffulc = fx, fnfc = fx;
    double xm = 0.5 * (a + b);
    double tol1 = sqrt_eps * fabs(xf) + xatol / 3.0, tol2 = 2.0 * tol1;
    while (fabs(xf - xm) > (tol2 - 0.5 * (b - a))) {
        bool golden = true;
        if (fabs(e) > tol1) {                      // try a parabolic step
            golden = false;
            double r = (xf - nfc) * (fx - ffulc);
            double q = (xf - fulc) * (fx - fnfc);
            double p = (xf - fulc) * q - (xf - nfc) * r;
            q = 2.0 * (q - r);
            if (q > 0.0) p = -p;
            q = fabs(q);
            r = e;
            e = rat;
            if ((fabs(p) < fabs(0.5 * q * r)) && (p > q * (a - xf)) && (p < q * (b - xf))) {
                rat = (p + 0.0) / q;
                x = xf + rat;
                if (((x - a) < tol2) || ((b - x) < tol2)) {
                    const double si = sgn(xm - xf) + ((xm - xf) == 0);
                    rat = tol1 * si;
                }
            } else {
                golden = true;
            }
        }
        if (golden) {
            e = (xf >= xm) ? a - xf : b - xf;
            rat = golden_mean * e;
        }
        const double si = sgn(rat) + (rat == 0);
        const double ar = fabs(rat);
        const double stepm = (ar != ar) ? ar : ((tol1 != tol1) ? tol1 : (ar > tol1 ? ar : tol1));   // np.maximum
        x = xf + si * stepm;
        fu = z_target(x, ssx, ssy, c);
        num++;
        if (fu <= fx) {
            if (x >= xf) a = xf; else b = xf;
            fulc = nfc; ffulc = fnfc;
            nfc = xf; fnfc = fx;
            xf = x; fx = fu;
        } else {
            if (x < xf) a = x; else b = x;
            if ((fu <= fnfc) || (nfc == xf)) {
                fulc = nfc; ffulc = fnfc;
                nfc = x; fnfc = fu;
            } else if ((fu <= ffulc) || (fulc == xf) || (fulc == nfc)) {
                fulc = x; ffulc = fu;
            }
        }
        xm = 0.5 * (a + b);
        tol1 = sqrt_eps * fabs(xf) + xatol / 3.0;
        tol2 = 2.0 * tol1;
        if (num >= maxfun) break;
    }
    z_out[i] = xf;
    sq_out[i] = fx;
}

__global__ void avgroi_kernel(const float *__restrict__ spots, int64_t N, const int64_t *__restrict__ d_n, int npix,
                              float *__restrict__ theta)
{
    int64_t n = N;
    if (d_n) { int64_t dn = *d_n; n = dn < n ? dn : n; }
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    const float *sp = spots + i * npix;
    for (int k = 0; k < npix; k++) s += (double)sp[k];      // row-major order, float64 accumulator
    float *t = theta + i * 6;
    t[0] = 0.f; t[1] = 0.f; t[2] = (float)s; t[3] = (float)s; t[4] = 1.f; t[5] = 1.f;
}

}  // namespace pmi

extern "C" {

int pmi_zfit_dev(const float *d_sx, const float *d_sy, int64_t N, const int64_t *d_n, const double *cx7,
                 const double *cy7, double *d_z, double *d_sq, void *stream)
{
    using namespace pmi;
    if (!cx7 || !cy7) { set_error("null calibration"); return PMI_ERR_ARG; }
    if (N <= 0) return PMI_OK;
    Calib c;
    for (int k = 0; k < 7; k++) { c.cx[k] = cx7[k]; c.cy[k] = cy7[k]; }
    hipLaunchKernelGGL(zfit_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_sx, d_sy,
                       N, d_n, c, d_z, d_sq);
    PMI_HIP(hipGetLastError());
    return PMI_OK;
}

int pmi_zfit(const float *sx, const float *sy, int64_t N, const double *cx7, const double *cy7, double *z, double *sq)
{
    using namespace pmi;
    if (pmi_device_count() < 1) { set_error("no HIP device"); return PMI_ERR_NODEVICE; }
    if (N == 0) return PMI_OK;
    if (!sx || !sy || !z || !sq) { set_error("null pointer"); return PMI_ERR_ARG; }
    void *d_in = nullptr, *d_out = nullptr;
    int rc;
    if ((rc = scratch(SCR_STAGE_A, (size_t)N * 8, &d_in)) != PMI_OK) return rc;
    if ((rc = scratch(SCR_STAGE_B, (size_t)N * 16, &d_out)) != PMI_OK) return rc;
    float *d_sx = (float *)d_in, *d_sy = d_sx + N;
    double *d_z = (double *)d_out, *d_sq = d_z + N;
    PMI_HIP(hipMemcpy(d_sx, sx, (size_t)N * 4, hipMemcpyHostToDevice));
    PMI_HIP(hipMemcpy(d_sy, sy, (size_t)N * 4, hipMemcpyHostToDevice));
    if ((rc = pmi_zfit_dev(d_sx, d_sy, N, nullptr, cx7, cy7, d_z, d_sq, nullptr)) != PMI_OK) return rc;
    PMI_HIP(hipMemcpy(z, d_z, (size_t)N * 8, hipMemcpyDeviceToHost));
    PMI_HIP(hipMemcpy(sq, d_sq, (size_t)N * 8, hipMemcpyDeviceToHost));
    return PMI_OK;
}

int pmi_avgroi_dev(const float *d_spots, int64_t N, const int64_t *d_n, int box, float *d_theta, void *stream)
{
    using namespace pmi;
    if (box < 1 || box > PMI_MAX_BOX) { set_error("bad box %d", box); return PMI_ERR_ARG; }
    if (N <= 0) return PMI_OK;
    hipLaunchKernelGGL(avgroi_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_spots, N,
                       d_n, box * box, d_theta);
    PMI_HIP(hipGetLastError());
    return PMI_OK;
}

int pmi_avgroi(const float *spots, int64_t N, int box, float *theta)
{
    using namespace pmi;
    if (pmi_device_count() < 1) { set_error("no HIP device"); return PMI_ERR_NODEVICE; }
    if (N == 0) return PMI_OK;
    if (!spots || !theta) { set_error("null pointer"); return PMI_ERR_ARG; }
    void *d_in = nullptr, *d_out = nullptr;
    int rc;
    const size_t in_bytes = (size_t)N * box * box * 4;
    if ((rc = scratch(SCR_STAGE_A, in_bytes, &d_in)) != PMI_OK) return rc;
    if ((rc = scratch(SCR_STAGE_B, (size_t)N * 24, &d_out)) != PMI_OK) return rc;
    PMI_HIP(hipMemcpy(d_in, spots, in_bytes, hipMemcpyHostToDevice));
    if ((rc = pmi_avgroi_dev((const float *)d_in, N, nullptr, box, (float *)d_out, nullptr)) != PMI_OK) return rc;
    PMI_HIP(hipMemcpy(theta, d_out, (size_t)N * 24, hipMemcpyDeviceToHost));
    return PMI_OK;
}

}  // extern "C"
