// xcorr.hip — image cross-correlation for redundant-cross-correlation (RCC) drift correction
// (picasso/imageprocess.py:27-50 xcorr, :53-161 get_image_shift up to the peak fit, :164-217 rcc).
//
//   xcorr(A, B) = fftshift(real(ifft2(fft2(A) * conj(fft2(B))))) / sqrt(A.size)      in float64
//
// The reference transforms both images again for every pair; here every segment image is
// transformed once (real-to-complex, half spectrum), and a pair costs one spectrum product, one
// complex-to-real inverse and one reduction kernel that applies the fftshift, the centre crop
// (`roi`), finds the first maximum in row-major order and cuts the box x box fit window — the
// inputs of the reference's scipy curve_fit, which stays on the host (25 numbers per pair).
// The transforms are hipFFT plans (double precision), cached per image size.
#include <hipfft/hipfft.h>

#include <algorithm>
#include <cmath>
#include <map>
#include <tuple>
#include <mutex>
#include <utility>
#include <vector>

#include "pmi_common.h"

#pragma clang fp contract(off)

namespace pmi {
int rcc_fit_peaks(const double *d_rois, const int32_t *h_peaks3, int64_t n_pairs, int box, int64_t Y, int64_t X, int64_t Y_,
                  int64_t X_, double *h_shift_yx, int32_t *h_status);      // peakfit.hip
namespace xc {

struct Plans { hipfftHandle fwd, inv; };
static std::map<std::tuple<int, int64_t, int64_t>, Plans> g_plans;      // a plan belongs to the device it was made on
static std::mutex g_plans_mu;        // pmi_fft_prewarm makes plans from a side thread of the host

static int get_plans(int64_t Y, int64_t X, Plans *out)
{
    std::lock_guard<std::mutex> lk(g_plans_mu);
    int dev = 0;
    PMI_HIP(hipGetDevice(&dev));
    auto key = std::make_tuple(dev, Y, X);
    auto it = g_plans.find(key);
    if (it == g_plans.end()) {
        Plans p;
        if (hipfftPlan2d(&p.fwd, (int)Y, (int)X, HIPFFT_D2Z) != HIPFFT_SUCCESS ||
            hipfftPlan2d(&p.inv, (int)Y, (int)X, HIPFFT_Z2D) != HIPFFT_SUCCESS) {
            set_error("hipfftPlan2d(%lld, %lld) failed", (long long)Y, (long long)X);
            return PMI_ERR_HIP;
        }
        it = g_plans.emplace(key, p).first;
    }
    *out = it->second;
    return PMI_OK;
}

// sums[i] != 0 when image i is not empty (picasso/imageprocess.py:85-86 tests np.sum(image) == 0; the
// images are non-negative renders, so that is "all pixels zero").  grid = (chunks, images); every block
// adds the float64 sum of its chunk of non-negative magnitudes to the image's total.
__global__ __launch_bounds__(256) void sum_kernel(const double *__restrict__ img, int64_t npix, double *__restrict__ sums)
{
    __shared__ double s[256];
    const double *p = img + (int64_t)blockIdx.y * npix;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) acc += fabs(p[i]);
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && s[0] != 0.0) atomicAdd(&sums[blockIdx.y], s[0]);
}

__global__ void product_kernel(const hipfftDoubleComplex *__restrict__ fa, const hipfftDoubleComplex *__restrict__ fb,
                               int64_t n, hipfftDoubleComplex *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double ar = fa[i].x, ai = fa[i].y, br = fb[i].x, bi = -fb[i].y;     // conj(FB)
    out[i].x = ar * br - ai * bi;
    out[i].y = ar * bi + ai * br;
}

// full shifted + scaled correlation image (for pmi_xcorr)
__global__ void shift_scale_kernel(const double *__restrict__ r, int64_t Y, int64_t X, double inv_n, double inv_sqrt,
                                   double *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Y * X) return;
    const int64_t ky = i / X, kx = i - ky * X;
    const int64_t sy = (ky - Y / 2 + Y) % Y, sx = (kx - X / 2 + X) % X;
    out[i] = (r[sy * X + sx] * inv_n) * inv_sqrt;
}

struct PeakOut {           // per pair
    int32_t y_max, x_max;  // first maximum of the cropped correlation, cropped coordinates
    int32_t valid;         // the box x box window lies inside the cropped image
    int32_t pad;
};

// one workgroup per pair: argmax (first in row-major order) over the cropped, shifted correlation
__global__ __launch_bounds__(256) void peak_kernel(const double *__restrict__ r, int64_t Y, int64_t X, int64_t Y_, int64_t X_,
                                                   int64_t cy, int64_t cx, double inv_n, double inv_sqrt, int box,
                                                   PeakOut *__restrict__ out, double *__restrict__ roi)
{
    __shared__ double s_v[256];
    __shared__ long long s_i[256];
    double best = -__builtin_inf();
    long long bi = -1;
    for (int64_t i = threadIdx.x; i < cy * cx; i += 256) {
        const int64_t ky = i / cx + Y_, kx = i % cx + X_;
        const int64_t sy = (ky - Y / 2 + Y) % Y, sx = (kx - X / 2 + X) % X;
        const double v = (r[sy * X + sx] * inv_n) * inv_sqrt;
        if (v > best || bi < 0) { best = v; bi = i; }          // ascending i per thread: keeps the first maximum; NaN never wins
    }
    s_v[threadIdx.x] = best; s_i[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const double v2 = s_v[threadIdx.x + o];
            const long long i2 = s_i[threadIdx.x + o];
            if (i2 >= 0 && (s_i[threadIdx.x] < 0 || v2 > s_v[threadIdx.x] || (v2 == s_v[threadIdx.x] && i2 < s_i[threadIdx.x]))) {
                s_v[threadIdx.x] = v2; s_i[threadIdx.x] = i2;
            }
        }
        __syncthreads();
    }
    const long long idx = s_i[0];
    const int64_t ym = idx / cx, xm = idx % cx;
    const int h = box / 2;
    const bool valid = ym - h >= 0 && ym + h < cy && xm - h >= 0 && xm + h < cx;    // numpy slicing would truncate or wrap otherwise
    if (threadIdx.x == 0) { out->y_max = (int32_t)ym; out->x_max = (int32_t)xm; out->valid = valid ? 1 : 0; out->pad = 0; }
    if ((int)threadIdx.x < box * box) {
        double v = 0.0;
        if (valid) {
            const int64_t ky = ym - h + threadIdx.x / box + Y_, kx = xm - h + threadIdx.x % box + X_;
            const int64_t sy = (ky - Y / 2 + Y) % Y, sx = (kx - X / 2 + X) % X;
            v = (r[sy * X + sx] * inv_n) * inv_sqrt;
        }
        roi[threadIdx.x] = v;
    }
}

}  // namespace xc

void release_fft_plans()
{
    std::lock_guard<std::mutex> lk(xc::g_plans_mu);
    for (auto &kv : xc::g_plans) { hipfftDestroy(kv.second.fwd); hipfftDestroy(kv.second.inv); }
    xc::g_plans.clear();
}

}  // namespace pmi

extern "C" {

// rocFFT compiles the kernels of a plan when the plan is made: 2.5 s for 2048 x 2048, the larger part of a first RCC
// undrift.  A host that knows the frame size early (it localizes the movie first) makes the plans from a side thread
// meanwhile; the correlations then find them in the cache.
int pmi_fft_prewarm(int64_t Y, int64_t X)
{
    using namespace pmi;
    if (pmi_device_count() < 1) { set_error("no HIP device"); return PMI_ERR_NODEVICE; }
    if (Y < 1 || X < 1 || Y > 0x7fffffff || X > 0x7fffffff) { set_error("fft prewarm: bad size"); return PMI_ERR_ARG; }
    xc::Plans pl;
    return xc::get_plans(Y, X, &pl);
}

int pmi_xcorr(const double *image_a, const double *image_b, int64_t Y, int64_t X, double *out)
{
    using namespace pmi;
    if (pmi_device_count() < 1) { set_error("no HIP device"); return PMI_ERR_NODEVICE; }
    if (!image_a || !image_b || !out || Y < 1 || X < 1 || Y > 0x7fffffff || X > 0x7fffffff) { set_error("xcorr: bad arguments"); return PMI_ERR_ARG; }
    xc::Plans pl;
    int rc = xc::get_plans(Y, X, &pl);
    if (rc != PMI_OK) return rc;
    const int64_t npix = Y * X, nspec = Y * (X / 2 + 1);
    void *d_img = nullptr, *d_spec = nullptr;
    if ((rc = scratch(SCR_STAGE_A, (size_t)npix * 8 * 2, &d_img)) != PMI_OK) return rc;
    if ((rc = scratch(SCR_STAGE_B, (size_t)nspec * 16 * 3, &d_spec)) != PMI_OK) return rc;
    double *da = (double *)d_img, *db = da + npix;
    hipfftDoubleComplex *fa = (hipfftDoubleComplex *)d_spec, *fb = fa + nspec, *fp = fb + nspec;
    PMI_HIP(hipMemcpy(da, image_a, (size_t)npix * 8, hipMemcpyHostToDevice));
    PMI_HIP(hipMemcpy(db, image_b, (size_t)npix * 8, hipMemcpyHostToDevice));
    if (hipfftExecD2Z(pl.fwd, da, fa) != HIPFFT_SUCCESS || hipfftExecD2Z(pl.fwd, db, fb) != HIPFFT_SUCCESS) { set_error("hipfftExecD2Z failed"); return PMI_ERR_HIP; }
    hipLaunchKernelGGL(xc::product_kernel, dim3((unsigned)((nspec + 255) / 256)), dim3(256), 0, 0, fa, fb, nspec, fp);
    if (hipfftExecZ2D(pl.inv, fp, da) != HIPFFT_SUCCESS) { set_error("hipfftExecZ2D failed"); return PMI_ERR_HIP; }
    hipLaunchKernelGGL(xc::shift_scale_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, 0, da, Y, X,
                       1.0 / (double)npix, 1.0 / std::sqrt((double)npix), db);
    PMI_HIP(hipGetLastError());
    PMI_HIP(hipMemcpy(out, db, (size_t)npix * 8, hipMemcpyDeviceToHost));
    return PMI_OK;
}

// *d_rois_out (optional): where the fit windows stay resident on the device until the next call that uses the
// thread's scratch bank — pmi_rcc_shifts fits them there without looking the buffer up a second time
static int rcc_pair_list_impl(const double *segments, int64_t n_seg, int64_t Y, int64_t X, int64_t roi, int box,
                              const int32_t *pairs, int64_t n_pairs, int32_t *peak_yx, int32_t *valid, double *fit_rois,
                              int32_t *crop_yx, const double **d_rois_out)
{
    using namespace pmi;
    if (d_rois_out) *d_rois_out = nullptr;
    if (pmi_device_count() < 1) { set_error("no HIP device"); return PMI_ERR_NODEVICE; }
    if (!segments || !peak_yx || !valid || !fit_rois || !crop_yx || (n_pairs > 0 && !pairs)) { set_error("null pointer"); return PMI_ERR_ARG; }
    if (n_seg < 1 || n_pairs < 0 || Y < 1 || X < 1 || Y > 0x7fffffff || X > 0x7fffffff || box < 1 || box > 15 || !(box & 1)) { set_error("rcc: bad arguments"); return PMI_ERR_ARG; }
    for (int64_t p = 0; p < n_pairs; p++)
        if (pairs[2 * p] < 0 || pairs[2 * p] >= n_seg || pairs[2 * p + 1] < 0 || pairs[2 * p + 1] >= n_seg) { set_error("rcc: pair %lld out of range", (long long)p); return PMI_ERR_ARG; }
    xc::Plans pl;
    int rc = xc::get_plans(Y, X, &pl);
    if (rc != PMI_OK) return rc;
    // centre crop of picasso/imageprocess.py:90-104
    int64_t Y_ = 0, X_ = 0;
    if (roi > 0) {
        Y_ = (int64_t)((double)(Y - roi) / 2.0); if (Y_ <= 0) Y_ = 0;
        X_ = (int64_t)((double)(X - roi) / 2.0); if (X_ <= 0) X_ = 0;
    }
    const int64_t cy = Y - 2 * Y_, cx = X - 2 * X_;
    crop_yx[0] = (int32_t)Y_; crop_yx[1] = (int32_t)X_;
    if (n_pairs == 0) return PMI_OK;
    const int64_t npix = Y * X, nspec = Y * (X / 2 + 1);
    void *d_img = nullptr, *d_spec = nullptr, *d_work = nullptr, *d_out = nullptr;
    if ((rc = scratch(SCR_STAGE_A, (size_t)n_seg * npix * 8, &d_img)) != PMI_OK) return rc;
    if ((rc = scratch(SCR_STAGE_B, (size_t)n_seg * nspec * 16, &d_spec)) != PMI_OK) return rc;
    if ((rc = scratch(SCR_STAGE_C, (size_t)nspec * 16 + (size_t)npix * 8 + 64, &d_work)) != PMI_OK) return rc;
    if ((rc = scratch(SCR_STAGE_D, (size_t)n_pairs * (sizeof(xc::PeakOut) + (size_t)box * box * 8) + (size_t)n_seg * 8 + 64, &d_out)) != PMI_OK) return rc;
    double *dseg = (double *)d_img;
    hipfftDoubleComplex *spec = (hipfftDoubleComplex *)d_spec, *prod = (hipfftDoubleComplex *)d_work;
    double *corr = (double *)(prod + nspec);
    double *d_rois = (double *)d_out;
    if (d_rois_out) *d_rois_out = d_rois;
    double *d_sums = d_rois + n_pairs * box * box;
    xc::PeakOut *d_peaks = (xc::PeakOut *)(d_sums + n_seg);
    PMI_HIP(hipMemcpy(dseg, segments, (size_t)n_seg * npix * 8, hipMemcpyHostToDevice));
    PMI_HIP(hipMemsetAsync(d_sums, 0, (size_t)n_seg * 8, 0));
    hipLaunchKernelGGL(xc::sum_kernel, dim3((unsigned)std::min<int64_t>(64, (npix + 255) / 256), (unsigned)n_seg), dim3(256), 0, 0,
                       dseg, npix, d_sums);
    std::vector<char> used((size_t)n_seg, 0);
    for (int64_t p = 0; p < 2 * n_pairs; p++) used[(size_t)pairs[p]] = 1;
    for (int64_t i = 0; i < n_seg; i++)
        if (used[(size_t)i] && hipfftExecD2Z(pl.fwd, dseg + i * npix, spec + i * nspec) != HIPFFT_SUCCESS) { set_error("hipfftExecD2Z failed"); return PMI_ERR_HIP; }
    std::vector<double> sums((size_t)n_seg);
    PMI_HIP(hipMemcpy(sums.data(), d_sums, (size_t)n_seg * 8, hipMemcpyDeviceToHost));
    const double inv_n = 1.0 / (double)npix, inv_sqrt = 1.0 / std::sqrt((double)npix);
    std::vector<int64_t> skipped;
    for (int64_t pidx = 0; pidx < n_pairs; pidx++) {
        const int64_t i = pairs[2 * pidx], j = pairs[2 * pidx + 1];
        if (sums[(size_t)i] == 0.0 || sums[(size_t)j] == 0.0) { skipped.push_back(pidx); continue; }   // shift (0, 0), imageprocess.py:85-86
        hipLaunchKernelGGL(xc::product_kernel, dim3((unsigned)((nspec + 255) / 256)), dim3(256), 0, 0, spec + i * nspec,
                           spec + j * nspec, nspec, prod);
        if (hipfftExecZ2D(pl.inv, prod, corr) != HIPFFT_SUCCESS) { set_error("hipfftExecZ2D failed"); return PMI_ERR_HIP; }
        hipLaunchKernelGGL(xc::peak_kernel, dim3(1), dim3(256), 0, 0, corr, Y, X, Y_, X_, cy, cx, inv_n, inv_sqrt, box,
                           d_peaks + pidx, d_rois + pidx * box * box);
    }
    PMI_HIP(hipGetLastError());
    std::vector<xc::PeakOut> peaks((size_t)n_pairs);
    PMI_HIP(hipMemcpy(peaks.data(), d_peaks, (size_t)n_pairs * sizeof(xc::PeakOut), hipMemcpyDeviceToHost));
    PMI_HIP(hipMemcpy(fit_rois, d_rois, (size_t)n_pairs * box * box * 8, hipMemcpyDeviceToHost));
    for (int64_t p = 0; p < n_pairs; p++) {
        peak_yx[2 * p] = peaks[(size_t)p].y_max; peak_yx[2 * p + 1] = peaks[(size_t)p].x_max;
        valid[p] = peaks[(size_t)p].valid;
    }
    for (int64_t p : skipped) {
        peak_yx[2 * p] = 0; peak_yx[2 * p + 1] = 0; valid[p] = -1;       // -1: an empty image, the shift is (0, 0) by definition
        for (int k = 0; k < box * box; k++) fit_rois[p * box * box + k] = 0.0;
    }
    return PMI_OK;
}

int pmi_rcc_pair_list(const double *segments, int64_t n_seg, int64_t Y, int64_t X, int64_t roi, int box,
                      const int32_t *pairs, int64_t n_pairs, int32_t *peak_yx, int32_t *valid, double *fit_rois,
                      int32_t *crop_yx)
{
    return rcc_pair_list_impl(segments, n_seg, Y, X, roi, box, pairs, n_pairs, peak_yx, valid, fit_rois, crop_yx, nullptr);
}

// all of get_image_shift (imageprocess.py:53-161) for a list of pairs: correlation, crop, peak, window AND the bounded
// Gaussian fit of the window (csrc/peakfit.hip), -> shift_yx (n_pairs, 2) = (-yc, -xc)
int pmi_rcc_shifts(const double *segments, int64_t n_seg, int64_t Y, int64_t X, int64_t roi, int box,
                   const int32_t *pairs, int64_t n_pairs, double *shift_yx, int32_t *fit_status)
{
    using namespace pmi;
    if (!shift_yx || !fit_status) { set_error("null pointer"); return PMI_ERR_ARG; }
    std::vector<int32_t> peak((size_t)std::max<int64_t>(n_pairs, 1) * 2), valid((size_t)std::max<int64_t>(n_pairs, 1));
    std::vector<double> rois((size_t)std::max<int64_t>(n_pairs, 1) * box * box);
    int32_t crop[2] = {0, 0};
    const double *d_rois = nullptr;
    int rc = rcc_pair_list_impl(segments, n_seg, Y, X, roi, box, pairs, n_pairs, peak.data(), valid.data(), rois.data(), crop, &d_rois);
    if (rc != PMI_OK || n_pairs == 0) return rc;
    if (!d_rois) { set_error("rcc: no resident fit windows"); return PMI_ERR_HIP; }
    std::vector<int32_t> pk3((size_t)n_pairs * 3);
    for (int64_t p = 0; p < n_pairs; p++) { pk3[(size_t)p * 3] = peak[(size_t)p * 2]; pk3[(size_t)p * 3 + 1] = peak[(size_t)p * 2 + 1]; pk3[(size_t)p * 3 + 2] = valid[(size_t)p]; }
    // empty-image pairs were zeroed on the host only: their `valid` is -1 and the kernel does not read their windows
    return rcc_fit_peaks(d_rois, pk3.data(), n_pairs, box, Y, X, crop[0], crop[1], shift_yx, fit_status);
}

int pmi_rcc_pairs(const double *segments, int64_t n_seg, int64_t Y, int64_t X, int64_t roi, int box,
                  int32_t *peak_yx, int32_t *valid, double *fit_rois, int32_t *crop_yx)
{
    if (n_seg < 2) { pmi::set_error("rcc: needs at least two segments"); return PMI_ERR_ARG; }
    std::vector<int32_t> pairs;                       // every pair i < j, in the order of imageprocess.py:197-205
    for (int64_t i = 0; i < n_seg - 1; i++)
        for (int64_t j = i + 1; j < n_seg; j++) { pairs.push_back((int32_t)i); pairs.push_back((int32_t)j); }
    return pmi_rcc_pair_list(segments, n_seg, Y, X, roi, box, pairs.data(), (int64_t)pairs.size() / 2, peak_yx, valid,
                             fit_rois, crop_yx);
}

}  // extern "C"
