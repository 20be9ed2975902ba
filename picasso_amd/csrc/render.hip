// render.hip — super-resolution rendering of a localization table
// (picasso/render.py:177-232 _render_setup, :451-467 _fill, :494-575 _draw_gaussian_loc /
//  _fill_gaussian, :798-853 _render_hist, :1020-1070 _render_gaussian without rotation).
//
// Histogram: one atomic float add of 1.0 per localization — exact, counts stay below 2^24.
//
// Gaussian: the reference adds every localization's separable footprint into the float32 image
// one after the other, so each pixel's value depends on the ORDER of the additions.  A scatter
// with float atomics would be correct only up to rounding; this kernel reproduces the order
// instead:
//   1. prep: per localization, image coordinates (float64, numba's promotion of the float32
//      columns), blur widths (float32), the clipped +-3 sigma footprint, the number of 32x32
//      image tiles it touches;
//   2. exclusive scan of those counts, emit (tile, localization) pairs in localization order,
//      stable radix sort by tile  ->  every tile gets its localizations in table order;
//   3. one workgroup per tile: 256 threads x 4 pixels in registers; the tile's list is walked
//      in chunks of 32 localizations whose 1-D profiles (float64 exp, rounded to float32 like
//      the reference's gx / gy arrays) are built cooperatively in LDS, then every pixel adds
//      gy*gx of each localization in order (0 outside the footprint: x + 0 == x).
// The result is the reference's image bit for bit, up to a 1-ulp difference between the
// device's and glibc's float64 exp surviving the rounding to float32.
#include <algorithm>
#include <cmath>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "pmi_common.h"

#pragma clang fp contract(off)

namespace pmi {
namespace rend {

constexpr int TILE = 32;
constexpr int CHUNK = 32;          // localizations per LDS batch in the tile kernel

struct View {
    double oversampling, y_min, x_min, y_max, x_max;
    float os_f, min_blur_f;
    int iso;                 // gaussian_iso: both widths = their mean (render.py:1148-1216)
    int64_t ny, nx;
    int tiles_x, tiles_y;
};

struct Loc {              // one localization in view, image coordinates
    double x, y;
    float sx, sy;
    int i_min, i_max, j_min, j_max;      // clipped footprint [min, max); empty when max <= min
};

__device__ __forceinline__ float np_maxf(float a, float b) { return (a != a) ? a : ((b != b) ? b : (a > b ? a : b)); }

// float64 -> int32 as the reference's compiled code does it on x86-64 (cvttsd2si): truncation, and
// INT_MIN for NaN or out-of-range values — a NaN precision then yields an empty footprint.
__device__ __forceinline__ int32_t to_int32(double v)
{
    return (v != v || v >= 2147483648.0 || v < -2147483648.0) ? (int32_t)0x80000000 : (int32_t)v;
}

__device__ __forceinline__ bool in_view(const View &v, float xf, float yf)
{
    const double xd = (double)xf, yd = (double)yf;
    return xd > v.x_min && yd > v.y_min && xd < v.x_max && yd < v.y_max;
}

__global__ void hist_kernel(const float *__restrict__ x, const float *__restrict__ y, int64_t N, View v,
                            float *__restrict__ image, unsigned long long *__restrict__ n_rendered)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool ok = false;
    if (i < N && in_view(v, x[i], y[i])) {
        ok = true;
        const int32_t xi = to_int32(v.oversampling * ((double)x[i] - v.x_min));      // astype(int32): truncation
        const int32_t yi = to_int32(v.oversampling * ((double)y[i] - v.y_min));
        atomicAdd(&image[(int64_t)yi * v.nx + xi], 1.0f);
    }
    const unsigned long long bal = __ballot(ok);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(n_rendered, (unsigned long long)__popcll(bal));
}

// footprint of picasso/render.py:505-524, including its asymmetric "+ 1"
__global__ void prep_kernel(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ lpx,
                            const float *__restrict__ lpy, int64_t N, View v, Loc *__restrict__ locs,
                            unsigned *__restrict__ ntiles, unsigned long long *__restrict__ n_rendered)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool ok = false;
    if (i < N) {
        unsigned cnt = 0;
        Loc L = {};
        if (in_view(v, x[i], y[i])) {
            ok = true;
            L.x = v.oversampling * ((double)x[i] - v.x_min);
            L.y = v.oversampling * ((double)y[i] - v.y_min);
            L.sx = v.os_f * np_maxf(lpx[i], v.min_blur_f);
            L.sy = v.os_f * np_maxf(lpy[i], v.min_blur_f);
            if (v.iso) { L.sy = (L.sy + L.sx) / 2.0f; L.sx = L.sy; }
            const double max_y_off = 3.0 * (double)L.sy, max_x_off = 3.0 * (double)L.sx;
            int64_t i_min = to_int32(L.y - max_y_off);
            if (i_min < 0) i_min = 0;
            int64_t i_max = to_int32(L.y + max_y_off + 1);
            if (i_max > v.ny) i_max = v.ny;
            int64_t j_min = to_int32(L.x - max_x_off);
            if (j_min < 0) j_min = 0;
            int64_t j_max = (int64_t)to_int32(L.x + max_x_off) + 1;
            if (j_max > v.nx) j_max = v.nx;
            L.i_min = (int)i_min; L.i_max = (int)i_max; L.j_min = (int)j_min; L.j_max = (int)j_max;
            if (i_max > i_min && j_max > j_min)
                cnt = (unsigned)(((i_max - 1) / TILE - i_min / TILE + 1) * ((j_max - 1) / TILE - j_min / TILE + 1));
        }
        locs[i] = L;
        ntiles[i] = cnt;
    }
    const unsigned long long bal = __ballot(ok);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(n_rendered, (unsigned long long)__popcll(bal));
}

__global__ void emit_kernel(const Loc *__restrict__ locs, const unsigned *__restrict__ ntiles,
                            const unsigned *__restrict__ offset, int64_t N, int tiles_x,
                            unsigned *__restrict__ keys, unsigned *__restrict__ vals)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N || ntiles[i] == 0) return;
    const Loc L = locs[i];
    unsigned o = offset[i];
    for (int ty = L.i_min / TILE; ty <= (L.i_max - 1) / TILE; ty++)
        for (int tx = L.j_min / TILE; tx <= (L.j_max - 1) / TILE; tx++) {
            keys[o] = (unsigned)(ty * tiles_x + tx);
            vals[o] = (unsigned)i;
            o++;
        }
}

__global__ void tile_bounds_kernel(const unsigned *__restrict__ keys, unsigned E, unsigned *__restrict__ start,
                                   unsigned *__restrict__ end)
{
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const unsigned k = keys[e];
    if (e == 0 || keys[e - 1] != k) start[k] = e;
    if (e + 1 == E || keys[e + 1] != k) end[k] = e + 1;
}

__global__ __launch_bounds__(256) void tile_kernel(const Loc *__restrict__ locs, const unsigned *__restrict__ vals,
                                                   const unsigned *__restrict__ start, const unsigned *__restrict__ end,
                                                   View v, float *__restrict__ image)
{
    __shared__ float s_gx[CHUNK][TILE], s_gy[CHUNK][TILE];
    const int tile = blockIdx.x;
    const unsigned e0 = start[tile], e1 = end[tile];
    const int ty = tile / v.tiles_x, tx = tile - ty * v.tiles_x;
    const int row0 = ty * TILE, col0 = tx * TILE;
    const int tid = threadIdx.x;
    const int pj = tid & (TILE - 1), pi = tid >> 5;           // pixel (pi + 8 k, pj), k = 0..3
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (unsigned c0 = e0; c0 < e1; c0 += CHUNK) {
        const int nloc = (int)min((unsigned)CHUNK, e1 - c0);
        // profiles: 2 * TILE values per localization, CHUNK localizations -> 8 values per thread
        for (int q = tid; q < CHUNK * 2 * TILE; q += 256) {
            const int l = q / (2 * TILE), r = q - l * (2 * TILE);
            float val = 0.f;
            if (l < nloc) {
                const Loc L = locs[vals[c0 + l]];
                if (r < TILE) {                                  // gx at image column col0 + r
                    const int j = col0 + r;
                    if (j >= L.j_min && j < L.j_max) {
                        const double inv_2sx2 = 1.0 / (2.0 * (double)L.sx * (double)L.sx);
                        const double dx = (double)j + 0.5 - L.x;
                        val = (float)exp(-dx * dx * inv_2sx2);
                    }
                } else {                                         // gy at image row row0 + r - TILE
                    const int i = row0 + r - TILE;
                    if (i >= L.i_min && i < L.i_max) {
                        const double inv_2sy2 = 1.0 / (2.0 * (double)L.sy * (double)L.sy);
                        const double norm = 1.0 / (6.283185307179586 * (double)L.sx * (double)L.sy);
                        const double dy = (double)i + 0.5 - L.y;
                        val = (float)(norm * exp(-dy * dy * inv_2sy2));
                    }
                }
            }
            if (r < TILE) s_gx[l][r] = val; else s_gy[l][r - TILE] = val;
        }
        __syncthreads();
        for (int l = 0; l < nloc; l++) {
            const float gx = s_gx[l][pj];
#pragma unroll
            for (int k = 0; k < 4; k++) acc[k] = acc[k] + s_gy[l][pi + 8 * k] * gx;
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = row0 + pi + 8 * k, j = col0 + pj;
        if (i < v.ny && j < v.nx) image[(int64_t)i * v.nx + j] = acc[k];
    }
}

static int make_view(double oversampling, double y_min, double x_min, double y_max, double x_max, double min_blur,
                     int64_t ny, int64_t nx, View *v)
{
    const int64_t ey = (int64_t)std::ceil(oversampling * (y_max - y_min));
    const int64_t ex = (int64_t)std::ceil(oversampling * (x_max - x_min));
    if (ny != ey || nx != ex) { set_error("render: image is %lld x %lld but the viewport needs %lld x %lld", (long long)ny, (long long)nx, (long long)ey, (long long)ex); return PMI_ERR_ARG; }
    if (ny <= 0 || nx <= 0 || ny > 0x7fffffff || nx > 0x7fffffff) { set_error("render: bad image size"); return PMI_ERR_ARG; }
    v->oversampling = oversampling; v->y_min = y_min; v->x_min = x_min; v->y_max = y_max; v->x_max = x_max;
    v->os_f = (float)oversampling; v->min_blur_f = (float)min_blur; v->iso = 0;
    v->ny = ny; v->nx = nx;
    v->tiles_x = (int)((nx + TILE - 1) / TILE); v->tiles_y = (int)((ny + TILE - 1) / TILE);
    if ((int64_t)v->tiles_x * v->tiles_y > 0x7fffffffLL) { set_error("render: image too large"); return PMI_ERR_ARG; }
    return PMI_OK;
}

}  // namespace rend
}  // namespace pmi

extern "C" {

int pmi_render_dims(double oversampling, double y_min, double x_min, double y_max, double x_max, int64_t *ny, int64_t *nx)
{
    if (!ny || !nx) { pmi::set_error("null pointer"); return PMI_ERR_ARG; }
    *ny = (int64_t)std::ceil(oversampling * (y_max - y_min));
    *nx = (int64_t)std::ceil(oversampling * (x_max - x_min));
    return PMI_OK;
}

int pmi_render_hist_dev(const float *d_x, const float *d_y, int64_t N, double oversampling, double y_min, double x_min,
                        double y_max, double x_max, float *d_image, int64_t ny, int64_t nx, int64_t *d_n_rendered,
                        void *stream)
{
    using namespace pmi;
    rend::View v;
    int rc = rend::make_view(oversampling, y_min, x_min, y_max, x_max, 0.0, ny, nx, &v);
    if (rc != PMI_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    PMI_HIP(hipMemsetAsync(d_image, 0, (size_t)ny * nx * sizeof(float), s));
    PMI_HIP(hipMemsetAsync(d_n_rendered, 0, 8, s));
    if (N > 0) {
        hipLaunchKernelGGL(rend::hist_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, d_x, d_y, N, v, d_image,
                           (unsigned long long *)d_n_rendered);
        PMI_HIP(hipGetLastError());
    }
    return PMI_OK;
}

int pmi_render_gaussian_dev(const float *d_x, const float *d_y, const float *d_lpx, const float *d_lpy, int64_t N,
                            double oversampling, double y_min, double x_min, double y_max, double x_max,
                            double min_blur_width, int iso, float *d_image, int64_t ny, int64_t nx,
                            int64_t *d_n_rendered, void *stream)
{
    using namespace pmi;
    rend::View v;
    int rc = rend::make_view(oversampling, y_min, x_min, y_max, x_max, min_blur_width, ny, nx, &v);
    if (rc != PMI_OK) return rc;
    v.iso = iso ? 1 : 0;
    if (N > 0x7fffffffLL) { set_error("render: too many localizations"); return PMI_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    PMI_HIP(hipMemsetAsync(d_n_rendered, 0, 8, s));
    const int64_t ntile = (int64_t)v.tiles_x * v.tiles_y;
    if (N == 0) { PMI_HIP(hipMemsetAsync(d_image, 0, (size_t)ny * nx * sizeof(float), s)); return PMI_OK; }

    // scratch: Loc[N], ntiles[N], offset[N], tile start/end
    void *p_loc = nullptr, *p_cnt = nullptr, *p_tile = nullptr;
    if ((rc = scratch(SCR_STAGE_B, (size_t)N * sizeof(rend::Loc), &p_loc)) != PMI_OK) return rc;
    if ((rc = scratch(SCR_STAGE_C, (size_t)N * 8 + 64, &p_cnt)) != PMI_OK) return rc;
    if ((rc = scratch(SCR_FRAME_COUNT, (size_t)ntile * 8, &p_tile)) != PMI_OK) return rc;
    rend::Loc *locs = (rend::Loc *)p_loc;
    unsigned *ntiles = (unsigned *)p_cnt, *offset = ntiles + N;
    unsigned *start = (unsigned *)p_tile, *end = start + ntile;
    const unsigned nb = (unsigned)((N + 255) / 256);
    hipLaunchKernelGGL(rend::prep_kernel, dim3(nb), dim3(256), 0, s, d_x, d_y, d_lpx, d_lpy, N, v, locs, ntiles,
                       (unsigned long long *)d_n_rendered);
    PMI_HIP(hipGetLastError());

    // exclusive scan -> offsets; the total decides the size of the pair buffers (one small D2H)
    size_t tmp_bytes = 0;
    PMI_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, ntiles, offset, 0u, (size_t)N, rocprim::plus<unsigned>(), s));
    void *p_tmp = nullptr;
    if ((rc = scratch(SCR_STAGE_D, tmp_bytes + 64, &p_tmp)) != PMI_OK) return rc;
    PMI_HIP(rocprim::exclusive_scan(p_tmp, tmp_bytes, ntiles, offset, 0u, (size_t)N, rocprim::plus<unsigned>(), s));
    unsigned last_off = 0, last_cnt = 0;
    PMI_HIP(hipMemcpyAsync(&last_off, offset + (N - 1), 4, hipMemcpyDeviceToHost, s));
    PMI_HIP(hipMemcpyAsync(&last_cnt, ntiles + (N - 1), 4, hipMemcpyDeviceToHost, s));
    PMI_HIP(hipStreamSynchronize(s));
    const uint64_t E = (uint64_t)last_off + last_cnt;
    if (E > 0x7fffffffULL) { set_error("render: footprints cover %llu tiles in total, more than this kernel handles", (unsigned long long)E); return PMI_ERR_ARG; }

    PMI_HIP(hipMemsetAsync(start, 0, (size_t)ntile * 8, s));
    if (E > 0) {
        void *p_pairs = nullptr;
        if ((rc = scratch(SCR_RECORDS, (size_t)E * 16 + 64, &p_pairs)) != PMI_OK) return rc;
        unsigned *keys = (unsigned *)p_pairs, *vals = keys + E, *keys2 = vals + E, *vals2 = keys2 + E;
        hipLaunchKernelGGL(rend::emit_kernel, dim3(nb), dim3(256), 0, s, locs, ntiles, offset, N, v.tiles_x, keys, vals);
        PMI_HIP(hipGetLastError());
        int bits = 1;
        while ((1LL << bits) < ntile) bits++;
        size_t sort_bytes = 0;
        PMI_HIP(rocprim::radix_sort_pairs(nullptr, sort_bytes, keys, keys2, vals, vals2, (size_t)E, 0, bits, s));
        void *p_sort = nullptr;
        if ((rc = scratch(SCR_RECORDS2, sort_bytes + 64, &p_sort)) != PMI_OK) return rc;
        PMI_HIP(rocprim::radix_sort_pairs(p_sort, sort_bytes, keys, keys2, vals, vals2, (size_t)E, 0, bits, s));   // stable
        hipLaunchKernelGGL(rend::tile_bounds_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, s, keys2, (unsigned)E, start, end);
        PMI_HIP(hipGetLastError());
        hipLaunchKernelGGL(rend::tile_kernel, dim3((unsigned)ntile), dim3(256), 0, s, locs, vals2, start, end, v, d_image);
    } else {
        PMI_HIP(hipMemsetAsync(d_image, 0, (size_t)ny * nx * sizeof(float), s));
    }
    PMI_HIP(hipGetLastError());
    return PMI_OK;
}

static int render_host(const float *x, const float *y, const float *lpx, const float *lpy, int64_t N, double oversampling,
                       double y_min, double x_min, double y_max, double x_max, double min_blur_width, int iso, bool gaussian,
                       float *image, int64_t ny, int64_t nx, int64_t *n_rendered)
{
    using namespace pmi;
    if (pmi_device_count() < 1) { set_error("no HIP device"); return PMI_ERR_NODEVICE; }
    if (!image || !n_rendered || (N > 0 && (!x || !y || (gaussian && (!lpx || !lpy))))) { set_error("null pointer"); return PMI_ERR_ARG; }
    void *d_in = nullptr, *d_img = nullptr;
    int rc;
    const size_t col = (size_t)std::max<int64_t>(N, 1) * 4;
    if ((rc = scratch(SCR_STAGE_A, col * 4 + 64, &d_in)) != PMI_OK) return rc;
    const size_t img_bytes = (((size_t)ny * nx * 4 + 7) / 8) * 8;
    if ((rc = scratch(SCR_IDS, img_bytes + 8, &d_img)) != PMI_OK) return rc;
    float *dx = (float *)d_in, *dy = dx + N, *dlx = dy + N, *dly = dlx + N;
    int64_t *dn = (int64_t *)((char *)d_img + img_bytes);
    if (N > 0) {
        PMI_HIP(hipMemcpy(dx, x, (size_t)N * 4, hipMemcpyHostToDevice));
        PMI_HIP(hipMemcpy(dy, y, (size_t)N * 4, hipMemcpyHostToDevice));
        if (gaussian) {
            PMI_HIP(hipMemcpy(dlx, lpx, (size_t)N * 4, hipMemcpyHostToDevice));
            PMI_HIP(hipMemcpy(dly, lpy, (size_t)N * 4, hipMemcpyHostToDevice));
        }
    }
    rc = gaussian ? pmi_render_gaussian_dev(dx, dy, dlx, dly, N, oversampling, y_min, x_min, y_max, x_max, min_blur_width, iso,
                                            (float *)d_img, ny, nx, dn, nullptr)
                  : pmi_render_hist_dev(dx, dy, N, oversampling, y_min, x_min, y_max, x_max, (float *)d_img, ny, nx, dn, nullptr);
    if (rc != PMI_OK) return rc;
    PMI_HIP(hipMemcpy(image, d_img, (size_t)ny * nx * 4, hipMemcpyDeviceToHost));
    PMI_HIP(hipMemcpy(n_rendered, dn, 8, hipMemcpyDeviceToHost));
    return PMI_OK;
}

int pmi_render_hist(const float *x, const float *y, int64_t N, double oversampling, double y_min, double x_min,
                    double y_max, double x_max, float *image, int64_t ny, int64_t nx, int64_t *n_rendered)
{
    return render_host(x, y, nullptr, nullptr, N, oversampling, y_min, x_min, y_max, x_max, 0.0, 0, false, image, ny, nx, n_rendered);
}

int pmi_render_gaussian(const float *x, const float *y, const float *lpx, const float *lpy, int64_t N,
                        double oversampling, double y_min, double x_min, double y_max, double x_max,
                        double min_blur_width, int iso, float *image, int64_t ny, int64_t nx, int64_t *n_rendered)
{
    return render_host(x, y, lpx, lpy, N, oversampling, y_min, x_min, y_max, x_max, min_blur_width, iso, true, image, ny, nx, n_rendered);
}

}  // extern "C"
