/*
 * picasso_oracle.c — CPU restatement of the reference's localization hot path.
 *
 * TEST INFRASTRUCTURE.  This file is the parity oracle and the timed CPU
 * baseline ("port").  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product path (picasso_amd/) never does.
 *
 * It restates, in plain C, the algorithm of these reference functions
 * (paths relative to the reference tree, jungmannlab/picasso v0.10.3):
 *
 *   picasso/localize.py:97-134    _local_maxima
 *   picasso/localize.py:153-181   _gradient_at
 *   picasso/localize.py:202-244   _net_gradient
 *   picasso/localize.py:247-292   identify_in_image
 *   picasso/localize.py:295-337   identify_in_frame (ROI crop, float32 cast)
 *   picasso/localize.py:340-421   identify_by_frame_number (frame bounds)
 *   picasso/localize.py:917-931   _cut_spots_numba
 *   picasso/localize.py:1101-1112 _to_photons
 *   picasso/gaussmle.py:28-168    initial parameters
 *   picasso/gaussmle.py:268-383   integrated-Gaussian model and derivatives
 *   picasso/gaussmle.py:533-742   _mlefit_sigma, _update_theta_sigma, CRLB
 *   picasso/gaussmle.py:745-954   _mlefit_sigmaxy, _update_theta_sigmaxy, CRLB
 *
 * Arithmetic follows numba's type promotion, which is what the reference
 * executes in production: int64 (op) float32 -> float64, float64 literal (op)
 * float32 -> float64, float32 ** int -> float32, results rounded to float32
 * only where the reference stores into a float32 array.  Each such place is
 * marked "f32 store".  Parity status: pinned against (1) golden vectors
 * minted by executing the reference's own source under NumPy semantics
 * (tests/golden/make_goldens.py; agreement <= 1e-4 px, iterations +-1 on
 * borderline convergence) and (2) the real-numba
 * identification table the reference bundles (tests/golden/
 * numba_identifications_testdata.npz; bit-exact).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAX_BOX 33
#define ORC_MAX_PIX (ORC_MAX_BOX * ORC_MAX_BOX)

enum { ORC_U16 = 0, ORC_U8 = 1, ORC_I16 = 2, ORC_U32 = 3, ORC_I32 = 4, ORC_F32 = 5 };
enum { ORC_SIGMA = 0, ORC_SIGMAXY = 1 };

static inline float px_as_f32(const void *base, int dtype, int64_t idx)
{
    switch (dtype) {
    case ORC_U16: return (float)((const uint16_t *)base)[idx];
    case ORC_U8:  return (float)((const uint8_t *)base)[idx];
    case ORC_I16: return (float)((const int16_t *)base)[idx];
    case ORC_U32: return (float)((const uint32_t *)base)[idx];
    case ORC_I32: return (float)((const int32_t *)base)[idx];
    default:      return ((const float *)base)[idx];
    }
}

/* numpy maximum/minimum: NaN-propagating */
static inline double np_max(double a, double b) { return (a != a) ? a : ((b != b) ? b : (a > b ? a : b)); }
static inline double np_min(double a, double b) { return (a != a) ? a : ((b != b) ? b : (a < b ? a : b)); }
static inline float np_maxf(float a, float b) { return (a != a) ? a : ((b != b) ? b : (a > b ? a : b)); }
static inline float np_minf(float a, float b) { return (a != a) ? a : ((b != b) ? b : (a < b ? a : b)); }
static inline float np_signf(float a) { return (a != a) ? a : (a > 0.0f ? 1.0f : (a < 0.0f ? -1.0f : 0.0f)); }

/* ------------------------------------------------------------------------
 * identify  (picasso/localize.py:97-134, 202-244, 247-292)
 * ---------------------------------------------------------------------- */

/* unit vectors ux[k][l] = (h-l)/r, uy[k][l] = (h-k)/r in float32
 * (picasso/localize.py:279-286; float32 array ** 2, sqrt and /= all stay
 * float32 under numba's mixed-input ufunc loop matching). */
void orc_unit_vectors(int box, float *ux, float *uy)
{
    int h = box / 2;
    for (int k = 0; k < box; k++)
        for (int l = 0; l < box; l++) {
            float vx = (float)(h - l), vy = (float)(h - k);
            float n2 = vx * vx + vy * vy;
            float n = sqrtf(n2);
            ux[k * box + l] = vx / n; /* centre: 0/0 = NaN, never read */
            uy[k * box + l] = vy / n;
        }
}

/* One frame, already cropped to the ROI and cast to float32.
 * Appends (y, x, ng) in np.where order (y-major).  Returns the number found
 * (may exceed cap; only the first cap are stored). */
static int64_t identify_image(const float *img, int Y, int X, int box, double min_ng,
                              const float *ux, const float *uy,
                              int64_t *oy, int64_t *ox, float *ong, int64_t cap)
{
    int h = box / 2;
    int64_t n = 0;
    for (int i = h; i < Y - (h + 1); i++) {          /* localize.py:122 */
        for (int j = h; j < X - (h + 1); j++) {      /* localize.py:123 */
            /* np.argmax over the window: first maximum in row-major order,
             * NaN counts as the maximum (numpy semantics). */
            float best = img[(int64_t)(i - h) * X + (j - h)];
            int bk = 0, bl = 0;
            for (int k = 0; k < box; k++)
                for (int l = 0; l < box; l++) {
                    float v = img[(int64_t)(i - h + k) * X + (j - h + l)];
                    if (v > best || (v != v && best == best)) { best = v; bk = k; bl = l; }
                }
            if (bk != h || bl != h) continue;
            /* net gradient, float32 accumulator, k then m (localize.py:233-243).
             * Negative indices wrap like numba's (row/col -1 -> last). */
            float ng = 0.0f;
            for (int k_index = 0; k_index < box; k_index++) {
                int k = i - h + k_index;
                for (int l_index = 0; l_index < box; l_index++) {
                    int m = j - h + l_index;
                    if (k == i && m == j) continue;
                    int km1 = k - 1 < 0 ? k - 1 + Y : k - 1;
                    int mm1 = m - 1 < 0 ? m - 1 + X : m - 1;
                    float gy = img[(int64_t)(k + 1) * X + m] - img[(int64_t)km1 * X + m];
                    float gx = img[(int64_t)k * X + (m + 1)] - img[(int64_t)k * X + mm1];
                    float t1 = gy * uy[k_index * box + l_index];
                    float t2 = gx * ux[k_index * box + l_index];
                    float s = t1 + t2;
                    ng = ng + s;
                }
            }
            if ((double)ng > min_ng) {               /* localize.py:288, strict */
                if (n < cap) { oy[n] = i; ox[n] = j; ong[n] = ng; }
                n++;
            }
        }
    }
    return n;
}

/* _net_gradient as a function of its own (picasso/localize.py:202-244): n pixels of one float32 image,
 * caller-supplied unit vectors; every index expression wraps on its own when negative, like numba's
 * unchecked indexing.  Returns 1 if a window would reach past the far edge (undefined in the reference). */
int orc_net_gradient(const float *img, int64_t Y, int64_t X, const int32_t *py, const int32_t *px, int64_t n, int box,
                     const float *uy, const float *ux, float *out)
{
    int h = box / 2;
    for (int64_t i = 0; i < n; i++) {
        int yi = py[i], xi = px[i];
        if (yi + h + 1 >= Y || xi + h + 1 >= X || yi - h - 1 < -Y || xi - h - 1 < -X) return 1;
        float ng = 0.0f;
        for (int kk = 0; kk < box; kk++) {
            int k = yi - h + kk;
            for (int ll = 0; ll < box; ll++) {
                int m = xi - h + ll;
                if (k == yi && m == xi) continue;
#define ORC_WRAP(v, N) ((v) < 0 ? (v) + (N) : (v))
                float gy = img[ORC_WRAP(k + 1, Y) * X + ORC_WRAP(m, X)] - img[ORC_WRAP(k - 1, Y) * X + ORC_WRAP(m, X)];
                float gx = img[ORC_WRAP(k, Y) * X + ORC_WRAP(m + 1, X)] - img[ORC_WRAP(k, Y) * X + ORC_WRAP(m - 1, X)];
#undef ORC_WRAP
                float t1 = gy * uy[kk * box + ll];
                float t2 = gx * ux[kk * box + ll];
                float s = t1 + t2;
                ng = ng + s;
            }
        }
        out[i] = ng;
    }
    return 0;
}

/* Whole movie.  roi = {y0, x0, y1, x1} (already normalised to the frame) or
 * NULL.  Frames outside [f_lo, f_hi] (inclusive, localize.py:401) are
 * skipped.  Output is ordered by frame, then y, then x.  Returns 0, or 1 if
 * the capacity was too small (out_n then holds the needed count). */
int orc_identify(const void *movie, int dtype, int64_t F, int64_t Y, int64_t X,
                 int box, double min_ng, const int64_t *roi, int64_t f_lo, int64_t f_hi,
                 int64_t *out_frame, int64_t *out_y, int64_t *out_x, float *out_ng,
                 int64_t cap, int64_t *out_n, int nthreads)
{
    if (box < 1 || box > ORC_MAX_BOX || (box & 1) == 0) return -1;
    int64_t y0 = 0, x0 = 0, y1 = Y, x1 = X;
    if (roi) { y0 = roi[0]; x0 = roi[1]; y1 = roi[2]; x1 = roi[3]; }
    int cy = (int)(y1 - y0), cx = (int)(x1 - x0);
    if (cy < 0) cy = 0;
    if (cx < 0) cx = 0;
    float ux[ORC_MAX_PIX], uy[ORC_MAX_PIX];
    orc_unit_vectors(box, ux, uy);
    if (f_lo < 0) f_lo = 0;
    if (f_hi > F - 1) f_hi = F - 1;
    int64_t nf = f_hi - f_lo + 1;
    if (nf <= 0 || cy == 0 || cx == 0) { *out_n = 0; return 0; }

    int h = box / 2;
    /* exact upper bound on maxima per frame: centres are > h apart */
    int64_t per_frame_cap = ((int64_t)cy / (h + 1) + 1) * ((int64_t)cx / (h + 1) + 1);
    int64_t *cnt = (int64_t *)calloc((size_t)nf, sizeof(int64_t));
    int64_t **fy = (int64_t **)calloc((size_t)nf, sizeof(int64_t *));
    int64_t **fx = (int64_t **)calloc((size_t)nf, sizeof(int64_t *));
    float **fg = (float **)calloc((size_t)nf, sizeof(float *));
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
#endif
    for (int64_t fi = 0; fi < nf; fi++) {
        int64_t f = f_lo + fi;
        float *img = (float *)malloc(sizeof(float) * (size_t)cy * (size_t)cx);
        for (int r = 0; r < cy; r++)
            for (int c = 0; c < cx; c++)
                img[(int64_t)r * cx + c] = px_as_f32(movie, dtype, (f * Y + (y0 + r)) * X + (x0 + c));
        int64_t *ty = (int64_t *)malloc(sizeof(int64_t) * (size_t)per_frame_cap);
        int64_t *tx = (int64_t *)malloc(sizeof(int64_t) * (size_t)per_frame_cap);
        float *tg = (float *)malloc(sizeof(float) * (size_t)per_frame_cap);
        cnt[fi] = identify_image(img, cy, cx, box, min_ng, ux, uy, ty, tx, tg, per_frame_cap);
        fy[fi] = ty; fx[fi] = tx; fg[fi] = tg;
        free(img);
    }
    int64_t n = 0;
    for (int64_t fi = 0; fi < nf; fi++) {
        for (int64_t q = 0; q < cnt[fi]; q++) {
            if (n < cap) {
                out_frame[n] = f_lo + fi;
                out_y[n] = fy[fi][q] + y0;           /* localize.py:334-336 */
                out_x[n] = fx[fi][q] + x0;
                out_ng[n] = fg[fi][q];
            }
            n++;
        }
        free(fy[fi]); free(fx[fi]); free(fg[fi]);
    }
    free(cnt); free(fy); free(fx); free(fg);
    *out_n = n;
    return n > cap ? 1 : 0;
}

/* ------------------------------------------------------------------------
 * get_spots = _cut_spots_numba + _to_photons
 * (picasso/localize.py:917-931, 1101-1112)
 * ---------------------------------------------------------------------- */
int orc_get_spots(const void *movie, int dtype, int64_t F, int64_t Y, int64_t X,
                  const int64_t *frame, const int64_t *y, const int64_t *x, int64_t N,
                  int box, double baseline, double sensitivity, double gain, float *spots)
{
    (void)F;
    int r = box / 2;
    float b = (float)baseline, s = (float)sensitivity, g = (float)gain;
    for (int64_t i = 0; i < N; i++)
        for (int a = 0; a < box; a++)
            for (int c = 0; c < box; c++) {
                float v = px_as_f32(movie, dtype, (frame[i] * Y + (y[i] - r + a)) * X + (x[i] - r + c));
                float t = v - b;      /* float32 array arithmetic, in this order */
                t = t * s;
                t = t / g;
                spots[(i * box + a) * box + c] = t;
            }
    return 0;
}

/* ------------------------------------------------------------------------
 * gaussmle  (picasso/gaussmle.py)
 * ---------------------------------------------------------------------- */
#define SQRT_2PI 2.5066282746310002 /* np.sqrt(2.0*np.pi) */
#define SQRT_2   1.4142135623730951
#define SQRT_PI  1.7724538509055159

/* gaussmle.py:28-48 */
static void sum_and_com(const float *spot, int size, double *sum, double *y, double *x)
{
    double sy = 0.0, sx = 0.0, s = 0.0;
    for (int i = 0; i < size; i++)
        for (int j = 0; j < size; j++) {
            double v = (double)spot[i * size + j];
            sy += v * (double)i;
            sx += v * (double)j;
            s += v;
        }
    if (s <= 0.0) { *sum = 0.01; *y = (size - 1) / 2.0; *x = (size - 1) / 2.0; return; }
    *sum = s; *y = sy / s; *x = sx / s;
}

/* gaussmle.py:61-91; returns the minimum of the filtered spot (float32) */
static float mean_filter_min(const float *spot, int size)
{
    float best = 0.0f;
    int first = 1;
    for (int k = 0; k < size; k++)
        for (int l = 0; l < size; l++) {
            int min_m = k - 1 > 0 ? k - 1 : 0, max_m = k + 2 < size ? k + 2 : size;
            int min_n = l - 1 > 0 ? l - 1 : 0, max_n = l + 2 < size ? l + 2 : size;
            int N = (max_m - min_m) * (max_n - min_n);
            double nsum = 0.0;
            for (int m = min_m; m < max_m; m++)
                for (int n = min_n; n < max_n; n++) nsum += (double)spot[m * size + n];
            float f = (float)(nsum / (double)N);          /* f32 store */
            /* np.min: NaN propagates */
            if (first) { best = f; first = 0; }
            else if (best == best && (f < best || f != f)) best = f;
        }
    return best;
}

/* gaussmle.py:94-139; theta6 = x, y, photons, bg, sx, sy as float32; sxy (optional) receives the
 * float64 sx, sy that _initial_theta_sigma averages BEFORE the float32 store (:150-153) */
static void initial_parameters_d(const float *spot, int size, float *theta6, double *sxy)
{
    double sum, y, x;
    sum_and_com(spot, size, &sum, &y, &x);
    float bg = mean_filter_min(spot, size);
    double photons = sum - (double)(size * size) * (double)bg;
    double photons_sane = np_max(1.0, photons);
    int size_half = size / 2;
    double sdy = 0.0, sdx = 0.0, sum_y = 0.0, sum_x = 0.0;
    for (int i = 0; i < size; i++) {
        double d2 = (double)((i - size_half) * (i - size_half));
        float vy = spot[i * size + size_half] - bg;       /* spot - bg is a float32 array */
        float vx = spot[size_half * size + i] - bg;
        sdy += (double)vy * d2;
        sdx += (double)vx * d2;
        sum_y += (double)vy;
        sum_x += (double)vx;
    }
    /* numba's default error model would raise ZeroDivisionError when
     * sum_y == 0; we follow IEEE (NaN/inf -> 0.01) like the NumPy execution
     * of the same source.  See DESIGN.md "degenerate spots". */
    double sy = sqrt(sdy / sum_y), sx = sqrt(sdx / sum_x);
    if (!isfinite(sy)) sy = 0.01;
    if (!isfinite(sx)) sx = 0.01;
    if (sx == 0) sx = 0.01;
    if (sy == 0) sy = 0.01;
    theta6[0] = (float)x; theta6[1] = (float)y; theta6[2] = (float)photons_sane;
    theta6[3] = bg; theta6[4] = (float)sx; theta6[5] = (float)sy;
    if (sxy) { sxy[0] = sx; sxy[1] = sy; }
}
static void initial_parameters(const float *spot, int size, float *theta6) { initial_parameters_d(spot, size, theta6, 0); }

/* gaussmle.py:268-280 */
static inline double gaussian_integral(int x, float mu, float sigma)
{
    double sq_norm = 0.70710678118654757 / (double)sigma;
    double d = (double)x - (double)mu;
    return 0.5 * (erf((d + 0.5) * sq_norm) - erf((d - 0.5) * sq_norm));
}

/* gaussmle.py:283-303 */
static inline void d_gaussian_integral(int x, float mu, float sigma, float photons, double PSFy,
                                       double *dudt, double *d2udt2)
{
    double d = (double)x - (double)mu;
    double ta = (d + 0.5) / (double)sigma, tb = (d - 0.5) / (double)sigma;
    double a = exp(-0.5 * (ta * ta));
    double b = exp(-0.5 * (tb * tb));
    float s3 = sigma * (sigma * sigma);                   /* float32 ** 3 stays float32 */
    *dudt = (double)photons * PSFy * (b - a) / (SQRT_2PI * (double)sigma);
    *d2udt2 = (double)photons * ((d - 0.5) * b - (d + 0.5) * a) * PSFy / (SQRT_2PI * (double)s3);
}

static inline double ipow_d(double a, int m) /* numba int power: square-and-multiply */
{
    double r = 1.0;
    while (m) { if (m & 1) r *= a; m >>= 1; a *= a; }
    return r;
}
static inline float ipow_f(float a, int m)
{
    float r = 1.0f;
    while (m) { if (m & 1) r *= a; m >>= 1; a *= a; }
    return r;
}

/* gaussmle.py:306-316 */
static inline double G(int n, int m, int x, float mu, float sigma_x)
{
    double a_minus = (double)x - (double)mu - 0.5;
    double a_plus = (double)x - (double)mu + 0.5;
    float s2 = sigma_x * sigma_x;
    double exp_minus = exp(-(a_minus * a_minus) / (2.0 * (double)s2));
    double exp_plus = exp(-(a_plus * a_plus) / (2.0 * (double)s2));
    return (ipow_d(a_minus, m) * exp_minus - ipow_d(a_plus, m) * exp_plus)
           / ((double)ipow_f(sigma_x, n) * SQRT_2PI);
}

/* gaussmle.py:319-336 */
static inline void d_gaussian_integral_sigma(int x, float mu, float sigma_x, float photons, double PSFy,
                                             double *dudt, double *d2udt2)
{
    *dudt = (double)photons * PSFy * G(2, 1, x, mu, sigma_x);
    *d2udt2 = (double)photons * PSFy * (G(5, 3, x, mu, sigma_x) - 2.0 * G(3, 1, x, mu, sigma_x));
}

/* gaussmle.py:339-383 (including the precedence quirk at :380-382) */
static inline void d_gaussian_integral_iso_sigma(int x, int y, float mu, float nu, float sigma, float photons,
                                                 double PSFx, double PSFy, double *dudt, double *d2udt2)
{
    double s = (double)sigma;
    double a_plus = ((double)x - (double)mu + 0.5) / (SQRT_2 * s);
    double a_minus = ((double)x - (double)mu - 0.5) / (SQRT_2 * s);
    double b_plus = ((double)y - (double)nu + 0.5) / (SQRT_2 * s);
    double b_minus = ((double)y - (double)nu - 0.5) / (SQRT_2 * s);
    double Fx = a_minus * exp(-(a_minus * a_minus)) - a_plus * exp(-(a_plus * a_plus));
    double Fy = b_minus * exp(-(b_minus * b_minus)) - b_plus * exp(-(b_plus * b_plus));
    double dPSFxdt = Fx / (SQRT_PI * s);
    double dPSFydt = Fy / (SQRT_PI * s);
    double dFxdt = (a_plus * exp(-(a_plus * a_plus)) * (1 - 2 * (a_plus * a_plus))
                    - a_minus * exp(-(a_minus * a_minus)) * (1 - 2 * (a_minus * a_minus))) / s;
    double dFydy = (b_plus * exp(-(b_plus * b_plus)) * (1 - 2 * (b_plus * b_plus))
                    - b_minus * exp(-(b_minus * b_minus)) * (1 - 2 * (b_minus * b_minus))) / s;
    float s2 = sigma * sigma;          /* float32 ** 2 */
    float sinv = 1.0f / sigma;         /* float32 ** (-1) */
    double d2PSFxdt2 = (1 / SQRT_PI) * ((-Fx / (double)s2) + (double)sinv * dFxdt);
    double d2PSFydt2 = (1 / SQRT_PI) * ((-Fy / (double)s2) + (double)sinv * dFydy);
    *dudt = (double)photons * (PSFy * dPSFxdt + PSFx * dPSFydt);
    *d2udt2 = (double)photons * PSFy * d2PSFxdt2 + 2 * dPSFxdt * dPSFydt + PSFx * d2PSFydt2;
}

/* symmetric pinv diagonal via cyclic Jacobi (np.linalg.pinv, rcond 1e-15:
 * singular values <= 1e-15 * max are dropped).  M is n x n, row-major. */
static void pinv_diag_sym(const double *Min, int n, double *diag)
{
    double A[36], V[36];
    int nonfinite = 0;
    for (int i = 0; i < n * n; i++) { A[i] = Min[i]; if (!isfinite(A[i])) nonfinite = 1; }
    if (nonfinite) { for (int i = 0; i < n; i++) diag[i] = NAN; return; }
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) V[i * n + j] = (i == j);
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int p = 0; p < n; p++) for (int q = p + 1; q < n; q++) off += A[p * n + q] * A[p * n + q];
        if (off == 0.0) break;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = A[p * n + q];
                if (apq == 0.0) continue;
                double app = A[p * n + p], aqq = A[q * n + q];
                double tau = (aqq - app) / (2.0 * apq);
                double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
                for (int k = 0; k < n; k++) {
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    double smax = 0.0;
    for (int i = 0; i < n; i++) if (fabs(A[i * n + i]) > smax) smax = fabs(A[i * n + i]);
    double cutoff = 1e-15 * smax;
    for (int i = 0; i < n; i++) {
        double acc = 0.0;
        for (int k = 0; k < n; k++) {
            double lam = A[k * n + k];
            if (fabs(lam) > cutoff) acc += V[i * n + k] * V[i * n + k] / lam;
        }
        diag[i] = acc;
    }
}

/* One spot, both methods.  thetas/crlbs rows of 6, as gaussmle.py:455-459
 * allocates them (the caller pre-fills crlbs with +inf, thetas with 0). */
static void mlefit_one(const float *spot, int size, int method, double eps, int max_it,
                       float *theta_out, float *crlb_out, float *ll_out, int32_t *it_out, float *closeness)
{
    /* closeness (optional, not part of the reference): min over the iterations of |D / eps - 1|, D = the largest
     * of the steps the convergence test looks at — how close the fit came to deciding the other way */
    float close = INFINITY;
    const int np_ = method == ORC_SIGMAXY ? 6 : 5;
    float theta[6], init[6];
    double sxy[2];
    initial_parameters_d(spot, size, init, sxy);
    theta[0] = init[0]; theta[1] = init[1]; theta[2] = init[2]; theta[3] = init[3];
    if (method == ORC_SIGMAXY) { theta[4] = init[4]; theta[5] = init[5]; }
    else { theta[4] = (float)((sxy[0] + sxy[1]) / 2); theta[5] = 0.0f; }   /* float64 mean, then the float32 store */

    float max_step[6];
    max_step[0] = theta[4]; max_step[1] = theta[4];
    max_step[2] = (float)(0.1 * (double)theta[2]);
    max_step[3] = (float)(0.1 * (double)theta[3]);
    max_step[4] = (float)(0.2 * (double)theta[4]);
    max_step[5] = (float)(0.2 * (double)theta[5]);

    float dudt[6], d2udt2[6], num[6], den[6];
    float old_x = theta[0], old_y = theta[1], old_sx = theta[4], old_sy = theta[5];
    int kk = 0;
    while (kk < max_it) {
        kk++;
        for (int l = 0; l < 6; l++) { num[l] = 0.0f; den[l] = 0.0f; }
        for (int ii = 0; ii < size; ii++)
            for (int jj = 0; jj < size; jj++) {
                float sgy = method == ORC_SIGMAXY ? theta[5] : theta[4];
                double PSFx = gaussian_integral(ii, theta[0], theta[4]);
                double PSFy = gaussian_integral(jj, theta[1], sgy);
                double a, b;
                d_gaussian_integral(ii, theta[0], theta[4], theta[2], PSFy, &a, &b);
                dudt[0] = (float)a; d2udt2[0] = (float)b;             /* f32 store */
                d_gaussian_integral(jj, theta[1], sgy, theta[2], PSFx, &a, &b);
                dudt[1] = (float)a; d2udt2[1] = (float)b;
                dudt[2] = (float)(PSFx * PSFy); d2udt2[2] = 0.0f;
                dudt[3] = 1.0f; d2udt2[3] = 0.0f;
                if (method == ORC_SIGMAXY) {
                    d_gaussian_integral_sigma(ii, theta[0], theta[4], theta[2], PSFy, &a, &b);
                    dudt[4] = (float)a; d2udt2[4] = (float)b;
                    d_gaussian_integral_sigma(jj, theta[1], theta[5], theta[2], PSFx, &a, &b);
                    dudt[5] = (float)a; d2udt2[5] = (float)b;
                } else {
                    d_gaussian_integral_iso_sigma(ii, jj, theta[0], theta[1], theta[4], theta[2],
                                                  PSFx, PSFy, &a, &b);
                    dudt[4] = (float)a; d2udt2[4] = (float)b;
                }
                double model = (double)theta[2] * PSFx * PSFy + (double)theta[3];
                double cf = 0.0, df = 0.0;
                float data = spot[jj * size + ii];
                if (model > 10e-3) {
                    cf = (double)data / model - 1;
                    df = (double)data / (model * model);
                }
                cf = np_min(cf, 10e4);
                df = np_min(df, 10e4);
                for (int l = 0; l < np_; l++) {
                    float du2 = dudt[l] * dudt[l];                     /* float32 ** 2 */
                    num[l] = (float)((double)num[l] + cf * (double)dudt[l]);
                    den[l] = (float)((double)den[l] + (cf * (double)d2udt2[l] - df * (double)du2));
                }
            }
        if (method == ORC_SIGMAXY) {                                   /* gaussmle.py:860-884 */
            for (int l = 0; l < 6; l++) {
                if (den[l] == 0.0f) theta[l] = theta[l] - np_signf(num[l]) * max_step[l];
                else theta[l] = theta[l] - np_minf(np_maxf(num[l] / den[l], -max_step[l]), max_step[l]);
            }
            theta[2] = (float)np_max((double)theta[2], 1.0);
            theta[3] = (float)np_max((double)theta[3], 0.01);
            theta[4] = (float)np_max((double)theta[4], 0.01);
            theta[5] = (float)np_max((double)theta[5], 0.01);
            int conv = ((double)fabsf(old_x - theta[0]) < eps) && ((double)fabsf(old_y - theta[1]) < eps)
                       && ((double)fabsf(old_sx - theta[4]) < eps) && ((double)fabsf(old_sy - theta[5]) < eps);
            {
                float D = fmaxf(fmaxf(fabsf(old_x - theta[0]), fabsf(old_y - theta[1])),
                                fmaxf(fabsf(old_sx - theta[4]), fabsf(old_sy - theta[5])));
                float c = (float)fabs((double)D / eps - 1.0);
                if (c < close) close = c;
            }
            if (conv) break;
            old_x = theta[0]; old_y = theta[1]; old_sx = theta[4]; old_sy = theta[5];
        } else {                                                       /* gaussmle.py:647-670 */
            for (int l = 0; l < 5; l++) {
                float update;
                if (den[l] == 0.0f) update = np_signf(num[l] * max_step[l]);   /* +-1, not +-max_step */
                else update = np_minf(np_maxf(num[l] / den[l], -max_step[l]), max_step[l]);
                theta[l] = theta[l] - update;
            }
            theta[2] = (float)np_max((double)theta[2], 1.0);
            theta[3] = (float)np_max((double)theta[3], 0.01);
            theta[4] = (float)np_max((double)theta[4], 0.01);
            theta[4] = (float)np_min((double)theta[4], (double)size);
            int conv = ((double)fabsf(old_x - theta[0]) < eps) && ((double)fabsf(old_y - theta[1]) < eps);
            {
                float D = fmaxf(fabsf(old_x - theta[0]), fabsf(old_y - theta[1]));
                float c = (float)fabs((double)D / eps - 1.0);
                if (c < close) close = c;
            }
            if (conv) break;
            old_x = theta[0]; old_y = theta[1];
        }
    }
    for (int l = 0; l < 5; l++) theta_out[l] = theta[l];
    theta_out[5] = method == ORC_SIGMAXY ? theta[5] : theta[4];
    *it_out = kk;
    if (closeness) *closeness = close;

    /* CRLB and log-likelihood (gaussmle.py:673-742, 887-954) */
    double M[36];
    for (int i = 0; i < 36; i++) M[i] = 0.0;
    double ll = 0.0;
    for (int l = 0; l < 6; l++) dudt[l] = 0.0f;
    for (int ii = 0; ii < size; ii++)
        for (int jj = 0; jj < size; jj++) {
            float sgy = method == ORC_SIGMAXY ? theta[5] : theta[4];
            double PSFx = gaussian_integral(ii, theta[0], theta[4]);
            double PSFy = gaussian_integral(jj, theta[1], sgy);
            double a, b;
            d_gaussian_integral(ii, theta[0], theta[4], theta[2], PSFy, &a, &b); dudt[0] = (float)a;
            d_gaussian_integral(jj, theta[1], sgy, theta[2], PSFx, &a, &b); dudt[1] = (float)a;
            if (method == ORC_SIGMAXY) {
                d_gaussian_integral_sigma(ii, theta[0], theta[4], theta[2], PSFy, &a, &b); dudt[4] = (float)a;
                d_gaussian_integral_sigma(jj, theta[1], theta[5], theta[2], PSFx, &a, &b); dudt[5] = (float)a;
            } else {
                d_gaussian_integral_iso_sigma(ii, jj, theta[0], theta[1], theta[4], theta[2], PSFx, PSFy, &a, &b);
                dudt[4] = (float)a;
            }
            dudt[2] = (float)(PSFx * PSFy);
            dudt[3] = 1.0f;
            double model = (double)theta[2] * PSFx * PSFy + (double)theta[3];
            for (int k = 0; k < np_; k++)
                for (int l = k; l < np_; l++) {
                    float prod = dudt[l] * dudt[k];                    /* float32 * float32 */
                    M[k * np_ + l] += (double)prod / model;
                    M[l * np_ + k] = M[k * np_ + l];
                }
            if (model > 0) {
                float data = spot[jj * size + ii];
                if (data > 0) {
                    float dlogd = data * logf(data);                   /* np.log(float32) is float32 */
                    ll += (double)data * log(model) - model - (double)dlogd + (double)data;
                } else {
                    ll += -model;
                }
            }
        }
    *ll_out = (float)ll;
    double diag[6];
    pinv_diag_sym(M, np_, diag);
    for (int k = 0; k < np_; k++) crlb_out[k] = (float)diag[k];
    if (method == ORC_SIGMA) crlb_out[5] = crlb_out[4];
}

/* gaussmle.py:409-475.  method: 0 "sigma", 1 "sigmaxy". */
int orc_gaussmle(const float *spots, int64_t N, int box, double eps, int max_it, int method,
                 float *thetas, float *crlbs, float *loglik, int32_t *iterations, int nthreads)
{
    if (box < 1 || box > ORC_MAX_BOX) return -1;
    if (method != ORC_SIGMA && method != ORC_SIGMAXY) return -2;   /* "Method not available." */
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads)
#endif
    for (int64_t i = 0; i < N; i++) {
        for (int l = 0; l < 6; l++) { thetas[i * 6 + l] = 0.0f; crlbs[i * 6 + l] = INFINITY; }
        mlefit_one(spots + i * box * box, box, method, eps, max_it,
                   thetas + i * 6, crlbs + i * 6, loglik + i, iterations + i, 0);
    }
    return 0;
}

/* The same fit, additionally reporting per spot how close its convergence test came to the other outcome
 * (see mlefit_one).  Used by the tests to bound the margin inside which the float32 device loop may disagree. */
int orc_gaussmle_closeness(const float *spots, int64_t N, int box, double eps, int max_it, int method,
                           float *thetas, float *crlbs, float *loglik, int32_t *iterations, float *closeness,
                           int nthreads)
{
    if (box < 1 || box > ORC_MAX_BOX) return -1;
    if (method != ORC_SIGMA && method != ORC_SIGMAXY) return -2;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads)
#endif
    for (int64_t i = 0; i < N; i++) {
        for (int l = 0; l < 6; l++) { thetas[i * 6 + l] = 0.0f; crlbs[i * 6 + l] = INFINITY; }
        mlefit_one(spots + i * box * box, box, method, eps, max_it,
                   thetas + i * 6, crlbs + i * 6, loglik + i, iterations + i, closeness + i);
    }
    return 0;
}

/* initial parameters only (for unit pinning of M1) */
int orc_initial_parameters(const float *spots, int64_t N, int box, float *theta6)
{
    for (int64_t i = 0; i < N; i++) initial_parameters(spots + i * box * box, box, theta6 + i * 6);
    return 0;
}

/* ------------------------------------------------------------------------
 * avgroi  (picasso/avgroi.py:24-41): theta = [0, 0, sum, sum, 1, 1], float64 sum
 * ---------------------------------------------------------------------- */
int orc_avgroi(const float *spots, int64_t N, int box, float *theta6)
{
    for (int64_t i = 0; i < N; i++) {
        double s = 0.0;
        for (int k = 0; k < box * box; k++) s += (double)spots[i * box * box + k];
        float *t = theta6 + i * 6;
        t[0] = 0.f; t[1] = 0.f; t[2] = (float)s; t[3] = (float)s; t[4] = 1.f; t[5] = 1.f;
    }
    return 0;
}

/* ------------------------------------------------------------------------
 * zfit  (picasso/zfit.py:254-291 _fit_z_target, :327-382 _fit_z)
 * The minimiser is third-party: scipy.optimize.minimize_scalar(bounds=...),
 * i.e. scipy/optimize/_optimize.py:_minimize_scalar_bounded (scipy 1.15.3,
 * xatol 1e-5, maxiter 500) — Brent's golden-section / parabolic fminbound,
 * restated here from its published algorithm; call site picasso/zfit.py:359-363.
 * ---------------------------------------------------------------------- */
static inline double zfit_target(double z, float sx, float sy, const double *cx, const double *cy)
{
    double z2 = z * z, z3 = z * z2, z4 = z * z3, z5 = z * z4, z6 = z * z5;
    double wx = cx[0] * z6 + cx[1] * z5 + cx[2] * z4 + cx[3] * z3 + cx[4] * z2 + cx[5] * z + cx[6];
    double wy = cy[0] * z6 + cy[1] * z5 + cy[2] * z4 + cy[3] * z3 + cy[4] * z2 + cy[5] * z + cy[6];
    double ax = pow((double)sx, 0.5) - pow(wx, 0.5), ay = pow((double)sy, 0.5) - pow(wy, 0.5);
    return ax * ax + ay * ay;
}

static inline double dsign(double v) { return (v > 0) - (v < 0); }

static void fminbound(float sx, float sy, const double *cx, const double *cy, double x1, double x2,
                      double xatol, int maxfun, double *xout, double *fout)
{
    const double sqrt_eps = sqrt(2.2e-16), golden_mean = 0.5 * (3.0 - sqrt(5.0));
    double a = x1, b = x2;
    double fulc = a + golden_mean * (b - a), nfc = fulc, xf = fulc;
    double rat = 0.0, e = 0.0;
    double x = xf, fx = zfit_target(x, sx, sy, cx, cy);
    int num = 1;
    double fu = INFINITY, ffulc = fx, fnfc = fx;
    double xm = 0.5 * (a + b);
    double tol1 = sqrt_eps * fabs(xf) + xatol / 3.0, tol2 = 2.0 * tol1;
    while (fabs(xf - xm) > (tol2 - 0.5 * (b - a))) {
        int golden = 1;
        if (fabs(e) > tol1) {
            golden = 0;
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
                    double si = dsign(xm - xf) + ((xm - xf) == 0);
                    rat = tol1 * si;
                }
            } else {
                golden = 1;
            }
        }
        if (golden) {
            e = (xf >= xm) ? a - xf : b - xf;
            rat = golden_mean * e;
        }
        double si = dsign(rat) + (rat == 0);
        double ar = fabs(rat);
        x = xf + si * ((ar != ar) ? ar : ((tol1 != tol1) ? tol1 : (ar > tol1 ? ar : tol1)));   /* np.maximum */
        fu = zfit_target(x, sx, sy, cx, cy);
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
    *xout = xf;
    *fout = fx;
}

/* z (before the magnification factor) and the squared calibration residual, float64 */
int orc_zfit(const float *sx, const float *sy, int64_t N, const double *cx7, const double *cy7,
             double *z, double *sqd, int nthreads)
{
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
    for (int64_t i = 0; i < N; i++)
        fminbound(sx[i], sy[i], cx7, cy7, -1000.0, 1000.0, 1e-5, 500, z + i, sqd + i);
    return 0;
}

/* ------------------------------------------------------------------------
 * gausslq  (picasso/gausslq.py:33-112 model + initial parameters, :206-244 fit_spot)
 *
 * The solver is third-party: scipy.optimize.leastsq(ftol=1e-2, xtol=1e-2) ->
 * MINPACK lmdif (scipy 1.15.3; call site picasso/gausslq.py:240-242).  lmdif,
 * lmpar, qrfac, qrsolv, fdjac2 and enorm are restated below from the published
 * MINPACK algorithm (More', Garbow, Hillstrom, ANL-80-74); tests validate this
 * restatement against scipy's own lmdif on the same residual function.
 * leastsq defaults that matter: gtol = 0, maxfev = 200*(n+1), factor = 100,
 * mode 1 (internal scaling), epsfcn = eps of the RESIDUAL dtype (float32).
 * ---------------------------------------------------------------------- */
#define LQ_N 6
#define LQ_MAXM ORC_MAX_PIX
static const double LQ_EPSMCH = 2.220446049250313e-16;
static const double LQ_DWARF = 2.2250738585072014e-308;

/* PROBE (diagnostics only, default 0 = MINPACK's order): 1 adds the long sums of lmdif — column norms, Householder
 * products, Q^T fvec — in REVERSE row order.  tools/probe_lq_order.py uses it to find the spots whose fit depends on
 * the order of those sums, i.e. the spots the device must fit with MINPACK's own order. */
static int g_lq_sum_reverse = 0, g_lq_trace = 0, g_lq_gs = 0;
/* reverse = 1: reversed row order; reverse = 2: the DEVICE's order (csrc/gausslq.hip): row r lives in lane r % gs, element
 * r / gs; a lane adds its elements in ascending order, the lanes are added by a butterfly (xor 1, 2, mirror of 8, mirror
 * of 16, xor 16, xor 32) */
void orc_lq_set_sum_order(int reverse) { g_lq_sum_reverse = reverse & 3; g_lq_gs = reverse >> 2; }
static double lq_device_sum(const double *t, int lo, int n)      /* sum of t[lo..n) in the device's order; t indexed by row */
{
    const int gs = g_lq_gs;
    double lane[64];
    for (int l = 0; l < gs; l++) {
        double a = 0;
        for (int r = l; r < n; r += gs) if (r >= lo) a += t[r];
        lane[l] = a;
    }
    double v[64], w[64];
    for (int l = 0; l < gs; l++) v[l] = lane[l];
    for (int l = 0; l < gs; l++) w[l] = v[l] + v[l ^ 1];
    for (int l = 0; l < gs; l++) v[l] = w[l] + w[l ^ 2];
    for (int l = 0; l < gs; l++) w[l] = v[l] + v[(l & ~7) | (7 - (l & 7))];
    if (gs == 8) return w[0];
    for (int l = 0; l < gs; l++) v[l] = w[l] + w[(l & ~15) | (15 - (l & 15))];
    if (gs == 16) return v[0];
    for (int l = 0; l < gs; l++) w[l] = v[l] + v[l ^ 16];
    if (gs == 32) return w[0];
    for (int l = 0; l < gs; l++) v[l] = w[l] + w[l ^ 32];
    return v[0];
}
void orc_lq_set_trace(int on) { g_lq_trace = on; }

static __thread int g_lq_enorm_rows = 0;      /* probe: total rows m of the Jacobian (0 outside lmdif) */
static double enorm(int n, const double *x)
{
    const double rdwarf = 3.834e-20, rgiant = 1.304e19;
    double s1 = 0, s2 = 0, s3 = 0, x1max = 0, x3max = 0;
    const double agiant = rgiant / (double)n;
    if (g_lq_sum_reverse == 2 && n > 6 && g_lq_enorm_rows > 0) {
        /* the rows of the vector are rows [rows - n, rows) of the residual numbering */
        double t[LQ_MAXM];
        const int rows = g_lq_enorm_rows, lo = rows - n;
        for (int r = 0; r < rows; r++) t[r] = 0;
        for (int i = 0; i < n; i++) t[lo + i] = x[i] * x[i];
        return sqrt(lq_device_sum(t, lo, rows));
    }
    for (int ii = 0; ii < n; ii++) {
        const int i = (g_lq_sum_reverse == 1 && n > 6) ? n - 1 - ii : ii;
        double xabs = fabs(x[i]);
        if (xabs > rdwarf && xabs < agiant) { s2 += xabs * xabs; }
        else if (xabs <= rdwarf) {
            if (xabs > x3max) { double r = x3max / xabs; s3 = 1 + s3 * r * r; x3max = xabs; }
            else if (xabs != 0) { double r = xabs / x3max; s3 += r * r; }
        } else {
            if (xabs > x1max) { double r = x1max / xabs; s1 = 1 + s1 * r * r; x1max = xabs; }
            else { double r = xabs / x1max; s1 += r * r; }
        }
    }
    if (s1 != 0) return x1max * sqrt(s1 + (s2 / x1max) / x1max);
    if (s2 != 0) {
        if (s2 >= x3max) return sqrt(s2 * (1 + (x3max / s2) * (x3max * s3)));
        return sqrt(x3max * ((s2 / x3max) + (x3max * s3)));
    }
    return x3max * sqrt(s3);
}

/* PROBE (diagnostics only, default 0 = libm's exp, what the reference's np.exp resolves to here): 1 evaluates the two
 * Gaussian profiles with exp correctly rounded to float64 (expl, 64-bit significand, then one rounding).  tools/probe_lq_exp.py
 * uses it on the spots where the device's strict mode differs from this oracle: libm's exp is within an ulp but not always
 * the correctly rounded value, the device's (ocml) is another such function, and where the two differ in the last bit of ONE
 * profile value a float32 rounding of the stored model (gausslq.py:203) can flip. */
static int g_lq_exp_variant = 0;
void orc_lq_set_exp(int which) { g_lq_exp_variant = which; }
static double lq_exp(double x) { return g_lq_exp_variant ? (double)expl((long double)x) : exp(x); }

/* residuals of the point-sampled Gaussian model, float32 stores as in gausslq.py:151-203 */
static void lq_residuals(const double *theta, const float *spot, int size, double *fvec)
{
    float mx[ORC_MAX_BOX], my[ORC_MAX_BOX];
    const int h = size / 2;
    const double nx = 0.3989422804014327 / theta[4], ny = 0.3989422804014327 / theta[5];
    for (int i = 0; i < size; i++) {
        double g = (double)(float)(i - h);                    /* grid is a float32 array */
        double tx = (g - theta[0]) / theta[4], ty = (g - theta[1]) / theta[5];
        mx[i] = (float)(nx * lq_exp(-0.5 * (tx * tx)));       /* f32 store */
        my[i] = (float)(ny * lq_exp(-0.5 * (ty * ty)));
    }
    for (int i = 0; i < size; i++)
        for (int j = 0; j < size; j++) {
            float model = (float)(theta[2] * (double)my[i] * (double)mx[j] + theta[3]);   /* f32 store */
            float res = spot[i * size + j] - model;           /* float32 array arithmetic */
            fvec[i * size + j] = (double)res;
        }
}

/* gausslq.py:95-112 */
static void lq_initial_parameters(const float *spot, int size, float *theta)
{
    const int h = size / 2;
    float mn = spot[0];
    for (int k = 1; k < size * size; k++) { float v = spot[k]; if (mn == mn && (v < mn || v != v)) mn = v; }
    theta[3] = mn;
    double y = 0.0, x = 0.0, sum = 0.0;
    for (int i = 0; i < size; i++)
        for (int j = 0; j < size; j++) {
            double v = (double)(float)(spot[i * size + j] - mn);
            y += v * (double)i; x += v * (double)j; sum += v;
        }
    if (sum <= 0.0) { sum = 0.01; y = (size - 1) / 2.0; x = (size - 1) / 2.0; }
    else { y /= sum; x /= sum; }
    theta[1] = (float)y; theta[0] = (float)x;
    theta[2] = (float)(1.0 > sum ? 1.0 : sum);
    double sdy = 0.0, sdx = 0.0;
    for (int i = 0; i < size; i++)
        for (int j = 0; j < size; j++) {
            double v = (double)(float)(spot[i * size + j] - mn);
            double dy = (double)i - (double)theta[1], dx = (double)j - (double)theta[0];
            sdy += v * (dy * dy);
            sdx += v * (dx * dx);
        }
    theta[5] = (float)sqrt(sdy / sum);
    theta[4] = (float)sqrt(sdx / sum);
    theta[0] = theta[0] - (float)h;
    theta[1] = theta[1] - (float)h;
}

static void qrsolv(int n, double *r, int ldr, const int *ipvt, const double *diag, const double *qtb,
                   double *x, double *sdiag, double *wa)
{
    for (int j = 0; j < n; j++) {
        for (int i = j; i < n; i++) r[i + j * ldr] = r[j + i * ldr];
        x[j] = r[j + j * ldr];
        wa[j] = qtb[j];
    }
    for (int j = 0; j < n; j++) {
        int l = ipvt[j];
        if (diag[l] != 0) {
            for (int k = j; k < n; k++) sdiag[k] = 0;
            sdiag[j] = diag[l];
            double qtbpj = 0;
            for (int k = j; k < n; k++) {
                if (sdiag[k] == 0) continue;
                double c, sn;
                if (fabs(r[k + k * ldr]) < fabs(sdiag[k])) {
                    double cotan = r[k + k * ldr] / sdiag[k];
                    sn = 0.5 / sqrt(0.25 + 0.25 * (cotan * cotan));
                    c = sn * cotan;
                } else {
                    double tn = sdiag[k] / r[k + k * ldr];
                    c = 0.5 / sqrt(0.25 + 0.25 * (tn * tn));
                    sn = c * tn;
                }
                r[k + k * ldr] = c * r[k + k * ldr] + sn * sdiag[k];
                double temp = c * wa[k] + sn * qtbpj;
                qtbpj = -sn * wa[k] + c * qtbpj;
                wa[k] = temp;
                for (int i = k + 1; i < n; i++) {
                    temp = c * r[i + k * ldr] + sn * sdiag[i];
                    sdiag[i] = -sn * r[i + k * ldr] + c * sdiag[i];
                    r[i + k * ldr] = temp;
                }
            }
        }
        sdiag[j] = r[j + j * ldr];
        r[j + j * ldr] = x[j];
    }
    int nsing = n;
    for (int j = 0; j < n; j++) {
        if (sdiag[j] == 0 && nsing == n) nsing = j;
        if (nsing < n) wa[j] = 0;
    }
    for (int k = 1; k <= nsing; k++) {
        int j = nsing - k;
        double sum = 0;
        for (int i = j + 1; i < nsing; i++) sum += r[i + j * ldr] * wa[i];
        wa[j] = (wa[j] - sum) / sdiag[j];
    }
    for (int j = 0; j < n; j++) x[ipvt[j]] = wa[j];
}

static void lmpar(int n, double *r, int ldr, const int *ipvt, const double *diag, const double *qtb, double delta,
                  double *par, double *x, double *sdiag, double *wa1, double *wa2)
{
    int nsing = n;
    for (int j = 0; j < n; j++) {
        wa1[j] = qtb[j];
        if (r[j + j * ldr] == 0 && nsing == n) nsing = j;
        if (nsing < n) wa1[j] = 0;
    }
    for (int k = 1; k <= nsing; k++) {
        int j = nsing - k;
        wa1[j] /= r[j + j * ldr];
        double temp = wa1[j];
        for (int i = 0; i < j; i++) wa1[i] -= r[i + j * ldr] * temp;
    }
    for (int j = 0; j < n; j++) x[ipvt[j]] = wa1[j];
    int iter = 0;
    for (int j = 0; j < n; j++) wa2[j] = diag[j] * x[j];
    double dxnorm = enorm(n, wa2);
    double fp = dxnorm - delta;
    if (fp <= 0.1 * delta) { *par = 0; return; }
    double parl = 0;
    if (nsing >= n) {
        for (int j = 0; j < n; j++) { int l = ipvt[j]; wa1[j] = diag[l] * (wa2[l] / dxnorm); }
        for (int j = 0; j < n; j++) {
            double sum = 0;
            for (int i = 0; i < j; i++) sum += r[i + j * ldr] * wa1[i];
            wa1[j] = (wa1[j] - sum) / r[j + j * ldr];
        }
        double temp = enorm(n, wa1);
        parl = ((fp / delta) / temp) / temp;
    }
    for (int j = 0; j < n; j++) {
        double sum = 0;
        for (int i = 0; i <= j; i++) sum += r[i + j * ldr] * qtb[i];
        wa1[j] = sum / diag[ipvt[j]];
    }
    double gnorm = enorm(n, wa1);
    double paru = gnorm / delta;
    if (paru == 0) paru = LQ_DWARF / (delta < 0.1 ? delta : 0.1);
    if (*par < parl) *par = parl;
    if (*par > paru) *par = paru;
    if (*par == 0) *par = gnorm / dxnorm;
    for (;;) {
        iter++;
        if (*par == 0) { double t = 0.001 * paru; *par = LQ_DWARF > t ? LQ_DWARF : t; }
        double temp = sqrt(*par);
        for (int j = 0; j < n; j++) wa1[j] = temp * diag[j];
        qrsolv(n, r, ldr, ipvt, wa1, qtb, x, sdiag, wa2);
        for (int j = 0; j < n; j++) wa2[j] = diag[j] * x[j];
        dxnorm = enorm(n, wa2);
        temp = fp;
        fp = dxnorm - delta;
        if (fabs(fp) <= 0.1 * delta || (parl == 0 && fp <= temp && temp < 0) || iter == 10) break;
        for (int j = 0; j < n; j++) { int l = ipvt[j]; wa1[j] = diag[l] * (wa2[l] / dxnorm); }
        for (int j = 0; j < n; j++) {
            wa1[j] /= sdiag[j];
            temp = wa1[j];
            for (int i = j + 1; i < n; i++) wa1[i] -= r[i + j * ldr] * temp;
        }
        temp = enorm(n, wa1);
        double parc = ((fp / delta) / temp) / temp;
        if (fp > 0 && parl < *par) parl = *par;
        if (fp < 0 && paru > *par) paru = *par;
        double np_ = *par + parc;
        *par = parl > np_ ? parl : np_;
    }
    if (iter == 0) *par = 0;
}

static void qrfac(int m, int n, double *a, int lda, int *ipvt, double *rdiag, double *acnorm, double *wa)
{
    for (int j = 0; j < n; j++) {
        acnorm[j] = enorm(m, a + j * lda);
        rdiag[j] = acnorm[j];
        wa[j] = rdiag[j];
        ipvt[j] = j;
    }
    int minmn = m < n ? m : n;
    for (int j = 0; j < minmn; j++) {
        int kmax = j;
        for (int k = j; k < n; k++) if (rdiag[k] > rdiag[kmax]) kmax = k;
        if (kmax != j) {
            for (int i = 0; i < m; i++) { double t = a[i + j * lda]; a[i + j * lda] = a[i + kmax * lda]; a[i + kmax * lda] = t; }
            rdiag[kmax] = rdiag[j];
            wa[kmax] = wa[j];
            int k = ipvt[j]; ipvt[j] = ipvt[kmax]; ipvt[kmax] = k;
        }
        double ajnorm = enorm(m - j, a + j + j * lda);
        if (ajnorm != 0) {
            if (a[j + j * lda] < 0) ajnorm = -ajnorm;
            for (int i = j; i < m; i++) a[i + j * lda] /= ajnorm;
            a[j + j * lda] += 1;
            for (int k = j + 1; k < n; k++) {
                double sum = 0;
                if (g_lq_sum_reverse == 2) {
                    double t[LQ_MAXM];
                    for (int i = 0; i < m; i++) t[i] = i >= j ? a[i + j * lda] * a[i + k * lda] : 0;
                    sum = lq_device_sum(t, j, m);
                } else if (g_lq_sum_reverse) for (int i = m - 1; i >= j; i--) sum += a[i + j * lda] * a[i + k * lda];
                else for (int i = j; i < m; i++) sum += a[i + j * lda] * a[i + k * lda];
                double temp = sum / a[j + j * lda];
                for (int i = j; i < m; i++) a[i + k * lda] -= temp * a[i + j * lda];
                if (rdiag[k] != 0) {
                    temp = a[j + k * lda] / rdiag[k];
                    double t2 = 1 - temp * temp;
                    rdiag[k] *= sqrt(t2 > 0 ? t2 : 0);
                    double q = rdiag[k] / wa[k];
                    if (0.05 * (q * q) <= LQ_EPSMCH) {
                        rdiag[k] = enorm(m - j - 1, a + (j + 1) + k * lda);
                        wa[k] = rdiag[k];
                    }
                }
            }
        }
        rdiag[j] = -ajnorm;
    }
}

/* lmdif for one spot; returns info, theta (float64[6]) in x */
static int lmdif_spot(const float *spot, int size, double *x, double ftol, double xtol, double gtol, int maxfev,
                      double epsfcn, double factor, int *nfev_out)
{
    const int n = LQ_N, m = size * size, ld = m;
    static __thread double fjac[LQ_MAXM * LQ_N], fvec[LQ_MAXM], wa4[LQ_MAXM];
    double diag[LQ_N], qtf[LQ_N], wa1[LQ_N], wa2[LQ_N], wa3[LQ_N];
    int ipvt[LQ_N];
    int info = 0, nfev = 0, iter = 1;
    double par = 0, delta = 0, xnorm = 0, gnorm = 0, fnorm, fnorm1, actred, prered, dirder, ratio, pnorm;
    lq_residuals(x, spot, size, fvec); nfev = 1;
    fnorm = enorm(m, fvec);
    const double eps = sqrt(epsfcn > LQ_EPSMCH ? epsfcn : LQ_EPSMCH);
    for (;;) {
        /* forward-difference Jacobian (fdjac2) */
        for (int j = 0; j < n; j++) {
            double temp = x[j], h = eps * fabs(temp);
            if (h == 0) h = eps;
            x[j] = temp + h;
            lq_residuals(x, spot, size, wa4);
            x[j] = temp;
            for (int i = 0; i < m; i++) fjac[i + j * ld] = (wa4[i] - fvec[i]) / h;
        }
        nfev += n;
        g_lq_enorm_rows = m;       /* (probe) the norms of qrfac are sums over Jacobian rows; fnorm / fnorm1 stay sequential */
        qrfac(m, n, fjac, ld, ipvt, wa1, wa2, wa3);
        g_lq_enorm_rows = 0;
        if (g_lq_trace) {
            double kap = 1e300;
            for (int j = 0; j < n; j++) { double k = fabs(wa1[j]) / wa2[ipvt[j]]; if (k < kap) kap = k; }
            fprintf(stderr, "  qrfac: min |R_jj| / |column| = %.3e  rdiag %.3e %.3e %.3e %.3e %.3e %.3e\n", kap, wa1[0], wa1[1], wa1[2], wa1[3], wa1[4], wa1[5]);
        }
        if (iter == 1) {
            for (int j = 0; j < n; j++) { diag[j] = wa2[j]; if (wa2[j] == 0) diag[j] = 1; }
            for (int j = 0; j < n; j++) wa3[j] = diag[j] * x[j];
            xnorm = enorm(n, wa3);
            delta = factor * xnorm;
            if (delta == 0) delta = factor;
        }
        for (int i = 0; i < m; i++) wa4[i] = fvec[i];
        for (int j = 0; j < n; j++) {
            if (fjac[j + j * ld] != 0) {
                double sum = 0;
                if (g_lq_sum_reverse == 2) {
                    double t[LQ_MAXM];
                    for (int i = 0; i < m; i++) t[i] = i >= j ? fjac[i + j * ld] * wa4[i] : 0;
                    sum = lq_device_sum(t, j, m);
                } else if (g_lq_sum_reverse) for (int i = m - 1; i >= j; i--) sum += fjac[i + j * ld] * wa4[i];
                else for (int i = j; i < m; i++) sum += fjac[i + j * ld] * wa4[i];
                double temp = -sum / fjac[j + j * ld];
                for (int i = j; i < m; i++) wa4[i] += fjac[i + j * ld] * temp;
            }
            fjac[j + j * ld] = wa1[j];
            qtf[j] = wa4[j];
        }
        gnorm = 0;
        if (fnorm != 0) {
            for (int j = 0; j < n; j++) {
                int l = ipvt[j];
                if (wa2[l] != 0) {
                    double sum = 0;
                    for (int i = 0; i <= j; i++) sum += fjac[i + j * ld] * (qtf[i] / fnorm);
                    double g = fabs(sum / wa2[l]);
                    if (g > gnorm) gnorm = g;
                }
            }
        }
        if (gnorm <= gtol) { info = 4; break; }
        for (int j = 0; j < n; j++) if (wa2[j] > diag[j]) diag[j] = wa2[j];
        for (;;) {
            lmpar(n, fjac, ld, ipvt, diag, qtf, delta, &par, wa1, wa2, wa3, wa4);
            for (int j = 0; j < n; j++) { wa1[j] = -wa1[j]; wa2[j] = x[j] + wa1[j]; wa3[j] = diag[j] * wa1[j]; }
            pnorm = enorm(n, wa3);
            if (iter == 1 && pnorm < delta) delta = pnorm;
            lq_residuals(wa2, spot, size, wa4); nfev++;
            fnorm1 = enorm(m, wa4);
            actred = -1;
            if (0.1 * fnorm1 < fnorm) { double r = fnorm1 / fnorm; actred = 1 - r * r; }
            for (int j = 0; j < n; j++) {
                wa3[j] = 0;
                double temp = wa1[ipvt[j]];
                for (int i = 0; i <= j; i++) wa3[i] += fjac[i + j * ld] * temp;
            }
            double temp1 = enorm(n, wa3) / fnorm, temp2 = (sqrt(par) * pnorm) / fnorm;
            prered = temp1 * temp1 + temp2 * temp2 / 0.5;
            dirder = -(temp1 * temp1 + temp2 * temp2);
            ratio = 0;
            if (prered != 0) ratio = actred / prered;
            if (ratio <= 0.25) {
                double temp = 0.5;
                if (actred < 0) temp = 0.5 * dirder / (dirder + 0.5 * actred);
                if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
                double pd = pnorm / 0.1;
                delta = temp * (delta < pd ? delta : pd);
                par = par / temp;
            } else if (par == 0 || ratio >= 0.75) {
                delta = pnorm / 0.5;
                par = 0.5 * par;
            }
            if (g_lq_trace)
                fprintf(stderr, "iter %d nfev %d par %.17g delta %.17g pnorm %.17g fnorm %.17g fnorm1 %.17g actred %.17g prered %.17g ratio %.17g xnorm %.17g ipvt %d%d%d%d%d%d\n",
                        iter, nfev, par, delta, pnorm, fnorm, fnorm1, actred, prered, ratio, xnorm, ipvt[0], ipvt[1], ipvt[2], ipvt[3], ipvt[4], ipvt[5]);
            if (ratio >= 1e-4) {
                for (int j = 0; j < n; j++) { x[j] = wa2[j]; wa2[j] = diag[j] * x[j]; }
                for (int i = 0; i < m; i++) fvec[i] = wa4[i];
                xnorm = enorm(n, wa2);
                fnorm = fnorm1;
                iter++;
            }
            if (fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1) info = 1;
            if (delta <= xtol * xnorm) info = 2;
            if (fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1 && info == 2) info = 3;
            if (info != 0) break;
            if (nfev >= maxfev) info = 5;
            if (fabs(actred) <= LQ_EPSMCH && prered <= LQ_EPSMCH && 0.5 * ratio <= 1) info = 6;
            if (delta <= LQ_EPSMCH * xnorm) info = 7;
            if (gnorm <= LQ_EPSMCH) info = 8;
            if (info != 0) break;
            if (ratio >= 1e-4) break;
        }
        if (info != 0) break;
    }
    if (nfev_out) *nfev_out = nfev;
    return info;
}

/* gausslq.fit_spots: theta (N,6) float32 [x, y, photons, bg, sx, sy], x/y relative to the box centre */
int orc_gausslq(const float *spots, int64_t N, int box, float *theta, int32_t *info_out, int32_t *nfev_out, int nthreads)
{
    if (box < 1 || box > ORC_MAX_BOX) return -1;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads)
#endif
    for (int64_t i = 0; i < N; i++) {
        float t0[6];
        double x[6];
        lq_initial_parameters(spots + i * box * box, box, t0);
        for (int k = 0; k < 6; k++) x[k] = (double)t0[k];
        int nfev = 0;
        int info = lmdif_spot(spots + i * box * box, box, x, 1e-2, 1e-2, 0.0, 200 * (LQ_N + 1),
                              1.1920928955078125e-07 /* finfo(float32).eps */, 100.0, &nfev);
        for (int k = 0; k < 6; k++) theta[i * 6 + k] = (float)x[k];
        if (info_out) info_out[i] = info;
        if (nfev_out) nfev_out[i] = nfev;
    }
    return 0;
}

/* lmdif from caller-supplied start values (float32, as gausslq.py:238 builds them) */
int orc_gausslq_from(const float *spots, int64_t N, int box, const float *theta0, float *theta, int nthreads)
{
    if (box < 1 || box > ORC_MAX_BOX) return -1;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads)
#endif
    for (int64_t i = 0; i < N; i++) {
        double x[6];
        for (int k = 0; k < 6; k++) x[k] = (double)theta0[i * 6 + k];
        lmdif_spot(spots + i * box * box, box, x, 1e-2, 1e-2, 0.0, 200 * (LQ_N + 1), 1.1920928955078125e-07, 100.0, NULL);
        for (int k = 0; k < 6; k++) theta[i * 6 + k] = (float)x[k];
    }
    return 0;
}

int orc_gausslq_initial(const float *spots, int64_t N, int box, float *theta)
{
    for (int64_t i = 0; i < N; i++) lq_initial_parameters(spots + i * box * box, box, theta + i * 6);
    return 0;
}

/* ------------------------------------------------------------------------
 * render  (picasso/render.py:177-232 _render_setup, :451-467 _fill,
 *          :494-575 _draw_gaussian_loc / _fill_gaussian, :798-853, :1020-1070)
 *
 * numba promotion: the float32 coordinates become float64 inside _render_setup
 * (array(float32) - float64 scalar), so the image coordinates, the footprint
 * bounds and the two 1-D profiles are float64, rounded to float32 when stored
 * into gx / gy; the outer product and the accumulation into the image are
 * float32, in localization order.  The blur widths are computed outside numba:
 * float32(oversampling) * max(lp, float32(min_blur_width)) in float32.
 * ---------------------------------------------------------------------- */
/* float64 -> int32 as compiled x86-64 code does it (cvttsd2si): INT_MIN for NaN / out of range */
static inline int32_t to_int32(double v)
{
    return (v != v || v >= 2147483648.0 || v < -2147483648.0) ? INT32_MIN : (int32_t)v;
}

static void render_dims(double oversampling, double y_min, double x_min, double y_max, double x_max,
                        int64_t *ny, int64_t *nx)
{
    *ny = (int64_t)ceil(oversampling * (y_max - y_min));
    *nx = (int64_t)ceil(oversampling * (x_max - x_min));
}

/* image (ny, nx) float32, zeroed by the caller.  Returns the number of localizations in view. */
int64_t orc_render_hist(const float *x, const float *y, int64_t N, double oversampling, double y_min, double x_min,
                        double y_max, double x_max, float *image, int64_t ny, int64_t nx)
{
    int64_t n = 0;
    for (int64_t i = 0; i < N; i++) {
        const double xd = (double)x[i], yd = (double)y[i];
        if (!(xd > x_min && yd > y_min && xd < x_max && yd < y_max)) continue;
        const int32_t xi = to_int32(oversampling * (xd - x_min)), yi = to_int32(oversampling * (yd - y_min));
        image[(int64_t)yi * nx + xi] += 1.0f;
        n++;
    }
    (void)ny;
    return n;
}

int64_t orc_render_gaussian(const float *x, const float *y, const float *lpx, const float *lpy, int64_t N,
                            double oversampling, double y_min, double x_min, double y_max, double x_max,
                            double min_blur_width, int iso, float *image, int64_t ny, int64_t nx)
{
    const float osf = (float)oversampling, mbw = (float)min_blur_width;
    float gx[4096], gy[4096];
    int64_t n = 0;
    for (int64_t i = 0; i < N; i++) {
        const double xd = (double)x[i], yd = (double)y[i];
        if (!(xd > x_min && yd > y_min && xd < x_max && yd < y_max)) continue;
        n++;
        const double x_ = oversampling * (xd - x_min), y_ = oversampling * (yd - y_min);
        float sx_ = osf * np_maxf(lpx[i], mbw), sy_ = osf * np_maxf(lpy[i], mbw);
        if (iso) { sy_ = (sy_ + sx_) / 2.0f; sx_ = sy_; }          /* gaussian_iso, render.py:1196-1198 */
        const double max_y_off = 3.0 * (double)sy_, max_x_off = 3.0 * (double)sx_;
        int64_t i_min = to_int32(y_ - max_y_off);
        if (i_min < 0) i_min = 0;
        int64_t i_max = to_int32(y_ + max_y_off + 1);
        if (i_max > ny) i_max = ny;
        int64_t j_min = to_int32(x_ - max_x_off);
        if (j_min < 0) j_min = 0;
        int64_t j_max = (int64_t)to_int32(x_ + max_x_off) + 1;
        if (j_max > nx) j_max = nx;
        const int64_t cx = j_max - j_min, cy = i_max - i_min;
        if (cx <= 0 || cy <= 0) continue;
        if (cx > 4096 || cy > 4096) return -1;                      /* footprint larger than this restatement allows */
        const double inv_2sx2 = 1.0 / (2.0 * (double)sx_ * (double)sx_);
        const double inv_2sy2 = 1.0 / (2.0 * (double)sy_ * (double)sy_);
        const double norm = 1.0 / (6.283185307179586 * (double)sx_ * (double)sy_);
        for (int64_t jj = 0; jj < cx; jj++) {
            const double dx = (double)(j_min + jj) + 0.5 - x_;
            gx[jj] = (float)exp(-dx * dx * inv_2sx2);
        }
        for (int64_t ii = 0; ii < cy; ii++) {
            const double dy = (double)(i_min + ii) + 0.5 - y_;
            gy[ii] = (float)(norm * exp(-dy * dy * inv_2sy2));
        }
        for (int64_t ii = 0; ii < cy; ii++) {
            float *row = image + (i_min + ii) * nx;
            for (int64_t jj = 0; jj < cx; jj++) row[j_min + jj] += gy[ii] * gx[jj];
        }
    }
    return n;
}

void orc_render_dims(double oversampling, double y_min, double x_min, double y_max, double x_max, int64_t *ny, int64_t *nx)
{
    render_dims(oversampling, y_min, x_min, y_max, x_max, ny, nx);
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------
 * Sub-pixel peak of a cross-correlation window  (picasso/imageprocess.py:121-141)
 *
 * The reference fits a*exp(-0.5*((x-xc)^2+(y-yc)^2)/s^2)+b to the box x box window around the
 * correlation maximum with scipy.optimize.curve_fit(p0=[max,0,0,1,min], bounds=([0,-inf,-inf,0,0],
 * inf)).  With bounds curve_fit calls least_squares(method='trf', jac='2-point', x_scale=1,
 * ftol=xtol=gtol=1e-8, max_nfev=100*n, tr_solver='exact'): the arithmetic is THIRD-PARTY — scipy
 * 1.15.3 (pyproject.toml:36 pins >= 1.15.3), scipy/optimize/_lsq/trf.py trf_bounds + common.py,
 * scipy/optimize/_numdiff.py approx_derivative — restated here from that published algorithm
 * (Branch, Coleman & Li 1999; More 1977 for the trust-region sub-problem).  The one deviation: the
 * SVD of the augmented Jacobian is a one-sided Jacobi SVD instead of LAPACK gesdd; both are
 * accurate to a few ulps, so the iterates agree to ~1e-12 and stop in the same iteration.
 * Pinned against scipy itself in tests/test_oracle_golden.py.
 * ---------------------------------------------------------------------- */
#define PK_N 5
#define PK_MAXM (ORC_MAX_BOX * ORC_MAX_BOX)
#define PK_EPS 2.220446049250313e-16

static void pk_fun(const double *x, const double *data, int box, double *f)
{
    const int h = box / 2;
    for (int i = 0; i < box; i++)
        for (int j = 0; j < box; j++) {
            const double xx = (double)(j - h) - x[1], yy = (double)(i - h) - x[2];
            const double e = -0.5 * (xx * xx + yy * yy) / (x[3] * x[3]);
            f[i * box + j] = (x[0] * exp(e) + x[4]) - data[i * box + j];
        }
}

static double pk_norm(const double *v, int n) { double s = 0; for (int i = 0; i < n; i++) s += v[i] * v[i]; return sqrt(s); }
static double pk_dot(const double *a, const double *b, int n) { double s = 0; for (int i = 0; i < n; i++) s += a[i] * b[i]; return s; }

/* approx_derivative(method='2-point', rel_step=None, bounds): J is m x n row-major */
static void pk_jac(const double *x0, const double *f0, const double *data, int box, const double *lb, const double *ub,
                   double *J, int *nfev)
{
    const int m = box * box;
    double x1[PK_N], f1[PK_MAXM];
    const double rstep = sqrt(PK_EPS);
    for (int i = 0; i < PK_N; i++) {
        const double sign_x0 = x0[i] >= 0 ? 1.0 : -1.0;
        double h = rstep * sign_x0 * fmax(1.0, fabs(x0[i]));
        /* _adjust_scheme_to_bounds, '1-sided', num_steps = 1 */
        const double lower_dist = x0[i] - lb[i], upper_dist = ub[i] - x0[i];
        const double xt = x0[i] + h;
        const int violated = (xt < lb[i]) || (xt > ub[i]);
        const int fitting = fabs(h) <= fmax(lower_dist, upper_dist);
        if (violated && fitting) h = -h;
        if (!fitting) h = upper_dist >= lower_dist ? upper_dist : -lower_dist;
        for (int k = 0; k < PK_N; k++) x1[k] = x0[k];
        x1[i] += h;
        const double dx = x1[i] - x0[i];
        pk_fun(x1, data, box, f1);
        (*nfev)++;          /* (counted by scipy's jac wrapper separately from nfev; kept for information) */
        for (int r = 0; r < m; r++) J[r * PK_N + i] = (f1[r] - f0[r]) / dx;
    }
}

/* one-sided Jacobi SVD of A (M x 5, row-major, overwritten by U*S); V (5x5 row-major) */
static void pk_svd(double *A, int M, double *V, double *s)
{
    for (int i = 0; i < PK_N; i++) for (int j = 0; j < PK_N; j++) V[i * PK_N + j] = (i == j);
    for (int sweep = 0; sweep < 60; sweep++) {
        int rotated = 0;
        for (int p = 0; p < PK_N - 1; p++)
            for (int q = p + 1; q < PK_N; q++) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int r = 0; r < M; r++) {
                    const double ap = A[r * PK_N + p], aq = A[r * PK_N + q];
                    alpha += ap * ap; beta += aq * aq; gamma += ap * aq;
                }
                if (gamma == 0.0 || fabs(gamma) <= PK_EPS * sqrt(alpha * beta)) continue;
                rotated = 1;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int r = 0; r < M; r++) {
                    const double ap = A[r * PK_N + p], aq = A[r * PK_N + q];
                    A[r * PK_N + p] = c * ap - sn * aq;
                    A[r * PK_N + q] = sn * ap + c * aq;
                }
                for (int r = 0; r < PK_N; r++) {
                    const double vp = V[r * PK_N + p], vq = V[r * PK_N + q];
                    V[r * PK_N + p] = c * vp - sn * vq;
                    V[r * PK_N + q] = sn * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    for (int i = 0; i < PK_N; i++) {
        double n2 = 0;
        for (int r = 0; r < M; r++) n2 += A[r * PK_N + i] * A[r * PK_N + i];
        s[i] = sqrt(n2);
    }
}

/* common.py solve_lsq_trust_region: suf[i] = s[i] * (U^T f)[i] */
static void pk_solve_tr(int m, const double *suf, const double *s, const double *V, double Delta, double *alpha_io, double *p)
{
    double smax = 0, smin = INFINITY;
    for (int i = 0; i < PK_N; i++) { if (s[i] > smax) smax = s[i]; if (s[i] < smin) smin = s[i]; }
    const int full_rank = smin > PK_EPS * m * smax;
    double t[PK_N];
    if (full_rank) {
        for (int i = 0; i < PK_N; i++) t[i] = (suf[i] / s[i]) / s[i];        /* uf / s */
        for (int i = 0; i < PK_N; i++) { double a = 0; for (int k = 0; k < PK_N; k++) a += V[i * PK_N + k] * t[k]; p[i] = -a; }
        if (pk_norm(p, PK_N) <= Delta) { *alpha_io = 0.0; return; }
    }
    double alpha_upper = pk_norm(suf, PK_N) / Delta, alpha_lower = 0.0;
    if (full_rank) {
        double pn2 = 0, dsum = 0;
        for (int i = 0; i < PK_N; i++) { const double den = s[i] * s[i]; const double q = suf[i] / den; pn2 += q * q; dsum += suf[i] * suf[i] / (den * den * den); }
        const double p_norm = sqrt(pn2), phi = p_norm - Delta, phi_prime = -dsum / p_norm;
        alpha_lower = -phi / phi_prime;
    }
    double alpha = *alpha_io;
    if (!full_rank && alpha == 0.0) alpha = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
    for (int it = 0; it < 10; it++) {
        if (alpha < alpha_lower || alpha > alpha_upper) alpha = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
        double pn2 = 0, dsum = 0;
        for (int i = 0; i < PK_N; i++) { const double den = s[i] * s[i] + alpha; const double q = suf[i] / den; pn2 += q * q; dsum += suf[i] * suf[i] / (den * den * den); }
        const double p_norm = sqrt(pn2), phi = p_norm - Delta, phi_prime = -dsum / p_norm;
        if (phi < 0) alpha_upper = alpha;
        const double ratio = phi / phi_prime;
        alpha_lower = fmax(alpha_lower, alpha - ratio);
        alpha -= (phi + Delta) * ratio / Delta;
        if (fabs(phi) < 0.01 * Delta) break;
    }
    for (int i = 0; i < PK_N; i++) t[i] = suf[i] / (s[i] * s[i] + alpha);
    for (int i = 0; i < PK_N; i++) { double a = 0; for (int k = 0; k < PK_N; k++) a += V[i * PK_N + k] * t[k]; p[i] = -a; }
    const double pn = pk_norm(p, PK_N);
    for (int i = 0; i < PK_N; i++) p[i] *= Delta / pn;
    *alpha_io = alpha;
}

static double pk_step_to_bound(const double *x, const double *sv, const double *lb, const double *ub, int *hits)
{
    double steps[PK_N], mn = INFINITY;
    for (int i = 0; i < PK_N; i++) {
        steps[i] = INFINITY;
        if (sv[i] != 0) steps[i] = fmax((lb[i] - x[i]) / sv[i], (ub[i] - x[i]) / sv[i]);
        if (steps[i] < mn) mn = steps[i];
    }
    if (hits) for (int i = 0; i < PK_N; i++) hits[i] = (steps[i] == mn) ? (sv[i] > 0) - (sv[i] < 0) : 0;
    return mn;
}

/* 0.5 * s^T (J_h^T J_h + diag) s + g^T s */
static double pk_quad(const double *Jh, int m, const double *diag, const double *g, const double *sv)
{
    double q = 0;
    for (int r = 0; r < m; r++) { double a = 0; for (int k = 0; k < PK_N; k++) a += Jh[r * PK_N + k] * sv[k]; q += a * a; }
    for (int k = 0; k < PK_N; k++) q += sv[k] * diag[k] * sv[k];
    return 0.5 * q + pk_dot(sv, g, PK_N);
}
static void pk_quad_1d(const double *Jh, int m, const double *diag, const double *g, const double *sv, const double *s0,
                       double *a, double *b, double *c)
{
    double aa = 0, bb = pk_dot(g, sv, PK_N), cc = 0, uu = 0, uv = 0;
    for (int r = 0; r < m; r++) {
        double v = 0, u = 0;
        for (int k = 0; k < PK_N; k++) { v += Jh[r * PK_N + k] * sv[k]; if (s0) u += Jh[r * PK_N + k] * s0[k]; }
        aa += v * v; uu += u * u; uv += u * v;
    }
    for (int k = 0; k < PK_N; k++) aa += sv[k] * diag[k] * sv[k];
    aa *= 0.5;
    if (s0) {
        bb += uv;
        cc = 0.5 * uu + pk_dot(g, s0, PK_N);
        for (int k = 0; k < PK_N; k++) { bb += s0[k] * diag[k] * sv[k]; cc += 0.5 * s0[k] * diag[k] * s0[k]; }
    }
    *a = aa; *b = bb; if (c) *c = cc;
}
static double pk_min_quad_1d(double a, double b, double lo, double hi, double c, double *tmin)
{
    double tt[3] = {lo, hi, 0}; int nt = 2;
    if (a != 0) { const double ex = -0.5 * b / a; if (lo < ex && ex < hi) tt[nt++] = ex; }
    double best = INFINITY; *tmin = lo;
    for (int i = 0; i < nt; i++) { const double y = tt[i] * (a * tt[i] + b) + c; if (y < best) { best = y; *tmin = tt[i]; } }
    return best;
}

/* trf.py select_step; p, p_h are modified like the numpy arrays are */
static double pk_select_step(const double *x, const double *Jh, int m, const double *diag_h, const double *g_h, double *p, double *p_h,
                             const double *d, double Delta, const double *lb, const double *ub, double theta,
                             double *step, double *step_h)
{
    double xp[PK_N];
    int inb = 1;
    for (int i = 0; i < PK_N; i++) { xp[i] = x[i] + p[i]; if (!(xp[i] >= lb[i] && xp[i] <= ub[i])) inb = 0; }
    if (inb) {
        for (int i = 0; i < PK_N; i++) { step[i] = p[i]; step_h[i] = p_h[i]; }
        return -pk_quad(Jh, m, diag_h, g_h, p_h);
    }
    int hits[PK_N];
    const double p_stride = pk_step_to_bound(x, p, lb, ub, hits);
    double r_h[PK_N], r[PK_N], x_on_bound[PK_N];
    for (int i = 0; i < PK_N; i++) { r_h[i] = hits[i] ? -p_h[i] : p_h[i]; r[i] = d[i] * r_h[i]; }
    for (int i = 0; i < PK_N; i++) { p[i] *= p_stride; p_h[i] *= p_stride; x_on_bound[i] = x[i] + p[i]; }
    /* intersect_trust_region(p_h, r_h, Delta): positive root */
    double to_tr;
    {
        const double a = pk_dot(r_h, r_h, PK_N), b = pk_dot(p_h, r_h, PK_N), c = pk_dot(p_h, p_h, PK_N) - Delta * Delta;
        const double dd = sqrt(b * b - a * c), q = -(b + copysign(dd, b));
        const double t1 = q / a, t2 = c / q;
        to_tr = t1 < t2 ? t2 : t1;
    }
    const double to_bound = pk_step_to_bound(x_on_bound, r, lb, ub, 0);
    double r_stride = fmin(to_bound, to_tr), r_stride_l, r_stride_u;
    if (r_stride > 0) { r_stride_l = (1 - theta) * p_stride / r_stride; r_stride_u = (r_stride == to_bound) ? theta * to_bound : to_tr; }
    else { r_stride_l = 0; r_stride_u = -1; }
    double r_value = INFINITY;
    if (r_stride_l <= r_stride_u) {
        double a, b, c;
        pk_quad_1d(Jh, m, diag_h, g_h, r_h, p_h, &a, &b, &c);
        r_value = pk_min_quad_1d(a, b, r_stride_l, r_stride_u, c, &r_stride);
        for (int i = 0; i < PK_N; i++) { r_h[i] = r_h[i] * r_stride + p_h[i]; r[i] = r_h[i] * d[i]; }
    }
    for (int i = 0; i < PK_N; i++) { p[i] *= theta; p_h[i] *= theta; }
    const double p_value = pk_quad(Jh, m, diag_h, g_h, p_h);
    double ag_h[PK_N], ag[PK_N];
    for (int i = 0; i < PK_N; i++) { ag_h[i] = -g_h[i]; ag[i] = d[i] * ag_h[i]; }
    const double to_tr2 = Delta / pk_norm(ag_h, PK_N);
    const double to_bound2 = pk_step_to_bound(x, ag, lb, ub, 0);
    double ag_stride = to_bound2 < to_tr2 ? theta * to_bound2 : to_tr2;
    double a, b;
    pk_quad_1d(Jh, m, diag_h, g_h, ag_h, 0, &a, &b, 0);
    const double ag_value = pk_min_quad_1d(a, b, 0, ag_stride, 0, &ag_stride);
    for (int i = 0; i < PK_N; i++) { ag_h[i] *= ag_stride; ag[i] *= ag_stride; }
    const double *ss, *sh; double val;
    if (p_value < r_value && p_value < ag_value) { ss = p; sh = p_h; val = p_value; }
    else if (r_value < p_value && r_value < ag_value) { ss = r; sh = r_h; val = r_value; }
    else { ss = ag; sh = ag_h; val = ag_value; }
    for (int i = 0; i < PK_N; i++) { step[i] = ss[i]; step_h[i] = sh[i]; }
    return -val;
}

/* roi: box x box float64 window (row-major, rows = y).  popt = a, xc, yc, s, b.  Returns scipy's status
 * (0 max_nfev, 1 gtol, 2 ftol, 3 xtol, 4 both). */
int orc_peak_fit(const double *roi, int box, double *popt, int *nfev_out)
{
    const int m = box * box, M = m + PK_N;
    const double lb[PK_N] = {0, -INFINITY, -INFINITY, 0, 0}, ub[PK_N] = {INFINITY, INFINITY, INFINITY, INFINITY, INFINITY};
    const double ftol = 1e-8, xtol = 1e-8, gtol = 1e-8;
    const int max_nfev = 100 * PK_N;
    double x[PK_N], mx = roi[0], mn = roi[0];
    for (int i = 1; i < m; i++) { if (roi[i] > mx) mx = roi[i]; if (roi[i] < mn) mn = roi[i]; }
    x[0] = mx; x[1] = 0; x[2] = 0; x[3] = 1; x[4] = mn;
    /* make_strictly_feasible(x0, lb, ub, rstep=1e-10) */
    for (int i = 0; i < PK_N; i++) {
        if (isfinite(lb[i]) && x[i] - lb[i] <= 1e-10 * fmax(1.0, fabs(lb[i]))) x[i] = lb[i] + 1e-10 * fmax(1.0, fabs(lb[i]));
        if (x[i] < lb[i] || x[i] > ub[i]) x[i] = 0.5 * (lb[i] + ub[i]);
    }
    double f[PK_MAXM], f_new[PK_MAXM], J[PK_MAXM * PK_N], Jaug[(PK_MAXM + PK_N) * PK_N], Jh[PK_MAXM * PK_N];
    double g[PK_N], v[PK_N], dv[PK_N], d[PK_N], diag_h[PK_N], g_h[PK_N], V[PK_N * PK_N], s[PK_N], suf[PK_N];
    int nfev = 1, njac = 0;
    pk_fun(x, roi, box, f);
    pk_jac(x, f, roi, box, lb, ub, J, &njac);
    double cost = 0.5 * pk_dot(f, f, m);
    for (int k = 0; k < PK_N; k++) { double a = 0; for (int r = 0; r < m; r++) a += J[r * PK_N + k] * f[r]; g[k] = a; }
    /* CL_scaling_vector */
#define PK_CL()                                                                                   \
    for (int i = 0; i < PK_N; i++) {                                                              \
        v[i] = 1; dv[i] = 0;                                                                      \
        if (g[i] < 0 && isfinite(ub[i])) { v[i] = ub[i] - x[i]; dv[i] = -1; }                    \
        if (g[i] > 0 && isfinite(lb[i])) { v[i] = x[i] - lb[i]; dv[i] = 1; }                     \
    }
    PK_CL();
    double Delta;
    { double t[PK_N]; for (int i = 0; i < PK_N; i++) t[i] = x[i] / sqrt(v[i]); Delta = pk_norm(t, PK_N); if (Delta == 0) Delta = 1.0; }
    double alpha = 0.0;
    int status = -1;
    for (;;) {
        PK_CL();
        double g_norm = 0;
        for (int i = 0; i < PK_N; i++) g_norm = fmax(g_norm, fabs(g[i] * v[i]));
        if (g_norm < gtol) status = 1;
        if (status != -1 || nfev == max_nfev) break;
        for (int i = 0; i < PK_N; i++) { d[i] = sqrt(v[i]); diag_h[i] = g[i] * dv[i]; g_h[i] = d[i] * g[i]; }
        for (int r = 0; r < m; r++) for (int k = 0; k < PK_N; k++) { Jh[r * PK_N + k] = J[r * PK_N + k] * d[k]; Jaug[r * PK_N + k] = Jh[r * PK_N + k]; }
        for (int r = 0; r < PK_N; r++) for (int k = 0; k < PK_N; k++) Jaug[(m + r) * PK_N + k] = (r == k) ? sqrt(diag_h[k]) : 0.0;
        pk_svd(Jaug, M, V, s);
        /* suf = s * (U^T f_aug) = (U S)^T f_aug; f_aug = (f, 0) */
        for (int k = 0; k < PK_N; k++) { double a = 0; for (int r = 0; r < m; r++) a += Jaug[r * PK_N + k] * f[r]; suf[k] = a; }
        const double theta = fmax(0.995, 1 - g_norm);
        double actual_reduction = -1, cost_new = cost, x_new[PK_N], step[PK_N], step_h[PK_N];
        while (actual_reduction <= 0 && nfev < max_nfev) {
            double p_h[PK_N], p[PK_N];
            pk_solve_tr(m, suf, s, V, Delta, &alpha, p_h);
            for (int i = 0; i < PK_N; i++) p[i] = d[i] * p_h[i];
            const double predicted = pk_select_step(x, Jh, m, diag_h, g_h, p, p_h, d, Delta, lb, ub, theta, step, step_h);
            /* make_strictly_feasible(x + step, rstep=0) */
            for (int i = 0; i < PK_N; i++) {
                x_new[i] = x[i] + step[i];
                if (x_new[i] <= lb[i]) x_new[i] = nextafter(lb[i], ub[i]);
                if (x_new[i] >= ub[i]) x_new[i] = nextafter(ub[i], lb[i]);
            }
            pk_fun(x_new, roi, box, f_new);
            nfev++;
            const double step_h_norm = pk_norm(step_h, PK_N);
            int finite = 1;
            for (int r = 0; r < m; r++) if (!isfinite(f_new[r])) finite = 0;
            if (!finite) { Delta = 0.25 * step_h_norm; continue; }
            cost_new = 0.5 * pk_dot(f_new, f_new, m);
            actual_reduction = cost - cost_new;
            /* update_tr_radius */
            double ratio, Delta_new = Delta;
            if (predicted > 0) ratio = actual_reduction / predicted;
            else if (predicted == 0 && actual_reduction == 0) ratio = 1;
            else ratio = 0;
            if (ratio < 0.25) Delta_new = 0.25 * step_h_norm;
            else if (ratio > 0.75 && step_h_norm > 0.95 * Delta) Delta_new = Delta * 2.0;
            const double step_norm = pk_norm(step, PK_N), x_norm = pk_norm(x, PK_N);
            const int ftol_ok = actual_reduction < ftol * cost && ratio > 0.25;
            const int xtol_ok = step_norm < xtol * (xtol + x_norm);
            if (ftol_ok && xtol_ok) status = 4; else if (ftol_ok) status = 2; else if (xtol_ok) status = 3;
            if (status != -1) break;
            alpha *= Delta / Delta_new;
            Delta = Delta_new;
        }
        if (actual_reduction > 0) {
            for (int i = 0; i < PK_N; i++) x[i] = x_new[i];
            for (int r = 0; r < m; r++) f[r] = f_new[r];
            cost = cost_new;
            pk_jac(x, f, roi, box, lb, ub, J, &njac);
            for (int k = 0; k < PK_N; k++) { double a = 0; for (int r = 0; r < m; r++) a += J[r * PK_N + k] * f[r]; g[k] = a; }
        }
    }
    if (status == -1) status = 0;
    for (int i = 0; i < PK_N; i++) popt[i] = x[i];
    if (nfev_out) *nfev_out = nfev;
    return status;
}
