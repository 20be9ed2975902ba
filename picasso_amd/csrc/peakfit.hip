// peakfit.hip — sub-pixel peak of a cross-correlation window on the device (picasso/imageprocess.py:121-141).
//
// The reference fits a*exp(-0.5*((x-xc)^2+(y-yc)^2)/s^2)+b to the box x box window around the correlation maximum with
// scipy.optimize.curve_fit(p0=[max,0,0,1,min], bounds=([0,-inf,-inf,0,0], inf)), i.e. least_squares(method='trf',
// jac='2-point', x_scale=1, ftol=xtol=gtol=1e-8, max_nfev=500, tr_solver='exact') — third-party arithmetic (scipy
// 1.15.3: _lsq/trf.py trf_bounds, _lsq/common.py, _numdiff.py), restated from the published Trust Region Reflective
// algorithm (Branch, Coleman & Li 1999; More 1977) with a one-sided Jacobi SVD in place of LAPACK's.  RCC has n(n-1)/2 of
// these 25-number fits (300 for config 4's 25 segments, 4 950 for 100), 13 ms each as a scipy call on the host and the bulk of
// the reference's undrift time.
//
// One WAVEFRONT per pair of segment images (round 5; rounds 2 - 4: one lane per pair with the residuals, the Jacobian and its
// SVD copy in 35 KB of scratch memory per lane).  The m = box^2 residual rows (<= 225) and the five rows the bounds add to
// the Jacobian sit on the lanes, row r on lane r % 64 as element r / 64 (PK_E = 4 per lane): the model evaluation, the
// forward differences and the column rotations of the Jacobi SVD are row-parallel; the m-long sums (J^T f, the rotation
// angles, the quadratic model) are per-lane partial sums + a butterfly over the wavefront, which leaves every lane the SAME
// bits (both partners of a stage add the same two numbers) — so the five-parameter trust-region logic is computed by
// every lane alike and the control flow never diverges.  No scratch, no LDS.
#include <math.h>

#include <vector>

#include "pmi_common.h"

#pragma clang fp contract(off)

namespace pmi {
namespace pk {

#define PK_N 5
#define PK_E 4                      // rows per lane: (15 * 15 + 5 + 63) / 64
#define PK_EPS 2.220446049250313e-16

__device__ __forceinline__ double wsum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ double wmax(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ double wmin(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmin(v, __shfl_xor(v, off));
    return v;
}

// the residuals of this lane's rows (rows >= m: 0)
__device__ __forceinline__ void pk_fun(const double (&x)[PK_N], const double (&data)[PK_E], int box, int m, int lane, double (&f)[PK_E])
{
    const int h = box / 2;
#pragma unroll
    for (int e = 0; e < PK_E; e++) {
        const int r = lane + 64 * e;
        double v = 0.0;
        if (r < m) {
            const int i = r / box, j = r - i * box;
            const double xx = (double)(j - h) - x[1], yy = (double)(i - h) - x[2];
            const double ex = -0.5 * (xx * xx + yy * yy) / (x[3] * x[3]);
            v = (x[0] * exp(ex) + x[4]) - data[e];
        }
        f[e] = v;
    }
}

__device__ __forceinline__ double pk_norm5(const double (&v)[PK_N]) { double s = 0; for (int i = 0; i < PK_N; i++) s += v[i] * v[i]; return sqrt(s); }
__device__ __forceinline__ double pk_dot5(const double (&a)[PK_N], const double (&b)[PK_N]) { double s = 0; for (int i = 0; i < PK_N; i++) s += a[i] * b[i]; return s; }

/* approx_derivative(method='2-point', rel_step=None, bounds): J[e][k], this lane's rows */
__device__ __forceinline__ void pk_jac(const double (&x0)[PK_N], const double (&f0)[PK_E], const double (&data)[PK_E], int box, int m, int lane,
                                       const double (&lb)[PK_N], const double (&ub)[PK_N], double (&J)[PK_E][PK_N])
{
    const double rstep = sqrt(PK_EPS);
#pragma unroll
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
        double x1[PK_N], f1[PK_E];
#pragma unroll
        for (int k = 0; k < PK_N; k++) x1[k] = x0[k];
        x1[i] += h;
        const double dx = x1[i] - x0[i];
        pk_fun(x1, data, box, m, lane, f1);
#pragma unroll
        for (int e = 0; e < PK_E; e++) J[e][i] = (lane + 64 * e < m) ? (f1[e] - f0[e]) / dx : 0.0;
    }
}

/* one-sided Jacobi SVD of A (this lane's rows of an M x 5 matrix, overwritten by U*S); V (5 x 5, every lane the same) */
__device__ __forceinline__ void pk_svd(double (&A)[PK_E][PK_N], double (&V)[PK_N][PK_N], double (&s)[PK_N])
{
#pragma unroll
    for (int i = 0; i < PK_N; i++)
#pragma unroll
        for (int j = 0; j < PK_N; j++) V[i][j] = (i == j);
    for (int sweep = 0; sweep < 60; sweep++) {
        int rotated = 0;
#pragma unroll
        for (int p = 0; p < PK_N - 1; p++)
#pragma unroll
            for (int q = p + 1; q < PK_N; q++) {
                double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
                for (int e = 0; e < PK_E; e++) {
                    const double ap = A[e][p], aq = A[e][q];
                    alpha += ap * ap; beta += aq * aq; gamma += ap * aq;
                }
                alpha = wsum(alpha); beta = wsum(beta); gamma = wsum(gamma);
                if (gamma == 0.0 || fabs(gamma) <= PK_EPS * sqrt(alpha * beta)) continue;
                rotated = 1;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
#pragma unroll
                for (int e = 0; e < PK_E; e++) {
                    const double ap = A[e][p], aq = A[e][q];
                    A[e][p] = c * ap - sn * aq;
                    A[e][q] = sn * ap + c * aq;
                }
#pragma unroll
                for (int r = 0; r < PK_N; r++) {
                    const double vp = V[r][p], vq = V[r][q];
                    V[r][p] = c * vp - sn * vq;
                    V[r][q] = sn * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
#pragma unroll
    for (int i = 0; i < PK_N; i++) {
        double n2 = 0;
#pragma unroll
        for (int e = 0; e < PK_E; e++) n2 += A[e][i] * A[e][i];
        s[i] = sqrt(wsum(n2));
    }
}

/* common.py solve_lsq_trust_region: suf[i] = s[i] * (U^T f)[i] */
__device__ __forceinline__ void pk_solve_tr(int m, const double (&suf)[PK_N], const double (&s)[PK_N], const double (&V)[PK_N][PK_N], double Delta,
                                            double *alpha_io, double (&p)[PK_N])
{
    double smax = 0, smin = INFINITY;
    for (int i = 0; i < PK_N; i++) { if (s[i] > smax) smax = s[i]; if (s[i] < smin) smin = s[i]; }
    const int full_rank = smin > PK_EPS * m * smax;
    double t[PK_N];
    if (full_rank) {
        for (int i = 0; i < PK_N; i++) t[i] = (suf[i] / s[i]) / s[i];        /* uf / s */
        for (int i = 0; i < PK_N; i++) { double a = 0; for (int k = 0; k < PK_N; k++) a += V[i][k] * t[k]; p[i] = -a; }
        if (pk_norm5(p) <= Delta) { *alpha_io = 0.0; return; }
    }
    double alpha_upper = pk_norm5(suf) / Delta, alpha_lower = 0.0;
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
    for (int i = 0; i < PK_N; i++) { double a = 0; for (int k = 0; k < PK_N; k++) a += V[i][k] * t[k]; p[i] = -a; }
    const double pn = pk_norm5(p);
    for (int i = 0; i < PK_N; i++) p[i] *= Delta / pn;
    *alpha_io = alpha;
}

__device__ __forceinline__ double pk_step_to_bound(const double (&x)[PK_N], const double (&sv)[PK_N], const double (&lb)[PK_N], const double (&ub)[PK_N], int *hits)
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

/* 0.5 * s^T (J_h^T J_h + diag) s + g^T s; J_h = J d, this lane's rows */
__device__ __forceinline__ double pk_quad(const double (&J)[PK_E][PK_N], const double (&d)[PK_N], const double (&diag)[PK_N], const double (&g)[PK_N],
                                          const double (&sv)[PK_N])
{
    double q = 0;
#pragma unroll
    for (int e = 0; e < PK_E; e++) { double a = 0; for (int k = 0; k < PK_N; k++) a += (J[e][k] * d[k]) * sv[k]; q += a * a; }
    q = wsum(q);
    for (int k = 0; k < PK_N; k++) q += sv[k] * diag[k] * sv[k];
    return 0.5 * q + pk_dot5(sv, g);
}
__device__ __forceinline__ void pk_quad_1d(const double (&J)[PK_E][PK_N], const double (&d)[PK_N], const double (&diag)[PK_N], const double (&g)[PK_N],
                                           const double (&sv)[PK_N], const double *s0, double *a, double *b, double *c)
{
    double aa = 0, bb = pk_dot5(g, sv), cc = 0, uu = 0, uv = 0;
#pragma unroll
    for (int e = 0; e < PK_E; e++) {
        double v = 0, u = 0;
        for (int k = 0; k < PK_N; k++) { const double jh = J[e][k] * d[k]; v += jh * sv[k]; if (s0) u += jh * s0[k]; }
        aa += v * v; uu += u * u; uv += u * v;
    }
    aa = wsum(aa);
    if (s0) { uu = wsum(uu); uv = wsum(uv); }
    for (int k = 0; k < PK_N; k++) aa += sv[k] * diag[k] * sv[k];
    aa *= 0.5;
    if (s0) {
        bb += uv;
        double gs0 = 0;
        for (int k = 0; k < PK_N; k++) gs0 += g[k] * s0[k];
        cc = 0.5 * uu + gs0;
        for (int k = 0; k < PK_N; k++) { bb += s0[k] * diag[k] * sv[k]; cc += 0.5 * s0[k] * diag[k] * s0[k]; }
    }
    *a = aa; *b = bb; if (c) *c = cc;
}
__device__ __forceinline__ double pk_min_quad_1d(double a, double b, double lo, double hi, double c, double *tmin)
{
    double tt[3] = {lo, hi, 0}; int nt = 2;
    if (a != 0) { const double ex = -0.5 * b / a; if (lo < ex && ex < hi) tt[nt++] = ex; }
    double best = INFINITY; *tmin = lo;
    for (int i = 0; i < nt; i++) { const double y = tt[i] * (a * tt[i] + b) + c; if (y < best) { best = y; *tmin = tt[i]; } }
    return best;
}

/* trf.py select_step; p, p_h are modified like the numpy arrays are */
__device__ __forceinline__ double pk_select_step(const double (&x)[PK_N], const double (&J)[PK_E][PK_N], const double (&diag_h)[PK_N], const double (&g_h)[PK_N],
                                                 double (&p)[PK_N], double (&p_h)[PK_N], const double (&d)[PK_N], double Delta, const double (&lb)[PK_N],
                                                 const double (&ub)[PK_N], double theta, double (&step)[PK_N], double (&step_h)[PK_N])
{
    double xp[PK_N];
    int inb = 1;
    for (int i = 0; i < PK_N; i++) { xp[i] = x[i] + p[i]; if (!(xp[i] >= lb[i] && xp[i] <= ub[i])) inb = 0; }
    if (inb) {
        for (int i = 0; i < PK_N; i++) { step[i] = p[i]; step_h[i] = p_h[i]; }
        return -pk_quad(J, d, diag_h, g_h, p_h);
    }
    int hits[PK_N];
    const double p_stride = pk_step_to_bound(x, p, lb, ub, hits);
    double r_h[PK_N], r[PK_N], x_on_bound[PK_N];
    for (int i = 0; i < PK_N; i++) { r_h[i] = hits[i] ? -p_h[i] : p_h[i]; r[i] = d[i] * r_h[i]; }
    for (int i = 0; i < PK_N; i++) { p[i] *= p_stride; p_h[i] *= p_stride; x_on_bound[i] = x[i] + p[i]; }
    /* intersect_trust_region(p_h, r_h, Delta): positive root */
    double to_tr;
    {
        const double a = pk_dot5(r_h, r_h), b = pk_dot5(p_h, r_h), c = pk_dot5(p_h, p_h) - Delta * Delta;
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
        pk_quad_1d(J, d, diag_h, g_h, r_h, p_h, &a, &b, &c);
        r_value = pk_min_quad_1d(a, b, r_stride_l, r_stride_u, c, &r_stride);
        for (int i = 0; i < PK_N; i++) { r_h[i] = r_h[i] * r_stride + p_h[i]; r[i] = r_h[i] * d[i]; }
    }
    for (int i = 0; i < PK_N; i++) { p[i] *= theta; p_h[i] *= theta; }
    const double p_value = pk_quad(J, d, diag_h, g_h, p_h);
    double ag_h[PK_N], ag[PK_N];
    for (int i = 0; i < PK_N; i++) { ag_h[i] = -g_h[i]; ag[i] = d[i] * ag_h[i]; }
    const double to_tr2 = Delta / pk_norm5(ag_h);
    const double to_bound2 = pk_step_to_bound(x, ag, lb, ub, 0);
    double ag_stride = to_bound2 < to_tr2 ? theta * to_bound2 : to_tr2;
    double a, b;
    pk_quad_1d(J, d, diag_h, g_h, ag_h, nullptr, &a, &b, nullptr);
    const double ag_value = pk_min_quad_1d(a, b, 0, ag_stride, 0, &ag_stride);
    for (int i = 0; i < PK_N; i++) { ag_h[i] *= ag_stride; ag[i] *= ag_stride; }
    double val;
    if (p_value < r_value && p_value < ag_value) { for (int i = 0; i < PK_N; i++) { step[i] = p[i]; step_h[i] = p_h[i]; } val = p_value; }
    else if (r_value < p_value && r_value < ag_value) { for (int i = 0; i < PK_N; i++) { step[i] = r[i]; step_h[i] = r_h[i]; } val = r_value; }
    else { for (int i = 0; i < PK_N; i++) { step[i] = ag[i]; step_h[i] = ag_h[i]; } val = ag_value; }
    return -val;
}

// data: this lane's rows of the box x box float64 window (row-major, rows = y).  popt = a, xc, yc, s, b (every lane the same).
// Returns scipy's status (0 max_nfev, 1 gtol, 2 ftol, 3 xtol, 4 both).
__device__ __forceinline__ int peak_fit(const double (&data)[PK_E], int box, int lane, double mx, double mn, double (&popt)[PK_N], int *nfev_out)
{
    const int m = box * box;
    const double lb[PK_N] = {0, -INFINITY, -INFINITY, 0, 0}, ub[PK_N] = {INFINITY, INFINITY, INFINITY, INFINITY, INFINITY};
    const double ftol = 1e-8, xtol = 1e-8, gtol = 1e-8;
    const int max_nfev = 100 * PK_N;
    double x[PK_N];
    x[0] = mx; x[1] = 0; x[2] = 0; x[3] = 1; x[4] = mn;
    /* make_strictly_feasible(x0, lb, ub, rstep=1e-10) */
    for (int i = 0; i < PK_N; i++) {
        if (isfinite(lb[i]) && x[i] - lb[i] <= 1e-10 * fmax(1.0, fabs(lb[i]))) x[i] = lb[i] + 1e-10 * fmax(1.0, fabs(lb[i]));
        if (x[i] < lb[i] || x[i] > ub[i]) x[i] = 0.5 * (lb[i] + ub[i]);
    }
    double f[PK_E], f_new[PK_E], J[PK_E][PK_N], A[PK_E][PK_N];
    double g[PK_N], v[PK_N], dv[PK_N], d[PK_N], diag_h[PK_N], g_h[PK_N], V[PK_N][PK_N], s[PK_N], suf[PK_N];
    int nfev = 1;
    pk_fun(x, data, box, m, lane, f);
    pk_jac(x, f, data, box, m, lane, lb, ub, J);
    auto dot_ff = [&](const double (&a)[PK_E]) { double t = 0; for (int e = 0; e < PK_E; e++) t += a[e] * a[e]; return wsum(t); };
    auto grad = [&]() { for (int k = 0; k < PK_N; k++) { double a = 0; for (int e = 0; e < PK_E; e++) a += J[e][k] * f[e]; g[k] = wsum(a); } };
    double cost = 0.5 * dot_ff(f);
    grad();
    /* CL_scaling_vector */
#define PK_CL()                                                                                   \
    for (int i = 0; i < PK_N; i++) {                                                              \
        v[i] = 1; dv[i] = 0;                                                                      \
        if (g[i] < 0 && isfinite(ub[i])) { v[i] = ub[i] - x[i]; dv[i] = -1; }                    \
        if (g[i] > 0 && isfinite(lb[i])) { v[i] = x[i] - lb[i]; dv[i] = 1; }                     \
    }
    PK_CL();
    double Delta;
    { double t[PK_N]; for (int i = 0; i < PK_N; i++) t[i] = x[i] / sqrt(v[i]); Delta = pk_norm5(t); if (Delta == 0) Delta = 1.0; }
    double alpha = 0.0;
    int status = -1;
    for (;;) {
        PK_CL();
        double g_norm = 0;
        for (int i = 0; i < PK_N; i++) g_norm = fmax(g_norm, fabs(g[i] * v[i]));
        if (g_norm < gtol) status = 1;
        if (status != -1 || nfev == max_nfev) break;
        for (int i = 0; i < PK_N; i++) { d[i] = sqrt(v[i]); diag_h[i] = g[i] * dv[i]; g_h[i] = d[i] * g[i]; }
        // J_h = J d, and below it the five rows diag(sqrt(diag_h)) (rows m .. m + 4, on the lanes like the others)
#pragma unroll
        for (int e = 0; e < PK_E; e++) {
            const int r = lane + 64 * e;
#pragma unroll
            for (int k = 0; k < PK_N; k++) A[e][k] = r < m ? J[e][k] * d[k] : (r - m == k ? sqrt(diag_h[k]) : 0.0);
        }
        pk_svd(A, V, s);
        /* suf = s * (U^T f_aug) = (U S)^T f_aug; f_aug = (f, 0) */
        for (int k = 0; k < PK_N; k++) { double a = 0; for (int e = 0; e < PK_E; e++) a += A[e][k] * f[e]; suf[k] = wsum(a); }
        const double theta = fmax(0.995, 1 - g_norm);
        double actual_reduction = -1, cost_new = cost, x_new[PK_N], step[PK_N], step_h[PK_N];
        while (actual_reduction <= 0 && nfev < max_nfev) {
            double p_h[PK_N], p[PK_N];
            pk_solve_tr(m, suf, s, V, Delta, &alpha, p_h);
            for (int i = 0; i < PK_N; i++) p[i] = d[i] * p_h[i];
            const double predicted = pk_select_step(x, J, diag_h, g_h, p, p_h, d, Delta, lb, ub, theta, step, step_h);
            /* make_strictly_feasible(x + step, rstep=0) */
            for (int i = 0; i < PK_N; i++) {
                x_new[i] = x[i] + step[i];
                if (x_new[i] <= lb[i]) x_new[i] = nextafter(lb[i], ub[i]);
                if (x_new[i] >= ub[i]) x_new[i] = nextafter(ub[i], lb[i]);
            }
            pk_fun(x_new, data, box, m, lane, f_new);
            nfev++;
            const double step_h_norm = pk_norm5(step_h);
            bool bad = false;
            for (int e = 0; e < PK_E; e++) bad = bad || !isfinite(f_new[e]);
            if (__builtin_amdgcn_ballot_w64(bad) != 0ull) { Delta = 0.25 * step_h_norm; continue; }
            cost_new = 0.5 * dot_ff(f_new);
            actual_reduction = cost - cost_new;
            /* update_tr_radius */
            double ratio, Delta_new = Delta;
            if (predicted > 0) ratio = actual_reduction / predicted;
            else if (predicted == 0 && actual_reduction == 0) ratio = 1;
            else ratio = 0;
            if (ratio < 0.25) Delta_new = 0.25 * step_h_norm;
            else if (ratio > 0.75 && step_h_norm > 0.95 * Delta) Delta_new = Delta * 2.0;
            const double step_norm = pk_norm5(step), x_norm = pk_norm5(x);
            const int ftol_ok = actual_reduction < ftol * cost && ratio > 0.25;
            const int xtol_ok = step_norm < xtol * (xtol + x_norm);
            if (ftol_ok && xtol_ok) status = 4; else if (ftol_ok) status = 2; else if (xtol_ok) status = 3;
            if (status != -1) break;
            alpha *= Delta / Delta_new;
            Delta = Delta_new;
        }
        if (actual_reduction > 0) {
            for (int i = 0; i < PK_N; i++) x[i] = x_new[i];
            for (int e = 0; e < PK_E; e++) f[e] = f_new[e];
            cost = cost_new;
            pk_jac(x, f, data, box, m, lane, lb, ub, J);
            grad();
        }
    }
    if (status == -1) status = 0;
    for (int i = 0; i < PK_N; i++) popt[i] = x[i];
    if (nfev_out) *nfev_out = nfev;
    return status;
}

struct PairIn { int32_t y_max, x_max, valid; };

constexpr int PK_WAVES = 4;           // wavefronts (pairs) per workgroup

// shift_yx[2p], [2p+1] = (-yc, -xc) of imageprocess.py:155-159; status: scipy's termination code, -1 no fit
// (empty image or truncated window: shift (0, 0)), -2 the window minimum is negative (curve_fit raises: p0 infeasible),
// -3 the window holds a NaN or an infinity (curve_fit(check_finite=True) raises ValueError); 0 = the fit ran into
// max_nfev, where curve_fit raises RuntimeError("Optimal parameters not found") — the host mirror raises both
__global__ __launch_bounds__(PK_WAVES * 64, 1) void peak_fit_kernel(const double *__restrict__ rois, const int32_t *__restrict__ peaks /* y, x, valid */,
                                int64_t n_pairs, int box, int64_t Y, int64_t X, int64_t Y_, int64_t X_,
                                double *__restrict__ shift_yx, double *__restrict__ popt_out, int32_t *__restrict__ status)
{
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * PK_WAVES + (threadIdx.x >> 6);       // one wavefront per pair
    if (p >= n_pairs) return;
    double sy = 0.0, sx = 0.0, popt[PK_N] = {0, 0, 0, 0, 0};
    int st = -1;
    if (peaks[3 * p + 2] == 1) {
        const int m = box * box;
        const double *roi = rois + p * m;
        double data[PK_E];
        double mn = INFINITY, mx = -INFINITY;
        bool finite = true;
#pragma unroll
        for (int e = 0; e < PK_E; e++) {
            const int r = lane + 64 * e;
            data[e] = r < m ? roi[r] : 0.0;
            if (r < m) { mn = fmin(mn, data[e]); mx = fmax(mx, data[e]); finite = finite && isfinite(data[e]); }      // fmin drops a NaN: test it
        }
        mn = wmin(mn); mx = wmax(mx);
        if (__builtin_amdgcn_ballot_w64(!finite) != 0ull) st = -3;
        else if (mn < 0.0) st = -2;
        else {
            int nfev = 0;
            st = peak_fit(data, box, lane, mx, mn, popt, &nfev);
            double xc = popt[1] + (double)X_ + (double)peaks[3 * p + 1];
            double yc = popt[2] + (double)Y_ + (double)peaks[3 * p];
            xc -= floor((double)X / 2.0);
            yc -= floor((double)Y / 2.0);
            sy = -yc; sx = -xc;
        }
    }
    if (lane == 0) {
        shift_yx[2 * p] = sy; shift_yx[2 * p + 1] = sx;
        if (popt_out) for (int k = 0; k < PK_N; k++) popt_out[p * PK_N + k] = popt[k];
        status[p] = st;
    }
}

}  // namespace pk

// fit the n_pairs windows that are resident at d_rois (peaks: y, x, valid triples on the host)
int rcc_fit_peaks(const double *d_rois, const int32_t *h_peaks3, int64_t n_pairs, int box, int64_t Y, int64_t X, int64_t Y_,
                  int64_t X_, double *h_shift_yx, int32_t *h_status)
{
    if (n_pairs <= 0) return PMI_OK;
    void *ptr = nullptr;
    int rc = scratch(SCR_FIT, (size_t)n_pairs * (3 * 4 + 2 * 8 + 4) + 64, &ptr);
    if (rc != PMI_OK) return rc;
    double *d_shift = (double *)ptr;
    int32_t *d_peaks = (int32_t *)(d_shift + 2 * n_pairs), *d_status = d_peaks + 3 * n_pairs;
    PMI_HIP(hipMemcpy(d_peaks, h_peaks3, (size_t)n_pairs * 12, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pk::peak_fit_kernel, dim3((unsigned)((n_pairs + pk::PK_WAVES - 1) / pk::PK_WAVES)), dim3(pk::PK_WAVES * 64), 0, 0, d_rois, d_peaks, n_pairs, box,
                       Y, X, Y_, X_, d_shift, (double *)nullptr, d_status);
    PMI_HIP(hipGetLastError());
    PMI_HIP(hipMemcpy(h_shift_yx, d_shift, (size_t)n_pairs * 16, hipMemcpyDeviceToHost));
    PMI_HIP(hipMemcpy(h_status, d_status, (size_t)n_pairs * 4, hipMemcpyDeviceToHost));
    return PMI_OK;
}

}  // namespace pmi

extern "C" {

// the fit alone, on host windows (n, box, box) float64: popt (n, 5) = a, xc, yc, s, b; status as above
int pmi_peak_fit(const double *rois, int64_t n, int box, double *popt, int32_t *status)
{
    using namespace pmi;
    if (pmi_device_count() < 1) { set_error("no HIP device"); return PMI_ERR_NODEVICE; }
    if (n == 0) return PMI_OK;
    if (!rois || !popt || !status || n < 0 || box < 3 || box > 15 || !(box & 1)) { set_error("peak fit: bad arguments"); return PMI_ERR_ARG; }
    void *ptr = nullptr;
    const size_t rb = (size_t)n * box * box * 8;
    int rc = scratch(SCR_STAGE_D, rb + (size_t)n * (12 + 16 + 40 + 4) + 64, &ptr);
    if (rc != PMI_OK) return rc;
    double *d_rois = (double *)ptr, *d_shift = d_rois + (size_t)n * box * box, *d_popt = d_shift + 2 * n;
    int32_t *d_peaks = (int32_t *)(d_popt + 5 * n), *d_status = d_peaks + 3 * n;
    std::vector<int32_t> pk3((size_t)n * 3, 0);
    for (int64_t i = 0; i < n; i++) pk3[(size_t)i * 3 + 2] = 1;
    PMI_HIP(hipMemcpy(d_rois, rois, rb, hipMemcpyHostToDevice));
    PMI_HIP(hipMemcpy(d_peaks, pk3.data(), (size_t)n * 12, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pk::peak_fit_kernel, dim3((unsigned)((n + pk::PK_WAVES - 1) / pk::PK_WAVES)), dim3(pk::PK_WAVES * 64), 0, 0, d_rois, d_peaks, n, box, (int64_t)0,
                       (int64_t)0, (int64_t)0, (int64_t)0, d_shift, d_popt, d_status);
    PMI_HIP(hipGetLastError());
    PMI_HIP(hipMemcpy(popt, d_popt, (size_t)n * 40, hipMemcpyDeviceToHost));
    PMI_HIP(hipMemcpy(status, d_status, (size_t)n * 4, hipMemcpyDeviceToHost));
    return PMI_OK;
}

}  // extern "C"
