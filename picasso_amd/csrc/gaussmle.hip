// gaussmle.hip — Poisson-MLE fit of a pixel-integrated 2D Gaussian, one
// wavefront (64 lanes) per spot, pixel(s)-per-lane, DPP wave reductions.
//
// Replaces picasso/gaussmle.py:28-168 (initial parameters), :268-383 (model and
// derivatives), :533-670 (_mlefit_sigma, _update_theta_sigma), :745-884
// (_mlefit_sigmaxy, _update_theta_sigmaxy), :673-742 / :887-954 (CRLB, log-
// likelihood) and, when fed from the movie, picasso/localize.py:917-931
// (_cut_spots_numba) + :1101-1112 (_to_photons).
//
// Arithmetic: the Newton iteration runs in float32 (the reference keeps its
// state in float32 and promotes intermediates to float64; the difference is
// below 1e-5 px after convergence).  Initial sums, the Fisher
// matrix and its inverse are float64 like the reference.  The quirks of the
// reference are kept: max_step from the INITIAL theta with theta[4] for both x
// and y; the 10e-3 / 10e4 literals; sign(num)*max_step vs sign(num*max_step)
// on a zero denominator; the un-multiplied terms of the isotropic second
// derivative (gaussmle.py:380-382).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fit_common.h"

namespace pmi {

// Spots to fit again in the reference's arithmetic (gaussmle_strict.hip): the largest tested step D of iteration
// kk (counted from 1) lies within the margin [eps_lo, eps_hi) of eps, widened with the iteration count (a slow fit
// takes many steps close to eps and the float32 loop drifts from the reference by more than a few ulps)
__device__ __forceinline__ unsigned borderline(float D, int kk, float eps_lo, float eps_hi)
{
    const float wide = fmaxf(1.0f, (float)(kk - 1) * FIT_MARGIN_GROWTH);
    const float epsf = 0.5f * (eps_lo + eps_hi), epsm = 0.5f * (eps_hi - eps_lo) * wide;
    return ((D >= epsf - epsm && D < epsf + epsm) ? FLAG_MARGIN : 0u) | (kk > FIT_SLOW_ITERATIONS ? FLAG_SLOW : 0u);
}
// the wobble test of newton_step (gaussmle_g8.hip) for one parameter: st = this iteration's step, h = its history
struct StepHistory { float prev, prev2, wprev; int run; };
__device__ __forceinline__ bool wobbles(float st, float value, int kk, StepHistory &h)
{
    const float w = kk >= 3 ? (st - 2.0f * h.prev) + h.prev2 : 0.0f;      // kk counts from 1 here
    const bool wob = w * h.wprev < 0.0f && fabsf(w) > FIT_WOBBLE_RATIO * fabsf(h.wprev) && fabsf(w) > FIT_WOBBLE_FLOOR * fabsf(value);
    h.run = wob ? h.run + 1 : 0;
    h.wprev = w; h.prev2 = h.prev; h.prev = st;
    return h.run >= FIT_WOBBLE_RUN || (h.run >= 2 && kk >= FIT_WOBBLE_LATE);
}

// NP = params (5: "sigma", 6: "sigmaxy"); PPL = pixels per lane = ceil(box^2/64)
template <int NP, int PPL, bool FROM_MOVIE>
__global__ __launch_bounds__(FIT_NT) void mle_fit_kernel(FitParams p, int stages)
{
    __shared__ float s_spot[FIT_WAVES][FIT_MAXPIX + 7];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float *sp = s_spot[wid];
    const int box = p.box, npix = box * box, h = box / 2;
    int64_t n = p.N;
    if (p.d_n) { int64_t dn = *p.d_n; n = dn < n ? dn : n; }

    // static per-lane pixel coordinates
    int pi[PPL], pj[PPL];
    bool act[PPL];
#pragma unroll
    for (int s = 0; s < PPL; s++) {
        int pix = lane + 64 * s;
        act[s] = pix < npix;
        int q = act[s] ? pix : 0;
        pj[s] = q / box;            // row (y)
        pi[s] = q - pj[s] * box;    // column (x)
    }

    for (;;) {
        unsigned long long sidx = 0;
        if (lane == 0) sidx = atomicAdd(p.queue, 1ull);
        sidx = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(sidx >> 32)) << 32) |
               (unsigned)__builtin_amdgcn_readfirstlane((int)(sidx & 0xffffffffu));
        if (p.final_list && !(stages & FIT_STAGE_NEWTON)) {      // the spots of the second re-fit only
            if (sidx >= (unsigned long long)*p.final_list_n) break;
            sidx = (unsigned long long)p.final_list[sidx];
        } else {
            sidx += (unsigned long long)p.first;
            if ((int64_t)sidx >= n) break;
        }

        // ---- load the spot (photons) -----------------------------------
        float data[PPL];
#pragma unroll
        for (int s = 0; s < PPL; s++) {
            float v = 0.f;
            if (act[s]) {
                if (FROM_MOVIE) {
                    int64_t fr = p.frame[sidx], yy = p.y[sidx], xx = p.x[sidx];
                    float raw = load_movie_px(p.movie, p.dtype, (fr * p.Y + (yy - h + pj[s])) * p.X + (xx - h + pi[s]));
                    // localize.py:1112, float32, in this order (no contraction possible: sub, mul, div)
                    v = div_const((raw - p.baseline) * p.sensitivity, p.gdiv);
                } else {
                    v = p.spots[(int64_t)sidx * npix + lane + 64 * s];
                }
                sp[lane + 64 * s] = v;
            }
            data[s] = v;
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();

        float th[6];
        int kk = 0;
        if (!(stages & FIT_STAGE_NEWTON)) {          // thetas in memory (re-fitted spots included): final stage only
#pragma unroll
            for (int l = 0; l < 6; l++) th[l] = p.thetas[sidx * 6 + l];
            kk = p.iterations[sidx];
        } else {
        // ---- initial parameters (gaussmle.py:28-139) -------------------
        double ds = 0.0, dsy = 0.0, dsx = 0.0;
        float fmin_local = INFINITY;
#pragma unroll
        for (int s = 0; s < PPL; s++)
            if (act[s]) {
                double v = (double)data[s];
                ds += v; dsy += v * (double)pj[s]; dsx += v * (double)pi[s];
                // 3x3 edge-clipped mean, float64 sum in (m, n) order, float32 store
                int k = pj[s], l = pi[s];
                int m0 = max(0, k - 1), m1 = min(box, k + 2), n0 = max(0, l - 1), n1 = min(box, l + 2);
                double nsum = 0.0;
                for (int m = m0; m < m1; m++)
                    for (int q = n0; q < n1; q++) nsum += (double)sp[m * box + q];
                float filt = (float)(nsum / (double)((m1 - m0) * (n1 - n0)));
                fmin_local = fminf(fmin_local, filt);
            }
        double sum = wave_sum_d(ds), sy_ = wave_sum_d(dsy), sx_ = wave_sum_d(dsx);
        const float bg0 = wave_min(fmin_local);
        double com_y, com_x;
        if (sum <= 0.0) { sum = 0.01; com_y = (box - 1) / 2.0; com_x = (box - 1) / 2.0; }
        else { com_y = sy_ / sum; com_x = sx_ / sum; }
        double photons = sum - (double)(box * box) * (double)bg0;
        photons = (photons != photons) ? photons : (photons > 1.0 ? photons : 1.0);
        // centre row / column second moments of (spot - bg) about box//2
        double a_sdy = 0.0, a_sdx = 0.0, a_sy = 0.0, a_sx = 0.0;
#pragma unroll
        for (int s = 0; s < PPL; s++)
            if (act[s]) {
                float vm = data[s] - bg0;
                if (pi[s] == h) { double d2 = (double)((pj[s] - h) * (pj[s] - h)); a_sdy += (double)vm * d2; a_sy += (double)vm; }
                if (pj[s] == h) { double d2 = (double)((pi[s] - h) * (pi[s] - h)); a_sdx += (double)vm * d2; a_sx += (double)vm; }
            }
        a_sdy = wave_sum_d(a_sdy); a_sy = wave_sum_d(a_sy); a_sdx = wave_sum_d(a_sdx); a_sx = wave_sum_d(a_sx);
        double isy = sqrt(a_sdy / a_sy), isx = sqrt(a_sdx / a_sx);
        if (!isfinite(isy)) isy = 0.01;
        if (!isfinite(isx)) isx = 0.01;
        if (isx == 0) isx = 0.01;
        if (isy == 0) isy = 0.01;

        th[0] = (float)com_x; th[1] = (float)com_y; th[2] = (float)photons; th[3] = bg0;
        if (NP == 6) { th[4] = (float)isx; th[5] = (float)isy; }
        else { th[4] = (float)((isx + isy) / 2); th[5] = 0.f; }
        float ms[6];
        ms[0] = th[4]; ms[1] = th[4];
        ms[2] = (float)(0.1 * (double)th[2]); ms[3] = (float)(0.1 * (double)th[3]);
        ms[4] = (float)(0.2 * (double)th[4]); ms[5] = (float)(0.2 * (double)th[5]);

        float old_x = th[0], old_y = th[1], old_sx = th[4], old_sy = th[5];
        StepHistory hist[6];
#pragma unroll
        for (int l = 0; l < 6; l++) hist[l] = {0.f, 0.f, 0.f, 0};
        unsigned flagged = 0u;
        bool conv = false;
        while (kk < p.max_it) {
            kk++;
            float num[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, den[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            float top = 0.f, was[6];
#pragma unroll
            for (int l = 0; l < 6; l++) was[l] = th[l];
            const float sgy = NP == 6 ? th[5] : th[4];
#pragma unroll
            for (int s = 0; s < PPL; s++) {
                Terms tx = gauss_terms((float)pi[s], th[0], th[4]);
                Terms ty = gauss_terms((float)pj[s], th[1], sgy);
                const float N_ = th[2];
                float du[6], d2[6];
                du[0] = N_ * ty.E * tx.A;  d2[0] = N_ * ty.E * tx.A2;
                du[1] = N_ * tx.E * ty.A;  d2[1] = N_ * tx.E * ty.A2;
                du[2] = tx.E * ty.E;       d2[2] = 0.f;
                du[3] = 1.f;               d2[3] = 0.f;
                if (NP == 6) {
                    du[4] = N_ * ty.E * tx.S;  d2[4] = N_ * ty.E * tx.S2;
                    du[5] = N_ * tx.E * ty.S;  d2[5] = N_ * tx.E * ty.S2;
                } else {
                    du[4] = N_ * (ty.E * tx.S + tx.E * ty.S);
                    d2[4] = N_ * ty.E * tx.S2 + 2.f * tx.S * ty.S + tx.E * ty.S2;   // gaussmle.py:380-382
                    du[5] = 0.f; d2[5] = 0.f;
                }
                const float model = N_ * tx.E * ty.E + th[3];
                float cf = 0.f, df = 0.f;
                if (model > 10e-3f) {
                    const float r = 1.0f / model;
                    cf = data[s] * r - 1.f;
                    df = data[s] * r * r;
                }
                if (!act[s]) { cf = 0.f; df = 0.f; }
                top = fmaxf(top, fmaxf(fabsf(cf), fabsf(df)));
                cf = np_minf(cf, 10e4f);
                df = np_minf(df, 10e4f);
#pragma unroll
                for (int l = 0; l < NP; l++) {
                    num[l] += cf * du[l];
                    den[l] += cf * d2[l] - df * du[l] * du[l];
                }
            }
#pragma unroll
            for (int l = 0; l < NP; l++) {
                num[l] = wave_sum(num[l]); den[l] = wave_sum(den[l]);
                flagged |= den[l] >= 0.0f ? FLAG_CURVATURE : 0u;          // curvature not negative: chaotic trajectory
            }
            flagged |= __any(top >= FIT_TOP_FLAG) ? FLAG_WILD : 0u;   // a pixel far off the model: the float32 sums are not trusted

            if (NP == 6) {                                  // gaussmle.py:860-884
#pragma unroll
                for (int l = 0; l < 6; l++) {
                    if (den[l] == 0.0f) th[l] = th[l] - np_signf(num[l]) * ms[l];
                    else th[l] = th[l] - np_minf(np_maxf(num[l] / den[l], -ms[l]), ms[l]);
                }
                th[2] = np_maxf(th[2], 1.0f); th[3] = np_maxf(th[3], 0.01f);
                th[4] = np_maxf(th[4], 0.01f); th[5] = np_maxf(th[5], 0.01f);
                // largest tested step; NaN (never converged in the reference) must win the maximum
                const float dx = fabsf(old_x - th[0]), dy = fabsf(old_y - th[1]), dsx_ = fabsf(old_sx - th[4]), dsy_ = fabsf(old_sy - th[5]);
                const float D = __uint_as_float(max(max(__float_as_uint(dx), __float_as_uint(dy)), max(__float_as_uint(dsx_), __float_as_uint(dsy_))));
                conv = (double)D < p.eps;
                flagged |= borderline(D, kk, p.eps_lo, p.eps_hi) | ((th[4] < FIT_NARROW_SIGMA || th[5] < FIT_NARROW_SIGMA) ? FLAG_NARROW : 0u);
#pragma unroll
                for (int l = 0; l < 6; l++) flagged |= wobbles(was[l] - th[l], th[l], kk, hist[l]) ? FLAG_SWING : 0u;
                if (conv) break;
                old_x = th[0]; old_y = th[1]; old_sx = th[4]; old_sy = th[5];
            } else {                                        // gaussmle.py:647-670
#pragma unroll
                for (int l = 0; l < 5; l++) {
                    float upd;
                    if (den[l] == 0.0f) upd = np_signf(num[l] * ms[l]);
                    else upd = np_minf(np_maxf(num[l] / den[l], -ms[l]), ms[l]);
                    th[l] = th[l] - upd;
                }
                th[2] = np_maxf(th[2], 1.0f); th[3] = np_maxf(th[3], 0.01f);
                th[4] = np_maxf(th[4], 0.01f); th[4] = np_minf(th[4], (float)box);
                const float dx = fabsf(old_x - th[0]), dy = fabsf(old_y - th[1]);
                const float D = __uint_as_float(max(__float_as_uint(dx), __float_as_uint(dy)));
                conv = (double)D < p.eps;
                flagged |= borderline(D, kk, p.eps_lo, p.eps_hi) | (th[4] < FIT_NARROW_SIGMA ? FLAG_NARROW : 0u);
#pragma unroll
                for (int l = 0; l < 5; l++) flagged |= wobbles(was[l] - th[l], th[l], kk, hist[l]) ? FLAG_SWING : 0u;
                if (conv) break;
                old_x = th[0]; old_y = th[1];
            }
        }
        if (lane == 0) {
            float *to = p.thetas + sidx * 6;
#pragma unroll
            for (int l = 0; l < 5; l++) to[l] = th[l];
            to[5] = NP == 6 ? th[5] : th[4];
            p.iterations[sidx] = kk;
            if (flagged && p.flag_list) {
                p.flag_list[atomicAdd(p.flag_count, 1u)] = (int32_t)sidx;
                count_flag_reasons(p.flag_reasons, flagged);
            }
        }
        }   // FIT_STAGE_NEWTON
        if (!(stages & FIT_STAGE_FINAL)) { __builtin_amdgcn_wave_barrier(); continue; }

        // ---- CRLB and log-likelihood (gaussmle.py:673-742, 887-954) ----
        double Mloc[21];
#pragma unroll
        for (int e = 0; e < 21; e++) Mloc[e] = 0.0;
        float ll_loc = 0.f;
        {
            const float sgy = NP == 6 ? th[5] : th[4];
#pragma unroll
            for (int s = 0; s < PPL; s++) {
                Terms tx = gauss_terms((float)pi[s], th[0], th[4]);
                Terms ty = gauss_terms((float)pj[s], th[1], sgy);
                const float N_ = th[2];
                float du[6];
                du[0] = N_ * ty.E * tx.A;
                du[1] = N_ * tx.E * ty.A;
                du[2] = tx.E * ty.E;
                du[3] = 1.f;
                if (NP == 6) { du[4] = N_ * ty.E * tx.S; du[5] = N_ * tx.E * ty.S; }
                else { du[4] = N_ * (ty.E * tx.S + tx.E * ty.S); du[5] = 0.f; }
                const float model = N_ * tx.E * ty.E + th[3];
                if (act[s]) {
                    const double inv = 1.0 / (double)model;
                    int e = 0;
#pragma unroll
                    for (int k = 0; k < NP; k++)
#pragma unroll
                        for (int l = k; l < NP; l++) { Mloc[e] += (double)(du[l] * du[k]) * inv; e++; }
                    if (model > 0.f) {
                        const float d = data[s];
                        if (d > 0.f) ll_loc += d * __logf(model / d) - (model - d);
                        else ll_loc += -model;
                    }
                }
            }
        }
        // Fisher matrix (upper triangle) -> crlb_kernel; theta, log-likelihood, iterations out
        double Msum[21];
        {
            int e = 0;
#pragma unroll
            for (int k = 0; k < NP; k++)
#pragma unroll
                for (int l = k; l < NP; l++) { Msum[e] = wave_sum_d(Mloc[e]); e++; }
        }
        const float ll = wave_sum(ll_loc);
        if (lane == 0) {
            p.loglik[sidx] = ll;
            double *fo = p.fisher + (sidx - (unsigned long long)p.first) * FISHER_STRIDE;
#pragma unroll
            for (int e = 0; e < NP * (NP + 1) / 2; e++) fo[e] = Msum[e];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// CRLB = diag(np.linalg.pinv(M)) (gaussmle.py:737, :950), one thread per spot.  Fast path:
// LDL^T inverse when trace(M) * trace(M^-1) < 1e12, which keeps every eigenvalue far above
// pinv's 1e-15 cutoff (pinv == inverse).  Otherwise a cyclic-Jacobi restatement of pinv's
// semantics: eigenvalues <= 1e-15 * max are dropped, so singular directions give 0, not inf.
template <int NP>
__device__ void pinv_diag_jacobi(const double *Min, double *diag)
{
    double A[36], V[36];
    bool nonfinite = false;
    for (int i = 0; i < NP * NP; i++) { A[i] = Min[i]; if (!isfinite(A[i])) nonfinite = true; }
    if (nonfinite) { for (int i = 0; i < NP; i++) diag[i] = NAN; return; }
    for (int i = 0; i < NP; i++) for (int j = 0; j < NP; j++) V[i * NP + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int a = 0; a < NP; a++) for (int b = a + 1; b < NP; b++) off += A[a * NP + b] * A[a * NP + b];
        if (off == 0.0) break;
        for (int a = 0; a < NP; a++)
            for (int b = a + 1; b < NP; b++) {
                double apq = A[a * NP + b];
                if (apq == 0.0) continue;
                double tau = (A[b * NP + b] - A[a * NP + a]) / (2.0 * apq);
                double tt = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                double c = 1.0 / sqrt(1.0 + tt * tt), s = tt * c;
                for (int k = 0; k < NP; k++) { double x = A[k * NP + a], y = A[k * NP + b]; A[k * NP + a] = c * x - s * y; A[k * NP + b] = s * x + c * y; }
                for (int k = 0; k < NP; k++) { double x = A[a * NP + k], y = A[b * NP + k]; A[a * NP + k] = c * x - s * y; A[b * NP + k] = s * x + c * y; }
                for (int k = 0; k < NP; k++) { double x = V[k * NP + a], y = V[k * NP + b]; V[k * NP + a] = c * x - s * y; V[k * NP + b] = s * x + c * y; }
            }
    }
    double smax = 0.0;
    for (int i = 0; i < NP; i++) smax = fmax(smax, fabs(A[i * NP + i]));
    const double cutoff = 1e-15 * smax;
    for (int i = 0; i < NP; i++) {
        double acc = 0.0;
        for (int k = 0; k < NP; k++) { double lam = A[k * NP + k]; if (fabs(lam) > cutoff) acc += V[i * NP + k] * V[i * NP + k] / lam; }
        diag[i] = acc;
    }
}

// list / list_n: only these spots (the second re-fit); unstable / unstable_n (first pass, re-fit mode): spots whose iteration
// does not contract at the fitted theta (FIT_UNSTABLE_LMAX, fit_common.h) are appended for that second re-fit
template <int NP>
__global__ __launch_bounds__(256) void crlb_kernel(const double *__restrict__ fisher, int64_t first, int64_t N,
                                                   const int64_t *__restrict__ d_n, float *__restrict__ crlbs,
                                                   const int32_t *__restrict__ list, const unsigned *__restrict__ list_n,
                                                   int32_t *__restrict__ unstable, unsigned *__restrict__ unstable_n,
                                                   unsigned *__restrict__ reasons, const unsigned char *__restrict__ refit_mark,
                                                   const unsigned char *__restrict__ accept)
{
    int64_t n = N;
    if (d_n) { int64_t dn = *d_n; n = dn < n ? dn : n; }
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t sidx = first + t;
    if (list) {
        if (t >= (int64_t)*list_n) return;
        sidx = (int64_t)list[t];
        t = sidx - first;
    } else if (sidx >= n) return;
    else if (accept && !accept[sidx]) return;            // a candidate identify's exact stage rejected: never fitted
    const double *f = fisher + t * FISHER_STRIDE;
    double M[36];
    {
        int e = 0;
#pragma unroll
        for (int k = 0; k < NP; k++)
#pragma unroll
            for (int l = k; l < NP; l++) { const double v = f[e++]; M[k * NP + l] = v; M[l * NP + k] = v; }
    }
    if (unstable && !(refit_mark && refit_mark[t])) {      // (a spot the Newton loop flagged carries the reference's bits already)
        // lambda_max of C = D^-1/2 M D^-1/2 by power iteration (C is symmetric, positive semi-definite, unit diagonal)
        double dinv[NP], v[NP], lam = 0.0;
        bool ok = true;
#pragma unroll
        for (int i = 0; i < NP; i++) { const double m = M[i * NP + i]; ok = ok && m > 0.0 && m < 1e300; dinv[i] = 1.0 / sqrt(m); v[i] = 1.0 + 0.125 * (double)i; }
        if (ok) {
            // A certificate first: lambda_max^8 <= trace(C^8) = ||C^4||_F^2 (C symmetric, positive semi-definite).  Below the
            // threshold with it, the power iteration — whose estimate never exceeds lambda_max — could not flag the spot
            // either: photon data has lambda_max 1.2 ... 1.7, trace(C^8)^(1/8) within 2 % of it, and whole wavefronts skip the
            // sixteen iterations (crlb_kernel 0.21 -> 0.14 ms per 1e6 spots).
            double C[NP * NP], C2[NP * NP], t8 = 0.0;
#pragma unroll
            for (int i = 0; i < NP; i++)
#pragma unroll
                for (int k = 0; k < NP; k++) C[i * NP + k] = M[i * NP + k] * dinv[i] * dinv[k];
#pragma unroll
            for (int i = 0; i < NP; i++)
#pragma unroll
                for (int k = i; k < NP; k++) {
                    double a = 0.0;
#pragma unroll
                    for (int q = 0; q < NP; q++) a += C[i * NP + q] * C[q * NP + k];
                    C2[i * NP + k] = a; C2[k * NP + i] = a;
                }
#pragma unroll
            for (int i = 0; i < NP; i++)
#pragma unroll
                for (int k = i; k < NP; k++) {
                    double a = 0.0;
#pragma unroll
                    for (int q = 0; q < NP; q++) a += C2[i * NP + q] * C2[q * NP + k];
                    t8 += (k == i ? 1.0 : 2.0) * a * a;
                }
            constexpr double L2 = (0.99 * FIT_UNSTABLE_LMAX) * (0.99 * FIT_UNSTABLE_LMAX), L8 = L2 * L2 * L2 * L2;
            ok = !(t8 <= L8);                          // (NaN: look closer)
        }
        if (ok) {
            for (int it = 0; it < 16; it++) {          // (second eigenvalue / first <= 0.7 on ill-conditioned fits: 0.7^16 = 3e-3)
                double u[NP], w[NP], nrm = 0.0, vn = 0.0;
#pragma unroll
                for (int i = 0; i < NP; i++) { u[i] = dinv[i] * v[i]; vn += v[i] * v[i]; }
#pragma unroll
                for (int i = 0; i < NP; i++) {
                    double a = 0.0;
#pragma unroll
                    for (int k = 0; k < NP; k++) a += M[i * NP + k] * u[k];
                    w[i] = dinv[i] * a; nrm += w[i] * w[i];
                }
                nrm = sqrt(nrm);
                lam = nrm / sqrt(vn);
                const double inv = 1.0 / nrm;
#pragma unroll
                for (int i = 0; i < NP; i++) v[i] = w[i] * inv;
            }
            if (lam > FIT_UNSTABLE_LMAX) {
                unstable[atomicAdd(unstable_n, 1u)] = (int32_t)sidx;
                if (reasons) atomicAdd(reasons + 6, 1u);
            }
        }
    }
    // (one reciprocal per pivot instead of a division per entry — 6 float64 divisions instead of 36; the results go to
    // float32, and are compared with the reference's pinv at 2e-3)
    double L[36], D[6], rD[6], diag[6];
    bool bad = false;
    double trM = 0.0;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        trM += M[i * NP + i];
#pragma unroll
        for (int c = 0; c <= i; c++) {
            double a = M[i * NP + c];
#pragma unroll
            for (int k = 0; k < c; k++) a -= L[i * NP + k] * L[c * NP + k] * D[k];
            if (c == i) { D[i] = a; rD[i] = 1.0 / a; if (!(a > 0.0)) bad = true; }
            else L[i * NP + c] = a * rD[c];
        }
    }
    // Linv = inverse of the unit lower-triangular L; diag(M^-1)_i = sum_k Linv[k][i]^2 / D[k]
    double Li[36];
#pragma unroll
    for (int i = 0; i < NP; i++)
#pragma unroll
        for (int c = 0; c < NP; c++) Li[i * NP + c] = (i == c) ? 1.0 : 0.0;
#pragma unroll
    for (int c = 0; c < NP; c++)
#pragma unroll
        for (int i = c + 1; i < NP; i++) {
            double a = 0.0;
#pragma unroll
            for (int k = c; k < i; k++) a -= L[i * NP + k] * Li[k * NP + c];
            Li[i * NP + c] = a;
        }
    double trInv = 0.0;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        double a = 0.0;
#pragma unroll
        for (int k = i; k < NP; k++) a += Li[k * NP + i] * Li[k * NP + i] * rD[k];
        diag[i] = a;
        trInv += a;
    }
    if (!(trM * trInv < 1e12)) bad = true;       // eigenvalue ratio could reach pinv's cutoff (or NaN)
    if (bad) pinv_diag_jacobi<NP>(M, diag);
    float *co = crlbs + sidx * 6;
#pragma unroll
    for (int l = 0; l < NP; l++) co[l] = (float)diag[l];
    if (NP == 5) co[5] = (float)diag[4];
}

template <int NP, bool FROM_MOVIE>
static void launch_fit_ppl(int ppl, dim3 grid, hipStream_t s, const FitParams &p, int stages)
{
    switch (ppl) {
    case 1: hipLaunchKernelGGL((mle_fit_kernel<NP, 1, FROM_MOVIE>), grid, dim3(FIT_NT), 0, s, p, stages); break;
    case 2: hipLaunchKernelGGL((mle_fit_kernel<NP, 2, FROM_MOVIE>), grid, dim3(FIT_NT), 0, s, p, stages); break;
    case 3: hipLaunchKernelGGL((mle_fit_kernel<NP, 3, FROM_MOVIE>), grid, dim3(FIT_NT), 0, s, p, stages); break;
    case 4: hipLaunchKernelGGL((mle_fit_kernel<NP, 4, FROM_MOVIE>), grid, dim3(FIT_NT), 0, s, p, stages); break;
    case 5: hipLaunchKernelGGL((mle_fit_kernel<NP, 5, FROM_MOVIE>), grid, dim3(FIT_NT), 0, s, p, stages); break;
    case 6: hipLaunchKernelGGL((mle_fit_kernel<NP, 6, FROM_MOVIE>), grid, dim3(FIT_NT), 0, s, p, stages); break;
    default: hipLaunchKernelGGL((mle_fit_kernel<NP, 7, FROM_MOVIE>), grid, dim3(FIT_NT), 0, s, p, stages); break;
    }
}


bool launch_fit_g8(const FitParams &p, int method, bool from_movie, int cu_count, float *state, int stages, hipStream_t s);   // gaussmle_g8.hip
void launch_fit_strict(const FitParams &p, int method, bool from_movie, const int32_t *list, const unsigned *list_n,
                       int64_t max_items, int cu_count, hipStream_t s);                                                    // gaussmle_strict.hip

constexpr int64_t FIT_BATCH = 1 << 22;      // spots per batch of fit_impl
// ---- the accepted candidates of a range of spots as an ascending list (deferred exact stage, fit_common.h) --------
// three small launches: counts per block of ACC_BLOCK spots, one workgroup turns them into offsets (and the total), the
// blocks write their spots' indices behind their offset
constexpr int ACC_BLOCK = 1024;
__global__ __launch_bounds__(256) void accept_count_kernel(const unsigned char *__restrict__ accept, int64_t first, int64_t N,
                                                           const int64_t *__restrict__ d_n, unsigned *__restrict__ blk_cnt)
{
    int64_t n = N;
    if (d_n) { const int64_t dn = *d_n; n = dn < n ? dn : n; }
    __shared__ unsigned s_c[4];
    const int64_t base = first + (int64_t)blockIdx.x * ACC_BLOCK;
    unsigned c = 0;
#pragma unroll
    for (int q = 0; q < ACC_BLOCK / 256; q++) {
        const int64_t sp = base + q * 256 + threadIdx.x;
        c += (unsigned)__popcll(__ballot(sp < n && accept[sp] != 0));
    }
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}
// counts -> exclusive offsets in place; *total = their sum (+ *carry_in when given: the accepted rows of an earlier range)
__global__ __launch_bounds__(1024) void accept_scan_kernel(unsigned *__restrict__ blk, int nblk, unsigned *__restrict__ total)
{
    __shared__ unsigned s_w[16];
    __shared__ unsigned s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int b0 = 0; b0 < nblk; b0 += 1024) {
        const int i = b0 + (int)threadIdx.x;
        const unsigned v = i < nblk ? blk[i] : 0u;
        unsigned inc = v;                                   // inclusive scan inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const unsigned t = __shfl_up(inc, off); if (lane >= off) inc += t; }
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        unsigned before = s_carry;
        for (int k = 0; k < w; k++) before += s_w[k];
        if (i < nblk) blk[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = s_carry;
}
__global__ __launch_bounds__(256) void accept_write_kernel(const unsigned char *__restrict__ accept, int64_t first, int64_t N,
                                                           const int64_t *__restrict__ d_n, const unsigned *__restrict__ blk_off,
                                                           int32_t *__restrict__ alist)
{
    int64_t n = N;
    if (d_n) { const int64_t dn = *d_n; n = dn < n ? dn : n; }
    __shared__ unsigned s_c[ACC_BLOCK / 64];
    const int64_t base = first + (int64_t)blockIdx.x * ACC_BLOCK;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long bal[ACC_BLOCK / 256];
#pragma unroll
    for (int q = 0; q < ACC_BLOCK / 256; q++) {
        const int64_t sp = base + q * 256 + threadIdx.x;
        bal[q] = __ballot(sp < n && accept[sp] != 0);
        if (lane == 0) s_c[q * 4 + w] = (unsigned)__popcll(bal[q]);
    }
    __syncthreads();
    const unsigned off0 = blk_off[blockIdx.x];
#pragma unroll
    for (int q = 0; q < ACC_BLOCK / 256; q++) {
        if ((bal[q] >> lane) & 1ull) {
            unsigned off = off0;
            for (int k = 0; k < q * 4 + w; k++) off += s_c[k];
            off += (unsigned)__popcll(bal[q] & ((1ull << lane) - 1ull));
            alist[off] = (int32_t)(base + q * 256 + threadIdx.x);
        }
    }
}
// blk: scratch of ceil(count / ACC_BLOCK) + 1 counters (the last one receives the total); list entries are absolute indices
static int build_accept_list(const unsigned char *accept, int64_t first, int64_t N, const int64_t *d_n, unsigned *blk,
                             int32_t *alist, hipStream_t s)
{
    const int64_t count = N - first;
    const int nblk = (int)((count + ACC_BLOCK - 1) / ACC_BLOCK);
    hipLaunchKernelGGL(accept_count_kernel, dim3((unsigned)nblk), dim3(256), 0, s, accept, first, N, d_n, blk);
    hipLaunchKernelGGL(accept_scan_kernel, dim3(1), dim3(1024), 0, s, blk, nblk, blk + nblk);
    hipLaunchKernelGGL(accept_write_kernel, dim3((unsigned)nblk), dim3(256), 0, s, accept, first, N, d_n, (const unsigned *)blk, alist);
    PMI_HIP(hipGetLastError());
    return PMI_OK;
}

// How the Newton loop is run (pmi_mle_set_mode):
//   PMI_MLE_FAST    float32 loop only (round-1 behaviour; a borderline |delta| < eps decision may fall the other way
//                   than in the reference's float64 arithmetic)
//   PMI_MLE_REFIT   float32 loop; spots whose decisive step came within `margin` (relative) of eps, or that ran into
//                   max_it, are re-fitted in the reference's arithmetic (default)
//   PMI_MLE_STRICT  every spot in the reference's arithmetic
static int g_mle_mode = PMI_MLE_REFIT;
static double g_mle_margin = 0.001;
// erf / exp of the reference-arithmetic kernel (pmi_mle_set_libm): glibc's bits (libm_glibc.h), the device library's, or
// glibc's where a result was seen to hang on them (see fit_impl)
static int g_mle_libm = PMI_LIBM_AUTO;
// flag statistics of the calling thread's last fit: a device buffer of its own (SCR_STATS of the thread's scratch bank:
// [0] = spots re-fitted, [1..FLAG_REASONS] = spots flagged per criterion), valid while the scratch generation stands
static thread_local const unsigned *g_last_stats[2] = {nullptr, nullptr};      // [1]: the second frame range of a fused call
static thread_local unsigned g_last_stats_generation = 0;
static thread_local int g_last_stats_device = 0;                                  // the device those buffers live on
static thread_local int g_last_stats_bank = 0;      // and the scratch bank they were taken from
static thread_local bool g_stats_second = false;                                  // the fit being queued is that second range

__global__ void zero_words_kernel(unsigned *__restrict__ a, int na, unsigned *__restrict__ b, int nb)
{
    for (int i = threadIdx.x; i < na; i += blockDim.x) a[i] = 0u;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) b[i] = 0u;
}

__global__ void flag_stats_kernel(const unsigned *__restrict__ flag_counts, int64_t nb, unsigned *__restrict__ stats)
{
    unsigned total = 0;
    for (int64_t b = 0; b < nb; b++) total += flag_counts[b];
    stats[0] = total;
}

static int mle_mode_now()
{
    static const char *env = getenv("PMI_MLE_MODE");      // "fast" | "refit" | "strict" overrides pmi_mle_set_mode
    if (env) {
        if (!strcmp(env, "fast")) return PMI_MLE_FAST;
        if (!strcmp(env, "strict")) return PMI_MLE_STRICT;
        if (!strcmp(env, "refit")) return PMI_MLE_REFIT;
    }
    return g_mle_mode;
}
static int mle_libm_now()
{
    static const char *env = getenv("PMI_MLE_LIBM");      // "auto" | "glibc" | "device" overrides pmi_mle_set_libm
    if (env) {
        if (!strcmp(env, "auto")) return PMI_LIBM_AUTO;
        if (!strcmp(env, "glibc")) return PMI_LIBM_GLIBC;
        if (!strcmp(env, "device")) return PMI_LIBM_DEVICE;
    }
    return g_mle_libm;
}

int fit_impl(FitParams p, int method, bool from_movie, hipStream_t s)
{
    if (p.box < 3 || p.box > PMI_MAX_BOX || (p.box & 1) == 0) { set_error("box must be odd, 3..%d (got %d)", PMI_MAX_BOX, p.box); return PMI_ERR_ARG; }
    if (method != PMI_MLE_SIGMA && method != PMI_MLE_SIGMAXY) { set_error("Method not available."); return PMI_ERR_ARG; }
    if (p.N < 0) { set_error("negative N"); return PMI_ERR_ARG; }
    if (p.N == 0) return PMI_OK;
    if (p.N > 0x7fffffffLL) { set_error("too many spots for one call"); return PMI_ERR_ARG; }
    const int g_cu_count = device_cu_count();
    const int mode = mle_mode_now();
    // Spots are processed in batches so that the Fisher scratch (168 B per spot) stays bounded;
    // every batch has its own queue words and flag counter.  With a device-side row count (d_n) the
    // batches past the count exit immediately.
    const int64_t BATCH = FIT_BATCH;
    const int64_t nb = (p.N + BATCH - 1) / BATCH;
    void *ptr = nullptr, *fptr = nullptr;
    int rc;
    // per batch: three queue words (Newton stage, final stage, final stage of the second re-fit) and two counters (spots
    // flagged by the Newton loop, spots found unstable by the Fisher pass)
    // (+ two queue words of the re-fit kernels over those two lists)
    if ((rc = scratch(SCR_FIT, (size_t)nb * 40 + 64, &ptr)) != PMI_OK) return rc;
    // per spot of a batch: 21 doubles of Fisher triangle + 12 floats of Newton start state + a slot in each of the two lists
    const size_t per_batch = (size_t)std::min<int64_t>(p.N, BATCH);
    // (+ with the deferred exact stage: the list of the accepted candidates of the batch and its block counters)
    const size_t acc_blocks = (per_batch + ACC_BLOCK - 1) / ACC_BLOCK + 2;
    if ((rc = scratch(SCR_STAGE_D, per_batch * (FISHER_STRIDE * sizeof(double) + 12 * sizeof(float) + 3 * sizeof(int32_t) + 1) + acc_blocks * sizeof(unsigned) + 32, &fptr)) != PMI_OK) return rc;
    float *state = reinterpret_cast<float *>((char *)fptr + per_batch * FISHER_STRIDE * sizeof(double));
    int32_t *flag_list = reinterpret_cast<int32_t *>(state + per_batch * 12);
    int32_t *unstable_list = flag_list + per_batch;
    int32_t *acc_list = unstable_list + per_batch;
    unsigned *acc_blk = reinterpret_cast<unsigned *>(acc_list + per_batch);
    unsigned char *refit_mark = reinterpret_cast<unsigned char *>(acc_blk + acc_blocks);
    unsigned long long *queues = (unsigned long long *)ptr;
    unsigned *flag_counts = (unsigned *)(queues + 3 * nb);           // [0, nb): flagged, [nb, 2 nb): unstable
    unsigned *strict_queues = flag_counts + 2 * nb;                  // [0, nb): first re-fit, [nb, 2 nb): second
    void *sptr = nullptr;
    if ((rc = scratch(SCR_STATS, 64, &sptr)) != PMI_OK) return rc;
    unsigned *stats = (unsigned *)sptr;
    // (one launch for the two small blocks; the marks of the re-fit are cleared by the start-value kernel where there is one)
    hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(256), 0, s, (unsigned *)ptr, (int)(nb * 10), stats, 16);
    p.flag_reasons = stats + 1;
    p.fisher = (double *)fptr;
    // erf / exp with glibc's bits: every launch (GLIBC), none (DEVICE), or the all-strict launch at any box and the re-fit
    // lists of boxes up to 5x5 (AUTO)
    const int libm = mle_libm_now();
    const bool libm_all = libm != PMI_LIBM_DEVICE;
    p.libm_glibc = libm == PMI_LIBM_GLIBC || (libm == PMI_LIBM_AUTO && p.box <= 5);
    static const char *menv = tuning_env("PMI_MLE_MARGIN");       // overrides the margin of pmi_mle_set_mode (tuning runs)
    const double margin = menv ? atof(menv) : g_mle_margin;
    // The tested step |delta| is a difference of float32 coordinates near box/2, i.e. a multiple of their ulp: the
    // two arithmetics differ by a few ulps there, so the margin is never smaller than four of them
    // (7x7: 9.5e-7 = eps/1000 at the default eps; 13x13: 1.9e-6; 17x17 and up: 3.8e-6).
    const double ulp = ldexp(1.0, (int)floor(log2(std::max(1.0, p.box / 2.0))) - 23);
    const double margin_abs = std::max(p.eps * margin, 4.0 * ulp);
    p.eps_lo = (float)(p.eps - margin_abs);
    p.eps_hi = (float)(p.eps + margin_abs);
    const int ppl = (p.box * p.box + 63) / 64;
    const int64_t Ntotal = p.N;
    ScopedKernelTimer tm(s, &g_last_times.fit_ms);
    static const bool force_wave_per_spot = tuning_env("PMI_FIT_WAVE_PER_SPOT") != nullptr;
    const bool g8 = !force_wave_per_spot && p.box <= 15;
    float *cut = nullptr;
    static const bool no_keep = tuning_env("PMI_MLE_NO_KEEP") != nullptr;
    if (g8 && from_movie && mode != PMI_MLE_STRICT && !no_keep) {
        void *cptr = nullptr;
        if ((rc = scratch(SCR_STAGE_C, per_batch * (size_t)(p.box * p.box) * sizeof(float), &cptr)) != PMI_OK) return rc;
        cut = (float *)cptr;
    }
    p.spots_out = nullptr;
    // the deferred exact stage lives in g8_init on the path that keeps the cut spots; anywhere else the rows are identifications
    if (p.ng_io && !(g8 && from_movie && cut)) { set_error("deferred exact stage outside the row-per-lane fit"); return PMI_ERR_ARG; }
    for (int64_t bi = 0; bi < nb; bi++) {
        p.first = bi * BATCH;
        p.N = std::min<int64_t>(Ntotal, p.first + BATCH);
        p.flag_list = mode == PMI_MLE_REFIT ? flag_list : nullptr;
        p.flag_count = flag_counts + bi;
        p.strict_queue = strict_queues + bi;
        p.refit_mark = mode == PMI_MLE_REFIT ? refit_mark : nullptr;
        if (p.refit_mark && !g8) PMI_HIP(hipMemsetAsync(refit_mark, 0, (size_t)(p.N - p.first), s));
        const int64_t count = p.N - p.first;
        const int64_t blocks = std::min<int64_t>((count + FIT_WAVES - 1) / FIT_WAVES, (int64_t)g_cu_count * 8);
        const dim3 grid((unsigned)std::max<int64_t>(blocks, 1));
        auto wave_per_spot = [&](int stages) {
            p.queue = queues + 3 * bi + (stages == FIT_STAGE_FINAL ? (p.final_list ? 2 : 1) : 0);
            if (method == PMI_MLE_SIGMAXY) {
                if (from_movie) launch_fit_ppl<6, true>(ppl, grid, s, p, stages); else launch_fit_ppl<6, false>(ppl, grid, s, p, stages);
            } else {
                if (from_movie) launch_fit_ppl<5, true>(ppl, grid, s, p, stages); else launch_fit_ppl<5, false>(ppl, grid, s, p, stages);
            }
        };
        // Newton stage: boxes <= 15 run eight or four spots per wavefront (gaussmle_g8.hip), boxes 17..21 one
        // wavefront per spot; then the flagged spots again in the reference's arithmetic; then Fisher matrix and
        // log-likelihood at the final thetas
        if (mode != PMI_MLE_STRICT && g8 && from_movie && cut) {
            // From a movie every stage would fetch the spot's B rows again — B cache lines of 128 B for B * B pixels, the
            // memory side of g8_init and a good part of the Fisher pass.  The start-value kernel keeps the photon values
            // it computed (2 lines per 7x7 spot, contiguous) and the later stages read those.
            FitParams pi = p;
            pi.spots_out = cut;
            launch_fit_g8(pi, method, true, g_cu_count, state, FIT_STAGE_INIT_ONLY, s);
            PMI_HIP(hipGetLastError());
            FitParams q = p;
            q.spots = cut - p.first * (int64_t)(p.box * p.box);      // indexed by the absolute spot number
            if (p.ng_io) {
                // g8_init decided the candidates: the Newton loop takes the accepted ones as a list
                int32_t *al = nb == 1 && p.alist_out ? p.alist_out : acc_list;
                unsigned *ab = nb == 1 && p.alist_out ? p.alist_blk_out : acc_blk;
                if ((rc = build_accept_list(p.accept, p.first, p.N, p.d_n, ab, al, s)) != PMI_OK) return rc;
                q.alist = al;
                q.alist_n = ab + (count + ACC_BLOCK - 1) / ACC_BLOCK;
            }
            launch_fit_g8(q, method, false, g_cu_count, state, FIT_STAGE_ITERATE_ONLY, s);
            PMI_HIP(hipGetLastError());
            if (mode == PMI_MLE_REFIT)
                launch_fit_strict(q, method, false, flag_list, q.flag_count, count, g_cu_count, s);
            PMI_HIP(hipGetLastError());
            launch_fit_g8(q, method, false, g_cu_count, state, FIT_STAGE_FINAL, s);
        } else {
        if (mode == PMI_MLE_STRICT) {
            FitParams ps = p;
            ps.libm_glibc = libm_all;
            launch_fit_strict(ps, method, from_movie, nullptr, nullptr, count, g_cu_count, s);
        } else {
            if (g8) launch_fit_g8(p, method, from_movie, g_cu_count, state, FIT_STAGE_NEWTON, s);
            else wave_per_spot(FIT_STAGE_NEWTON);
            PMI_HIP(hipGetLastError());
            if (mode == PMI_MLE_REFIT)
                launch_fit_strict(p, method, from_movie, flag_list, p.flag_count, count, g_cu_count, s);
        }
        PMI_HIP(hipGetLastError());
        if (g8) launch_fit_g8(p, method, from_movie, g_cu_count, state, FIT_STAGE_FINAL, s);
        else wave_per_spot(FIT_STAGE_FINAL);
        }
        PMI_HIP(hipGetLastError());
        const unsigned cb = (unsigned)((count + 255) / 256);
        unsigned *unstable_n = flag_counts + nb + bi;
        auto crlb = [&](const int32_t *list, const unsigned *list_n, int32_t *out, unsigned *out_n) {
            if (method == PMI_MLE_SIGMAXY)
                hipLaunchKernelGGL((crlb_kernel<6>), dim3(cb), dim3(256), 0, s, p.fisher, p.first, p.N, p.d_n, p.crlbs, list, list_n, out, out_n, p.flag_reasons, p.refit_mark, p.accept);
            else
                hipLaunchKernelGGL((crlb_kernel<5>), dim3(cb), dim3(256), 0, s, p.fisher, p.first, p.N, p.d_n, p.crlbs, list, list_n, out, out_n, p.flag_reasons, p.refit_mark, p.accept);
        };
        const bool second = mode == PMI_MLE_REFIT;
        crlb(nullptr, nullptr, second ? unstable_list : nullptr, second ? unstable_n : nullptr);
        PMI_HIP(hipGetLastError());
        if (second) {
            // Second re-fit: the spots whose iteration does not contract at the fitted theta (known only now, from the Fisher
            // matrix) — reference arithmetic, Fisher pass and inverse for these spots alone.  On photon data the list is empty
            // and the three launches exit at once.
            FitParams r = p;
            r.strict_queue = strict_queues + nb + bi;
            if (cut) { r.spots = cut - p.first * (int64_t)(p.box * p.box); }
            const bool r_movie = cut ? false : from_movie;
            launch_fit_strict(r, method, r_movie, unstable_list, unstable_n, count, g_cu_count, s);
            PMI_HIP(hipGetLastError());
            r.final_list = unstable_list; r.final_list_n = unstable_n;
            if (g8) launch_fit_g8(r, method, r_movie, g_cu_count, state, FIT_STAGE_FINAL, s);
            else {
                FitParams keep = p;
                p = r;
                wave_per_spot(FIT_STAGE_FINAL);
                p = keep;
            }
            PMI_HIP(hipGetLastError());
            crlb(unstable_list, unstable_n, nullptr, nullptr);
            PMI_HIP(hipGetLastError());
        }
    }
    hipLaunchKernelGGL(flag_stats_kernel, dim3(1), dim3(1), 0, s, flag_counts, mode == PMI_MLE_REFIT ? 2 * nb : 0, stats);
    PMI_HIP(hipGetLastError());
    g_last_stats[g_stats_second ? 1 : 0] = stats;
    if (!g_stats_second) g_last_stats[1] = nullptr;
    g_last_stats_device = current_device();
    g_last_stats_bank = scratch_user_bank();
    g_last_stats_generation = scratch_generation_of(g_last_stats_device, g_last_stats_bank, SCR_STATS);
    tm.stop();
    return PMI_OK;
}

static int read_last_stats(unsigned (&h)[16], hipStream_t s)
{
    for (unsigned &v : h) v = 0;
    if (!g_last_stats[0] || g_last_stats_generation != scratch_generation_of(g_last_stats_device, g_last_stats_bank, SCR_STATS)) return PMI_OK;      // no fit yet, or its buffers are gone
    PMI_HIP(hipStreamSynchronize(s));
    for (const unsigned *src : g_last_stats) {
        if (!src) continue;
        unsigned part[16];
        PMI_HIP(hipMemcpy(part, src, 64, hipMemcpyDeviceToHost));
        for (int i = 0; i < 16; i++) h[i] += part[i];
    }
    return PMI_OK;
}

// ---- get_spots -----------------------------------------------------------
__global__ void cut_spots_kernel(const void *__restrict__ movie, int dtype, int64_t Y, int64_t X,
                                 const int32_t *__restrict__ frame, const int32_t *__restrict__ y,
                                 const int32_t *__restrict__ x, int64_t N, const int64_t *__restrict__ d_n,
                                 int box, float baseline, float sensitivity, ConstDiv gdiv, float *__restrict__ spots)
{
    int64_t n = N;
    if (d_n) { int64_t dn = *d_n; n = dn < n ? dn : n; }
    const int npix = box * box, h = box / 2;
    int64_t total = n * npix;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t i = t / npix;
        int q = (int)(t - i * npix);
        int a = q / box, c = q - a * box;
        float raw = load_movie_px(movie, dtype, ((int64_t)frame[i] * Y + (y[i] - h + a)) * X + (x[i] - h + c));
        spots[t] = div_const((raw - baseline) * sensitivity, gdiv);
    }
}

// ---- locs_from_fits (gaussmle.py:957-1037) -------------------------------
struct LocCols { void *c[PMI_LOC_COLUMNS]; };
__global__ void locs_from_fits_kernel(const int32_t *__restrict__ frame, const int32_t *__restrict__ y,
                                      const int32_t *__restrict__ x, const float *__restrict__ ng,
                                      const float *__restrict__ th, const float *__restrict__ cr,
                                      const float *__restrict__ ll, const int32_t *__restrict__ it, int64_t N,
                                      const int64_t *__restrict__ d_n, int box, LocCols cols, const int64_t *__restrict__ d_row0,
                                      const int32_t *__restrict__ list)
{
    int64_t n = N;
    if (d_n) { int64_t dn = *d_n; n = dn < n ? dn : n; }
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int64_t src = list ? (int64_t)list[r] : r;                          // row of the fit arrays (list: the accepted candidates)
    const int64_t i = r + (d_row0 ? *d_row0 : 0);                              // row of the table: a second frame range follows the first
    const int off = box / 2;
    const float *t = th + src * 6, *c = cr + src * 6;
    // float32 theta + int64 coordinate -> float64 in pandas, then - offset, then cast
    ((uint32_t *)cols.c[0])[i] = (uint32_t)frame[src];
    ((float *)cols.c[1])[i] = (float)((double)t[0] + (double)x[src] - (double)off);
    ((float *)cols.c[2])[i] = (float)((double)t[1] + (double)y[src] - (double)off);
    ((float *)cols.c[3])[i] = t[2];
    ((float *)cols.c[4])[i] = t[4];
    ((float *)cols.c[5])[i] = t[5];
    ((float *)cols.c[6])[i] = t[3];
    ((float *)cols.c[7])[i] = sqrtf(c[0]);
    ((float *)cols.c[8])[i] = sqrtf(c[1]);
    float a = np_maxf(t[4], t[5]), b = np_minf(t[4], t[5]);
    ((float *)cols.c[9])[i] = (a - b) / a;
    ((float *)cols.c[10])[i] = ng[src];
    ((float *)cols.c[11])[i] = ll[src];
    ((uint32_t *)cols.c[12])[i] = (uint32_t)it[src];
    ((float *)cols.c[13])[i] = sqrtf(c[2]);
    ((float *)cols.c[14])[i] = sqrtf(c[3]);
    ((float *)cols.c[15])[i] = sqrtf(c[4]);
    ((float *)cols.c[16])[i] = sqrtf(c[5]);
}

int identify_impl(const void *d_movie, int dtype, int64_t F, int64_t Y, int64_t X, int box, double min_ng,
                  const int64_t *roi4, int64_t f_lo, int64_t f_hi, int64_t label_offset,
                  int32_t *d_frame, int32_t *d_y, int32_t *d_x, float *d_ng, int64_t cap, int64_t *d_out_n,
                  hipStream_t s);

}  // namespace pmi

extern "C" {

int pmi_mle_set_mode(int mode, double margin)
{
    using namespace pmi;
    if (mode != PMI_MLE_FAST && mode != PMI_MLE_REFIT && mode != PMI_MLE_STRICT) { set_error("unknown MLE mode %d", mode); return PMI_ERR_ARG; }
    if (!(margin >= 0.0 && margin < 1.0)) { set_error("margin must lie in [0, 1)"); return PMI_ERR_ARG; }
    g_mle_mode = mode;
    g_mle_margin = margin;
    return PMI_OK;
}

int pmi_mle_set_libm(int which)
{
    using namespace pmi;
    if (which != PMI_LIBM_DEVICE && which != PMI_LIBM_GLIBC && which != PMI_LIBM_AUTO) { set_error("unknown libm %d", which); return PMI_ERR_ARG; }
    g_mle_libm = which;
    return PMI_OK;
}

int pmi_mle_get_libm(int *which)
{
    if (which) *which = pmi::mle_libm_now();
    return PMI_OK;
}

int pmi_mle_get_mode(int *mode, double *margin)
{
    if (mode) *mode = pmi::mle_mode_now();
    if (margin) *margin = pmi::g_mle_margin;
    return PMI_OK;
}

int pmi_mle_last_refit_count(int64_t *n_refit, void *stream)
{
    using namespace pmi;
    unsigned h[16];
    int rc = read_last_stats(h, (hipStream_t)stream);
    if (rc != PMI_OK) return rc;
    if (n_refit) *n_refit = h[0];
    return PMI_OK;
}

int pmi_mle_last_flag_reasons(int64_t *counts, int n, void *stream)
{
    using namespace pmi;
    unsigned h[16];
    int rc = read_last_stats(h, (hipStream_t)stream);
    if (rc != PMI_OK) return rc;
    for (int i = 0; counts && i < n; i++) counts[i] = i < FLAG_REASONS ? h[1 + i] : 0;
    return PMI_OK;
}

int pmi_gaussmle_dev(const float *d_spots, int64_t N, const int64_t *d_n, int box, double eps, int max_it,
                     int method, float *d_thetas, float *d_crlbs, float *d_loglik, int32_t *d_iterations,
                     void *stream)
{
    pmi::FitParams p = {};
    p.spots = d_spots; p.N = N; p.d_n = d_n; p.box = box; p.eps = eps; p.max_it = max_it;
    p.thetas = d_thetas; p.crlbs = d_crlbs; p.loglik = d_loglik; p.iterations = d_iterations;
    return pmi::fit_impl(p, method, false, (hipStream_t)stream);
}

int pmi_gaussmle_movie_dev(const void *d_movie, int dtype, int64_t F, int64_t Y, int64_t X,
                           const int32_t *d_frame, const int32_t *d_y, const int32_t *d_x, int64_t N,
                           const int64_t *d_n, int box, double baseline, double sensitivity, double gain,
                           double eps, int max_it, int method, float *d_thetas, float *d_crlbs,
                           float *d_loglik, int32_t *d_iterations, void *stream)
{
    (void)F;
    if (dtype < 0 || dtype > PMI_F32) { pmi::set_error("unknown dtype code %d", dtype); return PMI_ERR_ARG; }
    pmi::FitParams p = {};
    p.movie = d_movie; p.dtype = dtype; p.Y = Y; p.X = X; p.frame = d_frame; p.y = d_y; p.x = d_x;
    p.baseline = (float)baseline; p.sensitivity = (float)sensitivity; p.gain = (float)gain; p.gdiv = pmi::make_const_div((float)gain);
    p.N = N; p.d_n = d_n; p.box = box; p.eps = eps; p.max_it = max_it;
    p.thetas = d_thetas; p.crlbs = d_crlbs; p.loglik = d_loglik; p.iterations = d_iterations;
    // a fused call whose scan handed the pixels on (these identifications, this thread, just now)
    if (pmi::g_handoff.used && pmi::g_handoff.d_slot && dtype == PMI_U16) { p.pix = pmi::g_handoff.pix; p.slot = pmi::g_handoff.d_slot; }
    return pmi::fit_impl(p, method, true, (hipStream_t)stream);
}

int pmi_gaussmle(const float *spots, int64_t N, int box, double eps, int max_it, int method, float *thetas,
                 float *crlbs, float *loglik, int32_t *iterations)
{
    using namespace pmi;
    if (pmi_device_count() < 1) { set_error("no HIP device"); return PMI_ERR_NODEVICE; }
    if (method != PMI_MLE_SIGMA && method != PMI_MLE_SIGMAXY) { set_error("Method not available."); return PMI_ERR_ARG; }
    if (N == 0) return PMI_OK;
    if (!spots || !thetas || !crlbs || !loglik || !iterations) { set_error("null pointer"); return PMI_ERR_ARG; }
    void *d_in = nullptr, *d_out = nullptr;
    int rc;
    size_t in_bytes = (size_t)N * box * box * sizeof(float);
    if ((rc = scratch(SCR_STAGE_A, in_bytes, &d_in)) != PMI_OK) return rc;
    if ((rc = scratch(SCR_STAGE_B, (size_t)N * 14 * 4, &d_out)) != PMI_OK) return rc;
    float *d_th = (float *)d_out, *d_cr = d_th + N * 6, *d_ll = d_cr + N * 6;
    int32_t *d_it = (int32_t *)(d_ll + N);
    PMI_HIP(hipMemcpy(d_in, spots, in_bytes, hipMemcpyHostToDevice));
    rc = pmi_gaussmle_dev((const float *)d_in, N, nullptr, box, eps, max_it, method, d_th, d_cr, d_ll, d_it, nullptr);
    if (rc != PMI_OK) return rc;
    PMI_HIP(hipMemcpy(thetas, d_th, (size_t)N * 24, hipMemcpyDeviceToHost));
    PMI_HIP(hipMemcpy(crlbs, d_cr, (size_t)N * 24, hipMemcpyDeviceToHost));
    PMI_HIP(hipMemcpy(loglik, d_ll, (size_t)N * 4, hipMemcpyDeviceToHost));
    PMI_HIP(hipMemcpy(iterations, d_it, (size_t)N * 4, hipMemcpyDeviceToHost));
    return PMI_OK;
}

int pmi_get_spots_dev(const void *d_movie, int dtype, int64_t F, int64_t Y, int64_t X, const int32_t *d_frame,
                      const int32_t *d_y, const int32_t *d_x, int64_t N, const int64_t *d_n, int box,
                      double baseline, double sensitivity, double gain, float *d_spots, void *stream)
{
    (void)F;
    using namespace pmi;
    if (box < 1 || box > PMI_MAX_BOX) { set_error("bad box %d", box); return PMI_ERR_ARG; }
    if (dtype < 0 || dtype > PMI_F32) { set_error("unknown dtype code %d", dtype); return PMI_ERR_ARG; }
    if (N <= 0) return PMI_OK;
    int64_t total = N * box * box;
    unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 64);
    hipLaunchKernelGGL(cut_spots_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_movie, dtype, Y, X,
                       d_frame, d_y, d_x, N, d_n, box, (float)baseline, (float)sensitivity, make_const_div((float)gain), d_spots);
    PMI_HIP(hipGetLastError());
    return PMI_OK;
}

static size_t px_size(int dtype) { return dtype == PMI_U8 ? 1 : ((dtype == PMI_U16 || dtype == PMI_I16) ? 2 : 4); }

// Host form: uploads only the frames the identifications touch, one chunk at a time.
int pmi_get_spots(const void *movie, int dtype, int64_t F, int64_t Y, int64_t X, const int32_t *frame,
                  const int32_t *y, const int32_t *x, int64_t N, int box, double baseline, double sensitivity,
                  double gain, float *out_spots)
{
    using namespace pmi;
    if (pmi_device_count() < 1) { set_error("no HIP device"); return PMI_ERR_NODEVICE; }
    if (N == 0) return PMI_OK;
    if (!movie || !frame || !y || !x || !out_spots) { set_error("null pointer"); return PMI_ERR_ARG; }
    if (dtype < 0 || dtype > PMI_F32) { set_error("unknown dtype code %d", dtype); return PMI_ERR_ARG; }
    const int r = box / 2;
    for (int64_t i = 0; i < N; i++)
        if (frame[i] < 0 || frame[i] >= F || y[i] - r < 0 || y[i] + r >= Y || x[i] - r < 0 || x[i] + r >= X) {
            set_error("identification %lld lies outside the movie", (long long)i);
            return PMI_ERR_ARG;
        }
    const size_t frame_bytes = (size_t)Y * X * px_size(dtype);
    const int64_t chunk = std::max<int64_t>(1, (int64_t)((size_t)1 << 30) / (int64_t)frame_bytes);
    void *d_chunk = nullptr, *d_ids = nullptr, *d_sp = nullptr;
    int rc;
    if ((rc = scratch(SCR_STAGE_A, (size_t)std::min<int64_t>(chunk, F) * frame_bytes, &d_chunk)) != PMI_OK) return rc;
    if ((rc = scratch(SCR_STAGE_B, (size_t)N * 12, &d_ids)) != PMI_OK) return rc;
    if ((rc = scratch(SCR_STAGE_C, (size_t)N * box * box * 4, &d_sp)) != PMI_OK) return rc;
    int32_t *d_f = (int32_t *)d_ids, *d_y = d_f + N, *d_x = d_y + N;
    // identifications are normally frame-sorted; process maximal runs that fit one chunk
    std::vector<int32_t> rel(N);
    int64_t i0 = 0;
    while (i0 < N) {
        int64_t fmin = frame[i0], fmax = frame[i0], i1 = i0 + 1;
        while (i1 < N) {
            int64_t lo = std::min<int64_t>(fmin, frame[i1]), hi = std::max<int64_t>(fmax, frame[i1]);
            if (hi - lo + 1 > chunk) break;
            fmin = lo; fmax = hi; i1++;
        }
        int64_t cnt = i1 - i0, nfc = fmax - fmin + 1;
        for (int64_t i = i0; i < i1; i++) rel[i - i0] = (int32_t)(frame[i] - fmin);
        PMI_HIP(hipMemcpy(d_chunk, (const char *)movie + (size_t)fmin * frame_bytes, (size_t)nfc * frame_bytes, hipMemcpyHostToDevice));
        PMI_HIP(hipMemcpy(d_f, rel.data(), (size_t)cnt * 4, hipMemcpyHostToDevice));
        PMI_HIP(hipMemcpy(d_y, y + i0, (size_t)cnt * 4, hipMemcpyHostToDevice));
        PMI_HIP(hipMemcpy(d_x, x + i0, (size_t)cnt * 4, hipMemcpyHostToDevice));
        rc = pmi_get_spots_dev(d_chunk, dtype, nfc, Y, X, d_f, d_y, d_x, cnt, nullptr, box, baseline, sensitivity, gain, (float *)d_sp, nullptr);
        if (rc != PMI_OK) return rc;
        PMI_HIP(hipMemcpy(out_spots + (size_t)i0 * box * box, d_sp, (size_t)cnt * box * box * 4, hipMemcpyDeviceToHost));
        i0 = i1;
    }
    return PMI_OK;
}

int pmi_locs_from_fits_dev(const int32_t *d_frame, const int32_t *d_y, const int32_t *d_x, const float *d_ng,
                           const float *d_thetas, const float *d_crlbs, const float *d_loglik,
                           const int32_t *d_iterations, int64_t N, const int64_t *d_n, int box,
                           void *const *d_cols, void *stream)
{
    using namespace pmi;
    if (N <= 0) return PMI_OK;
    LocCols cols;
    for (int c = 0; c < PMI_LOC_COLUMNS; c++) cols.c[c] = d_cols[c];
    unsigned blocks = (unsigned)((N + 255) / 256);
    hipLaunchKernelGGL(locs_from_fits_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_frame, d_y, d_x,
                       d_ng, d_thetas, d_crlbs, d_loglik, d_iterations, N, d_n, box, cols, (const int64_t *)nullptr, (const int32_t *)nullptr);
    PMI_HIP(hipGetLastError());
    return PMI_OK;
}

// ---- two frame ranges in flight ---------------------------------------------------------------------------------
// The scan is bound by memory requests, the fit by VALU issue: the scan of frame range B beside the fit of range A
// takes less than the two one after the other.  pmi_localize_mle_dev therefore cuts its frames
// in two: stream s runs scan A, fit A; a side stream of the library runs scan B (started when scan A is done) and
// fit B, with scratch from the inner bank; the table rows are written once both counts are known (A's rows, then
// B's), so the capacity contract holds for the sum.
//
// Deferred exact stage (default, pmi_localize_set_defer): on uint16 / uint8 / int16 movies and boxes up to 7x7
// the packed scan only emits CANDIDATES (window maximum, floor, neighbour rule) and the fit's start-value kernel, which
// reads a candidate's rows anyway, computes the float32 net gradient in the reference's order, applies the first-argmax
// rule and the threshold (gaussmle_g8.hip) — the scan no longer re-reads nine lines per candidate and never ends a chunk
// for a round.  The rejected candidates (one in seven on a DNA-PAINT movie) are skipped by the later stages and the table
// is compacted; the identification and fit arrays hold capc = cap + cap / 2 + 4096 candidates.  More candidates than
// that: *d_out_n reports their number (an upper bound of the rows), nothing is fitted, the table is untouched — the same
// contract as a table that is too small.
namespace pmi {
static bool g_localize_handoff = false;
static bool g_localize_defer = true;

// rows: [0] rows of A to fit, [1] rows of B to fit, [2] rows of A for the table, [3] rows of B for the table, [4] row offset of B
__global__ void range_rows_a_kernel(const int64_t *__restrict__ n_a, int64_t cap, int64_t *__restrict__ rows)
{
    rows[0] = *n_a > cap ? 0 : *n_a;
}
__global__ void range_rows_b_kernel(const int64_t *__restrict__ n_a, const int64_t *__restrict__ n_b, int64_t cap,
                                    int64_t *__restrict__ rows, int64_t *__restrict__ d_out_n)
{
    const int64_t a = *n_a, b = *n_b, total = a + b;
    const bool fits = total <= cap;
    rows[1] = fits ? b : 0;
    rows[2] = fits ? a : 0;
    rows[3] = fits ? b : 0;
    rows[4] = a;
    *d_out_n = total;
}
// deferred exact stage: the candidates of a range that may be fitted (all of them, or none when they overflow capc)
__global__ void cand_rows_kernel(const int64_t *__restrict__ n_cand, int64_t capc, int64_t *__restrict__ rows_fit)
{
    *rows_fit = *n_cand > capc ? 0 : *n_cand;
}
// ... and the table rows once the accepted candidates are counted (acc_b == nullptr: one range)
__global__ void accepted_rows_kernel(const int64_t *__restrict__ cand_a, const unsigned *__restrict__ acc_a,
                                     const int64_t *__restrict__ cand_b, const unsigned *__restrict__ acc_b, int64_t capc,
                                     int64_t cap, int64_t *__restrict__ rows, int64_t *__restrict__ d_out_n)
{
    const int64_t ca = *cand_a, cb = cand_b ? *cand_b : 0;
    const bool overflow = ca > capc || cb > capc;
    const int64_t a = overflow ? 0 : (int64_t)*acc_a, b = (overflow || !acc_b) ? 0 : (int64_t)*acc_b;
    const bool fits = !overflow && a + b <= cap;
    rows[2] = fits ? a : 0;
    rows[3] = fits ? b : 0;
    rows[4] = a;
    *d_out_n = overflow ? ca + cb : a + b;
}
}  // namespace pmi

int pmi_localize_set_handoff(int on)
{
    pmi::g_localize_handoff = on != 0;
    return PMI_OK;
}

int pmi_localize_set_defer(int on)
{
    pmi::g_localize_defer = on != 0;
    return PMI_OK;
}

int pmi_localize_set_ranges(int ranges)
{
    if (ranges != 1 && ranges != 2) { pmi::set_error("ranges in flight: 1 or 2"); return PMI_ERR_ARG; }
    pmi::g_localize_ranges = ranges;
    return PMI_OK;
}

int pmi_localize_mle_dev(const void *d_movie, int dtype, int64_t F, int64_t Y, int64_t X, int box, double min_ng,
                         const int64_t *roi4, int64_t f_lo, int64_t f_hi, double baseline, double sensitivity,
                         double gain, double eps, int max_it, int method, void *d_table, int64_t cap,
                         int64_t *d_out_n, void *stream)
{
    using namespace pmi;
    if (cap <= 0) { set_error("capacity must be positive"); return PMI_ERR_ARG; }
    if (dtype < 0 || dtype > PMI_F32) { set_error("unknown dtype code %d", dtype); return PMI_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    void *cols[PMI_LOC_COLUMNS];
    for (int c = 0; c < PMI_LOC_COLUMNS; c++) cols[c] = (char *)d_table + (size_t)c * cap * 4;
    // pixel hand-off from the scan's exact stage to the fit (uint16 movies, boxes of the row-per-lane fit): room for twice
    // the table's rows (candidates that fail the threshold take a slot too), spread over the eight record shards
    const bool hand = g_localize_handoff && dtype == PMI_U16 && box <= 15 && mle_mode_now() != PMI_MLE_STRICT;
    // the exact stage of identify in the fit's start-value kernel instead of the scan (see above)
    // (boxes up to 7: eight lanes per candidate in the start-value kernel.  With the 16-lane groups of boxes 9 ... 15 the stage
    // costs the fit more than it saves the scan — config 5, box 13: scan 7.0 -> 6.0 ms, fit 19.9 -> 22.0 ms — so those boxes
    // keep it in the scan, which is bound by instruction issue there, not by its memory requests)
    // (a tuning build's PMI_FIT_WAVE_PER_SPOT / PMI_MLE_NO_KEEP take the fit off the row-per-lane path that hosts the stage:
    // the stage then stays in the scan — the predicate fit_impl checks)
    static const bool off_g8 = tuning_env("PMI_FIT_WAVE_PER_SPOT") != nullptr || tuning_env("PMI_MLE_NO_KEEP") != nullptr;
    const bool defer = g_localize_defer && !hand && !off_g8 && box <= 7 && mle_mode_now() != PMI_MLE_STRICT &&
                       (dtype == PMI_U16 || dtype == PMI_U8 || dtype == PMI_I16) && min_ng > 0.0 && std::isfinite(min_ng);
    const int64_t capc = defer ? cap + cap / 2 + 4096 : cap;         // rows of the identification / fit arrays
    int64_t cy0 = 0, cx0 = 0, cy1 = Y, cx1 = X;
    if (roi4) { cy0 = roi4[0]; cx0 = roi4[1]; cy1 = roi4[2]; cx1 = roi4[3]; }
    struct Ids { int32_t *f, *y, *x; float *ng, *th, *cr, *ll; int32_t *it, *slot, *tlist; unsigned *blk; unsigned char *acc; };
    const size_t acc_blocks = (size_t)((capc + ACC_BLOCK - 1) / ACC_BLOCK) + 2;
    auto carve = [&](void *ptr) {
        Ids d;
        d.f = (int32_t *)ptr; d.y = d.f + capc; d.x = d.y + capc;
        d.ng = (float *)(d.x + capc);
        d.th = d.ng + capc; d.cr = d.th + capc * 6; d.ll = d.cr + capc * 6;
        d.it = (int32_t *)(d.ll + capc);
        d.slot = d.it + capc;
        d.tlist = d.slot + capc;
        d.blk = (unsigned *)(d.tlist + capc);
        d.acc = (unsigned char *)(d.blk + acc_blocks);
        return d;
    };
    const size_t ids_bytes = (size_t)capc * (16 + 16 * 4 + 1) + acc_blocks * sizeof(unsigned) + 16;
    const unsigned pix_cap = (unsigned)std::min<int64_t>(std::max<int64_t>(cap / 4, 4096), 0x0fffffff);
    const size_t pix_bytes = (size_t)8 * pix_cap * (size_t)(box * (box / 2 + 1)) * sizeof(uint32_t);
    auto arm_handoff = [&](int32_t *d_slot) -> int {         // in the scratch bank that is current
        g_handoff = PixHandoff();
        if (!hand) return PMI_OK;
        void *pp = nullptr;
        int r = scratch(SCR_PIX, pix_bytes, &pp);
        if (r != PMI_OK) return r;
        g_handoff.pix = (uint32_t *)pp; g_handoff.cap_per_shard = pix_cap; g_handoff.d_slot = d_slot;
        return PMI_OK;
    };
    // identify (candidates when deferred) over frames [lo_, hi_] into d, count at d_cnt
    auto scan_range = [&](const Ids &d, int64_t lo_, int64_t hi_, int64_t *d_cnt, hipStream_t st) -> int {
        g_defer_exact = defer;
        const int r = identify_impl(d_movie, dtype, F, Y, X, box, min_ng, roi4, lo_, hi_, 0, d.f, d.y, d.x, d.ng, capc, d_cnt, st);
        g_defer_exact = false;
        return r;
    };
    // the fit of the rows *d_rows of d (with the exact stage of identify in its start-value kernel when deferred)
    auto fit_range = [&](const Ids &d, const int64_t *d_rows, hipStream_t st) -> int {
        FitParams p = {};
        p.movie = d_movie; p.dtype = dtype; p.Y = Y; p.X = X; p.frame = d.f; p.y = d.y; p.x = d.x;
        p.baseline = (float)baseline; p.sensitivity = (float)sensitivity; p.gain = (float)gain; p.gdiv = make_const_div((float)gain);
        p.N = capc; p.d_n = d_rows; p.box = box; p.eps = eps; p.max_it = max_it;
        p.thetas = d.th; p.crlbs = d.cr; p.loglik = d.ll; p.iterations = d.it;
        if (g_handoff.used && g_handoff.d_slot && dtype == PMI_U16) { p.pix = g_handoff.pix; p.slot = g_handoff.d_slot; }
        if (defer) {
            p.ng_io = d.ng; p.accept = d.acc; p.min_ng = min_ng;
            p.crop_y0 = (int)cy0; p.crop_x0 = (int)cx0; p.crop_cy = (int)(cy1 - cy0); p.crop_cx = (int)(cx1 - cx0);
            p.alist_out = d.tlist; p.alist_blk_out = d.blk;
        }
        int r = fit_impl(p, method, true, st);
        if (r != PMI_OK || !defer) return r;
        // the table's source rows (total at blk[nblk]): the list the fit made for its only batch, or one pass over all flags
        if (capc <= FIT_BATCH) return PMI_OK;
        return build_accept_list(d.acc, 0, capc, d_rows, d.blk, d.tlist, st);
    };
    const unsigned *const no_acc = nullptr;
    const int nblk_c = (int)((capc + ACC_BLOCK - 1) / ACC_BLOCK);
    LocCols lc;
    for (int c = 0; c < PMI_LOC_COLUMNS; c++) lc.c[c] = cols[c];
    const unsigned tblocks = (unsigned)((cap + 255) / 256);

    const int64_t lo = f_lo < 0 ? 0 : f_lo, hi = f_hi > F - 1 ? F - 1 : f_hi, nf = hi - lo + 1;
    // two ranges pay when each keeps the chip busy for a while (below ~1e8 pixels a range is a few tens of microseconds)
    const bool two = g_localize_ranges == 2 && !g_kernel_timing && nf >= 16 && (double)nf * (double)Y * (double)X >= 2.5e8;
    void *ptr = nullptr, *cptr = nullptr;
    int rc;
    if ((rc = scratch(SCR_ROWS, 8 * sizeof(int64_t), &cptr)) != PMI_OK) return rc;
    int64_t *d_na = (int64_t *)cptr, *d_nb = d_na + 1, *rows = d_na + 2;
    if (!two) {
        if ((rc = scratch(SCR_IDS, ids_bytes, &ptr)) != PMI_OK) return rc;
        const Ids d = carve(ptr);
        if ((rc = arm_handoff(d.slot)) != PMI_OK) return rc;
        int64_t *d_cnt = defer ? d_na : d_out_n;
        rc = scan_range(d, f_lo, f_hi, d_cnt, s);
        if (rc != PMI_OK) { g_handoff = PixHandoff(); return rc; }
        hipLaunchKernelGGL(cand_rows_kernel, dim3(1), dim3(1), 0, s, (const int64_t *)d_cnt, capc, rows + 0);
        rc = fit_range(d, rows + 0, s);
        g_handoff = PixHandoff();
        if (rc != PMI_OK) return rc;
        if (defer) {
            hipLaunchKernelGGL(accepted_rows_kernel, dim3(1), dim3(1), 0, s, (const int64_t *)d_na, (const unsigned *)(d.blk + nblk_c),
                               (const int64_t *)nullptr, no_acc, capc, cap, rows, d_out_n);
            hipLaunchKernelGGL(locs_from_fits_kernel, dim3(tblocks), dim3(256), 0, s, d.f, d.y, d.x, d.ng, d.th, d.cr, d.ll, d.it, cap,
                               (const int64_t *)(rows + 2), box, lc, (const int64_t *)nullptr, (const int32_t *)d.tlist);
        } else {
            hipLaunchKernelGGL(locs_from_fits_kernel, dim3(tblocks), dim3(256), 0, s, d.f, d.y, d.x, d.ng, d.th, d.cr, d.ll, d.it, cap,
                               (const int64_t *)(rows + 0), box, lc, (const int64_t *)nullptr, (const int32_t *)nullptr);
        }
        PMI_HIP(hipGetLastError());
        return PMI_OK;
    }
    SideLane *side_p = nullptr;
    if ((rc = side_lane(0, &side_p)) != PMI_OK) return rc;
    SideLane &side = *side_p;
    const int64_t mid = lo + nf / 2 - 1;                      // A = [lo, mid], B = [mid + 1, hi]
    // counts of the two ranges and the row bookkeeping live in the OUTER bank (both streams read them)
    if ((rc = scratch(SCR_IDS, ids_bytes, &ptr)) != PMI_OK) return rc;
    const Ids a = carve(ptr);
    PMI_HIP(hipEventRecord(side.ev_start, s));               // the side stream joins the caller's stream order here
    PMI_HIP(hipStreamWaitEvent(side.s2, side.ev_start, 0));
    // From here on work may be queued on the side stream: whatever happens, the caller's stream is ordered after it
    // before this call returns (a caller that frees or reuses its buffers on an error must not race kernels on s2).
    struct Join {
        SideLane &sd; hipStream_t st; bool done = false;
        ~Join() { if (!done) { (void)hipEventRecord(sd.ev_b, sd.s2); (void)hipStreamWaitEvent(st, sd.ev_b, 0); } }
    } join{side, s};
    // ---- range A on the caller's stream
    if ((rc = arm_handoff(a.slot)) != PMI_OK) return rc;
    rc = scan_range(a, lo, mid, d_na, s);
    if (rc != PMI_OK) { g_handoff = PixHandoff(); return rc; }
    hipLaunchKernelGGL(range_rows_a_kernel, dim3(1), dim3(1), 0, s, (const int64_t *)d_na, capc, rows);
    PMI_HIP(hipEventRecord(side.ev_scan_a, s));
    rc = fit_range(a, rows + 0, s);
    g_handoff = PixHandoff();
    if (rc != PMI_OK) return rc;
    // ---- range B on the side stream, scratch from the inner bank; its scan starts when scan A is done
    PMI_HIP(hipStreamWaitEvent(side.s2, side.ev_scan_a, 0));
    const int outer = scratch_enter_inner();
    Ids b2 = {};
    rc = scratch(SCR_IDS, ids_bytes, &ptr);
    if (rc == PMI_OK) {
        b2 = carve(ptr);
        rc = arm_handoff(b2.slot);
    }
    if (rc == PMI_OK) rc = scan_range(b2, mid + 1, hi, d_nb, side.s2);
    if (rc == PMI_OK) {
        if (defer) hipLaunchKernelGGL(cand_rows_kernel, dim3(1), dim3(1), 0, side.s2, (const int64_t *)d_nb, capc, rows + 1);
        else hipLaunchKernelGGL(range_rows_b_kernel, dim3(1), dim3(1), 0, side.s2, (const int64_t *)d_na, (const int64_t *)d_nb, cap, rows, d_out_n);
        g_stats_second = true;
        rc = fit_range(b2, rows + 1, side.s2);
        g_stats_second = false;
    }
    g_handoff = PixHandoff();
    scratch_leave_inner(outer);
    if (rc != PMI_OK) return rc;
    PMI_HIP(hipEventRecord(side.ev_b, side.s2));
    PMI_HIP(hipStreamWaitEvent(s, side.ev_b, 0));
    join.done = true;
    // ---- the table, once both counts are known: A's rows, then B's
    if (defer)
        hipLaunchKernelGGL(accepted_rows_kernel, dim3(1), dim3(1), 0, s, (const int64_t *)d_na, (const unsigned *)(a.blk + nblk_c),
                           (const int64_t *)d_nb, (const unsigned *)(b2.blk + nblk_c), capc, cap, rows, d_out_n);
    hipLaunchKernelGGL(locs_from_fits_kernel, dim3(tblocks), dim3(256), 0, s, a.f, a.y, a.x, a.ng, a.th, a.cr, a.ll, a.it, cap,
                       (const int64_t *)(rows + 2), box, lc, (const int64_t *)nullptr, (const int32_t *)(defer ? a.tlist : nullptr));
    hipLaunchKernelGGL(locs_from_fits_kernel, dim3(tblocks), dim3(256), 0, s, b2.f, b2.y, b2.x, b2.ng, b2.th, b2.cr, b2.ll, b2.it, cap,
                       (const int64_t *)(rows + 3), box, lc, (const int64_t *)(rows + 4), (const int32_t *)(defer ? b2.tlist : nullptr));
    PMI_HIP(hipGetLastError());
    return PMI_OK;
}

}  // extern "C"
