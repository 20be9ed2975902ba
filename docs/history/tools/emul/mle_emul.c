/* TOOL (not product, not oracle): CPU study of how far the float32 Newton loop of the device kernels drifts from the
 * reference's arithmetic, iteration by iteration.  ref_trace = the oracle's loop (oracle/picasso_oracle.c mlefit_one)
 * recording theta after every iteration; fast_trace = a float32 restatement of csrc/gaussmle_g8.hip newton_step
 * (separable boundary terms, row-local sums, sums over rows) recording theta, num, den and the curvature part Q.
 * Build: gcc -O2 -fopenmp -ffp-contract=off -shared -fPIC -o tools/emul/_emul.so tools/emul/mle_emul.c -lm */
#include "../../oracle/picasso_oracle.c"

static void ref_trace_one(const float *spot, int size, int method, double eps, int max_it, int T, float *trace, int32_t *it_out)
{
    const int np_ = method == ORC_SIGMAXY ? 6 : 5;
    float theta[6], init[6];
    double sxy[2];
    initial_parameters_d(spot, size, init, sxy);
    theta[0] = init[0]; theta[1] = init[1]; theta[2] = init[2]; theta[3] = init[3];
    if (method == ORC_SIGMAXY) { theta[4] = init[4]; theta[5] = init[5]; }
    else { theta[4] = (float)((sxy[0] + sxy[1]) / 2); theta[5] = 0.0f; }
    float max_step[6];
    max_step[0] = theta[4]; max_step[1] = theta[4];
    max_step[2] = (float)(0.1 * (double)theta[2]);
    max_step[3] = (float)(0.1 * (double)theta[3]);
    max_step[4] = (float)(0.2 * (double)theta[4]);
    max_step[5] = (float)(0.2 * (double)theta[5]);
    for (int l = 0; l < 6; l++) trace[l] = theta[l];
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
                dudt[0] = (float)a; d2udt2[0] = (float)b;
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
                    d_gaussian_integral_iso_sigma(ii, jj, theta[0], theta[1], theta[4], theta[2], PSFx, PSFy, &a, &b);
                    dudt[4] = (float)a; d2udt2[4] = (float)b;
                }
                double model = (double)theta[2] * PSFx * PSFy + (double)theta[3];
                double cf = 0.0, df = 0.0;
                float data = spot[jj * size + ii];
                if (model > 10e-3) { cf = (double)data / model - 1; df = (double)data / (model * model); }
                cf = np_min(cf, 10e4);
                df = np_min(df, 10e4);
                for (int l = 0; l < np_; l++) {
                    float du2 = dudt[l] * dudt[l];
                    num[l] = (float)((double)num[l] + cf * (double)dudt[l]);
                    den[l] = (float)((double)den[l] + (cf * (double)d2udt2[l] - df * (double)du2));
                }
            }
        int conv;
        if (method == ORC_SIGMAXY) {
            for (int l = 0; l < 6; l++) {
                if (den[l] == 0.0f) theta[l] = theta[l] - np_signf(num[l]) * max_step[l];
                else theta[l] = theta[l] - np_minf(np_maxf(num[l] / den[l], -max_step[l]), max_step[l]);
            }
            theta[2] = (float)np_max((double)theta[2], 1.0);
            theta[3] = (float)np_max((double)theta[3], 0.01);
            theta[4] = (float)np_max((double)theta[4], 0.01);
            theta[5] = (float)np_max((double)theta[5], 0.01);
            conv = ((double)fabsf(old_x - theta[0]) < eps) && ((double)fabsf(old_y - theta[1]) < eps)
                   && ((double)fabsf(old_sx - theta[4]) < eps) && ((double)fabsf(old_sy - theta[5]) < eps);
        } else {
            for (int l = 0; l < 5; l++) {
                float update;
                if (den[l] == 0.0f) update = np_signf(num[l] * max_step[l]);
                else update = np_minf(np_maxf(num[l] / den[l], -max_step[l]), max_step[l]);
                theta[l] = theta[l] - update;
            }
            theta[2] = (float)np_max((double)theta[2], 1.0);
            theta[3] = (float)np_max((double)theta[3], 0.01);
            theta[4] = (float)np_max((double)theta[4], 0.01);
            theta[4] = (float)np_min((double)theta[4], (double)size);
            conv = ((double)fabsf(old_x - theta[0]) < eps) && ((double)fabsf(old_y - theta[1]) < eps);
        }
        if (kk < T) for (int l = 0; l < 6; l++) trace[kk * 6 + l] = theta[l];
        if (conv) break;
        old_x = theta[0]; old_y = theta[1]; old_sx = theta[4]; old_sy = theta[5];
    }
    *it_out = kk;
}

/* ---- float32 restatement of newton_step -------------------------------------------------- */
static float erf_f32(float a)
{
    const float t = fabsf(a), s = a * a;
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    const float big = copysignf(1.0f - expf(r), a);
    float q = -5.96761703e-4f;
    q = fmaf(q, s, 4.99119423e-3f);
    q = fmaf(q, s, -2.67681349e-2f);
    q = fmaf(q, s, 1.12819925e-1f);
    q = fmaf(q, s, -3.76125336e-1f);
    q = fmaf(q, s, 1.28379166e-1f);
    const float small = fmaf(q, a, a);
    return t > 0.921875f ? big : small;
}
typedef struct { float E, A, A2, S, S2; } BT;
static void bterms(int B, float mu, float sigma, BT *t)
{
    const float is = 1.0f / sigma;
    const float sn = 0.70710678118654757f * is, c1 = 0.3989422804014327f * is, is2 = is * is;
    float e[ORC_MAX_BOX + 2], g[ORC_MAX_BOX + 2], u[ORC_MAX_BOX + 2];
    for (int j = 0; j <= B; j++) {
        u[j] = (float)j - 0.5f - mu;
        e[j] = erf_f32(u[j] * sn);
        g[j] = expf(-0.5f * u[j] * u[j] * is2);
    }
    for (int j = 0; j < B; j++) {
        const float u0 = u[j], u1 = u0 + 1.0f, g0 = g[j], g1 = g[j + 1];
        const float q1 = u0 * g0 - u1 * g1;
        const float q3 = u0 * u0 * u0 * g0 - u1 * u1 * u1 * g1;
        t[j].E = 0.5f * (e[j + 1] - e[j]);
        t[j].A = (g0 - g1) * c1;
        t[j].A2 = q1 * c1 * is2;
        t[j].S = q1 * c1 * is;
        t[j].S2 = c1 * is2 * (q3 * is2 - 2.0f * q1);
    }
}
static float clip_np(float q, float lim) { const float c = fminf(fmaxf(q, -lim), lim); return (q != q) ? q : c; }

/* trace: theta per iteration; aux: per iteration AUXN floats: num[6], den[6], Q[6] (the df du^2 part of den), An[6] = sum |cf du|,
 * Ad[6] = sum |cf d2| + |df| du^2, top = max(|cf|, |df|) */
#define AUXN 31
static void fast_trace_one(const float *spot, int B, int method, double eps, int max_it, int T, float *trace, float *aux, int32_t *it_out)
{
    const int NP = method == ORC_SIGMAXY ? 6 : 5;
    float th[6], init[6], ms[6];
    double sxy[2];
    initial_parameters_d(spot, B, init, sxy);
    th[0] = init[0]; th[1] = init[1]; th[2] = init[2]; th[3] = init[3];
    if (NP == 6) { th[4] = init[4]; th[5] = init[5]; }
    else { th[4] = (float)((sxy[0] + sxy[1]) / 2); th[5] = 0.0f; }
    ms[0] = th[4]; ms[1] = th[4];
    ms[2] = (float)(0.1 * (double)th[2]); ms[3] = (float)(0.1 * (double)th[3]);
    ms[4] = (float)(0.2 * (double)th[4]); ms[5] = (float)(0.2 * (double)th[5]);
    for (int l = 0; l < 6; l++) trace[l] = th[l];
    const float floor_[6] = {-INFINITY, -INFINITY, 1.0f, 0.01f, 0.01f, NP == 6 ? 0.01f : -INFINITY};
    int kk = 0;
    while (kk < max_it) {
        BT tx[ORC_MAX_BOX], ty[ORC_MAX_BOX];
        const float sgy = NP == 6 ? th[5] : th[4];
        bterms(B, th[0], th[4], tx);
        bterms(B, th[1], sgy, ty);
        const float N_ = th[2], bg = th[3];
        float num[6] = {0, 0, 0, 0, 0, 0}, den[6] = {0, 0, 0, 0, 0, 0}, Q[6] = {0, 0, 0, 0, 0, 0};
        double An[6] = {0, 0, 0, 0, 0, 0}, Ad[6] = {0, 0, 0, 0, 0, 0}; float top = 0.f;
        for (int j = 0; j < B; j++) {           /* row j = one lane */
            const float NEy = N_ * ty[j].E;
            float a_cA = 0, a_cE = 0, a_cS = 0, a_cA2 = 0, a_cS2 = 0, a_c = 0, a_dA = 0, a_dE = 0, a_dS = 0, a_d = 0, a_dSE = 0;
            for (int i = 0; i < B; i++) {
                const float model = fmaf(NEy, tx[i].E, bg);
                const float r = 1.0f / model;
                const float d = spot[j * B + i];
                float cf = 0.f, df = 0.f;
                if (model > 10e-3f) { cf = fminf(fmaf(d, r, -1.f), 10e4f); df = fminf((d * r) * r, 10e4f); }
                a_cA = fmaf(cf, tx[i].A, a_cA); a_cE = fmaf(cf, tx[i].E, a_cE); a_cS = fmaf(cf, tx[i].S, a_cS);
                a_cA2 = fmaf(cf, tx[i].A2, a_cA2); a_cS2 = fmaf(cf, tx[i].S2, a_cS2); a_c = fmaf(cf, 1.0f, a_c);
                a_dA = fmaf(df, tx[i].A * tx[i].A, a_dA); a_dE = fmaf(df, tx[i].E * tx[i].E, a_dE);
                a_dS = fmaf(df, tx[i].S * tx[i].S, a_dS); a_d = fmaf(df, 1.0f, a_d);
                a_dSE = fmaf(df, tx[i].S * tx[i].E, a_dSE);
                {
                    top = fmaxf(top, fmaxf(fabsf(cf), fabsf(df)));
                    const double NA_y = N_ * ty[j].A, NS_y = N_ * ty[j].S;
                    double du[6], d2[6];
                    du[0] = NEy * tx[i].A; d2[0] = NEy * tx[i].A2;
                    du[1] = NA_y * tx[i].E; d2[1] = N_ * ty[j].A2 * tx[i].E;
                    du[2] = ty[j].E * tx[i].E; d2[2] = 0; du[3] = 1; d2[3] = 0;
                    if (NP == 6) { du[4] = NEy * tx[i].S; d2[4] = NEy * tx[i].S2; du[5] = NS_y * tx[i].E; d2[5] = N_ * ty[j].S2 * tx[i].E; }
                    else { du[4] = NEy * tx[i].S + NS_y * tx[i].E; d2[4] = NEy * tx[i].S2 + 2.0 * ty[j].S * tx[i].S + ty[j].S2 * tx[i].E; du[5] = 0; d2[5] = 0; }
                    for (int l = 0; l < 6; l++) { An[l] += fabs(cf * du[l]); Ad[l] += fabs(cf * d2[l]) + fabs(df) * du[l] * du[l]; }
                }
            }
            const float NAy = N_ * ty[j].A, NA2y = N_ * ty[j].A2, NSy = N_ * ty[j].S, NS2y = N_ * ty[j].S2;
            float n_[6], d_[6], q_[6];
            n_[0] = NEy * a_cA;   d_[0] = NEy * a_cA2 - NEy * NEy * a_dA;   q_[0] = NEy * NEy * a_dA;
            n_[1] = NAy * a_cE;   d_[1] = NA2y * a_cE - NAy * NAy * a_dE;   q_[1] = NAy * NAy * a_dE;
            n_[2] = ty[j].E * a_cE; d_[2] = -ty[j].E * ty[j].E * a_dE;      q_[2] = ty[j].E * ty[j].E * a_dE;
            n_[3] = a_c;          d_[3] = -a_d;                             q_[3] = a_d;
            if (NP == 6) {
                n_[4] = NEy * a_cS; d_[4] = NEy * a_cS2 - NEy * NEy * a_dS; q_[4] = NEy * NEy * a_dS;
                n_[5] = NSy * a_cE; d_[5] = NS2y * a_cE - NSy * NSy * a_dE; q_[5] = NSy * NSy * a_dE;
            } else {
                n_[4] = NEy * a_cS + NSy * a_cE;
                q_[4] = (NEy * NEy * a_dS + 2.f * NEy * NSy * a_dSE + NSy * NSy * a_dE);
                d_[4] = (NEy * a_cS2 + 2.f * ty[j].S * a_cS + ty[j].S2 * a_cE) - q_[4];
                n_[5] = 0; d_[5] = 0; q_[5] = 0;
            }
            for (int l = 0; l < 6; l++) { num[l] += n_[l]; den[l] += d_[l]; Q[l] += q_[l]; }
        }
        float nt[6];
        for (int l = 0; l < 6; l++) nt[l] = th[l];
        for (int l = 0; l < NP; l++) {
            const float stepz = NP == 6 ? np_signf(num[l]) * ms[l] : np_signf(num[l] * ms[l]);
            const float stepn = clip_np(num[l] * (1.0f / den[l]), ms[l]);
            float v = th[l] - (den[l] == 0.0f ? stepz : stepn);
            v = np_maxf(v, floor_[l]);
            if (NP == 5 && l == 4) v = np_minf(v, (float)B);
            nt[l] = v;
        }
        float D = fmaxf(fabsf(th[0] - nt[0]), fabsf(th[1] - nt[1]));
        int nan_ = (th[0] - nt[0]) != (th[0] - nt[0]) || (th[1] - nt[1]) != (th[1] - nt[1]);
        if (NP == 6) {
            D = fmaxf(D, fmaxf(fabsf(th[4] - nt[4]), fabsf(th[5] - nt[5])));
            nan_ = nan_ || (th[4] - nt[4]) != (th[4] - nt[4]) || (th[5] - nt[5]) != (th[5] - nt[5]);
        }
        const int conv = !nan_ && (double)D < eps;
        for (int l = 0; l < 6; l++) th[l] = nt[l];
        if (kk < T - 1) { for (int l = 0; l < 6; l++) { aux[kk * AUXN + l] = num[l]; aux[kk * AUXN + 6 + l] = den[l]; aux[kk * AUXN + 12 + l] = Q[l]; aux[kk * AUXN + 18 + l] = (float)An[l]; aux[kk * AUXN + 24 + l] = (float)Ad[l]; } aux[kk * AUXN + 30] = top; }
        kk++;
        if (kk < T) for (int l = 0; l < 6; l++) trace[kk * 6 + l] = th[l];
        if (conv) break;
    }
    *it_out = kk;
}

int emul_traces(const float *spots, int64_t N, int box, double eps, int max_it, int method, int T,
                float *trace_r, int32_t *it_r, float *trace_f, float *aux_f, int32_t *it_f, int nthreads)
{
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads)
    for (int64_t i = 0; i < N; i++) {
        ref_trace_one(spots + i * box * box, box, method, eps, max_it, T, trace_r + i * T * 6, it_r + i);
        fast_trace_one(spots + i * box * box, box, method, eps, max_it, T, trace_f + i * T * 6, aux_f + i * T * AUXN, it_f + i);
    }
    return 0;
}
