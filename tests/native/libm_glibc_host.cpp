// Host build of picasso_amd/csrc/libm_glibc.h beside the C library's exp / erf (tests/test_libm_glibc.py).
// TEST INFRASTRUCTURE: g++ -O2 -ffp-contract=off -std=c++17 -shared -fPIC
#include <math.h>
#include <stdint.h>

#include "libm_glibc.h"

static const uint64_t k_tab[PMI_GLIBC_EXP_TABLE_WORDS] = {
#include "libm_glibc_exp_table.inc"
};

extern "C" {
// number of arguments on which the bits differ (NaN results count as equal); first_bad: index of the first, or -1
int64_t cmp_exp(const double *x, int64_t n, int64_t *first_bad)
{
    int64_t bad = 0;
    *first_bad = -1;
    for (int64_t i = 0; i < n; i++) {
        const double a = pmi_glibc::exp(x[i], k_tab), b = ::exp(x[i]);
        if (pmi_glibc::to_bits(a) != pmi_glibc::to_bits(b) && !(a != a && b != b)) { if (!bad) *first_bad = i; bad++; }
    }
    return bad;
}
int64_t cmp_erf(const double *x, int64_t n, int64_t *first_bad)
{
    int64_t bad = 0;
    *first_bad = -1;
    for (int64_t i = 0; i < n; i++) {
        const double a = pmi_glibc::erf(x[i], k_tab), b = ::erf(x[i]);
        if (pmi_glibc::to_bits(a) != pmi_glibc::to_bits(b) && !(a != a && b != b)) { if (!bad) *first_bad = i; bad++; }
    }
    return bad;
}
// the C library's own values, for the device-side comparison (tests/test_gpu_parity.py)
// (parallel when built with -fopenmp: tools/libm_device_exhaustive.py)
void ref_exp(const double *x, int64_t n, double *out)
{
#pragma omp parallel for
    for (int64_t i = 0; i < n; i++) out[i] = ::exp(x[i]);
}
void ref_erf(const double *x, int64_t n, double *out)
{
#pragma omp parallel for
    for (int64_t i = 0; i < n; i++) out[i] = ::erf(x[i]);
}
double one_exp(double x) { return pmi_glibc::exp(x, k_tab); }
double one_erf(double x) { return pmi_glibc::erf(x, k_tab); }
}
