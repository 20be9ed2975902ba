// picasso_amd/csrc/libm_glibc.h against the C library on EVERY float32 argument (2^32 bit patterns, widened to float64) and on
// a stride of float64 patterns.  TEST INFRASTRUCTURE (tools/libm_glibc_exhaustive.sh):
//   g++ -O2 -fopenmp -ffp-contract=off -std=c++17 -I picasso_amd/csrc tests/native/libm_glibc_exhaustive.cpp -o /tmp/libm_exh && /tmp/libm_exh
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "libm_glibc.h"

static const uint64_t k_tab[PMI_GLIBC_EXP_TABLE_WORDS] = {
#include "libm_glibc_exp_table.inc"
};

static inline bool same(double a, double b) { return pmi_glibc::to_bits(a) == pmi_glibc::to_bits(b) || (a != a && b != b); }

int main()
{
    long long bad_exp = 0, bad_erf = 0;
#pragma omp parallel for reduction(+ : bad_exp, bad_erf) schedule(static)
    for (long long u = 0; u < (1LL << 32); u++) {
        const uint32_t bits = (uint32_t)u;
        float f;
        memcpy(&f, &bits, 4);
        const double x = (double)f;
        bad_exp += !same(pmi_glibc::exp(x, k_tab), ::exp(x));
        bad_erf += !same(pmi_glibc::erf(x, k_tab), ::erf(x));
    }
    printf("every float32 argument (4294967296): exp differs on %lld, erf on %lld\n", bad_exp, bad_erf);
    long long bad_exp64 = 0, bad_erf64 = 0, n64 = 0;
#pragma omp parallel for reduction(+ : bad_exp64, bad_erf64, n64) schedule(static)
    for (long long u = 0; u < (1LL << 32); u++) {
        // float64 patterns: the upper 32 bits run through every value, the lower 32 are a hash of them
        const uint64_t hi = (uint64_t)u, lo = (hi * 2654435761ull + 0x9e3779b9ull) & 0xffffffffull;
        const double x = pmi_glibc::from_bits((hi << 32) | lo);
        bad_exp64 += !same(pmi_glibc::exp(x, k_tab), ::exp(x));
        bad_erf64 += !same(pmi_glibc::erf(x, k_tab), ::erf(x));
        n64++;
    }
    printf("float64 arguments, one per upper word (%lld): exp differs on %lld, erf on %lld\n", n64, bad_exp64, bad_erf64);
    return (bad_exp || bad_erf || bad_exp64 || bad_erf64) ? 1 : 0;
}
