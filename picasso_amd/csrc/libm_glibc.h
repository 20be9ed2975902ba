// libm_glibc.h — exp() and erf() with the bits glibc gives them on x86-64.
//
// picasso/gaussmle.py calls math.erf and math.exp (gaussmle.py:279, 295, 313, 357): under numba these are the C library's, and
// both are faithful, not correctly rounded — another libm returns another last bit on a few inputs in a thousand.  A fit that
// contracts forgets such a bit when it rounds to float32; a fit that does not (3x3 boxes whose width collapses) carries it
// into another trajectory.  The strict MLE kernel therefore evaluates the two functions the way the reference's libm does —
// glibc >= 2.28 (exp: the table-driven algorithm of sysdeps/ieee754/dbl-64/e_exp.c in the variant the dynamic loader picks
// on a CPU with FMA, where the compiler fused four of its multiply-adds; erf: sysdeps/ieee754/dbl-64/s_erf.c, Sun's rational
// approximations in glibc's evaluation order, no fused operation) — operation for operation as the shipped binary
// (libm.so.6 of glibc 2.35, Ubuntu 22.04; read off its disassembly).  tests/test_libm_glibc.py compiles this header for the
// host and compares it with the C library's functions on 2e7 arguments, special values and range ends included.
//
// (The coefficients are the published ones of those two algorithms — Sun's fdlibm erf, freely distributable, and the exp of
// Arm's optimized-routines, MIT, which glibc adopted in 2.28 — checked here against the constants inside libm.so.6; the 2 KiB
// table is generated from its defining formula, tools/gen_glibc_exp_table.py.)
//
// Everything here must be compiled WITHOUT contraction (the including file sets `#pragma clang fp contract(off)`; the host
// test passes -ffp-contract=off); the fused operations are written out.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define PMI_LIBM_FN __host__ __device__ __forceinline__
#else
#define PMI_LIBM_FN static inline
#endif

namespace pmi_glibc {

// (the table lives with the caller: __constant__ memory on the device, a static array in the host test)
#define PMI_GLIBC_EXP_TABLE_WORDS 256

PMI_LIBM_FN uint64_t to_bits(double v) { uint64_t u; memcpy(&u, &v, 8); return u; }
PMI_LIBM_FN double from_bits(uint64_t u) { double v; memcpy(&v, &u, 8); return v; }

// exp(x); tab = the 256 words of libm_glibc_exp_table.inc
PMI_LIBM_FN double exp(double x, const uint64_t *tab)
{
    const double InvLn2N = 0x1.71547652b82fep+7, Shift = 0x1.8p+52;
    const double NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47;
    const double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
    const uint64_t ix = to_bits(x);
    uint32_t abstop = (uint32_t)(ix >> 52) & 0x7ffu;
    if (abstop - 0x3c9u > 0x3eu) {
        if ((int32_t)(abstop - 0x3c9u) < 0) return 1.0 + x;                  // |x| < 2^-54
        if (abstop > 0x408u) {                                                 // |x| >= 1024, infinities, NaN
            if (ix == 0xfff0000000000000ull) return 0.0;
            if (abstop == 0x7ffu) return 1.0 + x;
            return (int64_t)ix < 0 ? 0x1p-767 * 0x1p-767 : 0x1p769 * 0x1p769;
        }
        abstop = 0u;                                                           // 512 <= |x| < 1024: the scale needs care below
    }
    double kd = __builtin_fma(x, InvLn2N, Shift);
    const uint64_t ki = to_bits(kd);
    kd -= Shift;
    double r = __builtin_fma(kd, NegLn2hiN, x);
    r = __builtin_fma(kd, NegLn2loN, r);
    const uint64_t idx = 2u * (ki & 127u);
    const uint64_t top = ki << 45;
    const double p23 = __builtin_fma(r, C3, C2);
    const double tr = r + from_bits(tab[idx]);
    uint64_t sbits = tab[idx + 1] + top;
    const double r2 = r * r;
    const double p45 = __builtin_fma(r, C5, C4);
    const double t1 = __builtin_fma(p23, r2, tr);
    const double r4 = r2 * r2;
    const double tmp = __builtin_fma(r4, p45, t1);
    if (abstop == 0u) {
        if ((ki & 0x80000000ull) == 0) {                                       // k > 0: the result may overflow
            sbits -= 1009ull << 52;
            const double scale = from_bits(sbits);
            return 0x1p1009 * __builtin_fma(scale, tmp, scale);
        }
        sbits += 1022ull << 52;                                                // k < 0: the result may be subnormal
        const double scale = from_bits(sbits);
        const double st = scale * tmp;
        double y = scale + st;
        if (y < 1.0) {
            double lo = (scale - y) + st;
            const double hi = 1.0 + y;
            lo = ((1.0 - hi) + y) + lo;
            y = (hi + lo) - 1.0;
            if (y == 0.0) y = 0.0;                                             // (no -0)
        }
        return 0x1p-1022 * y;
    }
    const double scale = from_bits(sbits);
    return __builtin_fma(scale, tmp, scale);
}

// erf(x)
PMI_LIBM_FN double erf(double x, const uint64_t *tab)
{
    const uint64_t bx = to_bits(x);
    const int32_t hx = (int32_t)(bx >> 32);
    const int32_t ix = hx & 0x7fffffff;
    if (ix >= 0x7ff00000) return (double)(1 - 2 * (int32_t)((uint32_t)hx >> 31)) + 1.0 / x;      // +-1, NaN
    const double ax = from_bits(bx & 0x7fffffffffffffffull);
    if (ix < 0x3feb0000) {                                                     // |x| < 0.84375
        if (ix < 0x3e300000) {                                                 // |x| < 2^-28
            if (ix < 0x00800000) return 0.0625 * (16.0 * x + 0x1.06eba8214db69p+1 * x);
            return 0x1.06eba8214db69p-3 * x + x;
        }
        const double z = x * x, z2 = z * z, z4 = z2 * z2;
        const double r2 = -0x1.7a291236668e4p-8 * z - 0x1.d2a51dbd7194fp-6;
        const double r1 = -0x1.4cd7d691cb913p-2 * z + 0x1.06eba8214db68p-3;
        const double r = (r2 * z2 + r1) + -0x1.8ead6120016acp-16 * z4;
        const double s2 = 0x1.4d022c4d36b0fp-8 * z + 0x1.0a54c5536cebap-4;
        const double s1 = 0x1.97779cddadc09p-2 * z + 1.0;
        const double s3 = z * -0x1.09c4342a26120p-18 + 0x1.15dc9221c1a10p-13;
        const double s = s3 * z4 + (s2 * z2 + s1);
        const double y = r / s;
        return y * x + x;
    }
    if (ix < 0x3ff40000) {                                                     // 0.84375 <= |x| < 1.25
        const double s = ax - 1.0, s2 = s * s, s4 = s2 * s2, s6 = s2 * s4;
        const double p2 = 0x1.45fca805120e4p-2 * s - 0x1.7d240fbb8c3f1p-2;
        const double p1 = 0x1.a8d00ad92b34dp-2 * s - 0x1.359b8bef77538p-9;
        const double p3 = 0x1.22a36599795ebp-5 * s - 0x1.c63983d3e28ecp-4;
        const double P = ((p2 * s2 + p1) + p3 * s4) + -0x1.1bf380a96073fp-9 * s6;
        const double q2 = 0x1.2635cd99fe9a7p-4 * s + 0x1.14af092eb6f33p-1;
        const double q1 = 0x1.b3e6618eee323p-4 * s + 1.0;
        const double q3 = s * 0x1.bedc26b51dd1cp-7 + 0x1.02660e763351fp-3;
        const double Q = ((q2 * s2 + q1) + q3 * s4) + s6 * 0x1.88b545735151dp-7;
        const double pq = P / Q;
        return hx >= 0 ? pq + 0x1.b0ac160000000p-1 : -0x1.b0ac160000000p-1 - pq;
    }
    if (ix >= 0x40180000) return hx >= 0 ? 1.0 - 0x1.56e1fc2f8f359p-997 : 0x1.56e1fc2f8f359p-997 - 1.0;      // |x| >= 6
    // 1.25 <= |x| < 6: R / S in 1 / x^2, one coefficient set below 1 / 0.35 and one above (a term less in each polynomial)
    const double s = 1.0 / (x * x), s2 = s * s, s4 = s2 * s2, s6 = s2 * s4;
    double R, S;
    if (ix < 0x4006db6e) {
        R = (((s * -0x1.f300ae4cba38dp+5 - 0x1.51e0441b0e726p+3) * s2 + (-0x1.63416e4ba7360p-1 * s - 0x1.43412600d6435p-7))
             + (-0x1.7135cebccabb2p+7 * s - 0x1.44cb184282266p+7) * s4) + (-0x1.3a0efc69ac25cp+3 * s - 0x1.4526557e4d2f2p+6) * s6;
        const double S12 = s2 * (0x1.b290dd58a1a71p+8 * s + 0x1.1350c526ae721p+7) + (0x1.3a6b9bd707687p+4 * s + 1.0);
        S = ((s * 0x1.a47ef8e484a93p+2 + 0x1.b28a3ee48ae2cp+6) * s6 + (S12 + (0x1.ad02157700314p+8 * s + 0x1.42b1921ec2868p+9) * s4))
            + (s4 * s4) * -0x1.eeff2ee749a62p-5;
    } else {
        R = (((s * -0x1.4145d43c5ed98p+7 - 0x1.1c209555f995ap+4) * s2 + (-0x1.993ba70c285dep-1 * s - 0x1.4341239e86f4ap-7))
             + (-0x1.004616a2e5992p+10 * s - 0x1.3ec881375f228p+9) * s4) + -0x1.e384e9bdc383fp+8 * s6;
        const double S12 = s2 * (0x1.802eb189d5118p+10 * s + 0x1.45cae221b9f0ap+8) + (0x1.e568b261d5190p+4 * s + 1.0);
        S = (s * -0x1.670e242712d62p+4 + 0x1.da874e79fe763p+8) * s6 + (S12 + (0x1.3f219cedf3be6p+11 * s + 0x1.8ffb7688c246ap+11) * s4);
    }
    const double z = from_bits(to_bits(ax) & 0xffffffff00000000ull);
    const double e1 = exp((-z) * z - 0x1.2000000000000p-1, tab);
    const double e2 = exp((z - ax) * (z + ax) + R / S, tab);
    const double r = e2 * e1;
    return hx >= 0 ? 1.0 - r / ax : r / ax - 1.0;
}

}  // namespace pmi_glibc
