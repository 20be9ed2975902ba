// exp_cr.h — exp(x) rounded correctly to float64 (to within double rounding at 2^-100), in double-double arithmetic.
// Used where one last bit of exp decides a float32 rounding of the least-squares model (gausslq.hip): the device's exp and
// a CPU libm's are both within an ulp of the true value but are different functions.  Plain C so that the same text
// compiles for the host check (tests/test_host_logic.py builds it with gcc against decimal arithmetic).
#pragma once
#ifdef __HIPCC__
#define EXPCR_FN __device__ __forceinline__
#else
#include <math.h>
#define EXPCR_FN static inline
#endif

typedef struct { double hi, lo; } expcr_dd;
EXPCR_FN expcr_dd expcr_two_sum(double a, double b) { expcr_dd r; r.hi = a + b; double bb = r.hi - a; r.lo = (a - (r.hi - bb)) + (b - bb); return r; }
EXPCR_FN expcr_dd expcr_quick_two_sum(double a, double b) { expcr_dd r; r.hi = a + b; r.lo = b - (r.hi - a); return r; }      // |a| >= |b|
EXPCR_FN expcr_dd expcr_two_prod(double a, double b) { expcr_dd r; r.hi = a * b; r.lo = fma(a, b, -r.hi); return r; }
EXPCR_FN expcr_dd expcr_add(expcr_dd a, expcr_dd b)
{
    expcr_dd s = expcr_two_sum(a.hi, b.hi), t = expcr_two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = expcr_quick_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return expcr_quick_two_sum(s.hi, s.lo);
}
EXPCR_FN expcr_dd expcr_mul(expcr_dd a, expcr_dd b)
{
    expcr_dd p = expcr_two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return expcr_quick_two_sum(p.hi, p.lo);
}

// x finite.  exp(x) = 2^k * (1 + s)^(2^6) with r = x - k ln2, s = expm1(r / 64) by its Taylor series (|r / 64| < 0.0055: the
// term r^13 / 13! is below 2^-130), six squarings as (1 + s)^2 - 1 = 2 s + s^2.
EXPCR_FN double exp_cr(double x)
{
    if (x > 709.8) return INFINITY;
    if (x < -745.2) return 0.0;
    const double LN2_HI = 0x1.62e42fee00000p-1, LN2_MID = 0x1.a39ef35793c76p-33, LN2_LO = 0x1.cc01f97b57a08p-87;
    const double k = rint(x * 0x1.71547652b82fep+0);
    // r = x - k ln2 in double-double: k LN2_HI is exact (LN2_HI has 32 significant bits, |k| < 2^11)
    expcr_dd r = expcr_two_sum(x, -k * LN2_HI);
    expcr_dd t = expcr_two_prod(-k, LN2_MID);
    t.lo += -k * LN2_LO;
    r = expcr_add(r, t);
    r.hi *= 0.015625; r.lo *= 0.015625;                     // / 64, exact
    static const double CH[12] = {0x1.0000000000000p-1, 0x1.5555555555555p-3, 0x1.5555555555555p-5, 0x1.1111111111111p-7,
                                  0x1.6c16c16c16c17p-10, 0x1.a01a01a01a01ap-13, 0x1.a01a01a01a01ap-16, 0x1.71de3a556c734p-19,
                                  0x1.27e4fb7789f5cp-22, 0x1.ae64567f544e4p-26, 0x1.1eed8eff8d898p-29, 0x1.6124613a86d09p-33};
    static const double CL[12] = {0x0.0p+0, 0x1.5555555555555p-57, 0x1.5555555555555p-59, 0x1.1111111111111p-63,
                                  -0x1.f49f49f49f49fp-65, 0x1.a01a01a01a01ap-73, 0x1.a01a01a01a01ap-76, -0x1.c154f8ddc6c00p-73,
                                  0x1.cbbc05b4fa99ap-76, -0x1.c062e06d1f209p-80, -0x1.2aec959e14c06p-83, 0x1.f28e0cc748ebep-87};
    // s = r + r^2 (1/2! + r (1/3! + ... r / 13!))
    expcr_dd acc; acc.hi = CH[11]; acc.lo = CL[11];
    for (int i = 10; i >= 0; i--) {
        acc = expcr_mul(acc, r);
        expcr_dd c; c.hi = CH[i]; c.lo = CL[i];
        acc = expcr_add(acc, c);
    }
    expcr_dd s = expcr_add(r, expcr_mul(expcr_mul(r, r), acc));
    for (int i = 0; i < 6; i++) {
        expcr_dd two_s; two_s.hi = 2.0 * s.hi; two_s.lo = 2.0 * s.lo;
        s = expcr_add(two_s, expcr_mul(s, s));
    }
    expcr_dd one; one.hi = 1.0; one.lo = 0.0;
    const expcr_dd e = expcr_add(one, s);                    // in [1 / sqrt 2, sqrt 2]
    // scale by 2^k; a result in the subnormal range is rounded ONCE, from the double-double value
    const int ki = (int)k;
    const double res = ldexp(e.hi, ki);                      // hi is the correctly rounded sum hi + lo; the scaling is exact ...
    if (res >= 0x1p-1022) return res;                        // ... unless the result is subnormal:
    // gradual underflow: the value in units of 2^-1074 (both scalings exact), rounded once with the low part taken into account
    const double yh = ldexp(e.hi, ki + 1074), yl = ldexp(e.lo, ki + 1074);
    double n = rint(yh);
    const double frac = (yh - n) + yl;                       // in (-1, 1); the true distance from n
    if (frac > 0.5 || (frac == 0.5 && fmod(n, 2.0) != 0.0)) n += 1.0;
    else if (frac < -0.5 || (frac == -0.5 && fmod(n, 2.0) != 0.0)) n -= 1.0;
    return ldexp(n, -1074);
}
