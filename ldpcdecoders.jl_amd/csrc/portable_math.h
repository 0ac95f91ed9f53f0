/*
 * portable_math.h -- exp / log / tanh / atanh built from IEEE-754 double +,-,*,/ and bit moves only.
 *
 * Why: BP-OTS (src/decoders/bpots_decoder.jl:182-211) pushes messages through tanh and atanh and
 * then takes data-dependent decisions on them (sign of an LLR, arg-min |LLR| for the bias).  libm
 * on the host and OCML on the device each return tanh/atanh to ~1 ulp but not the SAME ulp, and on
 * symmetric Tanner graphs (the reference's cycle-matrix tests) near-ties are the rule, so two
 * correct libms can send the decoder down different paths.  With these functions the HIP kernel
 * and the CPU oracle execute the same operation sequence (compile with -ffp-contract=off) and
 * agree bit for bit.  Accuracy against libm is checked in tests/test_portable_math.py (a few ulp).
 *
 * Plain C99 and HIP device code alike.
 */
#ifndef LDPC_PORTABLE_MATH_H
#define LDPC_PORTABLE_MATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define PM_FN __host__ __device__ static inline
#else
#define PM_FN static inline
#endif

PM_FN double pm_from_bits(uint64_t u) { double d; memcpy(&d, &u, sizeof d); return d; }
PM_FN uint64_t pm_to_bits(double d) { uint64_t u; memcpy(&u, &d, sizeof u); return u; }
PM_FN double pm_inf(void) { return pm_from_bits(0x7FF0000000000000ull); }
PM_FN double pm_nan(void) { return pm_from_bits(0x7FF8000000000000ull); }
PM_FN double pm_fabs(double x) { return pm_from_bits(pm_to_bits(x) & 0x7FFFFFFFFFFFFFFFull); }
PM_FN double pm_copysign(double m, double s)
{
    return pm_from_bits((pm_to_bits(m) & 0x7FFFFFFFFFFFFFFFull) | (pm_to_bits(s) & 0x8000000000000000ull));
}
/* 2^k for -1022 <= k <= 1023 */
PM_FN double pm_pow2(int k) { return pm_from_bits((uint64_t)(k + 1023) << 52); }

/* exp(x): x = k ln2 + r, |r| <= ln2/2, Horner on the Taylor series to r^14/14! (|err| < 1e-18) */
PM_FN double pm_exp(double x)
{
    if (x != x) return x;
    if (x > 709.0) return pm_inf();
    if (x < -708.0) return 0.0;   /* the callers never go near the denormal range */
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double INV_LN2 = 1.44269504088896338700e+00;
    const double t = x * INV_LN2;
    const int k = (int)(t < 0.0 ? t - 0.5 : t + 0.5);
    const double r = (x - (double)k * LN2_HI) - (double)k * LN2_LO;
    double p = 1.0 + r * (1.0 / 14.0);
    p = 1.0 + (r * p) * (1.0 / 13.0);
    p = 1.0 + (r * p) * (1.0 / 12.0);
    p = 1.0 + (r * p) * (1.0 / 11.0);
    p = 1.0 + (r * p) * (1.0 / 10.0);
    p = 1.0 + (r * p) * (1.0 / 9.0);
    p = 1.0 + (r * p) * (1.0 / 8.0);
    p = 1.0 + (r * p) * (1.0 / 7.0);
    p = 1.0 + (r * p) * (1.0 / 6.0);
    p = 1.0 + (r * p) * (1.0 / 5.0);
    p = 1.0 + (r * p) * (1.0 / 4.0);
    p = 1.0 + (r * p) * (1.0 / 3.0);
    p = 1.0 + (r * p) * (1.0 / 2.0);
    p = 1.0 + r * p;
    return p * pm_pow2(k);
}

/* log(x): x = m 2^e, m in [sqrt(1/2), sqrt(2)); log m = 2 atanh z, z = (m-1)/(m+1), |z| <= 0.1716 */
PM_FN double pm_log(double x)
{
    if (x != x) return x;
    if (x < 0.0) return pm_nan();
    if (x == 0.0) return -pm_inf();
    if (x == pm_inf()) return x;
    int e = 0;
    uint64_t u = pm_to_bits(x);
    if ((u >> 52) == 0) {               /* denormal: scale into the normal range first */
        x = x * 18014398509481984.0;    /* 2^54 */
        u = pm_to_bits(x);
        e = -54;
    }
    e += (int)(u >> 52) - 1023;
    double m = pm_from_bits((u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);   /* [1, 2) */
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double z = (m - 1.0) / (m + 1.0);
    const double w = z * z;
    double s = 1.0 / 23.0;
    s = 1.0 / 21.0 + w * s;
    s = 1.0 / 19.0 + w * s;
    s = 1.0 / 17.0 + w * s;
    s = 1.0 / 15.0 + w * s;
    s = 1.0 / 13.0 + w * s;
    s = 1.0 / 11.0 + w * s;
    s = 1.0 / 9.0 + w * s;
    s = 1.0 / 7.0 + w * s;
    s = 1.0 / 5.0 + w * s;
    s = 1.0 / 3.0 + w * s;
    s = 1.0 + w * s;
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    return (double)e * LN2_HI + (2.0 * z * s + (double)e * LN2_LO);
}

/* tanh(x): odd series below 0.25, 1 - 2/(exp(2|x|)+1) above, saturated past 22 */
PM_FN double pm_tanh(double x)
{
    if (x != x) return x;
    const double a = pm_fabs(x);
    double t;
    if (a < 0.25) {
        const double w = a * a;
        /* a (1 - w/3 + 2w^2/15 - 17w^3/315 + 62w^4/2835 - 1382w^5/155925 + 21844w^6/6081075
               - 929569w^7/638512875 + 6404582w^8/10854718875 - 443861162w^9/1856156927625) */
        double s = -443861162.0 / 1856156927625.0;
        s = 6404582.0 / 10854718875.0 + w * s;
        s = -929569.0 / 638512875.0 + w * s;
        s = 21844.0 / 6081075.0 + w * s;
        s = -1382.0 / 155925.0 + w * s;
        s = 62.0 / 2835.0 + w * s;
        s = -17.0 / 315.0 + w * s;
        s = 2.0 / 15.0 + w * s;
        s = -1.0 / 3.0 + w * s;
        s = 1.0 + w * s;
        t = a * s;
    } else if (a > 22.0) {
        t = 1.0;
    } else {
        t = 1.0 - 2.0 / (pm_exp(2.0 * a) + 1.0);
    }
    return pm_copysign(t, x);
}

/* atanh(y): odd series below 0.25, log((1+|y|)/(1-|y|))/2 above; |y| = 1 -> Inf, > 1 -> NaN */
PM_FN double pm_atanh(double y)
{
    if (y != y) return y;
    const double a = pm_fabs(y);
    double t;
    if (a > 1.0) return pm_nan();
    if (a == 1.0) return pm_copysign(pm_inf(), y);
    if (a < 0.25) {
        const double w = a * a;
        double s = 1.0 / 29.0;
        s = 1.0 / 27.0 + w * s;
        s = 1.0 / 25.0 + w * s;
        s = 1.0 / 23.0 + w * s;
        s = 1.0 / 21.0 + w * s;
        s = 1.0 / 19.0 + w * s;
        s = 1.0 / 17.0 + w * s;
        s = 1.0 / 15.0 + w * s;
        s = 1.0 / 13.0 + w * s;
        s = 1.0 / 11.0 + w * s;
        s = 1.0 / 9.0 + w * s;
        s = 1.0 / 7.0 + w * s;
        s = 1.0 / 5.0 + w * s;
        s = 1.0 / 3.0 + w * s;
        s = 1.0 + w * s;
        t = a * s;
    } else {
        t = 0.5 * pm_log((1.0 + a) / (1.0 - a));
    }
    return pm_copysign(t, y);
}

#endif /* LDPC_PORTABLE_MATH_H */
