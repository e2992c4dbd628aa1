// Bit-reproducible FP32 transcendental functions for the scan matcher.
//
// Why: the reference's matchers (src/zhpsm/ZHPolar_Match.cpp) iterate 20-40 times
// through hard gates (integer shift arg-min, `<100 cm`, trimming, throw), so the
// "1e-4 m / 1e-4 rad" parity target is only reachable when every FP32 operation
// reproduces the reference CPU build bit for bit (SURVEY.md §7 "hard parts").
// The reference calls glibc's sinf / cosf / atan2f; the device has no libm, and
// the HIP device library's versions round differently.  These routines evaluate
// the same published algorithms glibc 2.35 uses, with the same operation order,
// so the same inputs give the same bits on the host and on gfx950:
//
//   sinf/cosf : ARM optimized-routines single-precision sin/cos (glibc
//               sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h): double
//               arithmetic, |x| < pi/4 direct polynomial, |x| < 120 one-step
//               reduction by pi/2.  glibc selects its FMA build on every CPU this
//               project runs on, so the contractions that build performs are made
//               explicit here with fma().
//   atan2f    : the Sun/FreeBSD float kernels glibc 2.35 still ships
//               (e_atan2f.c, s_atanf.c): pure FP32, no contraction.
//
// tests/test_math_bits.py checks these against the container's libm over every
// float in the supported domain.  Everything is compiled with -ffp-contract=off;
// the only fused operations are the explicit fma() calls below.
//
// Domain: |x| < 120 for sin/cos (angles here are a pose angle plus a bearing);
// outside it the result is NaN so a violation cannot pass silently.
#ifndef LSA_MATH_H
#define LSA_MATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LSA_HD __host__ __device__ __forceinline__
#else
#define LSA_HD static inline
#endif

namespace lsa {

LSA_HD uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
LSA_HD float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// 2/pi * 2^24, pi/2
#define LSA_HPI_INV 0x1.45F306DC9C883p+23
#define LSA_HPI     0x1.921FB54442D18p0

// polynomial kernels; `neg` selects the table whose cosine is negated (quadrants 2,3)
LSA_HD float sin_poly(double x, double x2)
{
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    double x3 = x * x2;
    double s1 = __builtin_fma(x2, S3, S2);
    double x7 = x3 * x2;
    double s = __builtin_fma(x3, S1, x);
    return (float)__builtin_fma(x7, s1, s);
}

LSA_HD float cos_poly(double x2, bool neg)
{
    double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5,
           C3 = -0x1.6c087e89a359dp-10, C4 = 0x1.99343027bf8c3p-16;
    if (neg) { C0 = -C0; C1 = -C1; C2 = -C2; C3 = -C3; C4 = -C4; }
    double x4 = x2 * x2;
    double t2 = __builtin_fma(x2, C4, C3);
    double t1 = __builtin_fma(x2, C1, C0);
    double x6 = x4 * x2;
    double c = __builtin_fma(x4, C2, t1);
    return (float)__builtin_fma(x6, t2, c);
}

LSA_HD uint32_t abstop12(float x) { return (f2u(x) >> 20) & 0x7ff; }

// which = 0: sine, 1: cosine
LSA_HD float sincos_core(float y, int which)
{
    double x = y;
    const uint32_t top = abstop12(y);
    if (top < abstop12(0x1.921FB6p-1f)) {
        if (top < abstop12(0x1p-12f)) return which ? 1.0f : y;
        return which ? cos_poly(x * x, false) : sin_poly(x, x * x);
    }
    if (top < abstop12(120.0f)) {
        double r = x * LSA_HPI_INV;
        int n = ((int32_t)r + 0x800000) >> 24;
        x = __builtin_fma(-(double)n, LSA_HPI, x);
        const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
        const bool neg = (n & 2) != 0;
        const int q = n ^ which;
        if ((q & 1) == 0) return sin_poly(x * sgn, x * x);
        return cos_poly(x * x, neg);
    }
    return u2f(0x7fc00000u);
}

LSA_HD float sinf_(float x) { return sincos_core(x, 0); }
LSA_HD float cosf_(float x) { return sincos_core(x, 1); }

// both at once: one range reduction, bit-identical to sinf_(y) and cosf_(y)
LSA_HD void sincosf_(float y, float& sn, float& cs)
{
    double x = y;
    const uint32_t top = abstop12(y);
    if (top < abstop12(0x1.921FB6p-1f)) {
        if (top < abstop12(0x1p-12f)) { sn = y; cs = 1.0f; return; }
        const double x2 = x * x;
        sn = sin_poly(x, x2);
        cs = cos_poly(x2, false);
        return;
    }
    if (top < abstop12(120.0f)) {
        double r = x * LSA_HPI_INV;
        int n = ((int32_t)r + 0x800000) >> 24;
        x = __builtin_fma(-(double)n, LSA_HPI, x);
        const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
        const bool neg = (n & 2) != 0;
        const double x2 = x * x;
        const float a = sin_poly(x * sgn, x2), b = cos_poly(x2, neg);
        if (n & 1) { sn = b; cs = a; } else { sn = a; cs = b; }
        return;
    }
    sn = cs = u2f(0x7fc00000u);
}

// ---- atanf / atan2f (FP32 kernels) ---------------------------------------

LSA_HD float fabsf_(float x) { return u2f(f2u(x) & 0x7fffffffu); }

LSA_HD float atanf_(float x)
{
    const float hi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float lo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f,
                          9.0908870101e-02f, -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f,
                          4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
    const int32_t hx = (int32_t)f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {                       // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? hi[3] + lo[3] : -hi[3] - lo[3];
    }
    if (ix < 0x3ee00000) {                        // |x| < 0.4375
        if (ix < 0x31000000) return x;            // |x| < 2^-29
        id = -1;
    } else {
        x = fabsf_(x);
        if (ix < 0x3f980000) {                    // |x| < 1.1875
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else                 { id = 1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else                 { id = 3; x = -1.0f / x; }
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = hi[id] - ((x * (s1 + s2) - lo[id]) - x);
    return hx < 0 ? -r : r;
}

LSA_HD float atan2f_(float y, float x)
{
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f,
                pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = (int32_t)f2u(x), hy = (int32_t)f2u(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return atanf_(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) {
        switch (m) {
        case 0: case 1: return y;
        case 2: return pi + tiny;
        default: return -pi - tiny;
        }
    }
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
            case 0: return pi_o_4 + tiny;
            case 1: return -pi_o_4 - tiny;
            case 2: return 3.0f * pi_o_4 + tiny;
            default: return -3.0f * pi_o_4 - tiny;
            }
        }
        switch (m) {
        case 0: return 0.0f;
        case 1: return -0.0f;
        case 2: return pi + tiny;
        default: return -pi - tiny;
        }
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = atanf_(fabsf_(y / x));
    switch (m) {
    case 0: return z;
    case 1: return u2f(f2u(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}

// ---- straight-line variants for the kernels' inner loops ---------------------
// Same results bit for bit (tests/native/math_bits_check.cpp), but without data
// dependent branches on the common path, so that the compiler can interleave the
// evaluation for several beams.  Rare inputs fall back to the functions above.

LSA_HD float cos_poly_pos(double x2)
{
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5,
                 C3 = -0x1.6c087e89a359dp-10, C4 = 0x1.99343027bf8c3p-16;
    double x4 = x2 * x2;
    double t2 = __builtin_fma(x2, C4, C3);
    double t1 = __builtin_fma(x2, C1, C0);
    double x6 = x4 * x2;
    double c = __builtin_fma(x4, C2, t1);
    return (float)__builtin_fma(x6, t2, c);
}

// `special` is set when the caller must use sincosf_ instead (|y| >= 120, NaN)
LSA_HD void sincosf_core(float y, float& sn, float& cs, bool& special)
{
    special = !(abstop12(y) < abstop12(120.0f));
    double x = special ? 0.0 : (double)y;
    const double r = x * LSA_HPI_INV;
    const int n = ((int32_t)r + 0x800000) >> 24;
    x = __builtin_fma(-(double)n, LSA_HPI, x);
    const double sgn = ((n + 1) & 2) ? -1.0 : 1.0;           // quadrants 1,2
    const double x2 = x * x;
    const float a = sin_poly(x * sgn, x2);
    float b = cos_poly_pos(x2);
    b = (n & 2) ? -b : b;                                    // the negated-cosine table, exactly
    const bool tiny = abstop12(y) < abstop12(0x1p-12f);     // keeps sin(-0) = -0
    sn = tiny ? y : ((n & 1) ? b : a);
    cs = tiny ? 1.0f : ((n & 1) ? a : b);
}

LSA_HD void sincosf_fast(float y, float& sn, float& cs)
{
    bool special;
    sincosf_core(y, sn, cs, special);
    if (special) sincosf_(y, sn, cs);
}

// `special` is set when the caller must use atan2f_ instead (zeros, infinities, NaN, extreme ratios)
LSA_HD float atan2f_core(float y, float x, bool& special)
{
    const int32_t hx = (int32_t)f2u(x), hy = (int32_t)f2u(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    const float v = fabsf_(y / x);
    const int32_t iv = (int32_t)f2u(v);
    const int32_t k = (iy - ix) >> 23;
    special = iy == 0 || ix == 0 || ix >= 0x7f800000 || iy >= 0x7f800000 || k > 60 || k < -60 ||
              iv >= 0x4c000000 || iv < 0x31000000;
    // argument reduction of atanf, as selects
    const int id = iv < 0x3ee00000 ? -1 : iv < 0x3f300000 ? 0 : iv < 0x3f980000 ? 1 : iv < 0x401c0000 ? 2 : 3;
    const float num = id < 0 ? v : id == 0 ? 2.0f * v - 1.0f : id == 1 ? v - 1.0f : id == 2 ? v - 1.5f : -1.0f;
    const float den = id < 0 ? 1.0f : id == 0 ? 2.0f + v : id == 1 ? v + 1.0f : id == 2 ? 1.0f + 1.5f * v : v;
    const float xr = num / den;
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    const float z = xr * xr;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    const float hi = id == 0 ? 4.6364760399e-01f : id == 1 ? 7.8539812565e-01f : id == 2 ? 9.8279368877e-01f : 1.5707962513e+00f;
    const float lo = id == 0 ? 5.0121582440e-09f : id == 1 ? 3.7748947079e-08f : id == 2 ? 3.4473217170e-08f : 7.5497894159e-08f;
    const float r_small = xr - xr * (s1 + s2);
    const float r_big = hi - ((xr * (s1 + s2) - lo) - xr);
    const float zt = id < 0 ? r_small : r_big;
    const float pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    const float q0 = zt, q1 = u2f(f2u(zt) ^ 0x80000000u), q2 = pi - (zt - pi_lo), q3 = (zt - pi_lo) - pi;
    return m == 0 ? q0 : m == 1 ? q1 : m == 2 ? q2 : q3;
}

LSA_HD float atan2f_fast(float y, float x)
{
    bool special;
    const float a = atan2f_core(y, x, special);
    return special ? atan2f_(y, x) : a;
}

} // namespace lsa
#endif
