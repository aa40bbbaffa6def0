// lrm_exact_math.h -- atan2f / sincosf whose results are bit-identical to glibc 2.35's
// (the libm behind the reference's host path), usable in device code.
//
// Why: the reference decides reachability with strict float comparisons on values that come
// out of atan2f and sincosf (one_leg.cu:26-29, :146-156; circles.cu.h:51).  A device libm
// that differs in the last ulp moves a few boundary points across a comparison.  Restating
// the two published algorithms glibc uses makes the strict kernels bit-identical to the
// reference's host path instead of "equal up to boundary points".
//
//  * atan2f/atanf: the FDLIBM single-precision algorithm (Sun Microsystems, 1993; glibc
//    sysdeps/ieee754/flt-32/e_atan2f.c, s_atanf.c): argument reduction to one of four
//    intervals + an 11-term odd polynomial, all in float, no FMA.
//  * sincosf: the double-precision polynomial algorithm of ARM Optimized Routines (Szabolcs
//    Nagy, 2018; glibc sysdeps/ieee754/flt-32/s_sincosf.h): reduction by pi/2 with a scaled
//    float->int conversion, degree-9/8 polynomials in double.  glibc's x86-64 build selects
//    its FMA variant on AVX2 hosts; with fused multiply-adds this restatement matches that
//    variant on all 2.2e9 floats with |x| < 120, and with or without FMA on |x| < 7 (the
//    only range the path produces: angles in [-2pi, 2pi]).  tests/test_exact_math.py checks it.
//    |x| >= 120 (never produced by the path) is answered with nan, like inf and nan.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "lrm_types.h"

LRM_HD uint32_t lrm_f2u(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
#endif
}
LRM_HD float lrm_u2f(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    memcpy(&f, &u, 4);
    return f;
#endif
}

// float atanf(float) -- FDLIBM s_atanf.c
LRM_HD float lrm_atanf(float x) {
    const float hi0 = 4.6364760399e-01f, hi1 = 7.8539812565e-01f, hi2 = 9.8279368877e-01f,
                hi3 = 1.5707962513e+00f;
    const float lo0 = 5.0121582440e-09f, lo1 = 3.7748947079e-08f, lo2 = 3.4473217170e-08f,
                lo3 = 7.5497894159e-08f;
    const float a0 = 3.3333334327e-01f, a1 = -2.0000000298e-01f, a2 = 1.4285714924e-01f,
                a3 = -1.1111110449e-01f, a4 = 9.0908870101e-02f, a5 = -7.6918758452e-02f,
                a6 = 6.6610731184e-02f, a7 = -5.8335702866e-02f, a8 = 4.9768779427e-02f,
                a9 = -3.6531571299e-02f, a10 = 1.6285819933e-02f;
    const int32_t hx = (int32_t)lrm_f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    if (ix >= 0x4c000000) { // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        return (hx > 0) ? hi3 + lo3 : -hi3 - lo3;
    }
    int id;
    float hi = 0.f, lo = 0.f;
    if (ix < 0x3ee00000) { // |x| < 0.4375
        if (ix < 0x31000000) return x;
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) { id = 0; hi = hi0; lo = lo0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else { id = 1; hi = hi1; lo = lo1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; hi = hi2; lo = lo2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else { id = 3; hi = hi3; lo = lo3; x = -1.0f / x; }
        }
    }
    float z = x * x;
    const float w = z * z;
    const float s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
    const float s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
    if (id < 0) return x - x * (s1 + s2);
    z = hi - ((x * (s1 + s2) - lo) - x);
    return (hx < 0) ? -z : z;
}

// float atan2f(float y, float x) -- FDLIBM e_atan2f.c
LRM_HD float lrm_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f,
                pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = (int32_t)lrm_f2u(x), hy = (int32_t)lrm_f2u(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return lrm_atanf(y);
    const int32_t m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) {
        if (m < 2) return y;
        return (m == 2) ? pi + tiny : -pi - tiny;
    }
    if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
            case 0: return pi_o_4 + tiny;
            case 1: return -pi_o_4 - tiny;
            case 2: return 3.0f * pi_o_4 + tiny;
            default: return -3.0f * pi_o_4 - tiny;
            }
        } else {
            switch (m) {
            case 0: return 0.0f;
            case 1: return -0.0f;
            case 2: return pi + tiny;
            default: return -pi - tiny;
            }
        }
    }
    if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = lrm_atanf(fabsf(y / x));
    switch (m) {
    case 0: return z;
    case 1: return lrm_u2f(lrm_f2u(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}

// void sincosf(float, float*, float*) -- ARM optimized routines / glibc s_sincosf.h
LRM_HD void lrm_sincosf(float y, float* sinp, float* cosp) {
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5,
                 C3 = -0x1.6c087e89a359dp-10, C4 = 0x1.99343027bf8c3p-16;
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    const uint32_t top = (lrm_f2u(y) >> 20) & 0x7ff;
    double x = (double)y;
    int n = 0;
    double csign = 1.0; // table 1 of the original negates the cosine polynomial
    if (top < 0x3f4u) { // |y| < pi/4
        if (top < 0x398u) { // |y| < 2^-12
            *sinp = y;
            *cosp = 1.0f;
            return;
        }
    } else if (top < 0x42fu) { // |y| < 120
        const double r = x * hpi_inv;
        n = ((int32_t)r + 0x800000) >> 24;
        x = __builtin_fma(-(double)n, hpi, x);
        const double s = ((n + 1) & 2) ? -1.0 : 1.0; // sign[] = {1,-1,-1,1}
        if (n & 2) csign = -1.0;
        x = x * s;
    } else {
        // |y| >= 120, inf, nan: outside the emulated range (the path only produces angles in
        // [-2pi, 2pi]).  inf/nan give nan as in glibc; a finite argument this large gives nan as
        // well, so that a misuse is loud instead of silently inexact.
        *sinp = *cosp = y - y + __builtin_nanf("");
        return;
    }
    const double x2 = x * x;
    const double x4 = x2 * x2;
    const double x3 = x2 * x;
    const double c2 = __builtin_fma(x2, csign * C4, csign * C3);
    const double s1 = __builtin_fma(x2, S3, S2);
    const double c1 = __builtin_fma(x2, csign * C1, csign * C0);
    const double x5 = x3 * x2;
    const double x6 = x4 * x2;
    const double s = __builtin_fma(x3, S1, x);
    const double c = __builtin_fma(x4, csign * C2, c1);
    const float sv = (float)__builtin_fma(x5, s1, s);
    const float cv = (float)__builtin_fma(x6, c2, c);
    if (n & 1) { *sinp = cv; *cosp = sv; }
    else { *sinp = sv; *cosp = cv; }
}
