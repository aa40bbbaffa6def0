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
//    only range the path produces: angles in [-2pi, 2pi]).  tests/test_capi_cpu.py (host build against this
//    machine's glibc) and tests/test_gpu_parity.py (device build against the host build) check it.
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

// Correctly rounded sqrtf.  Host: libm.  Device: the compiler's IEEE expansion (v_sqrt_f32, two
// fused residual tests for +-1 ulp, denormal scaling: ~16 instructions, 13 of them half-rate on
// gfx950) is replaced, for x in [2^-96, 2^96), by the Goldschmidt / Markstein sequence on v_rsq_f32
// (the sequence LLVM itself emits for IEEE sqrt when denormals are flushed): 8 instructions.
// Everything else (0, denormals, huge, inf, nan, negative) takes the compiler's expansion.
// tests/test_gpu_parity.py runs it against sqrtf on all 2^32 bit patterns on the device.
LRM_HD float lrm_sqrtf(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (__builtin_expect((lrm_f2u(x) - 0x0f800000u) >= 0x60000000u, 0)) return sqrtf(x);
    const float rs = __builtin_amdgcn_rsqf(x);
    float g = x * rs, h = 0.5f * rs;
    const float e = __builtin_fmaf(-h, g, 0.5f);
    h = __builtin_fmaf(h, e, h);
    g = __builtin_fmaf(g, e, g);
    const float r = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(r, h, g);
#else
    return sqrtf(x);
#endif
}

// float atanf(float) -- FDLIBM s_atanf.c, the five argument ranges evaluated through selects
// instead of branches (a wave would otherwise walk all of them one after the other).  Every range
// performs exactly the operations of the original in the original order:
//   reduction  x' = (fl(A*|x|) - B) / (C + fl(D*|x|))  with (A,B,C,D) = (2,1,2,1), (1,1,1,1), (1,1.5,1,1.5);
//   multiplying by 1.0f is exact, so the shared form changes no bit.
LRM_HD float lrm_atanf(float x) {
    const float a0 = 3.3333334327e-01f, a1 = -2.0000000298e-01f, a2 = 1.4285714924e-01f,
                a3 = -1.1111110449e-01f, a4 = 9.0908870101e-02f, a5 = -7.6918758452e-02f,
                a6 = 6.6610731184e-02f, a7 = -5.8335702866e-02f, a8 = 4.9768779427e-02f,
                a9 = -3.6531571299e-02f, a10 = 1.6285819933e-02f;
    const int32_t hx = (int32_t)lrm_f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    const float ax = fabsf(x);
    const bool small = ix < 0x3ee00000; // |x| < 0.4375: no reduction
    const bool r0 = ix < 0x3f300000, r1 = ix < 0x3f980000, r2 = ix < 0x401c0000;
    const float A = r0 ? 2.0f : 1.0f;
    const float B = r1 ? 1.0f : 1.5f; // ranges 0 and 1 subtract 1, range 2 subtracts 1.5
    const float C = r0 ? 2.0f : 1.0f;
    const float D = r1 ? 1.0f : 1.5f;
    float num = r2 ? (A * ax - B) : -1.0f;
    float den = r2 ? (C + D * ax) : ax;
    num = small ? x : num;
    den = small ? 1.0f : den;
    const float hi = r0 ? 4.6364760399e-01f : r1 ? 7.8539812565e-01f : r2 ? 9.8279368877e-01f : 1.5707962513e+00f;
    const float lo = r0 ? 5.0121582440e-09f : r1 ? 3.7748947079e-08f : r2 ? 3.4473217170e-08f : 7.5497894159e-08f;
    const float xr = num / den;
    const float z = xr * xr;
    const float w = z * z;
    const float s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
    const float s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
    const float t = xr * (s1 + s2);
    const float zz = hi - ((t - lo) - xr);
    float r = small ? (xr - t) : ((hx < 0) ? -zz : zz);
    r = (ix < 0x31000000) ? x : r; // |x| < 2^-29
    if (ix >= 0x4c000000) {        // |x| >= 2^25 (or nan)
        const float big = 1.5707962513e+00f + 7.5497894159e-08f;
        r = (ix > 0x7f800000) ? x + x : ((hx > 0) ? big : -big);
    }
    return r;
}

// the rare arguments of atan2f: zeros, infinities, nans (FDLIBM e_atan2f.c, verbatim structure)
LRM_HD float lrm_atan2f_special(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f;
    const int32_t hx = (int32_t)lrm_f2u(x), hy = (int32_t)lrm_f2u(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
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
        }
        switch (m) {
        case 0: return 0.0f;
        case 1: return -0.0f;
        case 2: return pi + tiny;
        default: return -pi - tiny;
        }
    }
    return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny; // iy == inf
}

// float atan2f(float y, float x) -- FDLIBM e_atan2f.c.  (Its x == 1 shortcut returns atanf(y), which
// is what the general path computes for x == 1, so it is not kept as a separate case.)
LRM_HD float lrm_atan2f(float y, float x) {
    const float pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = (int32_t)lrm_f2u(x), hy = (int32_t)lrm_f2u(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (__builtin_expect((ix == 0) | (iy == 0) | (ix >= 0x7f800000) | (iy >= 0x7f800000), 0))
        return lrm_atan2f_special(y, x);
    const int32_t k = (iy - ix) >> 23;
    float z = lrm_atanf(fabsf(y / x));
    z = (hx < 0 && k < -60) ? 0.0f : z;
    z = (k > 60) ? pi_o_2 + 0.5f * pi_lo : z;
    const float zp = z - pi_lo;
    const float r_neg = (hy < 0) ? (zp - pi) : (pi - zp);          // x < 0
    const float r_pos = lrm_u2f(lrm_f2u(z) ^ ((uint32_t)hy & 0x80000000u)); // x > 0: z with the sign of y
    return (hx < 0) ? r_neg : r_pos;
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
    // For n & 2 the original switches to a second coefficient table whose C0..C4 are negated: every
    // operation of the cosine polynomial then yields the exact negative (IEEE rounding is symmetric
    // in sign), so the polynomial runs once with the positive table and the float result changes sign.
    const double c2 = __builtin_fma(x2, C4, C3);
    const double s1 = __builtin_fma(x2, S3, S2);
    const double c1 = __builtin_fma(x2, C1, C0);
    const double x5 = x3 * x2;
    const double x6 = x4 * x2;
    const double s = __builtin_fma(x3, S1, x);
    const double c = __builtin_fma(x4, C2, c1);
    const float sv = (float)__builtin_fma(x5, s1, s);
    const float cv = lrm_u2f(lrm_f2u((float)__builtin_fma(x6, c2, c)) ^ (((uint32_t)n & 2u) << 30));
    if (n & 1) { *sinp = cv; *cosp = sv; }
    else { *sinp = sv; *cosp = cv; }
}
