/*
 * ss_float_steps.h -- the few floating-point steps inside the otherwise integer ORB path,
 * written so that host and gfx950 produce the same bits (SURVEY.md section 7 "hard parts").
 *
 *   ss_fast_atan2   cv::fastAtan2 (OpenCV mathfuncs_core scalar atan_f32): single-precision,
 *                   one IEEE operation per step, no contraction.
 *   ss_sincosf_deg  what computeOrbDescriptor does with kpt.angle: angle*factorPI, then
 *                   libm cosf / sinf.  The reference image is ubuntu:22.04
 *                   (/root/reference/dockerfile:1) = glibc 2.35, whose sinf/cosf are the
 *                   ARM optimized-routines double-precision polynomials, built with FMA
 *                   (the x86-64 ifunc variant every current host selects).  Restated here
 *                   from the published algorithm; tests/test_float_steps.py pins it against
 *                   this container's glibc 2.35 over every float in [2^-15, 120).
 *
 * Compile with -ffp-contract=off; fused steps are spelled fma().
 */
#ifndef SS_FLOAT_STEPS_H
#define SS_FLOAT_STEPS_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define SS_HD __host__ __device__ __forceinline__
#define SS_FMA(a, b, c) __builtin_fma((a), (b), (c))
#else
#include <math.h>
#define SS_HD static inline
#define SS_FMA(a, b, c) fma((a), (b), (c))
#endif

SS_HD float ss_fast_atan2(float y, float x)
{
    /* 0.9997878412794807f*(float)(180/CV_PI) etc.: float constants times a float */
    const float k = (float)(180 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * k, p3 = -0.3258083974640975f * k;
    const float p5 = 0.1555786518463281f * k, p7 = -0.04432655554792128f * k;
    const float eps = (float)2.2204460492503131e-16;
    const float ax = x < 0 ? -x : x, ay = y < 0 ? -y : y;
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* glibc 2.35 sysdeps/ieee754/flt-32/{s_sincosf.h,s_sincosf_data.c,s_sinf.c,s_cosf.c},
 * FMA build.  Valid for 0 <= |y| < 120 (the ORB angle is in [0, 2*pi + ulp]). */
SS_HD uint32_t ss_abstop12(float x)
{
    union { float f; uint32_t u; } v;
    v.f = x;
    return (v.u >> 20) & 0x7ff;
}

/* n odd: cosine polynomial, n even: sine polynomial; neg selects table entry 1 */
SS_HD float ss_sincosf_poly(double x, double x2, int n, int neg)
{
    const double c0 = neg ? -0x1p0 : 0x1p0;
    const double c1 = neg ? 0x1.ffffffd0c621cp-2 : -0x1.ffffffd0c621cp-2;
    const double c2 = neg ? -0x1.55553e1068f19p-5 : 0x1.55553e1068f19p-5;
    const double c3 = neg ? 0x1.6c087e89a359dp-10 : -0x1.6c087e89a359dp-10;
    const double c4 = neg ? -0x1.99343027bf8c3p-16 : 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double t1 = SS_FMA(x2, s3, s2);
        const double x7 = x3 * x2;
        const double s = SS_FMA(x3, s1, x);
        return (float)SS_FMA(x7, t1, s);
    } else {
        const double x4 = x2 * x2;
        const double t2 = SS_FMA(x2, c4, c3);
        const double t1 = SS_FMA(x2, c1, c0);
        const double x6 = x4 * x2;
        const double c = SS_FMA(x4, c2, t1);
        return (float)SS_FMA(x6, t2, c);
    }
}

SS_HD void ss_sincosf(float y, float *sin_out, float *cos_out)
{
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    double x = (double)y;
    if (ss_abstop12(y) < ss_abstop12(0x1.921FB6p-1f)) {
        const double x2 = x * x;
        if (ss_abstop12(y) < ss_abstop12(0x1p-12f)) {
            /* sinf returns y, cosf returns 1 (the underflow-raising branch has no value effect) */
            *sin_out = y;
            *cos_out = 1.0f;
            return;
        }
        *sin_out = ss_sincosf_poly(x, x2, 0, 0);
        *cos_out = ss_sincosf_poly(x, x2, 1, 0);
        return;
    }
    /* reduce_fast */
    const double r = x * hpi_inv;
    const int n = ((int32_t)r + 0x800000) >> 24;
    x = SS_FMA(-(double)n, hpi, x);
    const double sign = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0; /* sign[4] = {1,-1,-1,1} */
    const int neg = (n & 2) ? 1 : 0;
    const double xs = x * sign, x2 = x * x;
    *sin_out = ss_sincosf_poly(xs, x2, n, neg);
    *cos_out = ss_sincosf_poly(xs, x2, n ^ 1, neg);
}

/* a = cosf(angle_deg * factorPI), b = sinf(same); factorPI = (float)(CV_PI / 180.f) */
SS_HD void ss_sincosf_deg(float angle_deg, float *b_sin, float *a_cos)
{
    const float factor_pi = (float)(3.14159265358979323846 / 180.f);
    ss_sincosf(angle_deg * factor_pi, b_sin, a_cos);
}

#endif
