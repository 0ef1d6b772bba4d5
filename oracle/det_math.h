/* oracle/det_math.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Deterministic single-precision elementary functions for the CPU oracle.
 *
 * Why: the reference calls the OpenCL C builtins atan2/sincos/sin/cos/tan/acos/fmod/
 * remainder/hypot/length (reference shapes/simple2d.cl:16-46, simple3d.cl:23-97,
 * unsafe.cl:1-23, gears.cl:1-42, polygons2d.cl:1-74, cl_util/util.cl:1-15).  Those
 * live in whatever OpenCL runtime the user has (POCL on the author's CI,
 * .travis.yml:10-15) -- a third-party dependency that is absent from /root/reference
 * and unpinned by design: the program is built with -cl-fast-relaxed-math
 * (cl_util/opencl_manager.py:12-18), so no particular rounding is promised.
 *
 * This project therefore fixes ONE published single-precision algorithm per function
 * (the classic Cephes `single/` routines by S. L. Moshier: atanf.c, sinf.c, tanf.c,
 * asinf.c -- range reduction + short minimax polynomial, |rel err| <~ 2e-7) and writes
 * every operation explicitly (fmaf where fused, plain * and + where not).  The HIP
 * kernels implement the same sequence of IEEE-754 binary32 operations, so oracle and
 * GPU agree BIT FOR BIT, and both agree with any conforming libm/OpenCL runtime to
 * far better than the 1e-5 relative tolerance of the north star
 * (tests/test_oracle_math.py checks the oracle against float64 libm).
 *
 * Build rule: -ffp-contract=off, no -ffast-math.  fmaf() must be a true fused
 * multiply-add (glibc's fmaf is correctly rounded with or without hardware FMA).
 */
#ifndef ORACLE_DET_MATH_H
#define ORACLE_DET_MATH_H

#include <math.h>
#include <stdint.h>

#define DM_PI_F 3.14159274101257324f      /* (float)pi: OpenCL M_PI_F */
#define DM_PI_2_F 1.57079637050628662f    /* (float)(pi/2) */
#define DM_PI_4_F 0.785398185253143311f   /* (float)(pi/4) */
#define DM_2PI_F (2.0f * DM_PI_F)         /* util.h:4 M_2PI_F */

/* length / hypot: sqrt of an fma-accumulated sum of squares (IEEE sqrt). */
static inline float dm_length2(float x, float y) { return sqrtf(fmaf(y, y, x * x)); }
static inline float dm_length3(float x, float y, float z)
{
    return sqrtf(fmaf(z, z, fmaf(y, y, x * x)));
}
static inline float dm_hypot(float x, float y) { return dm_length2(x, y); }

/* atan for x >= 0 (Cephes atanf.c: two-step argument reduction, degree-4 in z=x^2). */
static inline float dm_atan_pos(float x)
{
    float y;
    if (x > 2.414213562373095f) { /* tan(3pi/8) */
        y = DM_PI_2_F;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) { /* tan(pi/8) */
        y = DM_PI_4_F;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        y = 0.0f;
    }
    float z = x * x;
    float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fmaf(p, z, 1.99777106478e-1f);
    p = fmaf(p, z, -3.33329491539e-1f);
    p = p * z;
    return y + fmaf(p, x, x);
}

/* atan2(y, x) with the usual quadrant rules; (0,0) -> 0. */
static inline float dm_atan2(float y, float x)
{
    float ax = fabsf(x), ay = fabsf(y);
    float t;
    if (ay == 0.0f)
        t = 0.0f;
    else if (ax == 0.0f)
        t = DM_PI_2_F;
    else
        t = dm_atan_pos(ay / ax);
    if (x < 0.0f)
        t = DM_PI_F - t;
    return copysignf(t, y);
}

/* float -> int32 defined for every input: NaN and values outside (-2^31, 2^31) give 0 (a plain cast is
 * undefined behaviour there; x86 gives INT_MIN, gfx950 0 or saturation).  Same rule in the kernels. */
static inline int32_t dm_to_int(float v) { return (v > -2147483648.0f && v < 2147483648.0f) ? (int32_t)v : 0; }

/* Cody-Waite reduction by pi/4 octants (Cephes sinf.c constants).  Returns r in
 * [-pi/4, pi/4] and the (even) octant count j so that |x| = r + j*pi/4. */
static inline float dm_reduce_pio4(float ax, int32_t *j_out)
{
    int32_t j = dm_to_int(ax * 1.27323954473516f); /* 4/pi */
    j += (j & 1);
    float y = (float)j;
    float r = fmaf(-y, 0.78515625f, ax);
    r = fmaf(-y, 2.4187564849853515625e-4f, r);
    r = fmaf(-y, 3.77489497744594108e-8f, r);
    *j_out = j;
    return r;
}

static inline float dm_sin_poly(float r, float z)
{
    float p = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    p = fmaf(p, z, -1.6666654611e-1f);
    p = p * z;
    return fmaf(p, r, r);
}

static inline float dm_cos_poly(float z)
{
    float p = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    p = fmaf(p, z, 4.166664568298827e-2f);
    p = p * z;
    p = fmaf(p, z, fmaf(-0.5f, z, 1.0f));
    return p;
}

/* sincos: *s = sin(x), *c = cos(x). */
static inline void dm_sincos(float x, float *s, float *c)
{
    int32_t j;
    float r = dm_reduce_pio4(fabsf(x), &j);
    float z = r * r;
    float ps = dm_sin_poly(r, z);
    float pc = dm_cos_poly(z);
    int32_t q = (j >> 1) & 3;
    float ss = (q & 1) ? pc : ps;
    float cc = (q & 1) ? ps : pc;
    if (q == 2 || q == 3) ss = -ss;
    if (q == 1 || q == 2) cc = -cc;
    if (x < 0.0f) ss = -ss;
    *s = ss;
    *c = cc;
}

static inline float dm_sin(float x) { float s, c; dm_sincos(x, &s, &c); return s; }
static inline float dm_cos(float x) { float s, c; dm_sincos(x, &s, &c); return c; }

/* tan (Cephes tanf.c polynomial on the reduced argument). */
static inline float dm_tan(float x)
{
    int32_t j;
    float r = dm_reduce_pio4(fabsf(x), &j);
    float z = r * r;
    float p = fmaf(9.38540185543e-3f, z, 3.11992232697e-3f);
    p = fmaf(p, z, 2.44301354525e-2f);
    p = fmaf(p, z, 5.34112807005e-2f);
    p = fmaf(p, z, 1.33387994085e-1f);
    p = fmaf(p, z, 3.33331568548e-1f);
    p = p * z;
    float y = fmaf(p, r, r);
    if (j & 2) y = -(1.0f / y);
    return (x < 0.0f) ? -y : y;
}

/* asin for 0 <= a <= 1 (Cephes asinf.c). */
static inline float dm_asin_pos(float a)
{
    float x, z;
    int flag = a > 0.5f;
    if (flag) {
        z = 0.5f * (1.0f - a);
        x = sqrtf(z);
    } else {
        x = a;
        z = x * x;
    }
    float p = fmaf(4.2163199048e-2f, z, 2.4181311049e-2f);
    p = fmaf(p, z, 4.5470025998e-2f);
    p = fmaf(p, z, 7.4953002686e-2f);
    p = fmaf(p, z, 1.6666752422e-1f);
    p = p * z;
    float r = fmaf(p, x, x);
    if (flag) r = DM_PI_2_F - (r + r);
    return r;
}

/* acos on [-1, 1]; arguments outside are clamped (the gear op can hit 1+eps). */
static inline float dm_acos(float x)
{
    if (x > 1.0f) x = 1.0f;
    if (x < -1.0f) x = -1.0f;
    if (x < -0.5f) {
        float t = dm_asin_pos(sqrtf(0.5f * (1.0f + x)));
        return DM_PI_F - (t + t);
    }
    if (x > 0.5f) {
        float t = dm_asin_pos(sqrtf(0.5f * (1.0f - x)));
        return t + t;
    }
    float t = dm_asin_pos(fabsf(x));
    return DM_PI_2_F - ((x < 0.0f) ? -t : t);
}

/* fmod(x, y) for y > 0 and finite x: truncated quotient, exact residual by fma,
 * one correction step each way for a mis-rounded quotient. */
static inline float dm_fmod(float x, float y)
{
    float q = truncf(x / y);
    float r = fmaf(-q, y, x);
    if (x >= 0.0f) {
        if (r < 0.0f) r = r + y;
        if (r >= y) r = r - y;
    } else {
        if (r > 0.0f) r = r - y;
        if (r <= -y) r = r + y;
    }
    return r;
}

/* remainder(x, y) = x - rint(x * (1/y)) * y with the reciprocal rounded once
 * (inv_y is a per-tape constant), residual exact by fma.  y = +inf (inv_y == 0)
 * returns x: reference shapes/unsafe.py:29-31 encodes "no repetition on this axis"
 * as an infinite spacing and relies on remainder(x, inf) == x. */
static inline float dm_remainder_inv(float x, float y, float inv_y)
{
    if (inv_y == 0.0f) return x;
    float n = rintf(x * inv_y);
    return fmaf(-n, y, x);
}

#endif /* ORACLE_DET_MATH_H */
