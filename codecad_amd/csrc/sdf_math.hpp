// codecad_amd/csrc/sdf_math.hpp
//
// Deterministic binary32 elementary functions for the gfx950 kernels and for the host-side
// tape pre-decoder (the same functions fold per-tape constants at upload time, so a value
// hoisted out of the per-voxel code is bit-identical to computing it per voxel).
//
// The reference calls OpenCL builtins (atan2, sincos, tan, acos, fmod, remainder, hypot,
// length; reference shapes/*.cl, cl_util/util.cl:1-15) under -cl-fast-relaxed-math
// (cl_util/opencl_manager.py:12-18); no rounding is promised there.  We fix one explicit
// IEEE-754 operation sequence per function (Cephes-style reduction + minimax polynomial,
// |rel err| ~ 2e-7) so results are reproducible on any IEEE machine.  Build flags that
// make this true: -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt.
// See DESIGN.md "Canonical arithmetic".
#pragma once

#ifndef __HIPCC_RTC__  // hipRTC (specialised tapes) has the runtime built in and no host headers
#include <hip/hip_runtime.h>
#include <cstdint>
#else
using __hip_internal::uint8_t;
using __hip_internal::int32_t;
using __hip_internal::uint32_t;
using __hip_internal::uint64_t;
typedef unsigned long size_t;
#endif

#define SDF_HD __host__ __device__ __forceinline__

namespace sdf {

constexpr float kPi = 3.14159274101257324f;     // (float)pi
constexpr float kPi2 = 1.57079637050628662f;    // (float)(pi/2)
constexpr float kPi4 = 0.785398185253143311f;   // (float)(pi/4)
constexpr float k2Pi = 2.0f * kPi;

SDF_HD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
SDF_HD float sqrt_(float x) { return __builtin_sqrtf(x); }
SDF_HD float abs_(float x) { return __builtin_fabsf(x); }
SDF_HD float copysign_(float m, float s) { return __builtin_copysignf(m, s); }

// float -> int32 that is defined for every input: NaN and values outside (-2^31, 2^31) give 0 (a plain cast
// is undefined behaviour there, and x86 and gfx950 disagree about it: INT_MIN vs 0 / saturation).  The
// oracle has the same function, so even garbage in (a NaN sample point) gives the same garbage out.
SDF_HD int32_t to_int_(float v) { return (v > -2147483648.0f && v < 2147483648.0f) ? (int32_t)v : 0; }

SDF_HD float length2(float x, float y) { return sqrt_(fma_(y, y, x * x)); }
SDF_HD float length3(float x, float y, float z) { return sqrt_(fma_(z, z, fma_(y, y, x * x))); }

SDF_HD float atan_pos(float x)
{
    float y;
    if (x > 2.414213562373095f) {
        y = kPi2;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) {
        y = kPi4;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        y = 0.0f;
    }
    float z = x * x;
    float p = fma_(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fma_(p, z, 1.99777106478e-1f);
    p = fma_(p, z, -3.33329491539e-1f);
    p = p * z;
    return y + fma_(p, x, x);
}

SDF_HD float atan2_(float y, float x)
{
    float ax = abs_(x), ay = abs_(y);
    float t;
    if (ay == 0.0f)
        t = 0.0f;
    else if (ax == 0.0f)
        t = kPi2;
    else
        t = atan_pos(ay / ax);
    if (x < 0.0f) t = kPi - t;
    return copysign_(t, y);
}

SDF_HD float reduce_pio4(float ax, int32_t& j_out)
{
    int32_t j = to_int_(ax * 1.27323954473516f);
    j += (j & 1);
    float y = (float)j;
    float r = fma_(-y, 0.78515625f, ax);
    r = fma_(-y, 2.4187564849853515625e-4f, r);
    r = fma_(-y, 3.77489497744594108e-8f, r);
    j_out = j;
    return r;
}

SDF_HD float sin_poly(float r, float z)
{
    float p = fma_(-1.9515295891e-4f, z, 8.3321608736e-3f);
    p = fma_(p, z, -1.6666654611e-1f);
    p = p * z;
    return fma_(p, r, r);
}

SDF_HD float cos_poly(float z)
{
    float p = fma_(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    p = fma_(p, z, 4.166664568298827e-2f);
    p = p * z;
    return fma_(p, z, fma_(-0.5f, z, 1.0f));
}

SDF_HD void sincos_(float x, float& s, float& c)
{
    int32_t j;
    float r = reduce_pio4(abs_(x), j);
    float z = r * r;
    float ps = sin_poly(r, z);
    float pc = cos_poly(z);
    int32_t q = (j >> 1) & 3;
    float ss = (q & 1) ? pc : ps;
    float cc = (q & 1) ? ps : pc;
    if (q == 2 || q == 3) ss = -ss;
    if (q == 1 || q == 2) cc = -cc;
    if (x < 0.0f) ss = -ss;
    s = ss;
    c = cc;
}

SDF_HD float sin_(float x) { float s, c; sincos_(x, s, c); return s; }
SDF_HD float cos_(float x) { float s, c; sincos_(x, s, c); return c; }

SDF_HD float tan_(float x)
{
    int32_t j;
    float r = reduce_pio4(abs_(x), j);
    float z = r * r;
    float p = fma_(9.38540185543e-3f, z, 3.11992232697e-3f);
    p = fma_(p, z, 2.44301354525e-2f);
    p = fma_(p, z, 5.34112807005e-2f);
    p = fma_(p, z, 1.33387994085e-1f);
    p = fma_(p, z, 3.33331568548e-1f);
    p = p * z;
    float y = fma_(p, r, r);
    if (j & 2) y = -(1.0f / y);
    return (x < 0.0f) ? -y : y;
}

SDF_HD float asin_pos(float a)
{
    float x, z;
    bool flag = a > 0.5f;
    if (flag) {
        z = 0.5f * (1.0f - a);
        x = sqrt_(z);
    } else {
        x = a;
        z = x * x;
    }
    float p = fma_(4.2163199048e-2f, z, 2.4181311049e-2f);
    p = fma_(p, z, 4.5470025998e-2f);
    p = fma_(p, z, 7.4953002686e-2f);
    p = fma_(p, z, 1.6666752422e-1f);
    p = p * z;
    float r = fma_(p, x, x);
    if (flag) r = kPi2 - (r + r);
    return r;
}

SDF_HD float acos_(float x)
{
    if (x > 1.0f) x = 1.0f;
    if (x < -1.0f) x = -1.0f;
    if (x < -0.5f) {
        float t = asin_pos(sqrt_(0.5f * (1.0f + x)));
        return kPi - (t + t);
    }
    if (x > 0.5f) {
        float t = asin_pos(sqrt_(0.5f * (1.0f - x)));
        return t + t;
    }
    float t = asin_pos(abs_(x));
    return kPi2 - ((x < 0.0f) ? -t : t);
}

SDF_HD float fmod_(float x, float y)
{
    float q = __builtin_truncf(x / y);
    float r = fma_(-q, y, x);
    if (x >= 0.0f) {
        if (r < 0.0f) r = r + y;
        if (r >= y) r = r - y;
    } else {
        if (r > 0.0f) r = r - y;
        if (r <= -y) r = r + y;
    }
    return r;
}

// remainder with a pre-rounded reciprocal; inv_y == 0 (y = +inf) returns x
// (reference shapes/unsafe.py:29-31 encodes "no repetition" as an infinite spacing).
SDF_HD float remainder_inv(float x, float y, float inv_y)
{
    float n = __builtin_rintf(x * inv_y);
    float r = fma_(-n, y, x);
    return (inv_y == 0.0f) ? x : r;
}

}  // namespace sdf
