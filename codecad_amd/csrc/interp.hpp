// codecad_amd/csrc/interp.hpp
//
// The per-voxel CSG tape interpreter for gfx950 and the op library it dispatches to.
// Replaces the reference's generated evaluate() (nodes/codegen.py:5-63, handlers :91-134)
// and the *_op functions of shapes/common.cl, simple2d.cl, simple3d.cl, polygons2d.cl,
// unsafe.cl, gears.cl.  Every op performs the operation sequence fixed in DESIGN.md
// "Canonical arithmetic" (same as oracle/sdf_oracle.c) so results match the oracle under ==.
//
// MI355X mapping:
//  * the program is wave-uniform: records are fetched through the scalar cache
//    (s_load_dwordx8/x4 into SGPRs), several records per fetch group; opcode dispatch is a
//    scalar compare tree + branch, parameters are SGPR operands of the VALU ops -- no VGPRs,
//    no LDS bandwidth and no VALU cycles are spent on instruction fetch/decode;
//  * the value registers (`registers[secondaryRegister]`, dynamically indexed, so they
//    cannot live in VGPRs) are per-lane float4 slots in LDS laid out [reg][lane]:
//    one ds_write_b128 / ds_read_b128 per access, consecutive lanes 16 B apart
//    (conflict-free); the reference keeps a fixed 8 KiB private array per work-item.
#pragma once

#include "tape.hpp"

namespace sdf {

__device__ __forceinline__ float4 f4(float x, float y, float z, float w) { return make_float4(x, y, z, w); }
__device__ __forceinline__ float4 neg(float4 a) { return f4(-a.x, -a.y, -a.z, -a.w); }
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
    return fma_(az, bz, fma_(ay, by, ax * bx));
}
__device__ __forceinline__ float dot2(float ax, float ay, float bx, float by) { return fma_(ay, by, ax * bx); }

// reference shapes/common.cl:1-6; k = w*w - dot(v,v) is folded at decode time
__device__ __forceinline__ void quat_xform(float qx, float qy, float qz, float qw, float k,
                                           float px, float py, float pz,
                                           float& ox, float& oy, float& oz)
{
    float d = dot3(qx, qy, qz, px, py, pz);
    float cx = fma_(qy, pz, -(qz * py));
    float cy = fma_(qz, px, -(qx * pz));
    float cz = fma_(qx, py, -(qy * px));
    float tx = fma_(cx, qw, qx * d);
    float ty = fma_(cy, qw, qy * d);
    float tz = fma_(cz, qw, qz * d);
    ox = fma_(px, k, tx + tx);
    oy = fma_(py, k, ty + ty);
    oz = fma_(pz, k, tz + tz);
}

// reference shapes/common.cl:15-31
__device__ __forceinline__ float4 perp_intersection(float4 a, float4 b)
{
    if (a.w > 0.0f && b.w > 0.0f) {
        float dist = length2(a.w, b.w);
        float inv = 1.0f / dist;
        float m1 = a.w * inv, m2 = b.w * inv;
        return f4(fma_(b.x, m2, a.x * m1), fma_(b.y, m2, a.y * m1), fma_(b.z, m2, a.z * m1), dist);
    }
    return (a.w > b.w) ? a : b;
}

// reference shapes/simple2d.cl:1-4 (slab_x/slab_y of common.cl:33-39 inlined).  Equal to
// perp_intersection(slab_x, slab_y) under ==: the zero components only drop exact zeros.
// Written with selects, not branches: a divergent branch inside the dispatch loop makes the
// compiler structurize the WHOLE loop (flag registers, phi copies, ~4x instruction bloat).
__device__ __forceinline__ float4 rectangle_op(float hw, float hh, float4 c)
{
    float sx = copysign_(1.0f, c.x), sy = copysign_(1.0f, c.y);
    float wx = abs_(c.x) - hw, wy = abs_(c.y) - hh;
    float dist = length2(wx, wy);
    float inv = 1.0f / dist;
    bool corner = (wx > 0.0f) & (wy > 0.0f);
    bool xs = wx > wy;
    return f4(corner ? sx * (wx * inv) : (xs ? sx : 0.0f), corner ? sy * (wy * inv) : (xs ? 0.0f : sy), 0.0f,
              corner ? dist : (xs ? wx : wy));
}

// Distance-only forms (DISTANCE_ONLY interpreter): the same operations that produce .w above,
// nothing else.  Valid for tapes without rounded blends, where no direction ever feeds a
// distance (tape.hpp: direction_feeds_distance).
__device__ __forceinline__ float perp_w(float a, float b)
{
    float dist = length2(a, b);
    bool corner = (a > 0.0f) & (b > 0.0f);
    return corner ? dist : ((a > b) ? a : b);
}

// reference shapes/simple3d.cl:18-21 = perp_intersection(slab_z(h, coords), in)
__device__ __forceinline__ float4 extrusion_op(float hh, float4 in, float4 coords)
{
    float sz = copysign_(1.0f, coords.z);
    float wz = abs_(coords.z) - hh;
    float dist = length2(wz, in.w);
    float inv = 1.0f / dist;
    float m1 = wz * inv, m2 = in.w * inv;
    bool corner = (wz > 0.0f) & (in.w > 0.0f);
    bool cap = wz > in.w;
    return f4(corner ? in.x * m2 : (cap ? 0.0f : in.x), corner ? in.y * m2 : (cap ? 0.0f : in.y),
              corner ? fma_(in.z, m2, sz * m1) : (cap ? sz : in.z), corner ? dist : (cap ? wz : in.w));
}

// reference shapes/common.cl:45-64
__device__ __forceinline__ float4 select4(bool c, float4 a, float4 b)
{
    return f4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w);
}

__device__ __noinline__ float4 rounded_blend(float r, float4 a, float4 b)
{
    float cos_alpha = dot3(a.x, a.y, a.z, b.x, b.y, b.z);
    float x1 = r - a.w, x2 = r - b.w;
    if (cos_alpha * x1 < x2 && cos_alpha * x2 < x1) {
        float num = fma_(-((2.0f * cos_alpha) * x1), x2, fma_(x2, x2, x1 * x1));
        float den = fma_(-cos_alpha, cos_alpha, 1.0f);
        return f4(0.0f, 0.0f, 0.0f, r - sqrt_(num / den));
    }
    return (a.w < b.w) ? a : b;
}

__device__ __forceinline__ float4 rounded_union(float r, float4 a, float4 b)
{
    if (r >= 0.0f) return rounded_blend(r, a, b);  // wave-uniform: r is a tape constant
    return select4(a.w < b.w, a, b);
}

// reference shapes/simple2d.cl:6-14
__device__ __forceinline__ float4 circle_op(float r, float4 c)
{
    float a = length2(c.x, c.y);
    float inv = 1.0f / a;
    bool zero = (a == 0.0f);
    return f4(zero ? 1.0f : c.x * inv, zero ? 0.0f : c.y * inv, 0.0f, a - r);
}

// reference shapes/simple3d.cl:1-12
__device__ __forceinline__ float4 sphere_op(float r, float4 c)
{
    float a = length3(c.x, c.y, c.z);
    float inv = 1.0f / a;
    bool zero = (a == 0.0f);
    return f4(zero ? 1.0f : c.x * inv, zero ? 0.0f : c.y * inv, zero ? 0.0f : c.z * inv, a - r);
}

__device__ __forceinline__ float sign_f(float s) { return (s > 0.0f) ? 1.0f : ((s < 0.0f) ? -1.0f : 0.0f); }

__device__ __forceinline__ float sector_alpha(float y, float x, float pi_over_n)
{
    return (atan2_(y, x) + 2.0f * kPi) + pi_over_n;
}

// reference shapes/simple2d.cl:16-46; p = {pi/n, r, 2pi/n, r*sin(pi/n), -(r*cos(pi/n))}
__device__ __noinline__ float4 regular_polygon2d_op(float pi_over_n, float r, float two_pi_over_n, float r_sin,
                                                     float neg_r_cos, float4 c)
{
    float len = length2(c.x, c.y);
    float alpha = sector_alpha(c.y, c.x, pi_over_n);
    int side = (int)__builtin_floorf(alpha / two_pi_over_n);
    float side2 = (float)(side * 2);
    float mod_alpha = (alpha - side2 * pi_over_n) - pi_over_n;
    float s, co;
    sincos_(mod_alpha, s, co);
    if (abs_(s * len) > r_sin) {
        float ny, nx;
        sincos_(fma_(sign_f(s), pi_over_n, side2 * pi_over_n), ny, nx);
        nx = nx * r;
        ny = ny * r;
        float dx = c.x - nx, dy = c.y - ny;
        float dist = length2(dx, dy);
        if (dist > 0.0f) {
            float inv = 1.0f / dist;
            return f4(dx * inv, dy * inv, 0.0f, dist);
        }
    }
    float dy, dx;
    sincos_(side2 * pi_over_n, dy, dx);
    return f4(dx, dy, 0.0f, fma_(len, co, neg_r_cos));
}

// reference shapes/polygons2d.cl:1-74; pts = n (x, y) pairs, wave-uniform (scalar loads)
__device__ __noinline__ float4 polygon2d_op(const float* __restrict__ pts, uint32_t n, float4 coords)
{
    float qx = coords.x, qy = coords.y;
    float nnx = 0.0f, nny = 0.0f;
    float nearest_d2 = __builtin_inff();
    bool nearest_is_vertex = false;
    float outside = 1.0f;
    float cx = pts[2 * (n - 1)], cy = pts[2 * (n - 1) + 1];
    for (uint32_t i = 0; i < n; ++i) {
        float px = cx, py = cy;
        cx = pts[2 * i];
        cy = pts[2 * i + 1];
        float dx = cx - px, dy = cy - py;
        float tqx = qx - px, tqy = qy - py;
        float snx = -dy, sny = dx;
        if (((py < qy) != (cy < qy)) && (dy * dot2(snx, sny, tqx, tqy) > 0.0f)) outside = -outside;
        float t = dot2(dx, dy, tqx, tqy) / dot2(dx, dy, dx, dy);
        if (t > 1.0f) continue;
        float cnx, cny, cd2;
        bool cvert;
        if (t >= 0.0f) {
            float tcx = fma_(-t, dx, tqx), tcy = fma_(-t, dy, tqy);
            cd2 = dot2(tcx, tcy, tcx, tcy);
            cnx = snx;
            cny = sny;
            cvert = false;
        } else {
            cnx = tqx;
            cny = tqy;
            cd2 = dot2(cnx, cny, cnx, cny);
            cvert = cd2 > 1.1920928955078125e-7f;  // FLT_EPSILON
            if (!cvert) {
                cnx = snx;
                cny = sny;
            }
        }
        if (cd2 < nearest_d2) {
            nearest_d2 = cd2;
            nnx = cnx;
            nny = cny;
            nearest_is_vertex = cvert;
        }
    }
    float distance = outside * sqrt_(nearest_d2);
    float inv = 1.0f / (nearest_is_vertex ? distance : length2(nnx, nny));
    return f4(nnx * inv, nny * inv, 0.0f, distance);
}

// reference shapes/simple3d.cl:28-39
__device__ __forceinline__ float4 revolution_from_op(float4 flat, float4 coords)
{
    float len = length2(coords.x, coords.z);
    bool zero = (len == 0.0f);
    float m = zero ? flat.x : flat.x / len;
    float cxx = zero ? 1.0f : coords.x;
    return f4(cxx * m, flat.y, coords.z * m, flat.w);
}

__device__ __forceinline__ void rot2(float c, float s, float px, float py, float& ox, float& oy)
{
    ox = fma_(c, px, -(s * py));
    oy = fma_(s, px, c * py);
}

// reference shapes/simple3d.cl:42-51
__device__ __noinline__ float4 twist_revolution_to_op(float r, float twist, float4 c)
{
    float alpha = fmod_(atan2_(c.z, c.x) + kPi, k2Pi);
    float beta = (twist * alpha) / k2Pi;
    float axis = length2(c.x, c.z);
    float s, co, ox, oy;
    sincos_(-beta, s, co);
    rot2(co, s, axis - r, c.y, ox, oy);
    return f4(ox, oy, 0.0f, 0.0f);
}

// reference shapes/simple3d.cl:53-97; p = {minorR, r, twist, 0.05*r, r-minorR, min(1, lipschitz)}
__device__ __noinline__ float4 twist_revolution_from_op(float minor_r, float r, float twist, float padding,
                                                         float r_minus_minor, float lipschitz, float4 res, float4 c)
{
    float axis = length2(c.x, c.z);
    float ipx = axis - r, ipy = c.y;
    float center = length2(ipx, ipy);
    float wrapper = center - minor_r;
    float bound, dx, dy;
    if (axis == 0.0f) return f4(1.0f, 0.0f, 0.0f, r_minus_minor);
    if (wrapper > padding) {
        bound = wrapper;
        float inv = 1.0f / center;
        dx = ipx * inv;
        dy = ipy * inv;
    } else {
        float alpha = fmod_(atan2_(c.z, c.x) + kPi, k2Pi);
        float beta = (twist * alpha) / k2Pi;
        bound = res.w * lipschitz;
        float s, co;
        sincos_(beta, s, co);
        rot2(co, s, res.x, res.y, dx, dy);
    }
    float m = dx / axis;
    return f4(c.x * m, dy, c.z * m, bound);
}

// reference shapes/unsafe.cl:8-15; p = {pi/n, 2pi/n}
__device__ __noinline__ float4 circular_repetition_to_op(float pi_over_n, float two_pi_over_n, float4 c)
{
    const float p[2] = {pi_over_n, two_pi_over_n};
    float len = length2(c.x, c.y);
    float alpha = sector_alpha(c.y, c.x, p[0]);
    int side = (int)__builtin_floorf(alpha / p[1]);
    float mod_alpha = (alpha - (float)(side * 2) * p[0]) - p[0];
    float s, co;
    sincos_(mod_alpha, s, co);
    return f4(len * co, len * s, c.z, 0.0f);
}

// reference shapes/unsafe.cl:17-23
__device__ __noinline__ float4 circular_repetition_from_op(float pi_over_n, float two_pi_over_n, float4 dist, float4 c)
{
    const float p[2] = {pi_over_n, two_pi_over_n};
    float alpha = sector_alpha(c.y, c.x, p[0]);
    int side = (int)__builtin_floorf(alpha / p[1]);
    float s, co, ox, oy;
    sincos_((float)(side * 2) * p[0], s, co);
    rot2(co, s, dist.x, dist.y, ox, oy);
    return f4(ox, oy, dist.z, dist.w);
}

// reference shapes/gears.cl:1-42; p = {teeth, pressure angle, base radius, tooth angle,
// half tooth base angle, 2*tooth angle, -(base radius)^2}
__device__ __noinline__ float4 involute_gear_op(float base_radius, float tooth_angle, float half_tooth_base,
                                                 float two_tooth_angle, float neg_base_sq, float4 c)
{
    float len = length2(c.x, c.y);
    float alpha = atan2_(c.y, c.x);
    float wrapped = fmod_(alpha + 2.0f * kPi, two_tooth_angle);
    float involute_alpha = half_tooth_base - abs_(wrapped - tooth_angle);
    if (len < base_radius) {
        float nx = c.y / len, ny = -c.x / len;
        if (wrapped > tooth_angle) {
            nx = -nx;
            ny = -ny;
        }
        float angular = abs_(wrapped - tooth_angle) - half_tooth_base;
        return f4(nx, ny, 0.0f, angular * len);
    }
    float phi = involute_alpha + acos_(base_radius / len);
    float normal_angle = (wrapped < tooth_angle) ? (kPi - phi) - (alpha - involute_alpha)
                                                 : phi - (alpha - involute_alpha);
    float nx, ny;
    sincos_(normal_angle, nx, ny);
    float distance = sqrt_(fma_(len, len, neg_base_sq)) - base_radius * phi;
    return f4(nx, ny, 0.0f, distance);
}

// ---------------------------------------------------------------------------------------
// The interpreter.  `regs` points at this lane's slot of register 0 in LDS; register r is
// regs[r * stride] (stride = lanes per workgroup).  `prog` and `extra` are wave-uniform.
//
// exec_one runs ONE decoded record; it returns true on _return.  run_tape fetches records
// in groups of kFetchGroup: all scalar loads of a group are issued together at the top of
// the loop and waited for once, so the scalar-cache latency is paid once per group instead
// of once per tape instruction (SMEM returns out of order, so a wave can only wait for
// "everything").  The build disables MachineSink (builder.py) so that the loads stay there.
// ---------------------------------------------------------------------------------------
#ifndef SDF_FETCH_GROUP
#define SDF_FETCH_GROUP 4
#endif
constexpr int kFetchGroup = SDF_FETCH_GROUP;  // tape.hpp pads the program with kFetchGroup _return records

template <bool DISTANCE_ONLY>
__device__ __forceinline__ bool exec_one(const Rec& cur, float4& last, const float* __restrict__ extra,
                                         float px, float py, float pz, float4* regs, uint32_t stride)
{
    const uint32_t op = cur.hdr & 0xffu;
    const uint32_t reg = cur.hdr >> 8;
    const float* p = cur.p;
    switch (op) {
    case OP_RETURN: return true;
    case OP_STORE: regs[reg * stride] = last; break;
    case OP_LOAD: last = regs[reg * stride]; break;
    case OP_RECTANGLE:
        if (DISTANCE_ONLY) last.w = perp_w(abs_(last.x) - p[0], abs_(last.y) - p[1]);
        else last = rectangle_op(p[0], p[1], last);
        break;
    case OP_CIRCLE:
        if (DISTANCE_ONLY) last.w = length2(last.x, last.y) - p[0];
        else last = circle_op(p[0], last);
        break;
    case OP_REGULAR_POLYGON2D: last = regular_polygon2d_op(p[0], p[1], p[2], p[3], p[4], last); break;
    case OP_POLYGON2D:
        last = polygon2d_op(extra + __float_as_uint(p[1]), __float_as_uint(p[0]), last);
        break;
    case OP_SPHERE:
        if (DISTANCE_ONLY) last.w = length3(last.x, last.y, last.z) - p[0];
        else last = sphere_op(p[0], last);
        break;
    case OP_HALF_SPACE:
        if (DISTANCE_ONLY) last.w = -last.y;
        else last = f4(0.0f, -1.0f, 0.0f, -last.y);
        break;
    case OP_REVOLUTION_TO: last = f4(length2(last.x, last.z), last.y, 0.0f, 0.0f); break;
    case OP_TWIST_REVOLUTION_TO: last = twist_revolution_to_op(p[0], p[1], last); break;
    case OP_INITIAL_TRANSFORMATION_TO: {
        float ox, oy, oz;
        quat_xform(p[0], p[1], p[2], p[3], p[7], px, py, pz, ox, oy, oz);
        last = f4(ox + p[4], oy + p[5], oz + p[6], 0.0f);
        break;
    }
    case OP_TRANSFORMATION_TO: {
        float ox, oy, oz;
        quat_xform(p[0], p[1], p[2], p[3], p[7], last.x, last.y, last.z, ox, oy, oz);
        last = f4(ox + p[4], oy + p[5], oz + p[6], 0.0f);
        break;
    }
    case OP_TRANSFORMATION_FROM: {
        if (DISTANCE_ONLY) {
            last.w = last.w * p[5];
            break;
        }
        float ox, oy, oz;
        quat_xform(p[0], p[1], p[2], p[3], p[4], last.x, last.y, last.z, ox, oy, oz);
        last = f4(ox * p[6], oy * p[6], oz * p[6], last.w * p[5]);
        break;
    }
    case OP_MIRROR: last.x = -last.x; break;
    case OP_SYMMETRICAL_TO: last.x = abs_(last.x); break;
    case OP_OFFSET: last.w = last.w - p[0]; break;
    case OP_SHELL: {
        if (DISTANCE_ONLY) {
            last.w = ((last.w >= 0.0f) ? last.w : -last.w) - p[0];
            break;
        }
        float4 s = select4(last.w >= 0.0f, last, neg(last));
        last = f4(s.x, s.y, s.z, s.w - p[0]);
        break;
    }
    case OP_REPETITION:
        last = f4(remainder_inv(last.x, p[0], p[3]), remainder_inv(last.y, p[1], p[4]),
                  remainder_inv(last.z, p[2], p[5]), 0.0f);
        break;
    case OP_CIRCULAR_REPETITION_TO: last = circular_repetition_to_op(p[0], p[1], last); break;
    case OP_CIRCULAR_REPETITION_FROM:
        if (!DISTANCE_ONLY) last = circular_repetition_from_op(p[0], p[1], last, regs[reg * stride]);
        break;
    case OP_INVOLUTE_GEAR: last = involute_gear_op(p[2], p[3], p[4], p[5], p[6], last); break;
    case OP_EXTRUSION:
        if (DISTANCE_ONLY) last.w = perp_w(abs_(regs[reg * stride].z) - p[0], last.w);
        else last = extrusion_op(p[0], last, regs[reg * stride]);
        break;
    case OP_REVOLUTION_FROM:
        if (!DISTANCE_ONLY) last = revolution_from_op(last, regs[reg * stride]);
        break;
    case OP_TWIST_REVOLUTION_FROM:
        last = twist_revolution_from_op(p[0], p[1], p[2], p[3], p[4], p[5], last, regs[reg * stride]);
        break;
    case OP_SYMMETRICAL_FROM: {
        if (DISTANCE_ONLY) break;
        float4 pt = regs[reg * stride];
        last.x = (pt.x < 0.0f) ? -last.x : last.x;
        break;
    }
    case OP_UNION:
        if (DISTANCE_ONLY) { float b = regs[reg * stride].w; last.w = (last.w < b) ? last.w : b; }
        else last = rounded_union(p[0], last, regs[reg * stride]);
        break;
    case OP_INTERSECTION:
        if (DISTANCE_ONLY) { float a = -last.w, b = -regs[reg * stride].w; last.w = -((a < b) ? a : b); }
        else last = neg(rounded_union(p[0], neg(last), neg(regs[reg * stride])));
        break;
    case OP_SUBTRACTION:
        if (DISTANCE_ONLY) { float a = -last.w, b = regs[reg * stride].w; last.w = -((a < b) ? a : b); }
        else last = neg(rounded_union(p[0], neg(last), regs[reg * stride]));
        break;
    default: return true;  // unreachable: tapes are validated at upload
    }
    return false;
}

template <bool DISTANCE_ONLY = false>
__device__ __forceinline__ float4 run_tape(const Rec* __restrict__ prog,
                                           const float* __restrict__ extra, float px, float py,
                                           float pz, float4* regs, uint32_t stride)
{
    float4 last = f4(0.0f, 0.0f, 0.0f, 0.0f);
    const Rec* pc = prog;
    for (;;) {
        Rec group[kFetchGroup];
#pragma unroll
        for (int k = 0; k < kFetchGroup; ++k) group[k] = pc[k];
        pc += kFetchGroup;
#pragma unroll
        for (int k = 0; k < kFetchGroup; ++k)
            if (exec_one<DISTANCE_ONLY>(group[k], last, extra, px, py, pz, regs, stride)) return last;
    }
}

}  // namespace sdf
