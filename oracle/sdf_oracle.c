/* oracle/sdf_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's hot path (bluecube/codecad): the tape
 * interpreter evaluate(), the 26 *_op device functions and the four kernels around
 * them.  Every function cites the reference file:line it follows (paths relative to
 * /root/reference/codecad/).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (codecad_amd/) never does.
 *
 * Pinning: the reference's OpenCL device code cannot be built in this image (it needs
 * an OpenCL runtime + its builtin library, both absent; we do not fake them), so this
 * restatement is pinned by the reference's OWN tests and fixtures restated in tests/:
 * analytic mass properties (reference tests/test_mass_properties.py:16-108), leaf-block
 * known answers (tests/test_subdivision.py:110-161), DSDF validity properties
 * (tests/test_dsdf.py:113-190), and by golden tapes / block tables produced by the
 * reference's pure-Python half (tests/golden/gen/make_golden.py).
 *
 * Canonical arithmetic (DESIGN.md section "Canonical arithmetic"): IEEE-754 binary32,
 * round-to-nearest-even, no implicit contraction (-ffp-contract=off), explicit fmaf()
 * exactly where written, IEEE sqrt and divide, elementary functions from det_math.h.
 * The reference is compiled with -cl-fast-relaxed-math (cl_util/opencl_manager.py:
 * 12-18), which permits exactly these liberties (fused multiply-add, x/u -> x*(1/u)
 * for a tape-constant u); the HIP kernels perform the same operation sequence.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "det_math.h"

#define EVAL_REGISTER_COUNT 512 /* nodes/__init__.py:6 */

typedef struct { float x, y, z, w; } f4;

static inline f4 mk4(float x, float y, float z, float w) { f4 r = { x, y, z, w }; return r; }
static inline f4 neg4(f4 a) { return mk4(-a.x, -a.y, -a.z, -a.w); }
static inline float dot3(f4 a, f4 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline float dot2(float ax, float ay, float bx, float by) { return fmaf(ay, by, ax * bx); }

/* shapes/common.cl:1-6 quaternion_transform:
 *   (v*dot(v,p) + cross(v,p)*w)*2 + p*(w*w - dot(v,v)) */
static inline f4 quaternion_transform(f4 q, f4 p)
{
    float d = dot3(q, p);
    float cx = fmaf(q.y, p.z, -(q.z * p.y));
    float cy = fmaf(q.z, p.x, -(q.x * p.z));
    float cz = fmaf(q.x, p.y, -(q.y * p.x));
    float k = fmaf(q.w, q.w, -dot3(q, q));
    float tx = fmaf(cx, q.w, q.x * d);
    float ty = fmaf(cy, q.w, q.y * d);
    float tz = fmaf(cz, q.w, q.z * d);
    return mk4(fmaf(p.x, k, tx + tx), fmaf(p.y, k, ty + ty), fmaf(p.z, k, tz + tz), 0.0f);
}

/* shapes/common.cl:8-11 */
static inline float quaternion_scale(f4 q)
{
    return fmaf(q.w, q.w, fmaf(q.z, q.z, fmaf(q.y, q.y, q.x * q.x)));
}

/* gfx9 v_min_f32 / v_max_f32 as the ISA documents them: a NaN operand yields the other operand, -0 orders below
 * +0.  The canonical distance of union / intersection / subtraction and of the nearer-slab case (DESIGN.md
 * section 3): the kernels compute it with that one instruction.  It differs from `a < b ? a : b` only for NaN
 * operands and for zeros of opposite sign. */
static inline float hw_min(float a, float b)
{
    if (a < b) return a;
    if (b < a) return b;
    if (a != a) return b;              /* unordered: the operand that is not NaN */
    if (b != b) return a;
    return signbit(a) ? a : b;         /* equal: -0 before +0 */
}
static inline float hw_max(float a, float b)
{
    if (a > b) return a;
    if (b > a) return b;
    if (a != a) return b;
    if (b != b) return a;
    return signbit(a) ? b : a;
}

/* shapes/common.cl:15-31 perpendicular_intersection (outside the corner: the direction of the nearer slab, the
 * distance as the hardware maximum) */
static inline f4 perpendicular_intersection(f4 a, f4 b)
{
    if (a.w > 0.0f && b.w > 0.0f) {
        float dist = dm_hypot(a.w, b.w);
        float inv = 1.0f / dist;
        float m1 = a.w * inv;
        float m2 = b.w * inv;
        return mk4(fmaf(b.x, m2, a.x * m1), fmaf(b.y, m2, a.y * m1), fmaf(b.z, m2, a.z * m1), dist);
    } else {
        f4 r = (a.w > b.w) ? a : b;
        r.w = hw_max(a.w, b.w);
        return r;
    }
}

/* shapes/common.cl:33-43 */
static inline f4 slab_x(float h, f4 p) { return mk4(copysignf(1.0f, p.x), 0, 0, fabsf(p.x) - h); }
static inline f4 slab_y(float h, f4 p) { return mk4(0, copysignf(1.0f, p.y), 0, fabsf(p.y) - h); }
static inline f4 slab_z(float h, f4 p) { return mk4(0, 0, copysignf(1.0f, p.z), fabsf(p.z) - h); }

/* shapes/common.cl:45-64 rounded_union */
static inline f4 rounded_union(float r, f4 a, f4 b)
{
    if (r >= 0.0f) {
        float cos_alpha = dot3(a, b);
        float x1 = r - a.w;
        float x2 = r - b.w;
        if (cos_alpha * x1 < x2 && cos_alpha * x2 < x1) {
            float num = fmaf(-((2.0f * cos_alpha) * x1), x2, fmaf(x2, x2, x1 * x1));
            float den = fmaf(-cos_alpha, cos_alpha, 1.0f);
            float d = r - sqrtf(num / den);
            return mk4(0, 0, 0, d);
        }
    }
    if (r >= 0.0f) return (a.w < b.w) ? a : b;
    f4 nearer = (a.w < b.w) ? a : b;   /* plain union: the direction of the nearer one, the distance as the hardware minimum */
    nearer.w = hw_min(a.w, b.w);
    return nearer;
}

/* shapes/common.cl:66-76 */
static inline f4 union_op(float r, f4 a, f4 b) { return rounded_union(r, a, b); }
static inline f4 intersection_op(float r, f4 a, f4 b) { return neg4(rounded_union(r, neg4(a), neg4(b))); }
static inline f4 subtraction_op(float r, f4 a, f4 b) { return neg4(rounded_union(r, neg4(a), b)); }

/* Rotation about one coordinate axis (quaternion with a single non-zero vector component q, scalar part w,
 * not normalised: |Q|^2 is the transform's scale) written as the 2x2 rotation-and-scale it is:
 *   along the axis  A = w^2 + q^2,   in the plane  B = w^2 - q^2 (0 for a quarter turn),  C = 2 q w,
 * folded in double (every product is exact there) and rounded once; div = 1 for transformation_to, |Q|^2
 * for transformation_from.  Same folding as the kernels' tape decoder (codecad_amd/csrc/tape.hpp). */
static inline void axis_constants(float q, float w, double div, float *A, float *B, float *C)
{
    const double qq = (double)q * (double)q, ww = (double)w * (double)w;
    *A = (float)((ww + qq) / div);
    *B = (float)((ww - qq) / div);
    *C = (float)((2.0 * ((double)q * (double)w)) / div);
}

/* (along, u, v) = the coordinate on the axis and the other two in cyclic order */
static inline void axis_rotate(float A, float B, float C, float along, float u, float v, float oa, float ou, float ov,
                               float *ra, float *ru, float *rv)
{
    *ra = fmaf(along, A, oa);
    *ru = fmaf(-v, C, ou);
    *rv = fmaf(u, C, ov);
    if (B != 0.0f) {
        *ru = fmaf(u, B, *ru);
        *rv = fmaf(v, B, *rv);
    }
}

static inline void axis_rotate_dir(float A, float B, float C, float along, float u, float v, float *ra, float *ru, float *rv)
{
    *ra = along * A;
    *ru = (-v) * C;
    *rv = u * C;
    if (B != 0.0f) {
        *ru = fmaf(u, B, *ru);
        *rv = fmaf(v, B, *rv);
    }
}

/* A general quaternion Q = (x, y, z, w), not normalised, as its matrix R = (w^2 - |v|^2) I + 2 v v^T + 2 w [v]x,
 * m[3*row + column], folded in double in exactly this order and rounded once (the kernels' decoder:
 * codecad_amd/csrc/tape.hpp matrix_constants). */
static inline void matrix_constants(f4 q, double div, float m[9])
{
    const double x = q.x, y = q.y, z = q.z, w = q.w;
    const double xx = x * x, yy = y * y, zz = z * z, ww = w * w;
    const double xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
    const double k = ww - ((xx + yy) + zz);
    const double e[9] = {k + 2.0 * xx,     2.0 * (xy - wz),  2.0 * (xz + wy),
                         2.0 * (xy + wz),  k + 2.0 * yy,     2.0 * (yz - wx),
                         2.0 * (xz - wy),  2.0 * (yz + wx),  k + 2.0 * zz};
    for (int i = 0; i < 9; ++i) m[i] = (float)(e[i] / div);
}

/* shapes/common.cl:78-98 (initial_)transformation_to_op.
 * Canonical arithmetic: a zero offset component counts as +0, so a transformed
 * coordinate is never -0 (the reference builds with -cl-no-signed-zeros and cannot tell); and when the
 * quaternion's vector part is zero (a scaling) or has a single non-zero component (a rotation about a
 * coordinate axis) the transform is evaluated as the scaling / 2x2 rotation it is, with constants folded
 * from the quaternion (axis_constants) -- the same linear map as the quaternion sandwich, each output one
 * or two fused multiply-adds instead of its ~20 roundings (difference ~1e-7 relative). */
static inline f4 transformation_to_op(const float *p, f4 point)
{
    const float ox = p[4] + 0.0f, oy = p[5] + 0.0f, oz = p[6] + 0.0f;
    const int zx = p[0] == 0.0f, zy = p[1] == 0.0f, zz = p[2] == 0.0f;
    f4 q = mk4(p[0], p[1], p[2], p[3]);
    if (!(zx + zy + zz >= 2)) {   /* general: the matrix, one fma chain per row */
        float m[9];
        matrix_constants(q, 1.0, m);
        return mk4(fmaf(point.x, m[0], fmaf(point.y, m[1], fmaf(point.z, m[2], ox))),
                   fmaf(point.x, m[3], fmaf(point.y, m[4], fmaf(point.z, m[5], oy))),
                   fmaf(point.x, m[6], fmaf(point.y, m[7], fmaf(point.z, m[8], oz))), 0.0f);
    }
    float A, B, C, x, y, z;
    if (zx && zy && zz) {
        axis_constants(0.0f, q.w, 1.0, &A, &B, &C);
        return mk4(fmaf(point.x, A, ox), fmaf(point.y, A, oy), fmaf(point.z, A, oz), 0.0f);
    }
    if (zy && zz) {          /* rotation about x */
        axis_constants(q.x, q.w, 1.0, &A, &B, &C);
        axis_rotate(A, B, C, point.x, point.y, point.z, ox, oy, oz, &x, &y, &z);
    } else if (zx && zz) {   /* about y */
        axis_constants(q.y, q.w, 1.0, &A, &B, &C);
        axis_rotate(A, B, C, point.y, point.z, point.x, oy, oz, ox, &y, &z, &x);
    } else {                 /* about z */
        axis_constants(q.z, q.w, 1.0, &A, &B, &C);
        axis_rotate(A, B, C, point.z, point.x, point.y, oz, ox, oy, &z, &x, &y);
    }
    return mk4(x, y, z, 0.0f);
}

/* shapes/common.cl:100-110 transformation_from_op.
 * Canonical arithmetic, like transformation_to_op: scalings and rotations about a
 * coordinate axis use the folded 2x2 form, constants divided by |Q|^2 so that unit directions stay unit, with
 * plain products (a component keeps the sign its source had).  The sign of a zero component is observable: a
 * rounded blend returns the direction (0, 0, 0), and the ray caster divides by dot(normal, ray). */
static inline f4 transformation_from_op(const float *p, f4 in)
{
    f4 q = mk4(p[0], p[1], p[2], p[3]);
    float scale = quaternion_scale(q);
    const int zx = p[0] == 0.0f, zy = p[1] == 0.0f, zz = p[2] == 0.0f;
    if (!(zx + zy + zz >= 2)) {   /* general: the matrix over |Q|^2 */
        float m[9];
        matrix_constants(q, (double)scale, m);
        return mk4(fmaf(in.x, m[0], fmaf(in.y, m[1], in.z * m[2])),
                   fmaf(in.x, m[3], fmaf(in.y, m[4], in.z * m[5])),
                   fmaf(in.x, m[6], fmaf(in.y, m[7], in.z * m[8])), in.w * scale);
    }
    float A, B, C, x, y, z;
    if (zx && zy && zz) {
        axis_constants(0.0f, q.w, (double)scale, &A, &B, &C);
        return mk4(in.x * A, in.y * A, in.z * A, in.w * scale);
    }
    if (zy && zz) {          /* rotation about x */
        axis_constants(q.x, q.w, (double)scale, &A, &B, &C);
        axis_rotate_dir(A, B, C, in.x, in.y, in.z, &x, &y, &z);
    } else if (zx && zz) {   /* about y */
        axis_constants(q.y, q.w, (double)scale, &A, &B, &C);
        axis_rotate_dir(A, B, C, in.y, in.z, in.x, &y, &z, &x);
    } else {                 /* about z */
        axis_constants(q.z, q.w, (double)scale, &A, &B, &C);
        axis_rotate_dir(A, B, C, in.z, in.x, in.y, &z, &x, &y);
    }
    return mk4(x, y, z, in.w * scale);
}

/* shapes/common.cl:112-131 */
static inline f4 mirror_op(f4 in) { return mk4(-in.x, in.y, in.z, in.w); }
static inline f4 symmetrical_to_op(f4 p) { return mk4(fabsf(p.x), p.y, p.z, p.w); }
static inline f4 symmetrical_from_op(f4 in, f4 point)
{
    return mk4(point.x < 0.0f ? -in.x : in.x, in.y, in.z, in.w);
}
static inline f4 offset_op(float d, f4 in) { return mk4(in.x, in.y, in.z, in.w - d); }
static inline f4 shell_op(float h, f4 in)
{
    f4 s = (in.w >= 0.0f) ? in : neg4(in);
    return offset_op(h, s);
}

/* shapes/simple2d.cl:1-4 */
static inline f4 rectangle_op(float hw, float hh, f4 c)
{
    return perpendicular_intersection(slab_x(hw, c), slab_y(hh, c));
}

/* shapes/simple2d.cl:6-14 */
static inline f4 circle_op(float r, f4 c)
{
    float a = dm_length2(c.x, c.y);
    float fx, fy;
    if (a == 0.0f) {
        fx = 1.0f; fy = 0.0f;
    } else {
        float inv = 1.0f / a;
        fx = c.x * inv; fy = c.y * inv;
    }
    return mk4(fx, fy, 0.0f, a - r);
}

static inline float sign_f(float s) { return (s > 0.0f) ? 1.0f : ((s < 0.0f) ? -1.0f : 0.0f); }

/* the "which sector" prologue shared by simple2d.cl:18-20 and unsafe.cl:10-12,18-19 */
static inline float sector_alpha(float y, float x, float pi_over_n)
{
    return (dm_atan2(y, x) + 2.0f * DM_PI_F) + pi_over_n;
}

/* shapes/simple2d.cl:16-46 */
static inline f4 regular_polygon2d_op(float pi_over_n, float r, f4 c)
{
    float len = dm_hypot(c.x, c.y);
    float alpha = sector_alpha(c.y, c.x, pi_over_n);
    int side = dm_to_int(floorf(alpha / (2.0f * pi_over_n)));
    float side2 = (2.0f * (float)side);
    float mod_alpha = (alpha - side2 * pi_over_n) - pi_over_n;
    float s, co;
    dm_sincos(mod_alpha, &s, &co);
    if (fabsf(s * len) > r * dm_sin(pi_over_n)) {
        float ny, nx;
        dm_sincos(fmaf(sign_f(s), pi_over_n, side2 * pi_over_n), &ny, &nx);
        nx = nx * r; ny = ny * r;
        float dx = c.x - nx, dy = c.y - ny;
        float dist = dm_length2(dx, dy);
        if (dist > 0.0f) {
            float inv = 1.0f / dist;
            return mk4(dx * inv, dy * inv, 0.0f, dist);
        }
    }
    float dy, dx;
    dm_sincos(side2 * pi_over_n, &dy, &dx);
    return mk4(dx, dy, 0.0f, fmaf(len, co, -(r * dm_cos(pi_over_n))));
}

/* shapes/polygons2d.cl:1-74; *pp points at [n, x0,y0, ...] and is advanced past it */
static inline f4 polygon2d_op(const float **pp, f4 coords)
{
    const float *params = *pp;
    uint32_t n = (uint32_t)params[0];
    const float *pts = params + 1;
    float qx = coords.x, qy = coords.y;
    float nnx = 0.0f, nny = 0.0f;
    float nearest_d2 = INFINITY;
    int nearest_is_vertex = 0;
    float outside = 1.0f;
    float cx = pts[2 * (n - 1)], cy = pts[2 * (n - 1) + 1];
    for (uint32_t i = 0; i < n; ++i) {
        float px = cx, py = cy;
        cx = pts[2 * i]; cy = pts[2 * i + 1];
        float dx = cx - px, dy = cy - py;
        float tqx = qx - px, tqy = qy - py;
        float snx = -dy, sny = dx;
        if (((py < qy) != (cy < qy)) && (dy * dot2(snx, sny, tqx, tqy) > 0.0f))
            outside = -outside;
        float t = dot2(dx, dy, tqx, tqy) / dot2(dx, dy, dx, dy);
        if (t > 1.0f)
            continue;
        float cnx, cny, cd2;
        int cvert;
        if (t >= 0.0f) {
            float tcx = fmaf(-t, dx, tqx), tcy = fmaf(-t, dy, tqy);
            cd2 = dot2(tcx, tcy, tcx, tcy);
            cnx = snx; cny = sny;
            cvert = 0;
        } else {
            cnx = tqx; cny = tqy;
            cd2 = dot2(cnx, cny, cnx, cny);
            cvert = cd2 > FLT_EPSILON;
            if (!cvert) { cnx = snx; cny = sny; }
        }
        if (cd2 < nearest_d2) {
            nearest_d2 = cd2; nnx = cnx; nny = cny; nearest_is_vertex = cvert;
        }
    }
    float distance = outside * sqrtf(nearest_d2);
    float inv;
    if (nearest_is_vertex)
        inv = 1.0f / distance;
    else
        inv = 1.0f / dm_length2(nnx, nny);
    *pp = pts + 2 * n;
    return mk4(nnx * inv, nny * inv, 0.0f, distance);
}

/* shapes/simple3d.cl:1-12 */
static inline f4 sphere_op(float r, f4 c)
{
    float a = dm_length3(c.x, c.y, c.z);
    float dist = a - r;
    if (a == 0.0f)
        return mk4(1, 0, 0, dist);
    float inv = 1.0f / a;
    return mk4(c.x * inv, c.y * inv, c.z * inv, dist);
}

/* shapes/simple3d.cl:14-16 */
static inline f4 half_space_op(f4 c) { return mk4(0, -1, 0, -c.y); }

/* shapes/simple3d.cl:18-21 */
static inline f4 extrusion_op(float hh, f4 in, f4 coords)
{
    return perpendicular_intersection(slab_z(hh, coords), in);
}

/* shapes/simple3d.cl:23-26 */
static inline f4 revolution_to_op(f4 c) { return mk4(dm_hypot(c.x, c.z), c.y, 0, 0); }

/* shapes/simple3d.cl:28-39 */
static inline f4 revolution_from_op(f4 flat, f4 coords)
{
    float len = dm_hypot(coords.x, coords.z);
    float m;
    if (len == 0.0f) { coords.x = 1.0f; m = flat.x; }
    else m = flat.x / len;
    return mk4(coords.x * m, flat.y, coords.z * m, flat.w);
}

/* cl_util/util.cl:11-15 rotated2d with (c, s) already known */
static inline void rot2(float c, float s, float px, float py, float *ox, float *oy)
{
    *ox = fmaf(c, px, -(s * py));
    *oy = fmaf(s, px, c * py);
}

/* shapes/simple3d.cl:42-51 */
static inline f4 twist_revolution_to_op(float r, float twist, f4 c)
{
    float alpha = dm_fmod(dm_atan2(c.z, c.x) + DM_PI_F, DM_2PI_F);
    float beta = (twist * alpha) / DM_2PI_F;
    float axis = dm_length2(c.x, c.z);
    float s, co, ox, oy;
    dm_sincos(-beta, &s, &co);
    rot2(co, s, axis - r, c.y, &ox, &oy);
    return mk4(ox, oy, 0, 0);
}

/* shapes/simple3d.cl:53-97 */
static inline f4 twist_revolution_from_op(float minor_r, float r, float twist, f4 res, f4 c)
{
    float axis = dm_length2(c.x, c.z);
    float ipx = axis - r, ipy = c.y;
    float center = dm_length2(ipx, ipy);
    float wrapper = center - minor_r;
    float padding = 0.05f * r;
    float bound, dx, dy;
    if (axis == 0.0f)
        return mk4(1, 0, 0, r - minor_r);
    else if (wrapper > padding) {
        bound = wrapper;
        float inv = 1.0f / center;
        dx = ipx * inv; dy = ipy * inv;
    } else {
        float alpha = dm_fmod(dm_atan2(c.z, c.x) + DM_PI_F, DM_2PI_F);
        float beta = (twist * alpha) / DM_2PI_F;
        float lip = (((r - minor_r) * 2.0f) *
                     dm_sin(fminf(DM_PI_F, (DM_PI_2_F * DM_PI_2_F) / fabsf(twist)))) / minor_r;
        bound = res.w * fminf(1.0f, lip);
        float s, co;
        dm_sincos(beta, &s, &co);
        rot2(co, s, res.x, res.y, &dx, &dy);
    }
    float m = dx / axis;
    return mk4(c.x * m, dy, c.z * m, bound);
}

/* shapes/unsafe.cl:1-6; the tape carries (ox, oy, oz); the reciprocals are rounded once */
static inline f4 repetition_op(float ox, float oy, float oz, f4 c)
{
    return mk4(dm_remainder_inv(c.x, ox, 1.0f / ox), dm_remainder_inv(c.y, oy, 1.0f / oy),
               dm_remainder_inv(c.z, oz, 1.0f / oz), 0.0f);
}

/* shapes/unsafe.cl:8-15 */
static inline f4 circular_repetition_to_op(float pi_over_n, f4 c)
{
    float len = dm_length2(c.x, c.y);
    float alpha = sector_alpha(c.y, c.x, pi_over_n);
    int side = dm_to_int(floorf(alpha / (2.0f * pi_over_n)));
    float mod_alpha = (alpha - (2.0f * (float)side) * pi_over_n) - pi_over_n;
    float s, co;
    dm_sincos(mod_alpha, &s, &co);
    return mk4(len * co, len * s, c.z, 0.0f);
}

/* shapes/unsafe.cl:17-23 */
static inline f4 circular_repetition_from_op(float pi_over_n, f4 dist, f4 c)
{
    float alpha = sector_alpha(c.y, c.x, pi_over_n);
    int side = dm_to_int(floorf(alpha / (2.0f * pi_over_n)));
    float s, co, ox, oy;
    dm_sincos((2.0f * (float)side) * pi_over_n, &s, &co);
    rot2(co, s, dist.x, dist.y, &ox, &oy);
    return mk4(ox, oy, dist.z, dist.w);
}

/* shapes/gears.cl:1-42 */
static inline f4 involute_gear_op(float tooth_count, float pressure_angle, f4 c)
{
    float base_radius = dm_cos(pressure_angle);
    float tooth_angle = DM_PI_F / tooth_count;
    float half_tooth_base = (tooth_angle / 2.0f + dm_tan(pressure_angle)) - pressure_angle;
    float len = dm_hypot(c.x, c.y);
    float alpha = dm_atan2(c.y, c.x);
    float wrapped = dm_fmod(alpha + 2.0f * DM_PI_F, 2.0f * tooth_angle);
    float involute_alpha = half_tooth_base - fabsf(wrapped - tooth_angle);
    if (len < base_radius) {
        float nx = c.y / len, ny = -c.x / len;
        if (wrapped > tooth_angle) { nx = -nx; ny = -ny; }
        float angular = fabsf(wrapped - tooth_angle) - half_tooth_base;
        return mk4(nx, ny, 0.0f, angular * len);
    } else {
        float phi = involute_alpha + dm_acos(base_radius / len);
        float normal_angle;
        if (wrapped < tooth_angle)
            normal_angle = (DM_PI_F - phi) - (alpha - involute_alpha);
        else
            normal_angle = phi - (alpha - involute_alpha);
        float nx, ny;
        dm_sincos(normal_angle, &nx, &ny); /* normal = (sin, cos), gears.cl:33-36 */
        float distance = sqrtf(fmaf(len, len, -(base_radius * base_radius))) - base_radius * phi;
        return mk4(nx, ny, 0.0f, distance);
    }
}

/* The interpreter generated by nodes/codegen.py:5-63 (handlers :91-134), opcode table
 * nodes/node.py:12-56.  Returns 0 on success, <0 on a malformed tape. */
static int evaluate(const float *program, const float *end, f4 point, f4 *result)
{
    f4 registers[EVAL_REGISTER_COUNT];
    f4 last = mk4(0, 0, 0, 0);
    f4 pt = mk4(point.x, point.y, point.z, 0.0f);
    while (program < end) {
        uint32_t instruction = (uint32_t)(*program++);
        uint32_t opcode = instruction / EVAL_REGISTER_COUNT;
        uint32_t reg = instruction % EVAL_REGISTER_COUNT;
        const float *p = program;
        switch (opcode) {
        case 0: *result = last; return 0;                                   /* _return */
        case 1: registers[reg] = last; break;                               /* _store */
        case 2: last = registers[reg]; break;                               /* _load */
        case 3: last = rectangle_op(p[0], p[1], last); program += 2; break;
        case 4: last = circle_op(p[0], last); program += 1; break;
        case 5: last = regular_polygon2d_op(p[0], p[1], last); program += 2; break;
        case 6: last = polygon2d_op(&program, last); break;
        case 7: last = sphere_op(p[0], last); program += 1; break;
        case 8: last = half_space_op(last); break;
        case 9: last = revolution_to_op(last); break;
        case 10: last = twist_revolution_to_op(p[0], p[1], last); program += 2; break;
        case 11: last = transformation_to_op(p, pt); program += 7; break;    /* initial_: arity 0 reads `point` */
        case 12: last = transformation_to_op(p, last); program += 7; break;
        case 13: last = transformation_from_op(p, last); program += 4; break;
        case 14: last = mirror_op(last); break;
        case 15: last = symmetrical_to_op(last); break;
        case 16: last = offset_op(p[0], last); program += 1; break;
        case 17: last = shell_op(p[0], last); program += 1; break;
        case 18: last = repetition_op(p[0], p[1], p[2], last); program += 3; break;
        case 19: last = circular_repetition_to_op(p[0], last); program += 1; break;
        case 20: last = circular_repetition_from_op(p[0], last, registers[reg]); program += 1; break;
        case 21: last = involute_gear_op(p[0], p[1], last); program += 2; break;
        case 22: last = extrusion_op(p[0], last, registers[reg]); program += 1; break;
        case 23: last = revolution_from_op(last, registers[reg]); break;
        case 24: last = twist_revolution_from_op(p[0], p[1], p[2], last, registers[reg]); program += 3; break;
        case 25: last = symmetrical_from_op(last, registers[reg]); break;
        case 26: last = union_op(p[0], last, registers[reg]); program += 1; break;
        case 27: last = intersection_op(p[0], last, registers[reg]); program += 1; break;
        case 28: last = subtraction_op(p[0], last, registers[reg]); program += 1; break;
        default: return -1;
        }
    }
    return -2; /* ran off the end without _return */
}

/* ------------------------------------------------------------------------------------
 * Exported entry points (ctypes).  `dims` = OpenCL global size (gx, gy, gz).
 * Sample position: point = corner + step * (float)gid, mul then add, not fused
 * (grid_eval.cl:31, subdivision.cl:22, mass_properties.cl:25-27; SURVEY.md note 5).
 * ---------------------------------------------------------------------------------- */
static inline f4 sample_point(const float *corner, float step, uint32_t x, uint32_t y, uint32_t z)
{
    return mk4(corner[0] + step * (float)x, corner[1] + step * (float)y,
               corner[2] + step * (float)z, 0.0f);
}

int oracle_evaluate_points(const float *tape, int n_tape, const float *pts, int n, float *out)
{
    for (int i = 0; i < n; ++i) {
        f4 r;
        int rc = evaluate(tape, tape + n_tape, mk4(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], 0), &r);
        if (rc) return rc;
        out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
    }
    return 0;
}

/* grid_eval.cl:23-34; output index INDEX3 = z + sz*(y + sy*x) (cl_util/indexing.h:4).
 * `threads` > 1 splits the x axis with OpenMP (CPU-baseline timing only). */
int oracle_grid_eval(const float *tape, int n_tape, const float *corner, float step,
                     const uint32_t *dims, float *out, int threads)
{
    int err = 0;
    uint32_t sx = dims[0], sy = dims[1], sz = dims[2];
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 1)
    for (uint32_t x = 0; x < sx; ++x)
        for (uint32_t y = 0; y < sy; ++y)
            for (uint32_t z = 0; z < sz; ++z) {
                f4 r;
                int rc = evaluate(tape, tape + n_tape, sample_point(corner, step, x, y, z), &r);
                if (rc) { err = rc; continue; }
                size_t idx = (size_t)z + (size_t)sz * ((size_t)y + (size_t)sy * x);
                out[4 * idx] = r.x; out[4 * idx + 1] = r.y; out[4 * idx + 2] = r.z; out[4 * idx + 3] = r.w;
            }
    return err;
}

/* grid_eval.cl:2-21; index = z + (x + (sy-1-y)*sx)*sz */
int oracle_grid_eval_pymcubes(const float *tape, int n_tape, const float *corner, float step,
                              const uint32_t *dims, float *out, int threads)
{
    int err = 0;
    uint32_t sx = dims[0], sy = dims[1], sz = dims[2];
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 1)
    for (uint32_t x = 0; x < sx; ++x)
        for (uint32_t y = 0; y < sy; ++y)
            for (uint32_t z = 0; z < sz; ++z) {
                f4 r;
                int rc = evaluate(tape, tape + n_tape, sample_point(corner, step, x, y, z), &r);
                if (rc) { err = rc; continue; }
                size_t idx = (size_t)z + ((size_t)x + (size_t)(sy - y - 1) * sx) * sz;
                out[idx] = r.w;
            }
    return err;
}

/* subdivision.cl:12-30.  The reference appends with a global atomic (order
 * nondeterministic); the oracle appends in gid order x-major.  list = uchar4. */
int oracle_subdivision_step(const float *tape, int n_tape, const float *corner, float step,
                            float thr, const uint32_t *dims, uint32_t *counter, uint8_t *list)
{
    uint32_t n = 0;
    for (uint32_t x = 0; x < dims[0]; ++x)
        for (uint32_t y = 0; y < dims[1]; ++y)
            for (uint32_t z = 0; z < dims[2]; ++z) {
                f4 r;
                int rc = evaluate(tape, tape + n_tape, sample_point(corner, step, x, y, z), &r);
                if (rc) return rc;
                if (r.w > -thr && r.w < thr) {
                    list[4 * n] = (uint8_t)x; list[4 * n + 1] = (uint8_t)y;
                    list[4 * n + 2] = (uint8_t)z; list[4 * n + 3] = 0;
                    ++n;
                }
            }
    *counter = n;
    return 0;
}

/* mass_properties.cl:7-56.  sum[10] order xx,xy,xz,x,yy,yz,y,zz,z,n
 * (mass_properties.py:125-127), uint32 wrap-around arithmetic like the device. */
int oracle_mass_properties(const float *tape, int n_tape, const float *corner, float step,
                           float thr, const uint32_t *dims, uint32_t *sum, uint32_t *counter,
                           uint8_t *list)
{
    uint32_t n = 0;
    for (int i = 0; i < 10; ++i) sum[i] = 0;
    for (uint32_t x = 0; x < dims[0]; ++x)
        for (uint32_t y = 0; y < dims[1]; ++y)
            for (uint32_t z = 0; z < dims[2]; ++z) {
                f4 r;
                int rc = evaluate(tape, tape + n_tape, sample_point(corner, step, x, y, z), &r);
                if (rc) return rc;
                if (r.w <= -thr) {
                    uint32_t c[4] = { x, y, z, 1 };
                    int i = 0;
                    for (int j = 0; j < 4; ++j)
                        for (int k = j; k < 4; ++k)
                            sum[i++] += c[j] * c[k];
                } else if (r.w < thr) {
                    list[4 * n] = (uint8_t)x; list[4 * n + 1] = (uint8_t)y;
                    list[4 * n + 2] = (uint8_t)z; list[4 * n + 3] = 0;
                    ++n;
                }
            }
    *counter = n;
    return 0;
}

/* det_math probes for tests/test_oracle_math.py: op selects the function. */
int oracle_det_math(int op, const float *a, const float *b, int n, float *out, float *out2)
{
    for (int i = 0; i < n; ++i) {
        switch (op) {
        case 0: out[i] = dm_atan2(a[i], b[i]); break;
        case 1: dm_sincos(a[i], &out[i], &out2[i]); break;
        case 2: out[i] = dm_tan(a[i]); break;
        case 3: out[i] = dm_acos(a[i]); break;
        case 4: out[i] = dm_fmod(a[i], b[i]); break;
        case 5: out[i] = dm_remainder_inv(a[i], b[i], 1.0f / b[i]); break;
        case 6: out[i] = dm_hypot(a[i], b[i]); break;
        default: return -1;
        }
    }
    return 0;
}

/* ====================================================================================
 * Renderers on top of evaluate() -- SURVEY.md section 8(f) rank 3 (ray caster) and the 2D
 * bitmap renderer.  Restated from reference rendering/ray_caster.cl:1-256 and
 * rendering/bitmap.cl:1-18.  Pinned by the reference's own 32 baseline images
 * (reference tests/baseline/rendered_*.png, tests/test_image.py:16-27, MSE <= 1e-3), which were
 * produced by the reference itself: tests/test_render_baselines.py.
 * Output: uchar RGB, index (y + h*x)*3 = INDEX2_GG*3 (cl_util/indexing.h:5,9).
 * Arithmetic: plain IEEE binary32 in the order written (no fma except inside evaluate()).
 * ================================================================================== */
typedef struct { float x, y, z; } f3;
static inline f3 mk3(float x, float y, float z) { f3 r = { x, y, z }; return r; }
static inline f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 mul3(f3 a, float k) { return mk3(a.x * k, a.y * k, a.z * k); }
static inline float dot3f(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline f3 normalize3(f3 a) { float inv = 1.0f / sqrtf(dot3f(a, a)); return mul3(a, inv); }
static inline float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
static inline float mixf(float a, float b, float t) { return a + (b - a) * t; }
static inline float smoothstepf(float e0, float e1, float x)
{
    float t = clampf((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return (t * t) * (3.0f - 2.0f * t);
}

#define RC_OVER_RELAXATION 0.5f
#define RC_PRIMARY_MAX_STEPS 1000
#define RC_LIGHT_MAX_STEPS 100
#define RC_AO_STEPS 4
#define RC_LIGHT_MIN_INFLUENCE (1.0f / 128.0f)
#define RC_FALSE_COLOR 1u
#define RC_ZEBRA 2u

typedef struct { const float *tape, *end; } scene_t;

/* The renderers below normally run over evaluate() above.  A test may put another evaluator in its place -- the
 * frozen literal-formula one of sdf_literal.c -- to render the reference's baseline images over it as well
 * (tests/test_render_baselines.py): fn(handle, point[3], out[4]).  NULL: evaluate().  Set before a render, not during. */
typedef int (*scene_hook_t)(void *handle, const float *p, float *out);
static scene_hook_t g_scene_hook;
static void *g_scene_handle;
void oracle_set_scene_evaluator(scene_hook_t fn, void *handle) { g_scene_hook = fn; g_scene_handle = handle; }

static inline f4 scene_eval(const scene_t *s, f3 p)
{
    f4 r = mk4(0, 0, 0, 0);
    if (g_scene_hook) {
        const float q[3] = { p.x, p.y, p.z };
        float o[4] = { 0, 0, 0, 0 };
        g_scene_hook(g_scene_handle, q, o);
        return mk4(o[0], o[1], o[2], o[3]);
    }
    evaluate(s->tape, s->end, mk4(p.x, p.y, p.z, 0), &r);
    return r;
}

/* ray_caster.cl:13-26 */
static inline float over_relaxation_step(f3 direction, f4 e)
{
    float over = RC_OVER_RELAXATION * fminf(1.0f, 1.0f + dot3f(direction, mk3(e.x, e.y, e.z)));
    return e.w * (1.0f + over);
}

/* ray_caster.cl:28-40: (diffuse, specular) */
static inline void light_no_trace(f3 normal, f3 to_light, f3 to_camera, float *diffuse, float *specular)
{
    f3 halfway = normalize3(add3(to_light, to_camera));
    float d = fmaxf(0.0f, dot3f(normal, to_light));
    float s = fmaxf(0.0f, dot3f(normal, halfway));
    s *= s; s *= s; s *= s;
    *diffuse = d; *specular = s;
}

/* ray_caster.cl:42-98 */
static inline void light_contribution(const scene_t *sc, f3 point, f3 normal, f3 to_light, f3 to_camera,
                                      float min_distance, float max_distance, uint32_t options,
                                      float *diffuse, float *specular)
{
    float d0, s0;
    light_no_trace(normal, to_light, to_camera, &d0, &s0);
    if (d0 <= 0.0f && s0 <= 0.0f) { *diffuse = 0.0f; *specular = 0.0f; return; }
    float threshold = RC_LIGHT_MIN_INFLUENCE / fmaxf(d0, s0);
    float visibility = 1.0f;
    float distance = min_distance, fallback = distance;
    uint32_t step;
    for (step = 0; step < RC_LIGHT_MAX_STEPS; ++step) {
        f4 e = scene_eval(sc, add3(point, mul3(to_light, distance)));
        visibility = fminf(visibility, e.w / distance);
        if (visibility < threshold) break;
        if (distance - fallback > e.w) { distance = fallback; continue; }
        fallback = distance + e.w;
        distance = distance + over_relaxation_step(to_light, e);
        if (distance > max_distance) break;
    }
    if (options & RC_FALSE_COLOR) { *diffuse = (float)step; *specular = 0.0f; }
    else { *diffuse = visibility * d0; *specular = visibility * s0; }
}

/* ray_caster.cl:100-116 */
static inline float ambient_occlusion(const scene_t *sc, f3 point, f3 normal, float distance_step)
{
    float occlusion = 0.0f, scale = 1.0f, distance = distance_step;
    for (uint32_t i = 0; i < RC_AO_STEPS; ++i) {
        f4 e = scene_eval(sc, add3(point, mul3(normal, distance)));
        occlusion += scale * (distance - e.w);
        scale /= 2.0f;
        distance += distance_step;
    }
    return clampf(1.0f - (occlusion * 0.5f) / (1.0f - scale), 0.0f, 1.0f);
}

/* ray_caster.cl:118-131 */
static inline f3 map_color(float ambient, float diffuse, float specular)
{
    float saturation = 0.75f * smoothstepf(0.0f, 0.25f, diffuse);
    float value = 0.1f + 0.8f * mixf(diffuse, ambient, 0.3f);
    float chroma = value * saturation;
    float X = chroma * 0.7f;
    float m = value - chroma;
    f3 color = mk3(255.0f * (X + m), 255.0f * (chroma + m), 255.0f * (0.0f + m));
    float sp = specular * 128.0f;
    return mk3(color.x + sp, color.y + sp, color.z + sp);
}

/* ray_caster.cl:133-144 */
static inline f3 map_color_zebra(f3 point, float ambient, float diffuse, float specular)
{
    int white = dm_to_int(floorf(point.y)) & 1;
    float color = 50.0f + 150.0f * (float)white;
    color *= ambient + diffuse;
    color += 128.0f * specular;
    return mk3(color, color, color);
}

/* ray_caster.cl:146-256, one pixel */
static void ray_caster_pixel(const scene_t *sc, uint32_t x, uint32_t y, uint32_t w, uint32_t h, f3 origin,
                             f3 forward, f3 up, f3 right, float pixel_tolerance, float box_radius,
                             float min_distance, float max_distance, float floor_z, uint32_t options,
                             uint8_t *out)
{
    float filmx = (float)x - (float)(w - 1) / 2.0f;
    float filmy = (float)y - (float)(h - 1) / 2.0f;
    f3 direction = normalize3(sub3(add3(forward, mul3(right, filmx)), mul3(up, filmy)));
    const f3 light_dir = normalize3(mk3(1, 2, -1));
    const f3 light2_dir = normalize3(mk3(-1, 1, 0));

    float distance = min_distance, fallback = min_distance;
    f4 e = mk4(0, 0, 0, 0);
    int hit = 0;
    uint32_t step;
    for (step = 0; step < RC_PRIMARY_MAX_STEPS; ++step) {
        e = scene_eval(sc, add3(origin, mul3(direction, distance)));
        if (distance - fallback > e.w) { distance = fallback; continue; }
        hit = e.w < pixel_tolerance * distance;
        if (hit) {
            f3 n = mk3(e.x, e.y, e.z);
            distance += e.w * clampf(1.0f / dot3f(n, mul3(direction, -1.0f)), 0.0f, 2.0f);
            break;
        }
        fallback = distance + e.w;
        distance = distance + over_relaxation_step(direction, e);
        if (distance > max_distance) { distance = INFINITY; break; }
    }

    f3 color;
    float local_eps = fmaxf(1e-4f, 2.0f * fabsf(e.w));
    f3 to_camera = mul3(direction, -1.0f);
    if (options & RC_FALSE_COLOR) {
        f3 point = add3(origin, mul3(direction, distance));
        f3 normal = mk3(e.x, e.y, e.z);
        float residual = hit ? fabsf(scene_eval(sc, point).w) : 0.0f;
        float steps = (float)step, d, s;
        light_contribution(sc, point, normal, mul3(light_dir, -1.0f), to_camera, local_eps, max_distance, options, &d, &s);
        steps += d;
        steps += (float)RC_AO_STEPS;
        color = mk3(steps, 1000.0f * residual, 0.0f);
    } else if (hit) {
        f3 point = add3(origin, mul3(direction, distance));
        f3 normal = mk3(e.x, e.y, e.z);
        float ambient = ambient_occlusion(sc, point, normal, box_radius / 100.0f);
        float d, s, d2, s2;
        light_contribution(sc, point, normal, mul3(light_dir, -1.0f), to_camera, local_eps, max_distance, options, &d, &s);
        light_no_trace(normal, mul3(light2_dir, -1.0f), to_camera, &d2, &s2);
        d = 0.8f * d + 0.2f * d2;
        s = 0.8f * s + 0.2f * s2;
        color = (options & RC_ZEBRA) ? map_color_zebra(point, ambient, d, s) : map_color(ambient, d, s);
    } else {
        color = mk3(230, 230, 241);
    }

    float floor_distance = (floor_z - origin.z) / direction.z;
    if (floor_distance > 0.0f && floor_distance < distance) {
        f3 floor_point = add3(origin, mul3(direction, floor_distance));
        float fd = scene_eval(sc, floor_point).w;
        float shadow = clampf((2.0f * fd) / box_radius, 0.0f, 1.0f);
        shadow = 1.0f - shadow;
        shadow *= shadow;
        shadow = 1.0f - shadow;
        float k = 0.4f + 0.6f * shadow;
        color = mk3(mixf(0.0f, color.x, k), mixf(0.0f, color.y, k), mixf(0.0f, color.z, k));
    }
    out[0] = (uint8_t)clampf(color.x, 0.0f, 255.0f);
    out[1] = (uint8_t)clampf(color.y, 0.0f, 255.0f);
    out[2] = (uint8_t)clampf(color.z, 0.0f, 255.0f);
}

int oracle_ray_caster(const float *tape, int n_tape, const float *origin, const float *forward, const float *up,
                      const float *right, float pixel_tolerance, float box_radius, float min_distance,
                      float max_distance, float floor_z, uint32_t options, uint32_t w, uint32_t h, uint8_t *out,
                      int threads)
{
    scene_t sc = { tape, tape + n_tape };
    f4 probe;
    if (evaluate(tape, tape + n_tape, mk4(0, 0, 0, 0), &probe)) return -1;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 4)
    for (uint32_t x = 0; x < w; ++x)
        for (uint32_t y = 0; y < h; ++y)
            ray_caster_pixel(&sc, x, y, w, h, mk3(origin[0], origin[1], origin[2]), mk3(forward[0], forward[1], forward[2]),
                             mk3(up[0], up[1], up[2]), mk3(right[0], right[1], right[2]), pixel_tolerance, box_radius,
                             min_distance, max_distance, floor_z, options, out + ((size_t)y + (size_t)h * x) * 3);
    return 0;
}

/* rendering/bitmap.cl:1-18 */
int oracle_bitmap(const float *tape, int n_tape, const float *origin, float step_size, uint32_t w, uint32_t h,
                  uint8_t *out)
{
    scene_t sc = { tape, tape + n_tape };
    f4 probe;
    if (evaluate(tape, tape + n_tape, mk4(0, 0, 0, 0), &probe)) return -1;
    for (uint32_t x = 0; x < w; ++x)
        for (uint32_t y = 0; y < h; ++y) {
            f3 p = mk3(origin[0] + step_size * (float)x, origin[1] + step_size * (float)(h - y - 1), origin[2] + step_size * 0.0f);
            float v = scene_eval(&sc, p).w;
            float t = (v < 0.0f) ? 0.0f : 1.0f; /* step(0, v) */
            uint8_t *o = out + ((size_t)y + (size_t)h * x) * 3;
            o[0] = (uint8_t)clampf(mixf(125.0f, 230.0f, t), 0.0f, 255.0f);
            o[1] = (uint8_t)clampf(mixf(179.0f, 230.0f, t), 0.0f, 255.0f);
            o[2] = (uint8_t)clampf(mixf(0.0f, 241.0f, t), 0.0f, 255.0f);
        }
    return 0;
}

/* ====================================================================================
 * 2D contouring -- SURVEY.md section 8(f) rank 4.  Restated from reference
 * rendering/polygon2d.cl:1-175 (encode_index :5-36, place_vertex :38-80, process_polygon :82-175).
 * The reference's tests never run this kernel (tests/test_polygons2d.py covers the polygon SHAPE
 * only) and there are no fixtures for it: PARITY UNPINNED beyond this line-by-line restatement;
 * tests/test_polygon2d_render.py checks geometric properties of the contours.
 * `corners`: float4[gx*gy] as written by grid_eval over (gx, gy, 1), index y + gy*x.
 * Global size of the reference launch = (gx-1, gy-1, 2); index = t + 2*(y + (gy-1)*x).
 * `starts` are appended in (x, y, t) scan order here (the reference's atomic order is unspecified).
 * Arithmetic: plain IEEE binary32 in the order written.
 * ================================================================================== */
static uint32_t pp_encode_index(int32_t cx, int32_t cy, uint32_t size_x, uint32_t size_y, uint32_t index)
{
    const uint32_t index_size = 20;
    index &= (1u << index_size) - 1u;
    int y;
    int32_t sx, sy;
    if (cx < 0 || (uint32_t)cx >= size_x) { y = 0; sx = cx; sy = cy; }
    else if (cy < 0 || (uint32_t)cy >= size_y) { y = 1; sx = cy; sy = cx; }
    else return index;
    return 0x80000000u | (y ? 0x40000000u : 0u) | (sx < 0 ? 0x20000000u : 0u) | ((uint32_t)sy << index_size) | index;
}

static void pp_place_vertex(const float pos[3][2], const f4 val[3], float out[2])
{
    float ax = 0.0f, ay = 0.0f, weight = 0.0f;
    for (int i = 0; i < 3; ++i) {
        float w = 1.0f / (1.0f + fabsf(val[i].w));
        ax += pos[i][0] * w;
        ay += pos[i][1] * w;
        weight += w;
    }
    ax /= weight;
    ay /= weight;
    float px = ax, py = ay;
    for (int i = 0; i < 8; ++i) {
        float gx = 0.0f, gy = 0.0f, residual = 0.0f;
        for (int j = 0; j < 3; ++j) {
            float nx = val[j].x, ny = val[j].y;
            float tmp = (nx * (px - pos[j][0]) + ny * (py - pos[j][1])) + val[j].w;
            residual += tmp * tmp;
            gx += nx * tmp;
            gy += ny * tmp;
        }
        if (residual < 1e-3f) break;
        float g2 = gx * gx + gy * gy;
        if (g2 < 1e-8f) break;
        float k = residual / g2;
        px -= gx * k;
        py -= gy * k;
    }
    out[0] = px;
    out[1] = py;
}

int oracle_process_polygon(const float *corners, uint32_t gx, uint32_t gy, const float *box_corner, float box_step,
                           float *vertices, uint32_t *links, uint32_t *starts, uint32_t *start_counter)
{
    if (gx < 2 || gy < 2) return -1;
    const uint32_t sx = gx - 1, sy = gy - 1;
    const f4 *c = (const f4 *)corners;
    for (uint32_t x = 0; x < sx; ++x)
        for (uint32_t y = 0; y < sy; ++y)
            for (uint32_t t = 0; t < 2; ++t) {
                const uint32_t off[3][2] = { { 0, 0 }, { 1, 1 }, { t, 1 - t } };
                uint32_t cell_type = 0;
                for (int i = 0; i < 3; ++i)
                    cell_type = (cell_type << 1) | (c[(y + off[i][1]) + (size_t)gy * (x + off[i][0])].w <= 0.0f ? 1u : 0u);
                const uint32_t index = t + 2u * (y + sy * x);
                if (cell_type == 0 || cell_type == 7) { links[index] = 0xffffffffu; continue; }
                int backwards = cell_type == 3 || cell_type == 5 || cell_type == 6;
                if (backwards) cell_type = 7 - cell_type;
                const int flip = t == 1;
                if (flip) backwards = !backwards;
                int32_t fx = 0, fy = 0, rx = 0, ry = 0;
                switch (cell_type) {
                case 1: fx = 0; fy = 1; rx = -1; ry = 0; break;
                case 2: fx = 0; fy = 0; rx = 0; ry = 1; break;
                case 4: fx = -1; fy = 0; rx = 0; ry = 0; break;
                }
                if (backwards) { int32_t a = fx, b = fy; fx = rx; fy = ry; rx = a; ry = b; }
                if (flip) { int32_t a = fx; fx = fy; fy = a; a = rx; rx = ry; ry = a; }
                fx += (int32_t)x; fy += (int32_t)y; rx += (int32_t)x; ry += (int32_t)y;
                /* INDEX3_G(fx, fy, 1 - t) in wrapping unsigned arithmetic, as the size_t expression truncates */
                const uint32_t fwd_index = (1u - t) + 2u * ((uint32_t)fy + sy * (uint32_t)fx);
                links[index] = pp_encode_index(fx, fy, sx, sy, fwd_index);
                const uint32_t start_index = pp_encode_index(rx, ry, sx, sy, index);
                if (start_index & 0x80000000u) starts[(*start_counter)++] = start_index ^ 0x20000000u;
                float pos[3][2];
                f4 val[3];
                for (int i = 0; i < 3; ++i) {
                    const uint32_t px = x + off[i][0], py = y + off[i][1];
                    pos[i][0] = box_corner[0] + (float)px * box_step;
                    pos[i][1] = box_corner[1] + (float)py * box_step;
                    val[i] = c[py + (size_t)gy * px];
                }
                pp_place_vertex(pos, val, vertices + 2 * (size_t)index);
            }
    return 0;
}

/* ====================================================================================
 * Marching cubes over one block of samples -- SURVEY.md section 8(f) rank 2, the consumer of
 * grid_eval_pymcubes (reference rendering/mesh.py:53-63: `mcubes.marching_cubes(block, 0)`).
 * The algorithm lives in a third-party dependency that is NOT in /root/reference: PyMCubes 0.0.6
 * (requirements.txt:11).  This restates the published algorithm (Lorensen & Cline 1987: one case
 * index per cell from the 8 corner signs, a case table of triangles over the 12 cube edges, vertices
 * by linear interpolation along edges, shared between cells) with a case table derived in
 * tools/gen_mc_table.py.  PARITY UNPINNED against PyMCubes itself (its table's choices on
 * ambiguous faces, its triangle order and its vertex order cannot be observed here); pinned by the
 * reference's own mesh test (tests/test_mesh.py:12-29: the mesh must be watertight) and by geometric
 * properties, tests/test_mesh.py.
 *   field: float[A0*A1*A2], index a2 + A2*(a1 + A1*a0); inside = value <= 0.
 *   vertices: array coordinates, double[.][3]; one per active edge, ordered by owning sample (linear
 *     index, the edge's lower end point) then axis; position = owner + (0 - f1) / (f2 - f1) along the axis.
 *   triangles: uint32[.][3] vertex ids, ordered by cell (linear index of its low corner) then table order;
 *     anticlockwise seen from outside the solid (in right-handed array coordinates).
 * Returns the counts; fills the arrays up to the given capacities.
 * ================================================================================== */
#include "mc_table.h"

static const unsigned char kMcCorner[8][3] = { {0,0,0}, {1,0,0}, {1,1,0}, {0,1,0}, {0,0,1}, {1,0,1}, {1,1,1}, {0,1,1} };
/* edge -> (owning corner, axis) */
static const unsigned char kMcEdgeOwner[12][2] = { {0,0}, {1,1}, {3,0}, {0,1}, {4,0}, {5,1}, {7,0}, {4,1}, {0,2}, {1,2}, {2,2}, {3,2} };

int oracle_marching_cubes(const float *field, uint32_t A0, uint32_t A1, uint32_t A2, double *vertices, uint64_t cap_v,
                          uint32_t *triangles, uint64_t cap_t, uint64_t *n_vertices, uint64_t *n_triangles)
{
    const size_t n = (size_t)A0 * A1 * A2;
    const size_t stride[3] = { (size_t)A1 * A2, A2, 1 };
    const uint32_t dims[3] = { A0, A1, A2 };
    uint32_t *info = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t)); /* first vertex id << 3 | active axes */
    if (!info) return -1;
    uint64_t nv = 0, nt = 0;
    for (uint32_t a0 = 0; a0 < A0; ++a0)
        for (uint32_t a1 = 0; a1 < A1; ++a1)
            for (uint32_t a2 = 0; a2 < A2; ++a2) {
                const size_t s = a2 + (size_t)A2 * (a1 + (size_t)A1 * a0);
                const uint32_t a[3] = { a0, a1, a2 };
                const float f1 = field[s];
                const int in1 = f1 <= 0.0f;
                uint32_t flags = 0;
                info[s] = (uint32_t)nv << 3;
                for (int axis = 0; axis < 3; ++axis) {
                    if (a[axis] + 1 >= dims[axis]) continue;
                    const float f2 = field[s + stride[axis]];
                    if ((f2 <= 0.0f) == in1) continue;
                    flags |= 1u << axis;
                    if (nv < cap_v) {
                        const double t = (1.0 * (0.0 - (double)f1)) / ((double)f2 - (double)f1);
                        for (int k = 0; k < 3; ++k) vertices[3 * nv + k] = (double)a[k] + (k == axis ? t : 0.0);
                    }
                    ++nv;
                }
                info[s] |= flags;
            }
    for (uint32_t a0 = 0; a0 + 1 < A0; ++a0)
        for (uint32_t a1 = 0; a1 + 1 < A1; ++a1)
            for (uint32_t a2 = 0; a2 + 1 < A2; ++a2) {
                const size_t s = a2 + (size_t)A2 * (a1 + (size_t)A1 * a0);
                uint32_t cube = 0;
                for (int m = 0; m < 8; ++m)
                    if (field[s + kMcCorner[m][0] * stride[0] + kMcCorner[m][1] * stride[1] + kMcCorner[m][2] * stride[2]] <= 0.0f)
                        cube |= 1u << m;
                for (int k = 0; kMcTriangles[cube][k] >= 0; k += 3) {
                    if (nt < cap_t)
                        for (int j = 0; j < 3; ++j) {
                            const int e = kMcTriangles[cube][k + j];
                            const unsigned char *c = kMcCorner[kMcEdgeOwner[e][0]];
                            const uint32_t axis = kMcEdgeOwner[e][1];
                            const uint32_t w = info[s + c[0] * stride[0] + c[1] * stride[1] + c[2] * stride[2]];
                            triangles[3 * nt + j] = (w >> 3) + (uint32_t)__builtin_popcount(w & ((1u << axis) - 1u));
                        }
                    ++nt;
                }
            }
    free(info);
    *n_vertices = nv;
    *n_triangles = nt;
    return 0;
}
