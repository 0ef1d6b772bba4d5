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
//    cannot live in VGPRs) are per-lane slots in LDS; the reference keeps a fixed 8 KiB
//    private array per work-item;
//  * a lane evaluates ONE voxel (T = float) or TWO adjacent voxels (T = f2, a 2-vector):
//    with two, every scalar instruction of the dispatch (fetch, decode, compare tree, branch --
//    measured to be the co-limiter next to VALU issue) is amortised over twice the work, and
//    the arithmetic maps to packed v_pk_fma/mul/add_f32, the only way to reach the FP32 peak.
//    Each component of a packed op is an IEEE binary32 op, so results are unchanged.
#pragma once

#include "tape_format.hpp"

namespace sdf {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i2 __attribute__((ext_vector_type(2)));

// ---- the small overload set the generic ops are written against --------------------------
template <class T> struct lanes_of { static constexpr int value = 1; };
template <> struct lanes_of<f2> { static constexpr int value = 2; };

template <class T> __device__ __forceinline__ T bc(float s) { return (T)(s); }  // broadcast a tape constant

__device__ __forceinline__ f2 fma_(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 sqrt_(f2 x) { return __builtin_elementwise_sqrt(x); }
__device__ __forceinline__ f2 abs_(f2 x) { return __builtin_elementwise_abs(x); }
__device__ __forceinline__ f2 copysign_(f2 m, f2 s) { return __builtin_elementwise_copysign(m, s); }
__device__ __forceinline__ float rint_(float x) { return __builtin_rintf(x); }
// |x| - h per voxel: the absolute value rides in the subtraction's source modifier (one instruction per voxel),
// where the packed form needs a v_and per voxel in front of the packed subtraction
// (inline asm: written as fabsf(x) - h the compiler re-packs the two voxels and puts the v_and back)
// SDF_ABS_MINUS_BUILTIN (per-tape code, HU_ABS_BUILTIN=1; an experiment that lost, specialise.hpp): the plain form, which
// the compiler can hoist out of the loop over a wavefront's bricks where x and y do not change -- an inline asm it will not.
#if defined(SDF_ABS_MINUS_BUILTIN) && SDF_ABS_MINUS_BUILTIN
__device__ __forceinline__ float abs_minus(float x, float h) { return __builtin_fabsf(x) - h; }
__device__ __forceinline__ f2 abs_minus(f2 x, float h) { return __builtin_elementwise_abs(x) - h; }
#else
__device__ __forceinline__ float abs_minus(float x, float h) { float r; asm("v_sub_f32 %0, |%1|, %2" : "=v"(r) : "v"(x), "s"(h)); return r; }
__device__ __forceinline__ f2 abs_minus(f2 x, float h) { f2 r; r.x = abs_minus(x.x, h); r.y = abs_minus(x.y, h); return r; }
#endif
// The hardware's v_min_f32 / v_max_f32 (ISA pseudocode: a NaN operand yields the other one, -0 orders below
// +0): one full-rate instruction where `a < b ? a : b` is a compare and a select.  The canonical distance of
// union / intersection / subtraction and of the nearer-slab case; the oracle restates it in
// plain C.  Inline asm, not fminf/fmaxf: llvm.minnum leaves the sign of a zero result open and, in IEEE mode,
// brings canonicalising instructions with it.
__device__ __forceinline__ float min_(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float max_(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// max(a, -b): the negation rides in the instruction's source modifier
__device__ __forceinline__ float max_neg_(float a, float b) { float r; asm("v_max_f32 %0, %1, -%2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f2 max_neg_(f2 a, f2 b) { f2 r; r.x = max_neg_(a.x, b.x); r.y = max_neg_(a.y, b.y); return r; }
__device__ __forceinline__ f2 min_(f2 a, f2 b) { f2 r; r.x = min_(a.x, b.x); r.y = min_(a.y, b.y); return r; }
__device__ __forceinline__ f2 max_(f2 a, f2 b) { f2 r; r.x = max_(a.x, b.x); r.y = max_(a.y, b.y); return r; }
__device__ __forceinline__ f2 rint_(f2 x) { return __builtin_elementwise_rint(x); }
// min(min(a, b), c) and max(max(a, b), c) in ONE instruction: the same bits as the two it replaces for every triple of
// special values (zeros of both signs, denormals, infinities, quiet and signalling NaNs) and 2^26 random triples --
// hu_selftest_minmax3 runs that comparison on the device (tests/test_gpu_math.py).  Per-tape code uses them where the
// inner result has no other reader (specialise.hpp render_variant).
__device__ __forceinline__ float min3_(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float max3_(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ f2 min3_(f2 a, f2 b, f2 c) { f2 r; r.x = min3_(a.x, b.x, c.x); r.y = min3_(a.y, b.y, c.y); return r; }
__device__ __forceinline__ f2 max3_(f2 a, f2 b, f2 c) { f2 r; r.x = max3_(a.x, b.x, c.x); r.y = max3_(a.y, b.y, c.y); return r; }
// Per-voxel predicates: the lane's own flag(s) PLUS the wavefront mask of each flag, built up
// alongside (`w` = ballot of `v`).  Every flag starts as a float compare, whose result IS its
// wavefront mask in an SGPR pair, and &, | combine the masks with scalar instructions; so
// "does any lane of the wavefront..." costs no vector instruction.  Asking the compiler for
// ballot(a && b) instead makes it rebuild the mask through a v_cndmask + v_cmp pair (measured: 4 VALU
// instructions per test in the rectangle op).  Unused masks are dead code.
struct m1 { bool v; uint64_t w; };
struct m2 { bool x, y; uint64_t wx, wy; };
__device__ __forceinline__ m1 mk(bool c) { return m1{c, __builtin_amdgcn_ballot_w64(c)}; }
__device__ __forceinline__ m2 mk(bool cx, bool cy) { return m2{cx, cy, __builtin_amdgcn_ballot_w64(cx), __builtin_amdgcn_ballot_w64(cy)}; }
__device__ __forceinline__ m1 operator&(m1 a, m1 b) { return m1{a.v && b.v, a.w & b.w}; }
__device__ __forceinline__ m1 operator|(m1 a, m1 b) { return m1{a.v || b.v, a.w | b.w}; }
__device__ __forceinline__ m2 operator&(m2 a, m2 b) { return m2{a.x && b.x, a.y && b.y, a.wx & b.wx, a.wy & b.wy}; }
__device__ __forceinline__ m2 operator|(m2 a, m2 b) { return m2{a.x || b.x, a.y || b.y, a.wx | b.wx, a.wy | b.wy}; }
// complement, and the masks of "all lanes" -- for the path masks of deferred directions (hip_util.hip generate_source)
__device__ __forceinline__ m1 operator~(m1 a) { return m1{!a.v, ~a.w}; }
__device__ __forceinline__ m2 operator~(m2 a) { return m2{!a.x, !a.y, ~a.wx, ~a.wy}; }
__device__ __forceinline__ m1 gt(float a, float b) { return mk(a > b); }
__device__ __forceinline__ m1 lt(float a, float b) { return mk(a < b); }
__device__ __forceinline__ m1 ge(float a, float b) { return mk(a >= b); }
__device__ __forceinline__ m1 eq(float a, float b) { return mk(a == b); }
__device__ __forceinline__ m1 not_ge(float a, float b) { return mk(!(a >= b)); }  // true for NaN
__device__ __forceinline__ m1 not_le(float a, float b) { return mk(!(a <= b)); }
__device__ __forceinline__ m2 gt(f2 a, f2 b) { return mk(a.x > b.x, a.y > b.y); }
__device__ __forceinline__ m2 lt(f2 a, f2 b) { return mk(a.x < b.x, a.y < b.y); }
__device__ __forceinline__ m2 ge(f2 a, f2 b) { return mk(a.x >= b.x, a.y >= b.y); }
__device__ __forceinline__ m2 eq(f2 a, f2 b) { return mk(a.x == b.x, a.y == b.y); }
__device__ __forceinline__ m2 gt(f2 a, float b) { return mk(a.x > b, a.y > b); }
__device__ __forceinline__ m2 lt(f2 a, float b) { return mk(a.x < b, a.y < b); }
__device__ __forceinline__ m2 ge(f2 a, float b) { return mk(a.x >= b, a.y >= b); }
__device__ __forceinline__ m2 eq(f2 a, float b) { return mk(a.x == b, a.y == b); }
__device__ __forceinline__ m2 not_ge(f2 a, float b) { return mk(!(a.x >= b), !(a.y >= b)); }
__device__ __forceinline__ m2 not_le(f2 a, float b) { return mk(!(a.x <= b), !(a.y <= b)); }
__device__ __forceinline__ float sel(m1 m, float a, float b) { return m.v ? a : b; }
__device__ __forceinline__ f2 sel(m2 m, f2 a, f2 b)
{
    f2 r;
    r.x = m.x ? a.x : b.x;
    r.y = m.y ? a.y : b.y;
    return r;
}
__device__ __forceinline__ float get(float v, int) { return v; }
__device__ __forceinline__ float get(f2 v, int i) { return i ? v.y : v.x; }
__device__ __forceinline__ f2 make_f2(float a, float b) { f2 r; r.x = a; r.y = b; return r; }  // NOT (f2)(a, b): in C++ that casts a comma expression

// Does any lane of the wavefront (any voxel of any lane) have the flag set?  Wave-uniform, so a
// branch on it is a scalar branch: used to skip the sqrt / reciprocal blocks of the
// perpendicular-intersection ops when no lane is in the corner region that needs them.
#ifndef SDF_SKIP_CORNER
#define SDF_SKIP_CORNER 1
#endif
#if SDF_SKIP_CORNER
__device__ __forceinline__ bool any_lane(m1 m) { return m.w != 0ull; }
__device__ __forceinline__ bool any_lane(m2 m) { return (m.wx | m.wy) != 0ull; }
#else
__device__ __forceinline__ bool any_lane(m1) { return true; }
__device__ __forceinline__ bool any_lane(m2) { return true; }
#endif

template <class T> struct V4 { T x, y, z, w; };
template <class T> __device__ __forceinline__ V4<T> v4(T x, T y, T z, T w) { V4<T> r = {x, y, z, w}; return r; }
template <class T> __device__ __forceinline__ V4<T> neg(V4<T> a) { return v4<T>(-a.x, -a.y, -a.z, -a.w); }
template <class T, class M> __device__ __forceinline__ V4<T> sel4(M m, V4<T> a, V4<T> b)
{
    return v4<T>(sel(m, a.x, b.x), sel(m, a.y, b.y), sel(m, a.z, b.z), sel(m, a.w, b.w));
}
__device__ __forceinline__ float4 f4(float x, float y, float z, float w) { return make_float4(x, y, z, w); }
__device__ __forceinline__ float4 voxel(const V4<float>& v, int) { return f4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float4 voxel(const V4<f2>& v, int i) { return f4(get(v.x, i), get(v.y, i), get(v.z, i), get(v.w, i)); }
__device__ __forceinline__ void join(V4<float>& o, float4 a, float4) { o = v4<float>(a.x, a.y, a.z, a.w); }
__device__ __forceinline__ void join(V4<f2>& o, float4 a, float4 b)
{
    o.x = make_f2(a.x, b.x); o.y = make_f2(a.y, b.y); o.z = make_f2(a.z, b.z); o.w = make_f2(a.w, b.w);
}

// Run a scalar (one-voxel) op on each voxel of the lane: used for the rare, branchy ops that
// are __noinline__ functions on float4.
template <class T, class F> __device__ __forceinline__ V4<T> per_voxel(const V4<T>& a, const V4<T>& b, F f)
{
    V4<T> o;
    if constexpr (lanes_of<T>::value == 1) {
        join(o, f(voxel(a, 0), voxel(b, 0)), f4(0, 0, 0, 0));
    } else {
        float4 r0 = f(voxel(a, 0), voxel(b, 0));
        float4 r1 = f(voxel(a, 1), voxel(b, 1));
        join(o, r0, r1);
    }
    return o;
}

template <class T> __device__ __forceinline__ T dot3(T ax, T ay, T az, T bx, T by, T bz)
{
    return fma_(az, bz, fma_(ay, by, ax * bx));
}

// ---- correctly rounded sqrt and reciprocal without the compiler's IEEE expansions ------------
// sqrt(x) and 1/sqrt_rounded(x) are what the canonical arithmetic asks for (sdf_math.hpp,
// oracle: sqrtf and 1.0f / s).  The compiler's correctly rounded expansions cost ~13 + ~10 VALU
// instructions (input scaling for denormals, v_div_scale/v_div_fmas/v_div_fixup); for sponge-class
// tapes that was a third of all VALU work.  For 2^-100 <= x <= 2^100 one Newton/Markstein step on
// the hardware seeds gives the SAME bits:
//     y = v_rsq(x); s0 = x*y; s = fma(fma(-s0, s0, x), y/2, s0)          == sqrtf(x)
//     r0 = v_rcp(s);          r = fma(fma(-s, r0, 1), r0, r0)            == 1.0f / s
// (Seeding the reciprocal with y instead of v_rcp(s) would save a quarter-rate instruction, but one Newton
// step from y is wrong for 200 of the 2^32 inputs, and two steps are still wrong for some and no faster.)  Not an approximation argument: tests/test_gpu_math.py runs hu_selftest_math, which compares
// these functions with the IEEE expansions on ALL 2^32 inputs on the device.  Outside that range
// (zeros, denormals, infinities, NaN, huge values) a wave-uniform branch takes the IEEE path.
#ifndef SDF_FAST_CR_MATH
#define SDF_FAST_CR_MATH 1
#endif
// In straight-line (per-tape) code the compiler otherwise speculates the small IEEE sqrt block and pays its
// 13 VALU instructions per voxel every time (measured: sponge(4) distance-only 1.35 -> 1.02 ms with the
// branch kept).  Inside the interpreter's dispatch loop the same marker has the opposite effect (3.0 ->
// 5.5 ms: it defeats the loop's uniform-region layout), so only the run-time-compiled code uses it.
#ifdef __HIPCC_RTC__
#define SDF_KEEP_BRANCH(why) asm volatile("; " why)
#else
#define SDF_KEEP_BRANCH(why)
#endif
constexpr float kFastLo = 0x1p-100f, kFastHi = 0x1p100f;
__device__ __forceinline__ float rsq_hw(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ f2 rsq_hw(f2 x) { return make_f2(__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y)); }
__device__ __forceinline__ float rcp_hw(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ f2 rcp_hw(f2 x) { return make_f2(__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)); }
__device__ __forceinline__ bool wave_any(m1 m) { return m.w != 0ull; }
__device__ __forceinline__ bool wave_any(m2 m) { return (m.wx | m.wy) != 0ull; }
template <class T> __device__ __forceinline__ auto outside_fast_range(T x) { return not_ge(x, kFastLo) | not_le(x, kFastHi); }
template <class T> struct mask_of { using type = m1; static __device__ __forceinline__ m1 all() { return m1{true, ~0ull}; } };
template <> struct mask_of<f2> { using type = m2; static __device__ __forceinline__ m2 all() { return m2{true, true, ~0ull, ~0ull}; } };

// sqrt(x), identical to sqrt_(x) in every lane/voxel where `used` holds
template <class T, class M> __device__ __forceinline__ T sqrt_cr(T x, M used)
{
#if SDF_FAST_CR_MATH
    const T y = rsq_hw(x);
    const T s0 = x * y, h = 0.5f * y;
    T s = fma_(fma_(-s0, s0, x), h, s0);
    if (wave_any(used & outside_fast_range(x))) {
        SDF_KEEP_BRANCH("IEEE sqrt for out-of-range input");
        s = sqrt_(x);
    }
    return s;
#else
    return sqrt_(x);
#endif
}
// s = sqrt(x), r = 1.0f / s, identical to the plain operations where `used` holds
template <class T, class M> __device__ __forceinline__ void sqrt_inv_cr(T x, M used, T& s, T& r)
{
#if SDF_FAST_CR_MATH
    const T y = rsq_hw(x);
    const T s0 = x * y, h = 0.5f * y;
    s = fma_(fma_(-s0, s0, x), h, s0);
    const T r0 = rcp_hw(s);
    r = fma_(fma_(-s, r0, bc<T>(1.0f)), r0, r0);
    if (wave_any(used & outside_fast_range(x))) {
        SDF_KEEP_BRANCH("IEEE sqrt and divide for out-of-range input");
        s = sqrt_(x);
        r = 1.0f / s;
    }
#else
    s = sqrt_(x);
    r = 1.0f / s;
#endif
}

template <class T> __device__ __forceinline__ T len2(T x, T y) { return sqrt_cr(fma_(y, y, x * x), mask_of<T>::all()); }
template <class T> __device__ __forceinline__ T len3(T x, T y, T z) { return sqrt_cr(fma_(z, z, fma_(y, y, x * x)), mask_of<T>::all()); }
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
    return fma_(az, bz, fma_(ay, by, ax * bx));
}
__device__ __forceinline__ float dot2(float ax, float ay, float bx, float by) { return fma_(ay, by, ax * bx); }

// reference shapes/common.cl:1-6; k = w*w - dot(v,v) is folded at decode time
template <class T>
__device__ __forceinline__ void quat_xform(float qx, float qy, float qz, float qw, float k, T px, T py, T pz,
                                           T& ox, T& oy, T& oz)
{
    const T QX = bc<T>(qx), QY = bc<T>(qy), QZ = bc<T>(qz), QW = bc<T>(qw), K = bc<T>(k);
    T d = fma_(QZ, pz, fma_(QY, py, QX * px));
    T cx = fma_(QY, pz, -(QZ * py));
    T cy = fma_(QZ, px, -(QX * pz));
    T cz = fma_(QX, py, -(QY * px));
    T tx = fma_(cx, QW, QX * d);
    T ty = fma_(cy, QW, QY * d);
    T tz = fma_(cz, QW, QZ * d);
    ox = fma_(px, K, tx + tx);
    oy = fma_(py, K, ty + ty);
    oz = fma_(pz, K, tz + tz);
}

// Rotation about a coordinate axis as the 2x2 rotation-and-scale it is (constants A, B, C folded by the
// decoder, tape.hpp axis_constants): `along` is the coordinate on the axis, (u, v) the other two in cyclic
// order; returns the transformed (along, u, v) plus offsets.  A quarter turn has B == 0 exactly.
template <class T>
__device__ __forceinline__ void axis_rotate(const float* p, T along, T u, T v, float oa, float ou, float ov, T& ra, T& ru, T& rv)
{
    const T A = bc<T>(p[0]), B = bc<T>(p[1]), C = bc<T>(p[2]);
    ra = fma_(along, A, bc<T>(oa));
    ru = fma_(-v, C, bc<T>(ou));
    rv = fma_(u, C, bc<T>(ov));
    if (p[1] != 0.0f) {  // wave-uniform: a tape constant
        ru = fma_(u, B, ru);
        rv = fma_(v, B, rv);
    }
}
// The same for directions (transformation_from: no offsets, constants over |Q|^2): plain products, so a
// component keeps the sign its source had and a constant that is exactly 1 costs nothing in per-tape code.
template <class T> __device__ __forceinline__ void axis_rotate_dir(const float* p, T along, T u, T v, T& ra, T& ru, T& rv)
{
    const T B = bc<T>(p[1]);
    ra = along * p[0];
    ru = (-v) * p[2];
    rv = u * p[2];
    if (p[1] != 0.0f) {
        ru = fma_(u, B, ru);
        rv = fma_(v, B, rv);
    }
}

// reference shapes/simple2d.cl:1-4 (slab_x/slab_y of common.cl:33-39 inlined).  Equal to
// perpendicular_intersection(slab_x, slab_y) (common.cl:15-31) under ==: the zero components
// only drop exact zeros.  Written with selects, not branches: a divergent branch inside the
// dispatch loop makes the compiler structurize the WHOLE loop (~4x instruction bloat).
// `wanted` (here and in extrusion_op / circle_op / sphere_op): the lanes whose result will be used.  The second phase
// of deferred directions evaluates a primitive for the lanes where it won; a corner block or an IEEE fallback that only
// other lanes would need is skipped (their values are discarded anyway).  Everywhere else: all lanes.
template <class T, class M = typename mask_of<T>::type>
__device__ __forceinline__ V4<T> rectangle_op(float hw, float hh, V4<T> c, M wanted = mask_of<T>::all())
{
    const T one = bc<T>(1.0f), zero = bc<T>(0.0f);
    T sx = copysign_(one, c.x), sy = copysign_(one, c.y);
    T wx = abs_minus(c.x, hw), wy = abs_minus(c.y, hh);
    auto corner = gt(wx, 0.0f) & gt(wy, 0.0f);
    auto xs = gt(wx, wy);
    // the nearer slab everywhere; the corner region (outside both slabs) is patched in only when some
    // lane of the wavefront is in it, so the common case pays three selects per voxel, not six
    V4<T> r = v4<T>(sel(xs, sx, zero), sel(xs, zero, sy), zero, max_(wx, wy));
    if (any_lane(corner & wanted)) {
        T dist, inv;
        sqrt_inv_cr(fma_(wy, wy, wx * wx), corner & wanted, dist, inv);
        r.x = sel(corner, sx * (wx * inv), r.x);
        r.y = sel(corner, sy * (wy * inv), r.y);
        r.w = sel(corner, dist, r.w);
    }
    return r;
}

// Distance-only forms (DISTANCE_ONLY interpreter): the same operations that produce .w above,
// nothing else.  Valid for tapes without rounded blends, where no direction ever feeds a
// distance (tape.hpp: direction_feeds_distance).
template <class T> __device__ __forceinline__ T perp_w(T a, T b)
{
    auto corner = gt(a, 0.0f) & gt(b, 0.0f);
    // (the corner region patched in only when some lane is in it, like rectangle_op: the common case pays no select)
    T r = max_(a, b);
    if (any_lane(corner)) r = sel(corner, sqrt_cr(fma_(b, b, a * a), corner), r);
    return r;
}

// ---- values of mixed width (per-tape code, specialise.hpp) ------------------------------------------------------
// Per-tape code gives every value the narrowest type that holds it: f2 where the two voxels of a lane may differ,
// float where they cannot -- in the brick kernels (kernels.hpp) a lane's two voxels differ in x only, so whatever
// is computed from y and z alone is ONE number per lane: half the instructions, half the registers.  Each helper
// below computes in the wider of its operands' types; a float operand is the same IEEE binary32 value in both
// halves of the packed form and every packed operation is two IEEE operations, so the bits are those of the all-f2
// evaluation (tests/test_gpu_bricks.py, test_gpu_random_shapes.py compare them with the oracle).
template <class X> struct plain { using type = X; };
template <class X> struct plain<const X> { using type = X; };
template <class X> using plain_t = typename plain<X>::type;     // decltype of a `const auto` value without its const
template <class A, class B> struct wider { using type = f2; };
template <> struct wider<float, float> { using type = float; };
template <class A, class B> using wider_t = typename wider<A, B>::type;
template <class R, class X> struct as_impl;
template <> struct as_impl<float, float> { static __device__ __forceinline__ float go(float x) { return x; } };
template <> struct as_impl<f2, float> { static __device__ __forceinline__ f2 go(float x) { return (f2)(x); } };
template <> struct as_impl<f2, f2> { static __device__ __forceinline__ f2 go(f2 x) { return x; } };
template <class R, class X> __device__ __forceinline__ R as(X x) { return as_impl<R, X>::go(x); }
template <class M> struct lanes_type { using type = float; };
template <> struct lanes_type<m2> { using type = f2; };
__device__ __forceinline__ m1 as_mask(m1 a, float) { return a; }
__device__ __forceinline__ m2 as_mask(m1 a, f2) { return m2{a.v, a.v, a.w, a.w}; }
__device__ __forceinline__ m2 as_mask(m2 a, f2) { return a; }
template <class A, class B, class C> __device__ __forceinline__ auto fma_x(A a, B b, C c)
{
    using R = wider_t<wider_t<A, B>, C>;
    return fma_(as<R>(a), as<R>(b), as<R>(c));
}
template <class A, class B> __device__ __forceinline__ auto min_x(A a, B b) { using R = wider_t<A, B>; return min_(as<R>(a), as<R>(b)); }
template <class A, class B> __device__ __forceinline__ auto max_x(A a, B b) { using R = wider_t<A, B>; return max_(as<R>(a), as<R>(b)); }
template <class A, class B, class C> __device__ __forceinline__ auto min3_x(A a, B b, C c) { using R = wider_t<wider_t<A, B>, C>; return min3_(as<R>(a), as<R>(b), as<R>(c)); }
template <class A, class B, class C> __device__ __forceinline__ auto max3_x(A a, B b, C c) { using R = wider_t<wider_t<A, B>, C>; return max3_(as<R>(a), as<R>(b), as<R>(c)); }
template <class A, class B> __device__ __forceinline__ auto max_neg_x(A a, B b) { using R = wider_t<A, B>; return max_neg_(as<R>(a), as<R>(b)); }
template <class A, class B> __device__ __forceinline__ auto lt_x(A a, B b) { using R = wider_t<A, B>; return lt(as<R>(a), as<R>(b)); }
// a point or a result whose components have different widths, widened for an op of the library (exec_one)
template <class X, class Y, class Z, class W> __device__ __forceinline__ auto v4x(X x, Y y, Z z, W w)
{
    using R = wider_t<wider_t<X, Y>, wider_t<Z, W>>;
    return v4<R>(as<R>(x), as<R>(y), as<R>(z), as<R>(w));
}
template <class R, class T> __device__ __forceinline__ V4<R> widen4(const V4<T>& v) { return v4<R>(as<R>(v.x), as<R>(v.y), as<R>(v.z), as<R>(v.w)); }

// sqrt_cr with what the launch knows (kernels.hpp JitEval::flags): bit 0 set = every sample coordinate of this launch
// is so far inside the fast range that no sum of squares of local coordinates can leave [2^-100, 2^100] unless it is
// exactly the square of a tape constant's difference (specialise.hpp coordinate_bound) -- then the four compares of
// the range test are skipped by a scalar branch on a kernel argument.
constexpr uint32_t kFlagInRange = 1u;
template <class T, class M> __device__ __forceinline__ T sqrt_cr(T x, M used, uint32_t flags)
{
#if SDF_FAST_CR_MATH
    const T y = rsq_hw(x);
    const T s0 = x * y, h = 0.5f * y;
    T s = fma_(fma_(-s0, s0, x), h, s0);
    if (!(flags & kFlagInRange)) {
        SDF_KEEP_BRANCH("range test of the fast sqrt");
        if (wave_any(used & outside_fast_range(x))) {
            SDF_KEEP_BRANCH("IEEE sqrt for out-of-range input");
            s = sqrt_(x);
        }
    }
    return s;
#else
    return sqrt_(x);
#endif
}
// perp_w for operands of different widths (same operations as perp_w above, in the wider type)
template <class A, class B> __device__ __forceinline__ wider_t<A, B> perp_w_x(A a, B b, uint32_t flags)
{
    using R = wider_t<A, B>;
    const auto corner = as_mask(gt(a, 0.0f), R()) & as_mask(gt(b, 0.0f), R());
    R r = max_(as<R>(a), as<R>(b));
    if (any_lane(corner)) r = sel(corner, sqrt_cr(fma_(as<R>(b), as<R>(b), as<R>(a * a)), corner, flags), r);
    return r;
}
// (a length can be exactly zero -- a sample on the axis --, so these keep the range test whatever the launch knows)
template <class A, class B> __device__ __forceinline__ wider_t<A, B> len2_x(A x, B y)
{
    using R = wider_t<A, B>;
    return sqrt_cr(fma_(as<R>(y), as<R>(y), as<R>(x * x)), mask_of<R>::all());
}
template <class A, class B, class C> __device__ __forceinline__ wider_t<wider_t<A, B>, C> len3_x(A x, B y, C z)
{
    using R = wider_t<wider_t<A, B>, C>;
    using RXY = wider_t<A, B>;
    return sqrt_cr(fma_(as<R>(z), as<R>(z), as<R>(fma_(as<RXY>(y), as<RXY>(y), as<RXY>(x * x)))), mask_of<R>::all());
}

// ---- directions of mixed width (the second phase of per-tape code, specialise.hpp) -------------------------------
// The direction parts of rectangle_op / circle_op / sphere_op / extrusion_op above, operation for operation, on
// coordinates of different widths; `act` = the lanes (voxels) of the path being evaluated, as `wanted` above.
__device__ __forceinline__ m1 narrow_mask(m1 a, float) { return a; }
__device__ __forceinline__ m2 narrow_mask(m2 a, f2) { return a; }
__device__ __forceinline__ m1 narrow_mask(m2 a, float) { return m1{a.x || a.y, a.wx | a.wy}; }   // "either voxel of the lane"
template <class M, class A, class B> __device__ __forceinline__ auto sel_x(M m, A a, B b)
{
    using R = typename lanes_type<M>::type;
    return sel(m, as<R>(a), as<R>(b));
}
template <class T, class M> __device__ __forceinline__ void sqrt_inv_cr(T x, M used, T& s, T& r, uint32_t flags)
{
#if SDF_FAST_CR_MATH
    const T y = rsq_hw(x);
    const T s0 = x * y, h = 0.5f * y;
    s = fma_(fma_(-s0, s0, x), h, s0);
    const T r0 = rcp_hw(s);
    r = fma_(fma_(-s, r0, bc<T>(1.0f)), r0, r0);
    if (!(flags & kFlagInRange)) {
        SDF_KEEP_BRANCH("range test of the fast sqrt and reciprocal");
        if (wave_any(used & outside_fast_range(x))) {
            SDF_KEEP_BRANCH("IEEE sqrt and divide for out-of-range input");
            s = sqrt_(x);
            r = 1.0f / s;
        }
    }
#else
    s = sqrt_(x);
    r = 1.0f / s;
#endif
}
// shell (a component of the direction, the distance that entered) and symmetrical_from (x of the direction, x of the point)
template <class C, class W> __device__ __forceinline__ wider_t<C, W> shell_dir_x(C c, W w)
{
    using R = wider_t<C, W>;
    return sel(as_mask(ge(w, 0.0f), R()), as<R>(c), -as<R>(c));
}
template <class C, class P> __device__ __forceinline__ wider_t<C, P> symm_dir_x(C c, P ptx)
{
    using R = wider_t<C, P>;
    return sel(as_mask(lt(ptx, 0.0f), R()), -as<R>(c), as<R>(c));
}
template <class R> struct D2 { R x, y; };
template <class R> struct D3 { R x, y, z; };
template <class X, class Y, class M>
__device__ __forceinline__ D2<wider_t<X, Y>> rect_dir_x(X cx, Y cy, float hw, float hh, M act, uint32_t flags)
{
    using R = wider_t<X, Y>;
    const R zero = bc<R>(0.0f);
    const R sx = as<R>(copysign_(bc<X>(1.0f), cx)), sy = as<R>(copysign_(bc<Y>(1.0f), cy));
    const X wx0 = abs_minus(cx, hw);
    const Y wy0 = abs_minus(cy, hh);
    const R wx = as<R>(wx0), wy = as<R>(wy0);
    const auto corner = as_mask(gt(wx0, 0.0f), R()) & as_mask(gt(wy0, 0.0f), R());
    const auto xs = gt(wx, wy);
    D2<R> r{sel(xs, sx, zero), sel(xs, zero, sy)};
    const auto wanted = narrow_mask(act, R());
    if (any_lane(corner & wanted)) {
        R dist, inv;
        sqrt_inv_cr(fma_(wy, wy, wx * wx), corner & wanted, dist, inv, flags);
        r.x = sel(corner, sx * (wx * inv), r.x);
        r.y = sel(corner, sy * (wy * inv), r.y);
    }
    return r;
}
template <class X, class Y, class M> __device__ __forceinline__ D2<wider_t<X, Y>> circle_dir_x(X cx, Y cy, M act)
{
    using R = wider_t<X, Y>;
    R a, inv;
    sqrt_inv_cr(fma_(as<R>(cy), as<R>(cy), as<R>(cx * cx)), narrow_mask(act, R()), a, inv);
    const auto zero = eq(a, 0.0f);
    return D2<R>{sel(zero, bc<R>(1.0f), as<R>(cx) * inv), sel(zero, bc<R>(0.0f), as<R>(cy) * inv)};
}
template <class X, class Y, class Z, class M>
__device__ __forceinline__ D3<wider_t<wider_t<X, Y>, Z>> sphere_dir_x(X cx, Y cy, Z cz, M act)
{
    using R = wider_t<wider_t<X, Y>, Z>;
    using RXY = wider_t<X, Y>;
    R a, inv;
    sqrt_inv_cr(fma_(as<R>(cz), as<R>(cz), as<R>(fma_(as<RXY>(cy), as<RXY>(cy), as<RXY>(cx * cx)))), narrow_mask(act, R()), a, inv);
    const auto zero = eq(a, 0.0f);
    return D3<R>{sel(zero, bc<R>(1.0f), as<R>(cx) * inv), sel(zero, bc<R>(0.0f), as<R>(cy) * inv), sel(zero, bc<R>(0.0f), as<R>(cz) * inv)};
}
// extrusion_op: (ix, iy, iz) = the direction that entered, iw its distance, cz = the z of the extruded point
template <class IX, class IY, class IZ, class IW, class CZ, class M>
__device__ __forceinline__ auto extrusion_dir_x(IX ix, IY iy, IZ iz, IW iw, CZ cz, float hh, M act, uint32_t flags)
{
    using R = wider_t<wider_t<wider_t<IX, IY>, wider_t<IZ, IW>>, CZ>;
    const R zero = bc<R>(0.0f);
    const R inx = as<R>(ix), iny = as<R>(iy), inz = as<R>(iz), inw = as<R>(iw);
    const R sz = as<R>(copysign_(bc<CZ>(1.0f), cz));
    const CZ wz0 = abs_minus(cz, hh);
    const R wz = as<R>(wz0);
    const auto corner = as_mask(gt(wz0, 0.0f), R()) & as_mask(gt(iw, 0.0f), R());
    const auto cap = gt(wz, inw);
    D3<R> r{sel(cap, zero, inx), sel(cap, zero, iny), sel(cap, sz, inz)};
    const auto wanted = narrow_mask(act, R());
    if (any_lane(corner & wanted)) {
        R dist, inv;
        sqrt_inv_cr(fma_(inw, inw, wz * wz), corner & wanted, dist, inv, flags);
        const R m1_ = wz * inv, m2_ = inw * inv;
        r.x = sel(corner, inx * m2_ + 0.0f, r.x);      // (+ 0: extrusion_op tells why)
        r.y = sel(corner, iny * m2_ + 0.0f, r.y);
        r.z = sel(corner, fma_(inz, m2_, sz * m1_), r.z);
    }
    return r;
}

// reference shapes/simple3d.cl:18-21 = perpendicular_intersection(slab_z(h, coords), in)
template <class T, class M = typename mask_of<T>::type>
__device__ __forceinline__ V4<T> extrusion_op(float hh, V4<T> in, V4<T> coords, M wanted = mask_of<T>::all())
{
    const T zero = bc<T>(0.0f);
    T sz = copysign_(bc<T>(1.0f), coords.z);
    T wz = abs_minus(coords.z, hh);
    auto corner = gt(wz, 0.0f) & gt(in.w, 0.0f);
    auto cap = gt(wz, in.w);
    V4<T> r = v4<T>(sel(cap, zero, in.x), sel(cap, zero, in.y), sel(cap, sz, in.z), max_(wz, in.w));
    if (any_lane(corner & wanted)) {
        T dist, inv;
        sqrt_inv_cr(fma_(in.w, in.w, wz * wz), corner & wanted, dist, inv);
        T m1 = wz * inv, m2 = in.w * inv;
        // + 0: the slab's direction has x = y = +0, and (+0)*m1 added to a product that is -0 gives +0
        // (perpendicular_intersection, common.cl:15-31, adds both terms); zero signs are observable
        r.x = sel(corner, in.x * m2 + 0.0f, r.x);
        r.y = sel(corner, in.y * m2 + 0.0f, r.y);
        r.z = sel(corner, fma_(in.z, m2, sz * m1), r.z);
        r.w = sel(corner, dist, r.w);
    }
    return r;
}

// reference shapes/common.cl:45-64
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


template <class T> __device__ __forceinline__ V4<T> rounded_union(float r, V4<T> a, V4<T> b)
{
    if (r >= 0.0f)  // wave-uniform: r is a tape constant
        return per_voxel(a, b, [r](float4 x, float4 y) { return rounded_blend(r, x, y); });
    const auto first = lt(a.w, b.w);   // the direction of the nearer one; the distance is the hardware minimum
    return v4<T>(sel(first, a.x, b.x), sel(first, a.y, b.y), sel(first, a.z, b.z), min_(a.w, b.w));
}

// reference shapes/simple2d.cl:6-14
template <class T, class M = typename mask_of<T>::type>
__device__ __forceinline__ V4<T> circle_op(float r, V4<T> c, M wanted = mask_of<T>::all())
{
    T a, inv;
    sqrt_inv_cr(fma_(c.y, c.y, c.x * c.x), wanted, a, inv);
    auto zero = eq(a, 0.0f);
    return v4<T>(sel(zero, bc<T>(1.0f), c.x * inv), sel(zero, bc<T>(0.0f), c.y * inv), bc<T>(0.0f), a - r);
}

// reference shapes/simple3d.cl:1-12
template <class T, class M = typename mask_of<T>::type>
__device__ __forceinline__ V4<T> sphere_op(float r, V4<T> c, M wanted = mask_of<T>::all())
{
    T a, inv;
    sqrt_inv_cr(fma_(c.z, c.z, fma_(c.y, c.y, c.x * c.x)), wanted, a, inv);
    auto zero = eq(a, 0.0f);
    return v4<T>(sel(zero, bc<T>(1.0f), c.x * inv), sel(zero, bc<T>(0.0f), c.y * inv), sel(zero, bc<T>(0.0f), c.z * inv),
                 a - r);
}

// remainder with a pre-rounded reciprocal; inv_y == 0 (y = +inf) returns x (sdf_math.hpp)
template <class T> __device__ __forceinline__ T remainder_t(T x, float y, float inv_y)
{
    T n = rint_(x * inv_y);
    T r = fma_(-n, bc<T>(y), x);
    return (inv_y == 0.0f) ? x : r;
}

// ---- rare ops: scalar, branchy, outlined ---------------------------------------------------
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
    int side = to_int_(__builtin_floorf(alpha / two_pi_over_n));
    float side2 = (2.0f * (float)side);
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
        // branch-free on purpose: this function lives in the translation unit that is built with
        // -structurizecfg-skip-uniform-regions (hip_util/builder.py), where a loop must not hold a divergent branch
        // beside its uniform exit; every value below is the one the reference's if / else / continue picks
        const bool crosses = ((py < qy) != (cy < qy)) && (dy * dot2(snx, sny, tqx, tqy) > 0.0f);
        outside = crosses ? -outside : outside;
        const float t = dot2(dx, dy, tqx, tqy) / dot2(dx, dy, dx, dy);
        const bool beyond = t > 1.0f;        // the reference's `continue`: the next edge owns this end
        const bool on_edge = t >= 0.0f;      // (false for NaN: a zero-length edge is its first vertex)
        const float tcx = fma_(-t, dx, tqx), tcy = fma_(-t, dy, tqy);
        const float edge_d2 = dot2(tcx, tcy, tcx, tcy), vertex_d2 = dot2(tqx, tqy, tqx, tqy);
        const bool cvert = !on_edge && vertex_d2 > 1.1920928955078125e-7f;  // FLT_EPSILON
        const float cd2 = on_edge ? edge_d2 : vertex_d2;
        const float cnx = cvert ? tqx : snx, cny = cvert ? tqy : sny;
        const bool nearer = !beyond && cd2 < nearest_d2;
        nearest_d2 = nearer ? cd2 : nearest_d2;
        nnx = nearer ? cnx : nnx;
        nny = nearer ? cny : nny;
        nearest_is_vertex = nearer ? cvert : nearest_is_vertex;
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
    int side = to_int_(__builtin_floorf(alpha / p[1]));
    float mod_alpha = (alpha - (2.0f * (float)side) * p[0]) - p[0];
    float s, co;
    sincos_(mod_alpha, s, co);
    return f4(len * co, len * s, c.z, 0.0f);
}

// reference shapes/unsafe.cl:17-23
__device__ __noinline__ float4 circular_repetition_from_op(float pi_over_n, float two_pi_over_n, float4 dist, float4 c)
{
    const float p[2] = {pi_over_n, two_pi_over_n};
    float alpha = sector_alpha(c.y, c.x, p[0]);
    int side = to_int_(__builtin_floorf(alpha / p[1]));
    float s, co, ox, oy;
    sincos_((2.0f * (float)side) * p[0], s, co);
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
// Value registers in LDS.  Slots are assigned by tape.hpp allocate_slots (liveness-packed).
// Two areas: `n4` float4 slots (points; in the full program also results), then scalar slots
// (distance-only program: a result is just its distance).
//   T = float: float4 area [slot][lane] (ds_*_b128, lanes 16 B apart), scalar area [slot][lane]
//              (ds_*_b32, lanes 4 B apart) -- both conflict-free.
//   T = f2:    float4 area [slot][component][lane] of 8-byte pairs (ds_*_b64), scalar area
//              [slot][lane] of 8-byte pairs -- conflict-free, and the packed register pairs go
//              out as they are, no repacking.
// ---------------------------------------------------------------------------------------
template <class T> struct Regs;
template <> struct Regs<float> {
    float4* base;  // this lane's slot of float4 register 0
    float* res;    // this lane's slot of scalar register 0
    uint32_t stride;
    static constexpr uint32_t kLaneBytes = 4;
    static __device__ __forceinline__ m1 wanted() { return mask_of<float>::all(); }
    __device__ __forceinline__ Regs(void* lds, uint32_t lane, uint32_t lanes, uint32_t n4)
        : base((float4*)lds + lane), res((float*)((float4*)lds + n4 * lanes) + lane), stride(lanes) {}
    __device__ __forceinline__ V4<float> load(uint32_t r) const { float4 v = base[r * stride]; return v4<float>(v.x, v.y, v.z, v.w); }
    __device__ __forceinline__ void store(uint32_t r, const V4<float>& v) const { base[r * stride] = f4(v.x, v.y, v.z, v.w); }
    __device__ __forceinline__ float load_x(uint32_t r) const { return base[r * stride].x; }
    __device__ __forceinline__ float load_z(uint32_t r) const { return base[r * stride].z; }
    __device__ __forceinline__ float load_res(uint32_t r) const { return res[r * stride]; }
    __device__ __forceinline__ void store_res(uint32_t r, float w) const { res[r * stride] = w; }
};
template <> struct Regs<f2> {
    f2* base;
    f2* res;
    uint32_t stride;
    static constexpr uint32_t kLaneBytes = 8;
    static __device__ __forceinline__ m2 wanted() { return mask_of<f2>::all(); }
    __device__ __forceinline__ Regs(void* lds, uint32_t lane, uint32_t lanes, uint32_t n4)
        : base((f2*)lds + lane), res((f2*)lds + n4 * 4u * lanes + lane), stride(lanes) {}
    __device__ __forceinline__ f2* slot(uint32_t r, uint32_t c) const { return base + (r * 4u + c) * stride; }
    __device__ __forceinline__ V4<f2> load(uint32_t r) const { return v4<f2>(*slot(r, 0), *slot(r, 1), *slot(r, 2), *slot(r, 3)); }
    __device__ __forceinline__ void store(uint32_t r, const V4<f2>& v) const
    {
        *slot(r, 0) = v.x; *slot(r, 1) = v.y; *slot(r, 2) = v.z; *slot(r, 3) = v.w;
    }
    __device__ __forceinline__ f2 load_x(uint32_t r) const { return *slot(r, 0); }
    __device__ __forceinline__ f2 load_z(uint32_t r) const { return *slot(r, 2); }
    __device__ __forceinline__ f2 load_res(uint32_t r) const { return res[r * stride]; }
    __device__ __forceinline__ void store_res(uint32_t r, f2 w) const { res[r * stride] = w; }
};

// Register file of SPECIALISED code (jit.hpp): every slot index is a compile-time constant
// once exec_one is inlined with a literal record, so the array dissolves into VGPRs.
template <class T, int SLOTS> struct RegsV {
    V4<T> v[SLOTS > 0 ? SLOTS : 1];
    static __device__ __forceinline__ typename mask_of<T>::type wanted() { return mask_of<T>::all(); }
    __device__ __forceinline__ V4<T> load(uint32_t r) const { return v[r]; }
    __device__ __forceinline__ void store(uint32_t r, const V4<T>& x) { v[r] = x; }
    __device__ __forceinline__ T load_x(uint32_t r) const { return v[r].x; }
    __device__ __forceinline__ T load_z(uint32_t r) const { return v[r].z; }
    __device__ __forceinline__ T load_res(uint32_t r) const { return v[r].w; }
    __device__ __forceinline__ void store_res(uint32_t r, T w) { v[r].w = w; }
};

// Register file of specialised DISTANCE-ONLY code: the distance-only program numbers its point slots and its
// result slots separately (tape.hpp allocate_slots), so they are two arrays here; both dissolve into VGPRs.
template <class T, int POINTS, int RESULTS> struct RegsDO {
    static __device__ __forceinline__ typename mask_of<T>::type wanted() { return mask_of<T>::all(); }
    V4<T> pt[POINTS > 0 ? POINTS : 1];
    T res[RESULTS > 0 ? RESULTS : 1];
    __device__ __forceinline__ V4<T> load(uint32_t r) const { return pt[r]; }
    __device__ __forceinline__ void store(uint32_t r, const V4<T>& x) { pt[r] = x; }
    __device__ __forceinline__ T load_x(uint32_t r) const { return pt[r].x; }
    __device__ __forceinline__ T load_z(uint32_t r) const { return pt[r].z; }
    __device__ __forceinline__ T load_res(uint32_t r) const { return res[r]; }
    __device__ __forceinline__ void store_res(uint32_t r, T w) { res[r] = w; }
};
// One operand, whatever slot the record names: the second phase of deferred directions re-executes single
// records along the path of the winning primitive and hands each its register operand here.
template <class T> struct RegsOne {
    V4<T> v;
    typename mask_of<T>::type act = mask_of<T>::all();   // the lanes the path being re-executed won
    __device__ __forceinline__ typename mask_of<T>::type wanted() const { return act; }
    __device__ __forceinline__ V4<T> load(uint32_t) const { return v; }
    __device__ __forceinline__ void store(uint32_t, const V4<T>&) {}
    __device__ __forceinline__ T load_x(uint32_t) const { return v.x; }
    __device__ __forceinline__ T load_z(uint32_t) const { return v.z; }
    __device__ __forceinline__ T load_res(uint32_t) const { return v.w; }
    __device__ __forceinline__ void store_res(uint32_t, T) {}
};
// An opaque copy: the compiler must not recognise values derived from it as the ones it already has.  The
// second phase recomputes a primitive's local coordinates from the sample point; without this, value numbering
// would instead keep EVERY primitive's coordinates of the first phase alive across it (sponge(4): 78 registers).
__device__ __forceinline__ float opaque(float x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ f2 opaque(f2 x) { asm volatile("" : "+v"(x)); return x; }

// ---------------------------------------------------------------------------------------
// The interpreter.  `prog` and `extra` are wave-uniform.
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
constexpr int kFetchGroup = SDF_FETCH_GROUP;  // tape.hpp pads the program with kTapePadding _return records

template <class T, bool DISTANCE_ONLY, class R>
__device__ __forceinline__ void exec_leaf(const Rec& cur, V4<T>& last, T px, T py, T pz, R& regs);   // below

// STATIC_OP >= 0 (per-tape code, where the generator knows each record's opcode): the switch below is on a
// compile-time constant, so the front end emits only that case.  Leaving it to the optimiser to discover
// that a literal record selects one case of 38 made hipRTC spend 94 % of its time in the inliner and in
// correlated-value-propagation over dead cases (planetary, 467 records: 18 s -> see DESIGN.md).
template <class T, bool DISTANCE_ONLY, class R, int STATIC_OP = -1>
__device__ __forceinline__ bool exec_one(const Rec& cur, V4<T>& last, const float* __restrict__ extra, T px, T py,
                                         T pz, R& regs)
{
    constexpr bool kStaticOp = STATIC_OP >= 0;
    const uint32_t op = kStaticOp ? (uint32_t)STATIC_OP : (cur.hdr & 0xffu);
    const uint32_t reg = (cur.hdr >> 8) & 0xffffu;
    const bool scalar_slot = DISTANCE_ONLY && (cur.hdr & kResultKind);  // wave-uniform
    const float* p = cur.p;
    const V4<T> none = v4<T>(bc<T>(0.0f), bc<T>(0.0f), bc<T>(0.0f), bc<T>(0.0f));
    const uint32_t fold = __float_as_uint(cur.p[kFoldParam]);  // wave-uniform (tape.hpp fold_moves)
    if (fold & kFoldLoad) {
        if (DISTANCE_ONLY && (fold & kFoldLoadResult)) last.w = regs.load_res(fold & 0xffu);
        else last = regs.load(fold & 0xffu);
    }
#ifndef SDF_LEAF_FIRST
#define SDF_LEAF_FIRST 1
#endif
    // the interpreter's programs are mostly fused leaves (tape.hpp fuse_leaves): one test instead of the walk down
    // the compare tree
    if (SDF_LEAF_FIRST && !kStaticOp && op == OPX_LEAF) {
        exec_leaf<T, DISTANCE_ONLY, R>(cur, last, px, py, pz, regs);
    } else
    switch (op) {
    case OP_RETURN: return true;
    case OP_STORE:
        if (scalar_slot) regs.store_res(reg, last.w);
        else regs.store(reg, last);
        break;
    case OP_LOAD:
        if (scalar_slot) last.w = regs.load_res(reg);
        else last = regs.load(reg);
        break;
    case OP_RECTANGLE:
        if (DISTANCE_ONLY) last.w = perp_w<T>(abs_minus(last.x, p[0]), abs_minus(last.y, p[1]));
        else last = rectangle_op(p[0], p[1], last, regs.wanted());
        break;
    case OP_CIRCLE:
        if (DISTANCE_ONLY) last.w = len2(last.x, last.y) - p[0];
        else last = circle_op(p[0], last, regs.wanted());
        break;
    case OP_REGULAR_POLYGON2D: {
        const float a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
        last = per_voxel(last, none, [=](float4 l, float4) { return regular_polygon2d_op(a, b, c, d, e, l); });
        break;
    }
    case OP_POLYGON2D: {
        const float* pts = extra + __float_as_uint(p[1]);
        const uint32_t n = __float_as_uint(p[0]);
        last = per_voxel(last, none, [=](float4 l, float4) { return polygon2d_op(pts, n, l); });
        break;
    }
    case OP_SPHERE:
        if (DISTANCE_ONLY) last.w = len3(last.x, last.y, last.z) - p[0];
        else last = sphere_op(p[0], last, regs.wanted());
        break;
    case OP_HALF_SPACE:
        if (DISTANCE_ONLY) last.w = -last.y;
        else last = v4<T>(bc<T>(0.0f), bc<T>(-1.0f), bc<T>(0.0f), -last.y);
        break;
    case OP_REVOLUTION_TO: last = v4<T>(len2(last.x, last.z), last.y, bc<T>(0.0f), bc<T>(0.0f)); break;
    case OP_TWIST_REVOLUTION_TO: {
        const float a = p[0], b = p[1];
        last = per_voxel(last, none, [=](float4 l, float4) { return twist_revolution_to_op(a, b, l); });
        break;
    }
#if !SDF_TO_SPECIAL || defined(SDF_EXP_KEEP_QUAT)   // (tape_format.hpp: the decoder leaves none of these)
    case OP_INITIAL_TRANSFORMATION_TO: {
        T ox, oy, oz;
        quat_xform<T>(p[0], p[1], p[2], p[3], p[7], px, py, pz, ox, oy, oz);
        last = v4<T>(ox + p[4], oy + p[5], oz + p[6], bc<T>(0.0f));
        break;
    }
    case OP_TRANSFORMATION_TO: {
        T ox, oy, oz;
        quat_xform<T>(p[0], p[1], p[2], p[3], p[7], last.x, last.y, last.z, ox, oy, oz);
        last = v4<T>(ox + p[4], oy + p[5], oz + p[6], bc<T>(0.0f));
        break;
    }
#endif
    case OPX_POINT: last = v4<T>(px, py, pz, bc<T>(0.0f)); break;
    // a general quaternion as its matrix (tape_format.hpp): x' is parked in w, then y', z' and the assembly
    case OPX_TO_ROW_X:
        last.w = fma_(last.x, bc<T>(p[0]), fma_(last.y, bc<T>(p[1]), fma_(last.z, bc<T>(p[2]), bc<T>(p[3]))));
        break;
    case OPX_TO_ROWS_YZ: {
        const T y = fma_(last.x, bc<T>(p[0]), fma_(last.y, bc<T>(p[1]), fma_(last.z, bc<T>(p[2]), bc<T>(p[3]))));
        const T z = fma_(last.x, bc<T>(p[4]), fma_(last.y, bc<T>(p[5]), fma_(last.z, bc<T>(p[6]), bc<T>(p[7]))));
        last = v4<T>(last.w, y, z, bc<T>(0.0f));
        break;
    }
    case OPX_TO_SCALE:  // p[0] = w*w
        last = v4<T>(fma_(last.x, bc<T>(p[0]), bc<T>(p[4])), fma_(last.y, bc<T>(p[0]), bc<T>(p[5])),
                     fma_(last.z, bc<T>(p[0]), bc<T>(p[6])), bc<T>(0.0f));
        break;
    case OPX_TO_AXIS_X: {
        T x, y, z;
        axis_rotate<T>(p, last.x, last.y, last.z, p[4], p[5], p[6], x, y, z);
        last = v4<T>(x, y, z, bc<T>(0.0f));
        break;
    }
    case OPX_TO_AXIS_Y: {
        T x, y, z;
        axis_rotate<T>(p, last.y, last.z, last.x, p[5], p[6], p[4], y, z, x);
        last = v4<T>(x, y, z, bc<T>(0.0f));
        break;
    }
    case OPX_TO_AXIS_Z: {
        T x, y, z;
        axis_rotate<T>(p, last.z, last.x, last.y, p[6], p[4], p[5], z, x, y);
        last = v4<T>(x, y, z, bc<T>(0.0f));
        break;
    }
#if !SDF_FROM_SPECIAL || defined(SDF_EXP_KEEP_QUAT)
    case OP_TRANSFORMATION_FROM: {
        if (DISTANCE_ONLY) {
            last.w = last.w * p[5];
            break;
        }
        T ox, oy, oz;
        quat_xform<T>(p[0], p[1], p[2], p[3], p[4], last.x, last.y, last.z, ox, oy, oz);
        last = v4<T>(ox * p[6], oy * p[6], oz * p[6], last.w * p[5]);
        break;
    }
#endif
    case OPX_INIT_ROW_X:
        last.w = fma_(px, bc<T>(p[0]), fma_(py, bc<T>(p[1]), fma_(pz, bc<T>(p[2]), bc<T>(p[3]))));
        break;
    case OPX_INIT_ROWS_YZ: {
        const T y = fma_(px, bc<T>(p[0]), fma_(py, bc<T>(p[1]), fma_(pz, bc<T>(p[2]), bc<T>(p[3]))));
        const T z = fma_(px, bc<T>(p[4]), fma_(py, bc<T>(p[5]), fma_(pz, bc<T>(p[6]), bc<T>(p[7]))));
        last = v4<T>(last.w, y, z, bc<T>(0.0f));
        break;
    }
    case OPX_FROM_MATRIX: {  // general quaternion: matrix over |Q|^2 (unit directions stay unit), p[9] = |Q|^2
        if (DISTANCE_ONLY) { last.w = last.w * p[9]; break; }
        const T x = fma_(last.x, bc<T>(p[0]), fma_(last.y, bc<T>(p[1]), last.z * p[2]));
        const T y = fma_(last.x, bc<T>(p[3]), fma_(last.y, bc<T>(p[4]), last.z * p[5]));
        const T z = fma_(last.x, bc<T>(p[6]), fma_(last.y, bc<T>(p[7]), last.z * p[8]));
        last = v4<T>(x, y, z, last.w * p[9]);
        break;
    }
    // transformation_from with an axis-aligned quaternion (decoder special cases, tape.hpp): the same 2x2
    // forms with the constants divided by |Q|^2, so unit directions stay unit; p[5] = |Q|^2 scales the distance.
    case OPX_FROM_SCALE:
        if (DISTANCE_ONLY) { last.w = last.w * p[5]; break; }
        last = v4<T>(last.x * p[0], last.y * p[0], last.z * p[0], last.w * p[5]);
        break;
    case OPX_FROM_AXIS_X: {
        if (DISTANCE_ONLY) { last.w = last.w * p[5]; break; }
        T x, y, z;
        axis_rotate_dir<T>(p, last.x, last.y, last.z, x, y, z);
        last = v4<T>(x, y, z, last.w * p[5]);
        break;
    }
    case OPX_FROM_AXIS_Y: {
        if (DISTANCE_ONLY) { last.w = last.w * p[5]; break; }
        T x, y, z;
        axis_rotate_dir<T>(p, last.y, last.z, last.x, y, z, x);
        last = v4<T>(x, y, z, last.w * p[5]);
        break;
    }
    case OPX_FROM_AXIS_Z: {
        if (DISTANCE_ONLY) { last.w = last.w * p[5]; break; }
        T x, y, z;
        axis_rotate_dir<T>(p, last.z, last.x, last.y, z, x, y);
        last = v4<T>(x, y, z, last.w * p[5]);
        break;
    }
    case OP_MIRROR: last.x = -last.x; break;
    case OP_SYMMETRICAL_TO: last.x = abs_(last.x); break;
    case OP_OFFSET: last.w = last.w - p[0]; break;
    case OP_SHELL: {
        if (DISTANCE_ONLY) {
            last.w = sel(ge(last.w, 0.0f), last.w, -last.w) - p[0];
            break;
        }
        V4<T> s = sel4(ge(last.w, 0.0f), last, neg(last));
        last = v4<T>(s.x, s.y, s.z, s.w - p[0]);
        break;
    }
    case OP_REPETITION:
        last = v4<T>(remainder_t(last.x, p[0], p[3]), remainder_t(last.y, p[1], p[4]), remainder_t(last.z, p[2], p[5]),
                     bc<T>(0.0f));
        break;
    case OP_CIRCULAR_REPETITION_TO: {
        const float a = p[0], b = p[1];
        last = per_voxel(last, none, [=](float4 l, float4) { return circular_repetition_to_op(a, b, l); });
        break;
    }
    case OP_CIRCULAR_REPETITION_FROM:
        if (!DISTANCE_ONLY) {
            const float a = p[0], b = p[1];
            last = per_voxel(last, regs.load(reg), [=](float4 l, float4 r) { return circular_repetition_from_op(a, b, l, r); });
        }
        break;
    case OP_INVOLUTE_GEAR: {
        const float a = p[2], b = p[3], c = p[4], d = p[5], e = p[6];
        last = per_voxel(last, none, [=](float4 l, float4) { return involute_gear_op(a, b, c, d, e, l); });
        break;
    }
    case OP_EXTRUSION:
        if (DISTANCE_ONLY) last.w = perp_w<T>(abs_minus(regs.load_z(reg), p[0]), last.w);
        else last = extrusion_op(p[0], last, regs.load(reg), regs.wanted());
        break;
    case OP_REVOLUTION_FROM:
        if (!DISTANCE_ONLY) last = per_voxel(last, regs.load(reg), [](float4 l, float4 r) { return revolution_from_op(l, r); });
        break;
    case OP_TWIST_REVOLUTION_FROM: {
        const float a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5];
        last = per_voxel(last, regs.load(reg),
                         [=](float4 l, float4 r) { return twist_revolution_from_op(a, b, c, d, e, f, l, r); });
        break;
    }
    case OP_SYMMETRICAL_FROM: {
        if (DISTANCE_ONLY) break;
        T ptx = regs.load_x(reg);
        last.x = sel(lt(ptx, 0.0f), -last.x, last.x);
        break;
    }
    case OP_UNION:
        if (DISTANCE_ONLY) last.w = min_(last.w, regs.load_res(reg));
        else last = rounded_union(p[0], last, regs.load(reg));
        break;
    case OP_INTERSECTION:
        if (DISTANCE_ONLY) last.w = max_(last.w, regs.load_res(reg));         // == -min(-a, -b), zeros and NaNs included
        else last = neg(rounded_union(p[0], neg(last), neg(regs.load(reg))));
        break;
    case OP_SUBTRACTION:
        if (DISTANCE_ONLY) last.w = max_neg_(last.w, regs.load_res(reg));     // == -min(-a, b)
        else last = neg(rounded_union(p[0], neg(last), regs.load(reg)));
        break;
    case OPX_LEAF:
        exec_leaf<T, DISTANCE_ONLY, R>(cur, last, px, py, pz, regs);
        break;
    default: return true;  // unreachable: tapes are validated at upload
    }
    if (fold & kFoldStore) {
        if (DISTANCE_ONLY && (fold & kFoldStoreResult)) regs.store_res((fold >> 16) & 0xffu, last.w);
        else regs.store((fold >> 16) & 0xffu, last);
    }
    return false;
}

// OPX_LEAF (tape_format.hpp, tape.hpp fuse_leaves): the parts of a transformed primitive back to back, each exactly
// the code of its single record above; which parts are present is wave-uniform (bits of the control word).
template <class T, bool DISTANCE_ONLY, class R>
__device__ __forceinline__ void exec_leaf(const Rec& cur, V4<T>& last, T px, T py, T pz, R& regs)
{
    const float* p = cur.p;
    const uint32_t c = __float_as_uint(p[kLeafControl]);
    const T zero = bc<T>(0.0f);
    V4<T> q = last;
    if (c & kLeafSample) q = v4<T>(px, py, pz, zero);
    const uint32_t to = (c >> kLeafToShift) & 7u;
    const float* t = p + kLeafTo;
    if (to == 1u) {
        q = v4<T>(fma_(q.x, bc<T>(t[0]), bc<T>(t[3])), fma_(q.y, bc<T>(t[0]), bc<T>(t[4])), fma_(q.z, bc<T>(t[0]), bc<T>(t[5])), zero);
    } else if (to == 2u) {
        T x, y, z;
        axis_rotate<T>(t, q.x, q.y, q.z, t[3], t[4], t[5], x, y, z);
        q = v4<T>(x, y, z, zero);
    } else if (to == 3u) {
        T x, y, z;
        axis_rotate<T>(t, q.y, q.z, q.x, t[4], t[5], t[3], y, z, x);
        q = v4<T>(x, y, z, zero);
    } else if (to == 4u) {
        T x, y, z;
        axis_rotate<T>(t, q.z, q.x, q.y, t[5], t[3], t[4], z, x, y);
        q = v4<T>(x, y, z, zero);
    }
    if (c & kLeafMidStore) regs.store((cur.hdr >> 8) & 0xffu, q);
    // the primitive
    const uint32_t prim = (c >> kLeafPrimShift) & 3u;
    V4<T> r = q;
    if (prim == LEAF_RECTANGLE) {
        if (DISTANCE_ONLY) r.w = perp_w<T>(abs_minus(q.x, p[kLeafPrim]), abs_minus(q.y, p[kLeafPrim + 1]));
        else r = rectangle_op(p[kLeafPrim], p[kLeafPrim + 1], q);
    } else if (prim == LEAF_CIRCLE) {
        if (DISTANCE_ONLY) r.w = len2(q.x, q.y) - p[kLeafPrim];
        else r = circle_op(p[kLeafPrim], q);
    } else if (prim == LEAF_SPHERE) {
        if (DISTANCE_ONLY) r.w = len3(q.x, q.y, q.z) - p[kLeafPrim];
        else r = sphere_op(p[kLeafPrim], q);
    } else {
        if (DISTANCE_ONLY) r.w = -q.y;
        else r = v4<T>(zero, bc<T>(-1.0f), zero, -q.y);
    }
    if (c & kLeafExtrusion) {
        if (DISTANCE_ONLY) r.w = perp_w<T>(abs_minus(q.z, p[kLeafExtrude]), r.w);
        else r = extrusion_op(p[kLeafExtrude], r, q);
    }
#if SDF_LEAF_FROM_LAST
    const uint32_t from = (c & kLeafFromLast) ? 0u : ((c >> kLeafFromShift) & 7u);
#else
    const uint32_t from = (c >> kLeafFromShift) & 7u;
#endif
    if (from != 0u) {
        const float* f = p + kLeafFrom;
        if (DISTANCE_ONLY) {
            r.w = r.w * p[kLeafScale];
        } else if (from == 1u) {
            r = v4<T>(r.x * f[0], r.y * f[0], r.z * f[0], r.w * p[kLeafScale]);
        } else if (from == 2u) {
            T x, y, z;
            axis_rotate_dir<T>(f, r.x, r.y, r.z, x, y, z);
            r = v4<T>(x, y, z, r.w * p[kLeafScale]);
        } else if (from == 3u) {
            T x, y, z;
            axis_rotate_dir<T>(f, r.y, r.z, r.x, y, z, x);
            r = v4<T>(x, y, z, r.w * p[kLeafScale]);
        } else {
            T x, y, z;
            axis_rotate_dir<T>(f, r.z, r.x, r.y, z, x, y);
            r = v4<T>(x, y, z, r.w * p[kLeafScale]);
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint32_t cb = c >> (k == 0 ? kLeafComb1Shift : kLeafComb2Shift);
        const uint32_t kind = cb & 3u, slot = (cb >> 2) & 0xffu;
        if (kind == 0u) continue;
        if (DISTANCE_ONLY) {
            const T b = regs.load_res(slot);
            if (kind == 1u) r.w = min_(r.w, b);
            else if (kind == 2u) r.w = max_(r.w, b);
            else r.w = max_neg_(r.w, b);
        } else {
            const V4<T> b = regs.load(slot);
            if (kind == 1u) r = rounded_union(-1.0f, r, b);
            else if (kind == 2u) r = neg(rounded_union(-1.0f, neg(r), neg(b)));
            else r = neg(rounded_union(-1.0f, neg(r), b));
        }
    }
#if SDF_LEAF_FROM_LAST
    if (c & kLeafFromLast) {   // a scaling applied to the combined value (OPX_FROM_SCALE after the selects)
        if (DISTANCE_ONLY) r.w = r.w * p[kLeafScale];
        else r = v4<T>(r.x * p[kLeafFrom], r.y * p[kLeafFrom], r.z * p[kLeafFrom], r.w * p[kLeafScale]);
    }
#endif
    last = r;
}

template <class T, bool DISTANCE_ONLY, class R>
__device__ __forceinline__ V4<T> run_tape(const Rec* __restrict__ prog, const float* __restrict__ extra, T px, T py,
                                          T pz, R& regs)
{
    V4<T> last = v4<T>(bc<T>(0.0f), bc<T>(0.0f), bc<T>(0.0f), bc<T>(0.0f));
    const Rec* pc = prog;
    for (;;) {
        Rec group[kFetchGroup];
#pragma unroll
        for (int k = 0; k < kFetchGroup; ++k) group[k] = pc[k];
        pc += kFetchGroup;
#pragma unroll
        for (int k = 0; k < kFetchGroup; ++k)
            if (exec_one<T, DISTANCE_ONLY, R>(group[k], last, extra, px, py, pz, regs)) return last;
    }
}

// TABLES (per-tape code, specialise.hpp): what the walks of a box read instead of recomputing statements that read one or
// two sample coordinates.  Columns are template arguments so that every read is one ds_read with an immediate offset (the
// pair of a lane's two voxels, two x entries apart: one ds_read2).
typedef __attribute__((address_space(3))) float lds_float;
// The tables of a 16^3 BOX (kernels.hpp box_eval): single-axis tables of 16 entries per column, and PAIR tables of 16 x 16
// entries per column -- what the tape computes from two coordinates, evaluated once per pair of samples (specialise.hpp
// "PAIR TABLES").  xy and xz are [y or z][x], yz is [y][z]; every pointer is this lane's entry of column 0.  X1: one
// entry (the builders of the pair tables fill one entry per lane) where the walks read a lane's two voxels.
// ROW STRIDES.  A wavefront reads a table at (its lane's y, x), (z, x) or (y, z) with lane -> (x: 2, y: 4, z: 8): half a
// wavefront asks xy for 4 rows, xz for 8 rows, yz for 4 rows of 8 entries.  With rows of 16 floats those rows start in
// LDS banks 0 and 16 only -- two-, four- and two-way bank conflicts on every read, and the walks do little besides
// reading (measured: 0.178 of the leaf blocks' 0.224 ms were the walks, at half the vector ALU's rate).  Rows of 17
// floats put the rows of xy and xz into different banks, rows of 24 floats those of yz (y * 24 mod 32 = 0, 24, 16, 8).
struct BoxTabs {
    static constexpr int kAxis = 16;                   // entries per column of a single-axis table
    static constexpr int kRowX = 17, kRowYZ = 24;      // floats per row of xy / xz, of yz
    static constexpr int kPairStep = 2;                // x entries between a lane's two voxels (two x planes apart)
    static constexpr int kPairX = 16 * kRowX, kPairYZ = 16 * kRowYZ;   // floats per column
    const lds_float* x;
    const lds_float* y;
    const lds_float* z;
    const lds_float* xy;
    const lds_float* xz;
    const lds_float* yz;
    template <int K> __device__ __forceinline__ f2 X() const { return make_f2(x[K * kAxis], x[K * kAxis + kPairStep]); }
    template <int K> __device__ __forceinline__ float X1() const { return x[K * kAxis]; }
    template <int K> __device__ __forceinline__ float Y() const { return y[K * kAxis]; }
    template <int K> __device__ __forceinline__ float Z() const { return z[K * kAxis]; }
    template <int K> __device__ __forceinline__ f2 XY() const { return make_f2(xy[K * kPairX], xy[K * kPairX + kPairStep]); }
    template <int K> __device__ __forceinline__ f2 XZ() const { return make_f2(xz[K * kPairX], xz[K * kPairX + kPairStep]); }
    template <int K> __device__ __forceinline__ float YZ() const { return yz[K * kPairYZ]; }
};

// ---- box pruning (per-tape code, specialise.hpp "BOX PRUNING") ------------------------------------------------------
// A box's mask: bit k set = scope k of the generated code is alive in this box.  The words are wave-uniform (one box per
// workgroup): they sit in scalar registers, and a test is a scalar bit test and branch.
template <int W> struct Prune {
    uint32_t w[W > 0 ? W : 1];
    template <int K> __device__ __forceinline__ bool alive() const { return (w[K >> 5] >> (K & 31)) & 1u; }
    // The same words as values the compiler has not seen before.  The walks call this once per brick: what the compiler
    // derives from the words it then cannot move out of the loop.  Left alone it evaluates every test of a kernel ONCE, ahead
    // of the walks, as 64-bit lane masks -- 131 of them for planetary, far beyond the scalar registers: each then lives in a
    // lane of a vector register and costs a v_readlane, its hazard s_nop and an s_and wherever it is used.  (A volatile asm
    // per TEST, round 4's first form, kept every test where it stood but cost an s_mov each and forbade merging the tests of
    // a scope that is entered twice: C4's leaf level 0.440 -> 0.385 ms with one s_mov per word and brick instead.)
    __device__ __forceinline__ Prune fresh() const
    {
        Prune p = *this;
#pragma unroll
        for (int i = 0; i < (W > 0 ? W : 1); ++i) asm volatile("" : "+s"(p.w[i]));
        return p;
    }
};
// a box's mask from the launch's mask buffer (NULL: nothing was decided, everything is alive)
template <int W> __device__ __forceinline__ Prune<W> load_prune(const uint32_t* __restrict__ masks, uint32_t box)
{
    Prune<W> pr;
#pragma unroll
    for (int i = 0; i < (W > 0 ? W : 1); ++i)
        pr.w[i] = (W > 0 && masks) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)masks[(size_t)box * W + i]) : 0xffffffffu;
    return pr;
}
// the comparison of a select one of whose operands is dead: all lanes chose the survivor
template <class M> __device__ __forceinline__ M mask_const(bool first);
template <> __device__ __forceinline__ m1 mask_const<m1>(bool first) { return m1{first, first ? ~0ull : 0ull}; }
template <> __device__ __forceinline__ m2 mask_const<m2>(bool first) { return m2{first, first, first ? ~0ull : 0ull, first ? ~0ull : 0ull}; }

// Bounds of a distance over a box (the mask function, one box per lane; plain float arithmetic of its own -- none of it
// reaches an output bit).  Every derived bound is moved outwards by 2^-18 of its magnitude: an order of magnitude more
// than the one or two roundings of the operation it mirrors.
struct Iv { float lo, hi; };
constexpr float kIvMargin = 64.0f * 0x1p-23f;      // a leaf's margin: 64 ulps of the largest magnitude its arithmetic sees
__device__ __forceinline__ float iv_down(float x) { return x - __builtin_fabsf(x) * 0x1p-18f; }
__device__ __forceinline__ float iv_up(float x) { return x + __builtin_fabsf(x) * 0x1p-18f; }
__device__ __forceinline__ float iv_hypot(float a, float b)
{
    a = __builtin_fabsf(a); b = __builtin_fabsf(b);
    const float hi = __builtin_fmaxf(a, b), lo = __builtin_fminf(a, b);
    if (!(hi > 0.0f) || hi == __builtin_inff()) return hi;
    const float q = lo / hi;
    return hi * __builtin_sqrtf(1.0f + q * q);
}
__device__ __forceinline__ Iv iv_unknown() { return Iv{-__builtin_inff(), __builtin_inff()}; }
__device__ __forceinline__ Iv iv_zero() { return Iv{0.0f, 0.0f}; }
__device__ __forceinline__ Iv iv_leaf(float centre, float radius, float margin)
{
    const float e = iv_up(radius + margin);
    return Iv{iv_down(centre - e), iv_up(centre + e)};
}
__device__ __forceinline__ Iv iv_neg(Iv a) { return Iv{-a.hi, -a.lo}; }
// does u, known to within +-e over the box, stay inside ONE cell of a repetition (remainder_t: n = rint(u * inv))?  With a
// margin of 1e-3 cells against the rounding of the product and a tie at the boundary.
__device__ __forceinline__ bool iv_same_cell(float u, float e, float inv)
{
    const float x = u * inv;
    return __builtin_fabsf(x - __builtin_rintf(x)) + e * __builtin_fabsf(inv) < 0.499f;
}
__device__ __forceinline__ Iv iv_scale(Iv a, float c) { return c > 0.0f ? Iv{iv_down(a.lo * c), iv_up(a.hi * c)} : Iv{iv_down(a.hi * c), iv_up(a.lo * c)}; }
__device__ __forceinline__ Iv iv_offset(Iv a, float c) { return Iv{iv_down(a.lo - c), iv_up(a.hi - c)}; }
__device__ __forceinline__ Iv iv_shell(Iv a, float c)
{
    const float least = __builtin_fmaxf(__builtin_fmaxf(a.lo, -a.hi), 0.0f), most = __builtin_fmaxf(__builtin_fabsf(a.lo), __builtin_fabsf(a.hi));
    return Iv{iv_down(least - c), iv_up(most - c)};
}
// perp_w: the maximum unless both operands are positive, then the hypotenuse -- non-decreasing in both
__device__ __forceinline__ float iv_perp1(float a, float b) { return (a > 0.0f && b > 0.0f) ? iv_hypot(a, b) : __builtin_fmaxf(a, b); }
__device__ __forceinline__ Iv iv_perp(Iv a, Iv b) { return Iv{iv_down(iv_perp1(a.lo, b.lo)), iv_up(iv_perp1(a.hi, b.hi))}; }
__device__ __forceinline__ Iv iv_min(Iv a, Iv b) { return Iv{__builtin_fminf(a.lo, b.lo), __builtin_fminf(a.hi, b.hi)}; }
__device__ __forceinline__ Iv iv_max(Iv a, Iv b) { return Iv{__builtin_fmaxf(a.lo, b.lo), __builtin_fmaxf(a.hi, b.hi)}; }

// The FULL form of a record on a widened value (the second phase of per-tape code, for the ops it does not restate):
// `act` = the lanes of the path, the record's register operand in `operand`
template <int OP, class TA, class TB, class M>
__device__ __forceinline__ auto run_record_full(const Rec& r, const float* __restrict__ extra, V4<TA> last, V4<TB> operand, M act)
{
    using R = wider_t<wider_t<TA, TB>, typename lanes_type<M>::type>;
    V4<R> v = widen4<R>(last);
    RegsOne<R> one;
    one.v = widen4<R>(operand);
    one.act = act;
    exec_one<R, false, RegsOne<R>, OP>(r, v, extra, bc<R>(0.0f), bc<R>(0.0f), bc<R>(0.0f), one);
    return v;
}

// One record of the library on values of mixed width (per-tape code, specialise.hpp: the ops its phase 1 does not
// restate component by component): `last` and the record's register operand widened to a common type, the
// distance-only form of the op run on them.
template <int OP, class TA, class TB>
__device__ __forceinline__ auto run_record(const Rec& r, const float* __restrict__ extra, V4<TA> last, V4<TB> operand)
{
    using R = wider_t<TA, TB>;
    V4<R> v = widen4<R>(last);
    RegsOne<R> one;
    one.v = widen4<R>(operand);
    exec_one<R, true, RegsOne<R>, OP>(r, v, extra, bc<R>(0.0f), bc<R>(0.0f), bc<R>(0.0f), one);
    return v;
}

}  // namespace sdf
