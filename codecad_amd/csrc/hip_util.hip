// codecad_amd/csrc/hip_util.hip -- gfx950 kernels + the C ABI declared in include/hip_util.h.
//
// Kernels (reference counterparts, paths relative to /root/reference/codecad/):
//   k_grid_eval            grid_eval.cl:2-34 (both layouts), dense slab of a logical grid
//   k_grid_eval_blocks     the per-leaf-block launches of rendering/mesh.py:53-60, batched
//   k_classify<MASS,BATCH> subdivision.cl:12-30 and mass_properties.cl:7-56, either one block
//                          (reference-shaped) or every parent of a level in one launch
//
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math
//        -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -shared (see codecad_amd/hip_util/builder.py)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hip_util.h"
#include "interp.hpp"

using sdf::Rec;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}

#define HU_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(HU_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------

// Sample position of reference grid_eval.cl:31 / subdivision.cl:22 / mass_properties.cl:25-27:
// corner + step * (float)gid, multiply then add, not fused.
__device__ __forceinline__ float sample(float corner, float step, uint32_t i) { return corner + step * (float)i; }

// N = voxels per lane (1: T = float, 2: T = packed float2, see interp.hpp)
template <int N> struct Pack { using T = float; };
template <> struct Pack<2> { using T = sdf::f2; };
__device__ __forceinline__ float pack(const float (&v)[1]) { return v[0]; }
__device__ __forceinline__ sdf::f2 pack(const float (&v)[2]) { return sdf::make_f2(v[0], v[1]); }

// The N consecutive cells (z fastest) a lane owns, starting at linear index lin0 of a grid
// with `n_cells` cells: coordinates by one divide for the first cell and carries for the rest.
template <int N> struct Cells {
    uint32_t x[N], y[N], z[N];
    bool active[N];
    __device__ __forceinline__ Cells(uint32_t lin0, uint32_t n_cells, uint32_t sy, uint32_t sz)
    {
        active[0] = lin0 < n_cells;
        const uint32_t l = active[0] ? lin0 : 0u;  // idle tail lanes follow the (uniform) tape harmlessly
        z[0] = l % sz;
        const uint32_t t = l / sz;
        y[0] = t % sy;
        x[0] = t / sy;
#pragma unroll
        for (int i = 1; i < N; ++i) {
            active[i] = active[0] && (lin0 + i < n_cells);
            const bool wrap_z = z[i - 1] + 1u == sz;
            const bool wrap_y = wrap_z && (y[i - 1] + 1u == sy);
            z[i] = wrap_z ? 0u : z[i - 1] + 1u;
            y[i] = wrap_y ? 0u : (wrap_z ? y[i - 1] + 1u : y[i - 1]);
            x[i] = wrap_y ? x[i - 1] + 1u : x[i - 1];
        }
    }
    __device__ __forceinline__ typename Pack<N>::T position(float corner, float step, const uint32_t (&c)[N], uint32_t c0 = 0) const
    {
        float v[N];
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = sample(corner, step, c0 + c[i]);
        return pack(v);
    }
};

// Workgroup-aggregated stream compaction of N flags per lane: 64-lane ballots + popcount
// prefixes inside each wavefront, wave totals combined through LDS, ONE global atomic per
// workgroup.  slot[i] is meaningful where flag[i] is set.  The reference does one global
// atomic_inc per surviving work-item (subdivision.cl:28).
template <int N>
__device__ __forceinline__ void wg_compact_slots(const bool (&flag)[N], uint32_t* __restrict__ counter, uint32_t* scratch,
                                                 uint32_t (&slot)[N])
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t below = (1ull << lane) - 1ull;
    uint32_t prefix = 0, total_w = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint64_t mask = __ballot(flag[i]);
        prefix += __popcll(mask & below);
        total_w += __popcll(mask);
    }
    if (lane == 0) scratch[wave] = total_w;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t nw = (blockDim.x + 63u) >> 6;
        uint32_t total = 0;
        for (uint32_t w = 0; w < nw; ++w) {
            uint32_t c = scratch[w];
            scratch[w] = total;
            total += c;
        }
        scratch[4] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    slot[0] = scratch[4] + scratch[wave] + prefix;
#pragma unroll
    for (int i = 1; i < N; ++i) slot[i] = slot[i - 1] + (flag[i - 1] ? 1u : 0u);
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ------------------------------------------------------------------------------------------
// dense grid evaluation
// ------------------------------------------------------------------------------------------
template <int LAYOUT, bool DO, int N>
__global__ void __launch_bounds__(256)
k_grid_eval(const Rec* __restrict__ prog, const float* __restrict__ extra, uint32_t n4, float cx, float cy,
            float cz, float step, uint32_t sx, uint32_t sy, uint32_t sz, uint32_t x0,
            uint32_t n_cells, void* __restrict__ out)
{
    using T = typename Pack<N>::T;
    extern __shared__ float4 lds[];
    const uint32_t lin0 = (blockIdx.x * blockDim.x + threadIdx.x) * N;
    const Cells<N> c(lin0, n_cells, sy, sz);
    const sdf::Regs<T> regs(lds, threadIdx.x, blockDim.x, n4);
    const sdf::V4<T> r = sdf::run_tape<T, DO>(prog, extra, c.position(cx, step, c.x, x0), c.position(cy, step, c.y),
                                              c.position(cz, step, c.z), regs);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (!c.active[i]) continue;
        if (LAYOUT == 0) {
            // INDEX3 = z + sz*(y + sy*x) (cl_util/indexing.h:4): inside a slab this is the linear
            // cell index; 64 lanes store 1 KiB (N = 2: 2 KiB) contiguous.
            static_cast<float4*>(out)[lin0 + i] = sdf::voxel(r, i);
        } else {
            // grid_eval.cl:18: z + (x + (sy-1-y)*sx)*sz; z-fastest so a wave stores contiguous runs
            const size_t idx = (size_t)c.z[i] + ((size_t)(x0 + c.x[i]) + (size_t)(sy - 1u - c.y[i]) * sx) * sz;
            static_cast<float*>(out)[idx] = sdf::get(r.w, i);
        }
    }
}

template <int LAYOUT, bool DO, int N>
__global__ void __launch_bounds__(256)
k_grid_eval_blocks(const Rec* __restrict__ prog, const float* __restrict__ extra, uint32_t n4,
                   const int4* __restrict__ blocks, uint32_t chunks, double res, double ox,
                   double oy, double oz, float step, uint32_t sx, uint32_t sy, uint32_t sz,
                   void* __restrict__ out)
{
    using T = typename Pack<N>::T;
    extern __shared__ float4 lds[];
    const uint32_t b = blockIdx.x / chunks, chunk = blockIdx.x - b * chunks;
    const uint32_t cells = sx * sy * sz;
    const int4 ic = blocks[b];
    // subdivision.py:100: pos = int_pos * resolution + origin (fp64), cast once (geometry.py:98-99)
    const float cx = (float)((double)ic.x * res + ox);
    const float cy = (float)((double)ic.y * res + oy);
    const float cz = (float)((double)ic.z * res + oz);
    const uint32_t lin0 = (chunk * blockDim.x + threadIdx.x) * N;
    const Cells<N> c(lin0, cells, sy, sz);
    const sdf::Regs<T> regs(lds, threadIdx.x, blockDim.x, n4);
    const sdf::V4<T> r = sdf::run_tape<T, DO>(prog, extra, c.position(cx, step, c.x), c.position(cy, step, c.y),
                                              c.position(cz, step, c.z), regs);
    const size_t base = (size_t)b * cells;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (!c.active[i]) continue;
        if (LAYOUT == 0)
            static_cast<float4*>(out)[base + lin0 + i] = sdf::voxel(r, i);
        else
            static_cast<float*>(out)[base + (size_t)c.z[i] + ((size_t)c.x[i] + (size_t)(sy - 1u - c.y[i]) * sx) * sz] =
                sdf::get(r.w, i);
    }
}

// ------------------------------------------------------------------------------------------
// classification kernels: subdivision_step / mass_properties, single block or whole level
// ------------------------------------------------------------------------------------------
struct ClassifyArgs {
    const Rec* prog;
    const float* extra;
    const void* parents;   // BATCH: int4[] (subdivision) or double4[] (mass); else unused
    uint32_t chunks;       // workgroups per parent
    uint32_t sx, sy, sz;
    float cx, cy, cz;      // !BATCH: sample corner as given by the caller
    float step, thr;
    int32_t int_step;      // BATCH subdivision: cell size of this level in resolution units
    int32_t dimension;
    double res, ox, oy, oz;  // BATCH subdivision: resolution + origin
    double s;                // BATCH mass: cell size of this level
    uint32_t* counter;
    void* list;            // !BATCH: uchar4[]; BATCH: int4[] / double4[] children
    uint32_t capacity;
    uint32_t* sums;        // MASS: uint32[10] per parent
    uint32_t scratch_offset;  // bytes of LDS taken by the register file (scratch follows)
    uint32_t n4;              // float4 slots of the register file (scalar slots follow them)
};

template <bool MASS, bool BATCH, bool DO, int N>
__global__ void __launch_bounds__(256) k_classify(const ClassifyArgs a)
{
    using T = typename Pack<N>::T;
    extern __shared__ float4 lds[];
    uint32_t* scratch = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(lds) + a.scratch_offset);  // [0..4] compaction, [8..17] sums
    const uint32_t b = BATCH ? blockIdx.x / a.chunks : 0u;
    const uint32_t chunk = BATCH ? blockIdx.x - b * a.chunks : blockIdx.x;
    const uint32_t cells = a.sx * a.sy * a.sz;

    float cx = a.cx, cy = a.cy, cz = a.cz;
    int4 ipar = make_int4(0, 0, 0, 0);
    double pcx = 0.0, pcy = 0.0, pcz = 0.0, pcw = 0.0;
    if (BATCH) {
        if (MASS) {
            // mass_properties.py:86: shifted_corner = box_corner + splat(box_step/2), fp64
            const double4 pc = static_cast<const double4*>(a.parents)[b];
            pcx = pc.x; pcy = pc.y; pcz = pc.z; pcw = pc.w;
            const double h = a.s / 2;
            cx = (float)(pcx + h); cy = (float)(pcy + h); cz = (float)(pcz + h);
        } else {
            // subdivision.py:56-65: (int_corner + int_step/2) * resolution + origin, fp64;
            // 2D shapes shift x and y only
            ipar = static_cast<const int4*>(a.parents)[b];
            const double h = (double)a.int_step / 2;
            cx = (float)(((double)ipar.x + h) * a.res + a.ox);
            cy = (float)(((double)ipar.y + h) * a.res + a.oy);
            cz = (float)(((double)ipar.z + (a.dimension == 3 ? h : 0.0)) * a.res + a.oz);
        }
    }
    if (MASS) {
        if (threadIdx.x < 10) scratch[8 + threadIdx.x] = 0u;
    }

    const uint32_t lin0 = (chunk * blockDim.x + threadIdx.x) * N;
    const Cells<N> c(lin0, cells, a.sy, a.sz);
    const sdf::Regs<T> regs(lds, threadIdx.x, blockDim.x, a.n4);
    const T w = sdf::run_tape<T, DO>(a.prog, a.extra, c.position(cx, a.step, c.x), c.position(cy, a.step, c.y),
                                     c.position(cz, a.step, c.z), regs).w;

    bool ambiguous[N];
    if (MASS) {
        // mass_properties.cl:31-52: inside (w <= -thr) -> moments of the integer cell index;
        // else w < thr -> ambiguous
        bool inside[N];
        bool any_inside = false;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float wi = sdf::get(w, i);
            inside[i] = c.active[i] && (wi <= -a.thr);
            ambiguous[i] = c.active[i] && !inside[i] && (wi < a.thr);
            any_inside |= inside[i];
        }
        const uint64_t imask = __ballot(any_inside);
        __syncthreads();  // scratch[8..17] zeroed
        if (imask != 0ull) {  // wave-uniform
            uint32_t v[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const uint32_t m = inside[i] ? 1u : 0u;
                const uint32_t x = c.x[i], y = c.y[i], z = c.z[i];
                const uint32_t xm = x * m, ym = y * m, zm = z * m;
                v[0] += xm * x; v[1] += xm * y; v[2] += xm * z; v[3] += xm;
                v[4] += ym * y; v[5] += ym * z; v[6] += ym;
                v[7] += zm * z; v[8] += zm; v[9] += m;
            }
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const uint32_t sum = wave_sum(v[i]);
                if ((threadIdx.x & 63u) == 0 && sum) atomicAdd(&scratch[8 + i], sum);
            }
        }
    } else {
        // subdivision.cl:25: -thr < w < thr
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float wi = sdf::get(w, i);
            ambiguous[i] = c.active[i] && (wi > -a.thr) && (wi < a.thr);
        }
    }

    uint32_t slot[N];
    wg_compact_slots<N>(ambiguous, a.counter, scratch, slot);  // has __syncthreads
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (!(ambiguous[i] && slot[i] < a.capacity)) continue;
        const uint32_t x = c.x[i], y = c.y[i], z = c.z[i];
        if (!BATCH) {
            static_cast<uchar4*>(a.list)[slot[i]] = make_uchar4((unsigned char)x, (unsigned char)y, (unsigned char)z, 0);
        } else if (MASS) {
            // mass_properties.py:155: Vector(i,j,k)*s + box_corner, fp64
            static_cast<double4*>(a.list)[slot[i]] =
                make_double4((double)x * a.s + pcx, (double)y * a.s + pcy, (double)z * a.s + pcz, pcw);
        } else {
            // subdivision.py:91-94: Vector(i,j,k)*int_box_step + int_box_corner
            static_cast<int4*>(a.list)[slot[i]] =
                make_int4(ipar.x + (int)x * a.int_step, ipar.y + (int)y * a.int_step, ipar.z + (int)z * a.int_step, ipar.w);
        }
    }
    if (MASS) {
        // wg_compact_slots' barriers ordered the LDS atomics before this read
        if (threadIdx.x < 10) {
            const uint32_t v = scratch[8 + threadIdx.x];
            if (v) atomicAdd(&a.sums[(size_t)b * 10 + threadIdx.x], v);
        }
    }
}

// ------------------------------------------------------------------------------------------
// mass_properties: per-parent index sums -> the ten integrals of this level, on the device.
// The reference does this on the host, block by block, in Python doubles with Kahan sums
// (mass_properties.py:119-148).  Same per-block formulas in fp64 (no contraction), summed
// deterministically: thread t Kahan-accumulates parents t, t+1024, ... and the 1024 partial
// sums are combined by a fixed tree, so the result does not depend on launch timing.
// One workgroup: a level has at most a few 100k parents, i.e. microseconds of work.
// out[10] order: 1, x, y, z, xx, yy, zz, xy, xz, yz.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
k_mass_integrals(const double4* __restrict__ parents, const uint32_t* __restrict__ sums, uint32_t n, double s,
                 double* __restrict__ out)
{
    __shared__ double part[10][1024];
    double acc[10], comp[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = comp[k] = 0.0;
    const double s2 = s * s, s3 = s * s2, h = s / 2, twelfth = s2 / 12;
    for (uint32_t p = threadIdx.x; p < n; p += 1024) {
        const double4 c = parents[p];
        const uint32_t* u = sums + (size_t)p * 10;
        const double sxx = u[0], sxy = u[1], sxz = u[2], sx = u[3], syy = u[4], syz = u[5], sy = u[6], szz = u[7],
                     sz = u[8], cnt = u[9];
        const double bx = c.x + h, by = c.y + h, bz = c.z + h;
        const double tx = s * sx, ty = s * sy, tz = s * sz;
        const double v[10] = {
            s3 * cnt,
            s3 * (cnt * bx + tx), s3 * (cnt * by + ty), s3 * (cnt * bz + tz),
            s3 * (cnt * (bx * bx + twelfth) + 2 * bx * tx + s2 * sxx),
            s3 * (cnt * (by * by + twelfth) + 2 * by * ty + s2 * syy),
            s3 * (cnt * (bz * bz + twelfth) + 2 * bz * tz + s2 * szz),
            s3 * (cnt * bx * by + bx * ty + by * tx + s2 * sxy),
            s3 * (cnt * bx * bz + bx * tz + bz * tx + s2 * sxz),
            s3 * (cnt * by * bz + by * tz + bz * ty + s2 * syz)};
#pragma unroll
        for (int k = 0; k < 10; ++k) {  // Kahan, like the reference's util.KahanSummation
            const double y = v[k] - comp[k];
            const double t = acc[k] + y;
            comp[k] = (t - acc[k]) - y;
            acc[k] = t;
        }
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) part[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (uint32_t stride = 512; stride > 0; stride >>= 1) {
        if (threadIdx.x < stride) {
#pragma unroll
            for (int k = 0; k < 10; ++k) part[k][threadIdx.x] += part[k][threadIdx.x + stride];
        }
        __syncthreads();
    }
    if (threadIdx.x < 10) out[threadIdx.x] = part[threadIdx.x][0];
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
constexpr size_t kMaxLds = 160 * 1024;
constexpr size_t kScratchBytes = 128;

struct LaunchShape {
    const Rec* prog;   // the program variant this launch runs
    uint32_t n4;       // its float4 slot count
    uint32_t block;
    size_t lds;
    size_t regfile_bytes;
    int voxels_per_lane;
};

}  // namespace

struct hu_tape_s {
    Rec* recs_dev = nullptr;     // full program
    Rec* recs_do_dev = nullptr;  // distance-only program (NULL when the tape has a rounded blend)
    float* extra_dev = nullptr;
    int n_instr = 0;
    int n_regs = 0;              // registers named by the tape
    int n_slots = 0;             // float4 slots of the full program after renaming
    int n_point_slots = 0, n_result_slots = 0;  // distance-only program
    int flags = 0;
};

namespace {

// Kernels that only consume the distance run the distance-only interpreter unless the tape has a
// rounded blend (the one op through which a direction feeds a distance) or the caller forces
// the full interpreter (HU_FULL_INTERPRETER=1, used by the parity tests to cover both).
bool distance_only(const hu_tape_s* t)
{
    static const bool forced_full = [] { const char* e = getenv("HU_FULL_INTERPRETER"); return e && e[0] == '1'; }();
    return !forced_full && t->recs_do_dev != nullptr;
}

// Voxels per lane and workgroup size from the register file.
// Two voxels per lane (packed float2) halve the scalar work per voxel (fetch, decode, compare
// tree, branch), which is what limits the interpreter once the VALU work is trimmed, but they
// double the LDS register file.  Measured on MI355X (tools/prof_shape.py):
// two win whenever a 256-lane workgroup's file still fits 48 KiB (>= 3 workgroups per CU):
// always for the distance-only program (48-52 B per voxel), for the full program up to 6 live
// float4 values (sponge(4): 5.3 vs 5.8 ms; sponge(5), 7 values: 8.0 vs 7.7 ms -> one voxel).
// HU_VOXELS_PER_LANE=1|2 forces a choice (the parity tests run both).
int launch_shape(const hu_tape_s* t, LaunchShape& ls, bool distance_only_kernel)
{
    static const int forced = [] { const char* e = getenv("HU_VOXELS_PER_LANE"); return e ? atoi(e) : 0; }();
    const size_t lane_bytes = distance_only_kernel ? (size_t)t->n_point_slots * 16 + (size_t)t->n_result_slots * 4
                                                   : (size_t)t->n_slots * 16;
    const int wanted = (forced == 1 || forced == 2) ? forced : ((lane_bytes * 2 * 256 <= 48 * 1024) ? 2 : 1);
    const Rec* prog = distance_only_kernel ? t->recs_do_dev : t->recs_dev;
    ls.prog = prog;
    ls.n4 = (uint32_t)(distance_only_kernel ? t->n_point_slots : t->n_slots);
    for (int n = wanted; n >= 1; --n) {
        const size_t per_lane = (distance_only_kernel ? (size_t)t->n_point_slots * 16 + (size_t)t->n_result_slots * 4
                                                      : (size_t)t->n_slots * 16) * n;
        uint32_t bs = 256;
        while (bs > 64 && per_lane * bs > 48 * 1024) bs >>= 1;
        const size_t regfile = per_lane * bs;
        if (regfile + kScratchBytes <= kMaxLds) {
            ls.block = bs;
            ls.regfile_bytes = regfile;
            ls.lds = regfile + kScratchBytes;
            ls.voxels_per_lane = n;
            return HU_OK;
        }
    }
    return fail(HU_ERR_UNSUPPORTED, "tape keeps " + std::to_string(t->n_slots) +
                                        " values live at once; at most 159 fit the 160 KiB LDS register file");
}

template <typename K>
int allow_big_lds(K kernel)
{
    HU_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
    return HU_OK;
}

template <int N>
int ensure_attrs_n()
{
    int rc;
    if ((rc = allow_big_lds(k_grid_eval<0, false, N>))) return rc;
    if ((rc = allow_big_lds(k_grid_eval<1, false, N>))) return rc;
    if ((rc = allow_big_lds(k_grid_eval<1, true, N>))) return rc;
    if ((rc = allow_big_lds(k_grid_eval_blocks<0, false, N>))) return rc;
    if ((rc = allow_big_lds(k_grid_eval_blocks<1, false, N>))) return rc;
    if ((rc = allow_big_lds(k_grid_eval_blocks<1, true, N>))) return rc;
    if ((rc = allow_big_lds(k_classify<false, false, false, N>))) return rc;
    if ((rc = allow_big_lds(k_classify<false, true, false, N>))) return rc;
    if ((rc = allow_big_lds(k_classify<true, false, false, N>))) return rc;
    if ((rc = allow_big_lds(k_classify<true, true, false, N>))) return rc;
    if ((rc = allow_big_lds(k_classify<false, false, true, N>))) return rc;
    if ((rc = allow_big_lds(k_classify<false, true, true, N>))) return rc;
    if ((rc = allow_big_lds(k_classify<true, false, true, N>))) return rc;
    if ((rc = allow_big_lds(k_classify<true, true, true, N>))) return rc;
    return HU_OK;
}

int ensure_attrs()
{
    static thread_local int done_for_device = -1;
    int dev = 0;
    HU_HIP(hipGetDevice(&dev));
    if (done_for_device == dev) return HU_OK;
    int rc;
    if ((rc = ensure_attrs_n<1>())) return rc;
    if ((rc = ensure_attrs_n<2>())) return rc;
    done_for_device = dev;
    return HU_OK;
}

int check_dims(const uint32_t dims[3], uint64_t& cells)
{
    if (!dims) return fail(HU_ERR_BAD_ARG, "dims is NULL");
    if (dims[0] == 0 || dims[1] == 0 || dims[2] == 0) return fail(HU_ERR_BAD_ARG, "dims must be >= 1 on every axis");
    cells = (uint64_t)dims[0] * dims[1] * dims[2];
    return HU_OK;
}

}  // namespace

extern "C" {

int hu_abi_version(void) { return HU_ABI_VERSION; }
const char* hu_last_error(void) { return g_last_error.c_str(); }

int hu_device_count(int* count)
{
    if (!count) return fail(HU_ERR_BAD_ARG, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(HU_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = n;
    return HU_OK;
}

int hu_set_device(int ordinal)
{
    HU_HIP(hipSetDevice(ordinal));
    return HU_OK;
}

int hu_device_name(int ordinal, char* buf, size_t buflen)
{
    if (!buf || !buflen) return fail(HU_ERR_BAD_ARG, "buf is NULL");
    hipDeviceProp_t prop;
    HU_HIP(hipGetDeviceProperties(&prop, ordinal));
    std::snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return HU_OK;
}

int hu_synchronize(void)
{
    HU_HIP(hipDeviceSynchronize());
    return HU_OK;
}

int hu_malloc(void** out_dev, size_t bytes)
{
    if (!out_dev) return fail(HU_ERR_BAD_ARG, "out_dev is NULL");
    *out_dev = nullptr;
    HU_HIP(hipMalloc(out_dev, bytes ? bytes : 1));
    return HU_OK;
}

int hu_free(void* dev)
{
    if (dev) HU_HIP(hipFree(dev));
    return HU_OK;
}

int hu_host_alloc(void** out_host, size_t bytes)
{
    if (!out_host) return fail(HU_ERR_BAD_ARG, "out_host is NULL");
    HU_HIP(hipHostMalloc(out_host, bytes ? bytes : 1, hipHostMallocDefault));
    return HU_OK;
}

int hu_host_free(void* host)
{
    if (host) HU_HIP(hipHostFree(host));
    return HU_OK;
}

int hu_memcpy_h2d(void* dst, const void* src, size_t n, void* stream)
{
    HU_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, (hipStream_t)stream));
    return HU_OK;
}

int hu_memcpy_d2h(void* dst, const void* src, size_t n, void* stream)
{
    HU_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return HU_OK;
}

int hu_memcpy_d2d(void* dst, const void* src, size_t n, void* stream)
{
    HU_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return HU_OK;
}

int hu_memset(void* dst, int value, size_t n, void* stream)
{
    HU_HIP(hipMemsetAsync(dst, value, n, (hipStream_t)stream));
    return HU_OK;
}

int hu_stream_create(void** out)
{
    if (!out) return fail(HU_ERR_BAD_ARG, "out is NULL");
    hipStream_t s;
    HU_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = s;
    return HU_OK;
}

int hu_stream_destroy(void* s)
{
    if (s) HU_HIP(hipStreamDestroy((hipStream_t)s));
    return HU_OK;
}

int hu_stream_synchronize(void* s)
{
    HU_HIP(hipStreamSynchronize((hipStream_t)s));
    return HU_OK;
}

int hu_stream_wait_event(void* s, void* ev)
{
    HU_HIP(hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)ev, 0));
    return HU_OK;
}

int hu_event_create(void** out)
{
    if (!out) return fail(HU_ERR_BAD_ARG, "out is NULL");
    hipEvent_t e;
    HU_HIP(hipEventCreate(&e));
    *out = e;
    return HU_OK;
}

int hu_event_destroy(void* e)
{
    if (e) HU_HIP(hipEventDestroy((hipEvent_t)e));
    return HU_OK;
}

int hu_event_record(void* e, void* s)
{
    HU_HIP(hipEventRecord((hipEvent_t)e, (hipStream_t)s));
    return HU_OK;
}

int hu_event_synchronize(void* e)
{
    HU_HIP(hipEventSynchronize((hipEvent_t)e));
    return HU_OK;
}

int hu_event_elapsed_ms(void* a, void* b, float* out_ms)
{
    if (!out_ms) return fail(HU_ERR_BAD_ARG, "out_ms is NULL");
    HU_HIP(hipEventElapsedTime(out_ms, (hipEvent_t)a, (hipEvent_t)b));
    return HU_OK;
}

int hu_tape_create(const float* tape, size_t n, hu_tape* out)
{
    if (!tape || !out) return fail(HU_ERR_BAD_ARG, "tape/out is NULL");
    *out = nullptr;
    sdf::DecodedTape d;
    std::string err = sdf::decode_tape(tape, n, d);
    if (!err.empty()) return fail(HU_ERR_BAD_TAPE, "malformed tape: " + err);
    hu_tape_s* t = new hu_tape_s();
    t->n_instr = (int)d.recs.size() - sdf::kTapePadding;
    t->n_regs = d.n_regs;
    t->n_slots = d.n_slots;
    t->n_point_slots = d.n_point_slots;
    t->n_result_slots = d.n_result_slots;
    t->flags = d.direction_feeds_distance ? 1 : 0;
    hipError_t e = hipMalloc((void**)&t->recs_dev, d.recs.size() * sizeof(Rec));
    if (e == hipSuccess && !d.recs_do.empty()) {
        e = hipMalloc((void**)&t->recs_do_dev, d.recs_do.size() * sizeof(Rec));
        if (e == hipSuccess) e = hipMemcpy(t->recs_do_dev, d.recs_do.data(), d.recs_do.size() * sizeof(Rec), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMalloc((void**)&t->extra_dev, d.extra.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(t->recs_dev, d.recs.data(), d.recs.size() * sizeof(Rec), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t->extra_dev, d.extra.data(), d.extra.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(t->recs_dev);
        (void)hipFree(t->recs_do_dev);
        (void)hipFree(t->extra_dev);
        delete t;
        return fail(HU_ERR_HIP, std::string("tape upload: ") + hipGetErrorString(e));
    }
    *out = t;
    return HU_OK;
}

int hu_tape_destroy(hu_tape t)
{
    if (!t) return HU_OK;
    (void)hipFree(t->recs_dev);
    (void)hipFree(t->recs_do_dev);
    (void)hipFree(t->extra_dev);
    delete t;
    return HU_OK;
}

int hu_tape_info(hu_tape t, int* n_instructions, int* n_registers, int* flags)
{
    if (!t) return fail(HU_ERR_BAD_ARG, "tape is NULL");
    if (n_instructions) *n_instructions = t->n_instr;
    if (n_registers) *n_registers = t->n_slots;
    if (flags) *flags = t->flags;
    return HU_OK;
}

int hu_grid_eval_slab(hu_tape t, const float corner[4], float step, const uint32_t dims[3],
                      uint32_t x0, uint32_t x_count, int layout, void* out_dev, void* stream)
{
    if (!t || !corner || !out_dev) return fail(HU_ERR_BAD_ARG, "NULL argument");
    if (layout != 0 && layout != 1) return fail(HU_ERR_BAD_ARG, "layout must be 0 (float4) or 1 (pymcubes float)");
    uint64_t cells;
    int rc;
    if ((rc = check_dims(dims, cells))) return rc;
    if ((uint64_t)x0 + x_count > dims[0]) return fail(HU_ERR_BAD_ARG, "slab exceeds the grid's x extent");
    const uint64_t plane = (uint64_t)dims[1] * dims[2];
    if (plane >= (1ull << 30)) return fail(HU_ERR_BAD_ARG, "dims[1]*dims[2] must be below 2^30");
    LaunchShape ls;
    if ((rc = launch_shape(t, ls, layout == 1 && distance_only(t)))) return rc;
    if ((rc = ensure_attrs())) return rc;
    // at most 2^30 cells per launch keeps every in-kernel index in 32 bits
    const uint32_t max_x = (uint32_t)((1ull << 30) / plane);
    for (uint32_t done = 0; done < x_count;) {
        const uint32_t nx = (x_count - done < max_x) ? (x_count - done) : max_x;
        const uint32_t n_cells = (uint32_t)(nx * plane);
        const uint32_t per_block = ls.block * ls.voxels_per_lane;
        const uint32_t blocks = (n_cells + per_block - 1) / per_block;
        void* o = (layout == 0) ? (void*)(static_cast<float4*>(out_dev) + (size_t)done * plane) : out_dev;
#define HU_LAUNCH_DENSE(L, D, NV)                                                                                  \
    hipLaunchKernelGGL((k_grid_eval<L, D, NV>), dim3(blocks), dim3(ls.block), ls.lds, (hipStream_t)stream, ls.prog,    \
                       t->extra_dev, ls.n4, corner[0], corner[1], corner[2], step, dims[0], dims[1], dims[2], x0 + done,  \
                       n_cells, o)
        const bool d_only = layout == 1 && distance_only(t);
        if (ls.voxels_per_lane == 2) {
            if (layout == 0) HU_LAUNCH_DENSE(0, false, 2);
            else if (d_only) HU_LAUNCH_DENSE(1, true, 2);
            else HU_LAUNCH_DENSE(1, false, 2);
        } else {
            if (layout == 0) HU_LAUNCH_DENSE(0, false, 1);
            else if (d_only) HU_LAUNCH_DENSE(1, true, 1);
            else HU_LAUNCH_DENSE(1, false, 1);
        }
#undef HU_LAUNCH_DENSE
        HU_HIP(hipGetLastError());
        done += nx;
    }
    return HU_OK;
}

int hu_grid_eval(hu_tape t, const float corner[4], float step, const uint32_t dims[3], void* out_dev, void* stream)
{
    if (!dims) return fail(HU_ERR_BAD_ARG, "dims is NULL");
    return hu_grid_eval_slab(t, corner, step, dims, 0, dims[0], 0, out_dev, stream);
}

int hu_grid_eval_pymcubes(hu_tape t, const float corner[4], float step, const uint32_t dims[3], void* out_dev, void* stream)
{
    if (!dims) return fail(HU_ERR_BAD_ARG, "dims is NULL");
    return hu_grid_eval_slab(t, corner, step, dims, 0, dims[0], 1, out_dev, stream);
}

int hu_grid_eval_blocks(hu_tape t, const int32_t* blocks_dev, uint32_t n_blocks, double resolution,
                        const double origin[3], float step, const uint32_t dims[3], int layout,
                        void* out_dev, void* stream)
{
    if (!t || !origin || !out_dev || (!blocks_dev && n_blocks)) return fail(HU_ERR_BAD_ARG, "NULL argument");
    if (layout != 0 && layout != 1) return fail(HU_ERR_BAD_ARG, "layout must be 0 or 1");
    uint64_t cells;
    int rc;
    if ((rc = check_dims(dims, cells))) return rc;
    if (cells > (1ull << 24)) return fail(HU_ERR_BAD_ARG, "a block may have at most 2^24 cells (256^3)");
    if (n_blocks == 0) return HU_OK;
    LaunchShape ls;
    if ((rc = launch_shape(t, ls, layout == 1 && distance_only(t)))) return rc;
    if ((rc = ensure_attrs())) return rc;
    const uint32_t per_block = ls.block * ls.voxels_per_lane;
    const uint32_t chunks = (uint32_t)((cells + per_block - 1) / per_block);
    if ((uint64_t)chunks * n_blocks > 0x7fffffffull) return fail(HU_ERR_BAD_ARG, "too many workgroups in one launch");
    const dim3 grid(chunks * n_blocks), block(ls.block);
#define HU_LAUNCH_BLOCKS(L, D, NV)                                                                                 \
    hipLaunchKernelGGL((k_grid_eval_blocks<L, D, NV>), grid, block, ls.lds, (hipStream_t)stream, ls.prog,          \
                       t->extra_dev, ls.n4, (const int4*)blocks_dev, chunks, resolution, origin[0], origin[1], origin[2], \
                       step, dims[0], dims[1], dims[2], out_dev)
    const bool d_only = layout == 1 && distance_only(t);
    if (ls.voxels_per_lane == 2) {
        if (layout == 0) HU_LAUNCH_BLOCKS(0, false, 2);
        else if (d_only) HU_LAUNCH_BLOCKS(1, true, 2);
        else HU_LAUNCH_BLOCKS(1, false, 2);
    } else {
        if (layout == 0) HU_LAUNCH_BLOCKS(0, false, 1);
        else if (d_only) HU_LAUNCH_BLOCKS(1, true, 1);
        else HU_LAUNCH_BLOCKS(1, false, 1);
    }
#undef HU_LAUNCH_BLOCKS
    HU_HIP(hipGetLastError());
    return HU_OK;
}

}  // extern "C"

namespace {

template <bool MASS, bool BATCH>
int launch_classify(hu_tape t, ClassifyArgs& a, uint32_t n_parents, const uint32_t dims[3], void* stream)
{
    uint64_t cells;
    int rc;
    if ((rc = check_dims(dims, cells))) return rc;
    if (cells > (1ull << 24)) return fail(HU_ERR_BAD_ARG, "at most 2^24 cells (256^3) per block: cell indices are uchar4");
    if (dims[0] > 256 || dims[1] > 256 || dims[2] > 256)
        return fail(HU_ERR_BAD_ARG, "grid size > 256 would overflow the uchar4 cell index (reference subdivision.py:206-208)");
    if (n_parents == 0) return HU_OK;
    LaunchShape ls;
    if ((rc = launch_shape(t, ls, distance_only(t)))) return rc;
    if ((rc = ensure_attrs())) return rc;
    a.prog = ls.prog;
    a.n4 = ls.n4;
    a.extra = t->extra_dev;
    a.sx = dims[0];
    a.sy = dims[1];
    a.sz = dims[2];
    const uint32_t per_block = ls.block * ls.voxels_per_lane;
    a.chunks = (uint32_t)((cells + per_block - 1) / per_block);
    a.scratch_offset = (uint32_t)ls.regfile_bytes;
    if ((uint64_t)a.chunks * n_parents > 0x7fffffffull) return fail(HU_ERR_BAD_ARG, "too many workgroups in one launch");
    const dim3 grid(a.chunks * n_parents), block(ls.block);
    const bool d_only = distance_only(t);
    if (ls.voxels_per_lane == 2) {
        if (d_only) hipLaunchKernelGGL((k_classify<MASS, BATCH, true, 2>), grid, block, ls.lds, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((k_classify<MASS, BATCH, false, 2>), grid, block, ls.lds, (hipStream_t)stream, a);
    } else {
        if (d_only) hipLaunchKernelGGL((k_classify<MASS, BATCH, true, 1>), grid, block, ls.lds, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((k_classify<MASS, BATCH, false, 1>), grid, block, ls.lds, (hipStream_t)stream, a);
    }
    HU_HIP(hipGetLastError());
    return HU_OK;
}

}  // namespace

extern "C" {

int hu_subdivision_step(hu_tape t, const float corner[4], float step, float threshold, const uint32_t dims[3],
                        uint32_t* counter_dev, void* list_dev, void* stream)
{
    if (!t || !corner || !counter_dev || !list_dev) return fail(HU_ERR_BAD_ARG, "NULL argument");
    ClassifyArgs a;
    std::memset(&a, 0, sizeof(a));
    a.cx = corner[0]; a.cy = corner[1]; a.cz = corner[2];
    a.step = step; a.thr = threshold;
    a.counter = counter_dev; a.list = list_dev; a.capacity = 0xffffffffu;
    return launch_classify<false, false>(t, a, 1, dims, stream);
}

int hu_mass_properties(hu_tape t, const float corner[4], float step, float threshold, const uint32_t dims[3],
                       uint32_t* sum_dev, uint32_t* counter_dev, void* list_dev, void* stream)
{
    if (!t || !corner || !sum_dev || !counter_dev || !list_dev) return fail(HU_ERR_BAD_ARG, "NULL argument");
    ClassifyArgs a;
    std::memset(&a, 0, sizeof(a));
    a.cx = corner[0]; a.cy = corner[1]; a.cz = corner[2];
    a.step = step; a.thr = threshold;
    a.counter = counter_dev; a.list = list_dev; a.capacity = 0xffffffffu;
    a.sums = sum_dev;
    return launch_classify<true, false>(t, a, 1, dims, stream);
}

int hu_subdivision_level(hu_tape t, const int32_t* parents_dev, uint32_t n_parents, int32_t int_step,
                         const uint32_t dims[3], int dimension, double resolution, const double origin[3],
                         float step, float threshold, uint32_t* counter_dev, int32_t* children_dev,
                         uint32_t capacity, void* stream)
{
    if (!t || !origin || !counter_dev || (!children_dev && capacity) || (!parents_dev && n_parents))
        return fail(HU_ERR_BAD_ARG, "NULL argument");
    if (dimension != 2 && dimension != 3) return fail(HU_ERR_BAD_ARG, "dimension must be 2 or 3");
    ClassifyArgs a;
    std::memset(&a, 0, sizeof(a));
    a.parents = parents_dev;
    a.int_step = int_step; a.dimension = dimension;
    a.res = resolution; a.ox = origin[0]; a.oy = origin[1]; a.oz = origin[2];
    a.step = step; a.thr = threshold;
    a.counter = counter_dev; a.list = children_dev; a.capacity = capacity;
    return launch_classify<false, true>(t, a, n_parents, dims, stream);
}

int hu_mass_properties_level(hu_tape t, const double* parents_dev, uint32_t n_parents, double s,
                             const uint32_t dims[3], float step, float threshold, uint32_t* sums_dev,
                             uint32_t* counter_dev, double* children_dev, uint32_t capacity, void* stream)
{
    if (!t || !sums_dev || !counter_dev || (!children_dev && capacity) || (!parents_dev && n_parents))
        return fail(HU_ERR_BAD_ARG, "NULL argument");
    ClassifyArgs a;
    std::memset(&a, 0, sizeof(a));
    a.parents = parents_dev;
    a.s = s;
    a.step = step; a.thr = threshold;
    a.counter = counter_dev; a.list = children_dev; a.capacity = capacity;
    a.sums = sums_dev;
    return launch_classify<true, true>(t, a, n_parents, dims, stream);
}

int hu_mass_integrals(const double* parents_dev, const uint32_t* sums_dev, uint32_t n_parents, double s,
                      double* out10_dev, void* stream)
{
    if (!out10_dev || ((!parents_dev || !sums_dev) && n_parents)) return fail(HU_ERR_BAD_ARG, "NULL argument");
    hipLaunchKernelGGL(k_mass_integrals, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const double4*)parents_dev,
                       sums_dev, n_parents, s, out10_dev);
    HU_HIP(hipGetLastError());
    return HU_OK;
}

}  // extern "C"
