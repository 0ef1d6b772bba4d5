// codecad_amd/csrc/kernels.hpp
//
// The gfx950 kernels of the hot path, generic over HOW a voxel is evaluated:
//   * InterpEval<DO>: the tape interpreter (interp.hpp run_tape), registers in LDS;
//   * a specialised evaluator generated per tape and compiled with hipRTC (jit in hip_util.hip).
// Reference counterparts (paths relative to /root/reference/codecad/):
//   k_grid_eval            grid_eval.cl:2-34 (both layouts), dense slab of a logical grid
//   k_grid_eval_blocks     the per-leaf-block launches of rendering/mesh.py:53-60, batched
//   k_classify<MASS,BATCH> subdivision.cl:12-30 and mass_properties.cl:7-56, either one block
//                          (reference-shaped) or every parent of a level in one launch
//   k_ray_caster           rendering/ray_caster.cl:146-256 as a per-lane state machine
//   k_bitmap               rendering/bitmap.cl:1-18
//   k_process_polygon      rendering/polygon2d.cl:82-175, one block (reference-shaped) or all blocks
#pragma once

#include "interp.hpp"

// Occupancy hint for experiments (-DSDF_WAVES_PER_EU=n): waves per SIMD the register allocator aims for.
#ifdef SDF_WAVES_PER_EU
#define SDF_KERNEL_ATTRS __attribute__((amdgpu_waves_per_eu(SDF_WAVES_PER_EU, SDF_WAVES_PER_EU)))
#else
#define SDF_KERNEL_ATTRS
#endif

namespace sdfk {

using sdf::Rec;

// Evaluate through the interpreter.  `prog` is the full or the distance-only program (DO).
template <bool DO> struct InterpEval {
    static constexpr bool kBricks = false;   // every primitive is evaluated anyway: runs along z (k_grid_eval)
    const Rec* prog;
    const float* extra;
    uint32_t n4;  // float4 slots of the LDS register file (scalar slots follow them)
    template <class T> __device__ __forceinline__ sdf::V4<T> operator()(T px, T py, T pz, void* lds) const
    {
        sdf::Regs<T> regs(lds, threadIdx.x, blockDim.x, n4);
        return sdf::run_tape<T, DO>(prog, extra, px, py, pz, regs);
    }
    // the distance alone (kernels that never look at the direction ask for this: per-tape code answers it
    // without its direction phase)
    template <class T> __device__ __forceinline__ T dist(T px, T py, T pz, void* lds) const { return (*this)(px, py, pz, lds).w; }
};

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------

// Sample position of reference grid_eval.cl:31 / subdivision.cl:22 / mass_properties.cl:25-27:
// corner + step * (float)gid, multiply then add, not fused.
__device__ __forceinline__ float sample(float corner, float step, uint32_t i) { return corner + step * (float)i; }

// A grid extent with the constants that divide by it: the kernels turn a linear cell index into (x, y, z)
// per lane, and a 32-bit division by a run-time value costs ~22 VALU instructions where
// multiply-high + shifts cost 5 (round-up method of Granlund & Montgomery, exact for every 32-bit x).
struct Dim {
    uint32_t n, m, s;
};
inline Dim make_dim(uint32_t n)
{
    if (n <= 1u) return Dim{n, 0u, 0u};
    uint32_t L = 0;
    while ((1ull << L) < n) ++L;
    return Dim{n, (uint32_t)((((1ull << L) - n) << 32) / n + 1ull), L - 1u};
}
__device__ __forceinline__ uint32_t div(uint32_t x, Dim d)
{
    const uint32_t t = __umulhi(d.m, x);
    const uint32_t q = (t + ((x - t) >> 1)) >> d.s;
    return d.n == 1u ? x : q;  // kernel-uniform select
}

// Grid stores.  Non-temporal stores (-DSDF_NT_STORES=1) were measured and are OFF: a 512^3 float4 grid of a
// store-bound tape takes 1.03 ms with `global_store_dwordx4 ... nt` (2.1 TB/s) against 0.40 ms with plain stores.
#ifndef SDF_NT_STORES
#define SDF_NT_STORES 0
#endif
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_voxel(float4* p, float4 v)
{
#if SDF_NT_STORES
    f4v t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(p));
#else
    *p = v;
#endif
}
__device__ __forceinline__ void store_voxel(float* p, float v)
{
#if SDF_NT_STORES
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// N = voxels per lane (1: T = float, 2: T = packed float2, see interp.hpp)
template <int N> struct Pack { using T = float; };
template <> struct Pack<2> { using T = sdf::f2; };
__device__ __forceinline__ float pack(const float (&v)[1]) { return v[0]; }
__device__ __forceinline__ sdf::f2 pack(const float (&v)[2]) { return sdf::make_f2(v[0], v[1]); }

// The N cells a lane owns of a grid with `n_cells` cells (z fastest): linear indices lin0, lin0 + stride, ...
// stride 1 (neighbours along z; the classification kernels): coordinates by one divide for the first cell and
// carries for the rest.  stride 64 (the grid kernels): CONSECUTIVE LANES own consecutive cells, so every store
// instruction of a wavefront writes one contiguous run (64 x 16 B = eight full 128-byte lines for a float4 grid)
// instead of every other 16 bytes of a run twice as long; each cell gets its own divide (5 instructions).
template <int N> struct Cells {
    uint32_t x[N], y[N], z[N];
    bool active[N];
    __device__ __forceinline__ Cells(uint32_t lin0, uint32_t n_cells, Dim dy, Dim dz, uint32_t stride = 1u)
    {
        const uint32_t sy = dy.n, sz = dz.n;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (i == 0 || stride != 1u) {
                const uint32_t lin = lin0 + (uint32_t)i * stride;
                active[i] = lin < n_cells;
                const uint32_t l = active[i] ? lin : 0u;  // idle tail lanes follow the (uniform) tape harmlessly
                const uint32_t t = div(l, dz);
                z[i] = l - t * sz;
                x[i] = div(t, dy);
                y[i] = t - x[i] * sy;
            } else {
                active[i] = active[0] && (lin0 + i < n_cells);
                const bool wrap_z = z[i - 1] + 1u == sz;
                const bool wrap_y = wrap_z && (y[i - 1] + 1u == sy);
                z[i] = wrap_z ? 0u : z[i - 1] + 1u;
                y[i] = wrap_y ? 0u : (wrap_z ? y[i - 1] + 1u : y[i - 1]);
                x[i] = wrap_y ? x[i - 1] + 1u : x[i - 1];
            }
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
// x extent of a slab with n_cells cells (the brick path bounds its wavefronts with it)
__device__ __forceinline__ uint32_t sx_slab(uint32_t n_cells, uint32_t sy, uint32_t sz) { return n_cells / (sy * sz); }

// linear index of a lane's first grid cell and the stride to its next one: a wavefront owns 64 * N consecutive cells
template <int N> __device__ __forceinline__ uint32_t first_cell(uint32_t block)
{
    return (block * blockDim.x + (threadIdx.x & ~63u)) * N + (threadIdx.x & 63u);
}
constexpr uint32_t kLaneStride = 64u;

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

// Sum over the 64 lanes of a wavefront; the total arrives in lane 63 (other lanes hold partial sums).
// Data-parallel-primitive adds (row shifts inside the rows of 16, then the two row broadcasts): six VALU
// instructions per value, where a __shfl_xor butterfly is six ds_bpermute round trips through the LDS crossbar.
__device__ __forceinline__ uint32_t wave_sum_to_last_lane(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8: lane 15 of a row = its total
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return v;
}

// ------------------------------------------------------------------------------------------
// boxes: per-tape code over compact bricks with the box's tables in LDS
// ------------------------------------------------------------------------------------------
// Where a box's voxels go: LAYOUT 0 float4[x][y][z] (z fastest) of a slab or block with extents (., sy, sz), LAYOUT 1 the
// PyMCubes order float[sy - 1 - y][xa + x][z] of a grid that is `sxa` wide (grid_eval.cl:18); `base` = elements before it.
struct BoxOut {
    void* out;
    size_t base;
    uint32_t sxa, sy, sz, xa;
    uint32_t nx;   // the slab's or block's extent along x (the boxes tile nx * sy * sz)
};
// A coordinate that the compiler must take as new in every iteration of a walk.  A lane's z never changes and its y only
// from column to column, so everything the tape computes from them alone is loop-invariant to the compiler -- and it hoists
// ALL of it (the partial sums of every general rotation: planetary's 27 frames held ~80 registers across the whole kernel,
// one or two wavefronts per SIMD), where the generator hoists what is worth a register (tape_pre_x) and tables the rest.
// Only for tapes pruned throughout (assemblies: many frames, few of them alive in a box); the others keep their code.
template <class E> __device__ __forceinline__ float walk_coordinate(float v)
{
    if constexpr (E::kPruneAll) return sdf::opaque(v);
    else return v;
}
// One workgroup, one BOX: up to 16 x 16 x 16 voxels at (x0, y0, z0) of a slab or block whose corner sample is (cx, cy, cz)
// [sample index of x: xs0 + x].  A wavefront evaluates compact 4 x 4 x 8 bricks (lane -> z: 8, y: 4, x: 2, its two voxels
// two x planes apart: a store instruction writes eight voxels along z per (x, y) row), the bricks of one (y, z) column of
// the box one after the other ALONG X.  Before the walks the workgroup fills the box's tables (specialise.hpp "AXIS
// TABLES", "PAIR TABLES"): what the tape computes from one coordinate, once per sample of that axis; then what it computes
// from two, once per PAIR of samples (16 x 16 evaluations where the walks would make 4096) -- the bars of a cross, any
// extruded profile.  The walks read, combine and store.  Extents: multiples of 4, 4 and 8 (the launchers).
// The tables of a box in LDS, filled by the whole workgroup (two barriers); -> where each table starts.
struct BoxTables {
    sdf::lds_float *x, *y, *z, *xy, *xz, *yz;
};
template <class E, class PR>
__device__ __forceinline__ BoxTables box_tables(const E& ev, float4* lds, float cx, float cy, float cz, float step, uint32_t xs0,
                                                uint32_t x0, uint32_t y0, uint32_t z0, uint32_t nx, uint32_t ny, uint32_t nz, const PR& pr)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    constexpr int NX = E::kTabXX, NY = E::kTabXY, NZ = E::kTabXZ, NXY = E::kPairXY, NXZ = E::kPairXZ, NYZ = E::kPairYZ;
    using Tabs = sdf::BoxTabs;
    sdf::lds_float* const tx = (sdf::lds_float*)lds;
    sdf::lds_float* const ty = tx + NX * Tabs::kAxis;
    sdf::lds_float* const tz = ty + NY * Tabs::kAxis;
    sdf::lds_float* const txy = tz + NZ * Tabs::kAxis;
    sdf::lds_float* const txz = txy + NXY * Tabs::kPairX;
    sdf::lds_float* const tyz = txz + NXZ * Tabs::kPairX;
    if constexpr (NX + NY + NZ > 0) {
        if (wave == 0u) {
            if constexpr (NX > 0)
                if (lane < nx) ev.template tab_x_x<Tabs::kAxis>(sample(cx, step, xs0 + x0 + lane), tx + lane);
        } else if (wave == 1u) {
            if constexpr (NY > 0)
                if (lane < ny) ev.template tab_x_y<Tabs::kAxis>(sample(cy, step, y0 + lane), ty + lane);
        } else if (wave == 2u) {
            if constexpr (NZ > 0)
                if (lane < nz) ev.template tab_x_z<Tabs::kAxis>(sample(cz, step, z0 + lane), tz + lane);
        }
        __syncthreads();
    }
    if constexpr (NXY + NXZ + NYZ > 0) {
        // one entry per lane: (row, column) = (thread / 16, thread % 16); the column is the table's fastest index
        const uint32_t r = threadIdx.x >> 4, c = threadIdx.x & 15u;
        if constexpr (NXY > 0)
            if (r < ny && c < nx)
                ev.template tab_x_xy<Tabs::kPairX>(sample(cx, step, xs0 + x0 + c), sample(cy, step, y0 + r), Tabs{tx + c, ty + r, tz, txy, txz, tyz}, pr,
                                                   txy + (r * Tabs::kRowX + c));
        if constexpr (NXZ > 0)
            if (r < nz && c < nx)
                ev.template tab_x_xz<Tabs::kPairX>(sample(cx, step, xs0 + x0 + c), sample(cz, step, z0 + r), Tabs{tx + c, ty, tz + r, txy, txz, tyz}, pr,
                                                   txz + (r * Tabs::kRowX + c));
        if constexpr (NYZ > 0)
            if (r < ny && c < nz)
                ev.template tab_x_yz<Tabs::kPairYZ>(sample(cy, step, y0 + r), sample(cz, step, z0 + c), Tabs{tx, ty + r, tz + c, txy, txz, tyz}, pr,
                                                    tyz + (r * Tabs::kRowYZ + c));
        __syncthreads();
    }
    return BoxTables{tx, ty, tz, txy, txz, tyz};
}

// RAGGED (round 4): the box may end anywhere -- a grid whose extents are no multiples of (4, 4, 8) --: the bricks at its rim
// are walked whole (their lanes beyond the rim read table entries nobody filled and compute on them: every value is per
// lane, the op library has no data-dependent loop) and only the voxels inside are stored.  Before, one extent of 250
// instead of 256 sent the WHOLE launch over runs of cells, in the form without tables and without pruning: planetary's
// float4 grid 0.55 -> 4.7 ms, sponge(4)'s 0.069 -> 0.123 ms (tools/experiments/ragged_time.py).
template <class E, int LAYOUT, int N, bool RAGGED = false>
__device__ __forceinline__ void box_eval(const E& ev, float4* lds, float cx, float cy, float cz, float step, uint32_t xs0,
                                         uint32_t x0, uint32_t y0, uint32_t z0, const BoxOut& o, const uint32_t* __restrict__ masks,
                                         uint32_t box = blockIdx.x)
{
    using T = typename Pack<N>::T;
    using Tabs = sdf::BoxTabs;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t sx = o.sxa, sy = o.sy, sz = o.sz;
    const uint32_t nx = min(16u, o.nx - x0), ny = min(16u, sy - y0), nz = min(16u, sz - z0);
    // box pruning: which operands of the tape's selects can win anywhere in this box (k_box_masks ran before this launch)
    const sdf::Prune<E::kPruneWords> pr = sdf::load_prune<E::kPruneWords>(masks, box);
    const BoxTables t = box_tables(ev, lds, cx, cy, cz, step, xs0, x0, y0, z0, nx, ny, nz, pr);
    sdf::lds_float* const tx = t.x; sdf::lds_float* const ty = t.y; sdf::lds_float* const tz = t.z;
    sdf::lds_float* const txy = t.xy; sdf::lds_float* const txz = t.xz; sdf::lds_float* const tyz = t.yz;
    // The box's (y, z) columns of bricks, at most eight: a wavefront takes column `wave` and the one four on -- the same
    // bricks along z, two (or, in a box eight voxels deep, four) rows of bricks further along y: what changes from its
    // first column to its second are y and a few pointers, by constants.
    const uint32_t nbz = RAGGED ? (nz + 7u) >> 3 : nz >> 3;        // 1 or 2
    const uint32_t bz = nbz == 2u ? (wave & 1u) : 0u, dby = 4u / nbz;
    const uint32_t zl = bz * 8u + (lane & 7u), z = z0 + zl, xl = lane >> 5;
    const bool z_inside = !RAGGED || zl < nz;
    const float pz = sample(cz, step, z);                          // (one number per lane: its voxels differ in x)
    uint32_t yl = (nbz == 2u ? (wave >> 1) : wave) * 4u + ((lane >> 3) & 3u);
    Tabs col{tx + xl, ty + yl, tz + zl, txy + (yl * Tabs::kRowX + xl), txz + (zl * Tabs::kRowX + xl), tyz + (yl * Tabs::kRowYZ + zl)};
    // where the lane's first voxel of a column goes, and the steps to its second voxel, the next brick, the next column
    const size_t plane = (size_t)sy * sz;
    size_t at = LAYOUT == 0 ? o.base + ((size_t)z + (size_t)sz * ((size_t)(y0 + yl) + (size_t)sy * (x0 + xl)))
                            : o.base + ((size_t)z + ((size_t)(o.xa + x0 + xl) + (size_t)(sy - 1u - (y0 + yl)) * sx) * sz);
    const size_t second = LAYOUT == 0 ? 2u * plane : 2u * (size_t)sz, brick = 2u * second;
    const size_t rows = (size_t)(4u * dby) * (LAYOUT == 0 ? (size_t)sz : (size_t)sx * sz);   // LAYOUT 1 runs backwards along y
    const uint32_t row_in_brick = (lane >> 3) & 3u, bricks_x = RAGGED ? (nx + 3u) >> 2 : nx >> 2;
    for (; (RAGGED ? yl - row_in_brick : yl) < ny; yl += 4u * dby) {     // (RAGGED: while the brick's first row is inside -- wave-uniform)
        const float py = sample(cy, step, y0 + yl);
        const bool row_inside = !RAGGED || (z_inside && yl < ny);
        Tabs tb = col;
        const auto hoisted = ev.hoist_x(py, walk_coordinate<E>(pz), tb, pr);
        size_t p = at;
#pragma unroll 1
        for (uint32_t j = 0; j < bricks_x; ++j) {
            // (the columns of y and z that `pre` does not hold are read again in every brick, which the compiler would
            // otherwise undo; unrolling the walk and batching its reads were measured: no gain)
            asm volatile("" ::: "memory");
            float xs[N];
#pragma unroll
            for (int i = 0; i < N; ++i) xs[i] = sample(cx, step, xs0 + x0 + j * 4u + xl + 2u * i);
            const T px = pack(xs);
            const float pyb = walk_coordinate<E>(py), pzb = walk_coordinate<E>(pz);
            const auto prb = pr.fresh();
            if (LAYOUT == 0) {
                const sdf::V4<T> r = ev.eval_hoisted_x(px, pyb, pzb, hoisted, tb, prb);
#pragma unroll
                for (int i = 0; i < N; ++i)
                    if (!RAGGED || (row_inside && j * 4u + xl + 2u * (uint32_t)i < nx))
                        store_voxel(static_cast<float4*>(o.out) + p + (size_t)i * second, sdf::voxel(r, i));
            } else {
                const T w = ev.dist_hoisted_x(px, pyb, pzb, hoisted, tb, prb);
#pragma unroll
                for (int i = 0; i < N; ++i)
                    if (!RAGGED || (row_inside && j * 4u + xl + 2u * (uint32_t)i < nx))
                        store_voxel(static_cast<float*>(o.out) + p + (size_t)i * second, sdf::get(w, i));
            }
            tb.x += 4; tb.xy += 4; tb.xz += 4;
            p += brick;
        }
        col.y += 4u * dby; col.xy += 4u * dby * Tabs::kRowX; col.yz += 4u * dby * Tabs::kRowYZ;
        at = LAYOUT == 0 ? at + rows : at - rows;
    }
}

// ------------------------------------------------------------------------------------------
// dense grid evaluation
// ------------------------------------------------------------------------------------------
// `boxes` != 0 (the launcher sets it for per-tape code with deferred directions when the slab's extents are multiples of
// 4, 4 and 8): a workgroup takes a 16^3 box of the slab (box_eval above) instead of runs of cells along z.  What compact
// bricks buy besides the tables: values that are uniform over a wavefront stay uniform far more often -- which primitive
// of a CSG tree is nearest (sponge(4) at 512^3: 1.8 distinct winners per brick against 3.1 per run; per-tape code
// computes the direction once per DISTINCT winner), and whether any lane is in the corner region of a rectangle.
// Evaluators with box code (E::kBricks) have TWO kernels per layout: this one over whole bricks, and k_grid_eval_ragged for
// slabs whose extents are no multiples of (4, 4, 8) (until late in round 4 that one went over RUNS, in the in-place form).
// One kernel holding two paths is allocated the registers of the hungrier one: planetary's in-place evaluation (256
// registers, two wavefronts per SIMD) sat on its box path (175) until they were split -- 1.86 against 1.03 ms.
template <class E, int LAYOUT, int N>
__device__ __forceinline__ void grid_eval_runs(const E& ev, float4* lds, float cx, float cy, float cz, float step, uint32_t sx, Dim dy, Dim dz,
                                               uint32_t x0, uint32_t n_cells, void* __restrict__ out);

// a slab over boxes: workgroup -> box (boxes along z fastest), in the order below
template <class E, int LAYOUT, int N, bool RAGGED>
__device__ __forceinline__ void grid_eval_boxes(const E& ev, float4* lds, float cx, float cy, float cz, float step, uint32_t sx, uint32_t sy, uint32_t sz,
                                                uint32_t x0, uint32_t n_cells, void* __restrict__ out, const uint32_t* __restrict__ masks)
{
    const uint32_t nx_slab = sx_slab(n_cells, sy, sz);
    const uint32_t boxes_z = (sz + 15u) >> 4, boxes_y = (sy + 15u) >> 4;
    // BOX ORDER.  Workgroup i runs on XCD i % 8, each XCD behind its own L2.  Boxes taken in launch order put neighbours
    // along z on different XCDs at the same time -- and in the float layout two boxes along z share every 128-byte line of
    // the output (a box's row is 16 floats): both L2s hold half-written lines.  So every XCD takes a CONTIGUOUS eighth of
    // the boxes, its workgroups walking it in order: neighbours meet in one L2.  Measured, 512^3, per-tape code over boxes
    // (profiles/r04_hbm_sweep.jsonl): float grids 0.16-0.18 -> 0.12-0.14 ms (csg_example 0.177 -> 0.116, sponge(4) 0.171 ->
    // 0.138), float4 grids of light tapes -2...-4 % (whole lines either way), sponge(4)'s unchanged.  -DSDF_BOX_ORDER=0: off.
    uint32_t b = blockIdx.x;
#if !defined(SDF_BOX_ORDER) || SDF_BOX_ORDER
    {
        const uint32_t k = b & 7u, q = gridDim.x >> 3, r = gridDim.x & 7u;     // XCD k takes q + (k < r) boxes
        b = k * q + (k < r ? k : r) + (b >> 3);
    }
#endif
    const uint32_t qz = b % boxes_z, qt = b / boxes_z, qy = qt % boxes_y, qx = qt / boxes_y;
    const BoxOut o{out, 0, sx, sy, sz, LAYOUT == 0 ? 0u : x0, nx_slab};
    box_eval<E, LAYOUT, N, RAGGED>(ev, lds, cx, cy, cz, step, x0, qx * 16u, qy * 16u, qz * 16u, o, masks, b);
}

template <class E, int LAYOUT, int N>
__global__ void __launch_bounds__(256) SDF_KERNEL_ATTRS
k_grid_eval(const E ev, float cx, float cy, float cz, float step, uint32_t sx, Dim dy, Dim dz, uint32_t x0,
            uint32_t n_cells, uint32_t boxes, void* __restrict__ out, const uint32_t* __restrict__ masks)
{
    extern __shared__ float4 lds[];
    if constexpr (E::kBricks && N == 2) {
        // (extents that are multiples of (4, 4, 8): the launchers send anything else to k_grid_eval_ragged)
        grid_eval_boxes<E, LAYOUT, N, false>(ev, lds, cx, cy, cz, step, sx, dy.n, dz.n, x0, n_cells, out, masks);
    } else {
        grid_eval_runs<E, LAYOUT, N>(ev, lds, cx, cy, cz, step, sx, dy, dz, x0, n_cells, out);
    }
}
// the same over boxes that may end anywhere (box_eval RAGGED): a kernel of its own, because its predicated stores and its
// longer live ranges are not for the aligned launches to pay
template <class E, int LAYOUT, int N>
__global__ void __launch_bounds__(256) SDF_KERNEL_ATTRS
k_grid_eval_ragged(const E ev, float cx, float cy, float cz, float step, uint32_t sx, Dim dy, Dim dz, uint32_t x0,
                   uint32_t n_cells, uint32_t boxes, void* __restrict__ out, const uint32_t* __restrict__ masks)
{
    extern __shared__ float4 lds[];
    if constexpr (E::kBricks && N == 2) grid_eval_boxes<E, LAYOUT, N, true>(ev, lds, cx, cy, cz, step, sx, dy.n, dz.n, x0, n_cells, out, masks);
}

// ... and over RUNS of cells, in the in-place form, for evaluators with box code: slabs in which boxes would be mostly padding
// (a 2D grid is one voxel deep: seven of a brick's eight z lanes idle -- 2048^2 float4: 0.081 ms over ragged boxes, slower
// than interpreted; the launchers take this kernel where less than half of the padded bricks' voxels exist)
template <class E, int LAYOUT, int N>
__global__ void __launch_bounds__(256) SDF_KERNEL_ATTRS
k_grid_eval_runs(const E ev, float cx, float cy, float cz, float step, uint32_t sx, Dim dy, Dim dz, uint32_t x0,
                 uint32_t n_cells, void* __restrict__ out)
{
    extern __shared__ float4 lds[];
    if constexpr (E::kBricks) grid_eval_runs<E, LAYOUT, N>(ev, lds, cx, cy, cz, step, sx, dy, dz, x0, n_cells, out);
}

template <class E, int LAYOUT, int N>
__device__ __forceinline__ void grid_eval_runs(const E& ev, float4* lds, float cx, float cy, float cz, float step, uint32_t sx, Dim dy, Dim dz,
                                               uint32_t x0, uint32_t n_cells, void* __restrict__ out)
{
    const uint32_t sy = dy.n, sz = dz.n;
    using T = typename Pack<N>::T;
    const uint32_t lin0 = first_cell<N>(blockIdx.x);
    const Cells<N> c(lin0, n_cells, dy, dz, kLaneStride);
    const T px = c.position(cx, step, c.x, x0), py = c.position(cy, step, c.y), pz = c.position(cz, step, c.z);
    if (LAYOUT == 0) {
        const sdf::V4<T> r = ev(px, py, pz, lds);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (!c.active[i]) continue;
            // INDEX3 = z + sz*(y + sy*x) (cl_util/indexing.h:4): inside a slab this is the linear
            // cell index; each store instruction of a wavefront writes 1 KiB contiguous.
            store_voxel(static_cast<float4*>(out) + lin0 + (uint32_t)i * kLaneStride, sdf::voxel(r, i));
        }
    } else {
        const T w = ev.dist(px, py, pz, lds);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (!c.active[i]) continue;
            // grid_eval.cl:18: z + (x + (sy-1-y)*sx)*sz; z-fastest so a wave stores contiguous runs
            const size_t idx = (size_t)c.z[i] + ((size_t)(x0 + c.x[i]) + (size_t)(sy - 1u - c.y[i]) * sx) * sz;
            store_voxel(static_cast<float*>(out) + idx, sdf::get(w, i));
        }
    }
}

template <class E, int LAYOUT, int N, bool BOXES, bool RAGGED = false>
__device__ __forceinline__ void grid_eval_blocks_body(const E& ev, float4* lds, const int4* __restrict__ blocks, const uint32_t* __restrict__ n_blocks_dev,
                                                      uint32_t b0, uint32_t chunks, uint32_t bricks, double res, double ox, double oy, double oz, float step,
                                                      uint32_t sx, Dim dy, Dim dz, void* __restrict__ out, const uint32_t* __restrict__ masks)
{
    const uint32_t sy = dy.n, sz = dz.n;
    using T = typename Pack<N>::T;
    // b0: the first block of this launch (a list too long for one grid is launched in pieces)
    const uint32_t chunk = blockIdx.x % chunks, b = b0 + blockIdx.x / chunks;
    // indirect form: the list length lives on the device (the launch is sized for its capacity), so a
    // traversal needs no host round trip between its levels; workgroup-uniform exit
    if (n_blocks_dev && b >= *n_blocks_dev) return;
    const uint32_t cells = sx * sy * sz;
    const int4 ic = blocks[b];
    // subdivision.py:100: pos = int_pos * resolution + origin (fp64), cast once (geometry.py:98-99)
    const float cx = (float)((double)ic.x * res + ox);
    const float cy = (float)((double)ic.y * res + oy);
    const float cz = (float)((double)ic.z * res + oz);
    if constexpr (BOXES) {
        // a workgroup takes one BOX of the block (box_eval): `chunks` = boxes per block, `bricks` = the boxes along y
        // and z packed as (boxes_y << 16 | boxes_z); a 16^3 block is one box
        const uint32_t boxes_z = bricks & 0xffffu, boxes_y = bricks >> 16;
        const uint32_t qz = chunk % boxes_z, qt = chunk / boxes_z, qy = qt % boxes_y, qx = qt / boxes_y;
        const BoxOut o{out, (size_t)b * cells, sx, sy, sz, 0u, sx};
        box_eval<E, LAYOUT, N, RAGGED>(ev, lds, cx, cy, cz, step, 0u, qx * 16u, qy * 16u, qz * 16u, o, masks);
        return;
    }
    const uint32_t lin0 = first_cell<N>(chunk);
    const Cells<N> c(lin0, cells, dy, dz, kLaneStride);
    const T px = c.position(cx, step, c.x), py = c.position(cy, step, c.y), pz = c.position(cz, step, c.z);
    const size_t base = (size_t)b * cells;
    if (LAYOUT == 0) {
        const sdf::V4<T> r = ev(px, py, pz, lds);
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (c.active[i]) store_voxel(static_cast<float4*>(out) + base + lin0 + (uint32_t)i * kLaneStride, sdf::voxel(r, i));
    } else {
        const T w = ev.dist(px, py, pz, lds);
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (c.active[i])
                store_voxel(static_cast<float*>(out) + base + (size_t)c.z[i] + ((size_t)c.x[i] + (size_t)(sy - 1u - c.y[i]) * sx) * sz, sdf::get(w, i));
    }
}
// (two kernels for evaluators with box code, like k_grid_eval / k_grid_eval_ragged)
template <class E, int LAYOUT, int N>
__global__ void __launch_bounds__(256) SDF_KERNEL_ATTRS
k_grid_eval_blocks(const E ev, const int4* __restrict__ blocks, const uint32_t* __restrict__ n_blocks_dev, uint32_t b0,
                   uint32_t chunks, uint32_t bricks, double res, double ox, double oy, double oz, float step, uint32_t sx,
                   Dim dy, Dim dz, void* __restrict__ out, const uint32_t* __restrict__ masks)
{
    extern __shared__ float4 lds[];
    grid_eval_blocks_body<E, LAYOUT, N, (E::kBricks && N == 2)>(ev, lds, blocks, n_blocks_dev, b0, chunks, bricks, res, ox, oy, oz, step, sx, dy, dz, out, masks);
}
template <class E, int LAYOUT, int N>
__global__ void __launch_bounds__(256) SDF_KERNEL_ATTRS
k_grid_eval_blocks_ragged(const E ev, const int4* __restrict__ blocks, const uint32_t* __restrict__ n_blocks_dev, uint32_t b0,
                          uint32_t chunks, uint32_t bricks, double res, double ox, double oy, double oz, float step, uint32_t sx,
                          Dim dy, Dim dz, void* __restrict__ out, const uint32_t* __restrict__ masks)
{
    extern __shared__ float4 lds[];
    if constexpr (E::kBricks && N == 2)     // blocks whose extents are no multiples of (4, 4, 8): boxes that end anywhere (box_eval RAGGED)
        grid_eval_blocks_body<E, LAYOUT, N, true, true>(ev, lds, blocks, n_blocks_dev, b0, chunks, bricks, res, ox, oy, oz, step, sx, dy, dz, out, masks);
}

// (... and over runs of cells, for blocks in which boxes would be mostly padding: k_grid_eval_runs)
template <class E, int LAYOUT, int N>
__global__ void __launch_bounds__(256) SDF_KERNEL_ATTRS
k_grid_eval_blocks_runs(const E ev, const int4* __restrict__ blocks, const uint32_t* __restrict__ n_blocks_dev, uint32_t b0,
                        uint32_t chunks, double res, double ox, double oy, double oz, float step, uint32_t sx, Dim dy, Dim dz,
                        void* __restrict__ out)
{
    extern __shared__ float4 lds[];
    if constexpr (E::kBricks)
        grid_eval_blocks_body<E, LAYOUT, N, false>(ev, lds, blocks, n_blocks_dev, b0, chunks, 0u, res, ox, oy, oz, step, sx, dy, dz, out, nullptr);
}

// ------------------------------------------------------------------------------------------
// classification kernels: subdivision_step / mass_properties, single block or whole level
// ------------------------------------------------------------------------------------------
struct ClassifyArgs {
    const void* parents;   // BATCH: int4[] (subdivision) or double4[] (mass); else unused
    uint32_t parent_base;           // BATCH: the first parent of this launch (a level too long for one grid is launched in pieces)
    const uint32_t* n_parents_dev;  // BATCH, optional: the number of parents, on the device (the launch is sized for the list's capacity)
    uint32_t chunks;       // workgroups per parent
    uint32_t sx, sy, sz;
    Dim dy, dz;            // sy, sz with their division constants
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
    uint32_t scratch_offset;  // bytes of LDS taken by the register file or a box's tables (scratch follows)
    uint32_t boxes;           // per-tape code over boxes (box_classify): boxes along y << 16 | boxes along z; chunks = boxes per parent
    const uint32_t* masks;    // boxes: the pruning masks of this launch's workgroups (k_box_masks), or NULL
    // OWNERSHIP (multi-GPU, codecad_amd/dist.py "replicated levels"): several ranks classify the SAME parents, and each keeps
    // only the cells it owns -- owner = mix(hash of the parent's row, the cell's linear index) mod own.n -- both in the list
    // and in the moment sums.  A cell's owner depends on what the cell is, not on where its parent stands in anybody's
    // list, so the ranks' lists partition the level however each rank's atomics ordered its own.  own.n <= 1: keep all.
    Dim own;
    uint32_t own_rank;
};
__device__ __forceinline__ uint32_t row_hash(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    uint32_t h = a * 0x9e3779b1u;
    h = (h ^ (h >> 15)) + b * 0x85ebca77u;
    h = (h ^ (h >> 13)) + c * 0xc2b2ae3du;
    h = (h ^ (h >> 16)) + d * 0x27d4eb2fu;
    return h ^ (h >> 15);
}
__device__ __forceinline__ bool owned(const ClassifyArgs& a, uint32_t parent_hash, uint32_t cell)
{
    // (straight-line: a mix -- plain (hash + cell) mod n would stripe a 16-wide grid over 8 ranks by z alone, and the sponge's
    // survivors are anything but uniform in z: shares of 3193..4499 where the mix gives 3850 +- 2 % --, then a multiply-high,
    // two shifts, a multiply; own.n <= 1 decides by a kernel-uniform select)
    uint32_t v = parent_hash + cell * 0x9e3779b1u;
    v = (v ^ (v >> 15)) * 0x85ebca77u;
    v ^= v >> 13;
    const bool mine = v - div(v, a.own) * a.own.n == a.own_rank;
    return a.own.n <= 1u ? true : mine;
}

template <class E, bool MASS, bool BATCH, int N>
__global__ void __launch_bounds__(256) k_classify(const E ev, const ClassifyArgs a)
{
    using T = typename Pack<N>::T;
    extern __shared__ float4 lds[];
    uint32_t* scratch = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(lds) + a.scratch_offset);  // [0..4] compaction, [8..17] sums
    const uint32_t b = BATCH ? a.parent_base + blockIdx.x / a.chunks : 0u;
    const uint32_t chunk = BATCH ? blockIdx.x % a.chunks : blockIdx.x;
    const uint32_t cells = a.sx * a.sy * a.sz;
    if (BATCH && a.n_parents_dev && b >= *a.n_parents_dev) return;   // workgroup-uniform, before any barrier

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
    // what identifies this parent on every rank that classifies it (ownership): its row
    const uint32_t phash = !BATCH ? 0u
                           : MASS ? row_hash((uint32_t)__double2loint(pcx) ^ (uint32_t)__double2hiint(pcx), (uint32_t)__double2loint(pcy) ^ (uint32_t)__double2hiint(pcy),
                                             (uint32_t)__double2loint(pcz) ^ (uint32_t)__double2hiint(pcz), (uint32_t)__double2loint(pcw) ^ (uint32_t)__double2hiint(pcw))
                                  : row_hash((uint32_t)ipar.x, (uint32_t)ipar.y, (uint32_t)ipar.z, (uint32_t)ipar.w);

    if constexpr (E::kBricks && N == 2) {
        if (a.boxes) {
            // Per-tape code over a BOX of the parent's grid (box_eval: tables in LDS, bricks walked along x); what differs from
            // the path below: cells are compacted wavefront by wavefront (one global atomic per brick that has ambiguous
            // cells: no barrier inside the walks), the moment sums are kept per lane over a wavefront's bricks.
            using Tabs = sdf::BoxTabs;
            const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
            const uint32_t boxes_z = a.boxes & 0xffffu, boxes_y = a.boxes >> 16;
            const uint32_t qz = chunk % boxes_z, qt = chunk / boxes_z, qy = qt % boxes_y, qx = qt / boxes_y;
            const uint32_t x0 = qx * 16u, y0 = qy * 16u, z0 = qz * 16u;
            const uint32_t nx = min(16u, a.sx - x0), ny = min(16u, a.sy - y0), nz = min(16u, a.sz - z0);
            const sdf::Prune<E::kPruneWords> pr = sdf::load_prune<E::kPruneWords>(a.masks, blockIdx.x);
            const BoxTables t = box_tables(ev, lds, cx, cy, cz, a.step, 0u, x0, y0, z0, nx, ny, nz, pr);
            if (MASS) __syncthreads();   // scratch[8..17] zeroed (a tape without tables has no barrier in box_tables)
            const bool nothing_ambiguous = MASS && a.thr == 0.0f;
            const uint64_t below = (1ull << lane) - 1ull;
            uint32_t v[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            constexpr uint32_t kStep = (uint32_t)Tabs::kPairStep;      // x distance of a lane's two voxels
            // one brick: the lane's two voxels at (xv, y, z) and (xv + kStep, y, z); `live`: the lane's row is inside the box
            // (a grid that is no multiple of (4, 4, 8): bricks at a box's rim are walked whole, `live` and x_end keep what lies
            // beyond the rim out of the sums and the lists -- as box_eval RAGGED does for the grid kernels' stores)
            const uint32_t x_end = x0 + nx;
            auto brick = [&](uint32_t xv, uint32_t y, uint32_t z, float py, float pz, const Tabs& tb, const auto& hoisted, bool live) {
                const T px = sdf::make_f2(sample(cx, a.step, xv), sample(cx, a.step, xv + kStep));
                const T w = ev.dist_hoisted_x(px, walk_coordinate<E>(py), walk_coordinate<E>(pz), hoisted, tb, pr.fresh());
                bool amb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float wi = sdf::get(w, i);
                    const uint32_t x = xv + kStep * (uint32_t)i;
                    const bool mine = live & (x < x_end) & owned(a, phash, z + a.sz * (y + a.sy * x));
                    if (MASS) {
                        // mass_properties.cl:31-52: inside (w <= -thr) -> moments of the integer cell index; else w < thr -> ambiguous
                        const bool in = wi <= -a.thr, inside = in && mine;
                        amb[i] = mine && !in && (wi < a.thr);
                        const uint32_t m = inside ? 1u : 0u;
                        const uint32_t xm = inside ? x : 0u, ym = inside ? y : 0u, zm = inside ? z : 0u;
                        v[0] += __umul24(xm, x); v[1] += __umul24(xm, y); v[2] += __umul24(xm, z); v[3] += xm;
                        v[4] += __umul24(ym, y); v[5] += __umul24(ym, z); v[6] += ym;
                        v[7] += __umul24(zm, z); v[8] += zm; v[9] += m;
                    } else {
                        amb[i] = mine && (wi > -a.thr) && (wi < a.thr);   // subdivision.cl:25
                    }
                }
                if (!nothing_ambiguous) {
                    const uint64_t m0 = __ballot(amb[0]), m1 = __ballot(amb[1]);
                    const uint32_t n0 = __popcll(m0), total = n0 + __popcll(m1);
                    if (total) {   // wave-uniform
                        uint32_t base = 0u;
                        if (lane == 0u) base = atomicAdd(a.counter, total);
                        base = __builtin_amdgcn_readfirstlane(base);
                        const uint32_t slot[2] = {base + (uint32_t)__popcll(m0 & below), base + n0 + (uint32_t)__popcll(m1 & below)};
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            if (!(amb[i] && slot[i] < a.capacity)) continue;
                            const uint32_t x = xv + kStep * (uint32_t)i;
                            if (!BATCH) {
                                static_cast<uchar4*>(a.list)[slot[i]] = make_uchar4((unsigned char)x, (unsigned char)y, (unsigned char)z, 0);
                            } else if (MASS) {
                                static_cast<double4*>(a.list)[slot[i]] =
                                    make_double4((double)x * a.s + pcx, (double)y * a.s + pcy, (double)z * a.s + pcz, pcw);
                            } else {
                                static_cast<int4*>(a.list)[slot[i]] =
                                    make_int4(ipar.x + (int)x * a.int_step, ipar.y + (int)y * a.int_step, ipar.z + (int)z * a.int_step, ipar.w);
                            }
                        }
                    }
                }
            };
            const uint32_t nbz = (nz + 7u) >> 3, bz = nbz == 2u ? (wave & 1u) : 0u, dby = 4u / nbz;
            const uint32_t zl = bz * 8u + (lane & 7u), z = z0 + zl, xl = lane >> 5;
            const float pz = sample(cz, a.step, z);
            const uint32_t row_in_brick = (lane >> 3) & 3u;
            for (uint32_t yl = (nbz == 2u ? (wave >> 1) : wave) * 4u + row_in_brick; yl - row_in_brick < ny; yl += 4u * dby) {   // (wave-uniform)
                const uint32_t y = y0 + yl;
                const bool live = (zl < nz) & (yl < ny);
                const float py = sample(cy, a.step, y);
                Tabs tb{t.x + xl, t.y + yl, t.z + zl, t.xy + (yl * Tabs::kRowX + xl), t.xz + (zl * Tabs::kRowX + xl), t.yz + (yl * Tabs::kRowYZ + zl)};
                const auto hoisted = ev.hoist_x(py, walk_coordinate<E>(pz), tb, pr);
#pragma unroll 1
                for (uint32_t j = 0; j < ((nx + 3u) >> 2); ++j) {
                    asm volatile("" ::: "memory");
                    brick(x0 + j * 4u + xl, y, z, py, pz, tb, hoisted, live);
                    tb.x += 4; tb.xy += 4; tb.xz += 4;
                }
            }
            if (MASS) {
#pragma unroll
                for (int i = 0; i < 10; ++i) {
                    const uint32_t sum = wave_sum_to_last_lane(v[i]);
                    if (lane == 63u && sum) atomicAdd(&scratch[8 + i], sum);
                }
                __syncthreads();
                if (threadIdx.x < 10) {
                    const uint32_t total = scratch[8 + threadIdx.x];
                    if (total) atomicAdd(&a.sums[(size_t)b * 10 + threadIdx.x], total);
                }
            }
            return;
        }
    }

    const uint32_t lin0 = (chunk * blockDim.x + threadIdx.x) * N;
    const Cells<N> c(lin0, cells, a.dy, a.dz);
    const T w = ev.dist(c.position(cx, a.step, c.x), c.position(cy, a.step, c.y), c.position(cz, a.step, c.z), lds);

    bool ambiguous[N];
    if (MASS) {
        // mass_properties.cl:31-52: inside (w <= -thr) -> moments of the integer cell index;
        // else w < thr -> ambiguous
        bool inside[N];
        bool any_inside = false;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float wi = sdf::get(w, i);
            // (no short-circuit around owned(): its division would become a divergent branch, in the unit built with
            // -structurizecfg-skip-uniform-regions of all places)
            const bool own = owned(a, phash, lin0 + (uint32_t)i);
            const bool mine = c.active[i] & own, in = wi <= -a.thr;
            inside[i] = mine & in;
            ambiguous[i] = mine & !in & (wi < a.thr);
            any_inside |= inside[i];
        }
        const uint64_t imask = __ballot(any_inside);
        __syncthreads();  // scratch[8..17] zeroed
        if (imask != 0ull) {  // wave-uniform
            uint32_t v[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < N; ++i) {
                // cell indices are below 256 (the launchers check): 24-bit multiplies, which are full rate
                const uint32_t m = inside[i] ? 1u : 0u;
                const uint32_t x = c.x[i], y = c.y[i], z = c.z[i];
                const uint32_t xm = inside[i] ? x : 0u, ym = inside[i] ? y : 0u, zm = inside[i] ? z : 0u;
                v[0] += __umul24(xm, x); v[1] += __umul24(xm, y); v[2] += __umul24(xm, z); v[3] += xm;
                v[4] += __umul24(ym, y); v[5] += __umul24(ym, z); v[6] += ym;
                v[7] += __umul24(zm, z); v[8] += zm; v[9] += m;
            }
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const uint32_t sum = wave_sum_to_last_lane(v[i]);
                if ((threadIdx.x & 63u) == 63u && sum) atomicAdd(&scratch[8 + i], sum);
            }
        }
    } else {
        // subdivision.cl:25: -thr < w < thr
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float wi = sdf::get(w, i);
            const bool own = owned(a, phash, lin0 + (uint32_t)i);
            ambiguous[i] = c.active[i] & own & (wi > -a.thr) & (wi < a.thr);
        }
    }

    // With a zero threshold (the leaf level of mass_properties) no sample can be ambiguous: nothing to compact,
    // one barrier (for the moment sums below) instead of the compaction's two.  Kernel-uniform.
    const bool nothing_ambiguous = MASS && a.thr == 0.0f;
    uint32_t slot[N];
    if (nothing_ambiguous) {
#pragma unroll
        for (int i = 0; i < N; ++i) slot[i] = 0u;
        __syncthreads();
    } else {
        wg_compact_slots<N>(ambiguous, a.counter, scratch, slot);  // has __syncthreads
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (nothing_ambiguous || !(ambiguous[i] && slot[i] < a.capacity)) continue;
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
// box pruning: the masks of a launch's boxes (specialise.hpp "BOX PRUNING")
// ------------------------------------------------------------------------------------------
// Runs BEFORE a launch of k_grid_eval / k_grid_eval_blocks / k_classify over boxes, on the same stream: one LANE per box
// (= per workgroup of that launch) evaluates the tape's bounded primitives at the box's centre and decides which operands of
// which selects can win anywhere in the box -> out[box * words ...].  What a box is -- which unit (slab, leaf block, parent)
// it belongs to, where its first sample lies -- is worked out exactly as the consuming kernel does.
struct MaskArgs {
    uint32_t mode;               // 0 dense slab, 1 leaf blocks (int4 rows), 2 one classified grid, 3 / 4 a level's parents (subdivision / mass)
    uint32_t n_boxes;            // workgroups of the launch this prepares
    uint32_t chunks;             // boxes per unit
    uint32_t boxes_y, boxes_z;
    uint32_t nx, ny, nz;         // a unit's extents in samples
    uint32_t xs0;                // mode 0: sample index of the slab's first plane
    uint32_t unit_base;          // first unit of the launch
    const void* units;           // modes 1, 3: int4[]; mode 4: double4[]
    const uint32_t* n_units_dev; // optional: the number of units, on the device
    float cx, cy, cz, step;      // modes 0, 2: the corner sample as given
    int32_t int_step, dimension; // mode 3
    double res, ox, oy, oz, s;   // modes 1, 3 (resolution + origin), 4 (cell size)
    uint32_t* out;
};
template <class E> __global__ void __launch_bounds__(64) k_box_masks(const E ev, const MaskArgs a)
{
    if constexpr (E::kBricks) {
        constexpr int W = E::kPruneWords;
        if constexpr (W > 0) {
            const uint32_t box = blockIdx.x * blockDim.x + threadIdx.x;
            if (box >= a.n_boxes) return;
            const uint32_t unit = a.unit_base + box / a.chunks, chunk = box % a.chunks;
            sdf::Prune<W> pr;
#pragma unroll
            for (int i = 0; i < W; ++i) pr.w[i] = 0xffffffffu;
            if (!(a.n_units_dev && unit >= *a.n_units_dev)) {
                float cx = a.cx, cy = a.cy, cz = a.cz;
                if (a.mode == 1u) {
                    const int4 ic = static_cast<const int4*>(a.units)[unit];
                    cx = (float)((double)ic.x * a.res + a.ox); cy = (float)((double)ic.y * a.res + a.oy); cz = (float)((double)ic.z * a.res + a.oz);
                } else if (a.mode == 3u) {
                    const int4 ip = static_cast<const int4*>(a.units)[unit];
                    const double h = (double)a.int_step / 2;
                    cx = (float)(((double)ip.x + h) * a.res + a.ox); cy = (float)(((double)ip.y + h) * a.res + a.oy);
                    cz = (float)(((double)ip.z + (a.dimension == 3 ? h : 0.0)) * a.res + a.oz);
                } else if (a.mode == 4u) {
                    const double4 pc = static_cast<const double4*>(a.units)[unit];
                    const double h = a.s / 2;
                    cx = (float)(pc.x + h); cy = (float)(pc.y + h); cz = (float)(pc.z + h);
                }
                const uint32_t qz = chunk % a.boxes_z, qt = chunk / a.boxes_z, qy = qt % a.boxes_y, qx = qt / a.boxes_y;
                const uint32_t x0 = qx * 16u, y0 = qy * 16u, z0 = qz * 16u;
                const float ex = 0.5f * (float)(min(16u, a.nx - x0) - 1u), ey = 0.5f * (float)(min(16u, a.ny - y0) - 1u),
                            ez = 0.5f * (float)(min(16u, a.nz - z0) - 1u);
                // the box's samples are corner + step * index: its centre and half extents in the same arithmetic
                const float hx = a.step * ex, hy = a.step * ey, hz = a.step * ez;
                ev.prune(sample(cx, a.step, a.xs0 + x0) + hx, sample(cy, a.step, y0) + hy, sample(cz, a.step, z0) + hz,
                         __builtin_fabsf(hx), __builtin_fabsf(hy), __builtin_fabsf(hz), pr);
            }
#pragma unroll
            for (int i = 0; i < W; ++i) a.out[(size_t)box * W + i] = pr.w[i];
        }
    }
}

// ------------------------------------------------------------------------------------------
// renderers (reference rendering/ray_caster.cl, rendering/bitmap.cl)
// ------------------------------------------------------------------------------------------
// Plain binary32 in the order written (no fma: the build has -ffp-contract=off), the same
// expression trees as oracle/sdf_oracle.c, so pixels compare equal byte for byte.
struct F3 { float x, y, z; };
__host__ __device__ __forceinline__ F3 mk3(float x, float y, float z) { F3 r = {x, y, z}; return r; }
__device__ __forceinline__ F3 add3(F3 a, F3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 sub3(F3 a, F3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 mul3(F3 a, float k) { return mk3(a.x * k, a.y * k, a.z * k); }
__device__ __forceinline__ float dot3(F3 a, F3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ F3 normalize3(F3 a) { return mul3(a, 1.0f / sdf::sqrt_(dot3(a, a))); }
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return __builtin_fminf(__builtin_fmaxf(v, lo), hi); }
__device__ __forceinline__ float mixf(float a, float b, float t) { return a + (b - a) * t; }
__device__ __forceinline__ float smoothstepf(float e0, float e1, float x)
{
    const float t = clampf((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return (t * t) * (3.0f - 2.0f * t);
}
__device__ __forceinline__ void store_rgb(uint8_t* out, F3 c)
{
    out[0] = (uint8_t)clampf(c.x, 0.0f, 255.0f);
    out[1] = (uint8_t)clampf(c.y, 0.0f, 255.0f);
    out[2] = (uint8_t)clampf(c.z, 0.0f, 255.0f);
}

struct RayCasterArgs {
    F3 origin, forward, up, right;
    float pixel_tolerance, box_radius, min_distance, max_distance, floor_z;
    uint32_t options, w, h;
    uint8_t* out;  // uchar RGB, index (y + h*x)*3 (INDEX2_GG, cl_util/indexing.h:5,9)
};

constexpr float kOverRelaxation = 0.5f;            // ray_caster.cl:5-11
constexpr uint32_t kPrimaryMaxSteps = 1000, kLightMaxSteps = 100, kAoSteps = 4;
constexpr float kLightMinInfluence = 1.0f / 128.0f;
constexpr uint32_t kFalseColor = 1u, kZebra = 2u;

// ray_caster.cl:13-26
__device__ __forceinline__ float over_relaxation_step(F3 direction, float4 e)
{
    const float over = kOverRelaxation * __builtin_fminf(1.0f, 1.0f + dot3(direction, mk3(e.x, e.y, e.z)));
    return e.w * (1.0f + over);
}
// ray_caster.cl:28-40
__device__ __forceinline__ void light_no_trace(F3 normal, F3 to_light, F3 to_camera, float& diffuse, float& specular)
{
    const F3 halfway = normalize3(add3(to_light, to_camera));
    diffuse = __builtin_fmaxf(0.0f, dot3(normal, to_light));
    float s = __builtin_fmaxf(0.0f, dot3(normal, halfway));
    s *= s; s *= s; s *= s;
    specular = s;
}
// ray_caster.cl:118-131
__device__ __forceinline__ F3 map_color(float ambient, float diffuse, float specular)
{
    const float saturation = 0.75f * smoothstepf(0.0f, 0.25f, diffuse);
    const float value = 0.1f + 0.8f * mixf(diffuse, ambient, 0.3f);
    const float chroma = value * saturation;
    const float X = chroma * 0.7f;
    const float m = value - chroma;
    const float sp = specular * 128.0f;
    return mk3(255.0f * (X + m) + sp, 255.0f * (chroma + m) + sp, 255.0f * (0.0f + m) + sp);
}
// ray_caster.cl:133-144
__device__ __forceinline__ F3 map_color_zebra(F3 point, float ambient, float diffuse, float specular)
{
    const int white = sdf::to_int_(__builtin_floorf(point.y)) & 1;
    float color = 50.0f + 150.0f * (float)white;
    color *= ambient + diffuse;
    color += 128.0f * specular;
    return mk3(color, color, color);
}

// One pixel per lane, 8x8 pixels per wavefront.  The reference kernel calls evaluate() from five
// places (primary march, false-colour residual, ambient occlusion, shadow march, floor shadow);
// here a pixel is a small state machine and the wavefront loops over ONE evaluate() site until
// every lane is done: lanes in different phases share each pass through the tape, the
// interpreter is instantiated once, and a pixel still sees exactly the reference's sequence of
// evaluations (pixels are independent).
template <class E> __global__ void __launch_bounds__(256) k_ray_caster(const E ev, const RayCasterArgs a)
{
    extern __shared__ float4 lds[];
    enum Phase : uint32_t { PRIMARY, RESIDUAL, AO, LIGHT, FLOOR, DONE };
    const uint32_t tiles_y = (a.h + 7u) >> 3;
    const uint32_t tile = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    const uint32_t px = (tile / tiles_y) * 8u + (lane >> 3), py = (tile % tiles_y) * 8u + (lane & 7u);
    const bool in_image = px < a.w && py < a.h;

    const float filmx = (float)px - (float)(a.w - 1u) / 2.0f;
    const float filmy = (float)py - (float)(a.h - 1u) / 2.0f;
    const F3 direction = normalize3(sub3(add3(a.forward, mul3(a.right, filmx)), mul3(a.up, filmy)));
    const F3 to_camera = mul3(direction, -1.0f);
    const F3 to_light = mul3(normalize3(mk3(1.0f, 2.0f, -1.0f)), -1.0f);
    const F3 to_light2 = mul3(normalize3(mk3(-1.0f, 1.0f, 0.0f)), -1.0f);
    const bool false_color = (a.options & kFalseColor) != 0u;

    uint32_t phase = in_image ? PRIMARY : DONE;
    // primary march
    float distance = a.min_distance, fallback = a.min_distance;
    float4 pe = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // last primary evaluation
    uint32_t step = 0;
    bool hit = false;
    // after the march
    F3 point = mk3(0, 0, 0), normal = mk3(0, 0, 0), color = mk3(0, 0, 0);
    float local_eps = 0.0f, residual = 0.0f, ambient = 0.0f;
    // ambient occlusion
    float occlusion = 0.0f, ao_scale = 1.0f, ao_distance = 0.0f;
    uint32_t ao_i = 0;
    // shadow march
    float d0 = 0.0f, s0 = 0.0f, threshold = 0.0f, visibility = 1.0f, ldistance = 0.0f, lfallback = 0.0f;
    uint32_t lstep = 0;
    float floor_distance = 0.0f;

    while (sdf::any_lane(sdf::mk(phase != DONE))) {
        F3 p = mk3(0.0f, 0.0f, 0.0f);
        switch (phase) {
        case PRIMARY: p = add3(a.origin, mul3(direction, distance)); break;
        case RESIDUAL: p = point; break;
        case AO: p = add3(point, mul3(normal, ao_distance)); break;
        case LIGHT: p = add3(point, mul3(to_light, ldistance)); break;
        case FLOOR: p = add3(a.origin, mul3(direction, floor_distance)); break;
        default: break;
        }
        const float4 e = sdf::voxel(ev(p.x, p.y, p.z, lds), 0);

        // 0 = stay in the phase; otherwise the transition this evaluation triggers
        enum Next : uint32_t { STAY, END_PRIMARY, BEGIN_LIGHT, END_LIGHT, BEGIN_FLOOR, WRITE };
        uint32_t next = STAY;
        float diffuse = 0.0f, specular = 0.0f;  // result of the shadow march (END_LIGHT)

        if (phase == PRIMARY) {  // ray_caster.cl:168-196
            pe = e;
            if (distance - fallback > e.w) {
                distance = fallback;
                if (++step == kPrimaryMaxSteps) next = END_PRIMARY;
            } else {
                hit = e.w < a.pixel_tolerance * distance;
                if (hit) {
                    distance += e.w * clampf(1.0f / dot3(mk3(e.x, e.y, e.z), to_camera), 0.0f, 2.0f);
                    next = END_PRIMARY;
                } else {
                    fallback = distance + e.w;
                    distance = distance + over_relaxation_step(direction, e);
                    if (distance > a.max_distance) {
                        distance = __builtin_inff();
                        next = END_PRIMARY;
                    } else if (++step == kPrimaryMaxSteps) {
                        next = END_PRIMARY;
                    }
                }
            }
        } else if (phase == RESIDUAL) {
            residual = sdf::abs_(e.w);
            next = BEGIN_LIGHT;
        } else if (phase == AO) {  // ray_caster.cl:100-116
            occlusion += ao_scale * (ao_distance - e.w);
            ao_scale /= 2.0f;
            ao_distance += a.box_radius / 100.0f;
            if (++ao_i == kAoSteps) {
                ambient = clampf(1.0f - (occlusion * 0.5f) / (1.0f - ao_scale), 0.0f, 1.0f);
                next = BEGIN_LIGHT;
            }
        } else if (phase == LIGHT) {  // ray_caster.cl:56-88
            visibility = __builtin_fminf(visibility, e.w / ldistance);
            bool finished = visibility < threshold;
            if (!finished) {
                if (ldistance - lfallback > e.w) {
                    ldistance = lfallback;
                    finished = ++lstep == kLightMaxSteps;
                } else {
                    lfallback = ldistance + e.w;
                    ldistance = ldistance + over_relaxation_step(to_light, e);
                    finished = ldistance > a.max_distance || ++lstep == kLightMaxSteps;
                }
            }
            if (finished) {
                diffuse = false_color ? (float)lstep : visibility * d0;
                specular = false_color ? 0.0f : visibility * s0;
                next = END_LIGHT;
            }
        } else if (phase == FLOOR) {  // ray_caster.cl:238-250
            float shadow = clampf((2.0f * e.w) / a.box_radius, 0.0f, 1.0f);
            shadow = 1.0f - shadow;
            shadow *= shadow;
            shadow = 1.0f - shadow;
            const float k = 0.4f + 0.6f * shadow;
            color = mk3(mixf(0.0f, color.x, k), mixf(0.0f, color.y, k), mixf(0.0f, color.z, k));
            next = WRITE;
        }

        if (next == END_PRIMARY) {  // ray_caster.cl:198-236
            local_eps = __builtin_fmaxf(1e-4f, 2.0f * sdf::abs_(pe.w));
            point = add3(a.origin, mul3(direction, distance));
            normal = mk3(pe.x, pe.y, pe.z);
            if (false_color) {
                if (hit) phase = RESIDUAL;
                else next = BEGIN_LIGHT;
            } else if (hit) {
                ao_distance = a.box_radius / 100.0f;
                phase = AO;
            } else {
                color = mk3(230.0f, 230.0f, 241.0f);
                next = BEGIN_FLOOR;
            }
        }
        if (next == BEGIN_LIGHT) {  // ray_caster.cl:42-55
            light_no_trace(normal, to_light, to_camera, d0, s0);
            if (d0 <= 0.0f && s0 <= 0.0f) {
                next = END_LIGHT;  // diffuse = specular = 0
            } else {
                threshold = kLightMinInfluence / __builtin_fmaxf(d0, s0);
                ldistance = lfallback = local_eps;
                phase = LIGHT;
            }
        }
        if (next == END_LIGHT) {
            if (false_color) {
                float steps = (float)step;
                steps += diffuse;
                steps += (float)kAoSteps;
                color = mk3(steps, 1000.0f * residual, 0.0f);
            } else {
                float d2, s2;
                light_no_trace(normal, to_light2, to_camera, d2, s2);
                const float d = 0.8f * diffuse + 0.2f * d2;
                const float s = 0.8f * specular + 0.2f * s2;
                color = (a.options & kZebra) ? map_color_zebra(point, ambient, d, s) : map_color(ambient, d, s);
            }
            next = BEGIN_FLOOR;
        }
        if (next == BEGIN_FLOOR) {
            floor_distance = (a.floor_z - a.origin.z) / direction.z;
            if (floor_distance > 0.0f && floor_distance < distance) phase = FLOOR;
            else next = WRITE;
        }
        if (next == WRITE) {
            store_rgb(a.out + ((size_t)py + (size_t)a.h * px) * 3, color);
            phase = DONE;
        }
    }
}

// bitmap.cl:1-18: one pixel per lane, y fastest like the output
template <class E>
__global__ void __launch_bounds__(256)
k_bitmap(const E ev, float ox, float oy, float oz, float step_size, uint32_t w, uint32_t h, uint8_t* __restrict__ out)
{
    extern __shared__ float4 lds[];
    const uint32_t lin = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = lin < w * h;
    const uint32_t x = active ? lin / h : 0u, y = active ? lin % h : 0u;
    const float v = ev.dist(ox + step_size * (float)x, oy + step_size * (float)(h - y - 1u), oz + step_size * 0.0f, lds);
    if (!active) return;
    const float t = (v < 0.0f) ? 0.0f : 1.0f;  // step(0, v)
    store_rgb(out + (size_t)lin * 3, mk3(mixf(125.0f, 230.0f, t), mixf(179.0f, 230.0f, t), mixf(0.0f, 241.0f, t)));
}

// ------------------------------------------------------------------------------------------
// 2D contouring (reference rendering/polygon2d.cl)
// ------------------------------------------------------------------------------------------
// encode_index, polygon2d.cl:5-36: a cell index, or where a link leaves the block
__device__ __forceinline__ uint32_t pp_encode_index(int32_t cx, int32_t cy, uint32_t size_x, uint32_t size_y, uint32_t index)
{
    constexpr uint32_t kIndexSize = 20;
    index &= (1u << kIndexSize) - 1u;
    bool y;
    int32_t sx, sy;
    if (cx < 0 || (uint32_t)cx >= size_x) { y = false; sx = cx; sy = cy; }
    else if (cy < 0 || (uint32_t)cy >= size_y) { y = true; sx = cy; sy = cx; }
    else return index;
    return 0x80000000u | (y ? 0x40000000u : 0u) | (sx < 0 ? 0x20000000u : 0u) | ((uint32_t)sy << kIndexSize) | index;
}

// place_vertex, polygon2d.cl:38-80: weighted average of the corners, then <= 8 steps towards the
// point where the three corners' tangent lines meet.  Plain binary32 in the oracle's order.
__device__ __forceinline__ float2 pp_place_vertex(const float (&px)[3], const float (&py)[3], const float4 (&val)[3])
{
    float ax = 0.0f, ay = 0.0f, weight = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float w = 1.0f / (1.0f + sdf::abs_(val[i].w));
        ax += px[i] * w;
        ay += py[i] * w;
        weight += w;
    }
    float x = ax / weight, y = ay / weight;
    for (int i = 0; i < 8; ++i) {
        float gx = 0.0f, gy = 0.0f, residual = 0.0f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float tmp = (val[j].x * (x - px[j]) + val[j].y * (y - py[j])) + val[j].w;
            residual += tmp * tmp;
            gx += val[j].x * tmp;
            gy += val[j].y * tmp;
        }
        if (residual < 1e-3f) break;
        const float g2 = gx * gx + gy * gy;
        if (g2 < 1e-8f) break;
        const float k = residual / g2;
        x -= gx * k;
        y -= gy * k;
    }
    return make_float2(x, y);
}

struct PolygonArgs {
    const float4* corners;   // float4[gx*gy] per block, index y + gy*x (grid_eval over (gx, gy, 1))
    uint32_t gx, gy;         // corner samples per block; cells = (gx-1)*(gy-1)*2
    float cx, cy, step;      // !BATCH: box corner as given
    const int4* blocks;      // BATCH: integer block corners
    double res, ox, oy;      // BATCH: corner = (float)(int_corner * res + origin)
    float2* vertices;        // [cells] per block
    uint32_t* links;         // [cells] per block
    uint32_t* starts;        // [(gx-1)+(gy-1)] per block
    uint32_t* start_counter; // one per block (caller zeroes)
};

// One lane per triangular half cell; lin = t + 2*(y + (gy-1)*x) = INDEX3_GG of the reference's
// (gx-1, gy-1, 2) launch, so a wavefront reads/writes contiguous runs.
template <bool BATCH> __global__ void __launch_bounds__(256) k_process_polygon(const PolygonArgs a)
{
    const uint32_t sx = a.gx - 1u, sy = a.gy - 1u, cells = sx * sy * 2u;
    const uint32_t b = BATCH ? blockIdx.y : 0u;
    const uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= cells) return;
    const uint32_t t = index & 1u, y = (index >> 1) % sy, x = (index >> 1) / sy;
    const float4* corners = a.corners + (size_t)b * a.gx * a.gy;
    float2* vertices = a.vertices + (size_t)b * cells;
    uint32_t* links = a.links + (size_t)b * cells;
    float bcx = a.cx, bcy = a.cy;
    if (BATCH) {
        const int4 ic = a.blocks[b];
        bcx = (float)((double)ic.x * a.res + a.ox);
        bcy = (float)((double)ic.y * a.res + a.oy);
    }
    const uint32_t offx[3] = {0u, 1u, t}, offy[3] = {0u, 1u, 1u - t};
    float4 val[3];
    uint32_t cell_type = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        val[i] = corners[(y + offy[i]) + (size_t)a.gy * (x + offx[i])];
        cell_type = (cell_type << 1) | (val[i].w <= 0.0f ? 1u : 0u);
    }
    if (cell_type == 0u || cell_type == 7u) {
        links[index] = 0xffffffffu;
        return;
    }
    bool backwards = cell_type == 3u || cell_type == 5u || cell_type == 6u;
    if (backwards) cell_type = 7u - cell_type;
    const bool flip = t == 1u;
    if (flip) backwards = !backwards;
    // cell_type is now 1, 2 or 4 (polygon2d.cl:125-139)
    int32_t fx = cell_type == 4u ? -1 : 0, fy = cell_type == 1u ? 1 : 0;
    int32_t rx = cell_type == 1u ? -1 : 0, ry = cell_type == 2u ? 1 : 0;
    if (backwards) { const int32_t u = fx, v = fy; fx = rx; fy = ry; rx = u; ry = v; }
    if (flip) { int32_t u = fx; fx = fy; fy = u; u = rx; rx = ry; ry = u; }
    fx += (int32_t)x; fy += (int32_t)y; rx += (int32_t)x; ry += (int32_t)y;
    const uint32_t fwd_index = (1u - t) + 2u * ((uint32_t)fy + sy * (uint32_t)fx);  // INDEX3_G, wrapping like the size_t expression
    links[index] = pp_encode_index(fx, fy, sx, sy, fwd_index);
    const uint32_t start_index = pp_encode_index(rx, ry, sx, sy, index);
    if (start_index & 0x80000000u) {
        const uint32_t slot = atomicAdd(&a.start_counter[b], 1u);
        if (slot < sx + sy) a.starts[(size_t)b * (sx + sy) + slot] = start_index ^ 0x20000000u;
    }
    float px[3], py[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        px[i] = bcx + (float)(x + offx[i]) * a.step;
        py[i] = bcy + (float)(y + offy[i]) * a.step;
    }
    vertices[index] = pp_place_vertex(px, py, val);
}

// ------------------------------------------------------------------------------------------
// self-test of the fast correctly rounded sqrt / reciprocal (interp.hpp sqrt_cr, sqrt_inv_cr):
// every binary32 bit pattern, one voxel per lane (T = float) and two (T = f2), against the
// compiler's IEEE expansions.  counts[0..3] = mismatches of {sqrt_cr, sqrt_inv_cr.s, sqrt_inv_cr.r}
// and the number of inputs that took the fast path.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool same_bits(float a, float b)
{
    return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
}

}  // namespace sdfk
