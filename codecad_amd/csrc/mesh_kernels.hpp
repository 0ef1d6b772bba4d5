// codecad_amd/csrc/mesh_kernels.hpp
//
// Marching cubes over ALL leaf blocks of a subdivision at once: the consumer of the scalar blocks
// that k_grid_eval_blocks<LAYOUT 1> writes (reference rendering/mesh.py:45-74, where each block is
// copied to the host and handed to PyMCubes one at a time).  Published algorithm (Lorensen & Cline
// 1987), case table derived in tools/gen_mc_table.py; output ordering and arithmetic are those of
// the oracle's restatement (oracle/sdf_oracle.c oracle_marching_cubes), so meshes compare equal.
//
// Indexed, deterministic output without atomics -- three passes over the samples, all HBM-bound:
//   k_mc_count      per workgroup: number of vertices (active edges owned by its samples) and of
//                   triangles (cells whose low corner it owns)
//   k_mc_scan_*     exclusive scan of the workgroup counts (tiles of 1024, their totals, add back)
//   k_mc_vertices   recompute, scan inside the workgroup, write vertex positions (fp64, world
//                   coordinates as mesh.py:65-68 computes them) and each sample's first vertex id
//   k_mc_triangles  recompute the case, scan, write triangles as global vertex ids
// A block is an array [A0][A1][A2] (a2 fastest): for the pymcubes layout A0 = sy (y flipped),
// A1 = sx, A2 = sz.  A lane owns kMcPerLane consecutive samples, a workgroup 256 times that, of one block.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

#include "mc_table.hpp"

namespace sdfk {

constexpr uint32_t kMcBlock = 256;

__device__ __constant__ unsigned char kMcCornerDev[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
__device__ __constant__ unsigned char kMcEdgeOwnerDev[12][2] = {{0, 0}, {1, 1}, {3, 0}, {0, 1}, {4, 0}, {5, 1}, {7, 0}, {4, 1}, {0, 2}, {1, 2}, {2, 2}, {3, 2}};
__device__ __constant__ signed char kMcTrianglesDev[256][MC_TABLE_WIDTH] = MC_TRIANGLES_INIT;
__device__ __constant__ unsigned char kMcTriangleCountDev[256] = MC_TRIANGLE_COUNT_INIT;

// x / d for x < 2^24 and d < 2^16 by one 64-bit multiply and a shift: m = floor(2^40 / d) + 1 is exact
// while x*d < 2^40 (the host checks both bounds), instead of a ~40-instruction division per lane.
struct FastDiv {
    uint32_t d;
    uint64_t m;
    __host__ __device__ __forceinline__ uint32_t div(uint32_t x) const { return d == 1u ? x : (uint32_t)((x * m) >> 40); }
};
inline FastDiv make_fast_div(uint32_t d) { return FastDiv{d, d > 1u ? (1ull << 40) / d + 1ull : 0ull}; }

struct McArgs {
    const float* fields;      // float[n_blocks][A0*A1*A2]
    uint32_t A0, A1, A2;
    FastDiv div_A1, div_A2;
    uint32_t chunks;          // workgroups per block
    const int4* blocks;       // integer block corners
    double res, ox, oy, oz;   // block corner = int_corner * res + origin  (subdivision.py:100)
    double step;              // sample spacing of the block (box_resolution)
    double y_offset;          // added to y: 0 reproduces mesh.py:65-68, (A0-1)*step gives true positions
    uint2* wg_counts;         // per workgroup (vertices, triangles); exclusive prefix after k_mc_scan
    uint32_t* info;           // per sample: first vertex id << 3 | active axes
    double* vertices;         // [.][3]
    uint32_t* triangles;      // [.][3]
};

// Samples per lane.  Measured on MI355X over the 126 M samples of the bench's leaf blocks (16^3):
// 1 per lane: count 0.84 ms, vertices 0.88 ms, triangles 1.62 ms; 4 per lane: 1.14 / 0.79 / 1.79 ms.
// The passes are bound by integer/address instructions (12 loads, 8 compares and the index math per
// sample), not by HBM (0.5 GB of samples per pass) nor by workgroup latency; the next step would be one
// inside-bit per sample shared through LDS row masks instead of 12 loads per sample.
constexpr uint32_t kMcPerLane = 1;
constexpr uint32_t kMcPerGroup = kMcBlock * kMcPerLane;

struct McSample {
    uint32_t s, a0, a1, a2;
    bool valid;
};

// block of this workgroup and the first sample of this lane
__device__ __forceinline__ void mc_lane(const McArgs& a, uint32_t& b, uint32_t& s0)
{
    b = blockIdx.x / a.chunks;  // wave-uniform
    const uint32_t chunk = blockIdx.x - b * a.chunks;
    s0 = (chunk * kMcBlock + threadIdx.x) * kMcPerLane;
}

__device__ __forceinline__ McSample mc_sample(const McArgs& a, uint32_t s)
{
    McSample m;
    m.s = s;
    m.valid = s < a.A0 * a.A1 * a.A2;
    const uint32_t v = m.valid ? s : 0u;
    const uint32_t t = a.div_A2.div(v);
    m.a2 = v - t * a.A2;
    m.a0 = a.div_A1.div(t);
    m.a1 = t - m.a0 * a.A1;
    return m;
}

// active axes of the sample's three owned edges, and the values needed to place their vertices
__device__ __forceinline__ uint32_t mc_edge_flags(const McArgs& a, const McSample& m, const float* f, float& f1, float (&f2)[3])
{
    f1 = 0.0f;
    f2[0] = f2[1] = f2[2] = 0.0f;
    if (!m.valid) return 0u;
    const uint32_t stride[3] = {a.A1 * a.A2, a.A2, 1u};
    const uint32_t pos[3] = {m.a0, m.a1, m.a2}, dims[3] = {a.A0, a.A1, a.A2};
    f1 = f[m.s];
    const bool in1 = f1 <= 0.0f;
    uint32_t flags = 0;
#pragma unroll
    for (int axis = 0; axis < 3; ++axis) {
        if (pos[axis] + 1u >= dims[axis]) continue;
        f2[axis] = f[m.s + stride[axis]];
        if ((f2[axis] <= 0.0f) != in1) flags |= 1u << axis;
    }
    return flags;
}

// case index of the cell whose low corner is the sample (0 when the sample owns no cell)
__device__ __forceinline__ uint32_t mc_case(const McArgs& a, const McSample& m, const float* f)
{
    if (!m.valid || m.a0 + 1u >= a.A0 || m.a1 + 1u >= a.A1 || m.a2 + 1u >= a.A2) return 0u;
    const uint32_t s0 = a.A1 * a.A2, s1 = a.A2;
    uint32_t cube = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const uint32_t off = ((c == 1 || c == 2 || c == 5 || c == 6) ? s0 : 0u) + ((c == 2 || c == 3 || c == 6 || c == 7) ? s1 : 0u) +
                             (c >= 4 ? 1u : 0u);
        if (f[m.s + off] <= 0.0f) cube |= 1u << c;
    }
    return cube;
}

// exclusive scan of one value per lane over the workgroup; `total` = sum.  scratch: >= 8 uint32 of LDS.
__device__ __forceinline__ uint32_t wg_exclusive_scan(uint32_t v, uint32_t* scratch, uint32_t& total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, 64);
        if (lane >= (uint32_t)off) incl += up;
    }
    __syncthreads();  // scratch may still be read from a previous call
    if (lane == 63u) scratch[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    total = 0;
    const uint32_t nw = (blockDim.x + 63u) >> 6;
    for (uint32_t w = 0; w < nw; ++w) {
        const uint32_t c = scratch[w];
        if (w < wave) base += c;
        total += c;
    }
    return base + incl - v;
}

__global__ void __launch_bounds__(256) k_mc_count(const McArgs a)
{
    __shared__ uint32_t scratch[8];
    uint32_t b, s0;
    mc_lane(a, b, s0);
    const float* f = a.fields + (size_t)b * a.A0 * a.A1 * a.A2;
    uint32_t nv = 0, cube[kMcPerLane];
#pragma unroll
    for (uint32_t i = 0; i < kMcPerLane; ++i) {
        const McSample m = mc_sample(a, s0 + i);
        float f1, f2[3];
        nv += __popc(mc_edge_flags(a, m, f, f1, f2));
        cube[i] = mc_case(a, m, f);
    }
    uint32_t nt = 0;
#pragma unroll
    for (uint32_t i = 0; i < kMcPerLane; ++i) nt += kMcTriangleCountDev[cube[i]];
    uint32_t total_v, total_t;
    wg_exclusive_scan(nv, scratch, total_v);
    wg_exclusive_scan(nt, scratch, total_t);
    if (threadIdx.x == 0) a.wg_counts[blockIdx.x] = make_uint2(total_v, total_t);
}

// Exclusive scan of counts[0..n) in place, counts[n] <- totals, in three small launches:
// k_mc_scan_tiles (each workgroup scans its tile of 1024 and records the tile total), k_mc_scan_totals
// (one workgroup scans the tile totals), k_mc_scan_add (tile offsets added back).
constexpr uint32_t kMcScanTile = 1024;

__global__ void __launch_bounds__(1024) k_mc_scan_tiles(uint2* counts, uint32_t n, uint2* tile_totals)
{
    __shared__ uint32_t scratch[16];
    const uint32_t i = blockIdx.x * kMcScanTile + threadIdx.x;
    const uint2 c = i < n ? counts[i] : make_uint2(0u, 0u);
    uint32_t tv, tt;
    const uint32_t pv = wg_exclusive_scan(c.x, scratch, tv);
    const uint32_t pt = wg_exclusive_scan(c.y, scratch, tt);
    if (i < n) counts[i] = make_uint2(pv, pt);
    if (threadIdx.x == 0) tile_totals[blockIdx.x] = make_uint2(tv, tt);
}

__global__ void __launch_bounds__(1024) k_mc_scan_totals(uint2* tile_totals, uint32_t n_tiles, uint2* grand_total)
{
    __shared__ uint32_t scratch[16];
    uint32_t base_v = 0, base_t = 0;
    for (uint32_t start = 0; start < n_tiles; start += blockDim.x) {
        const uint32_t i = start + threadIdx.x;
        const uint2 c = i < n_tiles ? tile_totals[i] : make_uint2(0u, 0u);
        uint32_t tv, tt;
        const uint32_t pv = wg_exclusive_scan(c.x, scratch, tv);
        const uint32_t pt = wg_exclusive_scan(c.y, scratch, tt);
        if (i < n_tiles) tile_totals[i] = make_uint2(base_v + pv, base_t + pt);
        base_v += tv;
        base_t += tt;
    }
    if (threadIdx.x == 0) *grand_total = make_uint2(base_v, base_t);
}

__global__ void __launch_bounds__(1024) k_mc_scan_add(uint2* counts, uint32_t n, const uint2* tile_totals)
{
    const uint32_t i = blockIdx.x * kMcScanTile + threadIdx.x;
    if (i >= n) return;
    const uint2 base = tile_totals[blockIdx.x];
    const uint2 c = counts[i];
    counts[i] = make_uint2(c.x + base.x, c.y + base.y);
}

__global__ void __launch_bounds__(256) k_mc_vertices(const McArgs a)
{
    __shared__ uint32_t scratch[8];
    uint32_t b, s0;
    mc_lane(a, b, s0);
    const size_t block_base = (size_t)b * a.A0 * a.A1 * a.A2;
    const float* f = a.fields + block_base;
    McSample m[kMcPerLane];
    uint32_t flags[kMcPerLane], count = 0;
    float f1[kMcPerLane], f2[kMcPerLane][3];
#pragma unroll
    for (uint32_t i = 0; i < kMcPerLane; ++i) {
        m[i] = mc_sample(a, s0 + i);
        flags[i] = mc_edge_flags(a, m[i], f, f1[i], f2[i]);
        count += __popc(flags[i]);
    }
    uint32_t total;
    uint32_t id = a.wg_counts[blockIdx.x].x + wg_exclusive_scan(count, scratch, total);
    // mesh.py:65-68 in numpy float64: swap the first two array axes, negate y, scale, add the corner
    const int4 ic = a.blocks[b];
    const double cx = (double)ic.x * a.res + a.ox, cy = (double)ic.y * a.res + a.oy, cz = (double)ic.z * a.res + a.oz;
#pragma unroll
    for (uint32_t i = 0; i < kMcPerLane; ++i) {
        if (!m[i].valid) continue;
        a.info[block_base + m[i].s] = (id << 3) | flags[i];
        const uint32_t pos[3] = {m[i].a0, m[i].a1, m[i].a2};
#pragma unroll
        for (int axis = 0; axis < 3; ++axis) {
            if (!(flags[i] & (1u << axis))) continue;
            const double t = (1.0 * (0.0 - (double)f1[i])) / ((double)f2[i][axis] - (double)f1[i]);
            double v[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) v[k] = (double)pos[k] + (k == axis ? t : 0.0);
            double* out = a.vertices + 3 * (size_t)id;
            out[0] = v[1] * a.step + cx;
            out[1] = ((-v[0]) * a.step + cy) + a.y_offset;
            out[2] = v[2] * a.step + cz;
            ++id;
        }
    }
}

__global__ void __launch_bounds__(256) k_mc_triangles(const McArgs a)
{
    __shared__ uint32_t scratch[8];
    uint32_t b, s0;
    mc_lane(a, b, s0);
    const size_t block_base = (size_t)b * a.A0 * a.A1 * a.A2;
    const float* f = a.fields + block_base;
    uint32_t cube[kMcPerLane], count = 0;
#pragma unroll
    for (uint32_t i = 0; i < kMcPerLane; ++i) cube[i] = mc_case(a, mc_sample(a, s0 + i), f);
#pragma unroll
    for (uint32_t i = 0; i < kMcPerLane; ++i) count += kMcTriangleCountDev[cube[i]];
    uint32_t total;
    uint32_t slot = a.wg_counts[blockIdx.x].y + wg_exclusive_scan(count, scratch, total);
    if (!count) return;
    const uint32_t stride[3] = {a.A1 * a.A2, a.A2, 1u};
#pragma unroll
    for (uint32_t i = 0; i < kMcPerLane; ++i) {
        const uint32_t nt = kMcTriangleCountDev[cube[i]];
        const uint32_t* info = a.info + block_base + s0 + i;
        for (uint32_t k = 0; k < 3u * nt; k += 3, ++slot) {
            uint32_t* out = a.triangles + 3 * (size_t)slot;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int e = kMcTrianglesDev[cube[i]][k + j];
                const unsigned char* c = kMcCornerDev[kMcEdgeOwnerDev[e][0]];
                const uint32_t axis = kMcEdgeOwnerDev[e][1];
                const uint32_t w = info[c[0] * stride[0] + c[1] * stride[1] + c[2] * stride[2]];
                out[j] = (w >> 3) + __popc(w & ((1u << axis) - 1u));
            }
        }
    }
}

}  // namespace sdfk
