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
//   k_mc_scan       exclusive scan of the workgroup counts (one workgroup; the list is tiny)
//   k_mc_vertices   recompute, scan inside the workgroup, write vertex positions (fp64, world
//                   coordinates as mesh.py:65-68 computes them) and each sample's first vertex id
//   k_mc_triangles  recompute the case, scan, write triangles as global vertex ids
// A block is an array [A0][A1][A2] (a2 fastest): for the pymcubes layout A0 = sy (y flipped),
// A1 = sx, A2 = sz.  One lane per sample; a workgroup owns 256 consecutive samples of one block.
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

struct McArgs {
    const float* fields;      // float[n_blocks][A0*A1*A2]
    uint32_t A0, A1, A2;
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

struct McSample {
    uint32_t b, s, a0, a1, a2;
    bool valid;
};

__device__ __forceinline__ McSample mc_sample(const McArgs& a)
{
    McSample m;
    m.b = blockIdx.x / a.chunks;
    const uint32_t chunk = blockIdx.x - m.b * a.chunks;
    const uint32_t n = a.A0 * a.A1 * a.A2;
    m.s = chunk * kMcBlock + threadIdx.x;
    m.valid = m.s < n;
    const uint32_t s = m.valid ? m.s : 0u;
    m.a2 = s % a.A2;
    const uint32_t t = s / a.A2;
    m.a1 = t % a.A1;
    m.a0 = t / a.A1;
    return m;
}

// active axes of the sample's three owned edges, and the values needed to place their vertices
__device__ __forceinline__ uint32_t mc_edge_flags(const McArgs& a, const McSample& m, const float* f, float& f1, float (&f2)[3])
{
    if (!m.valid) return 0u;
    const uint32_t stride[3] = {a.A1 * a.A2, a.A2, 1u};
    const uint32_t pos[3] = {m.a0, m.a1, m.a2}, dims[3] = {a.A0, a.A1, a.A2};
    f1 = f[m.s];
    const bool in1 = f1 <= 0.0f;
    uint32_t flags = 0;
#pragma unroll
    for (int axis = 0; axis < 3; ++axis) {
        f2[axis] = 0.0f;
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
    const McSample m = mc_sample(a);
    const float* f = a.fields + (size_t)m.b * a.A0 * a.A1 * a.A2;
    float f1, f2[3];
    const uint32_t nv = __popc(mc_edge_flags(a, m, f, f1, f2));
    const uint32_t nt = kMcTriangleCountDev[mc_case(a, m, f)];
    uint32_t total_v, total_t;
    wg_exclusive_scan(nv, scratch, total_v);
    wg_exclusive_scan(nt, scratch, total_t);
    if (threadIdx.x == 0) a.wg_counts[blockIdx.x] = make_uint2(total_v, total_t);
}

// Exclusive scan of counts[0..n) in place; counts[n] receives the totals.  One workgroup of 1024.
__global__ void __launch_bounds__(1024) k_mc_scan(uint2* counts, uint32_t n)
{
    __shared__ uint32_t scratch[16];
    uint32_t base_v = 0, base_t = 0;
    for (uint32_t start = 0; start < n; start += blockDim.x) {
        const uint32_t i = start + threadIdx.x;
        const uint2 c = i < n ? counts[i] : make_uint2(0u, 0u);
        uint32_t tv, tt;
        const uint32_t pv = wg_exclusive_scan(c.x, scratch, tv);
        const uint32_t pt = wg_exclusive_scan(c.y, scratch, tt);
        if (i < n) counts[i] = make_uint2(base_v + pv, base_t + pt);
        base_v += tv;
        base_t += tt;
    }
    if (threadIdx.x == 0) counts[n] = make_uint2(base_v, base_t);
}

__global__ void __launch_bounds__(256) k_mc_vertices(const McArgs a)
{
    __shared__ uint32_t scratch[8];
    const McSample m = mc_sample(a);
    const size_t block_base = (size_t)m.b * a.A0 * a.A1 * a.A2;
    const float* f = a.fields + block_base;
    float f1, f2[3];
    const uint32_t flags = mc_edge_flags(a, m, f, f1, f2);
    uint32_t total;
    const uint32_t first = a.wg_counts[blockIdx.x].x + wg_exclusive_scan(__popc(flags), scratch, total);
    if (!m.valid) return;
    a.info[block_base + m.s] = (first << 3) | flags;
    if (!flags) return;
    // mesh.py:65-68 in numpy float64: swap the first two array axes, negate y, scale, add the corner
    const int4 ic = a.blocks[m.b];
    const double cx = (double)ic.x * a.res + a.ox, cy = (double)ic.y * a.res + a.oy, cz = (double)ic.z * a.res + a.oz;
    const uint32_t pos[3] = {m.a0, m.a1, m.a2};
    uint32_t id = first;
#pragma unroll
    for (int axis = 0; axis < 3; ++axis) {
        if (!(flags & (1u << axis))) continue;
        const double t = (1.0 * (0.0 - (double)f1)) / ((double)f2[axis] - (double)f1);
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

__global__ void __launch_bounds__(256) k_mc_triangles(const McArgs a)
{
    __shared__ uint32_t scratch[8];
    const McSample m = mc_sample(a);
    const size_t block_base = (size_t)m.b * a.A0 * a.A1 * a.A2;
    const float* f = a.fields + block_base;
    const uint32_t cube = mc_case(a, m, f);
    const uint32_t nt = kMcTriangleCountDev[cube];
    uint32_t total;
    uint32_t slot = a.wg_counts[blockIdx.x].y + wg_exclusive_scan(nt, scratch, total);
    if (!nt) return;
    const uint32_t stride[3] = {a.A1 * a.A2, a.A2, 1u};
    const uint32_t* info = a.info + block_base + m.s;
    for (uint32_t k = 0; k < 3u * nt; k += 3, ++slot) {
        uint32_t* out = a.triangles + 3 * (size_t)slot;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int e = kMcTrianglesDev[cube][k + j];
            const unsigned char* c = kMcCornerDev[kMcEdgeOwnerDev[e][0]];
            const uint32_t axis = kMcEdgeOwnerDev[e][1];
            const uint32_t w = info[c[0] * stride[0] + c[1] * stride[1] + c[2] * stride[2]];
            out[j] = (w >> 3) + __popc(w & ((1u << axis) - 1u));
        }
    }
}

}  // namespace sdfk
