// codecad_amd/csrc/mesh_kernels.hpp
//
// Marching cubes over ALL leaf blocks of a subdivision at once: the consumer of the scalar blocks
// that k_grid_eval_blocks<LAYOUT 1> writes (reference rendering/mesh.py:45-74, where each block is
// copied to the host and handed to PyMCubes one at a time).  Published algorithm (Lorensen & Cline
// 1987), case table derived in tools/gen_mc_table.py; output ordering and arithmetic are those of
// the oracle's restatement (oracle/sdf_oracle.c oracle_marching_cubes), so meshes compare equal.
//
// Indexed, deterministic output without atomics.  All passes are over the samples, 64 consecutive
// samples of a block per wavefront:
//   k_mc_bits       one INSIDE bit per sample (value <= 0): each wavefront's ballot is one 64-bit word
//   k_mc_count      per workgroup: number of vertices (active edges owned by its samples) and of
//                   triangles (cells whose low corner it owns)
//   k_mc_scan_*     exclusive scan of the workgroup counts (tiles of 1024, their totals, add back)
//   k_mc_vertices   recompute, scan inside the workgroup, write vertex positions (fp64, world
//                   coordinates as mesh.py:65-68 computes them) and each sample's first vertex id
//   k_mc_triangles  recompute the case, scan, write triangles as global vertex ids
// The three counting/emitting passes never touch the float samples to classify: a wavefront fetches
// the eight 64-bit WINDOWS of the bit array that hold its lanes' eight cube corners (uniform loads),
// and if no corner differs from the lane's own sample anywhere in the wavefront -- most wavefronts: the
// surface crosses few of them -- it is done after a dozen scalar instructions.  Only surface wavefronts
// extract per-lane case indices (shifts of the windows), and only active edges load float samples.
// (The first version loaded 12 floats per sample in every pass and ran at 8 % of the HBM roofline.)
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
    const uint64_t* bits;     // inside bits: words_per_block 64-bit words per block (zero padded at the end)
    uint32_t words_per_block;
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
    m.b = blockIdx.x / a.chunks;  // wave-uniform
    const uint32_t chunk = blockIdx.x - m.b * a.chunks;
    m.s = chunk * kMcBlock + threadIdx.x;
    m.valid = m.s < a.A0 * a.A1 * a.A2;
    const uint32_t v = m.valid ? m.s : 0u;
    const uint32_t t = a.div_A2.div(v);
    m.a2 = v - t * a.A2;
    m.a0 = a.div_A1.div(t);
    m.a1 = t - m.a0 * a.A1;
    return m;
}

// Inside bits of the 64 samples at linear offset `off` from this wavefront's samples (bit i = sample
// first + i + off): two uniform word loads and a funnel shift.
__device__ __forceinline__ uint64_t mc_window(const uint64_t* words, uint32_t first, uint32_t off)
{
    const uint32_t bit = first + off, w = bit >> 6, sh = bit & 63u;
    const uint64_t lo = words[w], hi = words[w + 1];
    return sh ? (lo >> sh) | (hi << (64u - sh)) : lo;
}

// The wavefront's eight corner windows (cube corner numbering of the table) and whether any corner
// anywhere in the wavefront differs from the lane's own sample -- conservative: it ignores which
// neighbours exist, so a wavefront may be kept for nothing, never dropped wrongly.
struct McWindows {
    uint64_t w[8];
    bool any;
};
__device__ __forceinline__ McWindows mc_windows(const McArgs& a, const McSample& m)
{
    const uint64_t* words = a.bits + (size_t)m.b * a.words_per_block;
    const uint32_t first = __builtin_amdgcn_readfirstlane(m.s) & ~63u;  // the wavefront's first sample
    const uint32_t s0 = a.A1 * a.A2, s1 = a.A2;
    McWindows r;
    uint64_t diff = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const uint32_t off = ((c == 1 || c == 2 || c == 5 || c == 6) ? s0 : 0u) + ((c == 2 || c == 3 || c == 6 || c == 7) ? s1 : 0u) +
                             (c >= 4 ? 1u : 0u);
        r.w[c] = mc_window(words, first, off);
        diff |= r.w[c] ^ r.w[0];
    }
    r.any = diff != 0ull;
    return r;
}

// The lane's case index (0 when it owns no cell) and the active axes of its three owned edges.
__device__ __forceinline__ void mc_classify(const McArgs& a, const McSample& m, const McWindows& win, uint32_t& cube, uint32_t& flags)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t bits = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) bits |= (uint32_t)((win.w[c] >> lane) & 1ull) << c;
    const bool e0 = m.a0 + 1u < a.A0, e1 = m.a1 + 1u < a.A1, e2 = m.a2 + 1u < a.A2;
    const uint32_t own = bits & 1u;
    flags = 0;
    if (m.valid) {
        if (e0 && ((bits >> 1) & 1u) != own) flags |= 1u;  // corner 1 = +a0
        if (e1 && ((bits >> 3) & 1u) != own) flags |= 2u;  // corner 3 = +a1
        if (e2 && ((bits >> 4) & 1u) != own) flags |= 4u;  // corner 4 = +a2
    }
    cube = (m.valid && e0 && e1 && e2) ? bits : 0u;
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

__global__ void __launch_bounds__(256) k_mc_bits(const McArgs a, uint64_t* __restrict__ bits)
{
    const McSample m = mc_sample(a);
    const float* f = a.fields + (size_t)m.b * a.A0 * a.A1 * a.A2;
    const bool inside = m.valid && f[m.s] <= 0.0f;
    const uint64_t word = __builtin_amdgcn_ballot_w64(inside);
    if ((threadIdx.x & 63u) == 0u) bits[(size_t)m.b * a.words_per_block + (m.s >> 6)] = word;
}

__global__ void __launch_bounds__(256) k_mc_count(const McArgs a)
{
    __shared__ uint32_t scratch[8];
    const McSample m = mc_sample(a);
    const McWindows win = mc_windows(a, m);
    uint32_t nv = 0, nt = 0;
    if (win.any) {  // wave-uniform
        uint32_t cube, flags;
        mc_classify(a, m, win, cube, flags);
        nv = __popc(flags);
        nt = kMcTriangleCountDev[cube];
    }
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
    const McSample m = mc_sample(a);
    const size_t block_base = (size_t)m.b * a.A0 * a.A1 * a.A2;
    const McWindows win = mc_windows(a, m);
    uint32_t cube = 0, flags = 0;
    if (win.any) mc_classify(a, m, win, cube, flags);
    uint32_t total;
    const uint32_t first = a.wg_counts[blockIdx.x].x + wg_exclusive_scan(__popc(flags), scratch, total);
    if (!m.valid) return;
    a.info[block_base + m.s] = (first << 3) | flags;
    if (!flags) return;
    // mesh.py:65-68 in numpy float64: swap the first two array axes, negate y, scale, add the corner
    const float* f = a.fields + block_base;
    const float f1 = f[m.s];
    const int4 ic = a.blocks[m.b];
    const double cx = (double)ic.x * a.res + a.ox, cy = (double)ic.y * a.res + a.oy, cz = (double)ic.z * a.res + a.oz;
    const uint32_t pos[3] = {m.a0, m.a1, m.a2}, stride[3] = {a.A1 * a.A2, a.A2, 1u};
    uint32_t id = first;
#pragma unroll
    for (int axis = 0; axis < 3; ++axis) {
        if (!(flags & (1u << axis))) continue;
        const float f2 = f[m.s + stride[axis]];
        const double t = (1.0 * (0.0 - (double)f1)) / ((double)f2 - (double)f1);
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
    const McWindows win = mc_windows(a, m);
    uint32_t cube = 0, flags = 0;
    if (win.any) mc_classify(a, m, win, cube, flags);
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
