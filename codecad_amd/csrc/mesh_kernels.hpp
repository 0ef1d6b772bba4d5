// codecad_amd/csrc/mesh_kernels.hpp
//
// Marching cubes over ALL leaf blocks of a subdivision at once: the consumer of the scalar blocks
// that k_grid_eval_blocks<LAYOUT 1> writes (reference rendering/mesh.py:45-74, where each block is
// copied to the host and handed to PyMCubes one at a time).  Published algorithm (Lorensen & Cline
// 1987), case table derived in tools/gen_mc_table.py; output ordering and arithmetic are those of
// the oracle's restatement (oracle/sdf_oracle.c oracle_marching_cubes), so meshes compare equal.
//
// Indexed, deterministic output without atomics.  The unit of work is a SEGMENT: up to 32 consecutive
// samples along the fastest axis of a block (31 cells; a row of a 16^3 block is one segment), one
// segment per lane, 256 consecutive segments of one block per workgroup:
//   k_mc_masks      the only pass over the float samples: one inside bit per sample (value <= 0),
//                   a 32-bit mask per segment
//   k_mc_count      from the masks of a segment and of its +a0 / +a1 / +a0+a1 neighbours: the segment's
//                   active edges (three XORs, three popcounts) and its cells' triangles (only cells whose
//                   eight corner bits differ are looked up); reduced per workgroup
//   k_mc_scan_*     exclusive scan of the workgroup counts (tiles of 1024, their totals, add back)
//   k_mc_vertices   vertex ids from one scan per workgroup; per segment {first id, x/y/z edge masks};
//                   positions (fp64, as mesh.py:65-68 computes them) from the two samples of each ACTIVE edge
//   k_mc_triangles  per active cell the case, its triangles; a vertex id is the owning segment's first
//                   id plus popcounts of its edge masks
//                   (both emit through a workgroup queue in LDS: one vertex / one cell per lane)
// The first version classified every sample from 12 float loads in each of three passes with one sample
// per lane: 3.4 ms for the bench's 126 M samples, 8 % of the HBM roofline, bound by work per thread
// (DESIGN.md).  Here the floats are read once, everything else works on 4 bytes per 16-32 samples.
// A block is an array [A0][A1][A2] (a2 fastest): for the pymcubes layout A0 = sy (y flipped),
// A1 = sx, A2 = sz.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

#include "mc_table.hpp"

namespace sdfk {

constexpr uint32_t kMcBlock = 256;
constexpr uint32_t kMcSegCells = 31;  // cells per segment; a segment spans kMcSegCells + 1 samples

__device__ __constant__ unsigned char kMcTriangleCountDev[256] = MC_TRIANGLE_COUNT_INIT;

// The triangle table with each edge replaced by where its vertex lives: bits 0-2 = owning cube corner's
// offset along a0, a1, a2, bits 3-4 = the edge's axis; 0xff ends the list (byte 15: the triangle count).  One 16-byte row per case,
// so a cell's triangles cost one load instead of a chain of three byte lookups per triangle edge.
struct McPackedTable {
    unsigned char row[256][MC_TABLE_WIDTH];
};
constexpr McPackedTable mc_make_packed_table()
{
    constexpr signed char tri[256][MC_TABLE_WIDTH] = MC_TRIANGLES_INIT;
    constexpr unsigned char corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
    constexpr unsigned char owner[12][2] = {{0, 0}, {1, 1}, {3, 0}, {0, 1}, {4, 0}, {5, 1}, {7, 0}, {4, 1}, {0, 2}, {1, 2}, {2, 2}, {3, 2}};
    McPackedTable p{};
    for (int c = 0; c < 256; ++c)
        for (int k = 0; k < MC_TABLE_WIDTH; ++k) {
            const int e = tri[c][k];
            p.row[c][k] = e < 0 ? 0xffu
                                : (unsigned char)(corner[owner[e][0]][0] | (corner[owner[e][0]][1] << 1) |
                                                  (corner[owner[e][0]][2] << 2) | (owner[e][1] << 3));
        }
    // a case has at most five triangles, so the sixteenth byte of a row is free: the number of triangles of the case
    constexpr unsigned char count[256] = MC_TRIANGLE_COUNT_INIT;
    for (int c = 0; c < 256; ++c) p.row[c][MC_TABLE_WIDTH - 1] = count[c];
    return p;
}
static_assert(MC_TABLE_WIDTH == 16, "a table row is loaded as one 16-byte vector");
__device__ __constant__ McPackedTable kMcPackedDev = mc_make_packed_table();

// x / d for x < 2^24 and d < 2^16 by one 64-bit multiply and a shift: m = floor(2^40 / d) + 1 is exact
// while x*d < 2^40 (the host checks both bounds), instead of a ~40-instruction division per lane.
struct FastDiv {
    uint32_t d;
    uint64_t m;
    __host__ __device__ __forceinline__ uint32_t div(uint32_t x) const { return d == 1u ? x : (uint32_t)((x * m) >> 40); }
};
inline FastDiv make_fast_div(uint32_t d) { return FastDiv{d, d > 1u ? (1ull << 40) / d + 1ull : 0ull}; }

inline uint32_t mc_segments_per_row(uint32_t A2) { return A2 <= kMcSegCells + 1u ? 1u : (A2 - 1u + kMcSegCells - 1u) / kMcSegCells; }

struct McArgs {
    const float* fields;      // float[n_blocks][A0*A1*A2]
    uint32_t A0, A1, A2;
    uint32_t spr;             // segments per row (of A2 samples)
    uint32_t segments;        // per block = A0*A1*spr
    FastDiv div_A1, div_spr;
    uint32_t chunks;          // workgroups per block = ceil(segments / 256)
    const int4* blocks;       // integer block corners
    double res, ox, oy, oz;   // block corner = int_corner * res + origin  (subdivision.py:100)
    double step;              // sample spacing of the block (box_resolution)
    double y_offset;          // added to y: 0 reproduces mesh.py:65-68, (A0-1)*step gives true positions
    uint32_t* masks;          // [n_blocks][segments] inside bits
    uint2* wg_counts;         // per workgroup (vertices, triangles); exclusive prefix after k_mc_scan
    uint4* seg_info;          // [n_blocks][segments] {first vertex id, x-edge mask, y-edge mask, z-edge mask}
    double* vertices;         // [.][3]
    uint32_t* triangles;      // [.][3]
};

__device__ __forceinline__ uint32_t low_bits(uint32_t n) { return n >= 32u ? 0xffffffffu : (1u << n) - 1u; }

// One lane's segment: where it is, which samples it spans, which neighbours exist.
struct McSeg {
    uint32_t b, seg, a0, a1, z0, cnt;   // block, segment index in the block, position, first sample, samples
    bool valid, e0, e1, last;           // inside the block; +a0 / +a1 neighbour exists; last segment of its row
    uint32_t xy_own, z_own;             // samples whose x/y edges, z edges this segment owns
};
__device__ __forceinline__ McSeg mc_segment(const McArgs& a)
{
    McSeg g;
    g.b = blockIdx.x / a.chunks;  // wave-uniform
    const uint32_t chunk = blockIdx.x - g.b * a.chunks;
    g.seg = chunk * kMcBlock + threadIdx.x;
    g.valid = g.seg < a.segments;
    const uint32_t v = g.valid ? g.seg : 0u;
    const uint32_t row = a.div_spr.div(v), sg = v - row * a.spr;
    g.a0 = a.div_A1.div(row);
    g.a1 = row - g.a0 * a.A1;
    g.z0 = sg * kMcSegCells;
    const uint32_t left = a.A2 - g.z0;
    g.cnt = left < kMcSegCells + 1u ? left : kMcSegCells + 1u;
    g.last = g.z0 + g.cnt == a.A2;
    g.e0 = g.valid && g.a0 + 1u < a.A0;
    g.e1 = g.valid && g.a1 + 1u < a.A1;
    // the last sample of a segment that is not the last of its row is the first sample of the next one,
    // which owns its edges
    g.xy_own = g.valid ? low_bits(g.last ? g.cnt : g.cnt - 1u) : 0u;
    g.z_own = g.valid ? low_bits(g.cnt - 1u) : 0u;
    return g;
}
__device__ __forceinline__ uint32_t mc_first_sample(const McArgs& a, const McSeg& g) { return g.z0 + a.A2 * (g.a1 + a.A1 * g.a0); }

// masks of the segment and its three neighbours, and what follows from them
struct McMasks {
    uint32_t m00, m10, m01, m11;  // (a0, a1), (a0+1, a1), (a0, a1+1), (a0+1, a1+1)
    uint32_t ex, ey, ez;          // active owned edges along a0, a1, a2, one bit per sample of the segment
    uint32_t cells;               // cells (bit i: between samples i and i+1) whose eight corners differ
};
__device__ __forceinline__ McMasks mc_masks(const McArgs& a, const McSeg& g)
{
    const uint32_t* m = a.masks + (size_t)g.b * a.segments + g.seg;
    const uint32_t d0 = a.spr * a.A1, d1 = a.spr;
    McMasks r;
    r.m00 = g.valid ? m[0] : 0u;
    r.m10 = g.e0 ? m[d0] : r.m00;
    r.m01 = g.e1 ? m[d1] : r.m00;
    r.m11 = (g.e0 && g.e1) ? m[d0 + d1] : r.m00;
    r.ex = (r.m00 ^ r.m10) & g.xy_own;   // a missing neighbour was replaced by the segment itself: no edge
    r.ey = (r.m00 ^ r.m01) & g.xy_own;
    r.ez = (r.m00 ^ (r.m00 >> 1)) & g.z_own;
    const uint32_t any = r.m00 | r.m10 | r.m01 | r.m11, all = r.m00 & r.m10 & r.m01 & r.m11;
    r.cells = (g.e0 && g.e1) ? ((any | (any >> 1)) ^ (all & (all >> 1))) & g.z_own : 0u;
    return r;
}
// case index of cell i of the segment (corner numbering of the table)
__device__ __forceinline__ uint32_t mc_cube(const McMasks& k, uint32_t i)
{
    const uint32_t p00 = (k.m00 >> i) & 3u, p10 = (k.m10 >> i) & 3u, p11 = (k.m11 >> i) & 3u, p01 = (k.m01 >> i) & 3u;
    return (p00 & 1u) | ((p10 & 1u) << 1) | ((p11 & 1u) << 2) | ((p01 & 1u) << 3) | ((p00 >> 1) << 4) | ((p10 >> 1) << 5) |
           ((p11 >> 1) << 6) | ((p01 >> 1) << 7);
}

// exclusive scan of one value per lane over the workgroup of NW wavefronts; `total` = sum.  scratch: >= NW uint32 of
// LDS.  (NW is a template argument: with blockDim.x read at run time the loop over the wavefronts' sums became an
// eight-fold unrolled loop with remainder loops, ~100 vector instructions per scan.)
template <uint32_t NW>
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
#pragma unroll
    for (uint32_t w = 0; w < NW; ++w) {
        const uint32_t c = scratch[w];
        if (w < wave) base += c;
        total += c;
    }
    return base + incl - v;
}

// sum of one value per lane over the workgroup (every lane gets it).  scratch: >= NW uint32 of LDS.
template <uint32_t NW>
__device__ __forceinline__ uint32_t wg_sum(uint32_t v, uint32_t* scratch)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63u) == 0u) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t total = 0;
#pragma unroll
    for (uint32_t w = 0; w < NW; ++w) total += scratch[w];
    return total;
}

__device__ __forceinline__ uint32_t mc_row_mask(const McArgs& a, const McSeg& g);   // below

__global__ void __launch_bounds__(256) k_mc_masks(const McArgs a)
{
    const McSeg g = mc_segment(a);
    if (!g.valid) return;
    a.masks[(size_t)g.b * a.segments + g.seg] = mc_row_mask(a, g);
}

__global__ void __launch_bounds__(256) k_mc_count(const McArgs a)
{
    __shared__ uint32_t scratch[8];
    const McSeg g = mc_segment(a);
    const McMasks k = mc_masks(a, g);
    const uint32_t nv = __popc(k.ex) + __popc(k.ey) + __popc(k.ez);
    uint32_t nt = 0;
    for (uint32_t cells = k.cells; cells; cells &= cells - 1u) nt += kMcTriangleCountDev[mc_cube(k, __ffs(cells) - 1)];
    const uint32_t total_v = wg_sum<4>(nv, scratch), total_t = wg_sum<4>(nt, scratch);
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
    const uint32_t pv = wg_exclusive_scan<16>(c.x, scratch, tv);
    const uint32_t pt = wg_exclusive_scan<16>(c.y, scratch, tt);
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
        const uint32_t pv = wg_exclusive_scan<16>(c.x, scratch, tv);
        const uint32_t pt = wg_exclusive_scan<16>(c.y, scratch, tt);
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

// The emitting passes balance their work inside the workgroup: a lane first ENUMERATES what its
// segment produces (active edges / cells with triangles) into an LDS queue -- positions from one scan,
// so the global output order is unchanged -- and then every lane takes queue entries round robin: one
// vertex, or one cell's triangles, per lane.  Emitting straight from the per-segment loops left most
// lanes idle behind the busiest row of the wavefront (0.39 + 0.66 ms for the bench's leaf blocks).
constexpr uint32_t kMcQueue = 2048;   // entries per window; a workgroup with more loops over windows

__global__ void __launch_bounds__(256) k_mc_vertices(const McArgs a)
{
    __shared__ uint32_t scratch[8];
    __shared__ uint32_t queue[kMcQueue];
    const McSeg g = mc_segment(a);
    const McMasks k = mc_masks(a, g);
    uint32_t total;
    const uint32_t mine = wg_exclusive_scan<4>(__popc(k.ex) + __popc(k.ey) + __popc(k.ez), scratch, total);
    const uint32_t wg_first = a.wg_counts[blockIdx.x].x;
    if (g.valid) a.seg_info[(size_t)g.b * a.segments + g.seg] = make_uint4(wg_first + mine, k.ex, k.ey, k.ez);
    if (total == 0u) return;  // workgroup-uniform
    // mesh.py:65-68 in numpy float64: swap the first two array axes, negate y, scale, add the corner
    const uint32_t b = blockIdx.x / a.chunks, seg0 = (blockIdx.x - b * a.chunks) * kMcBlock;
    const float* block_f = a.fields + (size_t)b * a.A0 * a.A1 * a.A2;
    const int4 ic = a.blocks[b];
    const double cx = (double)ic.x * a.res + a.ox, cy = (double)ic.y * a.res + a.oy, cz = (double)ic.z * a.res + a.oz;
    const uint32_t stride[3] = {a.A1 * a.A2, a.A2, 1u}, edges[3] = {k.ex, k.ey, k.ez};
    for (uint32_t w0 = 0; w0 < total; w0 += kMcQueue) {
        // enumerate: sample by sample, axis by axis -- the order of the vertex ids
        uint32_t idx = mine;
        for (uint32_t active = k.ex | k.ey | k.ez; active; active &= active - 1u) {
            const uint32_t i = __ffs(active) - 1;
#pragma unroll
            for (uint32_t axis = 0; axis < 3; ++axis)
                if ((edges[axis] >> i) & 1u) {
                    if (idx - w0 < kMcQueue) queue[idx - w0] = threadIdx.x | (i << 8) | (axis << 13);  // unsigned: also idx >= w0
                    ++idx;
                }
        }
        __syncthreads();
        const uint32_t n = total - w0 < kMcQueue ? total - w0 : kMcQueue;
        for (uint32_t e = threadIdx.x; e < n; e += kMcBlock) {
            const uint32_t entry = queue[e], ls = entry & 255u, i = (entry >> 8) & 31u, axis = entry >> 13;
            // the owning segment's position (the division is cheap next to the fp64 work below)
            const uint32_t seg = seg0 + ls, row = a.div_spr.div(seg), sg = seg - row * a.spr;
            const uint32_t a0 = a.div_A1.div(row), a1 = row - a0 * a.A1, z = sg * kMcSegCells + i;
            const uint32_t s = z + a.A2 * (a1 + a.A1 * a0);
            const float f1 = block_f[s], f2 = block_f[s + stride[axis]];
            const double t = (1.0 * (0.0 - (double)f1)) / ((double)f2 - (double)f1);
            const uint32_t pos[3] = {a0, a1, z};
            double v[3];
#pragma unroll
            for (uint32_t c = 0; c < 3; ++c) v[c] = (double)pos[c] + (c == axis ? t : 0.0);
            double* out = a.vertices + 3 * (size_t)(wg_first + w0 + e);
            out[0] = v[1] * a.step + cx;
            out[1] = ((-v[0]) * a.step + cy) + a.y_offset;
            out[2] = v[2] * a.step + cz;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) k_mc_triangles(const McArgs a)
{
    __shared__ uint32_t scratch[8];
    __shared__ uint2 queue[kMcQueue];
    const McSeg g = mc_segment(a);
    const McMasks k = mc_masks(a, g);
    uint32_t nt = 0, ncell = 0;
    for (uint32_t cells = k.cells; cells; cells &= cells - 1u) {
        const uint32_t c = kMcTriangleCountDev[mc_cube(k, __ffs(cells) - 1)];
        nt += c;
        ncell += c ? 1u : 0u;
    }
    uint32_t total_t, total_c;
    const uint32_t my_slot = a.wg_counts[blockIdx.x].y + wg_exclusive_scan<4>(nt, scratch, total_t);
    const uint32_t my_cell = wg_exclusive_scan<4>(ncell, scratch, total_c);
    if (total_c == 0u) return;  // workgroup-uniform
    const uint32_t b = blockIdx.x / a.chunks, seg0 = (blockIdx.x - b * a.chunks) * kMcBlock;
    const uint4* block_info = a.seg_info + (size_t)b * a.segments;
    const uint32_t d0 = a.spr * a.A1, d1 = a.spr;
    for (uint32_t w0 = 0; w0 < total_c; w0 += kMcQueue) {
        uint32_t idx = my_cell, slot = my_slot;
        for (uint32_t cells = k.cells; cells; cells &= cells - 1u) {
            const uint32_t i = __ffs(cells) - 1, cube = mc_cube(k, i), c = kMcTriangleCountDev[cube];
            if (!c) continue;
            if (idx - w0 < kMcQueue)
                queue[idx - w0] = make_uint2(threadIdx.x | (i << 8) | (cube << 13) | (g.cnt << 21) | ((g.last ? 1u : 0u) << 27), slot);
            ++idx;
            slot += c;
        }
        __syncthreads();
        const uint32_t n = total_c - w0 < kMcQueue ? total_c - w0 : kMcQueue;
        for (uint32_t e = threadIdx.x; e < n; e += kMcBlock) {
            const uint2 entry = queue[e];
            const uint32_t ls = entry.x & 255u, i = (entry.x >> 8) & 31u, cube = (entry.x >> 13) & 255u;
            const uint32_t cnt = (entry.x >> 21) & 63u;
            const bool last = (entry.x >> 27) & 1u;
            const uint4* info = block_info + seg0 + ls;
            const uint4 row4 = *reinterpret_cast<const uint4*>(kMcPackedDev.row[cube]);
            const uint32_t row[4] = {row4.x, row4.y, row4.z, row4.w};
            uint32_t slot = entry.y;
#pragma unroll
            for (int t = 0; t < 15; t += 3) {  // at most five triangles per case
                if (((row[t >> 2] >> (8 * (t & 3))) & 0xffu) == 0xffu) break;
                uint32_t* out = a.triangles + 3 * (size_t)slot;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const uint32_t pk = (row[(t + j) >> 2] >> (8 * ((t + j) & 3))) & 0xffu;
                    const uint32_t axis = pk >> 3;
                    uint32_t q = ((pk & 1u) ? d0 : 0u) + ((pk & 2u) ? d1 : 0u), li = i + ((pk >> 2) & 1u);
                    if (li == cnt - 1u && !last) {  // that sample's edges belong to the next segment of the row
                        q += 1u;
                        li = 0u;
                    }
                    const uint4 w = info[q];
                    const uint32_t below = low_bits(li);
                    out[j] = w.x + __popc(w.y & below) + __popc(w.z & below) + __popc(w.w & below) +
                             (axis >= 1u ? (w.y >> li) & 1u : 0u) + (axis == 2u ? (w.z >> li) & 1u : 0u);
                }
                ++slot;
            }
        }
        __syncthreads();
    }
}

// ---- a whole block per workgroup -----------------------------------------------------------------------------
// Blocks of at most 256 rows of at most 32 samples (the 16^3 leaf blocks of a subdivision are that): one workgroup
// owns the block, so everything a lane needs from its neighbours -- their masks, their first vertex ids and edge
// masks -- lives in LDS, and the two emitting passes become one:
//   k_mc_block_count   = k_mc_masks + k_mc_count: the floats are read once, the neighbours' masks come from LDS
//   k_mc_block_emit    = k_mc_vertices + k_mc_triangles: per row {first id, edge masks} stay in LDS (the general
//                        kernels write 16 bytes per segment to memory and read them back three times per triangle),
//                        the case tables are copied to LDS once per workgroup, a lane emits exactly one vertex or one
//                        TRIANGLE at a time (cells have one to five), a triangle as one 12-byte store at a
//                        lane-consecutive address; empty blocks leave before loading anything.
// Same enumeration order, same arithmetic: the output is that of the general kernels.
__device__ __forceinline__ McMasks mc_masks_block(const uint32_t* lmask, const McArgs& a, const McSeg& g)
{
    McMasks r;
    r.m00 = lmask[threadIdx.x];           // 0 for a lane past the last row
    r.m10 = g.e0 ? lmask[threadIdx.x + a.A1] : r.m00;
    r.m01 = g.e1 ? lmask[threadIdx.x + 1u] : r.m00;
    r.m11 = (g.e0 && g.e1) ? lmask[threadIdx.x + a.A1 + 1u] : r.m00;
    r.ex = (r.m00 ^ r.m10) & g.xy_own;
    r.ey = (r.m00 ^ r.m01) & g.xy_own;
    r.ez = (r.m00 ^ (r.m00 >> 1)) & g.z_own;
    const uint32_t any = r.m00 | r.m10 | r.m01 | r.m11, all = r.m00 & r.m10 & r.m01 & r.m11;
    r.cells = (g.e0 && g.e1) ? ((any | (any >> 1)) ^ (all & (all >> 1))) & g.z_own : 0u;
    return r;
}

// bits 0-15 of m moved to bits 0, 4, 8, ... 60
__device__ __forceinline__ uint64_t mc_spread4(uint32_t m)
{
    uint32_t lo = m & 0xffu, hi = (m >> 8) & 0xffu;
    lo = (lo | (lo << 12)) & 0x000f000fu;
    hi = (hi | (hi << 12)) & 0x000f000fu;
    lo = (lo | (lo << 6)) & 0x03030303u;
    hi = (hi | (hi << 6)) & 0x03030303u;
    lo = (lo | (lo << 3)) & 0x11111111u;
    hi = (hi | (hi << 3)) & 0x11111111u;
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

__device__ __forceinline__ uint32_t mc_row_mask(const McArgs& a, const McSeg& g)
{
    const float* f = a.fields + (size_t)g.b * a.A0 * a.A1 * a.A2 + mc_first_sample(a, g);
    uint32_t mask = 0;
    if (g.cnt == 16u && (a.A2 & 3u) == 0u) {  // a row of a 16^3 block: four 16-byte loads in flight
        const float4* f4 = reinterpret_cast<const float4*>(f);
        const float4 v0 = f4[0], v1 = f4[1], v2 = f4[2], v3 = f4[3];
        const float v[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
#pragma unroll
        for (int i = 0; i < 16; ++i) mask |= (v[i] <= 0.0f ? 1u : 0u) << i;
    } else {
        for (uint32_t i = 0; i < g.cnt; ++i) mask |= (f[i] <= 0.0f ? 1u : 0u) << i;
    }
    return mask;
}

__global__ void __launch_bounds__(256) k_mc_block_count(const McArgs a)
{
    __shared__ uint32_t scratch[8];
    __shared__ uint32_t lmask[kMcBlock];
    __shared__ unsigned char lcount[256];
    const McSeg g = mc_segment(a);
    lcount[threadIdx.x] = kMcTriangleCountDev[threadIdx.x];
    const uint32_t mask = g.valid ? mc_row_mask(a, g) : 0u;
    if (g.valid) a.masks[(size_t)g.b * a.segments + g.seg] = mask;
    lmask[threadIdx.x] = mask;
    __syncthreads();
    const McMasks k = mc_masks_block(lmask, a, g);
    const uint32_t nv = __popc(k.ex) + __popc(k.ey) + __popc(k.ez);
    uint32_t nt = 0;
    if (a.A2 <= 16u) {   // (uniform)
        // The four masks interleaved, four bits per sample: the case of cell i is then ONE shift away -- bits
        // [4i, 4i + 8) -- instead of eight extractions (the loop runs as long as the busiest row of the wavefront has
        // cells: two thirds of this kernel's instructions).  Corner numbering of the table: 0 = (a0, a1), 1 = (a0+1, a1),
        // 2 = (a0+1, a1+1), 3 = (a0, a1+1), 4-7 the same one sample further.
        const uint64_t w = mc_spread4(k.m00) | (mc_spread4(k.m10) << 1) | (mc_spread4(k.m11) << 2) | (mc_spread4(k.m01) << 3);
        for (uint32_t cells = k.cells; cells; cells &= cells - 1u) nt += lcount[(uint32_t)(w >> (4u * (uint32_t)(__ffs(cells) - 1))) & 0xffu];
    } else {
        for (uint32_t cells = k.cells; cells; cells &= cells - 1u) nt += lcount[mc_cube(k, __ffs(cells) - 1)];
    }
    // one sum for both: a block of this form has at most 3 * 8192 vertices and 5 * kMcBlockCells triangles (16 bits each)
    const uint32_t both = wg_sum<4>(nv | (nt << 16), scratch);
    if (threadIdx.x == 0) a.wg_counts[blockIdx.x] = make_uint2(both & 0xffffu, both >> 16);
}

// Every pass over a block is balanced over the workgroup -- one lane per OUTPUT, not per row: a row has two
// active cells on average and up to fifteen, and a wavefront that walks its rows' cells in per-lane loops runs at
// the pace of its busiest row.  Only the listing of the active cells is such a loop (one LDS store per iteration);
// case indices and triangle counts are computed for a chunk of consecutive listed cells per lane, a vertex finds
// its row by bisection over the rows' first vertices, a triangle is told its cell by the cell.
// The kernel waits more than it computes (dependent LDS reads, then two samples per vertex from memory), so what
// it needs most is wavefronts to switch to: 20 KB of LDS per workgroup = eight workgroups per CU (at five, with the
// cells' cases kept in LDS: 0.365 ms for the bench's 30 800 blocks; at six: 0.298).
constexpr uint32_t kMcBlockCells = 3520;       // active cells of a block that fit the LDS list (a 16^3 block has 3375 cells)
constexpr uint32_t kMcTriangleWindow = 1024;   // triangles emitted per round (a block of the bench has ~900)

__device__ __forceinline__ uint32_t mc_bisect_steps(uint32_t n) { return n > 1u ? 32u - (uint32_t)__clz((int)(n - 1u)) : 0u; }

// case index of cell i of row `row` from the block's masks in LDS
__device__ __forceinline__ uint32_t mc_cube_block(const uint32_t* lmask, uint32_t row, uint32_t i, uint32_t A1)
{
    const uint32_t p00 = (lmask[row] >> i) & 3u, p10 = (lmask[row + A1] >> i) & 3u, p11 = (lmask[row + A1 + 1u] >> i) & 3u,
                   p01 = (lmask[row + 1u] >> i) & 3u;
    return (p00 & 1u) | ((p10 & 1u) << 1) | ((p11 & 1u) << 2) | ((p01 & 1u) << 3) | ((p00 >> 1) << 4) | ((p10 >> 1) << 5) |
           ((p11 >> 1) << 6) | ((p01 >> 1) << 7);
}

__global__ void __launch_bounds__(256) k_mc_block_emit(const McArgs a)
{
    __shared__ uint32_t scratch[8];
    __shared__ uint32_t lmask[kMcBlock];
    __shared__ uint4 linfo[kMcBlock];                    // per row: first vertex (within the block), x / y / z edge masks
    __shared__ unsigned short cells[kMcBlockCells];      // active cells in output order: row | i << 8
    __shared__ uint32_t which[kMcTriangleWindow];        // per triangle of the round: row | i << 8 | case << 13 | which of its triangles << 21
    __shared__ uint4 lrow[256];                          // the case table
    const uint2 first = a.wg_counts[blockIdx.x], next = a.wg_counts[blockIdx.x + 1u];   // the scan leaves the totals behind the last
    const uint32_t total_v = next.x - first.x, total_t = next.y - first.y;
    if (total_v == 0u) return;  // workgroup-uniform: nothing crosses zero in this block
    const McSeg g = mc_segment(a);
    lmask[threadIdx.x] = g.valid ? a.masks[(size_t)g.b * a.segments + g.seg] : 0u;
    lrow[threadIdx.x] = *reinterpret_cast<const uint4*>(kMcPackedDev.row[threadIdx.x]);
    __syncthreads();
    const unsigned char* row_bytes = reinterpret_cast<const unsigned char*>(lrow);
    const McMasks k = mc_masks_block(lmask, a, g);
    uint32_t total;   // (one scan for both, 16 bits each: at most 3 * 8192 vertices and kMcBlockCells cells per block)
    const uint32_t mine = wg_exclusive_scan<4>((__popc(k.ex) + __popc(k.ey) + __popc(k.ez)) | ((uint32_t)__popc(k.cells) << 16), scratch, total);
    const uint32_t my_vertex = mine & 0xffffu, n_cells = total >> 16;
    uint32_t my_cell = mine >> 16;
    linfo[threadIdx.x] = make_uint4(my_vertex, k.ex, k.ey, k.ez);
    for (uint32_t c = k.cells; c; c &= c - 1u) cells[my_cell++] = (unsigned short)(threadIdx.x | ((uint32_t)(__ffs(c) - 1) << 8));
    __syncthreads();

    // a chunk of consecutive listed cells per lane: their triangles, and the prefix of that over the workgroup
    const uint32_t per_lane = (n_cells + kMcBlock - 1u) / kMcBlock, chunk0 = threadIdx.x * per_lane;
    uint32_t chunk_triangles = 0;
    for (uint32_t j = 0; j < per_lane; ++j)
        if (chunk0 + j < n_cells) {
            const uint32_t entry = cells[chunk0 + j];
            chunk_triangles += row_bytes[mc_cube_block(lmask, entry & 255u, entry >> 8, a.A1) * 16u + 15u];
        }
    const uint32_t chunk_first = wg_exclusive_scan<4>(chunk_triangles, scratch, total);

    // ---- vertices (mesh.py:65-68 in numpy float64: swap the first two array axes, negate y, scale, add the corner)
    const uint32_t b = blockIdx.x;
    const float* block_f = a.fields + (size_t)b * a.A0 * a.A1 * a.A2;
    const int4 ic = a.blocks[b];
    const double cx = (double)ic.x * a.res + a.ox, cy = (double)ic.y * a.res + a.oy, cz = (double)ic.z * a.res + a.oz;
    const uint32_t stride[3] = {a.A1 * a.A2, a.A2, 1u};
    const uint32_t row_steps = mc_bisect_steps(a.segments);
    for (uint32_t e = threadIdx.x; e < total_v; e += kMcBlock) {
        // the last row whose first vertex is <= e: rows without vertices share their successor's first vertex
        uint32_t row = 0;
        for (uint32_t s = row_steps; s-- > 0u;) {
            const uint32_t probe = row + (1u << s);
            if (probe < a.segments && linfo[probe].x <= e) row = probe;
        }
        const uint4 w = linfo[row];
        // ... and in it the last sample with at most e - w.x edges before it (sample by sample, axis by axis)
        const uint32_t kth = e - w.x;
        uint32_t i = 0;
#pragma unroll
        for (int s = 4; s >= 0; --s) {
            const uint32_t probe = i + (1u << s), below = (1u << probe) - 1u;   // probe <= 31
            if (__popc(w.y & below) + __popc(w.z & below) + __popc(w.w & below) <= kth) i = probe;
        }
        const uint32_t below = (1u << i) - 1u, r = kth - (__popc(w.y & below) + __popc(w.z & below) + __popc(w.w & below));
        const uint32_t bx = (w.y >> i) & 1u, by = (w.z >> i) & 1u;
        const uint32_t axis = r == 0u ? (bx ? 0u : (by ? 1u : 2u)) : (r == 1u ? ((bx & by) ? 1u : 2u) : 2u);
        const uint32_t a0 = a.div_A1.div(row), a1 = row - a0 * a.A1;
        const uint32_t s = i + a.A2 * row;
        const float f1 = block_f[s], f2 = block_f[s + stride[axis]];
        const double t = (1.0 * (0.0 - (double)f1)) / ((double)f2 - (double)f1);
        const uint32_t pos[3] = {a0, a1, i};
        double v[3];
#pragma unroll
        for (uint32_t c = 0; c < 3; ++c) v[c] = (double)pos[c] + (c == axis ? t : 0.0);
        double* out = a.vertices + 3 * (size_t)(first.x + e);
        out[0] = v[1] * a.step + cx;
        out[1] = ((-v[0]) * a.step + cy) + a.y_offset;
        out[2] = v[2] * a.step + cz;
    }

    // ---- triangles: one lane, one 12-byte store each.  A round takes kMcTriangleWindow of the block's triangles: the
    // chunks' cells write {cell, case, which of its triangles} where the triangle's lane will look (a bisection over
    // the cells' first triangles, per triangle, cost a third of this loop)
    for (uint32_t w0 = 0; w0 < total_t; w0 += kMcTriangleWindow) {
        uint32_t running = chunk_first;
        for (uint32_t j = 0; j < per_lane; ++j)
            if (chunk0 + j < n_cells) {
                const uint32_t entry = cells[chunk0 + j], cube = mc_cube_block(lmask, entry & 255u, entry >> 8, a.A1);
                const uint32_t c = row_bytes[cube * 16u + 15u];
#pragma unroll
                for (uint32_t t = 0; t < 5u; ++t)   // at most five triangles per case
                    if (t < c && running + t - w0 < kMcTriangleWindow) which[running + t - w0] = entry | (cube << 13) | (t << 21);
                running += c;
            }
        __syncthreads();
        const uint32_t n = total_t - w0 < kMcTriangleWindow ? total_t - w0 : kMcTriangleWindow;
        for (uint32_t e = threadIdx.x; e < n; e += kMcBlock) {
            const uint32_t wh = which[e], row = wh & 255u, i = (wh >> 8) & 31u, cube = (wh >> 13) & 255u;
            const unsigned char* rb = row_bytes + cube * 16u + 3u * (wh >> 21);
            uint32_t id[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const uint32_t pk = rb[j], axis = pk >> 3;
                const uint32_t q = ((pk & 1u) ? a.A1 : 0u) + ((pk >> 1) & 1u), li = i + ((pk >> 2) & 1u);
                const uint4 w = linfo[row + q];
                const uint32_t below = low_bits(li);
                id[j] = first.x + w.x + __popc(w.y & below) + __popc(w.z & below) + __popc(w.w & below) +
                        (axis >= 1u ? (w.y >> li) & 1u : 0u) + (axis == 2u ? (w.z >> li) & 1u : 0u);
            }
            uint32_t* out = a.triangles + 3 * (size_t)(first.y + w0 + e);
            out[0] = id[0];
            out[1] = id[1];
            out[2] = id[2];
        }
        __syncthreads();
    }
}

// ---- binary STL records (reference rendering/stl_renderer.py:8-24 through numpy-stl 1.8.0) -------------
// One 50-byte record per triangle: normal, three corners (float32, little endian), a zero attribute word.
// Corners are the fp64 vertices rounded to float32 (the reference's assignment into the float32 `vectors`);
// normal = (v1 - v0) x (v2 - v0) in float32, not normalised, each component fl(fl(a*b) - fl(c*d)) as
// numpy.cross evaluates it (numpy-stl's update_normals on save).  Records are 2-byte aligned, so a workgroup
// assembles its 256 records in LDS as 16-bit halves and writes them out as aligned 16-byte words.
constexpr uint32_t kStlBlock = 256, kStlRecordBytes = 50;

__global__ __launch_bounds__(kStlBlock) void k_stl_records(const double* __restrict__ vertices,
                                                           const uint32_t* __restrict__ triangles, uint64_t n_triangles,
                                                           uint8_t* __restrict__ records)
{
    __shared__ __attribute__((aligned(16))) uint16_t rec[kStlBlock * (kStlRecordBytes / 2)];
    const uint64_t first = (uint64_t)blockIdx.x * kStlBlock, t = first + threadIdx.x;
    if (t < n_triangles) {
        float v[3][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double* src = vertices + 3 * (size_t)triangles[3 * t + c];
#pragma unroll
            for (int k = 0; k < 3; ++k) v[c][k] = (float)src[k];
        }
        const float ax = v[1][0] - v[0][0], ay = v[1][1] - v[0][1], az = v[1][2] - v[0][2];
        const float bx = v[2][0] - v[0][0], by = v[2][1] - v[0][1], bz = v[2][2] - v[0][2];
        const float f[12] = {ay * bz - az * by, az * bx - ax * bz, ax * by - ay * bx,
                             v[0][0], v[0][1], v[0][2], v[1][0], v[1][1], v[1][2], v[2][0], v[2][1], v[2][2]};
        uint16_t* r = rec + threadIdx.x * (kStlRecordBytes / 2);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const uint32_t u = __float_as_uint(f[i]);
            r[2 * i] = (uint16_t)(u & 0xffffu);
            r[2 * i + 1] = (uint16_t)(u >> 16);
        }
        r[24] = 0;
    }
    __syncthreads();
    const uint64_t left = n_triangles - first;
    const uint32_t bytes = (uint32_t)(left < kStlBlock ? left : kStlBlock) * kStlRecordBytes;
    uint8_t* out = records + first * kStlRecordBytes;   // a multiple of 12800: 16-byte aligned with the buffer
    const uint4* src16 = reinterpret_cast<const uint4*>(rec);
    for (uint32_t i = threadIdx.x; i < bytes / 16u; i += kStlBlock) reinterpret_cast<uint4*>(out)[i] = src16[i];
    const uint8_t* src8 = reinterpret_cast<const uint8_t*>(rec);
    for (uint32_t i = (bytes & ~15u) + threadIdx.x; i < bytes; i += kStlBlock) out[i] = src8[i];
}

}  // namespace sdfk
