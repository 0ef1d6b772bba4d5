// codecad_amd/csrc/mesh.hip
//
// hu_mesh_*: marching cubes over all leaf blocks of a subdivision and the binary STL records of the result -- the
// consumer the reference runs on the host, block by block (rendering/mesh.py:45-74 through PyMCubes,
// rendering/stl_renderer.py:8-24 through numpy-stl).  Kernels: mesh_kernels.hpp.
//
// Its own translation unit: nothing here touches the tape interpreter, and hip_util.hip is compiled with
// -mllvm -structurizecfg-skip-uniform-regions for the interpreter's dispatch loop (builder.py), an option that
// has miscompiled a divergent loop elsewhere (exchange.hip tells the story).  These kernels are loops over the set
// bits of per-lane masks -- exactly that shape -- so they are built without it.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <string>

#include "../../include/hip_util.h"
#include "mesh_kernels.hpp"

int hu_fail_external(int code, const char* message);   // hip_util.hip: sets the thread's last error

using namespace sdfk;

namespace {

int fail(int code, const char* message) { return hu_fail_external(code, message); }

#define HU_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return hu_fail_external(HU_ERR_HIP, (std::string(#expr) + ": " + hipGetErrorString(e_)).c_str()); \
    } while (0)

// Fills the shape part of McArgs; n_wg = workgroups of 256 segments over all blocks.
int mesh_shape(uint32_t n_blocks, const uint32_t dims[3], McArgs& a, uint64_t& n_wg)
{
    uint64_t samples;
    if (!dims) return fail(HU_ERR_BAD_ARG, "dims is NULL");
    if (dims[0] == 0 || dims[1] == 0 || dims[2] == 0) return fail(HU_ERR_BAD_ARG, "dims must be >= 1 on every axis");
    samples = (uint64_t)dims[0] * dims[1] * dims[2];
    if (samples > (1ull << 24)) return fail(HU_ERR_BAD_ARG, "a block may have at most 2^24 samples (256^3)");
    if (dims[0] > 65535u || dims[1] > 65535u || dims[2] > 65535u) return fail(HU_ERR_BAD_ARG, "block dims must be below 65536");
    a.A0 = dims[0];
    a.A1 = dims[1];
    a.A2 = dims[2];
    a.spr = mc_segments_per_row(dims[2]);
    a.segments = dims[0] * dims[1] * a.spr;
    a.div_A1 = make_fast_div(a.A1);
    a.div_spr = make_fast_div(a.spr);
    a.chunks = (a.segments + kMcBlock - 1) / kMcBlock;
    n_wg = (uint64_t)a.chunks * n_blocks;
    if (n_wg > 0x7fffffffull) return fail(HU_ERR_BAD_ARG, "too many workgroups in one launch");
    return HU_OK;
}

// a block of at most 256 rows of at most 32 samples: one workgroup owns it (mesh_kernels.hpp k_mc_block_*)
bool mesh_block_form(const McArgs& a)
{
    static const bool off = [] { const char* e = getenv("HU_MC_BLOCK_FORM"); return e && e[0] == '0'; }();
    return !off && a.chunks == 1u && a.spr == 1u && (uint64_t)(a.A0 - 1u) * (a.A1 - 1u) * (a.A2 - 1u) <= kMcBlockCells;
}

}  // namespace

extern "C" {

int hu_mesh_workgroups(uint32_t n_blocks, const uint32_t dims[3], uint64_t* n_workgroups, uint64_t* count_entries,
                       uint64_t* segments)
{
    if (!n_workgroups || !count_entries || !segments) return fail(HU_ERR_BAD_ARG, "NULL argument");
    McArgs a{};
    int rc;
    if ((rc = mesh_shape(n_blocks, dims, a, *n_workgroups))) return rc;
    *count_entries = *n_workgroups + 1 + (*n_workgroups + kMcScanTile - 1) / kMcScanTile;
    *segments = (uint64_t)a.segments * n_blocks;
    return HU_OK;
}

int hu_mesh_count(const float* fields_dev, uint32_t n_blocks, const uint32_t dims[3], uint32_t* masks_dev,
                  uint32_t* wg_counts_dev, void* stream)
{
    if (!wg_counts_dev || ((!fields_dev || !masks_dev) && n_blocks)) return fail(HU_ERR_BAD_ARG, "NULL argument");
    McArgs a{};
    uint64_t n_wg;
    int rc;
    if ((rc = mesh_shape(n_blocks, dims, a, n_wg))) return rc;
    a.fields = fields_dev;
    a.masks = masks_dev;
    a.wg_counts = reinterpret_cast<uint2*>(wg_counts_dev);
    // the tile totals of the scan live behind the totals entry: wg_counts_dev has n_wg + 1 + tiles entries
    const uint32_t n = (uint32_t)n_wg, tiles = (n + kMcScanTile - 1) / kMcScanTile;
    uint2* tile_totals = a.wg_counts + n + 1;
    if (n) {
        if (mesh_block_form(a)) {   // a block per workgroup: masks and counts in one pass (mesh_kernels.hpp)
            hipLaunchKernelGGL(k_mc_block_count, dim3(n), dim3(kMcBlock), 0, (hipStream_t)stream, a);
        } else {
            hipLaunchKernelGGL(k_mc_masks, dim3(n), dim3(kMcBlock), 0, (hipStream_t)stream, a);
            hipLaunchKernelGGL(k_mc_count, dim3(n), dim3(kMcBlock), 0, (hipStream_t)stream, a);
        }
        hipLaunchKernelGGL(k_mc_scan_tiles, dim3(tiles), dim3(kMcScanTile), 0, (hipStream_t)stream, a.wg_counts, n, tile_totals);
    }
    hipLaunchKernelGGL(k_mc_scan_totals, dim3(1), dim3(1024), 0, (hipStream_t)stream, tile_totals, tiles, a.wg_counts + n);
    if (n) hipLaunchKernelGGL(k_mc_scan_add, dim3(tiles), dim3(kMcScanTile), 0, (hipStream_t)stream, a.wg_counts, n, tile_totals);
    HU_HIP(hipGetLastError());
    return HU_OK;
}

int hu_mesh_emit(const float* fields_dev, const int32_t* blocks_dev, uint32_t n_blocks, double resolution,
                 const double origin[3], double step, const uint32_t dims[3], double y_offset,
                 const uint32_t* masks_dev, const uint32_t* wg_counts_dev, uint32_t* seg_info_dev, double* vertices_dev,
                 uint32_t* triangles_dev, void* stream)
{
    if (!origin) return fail(HU_ERR_BAD_ARG, "NULL argument");
    if (n_blocks == 0) return HU_OK;
    if (!fields_dev || !blocks_dev || !masks_dev || !wg_counts_dev || !seg_info_dev || !vertices_dev || !triangles_dev)
        return fail(HU_ERR_BAD_ARG, "NULL argument");
    McArgs a{};
    uint64_t n_wg;
    int rc;
    if ((rc = mesh_shape(n_blocks, dims, a, n_wg))) return rc;
    a.fields = fields_dev;
    a.blocks = reinterpret_cast<const int4*>(blocks_dev);
    a.res = resolution;
    a.ox = origin[0];
    a.oy = origin[1];
    a.oz = origin[2];
    a.step = step;
    a.y_offset = y_offset;
    a.masks = const_cast<uint32_t*>(masks_dev);
    a.wg_counts = reinterpret_cast<uint2*>(const_cast<uint32_t*>(wg_counts_dev));
    a.seg_info = reinterpret_cast<uint4*>(seg_info_dev);
    a.vertices = vertices_dev;
    a.triangles = triangles_dev;
    if (mesh_block_form(a)) {
        hipLaunchKernelGGL(k_mc_block_emit, dim3((uint32_t)n_wg), dim3(kMcBlock), 0, (hipStream_t)stream, a);
    } else {
        hipLaunchKernelGGL(k_mc_vertices, dim3((uint32_t)n_wg), dim3(kMcBlock), 0, (hipStream_t)stream, a);
        hipLaunchKernelGGL(k_mc_triangles, dim3((uint32_t)n_wg), dim3(kMcBlock), 0, (hipStream_t)stream, a);
    }
    HU_HIP(hipGetLastError());
    return HU_OK;
}

int hu_mesh_stl(const double* vertices_dev, const uint32_t* triangles_dev, uint64_t n_triangles, void* records_dev,
                void* stream)
{
    if (n_triangles == 0) return HU_OK;
    if (!vertices_dev || !triangles_dev || !records_dev) return fail(HU_ERR_BAD_ARG, "NULL argument");
    if (reinterpret_cast<uintptr_t>(records_dev) & 15u) return fail(HU_ERR_BAD_ARG, "records_dev must be 16-byte aligned");
    const uint64_t n_wg = (n_triangles + kStlBlock - 1) / kStlBlock;
    if (n_wg > 0x7fffffffull) return fail(HU_ERR_BAD_ARG, "too many triangles for one call");
    hipLaunchKernelGGL(k_stl_records, dim3((uint32_t)n_wg), dim3(kStlBlock), 0, (hipStream_t)stream, vertices_dev,
                       triangles_dev, n_triangles, static_cast<uint8_t*>(records_dev));
    HU_HIP(hipGetLastError());
    return HU_OK;
}

}  // extern "C"
