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
#include <hip/hiprtc.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

#include <dirent.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

extern char** environ;

#include "../../include/hip_util.h"
#include "kernels.hpp"
#include "launchers.hpp"
#include "tape.hpp"
#include "specialise.hpp"

using sdf::Rec;
using namespace sdfk;

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
        if (e_ != hipSuccess) {                                                              \
            (void)hipGetLastError(); /* reported through the return code: do not leave it sticky for the next launch check */ \
            return fail(HU_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
        }                                                                                    \
    } while (0)


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
    sdf::SpecProgram program;    // both programs on the host, kept for hu_tape_specialize (specialise.hpp)
    struct SpecKernels* spec = nullptr;
    std::string spec_source;     // the generated per-tape source, once it has been asked for (a tape's kernels are built and
    sdf::SpecMeta spec_meta;     // probed one by one: planetary's source takes tens of milliseconds to generate)
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
int launch_shape(const hu_tape_s* t, LaunchShape& ls, bool distance_only_kernel, int max_voxels_per_lane = 2,
                 bool lanes_are_independent = false)
{
    static const int forced = [] { const char* e = getenv("HU_VOXELS_PER_LANE"); return e ? atoi(e) : 0; }();
    const size_t lane_bytes = distance_only_kernel ? (size_t)t->n_point_slots * 16 + (size_t)t->n_result_slots * 4
                                                   : (size_t)t->n_slots * 16;
    // (with fused leaf records the scalar work per voxel fell and latency -- waves per SIMD -- took over for the full
    // program: sponge(4), 6 float4 slots: 3.17 ms with one voxel per lane (24 KiB per workgroup, 6 workgroups per
    // CU) against 3.74 ms with two (48 KiB, 3); csg_example, 3 slots: 0.96 against 1.06 ms the other way round)
    const size_t two_voxel_limit = distance_only_kernel ? 48 * 1024 : 32 * 1024;
    const int by_rule = (forced == 1 || forced == 2) ? forced : ((lane_bytes * 2 * 256 <= two_voxel_limit) ? 2 : 1);
    const int wanted = by_rule < max_voxels_per_lane ? by_rule : max_voxels_per_lane;
    const Rec* prog = distance_only_kernel ? t->recs_do_dev : t->recs_dev;
    ls.prog = prog;
    ls.n4 = (uint32_t)(distance_only_kernel ? t->n_point_slots : t->n_slots);
    for (int n = wanted; n >= 1; --n) {
        const size_t per_lane = (distance_only_kernel ? (size_t)t->n_point_slots * 16 + (size_t)t->n_result_slots * 4
                                                      : (size_t)t->n_slots * 16) * n;
        uint32_t bs = 256;
        while (bs > 64 && per_lane * bs > 48 * 1024) bs >>= 1;
        // The grid kernels' lanes share nothing, and the interpreter waits more than it computes (forcing 5 waves per SIMD
        // instead of 6 costs 14 %): where single-wavefront workgroups fit more wavefronts into a CU's LDS than workgroups of
        // 256 lanes, take them (sponge(4), 6 float4 values: 26 against 24 per CU, -1 %; sponge(5), 7 values: 22 against 20, -3 %).
        auto waves_per_cu = [&](uint32_t lanes) {
            const size_t groups = kMaxLds / (per_lane * lanes + kScratchBytes), waves = groups * (lanes / 64u);
            return waves < 32 ? waves : (size_t)32;
        };
        static const uint32_t forced_block = [] { const char* e = getenv("HU_BLOCK"); return e ? (uint32_t)atoi(e) : 0u; }();
        if (forced_block == 64u || forced_block == 128u || forced_block == 256u) bs = forced_block < bs ? forced_block : bs;
        else if (lanes_are_independent && bs == 256u && waves_per_cu(64u) > waves_per_cu(256u)) bs = 64u;
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
    if ((rc = allow_big_lds(k_grid_eval<InterpEval<false>, 0, N>))) return rc;
    if ((rc = allow_big_lds(k_grid_eval<InterpEval<false>, 1, N>))) return rc;
    if ((rc = allow_big_lds(k_grid_eval<InterpEval<true>, 1, N>))) return rc;
    if ((rc = allow_big_lds(k_grid_eval_blocks<InterpEval<false>, 0, N>))) return rc;
    if ((rc = allow_big_lds(k_grid_eval_blocks<InterpEval<false>, 1, N>))) return rc;
    if ((rc = allow_big_lds(k_grid_eval_blocks<InterpEval<true>, 1, N>))) return rc;
    if ((rc = allow_big_lds(k_classify<InterpEval<false>, false, false, N>))) return rc;
    if ((rc = allow_big_lds(k_classify<InterpEval<false>, false, true, N>))) return rc;
    if ((rc = allow_big_lds(k_classify<InterpEval<false>, true, false, N>))) return rc;
    if ((rc = allow_big_lds(k_classify<InterpEval<false>, true, true, N>))) return rc;
    if ((rc = allow_big_lds(k_classify<InterpEval<true>, false, false, N>))) return rc;
    if ((rc = allow_big_lds(k_classify<InterpEval<true>, false, true, N>))) return rc;
    if ((rc = allow_big_lds(k_classify<InterpEval<true>, true, false, N>))) return rc;
    if ((rc = allow_big_lds(k_classify<InterpEval<true>, true, true, N>))) return rc;
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
    HU_HIP(hu_render::allow_big_lds(kMaxLds));   // the ray caster and the bitmap kernels (render.hip)
    done_for_device = dev;
    return HU_OK;
}

// Launches of per-tape code go over 16^3 boxes of compact 4 x 4 x 8 bricks (kernels.hpp box_eval) when the slab's or the
// block's extents allow it without ragged bricks (the grid kernels take ragged boxes too: k_grid_eval_ragged).
uint32_t brick_tiles(uint32_t nx, uint32_t sy, uint32_t sz)
{
    return (nx % 4u == 0u && sy % 4u == 0u && sz % 8u == 0u) ? 1u : 0u;
}
// ... and are boxes worth it: do at least half of the voxels of the bricks they would walk exist?  (A 2D grid is one voxel deep:
// an eighth.)  Else the launch goes over runs of cells (k_grid_eval_runs).
bool boxes_worthwhile(uint64_t nx, uint64_t sy, uint64_t sz)
{
    const uint64_t padded = ((nx + 3u) & ~3ull) * ((sy + 3u) & ~3ull) * ((sz + 7u) & ~7ull);
    return padded <= 2u * nx * sy * sz;
}

// How many units (blocks / parents) of `chunks` workgroups of `threads` lanes go into one launch: a grid may
// have at most 2^31 - 1 workgroups and 2^32 - 1 work-items; longer lists are launched in pieces.
uint32_t units_per_launch(uint32_t chunks, uint32_t threads)
{
    const uint64_t by_items = 0xffffffffull / threads / chunks, by_groups = 0x7fffffffull / chunks;
    const uint64_t n = by_items < by_groups ? by_items : by_groups;
    return n < 1 ? 1u : (uint32_t)n;
}

int check_dims(const uint32_t dims[3], uint64_t& cells)
{
    if (!dims) return fail(HU_ERR_BAD_ARG, "dims is NULL");
    if (dims[0] == 0 || dims[1] == 0 || dims[2] == 0) return fail(HU_ERR_BAD_ARG, "dims must be >= 1 on every axis");
    cells = (uint64_t)dims[0] * dims[1] * dims[2];
    return HU_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// Per-tape specialisation (the analogue of the reference's generate_fixed_eval_source_code,
// nodes/codegen.py:137-204): the decoded program is unrolled into straight-line HIP source --
// one exec_one call per record with the record as a literal -- and compiled with hipRTC against
// the SAME op library (interp.hpp).  With a literal record the opcode switch folds to one case
// and every slot index is a constant, so the register file dissolves into VGPRs: no dispatch,
// no scalar fetch, no LDS.  Kernels that only read the distance get the direction arithmetic
// removed by dead-code elimination.  The arithmetic is the interpreter's, operation for
// operation, so results are identical (tests run the parity suite on specialised tapes).
// ------------------------------------------------------------------------------------------
// A tape's kernels may sit in several modules: a synchronous build (hu_tape_specialize_groups without a cached image)
// compiles the requested set as ONE module, the background builds (codecad_amd/hip_util/buffer.py) make one image per
// KERNEL, side by side in several processes.  (Rounds 1-3 built all kernels, then families, together: in the plain form
// hipRTC spent its time on the one straight-line tape function every kernel shared.  The deferred form over boxes
// instantiates its own functions per kernel -- sponge(4), family of five: 1.40 s, its kernels one by one: 0.09 + 0.47 +
// 0.55 + 0.22 + 0.27 s with the precompiled header below -- so a launch's kernel is ready in a third of the time.)
struct SpecKernels {
    std::vector<hipModule_t> modules;   // one per hu_tape_specialize_groups call that built something
    uint32_t groups = 0;                // the kernels that are loaded (bit i: kernel i of kSpecKernelNames)
    hipFunction_t dense[2] = {nullptr, nullptr};
    hipFunction_t blocks[2] = {nullptr, nullptr};
    // tapes with box code: the same over runs of cells, for extents that are no multiples of (4, 4, 8) (kernels.hpp k_grid_eval_ragged)
    hipFunction_t dense_ragged[2] = {nullptr, nullptr};
    hipFunction_t blocks_ragged[2] = {nullptr, nullptr};
    // ... and over runs of cells, in the in-place form, where boxes would be mostly padding (2D grids: kernels.hpp k_grid_eval_runs)
    hipFunction_t dense_runs[2] = {nullptr, nullptr};
    hipFunction_t blocks_runs[2] = {nullptr, nullptr};
    hipFunction_t classify[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // [MASS][BATCH]
    hipFunction_t ray_caster = nullptr, bitmap = nullptr;
    hipFunction_t box_masks = nullptr;   // k_box_masks (box pruning), part of every family that launches over boxes
    bool deferred = false;   // the module was generated with deferred directions (specialise.hpp): dense launches use bricks
    double coord_limit = 0.0;   // a launch whose sample coordinates all stay below this sets sdf::kFlagInRange (specialise.hpp)
    int tabs[6] = {0, 0, 0, 0, 0, 0};   // columns of a box's tables: x, y, z, xy, xz, yz (specialise.hpp)
    int prune_words = 0;     // 32-bit words of a box's pruning mask (0: nothing to prune in this tape)
    int prune_bits = 0;
    bool prune_all = false;  // the float4 code is guarded too (else only the distance walks: float4 launches skip the mask kernel)
    // the mask buffers of launches over boxes, one per stream that launched any (the mask kernel and the launch it prepares
    // are neighbours on their stream, so a stream's launches can share one buffer); grown when a launch needs more
    struct MaskBuffer { hipStream_t stream; uint32_t* ptr; size_t bytes; };
    std::vector<MaskBuffer> mask_buffers;
};

namespace {

constexpr int kSpecVoxelsPerLane = 2;
constexpr uint32_t kSpecBlock = 256;

// The flags a per-tape kernel is launched with: kFlagInRange when no sample coordinate of the launch can exceed the
// tape's limit (HU_INRANGE=0 never sets it: measurements)
uint32_t spec_flags(const hu_tape_s* t, double max_abs_coordinate)
{
    static const bool off = [] { const char* e = getenv("HU_INRANGE"); return e && e[0] == '0'; }();
    return (!off && t->spec && max_abs_coordinate < t->spec->coord_limit) ? sdf::kFlagInRange : 0u;
}
// |coordinate| of any sample of a grid: corner + step * [0, n)
double grid_reach(const float corner[3], float step, const uint32_t dims[3])
{
    double m = 0.0;
    for (int c = 0; c < 3; ++c) {
        const double a = std::fabs((double)corner[c]), b = std::fabs((double)corner[c] + (double)step * (double)dims[c]);
        m = std::max(m, std::max(a, b));
    }
    return m;
}
// ... of any sample of blocks whose integer corners (32-bit) come from a device list: |int| * resolution + origin + the block
double list_reach(double resolution, double ox, double oy, double oz, double block_extent)
{
    return 2147483648.0 * std::fabs(resolution) + std::max(std::fabs(ox), std::max(std::fabs(oy), std::fabs(oz))) + std::fabs(block_extent);
}

// HU_DEFER_DIRECTIONS=0 keeps the plain straight-line form of every tape (measurements, bisecting)
bool defer_directions()
{
    static const bool off = [] { const char* e = getenv("HU_DEFER_DIRECTIONS"); return e && e[0] == '0'; }();
    return !off;
}

std::string generate_source(const hu_tape_s* t, sdf::SpecMeta* meta = nullptr)
{
    return sdf::specialised_source(t->program, defer_directions(), meta);
}
// LDS bytes of a box's tables (sdf::BoxTabs: 16 entries per single-axis column; pair columns of 16 rows of 17 / 24 floats)
uint32_t box_table_bytes(const SpecKernels* k)
{
    return (uint32_t)((k->tabs[0] + k->tabs[1] + k->tabs[2]) * sdf::BoxTabs::kAxis + (k->tabs[3] + k->tabs[4]) * sdf::BoxTabs::kPairX +
                      k->tabs[5] * sdf::BoxTabs::kPairYZ) * 4u;
}

void keep_programs(hu_tape_s* t, const sdf::DecodedTape& d)
{
    t->program.full = d.recs;
    t->program.dist = d.recs_do;
    t->program.n_slots = d.n_slots;
    t->program.n_point_slots = d.n_point_slots;
    t->program.n_result_slots = d.n_result_slots;
}

struct SpecEval { const float* extra; uint32_t flags; };  // same layout as the generated sdfk::JitEval

constexpr int kSpecKernelCount = 19;
// bit i of a `groups` mask is kernel i below; the families of include/hip_util.h (HU_SPEC_*) are sets of them (the mask
// kernel of box pruning belongs to every family that launches over boxes)
constexpr uint32_t spec_bit(int i) { return 1u << i; }
constexpr uint32_t kSpecGroupOf[kSpecKernelCount] = {spec_bit(0), spec_bit(1), spec_bit(2), spec_bit(3), spec_bit(4), spec_bit(5), spec_bit(6), spec_bit(7),
                                                      spec_bit(8), spec_bit(9), spec_bit(10), spec_bit(11), spec_bit(12), spec_bit(13), spec_bit(14),
                                                      spec_bit(15), spec_bit(16), spec_bit(17), spec_bit(18)};
static_assert(HU_SPEC_DENSE == (spec_bit(0) | spec_bit(1) | spec_bit(10) | spec_bit(11) | spec_bit(12) | spec_bit(15) | spec_bit(16)), "hip_util.h");
static_assert(HU_SPEC_BLOCKS == (spec_bit(2) | spec_bit(3) | spec_bit(10) | spec_bit(13) | spec_bit(14) | spec_bit(17) | spec_bit(18)), "hip_util.h");
static_assert(HU_SPEC_CLASSIFY == (spec_bit(4) | spec_bit(5) | spec_bit(6) | spec_bit(7) | spec_bit(10)), "hip_util.h");
static_assert(HU_SPEC_RENDER == (spec_bit(8) | spec_bit(9)), "hip_util.h");
static_assert(HU_SPEC_ALL == (HU_SPEC_DENSE | HU_SPEC_BLOCKS | HU_SPEC_CLASSIFY | HU_SPEC_RENDER) && HU_SPEC_ALL == spec_bit(kSpecKernelCount) - 1u, "hip_util.h");
const char* const kSpecKernelNames[kSpecKernelCount] = {
    "sdfk::k_grid_eval<sdfk::JitEval, 0, 2>",           "sdfk::k_grid_eval<sdfk::JitEval, 1, 2>",
    "sdfk::k_grid_eval_blocks<sdfk::JitEval, 0, 2>",    "sdfk::k_grid_eval_blocks<sdfk::JitEval, 1, 2>",
    "sdfk::k_classify<sdfk::JitEval, false, false, 2>", "sdfk::k_classify<sdfk::JitEval, false, true, 2>",
    "sdfk::k_classify<sdfk::JitEval, true, false, 2>",  "sdfk::k_classify<sdfk::JitEval, true, true, 2>",
    "sdfk::k_ray_caster<sdfk::JitEval>",                "sdfk::k_bitmap<sdfk::JitEval>",
    "sdfk::k_box_masks<sdfk::JitEval>",
    "sdfk::k_grid_eval_ragged<sdfk::JitEval, 0, 2>",        "sdfk::k_grid_eval_ragged<sdfk::JitEval, 1, 2>",
    "sdfk::k_grid_eval_blocks_ragged<sdfk::JitEval, 0, 2>", "sdfk::k_grid_eval_blocks_ragged<sdfk::JitEval, 1, 2>",
    "sdfk::k_grid_eval_runs<sdfk::JitEval, 0, 2>",          "sdfk::k_grid_eval_runs<sdfk::JitEval, 1, 2>",
    "sdfk::k_grid_eval_blocks_runs<sdfk::JitEval, 0, 2>",   "sdfk::k_grid_eval_blocks_runs<sdfk::JitEval, 1, 2>"};

// Box pruning: run the tape's mask kernel for the `m.n_boxes` workgroups of the launch that follows on `stream` and hand
// back their masks -- or NULL (nothing to prune in this tape, HU_PRUNE_RUN=0, or a buffer that would have to grow while
// the stream is being captured into a graph): the launch then treats everything as alive.
int prepare_masks(hu_tape_s* t, MaskArgs& m, hipStream_t stream, const uint32_t** masks)
{
    *masks = nullptr;
    SpecKernels* k = t->spec;
    if (!k || !k->box_masks || k->prune_words <= 0 || m.n_boxes == 0) return HU_OK;
    static const bool off = [] { const char* e = getenv("HU_PRUNE_RUN"); return e && e[0] == '0'; }();
    if (off) return HU_OK;
    const size_t bytes = (size_t)m.n_boxes * (size_t)k->prune_words * sizeof(uint32_t);
    SpecKernels::MaskBuffer* buf = nullptr;
    for (auto& b : k->mask_buffers) if (b.stream == stream) buf = &b;
    if (!buf || buf->bytes < bytes) {
        hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
        if (stream && hipStreamIsCapturing(stream, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone) return HU_OK;
        (void)hipGetLastError();
        if (!buf) { k->mask_buffers.push_back(SpecKernels::MaskBuffer{stream, nullptr, 0}); buf = &k->mask_buffers.back(); }
        // (hipFree waits for the device: a launch still reading the old buffer has finished before it goes)
        if (buf->ptr) HU_HIP(hipFree(buf->ptr));
        buf->ptr = nullptr; buf->bytes = 0;
        const size_t want = bytes + bytes / 2 + 4096;
        HU_HIP(hipMalloc((void**)&buf->ptr, want));
        buf->bytes = want;
    }
    m.out = buf->ptr;
    SpecEval ev{t->extra_dev, 0u};
    void* args[] = {&ev, &m};
    HU_HIP(hipModuleLaunchKernel(k->box_masks, (m.n_boxes + 63u) / 64u, 1, 1, 64, 1, 1, 0, stream, args, nullptr));
    *masks = buf->ptr;
    return HU_OK;
}

}  // namespace

// for the other translation units of the library (sort.hip)
int hu_fail_external(int code, const char* message)
{
    if (code == HU_ERR_HIP) (void)hipGetLastError();  // reported through the return code, not left sticky
    return fail(code, message);
}

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
    t->n_instr = d.n_instructions;
    t->n_regs = d.n_regs;
    t->n_slots = d.n_slots;
    t->n_point_slots = d.n_point_slots;
    t->n_result_slots = d.n_result_slots;
    t->flags = d.direction_feeds_distance ? 1 : 0;
    keep_programs(t, d);
    // the device holds the interpreter's (fused) programs; per-tape code is generated from the unfused ones
    hipError_t e = hipMalloc((void**)&t->recs_dev, d.fused.size() * sizeof(Rec));
    if (e == hipSuccess && !d.fused_do.empty()) {
        e = hipMalloc((void**)&t->recs_do_dev, d.fused_do.size() * sizeof(Rec));
        if (e == hipSuccess) e = hipMemcpy(t->recs_do_dev, d.fused_do.data(), d.fused_do.size() * sizeof(Rec), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMalloc((void**)&t->extra_dev, d.extra.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(t->recs_dev, d.fused.data(), d.fused.size() * sizeof(Rec), hipMemcpyHostToDevice);
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
    if (t->spec) {
        for (auto& b : t->spec->mask_buffers) (void)hipFree(b.ptr);
        for (hipModule_t m : t->spec->modules) (void)hipModuleUnload(m);
        delete t->spec;
    }
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
    // (pieces of a slab too long for one launch start at multiples of 16 planes: they are ragged only where the slab is)
    const uint32_t spec_max_x = [&] { const uint32_t m = (uint32_t)((1ull << 30) / plane); return m >= 16u ? m & ~15u : m; }();
    // extents that are no multiples of (4, 4, 8) take the kernel whose boxes may end anywhere: an image of its own -- while it is
    // still being built the interpreter serves such launches
    const bool spec_ragged = t->spec && t->spec->deferred && !(brick_tiles(x_count, dims[1], dims[2]) && (x_count <= spec_max_x || spec_max_x % 4u == 0u));
    // ... unless boxes would be mostly padding: then over runs of cells, in the in-place form (a third image)
    const bool spec_runs = spec_ragged && !boxes_worthwhile(x_count < spec_max_x ? x_count : spec_max_x, dims[1], dims[2]);
    if (t->spec && t->spec->dense[layout] && (!spec_ragged || (spec_runs ? t->spec->dense_runs[layout] : t->spec->dense_ragged[layout]))) {
        const uint32_t max_x = spec_max_x;
        SpecEval ev{t->extra_dev, spec_flags(t, grid_reach(corner, step, dims))};
        float cx = corner[0], cy = corner[1], cz = corner[2];
        uint32_t sx = dims[0];
        Dim sy = make_dim(dims[1]), sz = make_dim(dims[2]);
        for (uint32_t done = 0; done < x_count;) {
            const uint32_t nx = (x_count - done < max_x) ? (x_count - done) : max_x;
            uint32_t n_cells = (uint32_t)(nx * plane), xs = x0 + done;
            void* o = (layout == 0) ? (void*)(static_cast<float4*>(out_dev) + (size_t)done * plane) : out_dev;
            // boxes of compact bricks pay where a wavefront's work depends on how many primitives win in it (deferred
            // directions) and where the tape has tables to fill; a tape that is bound by its store stream keeps the runs
            // along z (2 KiB contiguous per wavefront: sphere, 512^3 float4: 0.34 ms in runs, 0.42 ms in bricks)
            uint32_t boxes = (t->spec->deferred && !spec_runs) ? 1u : 0u;
            const bool ragged = boxes && !brick_tiles(nx, dims[1], dims[2]);
            // runs of cells (HU_RUN_BLOCK: 64 / 128 / 256 lanes per workgroup, for measurements: the lanes of a run kernel share
            // nothing, but single-wavefront workgroups were SLOWER on the store-bound tapes -- box, 512^3 float4: 0.409 against
            // 0.373 ms, its distance grid 0.265 against 0.180 ms)
            static const uint32_t run_block = [] { const char* e = getenv("HU_RUN_BLOCK"); const int v = e ? atoi(e) : 0; return (v == 64 || v == 128 || v == 256) ? (uint32_t)v : 256u; }();
            const uint32_t block = boxes ? kSpecBlock : run_block;
            // A tape of a primitive or two (a box, a sphere: per-tape code in the plain form, over runs) is bound by its
            // store stream, and that stream runs FASTER with fewer wavefronts in flight: box, 512^3 float4: 0.373 ms at eight
            // wavefronts per SIMD (what its 20-odd registers allow), 0.328 ms at four -- the interpreter's rate, whose LDS
            // register file holds it near there anyway (measured with -DSDF_WAVES_PER_EU: 2 / 4 / 6 / 8 -> 0.382 / 0.328 /
            // 0.353 / 0.373 ms).  So such a launch asks for 40 KiB of LDS it never touches: four workgroups, sixteen
            // wavefronts per CU.  (HU_STORE_BOUND_LDS=0: off; longer tapes -- sponge(4): 0.407 -> 0.429 ms at four -- keep all.)
            static const uint32_t store_bound_kib = [] { const char* e = getenv("HU_STORE_BOUND_LDS"); const int v = e ? atoi(e) : 40; return (uint32_t)((v >= 0 && v <= 64) ? v : 40); }();
            // (float4 launches only: the float grid of the same tape stores a quarter of the bytes and is slowed by the limit --
            // box, 512^3: 0.182 -> 0.206 ms; HU_STORE_BOUND_LDS=<KiB>, 0..64: 32-40 are the best, 53 -- three workgroups per CU,
            // the best for stores ALONE, tools/experiments/store_patterns.hip -- leaves the arithmetic too few wavefronts: 0.38 ms)
            const uint32_t idle_lds = (layout == 0 && !boxes && !t->spec->deferred && t->n_instr <= 16) ? store_bound_kib * 1024u : 0u;
            const uint32_t per_block = block * kSpecVoxelsPerLane;
            uint32_t grid = (n_cells + per_block - 1) / per_block;
            const uint32_t* masks = nullptr;
            if (boxes) {
                grid = ((nx + 15u) / 16u) * ((dims[1] + 15u) / 16u) * ((dims[2] + 15u) / 16u);   // a workgroup per 16^3 box
                MaskArgs m{};
                m.mode = 0u; m.n_boxes = grid; m.chunks = grid; m.boxes_y = (dims[1] + 15u) / 16u; m.boxes_z = (dims[2] + 15u) / 16u;
                m.nx = nx; m.ny = dims[1]; m.nz = dims[2]; m.xs0 = xs; m.cx = cx; m.cy = cy; m.cz = cz; m.step = step;
                if ((layout == 1 || t->spec->prune_all) && (rc = prepare_masks(t, m, (hipStream_t)stream, &masks))) return rc;
            }
            if (spec_runs) {
                void* args[] = {&ev, &cx, &cy, &cz, &step, &sx, &sy, &sz, &xs, &n_cells, &o};
                HU_HIP(hipModuleLaunchKernel(t->spec->dense_runs[layout], grid, 1, 1, block, 1, 1, 0u, (hipStream_t)stream, args, nullptr));
            } else {
                void* args[] = {&ev, &cx, &cy, &cz, &step, &sx, &sy, &sz, &xs, &n_cells, &boxes, &o, &masks};
                HU_HIP(hipModuleLaunchKernel(ragged ? t->spec->dense_ragged[layout] : t->spec->dense[layout], grid, 1, 1, block, 1, 1,
                                             boxes ? box_table_bytes(t->spec) : idle_lds, (hipStream_t)stream, args, nullptr));
            }
            done += nx;
        }
        return HU_OK;
    }
    LaunchShape ls;
    if ((rc = launch_shape(t, ls, layout == 1 && distance_only(t), 2, true))) return rc;
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
    hipLaunchKernelGGL((k_grid_eval<InterpEval<D>, L, NV>), dim3(blocks), dim3(ls.block), ls.lds, (hipStream_t)stream, \
                       (InterpEval<D>{ls.prog, t->extra_dev, ls.n4}), corner[0], corner[1], corner[2], step, dims[0],  \
                       make_dim(dims[1]), make_dim(dims[2]), x0 + done, n_cells,                                       \
                       0u /* the interpreter evaluates every primitive anyway: runs along z */, o, (const uint32_t*)nullptr)
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

// n_dev == NULL: exactly n_blocks blocks; else the launch covers n_blocks and workgroups past *n_dev leave at once
static int grid_eval_blocks_impl(hu_tape t, const int32_t* blocks_dev, uint32_t n_blocks, const uint32_t* n_dev, double resolution,
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
    const bool blocks_ragged = t->spec && t->spec->deferred && !brick_tiles(dims[0], dims[1], dims[2]);
    const bool blocks_runs = blocks_ragged && !boxes_worthwhile(dims[0], dims[1], dims[2]);     // (boxes would be mostly padding)
    if (t->spec && t->spec->blocks[layout] && (!blocks_ragged || (blocks_runs ? t->spec->blocks_runs[layout] : t->spec->blocks_ragged[layout]))) {
        const uint32_t per_block = kSpecBlock * kSpecVoxelsPerLane;
        uint32_t chunks = (uint32_t)((cells + per_block - 1) / per_block);
        // deferred-direction code over compact bricks (kernels.hpp): a workgroup per box of up to 16^3 voxels of the block,
        // its wavefronts walking 4 x 4 x 8 bricks along x; `bricks` carries the boxes along y and z
        uint32_t bricks = 0u;
        const bool ragged = blocks_ragged && !blocks_runs;      // boxes that may end anywhere (k_grid_eval_blocks_ragged)
        if (t->spec->deferred && !blocks_runs) {
            const uint32_t bxn = (dims[0] + 15u) / 16u, byn = (dims[1] + 15u) / 16u, bzn = (dims[2] + 15u) / 16u;
            bricks = (byn << 16) | bzn;
            chunks = bxn * byn * bzn;
        }
        const double extent = (double)step * (double)std::max(dims[0], std::max(dims[1], dims[2]));
        SpecEval ev{t->extra_dev, spec_flags(t, list_reach(resolution, origin[0], origin[1], origin[2], extent))};
        const int4* b = (const int4*)blocks_dev;
        double res = resolution, ox = origin[0], oy = origin[1], oz = origin[2];
        uint32_t sx = dims[0];
        Dim sy = make_dim(dims[1]), sz = make_dim(dims[2]);
        const uint32_t piece = units_per_launch(chunks, kSpecBlock);
        for (uint32_t b0 = 0; b0 < n_blocks; b0 += piece) {
            uint32_t first = b0;
            const uint32_t count = n_blocks - b0 < piece ? n_blocks - b0 : piece;
            const uint32_t* masks = nullptr;
            if (bricks) {
                MaskArgs m{};
                m.mode = 1u; m.n_boxes = chunks * count; m.chunks = chunks; m.boxes_y = bricks >> 16; m.boxes_z = bricks & 0xffffu;
                m.nx = dims[0]; m.ny = dims[1]; m.nz = dims[2]; m.unit_base = b0; m.units = b; m.n_units_dev = n_dev; m.step = step;
                m.res = res; m.ox = ox; m.oy = oy; m.oz = oz;
                if ((layout == 1 || t->spec->prune_all) && (rc = prepare_masks(t, m, (hipStream_t)stream, &masks))) return rc;
            }
            if (blocks_runs) {
                void* args[] = {&ev, &b, &n_dev, &first, &chunks, &res, &ox, &oy, &oz, &step, &sx, &sy, &sz, &out_dev};
                HU_HIP(hipModuleLaunchKernel(t->spec->blocks_runs[layout], chunks * count, 1, 1, kSpecBlock, 1, 1, 0u, (hipStream_t)stream, args, nullptr));
            } else {
                void* args[] = {&ev, &b, &n_dev, &first, &chunks, &bricks, &res, &ox, &oy, &oz, &step, &sx, &sy, &sz, &out_dev, &masks};
                HU_HIP(hipModuleLaunchKernel(ragged ? t->spec->blocks_ragged[layout] : t->spec->blocks[layout], chunks * count, 1, 1, kSpecBlock, 1, 1,
                                             bricks ? box_table_bytes(t->spec) : 0u, (hipStream_t)stream, args, nullptr));
            }
        }
        return HU_OK;
    }
    LaunchShape ls;
    if ((rc = launch_shape(t, ls, layout == 1 && distance_only(t), 2, true))) return rc;
    if ((rc = ensure_attrs())) return rc;
    const uint32_t per_block = ls.block * ls.voxels_per_lane;
    const uint32_t chunks = (uint32_t)((cells + per_block - 1) / per_block);
    const dim3 block(ls.block);
    const uint32_t piece = units_per_launch(chunks, ls.block);
#define HU_LAUNCH_BLOCKS(L, D, NV)                                                                                 \
    hipLaunchKernelGGL((k_grid_eval_blocks<InterpEval<D>, L, NV>), grid, block, ls.lds, (hipStream_t)stream,           \
                       (InterpEval<D>{ls.prog, t->extra_dev, ls.n4}), (const int4*)blocks_dev, n_dev, b0, chunks, 0u, resolution, \
                       origin[0], origin[1], origin[2], step, dims[0], make_dim(dims[1]), make_dim(dims[2]), out_dev, (const uint32_t*)nullptr)
    const bool d_only = layout == 1 && distance_only(t);
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += piece) {
        const dim3 grid(chunks * (n_blocks - b0 < piece ? n_blocks - b0 : piece));
        if (ls.voxels_per_lane == 2) {
            if (layout == 0) HU_LAUNCH_BLOCKS(0, false, 2);
            else if (d_only) HU_LAUNCH_BLOCKS(1, true, 2);
            else HU_LAUNCH_BLOCKS(1, false, 2);
        } else {
            if (layout == 0) HU_LAUNCH_BLOCKS(0, false, 1);
            else if (d_only) HU_LAUNCH_BLOCKS(1, true, 1);
            else HU_LAUNCH_BLOCKS(1, false, 1);
        }
    }
#undef HU_LAUNCH_BLOCKS
    HU_HIP(hipGetLastError());
    return HU_OK;
}

int hu_grid_eval_blocks(hu_tape t, const int32_t* blocks_dev, uint32_t n_blocks, double resolution,
                        const double origin[3], float step, const uint32_t dims[3], int layout,
                        void* out_dev, void* stream)
{
    return grid_eval_blocks_impl(t, blocks_dev, n_blocks, nullptr, resolution, origin, step, dims, layout, out_dev, stream);
}

int hu_grid_eval_blocks_indirect(hu_tape t, const int32_t* blocks_dev, const uint32_t* n_blocks_dev, uint32_t max_blocks,
                                 double resolution, const double origin[3], float step, const uint32_t dims[3], int layout,
                                 void* out_dev, void* stream)
{
    if (!n_blocks_dev) return fail(HU_ERR_BAD_ARG, "n_blocks_dev is NULL");
    return grid_eval_blocks_impl(t, blocks_dev, max_blocks, n_blocks_dev, resolution, origin, step, dims, layout, out_dev, stream);
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
    if (t->spec && t->spec->classify[MASS ? 1 : 0][BATCH ? 1 : 0]) {
        const uint32_t per_block = kSpecBlock * kSpecVoxelsPerLane;
        a.sx = dims[0]; a.sy = dims[1]; a.sz = dims[2];
        a.dy = make_dim(dims[1]); a.dz = make_dim(dims[2]);
        a.chunks = (uint32_t)((cells + per_block - 1) / per_block);
        a.scratch_offset = 0;
        a.boxes = 0;
        // Grids of more than one workgroup's worth of cells go over 16^3 boxes with their tables in LDS (kernels.hpp
        // k_classify / box_eval) where the extents allow it.  (Up to 256 cells the lane-order compaction of ONE workgroup

        // And only launches that fill the chip several times over: a box is one workgroup where the path below has eight,
        // and a level of a few hundred parents is a latency exercise (measured: C5's 704 parents of 16^3 0.21 -> 0.33 ms
        // over boxes; C3's mass properties at grid 8, 167 000 parents in the last level, 0.94 -> 0.80 ms).
        const uint64_t bxn = (dims[0] + 15u) / 16u, byn = (dims[1] + 15u) / 16u, bzn = (dims[2] + 15u) / 16u;
        // A tape with box pruning is another matter: only the box path prunes, the path below evaluates every primitive for
        // every sample (planetary, mass properties at grid 64, resolution 0.5: 45 parents = 2880 boxes: 0.93 -> 0.34 ms over boxes;
        // resolution 1.0: 8 parents, 0.32 -> 0.27 ms; tools/experiments/mass_scale.py): such tapes take the boxes from 64 on.
        uint64_t enough = t->spec->prune_bits > 0 ? 64u : 8192u;
        if (const char* e = getenv("HU_CLASSIFY_BOX_MIN")) enough = (uint64_t)atoll(e);   // (read per launch: the tests switch it)
        if (t->spec->deferred && cells > 256u && bxn * byn * bzn * n_parents >= enough && boxes_worthwhile(dims[0], dims[1], dims[2])) {   // (any extents: the rims of a box are predicated; not where boxes would be mostly padding: 2D)
            a.boxes = ((uint32_t)byn << 16) | (uint32_t)bzn;
            a.chunks = (uint32_t)(bxn * byn * bzn);
            a.scratch_offset = box_table_bytes(t->spec);
        }
        // (mass-property parents are fp64 corners on the device: their magnitude is not the host's to know)
        const double extent = (double)a.step * (double)std::max(dims[0], std::max(dims[1], dims[2]));
        const float c3[3] = {a.cx, a.cy, a.cz};
        const double reach = !BATCH ? grid_reach(c3, a.step, dims)
                             : MASS ? HUGE_VAL
                                    : list_reach(a.res, a.ox, a.oy, a.oz, extent + std::fabs((double)a.int_step * a.res));
        SpecEval ev{t->extra_dev, spec_flags(t, reach)};
        const uint32_t piece = units_per_launch(a.chunks, kSpecBlock);
        for (uint32_t p0 = 0; p0 < n_parents; p0 += piece) {
            a.parent_base = p0;
            a.masks = nullptr;
            if (a.boxes) {
                MaskArgs m{};
                m.mode = !BATCH ? 2u : MASS ? 4u : 3u;
                m.n_boxes = a.chunks * (n_parents - p0 < piece ? n_parents - p0 : piece); m.chunks = a.chunks;
                m.boxes_y = a.boxes >> 16; m.boxes_z = a.boxes & 0xffffu; m.nx = dims[0]; m.ny = dims[1]; m.nz = dims[2];
                m.unit_base = p0; m.units = a.parents; m.n_units_dev = a.n_parents_dev; m.cx = a.cx; m.cy = a.cy; m.cz = a.cz; m.step = a.step;
                m.int_step = a.int_step; m.dimension = a.dimension; m.res = a.res; m.ox = a.ox; m.oy = a.oy; m.oz = a.oz; m.s = a.s;
                if ((rc = prepare_masks(t, m, (hipStream_t)stream, &a.masks))) return rc;
            }
            void* args[] = {&ev, &a};
            HU_HIP(hipModuleLaunchKernel(t->spec->classify[MASS ? 1 : 0][BATCH ? 1 : 0],
                                         a.chunks * (n_parents - p0 < piece ? n_parents - p0 : piece), 1, 1, kSpecBlock, 1, 1,
                                         (unsigned)(a.scratch_offset + kScratchBytes), (hipStream_t)stream, args, nullptr));
        }
        return HU_OK;
    }
    LaunchShape ls;
    if ((rc = launch_shape(t, ls, distance_only(t)))) return rc;
    if ((rc = ensure_attrs())) return rc;
    a.sx = dims[0];
    a.sy = dims[1];
    a.sz = dims[2];
    a.dy = make_dim(dims[1]);
    a.dz = make_dim(dims[2]);
    const uint32_t per_block = ls.block * ls.voxels_per_lane;
    a.chunks = (uint32_t)((cells + per_block - 1) / per_block);
    a.scratch_offset = (uint32_t)ls.regfile_bytes;
    a.boxes = 0;
    a.masks = nullptr;
    const dim3 block(ls.block);
    const bool d_only = distance_only(t);
    const uint32_t piece = units_per_launch(a.chunks, ls.block);
#define HU_LAUNCH_CLASSIFY(D, NV)                                                                                \
    hipLaunchKernelGGL((k_classify<InterpEval<D>, MASS, BATCH, NV>), grid, block, ls.lds, (hipStream_t)stream,     \
                       (InterpEval<D>{ls.prog, t->extra_dev, ls.n4}), a)
    for (uint32_t p0 = 0; p0 < n_parents; p0 += piece) {
        a.parent_base = p0;
        const dim3 grid(a.chunks * (n_parents - p0 < piece ? n_parents - p0 : piece));
        if (ls.voxels_per_lane == 2) {
            if (d_only) HU_LAUNCH_CLASSIFY(true, 2);
            else HU_LAUNCH_CLASSIFY(false, 2);
        } else {
            if (d_only) HU_LAUNCH_CLASSIFY(true, 1);
            else HU_LAUNCH_CLASSIFY(false, 1);
        }
    }
#undef HU_LAUNCH_CLASSIFY
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

int hu_subdivision_level_indirect(hu_tape t, const int32_t* parents_dev, const uint32_t* n_parents_dev, uint32_t max_parents,
                                  int32_t int_step, const uint32_t dims[3], int dimension, double resolution,
                                  const double origin[3], float step, float threshold, uint32_t* counter_dev,
                                  int32_t* children_dev, uint32_t capacity, void* stream)
{
    if (!t || !origin || !counter_dev || !n_parents_dev || (!children_dev && capacity) || (!parents_dev && max_parents))
        return fail(HU_ERR_BAD_ARG, "NULL argument");
    if (dimension != 2 && dimension != 3) return fail(HU_ERR_BAD_ARG, "dimension must be 2 or 3");
    ClassifyArgs a;
    std::memset(&a, 0, sizeof(a));
    a.parents = parents_dev;
    a.n_parents_dev = n_parents_dev;
    a.int_step = int_step; a.dimension = dimension;
    a.res = resolution; a.ox = origin[0]; a.oy = origin[1]; a.oz = origin[2];
    a.step = step; a.thr = threshold;
    a.counter = counter_dev; a.list = children_dev; a.capacity = capacity;
    return launch_classify<false, true>(t, a, max_parents, dims, stream);
}

// ... and with OWNERSHIP (kernels.hpp ClassifyArgs::own): the launch of a level that `world` ranks classify in full, each keeping
// the cells it owns
int hu_subdivision_level_owned(hu_tape t, const int32_t* parents_dev, const uint32_t* n_parents_dev, uint32_t max_parents,
                               int32_t int_step, const uint32_t dims[3], int dimension, double resolution,
                               const double origin[3], float step, float threshold, uint32_t* counter_dev,
                               int32_t* children_dev, uint32_t capacity, uint32_t world, uint32_t rank, void* stream)
{
    if (!t || !origin || !counter_dev || !n_parents_dev || (!children_dev && capacity) || (!parents_dev && max_parents))
        return fail(HU_ERR_BAD_ARG, "NULL argument");
    if (dimension != 2 && dimension != 3) return fail(HU_ERR_BAD_ARG, "dimension must be 2 or 3");
    if (world == 0 || rank >= world) return fail(HU_ERR_BAD_ARG, "rank must be below world");
    ClassifyArgs a;
    std::memset(&a, 0, sizeof(a));
    a.parents = parents_dev;
    a.n_parents_dev = n_parents_dev;
    a.int_step = int_step; a.dimension = dimension;
    a.res = resolution; a.ox = origin[0]; a.oy = origin[1]; a.oz = origin[2];
    a.step = step; a.thr = threshold;
    a.counter = counter_dev; a.list = children_dev; a.capacity = capacity;
    a.own = make_dim(world); a.own_rank = rank;
    return launch_classify<false, true>(t, a, max_parents, dims, stream);
}

int hu_mass_properties_level_owned(hu_tape t, const double* parents_dev, const uint32_t* n_parents_dev, uint32_t max_parents, double s,
                                   const uint32_t dims[3], float step, float threshold, uint32_t* sums_dev,
                                   uint32_t* counter_dev, double* children_dev, uint32_t capacity, uint32_t world, uint32_t rank, void* stream)
{
    if (!t || !sums_dev || !counter_dev || !n_parents_dev || (!children_dev && capacity) || (!parents_dev && max_parents))
        return fail(HU_ERR_BAD_ARG, "NULL argument");
    if (world == 0 || rank >= world) return fail(HU_ERR_BAD_ARG, "rank must be below world");
    ClassifyArgs a;
    std::memset(&a, 0, sizeof(a));
    a.parents = parents_dev;
    a.n_parents_dev = n_parents_dev;
    a.s = s;
    a.step = step; a.thr = threshold;
    a.counter = counter_dev; a.list = children_dev; a.capacity = capacity;
    a.sums = sums_dev;
    a.own = make_dim(world); a.own_rank = rank;
    return launch_classify<true, true>(t, a, max_parents, dims, stream);
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
                      double* out_dev, uint32_t rows, void* stream)
{
    if (!out_dev || ((!parents_dev || !sums_dev) && n_parents)) return fail(HU_ERR_BAD_ARG, "NULL argument");
    if (rows == 0 || rows > 65535u) return fail(HU_ERR_BAD_ARG, "rows must be in 1..65535");
    const uint32_t per_row = (n_parents + rows - 1) / rows;   // rows past the end get an empty slice and write zeros
    HU_HIP(hu_render::mass_integrals((const double4*)parents_dev, sums_dev, n_parents, per_row, nullptr, s, out_dev, rows, (hipStream_t)stream));
    return HU_OK;
}

int hu_mass_properties_level_indirect(hu_tape t, const double* parents_dev, const uint32_t* n_parents_dev, uint32_t max_parents, double s,
                                      const uint32_t dims[3], float step, float threshold, uint32_t* sums_dev,
                                      uint32_t* counter_dev, double* children_dev, uint32_t capacity, void* stream)
{
    if (!t || !sums_dev || !counter_dev || !n_parents_dev || (!children_dev && capacity) || (!parents_dev && max_parents))
        return fail(HU_ERR_BAD_ARG, "NULL argument");
    ClassifyArgs a;
    std::memset(&a, 0, sizeof(a));
    a.parents = parents_dev;
    a.n_parents_dev = n_parents_dev;
    a.s = s;
    a.step = step; a.thr = threshold;
    a.counter = counter_dev; a.list = children_dev; a.capacity = capacity;
    a.sums = sums_dev;
    return launch_classify<true, true>(t, a, max_parents, dims, stream);
}

int hu_mass_integrals_indirect(const double* parents_dev, const uint32_t* sums_dev, const uint32_t* n_parents_dev, uint32_t max_parents,
                               double s, double* out_dev, uint32_t rows, void* stream)
{
    if (!out_dev || !n_parents_dev || ((!parents_dev || !sums_dev) && max_parents)) return fail(HU_ERR_BAD_ARG, "NULL argument");
    if (rows == 0 || rows > 65535u) return fail(HU_ERR_BAD_ARG, "rows must be in 1..65535");
    HU_HIP(hu_render::mass_integrals((const double4*)parents_dev, sums_dev, max_parents, 0u, n_parents_dev, s, out_dev, rows, (hipStream_t)stream));
    return HU_OK;
}

int hu_ray_caster(hu_tape t, const float origin[4], const float forward[4], const float up[4], const float right[4],
                  float pixel_tolerance, float box_radius, float min_distance, float max_distance, float floor_z,
                  uint32_t render_options, uint32_t width, uint32_t height, void* out_dev, void* stream)
{
    if (!t || !origin || !forward || !up || !right || !out_dev) return fail(HU_ERR_BAD_ARG, "NULL argument");
    if (width == 0 || height == 0) return fail(HU_ERR_BAD_ARG, "image must have at least one pixel");
    if (render_options > 3u) return fail(HU_ERR_BAD_ARG, "unknown render option bits");
    const uint64_t tiles = (uint64_t)((width + 7u) / 8u) * ((height + 7u) / 8u);
    LaunchShape ls;
    int rc;
    if ((rc = launch_shape(t, ls, false, 1))) return rc;  // directions steer the march: full program
    if ((rc = ensure_attrs())) return rc;
    const uint32_t waves_per_block = ls.block / 64u;
    const uint64_t blocks = (tiles + waves_per_block - 1) / waves_per_block;
    if (blocks > 0x7fffffffull) return fail(HU_ERR_BAD_ARG, "image too large for one launch");
    RayCasterArgs a;
    a.origin = mk3(origin[0], origin[1], origin[2]);
    a.forward = mk3(forward[0], forward[1], forward[2]);
    a.up = mk3(up[0], up[1], up[2]);
    a.right = mk3(right[0], right[1], right[2]);
    a.pixel_tolerance = pixel_tolerance;
    a.box_radius = box_radius;
    a.min_distance = min_distance;
    a.max_distance = max_distance;
    a.floor_z = floor_z;
    a.options = render_options;
    a.w = width;
    a.h = height;
    a.out = static_cast<uint8_t*>(out_dev);
    if (t->spec && t->spec->ray_caster) {
        SpecEval ev{t->extra_dev, 0u};
        void* args[] = {&ev, &a};
        const uint64_t spec_blocks = (tiles + kSpecBlock / 64u - 1) / (kSpecBlock / 64u);
        HU_HIP(hipModuleLaunchKernel(t->spec->ray_caster, (uint32_t)spec_blocks, 1, 1, kSpecBlock, 1, 1, 0, (hipStream_t)stream,
                                     args, nullptr));
        return HU_OK;
    }
    HU_HIP(hu_render::ray_caster(InterpEval<false>{ls.prog, t->extra_dev, ls.n4}, a, (uint32_t)blocks, ls.block, ls.lds, (hipStream_t)stream));
    return HU_OK;
}

int hu_bitmap(hu_tape t, const float origin[4], float step_size, uint32_t width, uint32_t height, void* out_dev,
              void* stream)
{
    if (!t || !origin || !out_dev) return fail(HU_ERR_BAD_ARG, "NULL argument");
    if (width == 0 || height == 0) return fail(HU_ERR_BAD_ARG, "image must have at least one pixel");
    const uint64_t pixels = (uint64_t)width * height;
    if (pixels > 0x7fffffffull) return fail(HU_ERR_BAD_ARG, "image too large for one launch");
    uint8_t* out = static_cast<uint8_t*>(out_dev);
    if (t->spec && t->spec->bitmap) {
        SpecEval ev{t->extra_dev, 0u};
        float ox = origin[0], oy = origin[1], oz = origin[2];
        void* args[] = {&ev, &ox, &oy, &oz, &step_size, &width, &height, &out};
        HU_HIP(hipModuleLaunchKernel(t->spec->bitmap, (uint32_t)((pixels + kSpecBlock - 1) / kSpecBlock), 1, 1, kSpecBlock, 1, 1, 0,
                                     (hipStream_t)stream, args, nullptr));
        return HU_OK;
    }
    const bool d_only = distance_only(t);
    LaunchShape ls;
    int rc;
    if ((rc = launch_shape(t, ls, d_only, 1))) return rc;
    if ((rc = ensure_attrs())) return rc;
    HU_HIP(hu_render::bitmap(d_only, ls.prog, t->extra_dev, ls.n4, origin[0], origin[1], origin[2], step_size, width, height, out,
                             (uint32_t)((pixels + ls.block - 1) / ls.block), ls.block, ls.lds, (hipStream_t)stream));
    return HU_OK;
}

// ---- specialised code objects: hipRTC build + optional on-disk cache ------------------------
// What hu_tape_specialize needs from a build: the code object and, per kernel of kSpecKernelNames, its
// lowered (mangled) name.  With a cache directory the image is stored under a key made of everything the
// build depends on -- generated source, the op library headers it includes, the compiler options, the
// hipRTC / HIP versions -- so a later process (or a later tape with the same program) loads it in
// milliseconds instead of compiling for seconds.  The cache is best effort: unreadable, truncated or
// foreign files are ignored and rebuilt, an unwritable directory is not an error.
struct SpecImage {
    std::vector<std::string> lowered;
    std::vector<char> code;
};

static const char* const kSpecHeaders[] = {"kernels.hpp", "interp.hpp", "tape_format.hpp", "sdf_math.hpp"};
static const char kSpecMagic[8] = {'H', 'U', 'S', 'P', 'E', 'C', '1', 0};

static uint64_t fnv1a(uint64_t h, const void* data, size_t n)
{
    const unsigned char* p = static_cast<const unsigned char*>(data);
    for (size_t i = 0; i < n; ++i) h = (h ^ p[i]) * 0x100000001b3ull;
    return h;
}

static bool read_file(const std::string& path, std::string& out)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    out.clear();
    char buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    const bool ok = !std::ferror(f);
    std::fclose(f);
    return ok;
}

// A source above this size is built with -O1: what takes the time in a kernel of 200 KB is code generation, the straight-line
// code the generator writes leaves the optimiser little to do, and -O1 spends a third less on it for the same kernels
// (planetary, 855 KB of source, MI355X box: first per-tape launch 4.0 -> 2.6 s after upload, all kernels 4.4 -> 2.9 s; C4 0.383
// ms, its 256^3 grids 0.19 / 0.57 ms, C3 and C5 at -O1: all unchanged; the parity tests pass either way).  Small sources gain
// nothing (sponge(4), 80 KB: 0.26 s either way) and keep -O3.  HU_RTC_BIG_KB: the threshold in KiB (default 256, 0: never).
static bool spec_source_is_big(size_t bytes)
{
    static const size_t limit = [] { const char* e = getenv("HU_RTC_BIG_KB"); const long v = e ? atol(e) : 256; return (size_t)(v > 0 ? v : 0) * 1024u; }();
    return limit != 0 && bytes > limit;
}

static std::vector<std::string> spec_options(const char* include_dir, bool big = false)
{
    // same numerical contract as the ahead-of-time build: no contraction, IEEE sqrt/divide (HIP default)
    std::vector<std::string> opts = {"--offload-arch=gfx950", big ? "-O1" : "-O3", "-std=c++17", "-ffp-contract=off",
                                     std::string("-I") + include_dir};
    if (const char* e = getenv("HU_RTC_FLAGS")) {  // extra compiler options, for tuning experiments
        std::istringstream in(e);
        for (std::string w; in >> w;) opts.push_back(w);
    }
    return opts;
}

// Two independent 64-bit hashes over everything the build depends on; false if a header cannot be read
// (then nothing is cached).
static bool spec_cache_key(const std::string& src, const char* include_dir, const std::vector<std::string>& opts, uint32_t groups,
                           uint64_t key[2])
{
    uint64_t h[2] = {0xcbf29ce484222325ull, 0x84222325cbf29ce4ull};
    auto mix = [&](const void* p, size_t n) {
        const uint64_t len = n;
        for (int i = 0; i < 2; ++i) {
            h[i] = fnv1a(h[i], &len, sizeof len);
            h[i] = fnv1a(h[i], p, n);
        }
    };
    int version[3] = {0, 0, HIP_VERSION};
    (void)hiprtcVersion(&version[0], &version[1]);
    mix(version, sizeof version);
    mix(src.data(), src.size());
    for (size_t i = 0; i < opts.size(); ++i)  // the include path itself does not matter, the headers' bytes do
        if (opts[i].compare(0, 2, "-I") != 0) mix(opts[i].data(), opts[i].size());
    for (int i = 0; i < kSpecKernelCount; ++i)   // the kernels of this build: a build of other families is another file
        if (kSpecGroupOf[i] & groups) mix(kSpecKernelNames[i], std::strlen(kSpecKernelNames[i]));
    std::string text;
    for (const char* name : kSpecHeaders) {
        if (!read_file(std::string(include_dir) + "/" + name, text)) return false;
        mix(text.data(), text.size());
    }
    key[0] = h[0];
    key[1] = h[1] ^ 0x9e3779b97f4a7c15ull;
    return true;
}

static std::string spec_cache_path(const char* cache_dir, const uint64_t key[2])
{
    char name[64];
    std::snprintf(name, sizeof name, "/%016llx%016llx.huspec", (unsigned long long)key[0], (unsigned long long)key[1]);
    return std::string(cache_dir) + name;
}

static uint32_t spec_kernels_in(uint32_t groups)
{
    uint32_t n = 0;
    for (int i = 0; i < kSpecKernelCount; ++i) n += (kSpecGroupOf[i] & groups) ? 1u : 0u;
    return n;
}

static bool spec_cache_load(const std::string& path, const uint64_t key[2], uint32_t groups, SpecImage& img)
{
    std::string blob;
    if (!read_file(path, blob)) return false;
    size_t pos = 0;
    auto take = [&](void* dst, size_t n) {
        if (blob.size() - pos < n) return false;
        std::memcpy(dst, blob.data() + pos, n);
        pos += n;
        return true;
    };
    char magic[8];
    uint64_t k[2], code_size, sum;
    uint32_t names;
    if (!take(magic, 8) || std::memcmp(magic, kSpecMagic, 8) != 0 || !take(k, 16) || k[0] != key[0] || k[1] != key[1] ||
        !take(&names, 4) || names != spec_kernels_in(groups))
        return false;
    img.lowered.clear();
    for (uint32_t i = 0; i < names; ++i) {
        uint32_t len;
        if (!take(&len, 4) || len == 0 || len > 4096 || blob.size() - pos < len) return false;
        img.lowered.emplace_back(blob.data() + pos, len);
        pos += len;
    }
    if (!take(&code_size, 8) || code_size == 0 || blob.size() - pos != code_size + 8) return false;
    img.code.assign(blob.begin() + pos, blob.begin() + pos + code_size);
    pos += code_size;
    return take(&sum, 8) && sum == fnv1a(0xcbf29ce484222325ull, blob.data(), blob.size() - 8);  // covers names and code
}

static void spec_cache_store(const char* cache_dir, const std::string& path, const uint64_t key[2], const SpecImage& img)
{
    (void)mkdir(cache_dir, 0700);  // one level; the caller creates parents
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    std::string blob(kSpecMagic, 8);
    auto put = [&](const void* p, size_t n) { blob.append(static_cast<const char*>(p), n); };
    put(key, 16);
    const uint32_t names = (uint32_t)img.lowered.size();
    put(&names, 4);
    for (const std::string& n : img.lowered) {
        const uint32_t len = (uint32_t)n.size();
        put(&len, 4);
        put(n.data(), len);
    }
    const uint64_t code_size = img.code.size();
    put(&code_size, 8);
    put(img.code.data(), img.code.size());
    const uint64_t sum = fnv1a(0xcbf29ce484222325ull, blob.data(), blob.size());
    put(&sum, 8);
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return;
    bool ok = std::fwrite(blob.data(), 1, blob.size(), f) == blob.size();
    ok = (std::fclose(f) == 0) && ok;
    if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) (void)std::remove(tmp.c_str());  // atomic publish
}

// Keep the cache bounded: beyond kSpecCacheFiles entries the oldest (by modification time) are removed.
constexpr size_t kSpecCacheFiles = 8192;    // (up to nineteen per tape)
static void spec_cache_prune(const char* cache_dir)
{
    DIR* d = opendir(cache_dir);
    if (!d) return;
    std::vector<std::pair<int64_t, std::string>> files;
    while (const dirent* e = readdir(d)) {
        const std::string name = e->d_name;
        if (name.size() < 8 || name.compare(name.size() - 7, 7, ".huspec") != 0) continue;
        struct stat st;
        const std::string path = std::string(cache_dir) + "/" + name;
        if (stat(path.c_str(), &st) == 0) files.emplace_back((int64_t)st.st_mtime, path);
    }
    closedir(d);
    if (files.size() <= kSpecCacheFiles) return;
    std::sort(files.begin(), files.end());
    for (size_t i = 0; i + kSpecCacheFiles * 3 / 4 < files.size(); ++i) (void)std::remove(files[i].second.c_str());
}

// ---- a precompiled header for the per-tape builds --------------------------------------------------------------------
// A per-tape build parses the same ~16 000 lines every time -- hipRTC's own runtime header (13 000) and the op library
// (kernels.hpp and what it includes) -- before it sees the first line that depends on the tape: a quarter of a family's
// build, and most of a single small kernel's.  hipRTC hands its options to clang, `-include-pch` among them; what it cannot
// do is WRITE one.  So the header is made once per (cache directory, op library, hipRTC installation) by the clang++ that
// sits next to the hipRTC in use (<lib>/llvm/bin/clang++: same compiler, or the file is refused and the build goes on
// without -- as it does when there is no such clang, e.g. under the hipRTC a PyTorch wheel brings along), from hipRTC's
// runtime header (libhiprtc-builtins.so exports its text) and with the options hipRTC itself passes.  Best effort all the
// way: no clang, no builtins library, a directory that cannot be written, a header another process is just making, a file
// clang refuses -- the build runs as before.  HU_RTC_PCH=0 switches it off, HU_CLANG names the compiler.
static std::atomic<bool> g_pch_refused{false};          // the compiler in this process refused a header once: do not offer it again
static std::atomic<bool> g_pch_beside_refused{false};   // ... the one beside the library (then: one of its own, in the cache directory)
static std::mutex g_pch_mutex;                          // builds may run on several threads of a process (buffer.py, servers off)

static std::string dir_of(const std::string& path)
{
    const size_t cut = path.rfind('/');
    return cut == std::string::npos ? std::string(".") : path.substr(0, cut);
}

static bool run_and_wait(const std::vector<std::string>& argv)
{
    std::vector<char*> av;
    for (const std::string& a : argv) av.push_back(const_cast<char*>(a.c_str()));
    av.push_back(nullptr);
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_addopen(&fa, 0, "/dev/null", O_RDONLY, 0);
    posix_spawn_file_actions_addopen(&fa, 1, "/dev/null", O_WRONLY, 0);   // (a compile server talks on its stdout)
    posix_spawn_file_actions_addopen(&fa, 2, "/dev/null", O_WRONLY, 0);
    pid_t pid = 0;
    const int rc = posix_spawn(&pid, av[0], &fa, nullptr, av.data(), environ);
    posix_spawn_file_actions_destroy(&fa);
    if (rc != 0) return false;
    int status = 0;
    while (waitpid(pid, &status, 0) < 0)
        if (errno != EINTR) return false;
    return WIFEXITED(status) && WEXITSTATUS(status) == 0;
}

// -> the path of a usable precompiled header, or "" (then the build runs without one): the one the library's build left
// next to the library (<directory of libhip_util.so>/pch, builder.py), else the one in `dir` (NULL: none), made now if need be
static std::string spec_pch(const char* include_dir, const char* dir, const std::vector<std::string>& options, bool only_in_dir = false)
{
    static const bool off = [] { const char* e = getenv("HU_RTC_PCH"); return e && e[0] == '0'; }();
    if (off || g_pch_refused) return "";
    Dl_info where{};
    if (!dladdr(reinterpret_cast<const void*>(&hiprtcCompileProgram), &where) || !where.dli_fname) return "";
    const std::string lib_dir = dir_of(where.dli_fname);
    std::string clang;
    if (const char* e = getenv("HU_CLANG")) clang = e;
    else
        for (const char* rel : {"/llvm/bin/clang++", "/../llvm/bin/clang++", "/../lib/llvm/bin/clang++"})
            if (clang.empty() && access((lib_dir + rel).c_str(), X_OK) == 0) clang = lib_dir + rel;
    if (clang.empty() || access(clang.c_str(), X_OK) != 0) return "";
    // its name: everything it depends on
    uint64_t h = 0xcbf29ce484222325ull;
    int version[3] = {0, 0, HIP_VERSION};
    (void)hiprtcVersion(&version[0], &version[1]);
    h = fnv1a(h, version, sizeof version);
    h = fnv1a(h, lib_dir.data(), lib_dir.size());
    h = fnv1a(h, clang.data(), clang.size());
    for (const std::string& o : options)
        if (o.compare(0, 2, "-I") != 0) h = fnv1a(h, o.data(), o.size() + 1);   // (not the include path: the headers' bytes)
    std::string text;
    for (const char* name : kSpecHeaders) {
        if (!read_file(std::string(include_dir) + "/" + name, text)) return "";
        h = fnv1a(h, text.data(), text.size());
    }
    char hex[32];
    std::snprintf(hex, sizeof hex, "%016llx", (unsigned long long)h);
    if (!only_in_dir) {
        Dl_info self{};
        if (dladdr(reinterpret_cast<const void*>(&hu_last_error), &self) && self.dli_fname) {
            const std::string beside = dir_of(self.dli_fname) + "/pch/pch_" + hex + ".pch";
            if (access(beside.c_str(), R_OK) == 0) return beside;
        }
    }
    if (!dir || !*dir) return "";
    const std::string base = std::string(dir) + "/pch_" + hex, pch = base + ".pch";
    if (access(pch.c_str(), R_OK) == 0) return pch;
    std::lock_guard<std::mutex> one_at_a_time(g_pch_mutex);
    if (access(pch.c_str(), R_OK) == 0) return pch;      // (another thread made it meanwhile)
    static std::vector<std::string> tried;     // one attempt per process and name
    if (std::find(tried.begin(), tried.end(), base) != tried.end()) return "";
    tried.push_back(base);
    // one process makes it; the others carry on without it meanwhile (a lock left behind by a crash expires)
    const std::string lock = base + ".lock";
    (void)mkdir(dir, 0700);
    int fd = open(lock.c_str(), O_CREAT | O_EXCL | O_WRONLY, 0600);
    if (fd < 0) {
        struct stat st;
        if (stat(lock.c_str(), &st) == 0 && time(nullptr) - st.st_mtime > 120) (void)unlink(lock.c_str());
        return "";
    }
    close(fd);
    bool ok = false;
    do {
        // hipRTC's runtime header, the text its own builds start from
        void* builtins = nullptr;
        for (const std::string& name : {lib_dir + "/libhiprtc-builtins.so", std::string("libhiprtc-builtins.so." + std::to_string(version[0])),
                                        std::string("libhiprtc-builtins.so")})
            if (!builtins) builtins = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!builtins) break;
        const char* header = static_cast<const char*>(dlsym(builtins, "__hipRTC_header"));
        const unsigned* header_size = static_cast<const unsigned*>(dlsym(builtins, "__hipRTC_header_size"));
        if (!header || !header_size || *header_size == 0) break;
        size_t n = *header_size;
        while (n > 0 && header[n - 1] == 0) --n;
        const std::string inc = base + "_include";
        (void)mkdir(inc.c_str(), 0755);
        const std::string tmp_tag = ".tmp" + std::to_string((long)getpid());
        FILE* f = std::fopen((inc + "/hiprtc_runtime.h" + tmp_tag).c_str(), "wb");
        if (!f) break;
        const bool wrote = std::fwrite(header, 1, n, f) == n;
        if ((std::fclose(f) != 0) || !wrote || std::rename((inc + "/hiprtc_runtime.h" + tmp_tag).c_str(), (inc + "/hiprtc_runtime.h").c_str()) != 0) break;
        f = std::fopen((base + ".hip").c_str(), "wb");
        if (!f) break;
        std::fputs("#include \"kernels.hpp\"\n", f);
        if (std::fclose(f) != 0) break;
        // the options hipRTC passes for a HIP source (amd_comgr: COMPILE_SOURCE_TO_RELOCATABLE), then ours
        const std::string v = std::to_string(HIP_VERSION_MAJOR) + "." + std::to_string(HIP_VERSION_MINOR) + "." + std::to_string(HIP_VERSION_PATCH);
        std::vector<std::string> argv = {clang, "-c", "-fhip-emit-relocatable", "-mllvm", "-amdgpu-internalize-symbols", "-I", inc, "-O3", "-x", "hip",
                                         "--offload-device-only", "--hip-version=" + v, "-DHIP_VERSION_MAJOR=" + std::to_string(HIP_VERSION_MAJOR),
                                         "-DHIP_VERSION_MINOR=" + std::to_string(HIP_VERSION_MINOR), "-DHIP_VERSION_PATCH=" + std::to_string(HIP_VERSION_PATCH),
                                         "-Wno-gnu-line-marker", "-Wno-missing-prototypes", "-D__HIPCC_RTC__", "-nogpuinc", "-include", "hiprtc_runtime.h"};
        for (const std::string& o : options) argv.push_back(o);
        for (const char* o : {"-Xclang", "-emit-pch", "-Xclang", "-fno-pch-timestamp", "-o"}) argv.push_back(o);
        argv.push_back(pch + tmp_tag);
        argv.push_back(base + ".hip");
        if (!run_and_wait(argv)) { (void)std::remove((pch + tmp_tag).c_str()); break; }
        ok = std::rename((pch + tmp_tag).c_str(), pch.c_str()) == 0;
    } while (false);
    (void)unlink(lock.c_str());
    return ok ? pch : "";
}

// Compile `src` with hipRTC (needs no device) into an image.
static int compile_specialised(const std::string& src, const std::vector<std::string>& options, uint32_t groups, SpecImage& img)
{
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "tape_specialised.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
        return fail(HU_ERR_UNSUPPORTED, "hiprtcCreateProgram failed");
    for (int i = 0; i < kSpecKernelCount; ++i)
        if (kSpecGroupOf[i] & groups) (void)hiprtcAddNameExpression(prog, kSpecKernelNames[i]);
    std::vector<const char*> opts;
    for (const std::string& w : options) opts.push_back(w.c_str());
    const hiprtcResult rc = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
    if (rc != HIPRTC_SUCCESS) {
        size_t n = 0;
        std::string log;
        if (hiprtcGetProgramLogSize(prog, &n) == HIPRTC_SUCCESS && n > 1) {
            log.resize(n);
            (void)hiprtcGetProgramLog(prog, &log[0]);
        }
        (void)hiprtcDestroyProgram(&prog);
        return fail(HU_ERR_UNSUPPORTED, std::string("hipRTC compile failed: ") + hiprtcGetErrorString(rc) + "\n" + log.substr(0, 4000));
    }
    size_t size = 0;
    (void)hiprtcGetCodeSize(prog, &size);
    img.code.resize(size);
    (void)hiprtcGetCode(prog, img.code.data());
    img.lowered.clear();
    for (int i = 0; i < kSpecKernelCount; ++i) {
        if (!(kSpecGroupOf[i] & groups)) continue;
        const char* name = kSpecKernelNames[i];
        const char* lowered = nullptr;
        if (hiprtcGetLoweredName(prog, name, &lowered) != HIPRTC_SUCCESS || !lowered) {
            (void)hiprtcDestroyProgram(&prog);
            return fail(HU_ERR_UNSUPPORTED, std::string("kernel missing from the specialised module: ") + name);
        }
        img.lowered.emplace_back(lowered);
    }
    (void)hiprtcDestroyProgram(&prog);
    return HU_OK;
}

// The image of `src`: from the cache when it is there, else built (and stored).  With only_if_cached a miss
// leaves img.code empty and is not an error.
static int specialised_image(const std::string& src, const char* include_dir, const char* cache_dir, bool only_if_cached, uint32_t groups,
                             SpecImage& img, int* from_cache, bool replace_cached = false)
{
    if (from_cache) *from_cache = 0;
    img.code.clear();
    const std::vector<std::string> options = spec_options(include_dir, spec_source_is_big(src.size()));
    uint64_t key[2];
    std::string path;
    const bool cached = cache_dir && *cache_dir && spec_cache_key(src, include_dir, options, groups, key);
    if (cached) {
        path = spec_cache_path(cache_dir, key);
        if (!replace_cached && spec_cache_load(path, key, groups, img)) {
            if (from_cache) *from_cache = 1;
            return HU_OK;
        }
        img.code.clear();
    }
    if (only_if_cached) return HU_OK;
    int rc = HU_ERR_UNSUPPORTED;
    for (int attempt = 0; attempt < 2 && rc != HU_OK; ++attempt) {
        const std::string pch = spec_pch(include_dir, cache_dir, options, g_pch_beside_refused);
        if (pch.empty()) break;
        std::vector<std::string> with = options;
        with.push_back("-include-pch");
        with.push_back(pch);
        if ((rc = compile_specialised(src, with, groups, img))) {
            // (whatever it was: the plain build below tells.)  The header beside the library may have been made under other
            // paths (a copied installation): then this process makes its own in the cache directory; one of the cache
            // directory that this compiler refuses goes, so that the next process makes a new one.
            const bool in_dir = cache_dir && *cache_dir && pch.compare(0, std::strlen(cache_dir), cache_dir) == 0;
            if (in_dir) {
                g_pch_refused = true;
                (void)std::remove(pch.c_str());
            } else {
                g_pch_beside_refused = true;
            }
        }
    }
    if (rc != HU_OK && (rc = compile_specialised(src, options, groups, img))) return rc;
    if (cached) {
        spec_cache_store(cache_dir, path, key, img);
        spec_cache_prune(cache_dir);
    }
    return HU_OK;
}

int hu_tape_compile_cached(const float* tape, size_t n, const char* include_dir, const char* cache_dir, size_t* code_bytes,
                           int* from_cache)
{
    return hu_tape_compile_groups(tape, n, include_dir, cache_dir, HU_SPEC_ALL, code_bytes, from_cache);
}

int hu_tape_compile_groups(const float* tape, size_t n, const char* include_dir, const char* cache_dir, uint32_t groups,
                           size_t* code_bytes, int* from_cache)
{
    if (!tape || !include_dir) return fail(HU_ERR_BAD_ARG, "NULL argument");
    if (groups == 0 || (groups & ~(uint32_t)HU_SPEC_ALL)) return fail(HU_ERR_BAD_ARG, "groups must be a non-empty set of HU_SPEC_* bits");
    sdf::DecodedTape d;
    const std::string err = sdf::decode_tape(tape, n, d);
    if (!err.empty()) return fail(HU_ERR_BAD_TAPE, "malformed tape: " + err);
    hu_tape_s t;  // host fields only
    t.n_slots = d.n_slots;
    keep_programs(&t, d);
    SpecImage img;
    int rc;
    if ((rc = specialised_image(generate_source(&t), include_dir, cache_dir, false, groups, img, from_cache))) return rc;
    if (code_bytes) *code_bytes = img.code.size();
    return HU_OK;
}

int hu_spec_pch_prepare(const char* include_dir, const char* dir, char* path, size_t capacity)
{
    if (!include_dir || !dir) return fail(HU_ERR_BAD_ARG, "NULL argument");
    // one per set of options the builds use: small sources (-O3) and big ones (-O1): clang refuses a header made at another level
    const std::string pch = spec_pch(include_dir, dir, spec_options(include_dir, false), true);
    const std::string pch_big = spec_pch(include_dir, dir, spec_options(include_dir, true), true);
    if (path && capacity) std::snprintf(path, capacity, "%s%s%s", pch.c_str(), (pch.empty() || pch_big.empty()) ? "" : "\n", pch_big.c_str());
    return HU_OK;
}

int hu_tape_compile_check(const float* tape, size_t n, const char* include_dir, size_t* code_bytes)
{
    return hu_tape_compile_cached(tape, n, include_dir, nullptr, code_bytes, nullptr);
}

int hu_tape_specialize_cached(hu_tape t, const char* include_dir, const char* cache_dir, int only_if_cached, int* from_cache)
{
    return hu_tape_specialize_groups(t, include_dir, cache_dir, only_if_cached, HU_SPEC_ALL, from_cache);
}

// Load `img` (the kernels of `set`) into the tape: those of them that are still missing take their slots.
static int load_specialised(hu_tape t, const SpecImage& img, uint32_t set, hipError_t* why)
{
    hipModule_t module = nullptr;
    hipFunction_t loaded[kSpecKernelCount] = {};
    hipError_t e = hipModuleLoadData(&module, img.code.data());
    size_t next = 0;
    for (int i = 0; i < kSpecKernelCount && e == hipSuccess; ++i)
        if (kSpecGroupOf[i] & set) e = (next < img.lowered.size()) ? hipModuleGetFunction(&loaded[i], module, img.lowered[next++].c_str()) : hipErrorNotFound;
    if (e != hipSuccess) {
        if (module) (void)hipModuleUnload(module);
        (void)hipGetLastError();  // the failed load must not surface at the next launch's error check
        if (why) *why = e;
        return HU_ERR_HIP;
    }
    if (!t->spec) t->spec = new SpecKernels();
    SpecKernels* k = t->spec;
    hipFunction_t* slots[kSpecKernelCount] = {&k->dense[0], &k->dense[1], &k->blocks[0], &k->blocks[1],
                                              &k->classify[0][0], &k->classify[0][1], &k->classify[1][0], &k->classify[1][1],
                                              &k->ray_caster, &k->bitmap, &k->box_masks,
                                              &k->dense_ragged[0], &k->dense_ragged[1], &k->blocks_ragged[0], &k->blocks_ragged[1],
                                              &k->dense_runs[0], &k->dense_runs[1], &k->blocks_runs[0], &k->blocks_runs[1]};
    const uint32_t missing = set & ~k->groups;
    for (int i = 0; i < kSpecKernelCount; ++i)
        if (kSpecGroupOf[i] & missing) *slots[i] = loaded[i];
    k->modules.push_back(module);
    k->groups |= missing;
    const sdf::SpecMeta& meta = t->spec_meta;
    k->deferred = meta.deferred;
    k->coord_limit = meta.coord_limit;
    k->prune_words = meta.prune_words;
    k->prune_bits = meta.prune_bits;
    k->prune_all = meta.prune_all;
    std::memcpy(k->tabs, meta.tabs, sizeof k->tabs);
    return HU_OK;
}

int hu_tape_specialize_groups(hu_tape t, const char* include_dir, const char* cache_dir, int only_if_cached, uint32_t groups, int* from_cache)
{
    if (from_cache) *from_cache = 0;
    if (!t || !include_dir) return fail(HU_ERR_BAD_ARG, "NULL argument");
    if (groups & ~(uint32_t)HU_SPEC_ALL) return fail(HU_ERR_BAD_ARG, "groups must be a set of HU_SPEC_* bits");
    auto missing = [&] { return t->spec ? (groups & ~t->spec->groups) : groups; };   // kernels that are loaded stay as they are
    if (missing() == 0) return HU_OK;
    if (t->spec_source.empty()) t->spec_source = generate_source(t, &t->spec_meta);
    const std::string& src = t->spec_source;
    const bool cache = cache_dir && *cache_dir;
    bool all_cached = true;
    // 1. the image of exactly this set (what a synchronous build of it left in the cache), 2. the images of its single
    // kernels (what the background builds leave), both only read; 3. what is still missing, built as one image
    for (int step = 0; step < 3 && missing(); ++step) {
        if (step < 2 && !cache) continue;
        if (step == 2 && only_if_cached) break;
        std::vector<uint32_t> sets;
        if (step == 1) {
            for (int i = 0; i < kSpecKernelCount; ++i)
                if ((kSpecGroupOf[i] & missing()) && kSpecGroupOf[i] != groups) sets.push_back(kSpecGroupOf[i]);
        } else {
            sets.push_back(step == 0 ? groups : missing());
        }
        for (uint32_t set : sets) {
            for (int attempt = 0; attempt < 2; ++attempt) {
                SpecImage img;
                int crc, cached = 0;
                // second attempt (step 3 only): the cached image did not load (e.g. written by an incompatible runtime): build and replace it
                if ((crc = specialised_image(src, include_dir, cache_dir, step < 2, set, img, &cached, attempt != 0))) return crc;
                if (img.code.empty()) break;  // not cached: still interpreted
                hipError_t e = hipSuccess;
                if (load_specialised(t, img, set, &e) == HU_OK) {
                    all_cached = all_cached && cached;
                    break;
                }
                if (step < 2) break;       // an unusable cached image is not the caller's problem
                if (!cached || attempt == 1) return fail(HU_ERR_HIP, std::string("loading the specialised module: ") + hipGetErrorString(e));
            }
        }
    }
    if (from_cache) *from_cache = (all_cached && missing() == 0) ? 1 : 0;
    return HU_OK;
}

int hu_tape_specialize(hu_tape t, const char* include_dir) { return hu_tape_specialize_cached(t, include_dir, nullptr, 0, nullptr); }

static int launch_process_polygon(bool batch, PolygonArgs& a, uint32_t n_blocks, void* stream)
{
    if (a.gx < 2 || a.gy < 2) return fail(HU_ERR_BAD_ARG, "the corner grid needs at least 2x2 samples");
    if (a.gx > 512 || a.gy > 512) return fail(HU_ERR_BAD_ARG, "corner grids above 512 overflow the link encoding (polygon2d.py:46)");
    if (n_blocks == 0) return HU_OK;
    if (n_blocks > 65535u) return fail(HU_ERR_BAD_ARG, "at most 65535 blocks per launch");
    const uint32_t cells = (a.gx - 1u) * (a.gy - 1u) * 2u;
    HU_HIP(hu_render::process_polygon(batch, a, dim3((cells + 255u) / 256u, n_blocks), (hipStream_t)stream));
    return HU_OK;
}

int hu_process_polygon(const float box_corner[2], float box_step, const void* corners_dev, const uint32_t grid[2],
                       void* vertices_dev, uint32_t* links_dev, uint32_t* starts_dev, uint32_t* start_counter_dev,
                       void* stream)
{
    if (!box_corner || !corners_dev || !grid || !vertices_dev || !links_dev || !starts_dev || !start_counter_dev)
        return fail(HU_ERR_BAD_ARG, "NULL argument");
    PolygonArgs a{};
    a.corners = static_cast<const float4*>(corners_dev);
    a.gx = grid[0] + 1u;  // the reference launches over (gx-1, gy-1, 2) triangles
    a.gy = grid[1] + 1u;
    a.cx = box_corner[0];
    a.cy = box_corner[1];
    a.step = box_step;
    a.vertices = static_cast<float2*>(vertices_dev);
    a.links = links_dev;
    a.starts = starts_dev;
    a.start_counter = start_counter_dev;
    return launch_process_polygon(false, a, 1, stream);
}

int hu_process_polygon_blocks(const void* corners_dev, const int32_t* blocks_dev, uint32_t n_blocks, double resolution,
                              const double origin[3], float step, const uint32_t dims[2], void* vertices_dev,
                              uint32_t* links_dev, uint32_t* starts_dev, uint32_t* start_counters_dev, void* stream)
{
    if (!origin || !dims) return fail(HU_ERR_BAD_ARG, "NULL argument");
    if (n_blocks && (!corners_dev || !blocks_dev || !vertices_dev || !links_dev || !starts_dev || !start_counters_dev))
        return fail(HU_ERR_BAD_ARG, "NULL argument");
    PolygonArgs a{};
    a.corners = static_cast<const float4*>(corners_dev);
    a.gx = dims[0];
    a.gy = dims[1];
    a.step = step;
    a.blocks = reinterpret_cast<const int4*>(blocks_dev);
    a.res = resolution;
    a.ox = origin[0];
    a.oy = origin[1];
    a.vertices = static_cast<float2*>(vertices_dev);
    a.links = links_dev;
    a.starts = starts_dev;
    a.start_counter = start_counters_dev;
    return launch_process_polygon(true, a, n_blocks, stream);
}

int hu_selftest_math(uint64_t counts[4])
{
    if (!counts) return fail(HU_ERR_BAD_ARG, "counts is NULL");
    unsigned long long* dev = nullptr;
    HU_HIP(hipMalloc((void**)&dev, 4 * sizeof(unsigned long long)));
    hipError_t e = hipMemset(dev, 0, 4 * sizeof(unsigned long long));
    if (e == hipSuccess) {
        e = hu_render::selftest_math(dev);
    }
    unsigned long long host[4] = {0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpy(host, dev, sizeof host, hipMemcpyDeviceToHost);
    (void)hipFree(dev);
    if (e != hipSuccess) return fail(HU_ERR_HIP, std::string("hu_selftest_math: ") + hipGetErrorString(e));
    for (int i = 0; i < 4; ++i) counts[i] = host[i];
    return HU_OK;
}

int hu_selftest_minmax3(uint64_t counts[3])
{
    if (!counts) return fail(HU_ERR_BAD_ARG, "counts is NULL");
    unsigned long long* dev = nullptr;
    HU_HIP(hipMalloc((void**)&dev, 3 * sizeof(unsigned long long)));
    hipError_t e = hipMemset(dev, 0, 3 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hu_render::selftest_minmax3(dev);
    unsigned long long host[3] = {0, 0, 0};
    if (e == hipSuccess) e = hipMemcpy(host, dev, sizeof host, hipMemcpyDeviceToHost);
    (void)hipFree(dev);
    if (e != hipSuccess) return fail(HU_ERR_HIP, std::string("hu_selftest_minmax3: ") + hipGetErrorString(e));
    for (int i = 0; i < 3; ++i) counts[i] = host[i];
    return HU_OK;
}

int hu_tape_source(const float* tape, size_t n, char* buf, size_t capacity, size_t* needed)
{
    if (!tape || !needed || (!buf && capacity)) return fail(HU_ERR_BAD_ARG, "NULL argument");
    sdf::DecodedTape d;
    const std::string err = sdf::decode_tape(tape, n, d);
    if (!err.empty()) return fail(HU_ERR_BAD_TAPE, "malformed tape: " + err);
    hu_tape_s t;  // host fields only: nothing touches a device
    t.n_slots = d.n_slots;
    keep_programs(&t, d);
    const std::string src = generate_source(&t);
    *needed = src.size() + 1;
    if (capacity >= src.size() + 1) std::memcpy(buf, src.c_str(), src.size() + 1);
    return HU_OK;
}

int hu_tape_listing(const float* tape, size_t n, int which, char* buf, size_t capacity, size_t* needed)
{
    if (!tape || !needed || (!buf && capacity)) return fail(HU_ERR_BAD_ARG, "NULL argument");
    if (which < 0 || which > 3) return fail(HU_ERR_BAD_ARG, "which must be 0..3");
    sdf::DecodedTape d;
    const std::string err = sdf::decode_tape(tape, n, d);
    if (!err.empty()) return fail(HU_ERR_BAD_TAPE, "malformed tape: " + err);
    const std::vector<Rec>& prog = which == 0 ? d.recs : which == 1 ? d.recs_do : which == 2 ? d.fused : d.fused_do;
    static const char* const internal[] = {"FROM_SCALE", "FROM_X", "FROM_Y", "FROM_Z", "POINT", "TO_SCALE", "TO_X", "TO_Y", "TO_Z",
                                           "TO_ROW_X", "TO_ROWS_YZ", "FROM_MATRIX", "INIT_ROW_X", "INIT_ROWS_YZ", "LEAF"};
    static const char* const kinds[] = {"-", "scale", "x", "y", "z"};
    static const char* const prims[] = {"rectangle", "circle", "sphere", "half_space"};
    static const char* const combs[] = {"", "union", "intersection", "subtraction"};
    std::ostringstream o;
    for (const Rec& r : prog) {
        const uint32_t op = r.hdr & 0xffu, slot = (r.hdr >> 8) & 0xffffu;
        uint32_t fold;
        std::memcpy(&fold, &r.p[sdf::kFoldParam], 4);
        if (fold & sdf::kFoldLoad) o << "[load " << (fold & 0xffu) << ((fold & sdf::kFoldLoadResult) ? "r" : "") << "] ";
        o << (op < sdf::OP_COUNT ? sdf::op_info(op).name : internal[op - sdf::OP_COUNT]);
        if (op == sdf::OPX_LEAF) {
            uint32_t c;
            std::memcpy(&c, &r.p[sdf::kLeafControl], 4);
            o << "(" << ((c & sdf::kLeafSample) ? "sample " : "") << "to:" << kinds[(c >> sdf::kLeafToShift) & 7u]
              << ((c & sdf::kLeafMidStore) ? " store-point:" + std::to_string(slot) : std::string()) << " "
              << prims[(c >> sdf::kLeafPrimShift) & 3u] << ((c & sdf::kLeafExtrusion) ? " extrusion" : "");
            if (!(c & sdf::kLeafFromLast)) o << " from:" << kinds[(c >> sdf::kLeafFromShift) & 7u];
            for (int k = 0; k < 2; ++k) {
                const uint32_t cb = c >> (k == 0 ? sdf::kLeafComb1Shift : sdf::kLeafComb2Shift);
                if (cb & 3u) o << " " << combs[cb & 3u] << ":" << ((cb >> 2) & 0xffu);
            }
            if (c & sdf::kLeafFromLast) o << " then-from:scale";
            o << ")";
        } else if (sdf::rec_arity(op) == 2 || op == sdf::OP_STORE || op == sdf::OP_LOAD) {
            o << " " << slot << ((r.hdr & sdf::kResultKind) ? "r" : "");
        }
        if (fold & sdf::kFoldStore) o << " [store " << ((fold >> 16) & 0xffu) << ((fold & sdf::kFoldStoreResult) ? "r" : "") << "]";
        o << "\n";
        if (op == sdf::OP_RETURN) break;
    }
    const std::string text = o.str();
    *needed = text.size() + 1;
    if (capacity >= text.size() + 1) std::memcpy(buf, text.c_str(), text.size() + 1);
    return HU_OK;
}

int hu_tape_prune_info(hu_tape t, int* bits, int* words)
{
    if (!t) return fail(HU_ERR_BAD_ARG, "tape is NULL");
    if (bits) *bits = t->spec ? t->spec->prune_bits : 0;
    if (words) *words = t->spec ? t->spec->prune_words : 0;
    return HU_OK;
}

int hu_tape_specialized(hu_tape t, int* out)
{
    if (!t || !out) return fail(HU_ERR_BAD_ARG, "NULL argument");
    *out = t->spec ? (int)t->spec->groups : 0;   // the HU_SPEC_* families that are loaded (0: interpreted)
    return HU_OK;
}

}  // extern "C"
