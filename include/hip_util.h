/* include/hip_util.h -- C ABI of libhip_util.so, the MI355X (gfx950) replacement for the
 * device half of bluecube/codecad's hot path.
 *
 * What it replaces (paths relative to /root/reference/codecad/):
 *   - cl_util/opencl_manager.py:73-85   `opencl_manager.k.<kernel>(global, local, *args)`
 *   - cl_util/cl_buffer.py:9-131        `Buffer` (device allocation + host transfers)
 *   - nodes/program.py:79-84            `make_program_buffer` (tape upload)
 *   - grid_eval.cl, subdivision.cl, mass_properties.cl and the generated evaluate()
 *   - rendering/ray_caster.cl, bitmap.cl, polygon2d.cl, and the PyMCubes call of rendering/mesh.py
 *
 * Conventions: every function returns 0 on success or a negative hu_status code; the
 * message for the last failure on the calling thread is hu_last_error().  All pointers
 * named *_dev are device pointers (from hu_malloc or any hipMalloc-compatible allocator,
 * e.g. a torch tensor's data_ptr()).  `stream` is a hipStream_t passed as void*
 * (NULL = the legacy default stream).  Launch functions are asynchronous and perform no
 * allocation or synchronisation, so they can be captured into a hipGraph (exceptions, each
 * documented at its declaration: hu_tape_create/specialize, hu_sort_blocks and hu_selftest_math
 * synchronise).  The caller owns every handle; there are no hidden global allocations.
 */
#ifndef HIP_UTIL_H
#define HIP_UTIL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HU_ABI_VERSION 1

enum hu_status {
    HU_OK = 0,
    HU_ERR_HIP = -1,        /* a HIP runtime call failed (message has hipGetErrorString) */
    HU_ERR_BAD_TAPE = -2,   /* malformed instruction tape */
    HU_ERR_BAD_ARG = -3,    /* NULL pointer, zero/oversized dims, ... */
    HU_ERR_NO_DEVICE = -4,  /* no gfx950 device / no HIP runtime */
    HU_ERR_UNSUPPORTED = -5 /* e.g. tape needs more value registers than fit in LDS */
};

typedef struct hu_tape_s* hu_tape; /* opaque: decoded program resident in HBM */

/* ---- runtime (replaces OpenCLManager.__init__, opencl_manager.py:88-98) ---------------- */
int hu_abi_version(void);
const char* hu_last_error(void);
int hu_device_count(int* count);
int hu_set_device(int ordinal);
int hu_device_name(int ordinal, char* buf, size_t buflen);
int hu_synchronize(void);

/* ---- memory (replaces cl_util.Buffer / pyopencl.enqueue_copy, cl_buffer.py:31-94) ------ */
int hu_malloc(void** out_dev, size_t bytes);
int hu_free(void* dev); /* NULL is a no-op */
int hu_host_alloc(void** out_host, size_t bytes); /* pinned host memory */
int hu_host_free(void* host);
int hu_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, void* stream);
int hu_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream);
int hu_memcpy_d2d(void* dst_dev, const void* src_dev, size_t bytes, void* stream);
int hu_memset(void* dst_dev, int value, size_t bytes, void* stream);

/* ---- streams and events (replaces the OOO queue + pyopencl.Event, wait_for=) ----------- */
int hu_stream_create(void** out_stream);
int hu_stream_destroy(void* stream);
int hu_stream_synchronize(void* stream);
int hu_stream_wait_event(void* stream, void* event);
int hu_event_create(void** out_event);
int hu_event_destroy(void* event);
int hu_event_record(void* event, void* stream);
int hu_event_synchronize(void* event);
int hu_event_elapsed_ms(void* start, void* stop, float* out_ms);

/* ---- tape (replaces nodes.make_program_buffer, nodes/program.py:79-84) ------------------ */
/* `tape`: the reference float32 instruction tape (opcode*512+register words + params,
 * nodes/program.py:55-71).  Validated, pre-decoded and uploaded; synchronous. */
int hu_tape_create(const float* tape, size_t n_floats, hu_tape* out);
int hu_tape_destroy(hu_tape t);
/* n_instructions, value registers used, flags bit0 = a rounded union/intersection is present */
int hu_tape_info(hu_tape t, int* n_instructions, int* n_registers, int* flags);

/* ---- reference-shaped kernels: same arguments as the OpenCL kernels --------------------- */
/* grid_eval.cl:23-25  out_dev: float4[dims[0]*dims[1]*dims[2]], index z + sz*(y + sy*x) */
int hu_grid_eval(hu_tape t, const float corner[4], float step, const uint32_t dims[3],
                 void* out_dev, void* stream);
/* grid_eval.cl:2-4    out_dev: float[...], index z + (x + (sy-1-y)*sx)*sz */
int hu_grid_eval_pymcubes(hu_tape t, const float corner[4], float step, const uint32_t dims[3],
                          void* out_dev, void* stream);
/* subdivision.cl:12-16  counter_dev: uint32 (caller zeroes it); list_dev: uchar4[cells] */
int hu_subdivision_step(hu_tape t, const float corner[4], float step, float threshold,
                        const uint32_t dims[3], uint32_t* counter_dev, void* list_dev,
                        void* stream);
/* mass_properties.cl:7-12  sum_dev: uint32[10] xx,xy,xz,x,yy,yz,y,zz,z,n (caller zeroes) */
int hu_mass_properties(hu_tape t, const float corner[4], float step, float threshold,
                       const uint32_t dims[3], uint32_t* sum_dev, uint32_t* counter_dev,
                       void* list_dev, void* stream);

/* ---- level-batched kernels (one launch per subdivision LEVEL instead of one per block) -- */
/* Dense grid sharded along x: evaluates x in [x0, x0+x_count) of the logical grid `dims`
 * and writes them at out_dev (which points at the FIRST voxel of the slab).  layout 0 =
 * float4 grid_eval, 1 = float grid_eval_pymcubes (then out_dev is the whole-grid base). */
int hu_grid_eval_slab(hu_tape t, const float corner[4], float step, const uint32_t dims[3],
                      uint32_t x0, uint32_t x_count, int layout, void* out_dev, void* stream);

/* Leaf blocks of subdivision() (subdivision.py:96-111): blocks_dev = int32[4]*n_blocks
 * integer corners; float corner = int*resolution + origin in fp64, cast once
 * (util/geometry.py:98-99).  out_dev: n_blocks consecutive grids of `dims`. */
int hu_grid_eval_blocks(hu_tape t, const int32_t* blocks_dev, uint32_t n_blocks,
                        double resolution, const double origin[3], float step,
                        const uint32_t dims[3], int layout, void* out_dev, void* stream);

/* One level of subdivision() for ALL parents at once (subdivision.py:48-113).
 * parents_dev: int32[4]*n_parents integer box corners; per parent the sample corner is
 * (int_corner + int_step/2)*resolution + origin in fp64 (z unshifted when dimension==2),
 * threshold = step*sqrt(dimension)/2 is computed by the caller.  Ambiguous cells are
 * appended (wavefront ballot scan, one atomic per workgroup) to children_dev as
 * int32[4] = parent + (x,y,z)*int_step, ready to be the next level's parents; the 4th
 * component is a caller tag (e.g. an object id) copied from parent to child untouched.
 * counter_dev (uint32, caller zeroes) ends up as the number of ambiguous cells even when it
 * exceeds `capacity` (extra cells are dropped: the caller re-runs with a larger list). */
int hu_subdivision_level(hu_tape t, const int32_t* parents_dev, uint32_t n_parents,
                         int32_t int_step, const uint32_t dims[3], int dimension,
                         double resolution, const double origin[3], float step, float threshold,
                         uint32_t* counter_dev, int32_t* children_dev, uint32_t capacity,
                         void* stream);

/* The same with the list lengths ON THE DEVICE, so that a whole traversal can be enqueued without a host
 * round trip between its levels: the launch covers max_parents (the list's capacity) and workgroups past
 * *n_parents_dev leave at once.  Typical chaining: counter_dev = word 0 of a [header row | rows...] buffer and
 * children_dev = its row 1, which makes the buffer the next level's (parents_dev = row 1, n_parents_dev = word
 * 0) and, on several GPUs, the fixed-size piece of the all-gather (hu_slice_rows).  The caller checks
 * counter <= capacity once, at the end of the traversal. */
int hu_subdivision_level_indirect(hu_tape t, const int32_t* parents_dev, const uint32_t* n_parents_dev,
                                  uint32_t max_parents, int32_t int_step, const uint32_t dims[3],
                                  int dimension, double resolution, const double origin[3], float step,
                                  float threshold, uint32_t* counter_dev, int32_t* children_dev,
                                  uint32_t capacity, void* stream);
int hu_grid_eval_blocks_indirect(hu_tape t, const int32_t* blocks_dev, const uint32_t* n_blocks_dev,
                                 uint32_t max_blocks, double resolution, const double origin[3], float step,
                                 const uint32_t dims[3], int layout, void* out_dev, void* stream);
/* Multi-GPU level exchange (the reference has one device and no counterpart).  After
 * an all-gather of fixed-size pieces gathered_dev holds world x piece_rows rows of row_bytes (16 or 32) each;
 * row 0 of a piece is its header (word 0 = number of rows that follow).  Writes this rank's balanced share of
 * the concatenated rows -- begin = rank*base + min(rank, extra), base/extra = divmod(total, world) -- to
 * out_dev as [header | rows] (out_capacity rows after the header).  stats_dev: uint32[2] <- {rows over all
 * ranks, 1 if a piece or the share was truncated}.  Asynchronous, no host involvement. */
int hu_slice_rows(const void* gathered_dev, uint32_t world, uint32_t piece_rows, uint32_t row_bytes,
                  uint32_t rank, void* out_dev, uint32_t out_capacity, uint32_t* stats_dev, void* stream);
/* The same for ONE piece that every rank holds identically -- a level small enough that each rank classified ALL of it
 * itself, in one workgroup per parent, whose compaction order is the lane order and hence the same everywhere (the top
 * level of a hierarchy: one parent, a few cells) -- shared out among `world` ranks without any collective. */
int hu_slice_rows_of(const void* piece_dev, uint32_t piece_rows, uint32_t row_bytes, uint32_t rank, uint32_t world,
                     void* out_dev, uint32_t out_capacity, uint32_t* stats_dev, void* stream);

/* One level of mass_properties() for ALL parents (mass_properties.py:69-157).
 * parents_dev: double[4]*n_parents box corners; sample corner = corner + s/2 (fp64), cast
 * once.  sums_dev: uint32[10]*n_parents (caller zeroes), same order as hu_mass_properties.
 * children_dev: double[4] = (i,j,k)*s + corner (fp64), 4th component = the parent's tag,
 * appended like above. */
int hu_mass_properties_level(hu_tape t, const double* parents_dev, uint32_t n_parents, double s,
                             const uint32_t dims[3], float step, float threshold,
                             uint32_t* sums_dev, uint32_t* counter_dev, double* children_dev,
                             uint32_t capacity, void* stream);

/* The ten integrals (1, x, y, z, xx, yy, zz, xy, xz, yz over the inside cells) of one level from
 * the per-parent index sums, with the per-block formulas of mass_properties.py:119-148 in fp64,
 * summed deterministically on the device: out_dev: double[rows][10], row g = the integrals of the g-th
 * of `rows` contiguous slices of the parents (Kahan per thread, fixed tree per workgroup); the caller adds
 * the rows in order.  rows = 1 gives the level's integrals directly; more rows spread a long level over
 * the chip (1 row per ~2048 parents is plenty). */
int hu_mass_integrals(const double* parents_dev, const uint32_t* sums_dev, uint32_t n_parents, double s,
                      double* out_dev, uint32_t rows, void* stream);

/* The two above with the number of parents ON THE DEVICE (the launches cover max_parents, the list's capacity), so
 * that mass_properties() enqueues all its levels without a host round trip, like hu_subdivision_level_indirect:
 * counter_dev = word 0 of the children's [header row | rows...] buffer (32-byte rows), children_dev its row 1; the
 * caller zeroes sums_dev for max_parents parents and checks counter <= capacity once, at the end.  The integrals'
 * rows are cut from the actual count: the same slices as hu_mass_integrals(n_parents = the count). */
int hu_mass_properties_level_indirect(hu_tape t, const double* parents_dev, const uint32_t* n_parents_dev,
                                      uint32_t max_parents, double s, const uint32_t dims[3], float step,
                                      float threshold, uint32_t* sums_dev, uint32_t* counter_dev,
                                      double* children_dev, uint32_t capacity, void* stream);
/* Levels that SEVERAL ranks classify in full (multi-GPU, codecad_amd/dist.py "replicated levels"; the reference has one
 * device, cl_util/opencl_manager.py:89-98): like the *_indirect forms, but of every parent's cells a rank lists -- and sums --
 * only those it OWNS: owner = mix(hash of the parent's row, the cell's linear index z + sz * (y + sy * x)) mod world.  The ranks'
 * lists then partition the level's survivors (their moment sums add up to the level's) without any exchange, whatever order
 * each rank's parents were in.  world = 1: the *_indirect forms. */
int hu_subdivision_level_owned(hu_tape t, const int32_t* parents_dev, const uint32_t* n_parents_dev, uint32_t max_parents,
                               int32_t int_step, const uint32_t dims[3], int dimension, double resolution,
                               const double origin[3], float step, float threshold, uint32_t* counter_dev,
                               int32_t* children_dev, uint32_t capacity, uint32_t world, uint32_t rank, void* stream);
int hu_mass_properties_level_owned(hu_tape t, const double* parents_dev, const uint32_t* n_parents_dev, uint32_t max_parents,
                                   double s, const uint32_t dims[3], float step, float threshold, uint32_t* sums_dev,
                                   uint32_t* counter_dev, double* children_dev, uint32_t capacity, uint32_t world,
                                   uint32_t rank, void* stream);
int hu_mass_integrals_indirect(const double* parents_dev, const uint32_t* sums_dev, const uint32_t* n_parents_dev,
                               uint32_t max_parents, double s, double* out_dev, uint32_t rows, void* stream);

/* ---- renderers on the same evaluate() -------------------- */
/* rendering/ray_caster.cl:146-159, launched by rendering/ray_caster.py:93-110 with global size
 * (width, height).  origin/forward/up/right: float4 as the reference passes them (forward already
 * scaled by the focal length; 4th component ignored).  render_options: bit0 false colour, bit1
 * zebra (ray_caster.cl:9-10).  out_dev: uchar[width*height*3], pixel (x, y) at (y + height*x)*3. */
int hu_ray_caster(hu_tape t, const float origin[4], const float forward[4], const float up[4],
                  const float right[4], float pixel_tolerance, float box_radius, float min_distance,
                  float max_distance, float floor_z, uint32_t render_options, uint32_t width,
                  uint32_t height, void* out_dev, void* stream);
/* rendering/bitmap.cl:1-4, launched by rendering/bitmap.py:22-26: inside/outside picture of a 2D
 * shape, sample (x, y) at origin + step_size*(x, height-y-1).  out_dev as above. */
int hu_bitmap(hu_tape t, const float origin[4], float step_size, uint32_t width, uint32_t height,
              void* out_dev, void* stream);

/* ---- 2D contouring ----------------------------------------- */
/* rendering/polygon2d.cl:82-93, launched by rendering/polygon2d.py:101-112 with global size
 * grid = (gx-1, gy-1, 2) over the float4 corner samples corners_dev[gx*gy] that hu_grid_eval wrote
 * for dims (gx, gy, 1).  Per triangular half cell (index t + 2*(y + (gy-1)*x)): vertices_dev
 * float2[cells] (written where the contour crosses the cell), links_dev uint32[cells] (next cell
 * along the contour, 0xffffffff = empty cell, top bits = where it leaves the block,
 * polygon2d.cl:5-36), starts_dev uint32[(gx-1)+(gy-1)] + *start_counter_dev (caller zeroes): cells
 * that begin a chain entering through the block's boundary, in unspecified order. */
int hu_process_polygon(const float box_corner[2], float box_step, const void* corners_dev,
                       const uint32_t grid[2], void* vertices_dev, uint32_t* links_dev,
                       uint32_t* starts_dev, uint32_t* start_counter_dev, void* stream);
/* The same for ALL leaf blocks of a 2D subdivision in one launch (the per-block loop of
 * polygon2d.py:84-126).  corners_dev: float4[n_blocks][dims[0]*dims[1]] as written by
 * hu_grid_eval_blocks(layout 0) over dims (gx, gy, 1); block corner = (float)(int_corner *
 * resolution + origin).  Outputs are per block, consecutive; start_counters_dev: uint32[n_blocks]. */
int hu_process_polygon_blocks(const void* corners_dev, const int32_t* blocks_dev, uint32_t n_blocks,
                              double resolution, const double origin[3], float step,
                              const uint32_t dims[2], void* vertices_dev, uint32_t* links_dev,
                              uint32_t* starts_dev, uint32_t* start_counters_dev, void* stream);

/* ---- leaf-block consumer: marching cubes ------------------- */
/* Replaces the per-block copy + `mcubes.marching_cubes(block, 0)` of rendering/mesh.py:53-63 (PyMCubes
 * 0.0.6, a dependency that is not part of the reference tree) for ALL leaf blocks at once.
 * fields_dev: float[n_blocks][dims[0]*dims[1]*dims[2]], each block an array [A0][A1][A2] (last index
 * fastest), inside = value <= 0; for blocks written by hu_grid_eval_blocks(layout 1) over (sx, sy, sz)
 * samples pass dims = (sy, sx, sz).  The unit of work is a segment (up to 32 consecutive samples along
 * the last axis: a row of a 16^3 block), 256 segments per workgroup: hu_mesh_workgroups gives the
 * number n of workgroups, the number of (uint32, uint32) entries wg_counts_dev must hold (n + 1 + scan
 * scratch) and the total number of segments.
 * hu_mesh_count: masks_dev uint32[segments] <- one inside bit per sample; wg_counts_dev[0..n) <-
 * exclusive prefix of (vertices, triangles) per workgroup; entry n holds the totals (read it back to
 * size the outputs; block b starts at workgroup b*n/n_blocks).  Asynchronous.
 * hu_mesh_emit (same fields, masks and counts): vertices_dev double[total_vertices][3] in world
 * coordinates exactly as mesh.py:65-68 computes them (swap the first two array axes, negate y, * step,
 * + block corner, all in fp64; block corner = int_corner*resolution + origin), plus y_offset on y (0 =
 * the reference's placement, which sits (A0-1)*step below the true one; (A0-1)*step = true positions);
 * triangles_dev uint32[total_triangles][3], global vertex ids, anticlockwise seen from outside the solid;
 * seg_info_dev: scratch uint32[4*segments].  Order: vertices by owning sample then axis, triangles by
 * cell -- deterministic, no atomics.  At most 2^32 - 1 vertices per call. */
int hu_mesh_workgroups(uint32_t n_blocks, const uint32_t dims[3], uint64_t* n_workgroups,
                       uint64_t* count_entries, uint64_t* segments);
int hu_mesh_count(const float* fields_dev, uint32_t n_blocks, const uint32_t dims[3],
                  uint32_t* masks_dev, uint32_t* wg_counts_dev, void* stream);
int hu_mesh_emit(const float* fields_dev, const int32_t* blocks_dev, uint32_t n_blocks,
                 double resolution, const double origin[3], double step, const uint32_t dims[3],
                 double y_offset, const uint32_t* masks_dev, const uint32_t* wg_counts_dev,
                 uint32_t* seg_info_dev, double* vertices_dev, uint32_t* triangles_dev, void* stream);
/* Binary STL records of an indexed mesh (rendering/stl_renderer.py:8-24, which fills a numpy-stl 1.8.0 mesh
 * triangle by triangle on the host and saves it): records_dev (16-byte aligned, 50*n_triangles bytes) <-
 * per triangle the normal, the three corners -- vertices_dev rounded to float32 -- and a zero attribute
 * word, little endian, i.e. the body of the file after its 80-byte header and uint32 count.  normal =
 * (v1-v0) x (v2-v0) in float32, unnormalised, as numpy-stl's update_normals computes it on save.  Every
 * index in triangles_dev must be a valid vertex.  Asynchronous. */
int hu_mesh_stl(const double* vertices_dev, const uint32_t* triangles_dev, uint64_t n_triangles,
                void* records_dev, void* stream);

/* Order a list of integer block corners (int32[4] rows, e.g. the leaf list of hu_subdivision_level) by
 * (x, y, z) on the device, in place, so that per-block output comes out in a reproducible order (the
 * kernels append survivors in the order workgroups finished).  Two calls: with scratch_dev NULL (or too
 * small) only *needed is set; then with scratch_bytes >= *needed the list is sorted; synchronises the
 * stream.  Corners must lie within +-2^20 resolution units. */
int hu_sort_blocks(int32_t* blocks_dev, uint32_t n_blocks, void* scratch_dev, size_t scratch_bytes,
                   size_t* needed, void* stream);

/* Per-tape specialisation (the reference's generate_fixed_eval_source_code, nodes/codegen.py:137-204):
 * unroll the decoded program into straight-line gfx950 code with hipRTC, using the op library
 * headers found in `include_dir` (codecad_amd/csrc).  Afterwards every launch with this tape runs
 * the specialised kernels; results are identical to the interpreter's.  Costs one compilation
 * (seconds); returns HU_ERR_UNSUPPORTED with the compiler log if hipRTC cannot build it. */
int hu_tape_specialize(hu_tape t, const char* include_dir);
/* The same with an on-disk cache of the compiled code objects: `cache_dir` (NULL or "" = no cache) holds one
 * file per (generated source, op library headers, compiler options, hipRTC/HIP version); a hit loads in
 * milliseconds instead of compiling for seconds -- the counterpart of pyopencl's program cache behind the
 * reference's `Program(...).build()` (cl_util/opencl_manager.py:116-141).  The directory is created if its
 * parent exists; unreadable, truncated or stale files are ignored and rebuilt; an unwritable directory is not
 * an error.  only_if_cached != 0: load when cached, otherwise leave the tape interpreted and return HU_OK.
 * `*from_cache` (may be NULL) <- 1 if the kernels came from the cache. */
int hu_tape_specialize_cached(hu_tape t, const char* include_dir, const char* cache_dir,
                              int only_if_cached, int* from_cache);
/* The same for a subset of the per-tape KERNELS: bit i of `groups` is kernel i of the list below, and the HU_SPEC_*
 * constants are the sets a kind of launch needs (its FAMILY).  Compiling only what is about to be used takes a fraction
 * of the time of all nineteen kernels, and a tape's kernels may be built one by one, side by side in several processes
 * (round 4: a tape's first kernel is ready after its OWN compilation, not after its family's).  Kernels that are loaded
 * already are skipped; a launch whose kernel is not loaded runs the interpreter (a launch over boxes whose mask kernel is
 * not loaded yet treats every operand as alive: same bits).
 *   bit 0, 1    k_grid_eval (float4, float)            bit 10      k_box_masks (box pruning, every launch over boxes)
 *   bit 2, 3    k_grid_eval_blocks (float4, float)     bit 11, 12  k_grid_eval_ragged (float4, float)
 *   bit 4..7    k_classify ([MASS][BATCH])             bit 13, 14  k_grid_eval_blocks_ragged (float4, float)
 *   bit 8, 9    k_ray_caster, k_bitmap                 bit 15, 16  k_grid_eval_runs (float4, float)
 *   bit 17, 18  k_grid_eval_blocks_runs (float4, float)
 * (ragged: boxes that may end anywhere, for extents that are no multiples of (4, 4, 8); runs: the in-place form over runs of
 * cells, where boxes would be mostly padding -- 2D grids)
 * With a cache directory: the image of exactly the requested set is taken when it is there, else the images of its single
 * kernels (what the background builds leave behind); what is still missing is built as ONE image (only_if_cached = 0). */
enum hu_spec_group {
    HU_SPEC_DENSE = 0x19c03,    /* hu_grid_eval, hu_grid_eval_pymcubes, hu_grid_eval_slab */
    HU_SPEC_BLOCKS = 0x6640c,   /* hu_grid_eval_blocks[_indirect] */
    HU_SPEC_CLASSIFY = 0x04f0,  /* hu_subdivision_step / _level[_indirect], hu_mass_properties / _level[_indirect] */
    HU_SPEC_RENDER = 0x0300,    /* hu_ray_caster, hu_bitmap */
    HU_SPEC_ALL = 0x7ffff
};
int hu_tape_specialize_groups(hu_tape t, const char* include_dir, const char* cache_dir, int only_if_cached,
                              uint32_t groups, int* from_cache);
/* *out_flag <- the per-tape kernels that are loaded (hu_spec_group bits; 0: everything is interpreted) */
int hu_tape_specialized(hu_tape t, int* out_flag);
/* Box pruning of the tape's per-tape code (no counterpart in the reference, whose kernels evaluate every primitive for
 * every sample): *bits <- the operands of min / max that can be decided per 16^3 box of a launch (0: nothing in this tape
 * can be bounded, or it is interpreted), *words <- 32-bit words of a box's mask.  Launches over boxes run the tape's mask
 * kernel first, on their stream; HU_PRUNE=0 in the environment builds tapes without it, HU_PRUNE_RUN=0 skips the mask
 * kernel (every operand is then taken as alive).  Results are bit-identical either way. */
int hu_tape_prune_info(hu_tape t, int* bits, int* words);
/* A precompiled header for the per-tape builds (host only): hipRTC's runtime header + the op library of `include_dir`,
 * made in `dir` by the clang++ that sits next to the hipRTC in use, if there is one; the builds then skip parsing those
 * ~16 000 lines (a quarter of a family's build, most of a small kernel's).  `path` (may be NULL) <- the file, or "" when
 * none could be made (no such clang, no libhiprtc-builtins.so, unwritable directory): the builds then run as before.
 * Two files, their paths separated by a newline: big sources are built with -O1 (HU_RTC_BIG_KB), and clang takes a header
 * only at the optimisation level it was made at.
 * The per-tape builds look for it in <directory of libhip_util.so>/pch (where the library's build puts it) and in their
 * cache directory (where they make it themselves when it is missing).  HU_RTC_PCH=0 switches it off.  No counterpart in
 * the reference (pyopencl's build has no headers to parse). */
int hu_spec_pch_prepare(const char* include_dir, const char* dir, char* path, size_t capacity);
/* The HIP source hu_tape_specialize would compile for this tape (host only, no device needed):
 * `*needed` receives its size including the terminator; it is copied when `capacity` suffices. */
int hu_tape_source(const float* tape, size_t n_floats, char* buf, size_t capacity, size_t* needed);
/* A readable listing of a tape's decoded programs, one record per line (host only, no device needed): which = 0 the
 * full program, 1 the distance-only program (empty for tapes with a rounded blend), 2 / 3 the same as the
 * interpreter runs them, transformed primitives fused into single records.  Same calling convention. */
int hu_tape_listing(const float* tape, size_t n_floats, int which, char* buf, size_t capacity, size_t* needed);
/* Compile that source with hipRTC without loading it (host only, no device needed): checks that the
 * op library headers in `include_dir` build under hipRTC and that all ten kernels are present.
 * `*code_bytes` (may be NULL) receives the code object size. */
int hu_tape_compile_check(const float* tape, size_t n_floats, const char* include_dir, size_t* code_bytes);
/* The same through the cache of hu_tape_specialize_cached (host only): a miss compiles and stores, a hit
 * only reads.  `*from_cache` (may be NULL) <- 1 on a hit. */
int hu_tape_compile_cached(const float* tape, size_t n_floats, const char* include_dir,
                           const char* cache_dir, size_t* code_bytes, int* from_cache);
/* ... for a subset of the kernels (hu_spec_group bits), as ONE image named after the set */
int hu_tape_compile_groups(const float* tape, size_t n_floats, const char* include_dir, const char* cache_dir,
                           uint32_t groups, size_t* code_bytes, int* from_cache);

/* Device self-test of the arithmetic contract: the kernels compute sqrt(x) and 1/sqrt(x) with a
 * short hardware-seeded sequence instead of the compiler's IEEE expansion (csrc/interp.hpp
 * sqrt_cr / sqrt_inv_cr).  This runs both on ALL 2^32 binary32 inputs, one and two voxels per
 * lane, and counts disagreements with the IEEE results: counts[0..2] = mismatches of sqrt_cr,
 * sqrt_inv_cr's root, sqrt_inv_cr's reciprocal (all must be 0); counts[3] = inputs on the fast
 * path.  Synchronous, about 0.1 s. */
int hu_selftest_math(uint64_t counts[4]);
/* ... and of the three-operand minimum / maximum per-tape code uses for min(min(a, b), c) / max(max(a, b), c)
 * (csrc/interp.hpp min3_ / max3_): every ordered triple of 64 special values (zeros of both signs, denormals,
 * infinities, quiet and signalling NaNs) and 2^26 random triples, one and two voxels per lane, against the two
 * instructions they replace: counts[0], counts[1] = disagreements of min3, max3 (must be 0); counts[2] = triples.
 * Synchronous. */
int hu_selftest_minmax3(uint64_t counts[3]);

#ifdef __cplusplus
}
#endif
#endif /* HIP_UTIL_H */
