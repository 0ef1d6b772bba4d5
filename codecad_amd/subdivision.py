"""Adaptive subdivision: skip provably empty / provably full space, return leaf blocks.

Entry points keep the reference's names and return shapes (reference subdivision.py:116-253):
`calculate_block_sizes(...)` and `subdivision(shape, resolution, overlap_edge_samples,
grid_size) -> (tape_buffer, max_grid_dims, [(dims, corner, step, int_corner, int_step)])`.

What changed underneath (MI355X-first): the reference walks the hierarchy one block at a
time -- per block a 4-byte H2D, a launch of <= 128^3 work-items, a blocking 4-byte read and
a blocking read of the WHOLE 8 MiB index list, then a Python object per survivor
(subdivision.py:48-113).  Here every LEVEL is one launch over all surviving parents
(`hu_subdivision_level`), survivors are compacted on the device with a wavefront
ballot/prefix scan, stay in HBM as the next level's parent list, and the host reads one
4-byte counter per level.  `subdivision_device()` exposes that form; `subdivision()`
converts the final list to the reference's tuples.
"""
import ctypes
import math
import os

import numpy

from . import util
from . import nodes
from . import hip_util
from .hip_util import manager as hip_manager, check


def calculate_block_sizes(box, dimension, resolution, grid_size, overlap, level_size_multiplier=1):
    """Hierarchy layout `[(cell_size_in_resolution_units, Vector grid_dims)]`, coarsest first.

    Every level but the top is a full grid_size^dimension grid; a level's cell is covered
    exactly by the next level's grid; with `overlap` the LEAF grids share their last sample
    plane with the neighbour (so a leaf cell spans grid_size-1 steps); the top level is
    cropped to the box (rounded up to `level_size_multiplier`).  Same results as reference
    subdivision.py:116-166 (tests/test_subdivision_host.py compares against golden tables).
    """
    if grid_size % level_size_multiplier != 0:
        raise ValueError("Grid size must be divisible by level_size_multiplier")
    if dimension == 2:
        full = util.Vector(grid_size, grid_size, 1)
        box = box.flattened()
    elif dimension == 3:
        full = util.Vector.splat(grid_size)
    else:
        raise AssertionError("dimension must be 2 or 3")

    extent = (box.size() / resolution).applyfunc(math.ceil)  # box size in resolution units
    longest = extent.max()

    cells = [1]  # cell size per level, finest first
    while True:
        shared = 1 if (overlap and len(cells) == 1) else 0
        coarser = cells[-1] * (grid_size - shared)
        if coarser >= longest:
            break
        cells.append(coarser)
    shared = 1 if (overlap and len(cells) == 1) else 0  # single level: the top IS the leaf

    top = cells[-1]
    top_dims = util.Vector(*(
        util.clamp(util.round_up_to(math.ceil(e / top) + shared, level_size_multiplier), 1, limit)
        for e, limit in zip(extent, full)))
    levels = [(c, full) for c in cells[:-1]] + [(top, top_dims)]
    levels.reverse()
    return levels


class LeafBlocks:
    """Device-resident result of `subdivision_device`: int32[n,4] leaf corners + metadata."""

    def __init__(self, tape, blocks, count, dims, step, int_step, resolution, origin, level_counts, samples):
        self.tape = tape
        self.blocks = blocks            # hip_util.Buffer int32 (capacity, 4); first `count` valid
        self.count = count
        self.dims = dims                # Vector: samples per leaf block
        self.step = step                # sample spacing inside a leaf block (float)
        self.int_step = int_step
        self.resolution = resolution
        self.origin = origin            # Vector: position of integer coordinate (0,0,0)
        self.level_counts = level_counts  # survivors after each evaluated level
        self.samples = samples          # SDF evaluations spent

    def sort(self):
        """Order the leaf list by integer corner (x, then y, then z), on the device (`hu_sort_blocks`).  The order
        the kernels leave depends on how workgroups raced for list space; consumers that emit per-block
        output (meshes, contours) sort first so that their output is reproducible."""
        if self.count > 1:
            lib = hip_manager.lib
            needed = ctypes.c_size_t(0)
            check(lib.hu_sort_blocks(self.blocks.device_ptr, self.count, None, 0, ctypes.byref(needed), None), "hu_sort_blocks")
            scratch = hip_util.Buffer(numpy.uint8, needed.value, queue=self.blocks.queue)
            check(lib.hu_sort_blocks(self.blocks.device_ptr, self.count, scratch.device_ptr, needed.value,
                                     ctypes.byref(needed), self.blocks.queue.handle), "hu_sort_blocks")
            scratch.release()
        return self

    def int_corners(self):
        """(count, 3) int32 numpy array, sorted lexicographically (deterministic order)."""
        if self.count == 0:
            return numpy.zeros((0, 3), dtype=numpy.int32)
        host = numpy.empty((self.blocks.shape[0], 4), dtype=numpy.int32)
        self.blocks.read(out=host)
        a = host[:self.count, :3]
        order = numpy.lexsort((a[:, 2], a[:, 1], a[:, 0]))
        return numpy.ascontiguousarray(a[order])


def child_capacity(n_parents, cells):
    """First size of a level's child list (the launch is repeated with the exact count if it is short): every
    cell when that is at most 2^21 rows -- a level of a fractal keeps most of its cells, and a repeat costs more
    than the memory --, else the surface estimate: it crosses O(cells^(2/3)) cells of a block, x4 headroom."""
    upper = int(n_parents) * int(cells)
    if upper <= 1 << 21:
        return upper
    return min(upper, max(1 << 21, 4 * int(n_parents) * int(round(cells ** (2.0 / 3.0)))))


def checked_capacity(rows):
    """A list capacity as the C ABI takes it (uint32).  Lists that long cannot be launched in one piece anyway."""
    rows = int(rows)
    if rows >= 2 ** 32 - 1:
        raise MemoryError("a level's list would hold %d rows: more than the 32-bit list lengths of the device code "
                          "(use a coarser resolution or a larger grid_size)" % rows)
    return rows


def list_budget_rows(row_bytes):
    """Rows a FIRST-GUESS list may hold: CODECAD_AMD_LIST_BUDGET_MB (default 64 MiB) per list, at least 65 536 rows."""
    budget = int(float(os.environ.get("CODECAD_AMD_LIST_BUDGET_MB", "64")) * (1 << 20))
    return max(1 << 16, budget // int(row_bytes))


def first_capacities(cells_per_level, n_top=1, row_bytes=16):
    """First guesses for the lists of a device-counted traversal, one per classified level.  Chained from level to
    level like the per-level driver's sizes, but every guess is capped by a memory budget: the chain runs on
    CAPACITIES (the counts are not known before the traversal has run), so uncapped it compounds -- 1/8192 at grid 16
    asked for a 1.7 GB list, 1/40000 for 32 GB, where the per-level driver sized each list from the count before it.
    A traversal that overflows a guess reports what it needed and is repeated with that (Overflow / the drivers'
    retry loops), so a tight cap costs one repeat, never a wrong result."""
    cap_rows = list_budget_rows(row_bytes)
    capacities, bound = [], int(n_top)
    for c in cells_per_level:
        capacities.append(checked_capacity(min(child_capacity(bound, c), cap_rows)))
        bound = capacities[-1]
    return capacities


def remembered_capacities(tape, key, cells_per_level, n_top=1, row_bytes=16):
    """Capacities for a traversal of `tape` under `key` (what fixes the hierarchy: driver, resolution, grid, box): what the
    SAME traversal needed last time, + 12 % -- a launch is sized for its lists' capacities (workgroups past a list's end
    cost their dispatch, and the launchers choose boxes or runs by it), so a first guess of 110 592 rows for a level of 27
    costs every call something; the first traversal of a kind starts from first_capacities()."""
    memo = getattr(tape, "_level_memo", None)
    if memo is None:
        memo = tape._level_memo = {}
    seen = memo.get(key)
    if seen is not None and len(seen) == len(cells_per_level):
        return [checked_capacity(int(n * 1.125) + 16) for n in seen]
    return first_capacities(cells_per_level, n_top, row_bytes)


def remember_counts(tape, key, counts):
    getattr(tape, "_level_memo", {}).__setitem__(key, [int(n) for n in counts])


def _level_launch(tape, parents, n_parents, int_step, dims, dimension, resolution, origin, counter, queue,
                  capacity_hint=None):
    """Run one level, growing the child list until everything fits.  Returns (children, count)."""
    lib = hip_manager.lib
    cells = int(dims[0]) * int(dims[1]) * int(dims[2])
    box_step = int_step * resolution
    thr = box_step * math.sqrt(dimension) / 2  # reference subdivision.py:67
    d = (ctypes.c_uint32 * 3)(int(dims[0]), int(dims[1]), int(dims[2]))
    o = (ctypes.c_double * 3)(origin.x, origin.y, origin.z)
    upper = n_parents * cells
    capacity = child_capacity(n_parents, cells) if capacity_hint is None else min(upper, capacity_hint)
    while True:
        children = hip_util.Buffer(numpy.int32, (max(capacity, 1), 4), queue=queue)
        counter.enqueue_fill(0)
        tape.note_samples(n_parents * int(dims[0]) * int(dims[1]) * int(dims[2]), hip_util.SPEC_CLASSIFY)
        check(lib.hu_subdivision_level(tape.device_ptr, parents.device_ptr, n_parents, int(int_step), d,
                                       dimension, float(resolution), o, numpy.float32(box_step),
                                       numpy.float32(thr), counter.device_ptr, children.device_ptr,
                                       capacity, queue.handle), "hu_subdivision_level")
        count = int(counter.read()[0])
        if count <= capacity:
            return children, count
        children.release()
        capacity = count


def _traverse_device_counted(tape, levels, dimension, resolution, origin, capacities, queue):
    """All classification levels enqueued back to back, no host round trip between them: every list is a
    `[header row | rows...]` buffer whose header word 0 is its length; a level counts into the header of its child
    list and the next launch -- sized for the capacity -- reads its parent count from there
    (`hu_subdivision_level_indirect`).  Returns (buffers, counts) after ONE synchronisation; counts above the
    capacities mean "repeat with larger lists"."""
    lib = hip_manager.lib
    o = (ctypes.c_double * 3)(origin.x, origin.y, origin.z)
    top = hip_util.Buffer(numpy.int32, (2, 4), queue=queue)
    top.enqueue_write(numpy.array([[1, 0, 0, 0], [0, 0, 0, 0]], dtype=numpy.int32))
    buffers, parents, max_parents = [], top, 1
    for (int_step, dims), capacity in zip(levels[:-1], capacities):
        children = hip_util.Buffer(numpy.int32, (capacity + 1, 4), queue=queue)
        check(lib.hu_memset(children.device_ptr, 0, 16, queue.handle), "hu_memset")
        box_step = int_step * resolution
        thr = box_step * math.sqrt(dimension) / 2  # reference subdivision.py:67
        d = (ctypes.c_uint32 * 3)(int(dims[0]), int(dims[1]), int(dims[2]))
        check(lib.hu_subdivision_level_indirect(tape.device_ptr, parents.device_ptr + 16, parents.device_ptr, max_parents,
                                                int(int_step), d, dimension, float(resolution), o, numpy.float32(box_step),
                                                numpy.float32(thr), children.device_ptr, children.device_ptr + 16, capacity,
                                                queue.handle), "hu_subdivision_level_indirect")
        buffers.append(children)
        parents, max_parents = children, capacity
    # the headers: small asynchronous reads, one wait
    heads = [numpy.zeros(4, dtype=numpy.int32) for _ in buffers]
    for b, h in zip(buffers, heads):
        check(lib.hu_memcpy_d2h(h.ctypes.data, b.device_ptr, 16, queue.handle), "hu_memcpy_d2h")
    queue.synchronize()
    top.release()
    return buffers, [int(h[0]) for h in heads]


def subdivision_device(shape, resolution, overlap_edge_samples=True, grid_size=None, queue=None):
    """Level-synchronous subdivision that leaves the leaf list on the GPU -> LeafBlocks."""
    if grid_size is None:
        grid_size = 128
    assert resolution > 0, "Non-positive resolution makes no sense"
    assert grid_size > 1, "Grid needs to be at least 2x2x2"
    assert grid_size <= 256, "Grid size > 256 would cause overflows in returned index list."
    queue = queue or hip_manager.queue
    tape = nodes.make_program_buffer(shape)
    dimension = shape.dimension()
    box = shape.bounding_box().expanded_additive(resolution / 2)
    if dimension == 2:
        box = box.flattened()
    levels = calculate_block_sizes(box, dimension, resolution, grid_size, overlap_edge_samples)
    leaf_int_step, leaf_dims = levels[-1]
    cells = [int(d[0]) * int(d[1]) * int(d[2]) for _, d in levels]
    if len(levels) == 1:
        parents = hip_util.Buffer(numpy.int32, (1, 4), queue=queue)
        parents.enqueue_write(numpy.zeros((1, 4), dtype=numpy.int32))
        return LeafBlocks(tape, parents, 1, leaf_dims, leaf_int_step * resolution, leaf_int_step, resolution, box.a, [], 0)

    # first sizes as before (every cell while that is small, else the surface estimate); the whole traversal is
    # repeated with the sizes it reported when a list was too short (fractal shapes keep most cells: rare)
    memo_key = ("subdivision", float(resolution), int(grid_size), bool(overlap_edge_samples), tuple(box.a), tuple(box.b))
    capacities = remembered_capacities(tape, memo_key, cells[:-1])
    while True:
        buffers, counts = _traverse_device_counted(tape, levels, dimension, resolution, box.a, capacities, queue)
        if all(n <= c for n, c in zip(counts, capacities)):
            remember_counts(tape, memo_key, counts)
            break
        for b in buffers:
            b.release()
        # (a level behind an overflowed one saw only the parents that fitted: its count may still grow next time)
        capacities = [checked_capacity(max(c, int(n * 1.125) + 16)) for n, c in zip(counts, capacities)]
    level_counts, samples, parents_n = [], 0, 1
    for n, c in zip(counts, cells[:-1]):
        samples += parents_n * c
        tape.note_samples(parents_n * c, hip_util.SPEC_CLASSIFY)
        level_counts.append(n)
        parents_n = n
        if n == 0:
            break
    count = level_counts[-1]
    # consumers take a plain row list: the leaf rows without their header row
    leaves = hip_util.Buffer(numpy.int32, (max(count, 1), 4), queue=queue)
    if count:
        check(hip_manager.lib.hu_memcpy_d2d(leaves.device_ptr, buffers[len(level_counts) - 1].device_ptr + 16, count * 16, queue.handle),
              "hu_memcpy_d2d")
    for b in buffers:
        b.release()      # stream-ordered: the pool hands the block to the next user of this queue
    return LeafBlocks(tape, leaves, count, leaf_dims, leaf_int_step * resolution, leaf_int_step, resolution,
                      box.a, level_counts, samples)


def subdivision(shape, resolution, overlap_edge_samples=True, grid_size=None):
    """Reference-shaped result: `(tape_buffer, max_grid_dims, final_blocks)`.

    `final_blocks` is a list of `(grid_dims, corner, spacing, int_corner, int_spacing)`;
    `corner = int_corner * resolution + origin` with origin the (expanded) bounding box
    corner, exactly as reference subdivision.py:96-111.  Blocks come sorted by integer
    corner (the reference's order depends on atomics and traversal and is unspecified).
    """
    leaves = subdivision_device(shape, resolution, overlap_edge_samples, grid_size)
    if leaves.int_step == 1 and not leaves.level_counts:
        # the whole shape fits one block (reference subdivision.py:222-227)
        return (leaves.tape, leaves.dims,
                [(leaves.dims, leaves.origin, resolution, util.Vector(0, 0, 0), 1)])
    blocks = []
    for ix, iy, iz in leaves.int_corners().tolist():
        ip = util.Vector(ix, iy, iz)
        blocks.append((leaves.dims, ip * resolution + leaves.origin, leaves.step, ip, leaves.int_step))
    leaves.blocks.release()
    return leaves.tape, leaves.dims, blocks
