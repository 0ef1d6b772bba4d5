"""Volume, centroid and inertia tensor by hierarchical voxel integration.

`mass_properties(shape, resolution, grid_size=None) -> MassProperties(volume, centroid,
inertia_tensor)` with the reference's semantics (reference mass_properties.py:30-229): cells
provably inside contribute closed-form moments of their integer indices, ambiguous cells are
subdivided, the finest level classifies by the sign at the cell centre.

Device side: one launch per LEVEL (`hu_mass_properties_level_indirect`) instead of one per block with
three fresh allocations and two blocking reads each (reference :75-114), all levels enqueued back to
back with their list lengths on the device and ONE synchronisation at the end; per-parent moment
sums stay uint32 exactly like the reference's kernel so the integers are identical, and the
fp64 conversion (`hu_mass_integrals_indirect`) applies the same formulas (reference :119-148).
"""
import collections
import ctypes
import math

import numpy

from . import util
from . import nodes
from . import hip_util
from . import subdivision
from .hip_util import manager as hip_manager, check


class MassProperties(collections.namedtuple("MassProperties", "volume centroid inertia_tensor")):
    """Volume, centroid (Vector) and 3x3 inertia tensor about the CENTROID (unit density)."""

    __slots__ = ()


_KEYS = ("1", "x", "y", "z", "xx", "yy", "zz", "xy", "xz", "yz")


def integrals_host(sums, corners, s):
    """Host (numpy) evaluation of the same formulas, used by the tests to cross-check the
    device reduction `hu_mass_integrals`; the driver does not call it.

    Per-parent index sums -> the ten integrals over the inside cells.

    sums: (n,10) uint32 in kernel order xx,xy,xz,x,yy,yz,y,zz,z,n; corners: (n,3) float64 box
    corners; s: cell size.  Formulas of reference mass_properties.py:119-148 (a cell
    contributes its centre moments plus s^2/12 on the diagonal second moments).
    """
    f = sums.astype(numpy.float64)
    sxx, sxy, sxz, sx, syy, syz, sy, szz, sz, n = (f[:, i] for i in range(10))
    s2 = s * s
    s3 = s * s2
    bx, by, bz = (corners[:, i] + s / 2 for i in range(3))
    tx, ty, tz = s * sx, s * sy, s * sz
    txx, tyy, tzz = s2 * sxx, s2 * syy, s2 * szz
    txy, txz, tyz = s2 * sxy, s2 * sxz, s2 * syz
    parts = {
        "1": s3 * n,
        "x": s3 * (n * bx + tx), "y": s3 * (n * by + ty), "z": s3 * (n * bz + tz),
        "xx": s3 * (n * (bx * bx + s2 / 12) + 2 * bx * tx + txx),
        "yy": s3 * (n * (by * by + s2 / 12) + 2 * by * ty + tyy),
        "zz": s3 * (n * (bz * bz + s2 / 12) + 2 * bz * tz + tzz),
        "xy": s3 * (n * bx * by + bx * ty + by * tx + txy),
        "xz": s3 * (n * bx * bz + bx * tz + bz * tx + txz),
        "yz": s3 * (n * by * bz + by * tz + bz * ty + tyz),
    }
    # extended-precision accumulation: at least as accurate as the reference's Kahan sums and
    # independent of the order in which parents were listed
    return {k: float(numpy.sum(v, dtype=numpy.longdouble)) for k, v in parts.items()}


def finish(total):
    """Integrals -> MassProperties (parallel-axis shift to the centroid), reference :179-229."""
    volume = total["1"]
    if volume == 0:
        return MassProperties(0, util.Vector.splat(0), numpy.zeros((3, 3)))
    c = util.Vector(total["x"], total["y"], total["z"]) / volume

    def second(key, a, ia, b, ib):
        return total[key] - a * ib - b * ia + a * b * volume

    xx = total["xx"] - 2 * c.x * total["x"] + c.x * c.x * volume
    yy = total["yy"] - 2 * c.y * total["y"] + c.y * c.y * volume
    zz = total["zz"] - 2 * c.z * total["z"] + c.z * c.z * volume
    xy = second("xy", c.x, total["x"], c.y, total["y"])
    xz = second("xz", c.x, total["x"], c.z, total["z"])
    yz = second("yz", c.y, total["y"], c.z, total["z"])
    tensor = numpy.array([[yy + zz, -xy, -xz], [-xy, xx + zz, -yz], [-xz, -yz, xx + yy]])
    return MassProperties(volume, c, tensor)


def _enqueue_levels(tape, levels, box_a, capacities, queue):
    """Every level of the integration enqueued back to back, no host round trip between them: a list is a
    `[header row | rows...]` buffer of 32-byte rows whose header word 0 is its length; a level counts its ambiguous cells
    into the header of its child list, the next launch -- sized for the capacity -- reads its parent count from there
    (`hu_mass_properties_level_indirect`), and the level's ten integrals are reduced on the device from the same count
    (`hu_mass_integrals_indirect`).  The reference waits for the host after every block (mass_properties.py:98-114).
    What the host wants to see afterwards -- every list's header and every level's rows of integrals -- is gathered in ONE
    small device buffer (a 32-byte device-to-device copy per level; the integrals are written there directly) and read
    once, into its pinned shadow.  (Round 3 copied headers and integrals level by level into pageable numpy arrays: such a
    copy waits for the stream, i.e. once per level.)
    Returns (survivor count per level, [rows x 10 array of integrals per level]) after ONE synchronisation; a count
    above its capacity means "repeat with larger lists"."""
    lib = hip_manager.lib
    top = hip_util.Buffer(numpy.float64, (2, 4), queue=queue)
    first = numpy.zeros((2, 4), dtype=numpy.float64)
    first.view(numpy.uint32)[0, 0] = 1
    first[1, :3] = box_a
    top.enqueue_write(first)
    # results: per level [header row: 4 doubles | rows x 10 doubles]
    rows_of, max_parents = [], 1
    for i in range(len(levels)):
        rows_of.append(integral_rows(max_parents))
        max_parents = 0 if i == len(levels) - 1 else capacities[i]
    offsets = [0]
    for r in rows_of:
        offsets.append(offsets[-1] + 4 + 10 * r)
    results = hip_util.Buffer(numpy.float64, (offsets[-1],), queue=queue)
    parents, max_parents = top, 1
    buffers = [top, results]
    for i, (s, dims) in enumerate(levels):
        leaf = i == len(levels) - 1
        capacity = 0 if leaf else capacities[i]
        thr = 0.0 if leaf else s * math.sqrt(3) / 2  # reference mass_properties.py:87-90
        d = (ctypes.c_uint32 * 3)(int(dims[0]), int(dims[1]), int(dims[2]))
        children = hip_util.Buffer(numpy.float64, (capacity + 1, 4), queue=queue)
        sums = hip_util.Buffer(numpy.uint32, (max_parents, 10), queue=queue)
        out_ptr = results.device_ptr + 8 * (offsets[i] + 4)
        check(lib.hu_memset(children.device_ptr, 0, 32, queue.handle), "hu_memset")
        check(lib.hu_memset(sums.device_ptr, 0, max_parents * 40, queue.handle), "hu_memset")
        check(lib.hu_mass_properties_level_indirect(tape.device_ptr, parents.device_ptr + 32, parents.device_ptr, max_parents, float(s), d,
                                                    numpy.float32(s), numpy.float32(thr), sums.device_ptr, children.device_ptr,
                                                    children.device_ptr + 32, capacity, queue.handle),
              "hu_mass_properties_level_indirect")
        check(lib.hu_mass_integrals_indirect(parents.device_ptr + 32, sums.device_ptr, parents.device_ptr, max_parents, float(s),
                                             out_ptr, rows_of[i], queue.handle), "hu_mass_integrals_indirect")
        check(lib.hu_memcpy_d2d(results.device_ptr + 8 * offsets[i], children.device_ptr, 32, queue.handle), "hu_memcpy_d2d")
        buffers += [children, sums]
        if leaf:
            break
        parents, max_parents = children, capacity
    got = results.read()            # one copy into the pinned shadow, one synchronisation
    heads = [int(got[offsets[i]:offsets[i] + 1].view(numpy.uint32)[0]) for i in range(len(levels))]
    partials = [got[offsets[i] + 4:offsets[i + 1]].reshape(rows_of[i], 10).copy() for i in range(len(levels))]
    for b in buffers:
        b.release()
    return heads, partials


def integral_rows(n_parents):
    """Rows (workgroups) hu_mass_integrals is asked for: one per ~2048 parents, at most 64."""
    return max(1, min(64, (int(n_parents) + 2047) // 2048))


def mass_properties(shape, resolution, grid_size=None):
    if grid_size is None:
        grid_size = 64
    assert shape.dimension() == 3, "2D objects are not supported yet"
    assert resolution > 0, "Non-positive resolution makes no sense"
    assert grid_size > 1, "Grid needs to be at least 2x2x2"
    assert grid_size ** 5 <= 2 ** 32, "Centroid coordinate sums would overflow"

    queue = hip_manager.queue
    tape = nodes.make_program_buffer(shape)
    box = shape.bounding_box()
    levels = [(resolution * cell, dims) for cell, dims in
              subdivision.calculate_block_sizes(box, 3, resolution, grid_size, overlap=False)]
    cells = [int(d[0]) * int(d[1]) * int(d[2]) for _, d in levels]
    # the whole hierarchy is enqueued at once (device-counted lists); it is repeated with the sizes it reported when a
    # list was too short -- the first guesses are generous, so that is rare
    memo_key = ("mass_properties", float(resolution), int(grid_size), tuple(box.a), tuple(box.b))
    capacities = subdivision.remembered_capacities(tape, memo_key, cells[:-1], row_bytes=32 + 40)
    while True:
        counts, partials = _enqueue_levels(tape, levels, (box.a.x, box.a.y, box.a.z), capacities, queue)
        if all(n <= c for n, c in zip(counts, capacities)):
            subdivision.remember_counts(tape, memo_key, counts[:len(capacities)])
            break
        capacities = [subdivision.checked_capacity(max(c, int(n * 1.125) + 16)) for n, c in zip(counts, capacities)]
    total = {k: util.KahanSummation() for k in _KEYS}
    stats = {"kernel_invocations": 0, "function_evaluations": 0}
    parents_n = 1
    for i, part in enumerate(partials):
        stats["kernel_invocations"] += 1
        stats["function_evaluations"] += parents_n * cells[i]
        tape.note_samples(parents_n * cells[i], hip_util.SPEC_CLASSIFY)
        for k, column in zip(_KEYS, part.T.tolist()):   # the rows of a level are added in order: deterministic
            total[k] += math.fsum(column)
        parents_n = counts[i] if i < len(counts) else 0
        if parents_n == 0:
            break
    result = finish({k: v.result for k, v in total.items()})
    mass_properties.last_stats = stats
    return result
