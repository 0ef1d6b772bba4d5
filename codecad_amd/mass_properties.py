"""Volume, centroid and inertia tensor by hierarchical voxel integration.

`mass_properties(shape, resolution, grid_size=None) -> MassProperties(volume, centroid,
inertia_tensor)` with the reference's semantics (reference mass_properties.py:30-229): cells
provably inside contribute closed-form moments of their integer indices, ambiguous cells are
subdivided, the finest level classifies by the sign at the cell centre.

Device side: one launch per LEVEL (`hu_mass_properties_level`) instead of one per block with
three fresh allocations and two blocking reads each (reference :75-114); per-parent moment
sums stay uint32 exactly like the reference's kernel so the integers are identical, and the
fp64 conversion below applies the same formulas (reference :119-148) to whole arrays.
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


def level_integrals(tape, parents, n_parents, s, dims, leaf, queue, counter):
    """Launch one level; returns (integral dict, children Buffer, child count)."""
    lib = hip_manager.lib
    cells = int(dims[0]) * int(dims[1]) * int(dims[2])
    thr = 0.0 if leaf else s * math.sqrt(3) / 2  # reference mass_properties.py:87-90
    d = (ctypes.c_uint32 * 3)(int(dims[0]), int(dims[1]), int(dims[2]))
    sums = hip_util.Buffer(numpy.uint32, (n_parents, 10), queue=queue)
    capacity = 0 if leaf else subdivision.child_capacity(n_parents, cells)
    # the ten integrals of this level are reduced on the device, one row per slice of ~2048 parents: at most
    # 5 KB come back, and the rows are added here in order (deterministic whatever the launch did)
    rows = integral_rows(n_parents)
    out = hip_util.Buffer(numpy.float64, (rows, 10), queue=queue)
    while True:
        children = hip_util.Buffer(numpy.float64, (max(capacity, 1), 4), queue=queue)
        sums.enqueue_fill(0)
        counter.enqueue_fill(0)
        tape.note_samples(n_parents * cells)
        check(lib.hu_mass_properties_level(tape.device_ptr, parents.device_ptr, n_parents, float(s), d,
                                           numpy.float32(s), numpy.float32(thr), sums.device_ptr,
                                           counter.device_ptr, children.device_ptr, capacity, queue.handle),
              "hu_mass_properties_level")
        # enqueued before the host waits for the counter: the reduction runs during that round trip
        check(lib.hu_mass_integrals(parents.device_ptr, sums.device_ptr, n_parents, float(s), out.device_ptr, rows,
                                    queue.handle), "hu_mass_integrals")
        count = int(counter.read()[0])
        if count <= capacity or leaf:
            break
        children.release()
        capacity = count
    values = [math.fsum(column) for column in out.read().T.tolist()]
    sums.release()
    out.release()
    return dict(zip(_KEYS, values)), children, (0 if leaf else count)


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

    parents = hip_util.Buffer(numpy.float64, (1, 4), queue=queue)
    parents.enqueue_write(numpy.array([[box.a.x, box.a.y, box.a.z, 0.0]], dtype=numpy.float64))
    counter = hip_util.Buffer(numpy.uint32, 1, queue=queue)
    total = {k: util.KahanSummation() for k in _KEYS}
    count = 1
    stats = {"kernel_invocations": 0, "function_evaluations": 0}
    for i, (s, dims) in enumerate(levels):
        leaf = (i == len(levels) - 1)
        stats["kernel_invocations"] += 1
        stats["function_evaluations"] += count * int(dims[0]) * int(dims[1]) * int(dims[2])
        part, children, n_children = level_integrals(tape, parents, count, s, dims, leaf, queue, counter)
        for k in total:
            total[k] += part[k]
        parents.release()
        parents, count = children, n_children
        if count == 0:
            break
    parents.release()
    counter.release()
    result = finish({k: v.result for k, v in total.items()})
    mass_properties.last_stats = stats
    return result
