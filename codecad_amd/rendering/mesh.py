"""Triangular surface mesh of a 3D shape (reference rendering/mesh.py:10-74).

The reference walks the leaf blocks of a subdivision one at a time: `grid_eval_pymcubes` of the block, a
blocking copy to the host, PyMCubes on the CPU.  Here all leaf blocks are evaluated in one launch
(`grid_eval_blocks`) and meshed on the device by marching cubes over all of them at once
(`hu_mesh_count` / `hu_mesh_emit`): only the mesh travels to the host.

PyMCubes 0.0.6 is not part of the reference tree; see DESIGN.md "Mesh" for what is and is not the same as its
output.  Vertex positions follow mesh.py:65-68 literally, including that they sit `(sy-1)*step` lower in y than
the samples they were computed from (a pure translation of the whole mesh; pass `true_positions=True` to
`mesh_arrays` for the untranslated mesh).
"""
import ctypes

import numpy

from .. import hip_util
from .. import subdivision
from .. import util
from ..hip_util import manager as hip_manager, check
from .. import grid_eval as _grid_eval

_BOX_TRIANGLES = [[0, 3, 1], [0, 2, 3], [1, 3, 5], [3, 7, 5], [4, 5, 6], [5, 7, 6],
                  [0, 6, 2], [0, 4, 6], [0, 1, 5], [0, 5, 4], [3, 2, 6], [3, 6, 7]]


class Mesh:
    """Indexed mesh of all leaf blocks: `vertices` float64 (n, 3), `triangles` uint32 (m, 3) (global
    vertex ids), and per block the first vertex / triangle (`block_vertex_start`, `block_triangle_start`,
    length blocks + 1) so that block b is `vertices[vs[b]:vs[b+1]]`, `triangles[ts[b]:ts[b+1]] - vs[b]`."""

    def __init__(self, vertices, triangles, block_vertex_start, block_triangle_start, samples, kernel_ms,
                 stl_records=None, n_triangles=None):
        self.vertices = vertices
        self.triangles = triangles
        self.n_triangles = len(triangles) if n_triangles is None else n_triangles
        self.stl_records = stl_records      # uint8 (m, 50): the body of a binary STL file (mesh_blocks(stl=True))
        self.block_vertex_start = block_vertex_start
        self.block_triangle_start = block_triangle_start
        self.samples = samples
        self.kernel_ms = kernel_ms


_STL_CHUNK = 1 << 20     # triangles per streamed piece of a binary STL (50 MiB)


def _stream_stl(lib, queue, vertices, triangles, total_t, sink):
    """Assemble the STL records piece by piece on the device and hand each piece to `sink` (a uint8 (k, 50)
    view of pinned memory, valid during the call): two staging buffers, so the copy of one piece overlaps
    whatever the sink does with the previous one (e.g. a file write)."""
    if total_t == 0:
        return 0.0
    bufs = [hip_util.Buffer(numpy.uint8, (min(_STL_CHUNK, total_t), 50), queue=queue) for _ in range(2)]
    pending, ms = None, 0.0
    for i, first in enumerate(range(0, total_t, _STL_CHUNK)):
        count = min(_STL_CHUNK, total_t - first)
        b = bufs[i & 1]
        ev = hip_util.Event(hip_manager, queue)
        check(lib.hu_mesh_stl(vertices.device_ptr, triangles.device_ptr + 12 * first, count, b.device_ptr, queue.handle),
              "hu_mesh_stl")
        ev._done()
        copied = b.enqueue_read()
        if pending is not None:
            pending[0].wait()
            sink(pending[1].array[:pending[2]])
            ms += pending[3].elapsed_ms()
        pending = (copied, b, count, ev)
    pending[0].wait()
    sink(pending[1].array[:pending[2]])
    ms += pending[3].elapsed_ms()
    for b in bufs:
        b.release()
    return ms


def mesh_blocks(leaves, true_positions=False, queue=None, download=True, stl=False, stl_sink=None):
    """Marching cubes over the leaf blocks of `subdivision_device` -> Mesh (device work + one download).
    stl=True also assembles the binary STL records on the device (`hu_mesh_stl`) and downloads them;
    stl_sink=callable streams them instead, 2^20 triangles at a time (see _stream_stl);
    with download=False the indexed mesh itself stays behind."""
    queue = queue or hip_manager.queue
    sx, sy, sz = (int(d) for d in leaves.dims)
    n = leaves.count
    lib = hip_manager.lib
    if n == 0 or min(sx, sy, sz) < 2:
        z = numpy.zeros(n + 1, dtype=numpy.int64)
        return Mesh(numpy.zeros((0, 3)), numpy.zeros((0, 3), numpy.uint32), z, z, 0, 0.0,
                    numpy.zeros((0, 50), numpy.uint8) if stl else None, 0)
    fields = _grid_eval.grid_eval_blocks(leaves, pymcubes=True, queue=queue)
    dims = (ctypes.c_uint32 * 3)(sy, sx, sz)    # array axes of the pymcubes layout: (flipped y, x, z)
    n_wg, entries, segments = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(0)
    check(lib.hu_mesh_workgroups(n, dims, ctypes.byref(n_wg), ctypes.byref(entries), ctypes.byref(segments)),
          "hu_mesh_workgroups")
    n_wg = n_wg.value
    counts = hip_util.Buffer(numpy.uint32, (entries.value, 2), queue=queue)
    masks = hip_util.Buffer(numpy.uint32, (segments.value,), queue=queue)   # one inside bit per sample, a mask per segment
    ev0 = hip_util.Event(hip_manager, queue)
    check(lib.hu_mesh_count(fields.device_ptr, n, dims, masks.device_ptr, counts.device_ptr, queue.handle), "hu_mesh_count")
    ev0._done()
    prefix = counts.read(wait_for=[ev0]).copy()
    total_v, total_t = int(prefix[n_wg, 0]), int(prefix[n_wg, 1])
    assert total_v < 2 ** 32, "too many vertices for one call"
    chunks = n_wg // n
    starts = numpy.concatenate([prefix[0:n_wg:chunks], prefix[n_wg:n_wg + 1]]).astype(numpy.int64)
    info = hip_util.Buffer(numpy.uint32, (segments.value, 4), queue=queue)   # per segment: first vertex id, edge masks
    vertices = hip_util.Buffer(numpy.float64, (max(total_v, 1), 3), queue=queue)
    triangles = hip_util.Buffer(numpy.uint32, (max(total_t, 1), 3), queue=queue)
    o = (ctypes.c_double * 3)(leaves.origin.x, leaves.origin.y, leaves.origin.z)
    step = float(leaves.step)
    ev1 = hip_util.Event(hip_manager, queue)
    check(lib.hu_mesh_emit(fields.device_ptr, leaves.blocks.device_ptr, n, float(leaves.resolution), o, step, dims,
                           (sy - 1) * step if true_positions else 0.0, masks.device_ptr, counts.device_ptr, info.device_ptr,
                           vertices.device_ptr, triangles.device_ptr, queue.handle), "hu_mesh_emit")
    ev1._done()
    records, ev2 = None, None
    if stl:
        rec = hip_util.Buffer(numpy.uint8, (max(total_t, 1), 50), queue=queue)
        ev2 = hip_util.Event(hip_manager, queue)
        check(lib.hu_mesh_stl(vertices.device_ptr, triangles.device_ptr, total_t, rec.device_ptr, queue.handle), "hu_mesh_stl")
        ev2._done()
        records = rec.read(wait_for=[ev2])[:total_t]
        rec.release()
    stl_ms = _stream_stl(lib, queue, vertices, triangles, total_t, stl_sink) if stl_sink is not None else 0.0
    if download:
        # views of the pinned shadows (kept alive by the arrays): no second 0.7 GB copy on the host
        v = vertices.read(wait_for=[ev1])[:total_v]
        t = triangles.read()[:total_t]
    else:
        ev1.wait()
        v = t = None
    ms = ev0.elapsed_ms() + ev1.elapsed_ms() + (ev2.elapsed_ms() if ev2 is not None else 0.0) + stl_ms
    for b in (fields, counts, masks, info, vertices, triangles):
        b.release()
    return Mesh(v, t, starts[:, 0].copy(), starts[:, 1].copy(), n * sx * sy * sz, ms, records, total_t)


def mesh_arrays(obj, subdivision_grid_size=None, true_positions=False, download=True, stl=False, stl_sink=None):
    """-> Mesh of the whole shape (all blocks; vertices are shared inside a block, not between blocks)."""
    obj.check_dimension(required=3)
    leaves = subdivision.subdivision_device(obj, obj.feature_size() / 2, grid_size=subdivision_grid_size).sort()
    mesh = mesh_blocks(leaves, true_positions=true_positions, download=download, stl=stl, stl_sink=stl_sink)
    leaves.blocks.release()
    return mesh


def triangular_mesh(obj, subdivision_grid_size=None, debug_subdivision_boxes=False):
    """Generate a triangular mesh of the surface of a 3D shape.  Yields (vertices, indices) per leaf block
    that has any triangle, like the reference's generator."""
    obj.check_dimension(required=3)
    if debug_subdivision_boxes:
        _tape, _dims, boxes = subdivision.subdivision(obj, obj.feature_size() / 2, grid_size=subdivision_grid_size)
        for box_size, box_corner, box_resolution, *_ in boxes:
            yield ([util.Vector(i, j, k).elementwise_mul(box_size) * box_resolution + box_corner
                    for k in range(2) for j in range(2) for i in range(2)], _BOX_TRIANGLES)
        return
    mesh = mesh_arrays(obj, subdivision_grid_size)
    vs, ts = mesh.block_vertex_start, mesh.block_triangle_start
    for b in range(len(vs) - 1):
        if ts[b + 1] == ts[b]:
            continue
        yield mesh.vertices[vs[b]:vs[b + 1]], (mesh.triangles[ts[b]:ts[b + 1]] - numpy.uint32(vs[b]))
