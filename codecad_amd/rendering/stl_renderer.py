"""Binary STL export (reference rendering/stl_renderer.py:8-24, which goes through numpy-stl 1.8.0).

The reference copies every corner of every triangle into a numpy-stl mesh in a Python loop and lets
`Mesh.save` compute the normals.  Here the 50-byte records are assembled on the device from the indexed mesh
(`hu_mesh_stl`) -- only the bytes of the file travel to the host."""
import struct

import numpy

from .. import hip_util
from ..hip_util import manager as hip_manager, check
from . import mesh as _mesh

RECORD = numpy.dtype([("normal", "<f4", 3), ("vectors", "<f4", (3, 3)), ("attr", "<u2")])


def stl_records(vertices, triangles, queue=None):
    """One 50-byte STL record per triangle of an indexed mesh held on the host (uploaded, assembled by
    `hu_mesh_stl`, downloaded): float32 corners, normal = (v1-v0) x (v2-v0) (not normalised, as numpy-stl's
    update_normals leaves it).  Returns a structured array of dtype RECORD."""
    queue = queue or hip_manager.queue
    vertices = numpy.ascontiguousarray(vertices, dtype=numpy.float64).reshape(-1, 3)
    triangles = numpy.ascontiguousarray(triangles, dtype=numpy.uint32).reshape(-1, 3)
    n = len(triangles)
    if n == 0:
        return numpy.zeros(0, dtype=RECORD)
    if int(triangles.max()) >= len(vertices):
        raise ValueError("triangle index out of range")
    v = hip_util.Buffer(numpy.float64, vertices.shape, queue=queue)
    t = hip_util.Buffer(numpy.uint32, triangles.shape, queue=queue)
    rec = hip_util.Buffer(numpy.uint8, (n, 50), queue=queue)
    v.enqueue_write(vertices)
    t.enqueue_write(triangles)
    check(hip_manager.lib.hu_mesh_stl(v.device_ptr, t.device_ptr, n, rec.device_ptr, queue.handle), "hu_mesh_stl")
    out = rec.read().copy().view(RECORD).reshape(n)
    for b in (v, t, rec):
        b.release()
    return out


def write_stl(filename, records):
    """records: uint8 (n, 50) or a RECORD array."""
    with open(filename, "wb") as fp:
        fp.write(b"codecad_amd binary STL".ljust(80, b" "))
        fp.write(struct.pack("<I", len(records)))
        fp.write(memoryview(numpy.ascontiguousarray(records)).cast("B"))
    return len(records)


def render_stl(obj, filename, subdivision_grid_size=None):
    """Mesh the shape and write a binary STL; the records stream from the device into the file piece by piece
    (the triangle count in the header is patched in at the end).  Returns the number of triangles."""
    with open(filename, "wb") as fp:
        fp.write(b"codecad_amd binary STL".ljust(80, b" "))
        fp.write(struct.pack("<I", 0))
        mesh = _mesh.mesh_arrays(obj, subdivision_grid_size, download=False,
                                 stl_sink=lambda piece: fp.write(memoryview(piece).cast("B")))
        if mesh.n_triangles >= 2 ** 32:
            raise ValueError("too many triangles for a binary STL")
        fp.seek(80)
        fp.write(struct.pack("<I", mesh.n_triangles))
    return mesh.n_triangles
