"""Binary STL export (reference rendering/stl_renderer.py:8-24, which goes through numpy-stl)."""
import struct

import numpy

from . import mesh as _mesh

_RECORD = numpy.dtype([("normal", "<f4", 3), ("vectors", "<f4", (3, 3)), ("attr", "<u2")])


def stl_records(vertices, triangles):
    """One 50-byte STL record per triangle: float32 corners, normal = (v1-v0) x (v2-v0) (not normalised,
    as numpy-stl's update_normals leaves it)."""
    rec = numpy.zeros(len(triangles), dtype=_RECORD)
    rec["vectors"] = vertices[triangles].astype(numpy.float32)
    v = rec["vectors"]
    rec["normal"] = numpy.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    return rec


def render_stl(obj, filename, subdivision_grid_size=None):
    mesh = _mesh.mesh_arrays(obj, subdivision_grid_size)
    rec = stl_records(mesh.vertices, mesh.triangles)
    with open(filename, "wb") as fp:
        fp.write(b"codecad_amd binary STL".ljust(80, b" "))
        fp.write(struct.pack("<I", len(rec)))
        fp.write(rec.tobytes())
    return len(rec)
