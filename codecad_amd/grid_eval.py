"""Dense SDF grids (reference grid_eval.cl:2-34; the reference's grid_eval.py only registers
the .cl file and callers launch `opencl_manager.k.grid_eval*` themselves).

The kernels are reachable in the reference's style through `hip_util.manager.k.grid_eval(...)`
and `.grid_eval_pymcubes(...)`.  The helpers below are the conveniences the reference's
consumers re-implement inline (rendering/mesh.py:16-63, rendering/polygon2d.py:86-99):
evaluate a whole grid, or all leaf blocks of a subdivision in ONE launch.
"""
import ctypes

import numpy

from . import nodes
from . import hip_util
from .hip_util import manager as hip_manager, check

FLOAT4 = hip_util.Buffer.quad_dtype(numpy.float32)


def grid_eval(shape, corner, step, dims, pymcubes=False, out=None, queue=None):
    """Evaluate `shape` at corner + step*(i,j,k) for a dims-sized grid.

    Returns a hip_util.Buffer: float4 (nx,ny,nz,distance) indexed [x][y][z]
    (z + sz*(y + sy*x)), or with pymcubes=True a float grid indexed z + (x + (sy-1-y)*sx)*sz.
    """
    queue = queue or hip_manager.queue
    tape = nodes.make_program_buffer(shape)
    dims = tuple(int(d) for d in dims)
    if out is None:
        out = hip_util.Buffer(numpy.float32 if pymcubes else FLOAT4, dims, queue=queue)
    kernel = hip_manager.k.grid_eval_pymcubes if pymcubes else hip_manager.k.grid_eval
    ev = kernel(dims, None, tape, _corner4(corner), numpy.float32(step), out, queue=queue)
    out.event = ev
    return out


def grid_eval_blocks(leaves, pymcubes=False, out=None, queue=None):
    """All leaf blocks of `subdivision_device()` in one launch.

    Returns a Buffer holding leaves.count consecutive grids of leaves.dims (float4, or float
    in the pymcubes layout).  Replaces the per-block launch + blocking copy loop of reference
    rendering/mesh.py:45-61.
    """
    queue = queue or hip_manager.queue
    dims = tuple(int(d) for d in leaves.dims)
    if out is None:
        out = hip_util.Buffer(numpy.float32 if pymcubes else FLOAT4, (max(leaves.count, 1),) + dims, queue=queue)
    d = (ctypes.c_uint32 * 3)(*dims)
    o = (ctypes.c_double * 3)(leaves.origin.x, leaves.origin.y, leaves.origin.z)
    ev = hip_util.Event(hip_manager, queue)
    leaves.tape.note_samples(leaves.count * dims[0] * dims[1] * dims[2], hip_util.SPEC_BLOCKS)
    check(hip_manager.lib.hu_grid_eval_blocks(leaves.tape.device_ptr, leaves.blocks.device_ptr, leaves.count,
                                              float(leaves.resolution), o, numpy.float32(leaves.step), d,
                                              1 if pymcubes else 0, out.device_ptr, queue.handle),
          "hu_grid_eval_blocks")
    out.event = ev._done()
    return out


def _corner4(corner):
    if hasattr(corner, "as_float4"):
        return corner.as_float4()
    a = numpy.zeros(4, dtype=numpy.float32)
    c = numpy.asarray(corner, dtype=numpy.float64).reshape(-1)
    a[:min(3, c.size)] = c[:3]
    return a
