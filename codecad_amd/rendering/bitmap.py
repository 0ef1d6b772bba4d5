"""Inside/outside picture of a 2D shape (reference rendering/bitmap.py:12-30, bitmap.cl:1-18)."""
import numpy

from .. import util
from .. import nodes
from .. import hip_util
from ..hip_util import manager as hip_manager


def kernel_arguments(obj, size):
    """(origin Vector, step) of the pixel grid: the bounding box centred in the image."""
    obj.check_dimension(required=2)
    box = obj.bounding_box().flattened()
    resolution = util.Vector(size[0], size[1], 1)   # the 1 avoids a division by zero
    step_size = box.size().elementwise_div(resolution).max()
    return box.midpoint() - resolution * step_size / 2, step_size


def render(obj, size):
    """-> uint8 array (height, width, 3)."""
    origin, step_size = kernel_arguments(obj, size)
    tape = nodes.make_program_buffer(obj)
    size = (int(size[0]), int(size[1]))
    output = hip_util.Buffer(numpy.uint8, size + (3,))
    ev = hip_manager.k.bitmap(size, None, tape, origin.as_float4(), numpy.float32(step_size), output)
    pixels = output.read(wait_for=[ev]).copy()
    output.release()
    return pixels.transpose((1, 0, 2))
