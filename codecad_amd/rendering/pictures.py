"""Raster pictures of a shape: `render_pixels` picks the sphere-traced view for 3D shapes (ray_caster.py) and
the inside/outside bitmap for 2D ones; `render_pil_image` / `render_image` wrap it for Pillow.
(Reference: rendering/image.py, rendering/bitmap.py:12-30 with kernel bitmap.cl:1-18.)"""
import numpy

from .. import util
from .. import nodes
from .. import hip_util
from ..hip_util import manager as hip_manager
from . import ray_caster

DEFAULT_SIZE = (1024, 768)


def bitmap_arguments(obj, size):
    """(origin Vector, step) of the pixel grid of a 2D shape: its bounding box centred in the image,
    one pixel = the larger of the two box-extent / image-extent ratios."""
    obj.check_dimension(required=2)
    box = obj.bounding_box().flattened()
    pixels = util.Vector(size[0], size[1], 1)   # z = 1 avoids a division by zero
    step = box.size().elementwise_div(pixels).max()
    return box.midpoint() - pixels * step / 2, step


def render_bitmap(obj, size):
    """Inside/outside picture of a 2D shape -> uint8 array (height, width, 3)."""
    origin, step = bitmap_arguments(obj, size)
    width, height = int(size[0]), int(size[1])
    out = hip_util.Buffer(numpy.uint8, (width, height, 3))
    ev = hip_manager.k.bitmap((width, height), None, nodes.make_program_buffer(obj), origin.as_float4(),
                              numpy.float32(step), out)
    pixels = out.read(wait_for=[ev]).copy()
    out.release()
    return pixels.transpose((1, 0, 2))


def render_pixels(obj, size=DEFAULT_SIZE, view_angle=None):
    """uint8 RGB array (height, width, 3): bitmap for 2D shapes, ray-cast view along +y for 3D shapes
    (`view_angle` in degrees; None = a normal lens, focal length = image diagonal)."""
    if obj.dimension() == 2:
        return render_bitmap(obj, size)
    camera = ray_caster.get_camera_params(obj.bounding_box(), size, view_angle)
    return ray_caster.render(obj, *camera, size=size)


def render_pil_image(obj, size=DEFAULT_SIZE, view_angle=None):
    import PIL.Image
    return PIL.Image.fromarray(render_pixels(obj, size, view_angle))


def render_image(obj, filename, size=DEFAULT_SIZE, view_angle=None):
    render_pil_image(obj, size, view_angle).save(filename)
