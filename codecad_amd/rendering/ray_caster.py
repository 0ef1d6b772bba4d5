"""Sphere-traced picture of a 3D shape (reference rendering/ray_caster.py:30-115, kernel
rendering/ray_caster.cl:146-256): over-relaxed sphere tracing, one soft-shadow ray, ambient
occlusion, a floor plane with a contact shadow."""
import enum
import math

import numpy

from .. import util
from .. import nodes
from .. import hip_util
from ..hip_util import manager as hip_manager


class RenderOptions(enum.IntFlag):
    """Bits of the kernel's renderOptions argument."""
    no_flags = 0
    false_color = 1   # R = steps taken, G = residual * 1000
    zebra = 2         # stripes along y instead of the flat colour


def _zero_if_inf(x):
    return 0 if math.isinf(x) else x


def kernel_arguments(obj, origin, direction, up, focal_length):
    """The camera frame and scene scalars the kernel takes, from a look-at description
    (host arithmetic of reference ray_caster.py:34-47).  Pure host code."""
    box = obj.bounding_box()
    obj.check_dimension(required=3)
    forward = direction.normalized()
    up = (up - forward * up.dot(forward)).normalized()
    right = forward.cross(up)
    origin_to_midpoint = abs(origin - box.midpoint())
    box_radius = abs(box.size()) / 2
    return {
        "origin": origin, "forward": forward * focal_length, "up": up, "right": right,
        "pixel_tolerance": 0.5 / focal_length,   # tangent of half a pixel
        "box_radius": box_radius,
        "min_distance": max(0, origin_to_midpoint - box_radius),
        "max_distance": origin_to_midpoint + box_radius,
        "floor_z": box.a.z - box.size().z / 20,
    }


def get_camera_params(box, size, view_angle):
    """(origin, direction, up, focal_length) looking along +y at the whole box with a 20 % margin;
    view_angle None = normal lens (focal length = image diagonal)."""
    box_size = box.size()
    diagonal = math.hypot(*size)
    focal_length = diagonal if view_angle is None else diagonal / (2 * math.tan(math.radians(view_angle) / 2))
    distance = focal_length * max(_zero_if_inf(box_size.x) / size[0], _zero_if_inf(box_size.z) / size[1])
    if distance == 0:
        distance = 1
    distance *= 1.2
    origin = box.midpoint() - util.Vector(0, distance + _zero_if_inf(box_size.y) / 2, 0)
    return origin, util.Vector(0, 1, 0), util.Vector(0, 0, 1), focal_length


def render(obj, origin, direction, up, focal_length, size, options=RenderOptions.no_flags):
    """-> uint8 array (height, width, 3)."""
    a = kernel_arguments(obj, origin, direction, up, focal_length)
    tape = nodes.make_program_buffer(obj)
    size = (int(size[0]), int(size[1]))
    output = hip_util.Buffer(numpy.uint8, size + (3,))
    ev = hip_manager.k.ray_caster(size, None, tape, a["origin"].as_float4(), a["forward"].as_float4(),
                                  a["up"].as_float4(), a["right"].as_float4(), numpy.float32(a["pixel_tolerance"]),
                                  numpy.float32(a["box_radius"]), numpy.float32(a["min_distance"]),
                                  numpy.float32(a["max_distance"]), numpy.float32(a["floor_z"]),
                                  numpy.uint32(int(options)), output)
    pixels = output.read(wait_for=[ev]).copy()
    output.release()
    return pixels.transpose((1, 0, 2))
