"""User-facing shape constructors (reference shapes/__init__.py:19-109).

`from codecad_amd.shapes import *` gives the same vocabulary as the reference:
rectangle, circle, half_plane, regular_polygon2d, polygon2d, polygon2d_builder, capsule,
box, sphere, cylinder, half_space, union, intersection, plus the `unsafe` and `gears`
modules.
"""
import math

from . import simple2d as _s2
from . import simple3d as _s3
from . import polygons2d as _polygons2d
from . import unsafe  # noqa: F401
from . import gears  # noqa: F401
from .base import TapeShape  # noqa: F401


def rectangle(x=1, y=None):
    return _s2.Rectangle(x, x if y is None else y)


def circle(d=1, r=None):
    return _s2.Circle(d, r)


def half_plane():
    return _s2.HalfPlane()


def regular_polygon2d(n, d=1, r=None, side_length=None, across_flats=None):
    return _s2.RegularPolygon2D(n, d, r, side_length, across_flats)


def polygon2d(points):
    return _polygons2d.Polygon2D(points)


def polygon2d_builder(origin_x, origin_y):
    return _polygons2d.Polygon2D.build(origin_x, origin_y)


def capsule(x1, y1, x2, y2, width):
    """Stadium between two points: a zero-height rectangle offset by width/2."""
    dx, dy = x2 - x1, y2 - y1
    return (rectangle(math.hypot(dx, dy), 0).offset(width / 2)
            .rotated(math.degrees(math.atan2(dy, dx))).translated((x1 + x2) / 2, (y1 + y2) / 2))


def box(x=1, y=None, z=None):
    if (y is None) != (z is None):
        raise ValueError("y and z must either both be None, or both be number")
    if y is None:
        y = z = x
    return rectangle(x, y).extruded(z)


def sphere(d=1, r=None):
    return _s3.Sphere(2 * r if r is not None else d)


def cylinder(h=1, d=1, r=None, symmetrical=True):
    return circle(d=d, r=r).extruded(h, symmetrical)


def half_space():
    return _s3.HalfSpace()


def _group(shapes, what, cls2, cls3, r):
    shapes = list(shapes)
    if not shapes:
        raise ValueError(what + " of empty set objects doesn't make much sense, does it?")
    if len(shapes) == 1:
        return shapes[0]
    dim = shapes[0].dimension()
    if any(s.dimension() != dim for s in shapes):
        raise ValueError(what + " needs shapes of identical dimensions")
    return (cls2 if dim == 2 else cls3)(shapes, r=r)


def union(shapes, r=-1):
    """Union; r >= 0 rounds the seams with that radius."""
    return _group(shapes, "Union", _s2.Union2D, _s3.Union, r)


def intersection(shapes, r=-1):
    return _group(shapes, "Intersection", _s2.Intersection2D, _s3.Intersection, r)
