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
    """Axis-aligned rectangle centred on the origin; one argument gives a square.
    Tape op `rectangle(half_w, half_h)`: exact distance, outward unit direction."""
    return _s2.Rectangle(x, x if y is None else y)


def circle(d=1, r=None):
    """Circle centred on the origin by diameter `d` or radius `r` (tape op `circle(r)`)."""
    return _s2.Circle(d, r)


def half_plane():
    """The half plane y > 0 (tape op `half_space`, shared with the 3D half space)."""
    return _s2.HalfPlane()


def regular_polygon2d(n, d=1, r=None, side_length=None, across_flats=None):
    """Regular n-gon with a vertex on +x, sized by exactly one of circumscribed diameter `d`,
    circumscribed radius `r`, `side_length` or `across_flats`."""
    return _s2.RegularPolygon2D(n, d, r, side_length, across_flats)


def polygon2d(points):
    """Simple polygon from (x, y) points, either winding; rejects self-intersections."""
    return _polygons2d.Polygon2D(points)


def polygon2d_builder(origin_x, origin_y):
    """Turtle-style builder: `.dx(..).dy(..).angle(..).close()` -> polygon."""
    return _polygons2d.Polygon2D.build(origin_x, origin_y)


def capsule(x1, y1, x2, y2, width):
    """Stadium between two points: a zero-height rectangle offset by width/2."""
    dx, dy = x2 - x1, y2 - y1
    return (rectangle(math.hypot(dx, dy), 0).offset(width / 2)
            .rotated(math.degrees(math.atan2(dy, dx))).translated((x1 + x2) / 2, (y1 + y2) / 2))


def box(x=1, y=None, z=None):
    """Cuboid centred on the origin (a rectangle extruded symmetrically); one argument gives a cube.
    `float("inf")` along z gives an infinite prism (the extrusion node is then omitted)."""
    if (y is None) != (z is None):
        raise ValueError("y and z must either both be None, or both be number")
    if y is None:
        y = z = x
    return rectangle(x, y).extruded(z)


def sphere(d=1, r=None):
    """Sphere centred on the origin by diameter or radius (tape op `sphere(r)`)."""
    return _s3.Sphere(2 * r if r is not None else d)


def cylinder(h=1, d=1, r=None, symmetrical=True):
    """Cylinder along z: a circle extruded by `h`, centred on z = 0 unless symmetrical=False
    (then it stands on the z = 0 plane)."""
    return circle(d=d, r=r).extruded(h, symmetrical)


def half_space():
    """The half space y > 0."""
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
    """Intersection; r >= 0 rounds the seams with that radius."""
    return _group(shapes, "Intersection", _s2.Intersection2D, _s3.Intersection, r)
