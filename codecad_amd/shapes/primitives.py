"""Primitive shapes: what each one contributes to the evaluation DAG, its bounding box and its
feature size.  2D: Rectangle, Circle, HalfPlane, RegularPolygon2D, InvoluteGearBase; 3D: Sphere,
HalfSpace; 2D -> 3D: Extrusion, Revolution.  (Reference: shapes/simple2d.py, simple3d.py, gears.py;
the public module names `simple2d`, `simple3d`, `gears` re-export from here.)
"""
import math

from .. import util
from . import base


# ---- 2D ------------------------------------------------------------------------------------
class Rectangle(base.Shape2D):
    def __init__(self, x=1, y=None):
        self.half_size = util.Vector(x, x if y is None else y) / 2

    def bounding_box(self):
        return util.BoundingBox(-self.half_size, self.half_size)

    def feature_size(self):
        return 2 * min(self.half_size.x, self.half_size.y)

    def get_node(self, point, cache):
        return cache.make_node("rectangle", [self.half_size.x, self.half_size.y], [point])


class Circle(base.Shape2D):
    def __init__(self, d=1, r=None):
        if r is None:
            self.d, self.r = d, d / 2
        else:
            self.d, self.r = 2 * r, r

    def bounding_box(self):
        v = util.Vector(self.r, self.r)
        return util.BoundingBox(-v, v)

    def feature_size(self):
        return self.d

    def get_node(self, point, cache):
        return cache.make_node("circle", [self.r], [point])


class HalfPlane(base.Shape2D):
    """The half plane y > 0."""

    def bounding_box(self):
        inf = float("inf")
        return util.BoundingBox(util.Vector(-inf, 0), util.Vector(inf, inf))

    def feature_size(self):
        return float("inf")

    def get_node(self, point, cache):
        return cache.make_node("half_space", [], [point])


class RegularPolygon2D(base.Shape2D):
    """Regular n-gon given by exactly one of d, r, side_length, across_flats."""

    def __init__(self, n, d=1, r=None, side_length=None, across_flats=None):
        if not util.at_most_one([d != 1, r is not None, side_length is not None, across_flats is not None]):
            raise ValueError("At most one of d, r, side_length and across_flats can be used at the same time")
        self.n = n
        c = math.cos(math.pi / n)
        flats_per_r = (c + 1) if n % 2 else 2 * c  # odd n: vertex-to-flat; even n: flat-to-flat
        if across_flats is not None:
            self.r = across_flats / flats_per_r
        elif side_length is not None:
            self.r = side_length / math.sin(math.pi / n) / 2
        elif r is not None:
            self.r = r
        else:
            self.r = d / 2
        self.d = 2 * self.r if side_length is None else side_length / math.sin(math.pi / n)
        self.across_flats = across_flats if across_flats is not None else self.r * flats_per_r
        self.side_length = side_length if side_length is not None else self.d * math.sin(math.pi / n)

    @staticmethod
    def calculate_n(r, side_length):
        """The (generally fractional) n for which radius r gives this side length."""
        return math.pi / math.asin(side_length / (2 * r))

    def bounding_box(self):
        v = util.Vector(self.r, self.r)
        return util.BoundingBox(-v, v)

    def feature_size(self):
        return self.side_length

    def get_node(self, point, cache):
        return cache.make_node("regular_polygon2d", [math.pi / self.n, self.r], [point])


class InvoluteGearBase(base.Shape2D):
    """Unit-pitch-radius external involute profile with sharp tips and no root land."""

    def __init__(self, tooth_count, pressure_angle):
        self.tooth_count = tooth_count
        self.pressure_angle = math.radians(pressure_angle)

    def bounding_box(self):
        return util.BoundingBox(util.Vector(-1.5, -1.5), util.Vector(1.5, 1.5))

    def feature_size(self):
        return 0.5 * math.pi / self.tooth_count  # half a tooth thickness

    def get_node(self, point, cache):
        return cache.make_node("involute_gear", [self.tooth_count, self.pressure_angle], [point])


# ---- 3D ------------------------------------------------------------------------------------
class Sphere(base.Shape3D):
    def __init__(self, d=1, r=None):
        if r is None:
            self.d, self.r = d, d / 2
        else:
            self.d, self.r = 2 * r, r

    def bounding_box(self):
        v = util.Vector.splat(self.r)
        return util.BoundingBox(-v, v)

    def feature_size(self):
        return self.d

    def get_node(self, point, cache):
        return cache.make_node("sphere", [self.r], [point])


class HalfSpace(base.Shape3D):
    """The half space y > 0."""

    def bounding_box(self):
        inf = float("inf")
        return util.BoundingBox(util.Vector(-inf, 0, -inf), util.Vector.splat(inf))

    def feature_size(self):
        return float("inf")

    def get_node(self, point, cache):
        return cache.make_node("half_space", [], [point])


# ---- 2D -> 3D ------------------------------------------------------------------------------
class Extrusion(base.Shape3D):
    """A 2D shape swept along z, symmetric about z = 0."""

    def __init__(self, s, height):
        self.check_dimension(s, required=2)
        self.s, self.h = s, height

    def bounding_box(self):
        b = self.s.bounding_box()
        return util.BoundingBox(util.Vector(b.a.x, b.a.y, -self.h / 2), util.Vector(b.b.x, b.b.y, self.h / 2))

    def feature_size(self):
        return min(self.s.feature_size(), self.h)

    def get_node(self, point, cache):
        flat = self.s.get_node(point, cache)
        if math.isinf(self.h):
            return flat  # an infinite prism is its own cross-section (reference simple3d.py:109-110)
        return cache.make_node("extrusion", [self.h / 2], [flat, point])


class Revolution(base.Shape3D):
    """A 2D shape revolved around the y axis at radius r, optionally twisted."""

    def __init__(self, s, r, twist):
        self.check_dimension(s, required=2)
        self.s, self.r = s, r
        self.twist = math.radians(twist)
        self.minor_r = max(abs(p) for p in s.bounding_box().points2d())
        if self.twist != 0 and self.minor_r >= 0.9 * r:
            raise ValueError("Radius of the revolved object around origin must be less than 90% of "
                             "revolution radius when twist is applied.")

    def bounding_box(self):
        b = self.s.bounding_box()
        if self.twist == 0:
            radius = self.r + max(-b.a.x, b.b.x)
            return util.BoundingBox(util.Vector(-radius, b.a.y, -radius), util.Vector(radius, b.b.y, radius))
        v = util.Vector(self.r + self.minor_r, self.minor_r, self.r + self.minor_r)
        return util.BoundingBox(-v, v)

    def feature_size(self):
        if self.twist > 2 * math.pi:
            # distance between two copies of a point meeting at the innermost radius
            return min(self.s.feature_size(),
                       2 * math.sin(2 * math.pi * math.pi / self.twist) * (self.r - self.minor_r))
        return self.s.feature_size()

    def get_node(self, point, cache):
        if self.twist == 0:
            flat_point = cache.make_node("revolution_to", [], [point])
            return cache.make_node("revolution_from", [], [self.s.get_node(flat_point, cache), point])
        flat_point = cache.make_node("twist_revolution_to", [self.r, self.twist], [point])
        return cache.make_node("twist_revolution_from", [self.minor_r, self.r, self.twist],
                               [self.s.get_node(flat_point, cache), point])


# ---------------------------------------------------------------------------------------------
# Constructor functions: the user-facing spelling of the classes above (exported by the package).
# ---------------------------------------------------------------------------------------------
def rectangle(x=1, y=None):
    """Axis-aligned rectangle centred on the origin; one argument gives a square.
    Tape op `rectangle(half_w, half_h)`: exact distance, outward unit direction."""
    return Rectangle(x, x if y is None else y)


def circle(d=1, r=None):
    """Circle centred on the origin by diameter `d` or radius `r` (tape op `circle(r)`)."""
    return Circle(d, r)


def half_plane():
    """The half plane y > 0 (tape op `half_space`, shared with the 3D half space)."""
    return HalfPlane()


def regular_polygon2d(n, d=1, r=None, side_length=None, across_flats=None):
    """Regular n-gon with a vertex on +x, sized by exactly one of circumscribed diameter `d`,
    circumscribed radius `r`, `side_length` or `across_flats`."""
    return RegularPolygon2D(n, d, r, side_length, across_flats)


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
    return Sphere(2 * r if r is not None else d)


def cylinder(h=1, d=1, r=None, symmetrical=True):
    """Cylinder along z: a circle extruded by `h`, centred on z = 0 unless symmetrical=False
    (then it stands on the z = 0 plane)."""
    return circle(d=d, r=r).extruded(h, symmetrical)


def half_space():
    """The half space y > 0."""
    return HalfSpace()
