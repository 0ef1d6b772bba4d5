"""3D primitives, extrusion/revolution and the 3D bindings (reference shapes/simple3d.py)."""
import math

from .. import util
from . import base, common


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


class Union(common.UnionMixin, base.Shape3D):
    pass


class Intersection(common.IntersectionMixin, base.Shape3D):
    pass


class Subtraction(common.SubtractionMixin, base.Shape3D):
    pass


class Offset(common.OffsetMixin, base.Shape3D):
    pass


class Shell(common.ShellMixin, base.Shape3D):
    pass


class Transformation(common.TransformationMixin, base.Shape3D):
    def bounding_box(self):
        return common.transformed_box(self, self.s.bounding_box())


class Mirror(common.MirrorMixin, base.Shape3D):
    pass


class Symmetrical(common.SymmetricalMixin, base.Shape3D):
    pass


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


UNION, INTERSECTION, SUBTRACTION = Union, Intersection, Subtraction
OFFSET, SHELL, MIRROR, SYMMETRICAL = Offset, Shell, Mirror, Symmetrical
