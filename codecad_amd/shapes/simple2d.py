"""2D primitives and the 2D bindings of the combinators (reference shapes/simple2d.py)."""
import math

from .. import util
from . import base, common


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


class Union2D(common.UnionMixin, base.Shape2D):
    pass


class Intersection2D(common.IntersectionMixin, base.Shape2D):
    pass


class Subtraction2D(common.SubtractionMixin, base.Shape2D):
    pass


class Offset2D(common.OffsetMixin, base.Shape2D):
    pass


class Shell2D(common.ShellMixin, base.Shape2D):
    pass


class Transformation2D(common.TransformationMixin, base.Shape2D):
    def bounding_box(self):
        box = common.transformed_box(self, self.s.bounding_box().flattened())
        return util.BoundingBox(util.Vector(box.a.x, box.a.y), util.Vector(box.b.x, box.b.y))


class Mirror2D(common.MirrorMixin, base.Shape2D):
    pass


class Symmetrical2D(common.SymmetricalMixin, base.Shape2D):
    pass


# registry used by ShapeBase's operators
UNION, INTERSECTION, SUBTRACTION = Union2D, Intersection2D, Subtraction2D
OFFSET, SHELL, MIRROR, SYMMETRICAL = Offset2D, Shell2D, Mirror2D, Symmetrical2D
