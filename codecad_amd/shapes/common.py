"""Dimension-independent CSG combinators (reference shapes/common.py:11-203).

Each class defines how bounding boxes / feature sizes combine and which DAG nodes it
emits.  simple2d.py / simple3d.py bind them to Shape2D / Shape3D.
"""
import functools

from .. import util


class UnionMixin:
    NODE = "union"

    def __init__(self, shapes, r=-1):
        self.shapes = list(shapes)
        self.check_dimension(*self.shapes)
        self.r = r  # r < 0: sharp (plain min); r >= 0: rounded blend of that radius

    def bounding_box(self):
        return functools.reduce(lambda a, b: a.union(b), (s.bounding_box() for s in self.shapes))

    def feature_size(self):
        return min(s.feature_size() for s in self.shapes)

    def get_node(self, point, cache):
        return cache.make_node(self.NODE, [self.r], [s.get_node(point, cache) for s in self.shapes])


class IntersectionMixin(UnionMixin):
    NODE = "intersection"

    def bounding_box(self):
        return functools.reduce(lambda a, b: a.intersection(b), (s.bounding_box() for s in self.shapes))


class SubtractionMixin:
    def __init__(self, s1, s2):
        self.check_dimension(s1, s2)
        self.s1, self.s2 = s1, s2

    def bounding_box(self):
        return self.s1.bounding_box()

    def feature_size(self):
        return min(self.s1.feature_size(), self.s2.feature_size())

    def get_node(self, point, cache):
        return cache.make_node("subtraction", [-1],
                               [self.s1.get_node(point, cache), self.s2.get_node(point, cache)])


class TransformationMixin:
    """Rotation + uniform scale + translation.

    The tape stores the INVERSE transform for the way in (`transformation_to`: sample point
    -> shape coordinates) and the forward quaternion for the way out (`transformation_from`:
    rotate the direction back, scale the distance).  Nested transforms collapse: a
    `transformation_to` directly on top of another (or on the initial one) is merged into
    it, and likewise for `transformation_from` (reference shapes/common.py:73-115), so a
    tower of .translated().rotated().scaled() costs one instruction each way.
    """

    def __init__(self, s, quaternion, translation):
        self.check_dimension(s)
        self.s = s
        self.transformation = util.Transformation(quaternion, translation)

    def feature_size(self):
        return self.s.feature_size() * self.transformation.quaternion.abs_squared()

    def get_node(self, point, cache):
        inverse = self.transformation.inverse()
        if point.name in ("transformation_to", "initial_transformation_to"):
            inverse = inverse * point.extra_data
            to_name, to_deps = point.name, point.dependencies
        else:
            to_name, to_deps = "transformation_to", [point]
        inner_point = cache.make_node(to_name, inverse.as_list(), to_deps, inverse)

        inner = self.s.get_node(inner_point, cache)

        quat = self.transformation.quaternion
        if inner.name == "transformation_from":
            quat = quat * inner.extra_data
            from_deps = inner.dependencies
        else:
            from_deps = [inner]
        return cache.make_node("transformation_from", quat.as_list(), from_deps, quat)


class MirrorMixin:
    def __init__(self, s):
        self.check_dimension(s)
        self.s = s

    def bounding_box(self):
        b = self.s.bounding_box()
        return util.BoundingBox(util.Vector(-b.b.x, b.a.y, b.a.z), util.Vector(-b.a.x, b.b.y, b.b.z))

    def feature_size(self):
        return self.s.feature_size()

    def get_node(self, point, cache):
        inner = self.s.get_node(cache.make_node("mirror", [], [point]), cache)
        return cache.make_node("mirror", [], [inner])


class SymmetricalMixin:
    def __init__(self, s):
        self.check_dimension(s)
        self.s = s

    def bounding_box(self):
        b = self.s.bounding_box()
        return util.BoundingBox(util.Vector(-b.b.x, b.a.y, b.a.z), util.Vector(b.b.x, b.b.y, b.b.z))

    def feature_size(self):
        return self.s.feature_size()

    def get_node(self, point, cache):
        inner = self.s.get_node(cache.make_node("symmetrical_to", [], [point]), cache)
        return cache.make_node("symmetrical_from", [], [inner, point])


class _GrownBoxMixin:
    def _grown(self, amount):
        box = self.s.bounding_box().expanded_additive(amount)
        return box.flattened() if self.dimension() == 2 else box


class OffsetMixin(_GrownBoxMixin):
    def __init__(self, s, distance):
        self.check_dimension(s)
        self.s = s
        self.distance = distance

    def bounding_box(self):
        return self._grown(self.distance)

    def feature_size(self):
        return max(0, self.s.feature_size() + self.distance * 2)

    def get_node(self, point, cache):
        return cache.make_node("offset", [self.distance], [self.s.get_node(point, cache)])


class ShellMixin(_GrownBoxMixin):
    def __init__(self, s, wall_thickness):
        self.check_dimension(s)
        self.s = s
        self.wall_thickness = wall_thickness

    def bounding_box(self):
        return self._grown(self.wall_thickness / 2)

    def feature_size(self):
        return self.wall_thickness  # finer features are swallowed by the wall

    def get_node(self, point, cache):
        return cache.make_node("shell", [self.wall_thickness / 2], [self.s.get_node(point, cache)])


def transformed_box(shape, box):
    """Bounding box of `box` after shape.transformation; infinite boxes stay infinite."""
    import math
    if any(math.isinf(v) for v in box.a) or any(math.isinf(v) for v in box.b):
        inf = float("inf")
        n = shape.dimension()
        hi = util.Vector(*([inf] * n))
        return util.BoundingBox(-hi, hi)
    return util.BoundingBox.containing(shape.transformation.transform_vector(v) for v in box.vertices())
