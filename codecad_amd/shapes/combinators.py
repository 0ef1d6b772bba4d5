"""Shapes made from other shapes: boolean combinations, rigid/similarity transforms, mirror and
symmetry, offset and shell, (circular) repetition, the involute gear assembly.  Each class says
how bounding boxes / feature sizes combine and which DAG nodes it emits; the dimension-independent
part lives in the *Mixin classes, bound to Shape2D / Shape3D below.  (Reference: shapes/common.py,
simple2d.py, simple3d.py, unsafe.py, gears.py; those module names re-export from here.)
"""
import functools
import math

from .. import util
from . import base
from .primitives import Circle, InvoluteGearBase


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
    if any(math.isinf(v) for v in box.a) or any(math.isinf(v) for v in box.b):
        inf = float("inf")
        n = shape.dimension()
        hi = util.Vector(*([inf] * n))
        return util.BoundingBox(-hi, hi)
    return util.BoundingBox.containing(shape.transformation.transform_vector(v) for v in box.vertices())


# ---- bound to 2D ---------------------------------------------------------------------------
class Union2D(UnionMixin, base.Shape2D):
    pass


class Intersection2D(IntersectionMixin, base.Shape2D):
    pass


class Subtraction2D(SubtractionMixin, base.Shape2D):
    pass


class Offset2D(OffsetMixin, base.Shape2D):
    pass


class Shell2D(ShellMixin, base.Shape2D):
    pass


class Transformation2D(TransformationMixin, base.Shape2D):
    def bounding_box(self):
        box = transformed_box(self, self.s.bounding_box().flattened())
        return util.BoundingBox(util.Vector(box.a.x, box.a.y), util.Vector(box.b.x, box.b.y))


class Mirror2D(MirrorMixin, base.Shape2D):
    pass


class Symmetrical2D(SymmetricalMixin, base.Shape2D):
    pass


# ---- bound to 3D ---------------------------------------------------------------------------
class Union(UnionMixin, base.Shape3D):
    pass


class Intersection(IntersectionMixin, base.Shape3D):
    pass


class Subtraction(SubtractionMixin, base.Shape3D):
    pass


class Offset(OffsetMixin, base.Shape3D):
    pass


class Shell(ShellMixin, base.Shape3D):
    pass


class Transformation(TransformationMixin, base.Shape3D):
    def bounding_box(self):
        return transformed_box(self, self.s.bounding_box())


class Mirror(MirrorMixin, base.Shape3D):
    pass


class Symmetrical(SymmetricalMixin, base.Shape3D):
    pass


# ---- repetition (valid only under unchecked preconditions) ----------------------------------
class _RepetitionMixin:
    """Infinite repetition along the axes with finite spacing.

    Valid when the repeated shape is symmetric about the planes through the origin
    perpendicular to each repeated axis and smaller than the spacing.  "No repetition on
    this axis" (spacing 0 or None) is encoded as an infinite spacing in the tape
    (reference unsafe.py:29-31); the device op returns the coordinate unchanged for it.
    """

    def __init__(self, s, spacing):
        self.check_dimension(s)
        self.s = s
        self.spacing = util.Vector(*(float("inf") if (v is None or v == 0) else v for v in spacing))
        if self.dimension() == 2 and self.spacing[2] != float("inf"):
            raise ValueError("Attempting repetition along Z axis for 2D shape")

    def bounding_box(self):
        # reference unsafe.py:36-45: every axis is reported unbounded
        inf = float("inf")
        return util.BoundingBox(util.Vector(-inf, -inf, -inf), util.Vector(inf, inf, inf))

    def feature_size(self):
        return min(self.s.feature_size(), self.spacing.min())

    def get_node(self, point, cache):
        return self.s.get_node(cache.make_node("repetition", list(self.spacing), [point]), cache)


class Repetition2D(_RepetitionMixin, base.Shape2D):
    pass


class Repetition(_RepetitionMixin, base.Shape3D):
    pass


class _CircularRepetitionMixin:
    """n copies rotated about the z axis at regular angles."""

    def __init__(self, s, n):
        self.check_dimension(s)
        self.s, self.n = s, n

    def bounding_box(self):
        v = util.Vector.splat(self.s.bounding_box().b.x)
        return util.BoundingBox(-v, v)

    def feature_size(self):
        return self.s.feature_size() / 2  # crude: features shrink towards the axis

    def get_node(self, point, cache):
        pi_over_n = math.pi / self.n
        sector_point = cache.make_node("circular_repetition_to", [pi_over_n], [point])
        inner = self.s.get_node(sector_point, cache)
        return cache.make_node("circular_repetition_from", [pi_over_n], [inner, point])


class CircularRepetition2D(_CircularRepetitionMixin, base.Shape2D):
    pass


class CircularRepetition(_CircularRepetitionMixin, base.Shape3D):
    pass


class Flatten(base.Shape2D):
    """The z = 0 slice of a 3D shape used as a 2D shape (directions are not corrected)."""

    def __init__(self, s):
        self.check_dimension(s, required=3)
        self.s = s

    def bounding_box(self):
        return self.s.bounding_box()

    def feature_size(self):
        return self.s.feature_size()

    def get_node(self, point, cache):
        return self.s.get_node(point, cache)


# ---- gears ----------------------------------------------------------------------------------
class InvoluteGear(Union2D):
    """External gear, or (internal=True) the negative of an internal gear.

    = (involute profile scaled to the pitch radius, offset by -backlash, clipped by the
    tip circle) united with the root circle.  Attributes as in the reference:
    n, module, addendum_modules, dedendum_modules, pressure_angle, backlash, clearance,
    internal, pitch_diameter, root_diameter and outside_diameter / inside_diameter.
    """

    def __init__(self, n, module, addendum_modules=1, dedendum_modules=1, pressure_angle=20,
                 backlash=0, clearance=0, internal=False):
        self.n, self.module = n, module
        self.addendum_modules, self.dedendum_modules = addendum_modules, dedendum_modules
        self.pressure_angle, self.backlash = pressure_angle, backlash
        self.clearance, self.internal = clearance, internal
        self.pitch_diameter = n * module
        pitch_radius = self.pitch_diameter / 2

        if internal:
            inner = pitch_radius - addendum_modules * module
            outer = pitch_radius + dedendum_modules * module + clearance
            self.inside_diameter, self.root_diameter = inner * 2, outer * 2
            backlash = -backlash
        else:
            inner = pitch_radius - dedendum_modules * module - clearance
            outer = pitch_radius + addendum_modules * module
            self.outside_diameter, self.root_diameter = outer * 2, inner * 2

        profile = InvoluteGearBase(n, pressure_angle).scaled(pitch_radius)
        if backlash != 0:
            profile = profile.offset(-backlash)
        super().__init__([profile & Circle(r=outer), Circle(r=inner)])


# ---------------------------------------------------------------------------------------------
# n-ary constructors (exported by the package)
# ---------------------------------------------------------------------------------------------
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
    return _group(shapes, "Union", Union2D, Union, r)


def intersection(shapes, r=-1):
    """Intersection; r >= 0 rounds the seams with that radius."""
    return _group(shapes, "Intersection", Intersection2D, Intersection, r)
