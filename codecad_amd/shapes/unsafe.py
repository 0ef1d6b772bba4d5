"""Operations that are only valid under unchecked preconditions (reference shapes/unsafe.py)."""
import math

from .. import util
from . import base


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
