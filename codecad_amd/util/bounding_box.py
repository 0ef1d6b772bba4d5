"""BoundingBox: an axis-aligned box as a pair of Vectors (API of reference util/geometry.py:123-180)."""
import itertools
from typing import NamedTuple

from .vector import Vector


class _BoxBase(NamedTuple):
    a: Vector
    b: Vector


class BoundingBox(_BoxBase):
    """Axis-aligned box [a, b] (reference geometry.py:121-183)."""

    __slots__ = ()

    def vertices(self):
        for pick in itertools.product((0, 1), repeat=3):
            yield Vector(*(self[which][axis] for axis, which in enumerate(pick)))

    @classmethod
    def containing(cls, vectors):
        inf = float("inf")
        lo, hi = Vector(inf, inf, inf), Vector(-inf, -inf, -inf)
        for v in vectors:
            lo, hi = lo.min(v), hi.max(v)
        return cls(lo, hi)

    def intersection(self, other):
        lo, hi = [], []
        for a1, b1, a2, b2 in zip(self.a, self.b, other.a, other.b):
            a = max(a1, a2)
            lo.append(a)
            hi.append(max(a, min(b1, b2)))  # an empty intersection collapses to a point
        return BoundingBox(Vector(*lo), Vector(*hi))

    def union(self, other):
        return BoundingBox(self.a.min(other.a), self.b.max(other.b))

    def expanded(self, factor):
        d = self.size() * factor
        return BoundingBox(self.a - d, self.b + d)

    def expanded_additive(self, amount):
        d = Vector.splat(amount)
        return BoundingBox(self.a - d, self.b + d)

    def size(self):
        return self.b - self.a

    def midpoint(self):
        return (self.a + self.b) / 2

    def volume(self):
        s = self.size()
        return s.x * s.y * s.z

    def flattened(self):
        return BoundingBox(self.a.flattened(), self.b.flattened())

    def points(self):
        for x in (self.a.x, self.b.x):
            for y in (self.a.y, self.b.y):
                for z in (self.a.z, self.b.z):
                    yield Vector(x, y, z)

    def points2d(self):
        for x in (self.a.x, self.b.x):
            for y in (self.a.y, self.b.y):
                yield Vector(x, y)
