"""Vector / BoundingBox / Quaternion / Transformation in Python doubles.

API and numerical conventions follow reference util/geometry.py:8-292: all values are
Python floats (fp64); they are rounded to fp32 exactly once, when a tape is encoded
(nodes/program.py) or when a corner is handed to a kernel (`Vector.as_float4`,
reference geometry.py:98-99).  A rotation+uniform scale is ONE quaternion whose squared
norm is the scale factor (reference geometry.py:186-196).
"""
import itertools
import math
import operator
from typing import NamedTuple

import numpy

FLOAT4 = numpy.dtype([("x", numpy.float32), ("y", numpy.float32),
                      ("z", numpy.float32), ("w", numpy.float32)])
FLOAT2 = numpy.dtype([("x", numpy.float32), ("y", numpy.float32)])


class _VectorBase(NamedTuple):
    x: float
    y: float
    z: float


def _each(op, *vectors):
    """Apply `op` component by component: the one place Vector arithmetic is spelled out."""
    return Vector(*map(op, *vectors))


class Vector(_VectorBase):
    """3-vector of Python numbers; 2D code simply leaves z = 0 (reference geometry.py:8-12).

    Immutable (a named tuple), so it hashes and compares by value: the drivers put corners
    into sets and the tests compare leaf-block corners exactly.
    """

    __slots__ = ()

    def __new__(cls, x, y, z=0):
        return super().__new__(cls, x, y, z)

    # constructors -------------------------------------------------------------------------
    @classmethod
    def splat(cls, value):
        return cls(value, value, value)

    @classmethod
    def zero(cls):
        return cls.splat(0)

    @classmethod
    def polar(cls, r, phi, rho=0):
        """Spherical coordinates in degrees: phi = longitude, rho = latitude."""
        lon, lat = math.radians(phi), math.radians(rho)
        ring = math.cos(lat)
        return cls(ring * math.cos(lon), ring * math.sin(lon), math.sin(lat)) * r

    # arithmetic: vector (+,-) vector, vector (*,/) scalar -------------------------------------
    def __add__(self, other):
        return _each(operator.add, self, other)

    def __sub__(self, other):
        return _each(operator.sub, self, other)

    def __mul__(self, k):
        return _each(lambda c: c * k, self)

    def __truediv__(self, k):
        return _each(lambda c: c / k, self)

    def __neg__(self):
        return _each(operator.neg, self)

    def __pos__(self):
        return self

    def __abs__(self):
        return math.sqrt(self.abs_squared())

    def dot(self, other):
        # left-to-right sum of the three products, like the reference, so fp64 results agree
        px, py, pz = map(operator.mul, self, other)
        return px + py + pz

    def abs_squared(self):
        return self.dot(self)

    def cross(self, o):
        (ax, ay, az), (bx, by, bz) = self, o
        return Vector(ay * bz - az * by, az * bx - ax * bz, ax * by - ay * bx)

    def normalized(self):
        return self / abs(self)

    def elementwise_abs(self):
        return _each(abs, self)

    def elementwise_mul(self, other):
        return _each(operator.mul, self, other)

    def elementwise_div(self, other):
        return _each(operator.truediv, self, other)

    def max(self, other=None):
        """Largest component, or the component-wise maximum with another vector."""
        return max(self) if other is None else _each(max, self, other)

    def min(self, other=None):
        return min(self) if other is None else _each(min, self, other)

    def applyfunc(self, f):
        return _each(f, self)

    def flattened(self):
        return Vector(self.x, self.y, 0)

    def perpendicular2d(self):
        return Vector(self.y, -self.x, self.z)

    # conversions ----------------------------------------------------------------------------
    def as_float4(self, w=0):
        """16-byte float4 kernel argument; the single fp64 -> fp32 rounding of a corner."""
        return numpy.array((self.x, self.y, self.z, w), dtype=FLOAT4)

    def as_float2(self):
        return numpy.array((self.x, self.y), dtype=FLOAT2)

    def as_tuple2(self):
        return self[:2]

    def as_matrix(self):
        """Homogeneous column vector (4x1)."""
        return numpy.array([[c] for c in self] + [[1]])

    def __str__(self):
        return "({}, {}, {})".format(*self)


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


class _QuatBase(NamedTuple):
    v: Vector
    w: float


class Quaternion(_QuatBase):
    """Rotation with uniform scale: |q|^2 is the scale (reference geometry.py:186-242)."""

    __slots__ = ()

    @classmethod
    def from_degrees(cls, axis, angle, scale=1):
        half = math.radians(angle) / 2
        m = math.sqrt(scale)
        return cls(Vector(*axis).normalized() * math.sin(half) * m, math.cos(half) * m)

    @classmethod
    def zero(cls):
        return cls(Vector.zero(), 1)

    def __mul__(self, o):
        return Quaternion(self.v * o.w + o.v * self.w + self.v.cross(o.v),
                          self.w * o.w - self.v.dot(o.v))

    def abs_squared(self):
        return self.w * self.w + self.v.abs_squared()

    def inverse(self):
        n = self.abs_squared()
        return Quaternion(-self.v / n, self.w / n)

    def conjugate(self):
        return Quaternion(-self.v, self.w)

    def transform_vector(self, p):
        # same expression as the device op (reference shapes/common.cl:1-6)
        return (self.v * self.v.dot(p) + self.v.cross(p) * self.w) * 2 + \
            p * (self.w * self.w - self.v.abs_squared())

    def as_list(self):
        return list(self.v) + [self.w]

    def as_matrix(self):
        cols = [self.transform_vector(Vector(*e)).as_matrix()
                for e in ((1, 0, 0), (0, 1, 0), (0, 0, 1))]
        return numpy.hstack(cols + [[[0], [0], [0], [1]]])


class _XformBase(NamedTuple):
    quaternion: Quaternion
    offset: Vector


class Transformation(_XformBase):
    """p -> q.transform_vector(p) + offset (reference geometry.py:245-292)."""

    __slots__ = ()

    @classmethod
    def from_degrees(cls, axis, angle, scale, offset):
        return cls(Quaternion.from_degrees(axis, angle, scale), Vector(*offset))

    @classmethod
    def zero(cls):
        return cls(Quaternion.zero(), Vector.zero())

    def __mul__(self, first):
        """`second * first`: apply `first`, then self."""
        return Transformation(self.quaternion * first.quaternion,
                              self.offset + self.quaternion.transform_vector(first.offset))

    def inverse(self):
        qi = self.quaternion.inverse()
        return Transformation(qi, -qi.transform_vector(self.offset))

    def transform_vector(self, p):
        return self.quaternion.transform_vector(p) + self.offset

    def as_list(self):
        """Tape parameters [qx, qy, qz, qw, ox, oy, oz]."""
        return self.quaternion.as_list() + list(self.offset)

    def as_matrix(self):
        m = self.quaternion.as_matrix()
        m[0, 3], m[1, 3], m[2, 3] = self.offset.x, self.offset.y, self.offset.z
        return m

    def is_2d(self):
        q = self.quaternion
        return q.v.x == 0 and q.v.y == 0 and self.offset.z == 0
