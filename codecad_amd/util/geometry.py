"""Vector / BoundingBox / Quaternion / Transformation in Python doubles.

API and numerical conventions follow reference util/geometry.py:8-292: all values are
Python floats (fp64); they are rounded to fp32 exactly once, when a tape is encoded
(nodes/program.py) or when a corner is handed to a kernel (`Vector.as_float4`,
reference geometry.py:98-99).  A rotation+uniform scale is ONE quaternion whose squared
norm is the scale factor (reference geometry.py:186-196).
"""
import itertools
import math
from typing import NamedTuple

import numpy

FLOAT4 = numpy.dtype([("x", numpy.float32), ("y", numpy.float32),
                      ("z", numpy.float32), ("w", numpy.float32)])
FLOAT2 = numpy.dtype([("x", numpy.float32), ("y", numpy.float32)])


class _VectorBase(NamedTuple):
    x: float
    y: float
    z: float


class Vector(_VectorBase):
    """3-vector; 2D code simply leaves z = 0 (reference geometry.py:8-12)."""

    __slots__ = ()

    def __new__(cls, x, y, z=0):
        return super().__new__(cls, x, y, z)

    @classmethod
    def splat(cls, value):
        return cls(value, value, value)

    @classmethod
    def zero(cls):
        return cls(0, 0, 0)

    @classmethod
    def polar(cls, r, phi, rho=0):
        """Spherical coordinates in degrees: phi = longitude, rho = latitude."""
        phi, rho = math.radians(phi), math.radians(rho)
        c = math.cos(rho)
        return cls(c * math.cos(phi), c * math.sin(phi), math.sin(rho)) * r

    # arithmetic ---------------------------------------------------------------------
    def __add__(self, o):
        return Vector(self.x + o.x, self.y + o.y, self.z + o.z)

    def __sub__(self, o):
        return Vector(self.x - o.x, self.y - o.y, self.z - o.z)

    def __mul__(self, k):
        return Vector(self.x * k, self.y * k, self.z * k)

    def __truediv__(self, k):
        return Vector(self.x / k, self.y / k, self.z / k)

    def __neg__(self):
        return Vector(-self.x, -self.y, -self.z)

    def __pos__(self):
        return self

    def __abs__(self):
        return math.sqrt(self.abs_squared())

    def abs_squared(self):
        return self.dot(self)

    def dot(self, o):
        return self.x * o.x + self.y * o.y + self.z * o.z

    def cross(self, o):
        return Vector(self.y * o.z - self.z * o.y,
                      self.z * o.x - self.x * o.z,
                      self.x * o.y - self.y * o.x)

    def normalized(self):
        return self / abs(self)

    def elementwise_abs(self):
        return Vector(abs(self.x), abs(self.y), abs(self.z))

    def elementwise_mul(self, o):
        return Vector(self.x * o.x, self.y * o.y, self.z * o.z)

    def elementwise_div(self, o):
        return Vector(self.x / o.x, self.y / o.y, self.z / o.z)

    def _fold(self, other, op):
        if other is None:
            return op(self.x, self.y, self.z)
        return Vector(op(self.x, other.x), op(self.y, other.y), op(self.z, other.z))

    def max(self, other=None):
        return self._fold(other, max)

    def min(self, other=None):
        return self._fold(other, min)

    def applyfunc(self, f):
        return Vector(f(self.x), f(self.y), f(self.z))

    def flattened(self):
        return Vector(self.x, self.y, 0)

    def perpendicular2d(self):
        return Vector(self.y, -self.x, self.z)

    # conversions --------------------------------------------------------------------
    def as_float4(self, w=0):
        """16-byte float4 kernel argument; the single fp64 -> fp32 rounding of a corner."""
        return numpy.array((self.x, self.y, self.z, w), dtype=FLOAT4)

    def as_float2(self):
        return numpy.array((self.x, self.y), dtype=FLOAT2)

    def as_tuple2(self):
        return (self.x, self.y)

    def as_matrix(self):
        return numpy.array([[self.x], [self.y], [self.z], [1]])

    def __str__(self):
        return "({}, {}, {})".format(self.x, self.y, self.z)


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
