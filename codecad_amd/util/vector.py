"""Vector: a 3-tuple of Python numbers (fp64), rounded to fp32 exactly once, when a tape is encoded
(nodes/program.py) or when a corner is handed to a kernel (`as_float4`).  API of reference
util/geometry.py:8-120; 2D code leaves z = 0."""
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
