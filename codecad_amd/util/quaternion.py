"""Quaternion and Transformation (API of reference util/geometry.py:183-292).  A rotation + uniform
scale is ONE quaternion whose squared norm is the scale factor (reference geometry.py:186-196)."""
import math
from typing import NamedTuple

import numpy

from .vector import Vector


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
