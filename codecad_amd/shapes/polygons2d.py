"""Simple polygons and a turtle-style builder (reference shapes/polygons2d.py:16-239)."""
import math
import sys

import numpy

from .. import util
from . import base

_EPS = 1e-12


def _cross(a, b):
    return a[0] * b[1] - a[1] * b[0]


def _dot(a, b):
    return a[0] * b[0] + a[1] * b[1]


def _sub(a, b):
    return (a[0] - b[0], a[1] - b[1])


def _check_simple(pts):
    """Raise ValueError unless the closed polyline `pts` is a simple polygon."""
    n = len(pts)
    edges = [(pts[i - 1], pts[i]) for i in range(n)]  # edge i ends at vertex i
    for (a, b) in edges:
        if a == b:
            raise ValueError("Zero length segments are not allowed in polygon")
    for i in range(n):
        p1, p2 = edges[i]
        d = _sub(p2, p1)
        for j in range(i + 1, n):
            q1, q2 = edges[j]
            e = _sub(q2, q1)
            denom = _cross(d, e)
            parallel = abs(denom) < _EPS
            adjacent = (j == i + 1) or (i == 0 and j == n - 1)
            if adjacent:
                if parallel and _dot(d, e) < 0:
                    raise ValueError("Polygon cannot be self intersecting (anti-parallel consecutive edges)")
                continue
            between = _sub(q1, p1)
            if parallel:
                if abs(_cross(between, d)) < _EPS:  # collinear: overlapping parameter ranges?
                    dd = _dot(d, d)
                    t1 = _dot(between, d) / dd
                    t2 = _dot(_sub(q2, p1), d) / dd
                    if max(t1, t2) >= 0 and min(t1, t2) <= 1:
                        raise ValueError("Polygon cannot be self intersecting (colinear segments)")
                continue
            t = _cross(between, e) / denom
            u = _cross(between, d) / denom
            if 0 <= t <= 1 and 0 <= u <= 1:
                raise ValueError("Polygon cannot be self intersecting")


class Polygon2D(base.Shape2D):
    """Simple polygon; the last point connects back to the first.

    Vertices are rounded to float32 up front (they go into the tape verbatim) and stored
    in the winding the device op expects (`polygon2d_op` takes (-dy, dx) as the outward
    normal of each edge, reference shapes/polygons2d.cl:24): the signed sum
    sum((x_i - x_{i-1}) * (y_{i-1} + y_i)) must not be negative.
    """

    def __init__(self, points):
        arr = numpy.asarray([util.wrap_vector_like(p).as_tuple2() for p in points], dtype=numpy.float32)
        if arr.ndim != 2 or arr.shape[0] < 3:
            raise ValueError("Polygon must have at least three vertices")
        pts = [(float(x), float(y)) for x, y in arr]
        _check_simple(pts)

        n = len(pts)
        winding = math.fsum((pts[i][0] - pts[i - 1][0]) * (pts[i - 1][1] + pts[i][1]) / 2 for i in range(n))
        if winding < 0:
            arr = numpy.flipud(arr)
        self.points = numpy.ascontiguousarray(arr)

        xs, ys = [p[0] for p in pts], [p[1] for p in pts]
        self.box = util.BoundingBox(util.Vector(min(xs), min(ys), 0), util.Vector(max(xs), max(ys), 0))
        self._feature_size = min(math.hypot(pts[i][0] - pts[j][0], pts[i][1] - pts[j][1])
                                 for i in range(n) for j in range(i + 1, n))

    def bounding_box(self):
        return self.box

    def feature_size(self):
        return self._feature_size

    def get_node(self, point, cache):
        flat = [float(v) for v in self.points.flat]
        return cache.make_node("polygon2d", [len(self.points)] + flat, [point])

    @classmethod
    def build(cls, origin_x, origin_y):
        """Start a Polygon2DBuilder at the given point."""
        return Polygon2DBuilder(cls, origin_x, origin_y)


class Polygon2DBuilder:
    """Build a vertex list by relative/absolute moves; `.close()` makes the polygon."""

    def __init__(self, close_callback, x, y):
        self._close_callback = close_callback
        self.points = [(x, y)]

    def close(self):
        return self._close_callback(self.points)

    # mirroring ---------------------------------------------------------------------------
    def symmetrical_x(self, center_x):
        """Append the existing points mirrored about x = center_x, in reverse order."""
        self.points.extend((2 * center_x - px, py) for px, py in reversed(list(self.points)))
        return self

    def symmetrical_y(self, center_y):
        self.points.extend((px, 2 * center_y - py) for px, py in reversed(list(self.points)))
        return self

    # nested blocks -----------------------------------------------------------------------
    def block(self, modifier=lambda pts: pts):
        """Sub-builder starting at the last point; closing it splices modifier(points) back."""
        outer = self

        def splice(new_points):
            outer.points.extend(modifier(new_points))
            return outer

        return type(self)(splice, *self.points.pop())

    def reversed_block(self):
        return self.block(reversed)

    # moves -------------------------------------------------------------------------------
    def xy(self, x, y):
        self.points.append((x, y))
        return self

    def x(self, x):
        return self.xy(x, self.points[-1][1])

    def y(self, y):
        return self.xy(self.points[-1][0], y)

    def dxdy(self, dx, dy):
        return self.xy(self.points[-1][0] + dx, self.points[-1][1] + dy)

    def dx(self, dx):
        return self.dxdy(dx, 0)

    def dy(self, dy):
        return self.dxdy(0, dy)

    def angle(self, angle, distance):
        """Move `distance` in the absolute direction `angle` (degrees)."""
        a = math.radians(angle)
        return self.dxdy(math.cos(a) * distance, math.sin(a) * distance)

    def angle_dx(self, angle, dx):
        return self.dxdy(dx, dx * math.tan(math.radians(angle)))

    def angle_dy(self, angle, dy):
        return self.dxdy(dy / math.tan(math.radians(angle)), dy)

    def tangent_point(self, center_x, center_y, radius):
        """Move to the tangent point on a circle; the sign of radius picks the side."""
        cx, cy = center_x - self.points[-1][0], center_y - self.points[-1][1]
        l2 = cx * cx + cy * cy
        s = 1 - radius * radius / l2
        t = radius * math.sqrt(s / l2)
        return self.dxdy(cx * s - cy * t, cx * t + cy * s)

    def print(self, file=sys.stdout):
        print(str(self.points), file=file)
        return self


def polygon2d(points):
    """Simple polygon from (x, y) points, either winding; rejects self-intersections."""
    return Polygon2D(points)


def polygon2d_builder(origin_x, origin_y):
    """Turtle-style builder: `.dx(..).dy(..).angle(..).close()` -> polygon."""
    return Polygon2D.build(origin_x, origin_y)
