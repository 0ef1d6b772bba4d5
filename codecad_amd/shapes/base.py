"""Shape base classes: the user-facing CSG methods (reference shapes/base.py:9-446).

A shape knows three things: its `bounding_box()`, a `feature_size()` estimate and how to
append itself to the evaluation DAG (`get_node(point, cache)`).  Everything else on this
page is sugar that builds other shapes (translated / rotated / scaled / mirrored /
symmetrical / offset / shell / extruded / revolved, and the operators `& + - | ^`).
The method names and argument conventions are the reference's so model scripts run
unchanged.
"""
import abc

from .. import util

_AXES = {"x": (1, 0, 0), "y": (0, 1, 0), "z": (0, 0, 1)}


class ShapeBase(abc.ABC):
    """Common protocol of 2D and 3D shapes."""

    @abc.abstractmethod
    def bounding_box(self):
        """A box that contains the whole shape."""

    @abc.abstractmethod
    def feature_size(self):
        """Estimate of the smallest feature; sampling at feature_size/2 loses nothing
        (a heuristic, see reference shapes/base.py:19-41)."""

    @abc.abstractmethod
    def get_node(self, point, cache):
        """Append this shape to the DAG; `point` is the node producing the sample point.
        Create nodes with `cache.make_node` so common subexpressions merge."""

    @staticmethod
    @abc.abstractmethod
    def dimension():
        """2 or 3."""

    # implemented by Shape2D / Shape3D through the class registry ------------------------
    def _family(self):
        from . import simple2d, simple3d
        return simple2d if self.dimension() == 2 else simple3d

    def __and__(self, other):
        return self._family().INTERSECTION([self, other])

    def __add__(self, other):
        return self._family().UNION([self, other])

    def __sub__(self, other):
        return self._family().SUBTRACTION(self, other)

    def __or__(self, other):
        return self.__add__(other)

    def __xor__(self, other):
        return (self - other) | (other - self)

    def offset(self, d):
        """Grow (d > 0) or shrink the shape by a distance."""
        return self._family().OFFSET(self, d)

    def shell(self, wall_thickness):
        """Shell of the given thickness centred on the surface."""
        return self._family().SHELL(self, wall_thickness)

    def mirrored_x(self):
        return self._family().MIRROR(self)

    def symmetrical_x(self):
        """Replace the half x < 0 by the mirror image of the half x > 0."""
        return self._family().SYMMETRICAL(self)

    def check_dimension(self, *shapes, required=None):
        required = self.dimension() if required is None else required
        for s in (shapes or (self,)):
            if s.dimension() != required:
                raise TypeError("Shape must be of dimension {}, but is {}".format(required, s.dimension()))

    def shape(self):
        """Return self (assemblies expose the same method)."""
        return self


class Shape2D(ShapeBase):
    """A shape in the xy plane."""

    @staticmethod
    def dimension():
        return 2

    def _with(self, quaternion, offset):
        from . import simple2d
        return simple2d.Transformation2D(self, quaternion, offset)

    def translated(self, x, y=None):
        v = util.wrap_vector_like(x) if y is None else util.Vector(x, y)
        return self._with(util.Quaternion.from_degrees((0, 0, 1), 0), v)

    def translated_x(self, distance):
        return self.translated(distance, 0)

    def translated_y(self, distance):
        return self.translated(0, distance)

    def rotated(self, angle, n=1):
        """Rotate by `angle` degrees; n > 1 gives the union of n copies at angle*i/n."""
        if n == 1:
            return self._with(util.Quaternion.from_degrees((0, 0, 1), angle), util.Vector(0, 0, 0))
        from . import simple2d
        return simple2d.Union2D([self.rotated((1 + i) * angle / n) for i in range(n)])

    def scaled(self, s):
        return self._with(util.Quaternion.from_degrees((0, 0, 1), 0, s), util.Vector(0, 0, 0))

    def transformed(self, transformation):
        if not transformation.is_2d():
            raise ValueError("Transformation needs to be 2D only")
        return self._with(transformation.quaternion, transformation.offset)

    def mirrored_y(self):
        return self.rotated(180).mirrored_x()

    def symmetrical_y(self):
        return self.rotated(-90).symmetrical_x().rotated(90)

    def extruded(self, height, symmetrical=True):
        """Extrude along z; symmetrical about z = 0 unless told otherwise."""
        from . import simple3d
        solid = simple3d.Extrusion(self, height)
        return solid if symmetrical else solid.translated(0, 0, height / 2)

    def revolved(self, r=0, twist=0):
        """Revolve the x > 0 half around the y axis, optionally twisting (degrees)."""
        from . import simple3d
        return simple3d.Revolution(self, r, twist)


class Shape3D(ShapeBase):
    """A solid."""

    @staticmethod
    def dimension():
        return 3

    def _with(self, quaternion, offset):
        from . import simple3d
        return simple3d.Transformation(self, quaternion, offset)

    def translated(self, x, y=None, z=None):
        if y is None and z is None:
            v = util.wrap_vector_like(x)
        elif y is not None and z is not None:
            v = util.Vector(x, y, z)
        else:
            raise ValueError("If y is specified, then z has to be too.")
        return self._with(util.Quaternion.from_degrees((0, 0, 1), 0), v)

    def translated_x(self, distance):
        return self.translated(distance, 0, 0)

    def translated_y(self, distance):
        return self.translated(0, distance, 0)

    def translated_z(self, distance):
        return self.translated(0, 0, distance)

    def rotated(self, axis, angle, n=1):
        """Rotate by `angle` degrees about `axis`; n > 1 as for Shape2D.rotated."""
        if n == 1:
            return self._with(util.Quaternion.from_degrees(util.wrap_vector_like(axis), angle),
                              util.Vector(0, 0, 0))
        from . import simple3d
        return simple3d.Union([self.rotated(axis, (1 + i) * angle / n) for i in range(n)])

    def rotated_x(self, angle):
        return self.rotated(_AXES["x"], angle)

    def rotated_y(self, angle):
        return self.rotated(_AXES["y"], angle)

    def rotated_z(self, angle):
        return self.rotated(_AXES["z"], angle)

    def scaled(self, s):
        return self._with(util.Quaternion.from_degrees((0, 0, 1), 0, s), util.Vector(0, 0, 0))

    def transformed(self, transformation):
        return self._with(transformation.quaternion, transformation.offset)

    def mirrored_y(self):
        return self.rotated_z(180).mirrored_x()

    def mirrored_z(self):
        return self.rotated_y(180).mirrored_x()

    def symmetrical_y(self):
        return self.rotated_z(-90).symmetrical_x().rotated_z(90)

    def symmetrical_z(self):
        return self.rotated_y(90).symmetrical_x().rotated_y(-90)


class TapeShape(Shape3D):
    """A pre-compiled instruction tape with its metadata, usable wherever a shape is.

    Lets a tape produced elsewhere (a golden fixture, a tape saved by another process or
    compiled by the reference) be fed to grid_eval / subdivision / mass_properties.
    """

    def __init__(self, tape, bounding_box, feature_size=None, dimension=3):
        import numpy
        self.raw_tape = numpy.ascontiguousarray(tape, dtype=numpy.float32)
        self._box = bounding_box
        self._feature_size = feature_size
        self._dimension = dimension

    def dimension(self):  # instance-level: a tape may describe a 2D shape
        return self._dimension

    def bounding_box(self):
        return self._box

    def feature_size(self):
        return self._feature_size

    def get_node(self, point, cache):
        raise TypeError("a TapeShape is already compiled and cannot be combined with other shapes")
