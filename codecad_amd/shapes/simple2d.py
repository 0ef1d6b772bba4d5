"""Public names of the 2D shape classes (reference module shapes/simple2d.py)."""
from .primitives import Rectangle, Circle, HalfPlane, RegularPolygon2D  # noqa: F401
from .combinators import (Union2D, Intersection2D, Subtraction2D, Offset2D, Shell2D, Transformation2D,  # noqa: F401
                          Mirror2D, Symmetrical2D)

# what ShapeBase's operators instantiate for a 2D shape
UNION, INTERSECTION, SUBTRACTION = Union2D, Intersection2D, Subtraction2D
OFFSET, SHELL, MIRROR, SYMMETRICAL = Offset2D, Shell2D, Mirror2D, Symmetrical2D
