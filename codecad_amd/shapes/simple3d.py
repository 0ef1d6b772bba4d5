"""Public names of the 3D shape classes (reference module shapes/simple3d.py)."""
from .primitives import Sphere, HalfSpace, Extrusion, Revolution  # noqa: F401
from .combinators import (Union, Intersection, Subtraction, Offset, Shell, Transformation, Mirror,  # noqa: F401
                          Symmetrical)

# what ShapeBase's operators instantiate for a 3D shape
UNION, INTERSECTION, SUBTRACTION = Union, Intersection, Subtraction
OFFSET, SHELL, MIRROR, SYMMETRICAL = Offset, Shell, Mirror, Symmetrical
