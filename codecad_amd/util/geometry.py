"""The geometric value types under their reference module name (reference util/geometry.py):
Vector (vector.py), BoundingBox (bounding_box.py), Quaternion and Transformation (quaternion.py)."""
from .vector import Vector, FLOAT2, FLOAT4  # noqa: F401
from .bounding_box import BoundingBox  # noqa: F401
from .quaternion import Quaternion, Transformation  # noqa: F401
