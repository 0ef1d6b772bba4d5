"""Host-side math types shared by shapes, the tape compiler and the drivers.

Same public names as the reference's `codecad.util` (reference util/__init__.py:1-4).
"""
from .geometry import Vector, BoundingBox, Quaternion, Transformation  # noqa: F401
from .math import KahanSummation, round_up_to, round_up_to_power_of_2, clamp  # noqa: F401
from .misc import status_block, Concatenate, at_most_one  # noqa: F401
from .types import wrap_number_like, wrap_vector_like  # noqa: F401
from . import types  # noqa: F401
