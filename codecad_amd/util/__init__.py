"""Host-side math types shared by shapes, the tape compiler and the drivers.

Same public names as the reference's `codecad.util` (reference util/__init__.py:1-4).
"""
import sys as _sys

from .geometry import Vector, BoundingBox, Quaternion, Transformation  # noqa: F401
from . import _support
from ._support import (KahanSummation, round_up_to, round_up_to_power_of_2, clamp, status_block,  # noqa: F401
                       Concatenate, at_most_one, wrap_number_like, wrap_vector_like)

types = _support   # `util.types.wrap_vector_like`, as in the reference
_sys.modules[__name__ + ".types"] = _support
