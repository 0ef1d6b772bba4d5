"""Involute gears (reference module shapes/gears.py)."""
from .primitives import InvoluteGearBase  # noqa: F401
from .combinators import InvoluteGear  # noqa: F401
