"""Public names of the dimension-independent combinators (reference module shapes/common.py)."""
from .combinators import (UnionMixin, IntersectionMixin, SubtractionMixin, TransformationMixin, MirrorMixin,  # noqa: F401
                          SymmetricalMixin, OffsetMixin, ShellMixin, transformed_box)
