"""Operations that are only valid under unchecked preconditions (reference module shapes/unsafe.py)."""
from .combinators import Repetition2D, Repetition, CircularRepetition2D, CircularRepetition, Flatten  # noqa: F401
