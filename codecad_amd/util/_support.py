"""Small host-side helpers behind `codecad_amd.util`: compensated summation, integer rounding, a timing
context manager, iterable helpers and the coercions the shape constructors use for their arguments.
The names are the reference's (`util/math.py`, `util/misc.py`, `util/types.py` are where its shape code looks
them up); the implementations are this package's own."""
import contextlib
import itertools
import numbers
import sys
import time


class KahanSummation:
    """Running sum with a carried low-order part (Neumaier's variant of compensated summation: the lost bits
    of whichever operand is smaller are kept, so a term larger than the running sum is handled too).
    `acc += x`, `acc -= x`; the sum is `acc.result`."""

    __slots__ = ("_high", "_low")

    def __init__(self, start=0):
        self._high = start
        self._low = 0

    @property
    def result(self):
        return self._high + self._low

    def __iadd__(self, term):
        total = self._high + term
        if abs(self._high) >= abs(term):
            self._low += (self._high - total) + term
        else:
            self._low += (term - total) + self._high
        self._high = total
        return self

    def __isub__(self, term):
        return self.__iadd__(-term)


def round_up_to(value, multiple):
    """The smallest multiple of `multiple` not below `value` (integers)."""
    return value + (-value) % multiple


def round_up_to_power_of_2(value):
    """The smallest power of two not below `value` (exact for integers; the reference goes through log2)."""
    if value != int(value):
        power = 1.0
        while power < value:
            power *= 2
        while power / 2 >= value:
            power /= 2
        return int(power) if power >= 1 else power
    return 1 if value <= 1 else 1 << (int(value) - 1).bit_length()


def clamp(value, lower, upper):
    if value < lower:
        return lower
    return upper if value > upper else value


@contextlib.contextmanager
def status_block(title, stream=None):
    """`title... 0.12 s` around a block of work."""
    out = stream or sys.stdout
    out.write("%s..." % title)
    out.flush()
    started = time.perf_counter()
    try:
        yield
    finally:
        out.write(" %.2f s\n" % (time.perf_counter() - started))
        out.flush()


class Concatenate:
    """Several iterables presented as one that can be walked repeatedly (a tape's variable-length parameters
    are measured first and written afterwards)."""

    def __init__(self, *parts):
        self._parts = parts

    def __iter__(self):
        return itertools.chain.from_iterable(self._parts)


def at_most_one(flags):
    """No more than one of `flags` is true."""
    return sum(1 for flag in flags if flag) <= 1


def wrap_number_like(value):
    """Numbers pass through; anything float()-able becomes a float; else TypeError."""
    if isinstance(value, numbers.Number):
        return value
    try:
        return float(value)
    except (TypeError, ValueError):
        raise TypeError("Value must be a number or convertible to float to be number-like")


def wrap_vector_like(value, max_dimension=3):
    """A Vector passes through; an iterable of 2..max_dimension numbers becomes a Vector."""
    if isinstance(value, _vector()):
        return value
    try:
        it = iter(value)
    except TypeError:
        raise TypeError("Value must be iterable to be vector-like")

    items = []
    for raw in it:
        if len(items) == max_dimension:
            # do not convert the surplus item: "too long" must win over "not a number"
            raise TypeError("Value must have at most three items to be vector-like")
        items.append(wrap_number_like(raw))
    if len(items) < 2:
        raise TypeError("Value must have at least two items to be vector-like")
    return _vector()(*items)


def _vector():
    from .geometry import Vector
    return Vector
