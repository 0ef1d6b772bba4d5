"""Small host-side helpers: compensated summation, rounding, status printing, iterable helpers and
duck-typed argument coercion (reference util/math.py, util/misc.py, util/types.py)."""
import contextlib
import math as _math
import numbers
import sys
import time


class KahanSummation:
    """Compensated running sum; `s += x` / `s -= x`, result in `.result`."""

    def __init__(self):
        self.result = 0
        self.correction = 0

    def __iadd__(self, x):
        y = x - self.correction
        t = self.result + y
        self.correction = (t - self.result) - y
        self.result = t
        return self

    def __isub__(self, x):
        self += -x
        return self


def round_up_to(x, multiple):
    """Smallest multiple of `multiple` that is >= x."""
    return ((x + multiple - 1) // multiple) * multiple


def round_up_to_power_of_2(x):
    return 2 ** _math.ceil(_math.log2(x))


def clamp(v, lower, upper):
    return max(lower, min(v, upper))


@contextlib.contextmanager
def status_block(title):
    """Print `title... <seconds> s` around a block."""
    print(title, end="...")
    sys.stdout.flush()
    t0 = time.perf_counter()
    try:
        yield
    finally:
        print(" {:0.2f} s".format(time.perf_counter() - t0))


class Concatenate:
    """Re-iterable chain of iterables (used for variable-length tape parameters)."""

    def __init__(self, *iterables):
        self._iterables = iterables

    def __iter__(self):
        for it in self._iterables:
            yield from it


def at_most_one(iterable):
    """True when at most one element is truthy."""
    it = iter(iterable)
    any(it)
    return not any(it)


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
