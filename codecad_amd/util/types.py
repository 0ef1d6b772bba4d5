"""Duck-typed argument coercion (reference util/types.py:5-59)."""
import numbers

from . import geometry


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
    if isinstance(value, geometry.Vector):
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
    return geometry.Vector(*items)
