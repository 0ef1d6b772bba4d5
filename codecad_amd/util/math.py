"""Small numeric helpers (reference util/math.py:4-34)."""
import math as _math


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
