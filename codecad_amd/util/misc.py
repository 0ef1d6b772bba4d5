"""Status printing and iterable helpers (reference util/misc.py:6-31)."""
import contextlib
import sys
import time


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
