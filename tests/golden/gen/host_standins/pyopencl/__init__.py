"""Host-only stand-in for the `pyopencl` *import name*, used ONLY by
tests/golden/gen/make_golden.py in the build container.

The reference package (bluecube/codecad, mounted read-only at /root/reference) does
`import pyopencl` at module scope in every file; pyopencl is not installed in this
image and cannot be fetched (no network).  The fixtures we generate come from the
reference's PURE-PYTHON half only (tape compiler, block-size calculation, bounding
boxes); none of that code calls into OpenCL.  This module provides just enough
names for those imports to succeed.  It implements NO OpenCL functionality: every
device entry point raises.  It is never imported by the product, the tests, the
benchmark, or anything that runs on the GPU box.
"""
from . import cltypes  # noqa: F401


class _NoDevice(RuntimeError):
    pass


def _no_device(*_a, **_k):
    raise _NoDevice("host-only stand-in: there is no OpenCL device or runtime here")


class _Bits(int):
    def __or__(self, other):
        return _Bits(int(self) | int(other))


class mem_flags:
    READ_WRITE = _Bits(1)
    WRITE_ONLY = _Bits(2)
    READ_ONLY = _Bits(4)
    USE_HOST_PTR = _Bits(8)
    ALLOC_HOST_PTR = _Bits(16)
    COPY_HOST_PTR = _Bits(32)
    HOST_WRITE_ONLY = _Bits(128)
    HOST_READ_ONLY = _Bits(256)
    HOST_NO_ACCESS = _Bits(512)


class map_flags:
    READ = _Bits(1)
    WRITE = _Bits(2)
    WRITE_INVALIDATE_REGION = _Bits(4)


class command_queue_properties:
    OUT_OF_ORDER_EXEC_MODE_ENABLE = _Bits(1)
    PROFILING_ENABLE = _Bits(2)


class Context:
    devices = ()


def create_some_context(*_a, **_k):
    return Context()


class CommandQueue:
    def __init__(self, context, properties=None):
        self.context = context


class Buffer:
    def __init__(self, context, flags, size=0, hostbuf=None):
        _no_device()


class Program:
    def __init__(self, *_a, **_k):
        _no_device()


class Event:
    pass


enqueue_copy = _no_device
enqueue_map_buffer = _no_device
