"""hip_util: the MI355X counterpart of the reference's `cl_util` package.

`from codecad_amd import hip_util` gives `manager` (context/queue/kernels singleton),
`Buffer`, `BufferList`, `Tape`, `Event`, `interleave`, `interleave2`, `mem_flags` --
the same names the reference's drivers use from `cl_util` (reference cl_util/__init__.py:1-5),
implemented over the C ABI in include/hip_util.h via ctypes.
"""
from .manager import instance as manager, HipManager, HipError, Event, Stream, check  # noqa: F401
from .buffer import Buffer, BufferList, Tape, interleave, interleave2, mem_flags, map_flags  # noqa: F401
from ._lib import SPEC_DENSE, SPEC_BLOCKS, SPEC_CLASSIFY, SPEC_RENDER, SPEC_ALL, SPEC_KERNELS  # noqa: F401  (hu_spec_group bits)
from . import _lib  # noqa: F401
from .builder import build  # noqa: F401


def format_c_string_literal(s):
    """C string literal for arbitrary text (reference cl_util/codegen.py:7-43); kept for
    callers that generate source, unused by the ahead-of-time HIP path."""
    out = ['"']
    for b in s.encode("utf-8"):
        ch = chr(b)
        if ch in '\\"':
            out.append("\\" + ch)
        elif 32 <= b < 127 and ch != "?":
            out.append(ch)
        else:
            out.append("\\%03o" % b)
    out.append('"')
    return "".join(out)
