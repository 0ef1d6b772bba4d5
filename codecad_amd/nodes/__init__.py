"""Tape compiler: CSG shape -> node DAG -> schedule -> float32 instruction tape.

Mirrors the reference's `codecad.nodes` surface (reference nodes/__init__.py:1-6).  The
reference also generates its OpenCL interpreter from the opcode table at import; here the
interpreter is the hand-written HIP in codecad_amd/csrc/interp.hpp.
"""
from .program import make_program, make_program_buffer, make_schedule, NodeCache  # noqa: F401
from . import node  # noqa: F401
from . import scheduler  # noqa: F401
