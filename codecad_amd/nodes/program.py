"""Shape -> float32 instruction tape (reference nodes/program.py:10-84).

Tape format: flat float32 array; one float `opcode*512 + secondaryRegister` per
instruction (exact in binary32) followed by that instruction's parameters; the last
instruction is `_return`.  This is the reference's format bit for bit, so tapes are
interchangeable with the reference's interpreter (tests/test_nodes.py runs the golden
reference tapes and our tapes through the same oracle and compares the results).
"""
import numpy

from .. import util
from . import node, scheduler


class NodeCache:
    """Hash-consing factory: structurally equal nodes become one object (CSE)."""

    def __init__(self):
        self._nodes = {}

    def make_node(self, name, params, dependencies, extra_data=None):
        candidate = node.Node(name, params, dependencies, extra_data)
        return self._nodes.setdefault(candidate, candidate)


def get_shape_nodes(shape):
    """DAG for `shape`: `_return(shape(initial_transformation_to(identity)(point)))`."""
    cache = NodeCache()
    identity = util.Transformation.zero()
    point = cache.make_node("initial_transformation_to", identity.as_list(), (), identity)
    return cache.make_node("_return", (), (shape.get_node(point, cache),))


def instruction_word(name, register):
    word = node.Node.node_types[name][2] * node.MAX_REGISTER_COUNT + register
    assert int(numpy.float32(word)) == word
    return word


def make_schedule(shape):
    """(registers_needed, [Instruction]) for a shape."""
    registers, code = scheduler.schedule(get_shape_nodes(shape))
    if registers > node.MAX_REGISTER_COUNT:
        raise ValueError("shape needs {} value registers, the tape format allows {}".format(
            registers, node.MAX_REGISTER_COUNT))
    return registers, code


def make_program(shape):
    """The float32 tape of a shape (cached on the shape object: compiling is pure)."""
    cached = getattr(shape, "_codecad_amd_tape", None)
    if cached is not None:
        return cached
    raw = getattr(shape, "raw_tape", None)
    if raw is not None:  # TapeShape: a pre-compiled tape
        tape = numpy.ascontiguousarray(raw, dtype=numpy.float32)
    else:
        _, code = make_schedule(shape)
        words = []
        for ins in code:
            words.append(instruction_word(ins.name, ins.register))
            words.extend(ins.params)
        tape = numpy.array(words, dtype=numpy.float32)
    tape.setflags(write=False)
    try:
        shape._codecad_amd_tape = tape
    except AttributeError:
        pass
    return tape


def make_program_buffer(shape):
    """Upload the tape to the current GPU -> `hip_util.Tape` (reference program.py:79-84).

    The handle is what `subdivision()` returns as element 0 and what every
    `hip_util.manager.k.<kernel>` accepts as its `scene` argument.  Needs the HIP library
    and a device; raises RuntimeError otherwise (there is no CPU fallback).
    """
    from .. import hip_util

    cached = getattr(shape, "_codecad_amd_tape_buffer", None)
    if cached is not None and cached.alive and cached.device == hip_util.manager.device:
        return cached
    buf = hip_util.Tape(make_program(shape))
    try:
        shape._codecad_amd_tape_buffer = buf
    except AttributeError:
        pass
    return buf
