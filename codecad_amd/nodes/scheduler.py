"""Linearise a node DAG into accumulator-machine code and allocate value registers.

The target machine (reference nodes/codegen.py:5-63) has one accumulator `lastValue`,
value registers written only by `_store` and read by `_load` or as the SECOND operand of a
binary op.  So a schedule is a post-order walk in which, for every binary node, the second
operand is computed first and parked in a register, and the first operand is computed last
so that it is still in the accumulator when the node executes.  A value needs a `_store`
when it has several consumers, is consumed as a second operand, or is not consumed
immediately.  Cost = (_store + _load + register-operand reads, registers), minimised over a
few operand orderings, like the reference's randomized scheduler (nodes/scheduler.py:171-178);
each store/load is a 16-byte LDS access per lane on the GPU and the register count sets
the LDS footprint of a workgroup, so both matter on MI355X too.
"""
import random
import sys


class Instruction:
    """One tape instruction: `name`, tape `params`, and the secondary register."""

    __slots__ = ("name", "params", "register")

    def __init__(self, name, params, register):
        self.name = name
        self.params = params
        self.register = register

    def __repr__(self):
        return "{}(r{}, {})".format(self.name, self.register, list(self.params))


class _Value:
    """Scheduling state of one (binarised) node."""

    __slots__ = ("name", "params", "operands", "consumers", "register", "emitted")

    def __init__(self, name, params, operands):
        self.name = name
        self.params = params
        self.operands = operands
        self.consumers = 0
        self.register = None
        self.emitted = False


class _Registers:
    def __init__(self):
        self.live = {}       # register -> outstanding reads
        self.high_water = 0

    def allocate(self, reads):
        r = 0
        while r in self.live:
            r += 1
        self.live[r] = reads
        self.high_water = max(self.high_water, r + 1)
        return r

    def read(self, r):
        self.live[r] -= 1
        if self.live[r] == 0:
            del self.live[r]


def _binarise(root, pick_order):
    """Copy the DAG into _Value objects, splitting n-ary nodes into binary chains."""
    memo = {}

    def convert(node):
        v = memo.get(id(node))
        if v is not None:
            return v
        deps = [convert(d) for d in node.dependencies]
        if len(deps) > 2:
            deps = list(pick_order(deps))
            chain = _Value(node.name, node.params, [deps[0], deps[1]])
            for d in deps[2:-1]:
                chain = _Value(node.name, node.params, [chain, d])
            # the chain is parked in a register, the last shape is evaluated last and
            # arrives in the accumulator
            deps = [deps[-1], chain]
        v = _Value(node.name, node.params, deps)
        memo[id(node)] = v
        return v

    top = convert(root)
    seen = set()

    def count(v):
        for d in v.operands:
            d.consumers += 1
            if id(d) not in seen:
                seen.add(id(d))
                count(d)

    count(top)
    return top


def _linearise(root, pick_order):
    top = _binarise(root, pick_order)
    regs = _Registers()
    out = []
    stats = {"mem": 0}
    accumulator = [None]  # the _Value currently held in lastValue

    def emit(v, must_store):
        n = len(v.operands)
        # which operand index is evaluated when: default second-first, first-last
        order = list(pick_order(list(reversed(range(n)))))
        for pos, i in enumerate(order):
            d = v.operands[i]
            if not d.emitted:
                passes_in_accumulator = (i == 0 and pos == n - 1 and d.consumers == 1)
                emit(d, not passes_in_accumulator)

        first = v.operands[0] if n else None
        if first is not None and accumulator[0] is not first:
            out.append(Instruction("_load", (), first.register))
            stats["mem"] += 1
        # operand registers are released BEFORE the result register is chosen so that a
        # result can overwrite one of its own inputs
        for d in v.operands:
            if d.register is not None:
                regs.read(d.register)
        second_reg = v.operands[1].register if n == 2 else 0
        if n == 2:
            stats["mem"] += 1
        out.append(Instruction(v.name, v.params, second_reg))
        v.emitted = True
        accumulator[0] = v
        if must_store:
            v.register = regs.allocate(v.consumers)
            out.append(Instruction("_store", (), v.register))
            stats["mem"] += 1

    old = sys.getrecursionlimit()
    sys.setrecursionlimit(max(old, 20000))
    try:
        emit(top, False)
    finally:
        sys.setrecursionlimit(old)
    return stats["mem"], regs.high_water, out


def _identity(x):
    return x


def schedule(root, random_passes=100, seed=0):
    """Best of (as given, reversed, `random_passes` shuffles) by (memory ops, registers).

    Deterministic: the shuffles come from a private `random.Random(seed)`, so the same
    shape always compiles to the same tape (the reference shuffles with the global
    `random` state, nodes/scheduler.py:165-168).
    Returns (registers_needed, [Instruction]).
    """
    rng = random.Random(seed)

    def shuffled(x):
        x = list(x)
        rng.shuffle(x)
        return x

    best = None
    for pick in [_identity, lambda x: list(reversed(x))] + [shuffled] * random_passes:
        mem, nregs, code = _linearise(root, pick)
        if best is None or (mem, nregs) < best[:2]:
            best = (mem, nregs, code)
    return best[1], best[2]
