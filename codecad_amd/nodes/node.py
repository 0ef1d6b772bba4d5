"""Computation-graph node and the opcode table of the instruction tape.

The opcode numbering, parameter counts and arities are the tape FORMAT and must equal the
reference's (reference nodes/node.py:12-56): a tape compiled by either implementation
runs on either interpreter.  `csrc/tape.hpp` holds the same table for the device side;
tests/test_nodes.py checks the two against each other and against the golden tapes.
"""

VARIABLE_COUNT = object()  # parameter count carried in the tape itself (polygon2d)

# name -> (parameter count, arity); the position in this list is the opcode.
_OPS = [
    ("_return", 0, 1), ("_store", 0, 1), ("_load", 0, 1),
    ("rectangle", 2, 1), ("circle", 1, 1), ("regular_polygon2d", 2, 1),
    ("polygon2d", VARIABLE_COUNT, 1),
    ("sphere", 1, 1), ("half_space", 0, 1), ("revolution_to", 0, 1),
    ("twist_revolution_to", 2, 1),
    ("initial_transformation_to", 7, 0), ("transformation_to", 7, 1),
    ("transformation_from", 4, 1), ("mirror", 0, 1), ("symmetrical_to", 0, 1),
    ("offset", 1, 1), ("shell", 1, 1),
    ("repetition", 3, 1), ("circular_repetition_to", 1, 1),
    ("circular_repetition_from", 1, 2), ("involute_gear", 2, 1),
    ("extrusion", 1, 2), ("revolution_from", 0, 2), ("twist_revolution_from", 3, 2),
    ("symmetrical_from", 0, 2),
    ("union", 1, 2), ("intersection", 1, 2), ("subtraction", 1, 2),
]

MAX_REGISTER_COUNT = 512  # instruction word = opcode * 512 + register (reference nodes/__init__.py:6)


class Node:
    """One operation of the CSG evaluation graph.

    `dependencies[0]` is the value that travels in the interpreter's accumulator
    (`lastValue`), `dependencies[1]` (binary ops) is read from a value register.  A node
    with more than two dependencies (union / intersection of many shapes) is commutative
    and associative and is split into a chain of binary nodes by the scheduler.
    Hash/equality are structural so that `NodeCache` can merge common subexpressions.
    """

    node_types = {name: (params, arity, code) for code, (name, params, arity) in enumerate(_OPS)}

    __slots__ = ("name", "params", "dependencies", "extra_data", "_hash")

    def __init__(self, name, params, dependencies, extra_data=None):
        n_params, arity, _ = self.node_types[name]
        self.name = name
        self.params = tuple(params)
        self.dependencies = tuple(dependencies)
        self.extra_data = extra_data
        if n_params is not VARIABLE_COUNT and len(self.params) != n_params:
            raise ValueError("{} takes {} parameters, got {}".format(name, n_params, len(self.params)))
        n_deps = len(self.dependencies)
        if not (n_deps == arity or (arity == 2 and n_deps > 2)):
            raise ValueError("{} takes {} inputs, got {}".format(name, arity, n_deps))
        self._hash = hash((name, self.params, tuple(id(d) for d in self.dependencies)))

    def __hash__(self):
        return self._hash

    def __eq__(self, other):
        # dependencies are compared by identity: they are already canonical (they came out
        # of the same NodeCache), which keeps equality O(1) instead of O(subtree)
        return (isinstance(other, Node) and self.name == other.name and self.params == other.params
                and len(self.dependencies) == len(other.dependencies)
                and all(a is b for a, b in zip(self.dependencies, other.dependencies)))

    def __repr__(self):
        return "Node({!r}, {!r}, {} deps)".format(self.name, self.params, len(self.dependencies))
