"""Host-only stand-in for the `flags` (py-flags) import name; see pyopencl/__init__.py.
Only what the reference touches at import time: class iteration, int(), no_flags,
to_simple_str()."""


class _Flag(int):
    _name = ""

    def to_simple_str(self):
        return self._name


class _Meta(type):
    def __new__(mcs, name, bases, ns):
        members = []
        bit = 1
        clean = {}
        for k, v in ns.items():
            if not k.startswith("_") and v == ():
                f = _Flag(bit)
                f._name = k
                bit <<= 1
                members.append(f)
                clean[k] = f
            else:
                clean[k] = v
        cls = super().__new__(mcs, name, bases, clean)
        cls._members = members
        cls.no_flags = _Flag(0)
        return cls

    def __iter__(cls):
        return iter(cls._members)


class Flags(metaclass=_Meta):
    pass
