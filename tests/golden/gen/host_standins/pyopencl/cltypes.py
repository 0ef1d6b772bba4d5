"""numpy dtypes under the names the reference reads from `pyopencl.cltypes`.
Host-only stand-in, see __init__.py."""
import numpy

char = numpy.int8
uchar = numpy.uint8
short = numpy.int16
ushort = numpy.uint16
int = numpy.int32
uint = numpy.uint32
long = numpy.int64
ulong = numpy.uint64
half = numpy.float16
float = numpy.float32
double = numpy.float64


def _vec(scalar, n):
    names = ["x", "y", "z", "w"][:n] if n <= 4 else ["s%x" % i for i in range(n)]
    fields = [(name, scalar) for name in names]
    if n == 3:  # OpenCL 3-vectors occupy 4 elements
        fields.append(("pad", scalar))
    return numpy.dtype(fields)


for _name, _scalar in list(globals().items()):
    if isinstance(_scalar, type) and issubclass(_scalar, numpy.generic):
        for _n in (2, 3, 4, 8, 16):
            globals()["%s%d" % (_name, _n)] = _vec(_scalar, _n)
