"""The benchmark shapes of BASELINE.json, written against this package's own API.

These are model definitions (data), restated from what the reference's example scripts
build: a Menger sponge (reference examples/menger_sponge.py:10-28) and the CSG tree of the
Wikipedia CSG figure (reference examples/csg_example.py:8-14).
"""
from . import shapes


def sponge(iteration):
    """Menger sponge of the unit cube after `iteration` rounds of hole punching.

    Round i removes an infinite 3-axis cross of square bars of side 3^-(i+1), repeated with
    period 3^-i in all three directions.
    """
    if iteration < 0:
        raise ValueError("Iteration must be positive or zero")
    cube = shapes.box()
    if iteration == 0:
        return cube
    bar = shapes.box(1 / 3, 1 / 3, float("inf"))
    cross = bar + bar.rotated_x(90) + bar.rotated_y(90)
    holes = shapes.union(shapes.unsafe.Repetition(cross.scaled(s), (s, s, s))
                         for s in ((1 / 3) ** i for i in range(iteration)))
    return cube - holes


def csg_example():
    """sphere(130) & (box(100) - three orthogonal cylinders d=40)."""
    cyl = shapes.cylinder(d=40, h=200)
    holes = cyl + cyl.rotated_x(90) + cyl.rotated_y(90)
    return shapes.sphere(130) & shapes.box(100) - holes


def sphere_plus_box():
    return shapes.sphere(130) + shapes.box(100)


def planetary():
    """The planetary gearbox assembly of BASELINE config C4 (reference examples/planetary.py:362-556,
    `Planetary(...).shape()`: 560 lines of model code) as its captured instruction tape -- 1,424 floats, 467
    instructions: 58 circles, 9 involute gears, 35 extrusions under 79 unions / intersections / subtractions --
    with its bounding box and feature size (codecad_amd/data/planetary.json, the same data as the golden fixture
    tests/golden/ref_tapes.json)."""
    import json
    import os
    import numpy
    from . import util
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "planetary.json")
    with open(path) as f:
        g = json.load(f)
    tape = numpy.array(g["tape_u32"], dtype=numpy.uint32).view(numpy.float32)
    box = util.BoundingBox(util.Vector(*[float(v) for v in g["bbox_a"]]), util.Vector(*[float(v) for v in g["bbox_b"]]))
    return shapes.TapeShape(tape, box, float(g["feature_size"]))
