"""Seeded random CSG trees and the grids they are sampled on -- shared by the GPU differential test
(tests/test_gpu_random_shapes.py: HIP == canonical oracle, bit for bit) and the CPU test of the canonical
arithmetic against the reference's literal formulas (tests/test_literal_oracle.py)."""
import numpy as np

from codecad_amd import shapes


def random_2d(rng, depth):
    if depth == 0 or rng.random() < 0.25:
        kind = rng.choice(["rectangle", "circle", "ngon", "polygon", "gear", "capsule", "half_plane"])
        if kind == "gear":
            return shapes.gears.InvoluteGear(rng.choice([9, 12]), rng.choice([0.5, 1])).scaled(0.5)
        if kind == "capsule":
            return shapes.capsule(-1, 0, 1, rng.choice([0, 1]), 0.5)
        if kind == "half_plane":
            return shapes.half_plane().translated_y(-0.5) & shapes.circle(d=3)
        if kind == "rectangle":
            return shapes.rectangle(rng.choice([1, 2, 3.5]), rng.choice([1, 2.5, 4]))
        if kind == "circle":
            return shapes.circle(d=rng.choice([1, 2, 3]))
        if kind == "ngon":
            return shapes.regular_polygon2d(rng.choice([3, 5, 6]), d=rng.choice([2, 3]))
        return shapes.polygon2d([(0, 0), (3, 0), (3, 1), (2, 2), (3, 3), (0, 3)])
    s = random_2d(rng, depth - 1)
    op = rng.choice(["translate", "rotate", "quarter", "scale", "mirror", "symmetrical", "offset", "union", "subtract", "intersect",
                     "repeat", "circular"])
    if op == "repeat":
        return shapes.unsafe.Repetition2D(s.scaled(0.25), (rng.choice([2, 4]), rng.choice([0, 2]), None))
    if op == "circular":
        return shapes.unsafe.CircularRepetition2D(s.scaled(0.5).translated_x(3), rng.choice([3, 4, 7]))
    if op == "translate":
        return s.translated(rng.choice([0, 0.5, -1, 2]), rng.choice([0, 1, -0.25]))
    if op == "rotate":
        return s.rotated(rng.choice([17, 33.3, -71]))
    if op == "quarter":
        return s.rotated(rng.choice([90, 180, 270, -90]))
    if op == "scale":
        return s.scaled(rng.choice([0.5, 2, 3]))
    if op == "mirror":
        return s.mirrored_x() if rng.random() < 0.5 else s.mirrored_y()
    if op == "symmetrical":
        return s.translated_x(1).symmetrical_x()
    if op == "offset":
        return s.offset(rng.choice([0.25, 0.5]))
    t = random_2d(rng, depth - 1)
    r = rng.choice([-1, -1, 0.3])
    if op == "union":
        return shapes.union([s, t], r=r)
    if op == "subtract":
        return s - t.scaled(0.5)
    return shapes.intersection([s, t.scaled(2)], r=r)


def random_3d(rng, depth):
    if depth == 0 or rng.random() < 0.2:
        kind = rng.choice(["box", "sphere", "cylinder", "extrude", "revolve", "twist", "half_space"])
        if kind == "twist":
            return shapes.rectangle(1, 2).revolved(r=rng.choice([3, 4]), twist=rng.choice([90, 180, 360]))
        if kind == "half_space":
            return shapes.half_space().translated_y(-0.5) & shapes.sphere(d=3)
        if kind == "box":
            return shapes.box(rng.choice([1, 2, 3]), rng.choice([1, 2.5]), rng.choice([1, 4]))
        if kind == "sphere":
            return shapes.sphere(d=rng.choice([1, 2, 3]))
        if kind == "cylinder":
            return shapes.cylinder(h=rng.choice([1, 3]), d=rng.choice([1, 2]), symmetrical=rng.random() < 0.5)
        if kind == "extrude":
            return random_2d(rng, 2).extruded(rng.choice([1, 2]))
        return shapes.rectangle(1, 2).translated_x(2).revolved()
    s = random_3d(rng, depth - 1)
    op = rng.choice(["translate", "rotate", "quarter", "quarter", "scale", "mirror", "symmetrical", "offset", "shell",
                     "union", "subtract", "intersect", "repeat", "circular"])
    if op == "repeat":
        return shapes.unsafe.Repetition(s.scaled(0.25), (rng.choice([2, 4]), rng.choice([0, 2]), rng.choice([None, 1, 4])))
    if op == "circular":
        return shapes.unsafe.CircularRepetition(s.scaled(0.5).translated_x(3), rng.choice([3, 4, 7]))
    if op == "translate":
        return s.translated(rng.choice([0, 0.5, -1]), rng.choice([0, 1]), rng.choice([0, -0.5, 2]))
    if op == "rotate":
        return s.rotated((rng.choice([1, 0, 2]), rng.choice([1, 3]), rng.choice([0, 1])), rng.choice([15, 40, -77]))
    if op == "quarter":
        return getattr(s, rng.choice(["rotated_x", "rotated_y", "rotated_z"]))(rng.choice([90, 180, 270, -90]))
    if op == "scale":
        return s.scaled(rng.choice([0.5, 2, 3]))
    if op == "mirror":
        return rng.choice([s.mirrored_x, s.mirrored_y, s.mirrored_z])()
    if op == "symmetrical":
        return s.translated_x(1).symmetrical_x()
    if op == "offset":
        return s.offset(rng.choice([0.25, 0.5]))
    if op == "shell":
        return s.shell(0.2)
    t = random_3d(rng, depth - 1)
    r = rng.choice([-1, -1, -1, 0.3])
    if op == "union":
        return shapes.union([s, t], r=r)
    if op == "subtract":
        return s - t.scaled(0.5)
    return shapes.intersection([s, t.scaled(2)], r=r)


def grids():
    """(corner, step, dims): binary-fraction steps through the origin (exact zeros and symmetric pairs),
    and an irrational-looking one."""
    return [(np.array([-4.0, -4.0, -4.0]), np.float32(0.5), (17, 17, 17)),
            (np.array([-3.03, -2.97, -3.11]), np.float32(0.37), (16, 17, 15))]
