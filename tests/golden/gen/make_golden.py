#!/usr/bin/env python3
"""Generate golden fixtures from the reference's PURE-PYTHON half (build container only).

Run (from anywhere, in the build container; /root/reference is read-only and absent on
the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/gen/make_golden.py

What it does: puts `host_standins/` (import-name stand-ins for the two third-party
modules the image lacks, `pyopencl` and `flags`; they implement no OpenCL) and
/root/reference on sys.path, imports the reference package, and records what its
host-side code computes for the hot path:

  * the float32 instruction tape  (nodes.make_program, reference nodes/program.py:74-76)
  * bounding box / feature size / dimension of each shape
  * subdivision.calculate_block_sizes tables (reference subdivision.py:116-166)

Outputs are DATA only (tests/golden/ref_tapes.json, ref_block_sizes.json).  No
reference source text is written anywhere.  The reference's OpenCL device code is NOT
run or compiled by this script (there is no OpenCL runtime in the image).
"""
import contextlib
import io
import itertools
import json
import math
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "host_standins"), REF,
                os.path.join(REF, "examples"), os.path.join(REF, "tests")]

import numpy  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    import codecad  # noqa: E402
    import codecad.subdivision  # noqa: E402
    from codecad.shapes import (box, sphere, cylinder, circle, rectangle,  # noqa: E402
                                half_space, union)
    import menger_sponge  # noqa: E402
    import csg_example  # noqa: E402
    import planetary  # noqa: E402
    import data as ref_test_data  # noqa: E402  (reference tests/data.py shape zoo)


def fnum(x):
    x = float(x)
    if math.isinf(x):
        return "inf" if x > 0 else "-inf"
    return x


def record(name, shape, group):
    random.seed(0)  # the reference scheduler shuffles (nodes/scheduler.py:165-178)
    tape = codecad.nodes.make_program(shape)
    assert tape.dtype == numpy.float32
    bb = shape.bounding_box()
    return {
        "name": name,
        "group": group,
        "dimension": shape.dimension(),
        "bbox_a": [fnum(v) for v in bb.a],
        "bbox_b": [fnum(v) for v in bb.b],
        "feature_size": fnum(shape.feature_size()),
        "tape_len": int(tape.size),
        # exact bit patterns of the float32 tape
        "tape_u32": tape.view(numpy.uint32).tolist(),
    }


def main():
    out = []

    # --- BASELINE configs -------------------------------------------------
    out.append(record("sphere_plus_box", sphere(130) + box(100), "config"))
    out.append(record("csg_example", csg_example.o, "config"))
    for n in (0, 1, 2, 3, 4, 5):
        out.append(record("sponge%d" % n, menger_sponge.sponge(n), "config"))
    with contextlib.redirect_stdout(io.StringIO()):
        p = planetary.Planetary(11, 60, 13, 41, 18, 53)
        asm = p.make_assembly()
    out.append(record("planetary", asm.shape(), "config"))

    # --- reference tests/data.py zoo (test_dsdf.py shapes) ------------------
    for k, v in sorted(ref_test_data.shapes_2d.items()):
        out.append(record(k, v, "zoo2d"))
    for k, v in sorted(ref_test_data.shapes_3d.items()):
        out.append(record(k, v, "zoo3d"))

    # --- reference tests/test_mass_properties.py shapes ---------------------
    mp = {
        "unit_box": box(1),
        "cylinder": cylinder(h=2, r=4, symmetrical=False),
        "sphere": sphere(d=2),
        "two_boxes": box(2).translated(-15, 0, 0) + box(2).translated(15, 0, 0),
        "hemisphere": sphere(r=2) - half_space(),
        "translated_sphere": sphere(d=2).translated(10, 11, 7),
        "translated_and_rotated_hemisphere":
            (sphere(r=2) - half_space()).translated(2, 0, 0).rotated((1, 0, 0), 90),
        "not_hammer": box(4).translated(0, 0, 2) + box(2, 2, 9).translated(0, 0, -3.5),
        "drunk_box": box(2, 3, 5).rotated((7, 11, 13), 17),
    }
    for k, v in mp.items():
        out.append(record("mp_" + k, v, "mass_properties"))

    # --- reference tests/test_subdivision.py KAT shapes ----------------------
    out.append(record("kat_box10", box(10), "subdivision_kat"))
    res, g = 0.1, 8
    diameter = g * (res * (g - 1)) - res
    out.append(record("kat_circle", circle(diameter), "subdivision_kat"))

    with open(os.path.join(HERE, "..", "ref_tapes.json"), "w") as f:
        json.dump({"generator": "tests/golden/gen/make_golden.py",
                   "seed": 0, "shapes": out}, f, indent=0, separators=(",", ":"))

    # --- calculate_block_sizes tables ----------------------------------------
    V = codecad.util.Vector
    rows = []

    def bs_row(a, b, dim, res, grid, overlap, mult):
        bb = codecad.util.BoundingBox(V(*a), V(*b))
        try:
            r = codecad.subdivision.calculate_block_sizes(bb, dim, res, grid, overlap, mult)
            result = [[c, [int(x) for x in dims]] for c, dims in r]
        except ValueError as e:
            result = "ValueError"
        rows.append({"a": list(a), "b": list(b), "dimension": dim, "resolution": res,
                     "grid_size": grid, "overlap": overlap, "multiplier": mult,
                     "result": result})

    # the 80 combinations of reference tests/test_subdivision.py:44-53
    for size in [(10, 20, 30), (16, 16, 16)]:
        for dim in (2, 3):
            for res in (1, 0.1):
                for grid, mult in [(2, 1), (2, 2), (21, 1), (256, 1), (256, 256)]:
                    for overlap in (True, False):
                        a = tuple(-s / 2 for s in size)
                        b = tuple(s / 2 for s in size)
                        bs_row(a, b, dim, res, grid, overlap, mult)
    # BASELINE configs
    for n_res, grids in [(512, (8, 16, 128)), (2048, (8, 16, 128)), (256, (8, 16, 64))]:
        res = 1.0 / n_res
        for grid in grids:
            bs_row((-0.5,) * 3, (0.5,) * 3, 3, res, grid, False, 1)
            e = res / 2
            bs_row((-0.5 - e,) * 3, (0.5 + e,) * 3, 3, res, grid, True, 1)
    bs_row((-5,) * 3, (5,) * 3, 3, 1, 4, True, 1)
    bs_row((-5.5,) * 3, (5.5,) * 3, 3, 1, 4, True, 1)
    with open(os.path.join(HERE, "..", "ref_block_sizes.json"), "w") as f:
        json.dump({"generator": "tests/golden/gen/make_golden.py", "rows": rows}, f,
                  indent=0, separators=(",", ":"))

    print("wrote", len(out), "tapes and", len(rows), "block-size rows")


if __name__ == "__main__":
    main()
