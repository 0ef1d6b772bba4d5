#!/usr/bin/env python3
"""Wall time of polygon() (2D contouring) for a few shapes, split into device part and host stitching."""
import os
import sys
import time

os.environ.setdefault("CODECAD_AMD_CACHE", "0")   # measurements and soaks choose their evaluator themselves

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import shapes_zoo  # noqa: E402
from codecad_amd import subdivision, shapes  # noqa: E402
from codecad_amd.rendering import polygon2d  # noqa: E402

cases = {"gear": shapes_zoo.shapes_2d["gear"], "rotated_pattern_2d": shapes_zoo.shapes_2d["rotated_pattern_2d"],
         "big_gear": shapes.gears.InvoluteGear(60, 2).shape() if hasattr(shapes.gears, "InvoluteGear") else shapes_zoo.shapes_2d["gear"]}
for name, shape in cases.items():
    for rep in range(2):
        t0 = time.perf_counter()
        leaves = subdivision.subdivision_device(shape, shape.feature_size() / 2)
        gx = int(leaves.dims[0])
        ic, v, l, s = polygon2d.contour_blocks(leaves)
        t1 = time.perf_counter()
        polys = list(polygon2d.stitch([(ic[i], v[i], l[i], s[i]) for i in range(len(ic))],
                                      leaves.int_step * (gx - 1) if leaves.count > 1 else None))
        t2 = time.perf_counter()
    print("%-20s blocks %4d of %d^2: device+download %.1f ms, stitching %.1f ms, %d polygons, %d vertices"
          % (name, leaves.count, gx, (t1 - t0) * 1e3, (t2 - t1) * 1e3, len(polys), sum(len(p) for p in polys)), flush=True)
