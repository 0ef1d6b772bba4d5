#!/usr/bin/env python3
"""One-off soak of 2D contouring on random 2D CSG trees: the batched GPU pipeline against the oracle-driven
per-block pipeline (polygons equal as lists).  Usage: python tools/soak_polygon.py [trees]"""
import importlib.util
import math
import os
import random
import sys

os.environ.setdefault("CODECAD_AMD_CACHE", "0")   # measurements and soaks choose their evaluator themselves

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import test_polygon2d_render as tp  # noqa: E402
from codecad_amd.rendering import polygon2d  # noqa: E402

spec = importlib.util.spec_from_file_location("trees", os.path.join(ROOT, "tests", "test_gpu_random_shapes.py"))
trees = importlib.util.module_from_spec(spec)
spec.loader.exec_module(trees)

count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
done = 0
for seed in range(count):
    rng = random.Random(9000 + seed)
    shape = trees.random_2d(rng, rng.choice([2, 3, 4]))
    box = shape.bounding_box()
    if not all(math.isfinite(v) for v in tuple(box.a)[:2] + tuple(box.b)[:2]):
        continue
    grid = rng.choice([16, 32, None])
    try:
        want_blocks, int_box_step, _ = tp.oracle_blocks(shape, grid or 128)
        want = list(polygon2d.stitch(want_blocks, int_box_step))
    except AssertionError:
        continue   # a contour that the reference's own stitching would refuse too (e.g. touching features)
    got = list(polygon2d.polygon(shape, subdivision_grid_size=grid))
    assert len(got) == len(want), (seed, len(got), len(want))
    for a, b in zip(got, want):   # a degenerate cell can yield a NaN vertex on both sides: NaN matches NaN
        a, b = np.array(a, dtype=np.float64), np.array(b, dtype=np.float64)
        assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True), seed
    done += 1
print("soak ok:", done, "contoured trees of", count)
