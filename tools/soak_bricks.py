#!/usr/bin/env python3
"""One-off soak of the box kernels of per-tape code (dense grids and leaf blocks: bricks walked along x over the tables of a box) on
seeded random CSG trees against the oracle: `soak_bricks.py [first_seed] [count]`.  tests/test_gpu_bricks.py holds the
cases that run every time."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import test_gpu_bricks as tb  # noqa: E402
from random_trees import random_3d  # noqa: E402
from codecad_amd import hip_util, nodes  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
hip = hip_util.manager
hip.lib  # raises loudly if the extension or the device is missing
bad = 0
for seed in range(first, first + count):
    rng = random.Random(seed)
    tape = nodes.make_program(random_3d(rng, rng.choice([2, 3, 4])))
    grids = [(np.array([-4.0, -4.0, -8.0]), np.float32(0.5), (16, 16, 32)),
             (np.array([-1.53 + 0.01 * (seed % 7), -0.97, -2.11]), np.float32(0.13), (8, 12, 64))]
    blocks = [([(-8, -8, -8), (0, -8, -8), (-3, 1, 2)], 0.25, (0.0, 0.0, 0.0)),
              ([(0, 0, 0), (5, -7, 3)], 0.07, (-0.31, 0.12, -0.55))]
    try:
        tb.run(hip, tape, grids, blocks)
    except AssertionError as e:
        bad += 1
        print("seed %d: %s" % (seed, e), flush=True)
    if (seed - first) % 50 == 49:
        print("%d trees, %d bad" % (seed - first + 1, bad), flush=True)
print("soak_bricks: %d trees, %d bad" % (count, bad))
sys.exit(1 if bad else 0)
