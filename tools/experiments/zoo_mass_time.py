#!/usr/bin/env python3
"""Run on the GPU box: mass_properties at the reference's default grid (64) of some golden 3D tapes with per-tape code;
HU_CLASSIFY_BOX_MIN switches the classification kernels' box path (default: from 8192 boxes per launch)."""
import json
import os
import sys
import time

os.environ.setdefault("CODECAD_AMD_CACHE", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import codecad_amd as cc  # noqa: E402
from codecad_amd import hip_util, util  # noqa: E402
from codecad_amd.shapes import TapeShape  # noqa: E402

names = sys.argv[1:] or ["csg_example", "csg_thing", "torus", "mirror_3d", "rotated_pattern_3d", "nested_transformations", "mp_not_hammer", "sponge3"]
shapes = {s["name"]: s for s in json.load(open(os.path.join(ROOT, "tests", "golden", "ref_tapes.json")))["shapes"]}
for name in names:
    g = shapes[name]
    tape = np.array(g["tape_u32"], dtype=np.uint32).view(np.float32)
    a, b = [float(v) for v in g["bbox_a"]], [float(v) for v in g["bbox_b"]]
    if not all(np.isfinite(a + b)):
        continue
    size = max(b[i] - a[i] for i in range(3))
    shape = TapeShape(tape, util.BoundingBox(util.Vector(*a), util.Vector(*b)), 1.0)
    cc.nodes.make_program_buffer(shape).specialize()
    res = size / 400
    best, mp = 1e9, None
    for _ in range(4):
        t0 = time.perf_counter()
        mp = cc.mass_properties(shape, res, 64)
        hip_util.manager.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("%-26s mass_properties at 1/400 of the size, grid 64: %.3f ms  volume %.6g" % (name, best * 1e3, mp.volume), flush=True)
