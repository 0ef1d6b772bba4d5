#!/usr/bin/env python3
"""Run on the GPU box: mass_properties of the planetary assembly at grid 64 for several resolutions (per-tape code): does
the rate rise with the size of the leaf level (a tail of heavy boxes at the end of the launch would show as a fixed cost)?"""
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

g = {s["name"]: s for s in json.load(open(os.path.join(ROOT, "tests", "golden", "ref_tapes.json")))["shapes"]}["planetary"]
tape = np.array(g["tape_u32"], dtype=np.uint32).view(np.float32)
box = util.BoundingBox(util.Vector(*[float(v) for v in g["bbox_a"]]), util.Vector(*[float(v) for v in g["bbox_b"]]))
shape = TapeShape(tape, box, float(g["feature_size"]))
cc.nodes.make_program_buffer(shape).specialize()
for res in (1.0, 0.5, 0.25, 0.125):
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        mp = cc.mass_properties(shape, res, 64)
        hip_util.manager.synchronize()
        best = min(best, time.perf_counter() - t0)
    n = cc.mass_properties.last_stats["function_evaluations"]
    print("resolution %.3f: %.3f ms, %d samples, %.1f Gsamples/s, volume %.3f" % (res, best * 1e3, n, n / best / 1e9, mp.volume), flush=True)
