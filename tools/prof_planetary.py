#!/usr/bin/env python3
"""BASELINE C4 on one GPU: mass_properties of the planetary assembly (its golden tape) at resolution 0.25, grid 64, and
its dense 256^3 grids, interpreter and per-tape code: `prof_planetary.py` (knobs through the environment, e.g.
HU_MAX_PATHS)."""
import json
import os
import sys
import time

os.environ.setdefault("CODECAD_AMD_CACHE", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import codecad_amd as cc  # noqa: E402
from codecad_amd import hip_util, util  # noqa: E402
from codecad_amd.shapes import TapeShape  # noqa: E402

g = {s["name"]: s for s in json.load(open(os.path.join(ROOT, "tests", "golden", "ref_tapes.json")))["shapes"]}["planetary"]
tape = np.array(g["tape_u32"], dtype=np.uint32).view(np.float32)
box = util.BoundingBox(util.Vector(*[float(v) for v in g["bbox_a"]]), util.Vector(*[float(v) for v in g["bbox_b"]]))
for policy in ("0", "1"):
    shape = TapeShape(tape, box, float(g["feature_size"]))
    t0 = time.perf_counter()
    buf = cc.nodes.make_program_buffer(shape)
    if policy == "1":
        buf.specialize()
    built = time.perf_counter() - t0
    best, mp = 1e9, None
    for _ in range(4):
        t0 = time.perf_counter()
        mp = cc.mass_properties(shape, 0.25, 64)
        hip_util.manager.synchronize()
        best = min(best, time.perf_counter() - t0)
    n = 256
    a, b = box.a, box.b
    size = max(b.x - a.x, b.y - a.y, b.z - a.z)
    corner = [a.x, a.y, a.z]
    out, dense = None, {}
    for pym in (True, False):
        for _ in range(2):
            out = cc.grid_eval.grid_eval(shape, corner, np.float32(size / n), (n, n, n), pymcubes=pym, out=None)
            out.event.wait()
        t0 = time.perf_counter()
        for _ in range(3):
            out = cc.grid_eval.grid_eval(shape, corner, np.float32(size / n), (n, n, n), pymcubes=pym, out=out)
        out.event.wait()
        dense[pym] = (time.perf_counter() - t0) / 3
        out.release()
    print("%-12s upload%s %.2f s; mass_properties res 0.25 grid 64: %.3f ms (volume %.3f, %d samples); 256^3 distance grid %.3f ms, float4 grid %.3f ms"
          % ("per-tape" if policy == "1" else "interpreter", " + hipRTC" if policy == "1" else "", built, best * 1e3, mp.volume,
             getattr(mp, "samples", 0), dense[True] * 1e3, dense[False] * 1e3), flush=True)
