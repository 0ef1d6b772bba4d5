#!/usr/bin/env python3
"""Run on the GPU box: dense 256^3 grids (float4 and float) of every 3D golden tape with per-tape code; run once as it is
and once with HU_BRICKS=0 (runs of cells along z: no boxes, no tables) to see what the boxes do to each tape."""
import json
import os
import sys

os.environ.setdefault("CODECAD_AMD_CACHE", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import codecad_amd as cc  # noqa: E402
from codecad_amd import hip_util, util  # noqa: E402
from codecad_amd.shapes import TapeShape  # noqa: E402

n = 256
for g in json.load(open(os.path.join(ROOT, "tests", "golden", "ref_tapes.json")))["shapes"]:
    if g["dimension"] != 3:
        continue
    tape = np.array(g["tape_u32"], dtype=np.uint32).view(np.float32)
    a = np.array([v if np.isfinite(v) else -2.0 for v in g["bbox_a"]], float)
    b = np.array([v if np.isfinite(v) else 2.0 for v in g["bbox_b"]], float)
    size = float(np.max(b - a)) * 1.1 + 1e-3
    corner = list((a + b) / 2 - size / 2)
    shape = TapeShape(tape, util.BoundingBox(util.Vector(*a), util.Vector(*b)), 1.0)
    try:
        cc.nodes.make_program_buffer(shape).specialize()
    except Exception as e:
        print("%-28s FAILED %s" % (g["name"], str(e)[:60]), flush=True)
        continue
    ms = {}
    for pym in (False, True):
        out = None
        for _ in range(3):
            out = cc.grid_eval.grid_eval(shape, corner, np.float32(size / n), (n, n, n), pymcubes=pym, out=out)
        out.event.wait()
        first = out.event
        for _ in range(5):
            out = cc.grid_eval.grid_eval(shape, corner, np.float32(size / n), (n, n, n), pymcubes=pym, out=out)
        out.event.wait()
        ms[pym] = (out.event.profile.end - first.profile.end) * 1e-6 / 5
        out.release()
    print("%-28s %4d floats  float4 %.4f ms  float %.4f ms" % (g["name"], tape.size, ms[False], ms[True]), flush=True)
