#!/usr/bin/env python3
"""Run on the GPU box: dense grids whose extents are / are no multiples of (4, 4, 8), per-tape code: planetary and sponge(4)."""
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

g = {s["name"]: s for s in json.load(open(os.path.join(ROOT, "tests", "golden", "ref_tapes.json")))["shapes"]}["planetary"]
tape = np.array(g["tape_u32"], dtype=np.uint32).view(np.float32)
box = util.BoundingBox(util.Vector(*[float(v) for v in g["bbox_a"]]), util.Vector(*[float(v) for v in g["bbox_b"]]))
for name, shape, bb in (("planetary", TapeShape(tape, box, float(g["feature_size"])), box), ("sponge4", cc.examples.sponge(4), cc.examples.sponge(4).bounding_box())):
    cc.nodes.make_program_buffer(shape).specialize()
    size = max(bb.b.x - bb.a.x, bb.b.y - bb.a.y, bb.b.z - bb.a.z)
    for n in ((256, 256, 256), (250, 250, 250), (256, 256, 250), (255, 256, 256)):
        for pym in (True, False):
            out, best = None, 1e9
            for _ in range(5):
                out = cc.grid_eval.grid_eval(shape, [bb.a.x, bb.a.y, bb.a.z], np.float32(size / 256), n, pymcubes=pym, out=out)
                out.event.wait()
                best = min(best, out.event.elapsed_ms())
            print("%-10s %s %-6s %.3f ms  %.1f Gvoxel/s" % (name, n, "float" if pym else "float4", best, n[0] * n[1] * n[2] / best / 1e6), flush=True)
            out.release()
