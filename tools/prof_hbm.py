#!/usr/bin/env python3
"""The HBM-bound regime and where it ends: dense 512^3 grid_eval (float4: 16 B/voxel, float: 4 B/voxel) over tapes
of growing length, interpreter and per-tape code, kernel time from HIP events -> one JSON line per (tape, evaluator,
layout) and a summary of the crossover (the first tape whose float4 launch falls below half of the 8 TB/s peak).
Usage (GPU box): python tools/prof_hbm.py [n] > gpurun_out/hbm_sweep.json"""
import json
import os
import sys

os.environ.setdefault("CODECAD_AMD_CACHE", "0")   # measurements choose their evaluator themselves

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import codecad_amd as cc  # noqa: E402
from codecad_amd import hip_util  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
PEAK = 8000.0
tapes = [("sphere", cc.shapes.sphere(130)), ("box", cc.shapes.box(100)), ("sphere_plus_box", cc.examples.sphere_plus_box()),
         ("csg_example", cc.examples.csg_example())] + [("sponge%d" % d, cc.examples.sponge(d)) for d in (1, 2, 3, 4, 5)]
rows = []
for name, shape in tapes:
    tape = cc.nodes.make_program(shape)
    bb = shape.bounding_box()
    extent = max(bb.b.x - bb.a.x, bb.b.y - bb.a.y, bb.b.z - bb.a.z)
    step = np.float32(extent / n)
    corner = [bb.a.x + extent / n / 2, bb.a.y + extent / n / 2, bb.a.z + extent / n / 2]
    for mode in ("interpreter", "specialised"):
        t = hip_util.Tape(tape, policy="0")
        if mode == "specialised":
            t.specialize()
        shape._codecad_amd_tape_buffer = t
        for pym, bpv in ((False, 16), (True, 4)):
            # ten back-to-back launches between two events (a launch timed on its own starts from idle clocks)
            out = None
            for i in range(3):
                out = cc.grid_eval.grid_eval(shape, corner, step, (n, n, n), pymcubes=pym, out=out)
            first = out.event
            for i in range(10):
                out = cc.grid_eval.grid_eval(shape, corner, step, (n, n, n), pymcubes=pym, out=out)
            out.event.wait()
            ms = (out.event.profile.end - first.profile.end) * 1e-6 / 10
            out.release()
            gbs = n ** 3 * bpv / (ms * 1e-3) / 1e9
            rows.append({"tape": name, "instructions": t.n_instructions, "evaluator": mode, "layout": "float" if pym else "float4",
                         "ms": round(ms, 4), "gvoxels_per_s": round(n ** 3 / ms / 1e6, 1), "GBps": round(gbs, 1),
                         "frac_of_8TBps": round(gbs / PEAK, 4)})
            print(json.dumps(rows[-1]), flush=True)
for mode in ("interpreter", "specialised"):
    f4 = [r for r in rows if r["evaluator"] == mode and r["layout"] == "float4"]
    over = [r for r in f4 if r["frac_of_8TBps"] < 0.5]
    print(json.dumps({"summary": mode, "best": max(f4, key=lambda r: r["GBps"]),
                      "first_tape_below_half_of_peak": over[0] if over else None}))
