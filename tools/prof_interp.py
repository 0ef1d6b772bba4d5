#!/usr/bin/env python3
"""The tape interpreter on the bench's dense launch (512^3 sponge(4), float4 and float) and on the planetary tape's
distance grid: ten back-to-back launches between two events.  CODECAD_AMD_LIB selects a library variant."""
import os
import sys

os.environ.setdefault("CODECAD_AMD_CACHE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
import codecad_amd as cc  # noqa: E402
from codecad_amd import hip_util  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512


def timed(shape, tape, corner, step, dims, pym):
    t = hip_util.Tape(tape, policy="0")
    shape._codecad_amd_tape_buffer = t
    out = None
    for i in range(3):
        out = cc.grid_eval.grid_eval(shape, corner, step, dims, pymcubes=pym, out=out)
    first = out.event
    for i in range(10):
        out = cc.grid_eval.grid_eval(shape, corner, step, dims, pymcubes=pym, out=out)
    out.event.wait()
    ms = (out.event.profile.end - first.profile.end) * 1e-6 / 10
    out.release()
    return ms


label = os.path.basename(os.environ.get("CODECAD_AMD_LIB", "default"))
for depth in (3, 4, 5):
    s = cc.examples.sponge(depth)
    tape = cc.nodes.make_program(s)
    for pym in (False, True):
        ms = timed(s, tape, [-0.5 + 0.5 / n] * 3, np.float32(1.0 / n), (n, n, n), pym)
        print("%-22s sponge(%d) %-6s %.3f ms  %.1f Gvoxel/s" % (label, depth, "float" if pym else "float4", ms, n ** 3 / ms / 1e6), flush=True)
csg = cc.examples.csg_example()
for pym in (False, True):
    ms = timed(csg, cc.nodes.make_program(csg), [-65 + 65.0 / n] * 3, np.float32(130.0 / n), (n, n, n), pym)
    print("%-22s csg_example %-6s %.3f ms  %.1f Gvoxel/s" % (label, "float" if pym else "float4", ms, n ** 3 / ms / 1e6), flush=True)
