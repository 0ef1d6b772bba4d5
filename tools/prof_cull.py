#!/usr/bin/env python3
"""One tape through the interpreter's dense kernels (512^3), for rocprofv3: `prof_cull.py [sponge4|csg] [float4|float]`."""
import os
import sys

os.environ.setdefault("CODECAD_AMD_CACHE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import codecad_amd as cc  # noqa: E402
from codecad_amd import hip_util  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "sponge4"
pym = (sys.argv[2] if len(sys.argv) > 2 else "float4") == "float"
n = 512
if which.startswith("sponge"):
    s = cc.examples.sponge(int(which[6:]))
    corner, step = [-0.5 + 0.5 / n] * 3, np.float32(1.0 / n)
else:
    s = cc.examples.csg_example()
    corner, step = [-65 + 65.0 / n] * 3, np.float32(130.0 / n)
t = hip_util.Tape(cc.nodes.make_program(s), policy="0")
s._codecad_amd_tape_buffer = t
out = None
for i in range(13):
    out = cc.grid_eval.grid_eval(s, corner, step, (n, n, n), pymcubes=pym, out=out)
    if i == 2:
        first = out.event
out.event.wait()
print("%s %s: %.3f ms per launch" % (which, "float" if pym else "float4", (out.event.profile.end - first.profile.end) * 1e-6 / 10))
