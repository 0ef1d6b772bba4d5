import os, sys
os.environ.setdefault("CODECAD_AMD_CACHE", "0")
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/codecad_amd") else ".")
import numpy as np
import codecad_amd as cc
from codecad_amd import hip_util
s = cc.shapes
shape = s.union([s.rectangle(1.0, 0.6).translated(0.2 * k - 1.0, 0.1 * (k % 3)) for k in range(6)]) - s.circle(d=0.5)
for policy in ("interp", "spec"):
    buf = cc.nodes.make_program_buffer(shape)
    if policy == "spec":
        buf.specialize()
    for dims in ((2048, 2048, 1), (2048, 2047, 1)):
        for pym in (True, False):
            out, best = None, 1e9
            for _ in range(4):
                out = cc.grid_eval.grid_eval(shape, [-2.0, -2.0, 0.0], np.float32(4.0 / 2048), dims, pymcubes=pym, out=out)
                out.event.wait()
                best = min(best, out.event.elapsed_ms())
            print(policy, dims, "float" if pym else "float4", "%.3f ms" % best, flush=True)
            out.release()
