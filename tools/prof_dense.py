#!/usr/bin/env python3
"""Run only the dense 512^3 sponge(4) grid_eval a few times (profiling target for rocprofv3)."""
import ctypes
import os
import sys

os.environ.setdefault("CODECAD_AMD_CACHE", "0")   # measurements and soaks choose their evaluator themselves

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import codecad_amd as cc  # noqa: E402
from codecad_amd import hip_util  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 4
shape = cc.examples.sponge(depth)
out = None
for _ in range(reps):
    out = cc.grid_eval.grid_eval(shape, [-0.5 + 0.5 / n] * 3, np.float32(1.0 / n), (n, n, n), out=out)
    ms = out.event.elapsed_ms()
print("dense %d^3 sponge(%d): %.3f ms  %.2f Gvoxel/s" % (n, depth, ms, n ** 3 / ms / 1e6))
