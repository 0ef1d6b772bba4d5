#!/usr/bin/env python3
"""hipRTC latency of per-tape specialisation: hu_tape_specialize (first kernel) and first use of the others."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import codecad_amd as cc  # noqa: E402
from codecad_amd import hip_util  # noqa: E402
from codecad_amd.shapes import TapeShape  # noqa: E402

shapes = {s["name"]: s for s in json.load(open(os.path.join(ROOT, "tests/golden/ref_tapes.json")))["shapes"]}
for name in ("sponge4", "csg_example", "planetary"):
    tape = np.array(shapes[name]["tape_u32"], dtype=np.uint32).view(np.float32)
    t = hip_util.Tape(tape)
    t0 = time.perf_counter()
    t.specialize()
    t1 = time.perf_counter()
    out = hip_util.Buffer(cc.grid_eval.FLOAT4, (32, 32, 32))
    c = np.zeros(4, np.float32)
    hip_util.manager.k.grid_eval((32, 32, 32), None, t, c, np.float32(0.1), out).wait()
    t2 = time.perf_counter()
    outf = hip_util.Buffer(np.float32, (32, 32, 32))
    hip_util.manager.k.grid_eval_pymcubes((32, 32, 32), None, t, c, np.float32(0.1), outf).wait()
    t3 = time.perf_counter()
    print("%-12s %4d instructions: specialize (dense float4 kernel) %.2f s, first float4 launch %.3f s, first use of the float kernel %.2f s"
          % (name, t.n_instructions, t1 - t0, t2 - t1, t3 - t2), flush=True)
