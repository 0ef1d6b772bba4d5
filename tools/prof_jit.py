#!/usr/bin/env python3
"""hipRTC latency of per-tape specialisation (cold, stored into a fresh cache directory) and the latency of taking
the same program from the on-disk cache afterwards."""
import json
import os
import sys
import tempfile
import time

os.environ["CODECAD_AMD_CACHE"] = tempfile.mkdtemp(prefix="codecad_amd_cache_")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import codecad_amd as cc  # noqa: E402
from codecad_amd import hip_util  # noqa: E402
from codecad_amd.shapes import TapeShape  # noqa: E402

shapes = {s["name"]: s for s in json.load(open(os.path.join(ROOT, "tests/golden/ref_tapes.json")))["shapes"]}
for name in ("sponge4", "csg_example", "planetary"):
    tape = np.array(shapes[name]["tape_u32"], dtype=np.uint32).view(np.float32)
    t = hip_util.Tape(tape, policy="0")
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
    t4 = time.perf_counter()
    again = hip_util.Tape(tape)          # default policy: from the cache at upload
    t5 = time.perf_counter()
    hip_util.manager.k.grid_eval((32, 32, 32), None, again, c, np.float32(0.1), out).wait()
    t6 = time.perf_counter()
    size = sum(os.path.getsize(os.path.join(os.environ["CODECAD_AMD_CACHE"], f)) for f in os.listdir(os.environ["CODECAD_AMD_CACHE"]))
    print("%-12s %4d instructions: specialize (compile, ten kernels) %.2f s, first float4 launch %.3f s, first float launch %.3f s; "
          "upload + load from the cache %.1f ms (from_cache=%s), first launch %.1f ms; cache now %.2f MB"
          % (name, t.n_instructions, t1 - t0, t2 - t1, t3 - t2, (t5 - t4) * 1e3, again.from_cache, (t6 - t5) * 1e3, size / 1e6), flush=True)
