#!/usr/bin/env python3
"""mass_properties of sponge(4) at 1/512, grid 8 (BASELINE C3), a few times: target for rocprofv3."""
import os
import sys
import time

os.environ.setdefault("CODECAD_AMD_CACHE", "0")   # measurements and soaks choose their evaluator themselves

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import codecad_amd as cc  # noqa: E402
from codecad_amd import hip_util  # noqa: E402

shape = cc.examples.sponge(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
if os.environ.get("CODECAD_AMD_SPECIALIZE") == "1":
    cc.nodes.make_program_buffer(shape).specialize()
for _ in range(4):
    t0 = time.perf_counter()
    mp = cc.mass_properties(shape, 1.0 / 512, grid_size=8)
    hip_util.manager.synchronize()
    print("volume %.9f  %.3f ms" % (mp.volume, (time.perf_counter() - t0) * 1e3), flush=True)
