#!/usr/bin/env python3
"""Specialised (hipRTC) vs interpreted evaluation of the same tape: identical bits, timings."""
import os
import sys
import time

os.environ.setdefault("CODECAD_AMD_CACHE", "0")   # measurements and soaks choose their evaluator themselves

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import codecad_amd as cc  # noqa: E402
from codecad_amd import hip_util  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for depth in (3, 4, 5):
    shape = cc.examples.sponge(depth)
    tape = cc.nodes.make_program(shape)
    res = {}
    for mode in ("interp", "spec"):
        t = hip_util.Tape(tape)
        if mode == "spec":
            t0 = time.perf_counter()
            t.specialize()
            jit_s = time.perf_counter() - t0
        shape._codecad_amd_tape_buffer = t
        for pym in (False, True):
            out = None
            for _ in range(4):
                out = cc.grid_eval.grid_eval(shape, [-0.5 + 0.5 / n] * 3, np.float32(1.0 / n), (n, n, n), pymcubes=pym, out=out)
                ms = out.event.elapsed_ms()
            res[(mode, pym)] = (ms, out.read().copy() if n <= 256 else None)
            out.release()
    for pym in (False, True):
        a, b = res[("interp", pym)], res[("spec", pym)]
        same = "" if a[1] is None else (" identical=%s" % np.array_equal(a[1].view(np.uint8), b[1].view(np.uint8)))
        print("sponge(%d) %-6s interp %.3f ms  spec %.3f ms (%.1f Gvoxel/s)  x%.2f  jit %.1f s%s" % (
            depth, "float" if pym else "float4", a[0], b[0], n ** 3 / b[0] / 1e6, a[0] / b[0], jit_s, same))
