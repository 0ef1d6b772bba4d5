#!/usr/bin/env python3
"""Dense float4 / float grid_eval throughput of a few shapes (HBM-bound vs VALU-bound regimes)."""
import os
import sys

os.environ.setdefault("CODECAD_AMD_CACHE", "0")   # measurements and soaks choose their evaluator themselves

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
import codecad_amd as cc  # noqa: E402
import bench  # noqa: E402
from codecad_amd.shapes import sphere, box  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
shapes = {"sphere": sphere(1.0), "box": box(1.0), "sphere+box": cc.examples.sphere_plus_box().scaled(1 / 140.0),
          "csg_example": cc.examples.csg_example().scaled(1 / 110.0), "sponge1": cc.examples.sponge(1),
          "sponge3": cc.examples.sponge(3), "sponge4": cc.examples.sponge(4), "sponge5": cc.examples.sponge(5)}
for name, shape in shapes.items():
    tape = cc.nodes.make_program(shape)
    for pym in (False, True):
        out = None
        for _ in range(4):
            out = cc.grid_eval.grid_eval(shape, [-0.5 + 0.5 / n] * 3, np.float32(1.0 / n), (n, n, n), pymcubes=pym, out=out)
            ms = out.event.elapsed_ms()
        bytes_per = 4 if pym else 16
        print("%-12s %-6s instrs=%3d flop=%5d  %.3f ms  %6.2f Gvoxel/s  %7.1f GB/s" % (
            name, "float" if pym else "float4", cc.nodes.make_program_buffer(shape).n_instructions,
            bench.tape_flop(tape), ms, n ** 3 / ms / 1e6, n ** 3 * bytes_per / ms / 1e6))
        out.release()
