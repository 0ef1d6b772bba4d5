#!/usr/bin/env python3
"""Run on the GPU box: the library drivers on the planetary assembly, per-tape code: subdivision at several grid sizes, the
leaf blocks' grids, and the mesh built from them -- a look for cliffs (paths that fall back to unpruned code)."""
import os
import sys
import time

os.environ.setdefault("CODECAD_AMD_CACHE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import codecad_amd as cc  # noqa: E402
from codecad_amd import hip_util  # noqa: E402

shape = cc.examples.planetary()
t0 = time.perf_counter()
cc.nodes.make_program_buffer(shape).specialize()
print("specialize %.2f s" % (time.perf_counter() - t0), flush=True)
for res in (1.0, 0.5, 0.25):
    for grid in (16, 32, 128):
        best, n = 1e9, 0
        for _ in range(3):
            t0 = time.perf_counter()
            leaves = cc.subdivision.subdivision_device(shape, res, grid_size=grid)
            hip_util.manager.synchronize()
            best = min(best, time.perf_counter() - t0)
            n, dims = leaves.count, tuple(int(d) for d in leaves.dims)
            if _ < 2:
                leaves.blocks.release()
        out, bl = None, 1e9
        for _ in range(3):
            out = cc.grid_eval.grid_eval_blocks(leaves, pymcubes=True, out=out)
            out.event.wait()
            bl = min(bl, out.event.elapsed_ms())
        samples = n * dims[0] * dims[1] * dims[2]
        print("resolution %.2f grid %3d: subdivision %.3f ms -> %6d leaf blocks of %s; their float grids %.3f ms (%.1f Gsamples/s)"
              % (res, grid, best * 1e3, n, dims, bl, samples / bl / 1e6), flush=True)
        out.release()
        leaves.blocks.release()
