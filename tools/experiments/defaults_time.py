#!/usr/bin/env python3
"""Run on the GPU box: the library drivers at the REFERENCE's default grid sizes (mass_properties: 64, subdivision: 128)
on sponge(4) at 1/512, per-tape code; HU_CLASSIFY_BOX_MIN switches the classification kernels' box path."""
import os
import sys
import time

os.environ.setdefault("CODECAD_AMD_CACHE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import codecad_amd as cc  # noqa: E402
from codecad_amd import hip_util  # noqa: E402

shape = cc.examples.sponge(4)
cc.nodes.make_program_buffer(shape).specialize()
for grid in (64, 16, 8):
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        mp = cc.mass_properties(shape, 1.0 / 512, grid_size=grid)
        hip_util.manager.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("mass_properties grid %3d: %.3f ms  volume %.9f" % (grid, best * 1e3, mp.volume), flush=True)
for grid in (128, 16):
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        leaves = cc.subdivision.subdivision_device(shape, 1.0 / 512, grid_size=grid)
        hip_util.manager.synchronize()
        best = min(best, time.perf_counter() - t0)
        n = leaves.count
        leaves.blocks.release()
    print("subdivision     grid %3d: %.3f ms  %d leaf blocks" % (grid, best * 1e3, n), flush=True)
