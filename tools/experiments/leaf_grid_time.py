#!/usr/bin/env python3
"""Run on the GPU box: grid_eval of ALL leaf blocks of sponge(4) at 1/512 for several grid sizes (blocks of 16^3 ... 128^3:
one ... 512 boxes per block), float and float4, per-tape code: kernel time by HIP events."""
import os
import sys

os.environ.setdefault("CODECAD_AMD_CACHE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import codecad_amd as cc  # noqa: E402
from codecad_amd import hip_util  # noqa: E402

shape = cc.examples.sponge(4)
cc.nodes.make_program_buffer(shape).specialize()
for grid in (16, 32, 64, 128):
    leaves = cc.subdivision.subdivision_device(shape, 1.0 / 512, grid_size=grid)
    for pym in (True, False):
        out, best = None, 1e9
        for _ in range(6):
            out = cc.grid_eval.grid_eval_blocks(leaves, pymcubes=pym, out=out)
            out.event.wait()
            best = min(best, out.event.elapsed_ms()) if hasattr(out.event, "elapsed_ms") else best
        n = leaves.count
        dims = leaves.dims if hasattr(leaves, "dims") else None
        print("grid %3d: %6d leaf blocks %s, %s: %.4f ms" % (grid, n, dims, "float" if pym else "float4", best), flush=True)
        out.release()
    leaves.blocks.release()
