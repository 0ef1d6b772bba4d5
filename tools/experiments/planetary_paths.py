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

# ---- the consumers: the whole mesh (subdivision at feature_size / 2, leaf grids, marching cubes; no download) and a render
from codecad_amd.rendering import mesh as mesh_mod, ray_caster  # noqa: E402
for grid in (16, 32):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        m = mesh_mod.mesh_arrays(shape, subdivision_grid_size=grid, download=False)
        hip_util.manager.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("mesh (feature size %.3f) grid %3d: %.2f ms wall, %.3f ms on the device, %d samples, %d triangles"
          % (shape.feature_size(), grid, best * 1e3, m.kernel_ms, m.samples if hasattr(m, "samples") else 0, m.n_triangles if hasattr(m, "n_triangles") else 0), flush=True)
box = shape.bounding_box()
cam = ray_caster.get_camera_params(box, (1024, 768), None)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    px = ray_caster.render(shape, *cam, (1024, 768))
    best = min(best, time.perf_counter() - t0)
print("ray caster 1024 x 768: %.2f ms wall (with the read-back)" % (best * 1e3), flush=True)
