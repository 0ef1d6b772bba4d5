#!/usr/bin/env python3
"""Time the mesh path on the bench workload: sponge(4) leaf blocks at 1/512 (grid 16) -> marching cubes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import codecad_amd as cc  # noqa: E402
from codecad_amd import subdivision  # noqa: E402
from codecad_amd.rendering import mesh  # noqa: E402

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 4
res = 1.0 / (int(sys.argv[2]) if len(sys.argv) > 2 else 512)
shape = cc.examples.sponge(depth)
for rep in range(3):
    t0 = time.perf_counter()
    leaves = subdivision.subdivision_device(shape, res, grid_size=16).sort()
    t1 = time.perf_counter()
    m = mesh.mesh_blocks(leaves, download=False)
    t2 = time.perf_counter()
    m2 = mesh.mesh_blocks(leaves, download=True)
    t3 = time.perf_counter()
    print("blocks %d samples %.1fM: subdivision %.1f ms, eval+marching cubes (device, wall) %.1f ms [mc kernels %.3f ms], with download %.1f ms; "
          "%d vertices %d triangles" % (leaves.count, m.samples / 1e6, (t1 - t0) * 1e3, (t2 - t1) * 1e3, m.kernel_ms, (t3 - t2) * 1e3,
                                        len(m2.vertices), len(m2.triangles)), flush=True)
    leaves.blocks.release()
