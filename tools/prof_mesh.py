#!/usr/bin/env python3
"""Time the mesh path on the bench workload: sponge(4) leaf blocks at 1/512 (grid 16) -> marching cubes."""
import os
import sys
import time

os.environ.setdefault("CODECAD_AMD_CACHE", "0")   # measurements and soaks choose their evaluator themselves

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
    t4 = time.perf_counter()
    m3 = mesh.mesh_blocks(leaves, download=False, stl=True)
    t5 = time.perf_counter()
    line = "  STL records on the device + download of %.2f GB: %.1f ms" % (m3.stl_records.nbytes / 1e9, (t5 - t4) * 1e3)
    if rep == 0:
        # what the host would do with the downloaded indexed mesh (numpy; the reference does it in a Python loop)
        t6 = time.perf_counter()
        rec = np.zeros(len(m2.triangles), dtype=[("normal", "<f4", 3), ("vectors", "<f4", (3, 3)), ("attr", "<u2")])
        rec["vectors"] = m2.vertices[m2.triangles].astype(np.float32)
        v = rec["vectors"]
        rec["normal"] = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
        t7 = time.perf_counter()
        same = rec.tobytes() == m3.stl_records.tobytes()
        line += "; the same records with numpy on the host: %.0f ms (identical: %s)" % ((t7 - t6) * 1e3, same)
        del rec, v
    print(line, flush=True)
    t8 = time.perf_counter()
    sink_bytes = [0]
    mesh.mesh_blocks(leaves, download=False, stl_sink=lambda piece: sink_bytes.__setitem__(0, sink_bytes[0] + piece.nbytes))
    t9 = time.perf_counter()
    with open("/tmp/prof_mesh.stl", "wb") as fp:
        mesh.mesh_blocks(leaves, download=False, stl_sink=lambda piece: fp.write(memoryview(piece).cast("B")))
    t10 = time.perf_counter()
    os.unlink("/tmp/prof_mesh.stl")
    print("  streamed in 50 MiB pieces: %.1f ms to a counting sink (%.2f GB), %.1f ms into a file in /tmp" %
          ((t9 - t8) * 1e3, sink_bytes[0] / 1e9, (t10 - t9) * 1e3), flush=True)
    del m, m2, m3
    leaves.blocks.release()
