#!/usr/bin/env python3
"""Time the BASELINE.json configs on one GPU (C2..C5 single-GPU forms) and print a table."""
import json
import os
import sys
import time

os.environ.setdefault("CODECAD_AMD_CACHE", "0")   # measurements and soaks choose their evaluator themselves

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import codecad_amd as cc  # noqa: E402
from codecad_amd import hip_util, util  # noqa: E402
from codecad_amd.shapes import TapeShape  # noqa: E402


def timed(fn, reps=3):
    fn()
    hip_util.manager.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        hip_util.manager.synchronize()
        best = min(best, time.perf_counter() - t0)
    return r, best


rows = []
# C2: sponge(3) 256^3 dense
s3 = cc.examples.sponge(3)
for n in (256, 512):
    out = None

    def dense():
        global out
        out = cc.grid_eval.grid_eval(s3, [-0.5 + 0.5 / n] * 3, np.float32(1.0 / n), (n, n, n), out=out)
        return out
    _, t = timed(dense)
    rows.append(("C2 sponge(3) %d^3 dense float4 grid_eval" % n, t, "%.1f Gvoxel/s" % (n ** 3 / t / 1e9)))
    out.release()

# C3: sponge(4) adaptive 512^3
s4 = cc.examples.sponge(4)
for grid in (8, 16, 128):
    leaves, t = timed(lambda: cc.subdivision.subdivision_device(s4, 1 / 512, True, grid))
    rows.append(("C3 sponge(4) subdivision res 1/512 grid %d: %d leaf blocks of %s, %d samples" % (
        grid, leaves.count, tuple(int(d) for d in leaves.dims), leaves.samples), t,
        "%.2f Gsamples/s" % (leaves.samples / t / 1e9)))
mp, t = timed(lambda: cc.mass_properties(s4, 1 / 512, 8))
rows.append(("C3 sponge(4) mass_properties res 1/512 grid 8: volume %.9f (exact %.9f), %d samples" % (
    mp.volume, (20 / 27) ** 4, cc.mass_properties.last_stats["function_evaluations"]), t,
    "%.2f Gsamples/s" % (cc.mass_properties.last_stats["function_evaluations"] / t / 1e9)))

# C4: planetary mass properties (golden tape)
g = {s["name"]: s for s in json.load(open(os.path.join(ROOT, "tests", "golden", "ref_tapes.json")))["shapes"]}["planetary"]
tape = np.array(g["tape_u32"], dtype=np.uint32).view(np.float32)
planetary = TapeShape(tape, util.BoundingBox(util.Vector(*[float(v) for v in g["bbox_a"]]),
                                             util.Vector(*[float(v) for v in g["bbox_b"]])), float(g["feature_size"]))
for res in (1.0, 0.25):
    mp, t = timed(lambda: cc.mass_properties(planetary, res, 64), reps=2)
    ev = cc.mass_properties.last_stats["function_evaluations"]
    rows.append(("C4 planetary mass_properties res %.2f grid 64: volume %.3f centroid (%.3f, %.3f, %.3f), %d samples" % (
        res, mp.volume, mp.centroid.x, mp.centroid.y, mp.centroid.z, ev), t, "%.2f Gsamples/s" % (ev / t / 1e9)))

# C5 (single-GPU form): sponge(5) effective 2048^3
s5 = cc.examples.sponge(5)
leaves, t = timed(lambda: cc.subdivision.subdivision_device(s5, 1 / 2048, True, 16), reps=2)
rows.append(("C5 sponge(5) subdivision res 1/2048 grid 16 (1 GPU): %d leaf blocks, levels %s, %d samples" % (
    leaves.count, leaves.level_counts, leaves.samples), t, "%.2f Gsamples/s" % (leaves.samples / t / 1e9)))
ev0 = hip_util.Event(hip_util.manager, hip_util.manager.queue)
out = cc.grid_eval.grid_eval_blocks(leaves, pymcubes=True)
ms = out.event.elapsed_ms()
nvox = leaves.count * 16 ** 3
rows.append(("C5 sponge(5) grid_eval of all %d leaf blocks (16^3, float): %d voxels (%.1f%% of 2048^3)" % (
    leaves.count, nvox, 100.0 * nvox / 2048 ** 3), ms / 1e3, "%.1f Gvoxel/s" % (nvox / ms / 1e6)))
for r in rows:
    print("%-130s %9.3f ms  %s" % (r[0], r[1] * 1e3, r[2]))
