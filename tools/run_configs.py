#!/usr/bin/env python3
"""Time the BASELINE.json configs on one GPU (C2..C5 single-GPU forms) through the library drivers, as a user of the
package meets them, and print a table with TWO times per config: the first calls on a new shape (the tape interpreter
serves them while hipRTC builds the tape's kernels in the background) and the same call once that build has been picked
up (`Tape.wait_specialized`, about a second later).  The on-disk code cache is off here, so every shape is new."""
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


def timed(fn, reps=3, shape=None):
    """-> (result, best time; and, with `shape`: ... of the first calls, best time once the background build is in)"""
    def best_of():
        best, r = 1e9, None
        for _ in range(reps):
            t0 = time.perf_counter()
            r = fn()
            hip_util.manager.synchronize()
            best = min(best, time.perf_counter() - t0)
        return r, best
    fn()
    hip_util.manager.synchronize()
    r, first = best_of()
    if shape is None:
        return r, first
    tape = cc.nodes.make_program_buffer(shape)
    tape.note_samples(1e12)             # (a tape whose calls were all small so far: ask for the build now)
    tape.wait_specialized(timeout=300)
    fn()
    hip_util.manager.synchronize()
    r, later = best_of()
    return r, (first, later, tape.specialized)


rows = []
# C2: sponge(3) 256^3 dense
for n in (256, 512):
    s3 = cc.examples.sponge(3)      # (a new shape per row: every row starts with a tape nobody has compiled)
    out = None

    def dense():
        global out
        out = cc.grid_eval.grid_eval(s3, [-0.5 + 0.5 / n] * 3, np.float32(1.0 / n), (n, n, n), out=out)
        return out
    _, t = timed(dense, shape=s3)
    rows.append(("C2 sponge(3) %d^3 dense float4 grid_eval" % n, t, lambda t, n=n: "%.1f Gvoxel/s" % (n ** 3 / t / 1e9)))
    out.release()

# C3: sponge(4) adaptive 512^3
for grid in (8, 16, 128):
    s4 = cc.examples.sponge(4)
    leaves, t = timed(lambda: cc.subdivision.subdivision_device(s4, 1 / 512, True, grid), shape=s4)
    rows.append(("C3 sponge(4) subdivision res 1/512 grid %d: %d leaf blocks of %s, %d samples" % (
        grid, leaves.count, tuple(int(d) for d in leaves.dims), leaves.samples), t,
        lambda t, n=leaves.samples: "%.2f Gsamples/s" % (n / t / 1e9)))
s4 = cc.examples.sponge(4)
mp, t = timed(lambda: cc.mass_properties(s4, 1 / 512, 8), shape=s4)
rows.append(("C3 sponge(4) mass_properties res 1/512 grid 8: volume %.9f (exact %.9f), %d samples" % (
    mp.volume, (20 / 27) ** 4, cc.mass_properties.last_stats["function_evaluations"]), t,
    lambda t, n=cc.mass_properties.last_stats["function_evaluations"]: "%.2f Gsamples/s" % (n / t / 1e9)))

# C4: planetary mass properties (golden tape)
g = {s["name"]: s for s in json.load(open(os.path.join(ROOT, "tests", "golden", "ref_tapes.json")))["shapes"]}["planetary"]
tape = np.array(g["tape_u32"], dtype=np.uint32).view(np.float32)
planetary = TapeShape(tape, util.BoundingBox(util.Vector(*[float(v) for v in g["bbox_a"]]),
                                             util.Vector(*[float(v) for v in g["bbox_b"]])), float(g["feature_size"]))
for res in (1.0, 0.25):
    planetary = TapeShape(tape, planetary.bounding_box(), float(g["feature_size"]))
    mp, t = timed(lambda: cc.mass_properties(planetary, res, 64), reps=2, shape=planetary)
    ev = cc.mass_properties.last_stats["function_evaluations"]
    rows.append(("C4 planetary mass_properties res %.2f grid 64: volume %.3f centroid (%.3f, %.3f, %.3f), %d samples" % (
        res, mp.volume, mp.centroid.x, mp.centroid.y, mp.centroid.z, ev), t, lambda t, n=ev: "%.2f Gsamples/s" % (n / t / 1e9)))

# C5 (single-GPU form): sponge(5) effective 2048^3
s5 = cc.examples.sponge(5)
t5 = hip_util.Tape(cc.nodes.make_program(s5), policy="0")      # this shape's first calls: interpreted for sure
s5._codecad_amd_tape_buffer = t5
leaves, t_first = timed(lambda: cc.subdivision.subdivision_device(s5, 1 / 2048, True, 16), reps=2)
out = cc.grid_eval.grid_eval_blocks(leaves, pymcubes=True)
ms_first = out.event.elapsed_ms()
out.release()
t5.specialize()
leaves, t_later = timed(lambda: cc.subdivision.subdivision_device(s5, 1 / 2048, True, 16), reps=2)
rows.append(("C5 sponge(5) subdivision res 1/2048 grid 16 (1 GPU): %d leaf blocks, levels %s, %d samples" % (
    leaves.count, leaves.level_counts, leaves.samples), (t_first, t_later, True), lambda t, n=leaves.samples: "%.2f Gsamples/s" % (n / t / 1e9)))
out = cc.grid_eval.grid_eval_blocks(leaves, pymcubes=True)
out = cc.grid_eval.grid_eval_blocks(leaves, pymcubes=True, out=out) if False else out
ms = out.event.elapsed_ms()
nvox = leaves.count * 16 ** 3
rows.append(("C5 sponge(5) grid_eval of all %d leaf blocks (16^3, float): %d voxels (%.1f%% of 2048^3)" % (
    leaves.count, nvox, 100.0 * nvox / 2048 ** 3), (ms_first / 1e3, ms / 1e3, True), lambda t, n=nvox: "%.1f Gvoxel/s" % (n / t / 1e9)))
print("%-122s %22s   %s" % ("config", "first calls (interpreter)", "once the background build is in (per-tape code)"))
for name, t, rate in rows:
    first, later, ok = t
    print("%-122s %9.3f ms %13s   %9.3f ms %13s%s" % (name, first * 1e3, rate(first), later * 1e3, rate(later), "" if ok else "  (not built)"))
