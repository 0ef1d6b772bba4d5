#!/usr/bin/env python3
"""One-off soak of the classification kernels (subdivision_step, mass_properties) on random CSG trees against the
oracle: counts, cell index sets and the ten uint32 moment sums exact.  Usage: python tools/soak_classify.py [trees]"""
import importlib.util
import math
import os
import random
import sys

os.environ.setdefault("CODECAD_AMD_CACHE", "0")   # measurements and soaks choose their evaluator themselves

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle  # noqa: E402
from codecad_amd import nodes, hip_util  # noqa: E402
from codecad_amd.hip_util import manager as hip  # noqa: E402

spec = importlib.util.spec_from_file_location("trees", os.path.join(ROOT, "tests", "test_gpu_random_shapes.py"))
trees = importlib.util.module_from_spec(spec)
spec.loader.exec_module(trees)

count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for seed in range(count):
    rng = random.Random(5000 + seed)
    shape = trees.random_3d(rng, rng.choice([2, 3, 4]))
    tape = nodes.make_program(shape)
    handle = hip_util.Tape(tape)
    if seed % 5 == 0:
        handle.specialize()
    box = shape.bounding_box()
    n = rng.choice([7, 16, 19])
    extent = max(box.size().x, box.size().y, box.size().z)
    if not math.isfinite(extent) or extent <= 0:
        extent = 8.0
    step = np.float32(extent * 1.1 / n)
    mid = box.midpoint()
    mid = [m if math.isfinite(m) else 0.0 for m in mid]
    corner = np.array([m - float(step) * (n - 1) / 2 for m in mid])
    dims = (n, n, n)
    c4 = np.zeros(4, np.float32)
    c4[:3] = corner
    for leaf in (False, True):
        thr = np.float32(0.0 if leaf else float(step) * math.sqrt(3) / 2)
        want_sums, want_n, want = oracle.mass_properties(tape, corner, step, thr, dims)
        sums, counter = hip_util.Buffer(np.uint32, 10), hip_util.Buffer(np.uint32, 1)
        lst = hip_util.Buffer(hip_util.Buffer.quad_dtype(np.uint8), n ** 3)
        sums.enqueue_fill(0)
        counter.enqueue_fill(0)
        hip.k.mass_properties(dims, None, handle, c4, step, thr, sums, counter, lst).wait()
        assert sums.read().tolist() == want_sums.tolist(), (seed, leaf)
        got_n = int(counter.read()[0])
        got = lst.read().view(np.uint8).reshape(-1, 4)[:got_n]
        assert got_n == want_n and sorted(map(tuple, got.tolist())) == sorted(map(tuple, want.tolist())), (seed, leaf)
        for b in (sums, counter, lst):
            b.release()
    thr = np.float32(float(step) * math.sqrt(3) / 2)
    want_n, want = oracle.subdivision_step(tape, corner, step, thr, dims)
    counter = hip_util.Buffer(np.uint32, 1)
    lst = hip_util.Buffer(hip_util.Buffer.quad_dtype(np.uint8), n ** 3)
    counter.enqueue_fill(0)
    hip.k.subdivision_step(dims, None, handle, c4, step, thr, counter, lst).wait()
    got_n = int(counter.read()[0])
    got = lst.read().view(np.uint8).reshape(-1, 4)[:got_n]
    assert got_n == want_n and sorted(map(tuple, got.tolist())) == sorted(map(tuple, want.tolist())), seed
    counter.release()
    lst.release()
    handle.release()
print("soak ok:", count, "trees")
