#!/usr/bin/env python3
"""One-off soak of the ray caster on random CSG trees against the oracle (all three render modes, byte equality).
Usage: python tools/soak_render.py [trees]"""
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
from codecad_amd import nodes  # noqa: E402
from codecad_amd.rendering import ray_caster  # noqa: E402

spec = importlib.util.spec_from_file_location("trees", os.path.join(ROOT, "tests", "test_gpu_random_shapes.py"))
trees = importlib.util.module_from_spec(spec)
spec.loader.exec_module(trees)

count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
size = (80, 60)
done = 0
for seed in range(count):
    rng = random.Random(7000 + seed)
    shape = trees.random_3d(rng, rng.choice([2, 3, 4]))
    box = shape.bounding_box()
    if not all(math.isfinite(v) for v in tuple(box.a) + tuple(box.b)):
        continue
    cam = ray_caster.get_camera_params(box, size, rng.choice([None, 40]))
    a = ray_caster.kernel_arguments(shape, *cam)
    tape = nodes.make_program(shape)
    for options in (0, 1, 2):
        got = ray_caster.render(shape, *cam, size=size, options=ray_caster.RenderOptions(options))
        want = oracle.ray_caster(tape, list(a["origin"]), list(a["forward"]), list(a["up"]), list(a["right"]),
                                 np.float32(a["pixel_tolerance"]), np.float32(a["box_radius"]), np.float32(a["min_distance"]),
                                 np.float32(a["max_distance"]), np.float32(a["floor_z"]), options, size, threads=16).transpose((1, 0, 2))
        bad = np.count_nonzero(np.any(got != want, axis=-1))
        assert bad == 0, (seed, options, bad)
    done += 1
    if done % 10 == 0:
        print("...", done, "trees", flush=True)
print("soak ok:", done, "rendered trees of", count)
