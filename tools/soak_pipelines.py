#!/usr/bin/env python3
"""One-off soak of the level-batched drivers on random 3D CSG trees against the oracle-driven per-block
reference traversals: subdivision leaf sets, mass properties, and the mesh pipeline (vertex bits, triangle ids).
Usage: python tools/soak_pipelines.py [trees]"""
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
import ref_driver  # noqa: E402
import test_mesh as tm  # noqa: E402
import codecad_amd as cc  # noqa: E402
from codecad_amd import nodes, subdivision  # noqa: E402
from codecad_amd.rendering import mesh  # noqa: E402

spec = importlib.util.spec_from_file_location("trees", os.path.join(ROOT, "tests", "test_gpu_random_shapes.py"))
trees = importlib.util.module_from_spec(spec)
spec.loader.exec_module(trees)

count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
done = 0
for seed in range(count):
    rng = random.Random(11000 + seed)
    shape = trees.random_3d(rng, rng.choice([2, 3]))
    box = shape.bounding_box()
    if not all(math.isfinite(v) for v in tuple(box.a) + tuple(box.b)):
        continue
    tape = nodes.make_program(shape)
    extent = max(box.size())
    res = extent / rng.choice([24, 40])
    grid = rng.choice([4, 8, 16])
    # subdivision: leaf sets
    leaves = subdivision.subdivision_device(shape, res, grid_size=grid)
    dims, want = ref_driver.subdivision(tape, box, 3, res, overlap=True, grid_size=grid)
    got = leaves.int_corners()
    assert sorted(map(tuple, got.tolist())) == sorted(b[2] for b in want), ("subdivision", seed)
    leaves.blocks.release()
    # mass properties
    g = rng.choice([4, 8])
    mp = cc.mass_properties(shape, res, grid_size=g)
    want_mp, _ = ref_driver.mass_properties(tape, box, res, grid_size=g)
    assert abs(mp.volume - want_mp.volume) <= 1e-12 * max(abs(want_mp.volume), 1e-30), ("volume", seed, mp.volume, want_mp.volume)
    assert np.allclose(mp.inertia_tensor, want_mp.inertia_tensor, rtol=1e-9, atol=1e-12 * max(1.0, abs(want_mp.volume))), ("inertia", seed)
    # mesh pipeline
    mg = rng.choice([8, 16, None])
    got_m = list(mesh.triangular_mesh(shape, subdivision_grid_size=mg))
    want_m = [(c, v, t) for c, v, t in tm.oracle_mesh(shape, mg)[0] if len(t)]
    assert len(got_m) == len(want_m), ("mesh blocks", seed, len(got_m), len(want_m))
    for (gv, gt), (_c, wv, wt) in zip(got_m, want_m):
        assert np.array_equal(gv.view(np.uint64), wv.view(np.uint64)) and np.array_equal(gt, wt), ("mesh", seed)
    done += 1
print("soak ok:", done, "trees of", count)
