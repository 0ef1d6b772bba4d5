#!/usr/bin/env python3
"""Known answers for the BASELINE configs at their STATED sizes, produced once in the build container by the
per-block reference traversal of tests/ref_driver.py over the CPU oracle's kernels (the restatement of
reference mass_properties.py:69-229 / subdivision.py:48-113) -> tests/golden/config_fixtures.json.

The GPU tests (tests/test_gpu_configs.py) run the level-batched HIP drivers at the same sizes and compare with
these numbers; nobody runs a 58 M-sample CPU traversal in the test suite.  Blocks are independent, so the
traversal is spread over worker processes (per-block sums are integers: the order does not matter; the fp64
integrals are accumulated per block exactly as ref_driver does and summed with math.fsum).

    python tests/golden/gen/make_config_fixtures.py [--workers 8]
"""
import argparse
import json
import math
import multiprocessing
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

_STATE = {}


def _init(tape):
    _STATE["tape"] = np.asarray(tape, dtype=np.float32)


def _block(args):
    """One mass_properties block: -> (ten fp64 integrals of the block, list of ambiguous child corners)."""
    import oracle
    corner, s, dims, leaf = args
    thr = 0.0 if leaf else s * math.sqrt(3) / 2
    shifted = [c + s / 2 for c in corner]
    sums, n, cells = oracle.mass_properties(_STATE["tape"], np.array(shifted, np.float64).astype(np.float32), np.float32(s),
                                            np.float32(thr), dims)
    sxx, sxy, sxz, sx, syy, syz, sy, szz, sz, cnt = (float(v) for v in sums)
    s2, s3 = s * s, s * s * s
    bx, by, bz = shifted
    tx, ty, tz = s * sx, s * sy, s * sz
    integrals = [s3 * cnt, s3 * (cnt * bx + tx), s3 * (cnt * by + ty), s3 * (cnt * bz + tz),
                 s3 * (cnt * (bx * bx + s2 / 12) + 2 * bx * tx + s2 * sxx),
                 s3 * (cnt * (by * by + s2 / 12) + 2 * by * ty + s2 * syy),
                 s3 * (cnt * (bz * bz + s2 / 12) + 2 * bz * tz + s2 * szz),
                 s3 * (cnt * bx * by + bx * ty + by * tx + s2 * sxy),
                 s3 * (cnt * bx * bz + bx * tz + bz * tx + s2 * sxz),
                 s3 * (cnt * by * bz + by * tz + bz * ty + s2 * syz)]
    children = [] if leaf else [[i * s + corner[0], j * s + corner[1], k * s + corner[2]] for i, j, k, _ in cells.tolist()]
    return integrals, children, int(cnt)


def mass_fixture(pool, box, resolution, grid):
    from codecad_amd.subdivision import calculate_block_sizes
    from codecad_amd.mass_properties import finish, _KEYS
    levels = [(resolution * c, tuple(int(v) for v in d)) for c, d in calculate_block_sizes(box, 3, resolution, grid, False)]
    parents = [[box.a.x, box.a.y, box.a.z]]
    columns = [[] for _ in range(10)]
    evaluations, level_parents, inside_cells = 0, [], []
    for level, (s, dims) in enumerate(levels):
        leaf = level + 1 == len(levels)
        level_parents.append(len(parents))
        evaluations += len(parents) * dims[0] * dims[1] * dims[2]
        results = pool.map(_block, [(p, s, dims, leaf) for p in parents], chunksize=1)
        parents, inside = [], 0
        for integrals, children, cnt in results:
            for col, v in zip(columns, integrals):
                col.append(v)
            parents += children
            inside += cnt
        inside_cells.append(inside)
        if not parents:
            break
    totals = dict(zip(_KEYS, [math.fsum(c) for c in columns]))
    mp = finish(totals)
    return {"resolution": resolution, "grid_size": grid, "levels": [[s, list(d)] for s, d in levels],
            "parents_per_level": level_parents, "inside_cells_per_level": inside_cells,
            "function_evaluations": evaluations, "volume": mp.volume,
            "centroid": [mp.centroid.x, mp.centroid.y, mp.centroid.z],
            "inertia_tensor": np.asarray(mp.inertia_tensor, dtype=np.float64).tolist()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=8)
    args = ap.parse_args()
    from conftest import load_golden_tapes
    from codecad_amd import util
    golden = load_golden_tapes()
    out = {"generator": "tests/golden/gen/make_config_fixtures.py (tests/ref_driver.py traversal over oracle/sdf_oracle.c)"}
    g = golden["planetary"]
    box = util.BoundingBox(util.Vector(*g["bbox_a"]), util.Vector(*g["bbox_b"]))
    t0 = time.time()
    with multiprocessing.Pool(args.workers, initializer=_init, initargs=(g["tape"],)) as pool:
        out["c4_planetary_mass_properties"] = mass_fixture(pool, box, 0.25, 64)
    out["c4_planetary_mass_properties"]["cpu_seconds_wall"] = round(time.time() - t0, 1)
    print("C4:", json.dumps(out["c4_planetary_mass_properties"])[:600])
    from codecad_amd import examples, nodes
    sponge = examples.sponge(4)
    t0 = time.time()
    with multiprocessing.Pool(args.workers, initializer=_init, initargs=(nodes.make_program(sponge),)) as pool:
        out["c3_sponge4_mass_properties"] = mass_fixture(pool, sponge.bounding_box(), 1.0 / 512, 8)
    out["c3_sponge4_mass_properties"]["cpu_seconds_wall"] = round(time.time() - t0, 1)
    print("C3:", json.dumps(out["c3_sponge4_mass_properties"])[:600])
    with open(os.path.join(ROOT, "tests", "golden", "config_fixtures.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
