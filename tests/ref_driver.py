"""Reference-shaped drivers over the ORACLE kernels (test infrastructure).

A restatement of how the reference's host code walks the hierarchy one block at a time
(reference subdivision.py:48-113,169-253 and mass_properties.py:69-229), calling the
oracle's per-block kernels.  Used to pin the oracle with the reference's own known-answer
and analytic tests, and as the per-block cross-check for the level-batched GPU drivers.
"""
import math

import numpy as np

import oracle
from codecad_amd import util
from codecad_amd.subdivision import calculate_block_sizes


def f32_corner(v):
    return np.array([v.x, v.y, v.z], dtype=np.float64).astype(np.float32)


def subdivision(tape, bbox, dimension, resolution, overlap=True, grid_size=128):
    """-> (leaf dims, [(corner Vector, step, int_corner tuple, int_step)]) via LIFO traversal."""
    box = bbox.expanded_additive(resolution / 2)
    if dimension == 2:
        box = box.flattened()
    levels = calculate_block_sizes(box, dimension, resolution, grid_size, overlap)
    if len(levels) == 1:
        return levels[0][1], [(box.a, resolution, (0, 0, 0), 1)]
    origin = box.a
    final, stack = [], [((0, 0, 0), 0)]
    while stack:
        (ix, iy, iz), level = stack.pop()
        int_step, dims = levels[level]
        half = int_step / 2
        shifted = util.Vector(ix + half, iy + half, iz + (half if dimension == 3 else 0))
        corner = shifted * resolution + origin
        step = int_step * resolution
        thr = step * math.sqrt(dimension) / 2
        n, cells = oracle.subdivision_step(tape, f32_corner(corner), np.float32(step), np.float32(thr),
                                           tuple(int(d) for d in dims))
        for i, j, k, _ in cells.tolist():
            child = (ix + i * int_step, iy + j * int_step, iz + k * int_step)
            if level + 1 == len(levels) - 1:
                nstep = levels[level + 1][0]
                pos = util.Vector(*child) * resolution + origin
                final.append((pos, nstep * resolution, child, nstep))
            else:
                stack.append((child, level + 1))
    return levels[-1][1], final


def mass_properties(tape, bbox, resolution, grid_size=64, kernel=None):
    """The reference's per-block traversal with Kahan-summed integrals -> (volume, centroid, inertia).
    `kernel`: the block kernel, oracle.mass_properties (canonical arithmetic) unless given."""
    kernel = kernel or oracle.mass_properties
    assert grid_size ** 5 <= 2 ** 32
    levels = [(resolution * c, d) for c, d in calculate_block_sizes(bbox, 3, resolution, grid_size, False)]
    acc = {k: util.KahanSummation() for k in ("1", "x", "y", "z", "xx", "yy", "zz", "xy", "xz", "yz")}
    stack = [(bbox.a, 0)]
    evaluations = 0
    while stack:
        corner, level = stack.pop()
        s, dims = levels[level]
        dims = tuple(int(d) for d in dims)
        leaf = level == len(levels) - 1
        thr = 0.0 if leaf else s * math.sqrt(3) / 2
        shifted = corner + util.Vector.splat(s / 2)
        sums, n, cells = kernel(tape, f32_corner(shifted), np.float32(s), np.float32(thr), dims)
        evaluations += dims[0] * dims[1] * dims[2]
        sxx, sxy, sxz, sx, syy, syz, sy, szz, sz, cnt = (float(v) for v in sums)
        s2, s3 = s * s, s * s * s
        b = shifted
        tx, ty, tz = s * sx, s * sy, s * sz
        acc["1"] += s3 * cnt
        acc["x"] += s3 * (cnt * b.x + tx)
        acc["y"] += s3 * (cnt * b.y + ty)
        acc["z"] += s3 * (cnt * b.z + tz)
        acc["xx"] += s3 * (cnt * (b.x * b.x + s2 / 12) + 2 * b.x * tx + s2 * sxx)
        acc["yy"] += s3 * (cnt * (b.y * b.y + s2 / 12) + 2 * b.y * ty + s2 * syy)
        acc["zz"] += s3 * (cnt * (b.z * b.z + s2 / 12) + 2 * b.z * tz + s2 * szz)
        acc["xy"] += s3 * (cnt * b.x * b.y + b.x * ty + b.y * tx + s2 * sxy)
        acc["xz"] += s3 * (cnt * b.x * b.z + b.x * tz + b.z * tx + s2 * sxz)
        acc["yz"] += s3 * (cnt * b.y * b.z + b.y * tz + b.z * ty + s2 * syz)
        assert not (leaf and n)
        for i, j, k, _ in cells.tolist():
            stack.append((util.Vector(i, j, k) * s + corner, level + 1))
    from codecad_amd.mass_properties import finish
    return finish({k: v.result for k, v in acc.items()}), evaluations
