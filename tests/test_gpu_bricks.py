"""Per-tape code on the grids where its box kernels run (csrc/kernels.hpp box_eval): dense grids and leaf blocks whose
extents are multiples of (4, 4, 8) -- a workgroup takes a box of up to 16^3 voxels, fills the box's single-axis and
pair tables in LDS (specialise.hpp "AXIS TABLES", "PAIR TABLES") and its wavefronts walk bricks along x reading them.
Every float must equal the oracle's: the tables move statements, they do not change them.  Full boxes, boxes cut by
the grid's edge in every direction, slabs that start inside a grid, blocks of several boxes.  (The other parity tests
use ragged grids, which take the kernels without bricks.)"""
import ctypes
import random

import numpy as np
import pytest

import oracle
import shapes_zoo
from conftest import load_golden_tapes, same_bits
from random_trees import random_3d

pytestmark = pytest.mark.gpu

GOLDEN = load_golden_tapes()
ZOO_3D = sorted(name for name in shapes_zoo.all_named if GOLDEN[name]["dimension"] == 3)


def check_dense(hip, handle, tape, corner, step, dims):
    from codecad_amd import hip_util
    c4 = np.zeros(4, np.float32)
    c4[:3] = corner
    out = hip_util.Buffer(hip_util.Buffer.quad_dtype(np.float32), dims)
    hip.k.grid_eval(dims, None, handle, c4, step, out).wait()
    got = out.read().view(np.float32).reshape(dims + (4,))
    want = oracle.grid_eval(tape, corner, step, dims)
    assert same_bits(got, want), "float4 grid differs: max |diff| = %g" % np.nanmax(np.abs(got - want))
    outw = hip_util.Buffer(np.float32, dims)
    hip.k.grid_eval_pymcubes(dims, None, handle, c4, step, outw).wait()
    assert same_bits(outw.read().reshape(-1), oracle.grid_eval_pymcubes(tape, corner, step, dims)), "distance grid differs"
    out.release()
    outw.release()


def check_blocks(hip, handle, tape, int_corners, resolution, origin, edge=16):
    """hu_grid_eval_blocks over blocks of edge^3 (or edge = (sx, sy, sz)) samples at integer corners (subdivision.py:100:
    corner * resolution + origin)"""
    from codecad_amd import hip_util
    from codecad_amd.hip_util import check
    n = len(int_corners)
    blocks = np.zeros((n, 4), np.int32)
    blocks[:, :3] = int_corners
    blocks_dev = hip_util.Buffer(np.int32, blocks.shape)
    blocks_dev.enqueue_write(blocks).wait()
    ext = (edge, edge, edge) if isinstance(edge, int) else tuple(edge)
    dims = (ctypes.c_uint32 * 3)(*ext)
    o = (ctypes.c_double * 3)(*origin)
    step = np.float32(resolution)
    for layout, per_voxel in ((0, 4), (1, 1)):
        out = hip_util.Buffer(np.float32, (n,) + ext + (per_voxel,) if layout == 0 else (n, ext[0] * ext[1] * ext[2]))
        check(hip.lib.hu_grid_eval_blocks(handle.device_ptr, blocks_dev.device_ptr, n, float(resolution), o, step, dims, layout,
                                          out.device_ptr, hip.queue.handle), "hu_grid_eval_blocks")
        hip.queue.finish()
        got = out.read()
        for i in range(n):
            corner = (np.array(int_corners[i], np.float64) * float(resolution) + np.array(origin, np.float64)).astype(np.float32)
            if layout == 0:
                want = oracle.grid_eval(tape, corner, step, ext)
                assert same_bits(got[i], want), "block %d float4 differs" % i
            else:
                want = oracle.grid_eval_pymcubes(tape, corner, step, ext)
                assert same_bits(got[i].reshape(-1), want.reshape(-1)), "block %d distances differ" % i
        out.release()
    blocks_dev.release()


def run(hip, tape, grids, block_sets, inspect=None):
    from codecad_amd import hip_util
    handle = hip_util.Tape(tape)
    handle.specialize(hip_util.SPEC_DENSE | hip_util.SPEC_BLOCKS)     # (the families these checks launch)
    if inspect is not None:
        inspect(handle)
    for corner, step, dims in grids:
        check_dense(hip, handle, tape, corner, step, dims)
    for block_set in block_sets:
        check_blocks(hip, handle, tape, *block_set)
    handle.release()


def check_slab(hip, handle, tape, corner, step, dims, x0, count):
    """hu_grid_eval_slab: planes [x0, x0 + count) of a grid, both layouts (what a rank of a multi-GPU job computes)"""
    import torch
    from codecad_amd.hip_util import check
    c4 = np.zeros(4, np.float32)
    c4[:3] = corner
    d = (ctypes.c_uint32 * 3)(*dims)
    fptr = ctypes.POINTER(ctypes.c_float)
    slab = torch.full((count, dims[1], dims[2], 4), float("nan"), dtype=torch.float32, device="cuda")
    check(hip.lib.hu_grid_eval_slab(handle.device_ptr, c4.ctypes.data_as(fptr), step, d, x0, count, 0, slab.data_ptr(), None), "slab")
    flat = torch.full((dims[1], dims[0], dims[2]), float("nan"), dtype=torch.float32, device="cuda")
    check(hip.lib.hu_grid_eval_slab(handle.device_ptr, c4.ctypes.data_as(fptr), step, d, x0, count, 1, flat.data_ptr(), None), "slab")
    torch.cuda.synchronize()
    want = oracle.grid_eval(tape, corner, step, dims)
    assert same_bits(slab.cpu().numpy(), want[x0:x0 + count]), "slab differs"
    wantw = oracle.grid_eval_pymcubes(tape, corner, step, dims).reshape(dims[1], dims[0], dims[2])
    got = flat.cpu().numpy()
    assert same_bits(got[:, x0:x0 + count], wantw[:, x0:x0 + count]), "slab of the distance grid differs"
    assert np.isnan(got[:, :x0]).all() and np.isnan(got[:, x0 + count:]).all(), "a slab wrote outside its planes"


@pytest.mark.parametrize("seed", range(40))
def test_random_trees_through_the_brick_kernels(hip, seed):
    from codecad_amd import nodes
    rng = random.Random(7000 + seed)
    tape = nodes.make_program(random_3d(rng, rng.choice([2, 3, 4])))
    grids = [(np.array([-4.0, -4.0, -8.0]), np.float32(0.5), (16, 16, 32)),       # through exact zeros, symmetric pairs
             (np.array([-1.53, -0.97, -2.11]), np.float32(0.13), (8, 12, 64))]     # four bricks along z per wavefront
    blocks = [([(-8, -8, -8), (0, -8, -8), (-3, 1, 2), (8, 8, -24)], 0.25, (0.0, 0.0, 0.0)),
              ([(0, 0, 0), (16, 0, 0), (5, -7, 3)], 0.07, (-0.31, 0.12, -0.55))]
    run(hip, tape, grids, blocks)


@pytest.mark.parametrize("seed", range(12))
def test_boxes_cut_by_the_edge_slabs_and_blocks_of_several_boxes(hip, seed):
    from codecad_amd import hip_util, nodes
    import codecad_amd as cc
    rng = random.Random(9100 + seed)
    tape = nodes.make_program(cc.examples.sponge(2 + seed % 3) if seed < 3 else random_3d(rng, rng.choice([2, 3, 4])))
    scale = 1.0 if seed < 3 else 8.0
    # 20 x 36 x 24: boxes of 16 and 4 along x, 16 / 16 / 4 along y, 16 and 8 along z
    grids = [(np.array([-0.47, -0.51, -0.49]) * scale, np.float32(0.027 * scale), (20, 36, 24))]
    blocks = [([(-16, -16, -16), (3, -5, 1)], 0.03 * scale, (0.01, -0.02, 0.0), 32),           # eight boxes per block
              ([(-24, -32, -20)], 0.02 * scale, (0.0, 0.0, 0.0), (48, 64, 40)) if seed < 4 else  # 3 x 4 x 3 boxes, cut along z
              ([(-2, -2, -4)], 0.2 * scale, (0.0, 0.0, 0.0), (4, 4, 8)),                           # one brick
              ([(-4, -10, -12), (0, 0, 0)], 0.05 * scale, (0.0, 0.0, 0.0), (8, 20, 24))]      # one box along x, two along y and z, cut
    run(hip, tape, grids, blocks)
    handle = hip_util.Tape(tape)
    handle.specialize()
    corner, step, dims = np.array([-0.5, -0.52, -0.48]) * scale, np.float32(0.031 * scale), (32, 16, 24)
    for x0, count in ((8, 12), (0, 32), (20, 12), (12, 4)):
        check_slab(hip, handle, tape, corner, step, dims, x0, count)
    handle.release()


@pytest.mark.parametrize("seed", range(10))
def test_boxes_that_end_anywhere(hip, seed):
    """Extents that are no multiples of (4, 4, 8): per-tape code walks the bricks at the rim whole and stores only what is
    inside (k_grid_eval_ragged / k_grid_eval_blocks_ragged over box_eval RAGGED; before the end of round 4 one such extent
    sent the whole launch over runs, in the form without tables and pruning).  Grids, slabs and blocks -- 17^3: leaf blocks
    with overlapping edge samples -- of sponges, random trees and assemblies with pruning bits, against the oracle; a grid
    that is ragged only in ONE extent; grids smaller than a brick; nothing written outside."""
    import ctypes as ct
    import torch
    from codecad_amd import hip_util, nodes
    from codecad_amd.hip_util import check
    import codecad_amd as cc
    rng = random.Random(9900 + seed)
    if seed < 2:
        shape, scale = cc.examples.sponge(2 + seed), 1.0
    elif seed < 5:
        from test_gpu_pruning import assembly
        shape, scale = assembly(rng, 6 + seed, 6.0), 16.0        # (parts scattered over +-6: the grids below span the lot)
    else:
        shape, scale = random_3d(rng, rng.choice([2, 3, 4])), 8.0
    tape = nodes.make_program(shape)
    dims_list = [(21, 19, 13), (16, 16, 9), (5, 7, 3), (33, 20, 40), (16, 18, 16), (1, 1, 1), (2, 35, 17)]
    grids = [(np.array([-0.47, -0.51, -0.49]) * scale, np.float32(0.9 * scale / max(d)), d) for d in (dims_list[seed % 3], dims_list[3 + seed % 4])]
    blocks = [([(-8, -8, -8), (3, -5, 1), (0, 0, 0)], 0.05 * scale, (0.01, -0.02, 0.0), 17),           # leaf blocks with overlapping edges
              ([(-4, -10, -12)], 0.04 * scale, (0.0, 0.0, 0.0), (9, 20, 30))]
    run(hip, tape, grids, blocks)
    handle = hip_util.Tape(tape)
    handle.specialize(hip_util.SPEC_DENSE)
    corner, step, dims = np.array([-0.5, -0.52, -0.48]) * scale, np.float32(0.031 * scale), (30, 18, 20)
    for x0, count in ((7, 13), (0, 30), (29, 1)):
        check_slab(hip, handle, tape, corner, step, dims, x0, count)
    # nothing outside the grid: a buffer with a guard plane on either side of a ragged grid
    d = (21, 19, 13)
    c4 = np.zeros(4, np.float32)
    c4[:3] = corner
    guard = torch.full((d[0] + 2, d[1], d[2], 4), float("nan"), dtype=torch.float32, device="cuda")
    check(hip.lib.hu_grid_eval_slab(handle.device_ptr, c4.ctypes.data_as(ct.POINTER(ct.c_float)), step, (ct.c_uint32 * 3)(*d), 0, d[0], 0,
                                    guard[1:].data_ptr(), None), "slab")
    torch.cuda.synchronize()
    got = guard.cpu().numpy()
    assert np.isnan(got[0]).all() and np.isnan(got[-1]).all() and not np.isnan(got[1:-1, :, :, 3]).any()
    handle.release()


@pytest.mark.parametrize("name", ZOO_3D)
def test_the_zoo_through_the_brick_kernels(hip, name):
    ref = GOLDEN[name]
    a, b = np.array(ref["bbox_a"], dtype=np.float64), np.array(ref["bbox_b"], dtype=np.float64)
    a = np.where(np.isfinite(a), a, -2.0)
    b = np.where(np.isfinite(b), b, 2.0)
    size = float(np.max(b - a)) * 1.2 + 1e-3
    mid = (a + b) / 2
    step = np.float32(size / 32)
    grids = [(mid - float(step) * np.array([8, 8, 16]) + float(step) / 2, step, (16, 16, 32))]
    res = size / 48
    blocks = [([(-24, -24, -24), (-8, -8, -8), (8, -8, 0)], res, tuple(mid))]
    run(hip, ref["tape"], grids, blocks)


@pytest.mark.parametrize("scale,offset", [(1.0, 0.0), (1e-9, 0.0), (1e-30, 0.0), (3e13, 0.0), (1e20, 0.0), (1.0, 7e14), (1.0, 3e37)])
def test_the_in_range_flag_at_its_edges(hip, scale, offset):
    """Per-tape code skips the range test of its fast sqrt when a launch's coordinates stay below the tape's
    `coordinate_limit` (specialise.hpp; kernels read sdf::kFlagInRange).  Both sides of that decision against the oracle:
    tiny shapes (half extents below 2^-25: the tape allows no skipping at all), huge shapes and grids far from the origin
    (coordinates beyond the limit: the launch keeps the test; sums of squares beyond 2^100 take the IEEE path), and the
    ordinary case.  Every float must still be the oracle's."""
    import codecad_amd as cc
    from codecad_amd import nodes
    shape = cc.examples.sponge(2).scaled(scale)
    if offset:
        shape = shape.translated(offset, 0, -offset / 3)
    tape = nodes.make_program(shape)
    step = np.float32(scale / 24)
    centre = np.array([offset, 0.0, -offset / 3])
    grids = [(centre - float(step) * np.array([8, 8, 16]) + float(step) / 2, step, (16, 16, 32))]
    res = float(scale) / 40
    blocks = [([(-16, -16, -16), (0, -8, 3)], res, tuple(centre))]
    run(hip, tape, grids, blocks)


def test_more_pair_statements_than_table_columns(hip):
    """Thirty cylinders along two axes: more statements of two coordinates than a box's tables have columns (specialise.hpp
    kMaxPairColumns, kMaxPairTotal) -- the rest is computed by the walks; every float the oracle's all the same."""
    import codecad_amd as cc
    from codecad_amd import nodes
    parts = [cc.shapes.cylinder(d=0.5 + 0.01 * i, h=40).translated(i * 0.7 - 7, (i % 5) * 0.9, 0) for i in range(20)]
    parts += [cc.shapes.cylinder(d=0.4 + 0.01 * i, h=40).rotated_x(90).translated(i * 0.7 - 3, 0, (i % 3) * 1.1) for i in range(10)]
    tape = nodes.make_program(cc.shapes.union(parts))
    grids = [(np.array([-7.3, -1.1, -1.6]), np.float32(0.11), (32, 16, 24))]
    blocks = [([(-60, -8, -8), (0, 0, 0), (20, 5, -3)], 0.1, (0.0, 0.0, 0.0))]
    run(hip, tape, grids, blocks)


def check_classify(hip, handle, tape, corner, step, dims, dimension=3):
    """subdivision_step and mass_properties (both thresholds) of per-tape code on one grid: count, index set and the ten
    moment sums exact (the order of the list is unspecified)."""
    import math
    from codecad_amd import hip_util
    c4 = np.zeros(4, np.float32)
    c4[:3] = corner
    cells = dims[0] * dims[1] * dims[2]
    counter = hip_util.Buffer(np.uint32, 1)
    lst = hip_util.Buffer(hip_util.Buffer.quad_dtype(np.uint8), cells)
    sums = hip_util.Buffer(np.uint32, 10)
    thr = np.float32(float(step) * math.sqrt(dimension) / 2)
    want_n, want = oracle.subdivision_step(tape, corner, step, thr, dims)
    ev = counter.enqueue_fill(0)
    hip.k.subdivision_step(dims, None, handle, c4, step, thr, counter, lst, wait_for=[ev]).wait()
    got_n = int(counter.read()[0])
    assert got_n == want_n
    assert sorted(map(tuple, lst.read().view(np.uint8).reshape(-1, 4)[:got_n].tolist())) == sorted(map(tuple, want.tolist()))
    for t in (thr, np.float32(0.0)):
        want_sums, want_n, want = oracle.mass_properties(tape, corner, step, t, dims)
        sums.enqueue_fill(0)
        counter.enqueue_fill(0)
        hip.k.mass_properties(dims, None, handle, c4, step, t, sums, counter, lst).wait()
        assert sums.read().tolist() == want_sums.tolist()
        got_n = int(counter.read()[0])
        assert got_n == want_n
        assert sorted(map(tuple, lst.read().view(np.uint8).reshape(-1, 4)[:got_n].tolist())) == sorted(map(tuple, want.tolist()))
    for b in (counter, lst, sums):
        b.release()


@pytest.mark.parametrize("seed", range(16))
def test_classification_over_boxes(hip, seed, monkeypatch):
    """The classification kernels of per-tape code take the boxes too (kernels.hpp k_classify, grids of more than 256
    cells): full boxes, cut boxes, several boxes per grid, and -- since the end of round 4 -- grids whose extents are no
    multiples of (4, 4, 8): what lies beyond a box's rim stays out of the sums and the lists."""
    from codecad_amd import hip_util, nodes
    import codecad_amd as cc
    rng = random.Random(9400 + seed)
    tape = nodes.make_program(cc.examples.sponge(2 + seed % 3) if seed < 3 else random_3d(rng, rng.choice([2, 3, 4])))
    scale = 1.0 if seed < 3 else 8.0
    handle = hip_util.Tape(tape)
    handle.specialize()
    monkeypatch.setenv("HU_CLASSIFY_BOX_MIN", "1")     # (by default only launches of thousands of boxes go over boxes)
    for corner, step, dims in ((np.array([-0.5, -0.5, -0.5]) * scale, np.float32(scale / 16), (16, 16, 16)),
                               (np.array([-0.47, -0.51, -0.49]) * scale, np.float32(0.033 * scale), (20, 12, 24)),
                               (np.array([-0.52, -0.5, -0.51]) * scale, np.float32(scale / 31), (32, 32, 32)),
                               (np.array([-0.5, -0.5, -0.25]) * scale, np.float32(scale / 8), (8, 8, 8)),
                               (np.array([-0.49, -0.5, -0.51]) * scale, np.float32(scale / 21), (21, 19, 13)),
                               (np.array([-0.5, -0.52, -0.5]) * scale, np.float32(scale / 33), (33, 17, 9)),
                               (np.array([-0.5, -0.5, -0.5]) * scale, np.float32(scale / 25), (25, 25, 25))):
        check_classify(hip, handle, tape, corner, step, dims)
    handle.release()
