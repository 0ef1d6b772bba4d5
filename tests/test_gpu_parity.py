"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, same inputs.

Bars: SDF floats compared with ==  (NaN == NaN), i.e. stricter than the north star's 1e-5
relative; survivor lists as sorted sets, counts and uint32 moment sums exact.
"""
import ctypes
import math

import numpy as np
import pytest

import oracle
import shapes_zoo
from conftest import load_golden_tapes, same_bits

pytestmark = pytest.mark.gpu

GOLDEN = load_golden_tapes()
ZOO = sorted(shapes_zoo.all_named)


def _grid_for(ref, n):
    """Cell-centred n^3 (or n x n x 1) grid over the bounding box padded by 10 %."""
    a, b = np.array(ref["bbox_a"]), np.array(ref["bbox_b"])
    a = np.where(np.isfinite(a), a, -2.0)
    b = np.where(np.isfinite(b), b, 2.0)
    size = float(np.max(b - a)) * 1.2 + 1e-3
    mid = (a + b) / 2
    step = size / n
    corner = mid - size / 2 + step / 2
    dims = (n, n, n)
    if ref["dimension"] == 2:
        corner[2] = 0.0
        dims = (n, n, 1)
    return corner, np.float32(step), dims


def _same(a, b):
    return same_bits(a, b)


@pytest.mark.parametrize("name", ZOO)
def test_grid_eval_matches_oracle(hip, name):
    """grid_eval (float4) on the golden REFERENCE tape: every float equal to the oracle's."""
    from codecad_amd import hip_util
    ref = GOLDEN[name]
    corner, step, dims = _grid_for(ref, 24)
    want = oracle.grid_eval(ref["tape"], corner, step, dims)
    tape = hip_util.Tape(ref["tape"])
    out = hip_util.Buffer(hip_util.Buffer.quad_dtype(np.float32), dims)
    c4 = np.zeros(4, np.float32)
    c4[:3] = corner
    hip.k.grid_eval(dims, None, tape, c4, step, out).wait()
    got = out.read().view(np.float32).reshape(dims + (4,))
    assert _same(got, want), "max |diff| = %g" % np.nanmax(np.abs(got - want))


@pytest.mark.parametrize("name", ["sphere_plus_box", "csg_example", "sponge3", "gear", "csg_thing", "planetary"])
def test_grid_eval_pymcubes_matches_oracle(hip, name):
    from codecad_amd import hip_util
    ref = GOLDEN[name]
    corner, step, _ = _grid_for(ref, 20)
    dims = (20, 13, 17) if ref["dimension"] == 3 else (20, 13, 1)   # ragged, not a multiple of 64
    want = oracle.grid_eval_pymcubes(ref["tape"], corner, step, dims)
    tape = hip_util.Tape(ref["tape"])
    out = hip_util.Buffer(np.float32, dims)
    c4 = np.zeros(4, np.float32)
    c4[:3] = corner
    hip.k.grid_eval_pymcubes(dims, None, tape, c4, step, out).wait()
    assert _same(out.read().reshape(-1), want)


def test_our_compiler_tape_equals_reference_tape_on_gpu(hip):
    """Tapes from OUR compiler and the golden reference tapes give identical GPU grids."""
    from codecad_amd import hip_util, nodes
    for name in ("csg_example", "sponge4", "mirror_3d", "nested_transformations"):
        ref = GOLDEN[name]
        corner, step, dims = _grid_for(ref, 16)
        c4 = np.zeros(4, np.float32)
        c4[:3] = corner
        grids = []
        for t in (ref["tape"], nodes.make_program(shapes_zoo.all_named[name])):
            out = hip_util.Buffer(hip_util.Buffer.quad_dtype(np.float32), dims)
            hip.k.grid_eval(dims, None, hip_util.Tape(t), c4, step, out).wait()
            grids.append(out.read().view(np.float32).copy())
        assert _same(grids[0], grids[1])


@pytest.mark.parametrize("name", ["sphere_plus_box", "csg_example", "sponge2", "sponge4", "torus", "gear",
                                  "kat_box10", "kat_circle", "planetary"])
def test_subdivision_step_matches_oracle(hip, name):
    """subdivision_step: count exact, cell index set exact (the order is unspecified)."""
    from codecad_amd import hip_util
    ref = GOLDEN[name]
    n = 16
    corner, step, dims = _grid_for(ref, n)
    thr = np.float32(float(step) * math.sqrt(ref["dimension"]) / 2)
    want_n, want = oracle.subdivision_step(ref["tape"], corner, step, thr, dims)
    tape = hip_util.Tape(ref["tape"])
    counter = hip_util.Buffer(np.uint32, 1)
    lst = hip_util.Buffer(hip_util.Buffer.quad_dtype(np.uint8), n * n * n)
    c4 = np.zeros(4, np.float32)
    c4[:3] = corner
    ev = counter.enqueue_fill(0)
    hip.k.subdivision_step(dims, None, tape, c4, step, thr, counter, lst, wait_for=[ev]).wait()
    got_n = int(counter.read()[0])
    got = lst.read().view(np.uint8).reshape(-1, 4)[:got_n]
    assert got_n == want_n
    assert sorted(map(tuple, got.tolist())) == sorted(map(tuple, want.tolist()))


@pytest.mark.parametrize("name", ["mp_unit_box", "mp_sphere", "mp_drunk_box", "csg_example", "sponge3", "planetary"])
@pytest.mark.parametrize("leaf", [False, True])
def test_mass_properties_kernel_matches_oracle(hip, name, leaf):
    """mass_properties kernel: the ten uint32 moment sums, the count and the index set exact."""
    from codecad_amd import hip_util
    ref = GOLDEN[name]
    n = 20
    corner, step, dims = _grid_for(ref, n)
    thr = np.float32(0.0 if leaf else float(step) * math.sqrt(3) / 2)
    want_sums, want_n, want = oracle.mass_properties(ref["tape"], corner, step, thr, dims)
    tape = hip_util.Tape(ref["tape"])
    sums = hip_util.Buffer(np.uint32, 10)
    counter = hip_util.Buffer(np.uint32, 1)
    lst = hip_util.Buffer(hip_util.Buffer.quad_dtype(np.uint8), n * n * n)
    c4 = np.zeros(4, np.float32)
    c4[:3] = corner
    sums.enqueue_fill(0)
    counter.enqueue_fill(0)
    hip.k.mass_properties(dims, None, tape, c4, step, thr, sums, counter, lst).wait()
    assert sums.read().tolist() == want_sums.tolist()
    got_n = int(counter.read()[0])
    assert got_n == want_n
    got = lst.read().view(np.uint8).reshape(-1, 4)[:got_n]
    assert sorted(map(tuple, got.tolist())) == sorted(map(tuple, want.tolist()))


def test_empty_and_degenerate_launches(hip):
    """1x1x1 grid, 1-D and 2-D global sizes, a shape that is empty everywhere."""
    from codecad_amd import hip_util
    ref = GOLDEN["empty_intersection"]
    tape = hip_util.Tape(ref["tape"])
    for dims in [(1,), (5,), (3, 2), (1, 1, 1), (2, 3, 65)]:
        d3 = tuple(dims) + (1,) * (3 - len(dims))
        out = hip_util.Buffer(hip_util.Buffer.quad_dtype(np.float32), d3)
        c4 = np.array([0.25, -0.5, 0.125, 0], np.float32)
        hip.k.grid_eval(dims, None, tape, c4, np.float32(0.37), out).wait()
        want = oracle.grid_eval(ref["tape"], c4[:3], np.float32(0.37), d3)
        assert _same(out.read().view(np.float32).reshape(d3 + (4,)), want)
    counter = hip_util.Buffer(np.uint32, 1)
    lst = hip_util.Buffer(hip_util.Buffer.quad_dtype(np.uint8), 8 ** 3)
    counter.enqueue_fill(0)
    # far away from the (empty) shape nothing is ambiguous
    hip.k.subdivision_step((8, 8, 8), None, tape, np.array([100, 100, 100, 0], np.float32), np.float32(0.1),
                           np.float32(0.0866), counter, lst).wait()
    assert int(counter.read()[0]) == 0


def test_bad_tapes_are_rejected(hip):
    from codecad_amd import hip_util
    for bad in ([], [512.0], [99999.0, 0.0], [3584.0], [float("nan"), 0.0], [1536.5, 1, 1, 0.0]):
        with pytest.raises(RuntimeError):
            hip_util.Tape(np.array(bad, dtype=np.float32))


def test_subdivision_driver_known_answers(hip):
    """Leaf-corner known answers of reference tests/test_subdivision.py:110-161."""
    import itertools
    import codecad_amd as cc
    _, _, blocks = cc.subdivision.subdivision(cc.shapes.box(10), 1, grid_size=4, overlap_edge_samples=True)
    assert blocks[0][2] == 1 and blocks[0][4] == 1
    corners = {tuple(b[1]) for b in blocks}
    expected = set(itertools.product([-5.5, -2.5, 0.5, 3.5], repeat=3)) - set(itertools.product([-2.5, 0.5], repeat=3))
    assert corners == expected

    resolution, grid = 0.1, 8
    step = resolution * (grid - 1)
    diameter = grid * step - resolution
    radius, thr = diameter / 2, math.sqrt(2) * step / 2
    _, _, blocks = cc.subdivision.subdivision(cc.shapes.circle(diameter), resolution, grid_size=grid,
                                              overlap_edge_samples=True)
    assert blocks[0][2] == resolution and blocks[0][4] == 1
    r = [-radius - 0.5 * resolution + i * step for i in range(grid)]
    expected = {(x, y) for x, y in itertools.product(r, repeat=2)
                if radius - thr < math.hypot(x + step / 2, y + step / 2) < radius + thr}
    got = {(b[1].x, b[1].y) for b in blocks}
    assert len(got) == len(expected)
    for g in got:
        assert any(abs(g[0] - e[0]) < 1e-9 and abs(g[1] - e[1]) < 1e-9 for e in expected)


@pytest.mark.parametrize("name", sorted(shapes_zoo.mass_property_cases))
def test_mass_properties_driver_analytic(hip, name):
    """Analytic volume / centroid / inertia of reference tests/test_mass_properties.py:98-108."""
    import codecad_amd as cc
    shape, volume, centroid, inertia = shapes_zoo.mass_property_cases[name]
    precision = 2e-3
    result = cc.mass_properties(shape, 10 * precision)
    assert result.volume == pytest.approx(volume, abs=1e-4, rel=precision)
    assert tuple(result.centroid) == pytest.approx(centroid, abs=1e-4, rel=precision)
    if inertia is not None:
        assert np.allclose(result.inertia_tensor, inertia, rtol=precision)


@pytest.mark.parametrize("name", sorted(shapes_zoo.rounded_shapes))
def test_rounded_blends_match_oracle(hip, name):
    """Tapes with rounded blends run the FULL interpreter in every kernel (a direction feeds a
    distance): grid_eval, pymcubes, subdivision_step and mass_properties against the oracle."""
    from codecad_amd import hip_util, nodes
    shape = shapes_zoo.rounded_shapes[name]
    tape_f = nodes.make_program(shape)
    bb = shape.bounding_box()
    ref = {"bbox_a": list(bb.a), "bbox_b": list(bb.b), "dimension": shape.dimension()}
    corner, step, dims = _grid_for(ref, 28)
    tape = hip_util.Tape(tape_f)
    assert tape.flags & 1
    c4 = np.zeros(4, np.float32)
    c4[:3] = corner
    out = hip_util.Buffer(hip_util.Buffer.quad_dtype(np.float32), dims)
    hip.k.grid_eval(dims, None, tape, c4, step, out).wait()
    want = oracle.grid_eval(tape_f, corner, step, dims)
    got = out.read().view(np.float32).reshape(dims + (4,))
    assert _same(got, want), "max |diff| = %g" % np.nanmax(np.abs(got - want))
    assert np.any((want[..., :3] == 0).all(axis=-1)), "the grid must cross a blend region"
    flat = hip_util.Buffer(np.float32, dims)
    hip.k.grid_eval_pymcubes(dims, None, tape, c4, step, flat).wait()
    assert _same(flat.read().reshape(-1), oracle.grid_eval_pymcubes(tape_f, corner, step, dims))
    thr = np.float32(float(step) * math.sqrt(shape.dimension()) / 2)
    counter = hip_util.Buffer(np.uint32, 1)
    lst = hip_util.Buffer(hip_util.Buffer.quad_dtype(np.uint8), dims[0] * dims[1] * dims[2])
    counter.enqueue_fill(0)
    hip.k.subdivision_step(dims, None, tape, c4, step, thr, counter, lst).wait()
    want_n, want_l = oracle.subdivision_step(tape_f, corner, step, thr, dims)
    n = int(counter.read()[0])
    assert n == want_n
    assert sorted(map(tuple, lst.read().view(np.uint8).reshape(-1, 4)[:n].tolist())) == sorted(map(tuple, want_l.tolist()))
    if shape.dimension() == 3:
        sums = hip_util.Buffer(np.uint32, 10)
        sums.enqueue_fill(0)
        counter.enqueue_fill(0)
        hip.k.mass_properties(dims, None, tape, c4, step, thr, sums, counter, lst).wait()
        ws, wn, _ = oracle.mass_properties(tape_f, corner, step, thr, dims)
        assert sums.read().tolist() == ws.tolist() and int(counter.read()[0]) == wn
