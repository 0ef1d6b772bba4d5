"""Per-brick culling of the dense interpreter kernels (csrc/cull.hpp, kernels.hpp k_brick_keep / k_grid_eval_culled).

Culling skips operands that cannot win inside a brick; what is evaluated is evaluated by the same code, so every
float must still equal the oracle's (bit for bit, NaN == NaN).  Grids here have extents that are multiples of 8,
so that the culled kernels run (other extents take the plain kernels), and steps from coarse -- a brick spans whole
primitives, little is culled -- to fine -- most operands are out."""
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


def listing(tape, which):
    from codecad_amd.hip_util import _lib
    lib = _lib.load()
    tape = np.ascontiguousarray(tape, dtype=np.float32)
    ptr = tape.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    need = ctypes.c_size_t()
    assert lib.hu_tape_listing(ptr, tape.size, which, None, 0, ctypes.byref(need)) == 0
    buf = ctypes.create_string_buffer(need.value)
    assert lib.hu_tape_listing(ptr, tape.size, which, buf, need.value, ctypes.byref(need)) == 0
    return buf.value.decode()


def check_grids(hip, tape, grids):
    from codecad_amd import hip_util
    handle = hip_util.Tape(tape)
    for corner, step, dims in grids:
        c4 = np.zeros(4, np.float32)
        c4[:3] = corner
        want = oracle.grid_eval(tape, corner, step, dims)
        out = hip_util.Buffer(hip_util.Buffer.quad_dtype(np.float32), dims)
        hip.k.grid_eval(dims, None, handle, c4, step, out).wait()
        got = out.read().view(np.float32).reshape(dims + (4,))
        assert same_bits(got, want), "float4 grid differs at step %g: max |diff| = %g" % (step, np.nanmax(np.abs(got - want)))
        want_w = oracle.grid_eval_pymcubes(tape, corner, step, dims)
        outw = hip_util.Buffer(np.float32, dims)
        hip.k.grid_eval_pymcubes(dims, None, handle, c4, step, outw).wait()
        assert same_bits(outw.read().reshape(-1), want_w), "distance grid differs at step %g" % step
        out.release()
        outw.release()
    handle.release()


def grids_around(ref, n, zooms):
    """n^3 grids centred on the shape's bounding box: the whole box (zoom 1) and ever smaller parts of it"""
    a, b = np.array(ref["bbox_a"], dtype=np.float64), np.array(ref["bbox_b"], dtype=np.float64)
    a = np.where(np.isfinite(a), a, -2.0)
    b = np.where(np.isfinite(b), b, 2.0)
    size = float(np.max(b - a)) * 1.2 + 1e-3
    mid = (a + b) / 2
    out = []
    for zoom in zooms:
        step = np.float32(size / zoom / n)
        # off-centre for the zoomed ones: towards a corner of the box, where surfaces are
        centre = mid + (0.0 if zoom == 1 else 0.3) * (b - a) * np.array([1.0, -1.0, 1.0])
        out.append((centre - float(step) * n / 2 + float(step) / 2, step, (n, n, n)))
    return out


@pytest.mark.parametrize("name", ZOO_3D)
def test_culled_grid_eval_matches_oracle_on_the_zoo(hip, name):
    ref = GOLDEN[name]
    check_grids(hip, ref["tape"], grids_around(ref, 16, (1, 4, 20)))


@pytest.mark.parametrize("seed", range(40))
def test_culled_grid_eval_matches_oracle_on_random_trees(hip, seed):
    from codecad_amd import nodes
    rng = random.Random(5000 + seed)
    shape = random_3d(rng, rng.choice([2, 3, 4]))
    tape = nodes.make_program(shape)
    grids = [(np.array([-4.0, -4.0, -4.0]), np.float32(0.5), (16, 16, 16)),           # exact zeros and symmetric pairs
             (np.array([-1.03, -0.97, -1.11]), np.float32(0.11), (16, 24, 16)),
             (np.array([0.21, -0.4, 0.13]), np.float32(0.013), (24, 16, 16))]
    check_grids(hip, tape, grids)


def test_culled_grid_eval_of_the_sponge_at_bench_resolution(hip):
    """A 64 x 64 x 128 window of the 512^3 bench grid of sponge(4): the resolution at which most operands are out."""
    import codecad_amd as cc
    from codecad_amd import nodes
    tape = nodes.make_program(cc.examples.sponge(4))
    text = listing(tape, 4)
    assert "culling on, 12 selects" in text
    step = np.float32(1.0 / 512)
    corner = np.array([-0.5 + 0.5 / 512 + 100 / 512, -0.5 + 0.5 / 512 + 200 / 512, -0.5 + 0.5 / 512 + 64 / 512])
    check_grids(hip, tape, [(corner, step, (64, 64, 128))])


def test_selects_a_tape_cannot_bound_are_left_alone():
    """Listing 4 shows what the decoder worked out: a twisted revolution has no Lipschitz bound here."""
    import codecad_amd as cc
    from codecad_amd import nodes, shapes
    tape = nodes.make_program(shapes.rectangle(1, 2).revolved(r=3, twist=180) + shapes.sphere(1))
    assert "La+Lb: inf" in listing(tape, 4)
    assert "culling off" in listing(tape, 4)
