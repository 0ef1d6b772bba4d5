"""Box pruning of per-tape code (csrc/specialise.hpp "BOX PRUNING", kernels.hpp k_box_masks): before a launch over 16^3 boxes
a mask kernel bounds the tape's distances over every box and marks the operands of min / max that cannot win anywhere in
it; the box kernels skip them.  Dropping a loser must not change one bit of any output -- distances, directions, survivor
lists, moment sums -- so every check here is the oracle's answer, bit for bit, on scenes built to make the pruning bite:
assemblies of many small parts far apart (most operands dead in most boxes), parts that touch, cut and contain each other
(boxes in which the decision is close), shells / offsets / scalings / mirrored and rotated frames between the selects, and
primitives the bounds know nothing about (gears, polygons, twists, circular repetitions) and plates perforated by a
repetition (bounded only in boxes that stay inside one cell of it) mixed in.  Both layouts of the dense
kernel, slabs, leaf blocks, the classification kernels over boxes; `hu_tape_prune_info` says whether a tape has anything
to prune at all (the sponge must not: it keeps the code it had)."""
import ctypes
import random

import numpy as np
import pytest

from conftest import load_golden_tapes
from test_gpu_bricks import check_classify, check_slab, run

pytestmark = pytest.mark.gpu
GOLDEN = load_golden_tapes()


def prune_info(hip, handle):
    bits, words = ctypes.c_int(0), ctypes.c_int(0)
    from codecad_amd.hip_util import check
    check(hip.lib.hu_tape_prune_info(handle.device_ptr, ctypes.byref(bits), ctypes.byref(words)), "hu_tape_prune_info")
    return bits.value, words.value


def random_part(rng):
    import codecad_amd as cc
    s = cc.shapes
    kind = rng.choice(["box", "sphere", "cylinder", "tube", "gear", "plate", "polygon", "shell", "capsule", "cone_stack", "twist", "ring_of_pins",
                       "perforated"])
    if kind == "box":
        p = s.box(rng.uniform(1, 4), rng.uniform(1, 4), rng.uniform(1, 4))
    elif kind == "sphere":
        p = s.sphere(d=rng.uniform(1, 4))
    elif kind == "cylinder":
        p = s.cylinder(h=rng.uniform(1, 5), d=rng.uniform(0.5, 3))
    elif kind == "tube":
        p = s.cylinder(h=rng.uniform(2, 5), d=3) - s.cylinder(h=10, d=rng.uniform(1, 2.5))
    elif kind == "gear":
        p = s.gears.InvoluteGear(rng.choice([9, 12, 17]), rng.choice([0.25, 0.5])).extruded(rng.uniform(0.5, 2))
    elif kind == "plate":
        p = (s.rectangle(4, 3) - s.circle(d=1).translated(1, 0.5) - s.circle(d=0.8).translated(-1, -0.5)).extruded(0.5)
    elif kind == "polygon":
        p = s.polygon2d([(0, 0), (3, 0), (3, 1), (2, 2), (3, 3), (0, 3)]).extruded(rng.uniform(0.5, 2))
    elif kind == "shell":
        p = s.sphere(d=3).shell(0.3) & s.half_space().translated_y(rng.uniform(-0.5, 0.5))
    elif kind == "capsule":
        p = s.capsule(-1, 0, 1, 0.5, 0.5).extruded(1).offset(0.2)
    elif kind == "cone_stack":
        p = s.union([s.cylinder(h=0.6, d=3 - 0.5 * i).translated_z(0.6 * i) for i in range(4)])
    elif kind == "perforated":
        # a plate with a grid of holes: the holes sit behind a REPETITION -- bounded only in boxes that stay inside one of its cells
        p = s.box(5, 4, 0.6) - s.unsafe.Repetition(s.cylinder(h=4, d=rng.uniform(0.3, 0.7)), (rng.choice([1.0, 1.5]), rng.choice([1.0, 1.25]), None))
    elif kind == "twist":
        p = s.rectangle(0.5, 1).revolved(r=1.5, twist=rng.choice([90, 180]))
    else:
        p = s.unsafe.CircularRepetition(s.cylinder(h=1, d=0.4).translated_x(1.2), rng.choice([5, 7]))
    for _ in range(rng.choice([0, 1, 1, 2])):
        op = rng.choice(["rx", "ry", "rz", "quarter", "scale", "mirror", "general"])
        if op == "rx":
            p = p.rotated_x(rng.uniform(-80, 80))
        elif op == "ry":
            p = p.rotated_y(rng.uniform(-80, 80))
        elif op == "rz":
            p = p.rotated_z(rng.uniform(-170, 170))
        elif op == "quarter":
            p = p.rotated_x(90) if rng.random() < 0.5 else p.rotated_y(-90)
        elif op == "scale":
            p = p.scaled(rng.choice([0.5, 1.5, 2]))
        elif op == "mirror":
            p = p.translated_x(0.5).mirrored_x()
        else:
            p = p.rotated((rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(0.2, 1)), rng.uniform(10, 170))
    return p


def assembly(rng, n_parts, spread):
    """A union of parts scattered over +-spread, some of them cut by a few long holes, the lot clipped by a big rounded body."""
    import codecad_amd as cc
    s = cc.shapes
    parts = [random_part(rng).translated(rng.uniform(-spread, spread), rng.uniform(-spread, spread), rng.uniform(-spread / 2, spread / 2))
             for _ in range(n_parts)]
    split = max(1, n_parts // 3)
    a = s.union(parts[:split])
    b = s.union(parts[split:]) if parts[split:] else a
    holes = s.union([s.cylinder(h=4 * spread, d=rng.uniform(0.5, 1.5)).rotated_x(90 * rng.choice([0, 1])).translated(
        rng.uniform(-spread, spread), rng.uniform(-spread, spread), 0) for _ in range(3)])
    body = (a + (b - holes)) & (s.sphere(d=3.2 * spread) + s.box(2.2 * spread, 2.2 * spread, spread))
    return body


@pytest.mark.parametrize("seed", range(10))
def test_assemblies_over_boxes(hip, seed):
    from codecad_amd import hip_util, nodes
    rng = random.Random(41000 + seed)
    spread = rng.choice([6.0, 10.0, 16.0])
    shape = assembly(rng, rng.choice([4, 8, 12, 16]), spread)
    tape = nodes.make_program(shape)

    def has_scopes(handle):
        bits, words = prune_info(hip, handle)
        assert bits > 0 and words == (bits + 31) // 32, "an assembly of bounded parts has operands to prune"
    n = 48
    step = np.float32(2.4 * spread / n)
    grids = [(np.array([-1.2 * spread, -1.2 * spread, -0.6 * spread]) + float(step) / 2, step, (n, n, n // 2)),
             (np.array([-0.37 * spread, -0.41 * spread, -0.2 * spread]), np.float32(float(step) / 5), (32, 16, 24))]      # a close-up: small boxes
    res = float(step) / 2
    blocks = [([(-40, -40, -16), (0, 0, 0), (13, -27, 5), (-64, 20, -8)], res, (0.0, 0.0, 0.0)),
              ([(-16, -16, -16)], res, (0.11 * spread, -0.07 * spread, 0.0), (32, 48, 40))]
    run(hip, tape, grids, blocks, inspect=has_scopes)


@pytest.mark.parametrize("seed", range(4))
def test_assemblies_classified_over_boxes_and_slabs(hip, seed, monkeypatch):
    from codecad_amd import hip_util, nodes
    rng = random.Random(42000 + seed)
    spread = rng.choice([6.0, 12.0])
    tape = nodes.make_program(assembly(rng, rng.choice([6, 10, 14]), spread))
    handle = hip_util.Tape(tape)
    handle.specialize(hip_util.SPEC_DENSE | hip_util.SPEC_CLASSIFY)
    assert prune_info(hip, handle)[0] > 0
    monkeypatch.setenv("HU_CLASSIFY_BOX_MIN", "1")
    step = np.float32(2.4 * spread / 32)
    corner = np.array([-1.2 * spread, -1.2 * spread, -0.6 * spread]) + float(step) / 2
    for c, st, dims in ((corner, step, (32, 32, 16)), (corner * 0.5, np.float32(float(step) / 2), (48, 20, 24)),
                        (np.array([-0.3, 0.2, -0.4]) * spread, np.float32(float(step) / 7), (16, 16, 16))):
        check_classify(hip, handle, tape, c, st, dims)
    for x0, count in ((0, 32), (12, 8), (16, 16)):
        check_slab(hip, handle, tape, corner, step, (32, 32, 16), x0, count)
    handle.release()


def test_planetary_dense_grids_and_blocks(hip, monkeypatch):
    """BASELINE C4's tape (80 (primitive, path) pairs, 9 gears, 35 extrusions) through the box kernels with its pruning masks:
    a coarse grid over the whole assembly (most of every box's tape dead), a fine grid across gear teeth, leaf blocks."""
    import codecad_amd as cc
    from codecad_amd import hip_util
    shape = cc.examples.planetary()
    tape = cc.nodes.make_program(shape)

    def has_scopes(handle):
        assert prune_info(hip, handle)[0] > 50
    grids = [(np.array([-52.0, -54.0, -2.0]), np.float32(2.2), (48, 48, 32)),
             (np.array([10.0, -8.0, 20.0]), np.float32(0.25), (32, 32, 32))]
    blocks = [([(-200, -200, 0), (0, 0, 100), (40, -120, 60), (80, 80, 200)], 0.25, (0.0, 0.0, 0.0)),
              ([(-64, -64, 0)], 0.5, (0.0, 0.0, 10.0), (32, 32, 32))]
    run(hip, tape, grids, blocks, inspect=has_scopes)


def test_tapes_without_bounds_have_nothing_to_prune(hip):
    """A tape that is mostly repetitions (the sponge: twelve of its thirteen primitives sit behind one) is left alone -- in a box its
    primitives are table reads, cheaper than the tests that would skip them --: no scopes, no mask kernel, the code it had."""
    import codecad_amd as cc
    from codecad_amd import hip_util
    for shape in (cc.examples.sponge(3), cc.shapes.sphere(3)):
        handle = hip_util.Tape(cc.nodes.make_program(shape))
        handle.specialize(hip_util.SPEC_DENSE)
        assert prune_info(hip, handle) == (0, 0)
        handle.release()


def test_pruned_launches_inside_a_hipgraph(hip):
    """A launch over boxes with its mask kernel, captured into a hipGraph and replayed: the mask buffer belongs to the tape
    handle and is only grown outside a capture (hip_util.hip prepare_masks) -- a capture that finds it too small runs that
    launch unpruned --, so both a capture after a warm launch (masks inside the graph) and a capture on a stream that never
    launched (no masks) must give the oracle's bits."""
    import torch
    import oracle
    import codecad_amd as cc
    from codecad_amd import hip_util
    from codecad_amd.hip_util import check
    from conftest import same_bits
    shape = cc.examples.planetary()
    host_tape = cc.nodes.make_program(shape)
    tape = hip_util.Tape(host_tape, policy="0")
    tape.specialize(hip_util.SPEC_DENSE)
    assert prune_info(hip, tape)[0] > 50
    n = 32
    corner = np.array([8.0, -10.0, 18.0, 0.0], np.float32)
    step = np.float32(0.5)
    dims = (ctypes.c_uint32 * 3)(n, n, n)
    fptr = ctypes.POINTER(ctypes.c_float)
    want = oracle.grid_eval(host_tape, corner[:3], step, (n, n, n), threads=4)
    dev = torch.device("cuda", 0)
    for warm in (True, False):
        stream = torch.cuda.Stream(device=dev)
        out = torch.full((n, n, n, 4), float("nan"), dtype=torch.float32, device=dev)

        def launch():
            check(hip.lib.hu_grid_eval_slab(tape.device_ptr, corner.ctypes.data_as(fptr), step, dims, 0, n, 0, out.data_ptr(), stream.cuda_stream), "slab")
        if warm:
            launch()
            stream.synchronize()
            assert same_bits(out.cpu().numpy(), want)
            out.fill_(float("nan"))
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            launch()
        for _ in range(2):
            out.fill_(float("nan"))
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            assert same_bits(out.cpu().numpy(), want), warm
    tape.release()
