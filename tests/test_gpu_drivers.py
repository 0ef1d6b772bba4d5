"""GPU drivers (level-batched kernels, torch interop, full-size properties) against the oracle."""
import ctypes
import math
import os

import numpy as np
import pytest

import oracle
import ref_driver
import shapes_zoo
from codecad_amd import util
from conftest import load_golden_tapes, same_bits

pytestmark = pytest.mark.gpu
GOLDEN = load_golden_tapes()


def _bbox(ref):
    return util.BoundingBox(util.Vector(*ref["bbox_a"]), util.Vector(*ref["bbox_b"]))


def _tape_shape(name):
    from codecad_amd.shapes import TapeShape
    ref = GOLDEN[name]
    return TapeShape(ref["tape"], _bbox(ref), float(ref["feature_size"]), ref["dimension"])


SUBDIV_CASES = [("sponge2", 1 / 54, 4, True), ("sponge2", 1 / 27, 8, False), ("csg_example", 2.0, 8, True),
                ("kat_box10", 1, 4, True), ("kat_circle", 0.1, 8, True), ("gear", 0.05, 8, True),
                ("torus", 0.1, 16, True), ("sponge4", 1 / 512, 16, True), ("planetary", 1.0, 8, True)]


@pytest.mark.parametrize("name, resolution, grid, overlap", SUBDIV_CASES)
def test_level_batched_subdivision_equals_per_block_reference_traversal(hip, name, resolution, grid, overlap):
    """One launch per LEVEL (GPU) == the reference's block-by-block traversal run on the oracle:
    identical leaf-block sets (integer corners), identical leaf dims."""
    import codecad_amd as cc
    ref = GOLDEN[name]
    want_dims, want = ref_driver.subdivision(ref["tape"], _bbox(ref), ref["dimension"], resolution, overlap, grid)
    leaves = cc.subdivision.subdivision_device(_tape_shape(name), resolution, overlap, grid)
    got = leaves.int_corners()
    assert tuple(int(d) for d in leaves.dims) == tuple(int(d) for d in want_dims)
    assert leaves.count == len(want)
    assert sorted(map(tuple, got.tolist())) == sorted(b[2] for b in want)
    # and through the reference-shaped return value
    _, dims, blocks = cc.subdivision.subdivision(_tape_shape(name), resolution, overlap, grid)
    assert len(blocks) == len(want)
    by_int = {b[2]: b for b in want}
    for bdims, corner, step, icorner, istep in blocks[:200]:
        w = by_int[tuple(icorner)]
        assert tuple(corner) == tuple(w[0]) and step == w[1] and istep == w[3]


MASS_CASES = [("sponge2", 1 / 27, 3), ("sponge2", 1 / 54, 8), ("csg_example", 2.0, 8), ("mp_drunk_box", 0.05, 16),
              ("planetary", 2.0, 8), ("mirror_3d", 0.1, 16)]


@pytest.mark.parametrize("name, resolution, grid", MASS_CASES)
def test_level_batched_mass_properties_equals_reference_traversal(hip, name, resolution, grid):
    """Same integer moment sums per block => volume / centroid / inertia agree to summation
    order (1e-12 relative; the north star asks for 1e-5)."""
    import codecad_amd as cc
    ref = GOLDEN[name]
    want, evaluations = ref_driver.mass_properties(ref["tape"], _bbox(ref), resolution, grid)
    got = cc.mass_properties(_tape_shape(name), resolution, grid)
    assert cc.mass_properties.last_stats["function_evaluations"] == evaluations
    assert got.volume == pytest.approx(want.volume, rel=1e-12)
    assert tuple(got.centroid) == pytest.approx(tuple(want.centroid), rel=1e-10, abs=1e-12)
    assert np.allclose(got.inertia_tensor, want.inertia_tensor, rtol=1e-10, atol=1e-9 * abs(want.volume))


def test_sponge_volume_is_exact_at_aligned_resolution(hip):
    """At resolution 3^-k the cell centres never touch the surface: volume == (20/27)^n."""
    import codecad_amd as cc
    for n, res in ((1, 1 / 9), (2, 1 / 27), (3, 1 / 81)):
        mp = cc.mass_properties(cc.examples.sponge(n), res, grid_size=9)
        assert mp.volume == pytest.approx((20 / 27) ** n, rel=1e-12)
        assert tuple(mp.centroid) == pytest.approx((0, 0, 0), abs=1e-12)


def test_leaf_block_grid_eval_matches_oracle(hip):
    import codecad_amd as cc
    shape = cc.examples.sponge(3)
    tape = cc.nodes.make_program(shape)
    leaves = cc.subdivision.subdivision_device(shape, 1 / 128, True, 8)
    corners = leaves.int_corners()
    assert leaves.count == len(corners) > 50
    # int_corners() sorts; re-upload in that order so block i is corners[i]
    from codecad_amd import hip_util
    blocks = np.zeros((len(corners), 4), np.int32)
    blocks[:, :3] = corners
    leaves.blocks.release()
    leaves.blocks = hip_util.Buffer(np.int32, blocks.shape)
    leaves.blocks.enqueue_write(blocks)
    dims = tuple(int(d) for d in leaves.dims)
    g4 = cc.grid_eval.grid_eval_blocks(leaves).read().view(np.float32).reshape((len(corners),) + dims + (4,))
    g1 = cc.grid_eval.grid_eval_blocks(leaves, pymcubes=True).read().reshape(len(corners), -1)
    for i in list(range(0, len(corners), max(1, len(corners) // 25))):
        corner = util.Vector(*corners[i].tolist()) * leaves.resolution + leaves.origin
        c32 = np.array(tuple(corner), np.float64).astype(np.float32)
        assert same_bits(g4[i], oracle.grid_eval(tape, c32, np.float32(leaves.step), dims))
        assert same_bits(g1[i], oracle.grid_eval_pymcubes(tape, c32, np.float32(leaves.step), dims).reshape(g1[i].shape))


def _dense_torch(hip, tape_obj, n, x0=0, x_count=None, layout=0):
    import torch
    x_count = n if x_count is None else x_count
    step = np.float32(1.0 / n)
    corner = np.array([-0.5 + 0.5 / n] * 3 + [0.0], np.float32)
    dims = (ctypes.c_uint32 * 3)(n, n, n)
    out = torch.empty((x_count, n, n, 4) if layout == 0 else (n, n, n), dtype=torch.float32, device="cuda")
    from codecad_amd.hip_util import check
    check(hip.lib.hu_grid_eval_slab(tape_obj.device_ptr, corner.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                    step, dims, x0, x_count, layout, out.data_ptr(), None), "hu_grid_eval_slab")
    torch.cuda.synchronize()
    return out, corner, step


@pytest.mark.parametrize("specialise", [False, True])
def test_full_size_512_dense_properties(hip, specialise):
    """BASELINE size (512^3 sponge(4), 2 GiB of float4): slabs tile the grid exactly, a random
    sample of 200k voxels equals the oracle bit for bit, the inside fraction is the sponge's
    volume, and the field is mirror-symmetric.  Both evaluators, each PINNED (policy "0": no background build can swap
    the evaluator under the test; specialise=True is the launch bench.py times -- k_grid_eval<JitEval, 0, 2> over boxes)."""
    import torch
    import codecad_amd as cc
    from codecad_amd import hip_util
    n = 512
    shape = cc.examples.sponge(4)
    tape = hip_util.Tape(cc.nodes.make_program(shape), policy="0")
    if specialise:
        tape.specialize()
    assert bool(tape.specialized) == specialise
    whole, corner, step = _dense_torch(hip, tape, n)
    # (a) x-slab sharding (what each rank of an 8-GPU job computes) reproduces the whole grid
    for x0, cnt in ((0, 64), (448, 64), (200, 37)):
        slab, _, _ = _dense_torch(hip, tape, n, x0, cnt)
        assert torch.equal(slab, whole[x0:x0 + cnt])
    # (b) sampled bit-parity with the oracle
    rng = np.random.default_rng(3)
    idx = rng.integers(0, n, size=(200000, 3))
    pts = corner[:3][None, :] + step * idx.astype(np.float32)
    want = oracle.evaluate_points(tape.host_tape, pts)
    ti = torch.from_numpy(idx).cuda()
    got = whole[ti[:, 0], ti[:, 1], ti[:, 2]].cpu().numpy()
    assert same_bits(got, want)
    # (c) volume fraction: cell centres of a 512 grid are never on a sponge(4) face
    inside = float((whole[..., 3] <= 0).double().mean().item())
    assert inside == pytest.approx((20 / 27) ** 4, rel=2e-3)
    # (d) the distance field is symmetric under x -> -x, y -> -y, z -> -z and axis swaps
    w = whole[..., 3]
    assert torch.allclose(w, w.flip(0), atol=1e-6) and torch.allclose(w, w.flip(2), atol=1e-6)
    assert torch.allclose(w, w.permute(2, 1, 0), atol=1e-6)
    # pymcubes layout of the same grid: out[z + (x + (n-1-y)*n)*n]
    flat, _, _ = _dense_torch(hip, tape, n, layout=1)
    assert torch.equal(flat.reshape(n, n, n), w.permute(1, 0, 2).flip(0).contiguous())


def test_more_than_2_pow_30_cells_in_one_call(hip):
    """The C ABI splits launches at 2^30 cells; indices stay exact across the seam."""
    import torch
    from codecad_amd import hip_util
    from codecad_amd.hip_util import check
    ref = GOLDEN["sphere_plus_box"]
    tape = hip_util.Tape(ref["tape"])
    dims_t = (1030, 1024, 1024)                      # 1.08e9 cells > 2^30
    dims = (ctypes.c_uint32 * 3)(*dims_t)
    step = np.float32(0.13)
    corner = np.array([-66.0, -66.0, -66.0, 0.0], np.float32)
    out = torch.empty(dims_t[0] * dims_t[1] * dims_t[2], dtype=torch.float32, device="cuda")
    check(hip.lib.hu_grid_eval_pymcubes(tape.device_ptr, corner.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), step,
                                        dims, out.data_ptr(), None), "hu_grid_eval_pymcubes")
    torch.cuda.synchronize()
    rng = np.random.default_rng(5)
    idx = np.concatenate([rng.integers(0, 1024, size=(50000, 3)),
                          np.array([[1023, 0, 0], [1024, 1023, 1023], [1029, 5, 7], [1024, 0, 0]])])
    idx[:50000, 0] = rng.integers(0, 1030, size=50000)
    pts = corner[:3][None, :] + step * idx.astype(np.float32)
    want = oracle.evaluate_points(ref["tape"], pts)[:, 3]
    lin = idx[:, 2] + (idx[:, 0] + (dims_t[1] - 1 - idx[:, 1]) * dims_t[0]) * dims_t[2]
    got = out[torch.from_numpy(lin).cuda()].cpu().numpy()
    assert same_bits(got, want)


def test_object_tags_and_torch_interop(hip):
    """N objects in one parent list (the multi-GPU layout of bench.py) == N single runs;
    parents/children live in torch tensors passed by data_ptr()."""
    import torch
    import codecad_amd as cc
    from codecad_amd import dist
    from codecad_amd.hip_util import check
    shape = cc.examples.sponge(2)
    tape = cc.nodes.make_program_buffer(shape)
    res = 1 / 54
    box = shape.bounding_box().expanded_additive(res / 2)
    levels = cc.subdivision.calculate_block_sizes(box, 3, res, 8, True)
    origin = (ctypes.c_double * 3)(box.a.x, box.a.y, box.a.z)
    counter = torch.zeros(1, dtype=torch.int32, device="cuda")

    def classify(level, parents):
        int_step, d = levels[level]
        dd = (ctypes.c_uint32 * 3)(int(d[0]), int(d[1]), int(d[2]))
        cap = int(parents.shape[0]) * int(d[0]) * int(d[1]) * int(d[2])
        children = torch.empty((cap, 4), dtype=torch.int32, device="cuda")
        counter.zero_()
        torch.cuda.synchronize()
        check(hip.lib.hu_subdivision_level(tape.device_ptr, parents.contiguous().data_ptr(), int(parents.shape[0]),
                                           int(int_step), dd, 3, res, origin, np.float32(int_step * res),
                                           np.float32(int_step * res * math.sqrt(3) / 2), counter.data_ptr(),
                                           children.data_ptr(), cap, None), "hu_subdivision_level")
        torch.cuda.synchronize()
        return children[:int(counter.item())]

    top = torch.zeros((3, 4), dtype=torch.int32, device="cuda")
    top[:, 3] = torch.arange(3, dtype=torch.int32)
    leaves, counts = dist.run_levels(top, len(levels) - 1, classify)
    rows = leaves.cpu().numpy()
    single = cc.subdivision.subdivision_device(shape, res, True, 8).int_corners()
    for obj in range(3):
        mine = rows[rows[:, 3] == obj][:, :3]
        assert sorted(map(tuple, mine.tolist())) == sorted(map(tuple, single.tolist()))
    assert counts[-1] == 3 * len(single)


def test_overflowing_child_list_is_detected_and_retried(hip):
    """capacity < survivors: the counter still reports the true count, nothing is written past
    the end, and the driver's retry path gives the full set."""
    import codecad_amd as cc
    from codecad_amd import hip_util, subdivision
    shape = cc.examples.sponge(2)
    tape = cc.nodes.make_program_buffer(shape)
    res = 1 / 54
    box = shape.bounding_box().expanded_additive(res / 2)
    levels = subdivision.calculate_block_sizes(box, 3, res, 8, True)
    parents = hip_util.Buffer(np.int32, (1, 4))
    parents.enqueue_write(np.zeros((1, 4), np.int32))
    counter = hip_util.Buffer(np.uint32, 1)
    full, n_full = subdivision._level_launch(tape, parents, 1, levels[0][0], levels[0][1], 3, res, box.a, counter,
                                             hip.queue)
    small, n_small = subdivision._level_launch(tape, parents, 1, levels[0][0], levels[0][1], 3, res, box.a, counter,
                                               hip.queue, capacity_hint=3)
    assert n_small == n_full > 3
    a = np.empty((full.shape[0], 4), np.int32)
    b = np.empty((small.shape[0], 4), np.int32)
    full.read(out=a)
    small.read(out=b)
    assert sorted(map(tuple, a[:n_full].tolist())) == sorted(map(tuple, b[:n_small].tolist()))


def test_buffer_surface(hip):
    """Buffer semantics of reference tests/test_clutil.py:18-111 that apply without OpenCL."""
    from codecad_amd import hip_util
    for shape, nitems in ((4, 4), ((4,), 4), ((4, 4), 16), ((4, 4, 4), 64)):
        for dtype, size in ((hip_util.Buffer.quad_dtype(np.uint32), 16), (np.uint8, 1), (np.float64, 8)):
            b = hip_util.Buffer(dtype, shape)
            b.create_host_side_array()
            assert b.nitems == nitems == len(b) and b.size == nitems * size and b.array.nbytes == b.size
    b = hip_util.Buffer(np.uint64, 1)
    b.create_host_side_array()
    b[0] = 42
    ev = b.enqueue_write()
    ev.wait()          # asynchronous copy from the pinned shadow: do not touch it before this
    b[0] = 0
    assert b.read(wait_for=[ev])[0] == 42
    with b.map(hip_util.map_flags.WRITE_INVALIDATE_REGION) as m:
        m[0] = 31
    with b.map(hip_util.map_flags.READ) as m:
        assert m[0] == 31
    with pytest.raises(RuntimeError):
        b.read(out=np.zeros(1, np.uint8))
    with pytest.raises(RuntimeError):
        b.enqueue_write(np.zeros(4, np.uint64))
    ev = b.enqueue_fill(0)
    assert ev.profile.end >= ev.profile.start
    assert b.read()[0] == 0
    b.release()
    b.release()


def test_device_integrals_match_host_formulas(hip):
    """hu_mass_integrals (fp64, deterministic device reduction) == the numpy evaluation of the
    reference's per-block formulas, on random sums/corners; and it is run-to-run identical."""
    from codecad_amd import hip_util
    from codecad_amd.hip_util import check
    from codecad_amd.mass_properties import integrals_host, _KEYS
    rng = np.random.default_rng(11)
    for n in (1, 7, 1024, 50021):
        sums = rng.integers(0, 2 ** 20, size=(n, 10), dtype=np.uint32)
        corners = np.zeros((n, 4))
        corners[:, :3] = rng.standard_normal((n, 3)) * 10
        s = 0.0371
        want = integrals_host(sums, corners[:, :3], s)
        pb = hip_util.Buffer(np.float64, (n, 4))
        sb = hip_util.Buffer(np.uint32, (n, 10))
        pb.enqueue_write(corners)
        sb.enqueue_write(sums)
        scale = max(abs(want[k]) for k in _KEYS) + 1e-300
        for rows in (1, 3, 64, 1000):          # one workgroup, ragged slices, more rows than slices hold parents
            ob = hip_util.Buffer(np.float64, (rows, 10))
            results = []
            for _ in range(2):
                ob.enqueue_fill(0xff)           # every row must be written, also the empty ones
                check(hip.lib.hu_mass_integrals(pb.device_ptr, sb.device_ptr, n, s, ob.device_ptr, rows, hip.queue.handle),
                      "integrals")
                results.append(ob.read().copy())
            assert np.array_equal(results[0], results[1])
            for k, v in zip(_KEYS, results[0].sum(axis=0)):
                assert v == pytest.approx(want[k], rel=1e-12, abs=1e-13 * scale * n)
            ob.release()
        assert hip.lib.hu_mass_integrals(pb.device_ptr, sb.device_ptr, n, s, pb.device_ptr, 0, hip.queue.handle) != 0
        # the indirect form: the count on the device, the launch sized for a larger capacity -> the same rows
        nb = hip_util.Buffer(np.uint32, 1)
        nb.enqueue_write(np.array([n], np.uint32))
        big_p, big_s = hip_util.Buffer(np.float64, (n + 100, 4)), hip_util.Buffer(np.uint32, (n + 100, 10))
        big_p.enqueue_fill(0xff)
        big_s.enqueue_fill(0xff)
        check(hip.lib.hu_memcpy_d2d(big_p.device_ptr, pb.device_ptr, n * 32, hip.queue.handle), "copy")
        check(hip.lib.hu_memcpy_d2d(big_s.device_ptr, sb.device_ptr, n * 40, hip.queue.handle), "copy")
        for rows in (1, 3, 64):
            direct, indirect = hip_util.Buffer(np.float64, (rows, 10)), hip_util.Buffer(np.float64, (rows, 10))
            check(hip.lib.hu_mass_integrals(pb.device_ptr, sb.device_ptr, n, s, direct.device_ptr, rows, hip.queue.handle), "integrals")
            check(hip.lib.hu_mass_integrals_indirect(big_p.device_ptr, big_s.device_ptr, nb.device_ptr, n + 100, s, indirect.device_ptr, rows,
                                                     hip.queue.handle), "integrals")
            assert np.array_equal(direct.read(), indirect.read())


def _wide_tape(k):
    """A hand-written tape that keeps k results live at once: k translated spheres are evaluated and
    stored in registers 0..k-1, then folded with unions.  (Exercises big LDS register files: the
    reference would just index its 512-entry private array.)"""
    t = []
    for i in range(k):
        t += [11 * 512, 0, 0, 0, 1, -0.37 * i, 0.11 * i, -0.05 * i]   # initial_transformation_to(translate)
        t += [7 * 512, 0.3 + 0.01 * i]                                  # sphere(r)
        t += [1 * 512 + i]                                              # _store i
    t += [2 * 512 + 0]                                                  # _load 0
    for i in range(1, k):
        t += [26 * 512 + i, -1.0]                                       # union(r=-1; reg i)
    t += [0]
    return np.array(t, dtype=np.float32)


@pytest.mark.parametrize("k", [3, 12, 40, 100, 159])
def test_wide_register_files(hip, k):
    """3..159 simultaneously live values: every workgroup-size / voxels-per-lane choice of
    launch_shape, the full and the distance-only program, against the oracle."""
    from codecad_amd import hip_util
    tape_f = _wide_tape(k)
    tape = hip_util.Tape(tape_f)
    assert tape.n_registers == k
    dims = (20, 9, 7)
    c4 = np.array([-1.0, -1.5, -1.2, 0], np.float32)
    step = np.float32(0.93 * k / 20 + 0.15)
    out = hip_util.Buffer(hip_util.Buffer.quad_dtype(np.float32), dims)
    hip.k.grid_eval(dims, None, tape, c4, step, out).wait()
    assert same_bits(out.read().view(np.float32).reshape(dims + (4,)), oracle.grid_eval(tape_f, c4[:3], step, dims))
    flat = hip_util.Buffer(np.float32, dims)
    hip.k.grid_eval_pymcubes(dims, None, tape, c4, step, flat).wait()
    assert np.array_equal(flat.read().reshape(-1), oracle.grid_eval_pymcubes(tape_f, c4[:3], step, dims))


def test_too_many_live_values_is_a_clean_error(hip):
    from codecad_amd import hip_util
    tape = hip_util.Tape(_wide_tape(200))
    out = hip_util.Buffer(hip_util.Buffer.quad_dtype(np.float32), (4, 4, 4))
    with pytest.raises(RuntimeError, match="live"):
        hip.k.grid_eval((4, 4, 4), None, tape, np.zeros(4, np.float32), np.float32(1), out)
    # the distance-only program needs 4 bytes per result, so the .w-only kernels still run
    flat = hip_util.Buffer(np.float32, (4, 4, 4))
    hip.k.grid_eval_pymcubes((4, 4, 4), None, tape, np.zeros(4, np.float32), np.float32(1), flat).wait()
    assert np.array_equal(flat.read().reshape(-1), oracle.grid_eval_pymcubes(_wide_tape(200), [0, 0, 0], np.float32(1), (4, 4, 4)))
    # and specialised code has no LDS register file at all
    small = hip_util.Tape(_wide_tape(24)).specialize()
    hip.k.grid_eval((4, 4, 4), None, small, np.zeros(4, np.float32), np.float32(1), out).wait()
    assert same_bits(out.read().view(np.float32).reshape(4, 4, 4, 4), oracle.grid_eval(_wide_tape(24), [0, 0, 0], np.float32(1), (4, 4, 4)))


def test_launches_are_graph_capturable(hip):
    """include/hip_util.h promises launches neither allocate nor synchronise: capture the dense
    kernel and a level launch into a hipGraph (through torch) and replay it."""
    import torch
    import codecad_amd as cc
    from codecad_amd.hip_util import check
    shape = cc.examples.sponge(2)
    tape = cc.nodes.make_program_buffer(shape)
    n = 48
    corner = np.array([-0.5 + 0.5 / n] * 3 + [0.0], np.float32)
    dims = (ctypes.c_uint32 * 3)(n, n, n)
    out = torch.zeros((n, n, n, 4), dtype=torch.float32, device="cuda")
    cptr = corner.ctypes.data_as(ctypes.POINTER(ctypes.c_float))

    def launch(stream):
        check(hip.lib.hu_grid_eval(tape.device_ptr, cptr, np.float32(1.0 / n), dims, out.data_ptr(), stream), "hu_grid_eval")

    launch(None)                      # warm-up outside capture (one-time function attributes)
    torch.cuda.synchronize()
    want = out.clone()
    out.zero_()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(graph, stream=side):
        launch(side.cuda_stream)
    torch.cuda.synchronize()
    assert float(out.abs().sum()) == 0.0          # captured, not executed
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)


@pytest.mark.gpu
def test_device_sort_of_block_lists(hip):
    """hu_sort_blocks orders int4 rows by (x, y, z) like numpy.lexsort; tags travel with their rows."""
    import ctypes
    import numpy as np
    from codecad_amd import hip_util
    rng = np.random.default_rng(9)
    for n in (2, 3, 257, 30800, 200001):
        rows = rng.integers(-5000, 5000, size=(n, 4)).astype(np.int32)
        rows[:, 3] = np.arange(n)
        rows[: n // 3, 0] = rows[0, 0]              # ties on x, then on (x, y)
        rows[: n // 7, 1] = rows[0, 1]
        # (x, y, z) must be unique for a unique expected order
        _, first = np.unique(rows[:, :3], axis=0, return_index=True)
        rows = rows[np.sort(first)]
        m = len(rows)
        buf = hip_util.Buffer(np.int32, (m, 4))
        buf.enqueue_write(rows).wait()
        needed = ctypes.c_size_t(0)
        assert hip.lib.hu_sort_blocks(buf.device_ptr, m, None, 0, ctypes.byref(needed), None) == 0
        scratch = hip_util.Buffer(np.uint8, needed.value)
        assert hip.lib.hu_sort_blocks(buf.device_ptr, m, scratch.device_ptr, needed.value, ctypes.byref(needed),
                                      hip.queue.handle) == 0, hip.lib.hu_last_error()
        want = rows[np.lexsort((rows[:, 2], rows[:, 1], rows[:, 0]))]
        assert np.array_equal(buf.read(), want)
        scratch.release()
        buf.release()
    # corners that do not fit the 21-bit key fields are refused, not mis-sorted
    rows = np.array([[0, 0, 0, 0], [1 << 21, 0, 0, 1]], dtype=np.int32)
    buf = hip_util.Buffer(np.int32, (2, 4))
    buf.enqueue_write(rows).wait()
    scratch = hip_util.Buffer(np.uint8, 1 << 20)
    needed = ctypes.c_size_t(0)
    assert hip.lib.hu_sort_blocks(buf.device_ptr, 2, scratch.device_ptr, 1 << 20, ctypes.byref(needed), hip.queue.handle) != 0
    assert b"2^20" in hip.lib.hu_last_error()


@pytest.mark.gpu
def test_config_c5_sponge5_at_2048_single_gpu_form(hip):
    """BASELINE config C5 (sponge(5), effective 2048^3, grid 16) in its single-GPU form: the level-batched
    traversal against the per-block reference traversal on the oracle (sorted leaf corners equal), and the
    size-independent property that the leaf blocks cover exactly the ambiguous cells' volume."""
    import ref_driver
    from codecad_amd import examples, nodes, subdivision
    shape = examples.sponge(5)
    res = 1.0 / 2048
    leaves = subdivision.subdivision_device(shape, res, grid_size=16).sort()
    assert leaves.level_counts == [704, 1340721] and tuple(int(d) for d in leaves.dims) == (16, 16, 16)
    got = leaves.int_corners()
    dims, want = ref_driver.subdivision(nodes.make_program(shape), shape.bounding_box(), 3, res, overlap=True, grid_size=16)
    want = np.array(sorted(b[2] for b in want), dtype=np.int32)
    assert tuple(int(d) for d in dims) == (16, 16, 16)
    assert np.array_equal(got, want)
    # every leaf corner sits on the lattice of its level (15 cells apart: blocks share their edge samples)
    assert np.all(got % 15 == 0)
    leaves.blocks.release()


def test_slice_rows_kernel_equals_the_reference_rule(hip):
    """hu_slice_rows (the device side of the multi-GPU level exchange) == dist.slice_rows_reference, for 16- and
    32-byte rows, ragged pieces, overflowing pieces and shares."""
    import torch
    from codecad_amd import dist
    from codecad_amd.hip_util import check
    lib = hip.lib
    rng = np.random.default_rng(11)
    dev = torch.device("cuda", 0)
    for world, piece_rows, k, dtype, cap in ((1, 9, 4, torch.int32, 8), (2, 700, 4, torch.int32, 699), (3, 5, 4, torch.int32, 3),
                                             (8, 300, 4, torch.int32, 299), (5, 40, 4, torch.float64, 39), (2, 6, 4, torch.int32, 1)):
        counts = rng.integers(0, piece_rows + 3, world)     # some headers claim more than a piece holds
        g = torch.zeros((world, piece_rows, k), dtype=dtype)
        for r in range(world):
            body = torch.from_numpy(rng.integers(-1000, 1000, (piece_rows - 1, k))).to(dtype)
            g[r, 1:] = body
            if dtype == torch.float64:
                g[r, 0].view(torch.int32)[0] = int(counts[r])
            else:
                g[r, 0, 0] = int(counts[r])
        for rank in range(world):
            want, wstats = torch.zeros((cap + 1, k), dtype=dtype), torch.zeros(2, dtype=torch.int32)
            gi = g if dtype != torch.float64 else g
            if dtype == torch.float64:
                # the reference rule reads the count as the tensor's element 0: give it the int view's value
                gref = g.clone()
                gref[:, 0, 0] = torch.tensor([float(c) for c in counts], dtype=torch.float64)
                dist.slice_rows_reference(gref, rank, want, wstats)
            else:
                dist.slice_rows_reference(gi, rank, want, wstats)
            gd = g.to(dev)
            out = torch.full((cap + 1, k), 77, dtype=dtype, device=dev)
            stats = torch.zeros(2, dtype=torch.int32, device=dev)
            check(lib.hu_slice_rows(gd.data_ptr(), world, piece_rows, k * g.element_size(), rank, out.data_ptr(), cap,
                                    stats.data_ptr(), None), "hu_slice_rows")
            torch.cuda.synchronize()
            n = int(want[0, 0]) if dtype != torch.float64 else int(want[0, 0].item())
            got_n = int(out[0].cpu().view(torch.int32)[0])
            assert got_n == n and stats.cpu().tolist() == wstats.tolist()
            assert torch.equal(out[1:1 + n].cpu(), want[1:1 + n])


@pytest.mark.parametrize("name, resolution, grid", [("sponge3", 1 / 243, 9), ("sponge4", 1 / 512, 16), ("csg_example", 1.0, 8)])
def test_pipeline_without_host_round_trips_equals_the_level_driver(hip, name, resolution, grid):
    """dist.LevelPipeline over hu_subdivision_level_indirect (list lengths read on the device, launches sized for
    the capacities) gives the leaf set of subdivision_device; the leaf blocks evaluated through
    hu_grid_eval_blocks_indirect equal the direct launch; a capacity that is too small is reported."""
    import torch
    import codecad_amd as cc
    from codecad_amd import dist, hip_util
    from codecad_amd.hip_util import check
    shape = _tape_shape(name)
    single = cc.subdivision.subdivision_device(shape, resolution, True, grid)
    want = sorted(map(tuple, single.int_corners().tolist()))
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    box = shape.bounding_box().expanded_additive(resolution / 2)
    levels = cc.subdivision.calculate_block_sizes(box, 3, resolution, grid, True)
    tape = cc.nodes.make_program_buffer(shape)
    counts = list(single.level_counts)
    pipe = dist.subdivision_pipeline(tape, levels, resolution, tuple(box.a), 3, [c + 5 for c in counts], dev, stream)
    for _ in range(2):     # a pipeline is reusable: headers are reset on every traversal
        mine = pipe.enqueue()
        assert pipe.check() == counts
        n = int(mine[0, 0])
        assert sorted(map(tuple, mine[1:1 + n, :3].cpu().tolist())) == want
    # the consumer, launched for the capacity with the length read on the device
    dims = tuple(int(d) for d in levels[-1][1])
    cells = dims[0] * dims[1] * dims[2]
    cap = int(mine.shape[0]) - 1
    out = torch.full((cap, cells), float("nan"), dtype=torch.float32, device=dev)
    ref = torch.empty((n, cells), dtype=torch.float32, device=dev)
    o = (ctypes.c_double * 3)(*tuple(box.a))
    d = (ctypes.c_uint32 * 3)(*dims)
    step = np.float32(levels[-1][0] * resolution)
    check(hip.lib.hu_grid_eval_blocks_indirect(tape.device_ptr, mine[1:].data_ptr(), mine.data_ptr(), cap, resolution, o, step, d, 1,
                                               out.data_ptr(), stream), "indirect")
    check(hip.lib.hu_grid_eval_blocks(tape.device_ptr, mine[1:].data_ptr(), n, resolution, o, step, d, 1, ref.data_ptr(), stream), "direct")
    torch.cuda.synchronize()
    assert same_bits(out[:n].cpu().numpy(), ref.cpu().numpy())
    assert bool(torch.isnan(out[n:]).all())      # nothing past the list's length was touched
    small = dist.subdivision_pipeline(tape, levels, resolution, tuple(box.a), 3, [max(c // 2, 1) for c in counts], dev, stream)
    small.enqueue()
    with pytest.raises(dist.Overflow) as info:
        small.check()
    assert info.value.needed[0] == counts[0]


def test_lists_longer_than_one_grid_are_launched_in_pieces(hip):
    """A launch may have at most 2^32 - 1 work-items: 17 M blocks of 2^3 samples (one workgroup each) exceed that,
    as does a capacity-sized launch over a short list.  Blocks beyond the first piece equal the oracle."""
    import torch
    from codecad_amd import hip_util
    from codecad_amd.hip_util import check
    ref = GOLDEN["sphere_plus_box"]
    tape = hip_util.Tape(ref["tape"], policy="0")
    n_blocks = (1 << 24) + 12345
    rng = np.random.default_rng(21)
    rows = torch.from_numpy(rng.integers(-60, 60, (n_blocks, 4)).astype(np.int32)).cuda()
    count = torch.tensor([n_blocks], dtype=torch.int32, device="cuda")
    out = torch.zeros((n_blocks, 8), dtype=torch.float32, device="cuda")
    dims = (ctypes.c_uint32 * 3)(2, 2, 2)
    origin = (ctypes.c_double * 3)(0.25, -0.5, 0.125)
    step = np.float32(0.75)
    for specialise in (False, True):
        if specialise:
            tape.specialize()
        out.zero_()
        check(hip.lib.hu_grid_eval_blocks_indirect(tape.device_ptr, rows.data_ptr(), count.data_ptr(), n_blocks + 1000, 1.0, origin, step,
                                                   dims, 1, out.data_ptr(), None), "indirect")
        torch.cuda.synchronize()
        for b in (0, 5, (1 << 24) - 1, 1 << 24, n_blocks - 1):
            corner = rows[b, :3].cpu().numpy().astype(np.float64) * 1.0 + np.array([0.25, -0.5, 0.125])
            want = oracle.grid_eval_pymcubes(ref["tape"], corner.astype(np.float32), step, (2, 2, 2))
            assert same_bits(out[b].cpu().numpy(), np.asarray(want).reshape(-1)), (specialise, b)


@pytest.mark.parametrize("forced", ["0", "1"])
def test_library_forms_over_a_process_group(hip, forced):
    """dist.mass_properties and dist.subdivision (tools/rehearse_dist.py compares them with the single-GPU drivers) --
    plainly, and with CODECAD_AMD_FORCE_COLLECTIVES=1: one rank in a real RCCL group, every level through
    all_gather_into_tensor -> hu_slice_rows -> indirect launches, the integrals through an all-reduce."""
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, CODECAD_AMD_FORCE_COLLECTIVES=forced)
    for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_RANK"):
        env.pop(k, None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rehearse_dist.py")], capture_output=True, text=True,
                          timeout=600, cwd=ROOT, env=env)
    assert proc.returncode == 0 and "rehearsal ok" in proc.stdout, proc.stdout[-2000:] + proc.stderr[-3000:]


def test_replicated_levels_partition_by_ownership(hip):
    """What replicated levels of a multi-rank traversal rest on (dist.LevelPipeline `replicate`, hu_*_level_owned): when
    `world` ranks classify the same parents, each listing the cells it owns, the lists partition the level -- whatever the
    order of a rank's parents and of its atomics, whichever evaluator or launch shape serves it -- and a cell's owner is
    dist.owner_of(parent row, linear cell index); the owned moment sums of mass_properties add up to the level's."""
    import torch
    import codecad_amd as cc
    from codecad_amd import subdivision, dist, hip_util
    from codecad_amd.hip_util import check
    shape = cc.examples.sponge(4)
    res = 1 / 512
    box = shape.bounding_box().expanded_additive(res / 2)
    levels = subdivision.calculate_block_sizes(box, 3, res, 16, True)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    origin = (ctypes.c_double * 3)(box.a.x, box.a.y, box.a.z)
    tape = hip_util.Tape(cc.nodes.make_program(shape), policy="0")

    def level(parents, lvl, own=None, capacity=40000):
        int_step, dims = levels[lvl]
        d = (ctypes.c_uint32 * 3)(int(dims[0]), int(dims[1]), int(dims[2]))
        n = torch.tensor([parents.shape[0]], dtype=torch.int32, device=dev)
        out = torch.zeros((capacity + 1, 4), dtype=torch.int32, device=dev)
        step = np.float32(int_step * res)
        args = (tape.device_ptr, parents.data_ptr(), n.data_ptr(), int(parents.shape[0]), int(int_step), d, 3, float(res), origin, step,
                np.float32(float(step) * 3 ** 0.5 / 2), out.data_ptr(), out[1:].data_ptr(), capacity)
        if own is None:
            check(hip.lib.hu_subdivision_level_indirect(*args, stream), "level")
        else:
            check(hip.lib.hu_subdivision_level_owned(*args, own[0], own[1], stream), "level owned")
        torch.cuda.synchronize()
        return out[1:1 + int(out[0, 0])].clone()

    top = torch.zeros((1, 4), dtype=torch.int32, device=dev)
    for evaluator in ("interpreter", "specialised"):
        if evaluator == "specialised":
            tape.specialize(hip_util.SPEC_CLASSIFY)
        l0 = level(top, 0)
        assert l0.shape[0] == 27
        whole = level(l0, 1)
        want = sorted(map(tuple, whole.cpu().tolist()))
        assert len(want) == 30800
        int_step, dims = levels[1]
        for world in (2, 3, 8):
            shares = []
            for rank in range(world):
                parents = l0[torch.randperm(27, device=dev)]          # every rank has its own order of the same parents
                shares.append(level(parents, 1, own=(world, rank)).cpu().numpy())
            assert sorted(map(tuple, np.concatenate(shares).tolist())) == want, (evaluator, world)
            sizes = [len(sh) for sh in shares]
            assert max(sizes) - min(sizes) <= 0.12 * len(want) / world + 8, sizes      # a cell's owner is as good as random
            # a cell's owner is the rule of dist.owner_of on its parent's row and its linear index in the parent's grid
            for rank in (0, world - 1):
                for row in shares[rank][:: max(1, len(shares[rank]) // 40)]:
                    cell = [int(row[c]) for c in range(3)]
                    # the parent is the level-0 survivor whose block holds the cell
                    for p in l0.cpu().tolist():
                        rel = [(cell[c] - p[c]) // int(int_step) for c in range(3)]
                        if all(0 <= rel[c] < int(dims[c]) and (cell[c] - p[c]) % int(int_step) == 0 for c in range(3)):
                            assert dist.owner_of(p, rel[2] + int(dims[2]) * (rel[1] + int(dims[1]) * rel[0]), world) == rank
                            break
                    else:
                        raise AssertionError("a listed cell belongs to no parent")
    # mass properties: the owned sums and lists of `world` ranks add up to the level's
    mlevels = [(res * cell, tuple(int(v) for v in dims)) for cell, dims in subdivision.calculate_block_sizes(shape.bounding_box(), 3, res, 8, overlap=False)]
    s, dims = mlevels[0]
    d = (ctypes.c_uint32 * 3)(*dims)
    parent = torch.zeros((1, 4), dtype=torch.float64, device=dev)
    parent[0, :3] = torch.tensor([shape.bounding_box().a.x, shape.bounding_box().a.y, shape.bounding_box().a.z], dtype=torch.float64)
    one = torch.tensor([1], dtype=torch.int32, device=dev)

    def mass(own=None):
        sums = torch.zeros((1, 10), dtype=torch.int32, device=dev)
        out = torch.zeros((600, 4), dtype=torch.float64, device=dev)
        args = (tape.device_ptr, parent.data_ptr(), one.data_ptr(), 1, float(s), d, np.float32(s), np.float32(s * 3 ** 0.5 / 2), sums.data_ptr(),
                out.data_ptr(), out[1:].data_ptr(), 599)
        if own is None:
            check(hip.lib.hu_mass_properties_level_indirect(*args, stream), "mass level")
        else:
            check(hip.lib.hu_mass_properties_level_owned(*args, own[0], own[1], stream), "mass level owned")
        torch.cuda.synchronize()
        n = int(out.view(torch.int32)[0, 0])
        return sums.cpu().numpy().astype(np.int64)[0], sorted(map(tuple, out[1:1 + n].cpu().tolist()))
    all_sums, all_cells = mass()
    for world in (2, 5):
        parts = [mass((world, r)) for r in range(world)]
        assert (sum(p[0] for p in parts) == all_sums).all()
        assert sorted(c for p in parts for c in p[1]) == all_cells and sum(len(p[1]) for p in parts) == len(all_cells)


def test_slice_rows_of_shares_one_piece_out(hip):
    """hu_slice_rows_of: ONE piece [header | rows] shared out among `world` ranks by dist.balanced_slice -- the shares tile it
    in order."""
    import torch
    from codecad_amd import dist
    from codecad_amd.hip_util import check
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    rng = np.random.default_rng(5)
    lists = [rng.integers(-1000, 1000, (27, 4)).astype(np.int32)]
    # the share rule on one replicated piece: the shares of 1..8 ranks tile it in order
    piece = torch.zeros((33, 4), dtype=torch.int32, device=dev)
    piece[0, 0] = 27
    piece[1:28] = torch.from_numpy(lists[0]).to(dev)
    for world in (1, 2, 3, 8):
        got = []
        for rank in range(world):
            out = torch.full((30, 4), -1, dtype=torch.int32, device=dev)
            stats = torch.zeros(2, dtype=torch.int32, device=dev)
            check(hip.lib.hu_slice_rows_of(piece.data_ptr(), 33, 16, rank, world, out.data_ptr(), 29, stats.data_ptr(), stream), "slice")
            n = int(out[0, 0])
            b, e = dist.balanced_slice(27, rank, world)
            assert n == e - b and stats.cpu().tolist() == [27, 0]
            got.append(out[1:1 + n].cpu().numpy())
        assert np.array_equal(np.concatenate(got), lists[0])
