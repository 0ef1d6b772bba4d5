"""The BASELINE.json configs at their STATED sizes on the HIP path ("assert the configs at full size").

  C1  csg_example / sphere+box, 64^3 dense grid_eval: every voxel against the oracle, both layouts, both evaluators
  C2  sponge(3), 256^3 dense float4: x-slabs tile the grid, 200 k sampled voxels against the oracle, volume fraction
  C3  sponge(4), 512^3: dense -> tests/test_gpu_drivers.py::test_full_size_512_dense_properties; subdivision at 1/512
      -> SUBDIV_CASES there; mass_properties(1/512, grid 8), 85.7 M samples: the known answer below
  C4  planetary assembly, mass_properties(resolution 0.25, grid 64), 57.7 M samples of a 467-instruction tape
  C5  sponge(5) at 1/2048 -> tests/test_gpu_drivers.py::test_config_c5_sponge5_at_2048_single_gpu_form

The known answers of C3 / C4 (tests/golden/config_fixtures.json) come from the reference's per-block traversal
restated in tests/ref_driver.py over the CPU oracle's kernels, run once in the build container by
tests/golden/gen/make_config_fixtures.py."""
import ctypes
import json
import os

import numpy as np
import pytest

import oracle
from codecad_amd import util
from conftest import load_golden_tapes, same_bits, ROOT

pytestmark = pytest.mark.gpu
GOLDEN = load_golden_tapes()
with open(os.path.join(ROOT, "tests", "golden", "config_fixtures.json")) as _f:
    FIXTURES = json.load(_f)


@pytest.mark.parametrize("name", ["csg_example", "sphere_plus_box"])
@pytest.mark.parametrize("specialise", [False, True])
def test_c1_dense_64_every_voxel(hip, name, specialise):
    import codecad_amd as cc
    from codecad_amd import hip_util, grid_eval
    shape = {"csg_example": cc.examples.csg_example, "sphere_plus_box": cc.examples.sphere_plus_box}[name]()
    tape = cc.nodes.make_program(shape)
    n = 64
    bb = shape.bounding_box()
    extent = max(bb.b.x - bb.a.x, bb.b.y - bb.a.y, bb.b.z - bb.a.z)
    step = np.float32(extent / n)                     # cell-centred samples over the bbox
    corner = np.array([bb.a.x + extent / n / 2, bb.a.y + extent / n / 2, bb.a.z + extent / n / 2], np.float32)
    t = hip_util.Tape(tape, policy="0")
    if specialise:
        t.specialize()
    c4 = np.zeros(4, np.float32)
    c4[:3] = corner
    out = hip_util.Buffer(grid_eval.FLOAT4, (n, n, n))
    hip.k.grid_eval((n, n, n), None, t, c4, step, out).wait()
    assert same_bits(out.read().view(np.float32).reshape(n, n, n, 4), oracle.grid_eval(tape, corner, step, (n, n, n), threads=8))
    w = hip_util.Buffer(np.float32, (n, n, n))
    hip.k.grid_eval_pymcubes((n, n, n), None, t, c4, step, w).wait()
    assert same_bits(w.read().reshape(-1), oracle.grid_eval_pymcubes(tape, corner, step, (n, n, n), threads=8).reshape(-1))
    out.release()
    w.release()


@pytest.mark.parametrize("specialise", [False, True])
def test_c2_sponge3_dense_256(hip, specialise):
    import torch
    import codecad_amd as cc
    from codecad_amd import hip_util
    from codecad_amd.hip_util import check
    n = 256
    shape = cc.examples.sponge(3)
    host_tape = cc.nodes.make_program(shape)
    t = hip_util.Tape(host_tape, policy="0")
    if specialise:
        t.specialize()
    step = np.float32(1.0 / n)
    corner = np.array([-0.5 + 0.5 / n] * 3 + [0.0], np.float32)
    dims = (ctypes.c_uint32 * 3)(n, n, n)
    fptr = ctypes.POINTER(ctypes.c_float)

    def slab(x0, count):
        out = torch.empty((count, n, n, 4), dtype=torch.float32, device="cuda")
        check(hip.lib.hu_grid_eval_slab(t.device_ptr, corner.ctypes.data_as(fptr), step, dims, x0, count, 0, out.data_ptr(), None), "slab")
        torch.cuda.synchronize()
        return out

    whole = slab(0, n)
    for rank in range(8):                              # the eight x-slabs of an 8-GPU job tile the grid exactly
        x0 = rank * 32
        assert torch.equal(slab(x0, 32), whole[x0:x0 + 32])
    rng = np.random.default_rng(9)
    idx = rng.integers(0, n, size=(200000, 3))
    pts = corner[:3][None, :] + step * idx.astype(np.float32)
    want = oracle.evaluate_points(host_tape, pts)
    ti = torch.from_numpy(idx).cuda()
    assert same_bits(whole[ti[:, 0], ti[:, 1], ti[:, 2]].cpu().numpy(), want)
    inside = float((whole[..., 3] <= 0).double().mean().item())
    assert inside == pytest.approx((20 / 27) ** 3, rel=5e-2)      # 1/256 cells against 1/27 features: a discretisation, not a parity, check


def _check_mass(mp, stats, want):
    assert stats["function_evaluations"] == want["function_evaluations"]
    assert stats["kernel_invocations"] == len(want["levels"])
    assert mp.volume == pytest.approx(want["volume"], rel=1e-12)
    scale = max(abs(v) for row in want["inertia_tensor"] for v in row)
    assert np.allclose([mp.centroid.x, mp.centroid.y, mp.centroid.z], want["centroid"], rtol=0, atol=1e-9 * (1 + abs(want["centroid"][2])))
    assert np.allclose(np.asarray(mp.inertia_tensor, dtype=np.float64), want["inertia_tensor"], rtol=0, atol=1e-11 * scale)


@pytest.mark.parametrize("specialise", [False, True])
def test_c3_sponge4_mass_properties_full_size(hip, specialise, monkeypatch):
    import codecad_amd as cc
    monkeypatch.setenv("CODECAD_AMD_SPECIALIZE", "1" if specialise else "0")
    want = FIXTURES["c3_sponge4_mass_properties"]
    mp = cc.mass_properties(cc.examples.sponge(4), want["resolution"], want["grid_size"])
    _check_mass(mp, cc.mass_properties.last_stats, want)
    assert mp.volume == pytest.approx((20 / 27) ** 4, rel=1e-3)      # 1/512 is not aligned with the sponge's 1/81 faces


@pytest.mark.parametrize("specialise", [False, True])
def test_c4_planetary_mass_properties_full_size(hip, specialise, monkeypatch):
    """57.7 M samples of the 467-instruction planetary tape: evaluation count (1 top block of 7x7x5 cells, 220
    ambiguous cells refined to 64^3 each), volume, centroid and inertia equal the reference traversal's."""
    import codecad_amd as cc
    from codecad_amd.shapes import TapeShape
    monkeypatch.setenv("CODECAD_AMD_SPECIALIZE", "1" if specialise else "0")
    want = FIXTURES["c4_planetary_mass_properties"]
    g = GOLDEN["planetary"]
    shape = TapeShape(g["tape"], util.BoundingBox(util.Vector(*g["bbox_a"]), util.Vector(*g["bbox_b"])), float(g["feature_size"]))
    mp = cc.mass_properties(shape, want["resolution"], want["grid_size"])
    assert want["parents_per_level"] == [1, 220] and want["function_evaluations"] == 7 * 7 * 5 + 220 * 64 ** 3
    _check_mass(mp, cc.mass_properties.last_stats, want)
