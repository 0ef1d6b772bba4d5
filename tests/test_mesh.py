"""Leaf-block consumer: marching cubes -> triangular mesh -> STL (reference rendering/mesh.py, stl_renderer.py;
SURVEY.md section 8(f) rank 2).

The algorithm is PyMCubes 0.0.6's, a dependency that is not in the reference tree, so the oracle restates the
published algorithm and is pinned by the reference's own mesh test (tests/test_mesh.py:12-29: box(10) and
sphere(10) at subdivision grid sizes 2, 12, 16 must give a watertight mesh) plus geometric properties; the
vertex / triangle ORDER of PyMCubes is unpinned.  On the GPU the HIP kernels are compared with the oracle
exactly (vertex bits, triangle ids)."""
import math
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

import oracle
import ref_driver
import shapes_zoo
from codecad_amd import nodes, shapes
from conftest import ROOT


def edge_report(triangles, n_vertices):
    """(every directed edge once, every directed edge has its reverse) -> closed, consistently oriented."""
    t = np.asarray(triangles, dtype=np.int64)
    e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]])
    key, rkey = e[:, 0] * n_vertices + e[:, 1], e[:, 1] * n_vertices + e[:, 0]
    return len(np.unique(key)) == len(key), np.array_equal(np.sort(key), np.sort(rkey))


def signed_volume(vertices, triangles):
    p = np.asarray(vertices)[np.asarray(triangles, dtype=np.int64)]
    return float(np.einsum("ij,ij->i", p[:, 0], np.cross(p[:, 1], p[:, 2])).sum()) / 6


def weld(vertices, triangles, tolerance=1e-6):
    """Merge vertices that coincide (what trimesh's mesh.process() does in the reference's test); drop
    triangles that collapse."""
    q = np.round(np.asarray(vertices) / tolerance).astype(np.int64)
    _, first, inverse = np.unique(q, axis=0, return_index=True, return_inverse=True)
    t = inverse.reshape(-1)[np.asarray(triangles, dtype=np.int64)]
    keep = (t[:, 0] != t[:, 1]) & (t[:, 1] != t[:, 2]) & (t[:, 2] != t[:, 0])
    return np.asarray(vertices)[first], t[keep]


def sphere_field(n, centre, radius):
    g = np.mgrid[0:n, 0:n, 0:n].astype(np.float64)
    return (np.sqrt(sum((g[i] - centre[i]) ** 2 for i in range(3))) - radius).astype(np.float32)


def oracle_mesh(shape, grid_size):
    """The reference's per-block loop (mesh.py:45-74) on the oracle -> list of (int_corner, vertices, triangles)."""
    tape = nodes.make_program(shape)
    resolution = shape.feature_size() / 2
    dims, boxes = ref_driver.subdivision(tape, shape.bounding_box(), 3, resolution, overlap=True, grid_size=grid_size or 128)
    sx, sy, sz = (int(d) for d in dims)
    out = []
    for corner, step, int_corner, _int_step in sorted(boxes, key=lambda b: b[2]):
        block = oracle.grid_eval_pymcubes(tape, ref_driver.f32_corner(corner), np.float32(step), (sx, sy, sz))
        v, t = oracle.marching_cubes(np.asarray(block).reshape(sy, sx, sz))
        # mesh.py:65-68 (float64 numpy); the reference also swaps the winding, PyMCubes' being the other way
        v[:, [0, 1]] = v[:, [1, 0]]
        v[:, 1] *= -1
        v *= step
        v += np.array([corner.x, corner.y, corner.z])
        out.append((int_corner, v, t))
    return out, ((sy - 1) * boxes[0][1] if boxes else 0.0)   # a shape without surface has no leaf blocks


def concatenate(blocks):
    vs, ts, base = [], [], 0
    for _c, v, t in blocks:
        vs.append(v)
        ts.append(t.astype(np.int64) + base)
        base += len(v)
    return np.concatenate(vs), np.concatenate(ts)


# ---------------------------------------------------------------------------------------------------
def test_case_tables_are_what_the_generator_derives():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_mc_table.py"), "--check"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr   # also runs the generator's own crack/orientation checks


@pytest.mark.parametrize("radius", [3.3, 8.2, 10.9])
def test_oracle_sphere_is_closed_oriented_and_has_the_volume(radius):
    f = sphere_field(24, (11.3, 11.7, 12.1), radius)
    v, t = oracle.marching_cubes(f)
    unique, paired = edge_report(t, len(v))
    assert unique and paired
    assert len(v) - 3 * len(t) // 2 + len(t) == 2                       # Euler characteristic of a sphere
    # a chordal approximation: the deficit shrinks with the square of the radius in cells
    assert signed_volume(v, t) == pytest.approx(4 / 3 * math.pi * radius ** 3, rel=0.7 / radius ** 2)
    g = np.linalg.norm(v - np.array([11.3, 11.7, 12.1]), axis=1)
    assert np.all(np.abs(g - radius) < 0.08)                           # vertices on the surface
    frac = v - np.floor(v)
    assert np.all(np.count_nonzero(frac > 0, axis=1) <= 1)              # each on a grid edge


def test_oracle_mesh_of_noise_has_no_cracks():
    """Random sign patterns exercise every case of the table, ambiguous faces included."""
    rng = np.random.default_rng(7)
    cases = set()
    for _ in range(6):
        f = rng.uniform(-1, 1, (11, 9, 10)).astype(np.float32)
        f[0], f[-1], f[:, 0], f[:, -1], f[:, :, 0], f[:, :, -1] = 1, 1, 1, 1, 1, 1
        v, t = oracle.marching_cubes(f)
        unique, paired = edge_report(t, len(v))
        assert unique and paired
        assert signed_volume(v, t) > 0
        inside = f <= 0
        c = sum(inside[dx:f.shape[0] - 1 + dx, dy:f.shape[1] - 1 + dy, dz:f.shape[2] - 1 + dz].astype(int) << m
                for m, (dx, dy, dz) in enumerate([(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]))
        cases |= set(np.unique(c).tolist())
    assert len(cases) == 256


def test_oracle_empty_and_degenerate_blocks():
    for f in (np.ones((4, 4, 4), np.float32), -np.ones((4, 4, 4), np.float32), np.ones((1, 5, 5), np.float32),
              np.zeros((3, 3, 3), np.float32)):
        v, t = oracle.marching_cubes(f)
        assert len(v) == 0 and len(t) == 0
    f = np.ones((3, 3, 3), np.float32)
    f[1, 1, 1] = 0.0   # a sample exactly on the surface counts as inside: an octahedron collapsed onto it
    v, t = oracle.marching_cubes(f)
    assert len(v) == 6 and len(t) == 8 and np.allclose(v, 1.0)


@pytest.mark.parametrize("grid_size", [2, 12, 16])
@pytest.mark.parametrize("name", ["box", "sphere"])
def test_oracle_watertight_like_the_reference_test(name, grid_size):
    """reference tests/test_mesh.py:12-29 on the oracle-driven pipeline."""
    shape = {"box": shapes.box(10), "sphere": shapes.sphere(10)}[name]
    blocks, y_shift = oracle_mesh(shape, grid_size)
    v, t = weld(*concatenate(blocks))
    unique, paired = edge_report(t, len(v))
    assert len(t) > 0 and unique and paired
    # the sample spacing is feature_size / 2 = 5 for both shapes, whatever the block size: a coarse
    # mesh (the cube's corners are cut), so only a loose volume bound
    expected = {"box": 1000.0, "sphere": 4 / 3 * math.pi * 125}[name]
    assert 0.5 * expected < signed_volume(v, t) < 1.05 * expected
    # the placement quirk of mesh.py:65-68: the mesh sits (sy-1)*step below the shape in y
    centre = (v.min(axis=0) + v.max(axis=0)) / 2
    assert centre[0] == pytest.approx(0, abs=0.3) and centre[2] == pytest.approx(0, abs=0.3)
    assert centre[1] == pytest.approx(-y_shift, abs=0.3)


# ---------------------------------------------------------------------------------------------------
def hip_marching_cubes(hip, fields, true_positions=False):
    """Run the device marching cubes on explicit fields (n, A0, A1, A2) with identity placement (corner 0, step 1)."""
    import ctypes
    from codecad_amd import hip_util
    f = np.ascontiguousarray(fields, dtype=np.float32)
    n, a0, a1, a2 = f.shape
    lib = hip.lib
    fields_dev = hip_util.Buffer(np.float32, f.shape)
    fields_dev.enqueue_write(f).wait()
    blocks = hip_util.Buffer(np.int32, (n, 4))
    blocks.enqueue_write(np.zeros((n, 4), np.int32)).wait()
    dims = (ctypes.c_uint32 * 3)(a0, a1, a2)
    n_wg, entries, segments = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(0)
    assert lib.hu_mesh_workgroups(n, dims, ctypes.byref(n_wg), ctypes.byref(entries), ctypes.byref(segments)) == 0
    counts = hip_util.Buffer(np.uint32, (entries.value, 2))
    masks = hip_util.Buffer(np.uint32, (segments.value,))
    assert lib.hu_mesh_count(fields_dev.device_ptr, n, dims, masks.device_ptr, counts.device_ptr, hip.queue.handle) == 0, lib.hu_last_error()
    prefix = counts.read().copy()
    tv, tt = int(prefix[n_wg.value, 0]), int(prefix[n_wg.value, 1])
    info = hip_util.Buffer(np.uint32, (segments.value, 4))
    vertices = hip_util.Buffer(np.float64, (max(tv, 1), 3))
    triangles = hip_util.Buffer(np.uint32, (max(tt, 1), 3))
    o = (ctypes.c_double * 3)(0.0, 0.0, 0.0)
    assert lib.hu_mesh_emit(fields_dev.device_ptr, blocks.device_ptr, n, 1.0, o, 1.0, dims, 0.0, masks.device_ptr, counts.device_ptr,
                            info.device_ptr, vertices.device_ptr, triangles.device_ptr, hip.queue.handle) == 0, lib.hu_last_error()
    v, t = vertices.read()[:tv].copy(), triangles.read()[:tt].copy()
    chunks = n_wg.value // n
    starts = np.concatenate([prefix[0:n_wg.value:chunks], prefix[n_wg.value:n_wg.value + 1]]).astype(np.int64)
    for b in (fields_dev, blocks, counts, masks, info, vertices, triangles):
        b.release()
    return v, t, starts


def oracle_placed(field):
    """Oracle mesh of one block under the identity placement the helper above asks for."""
    v, t = oracle.marching_cubes(field)
    return np.stack([v[:, 1] * 1.0 + 0.0, (-v[:, 0]) * 1.0 + 0.0 + 0.0, v[:, 2] * 1.0 + 0.0], axis=1), t


@pytest.mark.gpu
def test_hip_marching_cubes_matches_oracle_on_noise_and_spheres(hip):
    rng = np.random.default_rng(3)
    fields = rng.uniform(-1, 1, (5, 9, 13, 7)).astype(np.float32)
    fields[1] = sphere_field(13, (6.1, 5.9, 6.3), 4.4)[:9, :13, :7]
    fields[2] = 1.0                      # an empty block between non-empty ones
    fields[3, 2:4, 3:5, 1:3] = 0.0       # samples exactly on the surface
    v, t, starts = hip_marching_cubes(hip, fields)
    for b in range(5):
        want_v, want_t = oracle_placed(fields[b])
        got_v = v[starts[b, 0]:starts[b + 1, 0]]
        got_t = t[starts[b, 1]:starts[b + 1, 1]].astype(np.int64) - starts[b, 0]
        assert got_v.shape == want_v.shape and got_t.shape == want_t.shape
        assert np.array_equal(got_v.view(np.uint64), want_v.view(np.uint64))
        assert np.array_equal(got_t, want_t.astype(np.int64))
    assert starts[3, 0] == starts[2, 0] and starts[3, 1] == starts[2, 1]


@pytest.mark.gpu
def test_hip_marching_cubes_sizes_that_split_a_block_over_workgroups(hip):
    """257 and 1000+ samples per block: vertex ids and triangle slots cross workgroup boundaries."""
    rng = np.random.default_rng(5)
    # ... and rows longer than one 32-sample segment (33, 63, 70, 100 samples: segments share their end samples)
    # (16^3 noise: a whole block per workgroup, ~10^4 triangles: several windows of its triangle queue)
    for shape in ((2, 16, 16, 16), (2, 1, 257, 1), (3, 10, 10, 11), (2, 16, 16, 32), (2, 10, 10, 32), (1, 256, 1, 3), (1, 1, 256, 2), (1, 2, 2, 2), (2, 33, 5, 17), (2, 3, 4, 32), (2, 3, 4, 33),
                  (1, 5, 3, 63), (2, 4, 5, 70), (1, 20, 20, 100), (1, 1, 1, 40), (1, 2, 1, 40)):
        fields = rng.uniform(-1, 1, shape).astype(np.float32)
        v, t, starts = hip_marching_cubes(hip, fields)
        for b in range(shape[0]):
            want_v, want_t = oracle_placed(fields[b])
            assert np.array_equal(v[starts[b, 0]:starts[b + 1, 0]].view(np.uint64), want_v.view(np.uint64))
            assert np.array_equal(t[starts[b, 1]:starts[b + 1, 1]].astype(np.int64) - starts[b, 0], want_t.astype(np.int64))


@pytest.mark.gpu
def _tall_shape():
    """A thin tall box with a ball on it: with the default 128^3 blocks its rows (z is the fastest axis)
    are ~100 samples long, i.e. several 32-sample segments each."""
    return shapes.box(4, 4, 200) + shapes.sphere(6).translated_z(30)


@pytest.mark.gpu
@pytest.mark.parametrize("name,grid", [("box", 12), ("sphere", 16), ("csg_thing", 16), ("torus", 12), ("symmetrical_xyz", 16),
                                       ("csg_thing", None), ("tall", None), ("tall", 40)])
def test_hip_mesh_pipeline_matches_oracle_pipeline(hip, name, grid):
    from codecad_amd.rendering import mesh
    shape = _tall_shape() if name == "tall" else shapes_zoo.shapes_3d[name]
    got = list(mesh.triangular_mesh(shape, subdivision_grid_size=grid))
    want = [(c, v, t) for c, v, t in oracle_mesh(shape, grid)[0] if len(t)]
    assert len(got) == len(want) > 0
    for (gv, gt), (_c, wv, wt) in zip(got, want):
        assert np.array_equal(gv.view(np.uint64), wv.view(np.uint64))
        assert np.array_equal(gt, wt)


@pytest.mark.gpu
@pytest.mark.parametrize("grid_size", [2, 12, 16])
@pytest.mark.parametrize("name", ["box", "sphere"])
def test_hip_watertight_like_the_reference_test(hip, name, grid_size):
    from codecad_amd.rendering import mesh
    shape = {"box": shapes.box(10), "sphere": shapes.sphere(10)}[name]
    m = mesh.mesh_arrays(shape, subdivision_grid_size=grid_size, true_positions=True)
    v, t = weld(m.vertices, m.triangles)
    unique, paired = edge_report(t, len(v))
    assert len(t) > 0 and unique and paired
    expected = {"box": 1000.0, "sphere": 4 / 3 * math.pi * 125}[name]
    assert 0.5 * expected < signed_volume(v, t) < 1.05 * expected
    centre = (v.min(axis=0) + v.max(axis=0)) / 2
    assert np.allclose(centre, 0, atol=0.3)     # true_positions: no y translation


@pytest.mark.gpu
def test_stl_export(hip, tmp_path):
    from codecad_amd.rendering import stl_renderer, mesh
    path = tmp_path / "sphere.stl"
    n = stl_renderer.render_stl(shapes.sphere(10), str(path), subdivision_grid_size=16)
    data = path.read_bytes()
    assert n >= 20 and len(data) == 84 + 50 * n and struct.unpack("<I", data[80:84])[0] == n
    rec = np.frombuffer(data[84:], dtype=stl_renderer.RECORD)
    centre = rec["vectors"].reshape(-1, 3).mean(axis=0)
    out = rec["vectors"].mean(axis=1) - centre
    assert np.all(np.einsum("ij,ij->i", rec["normal"], out) > 0)   # normals point out of the sphere
    # the file is the oracle's records of the indexed mesh, byte for byte
    m = mesh.mesh_arrays(shapes.sphere(10), subdivision_grid_size=16)
    assert data[84:] == oracle.stl_records(m.vertices, m.triangles).tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", [32, 100, 129])
def test_stl_streams_in_pieces(hip, tmp_path, monkeypatch, chunk):
    """render_stl with pieces much smaller than the mesh (ragged last piece, pieces that are not multiples of
    a workgroup): the file does not depend on the piece size."""
    from codecad_amd.rendering import stl_renderer, mesh
    shape = shapes.sphere(20) - shapes.cylinder(h=30, d=8)
    m = mesh.mesh_arrays(shape, subdivision_grid_size=16)
    want = oracle.stl_records(m.vertices, m.triangles).tobytes()
    assert len(m.triangles) > 2 * chunk
    monkeypatch.setattr(mesh, "_STL_CHUNK", chunk)
    path = tmp_path / "piece.stl"
    n = stl_renderer.render_stl(shape, str(path), subdivision_grid_size=16)
    data = path.read_bytes()
    assert n == len(m.triangles) and struct.unpack("<I", data[80:84])[0] == n
    assert data[84:] == want


@pytest.mark.gpu
@pytest.mark.parametrize("n_triangles", [1, 7, 255, 256, 257, 1000, 70001])
def test_stl_records_parity(hip, n_triangles):
    """hu_mesh_stl == the oracle's records byte for byte: ragged last workgroup, shared and repeated
    vertices, values that round differently in float32, large / tiny / negative-zero coordinates (cross products
    stay finite: the sign of an inf - inf NaN is the one thing x86 and the GPU do not share)."""
    from codecad_amd.rendering import stl_renderer
    rng = np.random.default_rng(n_triangles)
    n_vertices = max(3, n_triangles // 2)
    v = rng.standard_normal((n_vertices, 3)) * 10.0 ** rng.integers(-3, 4, (n_vertices, 1))
    v[rng.integers(0, n_vertices, n_vertices // 8 + 1)] = np.array([-0.0, 1e-320, 1e15])
    v[rng.integers(0, n_vertices, n_vertices // 8 + 1)] *= 1e-20     # products that are float32 denormals
    v[0] = [1 + 2.0 ** -24, 1 + 2.0 ** -24 + 2.0 ** -50, -(1 + 3 * 2.0 ** -24)]   # ties and near-ties of the rounding
    t = rng.integers(0, n_vertices, (n_triangles, 3)).astype(np.uint32)
    got = stl_renderer.stl_records(v, t)
    want = oracle.stl_records(v, t)
    assert got.tobytes() == want.tobytes()


@pytest.mark.gpu
def test_stl_records_errors(hip):
    from codecad_amd.rendering import stl_renderer
    assert len(stl_renderer.stl_records(np.zeros((3, 3)), np.zeros((0, 3), np.uint32))) == 0
    with pytest.raises(ValueError):
        stl_renderer.stl_records(np.zeros((3, 3)), np.array([[0, 1, 3]], np.uint32))
    from codecad_amd import hip_util
    v = hip_util.Buffer(np.float64, (3, 3))
    t = hip_util.Buffer(np.uint32, (1, 3))
    rec = hip_util.Buffer(np.uint8, (2, 50))
    assert hip.lib.hu_mesh_stl(v.device_ptr, t.device_ptr, 1, rec.device_ptr + 2, None) != 0     # records must be 16-byte aligned
    assert b"aligned" in hip.lib.hu_last_error()
    assert hip.lib.hu_mesh_stl(None, t.device_ptr, 1, rec.device_ptr, None) != 0
    assert hip.lib.hu_mesh_stl(None, None, 0, None, None) == 0                                    # nothing to do
    for b in (v, t, rec):
        b.release()
