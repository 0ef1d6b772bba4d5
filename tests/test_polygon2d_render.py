"""2D contouring (reference rendering/polygon2d.cl + polygon2d.py), SURVEY.md section 8(f) rank 4.

The reference has no test or fixture for this renderer, so the oracle's restatement is checked through
properties of the contours it yields on the 2D shape zoo (CPU): every chain closes, every vertex lies on
the surface to within the sample spacing, the enclosed area matches the area measured by counting inside
samples, and contours are independent of how the plane is cut into blocks.  On the GPU the HIP kernel is
compared with the oracle cell by cell (exactly) and the batched pipeline with the oracle-driven one."""
import math

import numpy as np
import pytest

import oracle
import ref_driver
import shapes_zoo
from codecad_amd import nodes, util
from codecad_amd.rendering import polygon2d
from conftest import same_bits

SHAPES = sorted(shapes_zoo.shapes_2d)


def oracle_blocks(shape, grid_size):
    """The reference's per-block loop (polygon2d.py:84-126) on the oracle -> (blocks for stitch, int_box_step)."""
    tape = nodes.make_program(shape)
    resolution = shape.feature_size() / 2
    dims, boxes = ref_driver.subdivision(tape, shape.bounding_box(), 2, resolution, overlap=True, grid_size=grid_size)
    gx, gy = int(dims[0]), int(dims[1])
    assert int(dims[2]) == 1
    out = []
    for corner, step, int_corner, int_step in sorted(boxes, key=lambda b: b[2]):
        c = oracle.grid_eval(tape, ref_driver.f32_corner(corner), np.float32(step), (gx, gy, 1)).reshape(gx, gy, 4)
        v, l, s = oracle.process_polygon(c, ref_driver.f32_corner(corner)[:2], np.float32(step))
        out.append((int_corner[:2], v, l, np.sort(s)))
    int_box_step = boxes[0][3] * (gx - 1) if len(boxes) > 1 else None
    return out, int_box_step, resolution


def shoelace(poly):
    p = np.asarray(poly, dtype=np.float64)
    x, y = p[:, 0], p[:, 1]
    return 0.5 * float(np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y))


def sampled_area(shape, n=700):
    """Area by counting inside samples on an n x n grid over the bounding box (oracle evaluate)."""
    box = shape.bounding_box()
    size = box.size()
    step = max(size.x, size.y) / n
    nx, ny = int(math.ceil(size.x / step)), int(math.ceil(size.y / step))
    g = oracle.grid_eval(nodes.make_program(shape), np.array([box.a.x + step / 2, box.a.y + step / 2, 0], np.float32),
                         np.float32(step), (nx, ny, 1))
    return float(np.count_nonzero(g.reshape(-1, 4)[:, 3] <= 0)) * step * step


@pytest.mark.parametrize("name", SHAPES)
def test_oracle_contours_close_lie_on_the_surface_and_enclose_the_area(name):
    shape = shapes_zoo.shapes_2d[name]
    blocks, int_box_step, resolution = oracle_blocks(shape, 32)
    polygons = list(polygon2d.stitch(blocks, int_box_step))
    assert polygons, "no contour found"
    tape = nodes.make_program(shape)
    for poly in polygons:
        assert len(poly) >= 3
        pts = np.array([(x, y, 0.0) for x, y in poly], dtype=np.float32)
        d = oracle.evaluate_points(tape, pts)[:, 3]
        assert np.all(np.abs(d) <= 1.5 * resolution), float(np.abs(d).max())
        # consecutive vertices come from edge-adjacent half cells
        seg = np.linalg.norm(pts[:, :2] - np.roll(pts[:, :2], -1, axis=0), axis=1)
        assert float(seg.max()) <= 4.5 * resolution
    area = sum(shoelace(p) for p in polygons)   # holes wind the other way
    expected = sampled_area(shape)
    assert abs(abs(area) - expected) <= 0.03 * expected + 4 * resolution ** 2, (area, expected)


@pytest.mark.parametrize("name", ["circle", "gear", "nonconvex_shell2", "rotated_pattern_2d"])
def test_oracle_contours_do_not_depend_on_the_block_size(name):
    shape = shapes_zoo.shapes_2d[name]
    areas = []
    for grid in (16, 32, 128):
        blocks, int_box_step, _ = oracle_blocks(shape, grid)
        polys = list(polygon2d.stitch(blocks, int_box_step))
        areas.append((len(polys), sum(shoelace(p) for p in polys)))
    assert len({n for n, _ in areas}) == 1
    assert max(a for _, a in areas) - min(a for _, a in areas) <= 0.02 * abs(areas[0][1])


def test_encode_index_and_cell_tables_on_a_hand_made_grid():
    """A 3x3 corner grid with only the centre sample inside: the contour is one closed loop through the
    six half cells around the centre, no chain enters or leaves the block."""
    c = np.zeros((3, 3, 4), dtype=np.float32)
    c[..., 3] = 1.0
    c[1, 1, 3] = -1.0
    for x in range(3):
        for y in range(3):
            dx, dy = x - 1.0, y - 1.0
            n = math.hypot(dx, dy)
            c[x, y, 0], c[x, y, 1] = (dx / n, dy / n) if n else (1.0, 0.0)
            c[x, y, 3] = n - 0.5
    v, l, s = oracle.process_polygon(c, (0.0, 0.0), np.float32(1.0))
    assert len(s) == 0
    live = np.flatnonzero(l != 0xffffffff)
    assert len(live) == 6 and not np.any(l[live] & polygon2d.LINK_OVERFLOW_MASK)
    (poly,) = list(polygon2d.stitch([((0, 0), v, l, s)], None))
    assert len(poly) == 6
    r = np.hypot(*(np.array(poly).T - 1.0))
    assert np.all(r < 1.0)   # inside the ring of cells around the centre (the grid is far too coarse for more)
    assert shoelace(poly) > 0   # anticlockwise: inside on the left of the direction of travel


def test_stitch_detects_open_chains():
    v = np.zeros((2, 2), np.float32)
    l = np.array([0x80000000 | 1, 0xffffffff], dtype=np.uint32)
    with pytest.raises(AssertionError):
        list(polygon2d.stitch([((0, 0), v, l, np.array([], np.uint32))], 4))


# ---------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", SHAPES)
def test_hip_process_polygon_matches_oracle_cell_by_cell(hip, name):
    from codecad_amd import hip_util, grid_eval
    from codecad_amd.hip_util import manager as m
    shape = shapes_zoo.shapes_2d[name]
    tape = nodes.make_program(shape)
    box = shape.bounding_box().flattened()
    g = 67   # odd, not a multiple of the wavefront
    step = np.float32(max(box.size().x, box.size().y) * 1.1 / (g - 1))
    corner = util.Vector(box.midpoint().x - float(step) * (g - 1) / 2, box.midpoint().y - float(step) * (g - 1) / 2, 0)
    corners = grid_eval.grid_eval(shape, corner, step, (g, g, 1))
    cells = (g - 1) * (g - 1) * 2
    vertices = hip_util.Buffer(np.float32, (cells, 2))
    links = hip_util.Buffer(np.uint32, cells)
    starts = hip_util.Buffer(np.uint32, 2 * (g - 1))
    counter = hip_util.Buffer(np.uint32, 1)
    vertices.enqueue_fill(0xff)
    counter.enqueue_fill(0)
    ev = m.k.process_polygon((g - 1, g - 1, 2), None, corner.as_float2(), step, corners, vertices, links, starts, counter,
                             wait_for=[corners.event])
    got_v, got_l = vertices.read(wait_for=[ev]).copy(), links.read().copy()
    n = int(counter.read()[0])
    got_s = np.sort(starts.read()[:n].copy())
    c_host = oracle.grid_eval(tape, ref_driver.f32_corner(corner), step, (g, g, 1)).reshape(g, g, 4)
    assert same_bits(corners.read().view(np.float32).reshape(g, g, 4), c_host)
    want_v, want_l, want_s = oracle.process_polygon(c_host, ref_driver.f32_corner(corner)[:2], step)
    assert np.array_equal(got_l, want_l)
    live = want_l != 0xffffffff
    assert live.any()
    assert np.array_equal(got_v[live].view(np.uint32), want_v[live].view(np.uint32))
    assert np.all(got_v[~live].view(np.uint32) == 0xffffffff)   # empty cells are not written
    assert np.array_equal(got_s, np.sort(want_s))
    for b in (corners, vertices, links, starts, counter):
        b.release()


@pytest.mark.gpu
@pytest.mark.parametrize("name,grid", [(n, 32) for n in SHAPES] + [("gear", 16), ("gear", None), ("mirror_2d", 64)])
def test_hip_polygons_match_the_oracle_pipeline(hip, name, grid):
    shape = shapes_zoo.shapes_2d[name]
    got = list(polygon2d.polygon(shape, subdivision_grid_size=grid))
    blocks, int_box_step, _ = oracle_blocks(shape, grid or 128)
    want = list(polygon2d.stitch(blocks, int_box_step))
    assert got == want


@pytest.mark.gpu
def test_svg_export(hip, tmp_path):
    shape = shapes_zoo.shapes_2d["gear"]
    path = tmp_path / "gear.svg"
    polygon2d.render_svg(shape, str(path))
    text = path.read_text()
    assert text.startswith('<svg xmlns="http://www.w3.org/2000/svg"') and text.endswith('"/></svg>')
    assert text.count("M") == len(list(polygon2d.polygon(shape))) and text.count("L") > 100
