"""The renderers against the reference's OWN output: its 32 baseline images
(tests/golden/baseline/, copied from reference tests/baseline/; compared like reference
tests/tools.py:64-79, mean squared error <= 1e-3 on a 0..1 scale at 1024x768).

On the CPU this runs the ORACLE's ray caster / bitmap on our tapes -- the one place where the
oracle (and through it evaluate(), every op, the shape library and the tape compiler) is checked
against pixels the reference itself produced.  On the GPU the HIP renderers are checked against the
oracle (exactly) and against the same baselines."""
import os

import numpy as np
import pytest
from PIL import Image

import oracle
import shapes_zoo
from codecad_amd import nodes
from codecad_amd.rendering import ray_caster, pictures
from conftest import ROOT

SIZE = (1024, 768)
BASELINE = os.path.join(ROOT, "tests", "golden", "baseline")
ALL = sorted(shapes_zoo.shapes_2d) + sorted(shapes_zoo.shapes_3d)
CPU_SUBSET = ALL


def baseline(name):
    return np.asarray(Image.open(os.path.join(BASELINE, "rendered_%s.png" % name)).convert("RGB"), dtype=np.float32) / 255


def mse(pixels, name):
    ref = baseline(name)
    assert pixels.shape == ref.shape
    err = pixels.astype(np.float32) / 255 - ref
    return float(np.mean(err * err))


def oracle_render(name, size=SIZE, threads=None, literal=False):
    """literal=True: the same renderers over the frozen literal-formula evaluate() (oracle/sdf_literal.c)."""
    shape = {**shapes_zoo.shapes_2d, **shapes_zoo.shapes_3d}[name]
    tape = nodes.make_program(shape)
    if literal:
        with oracle.literal_scene(tape):
            return oracle_render(name, size, threads)
    if shape.dimension() == 2:
        origin, step = pictures.bitmap_arguments(shape, size)
        out = oracle.bitmap(tape, list(origin), np.float32(step), size)
    else:
        cam = ray_caster.get_camera_params(shape.bounding_box(), size, None)
        a = ray_caster.kernel_arguments(shape, *cam)
        out = oracle.ray_caster(tape, list(a["origin"]), list(a["forward"]), list(a["up"]), list(a["right"]),
                                np.float32(a["pixel_tolerance"]), np.float32(a["box_radius"]),
                                np.float32(a["min_distance"]), np.float32(a["max_distance"]), np.float32(a["floor_z"]),
                                0, size, threads=threads or min(8, os.cpu_count() or 1))
    return out.transpose((1, 0, 2))


def test_all_baselines_present():
    assert sorted(f[len("rendered_"):-4] for f in os.listdir(BASELINE) if f.endswith(".png")) == sorted(ALL)


@pytest.mark.parametrize("name", CPU_SUBSET)
def test_oracle_render_matches_reference_baseline(name):
    assert mse(oracle_render(name), name) <= 1e-3


@pytest.mark.parametrize("name", CPU_SUBSET)
def test_literal_formulas_render_the_reference_baseline(name):
    """The chain closed from the other side: the reference's formulas taken literally (the frozen oracle/sdf_literal.c,
    which the canonical arithmetic is held to within 1e-5 by tests/test_literal_oracle.py) render the reference's own
    images within the reference's tolerance too -- so the literal restatement is pinned by a reference-held fixture
    itself, not only through the canonical oracle."""
    pixels = oracle_render(name, literal=True)
    assert mse(pixels, name) <= 1e-3
    # ... and the two arithmetics give the same picture within that tolerance as well (a bitmap pixel on the contour may flip)
    canonical = oracle_render(name)
    assert float(np.mean((pixels.astype(np.float32) - canonical.astype(np.float32)) ** 2)) / 255 ** 2 <= 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("name", ALL)
def test_hip_render_matches_oracle_and_baseline(hip, name):
    shape = {**shapes_zoo.shapes_2d, **shapes_zoo.shapes_3d}[name]
    pixels = pictures.render_pixels(shape, SIZE)
    assert pixels.shape == (SIZE[1], SIZE[0], 3) and pixels.dtype == np.uint8
    assert mse(pixels, name) <= 1e-3
    small = (160, 120)
    got = pictures.render_pixels(shape, small)
    want = oracle_render(name, small, threads=16)
    differing = np.count_nonzero(np.any(got != want, axis=-1))
    assert differing == 0, "%d of %d pixels differ from the oracle" % (differing, small[0] * small[1])


@pytest.mark.gpu
def test_render_options(hip):
    from codecad_amd import shapes
    s = shapes.sphere(2) + shapes.box(1.5).translated_x(1)
    cam = ray_caster.get_camera_params(s.bounding_box(), (96, 64), 40)
    plain = ray_caster.render(s, *cam, size=(96, 64))
    zebra = ray_caster.render(s, *cam, size=(96, 64), options=ray_caster.RenderOptions.zebra)
    false_color = ray_caster.render(s, *cam, size=(96, 64), options=ray_caster.RenderOptions.false_color)
    assert plain.shape == zebra.shape == false_color.shape == (64, 96, 3)
    assert np.any(plain != zebra) and np.all(false_color[..., 2] == 0) and false_color[..., 0].max() > 4
    a = ray_caster.kernel_arguments(s, *cam)
    tape = nodes.make_program(s)
    for opt, got in ((2, zebra), (1, false_color)):
        want = oracle.ray_caster(tape, list(a["origin"]), list(a["forward"]), list(a["up"]), list(a["right"]),
                                 np.float32(a["pixel_tolerance"]), np.float32(a["box_radius"]),
                                 np.float32(a["min_distance"]), np.float32(a["max_distance"]), np.float32(a["floor_z"]),
                                 opt, (96, 64), threads=8).transpose((1, 0, 2))
        assert np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["csg_thing", "torus", "mirror_3d", "gear", "nonconvex_shell2"])
def test_specialised_tapes_render_the_same_bytes(hip, name):
    """hu_tape_specialize also builds the ray caster / bitmap kernels for the tape: same pixels, faster."""
    from codecad_amd import hip_util
    from codecad_amd.hip_util import manager as m
    shape = {**shapes_zoo.shapes_2d, **shapes_zoo.shapes_3d}[name]
    size = (200, 150)
    interpreted = pictures.render_pixels(shape, size)
    spec = hip_util.Tape(nodes.make_program(shape)).specialize()
    out = hip_util.Buffer(np.uint8, size + (3,))
    if shape.dimension() == 2:
        origin, step = pictures.bitmap_arguments(shape, size)
        m.k.bitmap(size, None, spec, origin.as_float4(), np.float32(step), out).wait()
    else:
        cam = ray_caster.get_camera_params(shape.bounding_box(), size, None)
        a = ray_caster.kernel_arguments(shape, *cam)
        m.k.ray_caster(size, None, spec, a["origin"].as_float4(), a["forward"].as_float4(), a["up"].as_float4(),
                       a["right"].as_float4(), a["pixel_tolerance"], a["box_radius"], a["min_distance"], a["max_distance"],
                       a["floor_z"], 0, out).wait()
    assert np.array_equal(out.read().transpose((1, 0, 2)), interpreted)
    out.release()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["extruded_pentagon", "gear_wheel", "ring_of_balls", "twisted", "quarter_turns"])
def test_false_colour_agrees_where_rays_miss(hip, name):
    """False colour evaluates the shape at points at infinity (missed rays march to INFINITY, 0 * inf = NaN):
    garbage in, but the SAME garbage out of the kernels and the oracle -- in particular float -> int
    conversions of NaN are defined identically (sdf_math.hpp to_int_ / det_math.h dm_to_int)."""
    from codecad_amd import shapes
    shape = {
        "extruded_pentagon": lambda: shapes.regular_polygon2d(5, d=3).extruded(2).rotated_x(30),
        "gear_wheel": lambda: shapes_zoo.shapes_2d["gear"].extruded(1).rotated_x(60),
        "ring_of_balls": lambda: shapes.unsafe.CircularRepetition(shapes.sphere(1).translated_x(3), 7).rotated_x(50),
        "twisted": lambda: shapes.rectangle(1, 2).revolved(r=4, twist=90).rotated_x(40),
        "quarter_turns": lambda: (shapes.box(1, 2, 3).rotated_x(90).scaled(2) + shapes.cylinder(h=4, d=1).rotated_y(90)
                                  + shapes.sphere(1).translated_z(3).rotated_z(180)),   # no general rotation on top
    }[name]()
    size = (96, 72)
    cam = ray_caster.get_camera_params(shape.bounding_box(), size, 50)
    a = ray_caster.kernel_arguments(shape, *cam)
    tape = nodes.make_program(shape)
    for options in (1, 0):
        got = ray_caster.render(shape, *cam, size=size, options=ray_caster.RenderOptions(options))
        want = oracle.ray_caster(tape, list(a["origin"]), list(a["forward"]), list(a["up"]), list(a["right"]),
                                 np.float32(a["pixel_tolerance"]), np.float32(a["box_radius"]),
                                 np.float32(a["min_distance"]), np.float32(a["max_distance"]), np.float32(a["floor_z"]),
                                 options, size, threads=8).transpose((1, 0, 2))
        assert np.array_equal(got, want), "%d pixels differ" % np.count_nonzero(np.any(got != want, axis=-1))
