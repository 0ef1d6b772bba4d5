"""Shape equality as the reference's test suite defines it (reference tests/tools.py:63-108):
equal bounding boxes, a symmetric difference whose volume is at most 0.1 % of the box, and
low-resolution renders whose mean squared error is at most 1e-3.  All three run on the HIP path
(mass_properties driver + ray caster); the helper exists for tests only."""
import numpy
import pytest

import codecad_amd
from codecad_amd.rendering import pictures


def assert_pictures_equal(tested, expected):
    a = numpy.asarray(tested, dtype=numpy.float32) / 255
    b = numpy.asarray(expected, dtype=numpy.float32) / 255
    assert a.shape == b.shape
    error = a - b
    assert float(numpy.mean(error * error)) <= 1e-3, "Mean squared error is too big."


def assert_shapes_equal(shape, expected, resolution=0.1):
    box, expected_box = shape.bounding_box(), expected.bounding_box()
    common = box.intersection(expected_box)
    assert box.volume() == pytest.approx(common.volume())
    assert expected_box.volume() == pytest.approx(common.volume())

    volume = codecad_amd.mass_properties(shape ^ expected, resolution).volume
    assert volume <= box.volume() * 0.001

    assert_pictures_equal(pictures.render_pil_image(shape, size=(800, 400)),
                          pictures.render_pil_image(expected, size=(800, 400)))
