"""The reference's shape-equality check (tests/test_test_tools.py:9-28) on the HIP path: the
volume of `a ^ b` from mass_properties() plus a render comparison."""
import pytest

from codecad_amd.shapes import sphere, box, cylinder

import shapes_zoo
from shape_compare import assert_shapes_equal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(shapes_zoo.shapes_3d))
def test_identity_equal(name):
    shape = shapes_zoo.shapes_3d[name]
    assert_shapes_equal(shape, shape)


def test_not_equal():
    with pytest.raises(AssertionError):
        assert_shapes_equal(sphere(), box())


def test_not_equal_volume_only():
    """The difference is hidden inside the sphere: only the volume check can see it."""
    s1 = sphere(r=1)
    with pytest.raises(AssertionError):
        assert_shapes_equal(s1, s1 - box(1))


def test_equal_by_construction():
    """Two different CSG trees of the same solid."""
    a = box(2, 3, 5).translated(1, 0, 0).rotated_z(90)
    b = box(3, 2, 5).translated(0, 1, 0)
    assert_shapes_equal(a, b)
    c = cylinder(h=4, d=2) & box(10, 10, 2)
    assert_shapes_equal(c, cylinder(h=2, d=2))
