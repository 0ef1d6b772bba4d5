"""calculate_block_sizes against the reference's tables and its invariants
(reference tests/test_subdivision.py:44-107)."""
import pytest

from codecad_amd import util
from codecad_amd.subdivision import calculate_block_sizes


def test_against_reference_tables(golden_block_sizes):
    assert len(golden_block_sizes) >= 100
    for row in golden_block_sizes:
        box = util.BoundingBox(util.Vector(*row["a"]), util.Vector(*row["b"]))
        args = (box, row["dimension"], row["resolution"], row["grid_size"], row["overlap"], row["multiplier"])
        if row["result"] == "ValueError":
            with pytest.raises(ValueError):
                calculate_block_sizes(*args)
            continue
        got = [[c, [int(v) for v in d]] for c, d in calculate_block_sizes(*args)]
        assert got == row["result"], row


@pytest.mark.parametrize("box_size", [util.Vector(10, 20, 30), util.Vector(16, 16, 16)])
@pytest.mark.parametrize("dimension", [2, 3])
@pytest.mark.parametrize("resolution", [1, 0.1])
@pytest.mark.parametrize("grid_size, multiplier", [(2, 1), (2, 2), (21, 1), (256, 1), (256, 256)])
@pytest.mark.parametrize("overlap", [True, False])
def test_invariants(box_size, dimension, resolution, grid_size, overlap, multiplier):
    box = util.BoundingBox(-box_size / 2, box_size / 2)
    bs = calculate_block_sizes(box, dimension, resolution, grid_size, overlap, multiplier)
    assert bs[-1][0] == 1
    for i, (cell, dims) in enumerate(bs):
        if i > 0:
            assert all(dims[j] == grid_size for j in range(dimension))
        if dimension == 2:
            assert dims[2] == 1
        assert max(dims) > 1
        assert all(dims[j] % multiplier == 0 for j in range(dimension))
    real = bs[0][0] * resolution
    shared = 1 if (overlap and len(bs) == 1) else 0
    for i in range(dimension):
        assert bs[0][1][i] >= box_size[i] / real + shared
        assert multiplier > 1 or box_size[i] / real + shared > bs[0][1][i] - 1
    for level, ((cell, dims), (finer, finer_dims)) in enumerate(zip(bs[:-1], bs[1:])):
        for i in range(dimension):
            shared = 1 if (overlap and level == len(bs) - 2) else 0
            assert cell == finer * (finer_dims[i] - shared)


def test_config_hierarchies():
    """The BASELINE configs."""
    unit = util.BoundingBox(util.Vector.splat(-0.5), util.Vector.splat(0.5))
    def dims(bs):
        return [(c, tuple(int(v) for v in d)) for c, d in bs]
    assert dims(calculate_block_sizes(unit, 3, 1 / 512, 8, False)) == [(64, (8,) * 3), (8, (8,) * 3), (1, (8,) * 3)]
    e = unit.expanded_additive(0.5 / 512)
    assert dims(calculate_block_sizes(e, 3, 1 / 512, 16, True)) == [(240, (3,) * 3), (15, (16,) * 3), (1, (16,) * 3)]
    assert dims(calculate_block_sizes(e, 3, 1 / 512, 128, True)) == [(127, (5,) * 3), (1, (128,) * 3)]
    e = unit.expanded_additive(0.5 / 2048)
    assert dims(calculate_block_sizes(e, 3, 1 / 2048, 16, True)) == [(240, (9,) * 3), (15, (16,) * 3), (1, (16,) * 3)]
    assert dims(calculate_block_sizes(unit, 3, 1 / 2048, 16, False)) == [(256, (8,) * 3), (16, (16,) * 3), (1, (16,) * 3)]
    with pytest.raises(ValueError):
        calculate_block_sizes(unit, 3, 0.1, 10, True, 3)
