"""The kernels' fast sqrt / reciprocal are the IEEE results, proved by enumeration: every one of the
2^32 binary32 inputs is run through csrc/interp.hpp sqrt_cr / sqrt_inv_cr (one and two voxels per
lane) on the device and compared with the compiler's correctly rounded sqrt and division, which the
parity tests in turn compare with the CPU oracle's sqrtf and 1.0f / s."""
import ctypes

import pytest

pytestmark = pytest.mark.gpu


def test_fast_sqrt_and_reciprocal_are_correctly_rounded_for_all_inputs(hip):
    counts = (ctypes.c_uint64 * 4)()
    assert hip.lib.hu_selftest_math(counts) == 0, hip.lib.hu_last_error()
    sqrt_bad, root_bad, reciprocal_bad, fast_inputs = list(counts)
    assert (sqrt_bad, root_bad, reciprocal_bad) == (0, 0, 0)
    # 2^-100 .. 2^100 inclusive: 200 binades of 2^23 values + the upper end point
    assert fast_inputs == 200 * 2 ** 23 + 1


def test_three_operand_min_and_max_equal_the_two_instructions_they_replace(hip):
    """min3 / max3 (per-tape code: min(min(a, b), c) where nothing else reads the inner result) give the bits of the
    nested v_min_f32 / v_max_f32 -- which the oracle restates -- on every triple of special values and 2^26 random ones."""
    counts = (ctypes.c_uint64 * 3)()
    assert hip.lib.hu_selftest_minmax3(counts) == 0, hip.lib.hu_last_error()
    min_bad, max_bad, triples = list(counts)
    assert (min_bad, max_bad) == (0, 0)
    assert triples == 64 ** 3 + 2 ** 26
