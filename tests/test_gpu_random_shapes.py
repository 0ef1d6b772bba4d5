"""Differential test on RANDOM CSG trees: every float of evaluate() from the HIP kernels (interpreter
and per-tape code) equals the oracle's, on grids that hit exact zeros, symmetry planes and repetition
boundaries.  Complements the 52 fixed tapes: the reduced transformation forms, the folded moves and the
short sqrt / reciprocal sequences see quarter turns, scalings, mirrors, nested repetitions and
extrusions in combinations nobody wrote down."""
import random

import numpy as np
import pytest

import oracle
from codecad_amd import nodes, shapes, grid_eval, hip_util
from conftest import same_bits

pytestmark = pytest.mark.gpu


from random_trees import random_2d, random_3d, grids  # noqa: E402


def check(shape, hip, specialise):
    tape = nodes.make_program(shape)
    handle = hip_util.Tape(tape)
    if specialise:
        handle.specialize()
    for corner, step, dims in grids():
        if shape.dimension() == 2:
            dims = (dims[0], dims[1], 1)
            corner = np.array([corner[0], corner[1], 0.0])
        want = oracle.grid_eval(tape, corner, step, dims)
        c4 = np.zeros(4, np.float32)
        c4[:3] = corner
        out = hip_util.Buffer(grid_eval.FLOAT4, dims)
        hip.k.grid_eval(dims, None, handle, c4, step, out).wait()
        got = out.read().view(np.float32).reshape(dims + (4,))
        assert same_bits(got, want), "float4 grid differs"
        want_w = oracle.grid_eval_pymcubes(tape, corner, step, dims)
        outw = hip_util.Buffer(np.float32, dims)
        hip.k.grid_eval_pymcubes(dims, None, handle, c4, step, outw).wait()
        assert same_bits(outw.read().reshape(-1), want_w), "distance grid differs"
        out.release()
        outw.release()
    handle.release()


import os

_EXTRA = int(os.environ.get("CODECAD_AMD_RANDOM_TREES", "0"))   # a one-off soak run: many more seeds


@pytest.mark.parametrize("seed", range(60 + _EXTRA))
def test_random_3d_tree_matches_oracle(hip, seed):
    rng = random.Random(1000 + seed)
    check(random_3d(rng, rng.choice([2, 3, 4])), hip, specialise=seed % 4 == 0)


@pytest.mark.parametrize("seed", range(36 + _EXTRA))
def test_random_2d_tree_matches_oracle(hip, seed):
    rng = random.Random(2000 + seed)
    check(random_2d(rng, rng.choice([2, 3, 4])), hip, specialise=seed % 4 == 0)
