"""The oracle's deterministic elementary functions against float64 libm (numpy)."""
import numpy as np
import pytest

import oracle

TOL = 4e-7   # the Cephes-style routines are good to ~2e-7 relative; the contract is 1e-5


def test_sincos():
    x = np.linspace(-200.0, 200.0, 400001).astype(np.float32)
    s, c = oracle.det_math("sincos", x)
    assert np.max(np.abs(s - np.sin(x.astype(np.float64)))) < TOL
    assert np.max(np.abs(c - np.cos(x.astype(np.float64)))) < TOL


def test_atan2_all_quadrants_and_axes():
    v = np.concatenate([np.linspace(-5, 5, 801), [0.0, -0.0, 1e-30, -1e-30, 1e30]]).astype(np.float32)
    y, x = np.meshgrid(v, v, indexing="ij")
    got = oracle.det_math("atan2", y.ravel(), x.ravel())
    want = np.arctan2(y.ravel().astype(np.float64), x.ravel().astype(np.float64))
    both_zero = (y.ravel() == 0) & (x.ravel() == 0)
    assert np.max(np.abs(got - want)[~both_zero]) < 2 * TOL
    # our convention at the origin: magnitude 0 for x >= 0, pi for x < 0 never happens (x == 0)
    assert np.all(np.abs(got[both_zero]) == 0)


def test_tan_acos():
    t = np.linspace(-1.55, 1.55, 200001).astype(np.float32)
    rel = np.abs(oracle.det_math("tan", t) - np.tan(t.astype(np.float64))) / np.maximum(1e-3, np.abs(np.tan(t.astype(np.float64))))
    assert rel.max() < 2 * TOL
    z = np.linspace(-1, 1, 200001).astype(np.float32)
    assert np.max(np.abs(oracle.det_math("acos", z) - np.arccos(z.astype(np.float64)))) < 2 * TOL
    assert oracle.det_math("acos", np.array([1.0000001, -1.0000001], np.float32)).tolist() == pytest.approx([0.0, np.pi], abs=1e-6)


def test_fmod_and_remainder():
    rng = np.random.default_rng(0)
    x = (rng.random(200000) * 40 - 20).astype(np.float32)
    y = (rng.random(200000) * 3 + 0.01).astype(np.float32)
    f = oracle.det_math("fmod", x, y)
    want = np.fmod(x.astype(np.float64), y.astype(np.float64))
    # equal, or off by exactly one period where the exact quotient sits on an integer
    d = np.abs(f - want)
    assert np.all((d < 1e-5) | (np.abs(d - y) < 1e-5))
    r = oracle.det_math("remainder", x, y)
    want = np.remainder(x.astype(np.float64) + y / 2.0, y.astype(np.float64)) - y / 2.0
    d = np.abs(r - want)
    assert np.all((d < 1e-5) | (np.abs(d - y) < 1e-5))
    assert np.all(np.abs(r) <= y / 2 * (1 + 1e-6))
    # infinite spacing = "not repeated along this axis" (reference shapes/unsafe.py:29-31)
    inf = np.full(5, np.inf, np.float32)
    v = np.array([-3.5, 0.0, 1e-20, 7.25, -0.0], np.float32)
    assert np.array_equal(oracle.det_math("remainder", v, inf), v)


def test_hypot():
    rng = np.random.default_rng(1)
    a = (rng.standard_normal(100000) * 10).astype(np.float32)
    b = (rng.standard_normal(100000) * 10).astype(np.float32)
    h = oracle.det_math("hypot", a, b)
    want = np.hypot(a.astype(np.float64), b.astype(np.float64))
    assert np.max(np.abs(h - want) / want) < 2e-7
