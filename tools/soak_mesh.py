#!/usr/bin/env python3
"""One-off soak of the device marching cubes against the oracle: random block shapes (rows of 1..130 samples,
1..4 blocks) and random fields with exact zeros.  Usage: python tools/soak_mesh.py [cases]"""
import os
import sys

os.environ.setdefault("CODECAD_AMD_CACHE", "0")   # measurements and soaks choose their evaluator themselves

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import test_mesh as tm  # noqa: E402
from codecad_amd.hip_util import manager as hip  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(123)
for case in range(cases):
    n = int(rng.integers(1, 5))
    dims = (int(rng.integers(1, 12)), int(rng.integers(1, 12)), int(rng.choice([1, 2, 3, 15, 16, 17, 31, 32, 33, 34, 63, 64, 65, 100, 130])))
    fields = rng.uniform(-1, 1, (n,) + dims).astype(np.float32)
    fields[rng.uniform(size=fields.shape) < 0.05] = 0.0
    if case % 5 == 0:
        fields = np.round(fields * 2) / 2      # many ties and zeros
    v, t, starts = tm.hip_marching_cubes(hip, fields)
    for b in range(n):
        want_v, want_t = tm.oracle_placed(fields[b])
        assert np.array_equal(v[starts[b, 0]:starts[b + 1, 0]].view(np.uint64), want_v.view(np.uint64)), (case, dims, b)
        assert np.array_equal(t[starts[b, 1]:starts[b + 1, 1]].astype(np.int64) - starts[b, 0], want_t.astype(np.int64)), (case, dims, b)
print("soak ok:", cases, "cases")
