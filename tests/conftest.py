import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
# The on-disk cache of per-tape code objects is off for the tests: a tape uploaded with the default policy must
# be INTERPRETED until a test specialises it, whatever an earlier test (or run) compiled.  The cache's own tests
# point CODECAD_AMD_CACHE at a temporary directory.
os.environ.setdefault("CODECAD_AMD_CACHE", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden_tapes():
    with open(os.path.join(ROOT, "tests", "golden", "ref_tapes.json")) as f:
        data = json.load(f)
    out = {}
    for s in data["shapes"]:
        s = dict(s)
        s["tape"] = np.array(s["tape_u32"], dtype=np.uint32).view(np.float32)
        s["bbox_a"] = [float(v) for v in s["bbox_a"]]
        s["bbox_b"] = [float(v) for v in s["bbox_b"]]
        out[s["name"]] = s
    return out


@pytest.fixture(scope="session")
def golden_tapes():
    return load_golden_tapes()


@pytest.fixture(scope="session")
def golden_block_sizes():
    with open(os.path.join(ROOT, "tests", "golden", "ref_block_sizes.json")) as f:
        return json.load(f)["rows"]


@pytest.fixture(scope="session")
def hip():
    """The opened HIP manager.  On a GPU box a missing library or device is a FAILURE."""
    from codecad_amd import hip_util
    hip_util.manager.lib  # raises loudly if the extension or the device is missing
    return hip_util.manager


def same_bits(a, b):
    """Every float identical BIT FOR BIT (so -0 differs from +0), except that any NaN matches any NaN.
    The sign of a zero is observable downstream (copysign in the slab ops, 1/dot(normal, ray) in the ray
    caster), so the parity tests hold the kernels to it."""
    import numpy as np
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    if a.shape != b.shape:
        return False
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))
