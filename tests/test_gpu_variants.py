"""Every interpreter variant against the oracle: {1, 2} voxels per lane x {distance-only, full}.

The library picks a variant per kernel (csrc/hip_util.hip launch_shape / distance_only); the
environment switches read at first use force one, so each combination runs in its own
process over the kernel-level parity tests."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("voxels_per_lane", ["1", "2"])
@pytest.mark.parametrize("full_interpreter", ["0", "1"])
def test_forced_variant_matches_oracle(hip, voxels_per_lane, full_interpreter):
    env = dict(os.environ, HU_VOXELS_PER_LANE=voxels_per_lane, HU_FULL_INTERPRETER=full_interpreter)
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_gpu_parity.py"),
           os.path.join(ROOT, "tests", "test_gpu_drivers.py"), "-k",
           "grid_eval or subdivision_step or mass_properties_kernel or level_batched or leaf_block or degenerate"]
    proc = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-2000:]
    assert " passed" in proc.stdout
