"""bench.py prints ONE JSON line with the fields the driver and the judge read (reduced grid so that it
takes seconds; the numbers themselves are not checked, only that they are there and consistent)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_bench(*extra):
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--n", "128"]
                          + list(extra), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, proc.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("evaluator", ["auto", "interpreter"])
def test_bench_line(hip, evaluator):
    line = run_bench("--evaluator", evaluator)
    for key, kind in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                      ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                      ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(line[key], kind), key
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1
    assert line["vs_baseline"] is None and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["unit"] == "Mvoxels/s" and line["value"] > 0 and "workload" in line["config"]
    assert ("specialised" in line["config"]["evaluator"]) == (evaluator == "auto")
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-3)
    assert r["traffic"] is None     # the committed PMC passes are of the 512^3 grid, not this one
    assert r["achieved"] == pytest.approx(128 ** 3 * 16 / (r["kernel_ms"] * 1e-3) / 1e9, rel=1e-2)
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Mvoxels/s" and c["sample"]
    # whole-job throughput = samples of one step / time of one step
    s = line["samples_per_step_per_gpu"]
    total = s["dense"] + s["subdivision"] + s["leaf_blocks"]
    assert line["value"] == pytest.approx(total / (line["ms_per_step"] * 1e-3) / 1e6, rel=2e-2)
