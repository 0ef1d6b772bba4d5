#!/usr/bin/env python3
"""hipRTC latency of per-tape specialisation (cold, stored into a fresh cache directory) and the latency of taking
the same program from the on-disk cache afterwards."""
import json
import os
import sys
import tempfile
import time

os.environ["CODECAD_AMD_CACHE"] = tempfile.mkdtemp(prefix="codecad_amd_cache_")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import codecad_amd as cc  # noqa: E402
from codecad_amd import hip_util  # noqa: E402
from codecad_amd.shapes import TapeShape  # noqa: E402

shapes = {s["name"]: s for s in json.load(open(os.path.join(ROOT, "tests/golden/ref_tapes.json")))["shapes"]}
for name in ("sponge4", "csg_example", "planetary"):
    tape = np.array(shapes[name]["tape_u32"], dtype=np.uint32).view(np.float32)
    if os.environ.get("PROF_JIT_POOL") == "1":      # the same through the compile servers, one image per kernel, in a cache of its own
        keep = os.environ["CODECAD_AMD_CACHE"]
        os.environ["CODECAD_AMD_CACHE"] = tempfile.mkdtemp(prefix="codecad_amd_cache_pool_")
        os.environ["CODECAD_AMD_SPECIALIZE_POOL"] = "1"
        os.environ["AMD_COMGR_CACHE"] = "0"
        tp = hip_util.Tape(tape, policy="0")
        t0 = time.perf_counter()
        tp.specialize()
        print("%-12s specialize() through the compile servers (%d servers, one image per kernel): %.2f s" % (name, len(hip_util.buffer._background.slots), time.perf_counter() - t0), flush=True)
        del os.environ["CODECAD_AMD_SPECIALIZE_POOL"], os.environ["AMD_COMGR_CACHE"]
        os.environ["CODECAD_AMD_CACHE"] = keep
    os.environ["CODECAD_AMD_SPECIALIZE_POOL"] = "0"      # in this process, as ONE image (the form of rounds 1-3)
    t = hip_util.Tape(tape, policy="0")
    t0 = time.perf_counter()
    t.specialize()
    t1 = time.perf_counter()
    del os.environ["CODECAD_AMD_SPECIALIZE_POOL"]
    out = hip_util.Buffer(cc.grid_eval.FLOAT4, (32, 32, 32))
    c = np.zeros(4, np.float32)
    hip_util.manager.k.grid_eval((32, 32, 32), None, t, c, np.float32(0.1), out).wait()
    t2 = time.perf_counter()
    outf = hip_util.Buffer(np.float32, (32, 32, 32))
    hip_util.manager.k.grid_eval_pymcubes((32, 32, 32), None, t, c, np.float32(0.1), outf).wait()
    t3 = time.perf_counter()
    t4 = time.perf_counter()
    again = hip_util.Tape(tape)          # default policy: from the cache at upload
    t5 = time.perf_counter()
    hip_util.manager.k.grid_eval((32, 32, 32), None, again, c, np.float32(0.1), out).wait()
    t6 = time.perf_counter()
    size = sum(os.path.getsize(os.path.join(os.environ["CODECAD_AMD_CACHE"], f)) for f in os.listdir(os.environ["CODECAD_AMD_CACHE"]))
    print("%-12s %4d instructions: specialize (in this process, all kernels as one image) %.2f s, first float4 launch %.3f s, first float launch %.3f s; "
          "upload + load from the cache %.1f ms (from_cache=%s), first launch %.1f ms; cache now %.2f MB"
          % (name, t.n_instructions, t1 - t0, t2 - t1, t3 - t2, (t5 - t4) * 1e3, again.from_cache, (t6 - t5) * 1e3, size / 1e6), flush=True)

# ---- the default policy as a library user meets it (round 3): a NEW tape in a fresh cache, launched over and over: no
# launch waits for the compiler; how long until the launches run per-tape code, and what each kind of launch costs.
# Round 4: with the compiler-support library's OWN cache off (ROCm 7.2's keeps the objects of sources it has seen: the tapes
# above), so that these are the times of tapes nobody has compiled before; and until ALL of a tape's kernels are loaded.
os.environ["CODECAD_AMD_CACHE"] = tempfile.mkdtemp(prefix="codecad_amd_cache2_")
os.environ["AMD_COMGR_CACHE"] = "0"
n = 256
for name in ("sponge4", "planetary", "sponge4 again (warm servers)"):
    tape = np.array(shapes[name.split()[0]]["tape_u32"], dtype=np.uint32).view(np.float32)
    if "again" in name:
        tape = tape.copy()
        os.environ["CODECAD_AMD_CACHE"] = tempfile.mkdtemp(prefix="codecad_amd_cache3_")
    t_start = time.perf_counter()
    t = hip_util.Tape(tape)                      # policy "auto"
    out = hip_util.Buffer(cc.grid_eval.FLOAT4, (n, n, n))
    c = np.array([-0.5, -0.5, -0.5, 0], np.float32)
    launches, first_ms, switched_at, slowest = 0, None, None, 0.0
    interpreted_ms, fast_ms = [], []
    while time.perf_counter() - t_start < 60:
        a = time.perf_counter()
        ev = hip_util.manager.k.grid_eval((n, n, n), None, t, c, np.float32(1.0 / n), out)
        ev.wait()
        b = time.perf_counter()
        launches += 1
        slowest = max(slowest, b - a)
        if first_ms is None:
            first_ms = (b - t_start) * 1e3
        (fast_ms if t.groups & 1 else interpreted_ms).append(ev.elapsed_ms())      # (bit 0: the float4 grid kernel over boxes)
        if t.groups & 1 and switched_at is None:
            switched_at = b - t_start
        if switched_at is not None and len(fast_ms) >= 5 and (t.groups == hip_util.SPEC_ALL or not t._jobs):
            break
    all_at = time.perf_counter() - t_start if t.groups == hip_util.SPEC_ALL else float("nan")
    print("%-12s default policy, %d^3 float4 grids back to back: first result after %.1f ms; per-tape code from launch %d on, %.2f s after "
          "upload, all %d kernels loaded %.2f s after upload; kernel %.3f ms interpreted -> %.3f ms; the slowest call of the run took %.1f ms "
          "(one that loads a finished build)"
          % (name, n, first_ms, len(interpreted_ms) + 1, switched_at if switched_at is not None else float("nan"), hip_util.SPEC_KERNELS, all_at,
             sorted(interpreted_ms)[len(interpreted_ms) // 2] if interpreted_ms else float("nan"),
             sorted(fast_ms)[len(fast_ms) // 2] if fast_ms else float("nan"), slowest * 1e3), flush=True)
    out.release()
