#!/usr/bin/env python3
"""Time the specialised dense kernels of sponge(4) at 512^3 under several HU_RTC_FLAGS settings."""
import os
import subprocess
import sys

os.environ.setdefault("CODECAD_AMD_CACHE", "0")   # measurements and soaks choose their evaluator themselves

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import codecad_amd as cc
from codecad_amd import hip_util
n = 512
shape = cc.examples.sponge(int(sys.argv[1]))
t = hip_util.Tape(cc.nodes.make_program(shape)); t.specialize()
shape._codecad_amd_tape_buffer = t
res = []
for pym in (False, True):
    out = None; best = 1e9
    for _ in range(6):
        out = cc.grid_eval.grid_eval(shape, [-0.5 + 0.5 / n] * 3, np.float32(1.0 / n), (n, n, n), pymcubes=pym, out=out)
        best = min(best, out.event.elapsed_ms())
    res.append(best); out.release()
print("float4 %%.3f ms   float %%.3f ms" %% tuple(res))
''' % ROOT

variants = [a for a in sys.argv[1:]] or ["", "-DSDF_FAST_CR_MATH=0", "-DSDF_WAVES_PER_EU=3", "-DSDF_WAVES_PER_EU=4"]
for depth in (4,):
    for v in variants:
        env = dict(os.environ, HU_RTC_FLAGS=v)
        out = subprocess.run([sys.executable, "-c", CHILD, str(depth)], env=env, capture_output=True, text=True)
        print("sponge(%d) %-40s %s" % (depth, v or "(default)", (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1]), flush=True)
