#!/usr/bin/env python3
"""Run ON THE GPU BOX: bench.py under a matrix of generator / compiler knobs (environment variables read when the
per-tape source is generated), one process each; prints the dense kernel's and the leaf-block kernel's time per setting.
Usage: python tools/exp_spec.py [c3|c5] NAME=VALUE[,NAME=VALUE...] ...   ("-" = the defaults)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
config = "c3"
settings = []
for a in sys.argv[1:]:
    if a in ("c3", "c5"):
        config = a
    else:
        settings.append(a)
for setting in settings or ["-"]:
    env = dict(os.environ, CODECAD_AMD_CACHE="0")
    if setting != "-":
        for kv in setting.split(","):
            k, v = kv.split("=", 1)
            env[k] = v.replace("+", " ")
    steps = "10" if config == "c3" else "3"
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--steps", steps, "--warmup", "2",
                           "--no-cpu-baseline", "--no-hbm-leg", "--no-graph"], capture_output=True, text=True, env=env, cwd=ROOT)
    try:
        d = json.loads(proc.stdout.strip().splitlines()[-1])
        print("%-6s %-44s step %.4f ms  dominant %.4f ms  subdivision %.4f  leaf blocks %.4f ms  value %.0f" %
              (config, setting, d["ms_per_step"], d["roofline"]["kernel_ms"], d["adaptive"]["subdivision_ms"], d["adaptive"]["leaf_blocks_ms"], d["value"]),
              flush=True)
    except Exception as e:
        print(config, setting, "FAILED", e, proc.stderr[-800:], flush=True)
