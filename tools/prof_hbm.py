#!/usr/bin/env python3
"""The HBM-bound regime and where it ends: dense 512^3 grid_eval (float4: 16 B/voxel, float: 4 B/voxel) over tapes
of growing length, interpreter and per-tape code -> one JSON line per (tape, evaluator, layout) and a summary of the
crossover (the first tape whose float4 launch falls below half of the 8 TB/s peak).  Timed exactly like the
`roofline_hbm` leg of bench.py (it IS bench.hbm_regime, over more tapes): ten back-to-back launches through the C ABI
between two HIP events after forty warm ones -- no Python driver call sits between the launches (round 2 timed
Python-driven calls here and read up to 10 % slow).
Usage (GPU box): python tools/prof_hbm.py [n] > gpurun_out/hbm_sweep.jsonl"""
import json
import os
import sys

os.environ.setdefault("CODECAD_AMD_CACHE", "0")   # measurements choose their evaluator themselves

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import codecad_amd as cc  # noqa: E402
from codecad_amd import hip_util  # noqa: E402
from codecad_amd.hip_util import check  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
hip_util.manager.use_device(0)
dev = torch.device("cuda", 0)
main = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(main)
tapes = [("sphere", cc.shapes.sphere(130)), ("box", cc.shapes.box(100)), ("sphere_plus_box", cc.examples.sphere_plus_box()),
         ("csg_example", cc.examples.csg_example())] + [("sponge%d" % d, cc.examples.sponge(d)) for d in (1, 2, 3, 4, 5)]
rows = bench.hbm_regime(hip_util.manager.lib, check, hip_util, cc, torch, np, dev, main.cuda_stream, n, "specialised", tapes=tapes)
for r in rows:
    print(json.dumps(r), flush=True)
for mode in ("interpreter", "specialised"):
    f4 = [r for r in rows if r["evaluator"] == mode and r["bytes"] == n ** 3 * 16]
    over = [r for r in f4 if r["frac"] < 0.5]
    print(json.dumps({"summary": mode, "best": max(f4, key=lambda r: r["achieved"]),
                      "first_tape_below_half_of_peak": over[0] if over else None}))
