#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_probe14; mkdir -p $O
for k in 0 32 40 48 53 64; do
  HU_STORE_BOUND_LDS=$k python3 - > $O/lds_$k.txt 2>&1 <<'PY'
import os, sys, json
os.environ.setdefault("CODECAD_AMD_CACHE", "0")
sys.path.insert(0, ".")
import numpy as np, torch
import codecad_amd as cc
from codecad_amd import hip_util
from codecad_amd.hip_util import check
import bench
hip_util.manager.use_device(0)
dev = torch.device("cuda", 0)
main = torch.cuda.Stream(device=dev); torch.cuda.set_stream(main)
tapes = [("sphere", cc.shapes.sphere(130)), ("box", cc.shapes.box(100)), ("sphere_plus_box", cc.examples.sphere_plus_box())]
rows = bench.hbm_regime(hip_util.manager.lib, check, hip_util, cc, torch, np, dev, main.cuda_stream, 512, "specialised", tapes=tapes)
for r in rows:
    if r["evaluator"] == "specialised": print(r["tape"], r["kernel"], r["ms"], r["frac"])
PY
  echo "== HU_STORE_BOUND_LDS=$k"; grep -v amdgpu.ids $O/lds_$k.txt
done
