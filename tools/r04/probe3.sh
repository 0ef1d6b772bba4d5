#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_probe3; rm -rf $O; mkdir -p $O
export HU_MAX_PATHS=400
python3 tools/prof_planetary.py > $O/prune.txt 2>&1; echo "prune rc=$?"
HU_PRUNE_RUN=0 python3 tools/prof_planetary.py > $O/prune_norun.txt 2>&1; echo "norun rc=$?"
HU_PRUNE=0 python3 tools/prof_planetary.py > $O/noprune.txt 2>&1; echo "noprune rc=$?"
grep per-tape $O/*.txt
python3 -m pytest tests/test_gpu_configs.py -x -q -m gpu > $O/pytest_configs.log 2>&1; echo "pytest configs rc=$?"; tail -5 $O/pytest_configs.log
python3 -m pytest tests/test_gpu_bricks.py -x -q -m gpu > $O/pytest_bricks.log 2>&1; echo "pytest bricks rc=$?"; tail -5 $O/pytest_bricks.log
