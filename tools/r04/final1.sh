#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python3 -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/r04_gputest.log 2>&1; echo "pytest rc=$?"; tail -14 gpurun_out/r04_gputest.log
PART=kernels bash tools/collect_round.sh r04
