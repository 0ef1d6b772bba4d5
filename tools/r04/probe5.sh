#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_probe5; rm -rf $O; mkdir -p $O
python3 tools/prof_planetary.py > $O/p1.txt 2>&1; grep per-tape $O/p1.txt
python3 tools/prof_planetary.py > $O/p2.txt 2>&1; grep per-tape $O/p2.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/prof_planetary.py > $O/stats.log 2>&1; echo "stats rc=$?"
CFGS=c4 bash tools/collect_kernels.sh r04b > $O/collect.log 2>&1; echo "collect rc=$?"
