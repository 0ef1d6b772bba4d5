#!/bin/bash
# Run ON THE GPU BOX: kernel timeline of bench.py's direct steps and of the same steps replayed from a hipGraph
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_graph_trace; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 32 --warmup 2 --no-cpu-baseline --no-hbm-leg --no-verify > $O/bench.json 2> $O/bench.err; echo "rc=$?"
python3 tools/r04/graph_trace.py $O/trace > $O/analysis.txt 2>&1; cat $O/analysis.txt
