#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
bash tools/collect_round.sh r04
bash tools/r04/graph_trace.sh > gpurun_out/r04_graph_trace_final.txt 2>&1; echo "graph trace rc=$?"
