#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_probe7; rm -rf $O; mkdir -p $O
bash tools/r04/graph_trace.sh > $O/graph_trace.txt 2>&1; grep -E "^==|A durations|rc=" $O/graph_trace.txt | cut -c1-400
for rb in 64 256; do HU_RUN_BLOCK=$rb python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-verify > $O/bench_rb$rb.json 2> $O/bench_rb$rb.err; python3 -c "
import json
d=json.load(open('$O/bench_rb$rb.json'))
print('run block $rb', d['ms_per_step'], d['graph_replay'])
for e in d.get('roofline_hbm',[]):
    if e['evaluator']=='specialised' and e['tape'] in ('box','sphere'): print('    ', e['tape'], e['kernel'], e['ms'], e['frac'])
"; done
