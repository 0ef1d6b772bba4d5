#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_probe4; rm -rf $O; mkdir -p $O
python3 bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 bench.py --config c4 --steps 20 --warmup 3 > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench c4 rc=$?"
python3 bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --no-hbm-leg > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench c5 rc=$?"
python3 -m pytest tests/test_gpu_pruning.py -x -q -m gpu --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -9 $O/pytest.log
python3 tools/prof_planetary.py > $O/planetary.txt 2>&1; grep per-tape $O/planetary.txt
for f in bench bench_c4 bench_c5; do python3 -c "
import json,sys
d=json.load(open('$O/$f.json'))
print('$f', d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'), d['roofline'].get('frac'), d.get('verified',{}).get('ok'), d.get('graph_replay',{}).get('ms_per_step'))
"; done
