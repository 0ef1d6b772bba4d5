#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_probe17; mkdir -p $O
for rep in 1 2 3; do
python3 bench.py --steps 32 --warmup 3 --no-cpu-baseline --no-hbm-leg > $O/base_$rep.json 2> $O/base_$rep.err; echo "base rc=$?"
BENCH_GRAPH_PRIORITIES=1 python3 bench.py --steps 32 --warmup 3 --no-cpu-baseline --no-hbm-leg > $O/prio_$rep.json 2> $O/prio_$rep.err; echo "prio rc=$?"
done
grep -h "graph:" $O/prio_1.err
python3 -c "
import json
for f in ('base_1','prio_1','base_2','prio_2','base_3','prio_3'):
    try:
        d=json.load(open('$O/%s.json'%f)); g=d.get('graph_replay') or {}
        print(f, 'direct', d['ms_per_step'], 'graph', g.get('ms_per_step'), g.get('captured'), g.get('error'))
    except Exception as e: print(f, 'failed', e)
"
