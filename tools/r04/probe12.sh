#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_probe12; rm -rf $O; mkdir -p $O
python3 -m pytest tests -x -q -m gpu --durations=6 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-hbm-leg > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline > $O/c4.json 2> $O/c4.err; echo "c4 rc=$?"
for f in bench c4; do python3 -c "
import json
d=json.load(open('$O/$f.json')); r=d['roofline']
print('$f', d['value'], d['ms_per_step'], r.get('kernel_ms'), r.get('frac'), d['verified']['ok'], r.get('note','')[:60])
"; done
