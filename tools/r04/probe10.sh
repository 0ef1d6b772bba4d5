#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_probe10; rm -rf $O; mkdir -p $O
for mode in "" "--serial-leaf"; do
  tag=$( [ -z "$mode" ] && echo overlap || echo serial )
  python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-hbm-leg $mode > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "bench $tag rc=$?"
  python3 bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --no-hbm-leg $mode > $O/c5_$tag.json 2> $O/c5_$tag.err; echo "c5 $tag rc=$?"
  for f in bench_$tag c5_$tag; do python3 -c "
import json
d=json.load(open('$O/$f.json'))
print('$f', d['value'], 'ms', d['ms_per_step'], 'A', d['roofline'].get('kernel_ms'), d['roofline'].get('frac'), 'C', d['adaptive']['leaf_blocks_ms'], d['adaptive'].get('leaf_blocks_ms_min_max'), 'B', d['adaptive']['subdivision_ms'], 'verified', d['verified']['ok'], 'graph', d['graph_replay'].get('ms_per_step'))
"; done
done
python3 -m pytest tests/test_gpu_bench_contract.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
