#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_probe11; rm -rf $O; mkdir -p $O
run() { # tag env...
  tag=$1; shift
  env "$@" python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-hbm-leg --no-graph > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "bench $tag rc=$?"
  env "$@" python3 bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --no-hbm-leg --no-graph > $O/c5_$tag.json 2> $O/c5_$tag.err; echo "c5 $tag rc=$?"
  for f in bench_$tag c5_$tag; do python3 -c "
import json
d=json.load(open('$O/$f.json'))
print('$f', d['value'], 'ms', d['ms_per_step'], 'A', d['roofline'].get('kernel_ms'), 'C', d['adaptive']['leaf_blocks_ms'], 'B', d['adaptive']['subdivision_ms'], 'verified', d['verified']['ok'])
"; done
}
run pruned X=1
run unpruned HU_PRUNE=0
run prunedall HU_PRUNE_EVAL_MIN=0
python3 -m pytest tests/test_gpu_bricks.py tests/test_gpu_pruning.py tests/test_gpu_random_shapes.py tests/test_gpu_drivers.py tests/test_gpu_configs.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
