#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_probe9; rm -rf $O; mkdir -p $O
for m in 0 1; do
  export HU_RTC_FLAGS="-DSDF_PRUNE_LAUNDER=$m"
  python3 bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline > $O/c4_$m.json 2> $O/c4_$m.err; echo "c4 launder=$m rc=$?"
  python3 -c "
import json
d=json.load(open('$O/c4_$m.json')); print('launder $m: c4 step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], d['verified']['ok'])"
  python3 tools/prof_planetary.py 2>&1 | grep per-tape
done
HU_RTC_FLAGS="-DSDF_PRUNE_LAUNDER=1" python3 -m pytest tests/test_gpu_pruning.py tests/test_gpu_configs.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
