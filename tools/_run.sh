set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-hbm-leg --no-cpu-baseline > gpurun_out/b_h.json
python - <<PY
import json
d=json.loads(open('gpurun_out/b_h.json').read().strip().splitlines()[-1]); print('hoisted', d['value'], d['ms_per_step'], d['adaptive']['leaf_blocks_ms'], d['roofline']['kernel_ms'])
PY
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -1 gpurun_out/gpu_tests.log
