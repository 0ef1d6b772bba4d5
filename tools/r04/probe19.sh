#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_probe19; mkdir -p $O
python3 -m pytest tests/test_gpu_bricks.py tests/test_gpu_pruning.py tests/test_gpu_configs.py tests/test_gpu_drivers.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python3 tools/prof_hbm.py > $O/hbm.jsonl 2>/dev/null; echo "hbm rc=$?"
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 tools/prof_planetary.py > $O/planetary.txt 2>&1; echo "planetary rc=$?"
python3 tools/run_configs.py > $O/configs.txt 2>&1; echo "configs rc=$?"
python3 -c "
import json
for l in open('$O/hbm.jsonl'):
    d=json.loads(l)
    if 'tape' in d and d['evaluator']=='specialised': print(d['tape'], d['kernel'], d['ms'], d['frac'])
d=json.load(open('$O/bench.json')); r=d['roofline']
print('bench', d['value'], d['ms_per_step'], r.get('kernel_ms'), r.get('frac'), d.get('verified',{}).get('ok'), d.get('adaptive',{}).get('leaf_blocks_ms'))
"
grep -v amdgpu $O/planetary.txt; grep "C2" $O/configs.txt | cut -c1-200
