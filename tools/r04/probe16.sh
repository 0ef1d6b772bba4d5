#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04_probe16; mkdir -p $O
python3 -m pytest tests/test_gpu_bricks.py tests/test_gpu_pruning.py tests/test_gpu_configs.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-hbm-leg > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench c4 rc=$?"
python3 bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline --no-hbm-leg > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench c5 rc=$?"
python3 -c "
import json
for f in ('bench','bench_c4','bench_c5'):
    try:
        d=json.load(open('$O/%s.json'%f)); r=d['roofline']
        print(f, d['value'], d['ms_per_step'], r.get('kernel_ms'), r.get('frac'), d.get('verified',{}).get('ok'), d.get('adaptive',{}).get('leaf_blocks_ms'), d.get('adaptive',{}).get('leaf_blocks_ms_min_max'))
    except Exception as e: print(f, 'failed', e)
"
